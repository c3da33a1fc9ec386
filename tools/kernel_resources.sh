#!/bin/bash
# VGPR / AGPR / scratch / LDS of every kernel in an object file: the code object's metadata notes
#   tools/kernel_resources.sh slice3d_amd/csrc/decode_f16.o [name filter]
set -e
OBJ=$(readlink -f ${1:-slice3d_amd/csrc/decode_f16.o})
TMP=$(mktemp -d)
cd $TMP
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $OBJ
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for blk in txt.split('  - .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = g('name')
    if flt in name:
        print('%-84s vgpr %s agpr %s scratch %s B lds %s B vspill %s' % (name[:84], g('vgpr_count'), blk.split()[0], g('private_segment_fixed_size'), g('group_segment_fixed_size'), g('vgpr_spill_count')))
" "$2"
rm -rf $TMP
