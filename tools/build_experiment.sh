#!/bin/bash
# build an experiment variant of libslice3d_hip.so from a patch under tools/patches (product sources stay untouched)
#   tools/build_experiment.sh tools/patches/attn_q_ablations.patch build/abl/lib_NOBAR.so -DAQ_ABL_NOBAR
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PATCH=$(realpath "$1"); OUT=$2; shift 2
TMP=$(mktemp -d /tmp/s3d_exp_XXXX)
mkdir -p "$TMP/slice3d_amd" "$TMP/include" && cp -r "$ROOT/slice3d_amd/csrc" "$ROOT/slice3d_amd/csrc_mesh" "$TMP/slice3d_amd/" && cp "$ROOT"/include/*.h "$TMP/include/"
(cd "$TMP" && patch -p1 < "$PATCH")
make -C "$TMP/slice3d_amd/csrc" -j8 clean all EXTRA="$*" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$(dirname "$OUT")" && cp "$TMP/slice3d_amd/csrc/libslice3d_hip.so" "$OUT" && rm -rf "$TMP"
echo "built $OUT"
