cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for ob in 0 2; do
rm -rf /tmp/ul; (cd /tmp && S3D_CONV_ONEBUF=$ob rocprofv3 --kernel-trace -d /tmp/ul -o u -- python $GRAFT_REPO_ROOT/tools/unet_layers_run.py 4 > /dev/null 2>&1)
echo "ONEBUF=$ob"; python tools/unet_layers.py $(find /tmp/ul -name "*.db" | head -1) 4 | grep -E "conv1_2|conv2_1|up3 3x3|up4 3x3|total" | cut -c1-130
done
