# 16x16x32 vs 32x32x16 f16 MFMA under the power cap: sustained TFLOP/s, clock and socket power of register-resident loops
cd $GRAFT_REPO_ROOT
for cfg in "16 6" "32 6" "16 6 zero" "32 6 zero"; do
  echo "=== shape $cfg"
  build/t_mfma_shapes $cfg > /tmp/ms.txt 2>&1 &
  BP=$!
  sleep 3
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket Graphics Package Power" | tr '\n' ' '; echo
    sleep 0.4
  done
  wait $BP
  tail -3 /tmp/ms.txt
done
