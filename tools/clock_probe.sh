# GPU clock / power while the decoder runs (is the dominant kernel clock-limited by power?)
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6
echo "--- under load (bench.py inference only, sampled every 0.25 s) ---"
python bench.py --steps 150 --warmup 5 --infer-only > /tmp/b.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '; echo
  sleep 0.25
done
wait $BP
python -c "import json; r=json.load(open('/tmp/b.json')); print('value', r['value'], 'ffn ms', r['roofline']['avg_launch_ms'])"
