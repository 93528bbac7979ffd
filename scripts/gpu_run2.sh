set -x
mkdir -p gpurun_out
rm -f gpurun_out/bench_sweep.jsonl
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
for cfg in "a16w4_4096_m1:" "a16w4_4096_m1:4,0" "a16w4_4096_m1:4,2" "a16w4_4096_m1:4,4" "a16w4_4096_m1_bf16:" "a16w4_4096_m8:" "a16w4_8192_m1:" "a16w4_8192_m1:4,0" "a16w2_16384_m1:" "a16w2_16384_m1:4,0" "a16w4_16384_m1:" "a16w4_4096_m16:" "a16w4_4096_m256:" "a16w4_8192_m256:"; do
  w=${cfg%%:*}; t=${cfg##*:}
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline ${t:+--tuning $t} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_m1.json 2> gpurun_out/bench_m1.err
tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest.log
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:40], '|', r['kernel'], 'value',d['value'],d['unit'],'kern_us',r['kernel_us'],'achieved',r['achieved'],'frac',r['frac'],'gap_us',r['us_per_launch_in_timed_region'])
PY
cat gpurun_out/bench_m1.json
