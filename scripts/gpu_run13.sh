mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -k "cfgA or golden or modes or bits or determinism or graph" > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
for cfg in "a16w4_4096_m1::" "a16w4_4096_m1:2,1,0,1:" "a16w4_4096_m1:2,1,0,2:" "a16w4_4096_m1:3,1,0,2:" "a16w4_4096_m1:3,2,0,2:" "a16w4_4096_m1:4,4,0,2:" "a16w4_4096_m1_bf16::" "a16w4_8192_m1::" "a16w4_8192_m1:3,1,0,2:" "a16w4_8192_m1:3,1,0,1:" "a16w4_16384_m1::" "a16w4_16384_m1:4,1,0,2:"; do
  IFS=: read w t mt <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline ${t:+--tuning $t} ${mt:+--matmul-type $mt} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
tail -5 gpurun_out/pytest.log
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:36], '|', r['kernel'], 'value',d['value'],d['unit'],'kern_us',r['kernel_us'],'achieved',r['achieved'],'frac',r['frac'],'gap_us',r['us_per_launch_in_timed_region'])
PY
