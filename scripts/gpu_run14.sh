mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
for cfg in "a16w4_8192_m256:0,4,4,0:" "a16w4_8192_m256:0,4,4,1:" "a16w4_4096_m256:0,0,4,0:" "a16w4_4096_m256:0,0,4,1:"; do
  IFS=: read w t mt <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline ${t:+--tuning $t} ${mt:+--matmul-type $mt} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
grep -E "FAILED|passed|failed|rc=" gpurun_out/pytest.log | tail -30
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:36], '|', r['kernel'], 'value',d['value'],d['unit'],'kern_us',r['kernel_us'],'achieved',r['achieved'],'frac',r['frac'],'gap_us',r['us_per_launch_in_timed_region'])
PY
