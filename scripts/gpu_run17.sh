mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 600 python scripts/timeline.py > gpurun_out/timeline_a.log 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 600 python scripts/timeline.py > gpurun_out/timeline_b.log 2>&1
for env in "" "HIP_FORCE_DEV_KERNARG=1"; do
 for cfg in "a16w4_4096_m1:::" "a16w4_4096_m1::--no-graph:" "a16w4_8192_m1:::" "a16w4_4096_m16:::" "a16w4_4096_m256:::"; do
  IFS=: read w t extra mt <<< "$cfg"
  env $env timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline $extra ${t:+--tuning $t} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
 done
done
echo "--- default kernarg"; grep -v amdgpu.ids gpurun_out/timeline_a.log; echo "--- HIP_FORCE_DEV_KERNARG=1"; grep -v amdgpu.ids gpurun_out/timeline_b.log
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:60], '|', r['kernel'], 'value',d['value'],d['unit'],'kern_us',r['kernel_us'],'frac',r['frac'],'gap_us',r['us_per_launch_in_timed_region'])
PY
