mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
for cfg in "a16w4_4096_m256:0,1,8,0" "a16w4_4096_m256:0,1,8,8" "a16w4_4096_m256:0,1,8,16" "a16w4_4096_m256:0,1,8,24" "a16w4_4096_m256:0,1,44,0" "a16w4_4096_m256:0,1,44,8" "a16w4_4096_m256:0,1,44,16" "a16w4_4096_m256:0,1,44,24" "a16w4_8192_m256:0,4,44,0" "a16w4_8192_m256:0,4,44,8" "a16w4_8192_m256:0,4,44,16" "a16w4_8192_m256:0,4,44,24"; do
  IFS=: read w t <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --kernel-samples 64 --tuning $t >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:44], '|', r['kernel'], 'kern_us',r['kernel_us'],'frac',r['frac'], d['config'].get('tuning'))
PY
grep -v amdgpu.ids gpurun_out/bench_sweep.err | tail -5
