mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "split_k or tiled or cfgB or M256 or odd_shapes" 2>&1 | tail -4
for cfg in "a16w4_4096_m256:0,1,8,0" "a16w4_4096_m256:0,0,8,0" "a16w4_4096_m256:0,4,8,0" "a16w4_4096_m256:0,1,44,0" "a16w4_4096_m256:0,0,44,0" "a16w4_8192_m256:0,0,8,0" "a16w4_8192_m256:0,2,8,0" "a16w4_8192_m256:0,0,44,0"; do
  IFS=: read w t <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --kernel-samples 64 --tuning $t >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:30], '|', r['kernel'], 'kern_us',r['kernel_us'],'frac',r['frac'], d['config'].get('tuning'))
PY
grep -v amdgpu.ids gpurun_out/bench_sweep.err | tail -5
