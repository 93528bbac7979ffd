mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "a8w8 or fp8 or fp16_fp16" 2>&1 | tail -5
for cfg in "a8w8_4096_m256:"; do
  IFS=: read w t <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --kernel-samples 64 ${t:+--tuning $t} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:34], '|', r['kernel'], 'kern_us',r['kernel_us'],'achieved',r['achieved'],r['unit'],'frac',r['frac'], d['config'].get('tuning'))
PY
grep -v amdgpu.ids gpurun_out/bench_sweep.err | tail -8
