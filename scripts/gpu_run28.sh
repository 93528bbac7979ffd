mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
for code in 0 256 512 768 1024 2048 4096 4864 7936; do
  timeout 300 python bench.py --workload a16w4_4096_m256 --steps 10 --warmup 2 --no-cpu-baseline --kernel-samples 32 --tuning 0,1,8,$code >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
timeout 300 python bench.py --workload a16w4_4096_m256 --steps 10 --warmup 2 --no-cpu-baseline --kernel-samples 32 --tuning 0,0,8,0 >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
timeout 300 python bench.py --workload a16w4_8192_m256 --steps 10 --warmup 2 --no-cpu-baseline --kernel-samples 32 --tuning 0,0,8,0 >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:30], '|', r['kernel'], 'kern_us',r['kernel_us'],'frac',r['frac'], d['config'].get('tuning'))
PY
grep -v amdgpu.ids gpurun_out/bench_sweep.err | tail -5
