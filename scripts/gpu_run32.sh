mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "M1- or config5 or split_k or golden or low_bits or bit_widths" 2>&1 | tail -3
for cfg in "a16w4_4096_m1:" "a16w4_8192_m1:" "a16w4_16384_m1:" "a16w2_16384_m1:" "a16w4_8192_m1:3,1,0,2"; do
  IFS=: read w t <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 30 --warmup 3 --no-cpu-baseline --kernel-samples 64 ${t:+--tuning $t} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:30], '|', r['kernel'], 'kern_us',r['kernel_us'],'frac',r['frac'], d['config'].get('tuning'))
PY
grep -v amdgpu.ids gpurun_out/bench_sweep.err | tail -5
