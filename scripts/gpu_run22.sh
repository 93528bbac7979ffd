mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "split_k or tiled or cfgB or M256 or M1-fp16 or M1-bf16 or odd_shapes or golden" 2>&1 | tail -8
for cfg in "a16w4_4096_m256::" "a16w4_4096_m256:0,0,4,0:" "a16w4_4096_m256:0,4,0,0:" "a16w4_4096_m256:0,2,0,0:" "a16w4_8192_m256::" "a16w4_8192_m256:0,0,4,0:" "a16w4_8192_m256:0,2,0,0:" "a16w4_8192_m256:0,1,0,0:" "a16w4_4096_m1::"; do
  IFS=: read w t extra <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline $extra ${t:+--tuning $t} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:44], '|', r['kernel'], 'value',d['value'],d['unit'],'kern_us',r['kernel_us'],'frac',r['frac'],'gap_us',r['us_per_launch_in_timed_region'], d['config'].get('tuning'))
PY
grep -v amdgpu.ids gpurun_out/bench_sweep.err | tail -5
