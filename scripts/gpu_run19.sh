mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "direct or cfgA or all_weight_modes or bit_widths or odd_shapes or low_bits or split_k or graph or golden or config5" 2>&1 | tail -15
timeout 300 python scripts/timeline.py 2>&1 | grep -v amdgpu.ids
for cfg in "a16w4_4096_m1::" "a16w4_8192_m1::" "a16w4_16384_m1::" "a16w4_4096_m16::" "a16w4_4096_m16:2,0,0,0:" "a16w4_4096_m16:4,0,0,0:" "a16w4_4096_m16:4,2,0,0:" "a16w4_4096_m1::--matmul-type GEMM_SPLITK" "a16w4_8192_m1::--matmul-type GEMM_SPLITK" ; do
  IFS=: read w t extra <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline $extra ${t:+--tuning $t} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:60], '|', r['kernel'], 'value',d['value'],d['unit'],'kern_us',r['kernel_us'],'frac',r['frac'],'gap_us',r['us_per_launch_in_timed_region'], d['config'].get('tuning'))
PY
grep -v amdgpu.ids gpurun_out/bench_sweep.err | tail -5
