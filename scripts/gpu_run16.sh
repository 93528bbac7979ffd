mkdir -p gpurun_out; rm -f gpurun_out/bench_sweep.jsonl
timeout 600 python scripts/timeline.py > gpurun_out/timeline.log 2>&1; echo "rc=$?" >> gpurun_out/timeline.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -k "tiled or cfgA" > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
for cfg in "a16w4_4096_m1:::" "a16w4_8192_m16:::" ; do
  IFS=: read w t mt <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline ${t:+--tuning $t} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
cat gpurun_out/timeline.log | grep -v amdgpu.ids; grep -E "FAILED|passed|failed|rc=" gpurun_out/pytest.log | tail -5
