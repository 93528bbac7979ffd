mkdir -p gpurun_out/prof_m256; export TMPDIR=/tmp; O=$PWD/gpurun_out/prof_m256
for w in a16w4_4096_m256 a16w4_8192_m256 a8w8_4096_m256; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o bench -- python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $O/$w.log 2>&1
  head -3 $O/$w/bench_kernel_stats.csv | cut -c1-200
  tail -1 $O/$w.log | cut -c1-400
done
