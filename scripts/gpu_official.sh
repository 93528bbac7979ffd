# Round artefacts: tests, smoke, default bench, rocprofv3 kernel stats of the SAME bench command, PMC traffic pass.
mkdir -p gpurun_out/official; export TMPDIR=/tmp; O=gpurun_out/official
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/rocprof_stats -o bench -- python bench.py --no-cpu-baseline > $O/rocprof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $PWD/$O/pmc_fetch -o bench -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 --no-graph --kernel-samples 8 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $PWD/$O/pmc_write -o bench -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 --no-graph --kernel-samples 8 > $O/pmc_write.log 2>&1
for w in a16w4_8192_m256 a16w4_4096_m256 a16w4_16384_m1 a16w4_8192_m1 a16w4_4096_m16 a16w4_4096_m8 a16w2_16384_m1 a16w4_4096_m1_bf16; do
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline >> $O/bench_others.jsonl 2>> $O/bench_others.err
done
tail -3 $O/smoke.log; tail -4 $O/pytest.log; cat $O/bench_default.json; find $O -name "*stats*.csv" | head; 
