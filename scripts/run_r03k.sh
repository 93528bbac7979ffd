export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03k; mkdir -p $O
timeout 900 python scripts/probe_a8w8_m256.py > $O/probe_a8w8_m256.log 2>&1; grep '^{' $O/probe_a8w8_m256.log | cut -c1-260
timeout 600 python -m pytest tests/test_ref_fullsize_gpu.py -m gpu -q -n 4 -p no:cacheprovider > $O/pytest_ref.log 2>&1; tail -4 $O/pytest_ref.log
timeout 300 python bench.py --workload a16w4_16384_m1 --single --no-cpu-baseline --steps 50 --warmup 5 --kernel-samples 0 > $O/b16384.json 2>/dev/null; cat $O/b16384.json | cut -c1-400
