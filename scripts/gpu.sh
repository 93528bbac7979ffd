#!/bin/bash
# One parameterised GPU runner (replaces the per-run scripts of round 1).  Usage, from the repo root:
#   gpurun --timeout 900 -- 'bash scripts/gpu.sh <step> [<step> ...]'
# Steps write under gpurun_out/<tag>/ (tag = $GL_TAG, default "run").
#   smoke | tests [pytest -k expr via $GL_K] | bench | bench_all | prof | pmc | ubench | probe:<script.py> | sh:<command>
export TMPDIR=/tmp
TAG=${GL_TAG:-run}; O=gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  case "$step" in
    smoke) python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log ;;
    tests) timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --timeout 600 ${GL_XDIST:+-n $GL_XDIST} ${GL_K:+-k "$GL_K"} > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log ;;
    bench) timeout 900 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json; tail -3 $O/bench_default.err ;;
    bench_all)
      for w in ${GL_WORKLOADS:-a16w4_4096_m256 a16w4_8192_m256 a16w4_8192_m2048 a16w4_4096_m2048 a16w4_16384_m1 a16w4_8192_m1 a16w4_4096_m16 a16w4_4096_m8 a16w2_16384_m1 a16w2_16384_m256 a16w4_4096_m1_bf16 a16w4_11008_m1 a8w8_4096_m1 a8w8_4096_m16 a8w8_4096_m256 fp8_16384_m1 fp8_16384_m256 a8w4_4096_m1 a8w4_4096_m256 a8w4_8192_m256 mx_a8w8_4096_m1 mx_a4w4_4096_m1 mx_a16w4_4096_m1 mx_a8w8_4096_m256 mx_a16w4_4096_m256 mx_a8w8_8192_m256 mx_a8w4_8192_m256 mx_a4w4_8192_m256 mx_a16w4_8192_m256 mx_a16w8_8192_m256 mx_a8w8_8192_m2048 mx_a8w4_8192_m2048 mx_a4w4_8192_m2048}; do
        timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --single >> $O/bench_others.jsonl 2>> $O/bench_others.err
      done; cat $O/bench_others.jsonl ;;
    prof)  # rocprofv3 kernel stats of the default bench command (same code, same process shape)
      timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/rocprof_stats -o bench -- python bench.py --no-cpu-baseline > $O/rocprof_stats.log 2>&1
      find $O/rocprof_stats -name "*kernel_trace.csv" -size +20M -delete
      find $O/rocprof_stats -name "*kernel_stats.csv" | head -3 | while read f; do head -12 "$f"; done ;;
    pmc)   # counter passes behind `traffic` / `mfma_util` of the bench line + FETCH_SIZE / WRITE_SIZE calibration
      bash scripts/pmc_official.sh $O/pmc ;;
    ubench) for b in ${GL_UBENCH:-launch_floor}; do timeout 300 scripts/ubench/$b ${GL_UBENCH_ARGS} > $O/ubench_$b.log 2>&1; cat $O/ubench_$b.log; done ;;
    probe:*) timeout 900 python scripts/${step#probe:} > $O/$(basename ${step#probe:} .py).log 2>&1; tail -60 $O/$(basename ${step#probe:} .py).log ;;
    sh:*) bash -c "${step#sh:}" ;;
  esac
done
