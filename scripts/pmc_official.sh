#!/bin/bash
# Counter passes behind `traffic` and `mfma_util` of the bench line (separate runs, never together with the trace domains gpurun
# refuses).  Per workload two passes of `bench.py --workload W --single --no-graph` (eager launches, 3 steps): A = FETCH_SIZE + 8 SQ
# counters, B = WRITE_SIZE.  Then the calibration microbenchmark under the same two counters.  Output: $O/pmc/<W>_{A,B}/, summary by
# scripts/pmc_official.py -> profiles/pmc_traffic.json, profiles/mfma_util.json.   usage: bash scripts/pmc_official.sh <outdir>
export TMPDIR=/tmp
O=${1:-gpurun_out/pmc}; mkdir -p $O
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
for W in ${GL_PMC_WORKLOADS:-a16w4_4096_m1 a16w4_4096_m256 a16w4_8192_m256 a16w4_8192_m2048 a16w4_8192_m1 a16w4_16384_m1 a8w8_4096_m256 a16w2_16384_m256 fp8_16384_m256}; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE $SQ --output-format csv -d $PWD/$O/${W}_A -o p -- python bench.py --workload $W --single --no-cpu-baseline --steps 3 --warmup 1 --no-graph --kernel-samples 0 > $O/${W}_A.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $PWD/$O/${W}_B -o p -- python bench.py --workload $W --single --no-cpu-baseline --steps 3 --warmup 1 --no-graph --kernel-samples 0 > $O/${W}_B.log 2>&1
  find $O/${W}_A $O/${W}_B -name "*kernel_trace.csv" -delete 2>/dev/null
done
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $PWD/$O/calib_A -o p -- scripts/ubench/fetch_calib > $O/calib_A.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $PWD/$O/calib_B -o p -- scripts/ubench/fetch_calib > $O/calib_B.log 2>&1
find $O/calib_A $O/calib_B -name "*kernel_trace.csv" -delete 2>/dev/null
python scripts/pmc_official.py $O | tee $O/summary.txt
# keep the summaries (*_counters_by_kernel.csv, *.json, summary.txt, logs); the raw per-dispatch CSVs exceed what gpurun copies back
for d in $O/*_A $O/*_B; do [ -d "$d" ] && rm -rf "$d"; done
