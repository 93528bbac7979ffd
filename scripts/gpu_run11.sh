mkdir -p gpurun_out gpurun_out/prof2; rm -f gpurun_out/bench_sweep.jsonl; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
for cfg in "a16w4_4096_m256::" "a16w4_4096_m256:0,2:" "a16w4_8192_m256::" "a16w4_8192_m256:0,4:" "a16w4_4096_m8::" "a16w4_4096_m16::" "a16w4_4096_m1::GEMM_SPLITK" "a16w4_16384_m1::GEMM_SPLITK"; do
  IFS=: read w t mt <<< "$cfg"
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline ${t:+--tuning $t} ${mt:+--matmul-type $mt} >> gpurun_out/bench_sweep.jsonl 2>> gpurun_out/bench_sweep.err
done
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $PWD/gpurun_out/prof2/p$i -o r -- python bench.py --workload a16w4_8192_m256 --steps 3 --warmup 1 --no-cpu-baseline --no-graph --kernel-samples 8 > gpurun_out/prof2/p$i.log 2>&1
done
tail -5 gpurun_out/pytest.log
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:36], '|', r['kernel'], 'value',d['value'],d['unit'],'kern_us',r['kernel_us'],'achieved',r['achieved'],'frac',r['frac'],'gap_us',r['us_per_launch_in_timed_region'])
PY
