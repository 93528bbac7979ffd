mkdir -p gpurun_out/prof; export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1
have() { grep -qw "$1" gpurun_out/prof/counters_list.txt; }
pick() { out=""; for c in "$@"; do if have $c; then out="$out $c"; fi; done; echo $out; }
P1=$(pick SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU)
P2=$(pick SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES)
P3=$(pick FETCH_SIZE TCC_HIT_sum)
P4=$(pick TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE GRBM_COUNT)
echo "P1=$P1"; echo "P2=$P2"; echo "P3=$P3"; echo "P4=$P4"
i=0
for W in "a16w4_16384_m1:GEMM_SPLITK" "a16w4_4096_m1:GEMM_SPLITK" "a16w4_4096_m1:"; do
  IFS=: read w mt <<< "$W"
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $PWD/gpurun_out/prof/p$i -o r -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-graph --kernel-samples 8 ${mt:+--matmul-type $mt} > gpurun_out/prof/p$i.log 2>&1
    echo "p$i: $w $mt :: $P" >> gpurun_out/prof/index.txt
  done
done
find gpurun_out/prof -name "*.csv" | head -40; cat gpurun_out/prof/index.txt
