export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03a; mkdir -p $O
scripts/ubench/graph_floor > $O/graph_floor.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gf_prof -o gf -- $R/scripts/ubench/graph_floor > $O/gf_prof.log 2>&1)
find $O/gf_prof -name "*kernel_trace.csv" -delete; find $O/gf_prof -name "*agent_info.csv" -delete
timeout 700 python oracle/run_ref_gpu.py --which ref --budget-s 400 > $O/ref.log 2>&1; echo "ref rc=$?" >> $O/ref.log
timeout 400 python oracle/run_ref_gpu.py --which hip --budget-s 300 > $O/hip.log 2>&1; echo "hip rc=$?" >> $O/hip.log
cat $O/graph_floor.log; tail -5 $O/ref.log; tail -3 $O/hip.log
