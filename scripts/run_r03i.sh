export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03i; mkdir -p $O
timeout 600 python scripts/timeline_mma3.py > $O/timeline_mma3.log 2>&1; grep '^{' $O/timeline_mma3.log; tail -3 $O/timeline_mma3.log | grep -v '^{'
