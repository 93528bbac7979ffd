export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03h; mkdir -p $O
timeout 300 python scripts/timeline_decode.py > $O/timeline_decode.log 2>&1; grep '^{' $O/timeline_decode.log
timeout 900 python scripts/probe_gemv3.py > $O/probe_gemv3.log 2>&1
