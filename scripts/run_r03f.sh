export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03f; mkdir -p $O
timeout 300 python scripts/timeline_decode.py > $O/timeline_decode.log 2>&1; grep '^{' $O/timeline_decode.log
timeout 900 python scripts/probe_gemv3.py a16w4_4096_m1 a16w4_4096_m1_bf16 a16w4_8192_m1 a16w4_16384_m1 a16w4_4096_m2 a16w4_4096_m8 a16w4_11008_m1 > $O/probe_gemv3.log 2>&1; grep '^{' $O/probe_gemv3.log | cut -c1-200
