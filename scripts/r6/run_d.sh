#!/bin/bash
# round 6, batch d: gemv_mfma with the exact planes on the round-5 request structure: parity + timing; g32 rows vs stream above 64 rows
export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py tests/test_ref_fullsize_gpu.py -q -m gpu -k "small_magnitude or mfma or gemv or decode or rows5_kernel_shapes or structured or reference_outputs" -p no:cacheprovider > $O/pytest_sub.log 2>&1; tail -5 $O/pytest_sub.log
timeout 900 python scripts/probe_gemv3.py a16w4_8192_m1 a16w2_16384_m1 a16w2_8192_m1 a16w2_4096_m1 a16w4_11008n_m1 '--tunings=[[0,0,0,0],[0,0,0,512],[0,0,0,1024]]' > $O/probe_gemv_r6.log 2>&1
timeout 600 python scripts/probe_gemv3.py a16w4_4096_m2 a16w4_4096_m4 a16w4_8192_m4 '--tunings=[[0,0,0,0],[0,0,0,512]]' >> $O/probe_gemv_r6.log 2>&1
grep "^{" $O/probe_gemv_r6.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('workload'), r.get('tuning'), r.get('kernel'), r.get('chained_us'), r.get('frac'), '%.2e' % r.get('rel_vs_first', -1) if 'rel_vs_first' in r else r.get('error'))"
GL_GS=32 GL_SHAPES="4096x4096,8192x8192,11008x4096" timeout 600 python scripts/probe_rows5.py 64 96 128 256 1024 > $O/probe_g32_rows_vs_stream.log 2>&1
grep "^{" $O/probe_g32_rows_vs_stream.log | cut -c1-400
