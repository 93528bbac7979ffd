#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
timeout 900 python scripts/r6/probe_mma_wl.py a16w4_4096_m256 a16w4_8192_m256 a16w4_4096_m128 a16w4_4096_m192 a16w4_11008_m256 a16w4_4096x11008_m256 > $O/probe_mma_wl_ring6.log 2>&1; grep "^{" $O/probe_mma_wl_ring6.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['workload'], r['path'], r.get('us'), r.get('bitwise_equal_first'), r.get('error'))"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py tests/test_ref_fullsize_gpu.py -q -m gpu -k "tiled or mma or m256 or cfgA or cfgB or structured or reference_outputs or narrow or prefill or bitwidth" -p no:cacheprovider > $O/pytest_sub.log 2>&1; tail -4 $O/pytest_sub.log
