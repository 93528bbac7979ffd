#!/bin/bash
# round 4, GPU call 1: converter-scale semantics, vendor dense reference, decode3 A/B + timeline, M = 1 parity tests
export TMPDIR=/tmp
O=gpurun_out/r4c1; mkdir -p $O
scripts/ubench/probe_cvt_scale > $O/probe_cvt_scale.log 2>&1; tail -12 $O/probe_cvt_scale.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py tests/test_ref_fullsize_gpu.py -m gpu -q -x -p no:cacheprovider -k "cfgA or all_weight_modes or golden_reference or odd_k or structured or fullsize or hip_graph or linearity" > $O/pytest_m1.log 2>&1; tail -5 $O/pytest_m1.log
timeout 300 python scripts/probe_gemv3.py a16w4_4096_m1 a16w4_4096_m1_bf16 a16w4_11008_m1 '--tunings=[[0,0,0,4096],[0,0,0,0],[0,0,0,4096],[0,0,0,0],[0,0,0,4096],[0,0,0,0]]' > $O/probe_decode3_ab.log 2>&1; grep -v amdgpu.ids $O/probe_decode3_ab.log | cut -c1-330
timeout 200 python scripts/timeline_decode.py a16w4_4096_m1 > $O/timeline_decode3.log 2>&1; grep -v amdgpu.ids $O/timeline_decode3.log
timeout 300 python scripts/probe_vendor_dense.py > $O/probe_vendor_dense.log 2>&1; cat $O/probe_vendor_dense.log
