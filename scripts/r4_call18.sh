#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=20 -p no:cacheprovider --timeout 600 -s -k "cooperative or small_magnitude or a8wn_fp8_activations or remaining_helper or fused_activation" > $O/pytest_sel.log 2>&1; grep "small-x" $O/pytest_sel.log; tail -8 $O/pytest_sel.log | cut -c1-400
timeout 600 python scripts/probe_fused_quant.py > $O/probe_fused_quant.log 2>&1; cat $O/probe_fused_quant.log | cut -c1-300
