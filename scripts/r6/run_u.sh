#!/bin/bash
# round 6, batch u: reference outputs (the reference's Triton kernels ON the MI355X) for the territory of w8_rows_lds_kernel and the group sizes that are not a power of two
export TMPDIR=/tmp
O=gpurun_out/r6u; mkdir -p $O
timeout 1500 python oracle/run_ref_gpu.py --which ref --only r6 --fixture fullsize_ref_r6.npz --budget-s 1300 --out $O > $O/ref.log 2>&1; echo "ref rc=$?"
timeout 600 python oracle/run_ref_gpu.py --which hip --only r6 --out $O > $O/hip.log 2>&1; echo "hip rc=$?"
grep -h "^{" $O/hip.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('case'), r.get('rel_mean_hip_vs_ref'), r.get('rel_max_hip_vs_ref'), r.get('graph_us'), r.get('error'))"
grep -h "^{" $O/ref.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('REF', r.get('case'), r.get('first_call_s'), r.get('graph_us'), r.get('error'))"
tail -3 $O/ref.log
