#!/bin/bash
# round 5, batch j: A/B on ONE box — the library with split accumulators in gemv_w4_decode_kernel / the 4-bit fp16 forms of gemv_wn_kernel ("new")
# against the previous build ("old": gemlite_amd/csrc/libgemlite_hip_old.so, built from HEAD~ of gemv_wn.hip), M = 1, single-workload bench
run() { for w in a16w4_16384_m1 a16w4_4096_m1; do timeout 200 python bench.py --workload $w --single --no-cpu-baseline --steps 60 --warmup 5 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); r=d[\"roofline\"]; print(\"$1\", r[\"kernel\"], r[\"kernel_us\"], r[\"frac\"])"; done; }
run new; run new_r3decode "--tuning 0,0,0,4096"; run new_r2 "--tuning 0,0,0,16"
cp gemlite_amd/csrc/libgemlite_hip.so /tmp/new.so; cp gemlite_amd/csrc/libgemlite_hip_old.so gemlite_amd/csrc/libgemlite_hip.so
run old; run old_r3decode "--tuning 0,0,0,4096"; run old_r2 "--tuning 0,0,0,16"
cp /tmp/new.so gemlite_amd/csrc/libgemlite_hip.so
run new_again
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "small_magnitude" 2>&1 | tail -2
