#!/bin/bash
# round 6, batch t: group sizes that are not a power of two on the tile kernel; the tile-kernel tests around it (the metadata row index changed form)
export TMPDIR=/tmp
O=gpurun_out/r6t; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_structured_exact_gpu.py tests/test_ref_fullsize_gpu.py -m gpu -q -p no:cacheprovider --timeout 900 -n 3 -k "power_of_two or groups_of_32 or mma or tiled or cfgA or cfgB or m256 or narrow or fullsize" > $O/pytest_sub.log 2>&1; tail -12 $O/pytest_sub.log
timeout 300 python bench.py --workload a16w4_4096_m256 --steps 50 --warmup 5 --no-cpu-baseline --single 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfgA', d['roofline']['kernel'], d['roofline']['kernel_us'])"
timeout 300 python bench.py --workload a16w4_8192_m256 --steps 50 --warmup 5 --no-cpu-baseline --single 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfgB', d['roofline']['kernel'], d['roofline']['kernel_us'])"
