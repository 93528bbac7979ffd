#!/bin/bash
# round 5, GPU call B: does requesting x as full 128-byte lines pay? (timing experiment of gemm_w4_rows_kernel), the failed tests again, bench
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
GL_XLINE=1 GL_SHAPES=4096x4096,8192x8192,4096x11008 timeout 600 python scripts/probe_rows5.py 4 16 32 64 > $O/probe_rows5_xline.log 2>&1; cat $O/probe_rows5_xline.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "rows5" -p no:cacheprovider --timeout 600 > $O/pytest_rows5.log 2>&1; echo "rc=$?" >> $O/pytest_rows5.log; tail -8 $O/pytest_rows5.log
timeout 600 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json; tail -3 $O/bench_default.err
