#!/bin/bash
# round 5, GPU call A: parity of the new rows kernel, its timing against the round-4 choice, the default bench line (nested roofline)
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "rows5" -p no:cacheprovider --timeout 600 > $O/pytest_rows5.log 2>&1; echo "rc=$?" >> $O/pytest_rows5.log; tail -8 $O/pytest_rows5.log
timeout 900 python -m pytest tests/test_structured_exact_gpu.py -q -k "packed_weight_families" -p no:cacheprovider --timeout 600 > $O/pytest_struct.log 2>&1; echo "rc=$?" >> $O/pytest_struct.log; tail -5 $O/pytest_struct.log
timeout 600 python scripts/probe_rows5.py > $O/probe_rows5.log 2>&1; tail -80 $O/probe_rows5.log
timeout 600 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json; tail -3 $O/bench_default.err
