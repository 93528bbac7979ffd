export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03d; mkdir -p $O
timeout 900 python scripts/probe_gemv3.py > $O/probe_gemv3.log 2>&1; grep '^{' $O/probe_gemv3.log | cut -c1-300
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 6 -p no:cacheprovider -k "cfgA or golden or odd or llm or modes or graph or linearity or bias or determinism or bit_widths" > $O/pytest_m1.log 2>&1 ); tail -15 $O/pytest_m1.log
