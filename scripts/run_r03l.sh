export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03l; mkdir -p $O
timeout 900 python scripts/probe_gemv3.py > $O/probe_gemv3.log 2>&1; grep '^{' $O/probe_gemv3.log | cut -c1-260
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 6 -p no:cacheprovider -k "low_bits or bit_widths or config5 or modes" > $O/pytest_w2.log 2>&1 ); tail -6 $O/pytest_w2.log
