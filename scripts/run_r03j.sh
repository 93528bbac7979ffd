export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03j; mkdir -p $O
timeout 900 python scripts/probe_mma3.py cfgA cfgB > $O/probe_mma3.log 2>&1; grep '^{' $O/probe_mma3.log | cut -c1-400; grep -v '^{' $O/probe_mma3.log | tail -5
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 6 -p no:cacheprovider -k "mma or cfg or split or tiled or determinism or odd or llm or config5 or low_bits or modes" > $O/pytest_mma.log 2>&1 ); tail -12 $O/pytest_mma.log
