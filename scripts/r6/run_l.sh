#!/bin/bash
# round 6, batch l: direct epilogue of the unsplit tiles: A/B is not possible in one library (the path is compile-time) — timing against batch h / k, then the tile-kernel tests
export TMPDIR=/tmp
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python scripts/r6/probe_mma_wl.py > $O/probe_mma_direct_epilogue.log 2>&1; grep "^{" $O/probe_mma_direct_epilogue.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
