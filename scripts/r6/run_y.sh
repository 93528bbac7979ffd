#!/bin/bash
# round 6, batch y: tests around the K rotation on the 8-bit narrow tiles of the tile kernel
export TMPDIR=/tmp
O=gpurun_out/r6y; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -n 3 -k "a16w8 or mx or mma or tile or structured or exact or processors or helper" > $O/pytest_sub.log 2>&1; tail -6 $O/pytest_sub.log
python scripts/r6/probe_a16w8_tiles_rotation.py 2>&1 | grep -v "amdgpu.ids\|^Loaded" | head -3
