#!/bin/bash
# round 6, batch p: re-validation of the planner's closed-form rules — default against forced candidates over the LLM shapes of the planner fixture
export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
timeout 900 python scripts/probe_m1_shapes.py 1 4 > $O/probe_m1_shapes_w4_m1.log 2>&1
timeout 900 python scripts/probe_m1_shapes.py 1 2 > $O/probe_m1_shapes_w2_m1.log 2>&1
timeout 900 python scripts/probe_m1_shapes.py 2 4 > $O/probe_m1_shapes_w4_m2.log 2>&1
timeout 900 python scripts/probe_m1_shapes.py 4 4 > $O/probe_m1_shapes_w4_m4.log 2>&1
timeout 1200 python scripts/probe_rows5.py 8 16 24 32 48 64 > $O/probe_rows5_g128.log 2>&1
GL_GS=64 timeout 1200 python scripts/probe_rows5.py 8 16 32 64 > $O/probe_rows5_g64.log 2>&1
timeout 1500 python scripts/probe_mma_narrow_shapes.py 256 128 > $O/probe_mma_narrow_shapes.log 2>&1
python scripts/r6/planner_regret.py $O > $O/planner_revalidation.json; python -c "
import json; d = json.load(open('$O/planner_revalidation.json'))['summary']; w = d.pop('worst'); print(d); [print(c) for c in w]"
