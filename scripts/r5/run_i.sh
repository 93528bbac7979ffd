#!/bin/bash
# round 5, batch i: groups of 64 at 17 .. 32 rows — the LDS-staged streaming kernel (round-4 default where the registers-only kernel refuses
# two row tiles) against the 8-wave MFMA kernel (tuning[0] = 3) and the rows kernel
mkdir -p gpurun_out/r5i
GL_SHAPES="1024x4096,5120x5120,8960x1536,4096x4096,11008x4096,14336x4096,4096x14336" GL_GS=64 timeout 900 python scripts/probe_rows5.py 17 24 32 > gpurun_out/r5i/probe_g64_m17_32_w4.log 2>&1
grep "^{" gpurun_out/r5i/probe_g64_m17_32_w4.log | cut -c1-330
GL_SHAPES="1024x4096,4096x4096,6144x4096,8192x8192,11008x4096,4096x11008" GL_GS=64 GL_BITS=2 timeout 900 python scripts/probe_rows5.py 17 24 32 > gpurun_out/r5i/probe_g64_m17_32_w2.log 2>&1
grep "^{" gpurun_out/r5i/probe_g64_m17_32_w2.log | cut -c1-330
