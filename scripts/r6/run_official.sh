#!/bin/bash
# round 6: the evidence set of the final code — smoke, GPU suite, bench line (+ full form), rocprofv3 kernel stats of the bench command, PMC passes
export TMPDIR=/tmp
export GL_TAG=official GL_COMMIT=e3fdfaf
export GL_PMC_WORKLOADS="a16w4_4096_m1 a16w4_4096_m16 a16w4_4096_m32 a16w4_4096_m64 a16w4_4096_m256 a16w4_8192_m256 a16w4_8192_m2048 a16w4_8192_m1 a16w4_16384_m1 a16w2_16384_m1 a16w2_16384_m256 a8w8_4096_m1 a8w8_4096_m16 a8w8_4096_m32 a8w8_4096_m64 a8w8_4096_m256 fp8_16384_m256 mx_a8w8_4096_m256 mx_a4w4_4096_m256 nvfp4_4096_m256"
bash scripts/gpu.sh smoke tests bench prof pmc
python scripts/make_profiles_summary.py gpurun_out/official 2>/dev/null | tail -5
ls gpurun_out/official gpurun_out/official/pmc | head -60
