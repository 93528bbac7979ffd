# round 6, late: one-box A/B of the tile kernel against the build of commit 9e0a5e3 (before the group division / K rotation entered gemm_wn_mma_kernel.inc) + the tests those changes touch
W="a16w4_4096_m256 a16w4_8192_m256 a16w4_4096_m256_fp16 a16w4_8192_m2048 a16w2_16384_m256"
for i in 1 2; do
python scripts/r6/ab_old_new.py _ab_old $W 2>&1 | grep "^{"
python scripts/r6/ab_old_new.py . $W 2>&1 | grep "^{"
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_ref_fullsize_gpu.py -m gpu -q -p no:cacheprovider --timeout 900 -n 3 -k "power_of_two or groups_of_32 or round6 or a16w8_tile or mma" 2>&1 | tail -3
python scripts/r6/probe_a16w8_tiles_rotation.py 2>&1 | grep "^{" | head -3
