// gemm_wn_mma.hip — host-side planning of the 8-wave MFMA tile kernel (kernel: gemm_wn_mma_kernel.inc, compiled per 16-bit type in
// gemm_wn_mma_f16.hip / gemm_wn_mma_bf16.hip).
#include "gemm_wn_mma_kernel.inc"

namespace gl {

bool k_rotation_pays(const gemlite_hip_forward_args& a, int64_t tile_bytes);  // gemm_a8w8.hip


const void* mma_lookup_f16(int kind, int nbits, int mi, int xdt, int xch);
const void* mma_lookup_bf16(int kind, int nbits, int mi, int xdt, int xch);

// When the narrow 64 x 64 tiles (KH = 4, 256-k steps) are the default: 0 = never, 1 = K unsplit, 2 = two K slices.  From sweeps over 20 LLM
// layer shapes x M = 40 .. 256 for 4-bit words under 16-bit activations (profiles/r04/probe_mma_narrow_llm_shapes.log, ..._m40_m64.log) and
// over six other geometries x four layer sizes (probe_narrow_geos.log: the same windows hold, 0.71 .. 0.97 of the 128-column time where
// they fire).  t64 = 64 x 64 tiles of the problem; every column tile re-reads its rows of x from L2, x_bytes in total:
//   * 192 <= t64 <= 256 (one row tile, M <= 64: from 140) and x_bytes <= 160 MB: K UNSPLIT (4096^2 M = 256: 19.8 -> 16.8 us; beyond 160 MB
//     the L2 -> LDS path is the limit: 4096 x 11008 M = 256, 344 MB, 33.3 -> 36.5; below 192 tiles too many CUs idle);
//   * 96 <= t64 <= 128 (one row tile with K <= 8192: from 64): TWO K slices (<= 256 blocks: one round; 4096^2 M = 128: 15.4 -> 13.7);
//   * the 128-row narrow tiles never beat the 128-column choice by more than 1 %: forced variants only.
// LDS of the round-6 epilogue of blocks whose K is not split over blocks: the KH partial tiles [KH][BM][BN + 4] fp32 (the kernel takes that
// path when they fit 144 KiB — DIRECT_EPI in gemm_wn_mma_kernel.inc)
static size_t direct_epi_bytes(int splitk, int kh, int bm, int bn) {
    const size_t b = (size_t)kh * bm * (bn + 4) * 4;
    return (splitk == 1 && kh >= 2 && b <= 144 * 1024) ? b : 0;
}

static int narrow_auto(int64_t M, int64_t N, int64_t K, int es) {
    if (M <= 32 || N % 64 != 0 || K % 256 != 0) return 0;
    const int64_t t64 = (N / 64) * ((M + 63) / 64);
    const int64_t x_bytes = (N / 64) * ((M + 63) / 64 * 64) * K * es;
    // (end of round 6: very few tiles take FOUR slices while those stay one round of CUs — 16-bit activations, 4- and 2-bit words alike:
    //  1536 x 8960 M = 96 / 128 16.4 / 16.5 -> 13.6 / 13.8 us, 2048 x 8192 16.4 / 16.9 -> 13.9 / 14.3, 1024 x 4096 M = 128 / 256 11.9 / 12.2 -> 10.7 / 11.4; with 80 or more
    //  tiles four slices lose (2560 x 9728 M = 128: 19.0 vs 24.2) — profiles/r06/probe_narrow_four_slices_w{4,2}.log)
    if (M > 64 && es == 2 && t64 * 4 <= 256 && K / 256 >= 16) return 4;
    if (t64 >= (M <= 64 ? 140 : 192) && t64 <= 256 && x_bytes <= (160ll << 20)) return 1;
    if (t64 >= (M <= 64 ? 64 : 96) && t64 <= 128 && K / 256 >= 8 && (M > 64 || K <= 8192)) return 2;
    // (late round 6, planner re-validation: a short K has nothing to slice — 4096 x 1024 M = 128, 128 tiles: 9.27 (32 x 128 tiles) -> 6.92 us unsplit)
    if (M > 64 && t64 >= 96 && t64 <= 128 && K / 256 < 8) return 1;
    // ... and narrow layers (N <= 1024: the k / v projections of grouped-query models) with 32 .. 64 tiles: two K slices — 1024 x 4096 M = 96 / 128 / 192 / 256
    // 12.4 / 12.5 / 13.5 / 12.8 -> 11.0 / 11.2 / 11.3 / 11.4 us (profiles/r06/probe_mma_narrow_shapes_more_m.log; wider layers with as few tiles — 1536 x 8960 — lose with it)
    // (2-bit words: 15.5 .. 17.6 -> 11.3 .. 11.7; K = 8192 the other way round, 14.6 vs 15.6: up to K = 4096 — profiles/r06/probe_narrow_layers_two_slices.log)
    if (M > 64 && N / 64 <= 16 && t64 >= 32 && t64 <= 64 && K / 256 >= 8 && K / 256 <= 16) return 2;
    return 0;
}

// 16-bit activations x block-scaled weights (layer formats MXFP16 / MXBF16: A16W8_MXFP, A16W4_MXFP): the same kernel with the
// K-contiguous weight geometry (Geo<MXW8 / MXW4>) — the weights are converted by v_cvt_scalef32_pk_* with their block scale, 4
// VALU per fragment instead of 23.  `p` arrives with x / w / scales / epilogue / M / N / K / strides filled in by the caller.
// tuning[1] = K slices, tuning[2] = tile rows / 32.
// nv (round 4): NVFP4 weights (e4m3 scale per 16 k) under the fp16 EXPANSION of NVFP4 activations — `a` then describes that expansion
// (input MXFP16, x [M, K] fp16 in the workspace); any of fp16 / bf16 / fp32 leaves through the untyped epilogue store.
bool plan_gemm_wn_mma_mx(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp, int mode) {
    // mode 0: block-scaled weights (e8m0 per 32 k), 1: NVFP4 weights, 2: plain K-contiguous 8-bit weights of an A16W8 layer (int8 /
    // fp8 e4m3 / e5m2, no block scale: Geo<KW8I / KW8F / KW8B>; the per-channel scale is the epilogue's, set by the caller)
    const bool nv = mode == 1, k8 = mode == 2;
    const bool f16 = k8 ? a.input_dtype == GEMLITE_DT_FP16 : a.input_dtype == GEMLITE_DT_MXFP16;
    if (!f16 && a.input_dtype != (k8 ? GEMLITE_DT_BF16 : GEMLITE_DT_MXBF16)) return false;
    if (nv && (!f16 || a.W_nbits != 4)) return false;
    if (mode == 0 && a.output_dtype != (f16 ? GEMLITE_DT_FP16 : GEMLITE_DT_BF16)) return false;  // typed epilogue
    if (k8 && (a.elements_per_sample != 1 || !(a.w_dtype == GEMLITE_DT_INT8 || a.w_dtype == GEMLITE_DT_FP8E4 || a.w_dtype == GEMLITE_DT_FP8E5))) return false;
    const int nb = k8 ? (a.w_dtype == GEMLITE_DT_INT8 ? mma::KW8I : (a.w_dtype == GEMLITE_DT_FP8E4 ? mma::KW8F : mma::KW8B))
                      : (nv ? mma::NVW4 : (a.W_nbits == 8 ? mma::MXW8 : mma::MXW4));
    const int sbk = nv ? 16 : 32;  // k per scale byte
    if ((!k8 && a.group_size != sbk) || a.stride_wk != 1 || a.stride_xk != 1 || a.stride_on != 1 || a.N % 64 != 0 || a.K % 128 != 0) return false;  // (128 columns: checked behind the narrow tiles)
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0 || ((uintptr_t)a.w_q % 16) != 0 || a.stride_wn % 16 != 0) return false;
    if (((uintptr_t)a.out % 8) != 0 || (a.stride_om * 2) % 8 != 0) return false;
    if ((int64_t)a.N * a.stride_wn >= (1ll << 31) || ((int64_t)a.M * a.stride_xm + a.K) * 2 >= (1ll << 31)) return false;
    if (!k8 && (int64_t)(a.K / sbk) * a.stride_meta_g + (int64_t)a.N * a.stride_meta_n >= (1ll << 31)) return false;
    if (nv || k8) {
        const int oal = a.output_dtype == GEMLITE_DT_FP32 ? 16 : 8;  // 4 outputs per (untyped) store
        if (((uintptr_t)a.out % oal) != 0 || (a.stride_om * (oal / 4)) % oal != 0) return false;
        if (k8 && p.epi.c_mode != 0 && p.epi.c_mode != 2 && ((uintptr_t)p.epi.scales_w % 16) != 0) return false;  // 4 channel scales per load
    }
    int nauto = (a.tuning[1] == 0 && a.tuning[2] == 0 && !(a.tuning[3] & 16384)) ? narrow_auto(a.M, a.N, a.K, 2) : 0;
    // (first fitted on the plain 8-bit weights, then found to hold for the block-scaled ones as well: scan_mxa16_w{4,8}_*.log — 22 / 16 of 60 cells within 3 % before)
    const bool k8_auto = a.M > 64 && a.tuning[1] == 0 && a.tuning[2] == 0 && !(a.tuning[3] & 16384);
    // plain 8-bit weights (late round 6, profiles/r06/scan_a16w8_*.log): very few narrow tiles take FOUR K slices while those stay one round
    // (1024 x 4096 M = 128: 14.8 -> 12.6 us; 1536 x 8960 / 2048 x 8192 M = 96 .. 128: 26 .. 28 -> 17.6 .. 18.1)
    if (k8_auto && a.N % 64 == 0 && a.K % 256 == 0 && (a.N / 64) * ((a.M + 63) / 64) * 4 <= 256 && a.K / 256 >= 16) nauto = 4;
    // ... and two slices of a very long K lose to seven of the 128 x 128 tiles (4096 x 11008 / x 14336 M = 96 .. 128: 33.8 .. 42.4 -> 30.0 .. 36.9)
    if (k8 && k8_auto && nauto == 2 && a.K > 10240 && a.N % 128 == 0) nauto = 0;
    if (a.tuning[2] == 32 || nauto) {  // narrow tiles (64 x 64, 256-k steps, KH = 4; round 4, late): tuning[2] = 32 forces them, [1] = K slices
        if (a.N % 64 != 0 || a.K % 256 != 0) return false;
        const int units = (int)(a.K / 256), splitk = a.tuning[2] == 32 ? (a.tuning[1] > 0 ? a.tuning[1] : 1) : nauto;
        if (splitk > units) return false;
        const int64_t tiles = (int64_t)(a.N / 64) * ((a.M + 63) / 64);
        if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
        const void* fn = f16 ? mma_lookup_f16(6, nb, 0, 0, 0) : mma_lookup_bf16(6, nb, 0, 0, 0);
        if (!fn) return false;
        p.splitk = splitk;
        p.rows_per_slice = (int)a.K;
        p.stride_wn_b = a.stride_wn;
        p.stride_meta_n = k8 ? 0 : a.stride_meta_n;
        p.stride_meta_g = k8 ? 0 : a.stride_meta_g;
        p.w_mode = 2;
        p.gs_shift = 5;
        p.combine = 0;
        // K rotation between the row tiles of a column tile (bit 30 of flags, gemm_wn_mma_kernel.inc): 8-bit weights, K unsplit, the L2 rule of the A8W8 tiles
        p.flags &= ~(1 << 30);
        if ((k8 || nb == mma::MXW8) && splitk == 1 && k_rotation_pays(a, (int64_t)64 * a.K)) p.flags |= (1 << 30);
        lp.fn = fn;
        lp.name = k8 ? "gemm_a16w8_kernel<64x64>" : (nv ? "gemm_nvfp4_f16_kernel<64x64>" : (nb == mma::MXW8 ? "gemm_a16w8_mxfp_kernel<64x64>" : "gemm_a16w4_mxfp_kernel<64x64>"));
        lp.grid = dim3((unsigned)tiles, splitk, 1);
        lp.block = dim3(512, 1, 1);
        const size_t stages = (size_t)((nb == mma::MXW8 || k8) ? 2 : 3) * 64 * 256 * 2;
        const size_t xch = (size_t)3 * 2 * 2 * 4 * 64 * 16;  // K-part exchange: [kh - 1][cg][mi][e4][lane] float4
        const size_t c_b = (size_t)64 * (64 + 4) * 4 + 16;
        lp.lds_bytes = stages > xch ? stages : xch;
        { const size_t de = direct_epi_bytes(splitk, 4, 64, 64); if (lp.lds_bytes < de) lp.lds_bytes = de; }
        if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
        lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * 64 * 64 * 4 : 0;
        lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
        return true;
    }
    if (a.N % mma::BN != 0) return false;
    auto kstep_of = [](int c) { return c >= 4 ? 128 : 256; };
    const int cap = a.M > 128 ? 8 : (a.M > 64 ? 4 : (a.M > 32 ? 2 : 1));
    // cheap conversion: the tallest tile that still gives >= 128 tiles (else >= 64, else the tallest M fills), then the
    // fewest K slices that give >= 224 blocks while a slice keeps >= 4 steps (the rule of the other non-4-bit widths)
    int mi = cap;
    for (int want = 128; want >= 64; want >>= 1) {
        int found = 0;
        for (int c = cap; c >= 2 && !found; c >>= 1)
            if (a.K % kstep_of(c) == 0 && (a.N / mma::BN) * ((a.M + 32 * c - 1) / (32 * c)) >= want) found = c;
        if (found) { mi = found; break; }
    }
    // Plain 8-bit weights above 64 rows (late round 6; 20 LLM layer shapes x M = 96 .. 512 against every forced form, profiles/r06/scan_a16w8_*.log):
    // the weight conversion is per 128 columns whatever the tile's rows, so the 64-row tiles never pay (5120 x 13824 M = 256: 106.6 us vs 56.8 on 128 rows);
    // these tiles hold one block per CU, so K slices never spill into a second round while half the chip stays busy (8960 x 1536 M = 256: 35.3 -> 19.3,
    // 11008 x 4096 M = 128: 45.7 -> 30.5) and a slice keeps >= 1024 k; 256 rows where the model below says so — per block and 1024 k: 8.2 us on 128 rows,
    // 17 on 256, 6 fixed, ~1 per slice — i.e. where the 128-row tiles leave the chip half idle or need two rounds (5120^2 M = 512: 62.7 -> 45.9).
    const int64_t resident = resident_block_limit();
    auto k8_slices = [&](int64_t tiles, int units) {
        int sk = 1;
        for (int s = 2; s <= units && s <= 32 && tiles * sk < 224; ++s) {
            if (units / s < (k8 ? 8 : 12)) break;  // (block-scaled: >= 1536 k — 8192 x 2048 M = 256 A16W4_MXFP two slices 19.4 us, one 16.1)
            if (tiles * s > resident && tiles * sk >= 128) break;
            sk = s;
        }
        return sk;
    };
    int k8_sk = 0;
    if (k8_auto) {
        double best = 0;
        for (int c = 4; c <= cap; c <<= 1) {
            const int64_t t = (int64_t)(a.N / mma::BN) * ((a.M + 32 * c - 1) / (32 * c));
            const int sk = k8_slices(t, (int)(a.K / 128));
            // per block: fixed us, us per 1024 k — plain 8-bit 6 + 8.2 / 17 (128 / 256 rows); block-scaled 8-bit 5 + 8.5 / 8.4 + 11.3; block-scaled 4-bit (MX, NVFP4) 4.2 + 6.4 / 7.4 + 10.2
            const double fix = k8 ? 6.0 : (nb == mma::MXW8 ? (c == 4 ? 5.0 : 8.4) : (c == 4 ? 4.2 : 7.4));
            const double per_k = k8 ? (c == 4 ? 8.2 : 17.0) : (nb == mma::MXW8 ? (c == 4 ? 8.5 : 11.3) : (c == 4 ? 6.4 : 10.2));
            // (a slice of a 256-row block-scaled tile costs ~5 us, slab and combine: 11008 x 4096 M = 256 A16W4_MXFP, 86 tiles x 2 slices 38.9 us vs 172 tiles of 128 rows 31.3)
            const double per_slice = (!k8 && c == 8) ? 5.0 : 1.0;
            const double est = (double)((t * sk + resident - 1) / resident) * (fix + per_k * (double)a.K / sk / 1024.0) + (sk > 1 ? per_slice * sk : 0);
            if (k8_sk == 0 || est < best) { best = est; mi = c; k8_sk = sk; }
        }
    }
    if (a.tuning[2] == 1 || a.tuning[2] == 2 || a.tuning[2] == 4 || a.tuning[2] == 8) mi = a.tuning[2];
    else if (a.tuning[2] != 0) return false;
    if (a.K % kstep_of(mi) != 0) {
        if (a.tuning[2] != 0) return false;
        mi = mi < 4 ? 4 : mi;
    }
    const int ks = kstep_of(mi), bm = 32 * mi;
    const int units = (int)(a.K / ks);
    const int64_t tiles = (int64_t)(a.N / mma::BN) * ((a.M + bm - 1) / bm);
    int splitk = 0;
    if (a.tuning[1] > 0) splitk = a.tuning[1];
    else if (k8_sk) splitk = k8_sk;
    else {
        for (int sk = 1; sk <= units && sk <= 32; ++sk) {
            if (sk > 1 && units / sk < 4) continue;
            splitk = sk;
            if (tiles * sk >= 224) break;
        }
        if (!splitk) splitk = 1;
    }
    if (splitk > units) return false;
    if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
    if ((uint64_t)splitk * bm * mma::BN * 4 >= (1ull << 31)) return false;
    const void* fn = f16 ? mma_lookup_f16(2, nb, mi, 0, 0) : mma_lookup_bf16(2, nb, mi, 0, 0);
    if (!fn) return false;
    p.splitk = splitk;
    p.rows_per_slice = (int)a.K;  // E = 1: "packed rows" are k
    p.stride_wn_b = a.stride_wn;  // 1-byte elements
    p.stride_meta_n = k8 ? 0 : a.stride_meta_n;
    p.stride_meta_g = k8 ? 0 : a.stride_meta_g;
    p.w_mode = 2;
    p.gs_shift = 5;
    lp.fn = fn;
    static const char* names[2][4] = {
        {"gemm_a16w8_mxfp_kernel<32x128>", "gemm_a16w8_mxfp_kernel<64x128>", "gemm_a16w8_mxfp_kernel<128x128>", "gemm_a16w8_mxfp_kernel<256x128>"},
        {"gemm_a16w4_mxfp_kernel<32x128>", "gemm_a16w4_mxfp_kernel<64x128>", "gemm_a16w4_mxfp_kernel<128x128>", "gemm_a16w4_mxfp_kernel<256x128>"}};
    static const char* nv_names[4] = {"gemm_nvfp4_f16_kernel<32x128>", "gemm_nvfp4_f16_kernel<64x128>", "gemm_nvfp4_f16_kernel<128x128>", "gemm_nvfp4_f16_kernel<256x128>"};
    static const char* k8_names[4] = {"gemm_a16w8_kernel<32x128>", "gemm_a16w8_kernel<64x128>", "gemm_a16w8_kernel<128x128>", "gemm_a16w8_kernel<256x128>"};
    if (k8) lp.name = k8_names[mi == 1 ? 0 : (mi == 2 ? 1 : (mi == 4 ? 2 : 3))];
    else lp.name = nv ? nv_names[mi == 1 ? 0 : (mi == 2 ? 1 : (mi == 4 ? 2 : 3))] : names[nb == mma::MXW8 ? 0 : 1][mi == 1 ? 0 : (mi == 2 ? 1 : (mi == 4 ? 2 : 3))];
    lp.grid = dim3((unsigned)tiles, splitk, 1);
    lp.block = dim3(512, 1, 1);
    const bool w8 = nb == mma::MXW8 || k8;
    const int nst = mi == 8 ? 2 : (mi == 4 ? 3 : (w8 ? 2 : (mi == 2 ? 3 : 4)));  // LDS stages of x (mma_pick_mi)
    const size_t stages = (size_t)nst * bm * ks * 2;
    const size_t xch = (size_t)4 * mi * 4 * 64 * 16;
    const size_t c_b = (size_t)(bm < mma::C_ROWS ? bm : mma::C_ROWS) * mma::C_PITCH * 4 + 16;
    lp.lds_bytes = stages > xch ? stages : xch;
    { const size_t de = direct_epi_bytes(splitk, 2, bm, mma::BN); if (lp.lds_bytes < de) lp.lds_bytes = de; }
    if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * bm * mma::BN * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

bool plan_gemm_wn_mma(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits;
    if (nbits != 4 && nbits != 2 && nbits != 1 && nbits != 8) return false;
    const int e = 32 / nbits;
    if (a.N % 64 != 0) return false;  // (128 for everything but the narrow tiles: checked behind them)
    // Activation type: the 16-bit float of the kernel's Tag, or 8 bits (fp8 e4m3 / int8: A8Wn dynamic, BitNet int8) with a
    // 16-bit output.  Tag = the type of the metadata read in the K loop = the output type (any of fp16 / bf16 / fp32 output
    // and fp32 channel scales go through the untyped epilogue store).
    const bool x16 = a.input_dtype == GEMLITE_DT_FP16 || a.input_dtype == GEMLITE_DT_BF16;
    const int xdt = x16 ? 0 : a.input_dtype;
    if (!x16 && a.input_dtype != GEMLITE_DT_FP8E4 && a.input_dtype != GEMLITE_DT_INT8) return false;
    if (!x16 && a.output_dtype != GEMLITE_DT_FP16 && a.output_dtype != GEMLITE_DT_BF16) return false;
    const int tag_dt = x16 ? a.input_dtype : a.output_dtype;
    const bool loop_s = a.W_group_mode >= 2, post_s = a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (loop_s && a.meta_dtype != tag_dt) return false;
    if (post_s && !loop_s && a.meta_dtype != GEMLITE_DT_FP32 && a.meta_dtype != GEMLITE_DT_FP16 && a.meta_dtype != GEMLITE_DT_BF16) return false;
    if (post_s && ((uintptr_t)a.scales % (a.meta_dtype == GEMLITE_DT_FP32 ? 16 : 8)) != 0) return false;  // vector loads of the channel scales
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != tag_dt) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if (xdt == GEMLITE_DT_INT8 && (a.W_group_mode >= 2 || (has_z && !a.zero_is_scalar))) return false;  // integer codes only
    const int es = x16 ? 2 : 1;
    if ((a.stride_xm * es) % 16 != 0 || ((uintptr_t)a.x % 16) != 0) return false;  // 16-byte LDS-DMA pieces
    const int oal = a.output_dtype == GEMLITE_DT_FP32 ? 16 : 8;  // 4 outputs per store
    if (((uintptr_t)a.out % oal) != 0 || (a.stride_om * (oal / 4)) % oal != 0) return false;
    if (p.group_size % 64 != 0 || p.gs_shift < 0) {  // (gs_shift < 0: group sizes that are not a power of two — multiples of 64 included — on the NGS = 2 form, the one that can divide)
        // Groups of 32 (round 6): 32-row tiles with TWO (scale, zero) pairs per column and 64-k sub-block (template parameter NGS = 2).
        // What reaches this: 1- and 8-bit packed words and fp8 activations x 4- / 2-bit words at M >= 2 (4- / 2-bit words under 16-bit activations
        // have the rows kernel in front) — shapes that ran on the coverage kernel until round 5 (4096^2 M = 64: ~3.8 ms).
        if (p.group_size % 32 != 0 || a.K % 256 != 0 || a.N % mma::BN != 0 || xdt == GEMLITE_DT_INT8) return false;  // (odd multiples of 32 above 32: gs_magic, api.hip)
        if (a.tuning[0] != 0 || a.tuning[2] != 0) return false;
        const int rows = (int)(a.K / e), units = (int)(a.K / 256);
        const int64_t tiles = (int64_t)(a.N / mma::BN) * ((a.M + 31) / 32);
        int splitk = 0;
        if (a.tuning[1] > 0) splitk = a.tuning[1];
        else {
            for (int sk = 1; sk <= units && sk <= 32; ++sk) {
                if (sk > 1 && units / sk < 4) continue;
                splitk = sk;
                if (tiles * sk >= 224) break;
            }
            if (!splitk) splitk = 1;
        }
        if (splitk > units) return false;
        if ((int64_t)rows * a.stride_wk * 4 >= (1ll << 31) || ((int64_t)a.M * a.stride_xm + a.K) * es >= (1ll << 31)) return false;
        if (((int64_t)(a.K / 32) * p.stride_meta_g + a.N) * 2 >= (1ll << 31)) return false;
        if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
        const bool f16 = tag_dt == GEMLITE_DT_FP16;
        const void* fn = f16 ? mma_lookup_f16(8, nbits, 1, xdt, 0) : mma_lookup_bf16(8, nbits, 1, xdt, 0);
        if (!fn) return false;
        p.splitk = splitk;
        p.rows_per_slice = rows;
        p.combine = 0;
        lp.fn = fn;
        static const char* g32_names[4] = {"gemm_w4_mma_kernel<32x128,g32>", "gemm_w2_mma_kernel<32x128,g32>", "gemm_w1_mma_kernel<32x128,g32>", "gemm_w8_mma_kernel<32x128,g32>"};
        lp.name = xdt ? (nbits == 4 ? "gemm_a8w4_mma_kernel<32x128,g32>" : "gemm_a8w2_mma_kernel<32x128,g32>") : g32_names[nbits == 4 ? 0 : (nbits == 2 ? 1 : (nbits == 1 ? 2 : 3))];
        lp.grid = dim3((unsigned)tiles, splitk, 1);
        lp.block = dim3(512, 1, 1);
        const size_t stages = (size_t)2 * 32 * 256 * es;
        const size_t xch = (size_t)4 * 1 * 4 * 64 * 16;
        const size_t c_b = (size_t)32 * mma::C_PITCH * 4 + 16;
        lp.lds_bytes = stages > xch ? stages : xch;
        { const size_t de = direct_epi_bytes(splitk, 2, 32, mma::BN); if (lp.lds_bytes < de) lp.lds_bytes = de; }
        if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
        lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * 32 * mma::BN * 4 : 0;
        lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
        return true;
    }
    // Tile rows (32 MI) and K slices.  A dequantised fragment feeds MI MFMAs, so tall tiles need the least unpack arithmetic
    // per MFMA; short tiles and K slices fill the 256 CUs (one 8-wave block each) — but every K slice costs slab traffic
    // through memory (the XCD L2s are not coherent) plus a serial tail in the last block to arrive, and narrow row tiles
    // re-read x from L2 once per column tile.  The choice is the minimum of a block-time model fitted to 66 shapes x 24
    // (MI, slices) pairs measured with HBM-cold weights (scripts/make_tuning_table.py, profiles/r02/autotune_report.json;
    // mean regret against the measured best 0.9 %, worst 18 %; the rule it replaces: 13 % / 90 %):
    //   us = rounds * (P + steps * S + [slices > 1] * (Q + slices * R)) + 0.096 * slab_MB + 0.048 * x_through_L2_MB
    // with rounds = ceil(blocks / 256) and steps = the K steps of the longest slice.
    const int cap = a.M > 128 ? 8 : (a.M > 64 ? 4 : (a.M > 32 ? 2 : 1));
    auto kstep_of = [](int c) { return c >= 4 ? 128 : 256; };
    auto cost_us = [&](int c, int sk) -> double {
        static const double P[4] = {1.46, 1.21, 1.67, 2.82}, S[4] = {0.945, 1.061, 0.695, 1.247};
        static const double Q[4] = {1.78, 1.20, 1.51, 1.09}, R[4] = {0.0, 0.222, 0.283, 0.516};
        const int i = c == 1 ? 0 : (c == 2 ? 1 : (c == 4 ? 2 : 3));
        const int un = (int)(a.K / kstep_of(c));
        const int steps = (un + sk - 1) / sk;
        const int64_t mpad = (a.M + 32 * c - 1) / (32 * c) * (32 * c);
        const int64_t blocks = (a.N / mma::BN) * (mpad / (32 * c)) * sk;
        const double rounds = (double)((blocks + 255) / 256);
        const double slab_mb = sk > 1 ? (double)sk * mpad * a.N * 8e-6 : 0.0;
        // (x rows past M are never requested — late round 6: the padded rows of a 256-row tile at M = 384 were priced as traffic, and 336 tiles of 128 rows in two
        //  rounds were chosen over 224 of 256 rows in one: 14336 x 4096 69.0 vs 60.5 us, profiles/r06/scan_a16w4_m384_m512.log)
        const int64_t m_req = (a.M + 31) / 32 * 32 < mpad ? (a.M + 31) / 32 * 32 : mpad;
        const double x_mb = (double)(a.N / mma::BN) * m_req * a.K * 2e-6;
        return rounds * (P[i] + steps * S[i] + (sk > 1 ? Q[i] + sk * R[i] : 0.0)) + 0.0955 * slab_mb + 0.0484 * x_mb;
    };
    int mi = 0, splitk = 0;
    // narrow tiles (32 MI x 64, KH = 4; round 4): tuning[2] = 32 + variant forces them (variants: gemm_wn_mma_kernel.inc, mma_pick_narrow);
    // tuning[3] & 16384 keeps the round-3 choice (A/B runs)
    // Automatic: narrow_auto() above (4- / 2-bit words under 16-bit or 8-bit activations)
    int narrow_v = -1, narrow_sk = 0;
    if (a.tuning[2] >= 32 && a.tuning[2] <= 35) {
        narrow_v = a.tuning[2] - 32;
        narrow_sk = a.tuning[1] > 0 ? a.tuning[1] : 1;
    } else if (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 && !(a.tuning[3] & 16384) && (nbits == 4 || nbits == 2)) {
        if (const int na = narrow_auto(a.M, a.N, a.K, es)) {
            narrow_v = 0;
            narrow_sk = na;
        }
    }
    if (narrow_v >= 0) {
        static const int V_MI[4] = {2, 2, 4, 4}, V_KS[4] = {256, 512, 256, 256}, V_NST[4] = {3, 2, 2, 2};
        const int v = narrow_v, vmi = V_MI[v], ks = V_KS[v], bm = 32 * vmi;
        if ((!x16 && v != 0) || (nbits != 4 && nbits != 2) || a.N % 64 != 0 || a.K % ks != 0) return false;  // (8-bit activations: variant 0 only)
        const int rows = (int)(a.K / e), units = (int)(a.K / ks);
        const int splitk = narrow_sk;
        if (splitk > units) return false;
        if ((int64_t)rows * a.stride_wk * 4 >= (1ll << 31) || ((int64_t)a.M * a.stride_xm + a.K) * 2 >= (1ll << 31)) return false;
        if (((int64_t)(a.K / (p.group_size > 0 ? p.group_size : a.K)) * p.stride_meta_g + a.N) * 2 >= (1ll << 31)) return false;
        const int64_t tiles = (int64_t)(a.N / 64) * ((a.M + bm - 1) / bm);
        if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
        const bool f16 = tag_dt == GEMLITE_DT_FP16;
        const int expv = (int)(((unsigned)a.tuning[3] >> 20) & 63u) << 8;  // (development builds: K-loop ablation, see mma_exp_lookup)
        // round 6: the 64 x 64 tiles fetch their packed words through LDS (one DMA request per wave and step instead of four register loads per
        // lane: cfgA M = 256 16.5 -> see profiles/r06/probe_mma_wl.log); tuning[3] & 131072 keeps the round-5 register path (A/B runs)
        const bool wl = v == 0 && x16 && nbits == 4 && expv == 0 && !(a.tuning[3] & 131072);
        const void* fn = f16 ? mma_lookup_f16(6, nbits, wl ? 4 : v, xdt, expv) : mma_lookup_bf16(6, nbits, wl ? 4 : v, xdt, expv);
        if (!fn) return false;
        p.splitk = splitk;
        p.rows_per_slice = rows;
        p.combine = 0;
        lp.fn = fn;
        static const char* nn[2][2] = {{"gemm_w4_mma_kernel<64x64>", "gemm_w4_mma_kernel<128x64>"}, {"gemm_w2_mma_kernel<64x64>", "gemm_w2_mma_kernel<128x64>"}};
        lp.name = xdt ? (nbits == 4 ? "gemm_a8w4_mma_kernel<64x64>" : "gemm_a8w2_mma_kernel<64x64>") : nn[nbits == 4 ? 0 : 1][vmi == 2 ? 0 : 1];
        lp.grid = dim3((unsigned)tiles, splitk, 1);
        lp.block = dim3(512, 1, 1);
        const size_t stages = (size_t)V_NST[v] * (bm * ks * (x16 ? 2 : 1) + (wl ? 8192 : 0));
        const size_t xch = (size_t)3 * 2 * vmi * 4 * 64 * 16;  // K-part exchange: [kh - 1][cg][mi][e4][lane] float4
        const size_t c_b = (size_t)(bm < mma::C_ROWS ? bm : mma::C_ROWS) * (64 + 4) * 4 + 16;
        lp.lds_bytes = stages > xch ? stages : xch;
        { const size_t de = direct_epi_bytes(splitk, 4, bm, 64); if (lp.lds_bytes < de) lp.lds_bytes = de; }
        if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
        lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * bm * 64 * 4 : 0;
        lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
        return true;
    }
    if (a.N % mma::BN != 0) return false;
    // wide tiles (32 MI x 256): tuning[2] = 16 + MI (20 / 24) forces them
    const bool wide_ok = x16 && (nbits == 4 || nbits == 2) && a.N % 256 == 0 && a.K % 64 == 0;
    bool wide = (a.tuning[2] == 20 || a.tuning[2] == 24);
    if (wide) {
        if (!wide_ok) return false;
        mi = a.tuning[2] - 16;
        const int un = (int)(a.K / 64);
        const int64_t tl = (int64_t)(a.N / 256) * ((a.M + 32 * mi - 1) / (32 * mi));
        if (a.tuning[1] > 0) splitk = a.tuning[1];
        else {
            for (int sk = 1; sk <= un && sk <= 16; ++sk) {
                if (sk > 1 && un / sk < 8) continue;
                splitk = sk;
                if (tl * sk >= 224) break;
            }
            if (!splitk) splitk = 1;
        }
    } else if (nbits != 4) {
        // other bit widths (not swept; their unpack arithmetic per weight differs): the rule of thumb the 4-bit model
        // replaced, measured on config 5 (A16W2 16384^2, M = 256: 256-row tiles x 2 slices, 137 us) — the tallest tile
        // (>= 64 rows) that still yields >= 128 tiles, else >= 64, else the tallest M fills; then the fewest K slices that
        // give >= 224 blocks while a slice keeps >= 4 steps.
        mi = cap;
        for (int want = 128; want >= 64; want >>= 1) {
            int found = 0;
            for (int c = cap; c >= 2 && !found; c >>= 1)
                if (a.K % kstep_of(c) == 0 && (a.N / mma::BN) * ((a.M + 32 * c - 1) / (32 * c)) >= want) found = c;
            if (found) { mi = found; break; }
        }
        if (a.tuning[2] == 1 || a.tuning[2] == 2 || a.tuning[2] == 4 || a.tuning[2] == 8) mi = a.tuning[2];
        else if (a.tuning[2] != 0) return false;
        if (a.K % kstep_of(mi) != 0) {
            if (a.K % 128 != 0 || a.tuning[2] != 0) return false;
            mi = mi < 4 ? 4 : mi;  // K = 128 * odd: only the 128-k-step variants apply
        }
        const int un = (int)(a.K / kstep_of(mi));
        const int64_t tl = (int64_t)(a.N / mma::BN) * ((a.M + 32 * mi - 1) / (32 * mi));
        if (a.tuning[1] > 0) splitk = a.tuning[1];
        else {
            for (int sk = 1; sk <= un && sk <= 32; ++sk) {
                if (un / sk < 4) continue;
                splitk = sk;
                if (tl * sk >= 224) break;
            }
            if (!splitk) splitk = 1;
        }
        // 2-bit words under 16-bit activations above 64 rows, late round 6 (20 LLM layer shapes x M = 128 .. 512 against every forced form,
        // profiles/r06/scan_a16w2_*.log: the rule of thumb above within 3 % of the best form on 25 of 60 cells — its slice count spills into a second
        // round of one-block-per-CU tiles (8960 x 1536 M = 256: two slices 30.8 us, one 16.0) and its 64-row tiles lose wherever 128 rows fill the
        // chip (5120 x 13824 M = 256: 77.2 vs 50.4)).  A block-time model instead, rounds x (fixed + us per 1024 k) + 1 us per slice, per block:
        // 64 rows 5 + 5.0, 128 rows 7 + 6.1, 256 rows 9 + 12; a slice keeps >= 1024 k.  60 cells after: 50 within 3 %, 58 within 10 %.
        if (nbits == 2 && x16 && a.M > 64 && a.tuning[1] == 0 && a.tuning[2] == 0 && !(a.tuning[3] & 16384)) {
            static const double FIX[3] = {5.0, 7.0, 9.0}, PER_K[3] = {5.0, 6.1, 12.0};
            const int64_t cus = resident_block_limit();
            double best = 1e30;
            for (int c = 2; c <= cap; c <<= 1) {
                if (a.K % kstep_of(c) != 0) continue;
                const int i = c == 2 ? 0 : (c == 4 ? 1 : 2);
                const int64_t t = (int64_t)(a.N / mma::BN) * ((a.M + 32 * c - 1) / (32 * c));
                for (int sk = 1; sk <= 16 && (sk == 1 || a.K / sk >= 1024); ++sk) {
                    // (+ the weights once per row tile at ~4 TB/s: the row tiles of a column tile run at the same time and each pulls its own copy —
                    //  BASELINE config 5, 16384^2 M = 256: 256 rows x 2 slices 140 us, 128 rows 151)
                    //  — for weights past the 32 MB of the eight L2s; smaller ones are re-read from there: 14336 x 4096 M = 256: 128 rows 39.0 us, 256 rows 44.5)
                    const double w_bytes = (double)a.N * a.K * nbits / 8.0;
                    const double w_us = w_bytes > 32.0 * 1048576.0 ? w_bytes * (double)((a.M + 32 * c - 1) / (32 * c)) / 4e6 : 0.0;
                    const double est = (double)((t * sk + cus - 1) / cus) * (FIX[i] + PER_K[i] * (double)a.K / sk / 1024.0) + (sk > 1 ? sk : 0) + w_us;
                    if (est < best) { best = est; mi = c; splitk = sk; }
                }
            }
        }
    } else {
        double best = 1e30;
        // second pass: K = 128 * odd (896, 640, ...) divides only the 128-k steps of the 128- / 256-row tiles — those then
        // apply whatever M is (rows >= M are never loaded nor stored)
        for (int pass = 0; pass < 2 && !mi; ++pass)
            for (int c = 1; c <= 8; c <<= 1) {
                if (a.tuning[2] != 0 ? c != a.tuning[2] : (pass == 0 ? c > cap : c != 4)) continue;
                if (a.K % kstep_of(c) != 0) continue;
                const int un = (int)(a.K / kstep_of(c));
                for (int sk = 1; sk <= un && sk <= 16; ++sk) {
                    if (a.tuning[1] > 0 ? sk != a.tuning[1] : (sk > 1 && un / sk < 2)) continue;
                    const double t = cost_us(c, sk);
                    if (t < best) { best = t; mi = c; splitk = sk; }
                }
            }
        if (!mi) return false;  // no tile variant divides K (K % 128 != 0), or an override that does not apply
    }
    // automatic choice of the wide tile: when 256 x 256 tiles alone fill the chip (prefill)
    if (!wide && wide_ok && a.tuning[1] == 0 && a.tuning[2] == 0 && (int64_t)(a.N / 256) * ((a.M + 255) / 256) >= 192) {
        wide = true;
        mi = 8;
        splitk = 1;
    }
    const int bn = wide ? 256 : mma::BN;
    const int ks = wide ? 64 : kstep_of(mi);
    const int bm = 32 * mi;
    const int rows = (int)(a.K / e), step_rows = ks / e;
    const int units = rows / step_rows;
    if (splitk > units) return false;
    // buffer descriptors: 32-bit byte offsets
    if ((int64_t)rows * a.stride_wk * 4 >= (1ll << 31) || ((int64_t)a.M * a.stride_xm + a.K) * es >= (1ll << 31)) return false;
    if (((int64_t)(a.K / (p.group_size > 0 ? p.group_size : a.K)) * p.stride_meta_g + a.N) * 2 >= (1ll << 31)) return false;
    const int64_t tiles = (int64_t)(a.N / bn) * ((a.M + bm - 1) / bm);
    if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
    if ((uint64_t)splitk * bm * bn * 4 >= (1ull << 31)) return false;  // slab buffer descriptor range
    const bool f16 = tag_dt == GEMLITE_DT_FP16;
    const int expv = (int)(((unsigned)a.tuning[3] >> 20) & 63u) << 8;  // (development builds: K-loop ablation, see mma_exp_lookup)
    const void* fn = f16 ? mma_lookup_f16(wide ? 1 : 0, nbits, mi, xdt, expv) : mma_lookup_bf16(wide ? 1 : 0, nbits, mi, xdt, expv);
    // K-slice combine: reduce-scatter between the slices of a tile when they are certain to be co-resident (every block of the
    // launch fits on the device at once: <= one block per CU), the slices divide the tile's row blocks, and the variant exists
    // (4- / 2-bit words, 16-bit activations); else slabs + ticket.  From 4 slices on: with 2 slices the ticket protocol is as fast
    // or faster (cfgA 64x128 x 2: 19.6 vs 21.2 us, cfgB 128x128 x 2: 43.9 vs 43.4; with 4 slices of 256-row tiles 27.5 -> 25.1 and
    // 47.9 -> 44.9, profiles/r03/probe_mma3_v3*.log).  tuning[3]: & 128 forces the ticket protocol, & 2048 takes the reduce-scatter
    // with 2 slices too, & 256 = & 2048 + every block hands its rows over after one poll (tests of that path).
    const bool use_xch = !wide && splitk > 1 && (splitk & (splitk - 1)) == 0 && mi >= 2 && splitk <= mi && !(a.tuning[3] & 128) &&
                     (splitk >= 4 || (a.tuning[3] & (2048 | 256))) &&
                     xdt == 0 && (nbits == 4 || nbits == 2) && tiles * splitk <= resident_block_limit();
    if (use_xch) fn = f16 ? mma_lookup_f16(4, nbits, mi, 0, 1) : mma_lookup_bf16(4, nbits, mi, 0, 1);
    // round 6: the 128 x 128 tiles of 4-bit words under 16-bit activations fetch their packed words through LDS as well (slab + ticket combine)
    const bool wl128 = !wide && !use_xch && mi == 4 && x16 && nbits == 4 && expv == 0 && !(a.tuning[3] & 131072);
    if (wl128) fn = f16 ? mma_lookup_f16(7, nbits, mi, 0, 0) : mma_lookup_bf16(7, nbits, mi, 0, 0);
    if (!fn) return false;
    p.splitk = splitk;
    p.rows_per_slice = rows;  // ALL packed rows: the kernel derives each slice's step range itself
    p.combine = use_xch ? 1 : 0;
    lp.fn = fn;
    static const char* names[4][4] = {
        {"gemm_w4_mma_kernel<32x128>", "gemm_w4_mma_kernel<64x128>", "gemm_w4_mma_kernel<128x128>", "gemm_w4_mma_kernel<256x128>"},
        {"gemm_w2_mma_kernel<32x128>", "gemm_w2_mma_kernel<64x128>", "gemm_w2_mma_kernel<128x128>", "gemm_w2_mma_kernel<256x128>"},
        {"gemm_w1_mma_kernel<32x128>", "gemm_w1_mma_kernel<64x128>", "gemm_w1_mma_kernel<128x128>", "gemm_w1_mma_kernel<256x128>"},
        {"gemm_w8_mma_kernel<32x128>", "gemm_w8_mma_kernel<64x128>", "gemm_w8_mma_kernel<128x128>", "gemm_w8_mma_kernel<256x128>"}};
    static const char* names8[2][4] = {
        {"gemm_a8w4_mma_kernel<32x128>", "gemm_a8w4_mma_kernel<64x128>", "gemm_a8w4_mma_kernel<128x128>", "gemm_a8w4_mma_kernel<256x128>"},
        {"gemm_a8w2_mma_kernel<32x128>", "gemm_a8w2_mma_kernel<64x128>", "gemm_a8w2_mma_kernel<128x128>", "gemm_a8w2_mma_kernel<256x128>"}};
    const int mix = mi == 1 ? 0 : (mi == 2 ? 1 : (mi == 4 ? 2 : 3));
    lp.name = xdt ? names8[nbits == 4 ? 0 : 1][mix] : names[nbits == 4 ? 0 : (nbits == 2 ? 1 : (nbits == 1 ? 2 : 3))][mix];
    static const char* names_wide[2][2] = {{"gemm_w4_mma_kernel<128x256>", "gemm_w4_mma_kernel<256x256>"},
                                           {"gemm_w2_mma_kernel<128x256>", "gemm_w2_mma_kernel<256x256>"}};
    if (wide) lp.name = names_wide[nbits == 4 ? 0 : 1][mi == 8 ? 1 : 0];
    lp.grid = dim3((unsigned)tiles, splitk, 1);
    lp.block = dim3(512, 1, 1);
    const int nst = mi == 8 ? 2 : (mi == 4 ? 3 : (nbits == 8 ? 2 : (mi == 2 ? 3 : 4)));  // LDS stages of x (mma_pick_mi)
    const size_t stages = (size_t)nst * (bm * ks * es + (wl128 ? 8192 : 0));
    const size_t xch = (size_t)4 * mi * 4 * 64 * 16;  // K-half exchange: [cg][mi][e4][lane] float4
    const size_t c_b = (size_t)(bm < mma::C_ROWS ? bm : mma::C_ROWS) * (bn + 4) * 4 + 16;
    lp.lds_bytes = stages > xch ? stages : xch;
    { const size_t de = direct_epi_bytes(splitk, wide ? 1 : 2, bm, bn); if (lp.lds_bytes < de) lp.lds_bytes = de; }
    if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
    if (p.combine == 1) {  // output staging tile of the owned rows + the S - 1 inbound copies of them
        const size_t owb = (size_t)mi / splitk;
        const size_t ex = ((owb * 32 * (bn + 4) * 4 + 1023) & ~(size_t)1023) + (size_t)(splitk - 1) * owb * 16 * 1024;
        if (lp.lds_bytes < ex) lp.lds_bytes = ex;
    }
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * bm * bn * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

}  // namespace gl
