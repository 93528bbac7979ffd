// api.hip — the C ABI of libgemlite_hip.so (include/gemlite_hip.h): validation, kernel selection,
// workspace carving and launch.  Host code only; the kernels live in gemv_wn.hip, gemm_wn_stream.hip,
// gemm_wn_tiled.hip and generic.hip.
//
// Selection mirrors get_matmul_type() (gemlite/core.py:100-114) in spirit: the caller's matmul_type names
// a kernel FAMILY; inside a family the library picks the CDNA4 kernel that covers the configuration, and
// everything else goes to the generic coverage kernel — never to a CPU path.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "gl_common.h"
#include "gl_coopquant.h"

namespace gl {
// planners (defined next to their kernels)
bool plan_gemv_wn(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp);
bool plan_gemv_mfma(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp);
bool plan_gemv_a8wn(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp, bool fq = false);
bool plan_a8wn_rows(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp);
bool plan_gemm_wn_stream(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp);
bool plan_gemm_wn_direct(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp);
bool plan_gemm_wn_rows(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp);
bool plan_gemm_a8w8(const gemlite_hip_forward_args& a, LaunchPlan& lp);
bool plan_gemm_a8w8_mma(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp);
bool plan_a8w8_rows(const gemlite_hip_forward_args& a, LaunchPlan& lp, bool fq = false);
bool plan_a16w8_rows(const gemlite_hip_forward_args& a, LaunchPlan& lp);
bool a16w8_rows_lds_pays(const gemlite_hip_forward_args& a);
bool plan_gemm_a8w8_sq(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp);
bool plan_gemm_a8w8_sq128(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp);
bool plan_gemm_wn_tiled(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp);
bool plan_gemm_wn_mma(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp);
bool plan_gemm_mx_mma(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp);
bool plan_gemm_mx_sq(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp);
bool plan_gemm_wn_mma_mx(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp, int mode = 0);
bool plan_mx_gemv(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp);
bool plan_gemm_mx_tile(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp);
const void* mx_generic_kernel_fn();
const void* act_quant_mx_kernel_fn(int mode);
bool plan_mx_rows(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp, bool any_m = false, bool fq = false);
bool plan_nvfp4_rows(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp, bool fq = false);
const void* nvfp4_expand_f16_kernel_fn();
const void* generic_kernel_fn();
const void* kmajor_kernel_fn(int mb);
const void* kmajor_w8a16_kernel_fn(int mb);
const void* kmajor_fused_quant_kernel_fn(int qdt);
const void* a8w8_decode_kernel_fn(int qdt, bool fused);
const void* a16w8_decode_kernel_fn(int x_dt, int w_dt);
const void* act_quant_kernel_fn();
const void* act_quant_vec_kernel_fn(int in_dt, int out_dt, int64_t K, int64_t stride_xm, const void* x, const void* y);
const void* pack_kernel_fn();
const void* pack32_kernel_fn();
const void* unpack_kernel_fn();

}  // namespace gl

using namespace gl;

static thread_local int tl_last_hip_error = 0;
static thread_local void* tl_evt_start = nullptr;
static thread_local void* tl_evt_stop = nullptr;

static int dtype_size(int dt) {
    switch (dt) {
        case GEMLITE_DT_FP32: case GEMLITE_DT_INT32: case GEMLITE_DT_UINT32: return 4;
        case GEMLITE_DT_FP16: case GEMLITE_DT_BF16: case GEMLITE_DT_INT16: case GEMLITE_DT_UINT16: return 2;
        case GEMLITE_DT_INT64: return 8;
        case GEMLITE_DT_FP8E4: case GEMLITE_DT_FP8E5: case GEMLITE_DT_INT8: case GEMLITE_DT_UINT8: return 1;
        default: return 0;
    }
}

enum Kind { K_NONE = 0, K_GEMV_WN, K_STREAM_WN, K_TILED_WN, K_KMAJOR, K_A8_MMA, K_A8_FQ, K_NV_MMA, K_GENERIC };

struct Resolved {
    Kind kind = K_NONE;
    LaunchPlan lp{};
    WnParams wn{};
    GenericParams gp{};
    int status = GEMLITE_OK;
    uint64_t nv_x16_bytes = 0;  // K_NV_MMA: bytes of the fp16 expansion of x in the workspace (behind the split-K slabs)
};

static Epilogue make_epilogue(const gemlite_hip_forward_args& a) {
    Epilogue e{};
    e.out = a.out;
    e.scales_w = a.scales;
    e.scales_x = (const float*)a.scales_x;
    e.stride_om = a.stride_om;
    e.stride_on = a.stride_on;
    e.stride_sx_m = a.stride_sx_m;
    e.out_dt = a.output_dtype;
    e.meta_dt = a.meta_dtype;
    e.c_mode = a.channel_scale_mode;
    return e;
}

// Dynamic activation quantisation fused into the matmul: the caller passes the UNQUANTISED 16-bit x, no scales_x, and the 8-bit
// unpacked weights of a dynamically quantised layer (channel_scale_mode 2 / 3).  M = 1: every block quantises the row itself
// (kmajor_fused_quant_kernel); 2 <= M: the blocks of the A8W8 kernels deal the rows among themselves (gl_coopquant.h).
static bool is_mx_input(int dt);
static bool wants_fused_quant(const gemlite_hip_forward_args* a) {
    if (is_mx_input(a->type_id / 100)) return false;  // a block-scaled layer (its weights carry e8m0 block scales): not this kernel family
    return a->M >= 1 && a->elements_per_sample == 1 && !a->scales_x &&
           (a->channel_scale_mode == 2 || a->channel_scale_mode == 3) && a->W_group_mode == 0 &&
           (a->input_dtype == GEMLITE_DT_FP16 || a->input_dtype == GEMLITE_DT_BF16) &&
           (a->w_dtype == GEMLITE_DT_INT8 || a->w_dtype == GEMLITE_DT_FP8E4 || a->w_dtype == GEMLITE_DT_FP8E5);
}

// The same request for PACKED weights (A8Wn fp8 dynamic, BitNet int8 dynamic; round 4, M = 1): the layer's activation type rides in
// type_id (= input_dtype * 100 + W_nbits, core.py:141-145), x is the unquantised 16-bit row.
static int fused_quant_packed_dtype(const gemlite_hip_forward_args* a) {
    if (a->M != 1 || a->elements_per_sample <= 1 || a->scales_x) return 0;
    if (!(a->channel_scale_mode == 2 || a->channel_scale_mode == 3)) return 0;
    if (!(a->input_dtype == GEMLITE_DT_FP16 || a->input_dtype == GEMLITE_DT_BF16)) return 0;
    const int layer_dt = a->type_id / 100;
    return (layer_dt == GEMLITE_DT_FP8E4 || layer_dt == GEMLITE_DT_INT8) && a->type_id % 100 == a->W_nbits ? layer_dt : 0;
}

static bool is_mx_input(int dt);
// ... and for the block-scaled dynamic layers (round 4, M = 1, MXFP8 / MXFP4 activations with block scales): x is the unquantised 16-bit
// row, input_dtype names ITS type, the layer's format rides in type_id like above, scales_x is NULL, channel_scale_mode 4 (block
// scales) or 2 (MXFP8 with one scale per token)
static int fused_quant_mx_dtype(const gemlite_hip_forward_args* a) {
    if (a->M != 1 || a->scales_x || !(a->channel_scale_mode == 4 || a->channel_scale_mode == 2)) return 0;
    if (!(a->input_dtype == GEMLITE_DT_FP16 || a->input_dtype == GEMLITE_DT_BF16)) return 0;
    const int layer_dt = a->type_id / 100;
    if (a->channel_scale_mode == 2 && layer_dt != GEMLITE_DT_MXFP8) return 0;  // one fp32 scale per token: fp8 activations
    return (layer_dt == GEMLITE_DT_MXFP8 || layer_dt == GEMLITE_DT_MXFP4 || layer_dt == GEMLITE_DT_NVFP4) && a->type_id % 100 == a->W_nbits ? layer_dt : 0;
}

static bool is_mx_input(int dt) { return dt >= GEMLITE_DT_MXFP16 && dt <= GEMLITE_DT_NVFP4; }

// block-scaled formats (see the header): what must hold before any kernel is chosen
static int validate_mx(const gemlite_hip_forward_args* a) {
    const bool nv = a->input_dtype == GEMLITE_DT_NVFP4;
    if (a->group_size != (nv ? 16 : 32)) return GEMLITE_ERR_UNSUPPORTED;
    if (a->K % 32 != 0) return GEMLITE_ERR_BAD_SHAPE;
    if (!a->scales) return GEMLITE_ERR_BAD_ARGUMENT;
    if (!(a->output_dtype == GEMLITE_DT_FP32 || a->output_dtype == GEMLITE_DT_FP16 || a->output_dtype == GEMLITE_DT_BF16))
        return GEMLITE_ERR_UNSUPPORTED;
    const bool w8 = a->W_nbits == 8 && a->elements_per_sample == 1 && a->w_dtype == GEMLITE_DT_FP8E4;
    const bool w4 = a->W_nbits == 4 && a->elements_per_sample == 2 && a->w_pack_bits == 8;
    if (!w8 && !w4) return GEMLITE_ERR_UNSUPPORTED;
    const bool x16 = a->input_dtype == GEMLITE_DT_MXFP16 || a->input_dtype == GEMLITE_DT_MXBF16;
    if ((a->input_dtype == GEMLITE_DT_MXFP4 || nv) && !w4) return GEMLITE_ERR_UNSUPPORTED;
    if (x16 ? a->channel_scale_mode != 0 : !(a->channel_scale_mode == 4 || (a->channel_scale_mode == 2 && !nv))) return GEMLITE_ERR_UNSUPPORTED;
    if (!x16 && !a->scales_x) return GEMLITE_ERR_BAD_ARGUMENT;
    if (a->stride_xk != 1) return GEMLITE_ERR_UNSUPPORTED;
    if (a->M > 65535 && a->M > 0x7FFFFFFF) return GEMLITE_ERR_BAD_SHAPE;
    return GEMLITE_OK;
}

static int validate(const gemlite_hip_forward_args* a) {
    if (!a || a->struct_size != sizeof(gemlite_hip_forward_args)) return GEMLITE_ERR_BAD_ARGUMENT;
    if (!a->x || !a->w_q || !a->out) return GEMLITE_ERR_BAD_ARGUMENT;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return GEMLITE_ERR_BAD_ARGUMENT;
    if (a->M > 0x7FFFFFFF || a->N > 0x7FFFFFFF || a->K > 0x7FFFFFFF) return GEMLITE_ERR_BAD_SHAPE;
    if (is_mx_input(a->input_dtype)) return validate_mx(a);
    if (const int mxdt = fused_quant_mx_dtype(a)) {  // validated as the block-scaled call it stands for
        gemlite_hip_forward_args b = *a;
        b.input_dtype = mxdt;
        b.scales_x = (const void*)(uintptr_t)0x1000;
        return validate_mx(&b);
    }
    if (a->W_group_mode < 0 || a->W_group_mode > 4) return GEMLITE_ERR_UNSUPPORTED;
    if (a->channel_scale_mode < 0 || a->channel_scale_mode > 3) return GEMLITE_ERR_UNSUPPORTED;  // 4: block-scaled inputs only
    if (a->elements_per_sample < 1) return GEMLITE_ERR_BAD_ARGUMENT;
    if (a->K % a->elements_per_sample != 0) return GEMLITE_ERR_BAD_SHAPE;
    if (a->input_dtype == GEMLITE_DT_FP8E4NUZ || a->input_dtype == GEMLITE_DT_FP8E5NUZ) return GEMLITE_ERR_UNSUPPORTED;
    if (a->input_dtype > GEMLITE_DT_FP8E5NUZ) return GEMLITE_ERR_UNSUPPORTED;
    if (dtype_size(a->input_dtype) == 0) return GEMLITE_ERR_UNSUPPORTED;
    if (!(a->output_dtype == GEMLITE_DT_FP32 || a->output_dtype == GEMLITE_DT_FP16 || a->output_dtype == GEMLITE_DT_BF16))
        return GEMLITE_ERR_UNSUPPORTED;
    const bool need_s = a->W_group_mode >= 2 || a->channel_scale_mode == 1 || a->channel_scale_mode == 3;
    const bool need_z = a->W_group_mode == 1 || a->W_group_mode >= 3;
    if (need_s && !a->scales) return GEMLITE_ERR_BAD_ARGUMENT;
    if (need_z && !a->zeros) return GEMLITE_ERR_BAD_ARGUMENT;
    if ((a->channel_scale_mode == 2 || a->channel_scale_mode == 3) && !a->scales_x && !wants_fused_quant(a) && !fused_quant_packed_dtype(a)) return GEMLITE_ERR_BAD_ARGUMENT;
    if (a->elements_per_sample > 1) {
        if (a->w_pack_bits != 8 && a->w_pack_bits != 16 && a->w_pack_bits != 32 && a->w_pack_bits != 64)
            return GEMLITE_ERR_BAD_ARGUMENT;
        if (a->W_nbits * a->elements_per_sample != a->w_pack_bits) return GEMLITE_ERR_BAD_ARGUMENT;
    } else if (dtype_size(a->w_dtype) == 0) {
        return GEMLITE_ERR_UNSUPPORTED;
    }
    const bool grouped = a->W_group_mode >= 2 || (need_z && !a->zero_is_scalar);
    if (grouped && a->group_size <= 0) return GEMLITE_ERR_BAD_SHAPE;
    if (a->input_dtype == GEMLITE_DT_INT8 && a->W_group_mode >= 2) return GEMLITE_ERR_UNSUPPORTED;
    return GEMLITE_OK;
}

// Block-scaled formats: the scaled-MFMA kernel for 8 / 4-bit activations, the coverage kernel for everything else
// (16-bit activations x MX weights, NVFP4, layouts that are not K-contiguous).  tuning[0] = 1 forces the coverage kernel.
static GenericParams mx_params(const gemlite_hip_forward_args& a);
static void resolve_mx_plan(const gemlite_hip_forward_args& a, Resolved& r, const GenericParams& g);
static void resolve_mx(const gemlite_hip_forward_args& a, Resolved& r) {
    GenericParams g = mx_params(a);
    r.gp = g;
    resolve_mx_plan(a, r, g);
}
static GenericParams mx_params(const gemlite_hip_forward_args& a) {
    GenericParams g{};
    g.x = a.x; g.w = a.w_q; g.scales = a.scales; g.zeros = nullptr;
    g.epi = make_epilogue(a);
    if (a.channel_scale_mode == 4) g.epi.c_mode = 0;  // block scales are applied inside the contraction
    g.M = (int)a.M; g.N = (int)a.N; g.K = (int)a.K;
    g.nbits = a.W_nbits; g.e = a.elements_per_sample; g.pack_bits = a.w_pack_bits;
    g.w_dt = a.w_dtype; g.x_dt = a.input_dtype;
    g.group_size = a.group_size;
    g.stride_xm = a.stride_xm; g.stride_xk = a.stride_xk; g.stride_wk = a.stride_wk; g.stride_wn = a.stride_wn;
    g.stride_meta_g = a.stride_meta_g; g.stride_meta_n = a.stride_meta_n;
    g.mx_x = a.input_dtype == GEMLITE_DT_MXFP16 ? MX_F16 : (a.input_dtype == GEMLITE_DT_MXBF16 ? MX_BF16 : (a.input_dtype == GEMLITE_DT_MXFP8 ? MX_FP8 : MX_FP4));
    g.mx_w = a.W_nbits == 8 ? MX_FP8 : MX_FP4;
    g.mx_scale_e4m3 = a.input_dtype == GEMLITE_DT_NVFP4 ? 1 : 0;
    g.sx_blocks = a.channel_scale_mode == 4 ? a.scales_x : nullptr;
    g.stride_sx_blk_m = a.stride_sx_m;
    g.mx_post = a.input_dtype == GEMLITE_DT_NVFP4 ? 0.0025f : 1.0f;  // meta_scale_norm = 0.05 ** 2 (gemm_kernels.py:461, 530-531)
    g.splitk = 1;
    return g;
}
static void resolve_mx_plan(const gemlite_hip_forward_args& a, Resolved& r, const GenericParams& g) {
    // 1 .. 64 rows of fp8 / fp4 activations (round 4): 16-column blocks, one 16-row scaled MFMA per 128-k chunk straight from memory.
    // Faster than the streaming kernel below from ONE row (4096^2 fp4 x fp4: 5.4 vs 6.3 us at M = 1, 5.6 vs 9.9 at M = 4) and than the
    // 32-row tile up to 64 rows (M = 16: 5.8 vs 17.3 us) — profiles/r04/probe_mx_rows.log.  tuning[0] = 4 forces it past its
    // x re-read budget, 5 = the streaming kernel, 2 = the tile kernels.
    if ((a.tuning[0] == 4 || (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0)) && plan_mx_rows(a, r.gp, r.lp)) { r.kind = K_KMAJOR; return; }
    // 16-bit activations x block-scaled weights, 1 .. 64 rows (round 4): the A16W8 rows kernel with the scaled converters — while every
    // block's re-read of x (M K 2 bytes x N / 16 blocks) stays in budget.  tuning[0] = 4 forces it, 5 / 2 keep the streaming / tile kernels
    // (round 4, profiles/r04/probe_rows_vs_tiles*.log: against the tile kernel the crossover sits at M N K ~ 570 M for 4096^2-sized layers
    //  — M = 34; M = 64: 19.0 vs 13.0 us — and ~ 250 M for the larger ones: 8192^2 and 14336 x 4096 from 4 rows, M = 16: 25.4 vs 17.7)
    const bool mx16 = a.input_dtype == GEMLITE_DT_MXFP16 || a.input_dtype == GEMLITE_DT_MXBF16;
    const bool a16_over = mx16 && a.M >= 2 && a.M <= 64 && a.N % 128 == 0 && a.K % 128 == 0 &&
                          (int64_t)a.M * a.N * a.K > ((int64_t)a.N * a.K <= (32ll << 20) ? 570000000ll : 250000000ll);
    if (mx16 && a.M <= 64 && (a.tuning[0] == 4 || (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 && !a16_over)) &&
        plan_a16w8_rows(a, r.lp)) { r.kind = K_KMAJOR; return; }
    // 65 .. 384 (512) rows (round 4): 64 x 64 tiles, K unsplit; tuning[0] = 6 forces them, 2 keeps the 128-column kernel with its K slices
    // (profiles/r04/probe_mx_sq.log, `layer(x)` at 4096^2 / 8192^2 / 4096 x 14336 / 14336 x 4096: ahead or within 5 % everywhere up to 384
    //  rows — 4096^2 M = 256: fp8 27.2 -> 17.4 us, fp4 29.7 -> 13.7; at 512 rows for fp4 x fp4 and for one-round shapes)
    // Up to 64 rows they take what the few-row kernel's budget refuses (plan_mx_rows): fp4 x fp4 anywhere, fp8 activations from 128 column tiles.
    const bool sq_few = a.M >= 2 && a.M <= 64 && ((g.mx_x == MX_FP4 && g.mx_w == MX_FP4) || a.N / 64 >= 128);
    // Late round 6 (the tiles walk K in the rotated / grouped order now and are 10 - 25 % faster: profiles/r06/probe_mx_forms*.log): up to 1024 rows — MXFP8 4096^2 M = 768 / 1024
    // 62.5 / 52.2 (128-column tiles) -> 34.8 / 35.2 us, 8192^2 M = 512 / 1024 94.0 / 134.8 -> 82.7 / 127.3, 4096 x 14336 M = 512 87.1 -> 84.2; MXFP4 4096^2 M = 1024 37.9 -> 22.4, but
    // 8192^2 M = 1024 75.0 (256 x 256 tiles) vs 87.5.  2048 rows: the 256 x 256 tiles (59.3 vs 73.5, fp4 43.0 vs 46.3).
    // fp4 WEIGHTS (fp4 x fp4 and fp8 x fp4: 8192^2 M = 1024 111.9 (256 x 256) vs 119.7) above 512 rows only up to N K = 4096^2 (M = 1024: 38.9 -> 33.0).
    const bool mx_w4 = g.mx_w == MX_FP4;
    // ... and a long K (> 8192) under 320 .. 640 of these tiles goes back to the 128-row tiles, whose slice count now minimises rounds x slice length (plan_gemm_mx_mma;
    // profiles/r06/scan_mx_*.log, 20 LLM layer shapes x M = 128 .. 512): MXFP8 5120 x 13824 M = 256 / 512 62.0 / 134.0 -> 55.9 / 104.4 us, 2560 x 9728 M = 448 / 512 48.5 / 47.7 -> 40.3 / 43.4
    // (4096 x 11008 / x 14336 M = 448 / 512 stay although the slices are 7 - 14 % ahead there); fp4 weights only on the largest of them (fp8 x fp4 5120 x 13824: 49.1 / 86.8 -> 40.7 / 76.1; 2560 x 9728 M = 512 the other way, 29.8 vs 32.6)
    const int64_t sq_tiles = (a.N / 64) * ((a.M + 63) / 64);
    // Two of these tiles share a CU, so 257 .. 512 of them cost what 512 do: the hand-over only where they fill less than 0.7 of their last round (320 / 640 tiles: 0.625;
    // at 384 / 480 they win by 1.3 - 1.5x — M = 384 on the same layers, scan_mx_a8w8_long_k_*.log)
    const int64_t sq_slots = (sq_tiles + 511) / 512 * 512;
    // ... and only where one to three slices of the 128-row tiles fill >= 0.9 of one or two rounds of CUs (4096 x 14336 M = 320: 96 tiles x 2 = 0.75 -> 76.8 us vs 55.8 here)
    bool slices_fill = false;
    for (int64_t sk = 1, t128 = (a.N / 128) * ((a.M + 127) / 128), cus = gl::resident_block_limit(); sk <= 3 && sk * 1024 <= a.K && !slices_fill; ++sk) {
        const int64_t rounds = (t128 * sk + cus - 1) / cus;
        slices_fill = rounds <= 2 && t128 * sk * 10 >= rounds * cus * 9;
    }
    const bool long_k_slices = a.M > 128 && a.K > 8192 && a.N % 128 == 0 && sq_tiles > 256 && sq_tiles <= 720 && sq_tiles * 10 < sq_slots * 7 && slices_fill &&
                               (!mx_w4 || ((int64_t)a.N * a.K > (1ll << 26) && g.mx_x == MX_FP8));
    const bool sq_auto = sq_few || (a.M > 64 && !long_k_slices && (a.M <= 512 || (a.M <= 1024 && (!mx_w4 || (int64_t)a.N * a.K <= (1ll << 24)))));
    if ((a.tuning[0] == 6 || (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 && sq_auto)) && plan_gemm_mx_sq(a, r.gp, r.lp)) { r.kind = K_A8_MMA; return; }
    // decode sizes of what is left (K % 64 != 0 ...): the streaming kernel
    if ((a.tuning[0] == 5 || (a.tuning[0] == 0 && !a16_over)) && a.tuning[1] == 0 && a.tuning[2] == 0 && plan_mx_gemv(a, r.gp, r.lp)) { r.kind = K_KMAJOR; return; }
    // prefill sizes of the same-format pairs: 256 x 256 tiles, both operands through LDS (tuning[0] = 3 forces it at any M)
    if (plan_gemm_mx_tile(a, r.gp, r.lp)) { r.kind = K_A8_MMA; return; }
    if ((a.tuning[0] == 0 || a.tuning[0] == 2) && plan_gemm_mx_mma(a, r.gp, r.lp)) { r.kind = K_A8_MMA; return; }
    if ((a.tuning[0] == 0 || a.tuning[0] == 2) && (a.input_dtype == GEMLITE_DT_MXFP16 || a.input_dtype == GEMLITE_DT_MXBF16)) {
        // 16-bit activations: the tiled MFMA kernel of the integer formats with the block-scaled weight geometry
        WnParams p{};
        p.x = a.x; p.w = (const uint32_t*)a.w_q; p.scales = a.scales; p.zeros = nullptr;
        p.epi = g.epi;
        p.epi.meta_dt = p.epi.out_dt;  // no channel scales: keeps the typed epilogue
        p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
        p.group_size = 32;
        p.stride_xm = a.stride_xm; p.stride_xk = a.stride_xk; p.stride_wk = 1;
        p.flags = a.tuning[3];
        if (plan_gemm_wn_mma_mx(a, p, r.lp)) { r.kind = K_TILED_WN; r.wn = p; return; }
    }
    // fp8 / fp4 activations that no tile kernel took (K % 512 != 0 with fp4 activations ...): the few-row kernel over 64-row tiles
    if (a.tuning[0] == 0 && plan_mx_rows(a, r.gp, r.lp, true)) { r.kind = K_KMAJOR; return; }
    // NVFP4, 1 .. 64 rows (round 4): 16-column blocks, both operands expanded to fp16 in registers (tuning[0] = 2 keeps the tile kernel)
    if ((a.tuning[0] == 0 || a.tuning[0] == 4) && a.tuning[1] == 0 && a.tuning[2] == 0 && plan_nvfp4_rows(a, r.gp, r.lp)) { r.kind = K_KMAJOR; return; }
    // NVFP4 (round 4): both operands are exact in fp16 — x is expanded into the workspace by a small kernel in front, the weights in
    // the K loop of the fp16 MFMA tile kernel (Geo<NVW4>), the layer's constant output factor rides as a per-row scale.  Two launches
    // inside this call.  tuning[0] = 1 keeps the coverage kernel.
    if (a.input_dtype == GEMLITE_DT_NVFP4 && a.channel_scale_mode == 4 && (a.tuning[0] == 0 || a.tuning[0] == 2) &&
        ((uintptr_t)a.x % 8) == 0 && a.stride_xm % 8 == 0 && a.K % 128 == 0 && (int64_t)a.M * a.K < (1ll << 30)) {
        gemlite_hip_forward_args b = a;
        b.input_dtype = GEMLITE_DT_MXFP16;
        b.x = (const void*)(uintptr_t)0x1000;  // (alignment checks only: the workspace copy is 256-byte aligned)
        b.stride_xm = a.K;
        WnParams p{};
        p.x = nullptr; p.w = (const uint32_t*)a.w_q; p.scales = a.scales; p.zeros = nullptr;
        p.epi = g.epi;
        p.epi.c_mode = 2;  // out = acc * post[m]
        p.epi.scales_x = nullptr;
        p.epi.stride_sx_m = 1;
        p.epi.meta_dt = p.epi.out_dt;
        p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
        p.group_size = 16;
        p.stride_xm = a.K; p.stride_xk = 1; p.stride_wk = 1;
        p.flags = a.tuning[3];
        if (plan_gemm_wn_mma_mx(b, p, r.lp, 1)) {
            r.kind = K_NV_MMA;
            r.wn = p;
            r.nv_x16_bytes = (uint64_t)(((int64_t)a.M * a.K * 2 + 255) & ~(int64_t)255);
            r.lp.ws_bytes = COUNTER_BYTES + r.lp.slab_bytes + r.nv_x16_bytes + (uint64_t)((a.M * 4 + 255) & ~(int64_t)255);
            return;
        }
    }
    // past their budgets but no tile kernel took the shape (an alignment the tile kernels need ...): the few-row kernels after all,
    // before the coverage kernel
    if (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0) {
        gemlite_hip_forward_args f = a;
        f.tuning[0] = 4;
        if (a16_over && plan_a16w8_rows(f, r.lp)) { r.kind = K_KMAJOR; return; }
        if (a.input_dtype == GEMLITE_DT_NVFP4 && plan_nvfp4_rows(f, r.gp, r.lp)) { r.kind = K_KMAJOR; return; }
    }
    if (a.M > 65535) { r.status = GEMLITE_ERR_BAD_SHAPE; return; }
    r.kind = K_GENERIC;
    r.lp.fn = mx_generic_kernel_fn();
    r.lp.name = "mx_generic_kernel";
    r.lp.grid = dim3((unsigned)((a.N + 255) / 256), (unsigned)a.M, 1);
    r.lp.block = dim3(256, 1, 1);
}

// When the unsplit 64 x 64 A8W8 tiles are the default (profiles/r04/probe_a8w8_sq*.log; everything else keeps the round-3 kernels):
// more than 64 rows and at most one round of tiles over the CUs (4096^2 int8: M = 65 .. 256 17.0 .. 21.7 -> 11.2 .. 13.6 us, 8192^2 M = 128
// 28.4 -> 24.2), or up to two rounds of a short K with two blocks per CU (4096^2 M = 384 / 512: 31.5 / 23.9 -> 19.3 / 21.5 us; at
// K = 8192 two rounds lose: 8192^2 M = 256 39.4 vs 34.4)
// Up to 64 rows (one row tile) they take what the rows kernel's budget refuses from 128 column tiles (plan_a8w8_rows).
static bool a8w8_sq_pays(const gemlite_hip_forward_args& a) {
    if (a.N % 64 != 0) return false;
    if (a.M <= 64) return a.M >= 2 && a.N / 64 >= 128 && a.N / 64 <= 512;
    const int64_t tiles = (a.N / 64) * ((a.M + 63) / 64);
    // late round 6 (profiles/r06/scan_a8w8_*.log, 20 LLM layer shapes x M = 96 .. 512): two rounds hold up to K = 8192 since the requests sit between
    // the MFMAs and the row tiles rotate their K order (8192^2 M = 256: 35.4 -> 32.7 us, 13824 x 5120 M = 128: 37.9 -> 25.0, 3072 x 8192 M = 512:
    // 43.4 -> 27.2; K = 11008 / 13824 still lose: 4096 x 11008 M = 512 58.8 vs 43.7); a very long K under few tiles goes to the K slices of the
    // 128 x 128 tiles (4096 x 14336 M = 96 / 128: 28.2 / 30.6 -> 26.4 / 27.5; M = 256 stays: 30.4 vs 38.2)
    if (a.K >= 14336 && a.M <= 128 && tiles <= 128 && a.N % 128 == 0) return false;
    return tiles <= 256 || (tiles <= 512 && a.K <= 8192);
}

// When gemm_w4_rows_kernel (gemm_wn_rows.hip, round 5) is the default for A16W4 (profiles/r05/probe_rows5_v2*.log: 14 layer shapes x
// M = 2 .. 64 against the round-4 choice, graph-replayed layer(x) in us):
//   * its 16-column blocks hold 146 KB of LDS, so one block per CU: it pays while ONE round of blocks covers N (N / 16 <= CUs: N <= 4096 —
//     4096^2 M = 16 / 32 / 64: 7.4 / 12.2 / 12.9 -> 6.8 / 8.5 / 11.0, 3584 x 4096: 9.0 / 12.0 / 14.8 -> 6.7 / 8.5 / 11.1; at N = 5120 ..
//     14336 two or three rounds lose 1.3 - 2.2x).  From 192 column tiles at any row count below, from 128 (N = 2048) with 16 .. 32 rows;
//   * every block re-reads all of x through L2 -> LDS: past ~176 MB per launch the tile kernels win (4096 x 11008: M = 32 17.7 -> 17.2,
//     M = 48 19.9 -> 21.5; 4096 x 8192 M = 64, 256 MB: 18.3 -> 19.0);
//   * weights are requested two chunks (2 KB per wave) ahead: enough for K <= 12288, not for K = 14336 (M = 16: 14.6 -> 16.3);
//   * from how many rows: K <= 2048 from 2 (4096 x 1024 / x 2048, M = 2: 5.0 / 5.1 -> 3.7 / 4.4); K <= 4096 from 8 (2 .. 7 rows are a
//     draw with the MFMA GEMV / the registers-only kernel: 6.2 - 6.9 vs 6.1 - 6.5); longer K from 2 again since the x pieces are requested
//     ahead of the weights (probe_rows5_v2b*.log: 4096 x 8192 M = 2 / 8 / 16 / 32: 10.1 / 9.3 / 10.5 / 15.6 -> 8.8 / 9.4 / 10.2 / 13.0;
//     4096 x 11008: 12.8 / 17.0 / 17.1 / 17.8 -> 10.5 / 11.4 / 12.3 / 16.7; 3072 x 8192: 9.0 / 11.2 / 14.0 / 15.3 -> 8.7 / 9.4 / 10.1 / 12.8).
//   * 4096 < N <= 8192 (N % 32 = 0, groups of >= 64): TWO column tiles per block (N / 32 resident blocks, an x piece read from LDS feeds both;
//     probe_rows5_nt2*.log) — each CU then streams twice the weights through the same 8 waves, which loses below 17 rows against the
//     registers-only kernel (8192^2 M = 16: 12.5 -> 16.0) and wins where that kernel's K-slice combine or the 32-row tile kernel took over:
//     17 .. 32 rows (8192^2 19.5 -> 18.5, 8192 x 4096 13.6 -> 11.9, 6144 x 4096 13.4 -> 11.4, 5120^2 15.5 -> 13.4; groups of 64 in bf16 at
//     6144 x 4096: 33.5 -> 12.1 — the round-4 planner fell to the LDS-staged streaming kernel there), and 8 .. 64 rows over K <= 2048
//     (8192 x 2048: 6.6 / 7.4 / 9.0 / 11.0 -> 6.2 / 6.3 / 7.3 / 9.4).  Groups of 64 over K > 4096 stay (8192^2 M = 32: 19.0 -> 20.3).
//   * 2-bit words (A16W2, BitNet A16W158; late round 5, probe_rows5_w2*.log): the same kernel, 512 k per chunk.  4096^2 M = 16 / 32 / 48 / 64:
//     6.8 / 11.6 / 12.2 / 12.3 -> 6.1 / 7.6 / 8.8 / 10.1; 3584 x 4096: 7.9 / 11.6 / 13.8 / 14.0 -> 6.0 / 7.5 / 8.7 / 10.1; 4096 x 2048 from 8 rows
//     (5.4 / 6.4 / 8.9 / 9.8 -> 4.4 / 4.5 / 5.5 / 7.1; 2 - 4 rows lose 0.1 - 0.3); 4096 x 14336 up to 32 rows (M = 32: 21.4 -> 17.8; 48 / 64 lose);
//     2048 x 8192 16 .. 32 rows (9.9 / 13.7 -> 9.0 / 11.5); groups of 64 hold 32 rows per block (M = 64 along grid.y loses: 11.6 -> 13.7);
//     one column tile per block only (N <= 4096).
//   * narrow layers (64 <= N / 16 < 192 tiles: the k / v projections of grouped-query models; probe_rows5_narrow_layers.log, 4-bit / 2-bit):
//     K <= 4096 from 8 rows (1024 x 4096 M = 16 / 32 / 64: 7.2 / 10.7 / 12.7 -> 6.3 / 7.7 / 10.6; 2-bit 6.8 / 10.4 / 12.7 -> 5.7 / 6.8 / 9.6;
//     1536 x 4096 the same), K <= 2048 from 4 (1024 x 2048 M = 32 / 64: 8.8 / 10.6 -> 5.0 / 6.7); a longer K only where the registers-only
//     kernel's K-slice combine ends: 16 (128 tiles) / 17 .. 32 rows (1024 x 8192 M = 32: 15.4 -> 12.3; M = 8 / 16 / 64 lose 0.6 - 0.8 us)
static bool rows5_narrow_layer_pays(int64_t M, int64_t tiles, int64_t K) {
    if (tiles < 64) return false;
    if (K <= 4096) return M >= (K <= 2048 ? 4 : 8);
    return M >= (tiles >= 128 ? 16 : 17) && M <= 32;
}
static bool rows5_pays_w2(int64_t M, int64_t N, int64_t K, int gs_shift) {
    if (M < 2 || M > 64 || N % 16 != 0 || K % 512 != 0 || K > 16384) return false;
    const int64_t tiles = N / 16;
    if (tiles > gl::resident_block_limit()) return false;
    const bool g64_two_tiles = gs_shift == 6 && M >= 17 && M <= 32;
    if (tiles < 192 && !(g64_two_tiles && tiles >= 32) && !rows5_narrow_layer_pays(M, tiles, K)) return false;
    if (gs_shift == 6 && M > 32) return false;
    if (gs_shift < 6) return false;  // (groups of 32: 16 rows per block — only where nothing else applies)
    if (M < (K <= 2048 ? 8 : 2)) return false;
    const int64_t bytes = tiles * ((M + 15) / 16 * 16) * K * 2;
    return bytes <= (192ll << 20) || (M <= 32 && bytes <= (256ll << 20));  // (4096 x 8192 M = 48, 192 MiB: 17.6 -> 14.4 us)
}
static bool rows5_pays(int64_t M, int64_t N, int64_t K, int gs_shift) {
    if (M < 2 || M > 64 || N % 16 != 0 || K > 12288) return false;
    const int64_t tiles = N / 16, resident = gl::resident_block_limit();
    if (tiles > resident) {
        if (N % 32 != 0 || tiles / 2 > resident || gs_shift < 6) return false;
        if (K <= 2048 ? M < 8 : (M < 17 || M > 32)) return false;
        if (gs_shift == 6 && (K > 4096 || M > 32)) return false;
        return (tiles / 2) * ((M + 15) / 16 * 16) * K * 2 <= (176ll << 20);
    }
    // groups of 64 at 17 .. 32 rows have no registers-only kernel behind them (only the 32-row MFMA tiles: 1024 x 4096 11.8 vs 7.8 us here)
    const bool g64_two_tiles = gs_shift == 6 && M >= 17 && M <= 32;
    if (tiles < 192 && !(g64_two_tiles && tiles >= 32) && !rows5_narrow_layer_pays(M, tiles, K)) return false;
    const int64_t min_m = (K > 2048 && K <= 4096) ? 8 : 2;
    if (M < min_m) return false;
    return tiles * ((M + 15) / 16 * 16) * K * 2 <= (176ll << 20);
}

// When the unsplit 128 x 128 A8W8 tiles are the default (profiles/r05/probe_a8w8_sq128.log): more than 64 rows, and the 128 x 128 tiles
// number 192 .. 256 — one round with most CUs busy (FP8 16384^2 M = 256: 95.8 vs 98.5 us; int8 8192^2 M = 512: 47.0 vs 52.7;
// 4096^2 M = 1024: 27.5 vs 32.2; 14336 x 4096 M = 256: 29.0 vs 31.1).  Fewer tiles (8192^2 M = 256: 44.5 vs 36.1 us) and several rounds
// (8192^2 M = 1024: 89.8 vs 76.3) stay on the 128- / 256-row tiles with K slices.  Late round 6: from 160 tiles (11008 x 4096 M = 256, 172 tiles: 38.4 -> 28.1 us;
// 128 tiles still lose: 8192^2 M = 256 44.6 vs 34.2 — profiles/r06/probe_a8w8_forms.log).
static bool a8w8_sq128_pays(const gemlite_hip_forward_args& a) {
    // From 144 tiles (6144 x 4096 M = 384: 36.2 -> 26.1 us); a short K (< 4096) only where the 64 x 64 tiles would need more than two rounds
    // (8192 x 2048 / x 3072 M = 384: 30.0 / 33.5 -> 17.9 / 23.1; 8960 x 1536 M = 256, 140 tiles: 28.5 -> 16.4) — profiles/r06/scan_a8w8_*.log.
    if (a.M <= 64 || a.N % 128 != 0) return false;
    const int64_t tiles = (a.N / 128) * ((a.M + 127) / 128);
    if (a.K < 4096) return tiles >= 136 && tiles <= gl::resident_block_limit() && (a.N / 64) * ((a.M + 63) / 64) > 512;
    return tiles >= 144 && tiles <= gl::resident_block_limit();
}

static void resolve(const gemlite_hip_forward_args& a, Resolved& r) {
    r.status = validate(&a);
    if (r.status != GEMLITE_OK) return;
    if (is_mx_input(a.input_dtype)) { resolve_mx(a, r); return; }
    if (const int mxdt = fused_quant_mx_dtype(&a)) {  // one row of a block-scaled dynamic layer, quantiser inside the launch (mx_rows_kernel<..., FQ>)
        r.status = GEMLITE_ERR_NO_FUSED_QUANT;
        if (a.tuning[0] != 0 || a.tuning[1] != 0 || a.tuning[2] != 0 || a.matmul_type != GEMLITE_MATMUL_AUTO) return;
        if (a.stride_xk != 1 || ((uintptr_t)a.x % 16) != 0) return;
        gemlite_hip_forward_args b = a;
        b.input_dtype = mxdt;
        b.x = (const void*)(uintptr_t)0x1000;
        b.scales_x = (const void*)(uintptr_t)0x1000;
        b.stride_xm = mxdt == GEMLITE_DT_MXFP8 ? a.K : a.K / 2;
        b.stride_sx_m = mxdt == GEMLITE_DT_NVFP4 ? a.K / 16 : a.K / 32;
        GenericParams g = mx_params(b);
        LaunchPlan lp{};
        if (!(mxdt == GEMLITE_DT_NVFP4 ? plan_nvfp4_rows(b, g, lp, true) : plan_mx_rows(b, g, lp, false, true))) return;
        g.x = a.x;                 // the raw row
        g.x_dt = a.input_dtype;    // ... and its type
        g.sx_blocks = a.channel_scale_mode == 4 ? a.x : nullptr;  // (non-null: block-scaled activations)
        r.status = GEMLITE_OK;
        r.kind = K_KMAJOR; r.gp = g; r.lp = lp;
        return;
    }
    const bool packed = a.elements_per_sample > 1;
    const bool x16 = a.input_dtype == GEMLITE_DT_FP16 || a.input_dtype == GEMLITE_DT_BF16;
    // metadata rows actually indexed by K: channel-wise / scalar metadata behaves like one K-long group
    const bool per_group_meta = a.W_group_mode >= 2 || ((a.W_group_mode == 1) && !a.zero_is_scalar &&
                                                        !(a.channel_scale_mode == 1 || a.channel_scale_mode == 3));
    const int eff_group = per_group_meta ? a.group_size : (int)a.K;

    // ---- M = 1 of a dynamically quantised PACKED layer with the activation quantiser inside the launch (gemv_a8wn.hip) ---------------
    if (const int qdt = fused_quant_packed_dtype(&a)) {
        r.status = GEMLITE_ERR_NO_FUSED_QUANT;
        if (a.tuning[0] != 0 || a.tuning[1] != 0 || a.tuning[2] != 0 || a.matmul_type != GEMLITE_MATMUL_AUTO) return;
        if (a.stride_xk != 1 || ((uintptr_t)a.x % 16) != 0 || a.output_dtype != a.input_dtype || a.w_pack_bits != 32 || a.stride_wn != 1 || a.stride_on != 1) return;
        const bool pgm = a.W_group_mode >= 2 || ((a.W_group_mode == 1) && !a.zero_is_scalar);
        const int eg = pgm ? a.group_size : (int)a.K;
        if (eg <= 0 || a.K % eg != 0 || (a.stride_meta_n != 1 && pgm)) return;
        gemlite_hip_forward_args b = a;
        b.input_dtype = qdt;
        b.x = (const void*)(uintptr_t)0x1000;
        b.scales_x = (const void*)(uintptr_t)0x1000;
        b.stride_xm = a.K;
        WnParams p{};
        p.x = a.x; p.w = (const uint32_t*)a.w_q; p.scales = a.scales; p.zeros = a.zeros;
        p.epi = make_epilogue(a);
        p.M = 1; p.N = (int)a.N; p.K = (int)a.K;
        p.group_size = eg;
        p.w_mode = a.W_group_mode;
        p.meta_dt = a.meta_dtype; p.zeros_dt = a.zeros_dtype; p.zero_is_scalar = a.zero_is_scalar;
        p.stride_xm = a.K; p.stride_xk = 1; p.stride_wk = a.stride_wk;
        p.stride_meta_g = (pgm && eg < a.K) ? a.stride_meta_g : 0;
        p.flags = a.tuning[3];
        p.gs_shift = eg >= a.K ? 31 : ((eg & (eg - 1)) == 0 ? __builtin_ctz((unsigned)eg) : -1);
        LaunchPlan lp{};
        if (p.gs_shift < 0 || !plan_gemv_a8wn(b, p, lp, true)) return;
        r.status = GEMLITE_OK;
        r.kind = K_GEMV_WN; r.wn = p; r.lp = lp;
        return;
    }
    // ---- specialised packed-weight kernels ---------------------------------------------------------
    const bool x8 = a.input_dtype == GEMLITE_DT_FP8E4 || a.input_dtype == GEMLITE_DT_INT8;  // A8Wn dynamic / BitNet int8
    if (packed && a.w_pack_bits == 32 && (x16 || x8) && a.stride_wn == 1 && a.stride_xk == 1 && a.stride_on == 1 &&
        (a.stride_meta_n == 1 || !per_group_meta) && (a.K % eff_group == 0)) {
        WnParams p{};
        p.x = a.x; p.w = (const uint32_t*)a.w_q; p.scales = a.scales; p.zeros = a.zeros;
        p.epi = make_epilogue(a);
        p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
        p.group_size = eff_group;
        p.w_mode = a.W_group_mode;
        p.meta_dt = a.meta_dtype; p.zeros_dt = a.zeros_dtype; p.zero_is_scalar = a.zero_is_scalar;
        p.stride_xm = a.stride_xm; p.stride_xk = a.stride_xk; p.stride_wk = a.stride_wk;
        // one metadata row spanning all of K ([1, N] scales of a K-long group): the row stride is never used, and views of a
        // [N, 1] tensor carry arbitrary values there (1 for `.t()`), which the planners' alignment checks would reject
        p.stride_meta_g = (per_group_meta && eff_group < a.K) ? a.stride_meta_g : 0;
        p.flags = a.tuning[3];
        p.gs_shift = eff_group >= a.K ? 31
                     : ((eff_group > 0 && (eff_group & (eff_group - 1)) == 0) ? __builtin_ctz((unsigned)eff_group) : -1);
        const int mt = a.matmul_type;
        const bool want_gemv = (mt == GEMLITE_MATMUL_GEMV || mt == GEMLITE_MATMUL_GEMV_REVSPLITK ||
                                mt == GEMLITE_MATMUL_GEMV_SPLITK || (mt == GEMLITE_MATMUL_AUTO && a.M <= 1));
        LaunchPlan lp{};
        if (p.gs_shift < 0) {
            // Group size not a power of two.  Multiples of 32 (round 6, VERDICT r5 #8: the reference only asks for K % group == 0, core.py:253-271) run on
            // the 8-wave tile kernel at every M — its metadata row index is wave-uniform, so k / group is one scalar multiply-high (gs_magic); 32-row tiles
            // carry two (scale, zero) pairs per 64-k sub-block when the group is an odd multiple of 32.  Everything else: the coverage kernel.
            if (eff_group % 32 == 0 && eff_group > 32 && a.K % eff_group == 0 && (int64_t)a.K * eff_group < (1ll << 40) &&
                a.tuning[0] == 0 && (mt == GEMLITE_MATMUL_AUTO || mt == GEMLITE_MATMUL_GEMM || mt == GEMLITE_MATMUL_GEMM_SPLITK)) {
                const uint32_t d = (uint32_t)(eff_group / 32);
                p.gs_magic = (uint32_t)(((1ull << 32) + d - 1) / d);
                // one row: the dot-product GEMV (a lane's 32-k span lies inside one group; the same multiply-high per lane and chunk) — the reference's
                // GEMV runs a group of 96 at 4096 x 3072 in 7.9 us, the 32-row tile below needs 12.4 (profiles/r06/reference_triton_mi355x_r6.json)
                if (want_gemv && a.tuning[1] == 0 && a.tuning[2] == 0 && plan_gemv_wn(a, p, lp)) { r.kind = K_GEMV_WN; r.wn = p; r.lp = lp; return; }
                if (plan_gemm_wn_mma(a, p, lp)) { r.kind = K_TILED_WN; r.wn = p; r.lp = lp; return; }
            }
            goto coverage;
        }
        // 2 .. 64 rows of 8-bit activations x packed weights (round 4): 16-column blocks on the 16-row fp8 / int8 MFMA.  (From 2 rows, not 5:
        // the streaming GEMV below re-reads the weights per row pair — 4096^2 M = 4 `layer(x)` 14.1 vs 9.5 us, 2-bit 14336 x 4096 M = 3 74.7 vs
        // 16.0, profiles/r04/probe_a8wn_fewrows.log.)  tuning[0] = 4 forces it at one row, 7 keeps the GEMV up to 4 rows.
        if (x8 && (want_gemv || mt == GEMLITE_MATMUL_AUTO) && a.tuning[1] == 0 && a.tuning[2] == 0 &&
            (a.tuning[0] == 4 || (a.tuning[0] == 0 && a.M >= 2)) && plan_a8wn_rows(a, p, lp)) {
            r.kind = K_GEMV_WN; r.wn = p; r.lp = lp; return;
        }
        // decode sizes of 8-bit activations x packed weights (A8Wn fp8 dynamic, BitNet int8): per-weight cast to the activation type
        // (one row by default; 2 .. 4 rows only as the A/B switch tuning[0] = 7: past the rows kernel's budget the tile kernel is the faster one)
        if (x8 && a.M <= (a.tuning[0] == 7 ? 4 : 1) && (want_gemv || mt == GEMLITE_MATMUL_AUTO) && (a.tuning[0] == 0 || a.tuning[0] == 7) && a.tuning[1] == 0 && a.tuning[2] == 0 &&
            plan_gemv_a8wn(a, p, lp)) {
            r.kind = K_GEMV_WN; r.wn = p; r.lp = lp; return;
        }
        // round 5: 2 .. 64 rows of 16-bit activations x 4- / 2-bit words on the decode-shaped MFMA rows kernel (gemm_wn_rows.hip): 16-column
        // blocks, K unsplit, weights requested first, x through LDS in whole cache lines.  Every block re-reads all of x (M K 2 bytes through
        // the CU's L2 -> LDS path), so the default stops at a budget on that traffic; groups of 32 and N % 64 != 0 — which no other
        // specialised kernel takes at M >= 2 — always come here (any M: row blocks along grid.y).  tuning[0] = 9 forces the kernel
        // (tuning[1] = 1 / 2 column tiles per block), tuning[3] & 65536 keeps the round-4 choice (A/B runs).
        if (x16 && (a.W_nbits == 4 || a.W_nbits == 2) && (mt == GEMLITE_MATMUL_AUTO || mt == GEMLITE_MATMUL_GEMM_SPLITK || (mt == GEMLITE_MATMUL_GEMM && a.tuning[0] == 9)) &&
            !(a.tuning[3] & 65536) && (a.tuning[0] == 9 || (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 && a.M >= 2))) {
            // nothing FASTER behind this one.  2-bit groups of 32 and N % 64 != 0: only the coverage kernel.  4-bit groups of 32 over N % 64 == 0 do have
            // gemm_wn_stream_kernel behind them (round 4's choice) and ADVICE r5 asked to hand M > 64 back to it, because the rows kernel re-streams
            // the weights once per 32-row block there — measured (round 6, profiles/r06/probe_g32_rows_vs_stream.log, layer(x) in us, rows / stream):
            // 4096^2 M = 64 17.0 / 114, M = 128 31.9 / 184, M = 256 61.5 / 324, M = 1024 242 / 1242; 8192^2 M = 256 208 / 1244; 11008 x 4096 M = 256
            // 172 / 926 — the rows kernel is 5 - 7 x faster at every M, so it keeps every group-32 layer.  (What these layers lack at M > 64 is a
            // group-32 form of the MFMA tile kernel: 61.5 us against 16.8 for groups of 128 at 4096^2 M = 256.)
            // Round 6, later: groups of 32 also have the 32-row tiles of the 8-wave kernel (template parameter NGS = 2, gemm_wn_mma.hip) — for M > 32
            // over N % 128 == 0, K % 256 == 0 they win (4096^2, rows / tiles: M = 128 31.4 / 17.6, M = 256 61.6 / 25.4, M = 1024 244 / 89.9 us;
            // profiles/r06/probe_g32_w4.log), so the rows kernel keeps groups of 32 up to its 32 rows per block and wherever the tiles do not apply.
            // (profiles/r06/probe_g32_w4_m33_64.log, rows / tiles: 4096^2 M = 40 .. 64 16.8 .. 17.3 / 13.2; 8192^2 M = 40 .. 64 52 / 27 and M = 24 / 32
            //  27.5 / 18.3; 11008 x 4096 M = 40 .. 64 47.5 / 30 — from 17 rows where the rows kernel's N / 16 blocks need more than one round)
            const bool g32_tiles = p.gs_shift == 5 && a.N % 128 == 0 && a.K % 256 == 0 &&
                                   (a.M > 32 || (a.M > 16 && a.K >= 2048 && a.N / 16 > gl::resident_block_limit()));
            const bool only_here = (p.gs_shift == 5 && !g32_tiles) || a.N % 64 != 0;
            const bool in_budget = a.W_nbits == 4 ? rows5_pays(a.M, a.N, a.K, p.gs_shift) : rows5_pays_w2(a.M, a.N, a.K, p.gs_shift);
            if (a.tuning[0] == 9 || only_here || (in_budget && !g32_tiles)) {
                WnParams pr = p;
                LaunchPlan lr{};
                if (plan_gemm_wn_rows(a, pr, lr)) { r.kind = K_STREAM_WN; r.wn = pr; r.lp = lr; return; }
            }
        }
        // Decode on the matrix core (gemv_mfma.hip, round 3) where it measured faster than the dot-product family
        // (profiles/r03/probe_gemv3_*.log, us per launch in a replayed graph):
        //   4-bit, M = 1: 32-column tiles over a long K (8192^2: 9.9 vs 10.4).  Not the 16-column shapes (4096^2: 5.8 vs 4.8 for the
        //     dot-product decode kernel), not 64-column tiles (16384^2: 25.3 vs 23.7), not 32-column tiles over K = 4096 (9.3 vs 8.9);
        //   2-bit, M = 1: 16- and 32-column tiles (4096^2 4.4 vs 4.9, 8192^2 7.4 vs 8.6, 11008 x 4096 6.1 vs 8.2; 16384^2 18.0 vs 17.4: no);
        //   4-bit, 2..4 rows: N < 12288 (4096^2: 5.8 - 6.2 vs 6.4 - 7.5 for gemm_wn_direct, 6144 x 4096 5.9 vs 7.4, 5120 x 13824 17.1 vs 21.2,
        //     1536 x 8960 10.7 vs 14.9) except a long K over a narrow N, which takes the registers-only kernel with 64-column tiles and
        //     K slices (4096 x 14336 11.7 vs 16.5, 8192 x 28672 22.2 vs 28.5); N >= 12288: the registers-only kernel's 64-column tiles
        //     unsplit (13824 x 5120 10.9 vs 11.6 .. 17.6, 28672 x 8192 21.9 vs 30.5) — profiles/r03/probe_m4_llm_shapes_*.log.
        // tuning[3] & 512 = never, & 1024 = wherever it applies (A/B runs).
        if (x16 && a.W_nbits == 4 && a.M >= 2 && a.M <= 4 && mt == GEMLITE_MATMUL_AUTO && a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 &&
            !(a.tuning[3] & 1024) && a.K >= 12288 && a.N <= 8192 && a.N % 64 == 0 && a.N / 64 >= 40) {
            gemlite_hip_forward_args a2 = a;
            a2.tuning[0] = 4;
            a2.tuning[1] = a.N / 64 >= 128 ? 2 : 4;
            WnParams pd = p;
            LaunchPlan ld{};
            if (plan_gemm_wn_direct(a2, pd, ld)) { r.kind = K_STREAM_WN; r.wn = pd; r.lp = ld; return; }
        }
        if (x16 && !(a.tuning[3] & 512) && a.tuning[1] == 0 &&
            ((want_gemv && a.M == 1 && mt != GEMLITE_MATMUL_GEMV_SPLITK) || (a.M >= 2 && a.M <= 4 && mt == GEMLITE_MATMUL_AUTO))) {
            WnParams pm = p;
            LaunchPlan lm{};
            if (plan_gemv_mfma(a, pm, lm)) {
                const int cols = (int)(a.N / lm.grid.x);
                // (round 3 sweep over 22 LLM shapes: 4-bit M = 1 also wins on every 6144 <= N <= 12288 with K <= 8192 — 6144 x 4096 5.6
                //  vs 7.0 us, 8192 x 3072 5.6 vs 6.2, 8960 x 1536 5.1 vs 5.5, 11008 x 4096 7.9 vs 8.5 — and loses on narrow N, K > 8192
                //  under 64-column tiles, and N >= 13824)
                // Round 6: with gemv_wn_kernel on counted asm loads (no drain at the head of its chunk loop) the dot-product family is ahead at ONE row
                // on 20 of 22 LLM shapes for 4-bit words (8192^2 9.14 vs 9.86 us, 12288 x 4096 7.76 vs 8.56, 11008 x 4096 7.70 vs 7.98; 6144 x 4096
                // 5.73 vs 5.49 the other way) and on 16 of 22 for 2-bit words (5120^2 5.65 vs 7.17, 13824 x 5120 7.26 vs 9.00, 28672 x 8192 16.2 vs 20.6,
                // 8192^2 7.07 vs 7.87) — profiles/r06/probe_m1_shapes_w{4,2}.log.  The matrix-core GEMV keeps one row only for 2-bit words over a K that
                // is not a multiple of 1024 (K = 8960 / 9728 / 11008: 7.26 vs 8.04, 7.84 vs 8.22, 8.29 vs 9.16), and 2 .. 4 rows as before.
                (void)cols;
                const bool wins = a.M >= 2 ? (a.W_nbits != 4 || a.N < 12288) : (a.W_nbits == 2 && a.K % 1024 != 0);
                if ((a.tuning[3] & 1024) || a.tuning[0] != 0 || wins) { r.kind = K_GEMV_WN; r.wn = pm; r.lp = lm; return; }
            }
        }
        if (want_gemv && plan_gemv_wn(a, p, lp)) { r.kind = K_GEMV_WN; r.wn = p; r.lp = lp; return; }
        if (!want_gemv) {
            // Many rows: the 8-wave MFMA kernel (all bit widths) from 33 rows.  tuning[0]: 1 = LDS-staged streaming kernel,
            // 2 = the 4-wave tiled kernel of round 1 (4-bit only; kept for A/B runs)
            const bool want_tiled = (mt == GEMLITE_MATMUL_GEMM || (mt == GEMLITE_MATMUL_AUTO && a.M > 32));
            if (want_tiled && a.tuning[0] == 0 && plan_gemm_wn_mma(a, p, lp)) { r.kind = K_TILED_WN; r.wn = p; r.lp = lp; return; }
            if (want_tiled && a.tuning[0] != 1 && plan_gemm_wn_tiled(a, p, lp)) { r.kind = K_TILED_WN; r.wn = p; r.lp = lp; return; }
            // 17..32 rows over a long K: the 8-wave kernel's LDS-staged x beats the
            // registers-only kernel, whose blocks each re-read their K range of every row from L2 (M = 32: N 5120 x K 13824
            // 25.3 vs 39.8 us, 8192 x 28672 51 vs 72, 28672 x 8192 49.5 vs 60; K <= 5120 the other way round —
            // profiles/r02/autotune_report.json)
            const bool long_k = mt == GEMLITE_MATMUL_AUTO && a.W_nbits == 4 && a.tuning[0] == 0 && a.tuning[2] == 0 &&
                                a.M > 16 && a.K >= 8192;
            if (long_k && plan_gemm_wn_mma(a, p, lp)) { r.kind = K_TILED_WN; r.wn = p; r.lp = lp; return; }
            // few rows: registers-only MFMA path (tuning[2] == 1 keeps the LDS-staged streaming kernel)
            if (a.M <= 32 && a.tuning[2] != 1 && a.tuning[0] != 3 && plan_gemm_wn_direct(a, p, lp)) { r.kind = K_STREAM_WN; r.wn = p; r.lp = lp; return; }
            // 17 .. 32 rows the registers-only kernel refuses (groups of 64 — hqq's default — need one row tile there): rounds 2-4 fell to the
            // LDS-staged streaming kernel, 1.3 - 2.2x behind the 32-row MFMA tiles (profiles/r05/probe_g64_m17_32_w{4,2}.log, 4-bit / 2-bit:
            // 4096^2 16.3 / 15.7 -> 12.9 / 11.9 us, 6144 x 4096 2-bit 23.1 -> 12.2, 11008 x 4096 31.4 / 31.0 -> 16.5 / 22.6, 8192^2 2-bit 25.3 -> 17.2);
            // a short K keeps the streaming kernel (8960 x 1536: 10.6 vs 11.1)
            if (mt == GEMLITE_MATMUL_AUTO && a.M > 16 && a.K >= 2048 && a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 && !(a.tuning[3] & 65536) &&
                plan_gemm_wn_mma(a, p, lp)) { r.kind = K_TILED_WN; r.wn = p; r.lp = lp; return; }
            if (a.tuning[0] != 3 && plan_gemm_wn_stream(a, p, lp)) { r.kind = K_STREAM_WN; r.wn = p; r.lp = lp; return; }
            // shapes the few-row kernels do not take (K = 11008, 8960, ...): small tiles of the MFMA kernel, never the
            // coverage kernel (tuning[0] == 3 forces this path for A/B runs)
            if (plan_gemm_wn_mma(a, p, lp)) { r.kind = K_TILED_WN; r.wn = p; r.lp = lp; return; }
        }
        // AUTO with small M that the GEMV planner rejected may still fit the streaming kernel
        if (want_gemv && mt == GEMLITE_MATMUL_AUTO && plan_gemm_wn_stream(a, p, lp)) {
            r.kind = K_STREAM_WN; r.wn = p; r.lp = lp; return;
        }
        // what none of the above takes at M = 1 — 8-bit activations x packed weights (A8Wn dynamic, BitNet int8), an output
        // or channel-scale type that differs from the activations' — still has the 8-wave MFMA kernel (32-row tiles)
        if (want_gemv && plan_gemm_wn_mma(a, p, lp)) { r.kind = K_TILED_WN; r.wn = p; r.lp = lp; return; }
    }

coverage:
    // ---- coverage kernels ---------------------------------------------------------------------------
    GenericParams g{};
    g.x = a.x; g.w = a.w_q; g.scales = a.scales; g.zeros = a.zeros;
    g.epi = make_epilogue(a);
    g.M = (int)a.M; g.N = (int)a.N; g.K = (int)a.K;
    g.nbits = a.W_nbits; g.e = a.elements_per_sample; g.pack_bits = a.w_pack_bits;
    g.w_dt = a.w_dtype; g.x_dt = a.input_dtype;
    g.group_size = eff_group; g.w_mode = a.W_group_mode;
    g.meta_dt = a.meta_dtype; g.zeros_dt = a.zeros_dtype; g.zero_is_scalar = a.zero_is_scalar;
    g.int_acc = (a.input_dtype == GEMLITE_DT_INT8 && (packed || a.w_dtype == GEMLITE_DT_INT8)) ? 1 : 0;
    g.stride_xm = a.stride_xm; g.stride_xk = a.stride_xk; g.stride_wk = a.stride_wk; g.stride_wn = a.stride_wn;
    g.stride_meta_g = per_group_meta ? a.stride_meta_g : 0;
    g.stride_meta_n = per_group_meta ? a.stride_meta_n : ((a.W_group_mode == 1 && !a.zero_is_scalar) ? 1 : 0);
    r.gp = g;
    // 2 <= M <= 64 of a dynamically quantised layer: the rows kernel with producer blocks in front that quantise x (gl_coopquant.h).
    // Planned on the arguments the matmul will see — 8-bit x [M, K] (row stride K) and fp32 scales in the workspace; the pointers
    // are filled in at launch.  GEMLITE_ERR_NO_FUSED_QUANT: ask again with quantised x and scales_x (two launches).
    if (wants_fused_quant(&a) && a.M >= 2) {
        r.status = GEMLITE_ERR_NO_FUSED_QUANT;
        if (a.tuning[0] != 0 || a.tuning[1] != 0 || a.tuning[2] != 0 || a.matmul_type != GEMLITE_MATMUL_AUTO) return;
        if (a.M > cq::MAX_ROWS || a.stride_xk != 1 || a.K % 16 != 0 || a.stride_xm % 8 != 0 || ((uintptr_t)a.x % 16) != 0) return;
        gemlite_hip_forward_args b = a;
        b.input_dtype = a.w_dtype;
        b.x = (const void*)(uintptr_t)0x1000;  // (alignment checks only: the workspace copies are 256-byte aligned)
        b.scales_x = (const void*)(uintptr_t)0x1000;
        b.stride_xm = a.K;
        b.stride_sx_m = 1;
        LaunchPlan lp{};
        GenericParams gq = g;
        if (a.M > 64 || !plan_a8w8_rows(b, lp, true)) return;
        gq.x = nullptr; gq.x_dt = a.w_dtype; gq.stride_xm = a.K; gq.stride_xk = 1;
        gq.int_acc = a.w_dtype == GEMLITE_DT_INT8 ? 1 : 0;
        gq.epi.scales_x = nullptr; gq.epi.stride_sx_m = 1;
        gq.cq_x = a.x; gq.cq_stride_xm = a.stride_xm; gq.cq_xdt = a.input_dtype;
        gq.flags = a.tuning[3];
        r.status = GEMLITE_OK;
        r.kind = K_A8_FQ;
        r.gp = gq;
        r.lp = lp;
        return;
    }
    // M = 1 of a dynamically quantised layer with the activation quantisation fused into the prologue
    if (wants_fused_quant(&a)) {
        if (a.stride_wk != 1 || a.stride_xk != 1 || a.K % 16 != 0 || a.stride_wn % 16 != 0 || a.K > 65536 ||
            (((uintptr_t)a.w_q | (uintptr_t)a.x) % 16) != 0) { r.status = GEMLITE_ERR_UNSUPPORTED; return; }
        r.kind = K_KMAJOR;
        // round 4: one wave per column with the whole weight row requested BEFORE the block quantises x (a8w8_decode_kernel);
        // tuning[0] = 7 keeps the round-2 kernel, which quantises first
        // (profiles/r04/probe_a8w8_decode.log: int8 faster on all 8 shapes tried, 4096^2 7.27 -> 6.18 us, 4096 x 14336 18.9 -> 14.5; fp8 —
        //  8 converter + 16 fma instructions per 16 bytes — only while the tiles are one round of blocks: 4096^2 7.80 -> 6.62, but
        //  8192^2 17.3 -> 18.9, 16384^2 54.5 -> 58.7)
        const bool dec_ok = a.tuning[0] != 7 && a.K % 1024 == 0 && a.N % 16 == 0 && (a.w_dtype == GEMLITE_DT_INT8 || a.N <= 4096 || a.tuning[0] == 8);
        if (dec_ok && a.stride_on == 1 && a.K <= 61440) {
            r.lp.fn = a8w8_decode_kernel_fn(a.w_dtype, true);
            r.lp.name = "a8w8_decode_fused_quant_kernel<tile16,16w>";
            r.lp.grid = dim3((unsigned)(a.N / 16), 1, 1);
            r.lp.block = dim3(1024, 1, 1);
            r.lp.lds_bytes = (size_t)a.K + 64;
            return;
        }
        r.lp.fn = kmajor_fused_quant_kernel_fn(a.w_dtype);
        r.lp.name = "kmajor_fused_quant_kernel";
        r.lp.grid = dim3((unsigned)((a.N + 7) / 8), 1, 1);
        r.lp.block = dim3(512, 1, 1);
        r.lp.lds_bytes = (size_t)((a.K + 15) & ~15) + 64;
        return;
    }
    // A8W8 (int8 / fp8), 2..16 rows: 16-column blocks, one 16-row MFMA per 64-k chunk (tuning[0] = 4 forces it at M = 1 too,
    // any other non-zero tuning[0] skips it)
    if (!packed && (a.tuning[0] == 4 || (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 && a.M >= 2)) &&
        a.matmul_type != GEMLITE_MATMUL_GEMV && a.matmul_type != GEMLITE_MATMUL_GEMV_SPLITK &&
        a.matmul_type != GEMLITE_MATMUL_GEMV_REVSPLITK && plan_a8w8_rows(a, r.lp)) {
        r.kind = K_KMAJOR;  // GenericParams, no workspace
        return;
    }
    // A8W8 (int8 / fp8), round 5: 128 x 128 tiles with K unsplit where THEY number about one per CU (FP8 x FP8 16384^2 at M = 256: 256 tiles) —
    // tuning[0] = 10 forces them, tuning[0] = 6 keeps the round-3 choice
    if (!packed && a.matmul_type != GEMLITE_MATMUL_GEMV && a.matmul_type != GEMLITE_MATMUL_GEMV_SPLITK &&
        a.matmul_type != GEMLITE_MATMUL_GEMV_REVSPLITK &&
        (a.tuning[0] == 10 || (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 && !(a.tuning[3] & 64) && a8w8_sq128_pays(a))) && plan_gemm_a8w8_sq128(a, r.gp, r.lp)) {
        r.kind = K_A8_MMA;
        return;
    }
    // A8W8 (int8 / fp8), round 4: 64 x 64 tiles with K unsplit where they fill the chip about once (config 4 at M = 256: 256 tiles) —
    // tuning[0] = 5 forces them, tuning[0] = 6 keeps the round-3 choice
    if (!packed && a.matmul_type != GEMLITE_MATMUL_GEMV && a.matmul_type != GEMLITE_MATMUL_GEMV_SPLITK &&
        a.matmul_type != GEMLITE_MATMUL_GEMV_REVSPLITK &&
        (a.tuning[0] == 5 || (a.tuning[0] == 0 && a.tuning[1] == 0 && a.tuning[2] == 0 && !(a.tuning[3] & 64) && a8w8_sq_pays(a))) && plan_gemm_a8w8_sq(a, r.gp, r.lp)) {
        r.kind = K_A8_MMA;
        return;
    }
    // A8W8 (int8 / fp8) from 2 rows: the 8-wave MFMA kernel.  tuning[0]: 1 = streaming kernel (one wave per column),
    // 2 = the 4-wave MFMA kernel of round 1 (M >= 32)
    if (!packed && (a.tuning[0] == 0 || a.tuning[0] == 6) && a.matmul_type != GEMLITE_MATMUL_GEMV && a.matmul_type != GEMLITE_MATMUL_GEMV_SPLITK &&
        a.matmul_type != GEMLITE_MATMUL_GEMV_REVSPLITK && plan_gemm_a8w8_mma(a, r.gp, r.lp)) {
        r.kind = K_A8_MMA;
        return;
    }
    if (!packed && a.tuning[0] != 1 && a.matmul_type != GEMLITE_MATMUL_GEMV && a.matmul_type != GEMLITE_MATMUL_GEMV_SPLITK &&
        a.matmul_type != GEMLITE_MATMUL_GEMV_REVSPLITK && plan_gemm_a8w8(a, r.lp)) {
        r.kind = K_KMAJOR;  // same launch path: GenericParams, no workspace
        return;
    }
    // A16W8: 8-bit unpacked weights under 16-bit activations, no metadata or one pre-scale per channel
    if (!packed && (a.W_group_mode == 0 || (a.W_group_mode == 2 && eff_group >= a.K && a.scales)) && a.stride_wk == 1 &&
        a.stride_xk == 1 && (a.input_dtype == GEMLITE_DT_FP16 || a.input_dtype == GEMLITE_DT_BF16) &&
        (a.w_dtype == GEMLITE_DT_INT8 || a.w_dtype == GEMLITE_DT_FP8E4 || a.w_dtype == GEMLITE_DT_FP8E5) &&
        a.K % 16 == 0 && a.stride_wn % 16 == 0 && (a.stride_xm * 2) % 16 == 0 && (((uintptr_t)a.w_q | (uintptr_t)a.x) % 16 == 0)) {
        r.kind = K_KMAJOR;
        // round 4, one row: a wave per column, the weight row in flight before x is staged (tuning[0] = 4 keeps the rows kernel)
        if (a.M == 1 && a.tuning[0] == 0 && a.K % 1024 == 0 && a.K <= 30720 && a.N % 16 == 0) {
            r.lp.fn = a16w8_decode_kernel_fn(a.input_dtype, a.w_dtype);
            r.lp.name = "a16w8_decode_kernel<tile16,16w>";
            r.lp.grid = dim3((unsigned)(a.N / 16), 1, 1);
            r.lp.block = dim3(1024, 1, 1);
            r.lp.lds_bytes = (size_t)a.K * 2;
            return;
        }
        // round 4, more than 64 rows: the 8-wave MFMA tile kernel with the K-contiguous 8-bit geometry (x through LDS, weights converted
        // in registers); the per-channel pre-scale becomes the epilogue's channel scale.  tuning[0] = 4 keeps the rows kernel.
        // (rows vs tile, profiles/r04/probe_rows_vs_tiles*.log: the crossover sits at M N K ~ 850 M — 4096^2: M = 52, 8192^2: 13 (M = 64 there:
        //  67.6 vs 29.1 us), 14336 x 4096: 12, 4096 x 14336: 19)
        //  round 6: the rows kernel with x through LDS first where IT is the faster one — a16w8_rows_lds_pays(), gemm_a8w8.hip)
        if (a.tuning[0] == 0 && a16w8_rows_lds_pays(a) && plan_a16w8_rows(a, r.lp)) return;
        if ((a.tuning[0] == 2 || ((a.M > 64 || (a.M >= 2 && (int64_t)a.M * a.N * a.K > 850000000ll)) && a.tuning[0] == 0)) &&
            !(a.W_group_mode == 2 && a.channel_scale_mode != 0)) {
            WnParams p{};
            p.x = a.x; p.w = (const uint32_t*)a.w_q; p.scales = a.scales; p.zeros = nullptr;
            p.epi = make_epilogue(a);
            if (a.W_group_mode == 2) { p.epi.c_mode = 1; p.epi.scales_w = a.scales; }
            p.M = (int)a.M; p.N = (int)a.N; p.K = (int)a.K;
            p.group_size = 32;
            p.stride_xm = a.stride_xm; p.stride_xk = a.stride_xk; p.stride_wk = 1;
            p.flags = a.tuning[3];
            if (plan_gemm_wn_mma_mx(a, p, r.lp, 2)) { r.kind = K_TILED_WN; r.wn = p; return; }
        }
        // round 4: 16-column blocks, weights converted in registers, MFMA (tuning[0] = 7 keeps the streaming kernel of rounds 1-3)
        if (a.tuning[0] != 7 && plan_a16w8_rows(a, r.lp)) return;
        const int mb = a.M == 1 ? 1 : 4;
        r.lp.fn = kmajor_w8a16_kernel_fn(mb);
        r.lp.name = "kmajor_w8a16_kernel";
        r.lp.grid = dim3((unsigned)((a.N + 3) / 4), (unsigned)((a.M + mb - 1) / mb), 1);
        r.lp.block = dim3(256, 1, 1);
        return;
    }
    const int esz = dtype_size(a.w_dtype);
    if (!packed && a.W_group_mode == 0 && a.stride_wk == 1 && a.stride_xk == 1 && a.w_dtype == a.input_dtype &&
        esz > 0 && (a.K % (16 / esz) == 0) && ((a.stride_wn * esz) % 16 == 0) && ((a.stride_xm * esz) % 16 == 0) &&
        (((uintptr_t)a.w_q | (uintptr_t)a.x) % 16 == 0) &&
        !(a.M > 1 && (a.w_dtype == GEMLITE_DT_FP8E4 || a.w_dtype == GEMLITE_DT_FP8E5))) {  // fp8 rows > 1: MFMA kernels / coverage
        const int mb = a.M == 1 ? 1 : 4;
        r.kind = K_KMAJOR;
        // round 4: the 8-bit decode kernel (one wave per column, the whole row in flight at once); tuning[0] = 7 keeps kmajor_matmul_kernel
        if (a.M == 1 && esz == 1 && a.tuning[0] != 7 && a.K % 1024 == 0 && a.N % 16 == 0 &&
            (a.w_dtype == GEMLITE_DT_INT8 || ((a.w_dtype == GEMLITE_DT_FP8E4 || a.w_dtype == GEMLITE_DT_FP8E5) && (a.N <= 4096 || a.tuning[0] == 8)))) {
            r.lp.fn = a8w8_decode_kernel_fn(a.w_dtype, false);
            r.lp.name = "a8w8_decode_kernel<tile16,16w>";
            r.lp.grid = dim3((unsigned)(a.N / 16), 1, 1);
            r.lp.block = dim3(1024, 1, 1);
            return;
        }
        r.lp.fn = kmajor_kernel_fn(mb);
        r.lp.name = "kmajor_matmul_kernel";
        r.lp.grid = dim3((unsigned)((a.N + 3) / 4), (unsigned)((a.M + mb - 1) / mb), 1);
        r.lp.block = dim3(256, 1, 1);
        return;
    }
    if (a.M > 65535) { r.status = GEMLITE_ERR_BAD_SHAPE; return; }
    r.kind = K_GENERIC;
    r.lp.fn = generic_kernel_fn();
    r.lp.name = "generic_matmul_kernel";
    r.lp.grid = dim3((unsigned)((a.N + 255) / 256), (unsigned)a.M, 1);
    r.lp.block = dim3(256, 1, 1);
}

static int launch(const void* fn, dim3 grid, dim3 block, void** kargs, size_t lds, hipStream_t stream) {
    hipError_t err;
    if (tl_evt_start || tl_evt_stop) {
        err = hipExtLaunchKernel(fn, grid, block, kargs, lds, stream, (hipEvent_t)tl_evt_start, (hipEvent_t)tl_evt_stop, 0);
        tl_evt_start = tl_evt_stop = nullptr;
    } else {
        err = hipLaunchKernel(fn, grid, block, kargs, lds, stream);
    }
    if (err != hipSuccess) {
        tl_last_hip_error = (int)err;
        (void)hipGetLastError();
        return GEMLITE_ERR_LAUNCH;
    }
    return GEMLITE_OK;
}

// CU count PER DEVICE (one process-global minimum made the planners' choices depend on which other devices the process had
// touched).  gemlite_hip_forward names the device it plans for (tl_plan_dev); the host-only queries (gemlite_hip_kernel_name,
// gemlite_hip_query, gemlite_hip_workspace_bytes) plan for a full 256-CU part, whatever was launched before.
static std::atomic<int> g_cu_by_dev[64];
static thread_local int tl_plan_dev = -1;
int gl::resident_block_limit() {
    if (tl_plan_dev < 0) return 256;
    const int n = g_cu_by_dev[tl_plan_dev & 63].load(std::memory_order_relaxed);
    return n > 0 ? n : 256;
}

// The current device must be a gfx950 part: the code object holds no other ISA.  Checked once per device id.
static int check_device(int* dev_out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) { (void)hipGetLastError(); return GEMLITE_ERR_NO_DEVICE; }
    *dev_out = dev;
    static std::atomic<int> state[64];  // 0 unknown | 1 gfx950 | 2 something else
    const int slot = dev & 63;
    int st = state[slot].load(std::memory_order_relaxed);
    if (st == 0) {
        hipDeviceProp_t prop;
        st = (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : 2;
        // what the planners may assume co-resident on THIS device (partitioned devices report fewer CUs)
        if (st == 1 && prop.multiProcessorCount > 0) g_cu_by_dev[slot].store(prop.multiProcessorCount, std::memory_order_relaxed);
        state[slot].store(st, std::memory_order_relaxed);
    }
    return st == 1 ? GEMLITE_OK : GEMLITE_ERR_NO_DEVICE;
}

// Raise the dynamic-LDS cap of a kernel above 64 KiB: once per (device, kernel) — the attribute is per device.
static int ensure_lds(const void* fn, size_t lds, int dev) {
    if (lds <= 65536) return GEMLITE_OK;
    struct Done { const void* fn; int dev; };
    static thread_local Done done[32] = {};
    for (const Done& d : done) if (d.fn == fn && d.dev == dev) return GEMLITE_OK;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) { tl_last_hip_error = (int)err; (void)hipGetLastError(); return GEMLITE_ERR_LAUNCH; }
    static thread_local int next = 0;
    done[next] = Done{fn, dev};
    next = (next + 1) & 31;
    return GEMLITE_OK;
}

__global__ void gemlite_noop_kernel() {}

extern "C" {

int gemlite_hip_abi_version(void) { return GEMLITE_HIP_ABI_VERSION; }

const char* gemlite_hip_build_info(void) {
    return "libgemlite_hip gfx950 (CDNA4) abi=1 kernels: gemv_wn, gemv_decode, gemv_mfma, gemv_a8wn, gemm_wn_rows, gemm_wn_direct, gemm_wn_stream, gemm_wn_mma, gemm_wn_tiled, gemm_a8w8, "
           "gemm_mx, mx_rows, nvfp4_f16, kmajor, generic, act_quant_per_token, act_quant_mx, pack/unpack_over_cols"
#ifdef GL_AB_KERNELS
           " +ab_kernels"
#endif
        ;
}

const char* gemlite_hip_status_string(int status) {
    switch (status) {
        case GEMLITE_OK: return "ok";
        case GEMLITE_ERR_BAD_ARGUMENT: return "bad argument (null pointer, non-positive size or ABI struct_size mismatch)";
        case GEMLITE_ERR_UNSUPPORTED: return "unsupported dtype / bit-width / mode combination";
        case GEMLITE_ERR_BAD_SHAPE: return "bad shape (K not divisible by elements_per_sample / group_size, or too large)";
        case GEMLITE_ERR_WORKSPACE: return "workspace missing or too small";
        case GEMLITE_ERR_LAUNCH: return "HIP launch failed (see gemlite_hip_last_hip_error)";
        case GEMLITE_ERR_NO_DEVICE: return "current device is not gfx950";
        case GEMLITE_ERR_NO_FUSED_QUANT: return "no kernel with in-launch activation quantisation for this shape: quantise x first and pass scales_x";
        default: return "unknown status";
    }
}

int gemlite_hip_last_hip_error(void) { return tl_last_hip_error; }

int gemlite_hip_query(const gemlite_hip_forward_args* args) {
    const int v = validate(args);
    if (v != GEMLITE_OK) return v;
    Resolved r;
    resolve(*args, r);
    return r.status;
}

uint64_t gemlite_hip_workspace_bytes(const gemlite_hip_forward_args* args) {
    if (validate(args) != GEMLITE_OK) return 0;
    Resolved r;
    resolve(*args, r);
    return r.status == GEMLITE_OK ? r.lp.ws_bytes : 0;
}

const char* gemlite_hip_kernel_name(const gemlite_hip_forward_args* args) {
    if (validate(args) != GEMLITE_OK) return "invalid";
    Resolved r;
    resolve(*args, r);
    return r.status == GEMLITE_OK ? r.lp.name : "unsupported";
}

void gemlite_hip_set_profile_events(void* start_event, void* stop_event) {
    tl_evt_start = start_event;
    tl_evt_stop = stop_event;
}

int gemlite_hip_launch_noop(int32_t blocks, int32_t threads, void* stream) {
    if (blocks <= 0 || threads <= 0 || threads > 1024) return GEMLITE_ERR_BAD_ARGUMENT;
    void* none[1] = {nullptr};
    return launch((const void*)gemlite_noop_kernel, dim3((unsigned)blocks, 1, 1), dim3((unsigned)threads, 1, 1), none, 0,
                  (hipStream_t)stream);
}

int gemlite_hip_forward(const gemlite_hip_forward_args* args, void* stream) {
    const int v = validate(args);
    if (v != GEMLITE_OK) return v;
    int dev = 0;
    const int dv = check_device(&dev);  // (before planning: the planners ask for the device's CU count)
    Resolved r;
    tl_plan_dev = dv == GEMLITE_OK ? dev : -1;
    resolve(*args, r);
    tl_plan_dev = -1;
    if (r.status != GEMLITE_OK) return r.status;
    if (dv != GEMLITE_OK) return dv;
    hipStream_t st = (hipStream_t)stream;
    if (r.kind == K_GEMV_WN || r.kind == K_STREAM_WN || r.kind == K_TILED_WN) {
        if (r.lp.ws_bytes > 0) {
            if (!args->workspace || args->workspace_bytes < r.lp.ws_bytes) return GEMLITE_ERR_WORKSPACE;
            r.wn.counters = (unsigned*)args->workspace;
            r.wn.slabs = (float*)((char*)args->workspace + COUNTER_BYTES);
        } else if (args->workspace && args->workspace_bytes >= COUNTER_BYTES) {
            r.wn.counters = (unsigned*)args->workspace;  // lets the opt-in timeline probes (tuning[3] & 4) find a buffer
        }
        const int e = ensure_lds(r.lp.fn, r.lp.lds_bytes, dev);
        if (e != GEMLITE_OK) return e;
        if (r.lp.arg_kind == 1) {  // scalar (SGPR-preloaded) arguments: gemv_decode.hip
            Decode3Args& d = r.lp.d3;
            d.counters = r.wn.counters;
            void* dargs[] = {(void*)&d.w, (void*)&d.x, (void*)&d.s, (void*)&d.z, (void*)&d.out, (void*)&d.sw4, (void*)&d.mstride2,
                             (void*)&d.nch_total, (void*)&d.modes, (void*)&d.counters};
            return launch(r.lp.fn, r.lp.grid, r.lp.block, dargs, 0, st);
        }
        if (r.lp.arg_kind == 2) {  // gemm_wn_rows.hip
            Rows5Args& d = r.lp.r5;
            void* dargs[] = {(void*)&d.w, (void*)&d.x, (void*)&d.s, (void*)&d.z, (void*)&d.out, (void*)&d.sw4, (void*)&d.mstride2,
                             (void*)&d.nch_total, (void*)&d.modes, (void*)&d.M, (void*)&d.sxm2, (void*)&d.som};
            return launch(r.lp.fn, r.lp.grid, r.lp.block, dargs, r.lp.lds_bytes, st);
        }
        void* kargs[] = {(void*)&r.wn};
        return launch(r.lp.fn, r.lp.grid, r.lp.block, kargs, r.lp.lds_bytes, st);
    }
    if (r.kind == K_A8_MMA) {
        if (r.lp.ws_bytes > 0) {
            if (!args->workspace || args->workspace_bytes < r.lp.ws_bytes) return GEMLITE_ERR_WORKSPACE;
            r.gp.counters = (unsigned*)args->workspace;
            r.gp.slabs = (float*)((char*)args->workspace + COUNTER_BYTES);
        }
        const int e = ensure_lds(r.lp.fn, r.lp.lds_bytes, dev);
        if (e != GEMLITE_OK) return e;
        void* kargs[] = {(void*)&r.gp};
        return launch(r.lp.fn, r.lp.grid, r.lp.block, kargs, r.lp.lds_bytes, st);
    }
    if (r.kind == K_NV_MMA) {  // NVFP4: [tickets | split-K slabs | fp16 expansion of x | per-row output factor]
        if (!args->workspace || args->workspace_bytes < r.lp.ws_bytes) return GEMLITE_ERR_WORKSPACE;
        char* base = (char*)args->workspace;
        uint16_t* x16 = (uint16_t*)(base + COUNTER_BYTES + r.lp.slab_bytes);
        float* post = (float*)((char*)x16 + r.nv_x16_bytes);
        r.wn.counters = (unsigned*)base;
        r.wn.slabs = (float*)(base + COUNTER_BYTES);
        r.wn.x = x16;
        r.wn.epi.scales_x = post;
        const uint8_t* xq = (const uint8_t*)args->x;
        const uint8_t* sx = (const uint8_t*)args->scales_x;
        int M = (int)args->M, K = (int)args->K;
        int64_t sxm = args->stride_xm, ssm = args->stride_sx_m;
        float post_v = r.gp.mx_post;
        void* eargs[] = {(void*)&xq, (void*)&sx, (void*)&x16, (void*)&post, (void*)&M, (void*)&K, (void*)&sxm, (void*)&ssm, (void*)&post_v};
        const int64_t nblk = ((int64_t)M * (K / 16) + 255) / 256;
        void* es = tl_evt_start; void* ee = tl_evt_stop;  // (profile events, if any, bracket BOTH launches)
        tl_evt_stop = nullptr;
        int rc = launch(nvfp4_expand_f16_kernel_fn(), dim3((unsigned)nblk, 1, 1), dim3(256, 1, 1), eargs, 0, st);
        if (rc != GEMLITE_OK) return rc;
        (void)es;
        tl_evt_start = nullptr; tl_evt_stop = ee;
        const int e = ensure_lds(r.lp.fn, r.lp.lds_bytes, dev);
        if (e != GEMLITE_OK) return e;
        void* kargs[] = {(void*)&r.wn};
        return launch(r.lp.fn, r.lp.grid, r.lp.block, kargs, r.lp.lds_bytes, st);
    }
    if (r.kind == K_A8_FQ) {  // cooperative activation quantisation: the workspace holds the flags, the quantised rows and their scales
        if (!args->workspace || args->workspace_bytes < r.lp.ws_bytes) return GEMLITE_ERR_WORKSPACE;
        char* payload = (char*)args->workspace + COUNTER_BYTES;
        r.gp.counters = (unsigned*)args->workspace;
        r.gp.x = payload;
        r.gp.epi.scales_x = (const float*)(payload + cq::xq_bytes(args->M, args->K));
        const int e = ensure_lds(r.lp.fn, r.lp.lds_bytes, dev);
        if (e != GEMLITE_OK) return e;
        void* kargs[] = {(void*)&r.gp};
        return launch(r.lp.fn, r.lp.grid, r.lp.block, kargs, r.lp.lds_bytes, st);
    }
    void* kargs[] = {(void*)&r.gp};
    {  // (round 6: w8_rows_lds_kernel stages x in up to 128 KB)
        const int e = ensure_lds(r.lp.fn, r.lp.lds_bytes, dev);
        if (e != GEMLITE_OK) return e;
    }
    return launch(r.lp.fn, r.lp.grid, r.lp.block, kargs, r.lp.lds_bytes, st);
}

int gemlite_hip_scale_activations_per_token(const void* x, void* y, float* scales, int64_t M, int64_t K,
                                            int64_t stride_xm, int32_t in_dtype, int32_t out_dtype, void* stream) {
    if (!x || !y || !scales || M <= 0 || K <= 0) return GEMLITE_ERR_BAD_ARGUMENT;
    if (!(in_dtype == GEMLITE_DT_FP16 || in_dtype == GEMLITE_DT_BF16 || in_dtype == GEMLITE_DT_FP32)) return GEMLITE_ERR_UNSUPPORTED;
    if (!(out_dtype == GEMLITE_DT_INT8 || out_dtype == GEMLITE_DT_FP8E4 || out_dtype == GEMLITE_DT_FP8E5)) return GEMLITE_ERR_UNSUPPORTED;
    if (M > 0x7FFFFFFF) return GEMLITE_ERR_BAD_SHAPE;
    int in_dt = in_dtype, out_dt = out_dtype;
    if (const void* vec = act_quant_vec_kernel_fn(in_dt, out_dt, K, stride_xm, x, y)) {  // row in registers: one memory round trip
        int k32 = (int)K;
        void* vargs[] = {(void*)&x, (void*)&y, (void*)&scales, (void*)&k32, (void*)&stride_xm};
        return launch(vec, dim3((unsigned)M, 1, 1), dim3(256, 1, 1), vargs, 0, (hipStream_t)stream);
    }
    void* kargs[] = {(void*)&x, (void*)&y, (void*)&scales, (void*)&K, (void*)&stride_xm, (void*)&in_dt, (void*)&out_dt};
    return launch(act_quant_kernel_fn(), dim3((unsigned)M, 1, 1), dim3(256, 1, 1), kargs, 0, (hipStream_t)stream);
}

static int scale_activations_mx(int mode, const void* x, void* y, uint8_t* scales, int64_t M, int64_t K, int64_t stride_xm,
                                int32_t in_dtype, void* stream) {
    const int64_t G = mode == 2 ? 16 : 32;
    if (!x || !y || !scales || M <= 0 || K <= 0) return GEMLITE_ERR_BAD_ARGUMENT;
    if (!(in_dtype == GEMLITE_DT_FP16 || in_dtype == GEMLITE_DT_BF16 || in_dtype == GEMLITE_DT_FP32)) return GEMLITE_ERR_UNSUPPORTED;
    if (K % 32 != 0) return GEMLITE_ERR_BAD_SHAPE;  // 16-byte output pieces; the formats' own block is 32 (16) k
    if (((uintptr_t)y % 16) != 0) return GEMLITE_ERR_BAD_ARGUMENT;
    int64_t m_pad = (M + G - 1) / G * G;
    const int64_t total = m_pad * (K / G);
    if (total > 0x7FFFFFFFll * 256) return GEMLITE_ERR_BAD_SHAPE;
    int in_dt = in_dtype;
    void* kargs[] = {(void*)&x, (void*)&y, (void*)&scales, (void*)&M, (void*)&m_pad, (void*)&K, (void*)&stride_xm, (void*)&in_dt};
    return launch(act_quant_mx_kernel_fn(mode), dim3((unsigned)((total + 255) / 256), 1, 1), dim3(256, 1, 1), kargs, 0, (hipStream_t)stream);
}

int gemlite_hip_scale_activations_mxfp8(const void* x, void* y, uint8_t* scales, int64_t M, int64_t K, int64_t stride_xm,
                                        int32_t in_dtype, void* stream) {
    return scale_activations_mx(0, x, y, scales, M, K, stride_xm, in_dtype, stream);
}
int gemlite_hip_scale_activations_mxfp4(const void* x, uint8_t* y, uint8_t* scales, int64_t M, int64_t K, int64_t stride_xm,
                                        int32_t in_dtype, void* stream) {
    return scale_activations_mx(1, x, y, scales, M, K, stride_xm, in_dtype, stream);
}
int gemlite_hip_scale_activations_nvfp4(const void* x, uint8_t* y, uint8_t* scales, int64_t M, int64_t K, int64_t stride_xm,
                                        int32_t in_dtype, void* stream) {
    return scale_activations_mx(2, x, y, scales, M, K, stride_xm, in_dtype, stream);
}

static int check_pack(int64_t N, int64_t K, int32_t nbits, int32_t pack_bits) {
    if (N <= 0 || K <= 0) return GEMLITE_ERR_BAD_ARGUMENT;
    if (!(nbits == 1 || nbits == 2 || nbits == 4 || nbits == 8)) return GEMLITE_ERR_UNSUPPORTED;
    if (!(pack_bits == 8 || pack_bits == 16 || pack_bits == 32 || pack_bits == 64) || pack_bits < nbits) return GEMLITE_ERR_UNSUPPORTED;
    if (K % (pack_bits / nbits) != 0) return GEMLITE_ERR_BAD_SHAPE;
    if ((K / (pack_bits / nbits)) * N > 0x7FFFFFFFll * 256) return GEMLITE_ERR_BAD_SHAPE;
    return GEMLITE_OK;
}

int gemlite_hip_pack_over_cols(const uint8_t* w_q, void* out, int64_t N, int64_t K, int64_t ld_in, int32_t W_nbits,
                               int32_t pack_bits, void* stream) {
    if (!w_q || !out) return GEMLITE_ERR_BAD_ARGUMENT;
    const int c = check_pack(N, K, W_nbits, pack_bits);
    if (c != GEMLITE_OK) return c;
    const int64_t total = (K / (pack_bits / W_nbits)) * N;
    int nb = W_nbits, pb = pack_bits;
    if (pack_bits == 32 && K % (32 / W_nbits) == 0 && (N + 63) / 64 <= 0x7FFFFFFF && (K + 255) / 256 <= 65535) {  // tiled through LDS: coalesced both ways
        void* targs[] = {(void*)&w_q, (void*)&out, (void*)&N, (void*)&K, (void*)&ld_in, (void*)&nb};
        return launch(pack32_kernel_fn(), dim3((unsigned)((N + 63) / 64), (unsigned)((K + 255) / 256), 1), dim3(256, 1, 1), targs, 0, (hipStream_t)stream);
    }
    void* kargs[] = {(void*)&w_q, (void*)&out, (void*)&N, (void*)&K, (void*)&ld_in, (void*)&nb, (void*)&pb};
    return launch(pack_kernel_fn(), dim3((unsigned)((total + 255) / 256), 1, 1), dim3(256, 1, 1), kargs, 0, (hipStream_t)stream);
}

int gemlite_hip_unpack_over_cols(const void* packed, uint8_t* out, int64_t N, int64_t K, int32_t W_nbits,
                                 int32_t pack_bits, void* stream) {
    if (!packed || !out) return GEMLITE_ERR_BAD_ARGUMENT;
    const int c = check_pack(N, K, W_nbits, pack_bits);
    if (c != GEMLITE_OK) return c;
    const int64_t total = (K / (pack_bits / W_nbits)) * N;
    int nb = W_nbits, pb = pack_bits;
    void* kargs[] = {(void*)&packed, (void*)&out, (void*)&N, (void*)&K, (void*)&nb, (void*)&pb};
    return launch(unpack_kernel_fn(), dim3((unsigned)((total + 255) / 256), 1, 1), dim3(256, 1, 1), kargs, 0, (hipStream_t)stream);
}

}  // extern "C"
