/*
 * gemlite_hip.h — C ABI of libgemlite_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for the fused unpack + dequant + matmul hot path of
 * mobiusml/gemlite.  Each entry point replaces one seam of the reference (paths relative
 * to the reference tree):
 *
 *   gemlite_hip_forward            <- GEMLITE_TRITON_MAPPING[matmul_type].forward(...)
 *                                     gemlite/core.py:184-190; the identical 17-argument
 *                                     launchers gemv_kernels.py:553-638,
 *                                     gemv_revsplitK_kernels.py:391-462,
 *                                     gemv_splitK_kernels.py:423-476,
 *                                     gemm_splitK_kernels.py:595-652, gemm_kernels.py:549-601
 *   gemlite_hip_workspace_bytes    <- the reference has no workspace: it zero-fills the
 *                                     output and uses atomics (gemv_revsplitK_kernels.py:422,
 *                                     gemm_splitK_kernels.py:141); we use caller-owned scratch
 *   gemlite_hip_scale_activations_per_token
 *                                  <- scale_activations_per_token_triton,
 *                                     gemlite/quant_utils.py:268-347 (spec: :231-253)
 *   gemlite_hip_scale_activations_mxfp8 / _mxfp4 / _nvfp4
 *                                  <- scale_activations_mxfp8 / mxfp4 / nvfp4 (= the *_triton_v2 launchers),
 *                                     gemlite/quant_utils.py:546-590, 820-855, 917-954
 *   gemlite_hip_pack_over_cols     <- pack_weights_over_cols_triton, gemlite/bitpack.py:77-144
 *                                     (bit layout spec: pack_weights_over_cols_torch :36-60)
 *   gemlite_hip_unpack_over_cols   <- unpack_over_cols_triton, gemlite/bitpack.py:175-241
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*); the library
 *     never allocates, frees or synchronises, so every call is hipGraph-capturable;
 *   - return value: 0 on success, a negative gemlite_status_t otherwise;
 *     gemlite_hip_status_string() maps it to text.  Nothing is launched on error;
 *   - dtype arguments are the integer codes of gemlite/dtypes.py:8-29 (gemlite_dtype_t);
 *   - strides are in ELEMENTS of the tensor's own dtype (what torch's .stride() returns).
 */
#ifndef GEMLITE_HIP_H
#define GEMLITE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEMLITE_HIP_ABI_VERSION 1

/* integer dtype codes — wire format shared with gemlite/dtypes.py:8-29 */
typedef enum gemlite_dtype_t {
    GEMLITE_DT_FP32 = 0,
    GEMLITE_DT_FP16 = 1,
    GEMLITE_DT_BF16 = 2,
    GEMLITE_DT_FP8E4 = 3, /* OCP e4m3fn: the gfx950 MFMA flavour */
    GEMLITE_DT_INT8 = 4,
    GEMLITE_DT_UINT8 = 5,
    GEMLITE_DT_INT32 = 6,
    GEMLITE_DT_UINT32 = 7,
    GEMLITE_DT_FP8E5 = 8, /* OCP e5m2 */
    GEMLITE_DT_INT16 = 9,
    GEMLITE_DT_UINT16 = 10,
    GEMLITE_DT_INT64 = 11,
    GEMLITE_DT_FP8E4NUZ = 12, /* MI300X flavour: rejected on gfx950 */
    GEMLITE_DT_FP8E5NUZ = 13,
    /* block-scaled ("microscaling") input formats, dtypes.py:24-29.  As input_dtype they name the FORMAT PAIR of a layer:
     * what x holds and how w_q / scales are laid out (see "Block-scaled formats" below) */
    GEMLITE_DT_MXFP16 = 14, /* x fp16,             w fp8 / fp4 + e8m0 scale per 32 k            */
    GEMLITE_DT_MXBF16 = 15, /* x bf16,             w fp8 / fp4 + e8m0 scale per 32 k            */
    GEMLITE_DT_MXFP8 = 16,  /* x fp8 e4m3,         w fp8 / fp4 + e8m0 scale per 32 k            */
    GEMLITE_DT_MXFP4 = 17,  /* x fp4 e2m1 (2/byte), w fp4 + e8m0 scale per 32 k                  */
    GEMLITE_DT_NVFP4 = 18,  /* x fp4 e2m1 (2/byte), w fp4 + e4m3 scale per 16 k, result * 0.05^2 */
    GEMLITE_DT_E8M0 = 19
} gemlite_dtype_t;

/* index into GEMLITE_MATMUL_TYPES (gemlite/core.py:56-66) */
typedef enum gemlite_matmul_type_t {
    GEMLITE_MATMUL_AUTO = -1, /* pick by M like get_matmul_type(), core.py:100-114 */
    GEMLITE_MATMUL_GEMV = 0,
    GEMLITE_MATMUL_GEMV_REVSPLITK = 1,
    GEMLITE_MATMUL_GEMV_SPLITK = 2,
    GEMLITE_MATMUL_GEMM_SPLITK = 3,
    GEMLITE_MATMUL_GEMM = 4
} gemlite_matmul_type_t;

typedef enum gemlite_status_t {
    GEMLITE_OK = 0,
    GEMLITE_ERR_BAD_ARGUMENT = -1, /* null pointer, non-positive size, bad struct_size  */
    GEMLITE_ERR_UNSUPPORTED = -2,  /* dtype / bit-width / mode combination not built     */
    GEMLITE_ERR_BAD_SHAPE = -3,    /* K not a multiple of elements_per_sample / group    */
    GEMLITE_ERR_WORKSPACE = -4,    /* workspace missing or smaller than required         */
    GEMLITE_ERR_LAUNCH = -5,       /* hipLaunchKernel failed; see gemlite_hip_last_hip_error */
    GEMLITE_ERR_NO_DEVICE = -6,    /* current device is not gfx950                       */
    GEMLITE_ERR_NO_FUSED_QUANT = -7 /* scales_x == NULL with 16-bit x asked for the in-launch activation quantisation, which this
                                    * shape / M has no kernel for: quantise x (gemlite_hip_scale_activations_per_token) and call
                                    * again with scales_x.  Not an error of the layer. */
} gemlite_status_t;

/* W_group_mode / channel_scale_mode follow gemlite/triton_kernels/utils.py:73-87 and
 * gemm_kernels.py:392-404:
 *   W_group_mode        0 none | 1 (q - z) | 2 q*s | 3 (q - z)*s | 4 fma(q, s, z')
 *   channel_scale_mode  0 none | 1 acc*s_w[n] | 2 acc*s_x[m] | 3 acc*s_x[m]*s_w[n]
 *                       4 block scales on the activations (MX / NV input formats only, gemm_kernels.py:508-509)
 *
 * Block-scaled formats (input_dtype GEMLITE_DT_MXFP16 .. GEMLITE_DT_NVFP4; gemm_MX_kernel, gemm_kernels.py:422-547;
 * layout produced by pack(), core.py:363-398, 489-497):
 *   w_q     W_nbits 8: fp8 e4m3 [K, N] view of an [N, K] tensor (elements_per_sample 1, w_dtype FP8E4);
 *           W_nbits 4: e2m1 codes, two per byte, k even in the low nibble: uint8 [K/2, N] (elements_per_sample 2,
 *           w_pack_bits 8).  Either way stride_wk / stride_wn are torch's strides of that tensor; the MFMA kernels take
 *           the K-contiguous layout pack() produces (stride_wk == 1), anything else runs on the coverage kernel
 *   scales  one byte per (32-k block, n): e8m0 (value 2^(b-127)); NVFP4: e4m3 per 16-k block; strides stride_meta_g
 *           (between blocks) / stride_meta_n;  W_group_mode is ignored, zeros unused
 *   x       MXFP16 / MXBF16: 16-bit floats, channel_scale_mode 0;  MXFP8: fp8 e4m3;  MXFP4 / NVFP4: uint8 [M, K/2]
 *   scales_x  channel_scale_mode 4: the block scales of x, uint8 [M_pad, K/32] (NVFP4: e4m3 [M_pad, K/16]), rows
 *           contiguous, row stride stride_sx_m, M_pad = M rounded up to 32 (16) — what
 *           gemlite_hip_scale_activations_mxfp8 / _mxfp4 / _nvfp4 write;  channel_scale_mode 2: fp32 [M] per token     */
typedef struct gemlite_hip_forward_args {
    uint32_t struct_size; /* = sizeof(gemlite_hip_forward_args), ABI guard */
    int32_t matmul_type;  /* gemlite_matmul_type_t */

    const void* x;        /* [M, K] activations, dtype input_dtype, row stride stride_xm   */
    const void* w_q;      /* packed: [K/e, N] words of w_pack_bits; unpacked: [K, N] view  */
    const void* scales;   /* [K/group, N] meta_dtype, or [N] channel scales, or NULL       */
    const void* zeros;    /* [K/group, N] meta_dtype, 1-elem int32 (scalar), or NULL       */
    const void* scales_x; /* [M] fp32 per-token activation scales, or NULL.  NULL together with
                           * channel_scale_mode 2/3, 16-bit float x and unpacked int8 / fp8 weights asks for
                           * the FUSED dynamic quantisation: x is quantised per token inside the matmul launch
                           * (same arithmetic as gemlite_hip_scale_activations_per_token, one launch instead of two).
                           * M == 1: always available.  2 <= M <= 64: producer blocks of the launch quantise the rows
                           * (needs the workspace: flags + M*K bytes + M floats; measured slower than two launches, the
                           * Python host does not use it by default); GEMLITE_ERR_NO_FUSED_QUANT where no such kernel
                           * applies (gemlite_hip_query() answers without launching).
                           * The same request exists at M == 1 for the layers whose activation format cannot be read off
                           * the weights — PACKED weights under fp8 / int8 activations (A8Wn dynamic, BitNet) and the
                           * block-scaled MXFP8 / MXFP4 layers (channel_scale_mode 4: block scales; 2: one scale per
                           * token): input_dtype names the type of the 16-bit x that is passed, the layer's activation
                           * format rides in type_id (= its DType code * 100 + W_nbits), scales_x is NULL. */
    void* out;            /* [M, N] output_dtype                                           */
    void* workspace;      /* >= gemlite_hip_workspace_bytes(); zero-filled ONCE by the owner */
    uint64_t workspace_bytes;

    int64_t M, N, K;

    int32_t W_nbits;             /* 1,2,4,8 packed; 8/16/32 unpacked                        */
    int32_t group_size;          /* K elements per scale/zero row (1 when there is none)    */
    int32_t unpack_mask;         /* 2^W_nbits - 1 (carried for signature parity; rederived) */
    int32_t elements_per_sample; /* e = w_pack_bits / W_nbits; 1 for unpacked weights       */
    int32_t w_pack_bits;         /* 8/16/32 for packed words; 0 when elements_per_sample==1 */
    int32_t w_dtype;             /* gemlite_dtype_t of w_q when unpacked (int8/fp8/fp16...) */

    int32_t input_dtype;  /* dtype of x as the kernel sees it (after activation quant)     */
    int32_t output_dtype;
    int32_t acc_dtype;    /* carried for parity; the HIP kernels accumulate fp32 / int32   */
    int32_t meta_dtype;   /* dtype of scales / tensor zeros                                */
    int32_t zeros_dtype;  /* meta_dtype for tensor zeros, GEMLITE_DT_INT32 for a scalar    */

    int32_t channel_scale_mode;
    int32_t W_group_mode;
    int32_t zero_is_scalar; /* zeros holds exactly one element (gemm_kernels.py:597)       */
    int32_t data_contiguous;
    int32_t type_id;        /* input_dtype*100 + W_nbits (core.py:141-145); tuning key only */

    int64_t stride_xm, stride_xk;
    int64_t stride_wk, stride_wn;       /* of w_q as given (packed rows or K)               */
    int64_t stride_om, stride_on;
    int64_t stride_meta_g, stride_meta_n;
    int64_t stride_sx_m;

    /* Planner overrides — the named values are the enums GEMLITE_T0_* / GEMLITE_T2_* / GEMLITE_TF_* below this struct (round 6, VERDICT r5 #9:
     * the numbers of rounds 1-5 stay valid, the names are what new callers should write).
     * (0 = library default everywhere; what helper.autotune_layer() searches and the tuning table
     * stores — the counterpart of the reference's per-shape Triton autotune configs).  Meaning per kernel family:
     *   packed GEMV (M = 1)        [0] 2/3/4 = 16-/32-/64-column tiles   [1] K slices   [2] 4/8/16 waves per block
     *                              (82 = 8 waves, 2 rows per lane; 48 = 4 waves, 8 rows per lane)   [3] & 3: 1 = x through LDS,
     *                              2 = x direct; & 16 = the round-2 kernel instead of the 16-column decode kernel,
     *                              & 32 = default-policy (not non-temporal) weight loads in the decode kernel,
     *                              & 4096 = the round-3 decode kernel (struct arguments) instead of gemv_w4_decode3_kernel
     *   MFMA GEMV (M = 1..4, 4- and 2-bit words under 16-bit activations; default where it measured faster)
     *                              [0] 21/22/24 = 16-/32-/64-column tiles   [2] 4/8/16 waves per block
     *                              [3] & 512 = never, & 1024 = wherever it applies
     *   few rows (2..32, MFMA)     [0] 1/2/4 = 16-/32-/64-column tiles, 3 = the 8-wave tiled kernel instead
     *                              [1] K slices   [2] 1 = LDS-staged streaming kernel, 4 / 8 = waves per block of the registers-only
     *                              kernel (8: one row tile, >= 32-column tiles; default with two K slices)
     *   decode batch (2..64 rows, 4- and 2-bit words under 16-bit activations; round 5: gemm_w{4,2}_rows_kernel, 16-column blocks, K unsplit,
     *                              raw integer codes through the MFMA)   [0] 9 = at any M >= 2 the kernel takes (row blocks along
     *                              grid.y above 16 MT rows; [2] must be 0 or 8; [1] = 1 / 2 column tiles per block, 4-bit); default for 2..64 rows where N / 16 blocks are
     *                              resident in one round and the x re-reads stay <= 176 MiB, and ALWAYS for group size 32 and
     *                              N % 64 != 0 (no other MFMA kernel takes them)   [3] & 65536 = never (the round-4 choice, A/B runs)
     *   tiled (M > 32, MFMA)       [0] 1 = streaming kernel, 2 = the 4-wave tiled kernel of round 1 (4-bit only; the planner's
     *                              fallback for K = 64 * odd)
     *                              [1] K slices (any count <= K steps; slices may be uneven)
     *                              [2] tile rows / 32: 1/2/4/8 (8-wave kernel, 128-column tiles); 20 / 24 = 128 / 256 rows x 256
     *                              columns (4- and 2-bit, 16-bit activations); 32 .. 35 = the narrow 64-COLUMN tiles of round 4 (32 also
     *                              for 8-bit activations, block-scaled / NVFP4 / K-contiguous 8-bit weights)
     *                              (32: 64 x 64, 256-k steps — the default where they fill the chip without K slices, e.g.
     *                              4096^2 at M = 256; 33: 64 x 64, 512-k steps; 34 / 35: 128 x 64); with [0] = 2: 4 = one-step-ahead,
     *                              8 = 256 rows (both only in a library built with `make AB=1`)
     *                              [3] K-slice combine: & 128 = slabs + ticket always, & 2048 = reduce-scatter with 2 slices too
     *                              (default: from 4 slices), & 256 = reduce-scatter with immediate hand-over (test switch);
     *                              & 16384 = never the narrow tiles (the round-3 choice, A/B runs)
     *   unpacked 8-bit (A8W8)      [0] 1 = streaming (one wave per column), 2 = the 4-wave MFMA kernel of round 1 (the planner's
     *                              fallback for K % 256 != 0), 4 = the 16-column few-row kernel (default for 2..64 rows while
     *                              M K N / 16 <= 88 MiB and, from 128 column tiles of 64, M N K <= 800 M) at any M <= 64 and at M = 1, 5 = the unsplit 64 x 64 tiles of round 4
     *                              (default from 65 rows where they fill the chip once or twice, and from 2 rows past the few-row budget; [2] = 2/3/4 LDS stages),
     *                              6 = the round-3 kernels instead; M = 1: 7 = the round-2 streaming kernels instead of
     *                              a8w8_decode_kernel, 8 = a8w8_decode_kernel also for fp8 with N > 4096;
     *                              10 = the unsplit 128 x 128 tiles of round 5 (default above 64 rows where they number 192 .. 256 and
     *                              K >= 4096; [2] = 4 / 5 LDS stages of 128-byte K steps)
     *                              [1] K slices   [2] tile rows / 32 (forces the 8-wave kernels at any M)
     *                              [3] & 64: 128- / 256-row tiles with the weights straight from memory (default: through LDS)
     *                              [3] & 32768: test switch of the in-launch activation quantisation (no producer block runs)
     *   block-scaled (MX / NVFP4)  [0] 1 = coverage kernel, 2 = the 8-wave scaled-MFMA tile kernels at any M, 3 = the 256 x 256 tile
     *                              kernel at any M, 4 = the few-row kernel (default for 1..64 rows of fp8 / fp4 activations) past
     *                              its x re-read budget, 5 = the streaming kernel of rounds 2-3 (M <= 4), 6 = the unsplit 64 x 64
     *                              tiles of round 4 at any M (default for 65..384 rows, to 512 for fp4 x fp4 and one-round
     *                              shapes; [2] = 2/3/4 LDS stages)
     *                              [1] K slices   [2] tile rows / 32   (NVFP4: [0] = 1 coverage kernel, else the fp16 tile kernel)
     *   8-bit x packed (A8Wn, BitNet int8)   [0] 4 = the 16-column few-row kernel also at one row (default for 2..64 rows),
     *                              7 = the streaming GEMV of rounds 2-3 up to 4 rows
     *   unpacked 8-bit under 16-bit x (A16W8)   [0] 4 = the 16-column rows kernel at any M (default for 2..64 rows; above: the 8-wave
     *                              tile kernel, [1] K slices, [2] tile rows / 32), 7 = the streaming kernel of rounds 1-3
     *   tiled, round 6             [3] & 131072 = the packed words of the 64 x 64 / 128 x 128 4-bit tiles as register loads (the round-5 path; default:
     *                              through LDS-DMA, one request per wave and step); groups of 32 run on 32-row tiles with two metadata pairs per
     *                              sub-block ("<32x128,g32>": [1] K slices, [0] and [2] must be 0)
     *   [3] & 4: development timeline stamps (needs a workspace)   [3] & 8: XCD-aware (tile, K slice) map (opt-in).
     *   [3] >> 20: K-loop ablation of development builds (make MMA_EXTRA=-DGL_MMA_EXPERIMENTS); ignored by the shipped library.
     *   A value that does not apply to the shape makes the planner fall through to its own choice or to another family;
     *   it never produces a wrong result. */
    int32_t tuning[4];
} gemlite_hip_forward_args;

/* ---- names for the tuning[] values (per kernel family; the comment inside the struct has what each one selects) ------------------------- */
enum gemlite_hip_tuning0 {                 /* tuning[0]: which kernel of the family */
    GEMLITE_T0_AUTO = 0,
    /* packed words, one row (dot-product GEMV family): tile width */
    GEMLITE_T0_GEMV_TILE16 = 2, GEMLITE_T0_GEMV_TILE32 = 3, GEMLITE_T0_GEMV_TILE64 = 4,
    /* packed words, matrix-core GEMV (1 .. 4 rows): tile width */
    GEMLITE_T0_MFMA_GEMV_TILE16 = 21, GEMLITE_T0_MFMA_GEMV_TILE32 = 22, GEMLITE_T0_MFMA_GEMV_TILE64 = 24,
    /* packed words, 2 .. 32 rows (registers-only MFMA kernel): tile width; 3 = the 8-wave tile kernel instead */
    GEMLITE_T0_FEWROWS_TILE16 = 1, GEMLITE_T0_FEWROWS_TILE32 = 2, GEMLITE_T0_FEWROWS_TILE64 = 4, GEMLITE_T0_FEWROWS_USE_TILES = 3,
    /* packed 4- / 2-bit words, 2 .. 64 rows: the decode-shaped rows kernel at any M it takes */
    GEMLITE_T0_ROWS_KERNEL = 9,
    /* packed words, many rows: 1 = LDS-staged streaming kernel, 2 = the 4-wave tile kernel of round 1 */
    GEMLITE_T0_TILED_STREAM = 1, GEMLITE_T0_TILED_ROUND1 = 2,
    /* unpacked 8-bit (A8W8) */
    GEMLITE_T0_A8W8_STREAM = 1, GEMLITE_T0_A8W8_ROUND1 = 2, GEMLITE_T0_A8W8_ROWS = 4, GEMLITE_T0_A8W8_SQ64 = 5, GEMLITE_T0_A8W8_ROUND3 = 6,
    GEMLITE_T0_A8W8_M1_ROUND2 = 7, GEMLITE_T0_A8W8_M1_DECODE_FP8_WIDE = 8, GEMLITE_T0_A8W8_SQ128 = 10,
    /* block-scaled (MX / NVFP4) */
    GEMLITE_T0_MX_COVERAGE = 1, GEMLITE_T0_MX_TILES = 2, GEMLITE_T0_MX_TILES_256 = 3, GEMLITE_T0_MX_ROWS = 4, GEMLITE_T0_MX_STREAM = 5, GEMLITE_T0_MX_SQ64 = 6,
    /* 8-bit activations x packed words (A8Wn, BitNet int8); unpacked 8-bit weights under 16-bit activations (A16W8) */
    GEMLITE_T0_A8WN_ROWS = 4, GEMLITE_T0_A8WN_GEMV = 7, GEMLITE_T0_A16W8_ROWS = 4, GEMLITE_T0_A16W8_STREAM = 7
};
enum gemlite_hip_tuning2 {                 /* tuning[2]: tile geometry of the 8-wave tile kernel (other families: waves per block, see the struct) */
    GEMLITE_T2_AUTO = 0,
    GEMLITE_T2_ROWS32 = 1, GEMLITE_T2_ROWS64 = 2, GEMLITE_T2_ROWS128 = 4, GEMLITE_T2_ROWS256 = 8,        /* x 128 columns */
    GEMLITE_T2_WIDE_128x256 = 20, GEMLITE_T2_WIDE_256x256 = 24,                                              /* 256-column tiles */
    GEMLITE_T2_NARROW_64x64 = 32, GEMLITE_T2_NARROW_64x64_K512 = 33, GEMLITE_T2_NARROW_128x64 = 34, GEMLITE_T2_NARROW_128x64_RING6 = 35,
    GEMLITE_T2_GEMV_8WAVES_2ROWS = 82, GEMLITE_T2_GEMV_4WAVES_8ROWS = 48
};
enum gemlite_hip_tuning_flags {            /* tuning[3]: bit flags (A/B switches and test hooks; 0 = the shipped defaults) */
    GEMLITE_TF_GEMV_X_THROUGH_LDS = 1, GEMLITE_TF_GEMV_X_DIRECT = 2,        /* (& 3) */
    GEMLITE_TF_TIMELINE = 4,                     /* development stamps (needs a workspace) */
    GEMLITE_TF_XCD_SLICE_MAP = 8,                /* XCD-aware (tile, K slice) map of the tile kernel */
    GEMLITE_TF_GEMV_ROUND2_KERNEL = 16, GEMLITE_TF_GEMV_DEFAULT_POLICY_LOADS = 32,
    GEMLITE_TF_A8W8_WEIGHTS_FROM_MEMORY = 64,
    GEMLITE_TF_COMBINE_TICKET = 128, GEMLITE_TF_COMBINE_HANDOVER_TEST = 256,
    GEMLITE_TF_NO_MFMA_GEMV = 512, GEMLITE_TF_FORCE_MFMA_GEMV = 1024,
    GEMLITE_TF_COMBINE_REDUCE_SCATTER_2 = 2048,
    GEMLITE_TF_GEMV_ROUND3_DECODE = 4096,
    GEMLITE_TF_NO_NARROW_TILES = 16384,
    GEMLITE_TF_QUANT_NO_PRODUCER_TEST = 32768,
    GEMLITE_TF_NO_ROWS_KERNEL = 65536,           /* the round-4 choice for 2 .. 64 rows */
    GEMLITE_TF_WORDS_AS_REGISTER_LOADS = 131072, /* round 6: the round-5 weight path of the 64 x 64 / 128 x 128 4-bit tiles */
    GEMLITE_TF_ROUND5_TILE_EPILOGUE = 262144,    /* round 6: the two-barrier K-part join of the unsplit tiles instead of the direct one */
    GEMLITE_TF_W8_ROWS_X_FROM_REGISTERS = 524288, /* round 6: the round-3 / round-4 few-row kernels of unpacked 8-bit weights (a8w8_rows_kernel / a16w8_rows_kernel)
                                                    instead of w8_rows_lds_kernel (x through LDS in whole cache lines) */
    GEMLITE_TF_W8_ROWS_LDS_BELOW_4_ROWS = 1048576, /* A16W8: take w8_rows_lds_kernel at 1 .. 3 rows too (default: from 4) */
    GEMLITE_TF_A8W8_TILE_REQUESTS_FIRST = 2097152, /* round 6: the round-4 order of a K step of the 64 x 64 A8W8 tile (DMA requests in front of the LDS reads) */
    GEMLITE_TF_NO_K_ROTATION = 4194304,            /* round 6: every row tile of the unsplit tiles of 8-bit weights walks K in the plain order */
    GEMLITE_TF_K_ORDER_GROUP_MASK = 0x0F000000     /* round 6: bits 24 .. 27 = 1 + log2(steps per group) of the grouped K order between the row tiles of a column tile (0 = the planner's choice) */
};

/* Library / ABI identification (host only, no device access). */
int gemlite_hip_abi_version(void);
const char* gemlite_hip_build_info(void);
const char* gemlite_hip_status_string(int status);
/* hipError_t of the last failed HIP runtime call made by this library on this thread. */
int gemlite_hip_last_hip_error(void);

/* 0 if the configuration has a native kernel, GEMLITE_ERR_* otherwise. Never launches. */
int gemlite_hip_query(const gemlite_hip_forward_args* args);

/* Scratch bytes gemlite_hip_forward needs for this configuration (split-K slabs + arrival
 * counters).  The caller allocates it once, zero-fills it once, and then only passes it in:
 * every launch leaves the counters at zero again.  One workspace per concurrently used stream. */
uint64_t gemlite_hip_workspace_bytes(const gemlite_hip_forward_args* args);

/* out[M,N] = epilogue( x[M,K] @ dequant(w_q, scales, zeros) ) — one fused launch. */
int gemlite_hip_forward(const gemlite_hip_forward_args* args, void* stream);

/* Name of the kernel gemlite_hip_forward would launch for `args` (for profiles/tests). */
const char* gemlite_hip_kernel_name(const gemlite_hip_forward_args* args);

/* Optional: the NEXT gemlite_hip_forward on this thread is launched with
 * hipExtLaunchKernel(start, stop) so that hipEventElapsedTime(start, stop) is that kernel's
 * own device duration.  Pass NULL, NULL to clear.  Events are hipEvent_t passed as void*. */
void gemlite_hip_set_profile_events(void* start_event, void* stop_event);

/* Measurement aid: launches an EMPTY kernel of `blocks` x `threads` on `stream` (honours
 * gemlite_hip_set_profile_events).  bench.py uses it to report the floor of the per-launch event clock next
 * to every kernel duration it quotes (on MI355X an empty kernel already reads ~4 us). */
int gemlite_hip_launch_noop(int32_t blocks, int32_t threads, void* stream);

/* Per-token dynamic activation quantisation: for each row m of x[M,K] (fp16/bf16/fp32)
 *   s[m] = max(amax(|x[m,:]|) / qmax, 1e-6) (fp32);  y = clamp(x / s, qmin, qmax);
 *   int8: round half away from zero;  fp8: round-to-nearest-even cast.
 * out_dtype: GEMLITE_DT_INT8, GEMLITE_DT_FP8E4 or GEMLITE_DT_FP8E5. */
int gemlite_hip_scale_activations_per_token(const void* x, void* y, float* scales, int64_t M,
                                            int64_t K, int64_t stride_xm, int32_t in_dtype,
                                            int32_t out_dtype, void* stream);

/* Block-scaled activation quantisers (scale_activations_mxfp8 / mxfp4 / nvfp4_triton_v2, gemlite/quant_utils.py:
 * 502-590, 769-855, 859-954).  x [M, K] fp16 / bf16 / fp32, row stride stride_xm, K % 32 == 0 (nvfp4: % 16).
 *   mxfp8: y fp8 e4m3 [M, K];       scales e8m0 [M_pad, K/32], M_pad = M rounded up to a multiple of 32
 *   mxfp4: y uint8 [M, K/2] (e2m1 codes, k even in the low nibble); scales e8m0 [M_pad, K/32]
 *   nvfp4: y uint8 [M, K/2];        scales fp8 e4m3 [M_pad, K/16], M_pad = M rounded up to a multiple of 16
 * The block scale is the next power of two of amax / qmax (exponent clamped to [-30, 127]); nvfp4: e4m3(amax / 0.3).
 * Rows M .. M_pad-1 of `scales` receive the scale of an all-zero block, like the reference's padded programs. */
int gemlite_hip_scale_activations_mxfp8(const void* x, void* y, uint8_t* scales, int64_t M, int64_t K,
                                        int64_t stride_xm, int32_t in_dtype, void* stream);
int gemlite_hip_scale_activations_mxfp4(const void* x, uint8_t* y, uint8_t* scales, int64_t M, int64_t K,
                                        int64_t stride_xm, int32_t in_dtype, void* stream);
int gemlite_hip_scale_activations_nvfp4(const void* x, uint8_t* y, uint8_t* scales, int64_t M, int64_t K,
                                        int64_t stride_xm, int32_t in_dtype, void* stream);

/* Bit-pack W_q[N, K] (uint8 values < 2^W_nbits, row stride ld_in) along K into words of
 * pack_bits: word(n, j) = OR_i W_q[n, j*e+i] << (W_nbits*i), e = pack_bits/W_nbits, stored
 * TRANSPOSED and contiguous: out[j*N + n]  (shape [K/e, N]) — the layout GemLiteLinear.pack()
 * produces with transpose=True + .contiguous() (core.py:384-398,478-480). */
int gemlite_hip_pack_over_cols(const uint8_t* w_q, void* out, int64_t N, int64_t K, int64_t ld_in,
                               int32_t W_nbits, int32_t pack_bits, void* stream);
/* Inverse: out[N, K] uint8 from packed [K/e, N]. */
int gemlite_hip_unpack_over_cols(const void* packed, uint8_t* out, int64_t N, int64_t K,
                                 int32_t W_nbits, int32_t pack_bits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEMLITE_HIP_H */
