// Internal interface of the encoder's fp16 MFMA GEMM (see gemm256_f16.hip).
#pragma once
#include "common.h"

namespace ance {

// EPI_*_F: the A-side LayerNorm is folded into the GEMM (encoder.hip, "LayerNorm without a kernel"): the token operand is
// fp16 of the PRE-LayerNorm row, the weight is fp16(gamma (.) W), and the epilogue finishes  r (acc - mu c) + b'  with the
// per-token (mu, r) and the per-feature c = csum.  EPI_RESLN: EPI_RES32 with the residual stream kept as an fp16 (hi, lo)
// pair and the per-row statistics of its OUTPUT left as partial (mean, M2) of every 64-column slice (part_out).
// The statistics a tile needs come from part_in: the slice partials of its 256 token rows are copied into LDS by LDS-DMA
// before the main loop (with the tile's bias / csum / gamma / beta vectors) and combined there when the epilogue starts --
// no LayerNorm kernel, no statistics kernel, no parameter load left on the epilogue's critical path.
// EPI_S_*: the SPLIT (fp32-grade) GEMM of gemm256_f16.hip -- operands are fp16 (hi, lo) pair rows, three MFMAs per k-step from four
// staged operand tiles (pipe256.h: PAIR3); epilogues in gemm256_epilogue.h.
enum { EPI_QK = 0, EPI_GELU = 1, EPI_RES32 = 2, EPI_VT = 3, EPI_RESLN = 4, EPI_QK_F = 5, EPI_GELU_F = 6, EPI_VT_F = 7,
       EPI_S_QKV = 8, EPI_S_GELU = 9, EPI_S_RESLN = 10, EPI_COUNT = 11 };

struct GemmArgs {
    const _Float16 *A;  // [M, K], row stride lda (halves)
    const _Float16 *B;  // [N, K], row stride ldb
    int lda, ldb;
    int M, N, K;        // M, N multiples of 256; K multiple of 64
    const float *bias;  // per column n (EPI_QK / GELU / RES32) or per row m (EPI_VT)
    _Float16 *out16;
    float *out32;
    const float *res32;  // EPI_RES32: residual, same layout as out32
    // EPI_RES32, optional: the residual is LayerNorm(res32) and is recomputed here from the pre-LN rows and the
    // per-row (mean, rstd) the LayerNorm kernel left -- the normalised fp32 rows are never written to HBM.
    const float *res_stats;  // [M][2] or null (res32 is then used as it is)
    const float *res_gamma, *res_beta;  // [N]
    int ldc;             // row stride of out16 / out32 / res32 (elements)
    float scale;         // EPI_QK: applied to columns n < scale_cols
    int scale_cols;
    const int *col_map;  // EPI_VT: token n -> destination column
    int n_valid;         // EPI_VT: columns n >= n_valid are not stored
    // folded LayerNorm: part_in[token][12][2] = (mean, M2) of the twelve 64-column slices of the token's pre-LayerNorm row
    // (tokens are the rows m for QK_F / GELU_F / RESLN's residual, the columns n for VT_F), ln_eps, and for EPI_*_F the
    // per-feature sum of the folded fp16 weight row
    const float *part_in;
    float ln_eps;
    const float *csum;
    // EPI_RESLN: residual = LayerNorm(res_hi + res_lo) with the statistics of part_in and res_gamma / res_beta; outputs
    // out16 (hi), out_lo and part_out[m][N / 64][2] = (mean, M2) of the 64 output columns each wave owns
    const _Float16 *res_hi, *res_lo;
    _Float16 *out_lo;
    float *part_out;
    // folded epilogues, optional: the lo halves of the token operand (same layout as the operand itself).  A tile with a token
    // whose |mean| rstd exceeds FOLD_WIDE_MEAN runs a second K loop over them, so that the operand carries 22 bits there.
    const _Float16 *tok_lo;
    // split GEMM: operand rows are pair rows (common.h) of 2 K halves (lda / ldb = 2 K or more); EPI_S_RESLN reads the residual pair
    // rows (2 N halves) at row stride ldr and writes pair rows at row stride ldc; wscale_inv: device scalar, the inverse of the
    // power of two the B operand (the weight) was stored with, or null (1)
    int ldr;
    const float *wscale_inv;
    unsigned *range_faults;  // split epilogues: sticky counter of threads that stored a value outside the fp16 range (common.h: range_report), or null
    int n_split;         // 2: N-split tile order (gemm256_f16.hip: tile_of_block; desc / split kernels only, N / 256 even); else 0
    int debug_mode;      // ance_debug_gemm ablations: 1 = no loads after tile 0, 2 = no MFMA, 4 = all blocks load tile (0,0)
};

// y = (x - mean) * rstd * gamma + beta -- the ONE expression every LayerNorm consumer uses, so that the fp32
// residual recomputed in a GEMM epilogue is bitwise the value the LayerNorm kernel rounded to fp16.
__device__ __forceinline__ f32x4 ln_apply4(f32x4 x, float mean, float rstd, f32x4 g, f32x4 b) {
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = (x[j] - mean) * rstd * g[j] + b[j];
    return y;
}

// sum over the 16 lanes of a DPP row (the 16 lanes that cover one 64-float slice of a row)
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_amdgcn_update_dpp(0.f, x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0.f, x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0.f, x, 0x124, 0xF, 0xF, true);  // row_ror:4
    x += __builtin_amdgcn_update_dpp(0.f, x, 0x128, 0xF, 0xF, true);  // row_ror:8
    return x;
}

// (mean, rstd) of a 768-wide row from the (mean, M2) of its twelve 64-column slices: Chan's combination with equal counts
__device__ __forceinline__ void stats_from_parts(const float *pp, float eps, float *mean, float *rstd) {
    float m = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) m += pp[2 * j];
    m *= 1.0f / 12.0f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const float d = pp[2 * j] - m;
        q += pp[2 * j + 1] + 64.0f * d * d;
    }
    *mean = m;
    *rstd = rsqrtf(q * (1.0f / 768.0f) + eps);
}

// epilogue parameter block in LDS, above the 128 KiB of stage buffers (floats)
constexpr int EPB_OFF = 32768;           // = 128 KiB
constexpr int EPB_PART = 0;              // [256 tokens][24]: slice partials (24 KiB)
constexpr int EPB_STATS = 256 * 24;      // [256][2]: (mean, rstd), filled when the epilogue starts
constexpr int EPB_VEC = EPB_STATS + 512; // three vectors of 256 floats: bias, csum | gamma, beta
constexpr int EPB_FLOATS = EPB_VEC + 768;

// 256 x 256 x 64 tile kernel of gemm256_f16.hip (M, N multiples of 256, K of 64).
int launch_gemm_f16(int epi, const GemmArgs &args, hipStream_t stream);
bool gemm256_applicable(const GemmArgs &args);
void reload_gemm_knobs();  // re-read ANCE_GEMM_STREAM (ance_reload_env)

}  // namespace ance
