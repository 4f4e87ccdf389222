// Internal interface of the encoder's fp16 MFMA GEMM (see gemm256_f16.hip).
#pragma once
#include "common.h"

namespace ance {

// EPI_*_F: the A-side LayerNorm is folded into the GEMM (encoder.hip, "LayerNorm without a kernel"): the token operand is
// fp16 of the PRE-LayerNorm row, the weight is fp16(gamma (.) W), and the epilogue finishes  r (acc - mu c) + b'  with the
// per-token (mu, r) = row_stats and the per-feature c = csum.  EPI_RESLN: EPI_RES32 with the residual stream kept as an
// fp16 (hi, lo) pair and the per-row statistics of its OUTPUT left as partial (mean, M2) of every 64-column slice.
enum { EPI_QK = 0, EPI_GELU = 1, EPI_RES32 = 2, EPI_VT = 3, EPI_RESLN = 4, EPI_QK_F = 5, EPI_GELU_F = 6, EPI_VT_F = 7, EPI_COUNT = 8 };

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
    // folded LayerNorm (EPI_*_F): token statistics (mean, rstd) -- tokens are the rows m (QK_F / GELU_F) or the columns n
    // (VT_F) -- and the per-feature sum of the folded fp16 weight row
    const float *row_stats;
    const float *csum;
    // EPI_RESLN: residual = LayerNorm(res_hi + res_lo) with res_stats / res_gamma / res_beta; outputs out16 (hi), out_lo
    // and part[m][N / 64][2] = (mean, M2) of the 64 output columns each wave owns
    const _Float16 *res_hi, *res_lo;
    _Float16 *out_lo;
    float *part;
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

// 256 x 256 x 64 tile kernel of gemm256_f16.hip (M, N multiples of 256, K of 64).
int launch_gemm_f16(int epi, const GemmArgs &args, hipStream_t stream);
bool gemm256_applicable(const GemmArgs &args);

}  // namespace ance
