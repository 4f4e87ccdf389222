// Internal interface of the encoder's fp16 MFMA GEMM (see gemm256_f16.hip).
#pragma once
#include "common.h"

namespace ance {

enum { EPI_QK = 0, EPI_GELU = 1, EPI_RES32 = 2, EPI_VT = 3 };

struct GemmArgs {
    const _Float16 *A;  // [M, K], row stride lda (halves)
    const _Float16 *B;  // [N, K], row stride ldb
    int lda, ldb;
    int M, N, K;        // M, N multiples of 256; K multiple of 64
    const float *bias;  // per column n (EPI_QK / GELU / RES32) or per row m (EPI_VT)
    _Float16 *out16;
    float *out32;
    const float *res32;  // EPI_RES32: residual, same layout as out32
    int ldc;             // row stride of out16 / out32 / res32 (elements)
    float scale;         // EPI_QK: applied to columns n < scale_cols
    int scale_cols;
    const int *col_map;  // EPI_VT: token n -> destination column
    int n_valid;         // EPI_VT: columns n >= n_valid are not stored
    int debug_mode;      // ance_debug_gemm ablations: 1 = no loads after tile 0, 2 = no MFMA, 4 = all blocks load tile (0,0)
};

// 256 x 256 x 64 tile kernel of gemm256_f16.hip (M, N multiples of 256, K of 64).
int launch_gemm_f16(int epi, const GemmArgs &args, hipStream_t stream);
bool gemm256_applicable(const GemmArgs &args);

}  // namespace ance
