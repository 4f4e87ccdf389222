// fp16 MFMA GEMM with fused epilogues for the encoder (gfx950).
//
//   C[m][n] = sum_k A[m][k] * B[n][k]      A [M,K] row-major f16, B [N,K] row-major f16 (nn.Linear
//                                           weight layout, so y = x W^T needs no transpose)
//
// Tile 128 x 128 x 64, 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_f16 blocks,
// fp32 accumulation.  LDS rows are 128 B (64 halves) with the 16-byte chunk index XOR-swizzled by
// ((row >> 1) & 7), which makes every ds_read_b128 of an MFMA fragment (16 lanes = 16 distinct
// rows, same chunk) hit 16 distinct 16-byte slots of the 256-byte bank row.  Global->LDS staging
// goes through registers (loads for tile t+1 are in flight while tile t is multiplied), one
// barrier per K-tile, two LDS stages.
//
// Epilogues (what the reference computes after each nn.Linear, fused here):
//   EPI_QK     out16 = (acc + bias[n]) * (n < scale_cols ? scale : 1)      Q | K projection, Q/8
//   EPI_GELU   out16 = gelu_erf(acc + bias[n])                            intermediate.dense
//   EPI_RES32  out32 = acc + bias[n] + res32[m][n]                        attention.output.dense / output.dense
//   EPI_VT     out16[m][col[n]] = acc + bias[m]   (n < n_valid)           V^T = Wv . h^T, key-contiguous
#include "common.h"
#include "gemm_f16.h"
#include <stdlib.h>
#include <string.h>

#ifndef ANCE_GEMM_DEFAULT
#define ANCE_GEMM_DEFAULT 2
#endif

namespace ance {
namespace {

constexpr int BM = 128, BN = 128, BKH = 64;   // BKH halves = 128 bytes per LDS row
constexpr int TILE_HALVES = BM * BKH;         // one operand tile
constexpr int GEMM_THREADS = 256;
constexpr size_t GEMM_LDS_BYTES = (size_t)2 * 2 * TILE_HALVES * sizeof(_Float16);  // 64 KiB

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// GELU(x) = x * Phi(x) with Phi from the Abramowitz-Stegun 7.1.26 erfc polynomial (|abs err| of
// erf <= 1.5e-7, far below the fp16 resolution of the stored result): ~14 VALU ops instead of the
// ~30 of ocml's erff, which matters because this epilogue runs on 3072 columns per token with no
// MFMA work to hide behind.  The erfc form keeps the negative tail free of cancellation.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float h = 0.5f * p * t * __expf(-az * az);  // 0.5 * erfc(|z|)
    return x * (z >= 0.0f ? 1.0f - h : h);
}

template <int EPI>
__global__ void __launch_bounds__(GEMM_THREADS, 2) gemm_f16_kernel(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    _Float16 *smem = reinterpret_cast<_Float16 *>(smem_f);

    // ---- block -> tile, XCD-aware: one XCD sweeps n for a fixed m-panel (A panel stays in its L2)
    const int NT = G.N / BN, MT = G.M / BM;
    // XCD-aware tile order (speed only): blocks b, b+8, ... share an XCD.  The dimension with more
    // tiles is dealt round-robin to the XCDs, the other one is swept fastest, so the panel of the
    // outer dimension stays in that XCD's L2 while the inner panels stream through it.
    const int b = blockIdx.x, xcd = b & 7, jx = b >> 3;
    int mt, nt;
    if (MT >= NT) {
        mt = (jx / NT) * 8 + xcd;
        nt = jx % NT;
    } else {
        nt = (jx / MT) * 8 + xcd;
        mt = jx % MT;
    }
    if (mt >= MT || nt >= NT) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, g = l >> 5, i = l & 31;
    const int wm = w >> 1, wn = w & 1;

    // staging: element e = tid + 256 j -> row e >> 3 (0..127), 16-byte chunk e & 7
    const int srow = tid >> 3, sch = tid & 7;
    const _Float16 *ga = G.A + (size_t)(m0 + srow) * G.lda + sch * 8;
    const _Float16 *gb = G.B + (size_t)(n0 + srow) * G.ldb + sch * 8;
    const size_t a_step = (size_t)32 * G.lda, b_step = (size_t)32 * G.ldb;
    int lds_w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = srow + 32 * j;
        lds_w[j] = row * BKH + swz_chunk(row, sch) * 8;
    }
    f16x8 ra[4], rb[4];
    auto load_regs = [&](int kt) {
        const int k0 = kt * BKH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const f16x8 *>(ga + j * a_step + k0);
            rb[j] = *reinterpret_cast<const f16x8 *>(gb + j * b_step + k0);
        }
    };
    auto write_lds = [&](int buf) {
        _Float16 *sa = smem + buf * 2 * TILE_HALVES;
        _Float16 *sb = sa + TILE_HALVES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f16x8 *>(sa + lds_w[j]) = ra[j];
            *reinterpret_cast<f16x8 *>(sb + lds_w[j]) = rb[j];
        }
    };

    // fragment rows of this lane
    const int ar0 = wm * 64 + i, ar1 = ar0 + 32;
    const int br0 = wn * 64 + i, br1 = br0 + 32;
    const int asw0 = (ar0 >> 1) & 7, asw1 = (ar1 >> 1) & 7;
    const int bsw0 = (br0 >> 1) & 7, bsw1 = (br1 >> 1) & 7;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int NK = G.K / BKH;
    load_regs(0);
    int buf = 0;
    for (int kt = 0; kt < NK; ++kt) {
        write_lds(buf);
        __syncthreads();
        if (kt + 1 < NK) load_regs(kt + 1);
        const _Float16 *sa = smem + buf * 2 * TILE_HALVES;
        const _Float16 *sb = sa + TILE_HALVES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = 2 * s + g;  // lane group g carries k = 16 s + 8 g .. + 8
            const f16x8 a0 = *reinterpret_cast<const f16x8 *>(sa + ar0 * BKH + ((ch ^ asw0) * 8));
            const f16x8 a1 = *reinterpret_cast<const f16x8 *>(sa + ar1 * BKH + ((ch ^ asw1) * 8));
            const f16x8 b0 = *reinterpret_cast<const f16x8 *>(sb + br0 * BKH + ((ch ^ bsw0) * 8));
            const f16x8 b1 = *reinterpret_cast<const f16x8 *>(sb + br1 * BKH + ((ch ^ bsw1) * 8));
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc11, 0, 0, 0);
        }
        buf ^= 1;
    }

    // ---- epilogue: C/D layout  col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -------
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const f32x16 &a = mb == 0 ? (nb == 0 ? acc00 : acc01) : (nb == 0 ? acc10 : acc11);
            const int n = n0 + wn * 64 + nb * 32 + i;
            const int mbase = m0 + wm * 64 + mb * 32 + 4 * g;
            if constexpr (EPI == EPI_QK) {
                const float bias = G.bias[n];
                const float sc = n < G.scale_cols ? G.scale : 1.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    G.out16[(size_t)m * G.ldc + n] = (_Float16)((a[r] + bias) * sc);
                }
            } else if constexpr (EPI == EPI_GELU) {
                const float bias = G.bias[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    G.out16[(size_t)m * G.ldc + n] = (_Float16)gelu_erf(a[r] + bias);
                }
            } else if constexpr (EPI == EPI_RES32) {
                const float bias = G.bias[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    const size_t o = (size_t)m * G.ldc + n;
                    G.out32[o] = a[r] + bias + G.res32[o];
                }
            } else {  // EPI_VT: rows are features (bias per row), columns are tokens scattered to
                      // their sequence's 8-aligned key column
                if (n < G.n_valid) {
                    const int col = G.col_map[n];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mbase + (r & 3) + 8 * (r >> 2);
                        G.out16[(size_t)m * G.ldc + col] = (_Float16)(a[r] + G.bias[m]);
                    }
                }
            }
        }
    }
}

}  // namespace

int launch_gemm128_f16(int epi, const GemmArgs &G, hipStream_t st) {
    if (G.M % BM || G.N % BN || G.K % BKH || G.M <= 0 || G.N <= 0 || G.K <= 0) {
        set_last_error("gemm_f16: M,N must be multiples of 128 and K of 64");
        return ANCE_E_INVALID;
    }
    const int MT = G.M / BM, NT = G.N / BN;
    const unsigned blocks = MT >= NT ? (unsigned)((MT + 7) / 8 * 8) * (unsigned)NT : (unsigned)((NT + 7) / 8 * 8) * (unsigned)MT;
    void (*k)(const GemmArgs) = nullptr;
    switch (epi) {
        case EPI_QK: k = gemm_f16_kernel<EPI_QK>; break;
        case EPI_GELU: k = gemm_f16_kernel<EPI_GELU>; break;
        case EPI_RES32: k = gemm_f16_kernel<EPI_RES32>; break;
        case EPI_VT: k = gemm_f16_kernel<EPI_VT>; break;
        default: set_last_error("gemm_f16: bad epilogue"); return ANCE_E_INVALID;
    }
    static bool attr_done[4] = {false, false, false, false};
    if (!attr_done[epi]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)GEMM_LDS_BYTES) != hipSuccess)
            return check_launch("gemm_f16 attr");
        attr_done[epi] = true;
    }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st, G);
    return ANCE_OK;
}

int launch_gemm_f16(int epi, const GemmArgs &G, hipStream_t st) {
    static int variant = -1;  // 0: 128 tile, 1: 256 register staged, 2: 256 direct-to-LDS
    if (variant < 0) {
        const char *e = getenv("ANCE_GEMM");
        variant = ANCE_GEMM_DEFAULT;
        if (e) {
            if (!strcmp(e, "128")) variant = 0;
            else if (!strcmp(e, "256reg")) variant = 1;
            else if (!strcmp(e, "256glds")) variant = 2;
        }
    }
    if (variant > 0 && gemm256_applicable(G)) return launch_gemm256_f16(epi, G, variant == 2, st);
    return launch_gemm128_f16(epi, G, st);
}

}  // namespace ance
