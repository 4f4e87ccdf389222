// fp32 reference-grade encoder path (ANCE_ENCODER_PRECISE=1): the reference's own arithmetic -- fp32 operands, fp32
// accumulation, exact erf GELU, fp32 softmax (model/models.py:149-157 runs the tower in fp32, no .half()) -- on the
// fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, 157 TFLOP/s peak = 1/16 of the fp16 rate).
// It exists so that the price of the default mode's fp16 MFMA operands (max |delta| 3e-3 on the embeddings, near-tie swaps
// in the negative lists) is a CHOICE with numbers: bench.py reports both modes side by side.  Structure is deliberately
// plain (one GEMM kernel with three epilogues, LayerNorm and attention as their own kernels, nothing fused across
// layers): this path is the audit, not the product.
#pragma once
#include "common.h"

namespace ance {
namespace {

constexpr int P_TM = 128, P_TN = 128, P_BK = 32;
constexpr int P_LD = P_BK + 4;  // floats per LDS row (144 B): 16-byte aligned rows, conflict-free ds_read_b128
constexpr size_t P_GEMM_LDS = (size_t)2 * (P_TM + P_TN) * P_LD * sizeof(float);  // two stages: 73,728 B
enum { P_EPI_BIAS = 0, P_EPI_GELU = 1, P_EPI_RES = 2 };

// out[m][n] = epi(sum_k A[m][k] B[n][k] + bias[n])   A [M,K], B [N,K] row-major fp32 (nn.Linear layout), M, N % 128 == 0,
// K % 32 == 0.  128 x 128 tile, 4 waves of 64 x 64 (2 x 2 MFMA 32x32 blocks), register-staged double-buffered LDS.
template <int EPI>
__global__ void __launch_bounds__(256) gemm32_kernel(const float *A, int lda, const float *B, int ldb, const float *bias,
                                                     const float *res, int ldr, float *out, int ldc, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem_p[];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 5, i = l & 31;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.y * P_TM, n0 = blockIdx.x * P_TN;
    // staging: element e = tid + 256 j -> row e >> 3, float4 column e & 7 (8 lanes = one 128-byte line)
    const int srow = tid >> 3, sc4 = tid & 7;
    const float *ap = A + (size_t)(m0 + srow) * lda + sc4 * 4;
    const float *bp = B + (size_t)(n0 + srow) * ldb + sc4 * 4;
    f32x4 ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const f32x4 *>(ap + (size_t)(32 * j) * lda + k0);
            rb[j] = *reinterpret_cast<const f32x4 *>(bp + (size_t)(32 * j) * ldb + k0);
        }
    };
    auto store = [&](int buf) {
        float *as = smem_p + buf * (P_TM + P_TN) * P_LD, *bs = as + P_TM * P_LD;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4 *>(as + (srow + 32 * j) * P_LD + sc4 * 4) = ra[j];
            *reinterpret_cast<f32x4 *>(bs + (srow + 32 * j) * P_LD + sc4 * 4) = rb[j];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
    const int NK = K / P_BK;
    load(0);
    store(0);
    __syncthreads();
    for (int kt = 0; kt < NK; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < NK) load((kt + 1) * P_BK);
        const float *as = smem_p + buf * (P_TM + P_TN) * P_LD + (wm * 64 + i) * P_LD + 4 * g;
        const float *bs = smem_p + buf * (P_TM + P_TN) * P_LD + (P_TM + wn * 64 + i) * P_LD + 4 * g;
        // k-step j of a 16-byte piece uses k = 8 s + 4 g + j on both operands
#pragma unroll
        for (int s = 0; s < P_BK / 8; ++s) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(as + s * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(as + 32 * P_LD + s * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bs + s * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bs + 32 * P_LD + s * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < NK) store(buf ^ 1);
        __syncthreads();
    }
    // acc[a][b][r]: row m0 + wm*64 + a*32 + (r & 3) + 8 (r >> 2) + 4 g, column n0 + wn*64 + b*32 + i
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 64 + b * 32 + i;
        const float bn = bias[n];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t m = (size_t)(m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                float v = acc[a][b][r] + bn;
                if constexpr (EPI == P_EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                if constexpr (EPI == P_EPI_RES) v += res[m * ldr + n];
                out[m * ldc + n] = v;
            }
    }
}

int launch_gemm32(int epi, const float *A, int lda, const float *B, int ldb, const float *bias, const float *res, int ldr,
                  float *out, int ldc, int M, int N, int K, hipStream_t st) {
    if (M % P_TM || N % P_TN || K % P_BK || M <= 0) {
        set_last_error("gemm32: M, N must be multiples of 128 and K of 32");
        return ANCE_E_INVALID;
    }
    void (*k)(const float *, int, const float *, int, const float *, const float *, int, float *, int, int) =
        epi == P_EPI_GELU ? gemm32_kernel<P_EPI_GELU> : (epi == P_EPI_RES ? gemm32_kernel<P_EPI_RES> : gemm32_kernel<P_EPI_BIAS>);
    static unsigned long long attr_done[3] = {0, 0, 0};
    if (attr_needed(&attr_done[epi])) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_GEMM_LDS) != hipSuccess)
            return check_launch("gemm32 attr");
        attr_mark(&attr_done[epi]);
    }
    hipLaunchKernelGGL(k, dim3(N / P_TN, M / P_TM), dim3(256), P_GEMM_LDS, st, A, lda, B, ldb, bias, res, ldr, out, ldc, K);
    return ANCE_OK;
}

// LayerNorm of 768-wide fp32 rows -> fp32 rows (+ the row statistics when stats != null); one wave per row, two passes
__global__ void __launch_bounds__(256) ln32_kernel(const float *pre, int rows, const float *gamma, const float *beta, float eps,
                                                   float *out, float *stats) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (t >= rows) return;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(pre + (size_t)t * 768);
    f32x4 v[3];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = x4[k * 64 + l];
        s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * (1.0f / 768.0f);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = v[k][j] - mean;
            q += a * a;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 768.0f) + eps);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c4 = k * 64 + l;
        const f32x4 gm = reinterpret_cast<const f32x4 *>(gamma)[c4], bt = reinterpret_cast<const f32x4 *>(beta)[c4];
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (v[k][j] - mean) * rstd * gm[j] + bt[j];
        reinterpret_cast<f32x4 *>(out + (size_t)t * 768)[c4] = y;
    }
    if (stats && l == 0) {
        stats[2 * (size_t)t] = mean;
        stats[2 * (size_t)t + 1] = rstd;
    }
}

// embeddings: (word + type) + position -> pre (fp32); the embedding LayerNorm is an ln32 launch
__global__ void __launch_bounds__(256) embed32_kernel(const int *tok_id, const int *tok_pos, int rows, const float *word,
                                                      const float *pos, const float *type0, int vocab, int max_pos, float *pre) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (t >= rows) return;
    int id = tok_id[t], p = tok_pos[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(word + (size_t)id * 768);
    const f32x4 *p4 = reinterpret_cast<const f32x4 *>(pos + (size_t)p * 768);
    const f32x4 *t4 = reinterpret_cast<const f32x4 *>(type0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c4 = k * 64 + l;
        reinterpret_cast<f32x4 *>(pre + (size_t)t * 768)[c4] = (w4[c4] + t4[c4]) + p4[c4];
    }
}

// Self-attention in fp32 on the vector units: one workgroup per (sequence, head), one thread per query row (two rounds
// for more than 256 queries), keys and values of the head staged 128 at a time in LDS and read as broadcasts; online
// softmax in fp32 with the full-precision exp.  qkv: [T, 2304] = Q | K | V rows.  ~2 % of the encoder's FLOPs.
constexpr int P_KC = 128;
constexpr size_t P_ATT_LDS = (size_t)2 * P_KC * 64 * sizeof(float);

// ctx_pair != null (split mode, encoder.hip): the output row is written as an fp16 pair row (common.h), the token operand of the
// split attention-output GEMM; cls_only: only query 0 of every sequence is
// computed and its row goes to row s (compact), as in attention.hip.
__global__ void __launch_bounds__(256) attention32_kernel(const float *qkv, float *ctx, _Float16 *ctx_pair, int cls_only,
                                                          const int *seq_off, int n_heads) {
    extern __shared__ __attribute__((aligned(16))) float smem_p[];
    float *Ks = smem_p, *Vs = smem_p + P_KC * 64;
    const int s = blockIdx.x / n_heads, h = blockIdx.x - s * n_heads;
    const int tok0 = seq_off[s], T = seq_off[s + 1] - tok0;
    const int tid = threadIdx.x;
    const int ld = 3 * 768;
    const int q_end = cls_only ? 1 : T;
    for (int q0 = 0; q0 < q_end; q0 += 256) {
        const int qi = q0 + tid;
        const bool qv = qi < q_end;
        float q[64], acc[64];
        float m = -INFINITY, lsum = 0.f;
        {
            const float *qp = qkv + (size_t)(tok0 + (qv ? qi : 0)) * ld + h * 64;
#pragma unroll
            for (int d = 0; d < 64; ++d) {
                q[d] = qp[d] * 0.125f;  // 1 / sqrt(64), exact
                acc[d] = 0.f;
            }
        }
        for (int c0 = 0; c0 < T; c0 += P_KC) {
            const int nk = (T - c0) < P_KC ? (T - c0) : P_KC;
            __syncthreads();  // the previous chunk (or round) is no longer read
            for (int e = tid; e < nk * 16; e += 256) {
                const int key = e >> 4, c4 = e & 15;
                const float *kp = qkv + (size_t)(tok0 + c0 + key) * ld + 768 + h * 64 + c4 * 4;
                *reinterpret_cast<f32x4 *>(Ks + key * 64 + c4 * 4) = *reinterpret_cast<const f32x4 *>(kp);
                *reinterpret_cast<f32x4 *>(Vs + key * 64 + c4 * 4) = *reinterpret_cast<const f32x4 *>(kp + 768);
            }
            __syncthreads();
            if (qv) {
                for (int key = 0; key < nk; ++key) {
                    const float *kr = Ks + key * 64, *vr = Vs + key * 64;
                    float sc = 0.f;
#pragma unroll
                    for (int d = 0; d < 64; ++d) sc = fmaf(q[d], kr[d], sc);
                    const float mn = fmaxf(m, sc);
                    const float a = expf(m - mn), p = expf(sc - mn);  // first key: m = -inf -> a = 0
                    lsum = lsum * a + p;
                    m = mn;
#pragma unroll
                    for (int d = 0; d < 64; ++d) acc[d] = fmaf(acc[d], a, p * vr[d]);
                }
            }
        }
        if (qv) {
            const float inv = 1.0f / lsum;
            const size_t orow = cls_only ? (size_t)s : (size_t)(tok0 + qi);
            if (ctx_pair) {
                _Float16 *ph = ctx_pair + orow * 1536;
#pragma unroll
                for (int d = 0; d < 64; d += 4) {
                    const f32x4 v = {acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv};
                    pair_store4(v, ph, 768, h * 64 + d);  // the stored hi and the hi of (v - hi) are the same bits (common.h)
                }
            } else {
                float *op = ctx + orow * 768 + h * 64;
#pragma unroll
                for (int d = 0; d < 64; ++d) op[d] = acc[d] * inv;
            }
        }
    }
}

int launch_attention32(const float *qkv, float *ctx, const int *seq_off, int n_seq, int n_heads, hipStream_t st,
                       _Float16 *ctx_pair = nullptr, int cls_only = 0) {
    static unsigned long long attr_done = 0;
    if (attr_needed(&attr_done)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(attention32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)P_ATT_LDS) != hipSuccess)
            return check_launch("attention32 attr");
        attr_mark(&attr_done);
    }
    hipLaunchKernelGGL(attention32_kernel, dim3((unsigned)n_seq * n_heads), dim3(256), P_ATT_LDS, st, qkv, ctx, ctx_pair, cls_only, seq_off, n_heads);
    return ANCE_OK;
}

}  // namespace
}  // namespace ance
