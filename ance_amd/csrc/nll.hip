// Forward of the reference's training objective on embeddings the encoder produced (first slice -- forward
// only, no backward, no optimizer): model/models.py:57-81 (NLL.forward) and :84-134 (NLL_MultiChunk.forward, MaxP)
//     logit_a[b] = q[b] . a[b]                                  FirstP
//                = max_c ( q[b] . a[b][c] + (1 - m_a[b][c]) (-9999) )   MaxP: m = the first attention-mask entry of chunk c
//     loss[b]    = -log_softmax([logit_a, logit_b])[0] = log(1 + exp(logit_b - logit_a))      (computed stably)
//     mean loss  = sum_b loss[b] / n
// HBM-bound by construction: (1 + 2 chunks) x d x 4 bytes read per triplet, 12 bytes written -- one wave per triplet, every
// row read once with 16-byte loads; the mean is a second, single-block, fixed-order reduction (no atomics: the same input
// gives the same bits).  Dot products are fp32 with a fixed summation order (lane-strided partial sums, xor-shuffle tree).
#include "common.h"

namespace ance {
namespace {

__device__ __forceinline__ float wave_dot(const float *x, const float *y, int d, int l) {
    float acc = 0.f;
    for (int c4 = l; c4 < d / 4; c4 += 64) {
        const f32x4 a = reinterpret_cast<const f32x4 *>(x)[c4], b = reinterpret_cast<const f32x4 *>(y)[c4];
        acc = __builtin_fmaf(a[0], b[0], acc);
        acc = __builtin_fmaf(a[1], b[1], acc);
        acc = __builtin_fmaf(a[2], b[2], acc);
        acc = __builtin_fmaf(a[3], b[3], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    return acc;
}

__global__ void __launch_bounds__(256) nll_rows_kernel(const float *q, const float *a, const float *b, const float *mask_a,
                                                       const float *mask_b, int64_t n, int d, int chunks, float *logits,
                                                       float *loss_rows) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (r >= n) return;
    const float *qr = q + r * d;
    float la = -INFINITY, lb = -INFINITY;
    for (int c = 0; c < chunks; ++c) {
        float sa = wave_dot(qr, a + (r * chunks + c) * d, d, l);
        float sb = wave_dot(qr, b + (r * chunks + c) * d, d, l);
        if (chunks > 1) {  // (1 - mask) * (-9999) added before the max (model/models.py:109-113, 121-127)
            sa += (1.0f - mask_a[r * chunks + c]) * -9999.0f;
            sb += (1.0f - mask_b[r * chunks + c]) * -9999.0f;
        }
        la = fmaxf(la, sa);
        lb = fmaxf(lb, sb);
    }
    if (l == 0) {
        const float m = fmaxf(la, lb);
        const float lse = m + logf(expf(la - m) + expf(lb - m));
        logits[2 * r] = la;
        logits[2 * r + 1] = lb;
        loss_rows[r] = lse - la;
    }
}

// fixed-order mean of n floats: 1024 threads accumulate strided partial sums, then a shared-memory tree
__global__ void __launch_bounds__(1024) mean_kernel(const float *x, int64_t n, float *out) {
    __shared__ float s[1024];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += x[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0] / (float)n;
}

}  // namespace
}  // namespace ance

extern "C" int ance_nll_forward(const float *d_q, const float *d_a, const float *d_b, const float *d_mask_a, const float *d_mask_b,
                                int64_t n, int d, int chunks, float *d_logits, float *d_loss_rows, float *d_loss_mean, void *stream) {
    using namespace ance;
    if (!d_q || !d_a || !d_b || !d_logits || !d_loss_rows || !d_loss_mean || n < 1 || d < 4 || d % 4 || chunks < 1 ||
        (chunks > 1 && (!d_mask_a || !d_mask_b))) {
        set_last_error("ance_nll_forward: invalid argument");
        return ANCE_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nll_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, d_q, d_a, d_b, d_mask_a, d_mask_b, n, d, chunks,
                       d_logits, d_loss_rows);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, st, (const float *)d_loss_rows, n, d_loss_mean);
    return check_launch("ance_nll_forward");
}
