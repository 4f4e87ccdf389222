// Variable-length multi-head self-attention for the encoder (gfx950), fp16 MFMA, fp32 softmax.
//
// One workgroup = one (sequence, head); 4 waves, each owning 32 query rows at a time.  Pad tokens
// do not exist in the packed layout, so "attention_mask" (data/msmarco_data.py:282) is simply the
// sequence boundary: keys >= len never enter the softmax.
//
// Swapped product S^T = K . Q^T (rows = keys, columns = queries): one lane owns one query column,
// so the running max / sum / rescale are per-lane scalars and the only cross-lane traffic is one
// exchange with lane ^ 32 per key block.  The C-layout of S^T (lane group g holds keys 4g..4g+3,
// 8+4g.. of every 16) is consumed DIRECTLY as the B operand of O^T = V^T . P^T; V arrives already
// transposed (key-contiguous) from the V^T GEMM epilogue, so no transpose is ever performed here.
//
// Measured and rejected (round 2): a workgroup owning 2-12 consecutive heads of a sequence with the next head's K / V^T / Q
// loads software-pipelined behind the current head's compute (48 more VGPRs: 2 workgroups per CU instead of 4) --
// 126-148 us per launch at the bench shape against 117 us for this kernel: the launch is bound by the dependent
// MFMA -> softmax -> MFMA chain inside each wave, which only residency (waves per SIMD) hides, not by the staging latency.
#include "common.h"
#include "attention.h"

namespace ance {
namespace {

constexpr int HD = 64;          // head dim
constexpr int ATT_THREADS = 256;

__device__ __forceinline__ int kswz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// One 32-query block (this wave's) of one (sequence, head) against all keys staged in LDS: online softmax in fp32,
// P in fp16, output rows written to ctx.  qf = the block's Q fragments (B operand layout).
__device__ __forceinline__ void attend_qblock(const AttnArgs &A, const _Float16 *Ks, const _Float16 *Vs, const f16x8 (&qf)[4], int s,
                                              int h, int tok0, int T, int Tk, int vld, int qb0, int q_end, int g, int i) {
    const int nkb = Tk >> 5;
    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 o0 = {0}, o1 = {0};
    for (int kb = 0; kb < nkb; ++kb) {
        const int krow = kb * 32 + i;
        const int ksw = (krow >> 1) & 7;
        f32x16 st = {0};
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const f16x8 kf = *reinterpret_cast<const f16x8 *>(Ks + krow * HD + (((4 * g + sx) ^ ksw) * 8));
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[sx], st, 0, 0, 0);
        }
        // st[r] = score(key kb*32 + (r&3) + 8 (r>>2) + 4 g, query i)   (Q already carries 1/8)
        const int key_base = kb * 32 + 4 * g;
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key_base + (r & 3) + 8 * (r >> 2);
            st[r] = key < T ? st[r] : -INFINITY;
            bm = fmaxf(bm, st[r]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);   // finite: key 0 of block 0 is always real
        const float alpha = __expf(m_run - m_new);
        float psum = 0.0f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __expf(st[r] - m_new);
            psum += p[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }
        // P^T fragments (B operand of O^T = V^T P^T): k-step u covers keys 16u..16u+15; this lane
        // group owns keys 16u + 4g + {0..3} and 16u + 8 + 4g + {0..3} = registers 8u..8u+7.
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (_Float16)p[8 * u + j];
            const int kc = kb * 32 + 16 * u + 4 * g;
            const _Float16 *v0 = Vs + i * vld + kc;
            const _Float16 *v1 = Vs + (i + 32) * vld + kc;
            const f16x4 a0 = *reinterpret_cast<const f16x4 *>(v0);
            const f16x4 a1 = *reinterpret_cast<const f16x4 *>(v0 + 8);
            const f16x4 c0 = *reinterpret_cast<const f16x4 *>(v1);
            const f16x4 c1 = *reinterpret_cast<const f16x4 *>(v1 + 8);
            const f16x8 vf0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const f16x8 vf1 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf0, pf, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf1, pf, o1, 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    // O^T[d][query]: d = db*32 + (r&3) + 8 (r>>2) + 4 g  ->  4 consecutive d per (db, r>>2)
    if (qb0 + i < q_end) {
        const size_t orow = A.cls_only ? (size_t)s : (size_t)(tok0 + qb0 + i);
        _Float16 *op = A.ctx + orow * A.ld_ctx + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const f32x16 &o = db == 0 ? o0 : o1;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f16x4 v = {(_Float16)(o[4 * rq + 0] * inv), (_Float16)(o[4 * rq + 1] * inv),
                                 (_Float16)(o[4 * rq + 2] * inv), (_Float16)(o[4 * rq + 3] * inv)};
                *reinterpret_cast<f16x4 *>(op + db * 32 + 8 * rq) = v;
            }
        }
    }
}

__global__ void __launch_bounds__(ATT_THREADS, 4) attention_kernel(const AttnArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int s = blockIdx.x / A.n_heads;
    const int h = blockIdx.x - s * A.n_heads;
    const int tok0 = A.seq_off[s];
    const int T = A.seq_off[s + 1] - tok0;          // 1..max_seq_len
    const int vcol0 = A.seq_vtcol[s];               // 8-aligned first key column in V^T
    const int Tk = (T + 31) & ~31;                   // keys padded to the MFMA block
    const int vld = Tk + 4;                          // V^T LDS row stride (halves): 8 * odd bytes
    _Float16 *Ks = reinterpret_cast<_Float16 *>(smem_f);  // [Tk][64], chunk-swizzled
    _Float16 *Vs = Ks + (size_t)Tk * HD;                   // [64][vld]

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, g = l >> 5, i = l & 31;
    const int H = A.n_heads * HD;

    // Q fragment of this wave's first query block: requested before the staging so that its latency
    // overlaps the K / V^T loads (B operand: lane (query i, group g) holds head dims 32 g + 8 s .. + 8)
    const int q_end = A.cls_only ? 1 : T;  // last layer: only the [CLS] query feeds the head
    f16x8 qf[4];
    if (w * 32 < q_end) {
        const _Float16 *qp = A.qk + (size_t)(tok0 + min(w * 32 + i, T - 1)) * A.ld_qk + h * HD + 32 * g;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) qf[sx] = *reinterpret_cast<const f16x8 *>(qp + sx * 8);
    }

    // ---- stage K (rows = keys) and V^T (rows = head dims) into LDS; rows/cols >= T are zeroed ---
    // Eight independent 16-byte loads per thread (4 of K, 4 of V^T: both tiles have Tk * 8 chunks) are in
    // flight before the first LDS store -- a plain load -> store loop waits one full memory latency per
    // iteration, and this kernel is latency-bound (a (sequence, head) is ~0.7 MFLOP).
    {
        const _Float16 *kbase = A.qk + (size_t)tok0 * A.ld_qk + H + h * HD;
        const _Float16 *vbase = A.vt + (size_t)(h * HD) * A.ld_vt + vcol0;
        const int nch = Tk >> 3;  // 16-byte chunks per V^T row
        const int ne = Tk * 8;    // chunks of K ([Tk][8]) = chunks of V^T ([64][Tk / 8])
        for (int e0 = tid; e0 < ne; e0 += 4 * ATT_THREADS) {
            f16x8 kv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * ATT_THREADS, row = e >> 3, ch = e & 7;
                const int dd = e / nch, key0 = (e - dd * nch) * 8;
                kv[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                vv[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (e < ne && row < T) kv[u] = *reinterpret_cast<const f16x8 *>(kbase + (size_t)row * A.ld_qk + ch * 8);
                if (e < ne && key0 < T) vv[u] = *reinterpret_cast<const f16x8 *>(vbase + (size_t)dd * A.ld_vt + key0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * ATT_THREADS, row = e >> 3, ch = e & 7;
                const int dd = e / nch, key0 = (e - dd * nch) * 8;
                if (e >= ne) continue;
                *reinterpret_cast<f16x8 *>(Ks + row * HD + kswz(row, ch) * 8) = kv[u];
                if (key0 + 8 > T) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (key0 + j >= T) vv[u][j] = (_Float16)0.0f;
                }
                _Float16 *dst = Vs + dd * vld + key0;  // 8-byte aligned
                *reinterpret_cast<f16x4 *>(dst) = f16x4{vv[u][0], vv[u][1], vv[u][2], vv[u][3]};
                *reinterpret_cast<f16x4 *>(dst + 4) = f16x4{vv[u][4], vv[u][5], vv[u][6], vv[u][7]};
            }
        }
    }
    __syncthreads();

    for (int qb0 = w * 32; qb0 < q_end; qb0 += 128) {
        if (qb0 != w * 32) {  // later query blocks of long sequences (the first one was prefetched above)
            const _Float16 *qp = A.qk + (size_t)(tok0 + min(qb0 + i, T - 1)) * A.ld_qk + h * HD + 32 * g;
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) qf[sx] = *reinterpret_cast<const f16x8 *>(qp + sx * 8);
        }

        attend_qblock(A, Ks, Vs, qf, s, h, tok0, T, Tk, vld, qb0, q_end, g, i);
    }
}


}  // namespace

size_t attention_lds_bytes(int max_seq_len) {
    const int Tk = (max_seq_len + 31) & ~31;
    return (size_t)Tk * HD * 2 + (size_t)HD * (Tk + 4) * 2;
}

int launch_attention(const AttnArgs &A, int n_seq, int max_seq_len, hipStream_t st) {
    if (n_seq <= 0) return ANCE_OK;
    const size_t lds = attention_lds_bytes(max_seq_len);
    if (lds > 160 * 1024) {
        set_last_error("attention: sequence too long for LDS");
        return ANCE_E_INVALID;
    }
    static size_t attr_set = 0;
    if (lds > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return check_launch("attention attr");
        attr_set = lds;
    }
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)n_seq * A.n_heads), dim3(ATT_THREADS), lds, st, A);
    return ANCE_OK;
}

}  // namespace ance
