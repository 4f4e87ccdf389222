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
#include <stdlib.h>

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
        // st[r] = score(key kb*32 + (r&3) + 8 (r>>2) + 4 g, query i) in the log2 domain (Q carries log2(e)/8)
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
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.0f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
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


// ------------------------------------------------------------------ short sequences (T <= 128) --
// Register-resident variant: one WAVE = one (sequence, head); the four waves of a workgroup are four independent
// units (no LDS, no barrier, no idle wave when a sequence has fewer than four query blocks).  All of K and V^T of the
// head sit in the wave's registers as MFMA A-operand fragments, loaded straight from global memory once, and the wave
// walks its query blocks with the next block's Q fragments in flight.  With N = ceil(T / 32) a template parameter the
// key loop is straight-line code.
//
// Key order inside a 32-key block.  MFMA row rho of S^T = K . Q^T is fed with key  kb*32 + pi(rho),  pi = swap bits
// 2 and 3: register r of lane group g then holds key  kb*32 + 16 (r >> 3) + 8 g + (r & 7),  i.e. the eight k-slots a
// lane group supplies to one k-step of O^T = V^T . P^T are EIGHT CONSECUTIVE keys -- the matching V^T fragment is one
// 16-byte load of a key-contiguous V^T row instead of two 8-byte pieces.
__device__ __forceinline__ int key_perm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// One instruction; an fmaxf chain gets a canonicalising v_max in front of every MFMA output.  hipcc does not pad the
// MFMA-result -> VALU-read hazard for an instruction inside an asm statement: the caller puts mfma_result_pad() between
// the last MFMA that wrote the operands and the first max3f that reads them (without it the maximum is read from
// registers the matrix pipe has not written yet -- on some waves, on some launches).
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// 12 wait states (8-pass XDL result -> any reader), tied to the accumulator by a read-write operand: the statement cannot
// move above the MFMA that produces it, and no reader of it can move above the statement.
__device__ __forceinline__ void mfma_result_pad(f32x16 &acc) { asm volatile("s_nop 11" : "+v"(acc)); }

template <int N>
__device__ __forceinline__ void attend_reg(const AttnArgs &A, int s, int h, int tok0, int T, int vcol0, int l) {
    const int g = l >> 5, i = l & 31;
    const int H = A.n_heads * HD;
    f16x8 kf[N][4], vf[N][2][2];
    {
        const _Float16 *kbase = A.qk + (size_t)tok0 * A.ld_qk + H + h * HD + 32 * g;
        const int pi = key_perm(i);
#pragma unroll
        for (int kb = 0; kb < N; ++kb) {
            const int key = kb * 32 + pi;
            const _Float16 *kp = kbase + (size_t)(key < T ? key : T - 1) * A.ld_qk;  // keys >= T are masked below
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) kf[kb][sx] = *reinterpret_cast<const f16x8 *>(kp + sx * 8);
        }
        const _Float16 *vbase = A.vt + (size_t)(h * HD + i) * A.ld_vt + vcol0 + 8 * g;
#pragma unroll
        for (int kb = 0; kb < N; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int key0 = kb * 32 + 16 * u + 8 * g;
                f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kb < N - 1 || key0 < T) {
                    a = *reinterpret_cast<const f16x8 *>(vbase + kb * 32 + 16 * u);
                    b = *reinterpret_cast<const f16x8 *>(vbase + (size_t)32 * A.ld_vt + kb * 32 + 16 * u);
                }
                if (kb == N - 1) {  // columns >= T are another sequence's keys (or stale): p is 0 there, keep 0 * x finite
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (key0 + e >= T) {
                            a[e] = (_Float16)0.0f;
                            b[e] = (_Float16)0.0f;
                        }
                }
                vf[kb][u][0] = a;
                vf[kb][u][1] = b;
            }
    }
    const int q_end = A.cls_only ? 1 : T;
    const _Float16 *qbase = A.qk + (size_t)tok0 * A.ld_qk + h * HD + 32 * g;
    f16x8 qf[4], qn[4];
    {
        const _Float16 *qp = qbase + (size_t)(i < T ? i : T - 1) * A.ld_qk;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) qf[sx] = *reinterpret_cast<const f16x8 *>(qp + sx * 8);
    }
    for (int qb0 = 0; qb0 < q_end; qb0 += 32) {
        const bool more = qb0 + 32 < q_end;
        if (more) {
            const int qr = qb0 + 32 + i;
            const _Float16 *qp = qbase + (size_t)(qr < T ? qr : T - 1) * A.ld_qk;
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) qn[sx] = *reinterpret_cast<const f16x8 *>(qp + sx * 8);
        }
        float m_run = -INFINITY, l_run = 0.0f;
        f32x16 o0 = {0}, o1 = {0};
#pragma unroll
        for (int kb = 0; kb < N; ++kb) {
            f32x16 st = {0};
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][sx], qf[sx], st, 0, 0, 0);
            // st[r] = score(key kb*32 + 16 (r>>3) + 8 g + (r&7), query qb0 + i), log2 domain
            if (kb == N - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + 16 * (r >> 3) + 8 * g + (r & 7);
                    st[r] = key < T ? st[r] : -INFINITY;
                }
            }
            mfma_result_pad(st);
            float bm = max3f(st[0], st[1], st[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) bm = max3f(bm, st[r], st[r + 1]);
            bm = fmaxf(bm, st[15]);
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            float m_new = bm;  // finite: every block holds at least one real key
            if (kb > 0) {
                m_new = fmaxf(m_run, bm);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            m_run = m_new;
            float p[16];
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
                psum += p[r];
            }
            l_run += psum;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (_Float16)p[8 * u + j];
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kb][u][0], pf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kb][u][1], pf, o1, 0, 0, 0);
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
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
        if (more) {
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) qf[sx] = qn[sx];
        }
    }
}

__global__ void __launch_bounds__(ATT_THREADS, 2) attention_reg_kernel(const AttnArgs A, int n_units) {
    const int unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (unit >= n_units) return;
    const int s = unit / A.n_heads, h = unit - s * A.n_heads;
    const int tok0 = A.seq_off[s];
    const int T = A.seq_off[s + 1] - tok0;  // 1..128
    const int vcol0 = A.seq_vtcol[s];
    const int l = threadIdx.x & 63;
    switch ((T + 31) >> 5) {
        case 1: attend_reg<1>(A, s, h, tok0, T, vcol0, l); break;
        case 2: attend_reg<2>(A, s, h, tok0, T, vcol0, l); break;
        case 3: attend_reg<3>(A, s, h, tok0, T, vcol0, l); break;
        default: attend_reg<4>(A, s, h, tok0, T, vcol0, l); break;
    }
}

}  // namespace

size_t attention_lds_bytes(int max_seq_len) {
    const int Tk = (max_seq_len + 31) & ~31;
    return (size_t)Tk * HD * 2 + (size_t)HD * (Tk + 4) * 2;
}

int launch_attention(const AttnArgs &A, int n_seq, int max_seq_len, hipStream_t st) {
    if (n_seq <= 0) return ANCE_OK;
    static const bool use_reg = [] {  // ANCE_ATTN_REG=0: the LDS kernel for every length (A/B)
        const char *e = getenv("ANCE_ATTN_REG");
        return !(e && e[0] == '0');
    }();
    if (use_reg && max_seq_len <= 128) {
        const int n_units = n_seq * A.n_heads;
        hipLaunchKernelGGL(attention_reg_kernel, dim3((unsigned)((n_units + 3) / 4)), dim3(ATT_THREADS), 0, st, A, n_units);
        return ANCE_OK;
    }
    const size_t lds = attention_lds_bytes(max_seq_len);
    if (lds > 160 * 1024) {
        set_last_error("attention: sequence too long for LDS");
        return ANCE_E_INVALID;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    static size_t attr_set[64] = {0};  // the attribute is per device
    if (dev < 0 || dev >= 64) dev = 0;
    if (lds > attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return check_launch("attention attr");
        attr_set[dev] = lds;
    }
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)n_seq * A.n_heads), dim3(ATT_THREADS), lds, st, A);
    return ANCE_OK;
}

}  // namespace ance
