// Exact inner-product top-k search for gfx950 (MI355X) -- replaces faiss.IndexFlatIP.add/search
// on the reference's hot path (drivers/run_ann_data_gen.py:269-276,303).
//
// Kernel 1  ip_topk_scan   : fp32 MFMA (v_mfma_f32_32x32x2_f32) tile product S^T = X . Q^T with a
//                            fused threshold filter + per-query candidate buffers + in-kernel
//                            radix-select prune.  Rows of the MFMA tile are passages, columns are
//                            queries, so one lane owns one query column: the running threshold is a
//                            single register and the filter is one compare per score.
// Kernel 2  topk_finalize  : per query, bitonic sort of the surviving candidates (all corpus
//                            splits) in LDS under the canonical order, decode to (D, I).
//
// Numerics: every score is an fp32 fmaf chain over k ascending (lanes 0-31 of the MFMA carry the
// even k, lanes 32-63 the odd k; the instruction accumulates k0 then k1), bit-identical to
// oracle/ip_topk_ref.c.  Ids are exact under (score desc, row asc): corpus rows are scanned in
// ascending order per split, so "s > threshold" (strict) is the correct admission test.
#include "common.h"
#include "topk_common.h"
#include <stdlib.h>
#include <string.h>

namespace ance {
namespace {

constexpr int TP = 128;       // passages per tile (MFMA rows)
constexpr int TQ = 128;       // queries per tile  (MFMA cols)
constexpr int BK = 32;        // k per LDS stage
constexpr int ROWF = BK + 4;  // floats per LDS row: 16 even-k | 16 odd-k | 4 pad  (144 B)
constexpr int STAGE_FLOATS = (TP + TQ) * ROWF;
constexpr int SCAN_THREADS = 256;
constexpr size_t SCAN_LDS_BYTES = (size_t)(2 * STAGE_FLOATS + TQ + TQ) * 4;  // 2 stages + tau + cnt

struct ScanParams {
    const float *x;
    const float *q;
    uint32_t n;        // rows in this shard
    uint32_t nq;       // queries in this launch
    int d;
    int k;
    int S;             // corpus splits (power of two, <= 64)
    int n_qt;          // query tiles in this launch
    int n_tiles_p;     // passage tiles in the shard
    int tiles_per_split;
    u64 *cand;         // [n_qt * S][TQ][C]
    u64 *part;         // [nq][S][k]
    const int *only_if;  // optional device flag: the whole launch is a no-op while it reads 0
    const int *nq_dev;   // optional device count: only the first min(nq, *nq_dev) queries are searched
};

template <int NPL>
__global__ void __launch_bounds__(SCAN_THREADS, 2) ip_topk_scan_kernel(const ScanParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int C = NPL * 64;
    float *tau_s = smem + 2 * STAGE_FLOATS;
    int *cnt_s = reinterpret_cast<int *>(tau_s + TQ);

    // ---- block -> (query tile, corpus split); XCD-aware grouping (speed only) -----------------
    // Blocks b, b+8, b+16, ... land on the same XCD.  64 consecutive blocks of one XCD form a
    // group of GQ = 64/S query tiles x S splits: they stream the same corpus ranges at the same
    // time (X tiles shared through that XCD's L2) and keep only GQ query tiles hot.
    if (P.only_if && *P.only_if == 0) return;
    uint32_t nq = P.nq;
    if (P.nq_dev) nq = min(nq, (uint32_t)max(*P.nq_dev, 0));
    if (nq == 0) return;
    const int n_qt = (int)((nq + TQ - 1) / TQ);
    const int b = blockIdx.x;
    const int xcd = b & 7, jx = b >> 3;
    const int gq = 64 / P.S;
    const int grp = (jx >> 6) * 8 + xcd;
    const int r64 = jx & 63;
    const int qt = grp * gq + r64 / P.S;
    const int split = r64 % P.S;
    if (qt >= n_qt) return;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, g = l >> 5, c = l & 31;
    const int wp = w >> 1, wq = w & 1;
    const uint32_t q0 = (uint32_t)qt * TQ;
    const int t0 = split * P.tiles_per_split;
    const int t1 = min(t0 + P.tiles_per_split, P.n_tiles_p);
    u64 *cand = P.cand + ((size_t)qt * P.S + split) * (size_t)TQ * C;

    if (tid < TQ) {
        tau_s[tid] = -INFINITY;
        cnt_s[tid] = 0;
    }

    // ---- staging geometry: thread handles 4 float4 of X and 4 of Q per stage -------------------
    // element e = tid + 256 j : row = e >> 3, float4 column = e & 7  (8 lanes = one 128 B line)
    const int d = P.d;
    const int NS = (d + BK - 1) / BK;
    const int c4 = tid & 7;
    const int row_base_t = tid >> 3;  // + 32 j
    // Loads are unconditional (clamped addresses) and masked when written to LDS: a guarded
    // load would make hipcc branch around -- and wait for -- every single load.
    const float *qrow[4];
    unsigned qmask = 0;  // bit j: query row j of this thread is real
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t qr = q0 + row_base_t + 32 * j;
        if (qr < nq) qmask |= 1u << j;
        qrow[j] = P.q + (size_t)min(qr, nq - 1) * d;
    }
    const int lds_w_even = c4 * 2;        // float offset of {k0,k2} inside the row
    const int lds_w_odd = 16 + c4 * 2;    // float offset of {k1,k3}

    f32x4 rx[4], rq[4];
    unsigned rmask = 0;  // bits 0-3: X row valid, bits 4-7: Q row valid (k-range folded in)
    auto load_regs = [&](int tile, int stage) {
        const int kb = stage * BK + c4 * 4;
        const bool kvalid = kb < d;
        const int kbc = kvalid ? kb : 0;
        const uint32_t p0 = (uint32_t)tile * TP;
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t pr = p0 + row_base_t + 32 * j;
            if (kvalid && pr < P.n) m |= 1u << j;
            rx[j] = *reinterpret_cast<const f32x4 *>(P.x + (size_t)min(pr, P.n - 1) * d + kbc);
            rq[j] = *reinterpret_cast<const f32x4 *>(qrow[j] + kbc);
        }
        if (kvalid) m |= qmask << 4;
        rmask = m;
    };
    auto write_lds = [&](int buf) {
        float *st = smem + buf * STAGE_FLOATS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = row_base_t + 32 * j;
            float *px = st + row * ROWF;
            float *pq = st + (TP + row) * ROWF;
            const float mx = (rmask >> j) & 1u ? 1.0f : 0.0f;
            const float mq = (rmask >> (4 + j)) & 1u ? 1.0f : 0.0f;
            const f32x4 vx = mx != 0.0f ? rx[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 vq = mq != 0.0f ? rq[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<float2 *>(px + lds_w_even) = make_float2(vx[0], vx[2]);
            *reinterpret_cast<float2 *>(px + lds_w_odd) = make_float2(vx[1], vx[3]);
            *reinterpret_cast<float2 *>(pq + lds_w_even) = make_float2(vq[0], vq[2]);
            *reinterpret_cast<float2 *>(pq + lds_w_odd) = make_float2(vq[1], vq[3]);
        }
    };

    // fragment read offsets (floats): lane (c, g) reads row c of its 32-row block, even-k half for
    // g = 0 / odd-k half for g = 1, 4 consecutive k-pairs per ds_read_b128.
    const int xoff0 = (wp * 64 + c) * ROWF + g * 16;
    const int xoff1 = xoff0 + 32 * ROWF;
    const int qoff0 = (TP + wq * 64 + c) * ROWF + g * 16;
    const int qoff1 = qoff0 + 32 * ROWF;

    int buf = 0;
    if (t0 < t1) load_regs(t0, 0);
    for (int t = t0; t < t1; ++t) {
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        for (int s = 0; s < NS; ++s) {
            write_lds(buf);
            __syncthreads();
            if (s + 1 < NS) load_regs(t, s + 1);
            else if (t + 1 < t1) load_regs(t + 1, 0);
            const float *st = smem + buf * STAGE_FLOATS;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(st + xoff0 + kk * 4);
                const f32x4 a1 = *reinterpret_cast<const f32x4 *>(st + xoff1 + kk * 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(st + qoff0 + kk * 4);
                const f32x4 b1 = *reinterpret_cast<const f32x4 *>(st + qoff1 + kk * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc11, 0, 0, 0);
                }
            }
            buf ^= 1;
        }

        // ---- fused top-k epilogue: threshold filter, rare append -----------------------------
        const uint32_t p0 = (uint32_t)t * TP + wp * 64 + 4 * g;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int ql = wq * 64 + qb * 32 + c;
            const bool qv = (q0 + ql) < nq;
            const float tau = tau_s[ql];
            u64 *cq = cand + (size_t)ql * C;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const f32x16 &a = pb == 0 ? (qb == 0 ? acc00 : acc01) : (qb == 0 ? acc10 : acc11);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t prow = p0 + pb * 32 + (r & 3) + 8 * (r >> 2);
                    const float sc = a[r];
                    if (qv && prow < P.n && sc > tau) {
                        const int slot = atomicAdd(&cnt_s[ql], 1);
                        cq[slot] = pack_key(sc, prow);
                    }
                }
            }
        }
        __syncthreads();
        // ---- prune queries whose buffer could overflow on the next tile -----------------------
        for (int i = 0; i < 32; ++i) {
            const int ql = w * 32 + i;
            const int n_c = __builtin_amdgcn_readfirstlane(cnt_s[ql]);
            if (n_c > C - TP) {
                u64 *cq = cand + (size_t)ql * C;
                float tau_new;
                const int kept = select_topk<NPL>(cq, n_c, P.k, cq, &tau_new);
                if (l == 0) {
                    cnt_s[ql] = kept;
                    tau_s[ql] = tau_new;
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- write this split's survivors (unsorted, <= k per query, 0-padded) ----------------------
    for (int i = 0; i < 32; ++i) {
        const int ql = w * 32 + i;
        const uint32_t qg = q0 + ql;
        if (qg >= nq) continue;  // wave-uniform
        const int n_c = __builtin_amdgcn_readfirstlane(cnt_s[ql]);
        const u64 *cq = cand + (size_t)ql * C;
        u64 *dst = P.part + ((size_t)qg * P.S + split) * (size_t)P.k;
        if (n_c > P.k) {
            float tau_new;
            select_topk<NPL>(cq, n_c, P.k, dst, &tau_new);
        } else {
            for (int e = l; e < P.k; e += 64) dst[e] = (e < n_c) ? cq[e] : 0ull;
        }
    }
}

struct Plan {
    int npl;        // candidate buffer = 64 * npl entries per query
    int S;          // corpus splits
    int64_t qc;     // queries per launch
    int n_tiles_p;
    int tiles_per_split;
    size_t cand_bytes, part_bytes;
};

bool make_plan(int64_t n, int64_t nq, int k, Plan *pl) {
    if (k < 1 || k > ANCE_TOPK_MAX_K || n < 0 || n >= (1ll << 32) || nq < 0) return false;
    pl->npl = k <= 256 ? 8 : (k <= 768 ? 16 : 32);
    const int C = pl->npl * 64;
    pl->n_tiles_p = (int)((n + TP - 1) / TP);
    const int64_t nqt = (nq + TQ - 1) / TQ;
    const int64_t qct = nqt < 512 ? (nqt > 0 ? nqt : 1) : 512;
    pl->qc = qct * TQ;
    int S = 1;
    while (qct * S < 1024 && S < 64) S <<= 1;
    if (S < 8) S = 8;
    while (S > 1 && (S * 4 > pl->n_tiles_p || next_pow2(S * k) > 8192)) S >>= 1;
    pl->S = S;
    pl->tiles_per_split = pl->n_tiles_p > 0 ? (pl->n_tiles_p + S - 1) / S : 0;
    pl->cand_bytes = (size_t)qct * S * TQ * C * sizeof(u64);
    pl->part_bytes = align_up((size_t)pl->qc * S * k * sizeof(u64), 256);
    return true;
}

}  // namespace

int launch_finalize_keys(const u64 *keys, int64_t nq, int m, int k, int64_t row_base, float *out_d, int64_t *out_i,
                         hipStream_t st, const FinalizeAlt *alt) {
    FinalizeAlt a;
    if (alt) a = *alt;
    int widest = m + (a.dd ? DEDUP_MAXC * k : 0);
    if (a.all_m > widest) widest = a.all_m;
    if (a.slot_m > widest) widest = a.slot_m;
    const int P2 = next_pow2(widest);
    if (P2 > 8192) {
        set_last_error("topk_finalize: more than 8192 survivors per query");
        return ANCE_E_INVALID;
    }
    static unsigned long long attr_done = 0;  // per device, for the largest list this kernel takes
    if (attr_needed(&attr_done)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(topk_finalize_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8192 * sizeof(u64))) != hipSuccess)
            return check_launch("topk_finalize attr");
        attr_mark(&attr_done);
    }
    ProfScope pf(PC_FINALIZE, st);
    hipLaunchKernelGGL(topk_finalize_kernel<false>, dim3((unsigned)nq), dim3(256), P2 * sizeof(u64), st, keys,
                       (const float *)nullptr, (const int64_t *)nullptr, 1, nq, m, P2, k, row_base, out_d, out_i, a);
    return ANCE_OK;
}

int launch_reduce_keys(const u64 *keys, int nq_max, int m, int k, u64 *out, const int *nq_dev, hipStream_t st) {
    const int P2 = next_pow2(m > k ? m : k);
    if (P2 > 8192) {
        set_last_error("topk_reduce_keys: more than 8192 keys per query");
        return ANCE_E_INVALID;
    }
    static unsigned long long attr_done = 0;
    if (attr_needed(&attr_done)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(topk_reduce_keys_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(8192 * sizeof(u64))) != hipSuccess)
            return check_launch("topk_reduce_keys attr");
        attr_mark(&attr_done);
    }
    hipLaunchKernelGGL(topk_reduce_keys_kernel, dim3((unsigned)nq_max), dim3(256), P2 * sizeof(u64), st, keys, m, P2, k, out, nq_dev);
    return ANCE_OK;
}

namespace {
int launch_scan(const Plan &pl, const float *d_x, int64_t n, const float *q, int64_t nqc, int d, int k, u64 *cand, u64 *part,
                const int *only_if, const int *nq_dev, hipStream_t st) {
    auto scan = pl.npl == 8 ? ip_topk_scan_kernel<8> : (pl.npl == 16 ? ip_topk_scan_kernel<16> : ip_topk_scan_kernel<32>);
    static unsigned long long attr_done[3] = {0, 0, 0};
    const int ai = pl.npl == 8 ? 0 : (pl.npl == 16 ? 1 : 2);
    if (attr_needed(&attr_done[ai])) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(scan), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)SCAN_LDS_BYTES) != hipSuccess)
            return check_launch("ip_topk_scan attr");
        attr_mark(&attr_done[ai]);
    }
    ScanParams P;
    P.x = d_x; P.q = q; P.n = (uint32_t)n; P.nq = (uint32_t)nqc; P.d = d; P.k = k; P.S = pl.S;
    P.n_qt = (int)((nqc + TQ - 1) / TQ); P.n_tiles_p = pl.n_tiles_p; P.tiles_per_split = pl.tiles_per_split;
    P.cand = cand; P.part = part; P.only_if = only_if; P.nq_dev = nq_dev;
    const int gq = 64 / pl.S;
    const int groups = (P.n_qt + gq - 1) / gq;
    const unsigned blocks = (unsigned)((groups + 7) / 8 * 8) * 64u;
    if (only_if || nq_dev) {  // device-side conditional redo of the fast path: normally a no-op, kept out of the profile
        hipLaunchKernelGGL(scan, dim3(blocks), dim3(SCAN_THREADS), SCAN_LDS_BYTES, st, P);
        return ANCE_OK;
    }
    ProfScope ps(PC_SCAN, st, 2.0 * (double)nqc * (double)n * (double)d);
    hipLaunchKernelGGL(scan, dim3(blocks), dim3(SCAN_THREADS), SCAN_LDS_BYTES, st, P);
    return ANCE_OK;
}
}  // namespace

size_t exact_scan_fallback_bytes(int64_t n, int64_t nq, int k) {
    Plan pl;
    if (!make_plan(n, nq, k, &pl) || nq > pl.qc) return 0;
    return pl.part_bytes + pl.cand_bytes + 256;
}

int exact_scan_fallback(const float *d_x, int64_t n, const float *d_q, int64_t nq, int64_t nq_plan, int d, int k, void *d_ws,
                        const int *only_if, const int *nq_dev, const u64 **part_out, int *m_out, hipStream_t st) {
    // The plan (corpus splits, buffer layout) is the one the workspace was sized for -- nq_plan queries -- whatever the
    // size of this chunk: a short last chunk planned on its own would choose more splits and outgrow the buffers.
    Plan pl;
    if (!make_plan(n, nq_plan, k, &pl) || nq_plan > pl.qc || nq > nq_plan) {
        set_last_error("exact_scan_fallback: chunk too large");
        return ANCE_E_INVALID;
    }
    u64 *part = reinterpret_cast<u64 *>(align_up((uintptr_t)d_ws, 256));
    u64 *cand = reinterpret_cast<u64 *>((char *)part + pl.part_bytes);
    *part_out = part;
    *m_out = pl.S * k;
    return launch_scan(pl, d_x, n, d_q, nq, d, k, cand, part, only_if, nq_dev, st);
}

}  // namespace ance

using namespace ance;

namespace ance {
size_t ip_index_bytes(int64_t n, int d);
int ip_index_build(const float *d_x, int64_t n, int d, void *d_index, size_t index_bytes, hipStream_t st);
size_t ip_topk_fast_workspace_bytes(int64_t n, int64_t nq, int d, int k, bool with_index);
int ip_topk_fast(const float *d_x, int64_t n, int64_t row_base, const void *d_index, const float *d_q, int64_t nq, int d, int k,
                 float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, hipStream_t st);
void set_fast_stamps(unsigned long long *d_stamps);
void reload_fast_knobs();
void reload_gemm_knobs();
}

extern "C" void ance_debug_search_stamps(void *d_stamps) { ance::set_fast_stamps(reinterpret_cast<unsigned long long *>(d_stamps)); }

// ANCE_SEARCH=exact forces the fp32-MFMA scan everywhere (A/B and cross-checks); default: the
// two-precision path whenever the shape is eligible (d % 128 == 0, d <= 2048, k <= 1024, n >= 4096).
static int g_search_exact = -1;  // ANCE_SEARCH, read once (ance_reload_env re-reads)
static bool fast_enabled() {
    if (g_search_exact < 0) {
        const char *e = getenv("ANCE_SEARCH");
        g_search_exact = (e && !strcmp(e, "exact")) ? 1 : 0;
    }
    return g_search_exact == 0;
}

extern "C" void ance_reload_env(void) {
    g_search_exact = -1;
    ance::reload_fast_knobs();
    ance::reload_gemm_knobs();
}

static size_t scan_workspace_bytes(int64_t n, int64_t nq, int k) {
    Plan pl;
    if (!make_plan(n, nq, k, &pl)) return 0;
    return pl.part_bytes + pl.cand_bytes + 256;
}

extern "C" size_t ance_ip_topk_workspace_bytes(int64_t n, int64_t nq, int d, int k) {
    size_t need = scan_workspace_bytes(n, nq, k);
    if (need && fast_enabled()) {
        const size_t f = ip_topk_fast_workspace_bytes(n, nq, d, k, true);
        if (f > need) need = f;
    }
    return need;
}

static int ip_topk_exact_scan(const float *d_x, int64_t n, int64_t row_base, const float *d_q, int64_t nq, int d, int k,
                              float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, void *stream) {
    Plan pl;
    if (!make_plan(n, nq, k, &pl) || d < 4 || (d & 3) || !d_out_d || !d_out_i || (nq > 0 && !d_q) || (n > 0 && !d_x) ||
        ((uintptr_t)d_x & 15) || ((uintptr_t)d_q & 15)) {
        set_last_error("ance_ip_topk: invalid argument");
        return ANCE_E_INVALID;
    }
    if (nq == 0) return ANCE_OK;
    if (!d_workspace || workspace_bytes < pl.part_bytes + pl.cand_bytes) {
        set_last_error("ance_ip_topk: workspace too small");
        return ANCE_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    u64 *part = reinterpret_cast<u64 *>(align_up((uintptr_t)d_workspace, 256));
    u64 *cand = reinterpret_cast<u64 *>((char *)part + pl.part_bytes);

    const int m = pl.S * k;
    for (int64_t q0 = 0; q0 < nq; q0 += pl.qc) {
        const int64_t nqc = (nq - q0) < pl.qc ? (nq - q0) : pl.qc;
        int rc = launch_scan(pl, d_x, n, d_q + (size_t)q0 * d, nqc, d, k, cand, part, nullptr, nullptr, st);
        if (rc) return rc;
        rc = launch_finalize_keys(part, nqc, m, k, row_base, d_out_d + (size_t)q0 * k, d_out_i + (size_t)q0 * k, st);
        if (rc) return rc;
    }
    return check_launch("ance_ip_topk");
}

static bool fast_args_ok(const float *d_x, const float *d_q, int64_t nq, float *d_out_d, int64_t *d_out_i, void *d_workspace) {
    return nq > 0 && d_x && d_q && d_out_d && d_out_i && d_workspace && !((uintptr_t)d_x & 15) && !((uintptr_t)d_q & 15);
}

extern "C" int ance_ip_topk(const float *d_x, int64_t n, int64_t row_base, const float *d_q, int64_t nq, int d, int k,
                            float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, void *stream) {
    if (fast_enabled() && fast_args_ok(d_x, d_q, nq, d_out_d, d_out_i, d_workspace) && ip_topk_fast_workspace_bytes(n, nq, d, k, true) > 0)
        return ip_topk_fast(d_x, n, row_base, nullptr, d_q, nq, d, k, d_out_d, d_out_i, d_workspace, workspace_bytes,
                            (hipStream_t)stream);
    return ip_topk_exact_scan(d_x, n, row_base, d_q, nq, d, k, d_out_d, d_out_i, d_workspace, workspace_bytes, stream);
}

// the fp32-MFMA scan alone, whatever the shape and the environment: the audit path of the two-precision kernel
extern "C" size_t ance_ip_topk_scan_workspace_bytes(int64_t n, int64_t nq, int d, int k) {
    (void)d;
    return scan_workspace_bytes(n, nq, k);
}
extern "C" int ance_ip_topk_scan(const float *d_x, int64_t n, int64_t row_base, const float *d_q, int64_t nq, int d, int k,
                                 float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, void *stream) {
    return ip_topk_exact_scan(d_x, n, row_base, d_q, nq, d, k, d_out_d, d_out_i, d_workspace, workspace_bytes, stream);
}

extern "C" size_t ance_ip_index_bytes(int64_t n, int d) { return fast_enabled() ? ip_index_bytes(n, d) : 0; }

extern "C" int ance_ip_index_build(const float *d_x, int64_t n, int d, void *d_index, size_t index_bytes, void *stream) {
    return ip_index_build(d_x, n, d, d_index, index_bytes, (hipStream_t)stream);
}

extern "C" size_t ance_ip_topk_indexed_workspace_bytes(int64_t n, int64_t nq, int d, int k) {
    size_t need = scan_workspace_bytes(n, nq, k);
    if (need && fast_enabled()) {
        const size_t f = ip_topk_fast_workspace_bytes(n, nq, d, k, false);
        if (f > need) need = f;
    }
    return need;
}

extern "C" int ance_ip_topk_indexed(const float *d_x, int64_t n, int64_t row_base, const void *d_index, const float *d_q, int64_t nq,
                                    int d, int k, float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes,
                                    void *stream) {
    if (d_index && fast_enabled() && fast_args_ok(d_x, d_q, nq, d_out_d, d_out_i, d_workspace) &&
        ip_topk_fast_workspace_bytes(n, nq, d, k, false) > 0)
        return ip_topk_fast(d_x, n, row_base, d_index, d_q, nq, d, k, d_out_d, d_out_i, d_workspace, workspace_bytes,
                            (hipStream_t)stream);
    if (!d_index) return ance_ip_topk(d_x, n, row_base, d_q, nq, d, k, d_out_d, d_out_i, d_workspace, workspace_bytes, stream);
    return ip_topk_exact_scan(d_x, n, row_base, d_q, nq, d, k, d_out_d, d_out_i, d_workspace, workspace_bytes, stream);
}

extern "C" size_t ance_topk_merge_workspace_bytes(int n_parts, int64_t nq, int k) {
    (void)n_parts; (void)nq; (void)k;
    return 256;  // the merge sorts in LDS; kept in the ABI so the contract can grow
}

extern "C" int ance_topk_merge(const float *d_parts_d, const int64_t *d_parts_i, int n_parts, int64_t nq, int k,
                               float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, void *stream) {
    (void)d_workspace; (void)workspace_bytes;
    if (n_parts < 1 || k < 1 || nq < 0 || !d_parts_d || !d_parts_i || !d_out_d || !d_out_i) {
        set_last_error("ance_topk_merge: invalid argument");
        return ANCE_E_INVALID;
    }
    const int64_t m64 = (int64_t)n_parts * k;
    if (m64 > 16384) {
        set_last_error("ance_topk_merge: n_parts * k > 16384 unsupported");
        return ANCE_E_INVALID;
    }
    if (nq == 0) return ANCE_OK;
    const int m = (int)m64, P2 = next_pow2(m);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(topk_finalize_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(P2 * sizeof(u64))) != hipSuccess)
        return check_launch("topk_merge attr");
    ProfScope pf(PC_FINALIZE, (hipStream_t)stream);
    hipLaunchKernelGGL(topk_finalize_kernel<true>, dim3((unsigned)nq), dim3(256), P2 * sizeof(u64), (hipStream_t)stream,
                       (const u64 *)nullptr, d_parts_d, d_parts_i, n_parts, nq, m, P2, k, (int64_t)0, d_out_d, d_out_i,
                       FinalizeAlt());
    return check_launch("ance_topk_merge");
}

// ---- restricted-candidate scoring (include/ance_amd.h: ance_ip_score_rows) -------------------------
namespace ance {
namespace {

// One workgroup per query: the query row is staged in LDS, each thread walks candidates j = tid, tid +
// 256, ... and runs the canonical chain s = fmaf(q[k], x[k], s), k ascending (the scan's arithmetic).
__global__ void __launch_bounds__(256) score_rows_kernel(const float *x, int64_t n, const float *q, int d, const int64_t *rows,
                                                         const int64_t *offsets, float *scores) {
    extern __shared__ __attribute__((aligned(16))) float qs[];
    const int64_t qi = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += 256) qs[k] = q[qi * d + k];
    __syncthreads();
    const int64_t j0 = offsets[qi], j1 = offsets[qi + 1];
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) {
        const int64_t r = rows[j];
        if (r < 0 || r >= n) {  // documented: an out-of-range candidate scores -inf
            scores[j] = -INFINITY;
            continue;
        }
        const float *xr = x + r * d;
        float s = 0.0f;
        for (int k = 0; k < d; k += 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + k);
            const f32x4 qv = *reinterpret_cast<const f32x4 *>(qs + k);
            s = __builtin_fmaf(qv[0], xv[0], s);
            s = __builtin_fmaf(qv[1], xv[1], s);
            s = __builtin_fmaf(qv[2], xv[2], s);
            s = __builtin_fmaf(qv[3], xv[3], s);
        }
        scores[j] = s;
    }
}

}  // namespace
}  // namespace ance

extern "C" int ance_ip_score_rows(const float *d_x, int64_t n, const float *d_q, int64_t nq, int d, const int64_t *d_rows,
                                  const int64_t *d_offsets, float *d_scores, void *stream) {
    using namespace ance;
    if (nq < 0 || n < 0 || d < 4 || d % 4 || d > 16384 || (nq > 0 && (!d_q || !d_offsets)) || (n > 0 && !d_x)) {
        set_last_error("ance_ip_score_rows: invalid argument (d must be a multiple of 4)");
        return ANCE_E_INVALID;
    }
    if (nq == 0) return ANCE_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(score_rows_kernel, dim3((unsigned)nq), dim3(256), (size_t)d * sizeof(float), st, d_x, n, d_q, d, d_rows,
                       d_offsets, d_scores);
    int rc = check_launch("ance_ip_score_rows");
    if (rc) return rc;
    return ANCE_OK;
}
