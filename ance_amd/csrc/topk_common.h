// Device code shared by the exact-scan (ip_topk.hip) and the split-precision (ip_topk_fast.hip)
// searches: wave-level exact k-th selection over packed keys, LDS bitonic sort, list finalisation.
#pragma once
#include "common.h"

namespace ance {

// k-th largest selection + compaction of one query's candidate list, by one wave.
// keys are distinct (distinct rows), 0 is the empty sentinel.
template <int NPL>
__device__ __forceinline__ int select_topk_regs(u64 (&keys)[NPL], int k, u64 *dst, float *tau_out) {
    const int l = lane_id();
    u64 T = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const u64 t2 = T | (1ull << bit);
        int ge = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) ge += __popcll(__ballot(keys[j] >= t2));
        if (ge >= k) T = t2;
    }
    int base = 0;
    const u64 lt_mask = (1ull << l) - 1ull;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const bool pr = keys[j] >= T;
        const u64 m = __ballot(pr);
        const int pos = base + __popcll(m & lt_mask);
        if (pr) dst[pos] = keys[j];
        base += __popcll(m);
    }
    *tau_out = key_score(T);
    return base;
}

template <int NPL>
__device__ __forceinline__ int select_topk(const u64 *src, int n_c, int k, u64 *dst, float *tau_out) {
    const int l = lane_id();
    u64 keys[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int idx = j * 64 + l;
        keys[j] = (idx < n_c) ? src[idx] : 0ull;
    }
    return select_topk_regs<NPL>(keys, k, dst, tau_out);
}

// ------------------------------------------------------------------------------------------------
// Bitonic sort (descending) of P2 keys in LDS by one 256-thread block.
__device__ __forceinline__ void bitonic_sort_desc(u64 *s, int P2) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int size = 2; size <= P2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P2 >> 1); i += nt) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 a = s[lo], b2 = s[hi];
                if ((a < b2) == desc) {
                    s[lo] = b2;
                    s[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

// ---- duplicate classes of a search image (ip_topk_fast.hip) ----------------------------------------------
constexpr int DEDUP_MAXC = 4;       // classes of bit-identical rows collapsed per shard
constexpr int DEDUP_MEMCAP = 1024;  // member ids kept per class (the smallest ones: no list needs more than k)
struct DedupHeader {                // first 256 bytes of a search image
    unsigned int xmax_bits;         // max norm of the CENTRED rows x - mu the fp16 image holds (float bits; +inf when not finite)
    unsigned int n_live;            // rows of the image (duplicates collapsed)
    int n_classes;
    unsigned int xmax_orig_bits;    // max norm of the rows themselves (the exact chain's own rounding scales with it)
    unsigned int guess[DEDUP_MAXC]; // sampled row that defines the class
    unsigned int rep[DEDUP_MAXC];   // smallest row id of the class: stays in the image
    unsigned int csize[DEDUP_MAXC]; // members besides the representative
    // stamp of a COMPLETED build: which matrix this image belongs to.  A search whose (n, d, rows pointer) do not match --
    // an image of another shard, a buffer that was never built -- is answered by the exact scan instead of gathering rows
    // through someone else's live2row (include/ance_amd.h: ance_ip_topk_indexed)
    unsigned int magic, d;
    unsigned long long n, x_ptr;
};
constexpr unsigned int DEDUP_MAGIC = 0x414E4345u;  // "ANCE"

// where topk_finalize takes a query's survivors from when the fast path handed the query (or its whole launch
// chunk) to the exact scan, and the duplicate classes to expand otherwise
struct FinalizeAlt {
    const int *sel_all = nullptr;   // != 0: every query of the chunk comes from all_keys [nq][all_m]
    const u64 *all_keys = nullptr;
    int all_m = 0;
    const int *slot = nullptr;      // [nq] >= 0: this query comes from slot_keys [slot][slot_m]
    const u64 *slot_keys = nullptr;
    int slot_m = 0;
    const DedupHeader *dd = nullptr;  // duplicate classes of the shard (survivors of the fast path only)
    const uint32_t *members = nullptr;  // [DEDUP_MAXC][DEDUP_MEMCAP] ascending member ids
};

// FROM_DI = false: entries are packed keys [nq][m]; true: entries are (D, I) parts [n_parts][nq][k]
template <bool FROM_DI>
__global__ void __launch_bounds__(256) topk_finalize_kernel(const u64 *keys, const float *pd, const int64_t *pi,
                                                            int n_parts, int64_t nq, int m, int P2, int k,
                                                            int64_t row_base, float *out_d, int64_t *out_i,
                                                            const FinalizeAlt alt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u64 *s = reinterpret_cast<u64 *>(smem);
    __shared__ unsigned int rep_hi[DEDUP_MAXC];  // ordered score bits of a class representative found among the survivors
    const int64_t qi = blockIdx.x;
    bool expand = false;
    if constexpr (!FROM_DI) {
        keys += (size_t)qi * m;
        expand = alt.dd && alt.dd->n_classes > 0;
        if (alt.sel_all && *alt.sel_all) {  // the launch chunk was redone by the exact scan: take its survivors
            keys = alt.all_keys + (size_t)qi * alt.all_m;
            m = alt.all_m;
            expand = false;  // the scan saw every row of the shard
        } else if (alt.slot && alt.slot[qi] >= 0) {  // this query was redone by the exact scan
            keys = alt.slot_keys + (size_t)alt.slot[qi] * alt.slot_m;
            m = alt.slot_m;
            expand = false;
        }
    }
    const int nc = expand ? alt.dd->n_classes : 0;
    const int m_all = m + nc * k;
    int p2 = 1;
    while (p2 < m_all) p2 <<= 1;  // <= P2 by construction of the launch
    if (threadIdx.x < DEDUP_MAXC) rep_hi[threadIdx.x] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        u64 v = 0ull;
        if constexpr (FROM_DI) {
            const int p = i / k, r = i - p * k;
            const size_t o = ((size_t)p * nq + qi) * k + r;
            const int64_t id = pi[o];
            if (id >= 0) v = pack_key(pd[o], (uint32_t)id);
        } else {
            v = keys[i];
            if (v != 0ull)
                for (int c = 0; c < nc; ++c)
                    if (key_row(v) == alt.dd->rep[c]) rep_hi[c] = (unsigned int)(v >> 32);
        }
        s[i] = v;
    }
    __syncthreads();
    // a class whose representative survived: its members tie with it and follow it in ascending id order; the
    // first k of them are all that can enter a top-k
    for (int i = m + threadIdx.x; i < p2; i += blockDim.x) {
        u64 v = 0ull;
        if (i < m_all) {
            const int c = (i - m) / k, r = (i - m) - c * k;
            if (rep_hi[c] != 0u && (unsigned int)r < alt.dd->csize[c] && r < DEDUP_MEMCAP)
                v = ((u64)rep_hi[c] << 32) | (u64)(0xFFFFFFFFu - alt.members[(size_t)c * DEDUP_MEMCAP + r]);
        }
        s[i] = v;
    }
    __syncthreads();
    bitonic_sort_desc(s, p2);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const u64 v = (i < p2) ? s[i] : 0ull;
        const size_t o = (size_t)qi * k + i;
        if (v == 0ull) {
            out_d[o] = -FLT_MAX;
            out_i[o] = -1;
        } else {
            out_d[o] = key_score(v);
            out_i[o] = row_base + (int64_t)key_row(v);
        }
    }
    (void)P2;
}

// top-k keys of nq lists of m keys each (unsorted in, sorted out): the per-query exact-scan lists of the fast path
static __global__ void __launch_bounds__(256) topk_reduce_keys_kernel(const u64 *keys, int m, int P2, int k, u64 *out, const int *nq_dev) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u64 *s = reinterpret_cast<u64 *>(smem);
    if (nq_dev && (int)blockIdx.x >= *nq_dev) return;
    for (int i = threadIdx.x; i < P2; i += blockDim.x) s[i] = i < m ? keys[(size_t)blockIdx.x * m + i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(s, P2);
    for (int i = threadIdx.x; i < k; i += blockDim.x) out[(size_t)blockIdx.x * k + i] = i < P2 ? s[i] : 0ull;
}


inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// launches topk_finalize_kernel<false> over nq queries whose m = S*k survivors are packed keys
int launch_finalize_keys(const u64 *keys, int64_t nq, int m, int k, int64_t row_base, float *out_d, int64_t *out_i,
                         hipStream_t st, const FinalizeAlt *alt = nullptr);
int launch_reduce_keys(const u64 *keys, int nq_max, int m, int k, u64 *out, const int *nq_dev, hipStream_t st);

// exact fp32-MFMA scan of one query chunk (nq <= 65,536), device-side conditional: a no-op while *only_if == 0
// (when given), over min(nq, *nq_dev) queries (when given); leaves m_out = S*k survivors per query at *part_out
// inside the given workspace, which exact_scan_fallback_bytes(n, nq_plan, k) sized (nq <= nq_plan).
size_t exact_scan_fallback_bytes(int64_t n, int64_t nq, int k);
int exact_scan_fallback(const float *d_x, int64_t n, const float *d_q, int64_t nq, int64_t nq_plan, int d, int k, void *d_ws,
                        const int *only_if, const int *nq_dev, const u64 **part_out, int *m_out, hipStream_t st);

}  // namespace ance
