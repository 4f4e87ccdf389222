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

// FROM_DI = false: entries are packed keys [nq][m]; true: entries are (D, I) parts [n_parts][nq][k]
template <bool FROM_DI>
__global__ void __launch_bounds__(256) topk_finalize_kernel(const u64 *keys, const float *pd, const int64_t *pi,
                                                            int n_parts, int64_t nq, int m, int P2, int k,
                                                            int64_t row_base, float *out_d, int64_t *out_i,
                                                            const int *sel_flag, const u64 *alt_keys, int alt_m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u64 *s = reinterpret_cast<u64 *>(smem);
    const int64_t qi = blockIdx.x;
    if (!FROM_DI && sel_flag && *sel_flag) {  // the launch chunk was redone by the exact scan: take its survivors
        keys = alt_keys;
        m = alt_m;
    }
    for (int i = threadIdx.x; i < P2; i += blockDim.x) {
        u64 v = 0ull;
        if (i < m) {
            if constexpr (FROM_DI) {
                const int p = i / k, r = i - p * k;
                const size_t o = ((size_t)p * nq + qi) * k + r;
                const int64_t id = pi[o];
                if (id >= 0) v = pack_key(pd[o], (uint32_t)id);
            } else {
                v = keys[(size_t)qi * m + i];
            }
        }
        s[i] = v;
    }
    __syncthreads();
    bitonic_sort_desc(s, P2);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const u64 v = (i < P2) ? s[i] : 0ull;
        const size_t o = (size_t)qi * k + i;
        if (v == 0ull) {
            out_d[o] = -FLT_MAX;
            out_i[o] = -1;
        } else {
            out_d[o] = key_score(v);
            out_i[o] = row_base + (int64_t)key_row(v);
        }
    }
}


inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// launches topk_finalize_kernel<false> over nq queries whose m = S*k survivors are packed keys
int launch_finalize_keys(const u64 *keys, int64_t nq, int m, int k, int64_t row_base, float *out_d, int64_t *out_i,
                         hipStream_t st, const int *sel_flag = nullptr, const u64 *alt_keys = nullptr, int alt_m = 0);

// exact fp32-MFMA scan of one query chunk (nq <= 65,536), run only if *only_if != 0 (device side);
// leaves m_out = S*k survivors per query at *part_out inside the given workspace.
size_t exact_scan_fallback_bytes(int64_t n, int64_t nq, int k);
int exact_scan_fallback(const float *d_x, int64_t n, const float *d_q, int64_t nq, int d, int k, void *d_ws, const int *only_if,
                        const u64 **part_out, int *m_out, hipStream_t st);

}  // namespace ance
