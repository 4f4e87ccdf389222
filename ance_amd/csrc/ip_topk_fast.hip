// Two-precision exact inner-product top-k for gfx950: the fast path behind ance_ip_topk.
//
// gfx950 has no reduced-precision path for fp32 inputs (no xf32), and the exact fp32 MFMA runs at
// 1/16 of the fp16 rate.  This kernel gets the fp16 rate WITHOUT giving up bit-exact results:
//
//   1. the corpus shard is turned ONCE into a search image (ance_ip_index_build): the shard's mean row mu, every row as
//      fp16(x - mu), the maximum norms of x - mu and of x, and -- when a sample of the shard shows heavy duplicate
//      classes (the all-pad MaxP chunks of model/models.py:165-199 are millions of bit-identical rows)
//      -- every class collapsed to its smallest row id; the image is compacted, `live2row` maps image
//      rows back to shard rows and the first ids of every class are kept for the expansion in step 6;
//   2. an approximate score  s~ = b + fp16(q - mq) . fp16(x - mu)  (b, mq: "Error bound" below) is an fp16 GEMM on the
//      256 x 256 x 64 direct-to-LDS main loop of pipe256.h (queries are the "m" side, so a lane owns a query), streamed
//      across the corpus tiles of a workgroup;
//   3. with eps a rigorous bound on the error of s~ (below) and t~ the k-th best APPROXIMATE score seen so
//      far, a row with s~ < t~ - 2 eps can never be in the exact top-k (k rows have s >= t~ - eps
//      > its s), so the per-query buffers keep exactly the rows with s~ >= t~ - 2 eps: about
//      k + 2 eps * density rows (~270 for k = 200 on LayerNorm-distributed rows).  The bound holds for
//      t~ taken over ANY subset of rows, so the workgroups that scan different corpus splits for the same
//      queries exchange their thresholds through global memory (stale values are merely weaker bounds);
//   4. the corpus is scanned in WINDOWS of ~100 MB that every workgroup of the launch finishes before any
//      moves on (a counter with a bounded spin: a scheduling hint, no data depends on it), and every list of every
//      workgroup is pruned at the same geometrically spaced tile counts: the workgroups stay within microseconds of each
//      other, the 16 of an XCD that scan the same corpus split find its tiles in their L2 (hit rate 15 % -> 58 %) and the
//      256 MB Infinity Cache serves the other XCDs;
//   5. when the scan is over, rescore_kernel (one wave per list) re-scores the rows within 2 eps of the final k-th best
//      approximate score (k-th over the list and its neighbour split's list together) with the exact fp32 fmaf chain over k ascending (the contract of oracle/ip_topk_ref.c) -- 64 rows
//      per round, one per lane, their fp32 data staged through LDS by coalesced 1 KiB LDS-DMAs -- and selects the exact
//      top-k under (score desc, row asc) from exact keys;
//   6. topk_finalize merges the splits and, for every duplicate class whose representative survived, adds
//      the class members (same exact score, ascending ids) before the final sort.
//   A query whose buffer cannot be pruned below its capacity (more than ~1,800 rows inside one 2 eps
//   band: pathologically clustered scores, fp16 overflow of that query) is appended to a device-side list
//   and redone by the exact fp32 scan -- only those queries; above 1,024 such queries per launch chunk the
//   whole chunk is redone.  Both are device-side conditionals, no host synchronisation.
//
// Error bound.  The image holds xh = fp16(x') with x' = fl32(x - mu), mu = the shard's mean row: q . x = q . (x - mu) + q . mu,
// and the second term is the same for every row of a query, so ranking by q . x' is ranking by q . x -- but |x'| is what the
// fp16 rounding error scales with.  Embeddings of one encoder share a large common component (random-init roberta-base:
// cosine 0.99 between any two passages, scores 737 +- 1.7): without the centring 2 eps is wider than the whole score
// distribution and every query overflows.  The queries get the same treatment: q = mq + dq with mq the mean query of the
// call, q . x' = mq . x' + dq . x'; the first term is a per-ROW constant b (one fp32 pass over the shard per call, the
// accumulators of a corpus tile start from it), and only dq meets the fp16 rounding.  With C the canonical fp32 chain score:
//   rounding dq and x' to fp16, normal range:  |dq . x' - dqh . xh| <= (2^-11 + 2^-11 + 2^-22) |dq| |x'|
//   fp32 accumulation inside / between MFMAs, starting from b:  <= 1.1 d 2^-24 (|dq| + |mq|) |x'|
//   b = fl32 chain of mq . x':                 <= d 2^-24 |mq| |x'|
//   x' = fl32(x - mu), dq = fl32(q - mq):      <= 2^-23 |q| |x'|
//   fp16 subnormal inputs, 2^-25 per element:  <= 2^-25 sqrt(d) (|dq| + |x'|)
//   the chain itself, C vs q . x:              <= d 2^-24 |q| |x|       (the un-centred norms)
// |s~ - (C - q . mu)| <= eps = 1.25 * [ (2^-10 + 1.1 d 2^-24) |dq| X' + 2.1 d 2^-24 |mq| X' + 2^-23 |q| X' + 2^-24 sqrt(d) (|dq| + X') + d 2^-24 |q| X ]
// with X' = max |x'|, X = max |x| (tests/test_eps_bound.py attacks it on the CPU).  The query mean is only used when it
// is a sizeable part of the queries (|mq| > 0.05 X); otherwise mq = 0, dq = q, b = 0 and the bias pass is skipped.
#include "common.h"
#include "topk_common.h"
#include "pipe256.h"
#include <stdlib.h>

namespace ance {
namespace {

constexpr int FQ = 256, FP = 256, FK = 64;
constexpr int F_OPER_HALVES = 256 * FK;
constexpr int F_STAGE_HALVES = 2 * F_OPER_HALVES;
constexpr int F_THREADS = 512;
constexpr int F_NPL = 32;
constexpr int F_C = F_NPL * 64;  // 2048 buffered rows per (block, query); a tile can add 256
constexpr size_t F_LDS_BYTES = (size_t)2 * F_STAGE_HALVES * sizeof(_Float16) + 3 * FQ * 4 + 16 + 2 * FP * 4;
constexpr int F_MAX_D = 2048;    // block-end re-scoring keeps 8 fp32 query rows + 8 position lists in the stage area
constexpr int F_MAX_K = 1024;
constexpr int OVF_CAP = 1024;    // overflowing queries per launch chunk that are redone one by one

// ---- search image of a shard (device memory, built by ance_ip_index_build) ----------------------------
constexpr int IDX_BLOCK_ROWS = 1024;
constexpr int IDX_SAMPLES = 2048;
constexpr int IDX_MIN_CLASS = 8;  // a duplicate class is collapsed when >= 8 of the 2,048 sampled rows fall in it

struct IndexLayout {
    size_t mu_off, x2_off, live_off, mem_off, cls_off, blk_off, samp_off, part_off, total;
    int64_t nb;
    int n_part;  // row ranges of the column-sum pass
};
IndexLayout index_layout(int64_t n, int d) {
    IndexLayout L;
    L.nb = (n + IDX_BLOCK_ROWS - 1) / IDX_BLOCK_ROWS;
    size_t o = 256;
    L.mu_off = o; o += align_up((size_t)d * sizeof(float), 256);
    L.x2_off = o; o += align_up((size_t)n * d * sizeof(_Float16), 256);
    L.live_off = o; o += align_up((size_t)n * 4, 256);
    L.mem_off = o; o += (size_t)DEDUP_MAXC * DEDUP_MEMCAP * 4;
    L.cls_off = o; o += align_up((size_t)n + 4, 256);
    L.blk_off = o; o += align_up((size_t)(1 + DEDUP_MAXC) * L.nb * 4, 256);
    L.samp_off = o; o += (size_t)IDX_SAMPLES * sizeof(u64);
    L.n_part = (int)(L.nb < 1024 ? L.nb : 1024);
    L.part_off = o; o += align_up((size_t)L.n_part * d * sizeof(float), 256);
    L.total = o;
    return L;
}

__device__ __forceinline__ u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// mean row of the shard, pass 1: block b sums the rows b, b + gridDim.x, ... per column (fp32 partials; the mean only has to be a
// fixed vector near the centre of the rows -- its own accuracy never enters the error bound)
__global__ void __launch_bounds__(256) idx_colsum_kernel(const float *x, int64_t n, int d, float *part) {
    for (int c4 = threadIdx.x; c4 * 4 < d; c4 += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int64_t r = blockIdx.x; r < n; r += gridDim.x) acc += *reinterpret_cast<const f32x4 *>(x + (size_t)r * d + c4 * 4);
        *reinterpret_cast<f32x4 *>(part + (size_t)blockIdx.x * d + c4 * 4) = acc;
    }
}
// pass 2: mu[c] = sum of the partials / n (double), or 0 when centring is off or a partial is not finite
__global__ void __launch_bounds__(256) idx_mean_kernel(const float *part, int n_part, int64_t n, int d, int center, float *mu) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (int p = 0; p < n_part; ++p) s += (double)part[(size_t)p * d + c];
    const float m = (float)(s / (double)n);
    mu[c] = (center && m == m && fabsf(m) < 3.0e38f) ? m : 0.0f;
}

// the mean query of a call: used only when it is a sizeable part of the queries (|mq| > 0.05 max|x|); decided on the device
struct QueryStat {
    float mq_norm;  // |mq| (0 when not used)
    int use_bias;   // != 0: dq = q - mq goes through the MFMAs, b = mq . x' is added per row
    int bad_image;  // != 0: the image's stamp does not match this call: no kernel touches it, every chunk is redone exactly
};
__global__ void __launch_bounds__(64) idx_stamp_kernel(DedupHeader *H, int64_t n, int d, const float *x) {
    if (threadIdx.x == 0) {
        H->d = (unsigned int)d;
        H->n = (unsigned long long)n;
        H->x_ptr = (unsigned long long)(uintptr_t)x;
        H->magic = DEDUP_MAGIC;
    }
}
// searches that found their image stamped for another matrix (or never built) and answered every chunk with the exact scan:
// correct results, several times slower -- counted so that a caller can notice (ance_search_bad_image_calls)
__device__ unsigned long long g_bad_image_calls = 0ull;

__global__ void __launch_bounds__(256) query_mean_decide_kernel(float *mq, int d, const DedupHeader *H, QueryStat *qs, int64_t n,
                                                                const float *x) {
    __shared__ float red[4];
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) s += mq[c] * mq[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]) * 1.0001f;
    const float xo = __builtin_bit_cast(float, H->xmax_orig_bits);
    const bool bad = H->magic != DEDUP_MAGIC || H->d != (unsigned int)d || H->n != (unsigned long long)n ||
                     H->x_ptr != (unsigned long long)(uintptr_t)x;
    const bool use = !bad && nrm == nrm && nrm < 3.0e38f && xo < 3.0e38f && nrm > 0.05f * xo;
    __syncthreads();
    if (!use)
        for (int c = threadIdx.x; c < d; c += 256) mq[c] = 0.0f;
    if (threadIdx.x == 0) {
        qs->mq_norm = use ? nrm : 0.0f;
        qs->use_bias = use ? 1 : 0;
        qs->bad_image = bad ? 1 : 0;
        if (bad) atomicAdd(&g_bad_image_calls, 1ull);
    }
}

// b[r] = mq . x'(image row r), x' = fl32(x - mu) recomputed from the fp32 shard row: one wave per image row
__global__ void __launch_bounds__(256) row_bias_kernel(const float *x, int d, const DedupHeader *H, const uint32_t *live2row, const float *mu,
                                                       const float *mq, const QueryStat *qs, float *bias) {
    if (!qs->use_bias) return;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t n_live = H->n_live;
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < n_live; r += (int64_t)gridDim.x * 4) {
        const float *s = x + (size_t)live2row[r] * d;
        float acc = 0.f;
        for (int k = l * 4; k < d; k += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(s + k), m4 = *reinterpret_cast<const f32x4 *>(mu + k),
                        q4 = *reinterpret_cast<const f32x4 *>(mq + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(q4[e], v[e] - m4[e], acc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (l == 0) bias[r] = acc;
    }
}

// one wave per sampled row: position-mixed 64-bit hash of the row's bits; low 11 bits carry the sample index
__global__ void __launch_bounds__(256) idx_sample_hash_kernel(const float *x, int64_t n, int d, u64 *samp) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + w;
    if (j >= IDX_SAMPLES) return;
    const int64_t row = (int64_t)(((unsigned __int128)(unsigned long long)j * (unsigned long long)n) / IDX_SAMPLES);
    const uint32_t *s = reinterpret_cast<const uint32_t *>(x + (size_t)row * d);
    u64 h = 0;
    for (int k = l; k < d; k += 64) h += mix64(((u64)s[k] << 20) ^ (u64)(k + 1) * 0x9E3779B97F4A7C15ull);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off);
    if (l == 0) samp[j] = (h & ~2047ull) | (u64)j;
}

// one block: sort the sample keys, every run of >= IDX_MIN_CLASS equal hashes defines a duplicate class
__global__ void __launch_bounds__(256) idx_find_classes_kernel(const u64 *samp, int64_t n, DedupHeader *H) {
    __shared__ u64 s[IDX_SAMPLES];
    __shared__ int ncls;
    for (int i = threadIdx.x; i < IDX_SAMPLES; i += 256) s[i] = samp[i];
    if (threadIdx.x == 0) ncls = 0;
    __syncthreads();
    bitonic_sort_desc(s, IDX_SAMPLES);
    for (int i = threadIdx.x; i < IDX_SAMPLES; i += 256) {
        if (i > 0 && (s[i] >> 11) == (s[i - 1] >> 11)) continue;  // not a run start
        int len = 1;
        while (i + len < IDX_SAMPLES && (s[i + len] >> 11) == (s[i] >> 11)) ++len;
        if (len >= IDX_MIN_CLASS) {
            const int c = atomicAdd(&ncls, 1);
            if (c < DEDUP_MAXC) {
                const int j = (int)(s[i] & 2047ull);
                H->guess[c] = (uint32_t)(((unsigned __int128)(unsigned long long)j * (unsigned long long)n) / IDX_SAMPLES);
                H->rep[c] = 0xFFFFFFFFu;
                H->csize[c] = 0;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        H->n_classes = ncls < DEDUP_MAXC ? ncls : DEDUP_MAXC;
        H->n_live = (uint32_t)n;
    }
}

// class of every row (0xFF: none): a row belongs to class c when it is BIT-identical to the class's sample row.
// A lane first compares the leading 16 bytes of its own row; only matches are compared in full by the wave.
__global__ void __launch_bounds__(256) idx_classify_kernel(const float *x, int64_t n, int d, DedupHeader *H, uint8_t *cls) {
    const int nc = H->n_classes;
    if (nc == 0) return;
    __shared__ uint32_t wmin[4][DEDUP_MAXC];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 head[DEDUP_MAXC];
    uint32_t mn[DEDUP_MAXC];
    for (int c = 0; c < DEDUP_MAXC; ++c) {
        mn[c] = 0xFFFFFFFFu;
        head[c] = *reinterpret_cast<const u32x4 *>(x + (size_t)H->guess[c < nc ? c : 0] * d);
    }
    const int64_t n_chunks = (n + 63) / 64;
    for (int64_t ch = (int64_t)blockIdx.x * 4 + w; ch < n_chunks; ch += (int64_t)gridDim.x * 4) {
        const int64_t row = ch * 64 + l;
        const bool rv = row < n;
        const u32x4 hv = *reinterpret_cast<const u32x4 *>(x + (size_t)(rv ? row : n - 1) * d);
        uint8_t mine = 0xFF;
        for (int c = 0; c < nc; ++c) {
            u64 m = __ballot(rv && hv[0] == head[c][0] && hv[1] == head[c][1] && hv[2] == head[c][2] && hv[3] == head[c][3]);
            const uint32_t *g = reinterpret_cast<const uint32_t *>(x + (size_t)H->guess[c] * d);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t *r = reinterpret_cast<const uint32_t *>(x + (size_t)(ch * 64 + b) * d);
                bool ne = false;
                for (int k = l * 4; k < d; k += 256) {
                    const u32x4 a = *reinterpret_cast<const u32x4 *>(r + k), bb = *reinterpret_cast<const u32x4 *>(g + k);
                    ne |= a[0] != bb[0] || a[1] != bb[1] || a[2] != bb[2] || a[3] != bb[3];
                }
                if (__ballot(ne) == 0ull && l == b && mine == 0xFF) {
                    mine = (uint8_t)c;
                    mn[c] = min(mn[c], (uint32_t)row);
                }
            }
        }
        if (rv) cls[row] = mine;
    }
    for (int c = 0; c < nc; ++c) {
        uint32_t v = mn[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off));
        if (l == 0) wmin[w][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < nc) {
        const int c = threadIdx.x;
        const uint32_t v = min(min(wmin[0][c], wmin[1][c]), min(wmin[2][c], wmin[3][c]));
        if (v != 0xFFFFFFFFu) atomicMin(&H->rep[c], v);
    }
}


// counter field of a row: 0 = stays in the image (no class, or the representative of its class), 1 + c = duplicate of class c
__device__ __forceinline__ int idx_row_field(const DedupHeader *H, uint8_t c, uint32_t row) {
    return (c == 0xFF || H->rep[c] == row) ? 0 : 1 + c;
}

// blk[f * nb + b] = rows of field f in block b (1,024 rows per block)
__global__ void __launch_bounds__(256) idx_count_kernel(int64_t n, int64_t nb, const DedupHeader *H, const uint8_t *cls,
                                                        uint32_t *blk) {
    if (H->n_classes == 0) return;
    __shared__ u64 ws[4];
    const int64_t r0 = (int64_t)blockIdx.x * IDX_BLOCK_ROWS + threadIdx.x * 4;
    u64 v = 0;  // five 12-bit fields (each <= 1024)
    for (int j = 0; j < 4; ++j)
        if (r0 + j < n) v += 1ull << (12 * idx_row_field(H, cls[r0 + j], (uint32_t)(r0 + j)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x <= DEDUP_MAXC) {
        const u64 t = ws[0] + ws[1] + ws[2] + ws[3];
        blk[(size_t)threadIdx.x * nb + blockIdx.x] = (uint32_t)((t >> (12 * threadIdx.x)) & 4095ull);
    }
}

// one block: exclusive scan of every field over the blocks, in place; totals go to the header
__global__ void __launch_bounds__(1024) idx_scan_kernel(int64_t nb, DedupHeader *H, uint32_t *blk) {
    if (H->n_classes == 0) return;
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x;
    for (int f = 0; f <= DEDUP_MAXC; ++f) {
        if (tid == 0) carry = 0;
        __syncthreads();
        for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
            const int64_t b = b0 + tid;
            const uint32_t mine = b < nb ? blk[(size_t)f * nb + b] : 0u;
            sh[tid] = mine;
            __syncthreads();
            for (int s = 1; s < 1024; s <<= 1) {
                const uint32_t t = tid >= s ? sh[tid - s] : 0u;
                __syncthreads();
                sh[tid] += t;
                __syncthreads();
            }
            if (b < nb) blk[(size_t)f * nb + b] = carry + sh[tid] - mine;
            __syncthreads();
            if (tid == 0) carry += sh[1023];
            __syncthreads();
        }
        if (tid == 0) {
            if (f == 0) H->n_live = carry;
            else H->csize[f - 1] = carry;
        }
        __syncthreads();
    }
}

// Builds the image: block b owns rows [1024 b, 1024 b + 1024).  Phase A ranks the block's rows inside their field
// (image position of a kept row, ordinal of a duplicate inside its class); phase B rounds the kept rows to fp16 at
// their image position (one wave per row) and folds their norms into the shard maximum.
__global__ void __launch_bounds__(256) idx_compact_round_kernel(const float *x, int64_t n, int d, int64_t nb, DedupHeader *H,
                                                                const uint8_t *cls, const uint32_t *blk, const float *mu,
                                                                _Float16 *x2, uint32_t *live2row, uint32_t *members) {
    __shared__ uint32_t pos_s[IDX_BLOCK_ROWS];  // image row of the block's rows, 0xFFFFFFFF for collapsed duplicates
    __shared__ u64 wtot[4];
    __shared__ float wmax[4], wmaxo[4];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * IDX_BLOCK_ROWS;
    const int nc = H->n_classes;
    if (nc == 0) {
        if (blockIdx.x == 0 && tid == 0) H->n_live = (uint32_t)n;  // (not set by anyone when the class search is off)
        for (int j = tid; j < IDX_BLOCK_ROWS; j += 256) {
            const int64_t row = r0 + j;
            pos_s[j] = row < n ? (uint32_t)row : 0xFFFFFFFFu;
            if (row < n) live2row[row] = (uint32_t)row;
        }
    } else {
        int fld[4];
        u64 v = 0;
        for (int j = 0; j < 4; ++j) {
            const int64_t row = r0 + tid * 4 + j;
            fld[j] = row < n ? idx_row_field(H, cls[row], (uint32_t)row) : -1;
            if (fld[j] >= 0) v += 1ull << (12 * fld[j]);
        }
        // exclusive prefix of the packed counters over the block's 256 threads (rows are in thread order)
        u64 inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u64 t = __shfl_up(inc, off);
            if (l >= off) inc += t;
        }
        if (l == 63) wtot[w] = inc;
        __syncthreads();
        u64 base = 0;
        for (int ww = 0; ww < w; ++ww) base += wtot[ww];
        u64 run = base + inc - v;
        for (int j = 0; j < 4; ++j) {
            if (fld[j] < 0) {  // past the end of the shard
                pos_s[tid * 4 + j] = 0xFFFFFFFFu;
                continue;
            }
            const uint32_t row = (uint32_t)(r0 + tid * 4 + j);
            const uint32_t rank = (uint32_t)((run >> (12 * fld[j])) & 4095ull);
            const uint32_t at = blk[(size_t)fld[j] * nb + blockIdx.x] + rank;
            if (fld[j] == 0) {
                pos_s[tid * 4 + j] = at;
                live2row[at] = row;
            } else {
                pos_s[tid * 4 + j] = 0xFFFFFFFFu;
                if (at < (uint32_t)DEDUP_MEMCAP) members[(size_t)(fld[j] - 1) * DEDUP_MEMCAP + at] = row;
            }
            run += 1ull << (12 * fld[j]);
        }
    }
    __syncthreads();
    float mymax = 0.0f, mymaxo = 0.0f;
    for (int j = w; j < IDX_BLOCK_ROWS; j += 4) {
        const uint32_t at = pos_s[j];
        if (at == 0xFFFFFFFFu) continue;  // wave-uniform
        const float *s = x + (size_t)(r0 + j) * d;
        _Float16 *hi = x2 + (size_t)at * d;
        float q = 0.f, qo = 0.f;
        for (int k = l * 4; k < d; k += 256) {
            const f32x4 vv = *reinterpret_cast<const f32x4 *>(s + k);
            const f32x4 m4 = *reinterpret_cast<const f32x4 *>(mu + k);
            f16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = vv[e] - m4[e];  // x' = fl32(x - mu): what the image holds, rounded to fp16
                h[e] = (_Float16)c;
                q = fmaf(c, c, q);
                qo = fmaf(vv[e], vv[e], qo);
            }
            *reinterpret_cast<f16x4 *>(hi + k) = h;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            q += __shfl_xor(q, off);
            qo += __shfl_xor(qo, off);
        }
        float nr = sqrtf(q) * 1.0001f, nro = sqrtf(qo) * 1.0001f;  // the norms only feed an upper bound
        if (!(nr == nr)) nr = INFINITY;  // a NaN row must not hide from the fp16-trust test of the filter
        if (!(nro == nro)) nro = INFINITY;
        mymax = fmaxf(mymax, nr);
        mymaxo = fmaxf(mymaxo, nro);
    }
    // ONE atomic per block and maximum (a single word saturates near 88 atomics/us)
    if (l == 0) {
        wmax[w] = mymax;
        wmaxo[w] = mymaxo;
    }
    __syncthreads();
    if (tid == 0) {
        atomicMax(&H->xmax_bits, __builtin_bit_cast(unsigned int, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
        atomicMax(&H->xmax_orig_bits, __builtin_bit_cast(unsigned int, fmaxf(fmaxf(wmaxo[0], wmaxo[1]), fmaxf(wmaxo[2], wmaxo[3]))));
    }
}

// fp16(q - mq) + the norms of dq = q - mq and of q for the query chunk: one wave per row
__global__ void __launch_bounds__(256) round_rows_kernel(const float *src, int64_t rows, int d, const float *mq, _Float16 *dst,
                                                         float *norm_c, float *norm_o) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
        const float *s = src + (size_t)row * d;
        _Float16 *hi = dst + (size_t)row * d;
        float q = 0.f, qo = 0.f;
        for (int k = l * 4; k < d; k += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(s + k), m4 = *reinterpret_cast<const f32x4 *>(mq + k);
            f16x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = v[j] - m4[j];
                h[j] = (_Float16)c;
                q = fmaf(c, c, q);
                qo = fmaf(v[j], v[j], qo);
            }
            *reinterpret_cast<f16x4 *>(hi + k) = h;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            q += __shfl_xor(q, off);
            qo += __shfl_xor(qo, off);
        }
        if (l == 0) {  // NaN stays NaN: the filter's trust test is false for it
            norm_c[row] = sqrtf(q) * 1.0001f;
            norm_o[row] = sqrtf(qo) * 1.0001f;
        }
    }
}

// 2 eps of one query (header of this file); INFINITY when fp16 cannot be trusted (a norm above 65504, or not finite)
struct EpsConst {
    float rel_c, acc_m, cen, abs_c, chain_o;  // 1.25 x: (2^-10 + 1.1 d 2^-24), 2.1 d 2^-24, 2^-23, 2^-24 sqrt(d), d 2^-24
};
// qc = |q - mq|, qo = |q|
__device__ __forceinline__ float two_eps(const EpsConst &E, float qc, float qo, const QueryStat *qs, const DedupHeader *H) {
    const float xc = __builtin_bit_cast(float, H->xmax_bits), xo = __builtin_bit_cast(float, H->xmax_orig_bits);
    const bool ok = qc <= 65504.0f && qo < 3.0e38f && xc <= 65504.0f && xo < 3.0e38f;  // false for NaN too
    return ok ? 2.0f * (E.rel_c * qc * xc + E.acc_m * qs->mq_norm * xc + E.cen * qo * xc + E.abs_c * (qc + xc) + E.chain_o * qo * xo)
              : INFINITY;
}

// ---- per-launch control block (device, zeroed before every launch chunk) ------------------------------
struct FastCtl {
    unsigned int win_arrived;  // workgroups that finished a corpus window (monotonic over the launch)
    int ovf_count;             // queries appended to ovf_list (may exceed OVF_CAP)
    int fb_nq;                 // queries the per-query exact scan redoes (0 when the whole chunk is redone)
    int fb_all;                // != 0: the whole chunk is redone by the exact scan
};

struct FastParams {
    const _Float16 *q2;  // [nq, d]  fp16(q)
    const _Float16 *x2;  // [n_live, d] fp16 image rows
    const float *q32;    // [nq, d]
    const float *x32;    // [n, d] shard rows
    const float *qnorm_c, *qnorm_o;  // [nq] |q - mq|, |q|
    const QueryStat *qstat;
    const float *bias;   // [n_live + 256] mq . x' per image row (read only when qstat->use_bias)
    const DedupHeader *hdr;
    const uint32_t *live2row;
    uint32_t nq;
    int d, k, S, n_qt;
    int Ws;              // corpus tiles per split per window (a window is S * Ws tiles)
    int share;           // exchange thresholds between the splits of a query tile
    unsigned int wait_ticks;  // bound of the window wait (100 MHz ticks)
    EpsConst eps;
    u64 *cand;     // [n_qt * S][FQ][F_C]
    u64 *part;     // [nq][S][k]
    float *thr_g;  // [n_qt * S][FQ] published thresholds (NaN = none yet)
    int *cnt_g;    // [n_qt * S][FQ] rows left in every buffer (for rescore_kernel)
    FastCtl *ctl;
    int *ovf_flag;  // [nq] 0 / 1
    int *ovf_list;  // [OVF_CAP]
    int prune_at;           // first scheduled prune once every list has this many rows (<= F_C - FP)
    int prune_growth;       // percent: the tile count between scheduled prunes grows by this factor (150 = 1.5x)
    int dbg;                     // STAMPS kernel only, timing experiments (results WRONG): 1 = no insertions after window 4,
                                 // 2 = no filter at all after window 4, 3 = as 2 and no prune check either
    unsigned long long *stamps;  // measurement: [workgroup][8] accumulated 100 MHz ticks (STAMPS kernel only)
};

// Source policy of the streamed main loop (pipe256.h).  Both operands go through buffer descriptors (wave-uniform
// SGPRs) + one 32-bit per-lane byte offset per staged piece that never changes during the kernel, + the K offset in an
// SGPR: 8 address VGPRs in all.  (With flat 64-bit addresses hipcc keeps a pointer pair per piece for the current AND
// the next corpus tile, spills, and every spill reload in the tile loop is a vmcnt(0) that drains the prefetch.)
// The descriptor of a corpus tile covers exactly its rows that exist (<= 256), the one of the query tile its real
// queries: rows past the end read as zeros (hardware range check) and are masked in the filter.
// K-tile t >= NK belongs to the NEXT corpus tile of this workgroup's sequence (descriptor rx1).
struct FastSrc {
    __amdgpu_buffer_rsrc_t rq, rx0, rx1;
    uint32_t voff[4][2];  // [A-half0, A-half1, B-half0, B-half1][piece]: (row of the 256-row tile) * d * 2 + chunk * 2 bytes
    int NK;
    template <int TYPE, int J>
    __device__ __forceinline__ void issue(int t, pipe_lds_t *dst) const {
        const bool nxt = t >= NK;
        const int so = (nxt ? t - NK : t) * (FK * 2);
        if constexpr (TYPE < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, dst, 16, voff[TYPE][J], so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(nxt ? rx1 : rx0, dst, 16, voff[TYPE][J], so, 0, 0);
    }
    // Round 6 (pipe256.h: PRECOMPUTE): the K offsets of the eight LDS-DMAs of a K-tile and the corpus descriptor of K-tile t + 2 are
    // computed ONCE per K-tile, in the read half-phase -- the compare / select / shift chains (and the four s_cselect of the descriptor)
    // used to sit in front of each DMA, between the MFMAs, fenced there by the schedule's sched_barriers: ~25 scalar instructions per
    // K-tile in the matrix pipe's shadow.
#ifdef ANCE_FAST_NO_PRECOMPUTE  // A/B builds only (make variant NAME=noprep DEFS=-DANCE_FAST_NO_PRECOMPUTE): the form of rounds 2-5
    static constexpr bool PRECOMPUTE = false;
#else
    static constexpr bool PRECOMPUTE = true;
#endif
    int so_1, so_2;                // K offset (bytes) of K-tile t + 1 (A-half1) and of K-tile t + 2 (A-half0, B-half0, B-half1)
    __amdgpu_buffer_rsrc_t rx_2;   // corpus descriptor of K-tile t + 2
    __device__ __forceinline__ void prepare(int t) {
        const int t1 = t + 1, t2 = t + 2;
        so_1 = (t1 >= NK ? t1 - NK : t1) * (FK * 2);
        so_2 = (t2 >= NK ? t2 - NK : t2) * (FK * 2);
        rx_2 = t2 >= NK ? rx1 : rx0;
    }
    template <int TYPE, int J>
    __device__ __forceinline__ void issue_pre(pipe_lds_t *dst) const {
        if constexpr (TYPE == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, dst, 16, voff[TYPE][J], so_1, 0, 0);
        else if constexpr (TYPE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, dst, 16, voff[TYPE][J], so_2, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_2, dst, 16, voff[TYPE][J], so_2, 0, 0);
    }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const _Float16 *base, uint32_t first_row, uint32_t n_rows, int d) {
    const uint32_t rows = first_row < n_rows ? min(n_rows - first_row, 256u) : 0u;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(base + (size_t)first_row * d), 0, (int)(rows * (uint32_t)d * 2u),
                                             0x00020000);
}

// v_max3_f32 without the canonicalisation (v_max_f32 x, x) hipcc puts in front of every fmaxf operand it cannot prove quiet.
// NaN operands lose against numbers, like fmaxf.  The caller pads the MFMA -> VALU hazard of the first use.
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float load_thr(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define STAMP(acc)                                      \
    if constexpr (STAMPS) {                             \
        const unsigned long long now_ = wall_clock64(); \
        acc += now_ - t_last;                           \
        t_last = now_;                                  \
    }

// One wave prunes one list in place: the k-th largest approximate key by radix select, then every row whose approximate
// score is below max(that score - 2 eps, thr_other) goes.  Returns the rows kept; *thr_out = the new filter threshold.
template <int NPL>
__device__ __forceinline__ int prune_list(u64 *cq, int n_c, int lp, int k, float eps2, float thr_other, float *thr_out) {
    u64 keys[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int idx = j * 64 + lp;
        keys[j] = (idx < n_c) ? cq[idx] : 0ull;
    }
    u64 T = 0;  // k-th largest approximate key
    for (int bit = 63; bit >= 0; --bit) {
        const u64 t2 = T | (1ull << bit);
        int ge = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) ge += __popcll(__ballot(keys[j] >= t2));
        if (ge >= k) T = t2;
    }
    const float thr_new = fmaxf(key_score(T) - eps2, thr_other);
    int base = 0;
    const u64 lt_mask = (1ull << lp) - 1ull;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const bool keep = keys[j] != 0ull && !(key_score(keys[j]) < thr_new);
        const u64 m = __ballot(keep);
        if (keep) cq[base + __popcll(m & lt_mask)] = keys[j];
        base += __popcll(m);
    }
    *thr_out = thr_new;
    return base;
}

// BIAS: the build that starts every corpus tile's accumulators from the per-row share of the mean query (header).  Both
// builds are launched for every chunk and the one the device-side decision (QueryStat) did not pick returns at once: the
// choice needs no host synchronisation, and the common case keeps the leaner kernel (the bias build is ~5 % slower).
template <bool STAMPS, bool BIAS>
__global__ void __launch_bounds__(F_THREADS, 2) ip_topk_fast_kernel(const FastParams P) {
    if ((P.qstat->use_bias != 0) != BIAS || P.qstat->bad_image) return;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    _Float16 *smem = reinterpret_cast<_Float16 *>(smem_f);
    float *thr_s = smem_f + (2 * F_STAGE_HALVES) / 2;  // after the 128 KiB of stages: filter threshold t~ - 2 eps
    float *eps2_s = thr_s + FQ;                         // 2 eps per query
    int *cnt_s = reinterpret_cast<int *>(eps2_s + FQ);

    // block -> (query tile, corpus split).  32 blocks of an XCD run at once (1 per CU): a group is
    // 32/S query tiles x S splits, so an XCD keeps few query tiles hot and shares each corpus tile.
    const int b = blockIdx.x, xcd = b & 7, jx = b >> 3;
    const int gq = 32 / P.S;
    const int grp = (jx >> 5) * 8 + xcd;
    const int r32 = jx & 31;
    const int qt = grp * gq + r32 / P.S;
    const int split = r32 % P.S;
    if (qt >= P.n_qt) return;

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int wm = w >> 2, wn = w & 3;  // wave tile: 128 queries x 64 passages
    const uint32_t q0 = (uint32_t)qt * FQ;
    const int d = P.d;
    const uint32_t n = P.hdr->n_live;  // rows of the image (device side: duplicates were collapsed there)
    const int n_tiles = (int)((n + FP - 1) / FP);
    const int W = P.Ws * P.S;
    const int n_win = (n_tiles + W - 1) / W;
    const unsigned n_part = (unsigned)(P.n_qt * P.S);
    u64 *cand = P.cand + ((size_t)qt * P.S + split) * (size_t)FQ * F_C;
    float *thr_mine = P.thr_g + ((size_t)qt * P.S + split) * FQ;
    const float *thr_tile = P.thr_g + (size_t)qt * P.S * FQ;

    if (tid < FQ) {
        const uint32_t qg = q0 + tid;
        thr_s[tid] = -INFINITY;
        cnt_s[tid] = 0;
        // the bound assumes no fp16 overflow: |x_j| <= ||x||, so norms <= 65504 exclude it.  Otherwise eps = inf
        // keeps every row until the buffer overflows and the query is redone by the exact scan.
        eps2_s[tid] = two_eps(P.eps, qg < P.nq ? P.qnorm_c[qg] : 0.0f, qg < P.nq ? P.qnorm_o[qg] : 0.0f, P.qstat, P.hdr);
    }

    // ---- main loop: the ping-pong pipeline of pipe256.h, streamed across this workgroup's corpus tiles ----
    // A operand = the block's 256 queries (re-read from L2 for every corpus tile), B operand = image rows.
    // Tile sequence: window by window, inside a window the Ws tiles of this split.  K-tile index t of the tile
    // being computed; t >= NK addresses the next tile of the sequence, so the LDS-DMA prefetch (5-6 phases
    // ahead) runs through the filter step into the next tile.
    Pipe256T<FastSrc, false, true, true> pipe;  // coarse schedule (two phases per K-tile, B-half0 fragments kept in registers)
    pipe.init(smem, w, l);
    {
        FastSrc &S = pipe.S;
        S.NK = d / FK;
        S.rq = tile_rsrc(P.q2, q0, P.nq, d);
        const int ch = pipe_stage_chunk(pipe_stage_row(w, l, 0), l);  // rows of piece 1 are 64 further: same swizzle
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = pipe_stage_row(w, l, j);
                S.voff[h][j] = (uint32_t)(pipe_a_tile_row(h, r) * d + ch) * 2u;
                S.voff[2 + h][j] = (uint32_t)(pipe_b_tile_row(h, r) * d + ch) * 2u;
            }
    }
    const int NK = d / FK;
    // Prune schedule.  The filter threshold is only as fresh as the last prune, and for most of the scan the insertion
    // rate is (rows kept at the last prune) / (rows seen at the last prune) per row: waiting for a full buffer (1,792
    // rows) lets the rows seen grow 6.6x between prunes and has ~1 insertion per (wave, query group) per tile.  Every
    // list of the workgroup is therefore pruned at the SAME geometrically spaced tile counts (x P.prune_growth / 100:
    // ~24 episodes over 17 k tiles at 1.5x): thresholds stay within 1.5x of fresh, and because every workgroup of the
    // launch follows the same schedule the episodes (a vmcnt(0) drain + ~0.3 ms of selection) coincide instead of
    // making a different workgroup the straggler of every window.  A buffer that fills up in between is pruned at once.
    int *epoch_s = cnt_s + FQ;  // last tile (1-based) in which some wave asked for an unscheduled prune
    if (tid == 0) *epoch_s = 0;
    // Per-row bias b = mq . x' of the corpus tile (header: the mean query's share of every score): two 1 KiB LDS slots,
    // filled by ONE LDS-DMA of wave 0 a whole tile ahead -- the instruction is older than every staging DMA the pipeline
    // counts, so the pipeline's own waits and barriers retire and publish it -- and read back as the accumulators' start.
    float *bias_s = reinterpret_cast<float *>(epoch_s + 4);
    int bbuf = 0;
    auto stage_bias = [&](int tile, int buf) {
        if (w == 0) {
            int lb = l;  // (laundered: keeps the per-lane address out of the tile loop's live registers, see the filter)
            asm volatile("" : "+v"(lb));
            __builtin_amdgcn_global_load_lds((pipe_glb_t *)(P.bias + (size_t)tile * FP + lb * 4), (pipe_lds_t *)(bias_s + buf * FP), 16, 0, 0);
        }
    };
    int n_done = 0, next_sched = max(1, (P.prune_at + FP - 1) / FP);

    unsigned long long t_last = 0, a_main = 0, a_filter = 0, a_prune = 0, a_sync = 0, a_end = 0, a_pro = 0;
    if constexpr (STAMPS) t_last = wall_clock64();

    int t = split * P.Ws, jw = 0, win = 0;
    bool have = t < n_tiles;
    if (have) {
        if constexpr (BIAS) stage_bias(t, 0);
        pipe.S.rx0 = tile_rsrc(P.x2, (uint32_t)t * FP, n, d);
        pipe.S.rx1 = pipe.S.rx0;
        pipe.prologue();  // also publishes thr_s / cnt_s / eps2_s / epoch_s
    } else {
        __syncthreads();
    }
    STAMP(a_pro)

    while (have) {
        int tn, jn = jw + 1, winn = win;
        if (jn < P.Ws) {
            tn = t + 1;
        } else {
            jn = 0;
            winn = win + 1;
            tn = winn * W + split * P.Ws;
        }
        const bool have_n = tn < n_tiles;
        const uint32_t p0 = (uint32_t)t * FP;
        pipe.S.rx0 = tile_rsrc(P.x2, p0, n, d);
        pipe.S.rx1 = tile_rsrc(P.x2, (uint32_t)tn * FP, n, d);
        f32x16 acc[2][4];
        if constexpr (BIAS) {
            if (have_n) stage_bias(tn, bbuf ^ 1);
            // acc[x][y][4 rq + j] <- b[row p0 + wn*64 + x*32 + 8 rq + 4 g + j], the same for the four query groups y
            int lb = l;
            asm volatile("" : "+v"(lb));
            const float *bs = bias_s + bbuf * FP + wn * 64 + 4 * (lb >> 5);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(bs + x * 32 + 8 * rq);
#pragma unroll
                    for (int y = 0; y < 4; ++y)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[x][y][4 * rq + j] = v[j];
                }
            bbuf ^= 1;
        } else {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = f32x16{0};
        }
        pipe.enter();
        if (have_n) pipe.tiles_streaming(NK, acc);
        else pipe.tiles_final(NK, acc);
        pipe.leave();
        STAMP(a_main)

        // ---- filter: keep every row whose approximate score is within 2 eps of the k-th best -------
        // acc[x][y][r]: passage = p0 + wn*64 + x*32 + (r&3) + 8 (r>>2) + 4 g ; query = q0 + wm*128 + y*32 + i
        // (the lane id is laundered through an empty asm: hipcc otherwise hoists every lane-derived address of this
        // section out of the tile loop, runs out of registers and reloads them from scratch here -- and a scratch
        // reload is a vmcnt(0) wait that drains the LDS-DMA prefetch of the next tile)
        int lf = l;
        asm volatile("" : "+v"(lf));
        const int gf = lf >> 5, qf = wm * 128 + (lf & 31);
        const uint32_t pw0 = p0 + wn * 64 + 4 * gf;
        const bool ragged = p0 + FP > n;  // uniform: rows past n were staged as zeros
        // MFMA results -> VALU reads inside asm statements: the compiler does not pad that hazard for us
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            if constexpr (STAMPS) {
                if (P.dbg >= 2 && win > 4) continue;
            }
            const int ql = qf + y * 32;
            const bool qv = (q0 + ql) < P.nq;
            const float thr = thr_s[ql];  // -inf until the first prune
            // Once the threshold is set few rows pass: take the maximum of the lane's 32 scores first and skip the whole
            // group when no lane of the wave has a candidate.  Scores are finite (fp16_ok above), so the maximum loses
            // nothing.  Not on a ragged last tile (clamped rows).  Quarter maxima (8 scores each) come out of the same
            // max tree (v_max3_f32: 18 instructions per 32 scores): the per-score compare-and-insert code only runs for
            // the quarters that hold a candidate somewhere in the wave.
            float mq[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int x = qd >> 1, rb = (qd & 1) * 8;
                float m = max3_f32(acc[x][y][rb], acc[x][y][rb + 1], acc[x][y][rb + 2]);
                m = max3_f32(m, acc[x][y][rb + 3], acc[x][y][rb + 4]);
                m = max3_f32(m, acc[x][y][rb + 5], acc[x][y][rb + 6]);
                mq[qd] = max3_f32(m, acc[x][y][rb + 7], acc[x][y][rb + 7]);
            }
            if (!ragged) {
                const float mx = max3_f32(max3_f32(mq[0], mq[1], mq[2]), mq[3], mq[3]);
                if (__ballot(qv && !(mx < thr)) == 0ull) continue;
            }
            if constexpr (STAMPS) {
                if (P.dbg == 1 && win > 4) continue;
            }
            u64 *cq = cand + (size_t)ql * F_C;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                if (!ragged && __ballot(qv && !(mq[qd] < thr)) == 0ull) continue;
                const int x = qd >> 1, rb = (qd & 1) * 8;
#pragma unroll
                for (int r = rb; r < rb + 8; ++r) {
                    const uint32_t prow = pw0 + x * 32 + (r & 3) + 8 * (r >> 2);
                    const float sc = acc[x][y][r];
                    if (qv && prow < n && !(sc < thr)) {
                        const int sl = atomicAdd(&cnt_s[ql], 1);
                        cq[sl] = pack_key(sc, prow);
                    }
                }
            }
        }
        // ---- prune buffers that could overflow on the next tile (approximate keys) --------------------
        // Barriers here are raw s_barriers: a __syncthreads would drain the LDS-DMA prefetch of the next
        // tile (vmcnt(0)).  Only when some buffer really needs a prune (a few times per query, early in
        // the scan) do all waves retire their candidate stores before anybody reads them back.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STAMP(a_filter)
        ++n_done;
        const bool sched = n_done == next_sched;  // uniform
        if (sched) next_sched = max(next_sched + 1, (int)(((long long)next_sched * P.prune_growth) / 100));
        {
            const int c32 = cnt_s[w * 32 + (l & 31)];
            if (__ballot(c32 > F_C - FP) != 0ull && l == 0) *epoch_s = t + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (sched || *epoch_s == t + 1) {  // block-uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int lp = l;  // laundered like lf above: keeps the 32 buffer positions j * 64 + lane out of the tile loop's registers
            asm volatile("" : "+v"(lp));
            // queries of this wave whose buffer passed its trigger (lane q < 32 looks at query w * 32 + q)
            u64 need = __ballot(lp < 32 && cnt_s[w * 32 + (lp & 31)] > (sched ? P.k + 32 : F_C - FP));
            while (need) {
                const int qq = __builtin_ctzll(need);
                need &= need - 1;
                const int ql = w * 32 + qq;
                const int n_c = __builtin_amdgcn_readfirstlane(cnt_s[ql]);
                {
                    u64 *cq = cand + (size_t)ql * F_C;
                    float thr_other = -INFINITY;
                    if (P.share) {  // what the other splits of this query have established is just as valid here
                        float o = (lp < P.S && lp != split) ? load_thr(thr_tile + (size_t)lp * FQ + ql) : -INFINITY;
#pragma unroll
                        for (int off = 16; off > 0; off >>= 1) o = fmaxf(o, __shfl_xor(o, off));  // S <= 32; NaN = none
                        thr_other = __shfl(o, 0);
                    }
                    float thr_new;
                    int base;  // (scheduled prunes see a few hundred rows: 8 keys per lane instead of 32)
                    if (n_c <= 8 * 64) base = prune_list<8>(cq, n_c, lp, P.k, eps2_s[ql], thr_other, &thr_new);
                    else if (n_c <= 16 * 64) base = prune_list<16>(cq, n_c, lp, P.k, eps2_s[ql], thr_other, &thr_new);
                    else base = prune_list<F_NPL>(cq, n_c, lp, P.k, eps2_s[ql], thr_other, &thr_new);
                    if (lp == 0) {
                        if (base > F_C - FP) {  // more than 1,792 rows inside one 2 eps band: this query is redone exactly
                            if (atomicExch(&P.ovf_flag[q0 + ql], 1) == 0) {
                                const int slot = atomicAdd(&P.ctl->ovf_count, 1);
                                if (slot < OVF_CAP) P.ovf_list[slot] = (int)(q0 + ql);
                            }
                            base = F_C - FP;
                        }
                        cnt_s[ql] = base;
                        thr_s[ql] = thr_new;
                        if (P.share) __hip_atomic_store(thr_mine + ql, thr_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        STAMP(a_prune)
        // ---- window boundary: wait (bounded) for the other workgroups, adopt their thresholds ----------
        if (winn != win && winn < n_win) {  // block-uniform
            if (tid == 0) {
                __hip_atomic_fetch_add(&P.ctl->win_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (have_n && P.wait_ticks) {
                    const unsigned target = (unsigned)(win + 1) * n_part;
                    const unsigned long long t_in = wall_clock64();
                    while (__hip_atomic_load(&P.ctl->win_arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        __builtin_amdgcn_s_sleep(64);
                        if (wall_clock64() - t_in > P.wait_ticks) break;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (P.share && tid < FQ) {
                float m = thr_s[tid];
                for (int s2 = 0; s2 < P.S; ++s2)
                    if (s2 != split) m = fmaxf(m, load_thr(thr_tile + (size_t)s2 * FQ + tid));
                thr_s[tid] = m;
            }
        }
        // thr_s / cnt_s updates are published by the barriers of the next tile's main loop
        t = tn; jw = jn; win = winn; have = have_n;
        STAMP(a_sync)
    }
    __syncthreads();

    // ---- hand the lists to rescore_kernel: rows buffered, final threshold --------------------------------------
    if (tid < FQ) {
        P.cnt_g[((size_t)qt * P.S + split) * FQ + tid] = cnt_s[tid];
        thr_mine[tid] = thr_s[tid];  // (thr_g doubles as the final-threshold array: nobody reads it during the scan any more
                                     //  once every split of this query tile is done, and a stale read is only a weaker bound)
    }
    if constexpr (STAMPS) {
        STAMP(a_end)
        if (tid == 0) {
            unsigned long long *o = P.stamps + (size_t)blockIdx.x * 8;
            o[0] = a_pro; o[1] = a_main; o[2] = a_filter; o[3] = a_prune; o[4] = a_sync; o[5] = a_end;
            o[6] = (unsigned long long)qt << 32 | (unsigned)split;
            o[7] = __builtin_amdgcn_s_getreg(0x1814) /* XCC_ID */;
        }
    }
}

// ---- exact re-scoring of the lists the filter kernel left: one wave (= one workgroup) per (query, split) ---------------
// A buffer ends the scan with a few hundred rows (everything above the LAST threshold), but only the rows within 2 eps
// of the final k-th best approximate score (about k + 66 per query) can be in the exact top-k.  The band is cut first: the
// k-th approximate score by radix select over this list and the next split's, the survivors' buffer positions compacted into an LDS list.  Then, 64 rows per
// round (one per lane), 256 floats of every row at a time: the wave copies the 64 row pieces into LDS with one
// 1 KiB LDS-DMA each -- every lane reading its own row straight from memory made 64 scattered 16-byte requests per
// load instruction and ran at 1.3 TB/s -- and each lane runs the canonical fmaf chain (k ascending) over its row's
// piece from LDS (row stride 1040 bytes: conflict-free ds_read_b128).  Exact keys stay in registers for the selection.
struct RescoreParams {
    const float *q32, *x32, *qnorm_c, *qnorm_o;
    const QueryStat *qstat;
    const DedupHeader *hdr;
    const uint32_t *live2row;
    const u64 *cand;     // [n_qt * S][FQ][F_C]
    const int *cnt_g;    // [n_qt * S][FQ]
    const float *thr_g;  // [n_qt * S][FQ] final filter thresholds
    u64 *part;           // [nq][S][k]
    uint32_t nq;
    int d, k, S;
    EpsConst eps;
};
constexpr int RS_CHUNK = 256;            // floats of a row staged per step
constexpr int RS_STRIDE = RS_CHUNK + 4;  // floats between the staged pieces of consecutive rows
inline size_t rescore_lds_bytes(int d) { return ((size_t)d + F_C / 2 + 64 * RS_STRIDE) * sizeof(float); }

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) rescore_kernel(const RescoreParams P) {  // LDS allows 2 waves per CU
    extern __shared__ __attribute__((aligned(16))) float rs_smem[];
    if (P.qstat->bad_image) return;
    const int l = threadIdx.x;
    const int d = P.d;
    float *qrow_lds = rs_smem;                                                   // d floats
    unsigned short *list = reinterpret_cast<unsigned short *>(qrow_lds + d);    // F_C buffer positions (4 KiB)
    float *stage = qrow_lds + d + F_C / 2;                                       // 64 x RS_STRIDE floats
    // list id -> (query tile, split, query): lists of one query tile and split are consecutive
    const size_t lid = blockIdx.x;
    const int ql = (int)(lid % FQ);
    const size_t ts = lid / FQ;  // qt * S + split
    const int split = (int)(ts % P.S);
    const uint32_t qg = (uint32_t)(ts / P.S) * FQ + ql;
    if (qg >= P.nq) return;
    const int n_c = P.cnt_g[lid];
    const u64 *cq = P.cand + lid * (size_t)F_C;
    u64 *dst = P.part + ((size_t)qg * P.S + split) * (size_t)P.k;
    const float *qsrc = P.q32 + (size_t)qg * d;
    for (int k4 = l * 4; k4 < d; k4 += 256) *reinterpret_cast<f32x4 *>(qrow_lds + k4) = *reinterpret_cast<const f32x4 *>(qsrc + k4);
    const float eps2 = two_eps(P.eps, P.qnorm_c[qg], P.qnorm_o[qg], P.qstat, P.hdr);  // as in the filter kernel
    const u64 lt_mask = (1ull << l) - 1ull;
    u64 keys[F_NPL];
#pragma unroll
    for (int j = 0; j < F_NPL; ++j) {
        const int idx = j * 64 + l;
        keys[j] = (idx < n_c) ? cq[idx] : 0ull;
    }
    float thr_band = P.thr_g[lid];  // rows buffered before the threshold rose (own prunes, other splits) are out as well
    if (!(thr_band == thr_band)) thr_band = -INFINITY;
    // The k-th best approximate score is taken over this list AND the list of the next split of the same query (with two
    // splits: over everything the filter kept for the query): a threshold from any subset of the rows is valid for all of
    // them, and the union's k-th is what the merged answer is cut at -- each list then keeps its share of the ~k + 66
    // band rows instead of k + 66 of its own.  Only the score halves of the keys take part (32 radix steps).
    const int n_own = (n_c + 63) >> 6;  // registers in use (wave-uniform)
    uint32_t sib[F_NPL];
    int n_c2 = 0, n_sib = 0;
    if (P.S > 1) {
        const size_t lid2 = (ts - split + (size_t)((split + 1) % P.S)) * FQ + ql;
        n_c2 = min(P.cnt_g[lid2], F_C);
        n_sib = (n_c2 + 63) >> 6;
        const u64 *cq2 = P.cand + lid2 * (size_t)F_C;
#pragma unroll
        for (int j = 0; j < F_NPL; ++j) {
            const int idx = j * 64 + l;
            sib[j] = (idx < n_c2) ? (uint32_t)(cq2[idx] >> 32) : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < F_NPL; ++j) sib[j] = 0u;
    }
    if (n_c + n_c2 >= P.k) {
        uint32_t T = 0;  // score half of the k-th largest approximate key of the union
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t t2 = T | (1u << bit);
            int ge = 0;
#pragma unroll
            for (int j = 0; j < F_NPL; ++j) {
                if (j < n_own) ge += __popcll(__ballot((uint32_t)(keys[j] >> 32) >= t2));
            }
#pragma unroll
            for (int j = 0; j < F_NPL; ++j) {
                if (j < n_sib) ge += __popcll(__ballot(sib[j] >= t2));
            }
            if (ge >= P.k) T = t2;
        }
        if (T != 0u) thr_band = fmaxf(thr_band, key_score((u64)T << 32) - eps2);
    }
    int n_band = 0;
#pragma unroll
    for (int j = 0; j < F_NPL; ++j) {
        const bool keep = keys[j] != 0ull && !(key_score(keys[j]) < thr_band);
        const u64 m = __ballot(keep);
        if (keep) list[n_band + __popcll(m & lt_mask)] = (unsigned short)(j * 64 + l);
        n_band += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < F_NPL; ++j) keys[j] = 0ull;
    for (int r0 = 0, rnd = 0; r0 < n_band; r0 += 64, ++rnd) {
        const int e = r0 + l;
        const bool valid = e < n_band;
        const uint32_t prow = valid ? P.live2row[key_row(cq[list[e]])] : 0u;  // image row -> shard row
        const int rows = min(64, n_band - r0);  // wave-uniform
        float sc = 0.0f;
        for (int c0 = 0; c0 < d; c0 += RS_CHUNK) {
            const int len = min(RS_CHUNK, d - c0);         // 256, or 128 for the last piece when d % 256 == 128
            const int lsrc = l * 4 < len ? l * 4 : 0;      // lanes past a short piece re-read its head (never past the row)
            for (int r = 0; r < rows; ++r) {
                const uint32_t row = __builtin_amdgcn_readlane(prow, r);
                const float *src = P.x32 + ((size_t)row * d + c0) + lsrc;
                __builtin_amdgcn_global_load_lds((pipe_glb_t *)src, (pipe_lds_t *)(stage + r * RS_STRIDE), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const float *xs = stage + l * RS_STRIDE;
                const float *qs = qrow_lds + c0;
#pragma unroll 16
                for (int j = 0; j < len / 4; ++j) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(xs + 4 * j);
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(qs + 4 * j);
                    sc = __builtin_fmaf(a[0], xv[0], sc);
                    sc = __builtin_fmaf(a[1], xv[1], sc);
                    sc = __builtin_fmaf(a[2], xv[2], sc);
                    sc = __builtin_fmaf(a[3], xv[3], sc);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next pieces overwrite the stage
            __builtin_amdgcn_wave_barrier();
        }
        const u64 v = valid ? pack_key(sc, prow) : 0ull;
#pragma unroll
        for (int j = 0; j < F_NPL; ++j) keys[j] = (j == rnd) ? v : keys[j];  // rnd is wave-uniform: register file stays static
    }
    if (n_band > P.k) {
        float tau_new;
        select_topk_regs<F_NPL>(keys, P.k, dst, &tau_new);
    } else {
#pragma unroll
        for (int j = 0; j < F_NPL; ++j) {
            const int e = j * 64 + l;
            if (e < P.k) dst[e] = keys[j];
        }
    }
}

// after the filter launch: turn the overflow list into the input of the per-query exact scan
__global__ void __launch_bounds__(256) gather_overflow_kernel(FastCtl *ctl, const QueryStat *qs, const int *ovf_list, const float *q32,
                                                              int d, float *qfb, int *fb_slot) {
    if (qs->bad_image) {  // nothing was filtered: the whole chunk goes to the exact scan
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctl->fb_all = 1;
            ctl->fb_nq = 0;
        }
        return;
    }
    const int cnt = ctl->ovf_count;
    if (cnt == 0) return;
    const int i = blockIdx.x;
    if (cnt > OVF_CAP) {
        if (i == 0 && threadIdx.x == 0) {
            ctl->fb_all = 1;
            ctl->fb_nq = 0;
        }
        return;
    }
    if (i == 0 && threadIdx.x == 0) ctl->fb_nq = cnt;
    if (i >= cnt) return;
    const int q = ovf_list[i];
    for (int k = threadIdx.x * 4; k < d; k += 1024)
        *reinterpret_cast<f32x4 *>(qfb + (size_t)i * d + k) = *reinterpret_cast<const f32x4 *>(q32 + (size_t)q * d + k);
    if (threadIdx.x == 0) fb_slot[q] = i;
}

struct FastPlan {
    int S, Ws;
    int64_t qc;  // queries per launch
    size_t q2_bytes, qn_bytes, qctr_bytes, bias_bytes, cand_bytes, part_bytes, thr_bytes, cnt_bytes, flag_bytes, qfb_bytes, fbk_bytes, fb_bytes, fball_bytes;
};

int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// Tuning knobs of the fast path (include/ance_amd.h lists them): read from the environment ONCE, when the library first
// needs them; ance_reload_env() re-reads (tests and sweeps that change a knob inside one process call it).
struct FastKnobs {
    int splits, window_tiles, dedup, center, share, wait_us, prune_at, prune_growth, debug;
    void load() {
        splits = env_int("ANCE_FAST_SPLITS", 0);
        window_tiles = env_int("ANCE_FAST_WINDOW_TILES", 256);
        dedup = env_int("ANCE_FAST_DEDUP", 1) != 0;
        center = env_int("ANCE_FAST_CENTER", 1) != 0;
        share = env_int("ANCE_FAST_SHARE", 1);
        wait_us = env_int("ANCE_FAST_WINDOW_WAIT_US", 200);
        prune_at = env_int("ANCE_FAST_PRUNE_AT", 512);
        prune_growth = env_int("ANCE_FAST_PRUNE_GROWTH", 150);
        if (prune_growth < 105) prune_growth = 105;
        debug = env_int("ANCE_FAST_DEBUG", 0);
    }
};
FastKnobs &fast_knobs() {
    static FastKnobs k = [] {
        FastKnobs x;
        x.load();
        return x;
    }();
    return k;
}


bool fast_shape_ok(int64_t n, int d, int k) {
    return d >= 128 && d % 128 == 0 && d <= F_MAX_D && k >= 1 && k <= F_MAX_K && n >= 4096 && n < (1ll << 32);
}

bool make_fast_plan(int64_t n, int64_t nq, int d, int k, FastPlan *pl) {
    if (!fast_shape_ok(n, d, k) || nq < 1) return false;
    const int n_tiles = (int)((n + FP - 1) / FP);
    const int64_t nqt = (nq + FQ - 1) / FQ;
    // One workgroup per CU: (query tiles per launch) x (corpus splits) = 256.  More splits = fewer query tiles per
    // XCD (better L2 reuse of the query side) but one more candidate list per query; ANCE_FAST_SPLITS overrides.
    int S = fast_knobs().splits;
    if (S < 1 || S > 32 || (S & (S - 1))) S = 2;
    while (nqt * S < 256 && S < 32) S <<= 1;
    while (S > 1 && (S * 8 > n_tiles || next_pow2((S + DEDUP_MAXC) * k) > 8192)) S >>= 1;
    pl->S = S;
    const int64_t qct = nqt < 256 / S ? nqt : 256 / S;
    pl->qc = qct * FQ;
    // window: ANCE_FAST_WINDOW_TILES corpus tiles of 256 rows (default 256 = 100 MB of fp16 rows at d = 768; 0 = one
    // window, i.e. every split scans its contiguous share as the first version of this kernel did)
    int Wt = fast_knobs().window_tiles;
    if (Wt <= 0 || Wt > n_tiles) Wt = n_tiles;
    pl->Ws = (Wt + S - 1) / S;
    pl->q2_bytes = align_up((size_t)pl->qc * d * sizeof(_Float16), 256);
    pl->qn_bytes = align_up((size_t)2 * pl->qc * sizeof(float), 256);  // |q - mq| and |q|
    pl->qctr_bytes = 256 + align_up((size_t)d * sizeof(float), 256) + align_up((size_t)1024 * d * sizeof(float), 256);  // QueryStat, mq, partials
    pl->bias_bytes = align_up(((size_t)n + FP) * sizeof(float), 256);
    pl->cand_bytes = (size_t)qct * S * FQ * F_C * sizeof(u64);
    pl->part_bytes = align_up((size_t)pl->qc * S * k * sizeof(u64), 256);
    pl->thr_bytes = align_up((size_t)qct * S * FQ * sizeof(float) + (size_t)pl->qc * sizeof(int), 256);  // thr_g + fb_slot (0xFF fill)
    pl->cnt_bytes = align_up((size_t)qct * S * FQ * sizeof(int), 256);
    pl->flag_bytes = align_up(256 + (size_t)pl->qc * sizeof(int) + OVF_CAP * sizeof(int), 256);  // ctl + ovf_flag + ovf_list (0 fill)
    pl->qfb_bytes = align_up((size_t)OVF_CAP * d * sizeof(float), 256);
    pl->fbk_bytes = align_up((size_t)OVF_CAP * k * sizeof(u64), 256);
    const int64_t nqc = nq < pl->qc ? nq : pl->qc;
    pl->fb_bytes = align_up(exact_scan_fallback_bytes(n, OVF_CAP, k), 256);
    pl->fball_bytes = align_up(exact_scan_fallback_bytes(n, nqc, k), 256);
    return pl->fb_bytes > 0 && pl->fball_bytes > 0;
}

size_t fast_search_bytes(const FastPlan &pl) {
    return 256 + pl.q2_bytes + pl.qn_bytes + pl.qctr_bytes + pl.bias_bytes + pl.part_bytes + pl.cand_bytes + pl.thr_bytes + pl.cnt_bytes + pl.flag_bytes + pl.qfb_bytes +
           pl.fbk_bytes + pl.fb_bytes + pl.fball_bytes;
}

unsigned long long *g_fast_stamps = nullptr;

}  // namespace

// measurement hook: while d_stamps != NULL the filter kernel is the instrumented build and every workgroup of the LAST launch
// chunk leaves uint64[8] = {prologue, main loop, filter, prune, sync, block end} ticks of the 100 MHz counter, (qt << 32 | split), XCC id
void set_fast_stamps(unsigned long long *d_stamps) { g_fast_stamps = d_stamps; }  // read by ANCE_MEASURE builds only
void reload_fast_knobs() { fast_knobs().load(); }

// ---- search image --------------------------------------------------------------------------------------
size_t ip_index_bytes(int64_t n, int d) {
    if (!fast_shape_ok(n, d, 1)) return 0;
    return index_layout(n, d).total + 256;
}

int ip_index_build(const float *d_x, int64_t n, int d, void *d_index, size_t index_bytes, hipStream_t st) {
    if (!fast_shape_ok(n, d, 1) || !d_x || !d_index || ((uintptr_t)d_x & 15)) {
        set_last_error("ance_ip_index_build: shape not eligible (d % 128 == 0, 128 <= d <= 2048, 4096 <= n < 2^32)");
        return ANCE_E_INVALID;
    }
    if (index_bytes < ip_index_bytes(n, d)) {
        set_last_error("ance_ip_index_build: index buffer too small");
        return ANCE_E_WORKSPACE;
    }
    const IndexLayout L = index_layout(n, d);
    char *base = reinterpret_cast<char *>(align_up((uintptr_t)d_index, 256));
    DedupHeader *H = reinterpret_cast<DedupHeader *>(base);
    _Float16 *x2 = reinterpret_cast<_Float16 *>(base + L.x2_off);
    uint32_t *live2row = reinterpret_cast<uint32_t *>(base + L.live_off);
    uint32_t *members = reinterpret_cast<uint32_t *>(base + L.mem_off);
    uint8_t *cls = reinterpret_cast<uint8_t *>(base + L.cls_off);
    uint32_t *blk = reinterpret_cast<uint32_t *>(base + L.blk_off);
    u64 *samp = reinterpret_cast<u64 *>(base + L.samp_off);
    float *mu = reinterpret_cast<float *>(base + L.mu_off);
    float *part = reinterpret_cast<float *>(base + L.part_off);
    const bool dedup = fast_knobs().dedup != 0;
    const int center = fast_knobs().center;
    ProfScope ps(PC_PLAN, st);
    (void)hipMemsetAsync(H, 0, 256, st);
    hipLaunchKernelGGL(idx_colsum_kernel, dim3((unsigned)L.n_part), dim3(256), 0, st, d_x, n, d, part);
    hipLaunchKernelGGL(idx_mean_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, part, L.n_part, n, d, center, mu);
    if (dedup) {
        hipLaunchKernelGGL(idx_sample_hash_kernel, dim3(IDX_SAMPLES / 4), dim3(256), 0, st, d_x, n, d, samp);
        hipLaunchKernelGGL(idx_find_classes_kernel, dim3(1), dim3(256), 0, st, samp, n, H);
        const unsigned cb = (unsigned)(((n + 63) / 64 + 3) / 4 < 4096 ? ((n + 63) / 64 + 3) / 4 : 4096);
        hipLaunchKernelGGL(idx_classify_kernel, dim3(cb), dim3(256), 0, st, d_x, n, d, H, cls);
        hipLaunchKernelGGL(idx_count_kernel, dim3((unsigned)L.nb), dim3(256), 0, st, n, L.nb, H, cls, blk);
        hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, L.nb, H, blk);
    }
    hipLaunchKernelGGL(idx_compact_round_kernel, dim3((unsigned)L.nb), dim3(256), 0, st, d_x, n, d, L.nb, H, cls, blk, mu, x2,
                       live2row, members);
    hipLaunchKernelGGL(idx_stamp_kernel, dim3(1), dim3(64), 0, st, H, n, d, d_x);  // last: marks the build complete
    return check_launch("ance_ip_index_build");
}

size_t ip_topk_fast_workspace_bytes(int64_t n, int64_t nq, int d, int k, bool with_index) {
    FastPlan pl;
    if (!make_fast_plan(n, nq, d, k, &pl)) return 0;
    return fast_search_bytes(pl) + (with_index ? ip_index_bytes(n, d) : 0);
}

// d_index: a search image built by ip_index_build for exactly (d_x, n, d), or NULL (then it is built inside the workspace)
int ip_topk_fast(const float *d_x, int64_t n, int64_t row_base, const void *d_index, const float *d_q, int64_t nq, int d, int k,
                 float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, hipStream_t st) {
    FastPlan pl;
    if (!make_fast_plan(n, nq, d, k, &pl)) {
        set_last_error("ip_topk_fast: shape not eligible");
        return ANCE_E_INVALID;
    }
    if (workspace_bytes < ip_topk_fast_workspace_bytes(n, nq, d, k, d_index == nullptr)) {
        set_last_error("ip_topk_fast: workspace too small");
        return ANCE_E_WORKSPACE;
    }
    char *p = reinterpret_cast<char *>(align_up((uintptr_t)d_workspace, 256));
    _Float16 *q2 = reinterpret_cast<_Float16 *>(p); p += pl.q2_bytes;
    float *qn = reinterpret_cast<float *>(p); p += pl.qn_bytes;
    float *qn_o = qn + pl.qc;
    QueryStat *qstat = reinterpret_cast<QueryStat *>(p);
    float *mq = reinterpret_cast<float *>(p + 256);
    float *qpart = reinterpret_cast<float *>(p + 256 + align_up((size_t)d * sizeof(float), 256));
    p += pl.qctr_bytes;
    float *bias = reinterpret_cast<float *>(p); p += pl.bias_bytes;
    u64 *part = reinterpret_cast<u64 *>(p); p += pl.part_bytes;
    u64 *cand = reinterpret_cast<u64 *>(p); p += pl.cand_bytes;
    char *ff_area = p; p += pl.thr_bytes;    // 0xFF-filled per chunk
    int *cnt_g = reinterpret_cast<int *>(p); p += pl.cnt_bytes;
    char *zero_area = p; p += pl.flag_bytes;  // zero-filled per chunk
    float *qfb = reinterpret_cast<float *>(p); p += pl.qfb_bytes;
    u64 *fb_keys = reinterpret_cast<u64 *>(p); p += pl.fbk_bytes;
    void *fb_ws = p; p += pl.fb_bytes;
    void *fball_ws = p; p += pl.fball_bytes;
    if (!d_index) {
        void *own = p;
        const int rc = ip_index_build(d_x, n, d, own, ip_index_bytes(n, d), st);
        if (rc) return rc;
        d_index = own;
    }
    const IndexLayout L = index_layout(n, d);
    const char *ibase = reinterpret_cast<const char *>(align_up((uintptr_t)d_index, 256));
    const DedupHeader *H = reinterpret_cast<const DedupHeader *>(ibase);
    const uint32_t *members = reinterpret_cast<const uint32_t *>(ibase + L.mem_off);

    const int64_t qct = pl.qc / FQ;
    float *thr_g = reinterpret_cast<float *>(ff_area);
    int *fb_slot = reinterpret_cast<int *>(ff_area + (size_t)qct * pl.S * FQ * sizeof(float));
    FastCtl *ctl = reinterpret_cast<FastCtl *>(zero_area);
    int *ovf_flag = reinterpret_cast<int *>(zero_area + 256);
    int *ovf_list = ovf_flag + pl.qc;

    static unsigned long long attr_done = 0;
    if (attr_needed(&attr_done)) {
        bool ok = true;
        for (const void *fn : {reinterpret_cast<const void *>(ip_topk_fast_kernel<false, false>),
#ifdef ANCE_MEASURE
                               reinterpret_cast<const void *>(ip_topk_fast_kernel<true, false>),
                               reinterpret_cast<const void *>(ip_topk_fast_kernel<true, true>),
#endif
                               reinterpret_cast<const void *>(ip_topk_fast_kernel<false, true>)})
            ok = ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES) == hipSuccess;
        if (!ok)
            return check_launch("ip_topk_fast attr");
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(rescore_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)rescore_lds_bytes(F_MAX_D)) != hipSuccess)
            return check_launch("rescore attr");
        attr_mark(&attr_done);
    }
    EpsConst eps;
    eps.rel_c = 1.25f * (9.765625e-4f + 1.1f * d * 5.9604645e-8f);
    eps.acc_m = 1.25f * 2.1f * d * 5.9604645e-8f;
    eps.cen = 1.25f * 1.1920929e-7f;
    eps.abs_c = 1.25f * 5.9604645e-8f * sqrtf((float)d);
    eps.chain_o = 1.25f * d * 5.9604645e-8f;
    // ---- the mean query of this call and its per-row share of every score (skipped on the device when |mq| is small) ----
    {
        const IndexLayout Li = index_layout(n, d);
        const char *ib = reinterpret_cast<const char *>(align_up((uintptr_t)d_index, 256));
        const int n_part_q = (int)(nq < 1024 ? nq : 1024);
        ProfScope ps(PC_PLAN, st);
        hipLaunchKernelGGL(idx_colsum_kernel, dim3((unsigned)n_part_q), dim3(256), 0, st, d_q, nq, d, qpart);
        hipLaunchKernelGGL(idx_mean_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, qpart, n_part_q, nq, d,
                           fast_knobs().center, mq);
        hipLaunchKernelGGL(query_mean_decide_kernel, dim3(1), dim3(256), 0, st, mq, d, reinterpret_cast<const DedupHeader *>(ib), qstat, n, d_x);
        (void)hipMemsetAsync(bias, 0, pl.bias_bytes, st);
        hipLaunchKernelGGL(row_bias_kernel, dim3(4096), dim3(256), 0, st, d_x, d, reinterpret_cast<const DedupHeader *>(ib),
                           reinterpret_cast<const uint32_t *>(ib + Li.live_off), reinterpret_cast<const float *>(ib + Li.mu_off), mq,
                           qstat, bias);
    }
    const FastKnobs &kn = fast_knobs();
    const int share = kn.share, wait_us = kn.wait_us, prune_at = kn.prune_at, prune_growth = kn.prune_growth;
#ifdef ANCE_MEASURE
    unsigned long long *stamps = g_fast_stamps;  // measurement hook (ance_debug_search_stamps)
#endif
    for (int64_t q0 = 0; q0 < nq; q0 += pl.qc) {
        const int64_t nqc = (nq - q0) < pl.qc ? (nq - q0) : pl.qc;
        (void)hipMemsetAsync(ff_area, 0xFF, pl.thr_bytes, st);
        (void)hipMemsetAsync(zero_area, 0, pl.flag_bytes, st);
        {
            ProfScope ps(PC_PLAN, st);
            hipLaunchKernelGGL(round_rows_kernel, dim3((unsigned)((nqc + 3) / 4 < 8192 ? (nqc + 3) / 4 : 8192)), dim3(256), 0, st,
                               d_q + (size_t)q0 * d, nqc, d, mq, q2, qn, qn_o);
        }
        FastParams P;
        P.q2 = q2; P.x2 = reinterpret_cast<const _Float16 *>(ibase + L.x2_off); P.q32 = d_q + (size_t)q0 * d; P.x32 = d_x; P.qnorm_c = qn; P.qnorm_o = qn_o; P.qstat = qstat; P.bias = bias;
        P.hdr = H; P.live2row = reinterpret_cast<const uint32_t *>(ibase + L.live_off);
        P.nq = (uint32_t)nqc; P.d = d; P.k = k; P.S = pl.S; P.Ws = pl.Ws;
        P.n_qt = (int)((nqc + FQ - 1) / FQ);
        P.share = share && pl.S > 1; P.wait_ticks = (unsigned)(wait_us > 0 ? wait_us * 100 : 0);
        P.eps = eps; P.cand = cand; P.part = part; P.thr_g = thr_g; P.ctl = ctl;
        P.ovf_flag = ovf_flag; P.ovf_list = ovf_list; P.cnt_g = cnt_g;
        P.prune_at = prune_at > k + 64 ? prune_at : k + 64;
        if (P.prune_at > F_C - FP) P.prune_at = F_C - FP;
        P.prune_growth = prune_growth;
#ifdef ANCE_MEASURE
        P.stamps = stamps; P.dbg = kn.debug;
#else
        P.stamps = nullptr; P.dbg = 0;
#endif
        const int gq = 32 / pl.S;
        const int groups = (P.n_qt + gq - 1) / gq;
        const unsigned blocks = (unsigned)((groups + 7) / 8 * 8) * 32u;
        {
            ProfScope ps(PC_SCAN, st, 2.0 * (double)nqc * (double)n * (double)d);
#ifdef ANCE_MEASURE  // the instrumented builds (per-workgroup time stamps, timing experiments) exist in the measurement library only
            if (stamps) {
                hipLaunchKernelGGL((ip_topk_fast_kernel<true, false>), dim3(blocks), dim3(F_THREADS), F_LDS_BYTES, st, P);
                hipLaunchKernelGGL((ip_topk_fast_kernel<true, true>), dim3(blocks), dim3(F_THREADS), F_LDS_BYTES, st, P);
            } else
#endif
            {
                hipLaunchKernelGGL((ip_topk_fast_kernel<false, false>), dim3(blocks), dim3(F_THREADS), F_LDS_BYTES, st, P);
                hipLaunchKernelGGL((ip_topk_fast_kernel<false, true>), dim3(blocks), dim3(F_THREADS), F_LDS_BYTES, st, P);
            }
        }
        {
            RescoreParams R;
            R.q32 = P.q32; R.x32 = d_x; R.qnorm_c = qn; R.qnorm_o = qn_o; R.qstat = qstat; R.hdr = H; R.live2row = P.live2row; R.cand = cand; R.cnt_g = cnt_g; R.thr_g = thr_g;
            R.part = part; R.nq = P.nq; R.d = d; R.k = k; R.S = pl.S; R.eps = eps;
            ProfScope ps(PC_RESCORE, st);
            hipLaunchKernelGGL(rescore_kernel, dim3((unsigned)(P.n_qt * pl.S * FQ)), dim3(64), rescore_lds_bytes(d), st, R);
        }
        // queries whose buffers overflowed: redone by the exact scan, one by one (<= OVF_CAP) or as a whole chunk
        hipLaunchKernelGGL(gather_overflow_kernel, dim3(OVF_CAP), dim3(256), 0, st, ctl, qstat, ovf_list, d_q + (size_t)q0 * d, d, qfb, fb_slot);
        const u64 *fb_part = nullptr, *fball_part = nullptr;
        int fb_m = 0, fball_m = 0;
        int rc = exact_scan_fallback(d_x, n, qfb, OVF_CAP, OVF_CAP, d, k, fb_ws, nullptr, &ctl->fb_nq, &fb_part, &fb_m, st);
        if (rc) return rc;
        rc = launch_reduce_keys(fb_part, OVF_CAP, fb_m, k, fb_keys, &ctl->fb_nq, st);
        if (rc) return rc;
        rc = exact_scan_fallback(d_x, n, d_q + (size_t)q0 * d, nqc, nq < pl.qc ? nq : pl.qc, d, k, fball_ws, &ctl->fb_all, nullptr,
                                 &fball_part, &fball_m, st);
        if (rc) return rc;
        FinalizeAlt alt;
        alt.sel_all = &ctl->fb_all; alt.all_keys = fball_part; alt.all_m = fball_m;
        alt.slot = fb_slot; alt.slot_keys = fb_keys; alt.slot_m = k;
        alt.dd = H; alt.members = members;
        rc = launch_finalize_keys(part, nqc, pl.S * k, k, row_base, d_out_d + (size_t)q0 * k, d_out_i + (size_t)q0 * k, st, &alt);
        if (rc) return rc;
    }
    return check_launch("ip_topk_fast");
}

}  // namespace ance

// include/ance_amd.h: how many ance_ip_topk_indexed calls (per launch chunk) on the current device ignored their image.
// Synchronises the device (a diagnostic, not a data-path call).
extern "C" int ance_search_bad_image_calls(unsigned long long *out) {
    using namespace ance;
    if (!out) {
        set_last_error("ance_search_bad_image_calls: invalid argument");
        return ANCE_E_INVALID;
    }
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bad_image_calls), sizeof(*out)) != hipSuccess) return check_launch("ance_search_bad_image_calls");
    return ANCE_OK;
}
