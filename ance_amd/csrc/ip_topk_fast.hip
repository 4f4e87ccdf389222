// Two-precision exact inner-product top-k for gfx950: the fast path behind ance_ip_topk.
//
// gfx950 has no reduced-precision path for fp32 inputs (no xf32), and the exact fp32 MFMA runs at
// 1/16 of the fp16 rate.  This kernel gets the fp16 rate WITHOUT giving up bit-exact results:
//
//   1. corpus and queries are rounded once to fp16 (xh = fp16(x));
//   2. an approximate score  s~ = qh . xh  is a plain fp16 GEMM on the 256 x 256 x 64 direct-to-LDS
//      main loop of gemm256_f16.hip (queries are the "m" side, so a lane owns a query);
//   3. with eps a rigorous bound on |s~ - s| (below) and t~ the k-th best APPROXIMATE score seen so
//      far, a row with s~ < t~ - 2 eps can never be in the exact top-k (k rows have s >= t~ - eps
//      > its s), so the per-query buffers keep exactly the rows with s~ >= t~ - 2 eps: about
//      k + 2 eps * density rows (~270 for k = 200 on LayerNorm-distributed rows);
//   4. when a block has scanned its corpus split, every kept row of every query is re-scored with
//      the exact fp32 fmaf chain over k ascending (the contract of oracle/ip_topk_ref.c) -- one
//      query per wave at a time, its fp32 row broadcast from LDS, 64 rows in flight -- and the exact
//      top-k under (score desc, row asc) is selected from exact keys.  Output scores and ids are
//      therefore bit-identical to the fp32-MFMA scan.
//   If a buffer cannot be pruned below its capacity (more than ~1,500 rows inside one 2 eps band:
//   pathologically clustered scores) the block raises a device-side flag and the launch chunk is
//   redone by the exact scan kernel (conditional on the flag, no host synchronisation).
//
// Error bound, B = sum_k |q_k||x_k| <= ||q|| ||x||, normal-range fp16 (|v| >= 2^-14):
//   rounding q and x to fp16:  |q.x - qh.xh| <= (2^-11 + 2^-11 + 2^-22) B
//   fp32 accumulation inside / between MFMAs: <= 1.1 d 2^-24 B ; the exact chain itself: <= d 2^-24 B
//   fp16 subnormal inputs add at most 2^-25 per element: <= 2^-25 sqrt(d) (||q|| + ||x||)
// eps = 1.25 * [ (2^-10 + 2.1 d 2^-24) ||q|| max||x||  +  2^-24 sqrt(d) (||q|| + max||x||) ].
#include "common.h"
#include "topk_common.h"
#include "pipe256.h"

namespace ance {
namespace {

constexpr int FQ = 256, FP = 256, FK = 64;
constexpr int F_OPER_HALVES = 256 * FK;
constexpr int F_STAGE_HALVES = 2 * F_OPER_HALVES;
constexpr int F_THREADS = 512;
constexpr int F_NPL = 32;
constexpr int F_C = F_NPL * 64;  // 2048 buffered rows per (block, query); a tile can add 256
constexpr size_t F_LDS_BYTES = (size_t)2 * F_STAGE_HALVES * sizeof(_Float16) + 3 * FQ * 4 + 16;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

struct FastParams {
    const _Float16 *q2;  // [nq, d]  fp16(q)
    const _Float16 *x2;  // [n, d]   fp16(x)
    const float *q32;    // [nq, d]
    const float *x32;    // [n, d]
    const float *qnorm;  // [nq]
    const float *xmax;   // [1] max row norm of the shard
    uint32_t n, nq;
    int d, k, S, n_qt, n_tiles_p, tiles_per_split;
    float slack_rel, slack_abs;
    u64 *cand;  // [n_qt * S][FQ][F_C]
    u64 *part;  // [nq][S][k]
    int *overflow;  // [1] raised when a buffer cannot be pruned (the chunk is then redone exactly)
};

// fp16 rounding + row norm (+ shard max): one wave per row, grid-stride, ONE atomic per block (a
// single word saturates near 88 atomics/us: one per row would cost 100 ms on 8.8 M rows)
__global__ void __launch_bounds__(256) round_rows_kernel(const float *src, int64_t rows, int d, _Float16 *dst, float *norm,
                                                         unsigned int *maxnorm_bits) {
    __shared__ float wmax[4];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    float mymax = 0.0f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
        const float *s = src + (size_t)row * d;
        _Float16 *hi = dst + (size_t)row * d;
        float q = 0.f;
        for (int k = l * 4; k < d; k += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(s + k);
            f16x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = (_Float16)v[j];
                q = fmaf(v[j], v[j], q);
            }
            *reinterpret_cast<f16x4 *>(hi + k) = h;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
        const float nr = sqrtf(q) * 1.0001f;  // the norm only feeds an upper bound
        if (l == 0 && norm) norm[row] = nr;
        mymax = fmaxf(mymax, nr);
    }
    if (maxnorm_bits) {
        if (l == 0) wmax[w] = mymax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            atomicMax(maxnorm_bits, __builtin_bit_cast(unsigned int, m));
        }
    }
}

// exact score: fp32 fmaf chain over k ascending from +0 (== v_mfma_f32_32x32x2_f32, == the oracle).
// q comes from LDS (every lane of the wave works on the same query: broadcast reads), x from global
// memory with 16 independent 16-byte loads in flight per lane.
__device__ __forceinline__ float exact_ip_lds(const float *q_lds, const float *x, int d) {
    float s = 0.0f;
    for (int k0 = 0; k0 < d; k0 += 64) {
        f32x4 xv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) xv[j] = *reinterpret_cast<const f32x4 *>(x + k0 + 4 * j);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(q_lds + k0 + 4 * j);
            s = __builtin_fmaf(a[0], xv[j][0], s);
            s = __builtin_fmaf(a[1], xv[j][1], s);
            s = __builtin_fmaf(a[2], xv[j][2], s);
            s = __builtin_fmaf(a[3], xv[j][3], s);
        }
    }
    return s;
}

// Source policy of the streamed main loop (pipe256.h): queries at fixed per-lane pointers, corpus rows
// addressed from the tile origin p0 (clamped to the last row; rows past n are masked in the filter).
// K-tile t >= NK belongs to the next corpus tile (p0 + 256).
struct FastSrc {
    const _Float16 *q2, *x2;  // uniform bases
    uint32_t qoff[2][2];      // per-lane offsets (halves) of the query pieces: (clamped row) * d + chunk
    uint32_t p0, n_last;
    int rs, ch, w, d, NK;     // rs = lane >> 3 (row inside a piece), ch = source chunk (same for both pieces)
    template <int TYPE, int J>
    __device__ __forceinline__ const _Float16 *addr(int t) const {
        const bool nxt = t >= NK;
        const int kk = nxt ? t - NK : t;
        if constexpr (TYPE < 2) {
            return q2 + (qoff[TYPE][J] + (uint32_t)(kk * 64));
        } else {
            const int r = (w + 8 * J) * 8 + rs;  // row of the half-tile
            const uint32_t row = min(p0 + (nxt ? 256u : 0u) + (uint32_t)pipe_b_tile_row(TYPE - 2, r), n_last);
            return x2 + ((size_t)row * d + (uint32_t)(kk * 64 + ch));
        }
    }
};

__global__ void __launch_bounds__(F_THREADS, 2) ip_topk_fast_kernel(const FastParams P) {
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
    const int l = tid & 63, g = l >> 5, i = l & 31;
    const int wm = w >> 2, wn = w & 3;  // wave tile: 128 queries x 64 passages
    const uint32_t q0 = (uint32_t)qt * FQ;
    const int t0 = split * P.tiles_per_split;
    const int t1 = min(t0 + P.tiles_per_split, P.n_tiles_p);
    const int d = P.d;
    u64 *cand = P.cand + ((size_t)qt * P.S + split) * (size_t)FQ * F_C;

    if (tid < FQ) {
        const uint32_t qg = q0 + tid;
        thr_s[tid] = -INFINITY;
        cnt_s[tid] = 0;
        const float qn = qg < P.nq ? P.qnorm[qg] : 0.0f, xm = P.xmax[0];
        // the bound assumes no fp16 overflow: |x_j| <= ||x||, so norms <= 65504 exclude it.  Otherwise eps = inf
        // keeps every row until the buffer overflows and the chunk is redone by the exact scan.
        const bool fp16_ok = qn <= 65504.0f && xm <= 65504.0f;  // false for NaN too
        eps2_s[tid] = fp16_ok ? 2.0f * (P.slack_rel * qn * xm + P.slack_abs * (qn + xm)) : INFINITY;
    }

    // ---- main loop: the ping-pong pipeline of pipe256.h, streamed across this split's corpus tiles ----
    // A operand = the block's 256 queries (re-read from L2 for every corpus tile), B operand = corpus
    // rows.  K-tile index t of the tile being computed; t >= NK addresses the next corpus tile, so the
    // LDS-DMA prefetch (5-6 phases ahead) runs through the filter step into the next tile.
    Pipe256T<FastSrc> pipe;
    pipe.init(smem, w, l);
    {
        FastSrc &S = pipe.S;
        S.q2 = P.q2; S.x2 = P.x2; S.d = d; S.n_last = P.n - 1; S.NK = d / FK; S.p0 = (uint32_t)t0 * FP;
        S.w = w; S.rs = l >> 3;
        S.ch = pipe_stage_chunk(pipe_stage_row(w, l, 0), l);  // rows of piece 1 are 64 further: same swizzle
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = pipe_stage_row(w, l, j);
                S.qoff[h][j] = min(q0 + (uint32_t)pipe_a_tile_row(h, r), P.nq - 1) * (uint32_t)d + S.ch;
            }
    }
    const int NK = d / FK;
    int *epoch_s = cnt_s + FQ;  // last tile (1-based) in which some wave asked for a prune
    if (tid == 0) *epoch_s = 0;
    pipe.prologue();  // also publishes thr_s / cnt_s / eps2_s / epoch_s

    for (int t = t0; t < t1; ++t) {
        const uint32_t p0 = (uint32_t)t * FP;
        pipe.S.p0 = p0;
        f32x16 acc[2][4];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = f32x16{0};
        pipe.enter();
        if (t + 1 < t1) pipe.tiles_streaming(NK, acc);
        else pipe.tiles_final(NK, acc);
        pipe.leave();

        // ---- filter: keep every row whose approximate score is within 2 eps of the k-th best -------
        // acc[x][y][r]: passage = p0 + wn*64 + x*32 + (r&3) + 8 (r>>2) + 4 g ; query = q0 + wm*128 + y*32 + i
        const uint32_t pw0 = p0 + wn * 64 + 4 * g;
        const bool ragged = p0 + FP > P.n;  // uniform: rows past n were staged as copies of row n-1
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int ql = wm * 128 + y * 32 + i;
            const bool qv = (q0 + ql) < P.nq;
            const float thr = thr_s[ql];  // -inf until the first prune
            // Once the threshold is set almost no row passes (about k + band of 8.8 M per query): take the
            // maximum of the lane's 32 scores first and skip the whole group when no lane of the wave has
            // a candidate -- 16 v_max3 instead of 32 compare-and-branch sequences.  Scores are finite
            // (fp16_ok above), so the maximum loses nothing.  Not on a ragged last tile (clamped rows).
            if (!ragged) {
                float mx = acc[0][y][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[0][y][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[1][y][r]);
                if (__ballot(qv && !(mx < thr)) == 0ull) continue;
            }
            u64 *cq = cand + (size_t)ql * F_C;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t prow = pw0 + x * 32 + (r & 3) + 8 * (r >> 2);
                    const float sc = acc[x][y][r];
                    if (qv && prow < P.n && !(sc < thr)) {
                        const int sl = atomicAdd(&cnt_s[ql], 1);
                        cq[sl] = pack_key(sc, prow);
                    }
                }
        }
        // ---- prune buffers that could overflow on the next tile (approximate keys) --------------------
        // Barriers here are raw s_barriers: a __syncthreads would drain the LDS-DMA prefetch of the next
        // tile (vmcnt(0)).  Only when some buffer really needs a prune (a few times per query, early in
        // the scan) do all waves retire their candidate stores before anybody reads them back.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int c32 = cnt_s[w * 32 + (l & 31)];
            if (__ballot(c32 > F_C - FP) != 0ull && l == 0) *epoch_s = t + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (*epoch_s != t + 1) continue;  // block-uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int qq = 0; qq < 32; ++qq) {
            const int ql = w * 32 + qq;
            const int n_c = __builtin_amdgcn_readfirstlane(cnt_s[ql]);
            if (n_c > F_C - FP) {
                u64 *cq = cand + (size_t)ql * F_C;
                u64 keys[F_NPL];
#pragma unroll
                for (int j = 0; j < F_NPL; ++j) {
                    const int idx = j * 64 + l;
                    keys[j] = (idx < n_c) ? cq[idx] : 0ull;
                }
                u64 T = 0;  // k-th largest approximate key
                for (int bit = 63; bit >= 0; --bit) {
                    const u64 t2 = T | (1ull << bit);
                    int ge = 0;
#pragma unroll
                    for (int j = 0; j < F_NPL; ++j) ge += __popcll(__ballot(keys[j] >= t2));
                    if (ge >= P.k) T = t2;
                }
                const float thr_new = key_score(T) - eps2_s[ql];
                int base = 0;
                const u64 lt_mask = (1ull << l) - 1ull;
#pragma unroll
                for (int j = 0; j < F_NPL; ++j) {
                    const bool keep = keys[j] != 0ull && !(key_score(keys[j]) < thr_new);
                    const u64 m = __ballot(keep);
                    if (keep) cq[base + __popcll(m & lt_mask)] = keys[j];
                    base += __popcll(m);
                }
                if (l == 0) {
                    if (base > F_C - FP) {  // more than 1,792 rows inside one 2 eps band: give up on this chunk
                        atomicExch(P.overflow, 1);
                        base = F_C - FP;
                    }
                    cnt_s[ql] = base;
                    thr_s[ql] = thr_new;
                }
            }
        }
        // thr_s / cnt_s updates are published by the barriers of the next tile's main loop
    }
    __syncthreads();

    // ---- block end: final band prune, exact re-scoring of the band, exact top-k of the split -----------
    // A buffer ends the scan with 700-1,700 rows (everything above the LAST threshold), but only the rows
    // within 2 eps of the final k-th best approximate score (about k + 66) can be in the exact top-k.
    // Re-scoring costs a 3 KB row read each, so the band is cut first: the k-th approximate key by the
    // same radix select, the survivors' buffer positions compacted into an LDS list (dense: every lane
    // re-scores one row per round), exact keys kept in registers and selected from there.
    float *qrow_lds = smem_f + w * 1024;  // 4 KiB per wave in the idle stage area (d <= 1024)
    unsigned short *list = reinterpret_cast<unsigned short *>(smem_f + 8 * 1024) + w * F_C;  // 4 KiB per wave
    const u64 lt_mask = (1ull << l) - 1ull;
    for (int qq = 0; qq < 32; ++qq) {
        const int ql = w * 32 + qq;
        const uint32_t qg = q0 + ql;
        if (qg >= P.nq) continue;  // wave-uniform
        const int n_c = __builtin_amdgcn_readfirstlane(cnt_s[ql]);
        const u64 *cq = cand + (size_t)ql * F_C;
        u64 *dst = P.part + ((size_t)qg * P.S + split) * (size_t)P.k;
        const float *qsrc = P.q32 + (size_t)qg * d;
        for (int k4 = l * 4; k4 < d; k4 += 256) *reinterpret_cast<f32x4 *>(qrow_lds + k4) = *reinterpret_cast<const f32x4 *>(qsrc + k4);
        u64 keys[F_NPL];
#pragma unroll
        for (int j = 0; j < F_NPL; ++j) {
            const int idx = j * 64 + l;
            keys[j] = (idx < n_c) ? cq[idx] : 0ull;
        }
        float thr_band = -INFINITY;
        if (n_c > P.k) {
            u64 T = 0;  // k-th largest approximate key
            for (int bit = 63; bit >= 0; --bit) {
                const u64 t2 = T | (1ull << bit);
                int ge = 0;
#pragma unroll
                for (int j = 0; j < F_NPL; ++j) ge += __popcll(__ballot(keys[j] >= t2));
                if (ge >= P.k) T = t2;
            }
            thr_band = key_score(T) - eps2_s[ql];
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
        // exact keys, dense: round r, lane l re-scores list entry 64 r + l
#pragma unroll
        for (int j = 0; j < F_NPL; ++j) keys[j] = 0ull;
        for (int r = 0; r * 64 < n_band; ++r) {
            const int e = r * 64 + l;
            u64 v = 0ull;
            if (e < n_band) {
                const uint32_t prow = key_row(cq[list[e]]);
                v = pack_key(exact_ip_lds(qrow_lds, P.x32 + (size_t)prow * d, d), prow);
            }
#pragma unroll
            for (int j = 0; j < F_NPL; ++j) keys[j] = (j == r) ? v : keys[j];  // r is wave-uniform: register file stays static
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
        __builtin_amdgcn_wave_barrier();  // the next query reuses qrow_lds and list
    }
}

struct FastPlan {
    int S, n_tiles_p, tiles_per_split;
    int64_t qc;  // queries per launch
    size_t x2_bytes, q2_bytes, qn_bytes, cand_bytes, part_bytes, fallback_bytes;
};

bool make_fast_plan(int64_t n, int64_t nq, int d, int k, FastPlan *pl) {
    if (d < 128 || d % 128 || k < 1 || k > 256 || n < 4096 || n >= (1ll << 32) || nq < 1) return false;
    pl->n_tiles_p = (int)((n + FP - 1) / FP);
    const int64_t nqt = (nq + FQ - 1) / FQ;
    const int64_t qct = nqt < 128 ? nqt : 128;
    pl->qc = qct * FQ;
    int S = 1;
    while (qct * S < 256 && S < 32) S <<= 1;  // one workgroup per CU: more splits only add prologues and re-scoring
    if (const char *e = getenv("ANCE_FAST_SPLITS")) {  // tuning knob (power of two, 1..32)
        const int v = atoi(e);
        if (v >= 1 && v <= 32 && (v & (v - 1)) == 0) S = v;
    }
    while (S > 1 && (S * 8 > pl->n_tiles_p || next_pow2(S * k) > 8192)) S >>= 1;
    pl->S = S;
    pl->tiles_per_split = (pl->n_tiles_p + S - 1) / S;
    pl->x2_bytes = align_up((size_t)n * d * sizeof(_Float16), 256);
    pl->q2_bytes = align_up((size_t)pl->qc * d * sizeof(_Float16), 256);
    pl->qn_bytes = align_up((size_t)pl->qc * sizeof(float), 256);
    pl->cand_bytes = (size_t)qct * S * FQ * F_C * sizeof(u64);
    pl->part_bytes = align_up((size_t)pl->qc * S * k * sizeof(u64), 256);
    const int64_t nqc = nq < pl->qc ? nq : pl->qc;
    pl->fallback_bytes = align_up(exact_scan_fallback_bytes(n, nqc, k), 256);
    return pl->fallback_bytes > 0;
}

}  // namespace

size_t ip_topk_fast_workspace_bytes(int64_t n, int64_t nq, int d, int k) {
    FastPlan pl;
    if (!make_fast_plan(n, nq, d, k, &pl)) return 0;
    return 256 + pl.x2_bytes + 256 + pl.q2_bytes + pl.qn_bytes + pl.cand_bytes + pl.part_bytes + pl.fallback_bytes;
}

int ip_topk_fast(const float *d_x, int64_t n, int64_t row_base, const float *d_q, int64_t nq, int d, int k, float *d_out_d,
                 int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, hipStream_t st) {
    FastPlan pl;
    if (!make_fast_plan(n, nq, d, k, &pl)) {
        set_last_error("ip_topk_fast: shape not eligible");
        return ANCE_E_INVALID;
    }
    if (workspace_bytes < ip_topk_fast_workspace_bytes(n, nq, d, k)) {
        set_last_error("ip_topk_fast: workspace too small");
        return ANCE_E_WORKSPACE;
    }
    char *p = reinterpret_cast<char *>(align_up((uintptr_t)d_workspace, 256));
    _Float16 *x2 = reinterpret_cast<_Float16 *>(p); p += pl.x2_bytes;
    unsigned int *xmax = reinterpret_cast<unsigned int *>(p); p += 256;
    _Float16 *q2 = reinterpret_cast<_Float16 *>(p); p += pl.q2_bytes;
    float *qn = reinterpret_cast<float *>(p); p += pl.qn_bytes;
    u64 *part = reinterpret_cast<u64 *>(p); p += pl.part_bytes;
    u64 *cand = reinterpret_cast<u64 *>(p); p += pl.cand_bytes;
    void *fb_ws = p;
    int *overflow = reinterpret_cast<int *>(xmax) + 16;  // same 256-byte cell as the max norm

    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(ip_topk_fast_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)F_LDS_BYTES) != hipSuccess)
            return check_launch("ip_topk_fast attr");
        attr_done = true;
    }
    (void)hipMemsetAsync(xmax, 0, 128, st);
    {
        ProfScope ps(PC_PLAN, st);
        hipLaunchKernelGGL(round_rows_kernel, dim3((unsigned)((n + 3) / 4 < 8192 ? (n + 3) / 4 : 8192)), dim3(256), 0, st, d_x, n, d, x2,
                           (float *)nullptr, xmax);
    }
    const float slack_rel = 1.25f * (9.765625e-4f + 2.1f * d * 5.9604645e-8f);
    const float slack_abs = 1.25f * 5.9604645e-8f * sqrtf((float)d);
    for (int64_t q0 = 0; q0 < nq; q0 += pl.qc) {
        const int64_t nqc = (nq - q0) < pl.qc ? (nq - q0) : pl.qc;
        {
            ProfScope ps(PC_PLAN, st);
            hipLaunchKernelGGL(round_rows_kernel, dim3((unsigned)((nqc + 3) / 4 < 8192 ? (nqc + 3) / 4 : 8192)), dim3(256), 0, st,
                               d_q + (size_t)q0 * d, nqc, d, q2,
                               qn, (unsigned int *)nullptr);
        }
        FastParams P;
        P.q2 = q2; P.x2 = x2; P.q32 = d_q + (size_t)q0 * d; P.x32 = d_x; P.qnorm = qn;
        P.xmax = reinterpret_cast<const float *>(xmax);
        P.n = (uint32_t)n; P.nq = (uint32_t)nqc; P.d = d; P.k = k; P.S = pl.S;
        P.n_qt = (int)((nqc + FQ - 1) / FQ); P.n_tiles_p = pl.n_tiles_p; P.tiles_per_split = pl.tiles_per_split;
        P.slack_rel = slack_rel; P.slack_abs = slack_abs; P.cand = cand; P.part = part; P.overflow = overflow;
        const int gq = 32 / pl.S;
        const int groups = (P.n_qt + gq - 1) / gq;
        const unsigned blocks = (unsigned)((groups + 7) / 8 * 8) * 32u;
        {
            ProfScope ps(PC_SCAN, st, 2.0 * (double)nqc * (double)n * (double)d);
            hipLaunchKernelGGL(ip_topk_fast_kernel, dim3(blocks), dim3(F_THREADS), F_LDS_BYTES, st, P);
        }
        // pathological score clustering: the chunk is redone by the exact scan, device-side conditional
        const u64 *fb_part = nullptr;
        int fb_m = 0;
        int rc = exact_scan_fallback(d_x, n, d_q + (size_t)q0 * d, nqc, d, k, fb_ws, overflow, &fb_part, &fb_m, st);
        if (rc) return rc;
        rc = launch_finalize_keys(part, nqc, pl.S * k, k, row_base, d_out_d + (size_t)q0 * k, d_out_i + (size_t)q0 * k, st, overflow,
                                  fb_part, fb_m);
        if (rc) return rc;
        if (q0 + pl.qc < nq) (void)hipMemsetAsync(overflow, 0, 4, st);
    }
    return check_launch("ip_topk_fast");
}

}  // namespace ance
