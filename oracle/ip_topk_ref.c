/*
 * CPU oracle for the exact inner-product top-k search -- TEST INFRASTRUCTURE ONLY.
 *
 * Restates what the reference asks of faiss.IndexFlatIP at
 *   drivers/run_ann_data_gen.py:269-276,303   (add; search(q, 100); search(q, topk_training))
 *   drivers/run_ann_data_gen_dpr.py:238-252
 * i.e. S = Q . X^T, per query the k largest scores sorted descending, D float32[nq,k],
 * I int64[nq,k] = row index into the added matrix, (-FLT_MAX, -1) padding when n < k.
 *
 * The arithmetic itself lives in faiss-cpu (unpinned, setup.py:22; not in /root/reference),
 * whose summation order is BLAS-defined.  PARITY UNPINNED at that boundary: the reference holds
 * no golden vectors for it.  This oracle fixes the two things FAISS leaves open so that results
 * are bit-reproducible:
 *   score  = fp32 fmaf chain over k = 0..d-1 ascending, starting from +0.0f
 *            (bit-for-bit what v_mfma_f32_32x32x2_f32 accumulates on gfx950);
 *   order  = total order (score descending, row id ascending).
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off; no -ffast-math).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLK 16

/* scores[q][j] for j in [0,n): fmaf chain, k ascending.  Rows are processed BLK at a time
 * through a transposed scratch so the compiler can vectorise ACROSS rows (independent chains);
 * the chain of each (q, row) pair stays strictly sequential in k. */
static void score_block(const float *x, int64_t ld, int d, const float *q, int nrows, float *xt, float *out)
{
    for (int r = 0; r < BLK; ++r)
        for (int k = 0; k < d; ++k)
            xt[(size_t)k * BLK + r] = (r < nrows) ? x[(size_t)r * ld + k] : 0.0f;
    float s[BLK];
    for (int r = 0; r < BLK; ++r) s[r] = 0.0f;
    for (int k = 0; k < d; ++k) {
        const float qk = q[k];
        const float *col = xt + (size_t)k * BLK;
        for (int r = 0; r < BLK; ++r) s[r] = __builtin_fmaf(qk, col[r], s[r]);
    }
    for (int r = 0; r < nrows; ++r) out[r] = s[r];
}

void ance_oracle_ip_scores(const float *x, int64_t n, const float *q, int64_t nq, int d, float *out)
{
#pragma omp parallel
    {
        float *xt = (float *)malloc(sizeof(float) * (size_t)d * BLK);
#pragma omp for schedule(static)
        for (int64_t j0 = 0; j0 < n; j0 += BLK) {
            int nrows = (int)((n - j0) < BLK ? (n - j0) : BLK);
            /* transpose once, reuse for all queries */
            for (int r = 0; r < BLK; ++r)
                for (int k = 0; k < d; ++k)
                    xt[(size_t)k * BLK + r] = (r < nrows) ? x[(size_t)(j0 + r) * d + k] : 0.0f;
            for (int64_t qi = 0; qi < nq; ++qi) {
                const float *qq = q + (size_t)qi * d;
                float s[BLK];
                for (int r = 0; r < BLK; ++r) s[r] = 0.0f;
                for (int k = 0; k < d; ++k) {
                    const float qk = qq[k];
                    const float *col = xt + (size_t)k * BLK;
                    for (int r = 0; r < BLK; ++r) s[r] = __builtin_fmaf(qk, col[r], s[r]);
                }
                for (int r = 0; r < nrows; ++r) out[(size_t)qi * n + j0 + r] = s[r];
            }
        }
        free(xt);
    }
    (void)score_block;
}

/* canonical total order: a "beats" b  <=>  a ranks before b in the result list */
static inline int beats(float sa, int64_t ia, float sb, int64_t ib)
{
    return (sa > sb) || (sa == sb && ia < ib);
}

typedef struct { float s; int64_t i; } ent_t;

/* min-heap on the canonical order: root = the WORST kept entry */
static void sift_down(ent_t *h, int n, int p)
{
    for (;;) {
        int l = 2 * p + 1, r = l + 1, w = p;
        if (l < n && beats(h[w].s, h[w].i, h[l].s, h[l].i)) w = l;
        if (r < n && beats(h[w].s, h[w].i, h[r].s, h[r].i)) w = r;
        if (w == p) return;
        ent_t t = h[p]; h[p] = h[w]; h[w] = t; p = w;
    }
}

static int cmp_rank(const void *a, const void *b)
{
    const ent_t *x = (const ent_t *)a, *y = (const ent_t *)b;
    if (beats(x->s, x->i, y->s, y->i)) return -1;
    if (beats(y->s, y->i, x->s, x->i)) return 1;
    return 0;
}

/* top-k of one score row under the canonical order; ids are row_base + position. */
void ance_oracle_topk_row(const float *scores, int64_t n, int64_t row_base, int k, float *D, int64_t *I)
{
    ent_t *h = (ent_t *)malloc(sizeof(ent_t) * (size_t)(k > 0 ? k : 1));
    int cnt = 0;
    for (int64_t j = 0; j < n; ++j) {
        float s = scores[j];
        if (s != s) continue; /* NaN never enters */
        int64_t id = row_base + j;
        if (cnt < k) {
            h[cnt].s = s; h[cnt].i = id; ++cnt;
            if (cnt == k) for (int p = k / 2 - 1; p >= 0; --p) sift_down(h, k, p);
        } else if (beats(s, id, h[0].s, h[0].i)) {
            h[0].s = s; h[0].i = id; sift_down(h, k, 0);
        }
    }
    qsort(h, (size_t)cnt, sizeof(ent_t), cmp_rank);
    for (int r = 0; r < k; ++r) {
        if (r < cnt) { D[r] = h[r].s; I[r] = h[r].i; }
        else { D[r] = -FLT_MAX; I[r] = -1; }
    }
    free(h);
}

/* Full search: scores by fmaf chain, canonical top-k.  Row-block streaming keeps memory O(nq*k). */
void ance_oracle_ip_topk(const float *x, int64_t n, int64_t row_base, const float *q, int64_t nq, int d, int k,
                         float *D, int64_t *I)
{
#pragma omp parallel
    {
        float *xt = (float *)malloc(sizeof(float) * (size_t)d * BLK);
        float *row = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            const float *qq = q + (size_t)qi * d;
            for (int64_t j0 = 0; j0 < n; j0 += BLK) {
                int nrows = (int)((n - j0) < BLK ? (n - j0) : BLK);
                score_block(x + (size_t)j0 * d, d, d, qq, nrows, xt, row + j0);
            }
            ance_oracle_topk_row(row, n, row_base, k, D + (size_t)qi * k, I + (size_t)qi * k);
        }
        free(xt); free(row);
    }
}

/* Merge n_parts per-shard lists [n_parts][nq][k] (each already canonical, padded with I=-1)
 * into one canonical list -- restates utils/eval_mrr.py:173-183 (all_gather (D,I), concat on
 * axis 1, argsort) with the canonical tie-break instead of argsort's unspecified one. */
void ance_oracle_topk_merge(const float *Dp, const int64_t *Ip, int n_parts, int64_t nq, int k, float *D, int64_t *I)
{
    ent_t *buf = (ent_t *)malloc(sizeof(ent_t) * (size_t)n_parts * (size_t)k);
    for (int64_t qi = 0; qi < nq; ++qi) {
        int cnt = 0;
        for (int p = 0; p < n_parts; ++p)
            for (int r = 0; r < k; ++r) {
                size_t o = ((size_t)p * nq + qi) * k + r;
                if (Ip[o] < 0) continue;
                buf[cnt].s = Dp[o]; buf[cnt].i = Ip[o]; ++cnt;
            }
        qsort(buf, (size_t)cnt, sizeof(ent_t), cmp_rank);
        for (int r = 0; r < k; ++r) {
            if (r < cnt) { D[qi * k + r] = buf[r].s; I[qi * k + r] = buf[r].i; }
            else { D[qi * k + r] = -FLT_MAX; I[qi * k + r] = -1; }
        }
    }
    free(buf);
}

/* ---- the algorithm of faiss-cpu's IndexFlatIP.search, for the timed CPU baseline (bench.py cpu_baseline, kind "port") ----
 * faiss computes blocks of the score matrix with sgemm (4,096 queries x 1,024 database rows, faiss/utils/distances.cpp:
 * distance_compute_blas_query_bs / _database_bs) and feeds every block row into a per-query k-heap (HeapResultHandler),
 * queries in parallel under OpenMP.  The sgemm is NumPy's BLAS (oracle/search_ref.py); this is the heap side: one size-k
 * min-heap per query under the canonical order, updated from a score block S[nq][nb] whose column j is row row_base + j.
 * heap_s / heap_i: [nq][k], heap_cnt: [nq] (0 before the first block). */
void ance_oracle_heap_update(const float *S, int64_t nq, int64_t nb, int64_t ld, int64_t row_base, int k, float *heap_s,
                             int64_t *heap_i, int32_t *heap_cnt)
{
#pragma omp parallel for schedule(static)
    for (int64_t qi = 0; qi < nq; ++qi) {
        const float *row = S + (size_t)qi * ld;
        float *hs = heap_s + (size_t)qi * k;
        int64_t *hi = heap_i + (size_t)qi * k;
        int cnt = heap_cnt[qi];
        for (int64_t j = 0; j < nb; ++j) {
            const float s = row[j];
            const int64_t id = row_base + j;
            if (cnt < k) {
                if (s != s) continue;
                hs[cnt] = s; hi[cnt] = id; ++cnt;
                if (cnt == k) {  /* heapify: root = the worst kept entry */
                    for (int p0 = k / 2 - 1; p0 >= 0; --p0) {
                        int p = p0;
                        for (;;) {
                            int l = 2 * p + 1, r = l + 1, w = p;
                            if (l < k && beats(hs[w], hi[w], hs[l], hi[l])) w = l;
                            if (r < k && beats(hs[w], hi[w], hs[r], hi[r])) w = r;
                            if (w == p) break;
                            float ts = hs[p]; hs[p] = hs[w]; hs[w] = ts;
                            int64_t ti = hi[p]; hi[p] = hi[w]; hi[w] = ti;
                            p = w;
                        }
                    }
                }
            } else if (s > hs[0] || (s == hs[0] && id < hi[0])) {
                hs[0] = s; hi[0] = id;
                int p = 0;
                for (;;) {
                    int l = 2 * p + 1, r = l + 1, w = p;
                    if (l < k && beats(hs[w], hi[w], hs[l], hi[l])) w = l;
                    if (r < k && beats(hs[w], hi[w], hs[r], hi[r])) w = r;
                    if (w == p) break;
                    float ts = hs[p]; hs[p] = hs[w]; hs[w] = ts;
                    int64_t ti = hi[p]; hi[p] = hi[w]; hi[w] = ti;
                    p = w;
                }
            }
        }
        heap_cnt[qi] = cnt;
    }
}

/* heaps -> sorted lists (score desc, id asc), padded with (-FLT_MAX, -1) */
void ance_oracle_heap_finish(int64_t nq, int k, const float *heap_s, const int64_t *heap_i, const int32_t *heap_cnt, float *D,
                             int64_t *I)
{
#pragma omp parallel for schedule(static)
    for (int64_t qi = 0; qi < nq; ++qi) {
        ent_t *buf = (ent_t *)malloc(sizeof(ent_t) * (size_t)(k > 0 ? k : 1));
        const int cnt = heap_cnt[qi];
        for (int r = 0; r < cnt; ++r) { buf[r].s = heap_s[(size_t)qi * k + r]; buf[r].i = heap_i[(size_t)qi * k + r]; }
        qsort(buf, (size_t)cnt, sizeof(ent_t), cmp_rank);
        for (int r = 0; r < k; ++r) {
            if (r < cnt) { D[(size_t)qi * k + r] = buf[r].s; I[(size_t)qi * k + r] = buf[r].i; }
            else { D[(size_t)qi * k + r] = -FLT_MAX; I[(size_t)qi * k + r] = -1; }
        }
        free(buf);
    }
}
