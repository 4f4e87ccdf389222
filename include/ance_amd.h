/*
 * ance_amd -- C ABI of the MI355X-native ANN hard-negative refresh path.
 *
 * The reference (microsoft/ANCE) has no FFI: its hot path calls two Python packages.  These
 * entry points are what a binding for that path replaces, one for one:
 *
 *   ance_encode_*      <- model.module.query_emb / body_emb
 *                         (drivers/run_ann_data_gen.py:171-180; model/models.py:149-157,165-199,
 *                          235-259) i.e. transformers' RobertaModel / BertModel forward + head
 *   ance_ip_topk       <- faiss.IndexFlatIP(dim).add(x); .search(q, k)
 *                         (drivers/run_ann_data_gen.py:269-276,303; run_ann_data_gen_dpr.py:238-252)
 *   ance_topk_merge    <- the shard-search-then-merge of utils/eval_mrr.py:173-183
 *                         (all_gather of (D, I), concat, argsort) under the canonical order
 *
 * Conventions: plain pointers and sizes, no exceptions, no torch types.  Every pointer named
 * d_* is DEVICE memory owned by the caller; nothing is allocated or freed behind the caller's
 * back except the small host-side handle of ance_encoder_create.  All work is enqueued on the
 * caller's hipStream_t (passed as void* so this header needs no HIP include) and is asynchronous;
 * the functions never synchronise the device.  Return value: 0 on success, negative ANCE_E_* on
 * error (nothing enqueued in that case).
 *
 * Canonical result order of every top-k list: score descending, then row id ascending.
 * Scores are exact fp32: an fmaf chain over k = 0..d-1 ascending starting from +0.0f.
 */
#ifndef ANCE_AMD_H
#define ANCE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANCE_OK 0
#define ANCE_E_INVALID (-1)   /* bad argument (null pointer, k/d/L out of the supported range) */
#define ANCE_E_WORKSPACE (-2) /* workspace too small */
#define ANCE_E_LAUNCH (-3)    /* HIP reported a launch error */
#define ANCE_E_NOMEM (-4)

#define ANCE_ABI_VERSION 5  /* 5: AnceEncoderDesc.precision (the arithmetic is an argument, not an environment variable), ance_encoder_range_faults,
                               ance_ip_topk_scan; 4: blocked pair rows in the split mode (ance_pair_layout; ance_debug_gemm_split + d_wscale_inv);
                               3: + ance_nll_forward, ance_search_bad_image_calls, ance_debug_gemm_split; split encoder mode */
int ance_abi_version(void);
/* last HIP error string seen by this library on the calling thread ("" if none) */
const char *ance_last_error(void);

/*
 * Measurement hook (bench.py's roofline): when enabled, every kernel launch of this library is
 * bracketed by HIP events on the launch stream.  ance_profile_read synchronises those events and
 * returns, per category, the summed kernel time (ms), the summed algorithmic work (FLOP) and the
 * launch count, then keeps accumulating.  Categories, in order:
 *   0 plan/pack  1 embed+LN  2 gemm Q|K  3 gemm V^T  4 attention  5 gemm attn-out  6 LayerNorm
 *   7 gemm FFN1+GELU  8 gemm FFN2  9 head  10 ip_topk scan  11 top-k finalize/merge
 *   12 exact re-scoring of the two-precision search
 * Returns the number of categories.  Not thread safe; off by default (no events are created).
 */
#define ANCE_PROFILE_CATEGORIES 13
void ance_profile_enable(int on);
int ance_profile_read(double *ms, double *work, long long *count, int n);

/* ------------------------------------------------------------------ exact IP top-k search -- */

#define ANCE_TOPK_MAX_K 1792

/* Bytes of scratch ance_ip_topk needs for (n rows, nq queries, dimension d, k).  0 if unsupported. */
size_t ance_ip_topk_workspace_bytes(int64_t n, int64_t nq, int d, int k);

/*
 * Exact inner-product top-k of nq queries against one shard of n rows.
 *   d_x   float32 [n, d] row-major, 16-byte aligned, d % 4 == 0
 *   d_q   float32 [nq, d]
 *   row_base  global id of the shard's first row (ids returned are row_base + local row)
 *   d_out_d   float32 [nq, k] scores, canonical order; -FLT_MAX where fewer than k rows exist
 *   d_out_i   int64   [nq, k] global row ids; -1 where fewer than k rows exist
 * Requires n < 2^32, 1 <= k <= ANCE_TOPK_MAX_K.
 *
 * Two kernels produce the same bits: an fp32-MFMA scan (any shape), and -- when d % 128 == 0,
 * 128 <= d <= 2048, k <= 1024 and n >= 4096 -- a two-precision path: fp16 MFMA scores filter the
 * corpus under a rigorous error slack, every survivor is re-scored with the exact fp32 fmaf chain.
 * A query whose candidates cannot be bounded that way (row or query norms above 65504 or NaN, i.e.
 * possible fp16 overflow; thousands of rows inside one error band) is detected on the device and
 * redone by the scan, alone; heavy classes of bit-identical rows (all-pad MaxP chunks) are scored
 * once and expanded in id order.  Environment (tuning / A-B only; read ONCE, the first time the library needs a
 * knob -- ance_reload_env() re-reads all of them):
 *   ANCE_SEARCH=exact            force the scan
 *   ANCE_FAST_SPLITS=<2^j>       corpus splits per query tile (default 2)
 *   ANCE_FAST_WINDOW_TILES=<n>   corpus window all workgroups finish together, in 256-row tiles
 *                                (default 256 = 100 MB of fp16 rows at d = 768; 0 = no windows)
 *   ANCE_FAST_WINDOW_WAIT_US     bound of the wait at a window boundary (default 200, 0 = none)
 *   ANCE_FAST_SHARE=0            do not exchange thresholds between the splits of a query
 *   ANCE_FAST_DEDUP=0            do not collapse duplicate classes when an image is built
 *   ANCE_FAST_CENTER=0           do not centre the fp16 image on the shard's mean row (the filter's error
 *                                slack then scales with |x| instead of |x - mean|)
 */
int ance_ip_topk(const float *d_x, int64_t n, int64_t row_base, const float *d_q, int64_t nq, int d, int k,
                 float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * The same search by the fp32-MFMA scan ALONE (csrc/ip_topk.hip: ip_topk_scan_kernel), whatever the shape and the environment:
 * the independent audit path of the two-precision kernel -- both must return the same bits (bench.py checks the headline-size
 * search against it on a sample of its queries; tests/test_gpu_search.py pins the scan to oracle/ip_topk_ref.c).  ~7 x slower.
 */
size_t ance_ip_topk_scan_workspace_bytes(int64_t n, int64_t nq, int d, int k);
int ance_ip_topk_scan(const float *d_x, int64_t n, int64_t row_base, const float *d_q, int64_t nq, int d, int k,
                      float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * Search image of a shard: what faiss.IndexFlatIP.add builds once and every .search reuses
 * (drivers/run_ann_data_gen.py:269-276 adds once, :276 and :303 search twice).  ance_ip_topk rebuilds
 * it inside its workspace on every call; a caller that searches the same rows repeatedly builds it once:
 *   bytes = ance_ip_index_bytes(n, d)        0: the shape has no image (the scan is used), pass NULL
 *   ance_ip_index_build(d_x, n, d, d_index, bytes, stream)
 *   ance_ip_topk_indexed(d_x, n, row_base, d_index, ...)   with ance_ip_topk_indexed_workspace_bytes
 * The image is stamped with the (n, d, d_x) it was built from when the build completes; a search whose arguments do not
 * match the stamp -- another shard's image, a buffer never built -- ignores the image on the device (no host
 * synchronisation) and answers every query with the exact scan: slower, never wrong rows.
 * The image is valid for exactly the (d_x, n, d) it was built from; d_x must stay alive and unchanged
 * (the exact re-scoring reads the fp32 rows).  Contents: the shard's mean row, fp16(row - mean) for every
 * row kept (n d 2 bytes), row map, duplicate classes.  Results are those of ance_ip_topk, bit for bit.
 */
size_t ance_ip_index_bytes(int64_t n, int d);
int ance_ip_index_build(const float *d_x, int64_t n, int d, void *d_index, size_t index_bytes, void *stream);
size_t ance_ip_topk_indexed_workspace_bytes(int64_t n, int64_t nq, int d, int k);
int ance_ip_topk_indexed(const float *d_x, int64_t n, int64_t row_base, const void *d_index, const float *d_q,
                         int64_t nq, int d, int k, float *d_out_d, int64_t *d_out_i, void *d_workspace,
                         size_t workspace_bytes, void *stream);

/* Bytes of scratch ance_topk_merge needs. */
size_t ance_topk_merge_workspace_bytes(int n_parts, int64_t nq, int k);

/*
 * Merge n_parts canonical lists (e.g. one per corpus shard / GPU) into one.
 *   d_parts_d float32 [n_parts, nq, k], d_parts_i int64 [n_parts, nq, k] (ids < 2^32, -1 = empty)
 */
int ance_topk_merge(const float *d_parts_d, const int64_t *d_parts_i, int n_parts, int64_t nq, int k,
                    float *d_out_d, int64_t *d_out_i, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * Restricted-candidate scoring (the rerank of evaluation/"Calculate Metrics.ipynb" cell 11 and
 * utils/eval_mrr.py:94-105, where a per-query faiss sub-index is built from the BM25 candidates):
 * d_scores[j] = <d_q[qi], d_x[d_rows[j]]> for d_offsets[qi] <= j < d_offsets[qi+1], with the same
 * fp32 fmaf chain (k ascending from +0) as ance_ip_topk, so scores are bitwise those of the full scan.
 * d_rows: int64 row ids in [0, n) (an id outside the range scores -inf); d_offsets: int64 [nq + 1]
 * ascending, d_offsets[0] = 0.  Enqueues on `stream` and returns.
 */
int ance_ip_score_rows(const float *d_x, int64_t n, const float *d_q, int64_t nq, int d, const int64_t *d_rows,
                       const int64_t *d_offsets, float *d_scores, void *stream);

/* ------------------------------------------------------------------------ dual encoder ------ */

#define ANCE_ARCH_ROBERTA 0 /* positions = cumsum(id != pad) * (id != pad) + pad, type 0     */
#define ANCE_ARCH_BERT 1    /* positions = 0..len-1, token type 0                              */

typedef struct AnceEncoderDesc {
    int32_t arch;         /* ANCE_ARCH_*                                                     */
    int32_t n_layers;     /* 12                                                              */
    int32_t hidden;       /* 768 (must be 768 in this build: 12 heads x 64)                  */
    int32_t n_heads;      /* 12                                                              */
    int32_t intermediate; /* 3072                                                            */
    int32_t vocab_size;
    int32_t max_position; /* rows of the position table (514 RoBERTa / 512 BERT)             */
    int32_t pad_token_id; /* 1 RoBERTa / 0 BERT                                              */
    float ln_eps;         /* 1e-5 RoBERTa / 1e-12 BERT (encoder LayerNorms)                  */
    int32_t has_head;     /* 1: LayerNorm_768(W h_cls + b), eps 1e-5 (models.py:145-153); 0: raw h_cls */
    int32_t max_seq_len;  /* longest single sequence (chunk) this handle will see, <= 512    */
    int32_t max_tokens;   /* token capacity of one micro-batch (workspace sizing)            */
    int32_t precision;    /* ANCE_PRECISION_*: the arithmetic of the handle (and of the two size queries)  */
} AnceEncoderDesc;

#define ANCE_PRECISION_DEFAULT 0 /* what the environment says (ANCE_ENCODER_* below); nothing set: split */
#define ANCE_PRECISION_SPLIT 1   /* fp16 pair operands, three MFMAs per k-step: fp32-grade (2e-5), the default */
#define ANCE_PRECISION_FP16 2    /* fp16 operands: the fast mode (5e-3) */
#define ANCE_PRECISION_FP32 3    /* fp32 operands on the fp32-input matrix cores: the audit path */

typedef struct AnceEncoder AnceEncoder;

/*
 * Order of the fp32 device weight pointers handed to ance_encoder_create (HF state-dict names,
 * prefix = "roberta." / "question_model." / "ctx_model."):
 *   [0] embeddings.word_embeddings.weight        [vocab, H]
 *   [1] embeddings.position_embeddings.weight    [max_position, H]
 *   [2] embeddings.token_type_embeddings.weight  [>=1, H]   (row 0 used)
 *   [3] embeddings.LayerNorm.weight  [4] embeddings.LayerNorm.bias
 *   then per layer i (16 pointers, base 5 + 16 i):
 *     +0 attention.self.query.weight  +1 .bias     +2 attention.self.key.weight   +3 .bias
 *     +4 attention.self.value.weight  +5 .bias     +6 attention.output.dense.weight +7 .bias
 *     +8 attention.output.LayerNorm.weight +9 .bias
 *     +10 intermediate.dense.weight   +11 .bias    +12 output.dense.weight +13 .bias
 *     +14 output.LayerNorm.weight     +15 .bias
 *   then, if has_head: embeddingHead.weight, embeddingHead.bias, norm.weight, norm.bias
 */
#define ANCE_ENCODER_N_WEIGHTS(n_layers, has_head) (5 + 16 * (n_layers) + ((has_head) ? 4 : 0))

/* Bytes of the packed weight arena (fp16 GEMM operands + fp32 vectors) and of the activation
 * workspace for desc->max_tokens.  Both are caller-allocated device buffers, 256-byte aligned.
 *
 * Arithmetic.  DEFAULT (nothing in the environment): the SPLIT mode -- an fp32-GRADE result from the fp16 matrix cores: every GEMM
 * operand an fp16 pair v = hi + lo, three MFMAs per k-step (hi hi + lo hi + hi lo) from four operand tiles staged once, fp32
 * accumulation, fp32 softmax, erf-GELU to fp32 grade, fp32 head; max |delta| 2e-5 (stated; 7e-6 measured at 12 layers) against the
 * reference's fp32 arithmetic (model/models.py:149-157 runs in fp32).  Preconditions of the split mode, both CHECKED on the device
 * (ance_encoder_range_faults): pre-LayerNorm values, Q | K | V and GELU outputs must stay below 65,504 (the hi half of a pair is an fp16)
 * -- the reference's fp32 has no such limit, --encoder_precision fp32 / ANCE_PRECISION_FP32 is the way out; and no NaN in the output
 * rows.  Activation magnitudes: the lo half of an element below 2^-3 is an fp16 subnormal, i.e. good to 2^-25 ABSOLUTE -- fp32-grade for
 * the O(1) values of the post-residual streams; the embedding sum (0.05 in trained BERT / RoBERTa checkpoints, followed by a LayerNorm
 * with rstd ~ 20) is therefore stored times 16 and its LayerNorm epsilon times 256 (exact: powers of two).  Weights may have any scale
 * (stored times a per-matrix power of two).
 * The arithmetic of a handle is AnceEncoderDesc.precision.  ANCE_PRECISION_DEFAULT (0) defers to the environment, read when the handle
 * is created (the two size queries resolve the mode the same way) -- a drivers' --encoder_precision flag overrides it:
 *   ANCE_ENCODER_FP16=1      the fp16 FAST mode (ANCE_ENCODER_SPLIT=0 is another spelling): fp16 MFMA operands, fp32 accumulation, fp32
 *                            softmax / statistics / head; LayerNorm folded into the GEMMs and the residual stream kept as fp16 (hi, lo)
 *                            pairs (22 mantissa bits) -- max |delta| 3e-3 on unit-variance embeddings (stated tolerance of the tests:
 *                            5e-3), 2.2 x the default's throughput.  A folded GEMM takes fp16(v) of the PRE-LayerNorm row v as its token
 *                            operand, so its rounding error scales with |v|, not |v - mean|: a GEMM tile with a token whose |mean| rstd
 *                            exceeds 2 runs a second K loop over the lo halves OF THOSE TOKENS (masked per row: a row's bits never
 *                            depend on its tile mates); below that threshold the error is at most sqrt(1 + 2^2) x the random-init figure
 *   ANCE_ENCODER_SPLIT=1     names the default explicitly; wins over ANCE_ENCODER_FP16
 *   ANCE_ENCODER_PRECISE=1   fp32 mode: fp32 operands on the fp32-input matrix cores, exact erf GELU, fp32 softmax -- the
 *                            reference's arithmetic (model/models.py:149-157); max |delta| 1e-5, 4.4 x slower than the default (the audit
 *                            path); wins over the other two switches
 *   ANCE_GEMM_NSPLIT=0       FFN1 without the N-split tile order (A/B switch)
 *   ANCE_ENCODER_STREAMS=n   internal streams / activation sets (1 or 2, default 2)
 *   ANCE_LN_FOLD=0 ANCE_HEAD_MFMA=0 ANCE_CLS_TAIL=0 ANCE_ATTN_COAL=0 ANCE_GEMM_DESC=0   A/B switches back to the previous
 *                            form of one piece each (LayerNorm kernels, per-sequence head, full last layer, per-lane
 *                            attention loads / stores, flat-pointer GEMM staging; ANCE_GEMM_DESC is read at the first GEMM
 *                            launch of the process, the others per handle) */
size_t ance_encoder_weight_bytes(const AnceEncoderDesc *desc);
size_t ance_encoder_workspace_bytes(const AnceEncoderDesc *desc);

/* Packs the fp32 weights into d_weight_arena (enqueued on stream; the fp32 sources may be freed
 * once the stream has passed this point) and returns a handle bound to the two buffers. */
int ance_encoder_create(const AnceEncoderDesc *desc, const void *const *d_weights_fp32, int n_weights,
                        void *d_weight_arena, size_t weight_bytes, void *d_workspace, size_t workspace_bytes,
                        void *stream, AnceEncoder **out);
void ance_encoder_destroy(AnceEncoder *enc);
/* The arithmetic the handle runs: ANCE_PRECISION_SPLIT / FP16 / FP32 (never DEFAULT). */
int ance_encoder_precision(const AnceEncoder *enc);

/*
 * Range guard.  The handle keeps two sticky device counters, updated by the kernels of every ance_encode_* call:
 *   [0] threads of the split mode's pair-forming stages (embedding sum, Q | K | V, GELU output, residual stream) that saw a
 *       value above 65,504 in magnitude or a non-finite one -- the split mode's precondition is violated, the embeddings of
 *       that call are NOT fp32-grade (the fp16 hi half overflowed);
 *   [1] output rows (any mode) whose statistics are NaN or infinite.
 * Enqueues on `stream` a copy of both to h_out (HOST pointer, uint32[2]; pinned memory keeps the copy asynchronous) and, if
 * reset != 0, zeroes them behind it.  Never synchronises: h_out is valid once `stream` has passed this point.
 */
int ance_encoder_range_faults(AnceEncoder *enc, uint32_t *h_out, int reset, void *stream);

/*
 * Encode n records.  Each record is L int32 token ids split into n_chunks chunks of L / n_chunks
 * tokens (n_chunks = 1: FirstP / queries; 4 with L = 2048: MaxP).  Pad tokens cost nothing: a
 * chunk contributes only its first len_c = clamp(len - c * L/n_chunks, 0, L/n_chunks) tokens
 * (an all-pad chunk is encoded as the single pad token it is equivalent to).
 *   d_out float32 [n * n_chunks, 768], row = record * n_chunks + chunk.
 *
 * ance_encode_records: d_records = raw rows of the reference's tokenised cache
 *   (utils/util.py:279-283): 4-byte BIG-endian length then L little-endian int32, record_bytes =
 *   4 + 4 L, so the cache file can be copied to HBM verbatim.
 * ance_encode_ids: d_ids int32 [n, L] (row stride ld_ids int32 elements), d_lens int32 [n].
 *
 * h_lens (HOST pointer, int32 [n], may be NULL): the same lengths the device will read, used by
 *   the host-side micro-batch planner.  With h_lens the call never synchronises; with NULL the
 *   library reads the lengths back once per 262,144 records (a stream synchronisation).
 *   Precondition: h_lens[i] equals the record's header / d_lens[i].
 */
int ance_encode_records(AnceEncoder *enc, const void *d_records, const int32_t *h_lens, int64_t n, int L,
                        int n_chunks, float *d_out, void *stream);
int ance_encode_ids(AnceEncoder *enc, const int32_t *d_ids, int64_t ld_ids, const int32_t *d_lens,
                    const int32_t *h_lens, int64_t n, int L, int n_chunks, float *d_out, void *stream);

/*
 * Test / measurement hook: C = A . B^T (+ epilogue) with the encoder's GEMM kernel on caller data.
 *   epi 0: out f16 = acc + bias[n]; 1: out f16 = gelu(acc + bias[n]); 2: out f32 = acc + bias[n] + res32
 *   d_a_f16 [M,K], d_b_f16 [N,K] fp16 row-major; M, N multiples of 256, K a multiple of 64, >= 128.
 *   ablate 0: the product kernel (ping-pong main loop).  Non-zero values are accepted by the MEASUREMENT library only
 *   (`make -C ance_amd/csrc measure`; the product library returns ANCE_E_INVALID) and select the two-phase loop it
 *   replaced, with measurement ablations by bit (results are then WRONG on purpose): 1 = no global
 *   loads after the first K-tile, 2 = no MFMA, 4 = every block loads tile (0,0); 8 = no ablation
 *   (correct results; the A/B reference for the main loop).  16 + bits: the ping-pong loop with
 *   ablations 1 = every K-tile re-reads tiles 0/1, 2 = no MFMA, 4 = tile (0,0), 8 = no staging at all.
 *   32 (epi 0 / 1): timeline -- correct results, and d_res32 receives uint64[M/256 * N/256][5] stamps of the
 *   100 MHz real-time counter per workgroup: start, prologue done, main loop done, epilogue issued, stores drained.
 */
int ance_debug_gemm(int ablate, int epi, const void *d_a_f16, const void *d_b_f16, int M, int N, int K,
                    const float *d_bias, void *d_out, const float *d_res32, void *stream);

/* Diagnostic: the number of ance_ip_topk_indexed launch chunks on the current device whose search image did not carry the stamp
 * of the matrix searched (moved / copied rows, a view at another address, a buffer that was never built).  Such calls are
 * answered by the exact scan -- same results, several times slower -- and nothing else reports it.  Synchronises the device. */
int ance_search_bad_image_calls(unsigned long long *out);

/* Forward of the training objective on embeddings the encoder produced -- the consumer side of the refresh's file contract
 * (forward only): replaces the tail of NLL.forward (model/models.py:71-81) and NLL_MultiChunk.forward
 * (:97-134) after the three query_emb / body_emb calls.
 *   d_q [n, d], d_a / d_b [n * chunks, d] fp32 (row = triplet * chunks + chunk); chunks = 1 for FirstP;
 *   d_mask_a / d_mask_b [n, chunks] fp32 = the attention mask's first entry of every chunk (MaxP: an all-pad chunk is biased by
 *   -9999 before the max over chunks, :109-113); may be NULL when chunks == 1;
 *   d_logits [n, 2] = (logit_a, logit_b); d_loss_rows [n] = -log_softmax(logits)[:, 0]; d_loss_mean [1] = their mean (fixed
 *   summation order: the same input gives the same bits).  d a multiple of 4. */
int ance_nll_forward(const float *d_q, const float *d_a, const float *d_b, const float *d_mask_a, const float *d_mask_b, int64_t n,
                     int d, int chunks, float *d_logits, float *d_loss_rows, float *d_loss_mean, void *stream);

/* Test hook: the SPLIT (fp32-grade) GEMM of the encoder with one of its epilogues.  acc[m][n] = sum_k a[m][k] b[n][k] with
 * a = a_hi + a_lo (b likewise; the lo x lo products are left out); d_a_pair [M, 2K] / d_b_pair [N, 2K] fp16 PAIR ROWS -- 32-column
 * blocks [hi (32) | lo (32)], lo = fp16(v - hi) unscaled: ance_pair_layout gives the positions; (mu_m, r_m) = mean and
 * 1 / sqrt(var + ln_eps) of row m combined from d_part [M][12][2], the (mean, M2) of its twelve 64-column slices;
 * w = *d_wscale_inv (device scalar; NULL: 1), the inverse of the power of two b was stored with.
 *   epi 8   d_out fp32 [M, N]      = r_m (w acc - mu_m vec1[n]) + bias[n]                             (vec1 = csum)
 *   epi 9   d_out fp16 pair [M, 2N] = pair(gelu_erf(r_m (w acc - mu_m vec1[n]) + bias[n]))
 *   epi 10  d_out fp16 pair [M, 2N] = pair(w acc + bias[n] + (res[m][n] - mu_m) r_m vec1[n] + vec2[n])   (vec1 = gamma, vec2 = beta,
 *           res = d_res_pair [M, 2N] pair rows; N = 768), d_part_out [M][12][2] = slice statistics of the output rows
 * M, N multiples of 256, K of 64, >= 128. */
int ance_debug_gemm_split(int epi, const void *d_a_pair, const void *d_b_pair, int M, int N, int K, const float *d_bias,
                          const float *d_vec1, const float *d_vec2, const float *d_part, float ln_eps, const void *d_res_pair,
                          void *d_out, float *d_part_out, const float *d_wscale_inv, void *stream);

/* Layout of the split mode's pair rows (for tests and tools that build or read them): column n of a W-wide fp32 row has its hi
 * half at *hi_col and its lo half at *lo_col of the 2 W-half pair row, lo = fp16((v - hi) * *lo_scale).  Product library:
 * hi_col = 64 (n / 32) + n % 32, lo_col = hi_col + 32, lo_scale = 1. */
void ance_pair_layout(int n, int W, int *hi_col, int *lo_col, float *lo_scale);

/* Re-reads every ANCE_* tuning knob from the environment (they are otherwise read once per process).  For tests and
 * sweeps that change a knob between two calls; not thread-safe against concurrent searches. */
void ance_reload_env(void);

/*
 * Measurement hook of the two-precision search -- a no-op in the product library; the instrumented kernel builds exist
 * only in the measurement library (`make -C ance_amd/csrc measure` -> libance_amd_measure.so, -DANCE_MEASURE; load it
 * with ANCE_AMD_LIB=<path>).  There: while d_stamps != NULL, the filter kernel runs as its instrumented
 * build and every workgroup of a launch chunk leaves uint64[8] at d_stamps + 8 * blockIdx: ticks of the 100 MHz
 * counter spent in {prologue, fp16 main loop, filter, prune, window waits, hand-over to the re-scoring kernel},
 * then (query tile << 32 | split) and the XCC id it ran on.  The buffer needs 8 * 8 * 2048 bytes.  NULL switches it off.
 */
void ance_debug_search_stamps(void *d_stamps);

/* Introspection for tests / bench: algorithmic FLOPs of the last ance_encode_* call cannot be
 * known without a sync, so the library exposes the pure function instead:
 * F_enc(T) = 169,869,312 T + 36,864 T^2 + 1,179,648 per sequence of T tokens. */
double ance_encoder_flops_per_sequence(int T);

/* ---------------------------------------------------------------------------------------------
 * Host-side post-search stage.  Pure host code, HOST pointers, no GPU work:
 * replaces the per-element Python of GenerateNegativePassaageID
 * (drivers/run_ann_data_gen.py:339-396) and of the ann_training_data_N writer (:314-327).
 *
 * mt_state: uint32[625] = CPython `random.getstate()[1]` (624 Mersenne-Twister words + index).
 * The functions draw exactly what the reference's `random.shuffle` calls would draw and leave the
 * advanced state in place (install it with `random.setstate`), so a seeded run produces the
 * reference's files byte for byte.
 * ------------------------------------------------------------------------------------------- */

/* out[0..n) = list(range(n)) after random.shuffle (the line order of ann_training_data_N, :316-317). */
int ance_host_py_shuffle(uint32_t *mt_state, int64_t n, int64_t *out);

/*
 * Negative selection for every query row r with active[r] != 0 (query id in effective_q_id):
 * candidates = I[r, order] with order = random.shuffle(range(k)) (select_topk == 0; one shuffle per
 * active row, row order) or I[r, :negative_sample+1] (select_topk != 0, --ann_measure_topk_mrr);
 * walk them as the reference does: pid = p2id[row] (negative row ids index from the end like
 * NumPy), skip pid == pos_pid[r] (rank <= 10 adds 1/rank to *out_mrr), skip pids already taken,
 * stop at negative_sample.  out_neg [nq, negative_sample] (-1 padded), out_cnt [nq] (-1 for
 * inactive rows).  *out_mrr = the reference's `mrr` accumulator (before the division), summed in
 * its order.  n_threads <= 0: up to 16.  mt_state may be NULL when select_topk != 0.
 */
int ance_host_select_negatives(uint32_t *mt_state, const int64_t *I, int64_t nq, int k, const int64_t *p2id,
                               int64_t n_rows, const int64_t *pos_pid, const uint8_t *active, int negative_sample,
                               int select_topk, int n_threads, int64_t *out_neg, int32_t *out_cnt, double *out_mrr);

/*
 * Writes "qid \t pos_pid \t neg,neg,...\n" for rows order[0..n_order) whose src_row[row] >= 0, taking
 * the negatives of row src_row[row] (the reference keys them by query id, so a repeated id shows
 * the negatives of its last row).  *out_lines = lines written.
 */
int ance_host_write_ann_training(const char *path, const int64_t *order, int64_t n_order, const int64_t *qid,
                                 const int64_t *pos_pid, const int64_t *src_row, const int64_t *neg, const int32_t *cnt,
                                 int negative_sample, int64_t *out_lines);

#ifdef __cplusplus
}
#endif
#endif /* ANCE_AMD_H */
