// Dual-encoder forward for gfx950: RoBERTa-base / BERT-base stack + ANCE head, variable-length
// packed (pad tokens are never materialised).  Replaces, on the reference's hot path,
//   model.module.query_emb / body_emb  (drivers/run_ann_data_gen.py:171-180)
//   = transformers RobertaModel/BertModel forward + embeddingHead + norm (model/models.py:149-157,
//     165-199, 235-259).
//
// Per micro-batch (<= max_tokens real tokens), per layer:
//   QK   = h16 Wqk^T + b          (gemm EPI_QK, Q pre-scaled by 1/8)        [T, 1536] f16
//   V^T  = Wv h16^T + b           (gemm EPI_VT, key-contiguous)             [768, cols] f16
//   ctx  = softmax(Q K^T) V       (attention.hip)                           [T, 768] f16
//   preA = ctx Wo^T + b + LN(preB)  (gemm EPI_RES32)                        [T, 768] f32
//   LayerNorm(preA)               -> h16 (next MFMA operand) and per-row (mean, rstd)
//   f    = gelu(h16 W1^T + b)     (gemm EPI_GELU)                           [T, 3072] f16
//   preB = f W2^T + b + LN(preA) ; LayerNorm(preB) -> h16, (mean, rstd)
// then  emb = LayerNorm(Wh LN(preB)[cls] + bh)  (fp32)  or raw LN(preB)[cls] for DPR's BERT.
// The fp32 residual stream h = LN(pre) is never stored: its consumers (the next RES32 epilogue, the
// [CLS] gather, the head) recompute it from the pre-LN row and the two row statistics with the one
// expression ln_apply4 -- 6 KB per token and layer less HBM traffic (LayerNorm kernel 72 -> ~46 us).
// Precision: fp16 MFMA operands, fp32 accumulation, fp32 residual stream / LayerNorm / softmax.
//
// "LayerNorm without a kernel" (default; ANCE_LN_FOLD=0 selects the form above): the two LayerNorm passes of a layer
// read 3 KB and write 1.5 KB per token at the HBM roofline for arithmetic every consumer can do on the fly.  Instead
//   * the RES GEMM epilogue (EPI_RESLN) writes its output row v as an fp16 pair (hi = fp16(v), lo = fp16(v - hi): the same
//     3 KB the fp32 row took, 22 mantissa bits) and the (mean, M2) of every 64-column slice (96 bytes per row);
//   * the consumer GEMMs take hi AS IT IS for their token operand and finish the normalisation algebraically:
//       LN(v) W^T + b = r (v (gamma (.) W)^T - mu c) + (b + W beta),   c[n] = sum_k fp16(gamma_k W[n][k])
//     with gamma folded into the fp16 weight when it is loaded, c summed over the ROUNDED weights (so that the identity is
//     exact for the products the MFMA actually forms) and b' in fp32;
//   * every consumer combines the 12 slices of a row into (mean, rstd) itself (Chan): a GEMM tile gets the partials of
//     its 256 tokens, with its bias / csum / gamma / beta vectors, by LDS-DMA ahead of its main loop and combines them
//     when its epilogue starts -- there is no LayerNorm kernel and no statistics kernel at all;
//   * the consumers of the fp32 value (next RES epilogue, [CLS] gather, head) recompute LN(hi + lo) from the pair.
// Rounding points that move: the token operand is fp16(v) instead of fp16(LN(v)) -- the same relative rounding of every
// element, taken before the mean is removed -- and gamma (.) W is rounded once instead of W.  Measured parity: DESIGN.md 4.
//
// SPLIT mode (ANCE_ENCODER_SPLIT=1; round 4): an fp32-GRADE result at a third of the fp16 MFMA rate instead of the sixteenth the
// fp32-input matrix cores run at (precise32.h).  Every GEMM operand is an fp16 pair  v = hi + lo' 2^-11  (rows [hi | lo']),
// a product is three fp16 MFMA passes on the pipeline of the default mode (gemm256_f16.hip: gemm256_split_kernel), the
// LayerNorm fold, the fp32 softmax (precise32.h attention, fp32 Q | K | V from the QKV epilogue), the exact-erf GELU and the
// fp32 head are the reference's arithmetic.  Stated tolerance 2e-5 (tests/test_split_model.py: 3.3e-6 on the CPU model).
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "attention.h"
#include "common.h"
#include "gemm_f16.h"
#include "precise32.h"

namespace ance {
namespace {

constexpr int H = 768;          // hidden size this build is specialised for
constexpr int HEAD_OUT = 768;   // embeddingHead output (model/models.py:145)
constexpr int S_CAP_MAX = 8192; // sequences per micro-batch
constexpr int FETCH_CHUNK = 262144;
constexpr int MAX_LANES = 2;     // activation sets / internal streams (3 and 4 lanes measured no gain: DESIGN.md 9)

// ------------------------------------------------------------------------------------ kernels --

__global__ void cvt_f32_f16_kernel(const float *src, _Float16 *dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (_Float16)src[i];
}

__device__ __forceinline__ int record_len(const int32_t *ids_or_rec, int64_t ld, const int32_t *lens, int64_t rec,
                                          int L) {
    int full;
    if (lens) full = lens[rec];
    else full = (int)__builtin_bswap32((uint32_t)ids_or_rec[rec * ld]);  // 4-byte big-endian header
    return full < 0 ? 0 : (full > L ? L : full);
}

// lengths of records [r0, r0 + n) -> out (for the host-side planner when it has no host copy)
__global__ void fetch_lens_kernel(const int32_t *base, int64_t ld, const int32_t *lens, int64_t r0, int n, int L,
                                  int32_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = record_len(base, ld, lens, r0 + i, L);
}

struct PlanArgs {
    const int32_t *base;  // records (header mode: row = [len_be, ids...]) or ids
    int64_t ld;           // row stride in int32
    const int32_t *lens;  // nullptr in header mode
    int hdr;              // 1: ids start at column 1
    int64_t g0;           // first global sequence (record * n_chunks + chunk) of this micro-batch
    int S;                // sequences in this micro-batch
    int L, n_chunks, Lc;
    int pad_id, arch;
    int T, Tpad;          // real tokens / padded to 128 (host computed, same arithmetic)
    int *seq_off, *seq_vtcol, *seq_len;
    int *tok_id, *tok_pos, *tok_vtcol;
    int4 *desc;           // attention descriptors (first token, length, V^T column, sequence), longest length bucket first
    int bstart[4];        // first descriptor of each bucket (host computed: the host knows every length)
};

__device__ __forceinline__ int len_bucket(int eff) {  // ceil(eff / 32) - 1, everything above 96 tokens in bucket 3
    const int b = (eff + 31) >> 5;
    return b > 4 ? 3 : b - 1;
}

// effective lengths + exclusive scans (token offsets; 8-aligned V^T columns; rank inside the length bucket).  One block.
__global__ void __launch_bounds__(1024) plan_kernel(const PlanArgs P) {
    __shared__ int s_tot[1024], s_tot8[1024];
    __shared__ unsigned long long s_bk[1024];  // four 16-bit bucket counts (a micro-batch has at most 8,192 sequences)
    const int tid = threadIdx.x;
    const int per = (P.S + 1023) / 1024;
    const int b0 = tid * per;
    int sum = 0, sum8 = 0;
    unsigned long long bk = 0;
    for (int j = 0; j < per; ++j) {
        const int s = b0 + j;
        if (s < P.S) {
            const int64_t gs = P.g0 + s;
            const int64_t rec = gs / P.n_chunks;
            const int c = (int)(gs - rec * P.n_chunks);
            const int full = record_len(P.base, P.ld, P.lens, rec, P.L);
            int lc = full - c * P.Lc;
            lc = lc < 0 ? 0 : (lc > P.Lc ? P.Lc : lc);
            P.seq_len[s] = lc;
            const int eff = lc > 0 ? lc : 1;
            sum += eff;
            sum8 += (eff + 7) & ~7;
            bk += 1ull << (16 * len_bucket(eff));
        }
    }
    s_tot[tid] = sum;
    s_tot8[tid] = sum8;
    s_bk[tid] = bk;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        int a = 0, a8 = 0;
        unsigned long long ab = 0;
        if (tid >= off) {
            a = s_tot[tid - off];
            a8 = s_tot8[tid - off];
            ab = s_bk[tid - off];
        }
        __syncthreads();
        s_tot[tid] += a;
        s_tot8[tid] += a8;
        s_bk[tid] += ab;
        __syncthreads();
    }
    int run = s_tot[tid] - sum, run8 = s_tot8[tid] - sum8;
    unsigned long long rbk = s_bk[tid] - bk;  // sequences of each bucket before this thread's
    for (int j = 0; j < per; ++j) {
        const int s = b0 + j;
        if (s < P.S) {
            const int lc = P.seq_len[s];
            const int eff = lc > 0 ? lc : 1;
            P.seq_off[s] = run;
            P.seq_vtcol[s] = run8;
            const int b = len_bucket(eff);
            P.desc[P.bstart[b] + (int)((rbk >> (16 * b)) & 0xFFFF)] = make_int4(run, eff, run8, s);
            rbk += 1ull << (16 * b);
            run += eff;
            run8 += (eff + 7) & ~7;
        }
    }
    if (tid == 1023) P.seq_off[P.S] = s_tot[1023];
}

// one wave per sequence: packed token ids, position ids, V^T columns
__global__ void __launch_bounds__(256) pack_kernel(const PlanArgs P) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (s < P.S) {
        const int64_t gs = P.g0 + s;
        const int64_t rec = gs / P.n_chunks;
        const int c = (int)(gs - rec * P.n_chunks);
        const int lc = P.seq_len[s];
        const int t0 = P.seq_off[s], v0 = P.seq_vtcol[s];
        if (lc == 0) {
            // all-pad chunk == one pad token attending to itself
            if (l == 0) {
                P.tok_id[t0] = P.pad_id;
                P.tok_pos[t0] = P.arch == ANCE_ARCH_ROBERTA ? P.pad_id : 0;
                P.tok_vtcol[t0] = v0;
            }
        } else {
            const int32_t *src = P.base + rec * P.ld + P.hdr + c * P.Lc;
            int before = 0;  // non-pad tokens seen so far (RoBERTa position ids)
            for (int j0 = 0; j0 < lc; j0 += 64) {
                const int j = j0 + l;
                const bool in = j < lc;
                const int id = in ? src[j] : P.pad_id;
                const bool nonpad = in && id != P.pad_id;
                const u64 m = __ballot(nonpad);
                if (in) {
                    int pos;
                    if (P.arch == ANCE_ARCH_ROBERTA)
                        pos = nonpad ? before + __popcll(m & ((2ull << l) - 1ull)) + P.pad_id : P.pad_id;
                    else
                        pos = j;
                    P.tok_id[t0 + j] = id;
                    P.tok_pos[t0 + j] = pos;
                    P.tok_vtcol[t0 + j] = v0 + j;
                }
                before += __popcll(m);
            }
        }
    }
    // rows T..Tpad exist only to fill the last GEMM tile
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt < P.Tpad - P.T) {
        P.tok_id[P.T + gt] = P.pad_id;
        P.tok_pos[P.T + gt] = P.arch == ANCE_ARCH_ROBERTA ? P.pad_id : 0;
        P.tok_vtcol[P.T + gt] = 0;
    }
}

// LayerNorm of one 768-wide row held as 12 floats per lane (3 x float4, lane-contiguous)
// LayerNorm of one 768-wide row held by one wave (3 float4 per lane): fp16 output for the next GEMM and the
// row statistics for the consumers of the fp32 value (see the file header).
__device__ __forceinline__ void ln_row_store(f32x4 v0, f32x4 v1, f32x4 v2, const float *gamma, const float *beta,
                                             float eps, _Float16 *out16, float *stats, int l) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v0[j] + v1[j] + v2[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v0[j] - mean, b = v1[j] - mean, c = v2[j] - mean;
        q += a * a + b * b + c * c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q * (1.0f / H) + eps);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gamma);
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(beta);
    f32x4 vin[3] = {v0, v1, v2};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int c4 = p * 64 + l;  // float4 index within the row
        const f32x4 y = ln_apply4(vin[p], mean, rstd, g4[c4], b4[c4]);
        reinterpret_cast<f16x4 *>(out16)[c4] = f16x4{(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
    }
    if (l == 0) {
        stats[0] = mean;
        stats[1] = rstd;
    }
}

__global__ void __launch_bounds__(256) embed_ln_kernel(const int *tok_id, const int *tok_pos, int Tpad, const float *word,
                                                       const float *pos, const float *type0, int vocab, int max_pos,
                                                       const float *gamma, const float *beta, float eps, float *pre,
                                                       _Float16 *h16, float *stats) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (t >= Tpad) return;
    int id = tok_id[t], p = tok_pos[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(word + (size_t)id * H);
    const f32x4 *p4 = reinterpret_cast<const f32x4 *>(pos + (size_t)p * H);
    const f32x4 *t4 = reinterpret_cast<const f32x4 *>(type0);
    f32x4 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c4 = k * 64 + l;
        v[k] = (w4[c4] + t4[c4]) + p4[c4];  // same association as the reference: (word + type) + pos
        reinterpret_cast<f32x4 *>(pre + (size_t)t * H)[c4] = v[k];
    }
    ln_row_store(v[0], v[1], v[2], gamma, beta, eps, h16 + (size_t)t * H, stats + 2 * (size_t)t, l);
}

__global__ void __launch_bounds__(256) ln_kernel(const float *pre, int Tpad, const float *gamma, const float *beta, float eps,
                                                 _Float16 *h16, float *stats) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (t >= Tpad) return;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(pre + (size_t)t * H);
    ln_row_store(x4[l], x4[64 + l], x4[128 + l], gamma, beta, eps, h16 + (size_t)t * H, stats + 2 * (size_t)t, l);
}

// last layer, CLS-only tail: compact residual rows  dst[s] = LN(pre[seq_off[s]])  (rows S..S_pad zeroed)
__global__ void __launch_bounds__(256) gather_cls_kernel(const float *pre, const float *stats, const float *gamma,
                                                         const float *beta, const int *seq_off, int S, int S_pad, float *dst) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (s >= S_pad) return;
    f32x4 *d4 = reinterpret_cast<f32x4 *>(dst + (size_t)s * H);
    if (s < S) {
        const size_t row = (size_t)seq_off[s];
        const f32x4 *s4 = reinterpret_cast<const f32x4 *>(pre + row * H);
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            d4[k * 64 + l] = ln_apply4(s4[k * 64 + l], mean, rstd, reinterpret_cast<const f32x4 *>(gamma)[k * 64 + l],
                                       reinterpret_cast<const f32x4 *>(beta)[k * 64 + l]);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) d4[k * 64 + l] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// hi / lo / ldp: the fp16 pair of the pre-LayerNorm row -- pair_w = 0: two planes (default mode, lo = fp16(v - hi)); pair_w = W: lo is
// null and hi points to pair rows of the split mode (common.h: pair_hi_col / pair_lo_col)
__global__ void __launch_bounds__(256) head_kernel(const float *pre, const _Float16 *hi, const _Float16 *lo, int ldp, int pair_w,
                                                   const float *stats, const float *part, float eps, const float *lng, const float *lnb,
                                                   const int *seq_off, int compact, const float *W, const float *b,
                                                   const float *gamma, const float *beta, int has_head, float *out, unsigned *faults) {
    __shared__ float cls[H];
    __shared__ float z[HEAD_OUT];
    __shared__ float red[8];
    const int s = blockIdx.x, tid = threadIdx.x;
    const size_t row = (size_t)(compact ? s : seq_off[s]);  // compact: row s already is the [CLS] row
    float mean_h, rstd_h;  // h = LN(pre), recomputed (file header)
    if (part) stats_from_parts(part + row * 24, eps, &mean_h, &rstd_h);
    else { mean_h = stats[2 * row]; rstd_h = stats[2 * row + 1]; }
    float *dst = out + (size_t)s * HEAD_OUT;
    auto src = [&](int j) {
        if (!hi) return pre[row * H + j];
        if (pair_w) return (float)hi[row * ldp + pair_hi_col(j, pair_w)] + (float)hi[row * ldp + pair_lo_col(j, pair_w)] * PAIR_LO_INV;
        return (float)hi[row * ldp + j] + (float)lo[row * ldp + j];
    };
    if (!has_head) {
        for (int j = tid; j < H; j += 256) dst[j] = (src(j) - mean_h) * rstd_h * lng[j] + lnb[j];
        // a NaN anywhere in the row (or an infinity: rstd 0) shows in its statistics: count the row (ance_encoder_range_faults [1])
        if (tid == 0 && faults && (!(fabsf(mean_h) < INFINITY) || !(rstd_h > 0.f && rstd_h < INFINITY))) atomicAdd(faults + 1, 1u);
        return;
    }
    for (int j = tid; j < H; j += 256) cls[j] = (src(j) - mean_h) * rstd_h * lng[j] + lnb[j];
    __syncthreads();
    // each wave computes output features n = w, w+4, ...: lanes split k, reduce by shuffle
    const int w = tid >> 6, l = tid & 63;
    for (int n = w; n < HEAD_OUT; n += 4) {
        const float *wr = W + (size_t)n * H;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < H / 64; ++k) acc = fmaf(wr[k * 64 + l], cls[k * 64 + l], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (l == 0) z[n] = acc + b[n];
    }
    __syncthreads();
    float sm = 0.f;
    for (int j = tid; j < HEAD_OUT; j += 256) sm += z[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sm += __shfl_xor(sm, off);
    if (l == 0) red[w] = sm;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) * (1.0f / HEAD_OUT);
    float q = 0.f;
    for (int j = tid; j < HEAD_OUT; j += 256) {
        const float a = z[j] - mean;
        q += a * a;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    if (l == 0) red[4 + w] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) * (1.0f / HEAD_OUT) + 1e-5f);
    for (int j = tid; j < HEAD_OUT; j += 256) dst[j] = (z[j] - mean) * rstd * gamma[j] + beta[j];
    if (tid == 0 && faults && (!(fabsf(mean) < INFINITY) || !(rstd > 0.f && rstd < INFINITY))) atomicAdd(faults + 1, 1u);
}


// ---- folded-LayerNorm path ---------------------------------------------------------------------
__device__ __forceinline__ void split_store(const f32x4 v, _Float16 *hi, _Float16 *lo, int c4) {
    const f16x4 h = cvt_f16x4_pinned(v);
    const f16x4 r = f16x4{(_Float16)(v[0] - (float)h[0]), (_Float16)(v[1] - (float)h[1]), (_Float16)(v[2] - (float)h[2]),
                          (_Float16)(v[3] - (float)h[3])};
    reinterpret_cast<f16x4 *>(hi)[c4] = h;
    reinterpret_cast<f16x4 *>(lo)[c4] = r;
}
__device__ __forceinline__ f32x4 pair_load(const _Float16 *hi, const _Float16 *lo, int c4) {
    const f16x4 h = reinterpret_cast<const f16x4 *>(hi)[c4], r = reinterpret_cast<const f16x4 *>(lo)[c4];
    return f32x4{(float)h[0] + (float)r[0], (float)h[1] + (float)r[1], (float)h[2] + (float)r[2], (float)h[3] + (float)r[3]};
}

// embeddings -> (hi, lo) pair of the pre-LayerNorm row + the (mean, M2) of its twelve 64-column slices (the format the
// RES epilogue leaves: gemm_f16.h); one wave per token, 16 lanes per slice
__global__ void __launch_bounds__(256) embed_fold_kernel(const int *tok_id, const int *tok_pos, int Tpad, const float *word,
                                                         const float *pos, const float *type0, int vocab, int max_pos,
                                                         _Float16 *hi, _Float16 *lo, float *part) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (t >= Tpad) return;
    int id = tok_id[t], p = tok_pos[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(word + (size_t)id * H);
    const f32x4 *p4 = reinterpret_cast<const f32x4 *>(pos + (size_t)p * H);
    const f32x4 *t4 = reinterpret_cast<const f32x4 *>(type0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c4 = k * 64 + l;  // columns 4 c4 .. 4 c4 + 3: slice 4 k + l / 16
        const f32x4 v = (w4[c4] + t4[c4]) + p4[c4];  // same association as the reference: (word + type) + pos
        split_store(v, hi + (size_t)t * H, lo + (size_t)t * H, c4);
        const float m64 = row16_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
        const float d0 = v[0] - m64, d1 = v[1] - m64, d2 = v[2] - m64, d3 = v[3] - m64;
        const float q64 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        if ((l & 15) == 0) {
            float *pp = part + ((size_t)t * 12 + 4 * k + (l >> 4)) * 2;
            pp[0] = m64;
            pp[1] = q64;
        }
    }
}

// last layer, CLS-only tail: compact (hi, lo, slice partials) rows of the [CLS] tokens; rows S..S_pad zeroed
__global__ void __launch_bounds__(256) gather_cls_fold_kernel(const _Float16 *hi, const _Float16 *lo, const float *part,
                                                              const int *seq_off, int S, int S_pad, _Float16 *chi, _Float16 *clo,
                                                              float *cpart) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (s >= S_pad) return;
    f16x4 *dh = reinterpret_cast<f16x4 *>(chi + (size_t)s * H), *dl = reinterpret_cast<f16x4 *>(clo + (size_t)s * H);
    if (s < S) {
        const size_t row = (size_t)seq_off[s];
        const f16x4 *sh = reinterpret_cast<const f16x4 *>(hi + row * H), *sl = reinterpret_cast<const f16x4 *>(lo + row * H);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dh[k * 64 + l] = sh[k * 64 + l];
            dl[k * 64 + l] = sl[k * 64 + l];
        }
        if (l < 24) cpart[(size_t)s * 24 + l] = part[row * 24 + l];
    } else {
        const f16x4 z = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dh[k * 64 + l] = z;
            dl[k * 64 + l] = z;
        }
        if (l < 24) cpart[(size_t)s * 24 + l] = (l & 1) ? 64.0f : 0.f;  // mean 0, variance 1: a finite rstd
    }
}

// weight load with the LayerNorm folded in (K = 768 columns): W16[n][k] = fp16(gamma[k] W[n][k]),
// csum[n] = sum_k W16[n][k] (over the ROUNDED values), bout[n] = b[n] + sum_k beta[k] W[n][k].  One wave per row.
__global__ void __launch_bounds__(256) fold_weight_kernel(const float *W, const float *b, const float *gamma, const float *beta,
                                                          int N, _Float16 *W16, float *csum, float *bout) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (n >= N) return;
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(W + (size_t)n * H);
    float cs = 0.f, bs = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c4 = k * 64 + l;
        const f32x4 w = w4[c4], g = reinterpret_cast<const f32x4 *>(gamma)[c4], be = reinterpret_cast<const f32x4 *>(beta)[c4];
        const f16x4 h = f16x4{(_Float16)(g[0] * w[0]), (_Float16)(g[1] * w[1]), (_Float16)(g[2] * w[2]), (_Float16)(g[3] * w[3])};
        reinterpret_cast<f16x4 *>(W16 + (size_t)n * H)[c4] = h;
        cs += ((float)h[0] + (float)h[1]) + ((float)h[2] + (float)h[3]);
        bs += (be[0] * w[0] + be[1] * w[1]) + (be[2] * w[2] + be[3] * w[3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cs += __shfl_xor(cs, off);
        bs += __shfl_xor(bs, off);
    }
    if (l == 0) {
        csum[n] = cs;
        bout[n] = b[n] + bs;
    }
}

// ---- split mode ---------------------------------------------------------------------------------
constexpr int HP = 2 * H;  // halves per pair row of a 768-wide stream (common.h: blocked [hi (32) | lo (32)] column blocks)

// embeddings -> pair rows of the pre-LayerNorm stream + the slice statistics (format of EPI_S_RESLN); one wave per token.
// The row is stored times EMB_SCALE: the lo half of a pair is unscaled (common.h), i.e. good to 2^-25 ABSOLUTE, which is fp32-grade for
// the O(1) post-residual streams but not for an embedding sum of magnitude 0.05 (trained BERT / RoBERTa checkpoints) that a LayerNorm
// with rstd ~ 20 then blows up.  A LayerNorm is scale-invariant up to its epsilon: the two consumers of this stream (layer
// 0's Q | K | V GEMM and the residual of its attention-output GEMM) run with eps EMB_SCALE^2 -- powers of two, so (v s - mean s) and
// rstd / s are the unscaled values' bits.
constexpr float EMB_SCALE = 16.0f;
__global__ void __launch_bounds__(256) embed_split_kernel(const int *tok_id, const int *tok_pos, int Tpad, const float *word,
                                                          const float *pos, const float *type0, int vocab, int max_pos,
                                                          _Float16 *xp, float *part, unsigned *faults) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (t >= Tpad) return;
    int id = tok_id[t], p = tok_pos[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(word + (size_t)id * H);
    const f32x4 *p4 = reinterpret_cast<const f32x4 *>(pos + (size_t)p * H);
    const f32x4 *t4 = reinterpret_cast<const f32x4 *>(type0);
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c4 = k * 64 + l;
        const f32x4 v = ((w4[c4] + t4[c4]) + p4[c4]) * EMB_SCALE;  // same association as the reference: (word + type) + pos
        range_track4(v, &mx);
        pair_store4(v, xp + (size_t)t * HP, H, c4 * 4);
        const float m64 = row16_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
        const float d0 = v[0] - m64, d1 = v[1] - m64, d2 = v[2] - m64, d3 = v[3] - m64;
        const float q64 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        if ((l & 15) == 0) {
            float *pp = part + ((size_t)t * 12 + 4 * k + (l >> 4)) * 2;
            pp[0] = m64;
            pp[1] = q64;
        }
    }
    range_report(mx, faults);
}

// Per-matrix scale of the split GEMM's weights: s = 2^p with max |g (.) W| s in [2^13, 2^14).  Pair halves are UNSCALED differences
// (common.h): lo = fp16(w s - hi) of an element below 2^-3 is an fp16 subnormal, good to 2^-25 absolute -- with the largest element
// at 2^13 that is 2^-38 of it, i.e. nothing (unscaled, a 0.02 weight would keep 19 bits).  wabsmax_kernel leaves max |g (.) W| as
// float bits (non-negative floats order as integers) with atomicMax; one slot per weight matrix, Q / K / V share one.
__global__ void __launch_bounds__(256) wabsmax_kernel(const float *W, const float *gamma, int N, int K, unsigned *slot) {
    const size_t total4 = (size_t)N * K / 4;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        f32x4 w = reinterpret_cast<const f32x4 *>(W)[i];
        if (gamma) w = w * reinterpret_cast<const f32x4 *>(gamma)[i % (size_t)(K / 4)];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fabsf(w[2]), fabsf(w[3]))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.f && m < INFINITY) atomicMax(slot, __builtin_bit_cast(unsigned, m));
}
__device__ __forceinline__ float weight_pair_scale(const unsigned *slot) {
    const float m = __builtin_bit_cast(float, *slot);
    if (!(m > 0.f)) return 1.0f;
    int e;
    (void)frexpf(m, &e);  // m in [2^(e-1), 2^e)
    int p = 14 - e;
    p = p < -60 ? -60 : (p > 60 ? 60 : p);
    return ldexpf(1.0f, p);
}

// weight load for the split GEMM: row n of W [N, K] -> pair row (common.h) of  s (g (.) W)  (g = the LayerNorm weight folded in, or
// null; s = the matrix's power-of-two scale, its inverse left in *winv for the GEMM epilogue), csum[n] = sum_k (hi + lo) / s
// (what the MFMAs actually multiply), bout[n] = b[n] + sum_k beta[k] W[n][k].  One wave per row.
__global__ void __launch_bounds__(256) split_weight_kernel(const float *W, const float *b, const float *gamma, const float *beta,
                                                           int N, int K, const unsigned *slot, _Float16 *Wp, float *csum, float *bout,
                                                           float *winv) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (n >= N) return;
    const float sc = weight_pair_scale(slot);
    if (n == 0 && l == 0) *winv = 1.0f / sc;
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(W + (size_t)n * K);
    _Float16 *row = Wp + (size_t)n * 2 * K;
    float cs = 0.f, bs = 0.f;
    for (int c4 = l; c4 < K / 4; c4 += 64) {
        f32x4 w = w4[c4];
        if (gamma) {
            const f32x4 g = reinterpret_cast<const f32x4 *>(gamma)[c4], be = reinterpret_cast<const f32x4 *>(beta)[c4];
            bs += (be[0] * w[0] + be[1] * w[1]) + (be[2] * w[2] + be[3] * w[3]);
            w = w * g;
        }
        w = w * sc;
        pair_store4(w, row, K, c4 * 4);
        const f32x4 back = pair_load4(row, K, c4 * 4);
        cs += (back[0] + back[1]) + (back[2] + back[3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cs += __shfl_xor(cs, off);
        bs += __shfl_xor(bs, off);
    }
    if (l == 0) {
        if (csum) csum[n] = cs * (1.0f / sc);
        if (bout) bout[n] = b[n] + bs;
    }
}

// last layer, CLS-only tail in split mode: compact pair rows + slice partials of the [CLS] tokens; rows S..S_pad zeroed
__global__ void __launch_bounds__(256) gather_cls_split_kernel(const _Float16 *xp, const float *part, const int *seq_off, int S,
                                                               int S_pad, _Float16 *cxp, float *cpart) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (s >= S_pad) return;
    f16x8 *d = reinterpret_cast<f16x8 *>(cxp + (size_t)s * HP);
    if (s < S) {
        const size_t row = (size_t)seq_off[s];
        const f16x8 *sp = reinterpret_cast<const f16x8 *>(xp + row * HP);
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k * 64 + l] = sp[k * 64 + l];
        if (l < 24) cpart[(size_t)s * 24 + l] = part[row * 24 + l];
    } else {
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k * 64 + l] = z;
        if (l < 24) cpart[(size_t)s * 24 + l] = (l & 1) ? 64.0f : 0.f;  // mean 0, variance 1: a finite rstd
    }
}

// ---- embeddingHead as one fp32 MFMA GEMM over the [CLS] rows (model/models.py:145-152) ----------
// z[s][n] = sum_k LN(cls_s)[k] W[n][k] + b[n]; 32 sequences x 128 features per workgroup, one 32 x 32 tile per wave
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation in k order).  The 32 normalised [CLS] rows are staged
// once in LDS (96.5 KiB); W rows stream from L2 as 16-byte pieces per lane.  A block per sequence re-read the whole
// 2.36 MB W (2 GB of L2 reads per 883 rows, 166 us); this reads it 28 times.
constexpr int HEAD_LDA = H + 4;  // floats; 16 lanes x stride 4 banks: conflict-free ds_read_b128
constexpr size_t HEAD_LDS_BYTES = (size_t)32 * HEAD_LDA * sizeof(float);

__global__ void __launch_bounds__(256) head_gemm_kernel(const float *pre32, const _Float16 *hi, const _Float16 *lo, int ldp,
                                                        int pair_w, const float *stats, const float *part, float eps, const float *lng,
                                                        const float *lnb, const int *seq_off, int compact, int S, const float *W,
                                                        const float *b, float *out) {
    extern __shared__ __attribute__((aligned(16))) float cls[];
    const int s0 = blockIdx.x * 32, n0 = blockIdx.y * 128;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 5, i = l & 31;
    // stage LN(pre)[cls] of 32 sequences: wave w takes rows w, w + 4, ...
    for (int r = w; r < 32; r += 4) {
        const int s = s0 + r;
        f32x4 *dst = reinterpret_cast<f32x4 *>(cls + r * HEAD_LDA);
        if (s < S) {
            const size_t row = (size_t)(compact ? s : seq_off[s]);
            float mean, rstd;
            if (part) stats_from_parts(part + row * 24, eps, &mean, &rstd);  // (every lane: 24 cached floats)
            else { mean = stats[2 * row]; rstd = stats[2 * row + 1]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c4 = k * 64 + l;
                const f32x4 x = !hi      ? reinterpret_cast<const f32x4 *>(pre32 + row * H)[c4]
                                : pair_w ? pair_load4(hi + row * ldp, pair_w, c4 * 4)   // split mode: pair rows (see head_kernel)
                                         : pair_load(hi + row * ldp, lo + row * ldp, c4);
                dst[c4] = ln_apply4(x, mean, rstd, reinterpret_cast<const f32x4 *>(lng)[c4],
                                    reinterpret_cast<const f32x4 *>(lnb)[c4]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[k * 64 + l] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    // MFMA rows (A operand) = sequences, columns (B operand) = features n0 + 32 w + i; k-step j of a 16-byte piece uses
    // k = 8 s' + 4 g + j on both sides
    const float *ap = cls + i * HEAD_LDA + 4 * g;
    const float *wp = W + (size_t)(n0 + 32 * w + i) * H + 4 * g;
    f32x16 acc = {0};
#pragma unroll 4
    for (int kk = 0; kk < H / 8; ++kk) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + kk * 8);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(wp + kk * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc, 0, 0, 0);
    }
    // acc[r]: sequence s0 + (r & 3) + 8 (r >> 2) + 4 g, feature n0 + 32 w + i
    const int n = n0 + 32 * w + i;
    const float bias = b[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int s = s0 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (s < S) out[(size_t)s * HEAD_OUT + n] = acc[r] + bias;
    }
}

// final LayerNorm (model/models.py:146,152 "norm") of the head output, in place; one wave per sequence
// (a NaN anywhere upstream of a row -- or an infinity -- shows in these statistics: such rows are counted in faults[1])
__global__ void __launch_bounds__(256) head_ln_kernel(float *out, int S, const float *gamma, const float *beta, unsigned *faults) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (s >= S) return;
    f32x4 *z4 = reinterpret_cast<f32x4 *>(out + (size_t)s * HEAD_OUT);
    f32x4 v[3];
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = z4[k * 64 + l];
        sm += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sm += __shfl_xor(sm, off);
    const float mean = sm * (1.0f / HEAD_OUT);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = v[k][j] - mean;
            q += a * a;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q * (1.0f / HEAD_OUT) + 1e-5f);
    if (l == 0 && faults && (!(fabsf(mean) < INFINITY) || !(rstd > 0.f && rstd < INFINITY))) atomicAdd(faults + 1, 1u);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c4 = k * 64 + l;
        z4[c4] = ln_apply4(v[k], mean, rstd, reinterpret_cast<const f32x4 *>(gamma)[c4], reinterpret_cast<const f32x4 *>(beta)[c4]);
    }
}

// ------------------------------------------------------------------------------- host layout --

// AnceEncoderDesc.precision names the arithmetic of a handle; ANCE_PRECISION_DEFAULT defers to these environment switches.
// ANCE_ENCODER_PRECISE=1: the fp32 path of precise32.h (the arena and workspace sizes grow by the fp32 weights and activations, so
// the size queries resolve the mode the same way)
bool precise_env() {
    const char *p = getenv("ANCE_ENCODER_PRECISE");
    return p && p[0] == '1';
}

// The encoder arithmetic of a handle, read from the environment when it is created (and by the two size queries: the modes differ
// in arena and workspace size).  DEFAULT = the split (fp32-grade) path: the reference runs its encoder in fp32
// (drivers/run_ann_data_gen.py:158,176-180, model/models.py:149-157 -- no .half()), and it is the mode in which the refresh
// reproduces the reference's negative ids (tests/test_gpu_config1.py).  ANCE_ENCODER_PRECISE=1 selects the fp32-operand audit
// path and wins over everything; ANCE_ENCODER_FP16=1 (or ANCE_ENCODER_SPLIT=0) selects the fp16-operand fast mode (3e-3 on the
// embeddings, 2.2 x the throughput); ANCE_ENCODER_SPLIT=1 names the default explicitly and wins over ANCE_ENCODER_FP16.
bool split_env() {
    if (precise_env()) return false;
    const char *s = getenv("ANCE_ENCODER_SPLIT");
    if (s && s[0] == '1') return true;
    if (s && s[0] == '0') return false;
    const char *f = getenv("ANCE_ENCODER_FP16");
    return !(f && f[0] == '1');
}

// The arithmetic of a descriptor: AnceEncoderDesc.precision, or -- ANCE_PRECISION_DEFAULT -- what the environment says.
int resolve_precision(const AnceEncoderDesc *d) {
    if (d->precision == ANCE_PRECISION_SPLIT || d->precision == ANCE_PRECISION_FP16 || d->precision == ANCE_PRECISION_FP32)
        return d->precision;
    if (precise_env()) return ANCE_PRECISION_FP32;
    return split_env() ? ANCE_PRECISION_SPLIT : ANCE_PRECISION_FP16;
}

struct LayerW {
    _Float16 *wqk, *wv, *wo, *w1, *w2;
    _Float16 *wqkv_s, *wo_s, *w1_s, *w2_s;  // split path: pair rows [hi (K) | lo' (K)]
    float *bqkv_s, *cqkv_s, *b1_s, *c1_s;   // split path: folded biases and row sums
    float *sc_s;  // split path: [0..3] max |g (.) W| of Wqkv, Wo, W1, W2 as float bits (weight load), [4..7] the inverse of their power-of-two scales
    float *bqk, *bv, *bo, *b1, *b2, *ln1w, *ln1b, *ln2w, *ln2b;
    float *cqk, *cv, *c1;  // folded LayerNorm: per-feature sums of the folded fp16 weight rows
    float *wqkv32, *bqkv32, *wo32, *w132, *w232;  // fp32 path only
};

struct Arena {
    size_t off = 0;
    char *base = nullptr;
    template <typename T>
    T *take(size_t n) {
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off = align_up(off + n * sizeof(T), 256);
        return p;
    }
};

}  // namespace
}  // namespace ance

using namespace ance;

struct AnceEncoder {
    AnceEncoderDesc d;
    // weights
    float *word, *pos, *type0, *eln_w, *eln_b;
    std::vector<LayerW> layers;
    float *head_w, *head_b, *norm_w, *norm_b;
    // workspace: two independent "lanes" (activation sets) so that consecutive micro-batches can run
    // on two streams -- the MFMA-bound GEMMs of one overlap the HBM-bound LayerNorm / attention /
    // epilogue phases of the other
    int tcap, vcap, scap;
    int *lens_fetch;
    unsigned *faults;  // device: [0] out-of-range stores of the split mode, [1] NaN output rows
    struct Lane {
        int *seq_off, *seq_vtcol, *seq_len, *tok_id, *tok_pos, *tok_vtcol;
        float *preA, *preB;      // pre-LayerNorm rows: attention block output / FFN block output (or embeddings)
        float *statsA, *statsB;  // (mean, rstd) per row of preA / preB
        _Float16 *h16, *qk16, *vt16, *ctx16, *ffn16;
        int4 *desc;              // attention descriptors in length-bucket order
        float *x32, *xa32, *qkv32, *ctx32, *ffn32;  // fp32 path only: hidden states, Q|K|V, attention output, FFN activation
        float *partA, *partB;    // folded LayerNorm: (mean, M2) of the twelve 64-column slices of every row of the two streams
        // folded LayerNorm: preA / preB hold the (hi, lo) fp16 pairs of the stream instead of fp32 rows
        _Float16 *xa_hi() const { return reinterpret_cast<_Float16 *>(preA); }
        _Float16 *xb_hi() const { return reinterpret_cast<_Float16 *>(preB); }
    } lane[MAX_LANES];
    int n_lanes;
    hipStream_t side[MAX_LANES];
    hipEvent_t ev_fork, ev_join[MAX_LANES];
    std::vector<int32_t> host_lens;
    bool cls_tail;  // run the last layer's post-attention part on the [CLS] rows only (ANCE_CLS_TAIL=0 disables)
    bool ln_fold;   // LayerNorm folded into the GEMMs (file header; ANCE_LN_FOLD=0 disables)
    bool head_mfma; // embeddingHead as one fp32 MFMA GEMM (ANCE_HEAD_MFMA=0: one block per sequence)
    bool precise;   // fp32 path (precise32.h)
    bool split;     // split (fp32-grade) path
    bool n_split;   // FFN1 with the N-split tile order (ANCE_GEMM_NSPLIT=0 disables)
    bool split_attn; // split mode: attention on the matrix cores (ANCE_SPLIT_ATTN=0: fp32 vector-unit kernel)
    bool attn_coal; // attention Q / output rows through LDS slabs (ANCE_ATTN_COAL=0: per-lane accesses)
};

namespace {

bool desc_ok(const AnceEncoderDesc *d) {
    return d && d->hidden == H && d->n_heads == 12 && d->intermediate > 0 && d->intermediate % 128 == 0 &&
           d->n_layers >= 1 && d->vocab_size > 0 && d->max_position > 0 && d->max_seq_len >= 1 &&
           d->max_seq_len <= 512 && d->max_tokens >= 512 && d->max_tokens % 256 == 0 && d->precision >= 0 && d->precision <= 3 &&
           (d->arch == ANCE_ARCH_ROBERTA || d->arch == ANCE_ARCH_BERT);
}

void layout_weights(const AnceEncoderDesc *d, Arena &a, AnceEncoder *e) {
    const size_t I = d->intermediate;
    const int mode = resolve_precision(d);
    float *word = a.take<float>((size_t)d->vocab_size * H);
    float *pos = a.take<float>((size_t)d->max_position * H);
    float *type0 = a.take<float>(H);
    float *ew = a.take<float>(H), *eb = a.take<float>(H);
    if (e) { e->word = word; e->pos = pos; e->type0 = type0; e->eln_w = ew; e->eln_b = eb; e->layers.resize(d->n_layers); }
    for (int i = 0; i < d->n_layers; ++i) {
        LayerW w;
        w.wqk = a.take<_Float16>((size_t)2 * H * H);
        w.bqk = a.take<float>(2 * H);
        w.wv = a.take<_Float16>((size_t)H * H);
        w.bv = a.take<float>(H);
        w.wo = a.take<_Float16>((size_t)H * H);
        w.bo = a.take<float>(H);
        w.ln1w = a.take<float>(H);
        w.ln1b = a.take<float>(H);
        w.w1 = a.take<_Float16>(I * H);
        w.b1 = a.take<float>(I);
        w.w2 = a.take<_Float16>((size_t)H * I);
        w.b2 = a.take<float>(H);
        w.ln2w = a.take<float>(H);
        w.ln2b = a.take<float>(H);
        w.cqk = a.take<float>(2 * H);
        w.cv = a.take<float>(H);
        w.c1 = a.take<float>(I);
        w.wqkv_s = w.wo_s = w.w1_s = w.w2_s = nullptr;
        w.bqkv_s = w.cqkv_s = w.b1_s = w.c1_s = w.sc_s = nullptr;
        if (mode == ANCE_PRECISION_SPLIT) {
            w.sc_s = a.take<float>(8);
            w.wqkv_s = a.take<_Float16>((size_t)3 * H * HP);
            w.bqkv_s = a.take<float>(3 * H);
            w.cqkv_s = a.take<float>(3 * H);
            w.wo_s = a.take<_Float16>((size_t)H * HP);
            w.w1_s = a.take<_Float16>(I * HP);
            w.b1_s = a.take<float>(I);
            w.c1_s = a.take<float>(I);
            w.w2_s = a.take<_Float16>((size_t)H * 2 * I);
        }
        w.wqkv32 = w.bqkv32 = w.wo32 = w.w132 = w.w232 = nullptr;
        if (mode == ANCE_PRECISION_FP32) {
            w.wqkv32 = a.take<float>((size_t)3 * H * H);
            w.bqkv32 = a.take<float>(3 * H);
            w.wo32 = a.take<float>((size_t)H * H);
            w.w132 = a.take<float>(I * H);
            w.w232 = a.take<float>((size_t)H * I);
        }
        if (e) e->layers[i] = w;
    }
    float *hw = nullptr, *hb = nullptr, *nw = nullptr, *nb = nullptr;
    if (d->has_head) {
        hw = a.take<float>((size_t)HEAD_OUT * H);
        hb = a.take<float>(HEAD_OUT);
        nw = a.take<float>(HEAD_OUT);
        nb = a.take<float>(HEAD_OUT);
    }
    if (e) { e->head_w = hw; e->head_b = hb; e->norm_w = nw; e->norm_b = nb; }
}

void layout_workspace(const AnceEncoderDesc *d, Arena &a, AnceEncoder *e) {
    const int tcap = d->max_tokens;
    const int scap = tcap < S_CAP_MAX ? tcap : S_CAP_MAX;
    const int vcap = (int)align_up((size_t)tcap + tcap / 4 + 256, 256);
    const int mode = resolve_precision(d);
    unsigned *faults = a.take<unsigned>(64);  // [0] out-of-range stores of the split mode, [1] NaN output rows (ance_encoder_range_faults)
    int *lens_fetch = a.take<int>(FETCH_CHUNK);
    if (e) {
        e->tcap = tcap; e->scap = scap; e->vcap = vcap; e->lens_fetch = lens_fetch; e->faults = faults;
    }
    for (int ln = 0; ln < MAX_LANES; ++ln) {
        AnceEncoder::Lane L;
        L.seq_off = a.take<int>(scap + 1); L.seq_vtcol = a.take<int>(scap); L.seq_len = a.take<int>(scap);
        L.tok_id = a.take<int>(tcap); L.tok_pos = a.take<int>(tcap); L.tok_vtcol = a.take<int>(tcap);
        L.desc = a.take<int4>(scap);
        L.preB = a.take<float>((size_t)tcap * H); L.preA = a.take<float>((size_t)tcap * H);
        L.statsA = a.take<float>((size_t)tcap * 2); L.statsB = a.take<float>((size_t)tcap * 2);
        L.h16 = a.take<_Float16>((size_t)tcap * H);
        L.qk16 = a.take<_Float16>((size_t)tcap * 2 * H);
        L.vt16 = a.take<_Float16>((size_t)H * vcap);
        L.ctx16 = a.take<_Float16>((size_t)tcap * H);
        L.ffn16 = a.take<_Float16>((size_t)tcap * d->intermediate);
        L.partA = a.take<float>((size_t)tcap * (H / 64) * 2);
        L.partB = a.take<float>((size_t)tcap * (H / 64) * 2);
        L.x32 = L.xa32 = L.qkv32 = L.ctx32 = L.ffn32 = nullptr;
        if (mode == ANCE_PRECISION_FP32) {
            L.x32 = a.take<float>((size_t)tcap * H);
            L.xa32 = a.take<float>((size_t)tcap * H);
        }
        if (mode == ANCE_PRECISION_FP32 || mode == ANCE_PRECISION_SPLIT) {
            // split mode: qkv32 = fp32 Q | K | V; ctx32 / ffn32 hold the PAIR rows of the attention output / FFN activation
            // (an fp16 pair row is as many bytes as the fp32 row)
            L.qkv32 = a.take<float>((size_t)tcap * 3 * H);
            L.ctx32 = a.take<float>((size_t)tcap * H);
            L.ffn32 = a.take<float>((size_t)tcap * d->intermediate);
        }
        if (e) e->lane[ln] = L;
    }
}

void cvt16(const void *src, _Float16 *dst, size_t n, hipStream_t st) {
    const unsigned blocks = (unsigned)((n + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(cvt_f32_f16_kernel, dim3(blocks > 4096 ? 4096 : (blocks ? blocks : 1)), dim3(256), 0, st,
                       (const float *)src, dst, n);
}
void cpy32(const void *src, float *dst, size_t n, hipStream_t st) {
    (void)hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
}

int encode_impl(AnceEncoder *e, const int32_t *base, int64_t ld, const int32_t *d_lens, const int32_t *h_lens, int hdr,
                int64_t n, int L, int n_chunks, float *d_out, hipStream_t caller_st) {
    hipStream_t st = caller_st;
    if (!e || !base || !d_out || n < 0 || L < 1 || n_chunks < 1 || L % n_chunks) {
        set_last_error("ance_encode: invalid argument");
        return ANCE_E_INVALID;
    }
    const int Lc = L / n_chunks;
    if (Lc > e->d.max_seq_len) {
        set_last_error("ance_encode: chunk length exceeds desc.max_seq_len");
        return ANCE_E_INVALID;
    }
    const AnceEncoderDesc &D = e->d;
    const int I = D.intermediate;
    int mb_index = 0;
    bool forked = false;

    for (int64_t r0 = 0; r0 < n; r0 += FETCH_CHUNK) {
        const int nr = (int)((n - r0) < FETCH_CHUNK ? (n - r0) : FETCH_CHUNK);
        const int32_t *hl;
        if (h_lens) {
            hl = h_lens + r0;
        } else {
            // no host copy of the lengths: read them back once (the only synchronising path)
            e->host_lens.resize(nr);
            hipLaunchKernelGGL(fetch_lens_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, base, ld, d_lens, r0, nr, L,
                               e->lens_fetch);
            if (hipMemcpyAsync(e->host_lens.data(), e->lens_fetch, (size_t)nr * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
                return check_launch("ance_encode: length read-back");
            hl = e->host_lens.data();
        }
        if (e->n_lanes > 1 && !forked) {  // side streams start after everything already queued by the caller
            (void)hipEventRecord(e->ev_fork, caller_st);
            for (int ln = 0; ln < e->n_lanes; ++ln) (void)hipStreamWaitEvent(e->side[ln], e->ev_fork, 0);
            forked = true;
        }
        // ---- greedy micro-batches over the sequences (record, chunk) of this block of records ---
        const int64_t gs_end = (int64_t)nr * n_chunks;
        int64_t gs = 0;
        while (gs < gs_end) {
            const AnceEncoder::Lane &LN = e->lane[mb_index % e->n_lanes];
            hipStream_t st = e->n_lanes > 1 ? e->side[mb_index % e->n_lanes] : caller_st;
            ++mb_index;
            int S = 0, T = 0, V = 0, maxlen = 1;
            int n_bucket[4] = {0, 0, 0, 0};
            int64_t g = gs;
            while (g < gs_end && S < e->scap) {
                const int64_t rec = g / n_chunks;
                const int c = (int)(g - rec * n_chunks);
                int full = hl[rec];
                full = full < 0 ? 0 : (full > L ? L : full);
                int lc = full - c * Lc;
                lc = lc < 0 ? 0 : (lc > Lc ? Lc : lc);
                const int eff = lc > 0 ? lc : 1;
                const int v8 = (eff + 7) & ~7;
                if (T + eff > e->tcap || V + v8 > e->vcap - 256) break;
                T += eff; V += v8; ++S; ++g;
                if (eff > maxlen) maxlen = eff;
                const int nb = (eff + 31) >> 5;
                ++n_bucket[nb > 4 ? 3 : nb - 1];  // = len_bucket(eff) of plan_kernel
            }
            if (S == 0) {
                set_last_error("ance_encode: max_tokens too small for one sequence");
                return ANCE_E_INVALID;
            }
            const int Tpad = (int)align_up((size_t)T, 256);
            const int ldvt = (int)align_up((size_t)V, 256);

            PlanArgs P;
            P.base = base; P.ld = ld; P.lens = d_lens; P.hdr = hdr;
            P.g0 = r0 * n_chunks + gs; P.S = S; P.L = L; P.n_chunks = n_chunks; P.Lc = Lc;
            P.pad_id = D.pad_token_id; P.arch = D.arch; P.T = T; P.Tpad = Tpad;
            P.seq_off = LN.seq_off; P.seq_vtcol = LN.seq_vtcol; P.seq_len = LN.seq_len;
            P.tok_id = LN.tok_id; P.tok_pos = LN.tok_pos; P.tok_vtcol = LN.tok_vtcol;
            P.desc = LN.desc;
            P.bstart[3] = 0;  // longest sequences first
            for (int b = 2; b >= 0; --b) P.bstart[b] = P.bstart[b + 1] + n_bucket[b + 1];
            {
                ProfScope ps(PC_PLAN, st);
                hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, st, P);
                const int nb_seq = (S + 3) / 4, nb_pad = (Tpad - T + 255) / 256;
                hipLaunchKernelGGL(pack_kernel, dim3(nb_seq > nb_pad ? nb_seq : nb_pad), dim3(256), 0, st, P);
            }
            if (e->precise) {
                // ---- fp32 path (precise32.h): plain sequence of fp32 kernels, every layer on every token ----
                int rc = ANCE_OK;
                {
                    ProfScope pe(PC_EMBED, st);
                    hipLaunchKernelGGL(embed32_kernel, dim3(Tpad / 4), dim3(256), 0, st, LN.tok_id, LN.tok_pos, Tpad, e->word, e->pos,
                                       e->type0, D.vocab_size, D.max_position, LN.preB);
                    hipLaunchKernelGGL(ln32_kernel, dim3(Tpad / 4), dim3(256), 0, st, LN.preB, Tpad, e->eln_w, e->eln_b, D.ln_eps,
                                       LN.x32, (float *)nullptr);
                }
                for (int li = 0; li < D.n_layers && !rc; ++li) {
                    const LayerW &W = e->layers[li];
                    const bool last = li == D.n_layers - 1;
                    {
                        ProfScope ps(PC_GEMM_QK, st, 2.0 * T * (3.0 * H) * H);
                        rc = launch_gemm32(P_EPI_BIAS, LN.x32, H, W.wqkv32, H, W.bqkv32, nullptr, 0, LN.qkv32, 3 * H, Tpad, 3 * H, H, st);
                    }
                    if (rc) break;
                    {
                        ProfScope ps(PC_ATTN, st);
                        rc = launch_attention32(LN.qkv32, LN.ctx32, LN.seq_off, S, D.n_heads, st);
                    }
                    if (rc) break;
                    {
                        ProfScope ps(PC_GEMM_OUT, st, 2.0 * T * (double)H * H);
                        rc = launch_gemm32(P_EPI_RES, LN.ctx32, H, W.wo32, H, W.bo, LN.x32, H, LN.preA, H, Tpad, H, H, st);
                    }
                    if (rc) break;
                    {
                        ProfScope ps(PC_LN, st);
                        hipLaunchKernelGGL(ln32_kernel, dim3(Tpad / 4), dim3(256), 0, st, LN.preA, Tpad, W.ln1w, W.ln1b, D.ln_eps, LN.xa32,
                                           (float *)nullptr);
                    }
                    {
                        ProfScope ps(PC_GEMM_FFN1, st, 2.0 * T * (double)I * H);
                        rc = launch_gemm32(P_EPI_GELU, LN.xa32, H, W.w132, H, W.b1, nullptr, 0, LN.ffn32, I, Tpad, I, H, st);
                    }
                    if (rc) break;
                    {
                        ProfScope ps(PC_GEMM_FFN2, st, 2.0 * T * (double)I * H);
                        rc = launch_gemm32(P_EPI_RES, LN.ffn32, I, W.w232, I, W.b2, LN.xa32, H, LN.preB, H, Tpad, H, I, st);
                    }
                    if (rc) break;
                    {
                        ProfScope ps(PC_LN, st);
                        hipLaunchKernelGGL(ln32_kernel, dim3(Tpad / 4), dim3(256), 0, st, LN.preB, Tpad, W.ln2w, W.ln2b, D.ln_eps, LN.x32,
                                           last ? LN.statsB : (float *)nullptr);
                    }
                }
                if (rc) return rc;
                {
                    ProfScope ps(PC_HEAD, st);
                    const LayerW &WL = e->layers[D.n_layers - 1];
                    float *dst = d_out + (size_t)(r0 * n_chunks + gs) * HEAD_OUT;
                    if (D.has_head) {
                        hipLaunchKernelGGL(head_gemm_kernel, dim3((S + 31) / 32, HEAD_OUT / 128), dim3(256), HEAD_LDS_BYTES, st, LN.preB,
                                           (const _Float16 *)nullptr, (const _Float16 *)nullptr, H, 0, LN.statsB, (const float *)nullptr,
                                           D.ln_eps, WL.ln2w, WL.ln2b, LN.seq_off, 0, S, e->head_w, e->head_b, dst);
                        hipLaunchKernelGGL(head_ln_kernel, dim3((S + 3) / 4), dim3(256), 0, st, dst, S, e->norm_w, e->norm_b, e->faults);
                    } else {
                        hipLaunchKernelGGL(head_kernel, dim3(S), dim3(256), 0, st, LN.preB, (const _Float16 *)nullptr,
                                           (const _Float16 *)nullptr, H, 0, LN.statsB, (const float *)nullptr, D.ln_eps, WL.ln2w, WL.ln2b,
                                           LN.seq_off, 0, e->head_w, e->head_b, e->norm_w, e->norm_b, 0, dst, e->faults);
                    }
                }
                gs = g;
                continue;
            }
            if (e->split) {
                // ---- split (fp32-grade) path: the default mode's schedule with pair operands and the fp32 attention ----
                _Float16 *const xa = reinterpret_cast<_Float16 *>(LN.preA), *const xb = reinterpret_cast<_Float16 *>(LN.preB);
                _Float16 *const ctxp = reinterpret_cast<_Float16 *>(LN.ctx32), *const ffnp = reinterpret_cast<_Float16 *>(LN.ffn32);
                {
                    ProfScope pe(PC_EMBED, st);
                    hipLaunchKernelGGL(embed_split_kernel, dim3(Tpad / 4), dim3(256), 0, st, LN.tok_id, LN.tok_pos, Tpad, e->word,
                                       e->pos, e->type0, D.vocab_size, D.max_position, xb, LN.partB, e->faults);
                }
                const bool cls_tail = e->cls_tail;
                const int S_pad = (int)align_up((size_t)S, 256);
                int rc = ANCE_OK;
                for (int li = 0; li < D.n_layers && !rc; ++li) {
                    const LayerW &W = e->layers[li];
                    const bool tail = cls_tail && li == D.n_layers - 1;
                    const int Mrows = tail ? S_pad : Tpad;
                    const double Mwork = tail ? (double)S : (double)T;
                    // layer 0 reads the embedding stream, stored times EMB_SCALE: its LayerNorm runs with eps EMB_SCALE^2 (embed_split_kernel)
                    const float eps_b = li == 0 ? D.ln_eps * (EMB_SCALE * EMB_SCALE) : D.ln_eps;
                    GemmArgs G;
                    memset(&G, 0, sizeof(G));
                    // Q | K | V projection -> fp32 (the LayerNorm that produces this layer's input is folded in)
                    G.A = xb; G.lda = HP; G.B = W.wqkv_s; G.ldb = HP; G.M = Tpad; G.N = 3 * H; G.K = H;
                    G.bias = W.bqkv_s; G.csum = W.cqkv_s; G.part_in = LN.partB; G.ln_eps = eps_b;
                    G.out32 = LN.qkv32; G.ldc = 3 * H; G.wscale_inv = W.sc_s + 4; G.range_faults = e->faults;
                    // CLS-only tail: the last layer's attention has ONE query per sequence.  K | V of every token, but Q of the [CLS]
                    // rows only: their pair rows and slice partials are compacted first (they are also the residual of the
                    // attention-output GEMM below) and projected by a second, small launch into the Q columns of rows 0 .. S_pad of
                    // qkv32 -- row s = the query of sequence s (attention_split_kernel, cls_only).  A third of this GEMM's work in one
                    // layer of twelve; every element is the same arithmetic as in the full launch.
                    const bool tail_q = tail && e->split_attn;
                    float *const cpt = reinterpret_cast<float *>(ffnp + (size_t)S_pad * HP);  // (the FFN buffer is dead here)
                    if (tail) {
                        ProfScope ps(PC_LN, st);
                        hipLaunchKernelGGL(gather_cls_split_kernel, dim3(S_pad / 4), dim3(256), 0, st, xb, LN.partB, LN.seq_off, S, S_pad,
                                           ffnp, cpt);
                    }
                    {
                        ProfScope ps(PC_GEMM_QK, st, tail_q ? 2.0 * T * (2.0 * H) * H + 2.0 * S * (double)H * H : 2.0 * T * (3.0 * H) * H);
                        if (tail_q) {
                            G.B = W.wqkv_s + (size_t)H * HP; G.N = 2 * H; G.bias = W.bqkv_s + H; G.csum = W.cqkv_s + H;
                            G.out32 = LN.qkv32 + H;
                            rc = launch_gemm_f16(EPI_S_QKV, G, st);
                            G.A = ffnp; G.part_in = cpt; G.M = S_pad;
                            G.B = W.wqkv_s; G.N = H; G.bias = W.bqkv_s; G.csum = W.cqkv_s; G.out32 = LN.qkv32;
                            if (!rc) rc = launch_gemm_f16(EPI_S_QKV, G, st);
                        } else {
                            rc = launch_gemm_f16(EPI_S_QKV, G, st);
                        }
                    }
                    if (rc) break;
                    {
                        ProfScope ps(PC_ATTN, st);
                        if (tail && S_pad > S)  // rows S..S_pad of the compact attention output feed the GEMM tile: keep them finite
                            (void)hipMemsetAsync(ctxp + (size_t)S * HP, 0, (size_t)(S_pad - S) * HP * sizeof(_Float16), st);
                        // (cls_only: one query per sequence, read from row s of the Q columns -- the tail's compact Q projection above)
                        rc = e->split_attn ? launch_attention_split(LN.qkv32, ctxp, LN.desc, S, D.n_heads, maxlen, tail ? 1 : 0, st)
                                           : ANCE_E_INVALID;
                        if (rc == ANCE_E_INVALID)  // ANCE_SPLIT_ATTN=0: the fp32 vector-unit kernel (A/B)
                            rc = launch_attention32(LN.qkv32, nullptr, LN.seq_off, S, D.n_heads, st, ctxp, tail ? 1 : 0);
                    }
                    if (rc) break;
                    // attention.output.dense + residual LayerNorm(x_b), x_b = the previous layer's output (or the embeddings)
                    memset(&G, 0, sizeof(G));
                    G.res_gamma = li == 0 ? e->eln_w : e->layers[li - 1].ln2w;
                    G.res_beta = li == 0 ? e->eln_b : e->layers[li - 1].ln2b;
                    G.res_hi = xb; G.ldr = HP; G.part_in = LN.partB; G.ln_eps = eps_b; G.range_faults = e->faults;
                    if (tail) {  // the compact pair rows + partials of the [CLS] tokens (gathered in front of the QKV GEMM)
                        G.res_hi = ffnp; G.part_in = cpt;
                    }
                    G.A = ctxp; G.lda = HP; G.B = W.wo_s; G.ldb = HP; G.M = Mrows; G.N = H; G.K = H;
                    G.bias = W.bo; G.out16 = xa; G.ldc = HP; G.part_out = LN.partA; G.wscale_inv = W.sc_s + 5;
                    {
                        ProfScope ps(PC_GEMM_OUT, st, 2.0 * Mwork * (double)H * H);
                        rc = launch_gemm_f16(EPI_S_RESLN, G, st);
                    }
                    if (rc) break;
                    // intermediate.dense + exact GELU (attention.output.LayerNorm folded in)
                    memset(&G, 0, sizeof(G));
                    G.A = xa; G.lda = HP; G.B = W.w1_s; G.ldb = HP; G.M = Mrows; G.N = I; G.K = H;
                    G.bias = W.b1_s; G.csum = W.c1_s; G.part_in = LN.partA; G.ln_eps = D.ln_eps;
                    G.out16 = ffnp; G.ldc = 2 * I; G.n_split = (e->n_split && (I / 256) % 2 == 0) ? 2 : 0; G.wscale_inv = W.sc_s + 6;
                    G.range_faults = e->faults;
                    {
                        ProfScope ps(PC_GEMM_FFN1, st, 2.0 * Mwork * (double)I * H);
                        rc = launch_gemm_f16(EPI_S_GELU, G, st);
                    }
                    if (rc) break;
                    // output.dense + residual LayerNorm(x_a)
                    memset(&G, 0, sizeof(G));
                    G.A = ffnp; G.lda = 2 * I; G.B = W.w2_s; G.ldb = 2 * I; G.M = Mrows; G.N = H; G.K = I;
                    G.bias = W.b2; G.res_gamma = W.ln1w; G.res_beta = W.ln1b; G.res_hi = xa; G.ldr = HP;
                    G.part_in = LN.partA; G.ln_eps = D.ln_eps; G.out16 = xb; G.ldc = HP; G.part_out = LN.partB; G.wscale_inv = W.sc_s + 7;
                    G.range_faults = e->faults;
                    {
                        ProfScope ps(PC_GEMM_FFN2, st, 2.0 * Mwork * (double)I * H);
                        rc = launch_gemm_f16(EPI_S_RESLN, G, st);
                    }
                }
                if (rc) return rc;
                {
                    ProfScope ps(PC_HEAD, st);
                    const LayerW &WL = e->layers[D.n_layers - 1];
                    float *dst = d_out + (size_t)(r0 * n_chunks + gs) * HEAD_OUT;
                    if (D.has_head) {
                        hipLaunchKernelGGL(head_gemm_kernel, dim3((S + 31) / 32, HEAD_OUT / 128), dim3(256), HEAD_LDS_BYTES, st,
                                           (const float *)nullptr, (const _Float16 *)xb, (const _Float16 *)nullptr, HP, H,
                                           (const float *)nullptr, (const float *)LN.partB, D.ln_eps, WL.ln2w, WL.ln2b, LN.seq_off,
                                           cls_tail ? 1 : 0, S, e->head_w, e->head_b, dst);
                        hipLaunchKernelGGL(head_ln_kernel, dim3((S + 3) / 4), dim3(256), 0, st, dst, S, e->norm_w, e->norm_b, e->faults);
                    } else {
                        hipLaunchKernelGGL(head_kernel, dim3(S), dim3(256), 0, st, (const float *)nullptr, (const _Float16 *)xb,
                                           (const _Float16 *)nullptr, HP, H, (const float *)nullptr, (const float *)LN.partB,
                                           D.ln_eps, WL.ln2w, WL.ln2b, LN.seq_off, cls_tail ? 1 : 0, e->head_w, e->head_b, e->norm_w,
                                           e->norm_b, 0, dst, e->faults);
                    }
                }
                gs = g;
                continue;
            }
            const bool fold = e->ln_fold;
            // folded LayerNorm: the two halves of the fp16 pair of a stream share the fp32 row's 3 KB
            _Float16 *const xa_hi = LN.xa_hi(), *const xa_lo = xa_hi + (size_t)e->tcap * H;
            _Float16 *const xb_hi = LN.xb_hi(), *const xb_lo = xb_hi + (size_t)e->tcap * H;
            {
            ProfScope pe(PC_EMBED, st);
            if (fold)
                hipLaunchKernelGGL(embed_fold_kernel, dim3(Tpad / 4), dim3(256), 0, st, LN.tok_id, LN.tok_pos, Tpad, e->word, e->pos,
                                   e->type0, D.vocab_size, D.max_position, xb_hi, xb_lo, LN.partB);
            else
                hipLaunchKernelGGL(embed_ln_kernel, dim3(Tpad / 4), dim3(256), 0, st, LN.tok_id, LN.tok_pos, Tpad, e->word, e->pos,
                                   e->type0, D.vocab_size, D.max_position, e->eln_w, e->eln_b, D.ln_eps, LN.preB, LN.h16, LN.statsB);
            }
            // Only the [CLS] row of the last layer reaches the head (model/models.py:49,152): after the
            // last layer's K / V projections everything runs on the S compact [CLS] rows.
            const bool cls_tail = e->cls_tail;
            const int S_pad = (int)align_up((size_t)S, 256);
            for (int li = 0; li < D.n_layers; ++li) {
                const LayerW &W = e->layers[li];
                const bool tail = cls_tail && li == D.n_layers - 1;
                const int Mrows = tail ? S_pad : Tpad;   // rows of the post-attention GEMMs / LayerNorms
                const double Mwork = tail ? (double)S : (double)T;
                GemmArgs G;
                memset(&G, 0, sizeof(G));
                // Q | K projection
                G.A = fold ? xb_hi : LN.h16; G.lda = H; G.B = W.wqk; G.ldb = H; G.M = Tpad; G.N = 2 * H; G.K = H;
                G.bias = W.bqk; G.out16 = LN.qk16; G.ldc = 2 * H; G.scale_cols = H;
                G.scale = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) and log2(e): the softmax runs on exp2
                G.part_in = LN.partB; G.ln_eps = D.ln_eps; G.csum = W.cqk; G.tok_lo = fold ? xb_lo : nullptr;
                // CLS-only tail (folded form): K of every token, Q of the compact [CLS] rows only -- row s of the Q columns is the query
                // of sequence s (attention_kernel, cls_only); the same arithmetic per element as the full launch (split mode: above)
                const bool tail_q = tail && fold;
                _Float16 *const chi = LN.ffn16, *const clo = chi + (size_t)S_pad * H;  // (the FFN buffer is dead here)
                float *const cpt = reinterpret_cast<float *>(clo + (size_t)S_pad * H);
                if (tail_q) {
                    ProfScope ps(PC_LN, st);
                    hipLaunchKernelGGL(gather_cls_fold_kernel, dim3(S_pad / 4), dim3(256), 0, st, xb_hi, xb_lo, LN.partB, LN.seq_off,
                                       S, S_pad, chi, clo, cpt);
                }
                int rc;
                {
                    ProfScope ps(PC_GEMM_QK, st, tail_q ? 2.0 * T * (double)H * H + 2.0 * S * (double)H * H : 2.0 * T * (2.0 * H) * H);
                    if (tail_q) {
                        G.B = W.wqk + (size_t)H * H; G.N = H; G.bias = W.bqk + H; G.csum = W.cqk + H; G.out16 = LN.qk16 + H; G.scale_cols = 0;
                        rc = launch_gemm_f16(EPI_QK_F, G, st);
                        G.A = chi; G.tok_lo = clo; G.part_in = cpt; G.M = S_pad;
                        G.B = W.wqk; G.bias = W.bqk; G.csum = W.cqk; G.out16 = LN.qk16; G.scale_cols = H;
                        if (!rc) rc = launch_gemm_f16(EPI_QK_F, G, st);
                    } else {
                        rc = launch_gemm_f16(fold ? EPI_QK_F : EPI_QK, G, st);
                    }
                }
                if (rc) return rc;
                // V^T = Wv h^T
                memset(&G, 0, sizeof(G));
                G.A = W.wv; G.lda = H; G.B = fold ? xb_hi : LN.h16; G.ldb = H; G.M = H; G.N = Tpad; G.K = H;
                G.bias = W.bv; G.out16 = LN.vt16; G.ldc = ldvt; G.col_map = LN.tok_vtcol; G.n_valid = T;
                G.part_in = LN.partB; G.ln_eps = D.ln_eps; G.csum = W.cv; G.tok_lo = fold ? xb_lo : nullptr;
                {
                    ProfScope ps(PC_GEMM_VT, st, 2.0 * T * (double)H * H);
                    rc = launch_gemm_f16(fold ? EPI_VT_F : EPI_VT, G, st);
                }
                if (rc) return rc;
                AttnArgs A;
                A.qk = LN.qk16; A.vt = LN.vt16; A.ctx = LN.ctx16; A.desc = LN.desc;
                A.ld_qk = 2 * H; A.ld_vt = ldvt; A.ld_ctx = H; A.n_heads = D.n_heads; A.cls_only = tail ? 1 : 0; A.q_compact = tail_q ? 1 : 0; A.coalesced = e->attn_coal ? 1 : 0;
                {
                    ProfScope ps(PC_ATTN, st, 0.0);
                    rc = launch_attention(A, S, maxlen, st);
                }
                if (rc) return rc;
                // attention.output.dense + residual.  The residual is LN(preB) with the LayerNorm that produced
                // this layer's input: the previous layer's output.LayerNorm, or the embedding LayerNorm.
                const float *rg = li == 0 ? e->eln_w : e->layers[li - 1].ln2w;
                const float *rb = li == 0 ? e->eln_b : e->layers[li - 1].ln2b;
                memset(&G, 0, sizeof(G));
                G.res_stats = LN.statsB; G.res_gamma = rg; G.res_beta = rb;
                if (fold) {
                    G.res_hi = xb_hi; G.res_lo = xb_lo; G.part_in = LN.partB; G.ln_eps = D.ln_eps;
                    if (tail) {  // the compact (hi, lo, partials) rows of the [CLS] tokens (gathered in front of the Q | K GEMM)
                        G.res_hi = chi; G.res_lo = clo; G.part_in = cpt;
                    }
                    G.out16 = xa_hi; G.out_lo = xa_lo; G.part_out = LN.partA;
                } else {
                    G.res32 = LN.preB;
                    if (tail) {  // compact residual rows (already normalised), parked in the (currently dead) FFN buffer
                        float *rc = reinterpret_cast<float *>(LN.ffn16);
                        ProfScope ps(PC_LN, st);
                        hipLaunchKernelGGL(gather_cls_kernel, dim3(S_pad / 4), dim3(256), 0, st, LN.preB, LN.statsB, rg, rb, LN.seq_off,
                                           S, S_pad, rc);
                        G.res32 = rc; G.res_stats = nullptr;
                    }
                    G.out32 = LN.preA;
                }
                G.A = LN.ctx16; G.lda = H; G.B = W.wo; G.ldb = H; G.M = Mrows; G.N = H; G.K = H;
                G.bias = W.bo; G.ldc = H;
                {
                    ProfScope ps(PC_GEMM_OUT, st, 2.0 * Mwork * (double)H * H);
                    rc = launch_gemm_f16(fold ? EPI_RESLN : EPI_RES32, G, st);
                }
                if (rc) return rc;
                if (!fold) {
                    ProfScope ps(PC_LN, st);
                    hipLaunchKernelGGL(ln_kernel, dim3(Mrows / 4), dim3(256), 0, st, LN.preA, Mrows, W.ln1w, W.ln1b, D.ln_eps,
                                       LN.h16, LN.statsA);
                }
                // intermediate.dense + GELU
                memset(&G, 0, sizeof(G));
                G.A = fold ? xa_hi : LN.h16; G.lda = H; G.B = W.w1; G.ldb = H; G.M = Mrows; G.N = I; G.K = H;
                G.bias = W.b1; G.out16 = LN.ffn16; G.ldc = I;
                G.part_in = LN.partA; G.ln_eps = D.ln_eps; G.csum = W.c1; G.tok_lo = fold ? xa_lo : nullptr;
                G.n_split = (fold && e->n_split && (I / 256) % 2 == 0) ? 2 : 0;
                {
                    ProfScope ps(PC_GEMM_FFN1, st, 2.0 * Mwork * (double)I * H);
                    rc = launch_gemm_f16(fold ? EPI_GELU_F : EPI_GELU, G, st);
                }
                if (rc) return rc;
                // output.dense + residual
                memset(&G, 0, sizeof(G));
                G.A = LN.ffn16; G.lda = I; G.B = W.w2; G.ldb = I; G.M = Mrows; G.N = H; G.K = I;
                G.bias = W.b2; G.ldc = H;
                G.res_stats = LN.statsA; G.res_gamma = W.ln1w; G.res_beta = W.ln1b;
                if (fold) {
                    G.res_hi = xa_hi; G.res_lo = xa_lo; G.out16 = xb_hi; G.out_lo = xb_lo;
                    G.part_in = LN.partA; G.ln_eps = D.ln_eps; G.part_out = LN.partB;
                } else {
                    G.res32 = LN.preA; G.out32 = LN.preB;
                }
                {
                    ProfScope ps(PC_GEMM_FFN2, st, 2.0 * Mwork * (double)I * H);
                    rc = launch_gemm_f16(fold ? EPI_RESLN : EPI_RES32, G, st);
                }
                if (rc) return rc;
                if (!fold) {
                    ProfScope ps(PC_LN, st);
                    hipLaunchKernelGGL(ln_kernel, dim3(Mrows / 4), dim3(256), 0, st, LN.preB, Mrows, W.ln2w, W.ln2b, D.ln_eps,
                                       LN.h16, LN.statsB);
                }
            }
            {
                ProfScope ps(PC_HEAD, st);
                const LayerW &WL = e->layers[D.n_layers - 1];
                float *dst = d_out + (size_t)(r0 * n_chunks + gs) * HEAD_OUT;
                const _Float16 *hh = fold ? xb_hi : nullptr, *hl = fold ? xb_lo : nullptr;
                if (D.has_head && e->head_mfma) {
                    hipLaunchKernelGGL(head_gemm_kernel, dim3((S + 31) / 32, HEAD_OUT / 128), dim3(256), HEAD_LDS_BYTES, st, LN.preB, hh,
                                       hl, H, 0, LN.statsB, fold ? LN.partB : (const float *)nullptr, D.ln_eps, WL.ln2w, WL.ln2b, LN.seq_off,
                                       cls_tail ? 1 : 0, S, e->head_w, e->head_b, dst);
                    hipLaunchKernelGGL(head_ln_kernel, dim3((S + 3) / 4), dim3(256), 0, st, dst, S, e->norm_w, e->norm_b, e->faults);
                } else {
                    hipLaunchKernelGGL(head_kernel, dim3(S), dim3(256), 0, st, LN.preB, hh, hl, H, 0, LN.statsB,
                                       fold ? LN.partB : (const float *)nullptr, D.ln_eps, WL.ln2w, WL.ln2b, LN.seq_off,
                                       cls_tail ? 1 : 0, e->head_w, e->head_b, e->norm_w, e->norm_b, D.has_head, dst, e->faults);
                }
            }
            gs = g;
        }
        if (forked && r0 + FETCH_CHUNK < n && !h_lens) {
            // the next block of records needs a length read-back on the caller's stream: join first
            for (int ln = 0; ln < e->n_lanes; ++ln) {
                (void)hipEventRecord(e->ev_join[ln], e->side[ln]);
                (void)hipStreamWaitEvent(caller_st, e->ev_join[ln], 0);
            }
            forked = false;
        }
    }
    if (forked) {  // the caller's stream continues only after both side streams are done
        for (int ln = 0; ln < e->n_lanes; ++ln) {
            (void)hipEventRecord(e->ev_join[ln], e->side[ln]);
            (void)hipStreamWaitEvent(caller_st, e->ev_join[ln], 0);
        }
    }
    return check_launch("ance_encode");
}

}  // namespace

extern "C" size_t ance_encoder_weight_bytes(const AnceEncoderDesc *desc) {
    if (!desc_ok(desc)) return 0;
    Arena a;
    layout_weights(desc, a, nullptr);
    return a.off + 256;
}

extern "C" size_t ance_encoder_workspace_bytes(const AnceEncoderDesc *desc) {
    if (!desc_ok(desc)) return 0;
    Arena a;
    layout_workspace(desc, a, nullptr);
    return a.off + 256;
}

extern "C" int ance_encoder_create(const AnceEncoderDesc *desc, const void *const *w, int n_weights, void *d_weight_arena,
                                   size_t weight_bytes, void *d_workspace, size_t workspace_bytes, void *stream,
                                   AnceEncoder **out) {
    if (!desc_ok(desc) || !w || !out || !d_weight_arena || !d_workspace ||
        n_weights != ANCE_ENCODER_N_WEIGHTS(desc->n_layers, desc->has_head)) {
        set_last_error("ance_encoder_create: invalid descriptor or weight list");
        return ANCE_E_INVALID;
    }
    for (int i = 0; i < n_weights; ++i)
        if (!w[i]) {
            set_last_error("ance_encoder_create: null weight pointer");
            return ANCE_E_INVALID;
        }
    if (weight_bytes < ance_encoder_weight_bytes(desc) || workspace_bytes < ance_encoder_workspace_bytes(desc)) {
        set_last_error("ance_encoder_create: arena or workspace too small");
        return ANCE_E_WORKSPACE;
    }
    AnceEncoder *e = new (std::nothrow) AnceEncoder();
    if (!e) return ANCE_E_NOMEM;
    e->d = *desc;
    {
        const char *ct = getenv("ANCE_CLS_TAIL");
        e->cls_tail = !(ct && ct[0] == '0');
        const char *lf = getenv("ANCE_LN_FOLD");
        e->ln_fold = !(lf && lf[0] == '0');
        const char *hm = getenv("ANCE_HEAD_MFMA");
        e->head_mfma = !(hm && hm[0] == '0');
        const char *ac = getenv("ANCE_ATTN_COAL");
        e->attn_coal = !(ac && ac[0] == '0');
        const int mode = resolve_precision(desc);
        e->precise = mode == ANCE_PRECISION_FP32;
        e->split = mode == ANCE_PRECISION_SPLIT;
        if (e->precise || e->split) e->ln_fold = false;  // these paths take the plain biases and LayerNorm parameters
        const char *nsp = getenv("ANCE_GEMM_NSPLIT");
        e->n_split = !(nsp && nsp[0] == '0');
        const char *sa = getenv("ANCE_SPLIT_ATTN");
        e->split_attn = !(sa && sa[0] == '0');
        const char *ns = getenv("ANCE_ENCODER_STREAMS");
        e->n_lanes = (ns && ns[0] >= '1' && ns[0] <= '0' + MAX_LANES) ? ns[0] - '0' : 2;
    }
    for (int ln = 0; ln < MAX_LANES; ++ln) {
        e->side[ln] = nullptr;
        e->ev_join[ln] = nullptr;
    }
    e->ev_fork = nullptr;
    if (e->n_lanes > 1) {
        bool ok = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int ln = 0; ln < e->n_lanes && ok; ++ln)
            ok = hipStreamCreateWithFlags(&e->side[ln], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&e->ev_join[ln], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            delete e;
            return check_launch("ance_encoder_create: streams");
        }
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(head_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)HEAD_LDS_BYTES) != hipSuccess) {  // per device: set for the device this handle lives on
        delete e;
        return check_launch("ance_encoder_create: head attr");
    }
    Arena wa, xa;
    wa.base = reinterpret_cast<char *>(align_up((uintptr_t)d_weight_arena, 256));
    xa.base = reinterpret_cast<char *>(align_up((uintptr_t)d_workspace, 256));
    layout_weights(desc, wa, e);
    layout_workspace(desc, xa, e);
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(xa.base, 0, xa.off, st);  // pad rows must never hold NaN bit patterns

    const size_t I = desc->intermediate;
    cpy32(w[0], e->word, (size_t)desc->vocab_size * H, st);
    cpy32(w[1], e->pos, (size_t)desc->max_position * H, st);
    cpy32(w[2], e->type0, H, st);
    cpy32(w[3], e->eln_w, H, st);
    cpy32(w[4], e->eln_b, H, st);
    for (int i = 0; i < desc->n_layers; ++i) {
        const void *const *p = w + 5 + 16 * i;
        LayerW &L = e->layers[i];
        if (e->ln_fold) {
            // the LayerNorm that produces this layer's input: embeddings.LayerNorm or the previous output.LayerNorm
            const float *gin = (const float *)(i == 0 ? w[3] : w[5 + 16 * (i - 1) + 14]);
            const float *bin = (const float *)(i == 0 ? w[4] : w[5 + 16 * (i - 1) + 15]);
            auto foldw = [&](const void *W, const void *b, const float *g, const float *be, int N, _Float16 *W16, float *cs,
                             float *bo) {
                hipLaunchKernelGGL(fold_weight_kernel, dim3((N + 3) / 4), dim3(256), 0, st, (const float *)W, (const float *)b, g,
                                   be, N, W16, cs, bo);
            };
            foldw(p[0], p[1], gin, bin, H, L.wqk, L.cqk, L.bqk);                                     // query
            foldw(p[2], p[3], gin, bin, H, L.wqk + (size_t)H * H, L.cqk + H, L.bqk + H);            // key
            foldw(p[4], p[5], gin, bin, H, L.wv, L.cv, L.bv);                                        // value
            foldw(p[10], p[11], (const float *)p[8], (const float *)p[9], (int)I, L.w1, L.c1, L.b1);  // intermediate.dense
        } else {
            cvt16(p[0], L.wqk, (size_t)H * H, st);                 // query
            cvt16(p[2], L.wqk + (size_t)H * H, (size_t)H * H, st);  // key
            cpy32(p[1], L.bqk, H, st);
            cpy32(p[3], L.bqk + H, H, st);
            cvt16(p[4], L.wv, (size_t)H * H, st);
            cpy32(p[5], L.bv, H, st);
            cvt16(p[10], L.w1, I * H, st);
            cpy32(p[11], L.b1, I, st);
        }
        cvt16(p[6], L.wo, (size_t)H * H, st);
        cpy32(p[7], L.bo, H, st);
        cpy32(p[8], L.ln1w, H, st);
        cpy32(p[9], L.ln1b, H, st);
        if (e->split) {
            const float *gin = (const float *)(i == 0 ? w[3] : w[5 + 16 * (i - 1) + 14]);
            const float *bin = (const float *)(i == 0 ? w[4] : w[5 + 16 * (i - 1) + 15]);
            unsigned *slots = reinterpret_cast<unsigned *>(L.sc_s);
            (void)hipMemsetAsync(L.sc_s, 0, 4 * sizeof(float), st);
            auto mx = [&](const void *W, const float *g, int N, int K, int slot) {
                hipLaunchKernelGGL(wabsmax_kernel, dim3(256), dim3(256), 0, st, (const float *)W, g, N, K, slots + slot);
            };
            auto sw = [&](const void *W, const void *b, const float *g, const float *be, int N, int K, int slot, _Float16 *Wp, float *cs,
                          float *bo) {
                hipLaunchKernelGGL(split_weight_kernel, dim3((N + 3) / 4), dim3(256), 0, st, (const float *)W, (const float *)b, g, be,
                                   N, K, (const unsigned *)(slots + slot), Wp, cs, bo, L.sc_s + 4 + slot);
            };
            mx(p[0], gin, H, H, 0);  // Q, K and V are one operand matrix: one scale
            mx(p[2], gin, H, H, 0);
            mx(p[4], gin, H, H, 0);
            mx(p[6], nullptr, H, H, 1);
            mx(p[10], (const float *)p[8], (int)I, H, 2);
            mx(p[12], nullptr, H, (int)I, 3);
            sw(p[0], p[1], gin, bin, H, H, 0, L.wqkv_s, L.cqkv_s, L.bqkv_s);                                             // query
            sw(p[2], p[3], gin, bin, H, H, 0, L.wqkv_s + (size_t)H * HP, L.cqkv_s + H, L.bqkv_s + H);                    // key
            sw(p[4], p[5], gin, bin, H, H, 0, L.wqkv_s + (size_t)2 * H * HP, L.cqkv_s + 2 * H, L.bqkv_s + 2 * H);        // value
            sw(p[6], nullptr, nullptr, nullptr, H, H, 1, L.wo_s, nullptr, nullptr);                                      // attention.output.dense
            sw(p[10], p[11], (const float *)p[8], (const float *)p[9], (int)I, H, 2, L.w1_s, L.c1_s, L.b1_s);            // intermediate.dense
            sw(p[12], nullptr, nullptr, nullptr, H, (int)I, 3, L.w2_s, nullptr, nullptr);                                // output.dense
        }
        if (e->precise) {
            cpy32(p[0], L.wqkv32, (size_t)H * H, st);
            cpy32(p[2], L.wqkv32 + (size_t)H * H, (size_t)H * H, st);
            cpy32(p[4], L.wqkv32 + (size_t)2 * H * H, (size_t)H * H, st);
            cpy32(p[1], L.bqkv32, H, st);
            cpy32(p[3], L.bqkv32 + H, H, st);
            cpy32(p[5], L.bqkv32 + 2 * H, H, st);
            cpy32(p[6], L.wo32, (size_t)H * H, st);
            cpy32(p[10], L.w132, I * H, st);
            cpy32(p[12], L.w232, (size_t)H * I, st);
        }
        cvt16(p[12], L.w2, (size_t)H * I, st);
        cpy32(p[13], L.b2, H, st);
        cpy32(p[14], L.ln2w, H, st);
        cpy32(p[15], L.ln2b, H, st);
    }
    if (desc->has_head) {
        const void *const *p = w + 5 + 16 * desc->n_layers;
        cpy32(p[0], e->head_w, (size_t)HEAD_OUT * H, st);
        cpy32(p[1], e->head_b, HEAD_OUT, st);
        cpy32(p[2], e->norm_w, HEAD_OUT, st);
        cpy32(p[3], e->norm_b, HEAD_OUT, st);
    }
    int rc = check_launch("ance_encoder_create");
    if (rc) {
        delete e;
        return rc;
    }
    *out = e;
    return ANCE_OK;
}

extern "C" void ance_encoder_destroy(AnceEncoder *enc) {
    if (!enc) return;
    for (int ln = 0; ln < MAX_LANES; ++ln) {
        if (enc->side[ln]) {
            (void)hipStreamSynchronize(enc->side[ln]);
            (void)hipStreamDestroy(enc->side[ln]);
        }
        if (enc->ev_join[ln]) (void)hipEventDestroy(enc->ev_join[ln]);
    }
    if (enc->ev_fork) (void)hipEventDestroy(enc->ev_fork);
    delete enc;
}

extern "C" int ance_encoder_precision(const AnceEncoder *enc) {
    if (!enc) return ANCE_E_INVALID;
    return enc->precise ? ANCE_PRECISION_FP32 : enc->split ? ANCE_PRECISION_SPLIT : ANCE_PRECISION_FP16;
}

extern "C" int ance_encoder_range_faults(AnceEncoder *enc, uint32_t *h_out, int reset, void *stream) {
    if (!enc || !h_out) {
        set_last_error("ance_encoder_range_faults: invalid argument");
        return ANCE_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(h_out, enc->faults, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
        return check_launch("ance_encoder_range_faults");
    if (reset) (void)hipMemsetAsync(enc->faults, 0, 2 * sizeof(uint32_t), st);
    return check_launch("ance_encoder_range_faults");
}

extern "C" int ance_encode_records(AnceEncoder *enc, const void *d_records, const int32_t *h_lens, int64_t n, int L,
                                   int n_chunks, float *d_out, void *stream) {
    return encode_impl(enc, (const int32_t *)d_records, (int64_t)L + 1, nullptr, h_lens, 1, n, L, n_chunks, d_out,
                       (hipStream_t)stream);
}

extern "C" int ance_encode_ids(AnceEncoder *enc, const int32_t *d_ids, int64_t ld_ids, const int32_t *d_lens,
                               const int32_t *h_lens, int64_t n, int L, int n_chunks, float *d_out, void *stream) {
    if (!d_lens) {
        set_last_error("ance_encode_ids: d_lens is required");
        return ANCE_E_INVALID;
    }
    return encode_impl(enc, d_ids, ld_ids, d_lens, h_lens, 0, n, L, n_chunks, d_out, (hipStream_t)stream);
}

extern "C" double ance_encoder_flops_per_sequence(int T) {
    const double t = T;
    return 169869312.0 * t + 36864.0 * t * t + 1179648.0;
}
