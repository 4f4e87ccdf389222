// Internal interface of the encoder's variable-length attention (see attention.hip).
#pragma once
#include "common.h"

namespace ance {

struct AttnArgs {
    const _Float16 *qk;    // [T, 2 H]: Q (pre-scaled by log2(e)/sqrt(64): the softmax runs on exp2) | K, row stride ld_qk
    const _Float16 *vt;    // [H, ld_vt]: V^T, row = head * 64 + dim, column = seq_vtcol[s] + key
    _Float16 *ctx;         // [T, H], row stride ld_ctx
    const int4 *desc;      // per sequence, longest length bucket first: (first token, length, first (8-aligned) V^T column, sequence index)
    int ld_qk, ld_vt, ld_ctx;
    int n_heads;
    int cls_only;          // 1: only query 0 of every sequence is computed; its row goes to ctx[s] (compact)
    int q_compact;         // with cls_only: that query is row s of the Q columns of qk (the encoder's compact [CLS] projection)
    int coalesced;         // 1 (default): Q rows and output rows through the wave-private LDS slabs; 0: per-lane loads / stores (A/B)
};

size_t attention_lds_bytes(int max_seq_len, int n_waves);
// split (fp32-grade) mode: fp32 Q | K | V rows in, fp16 pair rows out; any sequence length.  cls_only: one query per sequence -- the
// query of sequence s is row s of the Q columns (the compact [CLS] projection of the encoder's tail), its output row ctx_pair[s]
int launch_attention_split(const float *qkv, _Float16 *ctx_pair, const int4 *desc, int n_seq, int n_heads, int max_seq_len,
                           int cls_only, hipStream_t stream);
int launch_attention(const AttnArgs &args, int n_seq, int max_seq_len, hipStream_t stream);

}  // namespace ance
