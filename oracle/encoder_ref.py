"""Encoder oracle (test infrastructure): fp32 CPU restatement of the reference dual encoders.

Follows
  model/models.py:137-157   RobertaDot_NLL_LN.query_emb/body_emb  (FirstP: LN(W h_cls + b))
  model/models.py:160-199   RobertaDot_CLF_ANN_NLL_MultiChunk.body_emb (MaxP, 4 x 512 chunks)
  model/models.py:223-259   HFBertEncoder / BiEncoder (DPR: raw [CLS] of BERT-base)
and, for the transformer stack the reference delegates to ``transformers`` (pinned 2.3.0 in
setup.py:20, not vendored), the published RoBERTa/BERT encoder as stated in
transformers/models/roberta/modeling_roberta.py (5.x: embeddings :56-155, eager attention
:158-183, self-attention :186-250, output blocks :329-398).

Weights are a dict keyed by the HF state-dict names (``roberta.embeddings.word_embeddings.weight``
...) holding fp32 torch tensors.  Pinned against the imported reference classes by
tests/golden/make_golden.py (vectors committed under tests/golden/).
"""
import math

import torch
import torch.nn.functional as F


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def roberta_position_ids(ids, pad_id=1):
    """create_position_ids_from_input_ids (modeling_roberta.py:142-155)."""
    m = (ids != pad_id).to(torch.int64)
    return torch.cumsum(m, dim=1) * m + pad_id


def _encoder_stack(sd, prefix, x, add_mask, n_layers, n_heads, eps):
    B, L, H = x.shape
    dh = H // n_heads
    for i in range(n_layers):
        p = "%sencoder.layer.%d." % (prefix, i)
        q = F.linear(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"])
        k = F.linear(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"])
        v = F.linear(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"])
        q = q.view(B, L, n_heads, dh).transpose(1, 2)
        k = k.view(B, L, n_heads, dh).transpose(1, 2)
        v = v.view(B, L, n_heads, dh).transpose(1, 2)
        s = torch.matmul(q, k.transpose(2, 3)) * (1.0 / math.sqrt(dh)) + add_mask
        a = torch.softmax(s, dim=-1)
        ctx = torch.matmul(a, v).transpose(1, 2).reshape(B, L, H)
        o = F.linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        x = _ln(o + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
        f = F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"])
        f = F.gelu(f)  # exact erf GELU
        f = F.linear(f, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        x = _ln(f + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return x


def _additive_mask(mask):
    # masked keys get dtype-min (5.x) ; 2.3.0 used -10000: both underflow to exactly 0 weight in
    # fp32 whenever at least one key is kept, and give the same (uniform / identical-key) result
    # for an all-masked row of identical pad tokens.
    keep = mask.to(torch.bool)
    add = torch.zeros(mask.shape, dtype=torch.float32, device=mask.device)  # (device: the GPU-resident reference of tests/test_gpu_retrieval.py)
    add = add.masked_fill(~keep, torch.finfo(torch.float32).min)
    return add[:, None, None, :]


def roberta_hidden(sd, ids, mask, n_layers=12, n_heads=12, eps=1e-5, pad_id=1, prefix="roberta."):
    ids = ids.to(torch.int64)
    e = prefix + "embeddings."
    pos = roberta_position_ids(ids, pad_id)
    x = sd[e + "word_embeddings.weight"][ids] + sd[e + "token_type_embeddings.weight"][0] \
        + sd[e + "position_embeddings.weight"][pos]
    x = _ln(x, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps)
    return _encoder_stack(sd, prefix, x, _additive_mask(mask), n_layers, n_heads, eps)


def rdot_nll_ln_emb(sd, ids, mask, n_layers=12, n_heads=12, eps=1e-5):
    """RobertaDot_NLL_LN.query_emb == body_emb (model/models.py:149-157), use_mean=False."""
    h = roberta_hidden(sd, ids, mask, n_layers, n_heads, eps)
    cls = h[:, 0]
    z = F.linear(cls, sd["embeddingHead.weight"], sd["embeddingHead.bias"])
    return _ln(z, sd["norm.weight"], sd["norm.bias"], 1e-5)


def rdot_nll_multi_chunk_body_emb(sd, ids, mask, base_len=512, n_layers=12, n_heads=12, eps=1e-5):
    """RobertaDot_CLF_ANN_NLL_MultiChunk.body_emb (model/models.py:165-199) -> [B, C, 768]."""
    B, full = ids.shape
    C = full // base_len
    ids_c = ids.reshape(B * C, base_len)
    mask_c = mask.reshape(B * C, base_len)
    h = roberta_hidden(sd, ids_c, mask_c, n_layers, n_heads, eps)
    z = F.linear(h[:, 0], sd["embeddingHead.weight"], sd["embeddingHead.bias"])
    z = _ln(z, sd["norm.weight"], sd["norm.bias"], 1e-5)
    return z.reshape(B, C, -1)


def bert_cls(sd, ids, mask, prefix, n_layers=12, n_heads=12, eps=1e-12):
    """HFBertEncoder.forward -> sequence_output[:, 0, :] (model/models.py:235-240); DPR's
    BiEncoder.query_emb/body_emb use prefix 'question_model.' / 'ctx_model.' (:254-259)."""
    ids = ids.to(torch.int64)
    B, L = ids.shape
    e = prefix + "embeddings."
    pos = torch.arange(L)[None, :].expand(B, L)
    x = sd[e + "word_embeddings.weight"][ids] + sd[e + "token_type_embeddings.weight"][0] \
        + sd[e + "position_embeddings.weight"][pos]
    x = _ln(x, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps)
    h = _encoder_stack(sd, prefix, x, _additive_mask(mask), n_layers, n_heads, eps)
    return h[:, 0]


def mask_from_lengths(lengths, L):
    """GetProcessingFn's attention mask: 1 x len, 0 x pad (data/msmarco_data.py:280-282)."""
    return (torch.arange(L)[None, :] < torch.as_tensor(lengths)[:, None]).to(torch.int64)


def random_state_dict(kind="roberta", n_layers=12, hidden=768, inter=3072, vocab=50265, max_pos=514,
                      seed=0, head=True, prefixes=("roberta.",), std=0.02, ln_jitter=0.0):
    """Random-init weights with the reference's init (normal std 0.02 for Linear/Embedding,
    model/models.py:31-36; LayerNorm weight 1 / bias 0, Linear bias 0).  ``ln_jitter`` > 0
    perturbs LayerNorm/bias parameters so tests exercise non-trivial gamma/beta/bias."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def normal(*shape):
        return torch.randn(*shape, generator=g) * std

    def ln(name):
        sd[name + ".weight"] = torch.ones(hidden) + ln_jitter * torch.randn(hidden, generator=g)
        sd[name + ".bias"] = ln_jitter * torch.randn(hidden, generator=g)

    def lin(name, out_f, in_f):
        sd[name + ".weight"] = normal(out_f, in_f)
        sd[name + ".bias"] = ln_jitter * torch.randn(out_f, generator=g)

    for prefix in prefixes:
        e = prefix + "embeddings."
        sd[e + "word_embeddings.weight"] = normal(vocab, hidden)
        sd[e + "position_embeddings.weight"] = normal(max_pos, hidden)
        sd[e + "token_type_embeddings.weight"] = normal(1 if kind == "roberta" else 2, hidden)
        ln(e + "LayerNorm")
        for i in range(n_layers):
            p = "%sencoder.layer.%d." % (prefix, i)
            lin(p + "attention.self.query", hidden, hidden)
            lin(p + "attention.self.key", hidden, hidden)
            lin(p + "attention.self.value", hidden, hidden)
            lin(p + "attention.output.dense", hidden, hidden)
            ln(p + "attention.output.LayerNorm")
            lin(p + "intermediate.dense", inter, hidden)
            lin(p + "output.dense", hidden, inter)
            ln(p + "output.LayerNorm")
    if head:
        lin("embeddingHead", 768, hidden)
        sd["norm.weight"] = torch.ones(768) + ln_jitter * torch.randn(768, generator=g)
        sd["norm.bias"] = ln_jitter * torch.randn(768, generator=g)
    return sd
