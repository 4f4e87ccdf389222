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


# ---- deterministic weights that do not depend on any library's random stream ---------------------------------
# The golden vectors of the reference's own classes (tests/golden/) are only as durable as the weights they were
# generated with.  ``random_state_dict`` draws from torch's generator, whose stream another torch build may change
# (the pinned tests then degrade to skips).  ``det_state_dict`` has the same structure but draws every tensor from a
# counter-based generator written out here: splitmix64 of (element index, seed, crc32 of the tensor name) -> four
# 16-bit uniforms -> their sum (Irwin-Hall, n = 4: variance 1/3) -> scaled to the requested std.  Integer arithmetic
# mod 2^64, one exact int -> float64 conversion, one float64 multiply, one cast to float32: bit-identical on every
# IEEE-754 platform and every NumPy / torch version.
_SM_GOLDEN = 0x9E3779B97F4A7C15
_SM_M1 = 0xBF58476D1CE4E5B9
_SM_M2 = 0x94D049BB133111EB
_IH4_SCALE = 1.7320508075688772 / 65536.0  # sqrt(3) / 2^16: unit variance for the centred sum of four 16-bit uniforms


def det_normal(seed, name, shape, std=1.0):
    """float32 numpy array of ``shape``, mean 0, standard deviation ``std`` (bell-shaped, |x| <= 3.47 std)."""
    import zlib

    import numpy as np
    n = 1
    for s in shape:
        n *= int(s)
    key = (int(seed) * 0xD1342543DE82EF95 + zlib.crc32(name.encode("utf8")) * 0x2545F4914F6CDD1D + 0x632BE59BD9B4E019) & (2 ** 64 - 1)
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    for a in range(0, n, step):
        z = np.arange(a, min(n, a + step), dtype=np.uint64)
        z = z * np.uint64(_SM_GOLDEN) + np.uint64(key)
        z ^= z >> np.uint64(30)
        z *= np.uint64(_SM_M1)
        z ^= z >> np.uint64(27)
        z *= np.uint64(_SM_M2)
        z ^= z >> np.uint64(31)
        m = np.uint64(0xFFFF)
        s4 = (z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))
        v = (s4.astype(np.float64) - 131070.0) * (_IH4_SCALE * float(std))
        out[a:a + len(v)] = v.astype(np.float32)
    return out.reshape(shape)


def det_state_dict(kind="roberta", n_layers=12, hidden=768, inter=3072, vocab=50265, max_pos=514, seed=0, head=True,
                   prefixes=("roberta.",), std=0.02, ln_jitter=0.0):
    """``random_state_dict`` with ``det_normal`` as the generator (same names, shapes, init scale: normal-like std 0.02 for
    Linear / Embedding weights, model/models.py:31-36; LayerNorm weight 1 + jitter, biases jitter)."""
    sd = {}

    def put(name, shape, s, base=0.0):
        t = torch.from_numpy(det_normal(seed, name, shape, s)) if s > 0 else torch.zeros(*shape)
        sd[name] = t + base if base else t

    def ln(name):
        put(name + ".weight", (hidden,), ln_jitter, 1.0)
        put(name + ".bias", (hidden,), ln_jitter)

    def lin(name, out_f, in_f):
        put(name + ".weight", (out_f, in_f), std)
        put(name + ".bias", (out_f,), ln_jitter)

    for prefix in prefixes:
        e = prefix + "embeddings."
        put(e + "word_embeddings.weight", (vocab, hidden), std)
        put(e + "position_embeddings.weight", (max_pos, hidden), std)
        put(e + "token_type_embeddings.weight", (1 if kind == "roberta" else 2, hidden), std)
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
        put("norm.weight", (768,), ln_jitter, 1.0)
        put("norm.bias", (768,), ln_jitter)
    return sd


def state_dict_sha256(sd):
    """Content hash of a weight dict (names + raw float32 bytes): what the golden manifests record."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode("utf8"))
        h.update(sd[k].detach().cpu().contiguous().numpy().astype("<f4").tobytes())
    return h.hexdigest()
