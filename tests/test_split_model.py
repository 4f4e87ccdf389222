"""CPU model of the fp32-grade SPLIT encoder mode (csrc/gemm256_f16.hip: gemm256_split_kernel): every GEMM operand is an
fp16 pair  v = hi + lo,  hi = fp16(v), lo = fp16(v - hi)  (UNSCALED since round 5: one accumulator takes all three products, so
they must share one scale), and a product is formed on the fp16 matrix cores as

    A B^T  ~=  sum_k  A_hi B_hi + A_lo B_hi + A_hi B_lo                       (fp32 accumulation)

-- three fp16 MFMAs per k-step from four operand tiles staged once, every partial product exact in fp32 (11 x 11 bits), the
dropped lo x lo term 2^-22 relative.  lo of an element below 2^-3 is an fp16 subnormal (kept by the conversion and by the MFMA:
tests/test_gpu_gemm.py::test_mfma_keeps_f16_subnormals) and good to 2^-25 ABSOLUTE; activations are O(1) and take that as it is,
weights (0.02) are stored times a per-matrix power of two that puts their largest element in [2^13, 2^14) and the accumulator
is multiplied by its inverse in the epilogue.  Everything else (LayerNorm fold algebra, softmax, exact-erf GELU, residual
stream) is fp32 as in the reference (model/models.py:149-157).  The model restates these rounding points in torch and checks
the stated tolerance of the mode -- max |delta| <= 2e-5 on the unit-variance output rows at 12 layers -- against the fp32
oracle; it also pins the ALGEBRA (fold with a scaled split weight, pair reconstruction) that the HIP kernels implement, and
what the weight scale is worth (without it 6.3e-6 instead of 3.0e-6: the lo halves of 0.02-weights sit at 2^-17, 8 bits).  The kernels
themselves are tested on the GPU (tests/test_gpu_encoder.py::test_split_mode_*)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import encoder_ref, synth


def h16(x):
    return x.to(torch.float16).to(torch.float32)  # gradual underflow, as v_cvt_f16_f32 with the default denormal mode


def pair(v):
    """activation pair: (hi, lo, scale = 1)"""
    hi = h16(v)
    return hi, h16(v - hi), 1.0


def weight_scale(w):
    """2^p with max |w| 2^p in [2^13, 2^14) (encoder.hip: weight_pair_scale)"""
    m = float(w.abs().max())
    return 2.0 ** (14 - math.frexp(m)[1]) if m > 0 else 1.0


def pair_w(w, scaled=True):
    s = weight_scale(w) if scaled else 1.0
    hi = h16(w * s)
    return hi, h16(w * s - hi), s


def unpair(p):
    return (p[0] + p[1]) * (1.0 / p[2])


def split_matmul(a, b):
    """pairs -> fp32 [M, N]: one accumulator for the three products, the weight's scale undone at the end (exact)."""
    (ah, al, sa), (bh, bl, sb) = a, b
    return (ah @ bh.t() + (al @ bh.t() + ah @ bl.t())) * (1.0 / (sa * sb))


def stats_from_slices(v, eps):
    T, H = v.shape
    s = v.reshape(T, H // 64, 64)
    m_i = s.mean(-1)
    q_i = ((s - m_i[..., None]) ** 2).sum(-1)
    m = m_i.mean(-1)
    q = (q_i + 64.0 * (m_i - m[:, None]) ** 2).sum(-1)
    return m, torch.rsqrt(q / H + eps)


def folded_linear_split(xp, mu, r, W, b, gamma, beta, scaled=True):
    Wp = pair_w(gamma[None, :] * W, scaled)
    c = unpair(Wp).sum(1)
    bf = b + W @ beta
    acc = split_matmul(xp, Wp)
    return r[:, None] * (acc - mu[:, None] * c[None, :]) + bf[None, :]


def attention32(q, k, v, lens, n_heads=12):
    out = torch.zeros_like(q)
    off = 0
    for T in lens:
        for h in range(n_heads):
            sl = slice(h * 64, h * 64 + 64)
            s = (q[off:off + T, sl] * 0.125) @ k[off:off + T, sl].t()
            out[off:off + T, sl] = torch.softmax(s, dim=1) @ v[off:off + T, sl]
        off += T
    return out


def run_split(sd, ids, lens, n_layers, eps=1e-5, offset=0.0, scaled=True):
    pre = "roberta."
    e = pre + "embeddings."
    rows, pos = [], []
    for s, T in enumerate(lens):
        rows.append(ids[s, :T])
        pos.append(torch.arange(T) + 2)
    tok, p = torch.cat(rows).long(), torch.cat(pos)
    v = (sd[e + "word_embeddings.weight"][tok] + sd[e + "token_type_embeddings.weight"][0]) + sd[e + "position_embeddings.weight"][p]
    g_in, b_in = sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"]

    def ln_rows(x, mu, r, g, b):
        return (x - mu[:, None]) * r[:, None] * g[None, :] + b[None, :]

    xp = pair(v)
    mu, r = stats_from_slices(v, eps)
    for i in range(n_layers):
        L = "%sencoder.layer.%d." % (pre, i)
        W = lambda n: sd[L + n + ".weight"]
        B = lambda n: sd[L + n + ".bias"]
        q = folded_linear_split(xp, mu, r, W("attention.self.query"), B("attention.self.query"), g_in, b_in, scaled)
        k = folded_linear_split(xp, mu, r, W("attention.self.key"), B("attention.self.key"), g_in, b_in, scaled)
        vv = folded_linear_split(xp, mu, r, W("attention.self.value"), B("attention.self.value"), g_in, b_in, scaled)
        res = ln_rows(unpair(xp), mu, r, g_in, b_in)
        ctx = pair(attention32(q, k, vv, lens))
        va = split_matmul(ctx, pair_w(W("attention.output.dense"), scaled)) + B("attention.output.dense") + res
        g1, b1 = sd[L + "attention.output.LayerNorm.weight"], sd[L + "attention.output.LayerNorm.bias"]
        xa = pair(va)
        mua, ra = stats_from_slices(va, eps)
        f = pair(F.gelu(folded_linear_split(xa, mua, ra, W("intermediate.dense"), B("intermediate.dense"), g1, b1, scaled)))
        resa = ln_rows(unpair(xa), mua, ra, g1, b1)
        v = split_matmul(f, pair_w(W("output.dense"), scaled)) + B("output.dense") + resa
        g_in, b_in = sd[L + "output.LayerNorm.weight"], sd[L + "output.LayerNorm.bias"]
        xp = pair(v)
        mu, r = stats_from_slices(v, eps)
    x = ln_rows(unpair(xp), mu, r, g_in, b_in)
    first = np.concatenate([[0], np.cumsum(lens)[:-1]])
    cls = x[torch.as_tensor(first)]
    z = F.linear(cls, sd["embeddingHead.weight"], sd["embeddingHead.bias"])
    return F.layer_norm(z, (768,), sd["norm.weight"], sd["norm.bias"], 1e-5)


def _case(seed_w=5, offset=0.0):
    n_layers = 12
    sd = encoder_ref.det_state_dict(seed=seed_w, n_layers=n_layers, ln_jitter=0.1)
    if offset:
        # a common offset on every pre-LayerNorm row (embeddings and the dense biases of both residual branches): |mean| >> std
        sd = dict(sd)
        sd["roberta.embeddings.word_embeddings.weight"] = sd["roberta.embeddings.word_embeddings.weight"] + offset
        for i in range(n_layers):
            for n in ("attention.output.dense.bias", "output.dense.bias"):
                k = "roberta.encoder.layer.%d.%s" % (i, n)
                sd[k] = sd[k] + offset
    rng = np.random.default_rng(8)
    lens = np.array([1, 2, 31, 33, 64, 65, 96, 128, 70, 9, 100, 50], dtype=np.int64)
    ids = torch.from_numpy(synth.make_records(rng, len(lens), 128, lens))
    return sd, ids, lens


def test_split_scheme_is_fp32_grade():
    sd, ids, lens = _case()
    with torch.no_grad():
        sd64 = {k: v.double() for k, v in sd.items()}
        want64 = encoder_ref.rdot_nll_ln_emb(sd64, ids, encoder_ref.mask_from_lengths(lens, 128), n_layers=12)
        want32 = encoder_ref.rdot_nll_ln_emb(sd, ids, encoder_ref.mask_from_lengths(lens, 128), n_layers=12)
        got = run_split(sd, ids, lens, 12)
    e_split = float((got.double() - want64).abs().max())
    e_fp32 = float((want32.double() - want64).abs().max())
    print("max |delta| vs the fp64 oracle: split %.3e   plain fp32 %.3e" % (e_split, e_fp32))
    assert e_split <= 2e-5                      # the mode's stated tolerance
    assert e_split <= 4.0 * e_fp32 + 2e-6       # ... and it really is of the order of fp32 summation noise
    with torch.no_grad():
        e_unscaled = float((run_split(sd, ids, lens, 12, scaled=False).double() - want64).abs().max())
    print("without the per-matrix weight scale: %.3e" % e_unscaled)
    assert e_unscaled > 1.5 * e_split           # the weight scale is what keeps the 0.02-weights' lo halves at 11 bits


def test_split_scheme_with_large_row_means():
    """|mean| >> std on every pre-LayerNorm row (the single-fp16 fold loses 20-80x there): the pair carries 22
    bits of v itself, so the fold's  r (acc - mu c)  cancellation costs fp32 accumulation noise times |mu| r, nothing more."""
    sd, ids, lens = _case(offset=5.0)
    with torch.no_grad():
        sd64 = {k: v.double() for k, v in sd.items()}
        want64 = encoder_ref.rdot_nll_ln_emb(sd64, ids, encoder_ref.mask_from_lengths(lens, 128), n_layers=12)
        got = run_split(sd, ids, lens, 12)
    e = float((got.double() - want64).abs().max())
    print("offset +5: split %.3e" % e)
    assert e <= 1e-4
