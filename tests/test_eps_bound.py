"""The error slack of the two-precision search (ance_amd/csrc/ip_topk_fast.hip) restated and attacked on the CPU.

The fast search filters the corpus with  s~ = b + fp16(dq) . fp16(x')  (fp32 accumulation starting from b), where
x' = fl32(x - mu) is the row centred on the shard mean, dq = fl32(q - mq) the query centred on the mean query of the call
and b = fl32(mq . x') that mean query's share of the row's score; it keeps every row whose s~ is within 2 eps of the k-th
best.  Exactness of the final result needs |s~ - (C - q . mu)| <= eps for EVERY (query, row), where C is the canonical fp32
fmaf-chain score of (q, x) and q . mu the (real-number) constant the centring removes from every score of the query.
eps = rel_c |dq| X' + acc_m |mq| X' + cen |q| X' + abs_c (|dq| + X') + chain_o |q| X   (X' = max |x'|, X = max |x|).
This test recomputes s~ under several accumulation orders (the MFMA's internal order is not specified) on random and
adversarial vectors, with and without large common components, and checks the bound, and that the constants here are the
ones compiled into the kernel."""
import os
import re

import numpy as np
import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ance_amd", "csrc", "ip_topk_fast.hip")


def slack(d):
    u = np.float32(5.9604645e-8)
    f = np.float32(1.25)
    return dict(rel_c=float(f * (np.float32(9.765625e-4) + np.float32(1.1) * np.float32(d) * u)),
                acc_m=float(f * np.float32(2.1) * np.float32(d) * u), cen=float(f * np.float32(1.1920929e-7)),
                abs_c=float(f * u * np.float32(np.sqrt(np.float32(d)))), chain_o=float(f * np.float32(d) * u))


def test_constants_are_the_kernels():
    src = open(SRC).read()
    assert re.search(r"eps\.rel_c = 1\.25f \* \(9\.765625e-4f \+ 1\.1f \* d \* 5\.9604645e-8f\);", src)
    assert re.search(r"eps\.acc_m = 1\.25f \* 2\.1f \* d \* 5\.9604645e-8f;", src)
    assert re.search(r"eps\.cen = 1\.25f \* 1\.1920929e-7f;", src)
    assert re.search(r"eps\.abs_c = 1\.25f \* 5\.9604645e-8f \* sqrtf\(\(float\)d\);", src)
    assert re.search(r"eps\.chain_o = 1\.25f \* d \* 5\.9604645e-8f;", src)
    assert re.search(r"2\.0f \* \(E\.rel_c \* qc \* xc \+ E\.acc_m \* qs->mq_norm \* xc \+ E\.cen \* qo \* xc \+ E\.abs_c \* \(qc \+ xc\) \+ "
                     r"E\.chain_o \* qo \* xo\)", src)
    assert 9.765625e-4 == 2.0 ** -10 and abs(5.9604645e-8 - 2.0 ** -24) < 1e-15 and abs(1.1920929e-7 - 2.0 ** -23) < 1e-15


def chain(q, x):
    """Canonical score: fp32 fmaf chain, k ascending from +0 (product exact in float64, one rounding per step)."""
    s = np.float32(0.0)
    for a, b in zip(q.astype(np.float64), x.astype(np.float64)):
        s = np.float32(a * b + np.float64(s))
    return float(s)


def approx_scores(q, x, b0=0.0):
    """s~ = b0 + sum of the fp16-rounded operands' products, under several fp32 accumulation orders."""
    qh = q.astype(np.float16).astype(np.float32)
    xh = x.astype(np.float16).astype(np.float32)
    p = qh * xh  # exact in fp32: 11-bit x 11-bit significands
    b0 = np.float32(b0)
    out = []
    s = b0
    for v in p:
        s = np.float32(s + v)
    out.append(float(s))                                   # sequential, bias first (what the accumulator init does)
    s = np.float32(0)
    for v in p[::-1]:
        s = np.float32(s + v)
    out.append(float(np.float32(s + b0)))                  # reversed, bias last
    t = p.copy()
    while len(t) > 1:                                      # pairwise tree
        if len(t) % 2:
            t = np.append(t, np.float32(0))
        t = (t[0::2] + t[1::2]).astype(np.float32)
    out.append(float(np.float32(t[0] + b0)))
    blk = p.reshape(-1, 16).sum(axis=1, dtype=np.float32)  # 16-wide blocks (one MFMA k-step), then sequential from the bias
    s = b0
    for v in blk:
        s = np.float32(s + v)
    out.append(float(s))
    return out


def vectors(rng, d):
    ln = lambda v: ((v - v.mean()) / v.std()).astype(np.float32)
    yield ln(rng.standard_normal(d)), ln(rng.standard_normal(d))                       # the workload's distribution
    yield np.abs(ln(rng.standard_normal(d))), np.abs(ln(rng.standard_normal(d)))       # no cancellation: max accumulation error
    a = ln(rng.standard_normal(d))
    yield a, (-a + 1e-3 * rng.standard_normal(d)).astype(np.float32)                   # heavy cancellation around -|a|^2
    yield (rng.standard_normal(d) * 1e-6).astype(np.float32), ln(rng.standard_normal(d))  # fp16-subnormal operand
    yield (rng.standard_normal(d) * 200).astype(np.float32), (rng.standard_normal(d) * 200).astype(np.float32)  # large, < 65504
    yield (10.0 ** rng.uniform(-7, 2, d) * rng.choice([-1, 1], d)).astype(np.float32), ln(rng.standard_normal(d))  # mixed decades
    u = np.float32(1.0 + 2.0 ** -11)  # every element exactly on an fp16 rounding tie
    yield np.full(d, u, np.float32), np.full(d, u, np.float32)


def check(q, x, mu, mq, d, worst):
    E = slack(d)
    xc = (x - mu).astype(np.float32)   # fl32(x - mu), what the image rounds to fp16
    dq = (q - mq).astype(np.float32)   # fl32(q - mq), what the query chunk rounds to fp16
    n64 = lambda v: float(np.linalg.norm(v.astype(np.float64)))
    eps = E["rel_c"] * n64(dq) * n64(xc) + E["acc_m"] * n64(mq) * n64(xc) + E["cen"] * n64(q) * n64(xc) \
        + E["abs_c"] * (n64(dq) + n64(xc)) + E["chain_o"] * n64(q) * n64(x)
    target = chain(q, x) - float(np.dot(q.astype(np.float64), mu.astype(np.float64)))
    biases = [chain(mq, xc), float(np.float32(np.dot(mq.astype(np.float64), xc.astype(np.float64))))]  # fp32 chain / correctly rounded
    for b0 in biases:
        for st in approx_scores(dq, xc, b0):
            assert abs(st - target) <= eps, (d, st, target, eps)
            worst[0] = max(worst[0], abs(st - target) / eps)


@pytest.mark.parametrize("d", [128, 768, 1024, 2048])
def test_bound_holds(d):
    rng = np.random.default_rng(d)
    worst = [0.0]
    zero = np.zeros(d, np.float32)
    for rep in range(5):
        for q, x in vectors(rng, d):
            check(q, x, zero, zero, d, worst)  # no centring at all (mu = mq = 0)
    assert worst[0] < 0.95  # the slack is not razor-thin on any of these


@pytest.mark.parametrize("d", [128, 768])
def test_bound_holds_with_large_common_components(d):
    """rows = c + small deviation, queries = c' + small deviation (cosine ~0.99 inside each set, like the embeddings of one
    encoder / of DPR's two towers): the centred operands are ~10x smaller than the vectors and the bound shrinks with them
    -- and still holds, for exact and for sloppy means, with and without the query mean."""
    rng = np.random.default_rng(100 + d)
    worst = [0.0]
    unit = lambda v: (v / np.linalg.norm(v) * np.sqrt(d)).astype(np.float32)
    c = unit(rng.standard_normal(d))
    c2 = unit(c + 0.5 * unit(rng.standard_normal(d)))
    zero = np.zeros(d, np.float32)
    for rep in range(10):
        x = (c + 0.12 * rng.standard_normal(d)).astype(np.float32)
        for cq in (c, c2):
            q = (cq + 0.12 * rng.standard_normal(d)).astype(np.float32)
            for mu in (c, (c * np.float32(0.97)).astype(np.float32), (c + 0.05 * rng.standard_normal(d)).astype(np.float32)):
                for mq in (zero, cq, (cq + 0.05 * rng.standard_normal(d)).astype(np.float32)):
                    check(q, x, mu, mq, d, worst)
            check((-q).astype(np.float32), x, c, (-cq).astype(np.float32), d, worst)
    assert worst[0] < 0.95
