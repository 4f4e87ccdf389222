"""The error slack of the two-precision search (ance_amd/csrc/ip_topk_fast.hip) restated and attacked on the CPU.

The fast search filters the corpus with fp16-operand / fp32-accumulate scores s~ = fp16(q) . fp16(x') of the CENTRED rows
x' = fl32(x - mu) and keeps every row whose s~ is within 2 eps of the k-th best; exactness of the final result needs
|s~ - (C - q . mu)| <= eps for EVERY (query, row), where C is the canonical fp32 fmaf-chain score of (q, x) and q . mu is
the (real-number) constant the centring removes from every score of the query.
eps = rel_c |q| max|x'| + abs_c (|q| + max|x'|) + chain_o |q| max|x| with the three constants below.  This test recomputes
s~ under several accumulation orders (the MFMA's internal order is not specified) on random and adversarial vectors, with
and without a large common component, and checks the bound, and that the constants here are the ones compiled into the kernel."""
import os
import re

import numpy as np
import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ance_amd", "csrc", "ip_topk_fast.hip")


def slack(d):
    u = np.float32(5.9604645e-8)
    rel = np.float32(1.25) * (np.float32(9.765625e-4) + np.float32(1.1) * np.float32(d) * u + np.float32(1.1920929e-7))
    ab = np.float32(1.25) * u * np.float32(np.sqrt(np.float32(d)))
    ch = np.float32(1.25) * np.float32(d) * u
    return float(rel), float(ab), float(ch)


def test_constants_are_the_kernels():
    src = open(SRC).read()
    assert re.search(r"eps\.rel_c = 1\.25f \* \(9\.765625e-4f \+ 1\.1f \* d \* 5\.9604645e-8f \+ 1\.1920929e-7f\);", src)
    assert re.search(r"eps\.abs_c = 1\.25f \* 5\.9604645e-8f \* sqrtf\(\(float\)d\);", src)
    assert re.search(r"eps\.chain_o = 1\.25f \* d \* 5\.9604645e-8f;", src)
    assert re.search(r"2\.0f \* \(E\.rel_c \* qn \* xc \+ E\.abs_c \* \(qn \+ xc\) \+ E\.chain_o \* qn \* xo\)", src)
    assert 9.765625e-4 == 2.0 ** -10 and abs(5.9604645e-8 - 2.0 ** -24) < 1e-15 and abs(1.1920929e-7 - 2.0 ** -23) < 1e-15


def chain(q, x):
    """Canonical score: fp32 fmaf chain, k ascending from +0 (product exact in float64, one rounding per step)."""
    s = np.float32(0.0)
    for a, b in zip(q.astype(np.float64), x.astype(np.float64)):
        s = np.float32(a * b + np.float64(s))
    return float(s)


def approx_scores(q, x):
    """s~ under several fp32 accumulation orders of the fp16-rounded operands."""
    qh = q.astype(np.float16).astype(np.float32)
    xh = x.astype(np.float16).astype(np.float32)
    p = qh * xh  # exact in fp32: 11-bit x 11-bit significands
    out = []
    s = np.float32(0)
    for v in p:
        s = np.float32(s + v)
    out.append(float(s))                                   # sequential
    s = np.float32(0)
    for v in p[::-1]:
        s = np.float32(s + v)
    out.append(float(s))                                   # reversed
    t = p.copy()
    while len(t) > 1:                                      # pairwise tree
        if len(t) % 2:
            t = np.append(t, np.float32(0))
        t = (t[0::2] + t[1::2]).astype(np.float32)
    out.append(float(t[0]))
    blk = p.reshape(-1, 16).sum(axis=1, dtype=np.float32)  # 16-wide blocks (one MFMA k-step), then sequential
    s = np.float32(0)
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


def check(q, x, mu, d, worst):
    rel, ab, ch = slack(d)
    xc = (x - mu).astype(np.float32)  # fl32(x - mu), what the image rounds to fp16
    qn = float(np.linalg.norm(q.astype(np.float64)))
    eps = rel * qn * float(np.linalg.norm(xc.astype(np.float64))) + ab * (qn + float(np.linalg.norm(xc.astype(np.float64)))) \
        + ch * qn * float(np.linalg.norm(x.astype(np.float64)))
    target = chain(q, x) - float(np.dot(q.astype(np.float64), mu.astype(np.float64)))
    for st in approx_scores(q, xc):
        assert abs(st - target) <= eps, (d, st, target, eps)
        worst[0] = max(worst[0], abs(st - target) / eps)


@pytest.mark.parametrize("d", [128, 768, 1024, 2048])
def test_bound_holds(d):
    rng = np.random.default_rng(d)
    worst = [0.0]
    zero = np.zeros(d, np.float32)
    for rep in range(5):
        for q, x in vectors(rng, d):
            check(q, x, zero, d, worst)  # no centring (mu = 0): the bound of round 1 plus the chain term
    assert worst[0] < 0.95  # the slack is not razor-thin on any of these


@pytest.mark.parametrize("d", [128, 768])
def test_bound_holds_with_a_large_common_component(d):
    """rows = c + small deviation (cosine ~0.99 between rows, like the embeddings of one encoder): the centred image makes
    |x'| ~ 10x smaller than |x| and the bound shrinks with it -- and still holds, for exact and for sloppy means."""
    rng = np.random.default_rng(100 + d)
    worst = [0.0]
    c = (rng.standard_normal(d) * 1.0).astype(np.float32)
    c = (c / np.linalg.norm(c) * np.sqrt(d)).astype(np.float32)
    for rep in range(12):
        x = (c + 0.12 * rng.standard_normal(d)).astype(np.float32)
        q = (c + 0.12 * rng.standard_normal(d)).astype(np.float32)
        for mu in (c, (c * np.float32(0.97)).astype(np.float32), (c + 0.05 * rng.standard_normal(d)).astype(np.float32)):
            check(q, x, mu, d, worst)
        check((-q).astype(np.float32), x, c, d, worst)
    assert worst[0] < 0.95


