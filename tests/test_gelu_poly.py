"""The split (fp32-grade) mode's GELU (csrc/gemm256_epilogue.h: gelu_exact) restated in fp32 emulation: the erf form of the
reference (transformers "gelu", model/models.py via RobertaModel) computed as  x (x >= 0 ? 1 - e : e),  e = 2^q(z),
z = min(|x| / sqrt 2, 6.6),  q = a degree-9 fit of log2(erfc(z) / 2)  -- 9 fma and one hardware exp2 instead of ocml's two-branch
erff.  The claim pinned here: it is at least as close to the exact GELU as the reference's OWN fp32 arithmetic (torch's fp32
erf-GELU), so swapping it in does not leave the fp32 grade.  The coefficients below must stay identical to GELU_Q in the header
(checked textually)."""
import os
import re

import numpy as np
import torch
from scipy import special

Q = [-1.0, -1.627907395362854, -0.918441653251648, -0.14831341803073883, 0.02773732878267765, 6.778987153666094e-05,
     -0.002261603018268943, 0.0008423461113125086, -0.00015156660811044276, 1.1468856428109575e-05]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def gelu_poly32(x):
    x = x.astype(np.float32)
    z = np.minimum((np.abs(x) * np.float32(0.70710678118654752440)).astype(np.float32), np.float32(6.6))
    q = np.full_like(x, np.float32(Q[9]))
    for k in range(8, -1, -1):
        q = _fma32(q, z, np.full_like(x, np.float32(Q[k])))
    e = np.exp2(q.astype(np.float64)).astype(np.float32)  # v_exp_f32: 1 ulp
    return (x * np.where(x >= 0, (np.float32(1.0) - e).astype(np.float32), e)).astype(np.float32)


def test_header_carries_these_coefficients():
    src = open(os.path.join(ROOT, "ance_amd", "csrc", "gemm256_epilogue.h")).read()
    body = re.search(r"GELU_Q\[10\] = \{([^}]*)\}", src).group(1)
    got = [float(t.strip().rstrip("f")) for t in body.replace("\n", " ").split(",")]
    assert got == Q


def test_polynomial_gelu_is_fp32_grade():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-10, 10, 3_000_000), rng.normal(0, 1.5, 3_000_000), rng.uniform(-1, 1, 1_000_000) * 1e-3,
                         np.linspace(-9.5, 9.5, 1_000_001), np.array([0.0, -0.0, 9.4, -9.4, 50.0, -50.0, 1e-30, -1e-30])]).astype(np.float32)
    x64 = xs.astype(np.float64)
    want = 0.5 * x64 * special.erfc(-x64 / np.sqrt(2.0))
    got = gelu_poly32(xs).astype(np.float64)
    ref = torch.nn.functional.gelu(torch.from_numpy(xs)).numpy().astype(np.float64)  # the reference's own fp32 arithmetic
    assert np.isfinite(got).all()
    scale = np.maximum(np.abs(x64), 1e-30)
    ea, er = np.abs(got - want), np.abs(ref - want)
    print("max |err| / |x|: polynomial %.3e, torch fp32 %.3e;  mean |err|: %.3e / %.3e" % ((ea / scale).max(), (er / scale).max(), ea.mean(), er.mean()))
    assert (ea / scale).max() <= 1.5e-7          # error in Phi: a little over one fp32 ulp of 0.5 .. 1
    assert (ea / scale).max() <= (er / scale).max() and ea.mean() <= er.mean()
    for lo, hi in ((0, 0.5), (0.5, 1), (1, 2), (2, 3), (3, 5), (5, 10)):
        m = (np.abs(x64) >= lo) & (np.abs(x64) < hi)
        assert (ea / scale)[m].max() <= max(1.5 * (er / scale)[m].max(), 1.2e-7), (lo, hi)
    # the far tails: x itself for large x; for very negative x (exact: -0) |x| 2^q(6.6), where the fp32 Horner sum of terms of
    # magnitude 270 leaves q = -41.5 instead of -66: below 1e-12 |x| either way
    assert gelu_poly32(np.array([50.0], np.float32))[0] == 50.0 and abs(gelu_poly32(np.array([-50.0], np.float32))[0]) < 1e-10
