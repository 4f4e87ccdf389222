"""Oracle self-checks for the search restatement (CPU): the C fmaf-chain oracle against exact
arithmetic, the canonical order, merge, and the edge cases the domain has (ties, n < k, k = 1)."""
import numpy as np
import pytest

from oracle import search_ref as sr
from oracle import synth


def test_chain_scores_exact_on_dyadic_grid():
    rng = np.random.default_rng(0)
    x = synth.dyadic_rows(rng, 700)
    q = synth.dyadic_rows(rng, 9)
    S = sr.ip_scores_chain(x, q)
    exact = q.astype(np.float64) @ x.T.astype(np.float64)
    assert np.array_equal(S.astype(np.float64), exact)


def test_chain_is_a_k_ascending_fmaf_chain():
    # one pair, computed step by step in float64 with a single rounding per step == fmaf in fp32
    rng = np.random.default_rng(1)
    x = synth.ln_rows(rng, 3)
    q = synth.ln_rows(rng, 2)
    S = sr.ip_scores_chain(x, q)
    for qi in range(2):
        for j in range(3):
            s = np.float32(0.0)
            for k in range(768):
                s = np.float32(np.float64(q[qi, k]) * np.float64(x[j, k]) + np.float64(s))
            assert S[qi, j] == s


def test_canonical_order_and_ties():
    rng = np.random.default_rng(2)
    x = synth.dyadic_rows(rng, 500)
    x[400] = x[3]
    x[77] = x[3]
    q = x[3:4].copy()
    D, I = sr.flat_ip_topk_chain(x, q, 10)
    # the three identical rows tie at the top and come out in ascending id order
    assert I[0, :3].tolist() == [3, 77, 400]
    assert D[0, 0] == D[0, 1] == D[0, 2]
    assert np.all(np.diff(D[0]) <= 0)


def test_blas_and_chain_agree_on_exact_data():
    rng = np.random.default_rng(3)
    x = synth.dyadic_rows(rng, 3000)
    x[5] = x[1000]
    q = synth.dyadic_rows(rng, 17)
    D1, I1 = sr.flat_ip_topk_chain(x, q, 100)
    D2, I2 = sr.flat_ip_topk_blas(x, q, 100, x_block=1024)
    assert np.array_equal(I1, I2) and np.array_equal(D1, D2)


def test_fewer_rows_than_k_and_k1():
    rng = np.random.default_rng(4)
    x = synth.ln_rows(rng, 5)
    q = synth.ln_rows(rng, 3)
    D, I = sr.flat_ip_topk_chain(x, q, 8)
    assert np.all(I[:, 5:] == -1) and np.all(D[:, 5:] == sr.NEG_FILL)
    assert np.all(np.sort(I[:, :5], axis=1) == np.arange(5))
    D1, I1 = sr.flat_ip_topk_chain(x, q, 1)
    assert np.array_equal(I1[:, 0], I[:, 0])
    # empty corpus
    D0, I0 = sr.flat_ip_topk_chain(np.zeros((0, 768), np.float32), q, 4)
    assert np.all(I0 == -1)


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_shard_merge_invariance(shards):
    rng = np.random.default_rng(5)
    x = synth.ln_rows(rng, 2000)
    x[1500] = x[10]
    q = synth.ln_rows(rng, 12)
    Dc, Ic = sr.flat_ip_topk_chain(x, q, 64)
    per = (2000 + shards - 1) // shards
    Dp, Ip = [], []
    for s in range(shards):
        d, i = sr.flat_ip_topk_chain(x[s * per:(s + 1) * per], q, 64, row_base=s * per)
        Dp.append(d)
        Ip.append(i)
    Dm, Im = sr.topk_merge(np.stack(Dp), np.stack(Ip), 64)
    assert np.array_equal(Im, Ic) and np.array_equal(Dm, Dc)
