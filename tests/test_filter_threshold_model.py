"""CPU model of the threshold rules of the two-precision search (csrc/ip_topk_fast.hip): the filter keeps a row when its
approximate score is >= t~ - 2 eps, where t~ is the k-th best approximate score over ANY subset of the rows seen so far
(own list, another split's list, the union of two lists in rescore_kernel).  With |s~ - s| <= eps for every row, no row
of the exact top-k may ever be dropped, whatever the order of the rows, the prune points and the subsets are.  The model
replays the kernel's rules on random and adversarial (near-tie, duplicate-score) data with an adversarial error pattern."""
import numpy as np
import pytest


def kth_largest(v, k):
    return np.sort(v)[-k] if len(v) >= k else None


def run_model(s, s_apx, k, eps, S, rng, cap=64):
    """Rows are dealt to S splits; each split streams its rows through a bounded list with prunes at random points and
    takes thresholds from its own list or from a sibling's published one; rescore cuts at the union's k-th."""
    n = len(s)
    split_of = rng.integers(0, S, n)
    lists, thr = [[] for _ in range(S)], [-np.inf] * S
    order = rng.permutation(n)
    for r in order:
        sp = split_of[r]
        t_use = max(thr[sp], thr[rng.integers(0, S)])       # own threshold or a sibling's (thr_g sharing)
        if s_apx[r] >= t_use:
            lists[sp].append(r)
        if len(lists[sp]) >= cap or rng.random() < 0.02:     # capacity prune or a scheduled episode
            kth = kth_largest(s_apx[lists[sp]], k)
            if kth is not None:
                thr[sp] = max(thr[sp], kth - 2 * eps)
                lists[sp] = [x for x in lists[sp] if s_apx[x] >= thr[sp]]
    kept = []
    for sp in range(S):                                     # rescore_kernel: k-th over this list and the next split's
        sib = (sp + 1) % S
        pool = np.array(lists[sp] + (lists[sib] if S > 1 else []), dtype=np.int64)
        band = thr[sp]
        kth = kth_largest(s_apx[pool], k) if len(pool) else None
        if kth is not None:
            band = max(band, kth - 2 * eps)
        kept += [x for x in lists[sp] if s_apx[x] >= band]
    return np.array(kept, dtype=np.int64)


@pytest.mark.parametrize("S", [1, 2, 4])
@pytest.mark.parametrize("case", ["random", "near_ties", "duplicates", "adversarial_error"])
def test_exact_topk_survives_every_threshold(S, case):
    rng = np.random.default_rng(1000 * S + ["random", "near_ties", "duplicates", "adversarial_error"].index(case))
    for trial in range(12):
        n, k, eps = 3000, int(rng.choice([1, 5, 20])), 0.05
        if case == "random":
            s = rng.standard_normal(n)
        elif case == "near_ties":
            s = np.round(rng.standard_normal(n) * 4) / 4 + rng.uniform(-eps, eps, n) * 0.5
        elif case == "duplicates":
            s = np.round(rng.standard_normal(n) * 2) / 2
        else:
            s = rng.standard_normal(n)
        err = rng.uniform(-eps, eps, n)
        if case == "adversarial_error":   # push the true top-k down and everything else up, by the full eps
            top = np.argsort(-s, kind="stable")[:k]
            err = np.full(n, eps)
            err[top] = -eps
        s_apx = s + err
        kept = run_model(s, s_apx, k, eps, S, rng, cap=max(64, 4 * k))
        # exact top-k under (score desc, row asc) -- every one of them must have survived
        top = np.lexsort((np.arange(n), -s))[:k]
        assert set(top.tolist()) <= set(kept.tolist()), (S, case, trial)
        # and re-scoring the survivors exactly reproduces the exact answer
        ks = kept[np.lexsort((kept, -s[kept]))][:k]
        assert np.array_equal(ks, top)


def test_a_smaller_band_is_not_safe():
    """The 2 eps in the rule is tight: with a band of 1.5 eps the adversarial error pattern loses a top-k row."""
    rng = np.random.default_rng(3)
    n, k, eps = 400, 4, 0.05
    s = np.concatenate([np.full(k, 1.0), np.full(n - k, 1.0 - 1e-9)])   # k best rows barely above the rest
    err = np.full(n, eps)
    err[:k] = -eps
    s_apx = s + err
    kth = kth_largest(s_apx, k)
    assert np.all(s_apx[:k] >= kth - 2 * eps)           # the rule keeps them
    assert not np.all(s_apx[:k] >= kth - 1.5 * eps)     # a tighter band would not
