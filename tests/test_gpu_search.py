"""Parity of the HIP exact inner-product top-k (through the C ABI) with the oracle -- bit-exact
scores AND ids, on the edge cases the domain has, plus size-independent properties at sizes the
oracle cannot reach.  Needs an MI355X."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _diag(name, **kw):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "diag_%s.json" % name), "w") as f:
        json.dump({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in kw.items()}, f)


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment():
    """The library reads its ANCE_* knobs once per process; the tests below change them with monkeypatch.  This fixture is
    set up before monkeypatch, so its teardown runs after the environment has been restored."""
    yield
    from ance_amd import _lib
    _lib.reload_env()


def _search(x, q, k, row_base=0):
    from ance_amd import _lib
    from ance_amd.index import FlatIPIndex
    _lib.reload_env()
    idx = FlatIPIndex(x.shape[1], row_base=row_base)
    idx.add(x)
    return idx.search(q, k)


def _check_exact(name, x, q, k, row_base=0):
    from oracle import search_ref
    D, I = _search(x, q, k, row_base)
    Do, Io = search_ref.flat_ip_topk_chain(x, q, k, row_base=row_base)
    ok_i, ok_d = np.array_equal(I, Io), np.array_equal(D, Do)
    if not (ok_i and ok_d):
        bad = np.argwhere((I != Io) | (D != Do))
        r = int(bad[0][0])
        # which k-order does the hardware use?  compare with the reversed-pair chain too
        _diag(name, n=x.shape[0], nq=q.shape[0], k=k, n_bad=int(len(bad)), first_bad=bad[0], I=I[r], Io=Io[r],
              D=D[r].astype(float), Do=Do[r].astype(float))
    assert ok_i, "%s: ids differ from the oracle (see gpurun_out/diag_%s.json)" % (name, name)
    assert ok_d, "%s: scores differ bitwise from the fmaf-chain oracle" % name
    return D, I


@pytest.mark.parametrize("k", [1, 10, 100, 200])
def test_bit_exact_ln_rows(k):
    from oracle import synth
    rng = np.random.default_rng(10 + k)
    x = synth.ln_rows(rng, 5000)
    q = synth.ln_rows(rng, 200)
    _check_exact("ln_k%d" % k, x, q, k)


def test_bit_exact_with_duplicates_and_exact_arithmetic():
    from oracle import synth
    rng = np.random.default_rng(1)
    x = synth.dyadic_rows(rng, 6000)
    dup = rng.integers(0, 6000, size=120)
    x[rng.integers(0, 6000, size=120)] = x[dup]
    x[100:164] = x[7]  # a run of 64 identical rows: massive exact tie
    q = np.concatenate([synth.dyadic_rows(rng, 60), x[7:8], x[dup[:3]]])
    D, I = _check_exact("dyadic_dups", x, q, 200)
    # canonical order inside ties: ascending ids
    for r in range(I.shape[0]):
        same = D[r, 1:] == D[r, :-1]
        assert np.all(I[r, 1:][same] > I[r, :-1][same])


@pytest.mark.parametrize("n,nq,k,d", [(50, 3, 200, 768), (1, 1, 1, 768), (129, 129, 64, 768), (4097, 257, 200, 768),
                                      (3000, 40, 100, 100), (2000, 33, 50, 6), (777, 5, 300, 768)])
def test_ragged_shapes(n, nq, k, d):
    from oracle import synth
    rng = np.random.default_rng(n + nq)
    x = synth.ln_rows(rng, n, d=max(d, 8))[:, :d].copy()
    q = synth.ln_rows(rng, nq, d=max(d, 8))[:, :d].copy()
    D, I = _check_exact("ragged_%d_%d_%d_%d" % (n, nq, k, d), x, q, k)
    if n < k:
        assert np.all(I[:, n:] == -1)


def test_empty_corpus_and_no_queries():
    from ance_amd.index import FlatIPIndex
    idx = FlatIPIndex(768)
    q = np.zeros((3, 768), np.float32)
    D, I = idx.search(q, 5)
    assert np.all(I == -1) and np.all(D == np.float32(-3.4028234663852886e38))
    idx.add(np.ones((4, 768), np.float32))
    D, I = idx.search(np.zeros((0, 768), np.float32), 5)
    assert D.shape == (0, 5) and I.shape == (0, 5)


@pytest.mark.parametrize("k", [500, 1000])
def test_large_k(k):
    from oracle import synth
    rng = np.random.default_rng(k)
    x = synth.ln_rows(rng, 20000)
    q = synth.ln_rows(rng, 40)
    _check_exact("largek%d" % k, x, q, k)


def test_many_prunes_and_splits():
    """200k rows: every query's candidate buffer is pruned many times, several corpus splits."""
    from oracle import synth
    rng = np.random.default_rng(77)
    x = synth.ln_rows(rng, 200000)
    x[150000:150300] = x[5]
    q = np.concatenate([synth.ln_rows(rng, 150), x[5:6]])
    _check_exact("prunes", x, q, 200)


def test_ascending_scores_worst_case_for_filter():
    """Adversarial order: scores increase with the row id, so EVERY row passes the running
    threshold and the buffers overflow as fast as they can."""
    rng = np.random.default_rng(5)
    base = rng.standard_normal(768).astype(np.float32)
    n = 30000
    scale = (np.arange(n, dtype=np.float32) + 1) / np.float32(n)
    x = (base[None, :] * scale[:, None]).astype(np.float32)
    q = np.stack([base, -base, base * 0.5]).astype(np.float32)
    _check_exact("ascending", x, q, 200)


def test_shard_invariance_and_merge():
    import torch
    from ance_amd.index import topk_merge_device
    from oracle import search_ref, synth
    rng = np.random.default_rng(9)
    x = synth.ln_rows(rng, 30000)
    x[25000] = x[10]
    x[12345] = x[10]
    q = np.concatenate([synth.ln_rows(rng, 100), x[10:11]])
    D1, I1 = _search(x, q, 200)
    for shards in (2, 3, 8):
        per = (30000 + shards - 1) // shards
        Dp, Ip = [], []
        for s in range(shards):
            d, i = _search(x[s * per:(s + 1) * per], q, 200, row_base=s * per)
            Dp.append(d)
            Ip.append(i)
        Dm, Im = topk_merge_device(torch.from_numpy(np.stack(Dp)).cuda(), torch.from_numpy(np.stack(Ip)).cuda())
        assert np.array_equal(Im.cpu().numpy(), I1) and np.array_equal(Dm.cpu().numpy(), D1), shards
        # and the merge kernel agrees with the oracle merge
        Dmo, Imo = search_ref.topk_merge(np.stack(Dp), np.stack(Ip), 200)
        assert np.array_equal(Imo, I1) and np.array_equal(Dmo, D1)


def test_properties_at_scale():
    """2M x 768 corpus, 2048 queries, k = 200 (one eighth of the headline workload, the size that
    still leaves the box memory to spare): sortedness, uniqueness, exactness of the reported scores,
    and no missed row among a random sample."""
    import torch
    from ance_amd.index import FlatIPIndex
    from oracle import search_ref
    g = torch.Generator(device="cuda").manual_seed(3)
    n, nq, k = 2_000_000, 2048, 200
    x = torch.randn((n, 768), generator=g, device="cuda")
    x = torch.nn.functional.layer_norm(x, (768,))
    q = torch.nn.functional.layer_norm(torch.randn((nq, 768), generator=g, device="cuda"), (768,))
    x[1_500_000] = x[17]
    idx = FlatIPIndex(768)
    idx.add(x)
    D, I = idx.search(q, k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    assert np.all(I >= 0) and np.all(I < n)
    assert np.all(D[:, 1:] <= D[:, :-1])
    ties = D[:, 1:] == D[:, :-1]
    assert np.all(I[:, 1:][ties] > I[:, :-1][ties])
    for r in range(0, nq, 97):
        assert len(set(I[r].tolist())) == k
    # reported scores are the exact chain scores of the reported rows
    qs = q[::256].cpu().numpy()
    for j, r in enumerate(range(0, nq, 256)):
        rows = x[torch.from_numpy(I[r]).cuda()].cpu().numpy()
        S = search_ref.ip_scores_chain(rows, qs[j:j + 1])[0]
        assert np.array_equal(S, D[r])
    # no sampled row beats the k-th result
    samp = torch.randint(0, n, (50000,), generator=torch.Generator().manual_seed(1))
    xs = x[samp.cuda()].cpu().numpy()
    S = search_ref.ip_scores_chain(xs, qs)
    for j, r in enumerate(range(0, nq, 256)):
        kth_s, kth_i = D[r, -1], I[r, -1]
        better = (S[j] > kth_s) | ((S[j] == kth_s) & (samp.numpy() < kth_i))
        assert set(samp.numpy()[better].tolist()) <= set(I[r].tolist())


def test_exact_at_the_headline_size():
    """8,841,823 x 768 rows -- the size every queries/s number is quoted on -- against the independent fp32 scan
    (``ance_ip_topk_scan``, pinned bit-exactly to oracle/ip_topk_ref.c by the tests above at sizes the oracle reaches): 17 k
    corpus tiles, 35 windows, the full prune schedule.  2,048 queries through the two-precision path, 64 of them through the
    scan: ids AND scores bit-identical; 4 of those re-scored on the host by the oracle's fmaf chain.  LayerNorm rows, then
    encoder-like rows (the query-mean bias build).  Needs 60 GB of free HBM."""
    import torch
    from ance_amd.index import FlatIPIndex
    from oracle import search_ref
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("needs 60 GB of free device memory, %.0f GB free" % (free / 1e9))
    n, nq, k = 8_841_823, 2048, 200
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.empty((n, 768), device="cuda")
    for kind in ("layernorm", "encoder_like"):
        if kind == "layernorm":
            for b0 in range(0, n, 1 << 20):
                b1 = min(b0 + (1 << 20), n)
                x[b0:b1] = torch.nn.functional.layer_norm(torch.randn((b1 - b0, 768), generator=g, device="cuda"), (768,))
            q = torch.nn.functional.layer_norm(torch.randn((nq, 768), generator=g, device="cuda"), (768,))
        else:
            c = torch.randn((768,), generator=g, device="cuda")
            c = c / c.norm() * (768.0 ** 0.5)
            for b0 in range(0, n, 1 << 20):
                b1 = min(b0 + (1 << 20), n)
                x[b0:b1] = c[None, :] + 0.12 * torch.randn((b1 - b0, 768), generator=g, device="cuda")
            q = c[None, :] + 0.12 * torch.randn((nq, 768), generator=g, device="cuda")
        x[8_000_000] = x[17]  # one exact duplicate far away: a tie across windows
        idx = FlatIPIndex(768)
        idx.add(x)
        D, I = idx.search_device(q, k)
        sel = torch.arange(0, nq, 32, device="cuda")
        Ds, Is = idx.search_device(q[sel].contiguous(), k, exact_scan=True)
        assert torch.equal(Is, I[sel]), kind
        assert torch.equal(Ds.view(torch.int32), D[sel].view(torch.int32)), kind
        for j in sel[:4].tolist():
            S = search_ref.ip_scores_chain(x[I[j]].cpu().numpy(), q[j:j + 1].cpu().numpy())[0]
            assert np.array_equal(S, D[j].cpu().numpy()), (kind, j)
        del idx, D, I, Ds, Is
        torch.cuda.empty_cache()


def test_clustered_scores_every_row_is_a_candidate():
    """Rows differ by ~1e-3 noise around one vector: all scores lie far inside the fp16 filter's slack,
    so the two-precision path must re-score everything exactly -- and still return the exact lists."""
    from oracle import synth
    rng = np.random.default_rng(21)
    c = synth.ln_rows(rng, 1)[0]
    n = 8192
    x = (c[None, :] + 1e-3 * rng.standard_normal((n, 768))).astype(np.float32)
    x[4000] = x[17]
    q = np.concatenate([synth.ln_rows(rng, 40), c[None, :], -c[None, :]]).astype(np.float32)
    _check_exact("clustered", x, q, 200)


def test_tiny_and_mixed_magnitudes():
    """fp16-subnormal-sized corpus rows, and a corpus mixing magnitudes over six decades."""
    from oracle import synth
    rng = np.random.default_rng(22)
    x = synth.ln_rows(rng, 6000) * np.float32(1e-5)
    q = synth.ln_rows(rng, 48)
    _check_exact("tiny", x.astype(np.float32), q, 100)
    scale = (10.0 ** rng.uniform(-4, 2, size=(6000, 1))).astype(np.float32)
    _check_exact("mixed", (synth.ln_rows(rng, 6000) * scale).astype(np.float32), q, 200)


def test_fast_path_large_query_block():
    """20,000 queries x 60,000 rows: several query chunks / tiles per launch of the fast path."""
    from oracle import search_ref, synth
    rng = np.random.default_rng(23)
    x = synth.ln_rows(rng, 60000)
    q = synth.ln_rows(rng, 20000)
    D, I = _search(x, q, 100)
    pick = rng.integers(0, 20000, 300)
    Do, Io = search_ref.flat_ip_topk_chain(x, q[pick], 100)
    assert np.array_equal(I[pick], Io) and np.array_equal(D[pick], Do)
    assert np.all(D[:, 1:] <= D[:, :-1])


def test_fp16_overflow_is_not_trusted():
    """Values beyond the fp16 range (65504) break the two-precision bound; the launch must notice
    (row / query norms) and still return the exact result."""
    from oracle import synth
    rng = np.random.default_rng(24)
    x = synth.ln_rows(rng, 6000)
    q = synth.ln_rows(rng, 40)
    xb = x.copy()
    xb[1234, 5] = 1.0e5      # one corpus element overflows fp16
    xb[4321, 700] = -2.0e5
    _check_exact("ovf_corpus", xb, q, 100)
    qb = q.copy()
    qb[3, 17] = 9.0e4        # one query element overflows fp16
    _check_exact("ovf_query", x, qb, 100)


def test_sparse_candidates_after_threshold():
    """Many corpus tiles per query block with a planted cluster of very high scorers late in the scan:
    exercises the max-prefilter (groups with no candidate are skipped) together with late insertions."""
    from oracle import synth
    rng = np.random.default_rng(25)
    x = synth.ln_rows(rng, 40000)
    q = synth.ln_rows(rng, 300)
    x[39000:39050] = q[:50] * np.float32(1.5)   # late, far above every threshold, each matching one query
    x[20000:20010] = q[100:110] * np.float32(0.9)
    _check_exact("sparse_late", x, q, 50)


# ---- round 2: windows, threshold exchange, search image reuse, duplicate classes, per-query redo ----------


@pytest.mark.parametrize("splits,window", [("2", "8"), ("4", "16"), ("8", "8"), ("2", "0")])
def test_windows_and_split_counts(monkeypatch, splits, window):
    """60,000 rows = 235 corpus tiles scanned in windows of 8 / 16 tiles with 2, 4 and 8 splits per query tile
    (thresholds exchanged between the splits at every window boundary and prune), and without windows."""
    from oracle import synth
    monkeypatch.setenv("ANCE_FAST_SPLITS", splits)
    monkeypatch.setenv("ANCE_FAST_WINDOW_TILES", window)
    rng = np.random.default_rng(31)
    x = synth.ln_rows(rng, 60000)
    x[59000:59040] = x[3]
    q = np.concatenate([synth.ln_rows(rng, 299), x[3:4]])
    _check_exact("win_%s_%s" % (splits, window), x, q, 200)


def test_no_threshold_exchange_same_bits(monkeypatch):
    from oracle import synth
    rng = np.random.default_rng(32)
    x = synth.ln_rows(rng, 30000)
    q = synth.ln_rows(rng, 64)
    D1, I1 = _search(x, q, 100)
    monkeypatch.setenv("ANCE_FAST_SHARE", "0")
    monkeypatch.setenv("ANCE_FAST_WINDOW_WAIT_US", "0")
    D2, I2 = _search(x, q, 100)
    assert np.array_equal(I1, I2) and np.array_equal(D1, D2)


@pytest.mark.parametrize("d,k", [(2048, 50), (1152, 100), (4096, 20), (128, 200)])
def test_wide_rows(d, k):
    """d = 2048 is the widest row of the two-precision path (block-end re-scoring stages the fp32 query row in
    LDS), d = 1152 is not a multiple of 128 and d = 4096 is too wide: both take the fp32 scan.  Same bits."""
    rng = np.random.default_rng(d)
    x = rng.standard_normal((6000, d)).astype(np.float32)
    q = rng.standard_normal((70, d)).astype(np.float32)
    _check_exact("wide_%d" % d, x, q, k)


def test_k_1000_on_the_fast_path():
    from oracle import synth
    rng = np.random.default_rng(33)
    x = synth.ln_rows(rng, 50000)
    q = synth.ln_rows(rng, 48)
    _check_exact("k1000", x, q, 1000)
    _check_exact("k1024", x[:9000], q[:8], 1024)
    _check_exact("k1025_scan", x[:9000], q[:8], 1025)


def test_nan_row_is_not_trusted():
    """A NaN in one corpus row poisons the norm bound of the whole shard: the launch must notice and every query
    takes the exact scan, where (as in the oracle) a NaN score never enters a list."""
    from oracle import synth
    rng = np.random.default_rng(34)
    x = synth.ln_rows(rng, 6000)
    q = synth.ln_rows(rng, 20)
    xb = x.copy()
    xb[777, 3] = np.nan
    D, I = _check_exact("nan_row", xb, q, 50)
    assert not np.any(I == 777)


def test_search_image_is_reused_and_invalidated():
    import torch
    from ance_amd.index import FlatIPIndex
    from oracle import search_ref, synth
    rng = np.random.default_rng(35)
    x = synth.ln_rows(rng, 12000)
    q = synth.ln_rows(rng, 33)
    idx = FlatIPIndex(768)
    idx.add(x[:8000])
    D1, I1 = idx.search(q, 100)
    img = idx._image
    assert isinstance(img, torch.Tensor)
    D2, I2 = idx.search(q[:5], 10)
    assert idx._image is img  # second search: no rebuild
    Do, Io = search_ref.flat_ip_topk_chain(x[:8000], q, 100)
    assert np.array_equal(I1, Io) and np.array_equal(D1, Do)
    assert np.array_equal(I2, Io[:5, :10])
    idx.add(x[8000:])
    assert idx._image is None
    D3, I3 = idx.search(q, 100)
    Do, Io = search_ref.flat_ip_topk_chain(x, q, 100)
    assert np.array_equal(I3, Io) and np.array_equal(D3, Do)


def _dup_corpus(rng, n, frac0, n1):
    """LayerNorm rows with one heavy class of identical rows (fraction frac0 of the shard, scattered) and a second
    class of n1 rows."""
    from oracle import synth
    x = synth.ln_rows(rng, n)
    v0, v1 = synth.ln_rows(rng, 2)
    m0 = rng.random(n) < frac0
    m0[:5] = False       # the class does not start at row 0
    x[m0] = v0
    i1 = rng.choice(np.flatnonzero(~m0), size=n1, replace=False)
    x[i1] = v1
    return x, v0, v1, m0, i1


def test_duplicate_classes_are_expanded_in_id_order():
    """Half the shard is one vector v0 (the all-pad MaxP chunk of model/models.py:165-199), 3 % another: the image keeps one
    row per class.  Queries: random ones, v0 itself (the whole top-k is the class, ascending ids), a mix where the class sits
    mid-list, -v0 (class at the bottom: never returned)."""
    from oracle import synth
    rng = np.random.default_rng(36)
    n = 40000
    x, v0, v1, m0, i1 = _dup_corpus(rng, n, 0.5, 1200)
    q = np.concatenate([synth.ln_rows(rng, 60), v0[None], v1[None], (0.6 * v0 + 0.8 * synth.ln_rows(rng, 1)[0])[None],
                        -v0[None], (v0 + v1)[None]]).astype(np.float32)
    for k in (10, 200, 1000):
        D, I = _check_exact("dups_k%d" % k, x, q, k)
    # v0 as the query: all k results are class members, ascending
    assert np.all(m0[I[60]]) and np.all(np.diff(I[60]) > 0)


def test_duplicate_classes_with_shards_and_row_base():
    import torch
    from ance_amd.index import topk_merge_device
    from oracle import search_ref, synth
    rng = np.random.default_rng(37)
    n = 24000
    x, v0, v1, m0, i1 = _dup_corpus(rng, n, 0.4, 600)
    q = np.concatenate([synth.ln_rows(rng, 30), v0[None], (0.7 * v0 + 0.7 * v1)[None]]).astype(np.float32)
    Do, Io = search_ref.flat_ip_topk_chain(x, q, 200)
    per = n // 3
    Dp, Ip = [], []
    for s in range(3):
        d_, i_ = _search(x[s * per:(s + 1) * per], q, 200, row_base=s * per)
        Dp.append(d_)
        Ip.append(i_)
    Dm, Im = topk_merge_device(torch.from_numpy(np.stack(Dp)).cuda(), torch.from_numpy(np.stack(Ip)).cuda())
    assert np.array_equal(Im.cpu().numpy(), Io) and np.array_equal(Dm.cpu().numpy(), Do)


def test_dedup_off_same_bits(monkeypatch):
    from oracle import synth
    rng = np.random.default_rng(38)
    x, v0, v1, m0, i1 = _dup_corpus(rng, 20000, 0.3, 300)
    q = np.concatenate([synth.ln_rows(rng, 20), v0[None]]).astype(np.float32)
    D1, I1 = _search(x, q, 100)
    monkeypatch.setenv("ANCE_FAST_DEDUP", "0")   # the class then overflows the band of the v0 query: per-query redo
    D2, I2 = _search(x, q, 100)
    assert np.array_equal(I1, I2) and np.array_equal(D1, D2)


def test_only_the_overflowing_queries_are_redone():
    """3 of 400 queries see 6,000 rows inside one error band (rows = c + 1e-3 noise, too few to be sampled as a class);
    the other 397 must come out of the fast path untouched and everything must match the oracle."""
    from oracle import synth
    rng = np.random.default_rng(39)
    x = synth.ln_rows(rng, 50000)
    c = synth.ln_rows(rng, 1)[0]
    rows = rng.choice(50000, size=6000, replace=False)
    x[rows] = (c[None, :] + 1e-3 * rng.standard_normal((6000, 768))).astype(np.float32)
    q = synth.ln_rows(rng, 400)
    q[[7, 200, 399]] = c * np.array([[1.0], [0.9], [1.1]], np.float32)
    _check_exact("ovf_queries", x, q, 200)


def test_more_overflowing_queries_than_the_redo_list():
    """1,500 queries overflow (> 1,024): the whole launch chunk is redone by the scan."""
    from oracle import search_ref, synth
    rng = np.random.default_rng(40)
    c = synth.ln_rows(rng, 1)[0]
    x = (c[None, :] + 1e-3 * rng.standard_normal((8192, 768))).astype(np.float32)
    q = (c[None, :] * (1.0 + 0.1 * rng.random((1500, 1)))).astype(np.float32)
    D, I = _search(x, q, 100)
    pick = rng.integers(0, 1500, 40)
    Do, Io = search_ref.flat_ip_topk_chain(x, q[pick], 100)
    assert np.array_equal(I[pick], Io) and np.array_equal(D[pick], Do)


def test_dedup_at_scale_is_not_slower():
    """2 M rows of which 1 M are one vector -- bit-exact on sampled queries against the oracle run on the
    rows that can matter, and within 1.3x of the time of a corpus of distinct rows."""
    import time
    import torch
    from ance_amd.index import FlatIPIndex
    from oracle import search_ref
    g = torch.Generator(device="cuda").manual_seed(41)
    n, nq, k = 2_000_000, 4096, 200
    x = torch.nn.functional.layer_norm(torch.randn((n, 768), generator=g, device="cuda"), (768,))
    q = torch.nn.functional.layer_norm(torch.randn((nq, 768), generator=g, device="cuda"), (768,))

    def timed(xx):
        idx = FlatIPIndex(768)
        idx.add(xx)
        idx.search(q[:256], k)
        best = None
        for _ in range(3):  # best of three: the assertion below is about the kernel, not about a noisy neighbour
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            D, I = idx.search(q, k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, D.cpu().numpy(), I.cpu().numpy()

    t_distinct, _, _ = timed(x)
    v0 = x[123].clone()
    dup = torch.rand((n,), generator=g, device="cuda") < 0.5
    dup[:1000] = False
    xd = torch.where(dup[:, None], v0[None, :], x)
    q2 = q.clone()
    q2[5] = v0           # whole list = the class
    q2[6] = v0 + q[6]    # class somewhere in the list
    q = q2
    t_dup, D, I = timed(xd)
    _diag("dedup_scale", t_distinct=t_distinct, t_dup=t_dup, n_dup=int(dup.sum().item()))
    assert t_dup < 1.3 * t_distinct, (t_dup, t_distinct)
    # oracle on: every non-class row that could matter is unknown, so check sampled queries against the chain
    # scores of (their reported rows + a random sample + the first members of the class)
    dup_ids = torch.nonzero(dup).flatten()[:400].cpu().numpy()
    samp = torch.randint(0, n, (30000,), generator=torch.Generator().manual_seed(2)).numpy()
    for r in (0, 5, 6, 1000, 4095):
        rows = np.unique(np.concatenate([I[r], samp, dup_ids]))
        xs = xd[torch.from_numpy(rows).cuda()].cpu().numpy()
        Do, Io = search_ref.flat_ip_topk_chain(xs, q[r:r + 1].cpu().numpy(), k)
        assert np.array_equal(rows[Io[0]], I[r]), r
        assert np.array_equal(Do[0], D[r]), r


def _encoder_like(rng, n, spread=0.12):
    """rows of one encoder: a large common component + small deviations (random-init roberta-base gives cosine 0.99
    between any two passages, scores 737 +- 1.7): un-centred, 2 eps of the fp16 filter is wider than the whole score
    distribution."""
    c = rng.standard_normal(768).astype(np.float32)
    c = (c / np.linalg.norm(c) * np.sqrt(768.0)).astype(np.float32)
    x = (c[None, :] + spread * rng.standard_normal((n, 768))).astype(np.float32)
    return x, c


def test_rows_with_a_large_common_component(monkeypatch):
    rng = np.random.default_rng(50)
    x, c = _encoder_like(rng, 80000)
    x[70000] = x[5]
    q = (c[None, :] + 0.12 * rng.standard_normal((301, 768))).astype(np.float32)
    q[300] = x[5]
    D1, I1 = _check_exact("common_component", x, q, 200)
    monkeypatch.setenv("ANCE_FAST_CENTER", "0")  # un-centred: every query overflows and is redone by the scan -- same bits
    D2, I2 = _search(x, q, 200)
    assert np.array_equal(I1, I2) and np.array_equal(D1, D2)


def test_common_component_stays_on_the_fast_path():
    """2 M encoder-like rows: with the centred image the launch must not fall back to the exact scan (it did for every
    query before): compare the time with a corpus of LayerNorm-distributed rows of the same size."""
    import time
    import torch
    from ance_amd.index import FlatIPIndex
    g = torch.Generator(device="cuda").manual_seed(51)
    n, nq, k = 2_000_000, 8192, 200
    c = torch.randn((768,), generator=g, device="cuda")
    c = c / c.norm() * (768.0 ** 0.5)

    def timed(x, q):
        idx = FlatIPIndex(768)
        idx.add(x)
        idx.search(q[:256], k)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            D, I = idx.search(q, k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, D, I

    xl = torch.nn.functional.layer_norm(torch.randn((n, 768), generator=g, device="cuda"), (768,))
    ql = torch.nn.functional.layer_norm(torch.randn((nq, 768), generator=g, device="cuda"), (768,))
    t_ln, _, _ = timed(xl, ql)
    del xl
    xe = c[None, :] + 0.12 * torch.randn((n, 768), generator=g, device="cuda")
    qe = c[None, :] + 0.12 * torch.randn((nq, 768), generator=g, device="cuda")
    t_enc, D, I = timed(xe, qe)
    _diag("common_component_scale", t_ln=t_ln, t_enc=t_enc)
    assert t_enc < 1.5 * t_ln, (t_enc, t_ln)
    # spot check against the oracle on the rows that can matter
    from oracle import search_ref
    I = I.cpu().numpy()
    D = D.cpu().numpy()
    samp = torch.randint(0, n, (20000,), generator=torch.Generator().manual_seed(3)).numpy()
    for r in (0, 4000, 8191):
        rows = np.unique(np.concatenate([I[r], samp]))
        Do, Io = search_ref.flat_ip_topk_chain(xe[torch.from_numpy(rows).cuda()].cpu().numpy(), qe[r:r + 1].cpu().numpy(), k)
        assert np.array_equal(rows[Io[0]], I[r]) and np.array_equal(Do[0], D[r])


def test_whole_chunk_redo_with_a_short_last_chunk(monkeypatch):
    """6,980 queries (the MS MARCO dev set) = launch chunks of 4,096 + 2,884, every query overflowing (un-centred
    encoder-like rows): the exact-scan redo of the SHORT chunk must run inside the buffers sized for the full one
    (planned on its own it picks more corpus splits -- a memory fault at 2 M rows before the fix)."""
    from oracle import search_ref
    monkeypatch.setenv("ANCE_FAST_CENTER", "0")
    rng = np.random.default_rng(52)
    x, c = _encoder_like(rng, 60000)
    q = (c[None, :] + 0.12 * rng.standard_normal((6980, 768))).astype(np.float32)
    D, I = _search(x, q, 100)
    pick = np.concatenate([np.arange(0, 40), np.arange(4090, 4130), np.arange(6940, 6980)])
    Do, Io = search_ref.flat_ip_topk_chain(x, q[pick], 100)
    assert np.array_equal(I[pick], Io) and np.array_equal(D[pick], Do)


def test_two_towers_with_different_common_components():
    """DPR-like: queries share one large component, passages another (two encoders): the mean QUERY is taken out of the fp16
    operand too and comes back as a per-row bias; bit-exact results all the same."""
    rng = np.random.default_rng(53)
    x, c = _encoder_like(rng, 70000)
    c2 = rng.standard_normal(768).astype(np.float32)
    c2 = (0.8 * c + 0.6 * c2 / np.linalg.norm(c2) * np.sqrt(768.0)).astype(np.float32)
    q = (c2[None, :] + 0.12 * rng.standard_normal((700, 768))).astype(np.float32)
    _check_exact("two_towers", x, q, 100)
    # a handful of queries only (the mean query of a small call), and queries with no common component at all
    _check_exact("two_towers_few", x, q[:3], 200)
    from oracle import synth
    _check_exact("common_rows_ln_queries", x, synth.ln_rows(rng, 90), 50)


def test_image_of_another_matrix_is_not_trusted():
    """A search image carries the stamp of the matrix it was built from (n, d, rows pointer).  Handed to a search over
    other rows -- here: same shape, another tensor -- the library must not filter through it (stale fp16 rows, someone
    else's live2row): every query is answered by the exact scan, same bits as the oracle (the C entry point
    used to trust d_index completely)."""
    import ctypes
    import torch
    from ance_amd import _lib
    from ance_amd.index import FlatIPIndex
    from oracle import search_ref, synth
    rng = np.random.default_rng(61)
    x1, x2 = synth.ln_rows(rng, 20000), synth.ln_rows(rng, 20000)
    q = synth.ln_rows(rng, 70)
    idx = FlatIPIndex(768)
    idx.add(x1)
    idx.search(q, 50)                          # builds the image of x1
    img = idx._image
    assert img is not None and img is not False
    x2d = torch.from_numpy(x2).cuda()
    qd = torch.from_numpy(q).cuda()
    L = _lib.lib()
    bad0 = ctypes.c_ulonglong()
    _lib.check(L.ance_search_bad_image_calls(ctypes.byref(bad0)), "ance_search_bad_image_calls")
    idx.search(q, 50)                          # the normal path: the image matches, nothing is counted
    bad1 = ctypes.c_ulonglong()
    L.ance_search_bad_image_calls(ctypes.byref(bad1))
    assert bad1.value == bad0.value
    n, k = 20000, 50
    D = torch.empty((70, k), dtype=torch.float32, device="cuda")
    I = torch.empty((70, k), dtype=torch.int64, device="cuda")
    ws = torch.empty(L.ance_ip_topk_indexed_workspace_bytes(n, 70, 768, k), dtype=torch.uint8, device="cuda")
    rc = L.ance_ip_topk_indexed(ctypes.c_void_p(x2d.data_ptr()), n, 0, ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(qd.data_ptr()),
                                70, 768, k, ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                ws.numel(), _lib.current_stream_ptr())
    _lib.check(rc, "ance_ip_topk_indexed")
    torch.cuda.synchronize()
    Do, Io = search_ref.flat_ip_topk_chain(x2, q, k)
    assert np.array_equal(I.cpu().numpy(), Io) and np.array_equal(D.cpu().numpy(), Do)
    bad2 = ctypes.c_ulonglong()
    L.ance_search_bad_image_calls(ctypes.byref(bad2))
    assert bad2.value == bad1.value + 1        # ... and the silent fallback is visible to whoever asks
    # a buffer that was never built (all zero) is refused the same way
    blank = torch.zeros_like(img)
    rc = L.ance_ip_topk_indexed(ctypes.c_void_p(x2d.data_ptr()), n, 0, ctypes.c_void_p(blank.data_ptr()), ctypes.c_void_p(qd.data_ptr()),
                                70, 768, k, ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                ws.numel(), _lib.current_stream_ptr())
    _lib.check(rc, "ance_ip_topk_indexed")
    torch.cuda.synchronize()
    assert np.array_equal(I.cpu().numpy(), Io) and np.array_equal(D.cpu().numpy(), Do)
