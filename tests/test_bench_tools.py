"""bench.py's own data tools (CPU): the synthetic tokenised cache `bench.py --full` writes is the reference's cache format
(utils/util.py:257-307) as `ance_amd.cache.TokenCache` reads it, block boundaries included, and deterministic."""
import hashlib
import os

import numpy as np


def test_synthetic_cache_round_trip(tmp_path):
    import bench
    from ance_amd.cache import TokenCache
    p = str(tmp_path / "passages")
    secs, mean_len = bench.write_synthetic_cache(p, 5000, 128, 70.0, 0.45, 8, seed=1, block=2048)  # 3 blocks, last one short
    c = TokenCache(p)
    assert len(c) == 5000 and c.embedding_size == 128 and c.record_size == 4 + 4 * 128
    assert os.path.getsize(p) == 5000 * c.record_size
    L, ids = c.lengths(), c.ids()
    assert L.min() >= 8 and L.max() <= 128 and abs(float(L.mean()) - mean_len) < 1e-9
    rows = np.arange(5000)
    assert np.all(ids[:, 0] == 0) and np.all(ids[rows, L - 1] == 2)          # <s> ... </s>
    assert np.all((np.arange(128)[None, :] < L[:, None]) | (ids == 1))       # pad = 1 after the length
    inner = (np.arange(128)[None, :] > 0) & (np.arange(128)[None, :] < (L - 1)[:, None])
    assert np.all(ids[inner] >= 3) and np.all(ids[inner] < 50265)
    plen, tok = c[4999]                                                       # the reference's __getitem__ contract
    assert plen == int(L[4999]) and np.array_equal(tok, ids[4999])
    # deterministic in (seed, block)
    p2 = str(tmp_path / "again")
    bench.write_synthetic_cache(p2, 5000, 128, 70.0, 0.45, 8, seed=1, block=2048)
    assert hashlib.sha256(open(p, "rb").read()).digest() == hashlib.sha256(open(p2, "rb").read()).digest()


def test_bench_records_match_cache_layout():
    import bench
    rng = np.random.default_rng(0)
    rec, lens = bench.synthetic_records(rng, 64, 32)
    assert rec.shape == (64, 33) and rec.dtype == np.int32
    assert np.array_equal(rec[:, 0].view(">u4").astype(np.int64).reshape(-1), lens.astype(np.int64))  # big-endian header
