"""N > 1 path on CPU: two gloo processes run the job's host logic (contiguous sharding, query
all-gather, per-shard search, all-gather + canonical merge, rank-0 post-processing) with a
stand-in engine, and must produce byte-identical files to the single-process run.

The stand-in engine uses the oracle for the device operations -- it is test scaffolding for the
HOST logic only; the product never constructs anything but HipEngine."""
import json
import os
import random
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeEngine:
    def __init__(self):
        rng = np.random.default_rng(0)
        self.proj = rng.standard_normal((50265 % 997 + 64, 768)).astype(np.float32)

    def encode_cache(self, model, cache, r0, r1, is_query, chunks=1):
        ids = cache.ids(r0, r1)
        lens = cache.lengths(r0, r1)
        L = ids.shape[1] // chunks
        out = np.zeros(((r1 - r0) * chunks, 768), dtype=np.float32)
        for i in range(r1 - r0):
            for c in range(chunks):
                lc = int(np.clip(lens[i] - c * L, 0, L))
                toks = ids[i, c * L:c * L + lc]
                v = self.proj[toks % self.proj.shape[0]].sum(axis=0) if lc else self.proj[0] * 0 + 1.0
                v = v - v.mean()
                out[i * chunks + c] = v / np.sqrt((v * v).mean() + 1e-5)
        return torch.from_numpy(out)

    def search(self, x, row_base, q, k):
        from oracle import search_ref
        D, I = search_ref.flat_ip_topk_chain(x.numpy(), q.numpy(), k, row_base=row_base)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(self, Dp, Ip):
        from oracle import search_ref
        D, I = search_ref.topk_merge(Dp.numpy(), Ip.numpy(), Dp.shape[2])
        return torch.from_numpy(D), torch.from_numpy(I)

    def to_numpy(self, t):
        return t.numpy()


def _args(data, out):
    return types.SimpleNamespace(data_dir=data, output_dir=out, cache_dir=out, inference=False, topk_training=30,
                                 negative_sample=4, ann_chunk_factor=3, ann_measure_topk_mrr=False, model_type="rdot_nll",
                                 max_seq_length=32, max_query_length=16, device=None)


def _run(rank, world, data, out, port, chunks):
    sys.path.insert(0, ROOT)
    from ance_amd import ann_data_gen as adg
    from ance_amd import negatives
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    args = _args(data, out)
    train_pos, dev_pos = negatives.load_positive_ids(data)
    model = types.SimpleNamespace(chunks=chunks, q=None, b=None)
    random.seed(1234)
    res = adg.generate_new_ann(args, 1, "/m/checkpoint-500/", train_pos, dev_pos, 500, engine=FakeEngine(), model=model,
                               dist=adg.Dist())
    if rank == 0:
        assert res is not None and res[1] > 0
    else:
        assert res is None
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunks,world,search_chunk", [(1, 2, None), (2, 2, None), (1, 3, 5), (2, 2, 7)])
def test_multi_rank_refresh_equals_single_process(tmp_path, monkeypatch, chunks, world, search_chunk):
    """2 and 3 ranks; search_chunk: the exchange of sharded_search runs per launch chunk of queries (32,768 in production) --
    5 / 7 make the 47 train and 13 dev queries cross many chunk borders, with last chunks that do not divide by the world size."""
    from oracle import synth
    if search_chunk:
        monkeypatch.setenv("ANCE_SEARCH_CHUNK", str(search_chunk))
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, n_passages=301, n_train=47, n_dev=13, L=32, Lq=16, seed=3, len_median=14)
    out1, out2 = str(tmp_path / "w1"), str(tmp_path / "w2")
    _run(0, 1, data, out1, 0, chunks)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(world, data, out2, port, chunks), nprocs=world, join=True)
    for name in ("ann_training_data_1", "ann_ndcg_1"):
        assert open(os.path.join(out1, name)).read() == open(os.path.join(out2, name)).read(), name
    j = json.load(open(os.path.join(out2, "ann_ndcg_1")))
    assert j["checkpoint"] == "/m/checkpoint-500/" and 0.0 <= j["ndcg"] <= 1.0
    # the consumer's parser (data/msmarco_data.py:338-343) accepts every line
    n_pass = 301
    for line in open(os.path.join(out2, "ann_training_data_1")):
        qid, pos, negs = line.rstrip("\n").split("\t")
        negs = [int(x) for x in negs.split(",")]
        assert 0 <= int(qid) < 47 and 0 <= int(pos) < n_pass and all(0 <= x < n_pass for x in negs) and len(negs) <= 4
        assert int(pos) not in negs and len(set(negs)) == len(negs)
