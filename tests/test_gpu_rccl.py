"""RCCL on real devices (skipped only on a box with fewer than two GPUs): the collectives of the N > 1 refresh --
query all-gather, per-shard search with row_base, all-to-all by query owner, merge, gather on rank 0
(ance_amd/ann_data_gen.py: Dist, sharded_search; reference: drivers/run_ann_data_gen.py:637-640 init_process_group("nccl"),
utils/util.py:barrier_array_merge) -- on device tensors over backend "nccl".  The merged lists must be bit-identical to one
rank searching the whole corpus."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from ance_amd import ann_data_gen as adg
    from ance_amd.cache import shard_range
    from oracle import synth
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    rng = np.random.default_rng(3)
    n, nq, k = 30011, 257, 100
    x = synth.ln_rows(rng, n)
    q = synth.ln_rows(rng, nq)
    d = adg.Dist()
    assert d.world == world and d.rank == rank and dist.get_backend() == "nccl"
    eng = adg.HipEngine(dev)
    r0, r1 = shard_range(n, rank, world)
    # every rank holds only its query block before the gather, like the encode phase leaves them
    q0, q1 = shard_range(nq, rank, world)
    q_all = adg.gather_queries(d, torch.from_numpy(q[q0:q1]).to(dev), nq)
    assert torch.equal(q_all.cpu(), torch.from_numpy(q))
    res = adg.sharded_search(eng, d, torch.from_numpy(x[r0:r1]).to(dev), r0, q_all, k)
    if rank == 0:
        D, I = res
        np.savez(out_path, D=D.cpu().numpy(), I=I.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _job_rank(rank, world, port, data, ckpt, out):
    """One rank of a WHOLE refresh (ance_amd.ann_data_gen.generate_new_ann with the real HipEngine) over backend nccl, one device
    per rank -- what `python -m torch.distributed.run ... -m ance_amd.ann_data_gen` runs (drivers/run_ann_data_gen.py:637-640)."""
    import random
    import types
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from ance_amd import ann_data_gen as adg
    from ance_amd import negatives
    dev = torch.device("cuda", rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    args = types.SimpleNamespace(data_dir=data, output_dir=out, cache_dir=out, inference=False, topk_training=100,
                                 negative_sample=8, ann_chunk_factor=1, ann_measure_topk_mrr=False, model_type="rdot_nll",
                                 max_seq_length=64, max_query_length=32, device=dev, max_tokens=16384, encoder_precision=None)
    train_pos, dev_pos = negatives.load_positive_ids(data)
    random.seed(4321)
    d = adg.Dist()
    assert d.world == world and d.rank == rank
    res = adg.generate_new_ann(args, 0, ckpt, train_pos, dev_pos, 100, engine=adg.HipEngine(dev), dist=d)
    assert (res is not None) == (rank == 0)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: fewer than 2 GPUs on this box")
def test_refresh_job_over_rccl_equals_one_rank(tmp_path, monkeypatch):
    """The whole job on 2 (up to 4) devices over RCCL against the same job on one: both output files byte for byte (SURVEY 8e:
    results do not depend on the GPU count).  ANCE_SEARCH_CHUNK=512 makes the 1,500 train queries cross several exchange chunks, so
    the side-stream overlap of sharded_search (chunk i's all-to-all + merge + gather under chunk i + 1's scan) really runs."""
    from safetensors.torch import save_file
    from oracle import encoder_ref, synth
    monkeypatch.setenv("ANCE_SEARCH_CHUNK", "512")
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, n_passages=20000, n_train=1500, n_dev=301, L=64, Lq=32, seed=11, len_median=30)
    sd = encoder_ref.random_state_dict(seed=5, n_layers=2, ln_jitter=0.1)
    ckpt = tmp_path / "checkpoint-100"
    ckpt.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt / "model.safetensors"))
    outs = {}
    for world in (1, min(torch.cuda.device_count(), 4)):
        out = str(tmp_path / ("w%d" % world))
        port = 29800 + (os.getpid() + world) % 2000
        torch.multiprocessing.spawn(_job_rank, args=(world, port, data, str(ckpt) + "/", out), nprocs=world, join=True)
        outs[world] = {n: open(os.path.join(out, n)).read() for n in ("ann_training_data_0", "ann_ndcg_0")}
        assert outs[world]["ann_training_data_0"].count("\n") == 1500
    a, b = outs.values()
    assert a == b


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: fewer than 2 GPUs on this box")
def test_sharded_search_over_rccl(tmp_path):
    from oracle import search_ref, synth
    world = min(torch.cuda.device_count(), 4)
    out = str(tmp_path / "merged.npz")
    port = 29700 + os.getpid() % 2000
    torch.multiprocessing.spawn(_rank, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    rng = np.random.default_rng(3)
    x = synth.ln_rows(rng, 30011)
    q = synth.ln_rows(rng, 257)
    Do, Io = search_ref.flat_ip_topk_chain(x, q, 100)
    assert np.array_equal(got["I"], Io) and np.array_equal(got["D"], Do)
