"""N > 1 path on a real device: two ranks share cuda:0 (gloo rendezvous -- RCCL refuses two ranks on one GPU), each
with the real HipEngine and Dist: contiguous record sharding, per-rank HIP encode, query all-gather, per-shard HIP search
with row_base, all-to-all by query owner, HIP merge, gather on rank 0, native post-search stage.  The files must be
byte-identical to the single-rank run of the same job (SURVEY.md 8e invariant: results do not depend on the GPU count).
Also a three-rank run: a shard count that does not divide the collection sizes."""
import os
import random
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _job(rank, world, data, ckpt, out, port, L=64, max_tokens=16384, precision=None, fill_keep_gb=None):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ance_amd import ann_data_gen as adg
    from ance_amd import negatives
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if fill_keep_gb is not None:
        # memory pressure from the first search on: every rank in turn drops its cached blocks and takes its share of the free
        # device memory, so that about fill_keep_gb GB stay free -- the caching allocator must then recycle the per-chunk blocks of
        # sharded_search at once, across its two streams (a missing record_stream shows as a reuse-after-free: wrong files)
        real, state = adg.sharded_search, {}

        def under_pressure(engine, d, *a, **kw):
            if "fill" not in state:
                torch.cuda.synchronize()
                state["fill"] = []
                for r in range(d.world):
                    if d.rank == r:
                        torch.cuda.empty_cache()
                        free, _ = torch.cuda.mem_get_info()
                        take = int((free - fill_keep_gb * 2 ** 30) / (d.world - r))
                        if take > 0:
                            state["fill"].append(torch.empty(take, dtype=torch.uint8, device="cuda"))
                    d.barrier()
                state["free_after"] = torch.cuda.mem_get_info()[0]
                assert state["free_after"] < (fill_keep_gb + 0.5) * 2 ** 30, state["free_after"]
            return real(engine, d, *a, **kw)

        adg.sharded_search = under_pressure
    args = types.SimpleNamespace(data_dir=data, output_dir=out, cache_dir=out, inference=False, topk_training=100,
                                 negative_sample=8, ann_chunk_factor=1, ann_measure_topk_mrr=False, model_type="rdot_nll",
                                 max_seq_length=L, max_query_length=32, device=torch.device("cuda", 0), max_tokens=max_tokens,
                                 encoder_precision=precision)
    train_pos, dev_pos = negatives.load_positive_ids(data)
    random.seed(4321)
    d = adg.Dist()
    assert d.world == world and d.rank == rank
    res = adg.generate_new_ann(args, 0, ckpt, train_pos, dev_pos, 100, engine=adg.HipEngine(args.device), dist=d)
    assert (res is not None) == (rank == 0)
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", [None, "fp16"])  # None: the job's default = the split (fp32-grade) mode
def test_multi_rank_refresh_on_one_gpu(tmp_path, precision, monkeypatch):
    from safetensors.torch import save_file
    monkeypatch.setenv("ANCE_ENCODE_BLOCK", "3000")  # several blocks per rank through the 3-deep pinned ring of encode_cache
    monkeypatch.setenv("ANCE_SEARCH_CHUNK", "400")   # several exchange chunks per search (side-stream overlap of sharded_search)
    from oracle import encoder_ref, synth
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, n_passages=20000, n_train=1500, n_dev=301, L=64, Lq=32, seed=11, len_median=30)
    sd = encoder_ref.random_state_dict(seed=5, n_layers=2, ln_jitter=0.1)
    ckpt = tmp_path / "checkpoint-100"
    ckpt.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt / "model.safetensors"))
    outs = {}
    for world in (1, 2, 3):
        out = str(tmp_path / ("w%d" % world))
        port = 29600 + (os.getpid() + world) % 2000
        if world == 1:
            _job(0, 1, data, str(ckpt) + "/", out, port, 64, 16384, precision)
        else:
            torch.multiprocessing.spawn(_job, args=(world, data, str(ckpt) + "/", out, port, 64, 16384, precision), nprocs=world, join=True)
        outs[world] = {n: open(os.path.join(out, n)).read() for n in ("ann_training_data_0", "ann_ndcg_0")}
        assert outs[world]["ann_training_data_0"].count("\n") == 1500
    assert outs[2] == outs[1], "2 ranks"
    assert outs[3] == outs[1], "3 ranks"


def test_two_rank_refresh_at_512_tokens(tmp_path):
    """BASELINE configs[2] in miniature: FirstP at seq_len 512 (the long-sequence attention path: 128 queries per round,
    K / V^T of up to 512 keys in LDS), the corpus sharded over two ranks that share cuda:0, per-shard lists exchanged by
    query owner and merged -- files byte-identical to the single-rank run.  (fp16 fast mode; the split mode at 512 tokens is the
    next test.)"""
    from safetensors.torch import save_file
    from oracle import encoder_ref, synth
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, n_passages=3000, n_train=400, n_dev=101, L=512, Lq=32, seed=12, len_median=260, len_sigma=0.6)
    sd = encoder_ref.random_state_dict(seed=6, n_layers=2, ln_jitter=0.1)
    ckpt = tmp_path / "checkpoint-100"
    ckpt.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt / "model.safetensors"))
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / ("w%d" % world))
        port = 29650 + (os.getpid() + world) % 2000
        if world == 1:
            _job(0, 1, data, str(ckpt) + "/", out, port, 512, 32768, "fp16")
        else:
            torch.multiprocessing.spawn(_job, args=(world, data, str(ckpt) + "/", out, port, 512, 32768, "fp16"), nprocs=world, join=True)
        outs[world] = {n: open(os.path.join(out, n)).read() for n in ("ann_training_data_0", "ann_ndcg_0")}
        assert outs[world]["ann_training_data_0"].count("\n") == 400
    assert outs[2] == outs[1]


def test_two_rank_refresh_in_split_mode_at_512_tokens(tmp_path):
    """The same invariant for the fp32-grade split mode (--encoder_precision split), at seq_len 512 so that its long-sequence
    attention path (keys staged 256 at a time) and several micro-batches per rank are exercised: different shard boundaries
    mean different micro-batch compositions, and every row must still come out bit-identical (contraction is off in the split
    epilogues for exactly this reason)."""
    from safetensors.torch import save_file
    from oracle import encoder_ref, synth
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, n_passages=2500, n_train=300, n_dev=101, L=512, Lq=32, seed=13, len_median=200, len_sigma=0.7)
    sd = encoder_ref.random_state_dict(seed=7, n_layers=2, ln_jitter=0.1)
    ckpt = tmp_path / "checkpoint-100"
    ckpt.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt / "model.safetensors"))
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / ("w%d" % world))
        port = 29700 + (os.getpid() + world) % 2000
        if world == 1:
            _job(0, 1, data, str(ckpt) + "/", out, port, 512, 16384, "split")
        else:
            torch.multiprocessing.spawn(_job, args=(world, data, str(ckpt) + "/", out, port, 512, 16384, "split"), nprocs=world, join=True)
        outs[world] = {n: open(os.path.join(out, n)).read() for n in ("ann_training_data_0", "ann_ndcg_0")}
        assert outs[world]["ann_training_data_0"].count("\n") == 300
    assert outs[2] == outs[1]


def test_two_rank_refresh_under_memory_pressure(tmp_path, monkeypatch):
    """the per-chunk exchange of sharded_search allocates under a side stream what the caller's stream reads
    (record_stream on every gathered chunk).  Two ranks on cuda:0, 64-query exchange chunks (24 chunks for the train queries),
    and from the first search on less than 1 GB of free device memory: the allocator has to hand freed blocks out again
    immediately, on either stream.  The files must still equal the single-rank run's, byte for byte."""
    from safetensors.torch import save_file
    monkeypatch.setenv("ANCE_SEARCH_CHUNK", "64")
    from oracle import encoder_ref, synth
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, n_passages=12000, n_train=1500, n_dev=301, L=64, Lq=32, seed=21, len_median=30)
    sd = encoder_ref.random_state_dict(seed=8, n_layers=2, ln_jitter=0.1)
    ckpt = tmp_path / "checkpoint-100"
    ckpt.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt / "model.safetensors"))
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / ("w%d" % world))
        port = 29750 + (os.getpid() + world) % 2000
        if world == 1:
            _job(0, 1, data, str(ckpt) + "/", out, port, 64, 16384, None)
        else:
            torch.multiprocessing.spawn(_job, args=(world, data, str(ckpt) + "/", out, port, 64, 16384, None, 1.0), nprocs=world, join=True)
        outs[world] = {n: open(os.path.join(out, n)).read() for n in ("ann_training_data_0", "ann_ndcg_0")}
        assert outs[world]["ann_training_data_0"].count("\n") == 1500
    assert outs[2] == outs[1]
