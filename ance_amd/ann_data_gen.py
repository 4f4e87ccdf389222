"""The ANN hard-negative refresh job, MI355X-native (drop-in for drivers/run_ann_data_gen.py).

Same CLI flags (seam B3, drivers/run_ann_data_gen.py:443-627), same inputs (B2: tokenised caches,
qrels, ``checkpoint-N/`` dirs with the ``scheduler.pt`` commit marker) and the same outputs (B1:
``ann_training_data_N`` then ``ann_ndcg_N``), so ``drivers/run_ann.py`` consumes them unchanged.

What is different is where the work happens:
  * every rank encodes a CONTIGUOUS block of records and keeps the fp32 embeddings in its own HBM
    (the reference strides records over ranks and round-trips 27 GB through ``.npy`` files,
    utils/util.py:87-146);
  * search runs on every GPU against its resident shard (the reference searches on rank 0's CPU
    with faiss, :265-303); per-shard top-k lists are exchanged by query owner (one RCCL all-to-all),
    merged under the canonical order (score desc, row id asc) and gathered on rank 0, so ``I`` is
    independent of the GPU count;
  * row id == record offset (FirstP) or record * chunks + chunk (MaxP).

Launch: one process per GPU, e.g.
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m ance_amd.ann_data_gen ...
"""
import argparse
import logging
import os
import random
import time

import numpy as np

from . import negatives
from .cache import TokenCache, shard_range

logger = logging.getLogger(__name__)

# ------------------------------------------------------------------------------------- helpers --


def get_checkpoint_no(checkpoint_path):
    """Last integer in the name (utils/util.py:224-226)."""
    import re
    nums = re.findall(r"\d+", checkpoint_path)
    return int(nums[-1]) if nums else 0


def get_latest_ann_data(ann_data_path):
    """(n, path of ann_training_data_n, parsed ann_ndcg_n) of the newest refresh, or (-1, None, None)
    (utils/util.py:229-243)."""
    import json
    prefix = "ann_ndcg_"
    if not os.path.exists(ann_data_path):
        return -1, None, None
    nos = []
    for s in next(os.walk(ann_data_path))[2]:
        if s.startswith(prefix) and s[len(prefix):].isdigit():
            nos.append(int(s[len(prefix):]))
    if not nos:
        return -1, None, None
    no = max(nos)
    with open(os.path.join(ann_data_path, prefix + str(no)), "r") as f:
        ndcg_json = json.load(f)
    return no, os.path.join(ann_data_path, "ann_training_data_" + str(no)), ndcg_json


def get_latest_checkpoint(args):
    """Newest ``training_dir/checkpoint-N`` that contains ``scheduler.pt`` -- the last file the
    trainer writes, i.e. its commit marker -- else ``init_model_dir``
    (drivers/run_ann_data_gen.py:55-71)."""
    if not os.path.exists(args.training_dir):
        return args.init_model_dir, 0
    best = -1
    for sub in next(os.walk(args.training_dir))[1]:
        if os.path.exists(os.path.join(args.training_dir, sub, "scheduler.pt")):
            best = max(best, get_checkpoint_no(sub))
    if best >= 0:
        return os.path.join(args.training_dir, "checkpoint-" + str(best)) + "/", best
    return args.init_model_dir, 0


class Dist:
    """Thin view of torch.distributed that also works when no group is initialised."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1

    def barrier(self):
        if self.on:
            self.dist.barrier()

    def all_gather_scalars(self, v, device=None):
        """Every rank's integer ``v`` in rank order: one all-gather of an int64 (no pickling, no object collective)."""
        import torch
        if not self.on or self.world == 1:
            return [int(v)]
        dev = torch.device("cpu") if (device is None or self.dist.get_backend() == "gloo") else device
        mine = torch.tensor([int(v)], dtype=torch.int64, device=dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return [int(t.item()) for t in parts]

    # ``comm``: None, or a dict the caller installed -- every device collective is then bracketed by events on the stream it
    # is issued on and ``comm_ms()`` adds them up per collective (bench.py --gpus N prints it)
    comm = None

    def _timed(self, name, t, fn):
        if self.comm is None:
            return fn()
        if not t.is_cuda:  # host-staged collective (gloo): wall clock of the blocking call
            t0 = time.perf_counter()
            r = fn()
            self.comm.setdefault(name, []).append((None, 1e3 * (time.perf_counter() - t0), t.numel() * t.element_size()))
            return r
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.comm.setdefault(name, []).append((e0, e1, t.numel() * t.element_size()))
        return r

    def comm_ms(self):
        """{collective: {"ms": total, "calls": n, "bytes": payload bytes per rank}} of the collectives recorded since ``comm`` was
        installed (synchronises the device)."""
        import torch
        if not self.comm:
            return {}
        torch.cuda.synchronize()
        return {k: {"ms": sum(b if a is None else a.elapsed_time(b) for a, b, _ in v), "calls": len(v), "bytes": sum(n for _, _, n in v)}
                for k, v in self.comm.items()}

    def all_gather_rows(self, t, per):
        """Concatenate per-rank row blocks (each padded to ``per`` rows) in rank order."""
        import torch
        if not self.on or self.world == 1:
            return t
        pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        dev = pad.device
        if self._host_staged(pad):
            pad = pad.cpu()
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        self._timed("all_gather", pad, lambda: self.dist.all_gather(parts, pad))
        return torch.cat(parts, dim=0).to(dev)

    def _host_staged(self, t):
        """gloo moves host memory only: a device tensor is staged through the host for it (functional multi-rank
        tests on one GPU; RCCL -- backend "nccl" -- takes device tensors as they are)."""
        return self.on and t.is_cuda and self.dist.get_backend() == "gloo"

    def all_to_all_blocks(self, t):
        """``t`` = [world * per, ...]: block j goes to rank j.  Returns [world, per, ...] with [p] = the block rank p
        sent here (one ``all_to_all_single`` over xGMI)."""
        import torch
        if not self.on or self.world == 1:
            return t.unsqueeze(0)
        t = t.contiguous()
        dev = t.device
        src = t.cpu() if self._host_staged(t) else t
        out = torch.empty_like(src)
        self._timed("all_to_all", src, lambda: self.dist.all_to_all_single(out, src))
        out = out.to(dev)
        return out.view((self.world, t.shape[0] // self.world) + tuple(t.shape[1:]))

    def gather_rows_to_root(self, t):
        """Concatenation of every rank's equal-shaped ``t`` in rank order on rank 0; None elsewhere."""
        import torch
        if not self.on or self.world == 1:
            return t
        t = t.contiguous()
        dev = t.device
        src = t.cpu() if self._host_staged(t) else t
        parts = [torch.empty_like(src) for _ in range(self.world)] if self.rank == 0 else None
        self._timed("gather", src, lambda: self.dist.gather(src, parts, dst=0))
        return torch.cat(parts, dim=0).to(dev) if self.rank == 0 else None

    def broadcast_object(self, obj):
        """rank 0's ``obj`` on every rank."""
        if not self.on or self.world == 1:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]


# -------------------------------------------------------------------------------------- engine --


class HipEngine:
    """Device operations of the job on one MI355X (C ABI of include/ance_amd.h).  The only engine the
    product constructs; tests of the multi-process host logic inject a stand-in with this surface."""

    def __init__(self, device=None, block_records=None):
        import torch
        self.torch = torch
        self.device = torch.device(device if device is not None else "cuda")
        self.block_records = block_records
        self._index = None

    def encode_cache(self, model, cache, r0, r1, is_query, chunks=1):
        """Embeddings of records [r0, r1) of a TokenCache -> device fp32 [(r1-r0) * chunks, 768]."""
        torch = self.torch
        enc = model.q if is_query else model.b
        n = r1 - r0
        out = torch.empty((n * chunks, 768), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        rb = cache.record_size
        # records per encode call: every call ends with the two internal streams of the encoder joining (half a micro-batch of
        # idle lane on average) and with one partial micro-batch, so blocks are as large as a 64 MB pinned slot allows --
        # 65,536 records at seq_len 128 (75 micro-batches per call), 32 k at 512, 8 k at 2,048
        B = self.block_records or int(os.environ.get("ANCE_ENCODE_BLOCK", 0)) or max(4096, min(65536, (64 << 20) // rb))
        B = min(B, n)
        ring = [torch.empty((B, rb), dtype=torch.uint8).pin_memory() for _ in range(3)]
        done = [None, None, None]
        for bi, b0 in enumerate(range(0, n, B)):
            b1 = min(b0 + B, n)
            slot = bi % 3
            if done[slot] is not None:
                done[slot].synchronize()
            host = ring[slot][:b1 - b0]
            rec = cache.records(r0 + b0, r0 + b1)
            host.numpy()[...] = rec
            lens = cache.lengths(r0 + b0, r0 + b1)
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            done[slot] = ev
            enc.encode_records(dev, n_chunks=chunks, h_lens=lens, out=out[b0 * chunks:b1 * chunks])
        # range guard of the split arithmetic: blocks whose counters have arrived were checked as the loop went; the rest here
        # (one synchronisation per collection -- the caller is about to search or gather these rows anyway)
        enc.check_range(sync=True)
        return out

    def search(self, x, row_base, q, k):
        """Exact top-k of ``q`` over the resident shard ``x``.  The index (fp16 search image + workspace) of the last
        shard searched is kept: a refresh searches the same embeddings two or three times (dev, train, trivia --
        drivers/run_ann_data_gen.py:276,303) and faiss builds its index once too."""
        from .index import FlatIPIndex
        # the index keeps ``x`` alive, so a later tensor cannot reuse its address while this entry exists
        key = (x.data_ptr(), tuple(x.shape), int(row_base), getattr(x, "_version", 0))
        if self._index is None or self._index[0] != key:
            self._index = None  # drop the old image before the new one is allocated
            idx = FlatIPIndex(x.shape[1], device=self.device, row_base=row_base)
            idx.add(x)
            self._index = (key, idx)
        return self._index[1].search_device(q.contiguous(), k)

    def side_stream(self):
        """Stream the exchange of a multi-rank search runs on (sharded_search), beside the scan on the caller's stream."""
        if getattr(self, "_side", None) is None:
            self._side = self.torch.cuda.Stream(device=self.device)
        return self._side

    def release_index(self):
        """Frees the cached search image (call when the shard's embeddings are about to be replaced)."""
        self._index = None

    def merge(self, D_parts, I_parts):
        from .index import topk_merge_device
        return topk_merge_device(D_parts.contiguous(), I_parts.contiguous())

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)


DRIVER_MAX_TOKENS = 131072  # the refresh drivers' micro-batch (the library default is 65,536: ance_amd.encoder.Encoder)
SEARCH_CHUNK = 32768  # queries per exchange step = the search kernels' launch chunk (csrc/ip_topk_fast.hip)


def _pack_lists(D, I_local):
    """(D fp32 [n, k], ids int32 [n, k]) -> one int32 [n, k, 2] buffer: the wire format of a top-k list (8 bytes per entry)."""
    import torch
    return torch.stack((D.contiguous().view(torch.int32), I_local), dim=-1).contiguous()


def _unpack_lists(buf):
    import torch
    return buf[..., 0].contiguous().view(torch.float32), buf[..., 1]


def sharded_search(engine, dist, x_local, row_base, q_all, k, row_bases=None):
    """Exact top-k of ``q_all`` over the union of every rank's shard.  Every rank scans its shard for ALL queries; the
    per-shard lists are then exchanged by QUERY OWNER -- rank j receives every rank's lists for its block of the queries with one
    all-to-all and merges them under the canonical order (the reference's own shard-search-then-merge:
    utils/eval_mrr.py:137-183) -- and rank 0 gathers the merged blocks.  Per rank that is nq k 8 bytes received instead of
    world x as much with an all-gather, and nq / world merges instead of nq.

    The exchange runs per launch chunk of ``SEARCH_CHUNK`` queries on a side stream: chunk i's all-to-all + merge + gather
    overlap chunk i + 1's scan (with RCCL the collective synchronises with the stream it is issued on; gloo's host-staged
    collectives are simply synchronous).  On the wire a list entry is (score fp32, LOCAL row id int32): the receiver adds the
    sender's ``row_base``, so ids travel in 4 bytes whatever the corpus size.  ``row_bases``: every rank's row_base in rank order
    when the caller knows them (the refresh does: ``cache.shard_range`` is a pure function of the rank); else one int64 all-gather.
    Returns (D, I) on rank 0 and (None, None) elsewhere (only rank 0 post-processes)."""
    if dist.world == 1:
        return engine.search(x_local, row_base, q_all, k)
    import torch
    W, nq = dist.world, q_all.shape[0]
    dev = q_all.device
    on_gpu = q_all.is_cuda
    bases = [int(b) for b in row_bases] if row_bases is not None else dist.all_gather_scalars(int(row_base), dev)
    if len(bases) != W or bases[dist.rank] != int(row_base):
        raise ValueError("sharded_search: row_bases %r do not match this rank's row_base %d" % (bases, int(row_base)))
    base_t = torch.tensor(bases, dtype=torch.int64, device=dev).view(W, 1, 1)
    main = torch.cuda.current_stream(dev) if on_gpu else None
    side = engine.side_stream() if on_gpu and hasattr(engine, "side_stream") else None
    out_D, out_I = [], []
    chunk = int(os.environ.get("ANCE_SEARCH_CHUNK", SEARCH_CHUNK))  # (tests shrink it to cross many chunk borders)
    for c0 in range(0, nq, chunk):
        c1 = min(c0 + chunk, nq)
        n = c1 - c0
        D, I = engine.search(x_local, row_base, q_all[c0:c1], k)
        per = (n + W - 1) // W

        def exchange(D=D, I=I, n=n, per=per):
            I_loc = torch.where(I >= 0, I - row_base, I).to(torch.int32)  # -1 (fewer than k rows) stays -1
            buf = _pack_lists(D, I_loc)
            if per * W != n:  # empty lists for the padding queries of the last owner blocks
                padded = torch.empty((per * W, k, 2), dtype=torch.int32, device=buf.device)
                padded[:n] = buf
                padded[n:, :, 0] = torch.tensor(torch.finfo(torch.float32).min, dtype=torch.float32).view(torch.int32).item()
                padded[n:, :, 1] = -1
                buf = padded
            Dp, Ip = _unpack_lists(dist.all_to_all_blocks(buf))                     # [W, per, k] each, [p] = rank p's lists
            Ip = torch.where(Ip >= 0, Ip.to(torch.int64) + base_t, Ip.to(torch.int64))
            Dm, Im = engine.merge(Dp, Ip)                                           # [per, k], global ids
            Dg, Ig = dist.gather_rows_to_root(Dm), dist.gather_rows_to_root(Im)
            if dist.rank == 0:
                if side is not None:
                    # allocated and written under the side stream, concatenated below on the caller's: the caching allocator
                    # must not hand these blocks to a side-stream allocation of a later chunk before that read has run
                    Dg.record_stream(main)
                    Ig.record_stream(main)
                out_D.append(Dg[:n])
                out_I.append(Ig[:n])

        if side is not None:
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                D.record_stream(side)
                I.record_stream(side)
                exchange()
        else:
            exchange()
    if side is not None:
        main.wait_stream(side)
    if dist.rank != 0:
        return None, None
    if not out_D:  # no queries at all
        return (torch.empty((0, k), dtype=torch.float32, device=dev), torch.empty((0, k), dtype=torch.int64, device=dev))
    # (main has waited for the side stream above: the concatenation runs, and allocates, on the caller's stream)
    return torch.cat(out_D, dim=0), torch.cat(out_I, dim=0)


def shard_row_bases(n_records, world, chunks=1):
    """row_base of every rank's shard, in rank order: ``encode_collection`` shards by ``cache.shard_range``, a pure function."""
    return [shard_range(n_records, r, world)[0] * chunks for r in range(world)]


def encode_collection(engine, dist, model, cache, is_query, chunks=1, r_begin=0, r_end=None):
    """Each rank encodes a contiguous block; returns (local embeddings, first local row, n rows total)."""
    n = len(cache) if r_end is None else r_end
    s0, s1 = shard_range(n - r_begin, dist.rank, dist.world)
    with cache as c:
        emb = engine.encode_cache(model, c, r_begin + s0, r_begin + s1, is_query, chunks)
    return emb, s0 * chunks, (n - r_begin) * chunks


def gather_queries(dist, emb_local, n_total):
    """All ranks end up with the full [n_total, 768] query matrix in record order."""
    if dist.world == 1:
        return emb_local
    per = (n_total + dist.world - 1) // dist.world
    out = dist.all_gather_rows(emb_local, per)
    # blocks are contiguous and only the last ones can be short, so the real rows are a prefix of
    # every block; compact them
    if per * dist.world == n_total:
        return out
    keep = []
    for r in range(dist.world):
        s0, s1 = shard_range(n_total, r, dist.world)
        keep.append(out[r * per:r * per + (s1 - s0)])
    import torch
    return torch.cat(keep, dim=0)


# ----------------------------------------------------------------------------------- the job --


class _Phases:
    """Wall time per phase of a refresh (``args.timings``: a dict to fill, or absent).  A phase ends with a device
    synchronisation so that its kernels are charged to it -- only when somebody asked for timings."""

    def __init__(self, sink):
        self.sink = sink
        self.t = time.perf_counter()

    def mark(self, name):
        if self.sink is None:
            return
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
        now = time.perf_counter()
        self.sink[name] = self.sink.get(name, 0.0) + (now - self.t)
        logger.info("phase %s: %.2f s", name, now - self.t)
        self.t = now


def generate_new_ann(args, output_num, checkpoint_path, training_query_positive_id, dev_query_positive_id,
                     latest_step_num, engine=None, model=None, dist=None):
    """One refresh (drivers/run_ann_data_gen.py:231-336).  Returns (dev_ndcg, n_dev_queries) on rank 0."""
    dist = dist or Dist()
    ph = _Phases(getattr(args, "timings", None))
    if engine is None:
        engine = HipEngine(getattr(args, "device", None))
    if model is None:
        from .encoder import load_model
        model = load_model(args.model_type, checkpoint_path, max_seq_length=args.max_seq_length,
                           max_tokens=getattr(args, "max_tokens", None) or DRIVER_MAX_TOKENS, device=getattr(args, "device", None),
                           precision=getattr(args, "encoder_precision", None))
    chunks = getattr(model, "chunks", 1)
    ph.mark("load_model")

    logger.info("***** inference of dev query *****")
    dev_cache = TokenCache(os.path.join(args.data_dir, "dev-query"))
    dev_local, _, n_dev = encode_collection(engine, dist, model, dev_cache, True)
    ph.mark("encode_dev_queries")

    logger.info("***** inference of passages *****")
    p_cache = TokenCache(os.path.join(args.data_dir, "passages"))
    p_local, p_row0, n_rows = encode_collection(engine, dist, model, p_cache, False, chunks)
    logger.info("***** Done passage inference *****")
    ph.mark("encode_passages")

    if args.inference:
        _dump_inference(args, engine, dist, latest_step_num, dev_local, p_local, p_row0, chunks, None, 0)
        dist.barrier()
        return None

    logger.info("***** inference of train query *****")
    q_cache = TokenCache(os.path.join(args.data_dir, "train-query"))
    nq_all = len(q_cache)
    q_start, q_end = negatives.query_chunk(nq_all, output_num, args.ann_chunk_factor)
    # only the chunk that will be searched is encoded: rows are independent, the result is the same
    q_local, _, n_q = encode_collection(engine, dist, model, q_cache, True, 1, q_start, q_end)
    logger.info("Chunked %d query from %d", n_q, nq_all)
    ph.mark("encode_train_queries")

    dev_all = gather_queries(dist, dev_local, n_dev)
    q_all = gather_queries(dist, q_local, n_q)
    ph.mark("gather_queries")

    bases = shard_row_bases(n_rows // chunks, dist.world, chunks)
    _, dev_I = sharded_search(engine, dist, p_local, p_row0, dev_all, 100, row_bases=bases)
    ph.mark("search_dev")
    _, I = sharded_search(engine, dist, p_local, p_row0, q_all, args.topk_training, row_bases=bases)
    logger.info("***** Done ANN Index *****")
    ph.mark("search_train")

    result = None
    if dist.rank == 0:
        dev_I = engine.to_numpy(dev_I)
        I = engine.to_numpy(I)
        # row -> pid: identity for FirstP, row // chunks for MaxP (one vector per 512-token chunk)
        p2id = np.arange(n_rows, dtype=np.int64) // chunks
        dev_q2id = np.arange(n_dev, dtype=np.int64)
        q2id = np.arange(q_start, q_end, dtype=np.int64)
        dev_ndcg, n_dev_eval = negatives.eval_dev_query(dev_q2id, p2id, dev_query_positive_id, dev_I)
        print("Rank:" + str(dist.rank) + " --- ANN NDCG@10:" + str(dev_ndcg))
        effective_q_id = set(q2id.tolist())
        neg = negatives.select_negatives(q2id, p2id, training_query_positive_id, I, effective_q_id,
                                         args.negative_sample, args.ann_measure_topk_mrr)
        if args.ann_measure_topk_mrr:
            print("Rank:" + str(dist.rank) + " --- ANN MRR:" + str(neg.mrr))
        logger.info("***** Construct ANN Triplet *****")
        os.makedirs(args.output_dir, exist_ok=True)
        negatives.write_ann_files(args.output_dir, output_num, I.shape[0], q2id, effective_q_id,
                                  training_query_positive_id, neg, dev_ndcg, checkpoint_path)
        result = (dev_ndcg, n_dev_eval)
    if hasattr(engine, "release_index"):
        engine.release_index()  # the shard's embeddings die with this refresh
    ph.mark("host_stage")
    dist.barrier()
    return result


def _dump_inference(args, engine, dist, step, dev_local, p_local, p_row0, chunks, q_local, q_row0):
    """``--inference`` dumps (seam B6): per-rank ``{prefix}_data_obj_{rank}.npy`` with the prefixes of
    drivers/run_ann_data_gen.py:215-224,245,252 (consumed by evaluation/Calculate Metrics.ipynb)."""
    os.makedirs(args.output_dir, exist_ok=True)

    def dump(prefix, emb, row0, per_id):
        e = engine.to_numpy(emb)
        ids = np.arange(row0, row0 + e.shape[0], dtype=np.int64) // per_id
        np.save(os.path.join(args.output_dir, "{}_emb_p__data_obj_{}.npy".format(prefix, dist.rank)), e,
                allow_pickle=False)
        np.save(os.path.join(args.output_dir, "{}_embid_p__data_obj_{}.npy".format(prefix, dist.rank)), ids,
                allow_pickle=False)

    dev_n = len(TokenCache(os.path.join(args.data_dir, "dev-query")))
    dump("dev_query_" + str(step) + "_", dev_local, shard_range(dev_n, dist.rank, dist.world)[0], 1)
    dump("passage_" + str(step) + "_", p_local, p_row0, chunks)


def ann_data_gen(args, engine=None, dist=None):
    """Poll loop: one refresh per new checkpoint (drivers/run_ann_data_gen.py:663-702)."""
    dist = dist or Dist()
    last_checkpoint = args.last_checkpoint_dir
    ann_no, _, _ = get_latest_ann_data(args.output_dir)
    output_num = ann_no + 1
    logger.info("starting output number %d", output_num)
    if dist.rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        os.makedirs(args.cache_dir, exist_ok=True)
    training_positive_id, dev_positive_id = negatives.load_positive_ids(args.data_dir)

    while args.end_output_num == -1 or output_num <= args.end_output_num:
        # rank 0 looks at the training directory and every rank follows its decision: two ranks scanning on their own
        # can straddle a checkpoint commit, and then one of them enters the refresh's collectives while the other
        # sleeps and goes for the barrier
        next_checkpoint, latest_step_num = dist.broadcast_object(get_latest_checkpoint(args) if dist.rank == 0 else None)
        if args.only_keep_latest_embedding_file:
            latest_step_num = 0
        if next_checkpoint == last_checkpoint:
            time.sleep(getattr(args, "poll_seconds", 60))
        else:
            logger.info("start generate ann data number %d", output_num)
            logger.info("next checkpoint at " + next_checkpoint)
            generate_new_ann(args, output_num, next_checkpoint, training_positive_id, dev_positive_id,
                             latest_step_num, engine=engine, dist=dist)
            if args.inference:
                break
            logger.info("finished generating ann data number %d", output_num)
            output_num += 1
            last_checkpoint = next_checkpoint
        dist.barrier()


def get_arguments(argv=None):
    """Flags of drivers/run_ann_data_gen.py:443-627 (same names, defaults and meaning)."""
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", required=True, type=str)
    p.add_argument("--training_dir", required=True, type=str)
    p.add_argument("--init_model_dir", required=True, type=str)
    p.add_argument("--last_checkpoint_dir", default="", type=str)
    p.add_argument("--model_type", required=True, type=str)
    p.add_argument("--output_dir", required=True, type=str)
    p.add_argument("--cache_dir", required=True, type=str)
    p.add_argument("--end_output_num", default=-1, type=int)
    p.add_argument("--max_seq_length", default=128, type=int)
    p.add_argument("--max_query_length", default=64, type=int)
    p.add_argument("--max_doc_character", default=10000, type=int)
    p.add_argument("--per_gpu_eval_batch_size", default=128, type=int)
    p.add_argument("--ann_chunk_factor", default=5, type=int)
    p.add_argument("--topk_training", default=500, type=int)
    p.add_argument("--negative_sample", default=5, type=int)
    p.add_argument("--ann_measure_topk_mrr", default=False, action="store_true")
    p.add_argument("--only_keep_latest_embedding_file", default=False, action="store_true")
    p.add_argument("--no_cuda", action="store_true")
    p.add_argument("--local_rank", "--local-rank", type=int, default=-1)
    p.add_argument("--server_ip", type=str, default="")
    p.add_argument("--server_port", type=str, default="")
    p.add_argument("--inference", default=False, action="store_true")
    p.add_argument("--config_name", default="", type=str)
    p.add_argument("--tokenizer_name", default="", type=str)
    # additions (not in the reference)
    p.add_argument("--max_tokens", default=DRIVER_MAX_TOKENS, type=int,
                   help="tokens per encoder micro-batch (131,072: half as many launch ramps and partial last GEMM rounds as 65,536, +1.3 %; "
                        "the activation workspace scales with it: 11 GB for the two lanes of the default arithmetic)")
    p.add_argument("--seed", default=None, type=int, help="seed `random` before negative sampling")
    p.add_argument("--encoder_precision", default=None, choices=["fp16", "split", "fp32"],
                   help="encoder arithmetic (overrides the ANCE_ENCODER_* environment switches; with neither, split): split = fp16-pair operands on the fp16 matrix cores, fp32-grade like the reference's "
                        "own fp32 forward (2e-5 on the embeddings; reproduces the reference's negative ids up to proven near-ties); fp16 = "
                        "fp16 MFMA operands, the fast mode (3e-3, ~2.2 x the throughput, about half of the negative lists identical); "
                        "fp32 = fp32 operands (audit path)")
    return p.parse_args(argv)


def set_env(args):
    """One process per GPU over RCCL (backend name "nccl" on ROCm) -- drivers/run_ann_data_gen.py:630-660."""
    import torch
    if args.no_cuda:
        raise RuntimeError("this job has no CPU path: the HIP kernels are the product (--no_cuda is not supported)")
    if args.local_rank == -1 and "LOCAL_RANK" in os.environ:
        args.local_rank = int(os.environ["LOCAL_RANK"])
    if args.local_rank != -1:
        torch.cuda.set_device(args.local_rank)
        args.device = torch.device("cuda", args.local_rank)
        torch.distributed.init_process_group(backend="nccl")
        args.world_size = torch.distributed.get_world_size()
        args.rank = torch.distributed.get_rank()
    else:
        args.device = torch.device("cuda")
        args.world_size, args.rank = 1, 0
    args.n_gpu = 1
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s -   %(message)s", datefmt="%m/%d/%Y %H:%M:%S",
                        level=logging.INFO if args.local_rank in [-1, 0] else logging.WARN)


def main(argv=None):
    args = get_arguments(argv)
    set_env(args)
    if args.seed is not None:
        random.seed(args.seed)
    ann_data_gen(args)


if __name__ == "__main__":
    main()
