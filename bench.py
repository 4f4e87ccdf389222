#!/usr/bin/env python
"""Benchmark of the ANN refresh hot path on MI355X (BASELINE.json metric:
passages-encoded/sec + top-200 queries/sec, 8.8M x 768-d MS MARCO shape, 1/2/4/8 GPU).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input, timed per leg:
  encode leg : every rank encodes --encode-block passages (token ids ~ config 2 of SURVEY.md 8d:
               lengths lognormal(median 70, sigma .45) clipped to [8,128], random-init roberta-base
               rdot_nll, records already resident in HBM) -> value = passages/s over all ranks;
  search leg : --query-block queries, exact IP top-200 against the 8,841,823 x 768 fp32 corpus that
               is resident in HBM, sharded over the ranks (contiguous row blocks), per-shard lists
               exchanged by query owner over RCCL (all-to-all), merged, gathered on rank 0 -> queries/s
               (the refresh searches 100k-503k queries, in launch chunks of 32,768: that is the step).
    python bench.py --full   measures ONE real refresh end to end instead (caches on disk -> files), per phase.
Both legs: W untimed warm-up steps, then exactly K steps between barrier + synchronize on both
sides, MAX over ranks.  Rank 0 prints ONE JSON line.  `roofline` comes from HIP events the library
records around every kernel launch on the launch stream during the timed steps; `cpu_baseline` is
the oracle (a port of the reference CPU path) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_PASSAGES = 8841823
N_TRAIN_QUERIES = 502939
PEAK_F16_TF = 2500.0   # dense fp16/bf16 MFMA, MI355X_MICROARCH.md chip-level table
PEAK_F32_TF = 157.3    # fp32-input MFMA (= fp32 vector peak)
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--encode-block", type=int, default=16384, help="passages per rank per encode step")
    p.add_argument("--query-block", type=int, default=32768, help="queries per search step")
    p.add_argument("--n-passages", type=int, default=N_PASSAGES, help="rows of the resident corpus (all ranks)")
    p.add_argument("--seq-len", type=int, default=128)
    p.add_argument("--topk", type=int, default=200)
    p.add_argument("--max-tokens", type=int, default=65536)
    p.add_argument("--layers", type=int, default=12)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget per CPU-baseline leg")
    p.add_argument("--skip-search", action="store_true")
    p.add_argument("--skip-encode", action="store_true")
    p.add_argument("--skip-precise", action="store_true", help="skip the fp32-mode encoder sample")
    p.add_argument("--skip-encoder-like", action="store_true", help="skip the second search leg (encoder-like rows)")
    p.add_argument("--full", action="store_true",
                   help="measure ONE real refresh instead of the step benchmark: write the tokenised caches of "
                        "--n-passages passages / --full-queries train queries to --full-dir, then run "
                        "ance_amd.ann_data_gen.generate_new_ann on them (stream from disk, encode, search, host stage, "
                        "files) and print the wall time per phase")
    p.add_argument("--full-dir", type=str, default="/tmp/ance_full")
    p.add_argument("--full-queries", type=int, default=N_TRAIN_QUERIES)
    p.add_argument("--full-dev-queries", type=int, default=6980)
    p.add_argument("--negative-sample", type=int, default=20)
    p.add_argument("--encoder-precision", type=str, default=None, choices=["fp16", "split", "fp32"],
                   help="--full only: the encoder arithmetic of the refresh (ance_amd.ann_data_gen --encoder_precision)")
    return p.parse_args()


def write_synthetic_cache(path, n, L, median, sigma, lo, seed, block=1 << 20):
    """Reference cache format (utils/util.py:257-307 reads it; ance_amd.cache.TokenCache maps it), written in
    blocks so that the 4.56 GB passage file never sits in host memory.  Returns (seconds, mean length)."""
    t0 = time.perf_counter()
    rng = np.random.default_rng(seed)
    tot = 0
    with open(path, "wb") as f:
        for b0 in range(0, n, block):
            m = min(block, n - b0)
            lens = np.clip(np.rint(rng.lognormal(np.log(median), sigma, size=m)), lo, L).astype(np.int32)
            ids = rng.integers(3, 50265, size=(m, L), dtype=np.int32)
            ids[:, 0] = 0
            ids[np.arange(m), lens - 1] = 2
            ids[np.arange(L)[None, :] >= lens[:, None]] = 1
            rec = np.empty((m, 1 + L), dtype=np.int32)
            rec[:, 0] = lens.astype(">u4").view(np.int32)
            rec[:, 1:] = ids
            f.write(rec.tobytes())
            tot += int(lens.sum())
    with open(path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": int(n), "embedding_size": int(L)}, f)
    return time.perf_counter() - t0, tot / max(n, 1)


def full_refresh(a):
    """the path the reference actually runs (drivers/run_ann_data_gen.py:231-336 over
    utils/util.py:257-329), measured end to end on this box instead of extrapolated from one resident block."""
    import types
    import torch
    from safetensors.torch import save_file
    from ance_amd import ann_data_gen as adg
    from ance_amd import negatives
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        torch.distributed.init_process_group(backend=os.environ.get("ANCE_BENCH_BACKEND", "nccl"))
    d = a.full_dir
    data, ckpt, outd = os.path.join(d, "data"), os.path.join(d, "checkpoint-1"), os.path.join(d, "out")
    prep = {}
    if rank == 0:
        for sub in (data, ckpt, outd):
            os.makedirs(sub, exist_ok=True)
        prep["write_passages_s"], mean_p = write_synthetic_cache(os.path.join(data, "passages"), a.n_passages, a.seq_len, 70.0,
                                                                 0.45, 8, 1)
        prep["write_train_queries_s"], mean_q = write_synthetic_cache(os.path.join(data, "train-query"), a.full_queries, 64,
                                                                      9.0, 0.35, 4, 2)
        prep["write_dev_queries_s"], _ = write_synthetic_cache(os.path.join(data, "dev-query"), a.full_dev_queries, 64, 9.0,
                                                               0.35, 4, 3)
        prep["mean_passage_len"], prep["mean_query_len"] = mean_p, mean_q
        rng = np.random.default_rng(4)
        with open(os.path.join(data, "train-qrel.tsv"), "w") as f:
            pos = rng.integers(0, a.n_passages, size=a.full_queries)
            f.write("".join("%d\t%d\t1\n" % (q, p) for q, p in enumerate(pos.tolist())))
        with open(os.path.join(data, "dev-qrel.tsv"), "w") as f:
            pos = rng.integers(0, a.n_passages, size=a.full_dev_queries)
            f.write("".join("%d\t%d\t1\n" % (q, p) for q, p in enumerate(pos.tolist())))
        save_file({k: v.contiguous() for k, v in random_init_roberta_base(torch, a.layers, seed=0).items()},
                  os.path.join(ckpt, "model.safetensors"))
    dist = adg.Dist()
    dist.barrier()
    timings = {}
    args = types.SimpleNamespace(data_dir=data, output_dir=outd, cache_dir=outd, inference=False, topk_training=a.topk,
                                 negative_sample=a.negative_sample, ann_chunk_factor=1, ann_measure_topk_mrr=False,
                                 model_type="rdot_nll", max_seq_length=a.seq_len, max_query_length=64, device=dev,
                                 max_tokens=a.max_tokens, timings=timings, encoder_precision=a.encoder_precision)
    import logging
    import random
    logging.basicConfig(format="%(asctime)s %(name)s %(message)s", level=logging.INFO, stream=sys.stderr)
    random.seed(0)
    t0 = time.perf_counter()
    train_pos, dev_pos = negatives.load_positive_ids(data)
    t_qrels = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = adg.generate_new_ann(args, 0, ckpt + "/", train_pos, dev_pos, 1, dist=dist)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if rank == 0:
        lines = sum(1 for _ in open(os.path.join(outd, "ann_training_data_0")))
        enc_s = timings.get("encode_passages", 0.0)
        mode = a.encoder_precision or ("fp32" if os.environ.get("ANCE_ENCODER_PRECISE", "")[:1] == "1" else
                                       "split" if os.environ.get("ANCE_ENCODER_SPLIT", "")[:1] == "1" else "fp16")
        out = {"metric": "full_refresh_seconds", "value": wall, "unit": "s", "n_gpus": world, "higher_is_better": False,
               "dtype": {"fp16": "f16", "split": "f16 pairs (fp32-grade)", "fp32": "f32"}[mode], "encoder_precision": mode, "data": "synthetic",
               "config": {"workload": "one ANN refresh: %d passages (seq_len %d, streamed from %s) + %d train queries "
                                      "(ann_chunk_factor 1) + %d dev queries, top-%d, %d negatives, roberta-base rdot_nll "
                                      "random init" % (a.n_passages, a.seq_len, d, a.full_queries, a.full_dev_queries, a.topk,
                                                       a.negative_sample)},
               "phases_s": {k: round(v, 3) for k, v in timings.items()}, "load_qrels_s": round(t_qrels, 3),
               "prepare_s": {k: round(v, 3) for k, v in prep.items()},
               "passages_per_sec_measured": a.n_passages / enc_s if enc_s > 0 else None,
               "train_queries_per_sec_measured": a.full_queries / timings["search_train"] if timings.get("search_train") else None,
               "lines_written": lines, "dev_ndcg": res[0] if res else None}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()



def synthetic_records(rng, n, L):
    """Tokenised-cache rows [n, 1+L] int32 (big-endian length header, <s> ... </s>, pad = 1) with the
    length distribution of SURVEY.md 8d config 2.  Returns (records, lengths)."""
    lens = np.clip(np.rint(rng.lognormal(np.log(70.0), 0.45, size=n)), 8, L).astype(np.int32)
    ids = rng.integers(3, 50265, size=(n, L), dtype=np.int64).astype(np.int32)
    ids[:, 0] = 0
    ids[np.arange(n), lens - 1] = 2
    ids = np.where(np.arange(L)[None, :] < lens[:, None], ids, 1).astype(np.int32)
    rec = np.empty((n, 1 + L), dtype=np.int32)
    rec[:, 0] = lens.astype(">u4").view(np.int32)
    rec[:, 1:] = ids
    return rec, lens


def random_init_roberta_base(torch, n_layers, seed=0):
    """Random-init rdot_nll weights (normal std 0.02 / LayerNorm 1,0 / bias 0 -- the reference's
    _init_weights, model/models.py:31-36); there are no checkpoints offline."""
    g = torch.Generator().manual_seed(seed)
    H, I = 768, 3072
    sd = {}

    def lin(name, o, i):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * 0.02
        sd[name + ".bias"] = torch.zeros(o)

    def ln(name):
        sd[name + ".weight"] = torch.ones(H)
        sd[name + ".bias"] = torch.zeros(H)

    e = "roberta.embeddings."
    sd[e + "word_embeddings.weight"] = torch.randn(50265, H, generator=g) * 0.02
    sd[e + "position_embeddings.weight"] = torch.randn(514, H, generator=g) * 0.02
    sd[e + "token_type_embeddings.weight"] = torch.randn(1, H, generator=g) * 0.02
    ln(e + "LayerNorm")
    for i in range(n_layers):
        p = "roberta.encoder.layer.%d." % i
        lin(p + "attention.self.query", H, H)
        lin(p + "attention.self.key", H, H)
        lin(p + "attention.self.value", H, H)
        lin(p + "attention.output.dense", H, H)
        ln(p + "attention.output.LayerNorm")
        lin(p + "intermediate.dense", I, H)
        lin(p + "output.dense", H, I)
        ln(p + "output.LayerNorm")
    lin("embeddingHead", 768, H)
    ln("norm")
    return sd


CURRENT_ROUND = "r04"  # counters and kernel traces under profiles/ are quoted only when they were taken on this round's tree
KERNEL_OF_SPLIT = {"gemm_ffn1": "gemm256_split_kernel<9>", "gemm_qk": "gemm256_split_kernel<8>",
                   "gemm_attn_out": "gemm256_split_kernel<10>", "gemm_ffn2": "gemm256_split_kernel<10>"}
KERNEL_OF_FP32 = {"gemm_ffn1": "gemm32_kernel<1>", "gemm_qk": "gemm32_kernel<0>", "gemm_attn_out": "gemm32_kernel<2>",
                  "gemm_ffn2": "gemm32_kernel<2>"}
KERNEL_OF = {"gemm_ffn1": "gemm256_f16_desc_kernel<6>", "gemm_qk": "gemm256_f16_desc_kernel<5>", "gemm_vt": "gemm256_f16_desc_kernel<7>",
             "gemm_attn_out": "gemm256_f16_desc_kernel<4>", "gemm_ffn2": "gemm256_f16_desc_kernel<4>"}


def trace_avg_ns(csv_name, needle):
    """AverageNs of the kernel whose name contains `needle` in a committed rocprofv3 --stats CSV (None if absent)."""
    import csv
    try:
        with open(os.path.join(ROOT, "profiles", csv_name)) as f:
            for row in csv.DictReader(f):
                if needle in row["Name"]:
                    return float(row["AverageNs"])
    except Exception:
        pass
    return None


def pmc_traffic(leg, kernel):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 for wide
    reads on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md HBM section).  PMC cannot be collected from
    inside this process; the numbers are read from profiles/pmc_traffic.json, which
    scripts/gpu_pmc.sh regenerates on the same workload (null if the file has no entry)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get(leg, {}).get("gemm_res" if kernel in ("gemm_attn_out", "gemm_ffn2") else kernel)
        return ent if ent and str(ent.get("round", "")).startswith(CURRENT_ROUND) else None  # never quote another round's counters
    except Exception:
        return None


def timed_steps(fn, steps, warmup, dist_on, torch):
    for _ in range(warmup):
        fn()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_encode_baseline(seq_len, seconds, layers):
    """Reference CPU path, encode: what RobertaDot_NLL_LN.body_emb runs (model/models.py:149-157) -- transformers' own
    RobertaModel forward (the third-party library the reference calls; it is installed on the GPU box, /root/reference is
    not) + the reference's head (first token -> Linear(768, 768) -> LayerNorm) restated in three lines; fp32, random init,
    batch 16 (the recipe's --per_gpu_eval_batch_size) and 128, padded to seq_len like the reference pads.  torch's intra-op
    pool degrades badly when every logical core of a large host joins a small matmul, so a short probe picks the best thread
    count (reported as `cores`) before the timed sample.  Falls back to the oracle's restatement of the same arithmetic
    (oracle/encoder_ref.py, pinned to the reference's classes by tests/golden) when transformers cannot be imported."""
    import torch
    from oracle import encoder_ref, synth
    ncpu = os.cpu_count() or 1
    rng = np.random.default_rng(1234)
    impl = None
    try:
        from transformers import RobertaConfig, RobertaModel
        cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1,
                            bos_token_id=0, eos_token_id=2, num_hidden_layers=layers)
        torch.manual_seed(0)
        hf = RobertaModel(cfg, add_pooling_layer=False).eval()
        head = torch.nn.Linear(768, 768)
        norm = torch.nn.LayerNorm(768)

        def forward(ids, mask):
            h = hf(input_ids=ids.long(), attention_mask=mask)[0]
            return norm(head(h[:, 0]))
        impl = "transformers %s RobertaModel (the library the reference calls) + the reference's head restated" % __import__("transformers").__version__
    except Exception:
        sd = encoder_ref.random_state_dict(seed=0, n_layers=layers)

        def forward(ids, mask):
            return encoder_ref.rdot_nll_ln_emb(sd, ids, mask, n_layers=layers)
        impl = "oracle/encoder_ref.py (torch restatement of the same forward)"

    def batch(bs):
        lens = synth.lognormal_lengths(rng, bs, 70, 0.45, 8, seq_len)
        return torch.from_numpy(synth.make_records(rng, bs, seq_len, lens)), encoder_ref.mask_from_lengths(lens, seq_len)

    best = None
    with torch.no_grad():
        for bs in (16, 128):
            ids, mask = batch(bs)
            for th in sorted({t for t in (16, 32, 64, 128, ncpu) if t <= ncpu}):
                torch.set_num_threads(th)
                forward(ids, mask)
                t0 = time.perf_counter()
                forward(ids, mask)
                rate = bs / (time.perf_counter() - t0)
                if best is None or rate > best[0]:
                    best = (rate, bs, th)
        _, bs, th = best
        torch.set_num_threads(th)
        ids, mask = batch(bs)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            forward(ids, mask)
            n += bs
        dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="passages/s", cores=th, kind="port", implementation=impl,
                sample="%d passages, batch %d x %d tokens (padded), fp32 torch CPU with %d of %d logical cores "
                       "(best of a 16..%d thread probe)" % (n, bs, seq_len, th, ncpu, ncpu))


def cpu_search_baseline(n_rows_total, k, seconds):
    """Reference CPU path, search: what faiss-cpu's IndexFlatIP.search does -- 4,096-query blocks of the score matrix by BLAS
    sgemm streamed into one k-heap per query under OpenMP (oracle/search_ref.py: flat_ip_topk_faisslike; SURVEY.md 8d:
    4,096 x 65,536 blocks) -- on a bounded slice, extrapolated linearly in corpus rows.  The reference pins faiss to 16
    threads (drivers/run_ann_data_gen.py:269); a short probe picks the BLAS thread count that wins on this host."""
    from oracle import search_ref, synth
    ncpu = os.cpu_count() or 1
    rng = np.random.default_rng(4321)
    n_s, nq_s = 262144, 4096
    x = synth.ln_rows(rng, n_s)
    q = synth.ln_rows(rng, nq_s)
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    best = None
    cands = sorted({t for t in (16, 32, 64, 128, ncpu) if t <= ncpu}) if threadpool_limits else [ncpu]
    for th in cands:
        ctx = threadpool_limits(limits=th, user_api="blas") if threadpool_limits else None
        try:
            search_ref.flat_ip_topk_faisslike(x[:16384], q[:1024], k, x_block=65536)
            t0 = time.perf_counter()
            search_ref.flat_ip_topk_faisslike(x[:65536], q, k, x_block=65536)
            rate = 1.0 / (time.perf_counter() - t0)
        finally:
            if ctx is not None:
                ctx.unregister() if hasattr(ctx, "unregister") else ctx.restore_original_limits()
        if best is None or rate > best[0]:
            best = (rate, th)
    th = best[1]
    ctx = threadpool_limits(limits=th, user_api="blas") if threadpool_limits else None
    try:
        done, t0 = 0, time.perf_counter()
        while True:
            search_ref.flat_ip_topk_faisslike(x, q, k, x_block=65536)
            done += nq_s
            if time.perf_counter() - t0 > seconds:
                break
        dt = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.unregister() if hasattr(ctx, "unregister") else ctx.restore_original_limits()
    qps_sample = done / dt
    return dict(value=qps_sample * n_s / n_rows_total, unit="queries/s", cores=th, kind="port",
                tflops=2.0 * done * n_s * 768 / dt / 1e12,
                sample="%d queries x %d rows in 4,096 x 65,536 blocks: BLAS sgemm (%d of %d logical cores, best of a probe) + one "
                       "top-%d heap per query under OpenMP (oracle/search_ref.py flat_ip_topk_faisslike = faiss-cpu "
                       "IndexFlatIP's algorithm; faiss itself is not installable here), scaled by rows to %d"
                       % (done, n_s, th, ncpu, k, n_rows_total))


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU, RCCL) under torch.distributed.run --
    the reference's own launch is `python -m torch.distributed.launch --nproc_per_node=N` (commands/run_ann_data_gen.sh:44-49,
    drivers/run_ann_data_gen.py:637-640).  Refuses to measure fewer GPUs than were asked for."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if a.gpus > have and os.environ.get("ANCE_BENCH_BACKEND", "nccl") == "nccl":
        sys.stderr.write("bench.py: --gpus %d but this node has %d visible GPU(s); RCCL needs one device per rank\n" % (a.gpus, have))
        sys.exit(2)
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    # torch and the oracle's C library share libgomp: spinning OpenMP workers fight the BLAS pool of the CPU baseline
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    a = parse()
    if a.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a)
    if a.full:
        return full_refresh(a)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: the launcher must start one rank per GPU\n" % (a.gpus, world))
        sys.exit(2)
    dist_on = world > 1
    # one rank per GPU; ANCE_BENCH_BACKEND=gloo lets two ranks share a GPU to exercise the N > 1 code path on a
    # one-GPU box (RCCL refuses duplicate devices) -- a functional check, not a measurement
    backend = os.environ.get("ANCE_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        if backend == "nccl":
            torch.distributed.init_process_group(backend="nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend=backend)
    from ance_amd import _lib
    from ance_amd import ann_data_gen as adg
    from ance_amd.cache import shard_range
    from ance_amd.encoder import ARCH_ROBERTA, Encoder
    dist = adg.Dist()
    eng = adg.HipEngine(dev)
    errors = {}
    out = {"metric": "passages_encoded_per_sec", "value": None, "unit": "passages/s", "n_gpus": world,
           "rccl_ranks": (torch.distributed.get_world_size() if dist_on else 1),
           "backend": (torch.distributed.get_backend() if dist_on else None),
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": "MS MARCO passage %d x 768-d, roberta-base rdot_nll FirstP seq_len=%d, encode + "
                                  "brute-force IP top-%d (BASELINE configs[1])" % (a.n_passages, a.seq_len, a.topk),
                      "encode_block_per_gpu": a.encode_block, "query_block": a.query_block, "layers": a.layers,
                      "parallelism": "dp%d (corpus rows sharded, top-k all-to-all by query owner + merge)" % world}}

    # ------------------------------------------------------------------------------ encode leg --
    if not a.skip_encode:
        try:
            sd = random_init_roberta_base(torch, a.layers, seed=0)
            enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=min(a.seq_len, 512),
                          max_tokens=a.max_tokens, device=dev)
            sd_for_probe = sd
            rng = np.random.default_rng(1234 + rank)
            rec, lens = synthetic_records(rng, a.encode_block, a.seq_len)
            rec_d = torch.from_numpy(rec).to(dev)
            emb = torch.empty((a.encode_block, 768), dtype=torch.float32, device=dev)
            flops_alg = float(sum(169869312.0 * t + 36864.0 * t * t + 1179648.0 for t in lens.astype(np.float64)))
            flops_pad = a.encode_block * (169869312.0 * a.seq_len + 36864.0 * a.seq_len ** 2 + 1179648.0)

            def step_enc():
                enc.encode_records(rec_d, h_lens=lens, out=emb)

            for _ in range(max(a.warmup, 1)):
                step_enc()
            torch.cuda.synchronize()
            dt = timed_steps(step_enc, a.steps, 0, dist_on, torch)
            # Roofline pass.  The product overlaps consecutive micro-batches on two internal streams, so
            # inside the timed region kernels of two micro-batches share the chip and a per-kernel
            # duration is not a property of that kernel.  The same K steps are therefore repeated on a
            # single-stream handle with the library's HIP events on (launch stream), kernels isolated.
            os.environ["ANCE_ENCODER_STREAMS"] = "1"
            enc1 = Encoder(sd_for_probe, ARCH_ROBERTA, "roberta.", True, max_seq_len=min(a.seq_len, 512),
                           max_tokens=a.max_tokens, device=dev)
            os.environ.pop("ANCE_ENCODER_STREAMS", None)
            enc1.encode_records(rec_d, h_lens=lens, out=emb)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            t1 = time.perf_counter()
            for _ in range(a.steps):
                enc1.encode_records(rec_d, h_lens=lens, out=emb)
            torch.cuda.synchronize()
            dt_iso = time.perf_counter() - t1
            prof = _lib.profile_read()
            _lib.profile_enable(False)
            del enc1
            pps = world * a.encode_block * a.steps / dt
            out["value"] = pps
            out["ms_per_step"] = 1e3 * dt / a.steps
            gemm_cats = ["gemm_qk", "gemm_vt", "gemm_attn_out", "gemm_ffn1", "gemm_ffn2"]
            dom = max(gemm_cats, key=lambda c: prof[c]["ms"])
            by_kernel = {c: dict(ms_per_launch=(v["ms"] / v["count"]) if v["count"] else None, launches=v["count"],
                                 total_ms=v["ms"], tflops=(v["work"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["work"] > 0 else None)
                         for c, v in prof.items() if v["count"]}
            ach = prof[dom]["work"] / (prof[dom]["ms"] * 1e-3) / 1e12 if prof[dom]["ms"] > 0 else None
            all_gemm_ms = sum(prof[c]["ms"] for c in gemm_cats)
            all_gemm_work = sum(prof[c]["work"] for c in gemm_cats)
            # the same fraction from the committed kernel trace of this round (profiles/): FLOPs of one launch / its average
            # duration in the single-stream rocprofv3 --stats CSV
            t_ns = trace_avg_ns("%s_rocprofv3_encode_single_stream_kernel_stats.csv" % CURRENT_ROUND, KERNEL_OF.get(dom, "?"))
            flop_launch = prof[dom]["work"] / max(prof[dom]["count"], 1)
            out["roofline"] = {"bound": "mfma", "kernel": "%s (%s)" % (KERNEL_OF.get(dom, "gemm256_f16_desc_kernel"), dom), "achieved": ach,
                               "peak": PEAK_F16_TF, "unit": "TFLOP/s", "frac": (ach / PEAK_F16_TF) if ach else None,
                               "frac_from_profiles": (flop_launch / (t_ns * 1e-9) / 1e12 / PEAK_F16_TF) if t_ns else None,
                               "traffic": pmc_traffic("encode", dom),
                               "timing": "HIP events on the launch stream, single-stream pass of the same %d steps "
                                         "(%.1f ms/step isolated vs %.1f ms/step overlapped)" % (a.steps, 1e3 * dt_iso / a.steps, 1e3 * dt / a.steps),
                               "all_gemm_tflops": all_gemm_work / (all_gemm_ms * 1e-3) / 1e12 if all_gemm_ms > 0 else None,
                               "by_kernel": by_kernel}
            out["encode"] = {"passages_per_sec": pps, "tokens_per_sec": pps * float(lens.mean()),
                             "mean_len": float(lens.mean()),
                             "algorithmic_tflops": world * flops_alg * a.steps / dt / 1e12,
                             "padded_equiv_tflops": world * flops_pad * a.steps / dt / 1e12,
                             "end_to_end_mfma_frac": world * flops_alg * a.steps / dt / 1e12 / (PEAK_F16_TF * world),
                             "hbm_min_bytes_per_passage": 4 + 4 * a.seq_len + 3072,
                             "full_corpus_seconds_est": N_PASSAGES / pps}
            # ---- the two fp32-grade modes beside the default, as first-class measurements: the same block, the same K
            # steps, their own roofline (single-stream pass with the library's HIP events), and how far the default's
            # fp16-operand embeddings are from each on the same records (the stated tolerance, measured live)
            # (single-GPU runs only: the multi-GPU runs are the driver's scaling curve of `value`, and the audit path would
            # double their length)
            if not a.skip_precise and world == 1:
                enc.encode_records(rec_d, h_lens=lens, out=emb)
                torch.cuda.synchronize()
                emb_default = emb.clone()
                modes = {}
                for label, env, kernels, peak in (("encode_split", "ANCE_ENCODER_SPLIT", KERNEL_OF_SPLIT, PEAK_F16_TF),
                                                  ("encode_fp32", "ANCE_ENCODER_PRECISE", KERNEL_OF_FP32, PEAK_F32_TF)):
                    os.environ[env] = "1"
                    try:
                        encp = Encoder(sd_for_probe, ARCH_ROBERTA, "roberta.", True, max_seq_len=min(a.seq_len, 512),
                                       max_tokens=a.max_tokens, device=dev)
                        os.environ["ANCE_ENCODER_STREAMS"] = "1"
                        encp1 = Encoder(sd_for_probe, ARCH_ROBERTA, "roberta.", True, max_seq_len=min(a.seq_len, 512),
                                        max_tokens=a.max_tokens, device=dev)
                    finally:
                        os.environ.pop(env, None)
                        os.environ.pop("ANCE_ENCODER_STREAMS", None)
                    embp = torch.empty_like(emb)

                    def step_mode():
                        encp.encode_records(rec_d, h_lens=lens, out=embp)

                    step_mode()
                    torch.cuda.synchronize()
                    # the same K steps as the default leg; the 9 x slower audit path is capped at 8 steps (2.2 s each) so that a
                    # driver run with a large K still finishes within minutes -- the JSON carries the count that was timed
                    steps_m = a.steps if label == "encode_split" else min(a.steps, 8)
                    dtm = timed_steps(step_mode, steps_m, 0, dist_on, torch)
                    n_iso = max(1, min(a.steps, 3))
                    encp1.encode_records(rec_d, h_lens=lens, out=embp)
                    torch.cuda.synchronize()
                    _lib.profile_enable(True)
                    for _ in range(n_iso):
                        encp1.encode_records(rec_d, h_lens=lens, out=embp)
                    torch.cuda.synchronize()
                    profm = _lib.profile_read()
                    _lib.profile_enable(False)
                    step_mode()
                    torch.cuda.synchronize()
                    gcats = [c for c in ("gemm_qk", "gemm_attn_out", "gemm_ffn1", "gemm_ffn2") if profm[c]["count"]]
                    domm = max(gcats, key=lambda c: profm[c]["ms"])
                    alg = profm[domm]["work"] / (profm[domm]["ms"] * 1e-3) / 1e12
                    passes = 3.0 if label == "encode_split" else 1.0  # MFMA passes per algorithmic product
                    t_ns = trace_avg_ns("%s_rocprofv3_%s_single_stream_kernel_stats.csv" % (CURRENT_ROUND, label), kernels.get(domm, "?"))
                    fl = profm[domm]["work"] / max(profm[domm]["count"], 1)
                    diff = (emb_default - embp).abs()
                    pps_m = world * a.encode_block * steps_m / dtm
                    modes[label] = {
                        "value": pps_m, "unit": "passages/s", "ms_per_step": 1e3 * dtm / steps_m, "steps": steps_m,
                        "block": a.encode_block,
                        "algorithmic_tflops": world * flops_alg * steps_m / dtm / 1e12,
                        "max_abs_vs_default": float(diff.max().item()), "mean_abs_vs_default": float(diff.mean().item()),
                        "roofline": {"bound": "mfma", "kernel": "%s (%s)" % (kernels.get(domm, "?"), domm),
                                     "achieved": alg * passes, "algorithmic": alg, "mfma_passes_per_product": passes,
                                     "peak": peak, "unit": "TFLOP/s", "frac": alg * passes / peak,
                                     "frac_from_profiles": (fl * passes / (t_ns * 1e-9) / 1e12 / peak) if t_ns else None,
                                     "timing": "HIP events on the launch stream, single-stream pass of %d steps" % n_iso,
                                     "by_kernel": {c: dict(ms_per_launch=v["ms"] / v["count"], launches=v["count"],
                                                           algorithmic_tflops=(v["work"] / (v["ms"] * 1e-3) / 1e12) if v["work"] > 0 and v["ms"] > 0 else None)
                                                   for c, v in profm.items() if v["count"]}}}
                    if label == "encode_split":
                        modes[label]["arithmetic"] = ("fp16 (hi, lo') pair operands, three fp16 MFMA passes per product, fp32 accumulation, fp32 "
                                                      "softmax, exact erf GELU: fp32-grade (stated 2e-5); algorithmic TF above the 157.3 TF "
                                                      "fp32-MFMA peak means it beats what fp32 operands could reach")
                        modes[label]["algorithmic_vs_fp32_mfma_peak"] = alg / PEAK_F32_TF
                        emb_split = embp.clone()
                    else:
                        modes[label]["arithmetic"] = "fp32 operands on v_mfma_f32_32x32x2_f32, exact erf GELU, fp32 softmax (the reference's arithmetic; the audit path)"
                        if "encode_split" in modes:
                            modes["encode_split"]["max_abs_vs_fp32_mode"] = float((emb_split - embp).abs().max().item())
                    del encp, encp1, embp
                    torch.cuda.empty_cache()
                out.update(modes)
                out["encoder_modes"] = {
                    "default": {"operands": "fp16 MFMA operands, fp32 accumulation, LayerNorm folded into the GEMMs, residual stream as fp16 (hi, lo) pairs",
                                "passages_per_sec": pps},
                    "split (ANCE_ENCODER_SPLIT=1)": {"passages_per_sec": modes["encode_split"]["value"]},
                    "fp32 (ANCE_ENCODER_PRECISE=1)": {"passages_per_sec": modes["encode_fp32"]["value"]},
                    "max_abs_default_vs_fp32": modes["encode_fp32"]["max_abs_vs_default"],
                    "mean_abs_default_vs_fp32": modes["encode_fp32"]["mean_abs_vs_default"]}
                del emb_default
            del enc, rec_d, emb, sd_for_probe
            torch.cuda.empty_cache()
        except Exception as e:  # keep going: a bench line with the other leg is still informative
            import traceback
            traceback.print_exc()
            errors["encode"] = repr(e)

    # ------------------------------------------------------------------------------ search leg --
    if not a.skip_search:
        try:
            r0, r1 = shard_range(a.n_passages, rank, world)
            n_loc = r1 - r0
            g = torch.Generator(device=dev).manual_seed(4321 + rank)
            x = torch.empty((n_loc, 768), dtype=torch.float32, device=dev)
            for b0 in range(0, n_loc, 1 << 20):
                b1 = min(b0 + (1 << 20), n_loc)
                z = torch.randn((b1 - b0, 768), generator=g, device=dev)
                x[b0:b1] = torch.nn.functional.layer_norm(z, (768,))
            gq = torch.Generator(device=dev).manual_seed(99)
            q = torch.nn.functional.layer_norm(torch.randn((a.query_block, 768), generator=gq, device=dev), (768,))
            res = {}

            def step_search():
                res["DI"] = adg.sharded_search(eng, dist, x, r0, q, a.topk)

            for _ in range(max(a.warmup, 1)):
                step_search()
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            dt = timed_steps(step_search, a.steps, 0, dist_on, torch)
            prof = _lib.profile_read()
            _lib.profile_enable(False)
            qps = a.query_block * a.steps / dt
            scan, resc, fin = prof["ip_topk_scan"], prof["ip_topk_rescore"], prof["topk_finalize"]
            n_scan = max(scan["count"], 1)  # (the device-side conditional redo launches of the fast path are not profiled)
            ach = scan["work"] / (scan["ms"] * 1e-3) / 1e12 if scan["ms"] > 0 else None
            D, I = res["DI"]  # rank 0 holds the merged lists
            ok = None
            if rank == 0:
                ok = bool((D[:, 1:] <= D[:, :-1]).all().item()) and bool((I >= 0).all().item())
            # the search image (fp16 rows, duplicate classes) is built once per refresh, not per step: time it alone
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            from ance_amd.index import FlatIPIndex
            probe = FlatIPIndex(768, device=dev)
            probe.add(x)
            probe._search_image(_lib.lib(), x)
            torch.cuda.synchronize()
            build_ms = 1e3 * (time.perf_counter() - t_b)
            del probe
            traffic = pmc_traffic("search", "ip_topk_fast")
            # what the counters say limits the filter kernel (profiles/pmc_traffic.json, scripts/gpu_pmc.sh): the MFMA pipe
            # when it is busy most of the cycles, else the memory side feeding it
            bound = "mfma"
            if traffic and traffic.get("cycles") and (traffic["cycles"].get("mfma_busy_frac") or 1.0) < 0.5:
                bound = "mfma (pipe busy %.0f %% of the cycles at %.2f GHz: the rest is LDS-DMA issue, barriers and the filter between " \
                        "corpus tiles; L2-fill traffic %.2f TB per launch)" % (100 * traffic["cycles"]["mfma_busy_frac"],
                                                                                traffic["cycles"]["clock_ghz"], traffic["fetch_bytes_x2"] / 1e12)
            out["search"] = {"metric": "top%d_queries_per_sec" % a.topk, "value": qps, "unit": "queries/s",
                             "ms_per_step": 1e3 * dt / a.steps, "dtype": "f16 filter + f32 exact re-score (results bit-identical to the f32 scan)", "scaling": "strong (corpus sharded)",
                             "rows_total": a.n_passages, "rows_per_gpu": n_loc, "sorted_and_valid": ok,
                             "full_train_queries_seconds_est": N_TRAIN_QUERIES / qps,
                             "search_image_build_ms": build_ms,
                             "roofline": {"bound": bound,
                                          "kernel": "ip_topk_fast_kernel (fp16 MFMA 32x32x16 filter; algorithmic FLOPs = 2 nq n d = "
                                                    "1,536 per query-row pair)",
                                          "achieved": ach, "peak": PEAK_F16_TF, "unit": "TFLOP/s",
                                          "frac": (ach / PEAK_F16_TF) if ach else None,
                                          "frac_from_profiles": (lambda t_: (2.0 * a.query_block * n_loc * 768 / (t_ * 1e-9) / 1e12 / PEAK_F16_TF)
                                                                 if t_ and world == 1 else None)(
                                              trace_avg_ns("%s_rocprofv3_search_kernel_stats.csv" % CURRENT_ROUND, "ip_topk_fast_kernel<false, false>")),
                                          "traffic": traffic,
                                          "ms_per_launch": scan["ms"] / n_scan,
                                          "rescore_ms_per_launch": resc["ms"] / max(resc["count"], 1),
                                          "rescore_kernel": "rescore_kernel (exact fp32 fmaf chains of the ~k + 66 band rows per query, shared between its split lists; "
                                                            "HBM-bound gather of 3 KB rows)",
                                          "finalize_ms_per_launch": fin["ms"] / max(fin["count"], 1),
                                          "whole_step_tflops": 2.0 * a.query_block * a.n_passages * 768 * a.steps / dt / 1e12,
                                          "hbm_read_gbs_min": ((n_loc * 768 * 2.0) / (scan["ms"] / n_scan * 1e-3) / 1e9)
                                          if scan["ms"] > 0 else None}}
            # ---- the same leg on ENCODER-LIKE rows: one large common component + small deviations (what a dual encoder
            # really emits: random-init roberta-base gives cosine 0.99 between any two passages) -- the filter's hard case:
            # the error slack is a larger share of the score spread, and the query-mean bias build of the kernel runs
            if not a.skip_encoder_like:
                del res["DI"], D, I
                gc_ = torch.Generator(device=dev).manual_seed(777)
                c = torch.randn((768,), generator=gc_, device=dev)
                c = c / c.norm() * (768.0 ** 0.5)
                for b0 in range(0, n_loc, 1 << 20):
                    b1 = min(b0 + (1 << 20), n_loc)
                    x[b0:b1] = c[None, :] + 0.12 * torch.randn((b1 - b0, 768), generator=g, device=dev)
                q.copy_(c[None, :] + 0.12 * torch.randn((a.query_block, 768), generator=gq, device=dev))
                x.add_(0.0)  # bumps the tensor's version counter: the engine rebuilds its search image
                for _ in range(max(a.warmup, 1)):
                    step_search()
                torch.cuda.synchronize()
                _lib.profile_enable(True)
                dt2 = timed_steps(step_search, a.steps, 0, dist_on, torch)
                prof2 = _lib.profile_read()
                _lib.profile_enable(False)
                scan2 = prof2["ip_topk_scan"]
                D2, _ = res["DI"]
                out["search"]["encoder_like"] = {
                    "rows": "common component (norm sqrt 768) + 0.12 N(0,1) per row and per query (tests/test_gpu_search.py:_encoder_like)",
                    "value": a.query_block * a.steps / dt2, "unit": "queries/s", "ms_per_step": 1e3 * dt2 / a.steps,
                    "filter_ms_per_launch": scan2["ms"] / max(scan2["count"], 1),
                    "filter_frac_of_peak": (scan2["work"] / (scan2["ms"] * 1e-3) / 1e12 / PEAK_F16_TF) if scan2["ms"] > 0 else None,
                    "rescore_ms_per_launch": prof2["ip_topk_rescore"]["ms"] / max(prof2["ip_topk_rescore"]["count"], 1),
                    "sorted": bool((D2[:, 1:] <= D2[:, :-1]).all().item()) if rank == 0 else None,
                    "relative_to_layernorm_rows": (a.query_block * a.steps / dt2) / qps}
            del x, q
        except Exception as e:
            import traceback
            traceback.print_exc()
            errors["search"] = repr(e)

    # ---------------------------------------------------------------------------- CPU baseline --
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            if not a.skip_encode:
                out["cpu_baseline"] = cpu_encode_baseline(a.seq_len, a.cpu_seconds, a.layers)
            if not a.skip_search and "search" in out:
                out["search"]["cpu_baseline"] = cpu_search_baseline(a.n_passages, a.topk, a.cpu_seconds)
        except Exception as e:
            errors["cpu_baseline"] = repr(e)
    try:  # measured by tests/test_gpu_retrieval.py on an MI355X (fp16-operand encoder vs the fp32 reference arithmetic)
        src = next(n for n in ("r04_retrieval_agreement.json", "r03_retrieval_agreement.json", "r02_retrieval_agreement.json")
                   if os.path.exists(os.path.join(ROOT, "profiles", n)))
        with open(os.path.join(ROOT, "profiles", src)) as f:
            ra = json.load(f)
        keys = ("n_passages", "n_queries", "layers", "k", "max_abs_passage", "recall_at_200", "identical_top1",
                "first_20_negatives_overlap", "identical_first_20_negatives")
        out["retrieval_agreement"] = {k_: ra[k_] for k_ in keys}
        if "precise_mode" in ra:
            out["retrieval_agreement"]["fp32_mode"] = {k_: ra["precise_mode"][k_] for k_ in keys}
        if "split_mode" in ra:
            out["retrieval_agreement"]["split_mode"] = {k_: ra["split_mode"][k_] for k_ in keys}
        out["retrieval_agreement"]["source"] = "profiles/%s (tests/test_gpu_retrieval.py: both encoder modes against the fp32 oracle)" % src
    except Exception:
        pass
    if errors:
        out["errors"] = errors
    if dist_on:
        torch.distributed.barrier()
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
