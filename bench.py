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
the oracle (a port of the reference CPU path) timed on this box's host cores on a bounded sample;
`power` (and `search.power`) are the board's socket power and shader clock sampled from the amdgpu
hwmon files while the timed steps run -- reported only: both matrix-core encoder modes sit at the cap.
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
    p.add_argument("--encode-block", type=int, default=16384,
                   help="passages per rank per encode step (one encode call; the refresh job itself encodes 65,536-record blocks at "
                        "seq_len 128: every call ends with the encoder's two internal streams joining and one partial micro-batch)")
    p.add_argument("--query-block", type=int, default=32768, help="queries per search step")
    p.add_argument("--n-passages", type=int, default=N_PASSAGES, help="rows of the resident corpus (all ranks)")
    p.add_argument("--seq-len", type=int, default=128)
    p.add_argument("--topk", type=int, default=200)
    p.add_argument("--max-tokens", type=int, default=131072, help="tokens per encoder micro-batch (the job's --max_tokens default)")
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
    p.add_argument("--skip-mfma-probe", action="store_true",
                   help="do not run tools/power_probe (the sustained MFMA rate of this board at its power cap, ~12 s)")
    p.add_argument("--skip-other-configs", action="store_true",
                   help="skip the encode legs of BASELINE configs[2..4] (L=512, MaxP 4x512, DPR BERT L=256)")
    p.add_argument("--other-config-steps", type=int, default=3, help="timed steps of each other-config leg (at most 6)")
    p.add_argument("--other-config-tokens", type=int, default=1 << 20, help="real tokens per step of each other-config leg")
    p.add_argument("--exact-check-queries", type=int, default=64,
                   help="queries of the search step re-run through the independent fp32 scan at full corpus size (0: skip)")
    p.add_argument("--skip-slice", action="store_true", help="skip the whole-refresh slice of the default run")
    p.add_argument("--slice-passages", type=int, default=500000)
    p.add_argument("--slice-queries", type=int, default=32768)
    p.add_argument("--slice-dev-queries", type=int, default=6980)
    p.add_argument("--slice-dir", type=str, default="/tmp/ance_slice")
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


def run_refresh(a, d, n_passages, n_train, n_dev, dist, dev, rank, precision):
    """Write the tokenised caches of a synthetic collection to ``d`` (rank 0), then run ONE refresh on them with the product's
    own job function (ance_amd.ann_data_gen.generate_new_ann: stream from disk, encode, search, host stage, files).
    Returns (wall seconds of the refresh, per-phase seconds, preparation seconds, lines written, dev NDCG@10)."""
    import types
    import torch
    from safetensors.torch import save_file
    from ance_amd import ann_data_gen as adg
    from ance_amd import negatives
    data, ckpt, outd = os.path.join(d, "data"), os.path.join(d, "checkpoint-1"), os.path.join(d, "out")
    prep = {}
    if rank == 0:
        for sub in (data, ckpt, outd):
            os.makedirs(sub, exist_ok=True)
        prep["write_passages_s"], mean_p = write_synthetic_cache(os.path.join(data, "passages"), n_passages, a.seq_len, 70.0,
                                                                 0.45, 8, 1)
        prep["write_train_queries_s"], mean_q = write_synthetic_cache(os.path.join(data, "train-query"), n_train, 64, 9.0, 0.35, 4, 2)
        prep["write_dev_queries_s"], _ = write_synthetic_cache(os.path.join(data, "dev-query"), n_dev, 64, 9.0, 0.35, 4, 3)
        prep["mean_passage_len"], prep["mean_query_len"] = mean_p, mean_q
        rng = np.random.default_rng(4)
        with open(os.path.join(data, "train-qrel.tsv"), "w") as f:
            pos = rng.integers(0, n_passages, size=n_train)
            f.write("".join("%d\t%d\t1\n" % (q, p) for q, p in enumerate(pos.tolist())))
        with open(os.path.join(data, "dev-qrel.tsv"), "w") as f:
            pos = rng.integers(0, n_passages, size=n_dev)
            f.write("".join("%d\t%d\t1\n" % (q, p) for q, p in enumerate(pos.tolist())))
        save_file({k: v.contiguous() for k, v in random_init_roberta_base(torch, a.layers, seed=0).items()},
                  os.path.join(ckpt, "model.safetensors"))
    dist.barrier()
    timings = {}
    args = types.SimpleNamespace(data_dir=data, output_dir=outd, cache_dir=outd, inference=False, topk_training=a.topk,
                                 negative_sample=a.negative_sample, ann_chunk_factor=1, ann_measure_topk_mrr=False,
                                 model_type="rdot_nll", max_seq_length=a.seq_len, max_query_length=64, device=dev,
                                 max_tokens=a.max_tokens, timings=timings, encoder_precision=precision)
    import random
    random.seed(0)
    t0 = time.perf_counter()
    train_pos, dev_pos = negatives.load_positive_ids(data)
    prep["load_qrels_s"] = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = adg.generate_new_ann(args, 0, ckpt + "/", train_pos, dev_pos, 1, dist=dist)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    lines = sum(1 for _ in open(os.path.join(outd, "ann_training_data_0"))) if rank == 0 else None
    return wall, timings, prep, lines, (res[0] if res else None)


def full_refresh(a):
    """the path the reference actually runs (drivers/run_ann_data_gen.py:231-336 over
    utils/util.py:257-329), measured end to end on this box instead of extrapolated from one resident block."""
    import torch
    from ance_amd import ann_data_gen as adg
    from ance_amd.encoder import precision_from_env
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        torch.distributed.init_process_group(backend=os.environ.get("ANCE_BENCH_BACKEND", "nccl"))
    import logging
    logging.basicConfig(format="%(asctime)s %(name)s %(message)s", level=logging.INFO, stream=sys.stderr)
    d = a.full_dir
    mode = a.encoder_precision or precision_from_env()
    wall, timings, prep, lines, ndcg = run_refresh(a, d, a.n_passages, a.full_queries, a.full_dev_queries, adg.Dist(), dev, rank, mode)
    if rank == 0:
        enc_s = timings.get("encode_passages", 0.0)
        out = {"metric": "full_refresh_seconds", "value": wall, "unit": "s", "n_gpus": world, "higher_is_better": False,
               "dtype": DTYPE_OF[mode], "encoder_precision": mode, "data": "synthetic",
               "config": {"workload": "one ANN refresh: %d passages (seq_len %d, streamed from %s) + %d train queries "
                                      "(ann_chunk_factor 1) + %d dev queries, top-%d, %d negatives, roberta-base rdot_nll "
                                      "random init" % (a.n_passages, a.seq_len, d, a.full_queries, a.full_dev_queries, a.topk,
                                                       a.negative_sample)},
               "phases_s": {k: round(v, 3) for k, v in timings.items()},
               "prepare_s": {k: round(v, 3) for k, v in prep.items()},
               "passages_per_sec_measured": a.n_passages / enc_s if enc_s > 0 else None,
               "train_queries_per_sec_measured": a.full_queries / timings["search_train"] if timings.get("search_train") else None,
               "lines_written": lines, "dev_ndcg": ndcg}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def refresh_slice(a, dist, dev, pps_step):
    """A whole refresh INSIDE the default run, so that the streaming path (R5-R9: tokenised cache on disk ->
    pinned ring -> HBM -> encoder) and the host stages (R15-R18: chunking, dev NDCG, negatives, writers) sit inside a time the
    driver observes: a --slice-passages collection in the headline arithmetic, per-phase seconds, and the slice's own
    passages/s beside the step benchmark's."""
    import shutil
    d = a.slice_dir
    shutil.rmtree(d, ignore_errors=True)
    try:
        wall, timings, prep, lines, ndcg = run_refresh(a, d, a.slice_passages, a.slice_queries, a.slice_dev_queries, dist, dev, 0,
                                                       HEADLINE_MODE)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    enc_s = timings.get("encode_passages", 0.0)
    pps = a.slice_passages / enc_s if enc_s > 0 else None
    return {"workload": "one ANN refresh of %d passages (seq_len %d) + %d train + %d dev queries, top-%d, %d negatives, caches "
                        "streamed from %s, ance_amd.ann_data_gen.generate_new_ann, --encoder_precision %s"
                        % (a.slice_passages, a.seq_len, a.slice_queries, a.slice_dev_queries, a.topk, a.negative_sample, d, HEADLINE_MODE),
            "wall_s": wall, "phases_s": {k: round(v, 3) for k, v in timings.items()},
            "prepare_s": {k: round(v, 3) for k, v in prep.items()},
            "passages_per_sec": pps, "passages_per_sec_vs_step_benchmark": (pps / pps_step) if pps and pps_step else None,
            "train_queries_per_sec": a.slice_queries / timings["search_train"] if timings.get("search_train") else None,
            "lines_written": lines, "dev_ndcg": ndcg, "measured_in_this_run": True}


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


CURRENT_ROUND = "r06"  # counters and kernel traces under profiles/ are quoted only when they were taken on this round's tree
HEADLINE_MODE = "split"  # the library's default arithmetic: fp32-grade, the precision the reference computes in
DTYPE_OF = {"split": "f16 pair operands on the fp16 MFMA, f32 accumulate (fp32-grade: <= 2e-5 from the reference's fp32 forward)",
            "fp16": "f16", "fp32": "f32"}
ARITHMETIC_OF = {
    "split": "every GEMM operand an fp16 (hi, lo) pair, three fp16 MFMAs per k-step (hi hi + lo hi + hi lo) from four operand tiles staged "
             "once, fp32 accumulation, fp32 softmax, erf-GELU to fp32 grade (polynomial erfc), fp32 head: fp32-grade (stated 2e-5, measured "
             "7e-6 at 12 layers against the reference's own class)",
    "fp16": "fp16 MFMA operands, fp32 accumulation, LayerNorm folded into the GEMMs, residual stream as fp16 (hi, lo) pairs (stated 5e-3)",
    "fp32": "fp32 operands on v_mfma_f32_32x32x2_f32, exact erf GELU, fp32 softmax (the reference's arithmetic; the audit path)"}
# (round 6: QKV and FFN1 run the persistent streaming kernel, the two RESLN GEMMs the launch-per-tile kernel: csrc/gemm256_f16.hip)
KERNEL_OF_SPLIT = {"gemm_ffn1": "gemm256_split_stream_kernel<9>", "gemm_qk": "gemm256_split_stream_kernel<8>",
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
        if not (ent and str(ent.get("round", "")).startswith(CURRENT_ROUND)):
            return None  # never quote another round's counters
        return dict(ent, measured_in_this_run=False, source="profiles/pmc_traffic.json (scripts/gpu_pmc.sh, rocprofv3 --pmc passes)")
    except Exception:
        return None


class PowerSampler:
    """Socket power and shader clock of the device while a timed region runs (a thread reading the amdgpu hwmon files every 0.25 s;
    `rocm-smi --json` when they are not readable): the split and fp16 encoder modes run AT the board's power cap, which is what
    sets their clock.  Reported, never used; None when the platform offers neither source."""

    def __init__(self, torch, dev_index, hwmon_dirs=None):
        import glob
        self.files, self.samples, self.stop, self.thread, self.index = None, [], False, None, dev_index
        dirs = list(hwmon_dirs or [])
        if not dirs:
            try:  # the hwmon directory of THIS device (a box lists every card of the node under /sys/class/drm)
                p = torch.cuda.get_device_properties(dev_index)
                bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
                dirs = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
            except Exception:
                dirs = []
        dirs = dirs or sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        for d in dirs:
            pw = [f for f in (d + "/power1_average", d + "/power1_input") if os.path.exists(f)]
            if pw:
                self.files = dict(power=pw[0], sclk=d + "/freq1_input", cap=d + "/power1_cap")
                break

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _smi(self):
        import subprocess
        try:
            out = json.loads(subprocess.run(["rocm-smi", "-d", str(self.index), "--showpower", "--showclocks", "--json"],
                                            capture_output=True, text=True, timeout=5).stdout)
            card = next(iter(out.values()))
            pw = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
            sc = next((float(v.strip("()Mhz")) for k, v in card.items() if k.startswith("sclk clock speed")), None)
            return pw, sc
        except Exception:
            return None, None

    def _loop(self):
        while not self.stop:
            if self.files:
                pw, sc = self._read(self.files["power"]), self._read(self.files["sclk"])
                pw, sc = (pw / 1e6 if pw is not None else None), (sc / 1e6 if sc is not None else None)
            else:
                pw, sc = self._smi()
            if pw is not None:
                self.samples.append((pw, sc))
            time.sleep(0.25)

    def __enter__(self):
        import threading
        self.thread = threading.Thread(target=self._loop, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        self.thread.join(timeout=10)

    def report(self):
        if not self.samples:
            return None
        pw = [x[0] for x in self.samples]
        sc = [x[1] for x in self.samples if x[1] is not None]
        cap = self._read(self.files["cap"]) if self.files else None
        return {"socket_power_w": {"min": min(pw), "mean": sum(pw) / len(pw), "max": max(pw)},
                "sclk_mhz": {"min": min(sc), "mean": sum(sc) / len(sc), "max": max(sc)} if sc else None,
                "power_cap_w": cap / 1e6 if cap else None, "samples": len(pw),
                "source": "amdgpu hwmon (sysfs), 0.25 s period, over the timed steps" if self.files else "rocm-smi --json over the timed steps",
                "measured_in_this_run": True}


def sustained_mfma_rate(seconds=2.0):
    """tools/power_probe (built by __graft_entry__.build): the fp16 MFMA rate THIS board sustains -- 8 waves per CU, the split GEMM's
    24 MFMAs per step -- bare, on zero operands, with the main loop's LDS fragment reads, with LDS-DMA traffic on top; each with the
    socket power and shader clock it ran at.  None when the tool is not there."""
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "power_probe")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, str(seconds)], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    res = {}
    for m in re.finditer(r"^(\S+)\s+([0-9.]+) TF executed.*?power\s+([0-9.-]+) W\s+sclk\s+([0-9.-]+) MHz", out, flags=re.M):
        res[m.group(1)] = {"tflops": float(m.group(2)), "socket_power_w": float(m.group(3)), "sclk_mhz": float(m.group(4))}
    return res or None


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


def search_exact_check(torch, x, q, D, I, k, n_queries, n_host=8):
    """the lists the timed search step returned, compared at the FULL corpus size with an independent path --
    ``n_queries`` of the step's queries re-run through the fp32-MFMA scan alone (``ance_ip_topk_scan``: no fp16 image, no
    filter, no re-scoring; the kernel tests/test_gpu_search.py pins bit-exactly to oracle/ip_topk_ref.c), ids AND scores
    required bit-identical; for ``n_host`` of them the 200 reported scores are also recomputed on the host with the oracle's
    fmaf chain (the oracle is the checker here, outside every timed region).  Matches drivers/run_ann_data_gen.py:269-276,303:
    faiss' IndexFlatIP is exact at any size."""
    from ance_amd.index import FlatIPIndex
    nq = q.shape[0]
    sel = torch.unique(torch.linspace(0, nq - 1, min(n_queries, nq), device=q.device).round().long())
    t0 = time.perf_counter()
    idx = FlatIPIndex(x.shape[1], device=x.device)
    idx.add(x)
    Ds, Is = idx.search_device(q[sel].contiguous(), k, exact_scan=True)
    same_i = bool(torch.equal(Is, I[sel]))
    same_d = bool(torch.equal(Ds.view(torch.int32), D[sel].view(torch.int32)))
    torch.cuda.synchronize()
    scan_s = time.perf_counter() - t0
    host_ok, host_n = None, 0
    try:
        from oracle import search_ref
        host_ok = True
        for j in sel[:n_host].tolist():
            rows = x[I[j]].cpu().numpy()
            S = search_ref.ip_scores_chain(rows, q[j:j + 1].cpu().numpy())[0]
            host_ok = host_ok and bool(np.array_equal(S, D[j].cpu().numpy()))
            host_n += 1
    except Exception as e:  # the oracle's C library is test infrastructure: its absence does not void the device comparison
        host_ok = "oracle unavailable: %r" % (e,)
    return {"queries": int(sel.numel()), "rows": int(x.shape[0]), "k": k, "identical": same_i and same_d,
            "identical_ids": same_i, "identical_scores_bitwise": same_d,
            "against": "ance_ip_topk_scan (fp32-MFMA scan alone, csrc/ip_topk.hip) on the same resident rows",
            "host_chain_rescored_queries": host_n, "host_chain_scores_identical": host_ok,
            "scan_seconds": scan_s, "measured_in_this_run": True}


def synthetic_records_of(rng, lens, L, first, last, pad, lo, hi):
    """Tokenised-cache rows [n, 1+L] for given lengths and special ids (RoBERTa: 0 / 2 / 1, BERT: 101 / 102 / 0)."""
    n = len(lens)
    ids = rng.integers(lo, hi, size=(n, L), dtype=np.int64).astype(np.int32)
    ids[:, 0] = first
    ids[np.arange(n), lens - 1] = last
    ids = np.where(np.arange(L)[None, :] < lens[:, None], ids, pad).astype(np.int32)
    rec = np.empty((n, 1 + L), dtype=np.int32)
    rec[:, 0] = lens.astype(">u4").view(np.int32)
    rec[:, 1:] = ids
    return rec


def random_init_bert_base(torch, prefix, seed=0):
    """Random-init bert-base-uncased tower under a DPR prefix (question_model. / ctx_model.)."""
    g = torch.Generator().manual_seed(seed)
    H, I = 768, 3072
    sd = {}

    def lin(name, o, i):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * 0.02
        sd[name + ".bias"] = torch.zeros(o)

    def ln(name):
        sd[name + ".weight"] = torch.ones(H)
        sd[name + ".bias"] = torch.zeros(H)

    e = prefix + "embeddings."
    sd[e + "word_embeddings.weight"] = torch.randn(30522, H, generator=g) * 0.02
    sd[e + "position_embeddings.weight"] = torch.randn(512, H, generator=g) * 0.02
    sd[e + "token_type_embeddings.weight"] = torch.randn(2, H, generator=g) * 0.02
    ln(e + "LayerNorm")
    for i in range(12):
        p = "%sencoder.layer.%d." % (prefix, i)
        for k in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            lin(p + k, H, H)
        ln(p + "attention.output.LayerNorm")
        lin(p + "intermediate.dense", I, H)
        lin(p + "output.dense", H, I)
        ln(p + "output.LayerNorm")
    return sd


def other_config_cases(rng):
    """Encode workloads of BASELINE.json configs[2..4] (SURVEY.md 8d length distributions): (name, tower, L, chunks, lengths,
    (first, last, pad, lo, hi) token ids).  commands/run_ann_data_gen.sh:31,44-49 (MaxP), run_ann_data_gen_dpr.sh:17-26."""
    rob, bert = (0, 2, 1, 3, 50265), (101, 102, 0, 1000, 30522)
    return [
        ("configs[2]: passage L=512 (lognormal lengths, median 70)", "roberta", 512, 1,
         lambda n: np.clip(np.rint(rng.lognormal(np.log(70.0), 0.45, size=n)), 8, 512), rob),
        ("configs[2]: passage L=512 (all 512 tokens, worst case)", "roberta", 512, 1, lambda n: np.full(n, 512.0), rob),
        ("configs[3]: document MaxP 4x512 (lognormal lengths, median 1100)", "roberta", 2048, 4,
         lambda n: np.clip(np.rint(rng.lognormal(np.log(1100.0), 0.9, size=n)), 32, 2048), rob),
        ("configs[4]: DPR BERT-base ctx tower L=256 (lognormal lengths, median 140)", "bert", 256, 1,
         lambda n: np.clip(np.rint(rng.lognormal(np.log(140.0), 0.3, size=n)), 16, 256), bert),
    ]


def measure_other_configs(torch, dev, steps, tokens_per_step, precision, max_tokens, layers=12, with_kernels=True):
    """One encode leg per other BASELINE configuration, on ONE GPU, inputs resident in HBM, random-init weights: items/s,
    algorithmic TFLOP/s and -- from a single-stream pass with the library's HIP events on -- the dominant GEMM's algorithmic
    fraction of the fp16 MFMA peak (the same figures the headline leg carries for configs[1])."""
    from ance_amd import _lib
    from ance_amd.encoder import ARCH_BERT, ARCH_ROBERTA, Encoder
    rng = np.random.default_rng(99)
    towers = {"roberta": (random_init_roberta_base(torch, layers, seed=0), ARCH_ROBERTA, "roberta.", True),
              "bert": (None, ARCH_BERT, "ctx_model.", False)}
    rows = []
    for name, tower, L, chunks, lens_fn, (first, last, pad, lo, hi) in other_config_cases(rng):
        sd, arch, prefix, head = towers[tower]
        if sd is None:
            sd = random_init_bert_base(torch, prefix)
            towers[tower] = (sd, arch, prefix, head)
        n = max(256, int(tokens_per_step / float(lens_fn(2000).mean())) // 64 * 64)
        lens = lens_fn(n).astype(np.int32)
        rec = torch.from_numpy(synthetic_records_of(rng, lens, L, first, last, pad, lo, hi)).to(dev)
        out = torch.empty((n * chunks, 768), dtype=torch.float32, device=dev)
        enc = Encoder(sd, arch, prefix, head, max_seq_len=min(L, 512), max_tokens=max_tokens, device=dev, precision=precision)
        enc.encode_records(rec, n_chunks=chunks, h_lens=lens, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            enc.encode_records(rec, n_chunks=chunks, h_lens=lens, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        enc.check_range(sync=True)
        finite = bool(torch.isfinite(out).all())
        # algorithmic FLOPs: per chunk of T real tokens 169,869,312 T + 36,864 T^2
        if chunks == 1:
            tl = lens.astype(np.float64)
        else:
            tl = np.concatenate([np.clip(lens.astype(np.float64) - 512 * c, 0, 512) for c in range(chunks)])
            tl = tl[tl > 0]
        flops = float((169869312.0 * tl + 36864.0 * tl * tl).sum())
        row = {"config": name, "encoder_precision": enc.precision, "items_per_sec": n / dt, "vectors_per_sec": n * chunks / dt,
               "tokens_per_sec": float(lens.sum()) / dt, "mean_len": float(lens.mean()), "algorithmic_tflops": flops / dt / 1e12,
               "items": n, "steps": steps, "ms_per_step": 1e3 * dt, "finite": finite, "measured_in_this_run": True}
        del enc
        if with_kernels:
            os.environ["ANCE_ENCODER_STREAMS"] = "1"
            try:
                enc1 = Encoder(sd, arch, prefix, head, max_seq_len=min(L, 512), max_tokens=max_tokens, device=dev, precision=precision)
            finally:
                os.environ.pop("ANCE_ENCODER_STREAMS", None)
            enc1.encode_records(rec, n_chunks=chunks, h_lens=lens, out=out)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            enc1.encode_records(rec, n_chunks=chunks, h_lens=lens, out=out)
            torch.cuda.synchronize()
            prof = _lib.profile_read()
            _lib.profile_enable(False)
            cats = [c for c in ("gemm_qk", "gemm_vt", "gemm_attn_out", "gemm_ffn1", "gemm_ffn2") if prof[c]["count"]]
            dom = max(cats, key=lambda c: prof[c]["ms"])
            alg = prof[dom]["work"] / (prof[dom]["ms"] * 1e-3) / 1e12
            tot = sum(v["ms"] for v in prof.values())
            row["dominant_kernel"] = {"category": dom, "algorithmic_tflops": alg, "frac_of_fp16_mfma_peak": alg / PEAK_F16_TF,
                                      "ms_per_launch": prof[dom]["ms"] / prof[dom]["count"],
                                      "attention_share_of_kernel_time": prof["attention"]["ms"] / tot if tot > 0 else None}
            del enc1
        rows.append(row)
        del rec, out
        torch.cuda.empty_cache()
    return rows


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
           "vs_baseline": None, "dtype": DTYPE_OF[HEADLINE_MODE], "data": "synthetic",
           "config": {"workload": "MS MARCO passage %d x 768-d, roberta-base rdot_nll FirstP seq_len=%d, encode + "
                                  "brute-force IP top-%d (BASELINE configs[1])" % (a.n_passages, a.seq_len, a.topk),
                      "encode_block_per_gpu": a.encode_block, "query_block": a.query_block, "layers": a.layers,
                      "parallelism": "dp%d (corpus rows sharded, top-k all-to-all by query owner + merge)" % world}}

    # ------------------------------------------------------------------------------ encode leg --
    # Three arithmetic modes of the same encoder on the same block.  The HEADLINE (top-level value / ms_per_step / dtype /
    # roofline) is the library's default, the split mode: fp32-grade like the reference's own fp32 forward
    # (model/models.py:149-157 has no .half()), so it is the precision-matched number.  The fp16 fast mode and the fp32 audit
    # mode are reported beside it (`encode_fp16_fast`, `encode_fp32`), each with its own roofline and its measured distance
    # from the headline mode's embeddings.
    if not a.skip_encode:
        try:
            sd = random_init_roberta_base(torch, a.layers, seed=0)
            rng = np.random.default_rng(1234 + rank)
            rec, lens = synthetic_records(rng, a.encode_block, a.seq_len)
            rec_d = torch.from_numpy(rec).to(dev)
            flops_alg = float(sum(169869312.0 * t + 36864.0 * t * t + 1179648.0 for t in lens.astype(np.float64)))
            flops_pad = a.encode_block * (169869312.0 * a.seq_len + 36864.0 * a.seq_len ** 2 + 1179648.0)

            want_emb = not a.skip_precise and world == 1
            # the rate the matrix pipe sustains on THIS board (tools/power_probe: ~12 s), single-GPU runs only, before anything is timed
            probe_live = [sustained_mfma_rate() if (world == 1 and not a.skip_mfma_probe) else None]

            def measure_mode(mode, steps, kernels, peak, mfma_per_product, trace_label):
                """K timed steps of `mode` on the product handle (two internal streams), then the same records on a single-stream
                handle with the library's HIP events on: inside the timed region kernels of two micro-batches share the chip,
                so a per-kernel duration is a property of the kernel only in the single-stream pass."""
                enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=min(a.seq_len, 512), max_tokens=a.max_tokens,
                              device=dev, precision=mode)
                assert enc.precision == mode
                emb = torch.empty((a.encode_block, 768), dtype=torch.float32, device=dev)

                def step():
                    enc.encode_records(rec_d, h_lens=lens, out=emb)

                for _ in range(max(a.warmup, 1) if mode == HEADLINE_MODE else 1):
                    step()
                torch.cuda.synchronize()
                with PowerSampler(torch, dev.index or 0) as ps:
                    dt = timed_steps(step, steps, 0, dist_on, torch)
                power = ps.report()
                os.environ["ANCE_ENCODER_STREAMS"] = "1"
                try:
                    enc1 = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=min(a.seq_len, 512), max_tokens=a.max_tokens,
                                   device=dev, precision=mode)
                finally:
                    os.environ.pop("ANCE_ENCODER_STREAMS", None)
                emb1 = torch.empty_like(emb)
                enc1.encode_records(rec_d, h_lens=lens, out=emb1)
                torch.cuda.synchronize()
                n_iso = steps if mode == HEADLINE_MODE else max(1, min(steps, 3))
                _lib.profile_enable(True)
                t1 = time.perf_counter()
                for _ in range(n_iso):
                    enc1.encode_records(rec_d, h_lens=lens, out=emb1)
                torch.cuda.synchronize()
                dt_iso = time.perf_counter() - t1
                prof = _lib.profile_read()
                _lib.profile_enable(False)
                del enc1, emb1
                gemm_cats = [c for c in ("gemm_qk", "gemm_vt", "gemm_attn_out", "gemm_ffn1", "gemm_ffn2") if prof[c]["count"]]
                dom = max(gemm_cats, key=lambda c: prof[c]["ms"])
                alg = prof[dom]["work"] / (prof[dom]["ms"] * 1e-3) / 1e12   # ALGORITHMIC TFLOP/s: 2 M N K per launch / its duration
                all_ms = sum(prof[c]["ms"] for c in gemm_cats)
                all_work = sum(prof[c]["work"] for c in gemm_cats)
                t_ns = trace_avg_ns("%s_rocprofv3_%s_single_stream_kernel_stats.csv" % (CURRENT_ROUND, trace_label), kernels.get(dom, "?"))
                fl = prof[dom]["work"] / max(prof[dom]["count"], 1)
                pps = world * a.encode_block * steps / dt
                roof = {"bound": "mfma", "kernel": "%s (%s)" % (kernels.get(dom, "?"), dom), "achieved": alg, "peak": peak,
                        "unit": "TFLOP/s", "frac": alg / peak,
                        "achieved_is": "algorithmic FLOPs (2 M N K of the launch) / average launch duration",
                        "mfma_products_per_algorithmic_product": mfma_per_product,
                        "frac_algorithmic": alg / peak, "frac_executed": alg * mfma_per_product / peak,
                        "frac_from_profiles": {"value": (fl / (t_ns * 1e-9) / 1e12 / peak) if t_ns else None,
                                               "measured_in_this_run": False,
                                               "source": "profiles/%s_rocprofv3_%s_single_stream_kernel_stats.csv" % (CURRENT_ROUND, trace_label)},
                        "traffic": pmc_traffic({"encode_fp16": "encode"}.get(trace_label, trace_label), dom),
                        "timing": "HIP events on the launch stream, single-stream pass of %d steps (%.1f ms/step isolated vs %.1f "
                                  "ms/step with the product's two internal streams)" % (n_iso, 1e3 * dt_iso / n_iso, 1e3 * dt / steps),
                        "all_gemm_tflops": all_work / (all_ms * 1e-3) / 1e12 if all_ms > 0 else None,
                        "by_kernel": {c: dict(ms_per_launch=v["ms"] / v["count"], launches=v["count"], total_ms=v["ms"],
                                              tflops=(v["work"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["work"] > 0 else None)
                                      for c, v in prof.items() if v["count"]}}
                if peak != PEAK_F32_TF:
                    roof["algorithmic_vs_fp32_mfma_peak"] = alg / PEAK_F32_TF
                    # what the board sustains at its power cap (tools/power_probe.cpp): `peak` assumes 2.4 GHz, random operands allow ~1.7
                    live = probe_live[0]
                    bare = live["mfma"]["tflops"] if live else 1678.0
                    with_reads = live["mfma+lds"]["tflops"] if live else 1525.0
                    roof["sustained_mfma_rate_at_the_power_cap"] = {
                        "bare_mfma_random_operands_tflops": bare, "with_the_main_loops_fragment_reads_tflops": with_reads,
                        "executed_frac_of_the_latter": alg * mfma_per_product / with_reads, "measured_in_this_run": bool(live),
                        "cases": live,
                        "source": "tools/power_probe run by this bench on this box (2 s per case, before the timed legs)" if live else
                                  "profiles/r06_power_probe.txt (tools/power_probe.cpp, scripts/gpu_r6_power_probe.sh)"}
                leg = {"value": pps, "unit": "passages/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "block": a.encode_block,
                       "encoder_precision": mode, "dtype": DTYPE_OF[mode], "arithmetic": ARITHMETIC_OF[mode],
                       "tokens_per_sec": pps * float(lens.mean()), "algorithmic_tflops": world * flops_alg * steps / dt / 1e12,
                       "roofline": roof, "power": power}
                e = None
                if want_emb:  # the mode's embeddings of the block, for the distance between the modes
                    step()
                    torch.cuda.synchronize()
                    e = emb.clone()
                del enc, emb
                torch.cuda.empty_cache()
                return leg, e

            head, emb_head = measure_mode(HEADLINE_MODE, a.steps, KERNEL_OF_SPLIT, PEAK_F16_TF, 3.0, "encode_split")
            pps = head["value"]
            out["value"] = pps
            out["ms_per_step"] = head["ms_per_step"]
            out["dtype"] = head["dtype"]
            out["encoder_precision"] = HEADLINE_MODE
            out["roofline"] = head["roofline"]
            out["power"] = head["power"]  # the board while the timed steps ran (PowerSampler): both matrix-core modes sit at the cap
            out["encode"] = {"passages_per_sec": pps, "tokens_per_sec": head["tokens_per_sec"], "mean_len": float(lens.mean()),
                             "arithmetic": head["arithmetic"], "algorithmic_tflops": head["algorithmic_tflops"],
                             "padded_equiv_tflops": world * flops_pad * a.steps / (head["ms_per_step"] * 1e-3 * a.steps) / 1e12,
                             "end_to_end_mfma_frac": head["algorithmic_tflops"] / (PEAK_F16_TF * world),
                             "hbm_min_bytes_per_passage": 4 + 4 * a.seq_len + 3072,
                             "full_corpus_seconds_est": N_PASSAGES / pps}
            # (single-GPU runs only: the multi-GPU runs are the driver's scaling curve of `value`, and the other two modes would
            # double their length)
            if not a.skip_precise and world == 1:
                fast, emb_fast = measure_mode("fp16", a.steps, KERNEL_OF, PEAK_F16_TF, 1.0, "encode_fp16")
                d = (emb_fast - emb_head).abs()
                fast["max_abs_vs_split"], fast["mean_abs_vs_split"] = float(d.max().item()), float(d.mean().item())
                fast["speedup_vs_headline"] = fast["value"] / pps
                out["encode_fp16_fast"] = fast
                # the 4.4 x slower audit path is capped at 4 steps so that a driver run with a large K still finishes within
                # minutes -- the leg carries the count that was timed
                p32, emb_32 = measure_mode("fp32", min(a.steps, 4), KERNEL_OF_FP32, PEAK_F32_TF, 1.0, "encode_fp32")
                d = (emb_32 - emb_head).abs()
                p32["max_abs_vs_split"], p32["mean_abs_vs_split"] = float(d.max().item()), float(d.mean().item())
                out["encode_fp32"] = p32
                out["encoder_modes"] = {
                    "split (default; headline)": {"passages_per_sec": pps},
                    "fp16 (ANCE_PRECISION_FP16 / --encoder_precision fp16)": {"passages_per_sec": fast["value"],
                                                                              "max_abs_vs_split": fast["max_abs_vs_split"]},
                    "fp32 (ANCE_PRECISION_FP32 / --encoder_precision fp32)": {"passages_per_sec": p32["value"],
                                                                                 "max_abs_vs_split": p32["max_abs_vs_split"]}}
                del emb_fast, emb_32
            del emb_head, rec_d
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
            if dist_on:
                dist.comm = {}  # every device collective of the timed steps bracketed by events on its stream (Dist._timed)
            with PowerSampler(torch, dev.index or 0) as ps_search:
                dt = timed_steps(step_search, a.steps, 0, dist_on, torch)
            prof = _lib.profile_read()
            _lib.profile_enable(False)
            comm = dist.comm_ms() if dist_on else {}
            dist.comm = None
            qps = a.query_block * a.steps / dt
            scan, resc, fin = prof["ip_topk_scan"], prof["ip_topk_rescore"], prof["topk_finalize"]
            n_scan = max(scan["count"], 1)  # (the device-side conditional redo launches of the fast path are not profiled)
            ach = scan["work"] / (scan["ms"] * 1e-3) / 1e12 if scan["ms"] > 0 else None
            D, I = res["DI"]  # rank 0 holds the merged lists
            ok = None
            if rank == 0:
                ok = bool((D[:, 1:] <= D[:, :-1]).all().item()) and bool((I >= 0).all().item())
            exact = None
            if world == 1 and a.exact_check_queries > 0:
                exact = search_exact_check(torch, x, q, D, I, a.topk, a.exact_check_queries)
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
                             "rows_total": a.n_passages, "rows_per_gpu": n_loc, "sorted_and_valid": ok, "exact_check": exact,
                             "full_train_queries_seconds_est": N_TRAIN_QUERIES / qps,
                             "comm_ms_per_step": {k_: {"ms": v_["ms"] / a.steps, "calls": v_["calls"] // a.steps, "bytes_per_rank": v_["bytes"] // a.steps}
                                                  for k_, v_ in comm.items()} if dist_on else None,
                             "comm_note": "rank 0's collectives of one step (all-to-all of the per-shard lists by query owner, gather of the merged "
                                          "blocks), issued on the exchange stream beside the next chunk's scan" if dist_on else None,
                             "search_image_build_ms": build_ms, "power": ps_search.report(),
                             "roofline": {"bound": bound,
                                          "kernel": "ip_topk_fast_kernel (fp16 MFMA 32x32x16 filter; algorithmic FLOPs = 2 nq n d = "
                                                    "1,536 per query-row pair)",
                                          "achieved": ach, "peak": PEAK_F16_TF, "unit": "TFLOP/s",
                                          "frac": (ach / PEAK_F16_TF) if ach else None,
                                          "frac_from_profiles": {"value": (lambda t_: (2.0 * a.query_block * n_loc * 768 / (t_ * 1e-9) / 1e12 / PEAK_F16_TF)
                                                                           if t_ and world == 1 else None)(
                                              trace_avg_ns("%s_rocprofv3_search_kernel_stats.csv" % CURRENT_ROUND, "ip_topk_fast_kernel<false, false>")),
                                              "measured_in_this_run": False, "source": "profiles/%s_rocprofv3_search_kernel_stats.csv" % CURRENT_ROUND},
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
                D2, I2 = res["DI"]
                exact2 = None
                if world == 1 and a.exact_check_queries > 0:
                    exact2 = search_exact_check(torch, x, q, D2, I2, a.topk, a.exact_check_queries)
                out["search"]["encoder_like"] = {
                    "rows": "common component (norm sqrt 768) + 0.12 N(0,1) per row and per query (tests/test_gpu_search.py:_encoder_like)",
                    "value": a.query_block * a.steps / dt2, "unit": "queries/s", "ms_per_step": 1e3 * dt2 / a.steps,
                    "filter_ms_per_launch": scan2["ms"] / max(scan2["count"], 1),
                    "filter_frac_of_peak": (scan2["work"] / (scan2["ms"] * 1e-3) / 1e12 / PEAK_F16_TF) if scan2["ms"] > 0 else None,
                    "rescore_ms_per_launch": prof2["ip_topk_rescore"]["ms"] / max(prof2["ip_topk_rescore"]["count"], 1),
                    "sorted": bool((D2[:, 1:] <= D2[:, :-1]).all().item()) if rank == 0 else None, "exact_check": exact2,
                    "relative_to_layernorm_rows": (a.query_block * a.steps / dt2) / qps}
            del x, q
        except Exception as e:
            import traceback
            traceback.print_exc()
            errors["search"] = repr(e)

    # ------------------------------------------------------- other BASELINE configurations --
    if world == 1 and not a.skip_encode and not a.skip_other_configs:
        try:
            out["other_configs"] = measure_other_configs(torch, dev, max(1, min(a.other_config_steps, 6)), a.other_config_tokens,
                                                         HEADLINE_MODE, a.max_tokens, a.layers)
        except Exception as e:
            import traceback
            traceback.print_exc()
            errors["other_configs"] = repr(e)

    # --------------------------------------------------------------------- whole-refresh slice --
    if world == 1 and not a.skip_slice and not a.skip_encode and out.get("value"):
        try:
            out["full_refresh_slice"] = refresh_slice(a, dist, dev, out["value"])
        except Exception as e:
            import traceback
            traceback.print_exc()
            errors["full_refresh_slice"] = repr(e)

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
        src = next(n for n in ("r06_retrieval_agreement.json", "r05_retrieval_agreement.json")
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
        out["retrieval_agreement"]["measured_in_this_run"] = False
        out["retrieval_agreement"]["source"] = "profiles/%s (tests/test_gpu_retrieval.py: the three encoder modes against the fp32 oracle)" % src
    except Exception:
        pass
    if dist_on and out["rccl_ranks"] != a.gpus:
        errors["ranks"] = "process group has %d ranks, --gpus %d" % (out["rccl_ranks"], a.gpus)
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
