#!/usr/bin/env python
"""Per-workgroup stamps of the split GEMMs inside the real encoder (measurement library, `make -C ance_amd/csrc measure`): where a
256 x 256 tile's time goes -- prologue + main loop, statistics, epilogue incl. the drain of its stores -- for the QKV (EPI 8),
FFN1 / GELU (9) and FFN2 / RESLN (10) launches of the last layer of one 65,536-token micro-batch (ANCE_CLS_TAIL=0: full size).
    python scripts/gemm_split_stamps.py  > gpurun_out/gemm_split_stamps.jsonl"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ANCE_AMD_LIB", os.path.join(ROOT, "ance_amd", "libance_amd_measure.so"))
os.environ["ANCE_CLS_TAIL"] = "0"
os.environ["ANCE_ENCODER_STREAMS"] = "1"
import bench  # noqa: E402
from ance_amd import _lib  # noqa: E402
from ance_amd.encoder import ARCH_ROBERTA, Encoder  # noqa: E402

L = _lib.lib()
L.ance_debug_gemm_stamps.restype = None
L.ance_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
L.ance_debug_gemm_stamps_epi.restype = None
L.ance_debug_gemm_stamps_epi.argtypes = [ctypes.c_int]
sd = bench.random_init_roberta_base(torch, 2, seed=0)
enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=128, max_tokens=65536, precision="split")
rec, lens = bench.synthetic_records(np.random.default_rng(1), 880, 128)
rec_d = torch.from_numpy(rec).cuda()
out = torch.empty((880, 768), device="cuda")
enc.encode_records(rec_d, h_lens=lens, out=out)
torch.cuda.synchronize()
NAMES = {8: "QKV (N = 2304, K = 768: 24 K-tiles)", 9: "FFN1 / GELU (N = 3072, K = 768: 24 K-tiles)", 10: "FFN2 / RESLN (N = 768, K = 3072: 96 K-tiles)"}
KT = {8: 24, 9: 24, 10: 96}
for rep in range(2):
    for epi in (8, 9, 10):
        st = torch.zeros((8192, 8), dtype=torch.int64, device="cuda")
        L.ance_debug_gemm_stamps_epi(epi)
        L.ance_debug_gemm_stamps(ctypes.c_void_p(st.data_ptr()))
        enc.encode_records(rec_d, h_lens=lens, out=out)
        torch.cuda.synchronize()
        L.ance_debug_gemm_stamps(None)
        t = st.cpu().numpy().astype(np.float64) / 100.0  # us (100 MHz counter)
        t = t[t[:, 0] > 0]
        seg = {"prologue_plus_main_loop": t[:, 1] - t[:, 0], "statistics": t[:, 2] - t[:, 1], "epilogue_incl_store_drain": t[:, 3] - t[:, 2],
               "total": t[:, 3] - t[:, 0]}
        # rounds: workgroups sorted by start time, 256 per round
        order = np.argsort(t[:, 0])
        n_rounds = int(np.ceil(len(t) / 256.0))
        res = {"rep": rep, "epi": epi, "kernel": NAMES[epi], "workgroups": int(len(t)), "rounds": n_rounds,
               "launch_us": float(t[:, 3].max() - t[:, 0].min()),
               "us_per_k_tile_incl_prologue": round(float(seg["prologue_plus_main_loop"].mean()) / KT[epi], 3),
               **{k: {"mean": round(float(v.mean()), 2), "p10": round(float(np.percentile(v, 10)), 2), "p90": round(float(np.percentile(v, 90)), 2)}
                  for k, v in seg.items()}}
        first = order[:256]
        res["first_round"] = {k: round(float(v[first].mean()), 2) for k, v in seg.items()}
        res["non_main_share"] = round(float(1.0 - seg["prologue_plus_main_loop"].sum() / seg["total"].sum()), 3)
        print(json.dumps(res))
