#!/usr/bin/env python
"""Per-workgroup phase stamps of the encoder GEMM (ance_debug_gemm timeline mode, include/ance_amd.h) on the encoder's own
shapes: how much of a launch is main loop and how much is prologue / epilogue / store drain, and whether the workgroups
of a launch run in lockstep (all epilogues at the same time = HBM write bursts)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the timeline build of the kernel exists in the measurement library only (make -C ance_amd/csrc measure)
os.environ.setdefault("ANCE_AMD_LIB", os.path.join(ROOT, "ance_amd", "libance_amd_measure.so"))
from ance_amd import _lib  # noqa: E402


def run(M, N, K, epi, name):
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    a = (torch.randn((M, K), generator=g, device=dev) * 0.5).half()
    b = (torch.randn((N, K), generator=g, device=dev) * 0.05).half()
    bias = torch.zeros(N, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    tiles = (M // 256) * (N // 256)
    blocks = tiles  # M, N multiples of 256 x 8 here
    ts = torch.zeros((blocks + 64, 5), dtype=torch.int64, device=dev)
    st = _lib.current_stream_ptr()
    for _ in range(3):
        rc = L.ance_debug_gemm(32, epi, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), M, N, K,
                               ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ts.data_ptr()), st)
        _lib.check(rc, "ance_debug_gemm")
    torch.cuda.synchronize()
    t = ts.cpu().numpy()[:blocks].astype(np.float64) / 100.0  # us (100 MHz counter)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    d = {"prologue": t[:, 1] - t[:, 0], "main": t[:, 2] - t[:, 1], "epilogue_issue": t[:, 3] - t[:, 2], "drain": t[:, 4] - t[:, 3],
         "total": t[:, 4] - t[:, 0]}
    res = {"shape": name, "M": M, "N": N, "K": K, "epi": epi, "workgroups": int(len(t)), "launch_us": float(t[:, 4].max() - t0)}
    for k, v in d.items():
        res[k + "_us"] = {"mean": float(v.mean()), "p10": float(np.percentile(v, 10)), "p90": float(np.percentile(v, 90))}
    # lockstep: how many workgroups are in their epilogue at the same moment (peak and mean over 1 us bins)
    bins = np.arange(0, res["launch_us"] + 1, 1.0)
    inside = np.zeros(len(bins))
    for s, e in zip(t[:, 2] - t0, t[:, 4] - t0):
        inside[int(s):int(e) + 1] += 1
    res["epilogue_overlap"] = {"peak_workgroups": float(inside.max()), "mean_when_any": float(inside[inside > 0].mean())}
    order = np.argsort(t[:, 0])
    res["start_spread_first_256_us"] = float(t[order[:256], 0].max() - t0)
    print(json.dumps(res))


if __name__ == "__main__":
    run(65536, 3072, 768, 1, "ffn1 (bias + GELU)")
    run(65536, 1536, 768, 0, "qk (bias, Q scale)")
    run(65536, 768, 3072, 0, "ffn2 shape with the fp16 epilogue")
