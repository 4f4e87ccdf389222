#!/usr/bin/env python
"""A/B sweep of the search kernel's scheduling knobs on the headline shape (one process, one corpus):
splits per query tile x corpus window x threshold exchange.  Every configuration must return the same (D, I)
as the first one; prints one JSON line per configuration.   python scripts/sweep_search.py [--rows N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8841823)
    ap.add_argument("--queries", type=int, default=32768)
    ap.add_argument("--topk", type=int, default=200)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--configs", type=str, default="")
    a = ap.parse_args()
    import torch
    from ance_amd.index import FlatIPIndex
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(4321)
    x = torch.empty((a.rows, 768), dtype=torch.float32, device=dev)
    for b0 in range(0, a.rows, 1 << 20):
        b1 = min(b0 + (1 << 20), a.rows)
        x[b0:b1] = torch.nn.functional.layer_norm(torch.randn((b1 - b0, 768), generator=g, device=dev), (768,))
    q = torch.nn.functional.layer_norm(torch.randn((a.queries, 768), generator=torch.Generator(device=dev).manual_seed(99),
                                                   device=dev), (768,))
    idx = FlatIPIndex(768, device=dev)
    idx.add(x)
    # every configuration: ANCE_FAST_* environment of the library (read at every call), short names below
    names_ = dict(S="ANCE_FAST_SPLITS", W="ANCE_FAST_WINDOW_TILES", share="ANCE_FAST_SHARE", wait="ANCE_FAST_WINDOW_WAIT_US",
                  prune="ANCE_FAST_PRUNE_AT", dbg="ANCE_FAST_DEBUG", grow="ANCE_FAST_PRUNE_GROWTH")
    base = dict(S=2, W=256, share=1, wait=200, prune=512, dbg=0, grow=150)
    configs = ["", "grow=200", "grow=125", "wait=50", "W=512", "W=0,wait=0", "S=4", "share=0"]
    if a.configs:
        configs = a.configs.split(";")
    from ance_amd import _lib
    import ctypes
    stamps = torch.zeros((2048, 8), dtype=torch.int64, device=dev)
    ref = None
    for cfg in configs:
        kv = dict(base)
        for item in filter(None, cfg.split(",")):
            k_, v_ = item.split("=")
            kv[k_] = int(v_)
        for k_, v_ in kv.items():
            os.environ[names_[k_]] = str(v_)
        _lib.reload_env()
        D, I = idx.search_device(q, a.topk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            D, I = idx.search_device(q, a.topk)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        _lib.profile_enable(True)   # per-kernel HIP-event times of one more step (events serialise nothing here: one stream)
        idx.search_device(q, a.topk)
        torch.cuda.synchronize()
        prof = {k_: round(v_["ms"], 2) for k_, v_ in _lib.profile_read().items() if v_["count"]}
        _lib.profile_enable(False)
        same = None
        if ref is None:
            ref = (D.clone(), I.clone())
        else:
            same = bool(torch.equal(I, ref[1]) and torch.equal(D, ref[0]))
        # where the time of a workgroup goes: one more search with the instrumented kernel (last launch chunk)
        stamps.zero_()
        _lib.lib().ance_debug_search_stamps(ctypes.c_void_p(stamps.data_ptr()))
        idx.search_device(q, a.topk)
        torch.cuda.synchronize()
        _lib.lib().ance_debug_search_stamps(None)
        st = stamps.cpu().numpy()
        st = st[st[:, 1] > 0]
        names = ["prologue", "main", "filter", "prune", "sync", "block_end"]
        per_wg = {n_: round(float(st[:, i].mean()) / 1e5, 2) for i, n_ in enumerate(names)}  # ms
        per_wg["main_max"] = float(st[:, 1].max()) / 1e5
        per_wg["sync_max"] = float(st[:, 4].max()) / 1e5
        xcds = sorted(set(int(v) & 0xF for v in st[:, 7]))
        print(json.dumps(dict(cfg=cfg or "default", ms=1e3 * dt, qps=a.queries / dt, tflops=2.0 * a.queries * a.rows * 768 / dt / 1e12,
                              same_as_first=same, kernels_ms=prof, wg_ms=per_wg, n_wg=int(len(st)), xcc_ids=xcds)), flush=True)


if __name__ == "__main__":
    main()
