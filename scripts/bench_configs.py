"""Encode throughput of the other BASELINE.json configurations on ONE MI355X (inputs resident in HBM,
random-init weights, synthetic token ids with the length distributions of SURVEY.md 8d):

  3   MS MARCO passage, L=512            lengths clip(LogNormal(ln 70, .45), 8, 512)   (and all-512 worst case)
  4   MS MARCO document, MaxP 4 x 512    lengths clip(LogNormal(ln 1100, .9), 32, 2048)
  5   DPR Wikipedia, BERT-base, L=256    lengths clip(LogNormal(ln 140, .3), 16, 256)  (100-word passages + title)

and the SEARCH of configuration 4: 3,213,835 documents x 4 chunks = 12,855,340 vectors of which every chunk past a
document's end is the same all-pad vector v0 (model/models.py:165-199) -- the duplicate class the search image collapses.

The encode legs are bench.py's `other_configs` leg (bench.measure_other_configs: the driver's run carries them); this script
runs them with more tokens per step and adds the search of configuration 4.  Prints one JSON line per configuration for
BASELINE.md section 4.  Usage on the GPU box:  python scripts/bench_configs.py [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--tokens-per-step", type=int, default=1 << 21)
    ap.add_argument("--search-only", action="store_true")
    ap.add_argument("--skip-search", action="store_true")
    ap.add_argument("--max-tokens", type=int, default=131072)
    ap.add_argument("--precision", default=None, choices=["fp16", "split", "fp32"],
                    help="encoder arithmetic (default: the library's default = split, fp32-grade)")
    a = ap.parse_args()
    import torch
    import bench
    dev = torch.device("cuda", 0)
    if not a.search_only:
        # the encode legs are bench.py's own `other_configs` leg (the driver's run carries them); here with more tokens per step
        for row in bench.measure_other_configs(torch, dev, a.steps, a.tokens_per_step, a.precision, a.max_tokens):
            print(json.dumps(row))
    if not a.skip_search:
        search_config4(torch, dev, a.steps)


def search_config4(torch, dev, steps, n_docs=3213835, nq=32768, k=200):
    """Exact top-200 over the 12.86 M chunk vectors of configuration 4 (rows: LayerNorm-distributed random vectors; all-pad
    chunks -- those past the document's length, lengths as in the encode case above -- are one identical vector)."""
    from ance_amd import _lib
    from ance_amd.index import FlatIPIndex
    rng = np.random.default_rng(4)
    lens = np.clip(np.rint(rng.lognormal(np.log(1100.0), 0.9, size=n_docs)), 32, 2048).astype(np.int64)
    pad_chunk = (np.arange(4)[None, :] * 512 >= lens[:, None]).reshape(-1)  # chunk c is all-pad when len <= 512 c
    n = n_docs * 4
    g = torch.Generator(device=dev).manual_seed(44)
    x = torch.empty((n, 768), dtype=torch.float32, device=dev)
    for b0 in range(0, n, 1 << 20):
        b1 = min(b0 + (1 << 20), n)
        x[b0:b1] = torch.nn.functional.layer_norm(torch.randn((b1 - b0, 768), generator=g, device=dev), (768,))
    v0 = torch.nn.functional.layer_norm(torch.randn((768,), generator=g, device=dev), (768,))
    x[torch.from_numpy(pad_chunk).to(dev)] = v0
    q = torch.nn.functional.layer_norm(torch.randn((nq, 768), generator=g, device=dev), (768,))
    q[7] = v0  # a query whose whole list is the class
    res = {}
    for dedup in ("1", "0"):
        os.environ["ANCE_FAST_DEDUP"] = dedup
        _lib.reload_env()
        idx = FlatIPIndex(768, device=dev)
        idx.add(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D, I = idx.search_device(q[:256], k)
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0  # image build + one small search
        nsteps = steps if dedup == "1" else 1
        t0 = time.perf_counter()
        for _ in range(nsteps):
            D, I = idx.search_device(q, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nsteps
        res[dedup] = (dt, t_first, I.clone())
        del idx
        torch.cuda.empty_cache()
    os.environ.pop("ANCE_FAST_DEDUP", None)
    _lib.reload_env()
    same = bool(torch.equal(res["1"][2], res["0"][2]))
    I7 = res["1"][2][7].cpu().numpy()
    print(json.dumps({"config": "4: search, 12,855,340 MaxP chunk vectors", "rows": n, "all_pad_rows": int(pad_chunk.sum()),
                      "queries_per_sec": nq / res["1"][0], "ms_per_32768_queries": 1e3 * res["1"][0],
                      "image_build_plus_first_search_s": res["1"][1],
                      "without_duplicate_collapse": {"queries_per_sec": nq / res["0"][0], "ms_per_32768_queries": 1e3 * res["0"][0]},
                      "identical_results_with_and_without_collapse": same,
                      "v0_query_returns_ascending_class_ids": bool(pad_chunk[I7].all() and (np.diff(I7) > 0).all())}))


if __name__ == "__main__":
    main()
