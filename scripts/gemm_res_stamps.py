#!/usr/bin/env python
"""Per-workgroup stamps of the RES GEMM (EPI_RESLN) inside the real encoder (measurement library): main loop, statistics
pre-pass, each of the four epilogue passes, store drain.  Last RES launch of one micro-batch (FFN2 of the last layer with
ANCE_CLS_TAIL=0: 768 tiles)."""
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
sd = bench.random_init_roberta_base(torch, 2, seed=0)
enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=128, max_tokens=65536)
rec, lens = bench.synthetic_records(np.random.default_rng(1), 880, 128)
rec_d = torch.from_numpy(rec).cuda()
out = torch.empty((880, 768), device="cuda")
enc.encode_records(rec_d, h_lens=lens, out=out)
torch.cuda.synchronize()
names = ["main_loop", "stats_prepass", "pass0", "pass1", "pass2", "pass3", "drain"]


def stamped(bits):
    """bits: measurement ablation of the RESLN epilogue (0 = the product's code; 1 = residual reads from cache-resident rows,
    2 = output rows written to cache-resident rows, 3 = both -- WRONG results, timing only)."""
    if bits:
        L.ance_debug_gemm_res_ablate.restype = None
        L.ance_debug_gemm_res_ablate.argtypes = [ctypes.c_int]
        L.ance_debug_gemm_res_ablate(bits)
    st = torch.zeros((1024, 8), dtype=torch.int64, device="cuda")
    L.ance_debug_gemm_stamps(ctypes.c_void_p(st.data_ptr()))
    enc.encode_records(rec_d, h_lens=lens, out=out)
    torch.cuda.synchronize()
    L.ance_debug_gemm_stamps(None)
    if bits:
        L.ance_debug_gemm_res_ablate(0)
    t = st.cpu().numpy().astype(np.float64) / 100.0  # us
    t = t[t[:, 0] > 0]
    # slots: 0 start, 1 main done, 2 stats ready, 3 end (after vmcnt(0)), 4..7 end of pass 0..3
    seg = {"main_loop": t[:, 1] - t[:, 0], "stats_prepass": t[:, 2] - t[:, 1], "pass0": t[:, 4] - t[:, 2], "pass1": t[:, 5] - t[:, 4],
           "pass2": t[:, 6] - t[:, 5], "pass3": t[:, 7] - t[:, 6], "drain": t[:, 3] - t[:, 7], "epilogue": t[:, 3] - t[:, 2],
           "total": t[:, 3] - t[:, 0]}
    return {"ablate_bits": bits, "workgroups": int(len(t)), "launch_us": float(t[:, 3].max() - t[:, 0].min()),
            **{k: {"mean": round(float(v.mean()), 2), "p10": round(float(np.percentile(v, 10)), 2), "p90": round(float(np.percentile(v, 90)), 2)}
               for k, v in seg.items()}}


for bits in [int(b) for b in os.environ.get("RES_ABLATE", "0").split(",")]:
    print(json.dumps(stamped(bits)))
