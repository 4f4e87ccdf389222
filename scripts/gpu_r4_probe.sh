#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/encoder_parity.jsonl
timeout 600 python -m pytest tests/test_gpu_encoder.py -q -p no:cacheprovider -k "split" 2>&1 | tail -3
grep split gpurun_out/encoder_parity.jsonl
timeout 300 python - <<'PY'
# split mode at L = 512 (all sequences 512 tokens) and MaxP-like: rate with the MFMA attention (long-sequence path)
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench
from ance_amd.encoder import ARCH_ROBERTA, Encoder
sd = bench.random_init_roberta_base(torch, 12, seed=0)
for attn in ("1", "0"):
    import os
    os.environ["ANCE_SPLIT_ATTN"] = attn
    enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=512, max_tokens=65536, precision="split")
    rng = np.random.default_rng(5)
    for name, lens in (("all-512", np.full(2048, 512, np.int32)), ("lognormal<=512", np.clip(np.rint(rng.lognormal(np.log(70.0), 0.45, 8192)), 8, 512).astype(np.int32))):
        n = len(lens)
        ids = rng.integers(3, 50265, size=(n, 512), dtype=np.int64).astype(np.int32); ids[:, 0] = 0
        ids[np.arange(n), lens - 1] = 2
        ids = np.where(np.arange(512)[None, :] < lens[:, None], ids, 1).astype(np.int32)
        di, dl = torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()
        out = enc.encode_ids(di, dl, h_lens=lens); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): out = enc.encode_ids(di, dl, h_lens=lens)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print("ANCE_SPLIT_ATTN=%s %s: %.0f passages/s (%.1f ms), finite %s" % (attn, name, n / dt, dt * 1e3, bool(torch.isfinite(out).all())))
    del enc
PY
