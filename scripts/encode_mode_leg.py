"""One encode step of a given encoder mode (fp16 | split | fp32) on the bench's records, for rocprofv3 passes
(scripts/gpu_pmc.sh): python scripts/encode_mode_leg.py split [steps] [block] [max_tokens]."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench  # noqa: E402
from ance_amd.encoder import ARCH_ROBERTA, Encoder  # noqa: E402
mode = sys.argv[1] if len(sys.argv) > 1 else "split"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
block = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
max_tokens = int(sys.argv[4]) if len(sys.argv) > 4 else 131072
sd = bench.random_init_roberta_base(torch, 12, seed=0)
enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=128, max_tokens=max_tokens, precision=mode)
rec, lens = bench.synthetic_records(np.random.default_rng(1234), block, 128)
rec_d = torch.from_numpy(rec).cuda()
emb = torch.empty((block, 768), dtype=torch.float32, device="cuda")
import time  # noqa: E402
enc.encode_records(rec_d, h_lens=lens, out=emb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    enc.encode_records(rec_d, h_lens=lens, out=emb)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("ok", mode, "max_tokens", max_tokens, "lanes", __import__("os").environ.get("ANCE_ENCODER_STREAMS", "2"), float(emb.abs().mean()), "ms_per_step %.2f passages_per_sec %.0f" % (dt * 1e3, block / dt))
