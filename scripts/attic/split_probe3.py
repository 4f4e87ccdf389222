import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from ance_amd.encoder import ARCH_ROBERTA, Encoder
from oracle import encoder_ref, synth
rng = np.random.default_rng(8)
lens = np.array([1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 128, 70, 9, 100, 50, 77, 128, 3, 45], dtype=np.int32)
ids = synth.make_records(rng, len(lens), 128, lens.astype(np.int64))
def enc(sd, mt, **env):
    for k, v in env.items(): os.environ[k] = v
    e = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=128, max_tokens=mt)
    for k in list(env): os.environ.pop(k)
    return e.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda(), h_lens=lens).cpu().numpy()
for nl in (1, 2):
    sd = encoder_ref.random_state_dict(seed=5, n_layers=nl, ln_jitter=0.1)
    for mode in ({}, {"ANCE_ENCODER_SPLIT": "1"}, {"ANCE_ENCODER_PRECISE": "1"}, {"ANCE_ENCODER_SPLIT": "1", "ANCE_CLS_TAIL": "0"}):
        a = enc(sd, 2048, **mode); b = enc(sd, 512, **mode)
        d = np.abs(a - b)
        print(nl, mode, "rows differing:", [int(i) for i in np.flatnonzero(d.max(1) > 0)], "max %.2e" % d.max(), "frac of elements differing in those rows %.2f" % (float((d > 0).sum()) / max(1, 768 * int((d.max(1) > 0).sum()))))
