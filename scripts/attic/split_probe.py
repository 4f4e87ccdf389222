"""Diagnostic: where does the split mode lose its fp32 grade? (layers x CLS tail x N-split)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from ance_amd.encoder import ARCH_ROBERTA, Encoder
from oracle import encoder_ref, synth
rng = np.random.default_rng(8)
lens = np.array([1, 2, 31, 33, 64, 65, 96, 128, 70, 9, 100, 50], dtype=np.int32)
ids = synth.make_records(rng, len(lens), 128, lens.astype(np.int64))
for n_layers in (1, 2, 4):
    sd = encoder_ref.random_state_dict(seed=5, n_layers=n_layers, ln_jitter=0.1)
    with torch.no_grad():
        sd64 = {k: v.double() for k, v in sd.items()}
        want = encoder_ref.rdot_nll_ln_emb(sd64, torch.from_numpy(ids), encoder_ref.mask_from_lengths(lens, 128), n_layers=n_layers).float().numpy()
    for env in ({}, {"ANCE_CLS_TAIL": "0"}, {"ANCE_GEMM_NSPLIT": "0"}, {"ANCE_CLS_TAIL": "0", "ANCE_GEMM_NSPLIT": "0"}, {"ANCE_ENCODER_STREAMS": "1"}):
        os.environ["ANCE_ENCODER_SPLIT"] = "1"
        for k, v in env.items():
            os.environ[k] = v
        enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=128, max_tokens=2048)
        for k in list(env) + ["ANCE_ENCODER_SPLIT"]:
            os.environ.pop(k)
        got = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda(), h_lens=lens).cpu().numpy()
        got2 = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda(), h_lens=lens).cpu().numpy()
        d = np.abs(got - want).max(1)
        print(n_layers, env, "max %.3e" % d.max(), "per-row", ["%.1e" % x for x in d], "repeatable", bool(np.array_equal(got, got2)))
        del enc
