"""Where does the default mode lose accuracy on rows with a large mean?  (diagnostic for tests/test_gpu_encoder.py::
test_rows_with_a_large_mean_keep_the_default_tolerance)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ance_amd.encoder import ARCH_ROBERTA, Encoder  # noqa: E402
from oracle import encoder_ref, synth  # noqa: E402

rng = np.random.default_rng(8)
lens = np.array([1, 2, 31, 33, 64, 65, 96, 128, 70, 9, 100, 50], dtype=np.int32)
ids = synth.make_records(rng, len(lens), 128, lens.astype(np.int64))
out = []
for n_layers in (1, 4):
    for offset, emb in ((30.0, True), (30.0, False), (5.0, True), (0.0, False)):
        sd = dict(encoder_ref.random_state_dict(seed=5, n_layers=n_layers, ln_jitter=0.1))
        if emb:
            sd["roberta.embeddings.word_embeddings.weight"] = sd["roberta.embeddings.word_embeddings.weight"] + offset
        for i in range(n_layers):
            for n in ("attention.output.dense.bias", "output.dense.bias"):
                k = "roberta.encoder.layer.%d.%s" % (i, n)
                sd[k] = sd[k] + offset
        with torch.no_grad():
            sd64 = {k: v.double() for k, v in sd.items()}
            want = encoder_ref.rdot_nll_ln_emb(sd64, torch.from_numpy(ids), encoder_ref.mask_from_lengths(lens, 128), n_layers=n_layers).float().numpy()
        row = dict(n_layers=n_layers, offset=offset, emb_offset=emb)
        for mode, env in (("default", {}), ("ln_fold0", {"ANCE_LN_FOLD": "0"}), ("no_tail", {"ANCE_CLS_TAIL": "0"}),
                          ("split", {"ANCE_ENCODER_SPLIT": "1"}), ("fp32", {"ANCE_ENCODER_PRECISE": "1"})):
            for k, v in env.items():
                os.environ[k] = v
            enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=128, max_tokens=2048)
            for k in env:
                os.environ.pop(k)
            got = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda(), h_lens=lens).cpu().numpy()
            d = np.abs(got - want)
            row[mode] = float(d.max())
            row[mode + "_worst_row"] = int(d.max(1).argmax())
            del enc
        out.append(row)
        print(row)
json.dump(out, open("gpurun_out/offset_probe.json", "w"), indent=1)
