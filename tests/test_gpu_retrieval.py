"""What the fp16-operand encoder costs in RETRIEVAL terms.

The reference encodes in fp32 (model/models.py:149-157, no .half()); the HIP encoder feeds fp16 operands to the MFMAs
(fp32 accumulation, fp32 residual stream).  Embeddings agree to ~3e-3; this test states what that does to the lists the
job is about: 12 layers, 100,000 passages x 2,048 queries, exact top-200 of both embedding sets (same exact search), and

  recall     = |top-200(fp32 encoder) & top-200(fp16-operand encoder)| / 200, mean over queries
  same_neg   = fraction of queries whose first 20 negatives (the --ann_measure_topk_mrr selection of
               drivers/run_ann_data_gen.py:383: the top of the list, positives skipped) are the same list

The fp32 side is the torch restatement of the reference (oracle/encoder_ref.py, pinned to the reference's own classes by
tests/golden) run on the GPU in fp32 so that 100 k passages take seconds.  Numbers go to gpurun_out/retrieval_agreement.json
(copied to profiles/, quoted by bench.py); the asserts are set just under what was measured."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _fp32_reference(sd_dev, ids, lens, L, batch=512):
    """oracle encoder on the device, fp32, sequences sorted by length and padded per batch (pad columns are masked:
    the result does not depend on the padding)."""
    from oracle import encoder_ref
    order = np.argsort(lens, kind="stable")
    out = torch.empty((len(lens), 768), dtype=torch.float32, device="cuda")
    with torch.no_grad():
        for b0 in range(0, len(order), batch):
            sel = order[b0:b0 + batch]
            Lb = int(lens[sel].max())
            bi = torch.from_numpy(ids[sel, :Lb]).cuda()
            bm = torch.from_numpy(encoder_ref.mask_from_lengths(lens[sel], Lb).numpy()).cuda()
            out[torch.from_numpy(sel).cuda()] = encoder_ref.rdot_nll_ln_emb(sd_dev, bi, bm)
    return out


def test_retrieval_agreement_fp16_operands_vs_fp32_reference(monkeypatch):
    from ance_amd.encoder import ARCH_ROBERTA, Encoder
    from ance_amd.index import FlatIPIndex
    from oracle import encoder_ref, synth
    assert not torch.backends.cuda.matmul.allow_tf32
    n_p, n_q, L, Lq, k, neg = 100_000, 2048, 128, 64, 200, 20
    sd = encoder_ref.random_state_dict(seed=3, n_layers=12)
    rng = np.random.default_rng(17)
    plen = synth.lognormal_lengths(rng, n_p, 70, 0.45, 8, L).astype(np.int32)
    pids = synth.make_records(rng, n_p, L, plen.astype(np.int64))
    qlen = synth.lognormal_lengths(rng, n_q, 9, 0.35, 4, Lq).astype(np.int32)
    qids = synth.make_records(rng, n_q, Lq, qlen.astype(np.int64))
    pos = rng.integers(0, n_p, size=n_q)
    for q in range(n_q // 2):  # half of the queries repeat the head of their positive passage (a retrievable signal)
        m = int(min(qlen[q], plen[pos[q]])) - 1
        qids[q, 1:m] = pids[pos[q], 1:m]

    def encode_all():
        enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=L, max_tokens=65536)
        p = torch.empty((n_p, 768), dtype=torch.float32, device="cuda")
        for b0 in range(0, n_p, 16384):
            b1 = min(b0 + 16384, n_p)
            p[b0:b1] = enc.encode_ids(torch.from_numpy(pids[b0:b1]).cuda(), torch.from_numpy(plen[b0:b1]).cuda(), h_lens=plen[b0:b1])
        q = enc.encode_ids(torch.from_numpy(np.pad(qids, ((0, 0), (0, L - Lq)), constant_values=1)).cuda(),
                           torch.from_numpy(qlen).cuda(), h_lens=qlen)
        del enc
        return p, q

    sd_dev = {k_: v.cuda() for k_, v in sd.items()}
    p32 = _fp32_reference(sd_dev, pids, plen, L)
    q32 = _fp32_reference(sd_dev, qids, qlen, Lq)

    def topk(x, q):
        idx = FlatIPIndex(768)
        idx.add(x)
        D, I = idx.search(q, k)
        return D.cpu().numpy(), I.cpu().numpy()

    D32, I32 = topk(p32, q32)

    def negs(I):
        return [[int(p) for p in I[r, :neg + 1] if p != pos[r]][:neg] for r in range(n_q)]

    n32 = negs(I32)

    def agreement(p16, q16):
        d_p = (p16 - p32).abs().max().item()
        d_q = (q16 - q32).abs().max().item()
        _, I16 = topk(p16, q16)
        n16 = negs(I16)
        return dict(n_passages=n_p, n_queries=n_q, layers=12, k=k, max_abs_passage=d_p, max_abs_query=d_q,
                    recall_at_200=float(np.mean([len(np.intersect1d(I16[r], I32[r])) / k for r in range(n_q)])),
                    identical_top200_set=float(np.mean([np.array_equal(np.sort(I16[r]), np.sort(I32[r])) for r in range(n_q)])),
                    identical_top200_list=float(np.mean(np.all(I16 == I32, axis=1))),
                    identical_top1=float(np.mean(I16[:, 0] == I32[:, 0])),
                    identical_first_20_negatives=float(np.mean([a == b for a, b in zip(n16, n32)])),
                    first_20_negatives_overlap=float(np.mean([len(set(a) & set(b)) / neg for a, b in zip(n16, n32)])),
                    median_score_span_top200=float(np.median(D32[:, 0] - D32[:, -1])),
                    planted_found=float(np.mean(I16[:n_q // 2, 0] == pos[:n_q // 2])),
                    planted_found_fp32=float(np.mean(I32[:n_q // 2, 0] == pos[:n_q // 2])))

    # fp16 fast mode: fp16 MFMA operands (folded LayerNorm, fp16 (hi, lo) residual stream)
    monkeypatch.delenv("ANCE_ENCODER_PRECISE", raising=False)
    monkeypatch.setenv("ANCE_ENCODER_FP16", "1")
    res = agreement(*encode_all())
    monkeypatch.delenv("ANCE_ENCODER_FP16", raising=False)
    # fp32 mode (csrc/precise32.h): what is left is the summation order of two fp32 implementations
    monkeypatch.setenv("ANCE_ENCODER_PRECISE", "1")
    res_p = agreement(*encode_all())
    monkeypatch.delenv("ANCE_ENCODER_PRECISE", raising=False)
    # split mode, the library's default (fp16 pair operands, three MFMAs per k-step): fp32-grade
    monkeypatch.setenv("ANCE_ENCODER_SPLIT", "1")
    res_s = agreement(*encode_all())
    monkeypatch.delenv("ANCE_ENCODER_SPLIT", raising=False)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "retrieval_agreement.json"), "w") as f:
        json.dump(dict(res, precise_mode=res_p, split_mode=res_s), f, indent=1)
    assert res["max_abs_passage"] <= 5e-3 and res["max_abs_query"] <= 5e-3, res
    assert res["recall_at_200"] >= 0.985, res
    assert res["first_20_negatives_overlap"] >= 0.97, res
    assert res["identical_top1"] >= 0.99, res
    assert res_p["max_abs_passage"] <= 5e-5 and res_p["max_abs_query"] <= 5e-5, res_p
    assert res_p["recall_at_200"] >= 0.9995, res_p
    assert res_p["identical_first_20_negatives"] >= 0.9, res_p
    assert res_s["max_abs_passage"] <= 2e-5 and res_s["max_abs_query"] <= 2e-5, res_s
    assert res_s["recall_at_200"] >= 0.9995, res_s
    assert res_s["identical_first_20_negatives"] >= 0.95, res_s
