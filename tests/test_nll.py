"""SURVEY.md 8(f).4, first slice (forward only): the training objective of the reference -- NLL.forward / NLL_MultiChunk.forward,
model/models.py:57-134 -- against golden values of the reference's own classes (tests/golden/nll.*, make_golden.py::golden_nll).
CPU part: the NumPy restatement (oracle/nll_ref.py) on the reference's embeddings reproduces the reference's loss.  GPU part:
ance_amd's AnceModel.forward (three encodes on the HIP encoder + the fused ance_nll_forward kernel) on the same token ids."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import golden_weights
from oracle import nll_ref


def _golden(golden_dir):
    with open(os.path.join(golden_dir, "nll.json")) as f:
        return json.load(f), np.load(os.path.join(golden_dir, "nll.npz"))


def _chunk_mask(lens, chunks, base=512):
    return (np.asarray(lens)[:, None] > np.arange(chunks)[None, :] * base).astype(np.float32)


def test_oracle_reproduces_the_reference_loss(golden_dir):
    j, g = _golden(golden_dir)
    logits, rows, mean = nll_ref.nll_forward(g["f_q"], g["f_a"], g["f_b"])
    assert abs(mean - j["firstp"]["loss"]) <= 1e-4, (mean, j["firstp"]["loss"])
    ma, mb = _chunk_mask(g["m_a_len"], 4), _chunk_mask(g["m_b_len"], 4)
    assert ma.sum() < ma.size  # the set has all-pad chunks
    logits2, rows2, mean2 = nll_ref.nll_forward(g["m_q"], g["m_a"], g["m_b"], ma, mb)
    assert abs(mean2 - j["maxp"]["loss"]) <= 1e-4, (mean2, j["maxp"]["loss"])
    # the -9999 bias keeps an all-pad chunk from winning the max over chunks (model/models.py:109-113)
    q = np.ones((1, 8)); a = np.zeros((1, 2, 8)); a[0, 1] = 1.0; b = np.zeros((1, 2, 8))
    lg, _, _ = nll_ref.nll_forward(q, a, b, np.array([[1.0, 0.0]]), np.array([[1.0, 1.0]]))
    assert lg[0, 0] == 0.0 and nll_ref.nll_forward(q, a, b, np.ones((1, 2)), np.ones((1, 2)))[0][0, 0] == 8.0


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol", [("fp32", 2e-3), ("split", 2e-3), ("fp16", 0.15)])
def test_forward_on_the_gpu_matches_the_reference(golden_dir, monkeypatch, mode, tol):
    """Loss of the reference's forward on the same ids.  The logits are inner products of ~740 (random-init embeddings share a
    large common component), so the loss -- log(1 + exp(logit_b - logit_a)) -- sees the embeddings' error amplified by
    |q| |a| ~ 768: 2e-3 for the two fp32-grade modes, 0.15 for the fp16-operand fast mode (its 3e-3 embedding tolerance)."""
    from ance_amd.encoder import ARCH_ROBERTA, AnceModel, Encoder
    if mode == "fp32":
        monkeypatch.setenv("ANCE_ENCODER_PRECISE", "1")
    if mode == "split":
        monkeypatch.setenv("ANCE_ENCODER_SPLIT", "1")
    if mode == "fp16":
        monkeypatch.setenv("ANCE_ENCODER_FP16", "1")
    j, g = _golden(golden_dir)
    T = lambda x: torch.from_numpy(np.asarray(x)).cuda()  # noqa: E731

    def mask(lens, L):
        return (torch.arange(L)[None, :] < torch.from_numpy(np.asarray(lens))[:, None]).long().cuda()

    sd = golden_weights(j["firstp"]["weights"])
    model = AnceModel("rdot_nll", Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=64, max_tokens=2048))
    (loss,) = model(T(g["f_q_ids"]).long(), mask(g["f_q_len"], 32), T(g["f_a_ids"]).long(), mask(g["f_a_len"], 64),
                    T(g["f_b_ids"]).long(), mask(g["f_b_len"], 64))
    want_logits, want_rows, _ = nll_ref.nll_forward(g["f_q"], g["f_a"], g["f_b"])
    assert abs(float(loss) - j["firstp"]["loss"]) <= tol, (float(loss), j["firstp"]["loss"])
    assert np.abs(model.last_loss_rows.cpu().numpy() - want_rows).max() <= 4 * tol
    # one input: the embedding of the right tower (model/models.py:66-69)
    e = model(T(g["f_q_ids"]).long(), mask(g["f_q_len"], 32))
    assert e.shape == (12, 768) and np.abs(e.cpu().numpy() - g["f_q"]).max() <= (5e-3 if mode == "fp16" else 2e-5)
    # the kernel alone on the reference's own embeddings: fp32 dot products, fixed order
    from ance_amd import _lib
    import ctypes
    q, a, b = T(g["f_q"]), T(g["f_a"]), T(g["f_b"])
    lg, rw, mn = torch.empty((12, 2), device="cuda"), torch.empty(12, device="cuda"), torch.empty(1, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(_lib.lib().ance_nll_forward(P(q), P(a), P(b), None, None, 12, 768, 1, P(lg), P(rw), P(mn), _lib.current_stream_ptr()),
               "ance_nll_forward")
    assert np.abs(lg.cpu().numpy() - want_logits).max() <= 1e-3 and abs(float(mn) - j["firstp"]["loss"]) <= 1e-4
    del model

    sd2 = golden_weights(j["maxp"]["weights"])
    model2 = AnceModel("rdot_nll_multi_chunk", Encoder(sd2, ARCH_ROBERTA, "roberta.", True, max_seq_len=512, max_tokens=8192), chunks=4)
    (loss2,) = model2(T(g["m_q_ids"]).long(), mask(g["m_q_len"], 32), T(g["m_a_ids"]).long(), mask(g["m_a_len"], 2048),
                      T(g["m_b_ids"]).long(), mask(g["m_b_len"], 2048))
    assert abs(float(loss2) - j["maxp"]["loss"]) <= tol, (float(loss2), j["maxp"]["loss"])
