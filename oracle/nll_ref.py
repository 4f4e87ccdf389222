"""Training-objective oracle (test infrastructure): NumPy restatement of the tail of the reference's forward passes,

  model/models.py:71-81    NLL.forward            logits = [q.a, q.b]; loss = -log_softmax(logits)[:, 0]; mean
  model/models.py:97-134   NLL_MultiChunk.forward logit = max over chunks of (q.a_c + (1 - m_c)(-9999)), m_c = the attention
                                                  mask's first entry of chunk c

given the three embedding sets.  Pinned to the reference's own classes by tests/golden/nll.npz (make_golden.py::golden_nll)."""
import numpy as np


def nll_forward(q, a, b, mask_a=None, mask_b=None):
    """q [n, d]; a, b [n, d] or [n, C, d]; mask_* [n, C] (first mask entry of every chunk).  Returns (logits [n, 2], loss rows
    [n], mean loss) in fp64 arithmetic on the given values."""
    q = np.asarray(q, np.float64)
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.ndim == 2:
        la, lb = (q * a).sum(-1), (q * b).sum(-1)
    else:
        la = (np.einsum("nd,ncd->nc", q, a) + (1.0 - np.asarray(mask_a, np.float64)) * -9999.0).max(-1)
        lb = (np.einsum("nd,ncd->nc", q, b) + (1.0 - np.asarray(mask_b, np.float64)) * -9999.0).max(-1)
    logits = np.stack([la, lb], axis=1)
    m = logits.max(1)
    lse = m + np.log(np.exp(la - m) + np.exp(lb - m))
    rows = lse - la
    return logits, rows, float(rows.mean())
