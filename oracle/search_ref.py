"""Search oracle (test infrastructure): exact inner-product top-k under the canonical order.

Two restatements of what the reference asks of ``faiss.IndexFlatIP``
(``drivers/run_ann_data_gen.py:269-276,303``; ``drivers/run_ann_data_gen_dpr.py:238-252``):

* ``flat_ip_topk_chain``  -- C (``ip_topk_ref.c``): fp32 ``fmaf`` chain, k ascending; bit-exact
  target for the HIP kernel.
* ``flat_ip_topk_blas``   -- NumPy/BLAS ``q @ x.T`` + canonical selection; what a faiss-cpu flat
  index does up to BLAS summation order.  Used as the ``faiss`` stand-in for the reference
  harness and as the timed CPU baseline ("port").

FAISS itself is absent (unpinned ``faiss-cpu`` in ``setup.py:22``): parity unpinned there.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NEG_FILL = np.float32(-3.4028234663852886e38)  # -FLT_MAX, faiss' fill for missing results


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "ip_topk_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.ance_oracle_ip_scores.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int64, ctypes.c_int, f32p]
        L.ance_oracle_ip_topk.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, f32p, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_int, f32p, i64p]
        L.ance_oracle_topk_row.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, f32p, i64p]
        L.ance_oracle_topk_merge.argtypes = [f32p, i64p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, f32p, i64p]
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.ance_oracle_heap_update.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                              f32p, i64p, i32p]
        L.ance_oracle_heap_finish.argtypes = [ctypes.c_int64, ctypes.c_int, f32p, i64p, i32p, f32p, i64p]
        for fn in (L.ance_oracle_ip_scores, L.ance_oracle_ip_topk, L.ance_oracle_topk_row, L.ance_oracle_topk_merge,
                   L.ance_oracle_heap_update, L.ance_oracle_heap_finish):
            fn.restype = None
        _LIB = L
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def ip_scores_chain(x, q):
    """fp32 fmaf-chain scores [nq, n]."""
    x, xp = _f32(x)
    q, qp = _f32(q)
    n, d = x.shape
    nq = q.shape[0]
    out = np.empty((nq, n), dtype=np.float32)
    lib().ance_oracle_ip_scores(xp, n, qp, nq, d, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def flat_ip_topk_chain(x, q, k, row_base=0):
    x, xp = _f32(x)
    q, qp = _f32(q)
    n, d = x.shape
    nq = q.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().ance_oracle_ip_topk(xp, n, row_base, qp, nq, d, k,
                              D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                              I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return D, I


def canonical_topk_from_scores(S, k, row_base=0):
    """Canonical (score desc, id asc) top-k of a dense score matrix [nq, n] (NumPy)."""
    nq, n = S.shape
    kk = min(k, n)
    D = np.full((nq, k), NEG_FILL, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if kk == 0:
        return D, I
    ids = np.arange(n, dtype=np.int64)
    for r in range(nq):
        s = S[r]
        if kk < n:
            # candidates: everything >= the kk-th largest value (keeps all ties at the boundary)
            kth = np.partition(s, n - kk)[n - kk]
            cand = np.nonzero(s >= kth)[0]
        else:
            cand = ids
        order = np.lexsort((cand, -s[cand].astype(np.float64)))[:kk]
        sel = cand[order]
        D[r, :kk] = s[sel]
        I[r, :kk] = sel + row_base
    return D, I


def flat_ip_topk_blas(x, q, k, row_base=0, q_block=1024, x_block=262144):
    """BLAS restatement of IndexFlatIP.search: blocked sgemm + canonical selection."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n = x.shape[0]
    nq = q.shape[0]
    D = np.full((nq, k), NEG_FILL, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for q0 in range(0, nq, q_block):
        qb = q[q0:q0 + q_block]
        partD, partI = [], []
        for x0 in range(0, n, x_block):
            S = qb @ x[x0:x0 + x_block].T
            d_, i_ = canonical_topk_from_scores(S, k, row_base + x0)
            partD.append(d_)
            partI.append(i_)
        if len(partD) == 1:
            D[q0:q0 + q_block], I[q0:q0 + q_block] = partD[0], partI[0]
        else:
            d_, i_ = topk_merge(np.stack(partD), np.stack(partI), k)
            D[q0:q0 + q_block], I[q0:q0 + q_block] = d_, i_
    return D, I


def flat_ip_topk_faisslike(x, q, k, row_base=0, q_block=4096, x_block=8192):
    """faiss-cpu IndexFlatIP.search as faiss runs it: blocks of the score matrix by BLAS sgemm (4,096 queries per block,
    faiss/utils/distances.cpp) streamed into one k-heap per query, queries in parallel under OpenMP (ip_topk_ref.c:
    ance_oracle_heap_update).  Same results as ``flat_ip_topk_blas`` up to BLAS rounding of a score by block shape; this is
    the form timed as the CPU baseline (selection by ``np.partition`` per row is several times slower than a heap)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n, nq = x.shape[0], q.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    L = lib()
    f32p, i64p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)
    for q0 in range(0, nq, q_block):
        qb = q[q0:q0 + q_block]
        m = qb.shape[0]
        hs = np.empty((m, k), dtype=np.float32)
        hi = np.empty((m, k), dtype=np.int64)
        hc = np.zeros(m, dtype=np.int32)
        S = np.empty((m, min(x_block, max(n, 1))), dtype=np.float32)
        for x0 in range(0, n, x_block):
            xb = x[x0:x0 + x_block]
            Sb = S[:, :xb.shape[0]]
            np.matmul(qb, xb.T, out=Sb)
            L.ance_oracle_heap_update(Sb.ctypes.data_as(f32p), m, xb.shape[0], S.shape[1], row_base + x0, k,
                                      hs.ctypes.data_as(f32p), hi.ctypes.data_as(i64p), hc.ctypes.data_as(i32p))
        Db, Ib = D[q0:q0 + m], I[q0:q0 + m]
        L.ance_oracle_heap_finish(m, k, hs.ctypes.data_as(f32p), hi.ctypes.data_as(i64p), hc.ctypes.data_as(i32p),
                                  Db.ctypes.data_as(f32p), Ib.ctypes.data_as(i64p))
    return D, I


def topk_merge(D_parts, I_parts, k):
    """Merge per-shard canonical lists [P, nq, k] -> [nq, k] (C oracle)."""
    Dp, dpp = _f32(D_parts)
    Ip, ipp = _i64(I_parts)
    P, nq, kk = Dp.shape
    assert kk == k
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().ance_oracle_topk_merge(dpp, ipp, P, nq, k,
                                 D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                 I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return D, I


class OracleIndexFlatIP:
    """Stand-in for ``faiss.IndexFlatIP`` with the same call shape (seam B5)."""

    def __init__(self, d, chain=False):
        self.d = int(d)
        self.chain = chain
        self._x = np.zeros((0, self.d), dtype=np.float32)

    @property
    def ntotal(self):
        return self._x.shape[0]

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._x = np.concatenate([self._x, x], axis=0)

    def search(self, q, k):
        fn = flat_ip_topk_chain if self.chain else flat_ip_topk_blas
        return fn(self._x, q, int(k))
