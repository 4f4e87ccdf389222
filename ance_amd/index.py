"""``FlatIPIndex`` -- MI355X replacement for ``faiss.IndexFlatIP`` with the same call shape
(seam B5): ``idx = FlatIPIndex(d); idx.add(x); D, I = idx.search(q, k)``.

Call sites it replaces: drivers/run_ann_data_gen.py:269-276,303 and
drivers/run_ann_data_gen_dpr.py:238-252.  The corpus lives in HBM as fp32 [n, d].  ``add`` is lazy;
the first search builds the shard's search image once (``ance_ip_index_build``: fp16 rows, duplicate
classes) and every later search reuses it, like faiss reuses what ``add`` built.  Search itself is
``ance_ip_topk_indexed``: the two-precision kernel of csrc/ip_topk_fast.hip (fp16 MFMA filter, exact
fp32 re-scoring) where the shape allows it -- d % 128 == 0, d <= 2048, k <= ``FAST_MAX_K``, n >= 4096 --
and the fp32-MFMA scan of csrc/ip_topk.hip otherwise (k up to ``MAX_K``); both return the same bits.
Results follow the canonical order (score desc, row id asc), ``I = -1`` / ``D = -FLT_MAX`` when
fewer than k rows exist (faiss' convention).  There is no CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib


class FlatIPIndex:
    MAX_K = 1792       # ANCE_TOPK_MAX_K
    FAST_MAX_K = 1024  # above it (or d % 128 != 0, d > 2048, n < 4096) the fp32 scan answers: same bits, ~7x slower

    def __init__(self, d, device=None, row_base=0):
        import torch
        self.d = int(d)
        self.dp = (self.d + 3) // 4 * 4  # kernel needs d % 4 == 0; zero padding keeps scores exact
        self.device = torch.device(device if device is not None else "cuda")
        self.row_base = int(row_base)
        self._parts = []
        self._x = None
        self._ws = None
        self._image = None  # search image of self._x (device bytes), built on first use
        self._image_key = None
        self._image_event = None

    # -- faiss-like surface ---------------------------------------------------------------------
    @property
    def ntotal(self):
        return sum(p.shape[0] for p in self._parts)

    def reset(self):
        self._parts, self._x, self._image = [], None, None

    def _to_device(self, a, what):
        import torch
        if isinstance(a, np.ndarray):
            if a.ndim != 2 or a.shape[1] != self.d:
                raise ValueError("%s must be [n, %d]" % (what, self.d))
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        elif isinstance(a, torch.Tensor):
            if a.dim() != 2 or a.shape[1] != self.d:
                raise ValueError("%s must be [n, %d]" % (what, self.d))
            t = a.to(device=self.device, dtype=torch.float32).contiguous()
        else:
            raise TypeError("%s must be a numpy array or torch tensor" % what)
        if self.dp != self.d:
            t = torch.nn.functional.pad(t, (0, self.dp - self.d)).contiguous()
        return t

    def add(self, x):
        """Append rows.  A contiguous fp32 CUDA tensor is kept by reference (zero copy) -- this is
        how embeddings stay in the HBM of the GPU that encoded them."""
        self._parts.append(self._to_device(x, "x"))
        self._x = None
        self._image = None

    def invalidate(self):
        """Forget the search image of the added rows.  Needed only after the rows were rewritten through a raw pointer by
        code that does not bump the tensor's version counter (the image is keyed on it; the library's own
        ``Encoder.encode_records(out=...)`` / ``encode_ids(out=...)`` do bump it): the filter would otherwise run on the stale
        fp16 rows while the exact re-scoring reads the new ones, and true neighbours could be filtered out silently.  The old
        image stays alive for the searches already queued on it (every search records its stream on the image it uses)."""
        self._image = None

    def _matrix(self):
        import torch
        if self._x is None:
            if not self._parts:
                self._x = torch.zeros((0, self.dp), dtype=torch.float32, device=self.device)
            elif len(self._parts) == 1:
                self._x = self._parts[0]
            else:
                self._x = torch.cat(self._parts, dim=0)
                self._parts = [self._x]
        return self._x

    def search(self, q, k):
        """(D float32 [nq,k], I int64 [nq,k]); numpy in -> numpy out, torch in -> CUDA tensors out."""
        import torch
        as_numpy = isinstance(q, np.ndarray)
        D, I = self.search_device(self._to_device(q, "q"), int(k))
        if as_numpy:
            torch.cuda.synchronize(self.device)
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

    # -- device-resident path -------------------------------------------------------------------
    def _search_image(self, L, x):
        """Device buffer holding the search image of ``x`` (None when the shape has none)."""
        import torch
        n = x.shape[0]
        # the image is a function of the rows' VALUES: rows changed in place since it was built (tensor version counter)
        # or a different storage invalidate it -- the filter would otherwise run on stale fp16 rows while the exact
        # re-scoring reads the new ones
        key = (x.data_ptr(), x._version, n)
        if self._image is not None and self._image_key != key:
            self._image = None  # (searches queued on the old image keep it alive: search_device records their stream on it)
        if self._image is None:
            self._image_key = key
            self._image_event = None
            need = L.ance_ip_index_bytes(n, self.dp) if n else 0
            if need == 0:
                self._image = False
            else:
                buf = torch.empty(need, dtype=torch.uint8, device=self.device)
                with torch.cuda.device(self.device):
                    rc = L.ance_ip_index_build(ctypes.c_void_p(x.data_ptr()), n, self.dp, ctypes.c_void_p(buf.data_ptr()),
                                               buf.numel(), _lib.current_stream_ptr())
                _lib.check(rc, "ance_ip_index_build")
                self._image = buf
                # searches on another stream must not start before the build has finished
                self._image_event = torch.cuda.Event()
                self._image_event.record(torch.cuda.current_stream(self.device))
        elif self._image_event is not None:
            torch.cuda.current_stream(self.device).wait_event(self._image_event)
        return self._image if self._image is not False else None

    def search_device(self, qd, k, exact_scan=False):
        """exact_scan: answer with the fp32-MFMA scan alone (``ance_ip_topk_scan``) -- the independent audit path of the
        two-precision kernel: same bits, ~7 x slower, no search image."""
        import torch
        L = _lib.lib()
        x = self._matrix()
        n, nq = x.shape[0], qd.shape[0]
        if not 1 <= k <= self.MAX_K:
            raise ValueError("k must be in [1, %d]" % self.MAX_K)
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        if nq == 0:
            return D, I
        if exact_scan:
            need = L.ance_ip_topk_scan_workspace_bytes(n, nq, self.dp, k)
            if need == 0:
                raise _lib.AnceLibraryError("ance_ip_topk_scan: unsupported (n=%d, nq=%d, k=%d)" % (n, nq, k))
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                rc = L.ance_ip_topk_scan(ctypes.c_void_p(x.data_ptr() if n else 0), n, self.row_base, ctypes.c_void_p(qd.data_ptr()),
                                         nq, self.dp, k, ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                         ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.current_stream_ptr())
            _lib.check(rc, "ance_ip_topk_scan")
            ws.record_stream(torch.cuda.current_stream(self.device))
            return D, I
        img = self._search_image(L, x)
        if img is not None:
            # at USE time and on the stream that uses it: the caching allocator must not hand the image's memory out again
            # while this search is queued, whichever stream later drops or rebuilds the image
            img.record_stream(torch.cuda.current_stream(self.device))
        need = L.ance_ip_topk_indexed_workspace_bytes(n, nq, self.dp, k)
        if need == 0:
            raise _lib.AnceLibraryError("ance_ip_topk: unsupported (n=%d, nq=%d, k=%d)" % (n, nq, k))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            if img is not None:
                rc = L.ance_ip_topk_indexed(ctypes.c_void_p(x.data_ptr()), n, self.row_base, ctypes.c_void_p(img.data_ptr()),
                                            ctypes.c_void_p(qd.data_ptr()), nq, self.dp, k,
                                            ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                            ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                            _lib.current_stream_ptr())
            else:
                need = L.ance_ip_topk_workspace_bytes(n, nq, self.dp, k)
                if self._ws.numel() < need:
                    self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                rc = L.ance_ip_topk(ctypes.c_void_p(x.data_ptr() if n else 0), n, self.row_base,
                                    ctypes.c_void_p(qd.data_ptr()), nq, self.dp, k,
                                    ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                    ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "ance_ip_topk")
        return D, I


def topk_merge_device(D_parts, I_parts):
    """Merge canonical lists [P, nq, k] (CUDA tensors) -> ([nq, k], [nq, k]) via ``ance_topk_merge``."""
    import torch
    L = _lib.lib()
    D_parts = _lib.require_cuda_tensor(D_parts, torch.float32, "D_parts")
    I_parts = _lib.require_cuda_tensor(I_parts, torch.int64, "I_parts")
    P, nq, k = D_parts.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=D_parts.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=D_parts.device)
    if nq == 0:
        return D, I
    with torch.cuda.device(D_parts.device):
        rc = L.ance_topk_merge(ctypes.c_void_p(D_parts.data_ptr()), ctypes.c_void_p(I_parts.data_ptr()), P, nq, k,
                               ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), None, 0,
                               _lib.current_stream_ptr())
    _lib.check(rc, "ance_topk_merge")
    return D, I
