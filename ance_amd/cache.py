"""Tokenised binary cache reader (the reference's ``EmbeddingCache``, utils/util.py:257-307).

On-disk format (written by data/msmarco_data.py:160-176,258,272 and data/DPR_data.py):
``<name>_meta`` = JSON ``{'type': 'int32', 'total_number': N, 'embedding_size': L}``;
``<name>`` = N records of ``4-byte big-endian passage_len`` + ``L`` little-endian int32.

Instead of a Python ``seek`` + ``read`` per record, the file is memory-mapped as uint8 [N, 4+4L]:
a block of records is one contiguous slice that goes to HBM verbatim (``ance_encode_records``
byte-swaps the header on the device), and lengths are one strided NumPy view.
"""
import json

import numpy as np


class TokenCache:
    def __init__(self, base_path):
        self.base_path = base_path
        with open(base_path + "_meta", "r") as f:
            meta = json.load(f)
        self.dtype = np.dtype(meta["type"])
        if self.dtype != np.dtype("int32"):
            raise ValueError("unsupported cache dtype %s" % self.dtype)
        self.total_number = int(meta["total_number"])
        self.embedding_size = int(meta["embedding_size"])
        self.record_size = self.embedding_size * self.dtype.itemsize + 4
        self._mm = None

    # context-manager protocol kept for call-shape parity with EmbeddingCache
    def open(self):
        if self._mm is None:
            if self.total_number == 0:
                self._mm = np.zeros((0, self.record_size), dtype=np.uint8)
            else:
                self._mm = np.memmap(self.base_path, dtype=np.uint8, mode="r",
                                     shape=(self.total_number, self.record_size))
        return self

    def close(self):
        self._mm = None

    def __enter__(self):
        return self.open()

    def __exit__(self, *exc):
        self.close()

    def __len__(self):
        return self.total_number

    def records(self, r0=0, r1=None):
        """uint8 view [r1-r0, 4+4L] of raw records (zero copy)."""
        self.open()
        r1 = self.total_number if r1 is None else r1
        return self._mm[r0:r1]

    def lengths(self, r0=0, r1=None):
        """int32 passage_len of records [r0, r1), clamped to [0, L] like the consumers do."""
        rec = self.records(r0, r1)
        be = np.ascontiguousarray(rec[:, :4]).view(">u4").reshape(-1)
        return np.minimum(be.astype(np.int64), self.embedding_size).astype(np.int32)

    def ids(self, r0=0, r1=None):
        """int32 [n, L] token ids (copy)."""
        rec = self.records(r0, r1)
        return np.ascontiguousarray(rec[:, 4:]).view("<i4").reshape(rec.shape[0], self.embedding_size)

    def __getitem__(self, key):
        # same contract (and the same off-by-one tolerance) as EmbeddingCache.__getitem__
        if key < 0 or key > self.total_number:
            raise IndexError("Index {} is out of bound for cached embeddings of size {}".format(
                key, self.total_number))
        rec = self.records(key, key + 1)
        passage_len = int.from_bytes(bytes(rec[0, :4]), "big")
        passage = np.frombuffer(bytes(rec[0, 4:]), dtype=self.dtype)
        return passage_len, passage

    def __iter__(self):
        for i in range(self.total_number):
            yield self[i]


def shard_range(n, rank, world_size):
    """Contiguous row block of ``rank``: rows [g*ceil(n/G), (g+1)*ceil(n/G))."""
    per = (n + world_size - 1) // world_size
    r0 = min(rank * per, n)
    return r0, min(r0 + per, n)
