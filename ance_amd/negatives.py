"""Host-side stages around the search: query chunking, dev NDCG@10, hard-negative selection and
the ``ann_training_data_N`` / ``ann_ndcg_N`` writers (the file contract, seam B1).

Same behaviour as the reference functions they stand in for -- cited per function.  The O(nq k)
negative selection and the training-file writer run in the native host stage of libance_amd.so
(csrc/host_postsearch.hip) instead of per-element Python.  Randomness is drawn
from the module-level ``random`` stream exactly where the reference consumes it (one
``random.shuffle`` of ``range(k)`` per query in row order, then one shuffle of the query order):
the native code continues CPython's Mersenne Twister state in place, so a run is reproducible
under ``random.seed`` and byte-comparable with a seeded reference run.
"""
import ctypes
import json
import os
import random

import numpy as np

from . import _lib


def query_chunk(num_queries, output_num, chunk_factor):
    """Slice of the train queries refreshed at ``output_num`` (drivers/run_ann_data_gen.py:281-296).
    As in the reference the modulo comes first, so ``chunk_factor == 0`` raises."""
    effective_idx = output_num % chunk_factor
    if chunk_factor <= 0:
        chunk_factor = 1
    per = num_queries // chunk_factor
    start = per * effective_idx
    end = num_queries if effective_idx == chunk_factor - 1 else start + per
    return start, end


def _first_occurrence_mask(a):
    """Boolean mask of the first occurrence of each value, per row of a 2-D int array."""
    order = np.argsort(a, axis=1, kind="stable")
    s = np.take_along_axis(a, order, axis=1)
    first_sorted = np.ones_like(s, dtype=bool)
    first_sorted[:, 1:] = s[:, 1:] != s[:, :-1]
    mask = np.zeros_like(first_sorted)
    np.put_along_axis(mask, order, first_sorted, axis=1)
    return mask


def eval_dev_query(query_embedding2id, passage_embedding2id, dev_query_positive_id, I):
    """NDCG@10 of the dev queries (drivers/run_ann_data_gen.py:399-440): first 50 neighbours,
    row -> pid, keep the first row of each pid (MaxP), score = -rank; trec_eval ``ndcg_cut_10``
    (gain = rel, discount log2(rank + 1), ideal = judged rels sorted descending) averaged over the
    queries present in both the run and the qrels.  Returns (ndcg, n_queries)."""
    I = np.asarray(I)[:, :50]
    p2id = np.asarray(passage_embedding2id)
    valid = I >= 0
    pids = np.where(valid, p2id[np.where(valid, I, 0)], -1)
    keep = _first_occurrence_mask(pids) & valid
    disc = 1.0 / np.log2(np.arange(2, 12))  # ranks 1..10
    total, cnt = 0.0, 0
    # a later row with the same query id overwrites an earlier one (dict semantics of the reference)
    last_row = {}
    for row, qid in enumerate(np.asarray(query_embedding2id).tolist()):
        last_row[qid] = row
    for qid, row in last_row.items():
        rels = dev_query_positive_id.get(qid)
        if rels is None:
            continue
        ranked = pids[row][keep[row]][:10]
        dcg = 0.0
        for r, pid in enumerate(ranked.tolist()):
            g = rels.get(pid, 0)
            if g > 0:
                dcg += g * disc[r]
        ideal = sorted((g for g in rels.values() if g > 0), reverse=True)[:10]
        idcg = sum(g * disc[r] for r, g in enumerate(ideal))
        total += dcg / idcg if idcg > 0 else 0.0
        cnt += 1
    return (total / cnt if cnt else 0.0), cnt


class NegativeSelection:
    """Row-wise result of the negative selection: ``neg`` int64 [nq, negative_sample] (-1 padded),
    ``cnt`` int32 [nq] (-1 for rows whose query is not effective), ``pos_pid`` int64 [nq] (-1 when the
    query has no positive), ``mrr`` (the reference's accumulator / num_queries)."""

    def __init__(self, q2id, pos_pid, active, neg, cnt, mrr, num_queries):
        self.q2id, self.pos_pid, self.active, self.neg, self.cnt = q2id, pos_pid, active, neg, cnt
        self.mrr, self.num_queries = mrr, num_queries

    def as_dict(self):
        """{qid: [negative pid, ...]} -- a repeated query id keeps its last row, like the reference's dict."""
        out = {}
        rows = np.nonzero(self.active)[0]
        negs = self.neg[rows].tolist()
        for qid, c, lst in zip(self.q2id[rows].tolist(), self.cnt[rows].tolist(), negs):
            out[qid] = lst[:c]
        return out


def _mt_state():
    st = random.getstate()
    if st[0] != 3 or len(st[1]) != 625:
        raise RuntimeError("unexpected random.getstate() layout")
    return st, np.array(st[1], dtype=np.uint32)


def _mt_install(st, words):
    random.setstate((st[0], tuple(words.tolist()), st[2]))


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def py_shuffled_range(n):
    """``x = list(range(n)); random.shuffle(x)`` as an int64 array, drawn natively from (and advancing)
    the module-level ``random`` state."""
    st, words = _mt_state()
    out = np.empty(int(n), dtype=np.int64)
    _lib.check(_lib.lib().ance_host_py_shuffle(_ptr(words), int(n), _ptr(out)), "ance_host_py_shuffle")
    _mt_install(st, words)
    return out


def select_negatives(query_embedding2id, passage_embedding2id, training_query_positive_id, I, effective_q_id,
                     negative_sample, select_topk, n_threads=0):
    """Hard-negative selection of drivers/run_ann_data_gen.py:339-396 through the native host stage
    ``ance_host_select_negatives`` (include/ance_amd.h).

    Per query row whose id is in ``effective_q_id``: candidates = the k neighbours in a
    ``random.shuffle``d order (default) or the first ``negative_sample + 1`` in rank order
    (``--ann_measure_topk_mrr``); walk them, skip the positive (adding 1/rank to the MRR if
    rank <= 10), skip repeated pids, stop at ``negative_sample`` negatives.  The shuffles consume the
    module-level ``random`` stream exactly as the reference does."""
    I = np.ascontiguousarray(np.asarray(I), dtype=np.int64)
    p2id = np.ascontiguousarray(np.asarray(passage_embedding2id).reshape(-1), dtype=np.int64)
    q2id = np.ascontiguousarray(np.asarray(query_embedding2id).reshape(-1), dtype=np.int64)
    nq, k = I.shape
    qlist = q2id[:nq].tolist()
    active = np.fromiter((q in effective_q_id for q in qlist), dtype=np.uint8, count=nq)
    get = training_query_positive_id.get
    pos_l = [get(q, -1) for q in qlist]
    pos_pid = np.asarray(pos_l, dtype=np.int64) if nq else np.zeros(0, np.int64)
    missing = np.nonzero((active != 0) & (pos_pid < 0))[0]
    for r in missing.tolist():
        if qlist[r] not in training_query_positive_id:
            raise KeyError(qlist[r])  # the reference indexes the dict for every effective query
    neg = np.empty((nq, negative_sample), dtype=np.int64)
    cnt = np.empty(nq, dtype=np.int32)
    mrr = ctypes.c_double(0.0)
    st, words = _mt_state()
    rc = _lib.lib().ance_host_select_negatives(_ptr(words), _ptr(I), nq, k, _ptr(p2id), p2id.shape[0], _ptr(pos_pid),
                                               _ptr(active), int(negative_sample), 1 if select_topk else 0,
                                               int(n_threads), _ptr(neg), _ptr(cnt), ctypes.byref(mrr))
    _lib.check(rc, "ance_host_select_negatives")
    if not select_topk:
        _mt_install(st, words)
    num_queries = int(active.sum())
    return NegativeSelection(q2id[:nq], pos_pid, active, neg, cnt, mrr.value / max(num_queries, 1), num_queries)


def generate_negative_passage_ids(query_embedding2id, passage_embedding2id, training_query_positive_id, I,
                                  effective_q_id, negative_sample, select_topk, rank=0, verbose=True):
    """{qid: [negative pid, ...]} (drivers/run_ann_data_gen.py:339-396); see ``select_negatives``."""
    sel = select_negatives(query_embedding2id, passage_embedding2id, training_query_positive_id, I, effective_q_id,
                           negative_sample, select_topk)
    if select_topk and verbose:
        print("Rank:" + str(rank) + " --- ANN MRR:" + str(sel.mrr))
    return sel.as_dict()


def write_ann_files(output_dir, output_num, n_rows, query_embedding2id, effective_q_id, training_query_positive_id,
                    query_negative_passage, dev_ndcg, checkpoint_path, extra_metrics=None):
    """``ann_training_data_N`` then ``ann_ndcg_N`` -- data file first, the trainer discovers a
    refresh by the ndcg file (drivers/run_ann_data_gen.py:314-334; utils/util.py:229-243).
    Lines: ``qid \\t pos_pid \\t neg,neg,...`` in a ``random.shuffle``d query order.
    ``query_negative_passage``: a ``NegativeSelection`` or the reference's {qid: [pid, ...]} dict."""
    q2id = np.ascontiguousarray(np.asarray(query_embedding2id).reshape(-1)[:n_rows], dtype=np.int64)
    qlist = q2id.tolist()
    get = training_query_positive_id.get
    if isinstance(query_negative_passage, NegativeSelection):
        sel = query_negative_passage
        neg, cnt, pos_pid = sel.neg, sel.cnt, sel.pos_pid
        writable = (sel.active != 0) & np.fromiter((q in training_query_positive_id for q in qlist), dtype=bool,
                                                   count=n_rows)
        # negatives are keyed by query id in the reference: a repeated id prints its last row's list
        uq, inv = np.unique(q2id, return_inverse=True)
        last = np.full(uq.shape[0], -1, dtype=np.int64)
        act_rows = np.nonzero(sel.active)[0]
        np.maximum.at(last, inv[act_rows], act_rows)
        src_row = np.where(writable, last[inv], -1).astype(np.int64)
    else:
        lists = [query_negative_passage[q] if (q in effective_q_id and q in training_query_positive_id) else None
                 for q in qlist]
        width = max([len(x) for x in lists if x is not None] + [0])
        neg = np.full((n_rows, width), -1, dtype=np.int64)
        cnt = np.zeros(n_rows, dtype=np.int32)
        for r, x in enumerate(lists):
            if x is not None:
                cnt[r] = len(x)
                neg[r, :len(x)] = x
        pos_pid = np.asarray([get(q, -1) for q in qlist], dtype=np.int64) if n_rows else np.zeros(0, np.int64)
        src_row = np.asarray([r if x is not None else -1 for r, x in enumerate(lists)], dtype=np.int64)
    neg = np.ascontiguousarray(neg, dtype=np.int64)
    cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    pos_pid = np.ascontiguousarray(pos_pid, dtype=np.int64)
    src_row = np.ascontiguousarray(src_row, dtype=np.int64)
    order = py_shuffled_range(n_rows)
    train_path = os.path.join(output_dir, "ann_training_data_" + str(output_num))
    tmp = train_path + ".tmp"
    lines = ctypes.c_int64(0)
    rc = _lib.lib().ance_host_write_ann_training(tmp.encode(), _ptr(order), n_rows, _ptr(q2id), _ptr(pos_pid),
                                                 _ptr(src_row), _ptr(neg), _ptr(cnt), int(neg.shape[1]),
                                                 ctypes.byref(lines))
    _lib.check(rc, "ance_host_write_ann_training")
    os.replace(tmp, train_path)
    payload = {"ndcg": dev_ndcg, "checkpoint": checkpoint_path}
    if extra_metrics:
        payload.update(extra_metrics)
    ndcg_path = os.path.join(output_dir, "ann_ndcg_" + str(output_num))
    with open(ndcg_path + ".tmp", "w") as f:
        json.dump(payload, f)
    os.replace(ndcg_path + ".tmp", ndcg_path)
    return train_path, ndcg_path


def load_positive_ids(data_dir):
    """train: {qid_offset: pid_offset} (last wins, rel must be "1"); dev: {qid: {pid: rel}}
    (drivers/run_ann_data_gen.py:74-100)."""
    train = {}
    with open(os.path.join(data_dir, "train-qrel.tsv"), "r", encoding="utf8") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) != 3:
                raise ValueError("train-qrel.tsv: expected 3 tab-separated columns, got %r" % line)
            assert parts[2] == "1"
            train[int(parts[0])] = int(parts[1])
    dev = {}
    with open(os.path.join(data_dir, "dev-qrel.tsv"), "r", encoding="utf8") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) != 3:
                raise ValueError("dev-qrel.tsv: expected 3 tab-separated columns, got %r" % line)
            dev.setdefault(int(parts[0]), {})[int(parts[1])] = int(parts[2])
    return train, dev
