"""Offline retrieval metrics from ``--inference`` dumps: what
``evaluation/Calculate Metrics.ipynb`` computes -- full-rank and BM25-rerank NDCG@10 / MAP@10 /
MRR / recall@N / hole rates / MS MARCO MRR@10 -- with the two searches on the MI355X:

  full rank   one ``FlatIPIndex.search`` over every passage vector (notebook cell 13)
  rerank      ``ance_ip_score_rows`` on each query's BM25 candidate rows (cell 11 builds a faiss
              sub-index per query), scores bitwise those of the full scan

    python -m ance_amd.metrics --checkpoint_path OUT/ --checkpoint 200000 --data_type 1 --test_set 0 \\
        --raw_data_dir RAW --processed_data_dir DATA

Measure definitions are trec_eval's (what ``pytrec_eval`` computes): a document is relevant when
rel > 0; ``ndcg_cut_10`` gain = rel, discount log2(rank + 1), ideal = judged rels descending;
``map_cut_10`` = sum of precision at relevant ranks <= 10 / #relevant; ``recip_rank`` over the
whole returned list; ``recall_N`` = relevant in the first N / #relevant.  ``ms_mrr`` is the
official MS MARCO MRR@10 (utils/msmarco_eval.py:100-131: first hit within 10, averaged over ALL
reference queries).
"""
import argparse
import collections
import csv
import ctypes
import glob
import gzip
import math
import os
import pickle

import numpy as np

from . import _lib
from .index import FlatIPIndex


# ---------------------------------------------------------------------------------- inputs
def load_inference_dumps(output_dir, checkpoint):
    """Concatenate the per-rank dumps of ``--inference`` (drivers/run_ann_data_gen.py:215-224 via
    utils/util.py:108-113): ``{dev_query,passage}_{ckpt}__{emb,embid}_p__data_obj_{rank}.npy``."""
    def cat(prefix):
        parts = []
        r = 0
        while True:
            f = os.path.join(output_dir, "%s_data_obj_%d.npy" % (prefix, r))
            if not os.path.exists(f):
                break
            parts.append(np.load(f, allow_pickle=False))
            r += 1
        if not parts:
            raise FileNotFoundError("no dumps %s_data_obj_*.npy under %s" % (prefix, output_dir))
        return np.concatenate(parts, axis=0)

    c = str(checkpoint)
    return (cat("dev_query_" + c + "__emb_p_"), cat("dev_query_" + c + "__embid_p_"),
            cat("passage_" + c + "__emb_p_"), cat("passage_" + c + "__embid_p_"))


def load_dev_qrels(processed_data_dir):
    """{qid_offset: {pid_offset: rel}} from ``dev-qrel.tsv`` (notebook cell 4)."""
    out = {}
    with open(os.path.join(processed_data_dir, "dev-qrel.tsv"), "r", encoding="utf8") as f:
        for topicid, docid, rel in csv.reader(f, delimiter="\t"):
            out.setdefault(int(topicid), {})[int(docid)] = int(rel)
    return out


def _open_text(path):
    return gzip.open(path, "rt", encoding="utf-8") if path[-2:] == "gz" else open(path, "rt", encoding="utf-8")


def load_bm25_candidates(query_path, candidate_path, qidmap, pidmap, data_type):
    """{qid_offset: set(pid_offset)} of the first-stage candidates (notebook cell 6): TREC run lines for
    documents (``qid Q0 Dpid rank score run``), ``qid \\t pid \\t query \\t passage`` for passages."""
    qset = set()
    with _open_text(query_path) as f:
        for qid, _ in csv.reader(f, delimiter="\t"):
            qset.add(qid)
    bm25 = collections.defaultdict(set)
    with _open_text(candidate_path) as f:
        for line in f:
            if data_type == 0:
                qid, _, pid, _, _, _ = line.split(" ")
                pid = pid[1:]
            else:
                qid, pid, _, _ = line.split("\t")
            if qid in qset and int(qid) in qidmap:
                bm25[qidmap[int(qid)]].add(pidmap[int(pid)])
    return bm25


# ---------------------------------------------------------------------------------- searches
def full_rank(dev_query_embedding, passage_embedding, topN, device=None):
    """Row ids [nq, topN] of the exact inner-product top-N (cell 13)."""
    idx = FlatIPIndex(passage_embedding.shape[1], device=device)
    idx.add(passage_embedding)
    _, I = idx.search(np.ascontiguousarray(dev_query_embedding, dtype=np.float32), int(topN))
    return I


def rerank(dev_query_embedding, dev_query_embedding2id, passage_embedding, passage_embedding2id, bm25, device=None):
    """Per query: the rows of its candidate pids (every vector of a pid for MaxP) ordered by exact score
    (cell 11).  Returns a list of int64 arrays (ragged).  Ties: lower row id first."""
    import torch
    dev = torch.device(device if device is not None else "cuda")
    p2id = np.asarray(passage_embedding2id).reshape(-1)
    order = np.argsort(p2id, kind="stable")
    sorted_ids = p2id[order]
    rows, offsets = [], [0]
    for qid in np.asarray(dev_query_embedding2id).reshape(-1).tolist():
        cand = np.fromiter(bm25.get(qid, ()), dtype=np.int64)
        lo = np.searchsorted(sorted_ids, cand, side="left")
        hi = np.searchsorted(sorted_ids, cand, side="right")
        r = np.concatenate([order[a:b] for a, b in zip(lo.tolist(), hi.tolist())]) if len(cand) else np.zeros(0, np.int64)
        rows.append(np.sort(r))
        offsets.append(offsets[-1] + len(r))
    flat = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    d = passage_embedding.shape[1]
    dp = (d + 3) // 4 * 4

    def dev_f32(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        return torch.nn.functional.pad(t, (0, dp - d)).contiguous() if dp != d else t

    x, q = dev_f32(passage_embedding), dev_f32(dev_query_embedding)
    rows_d = torch.from_numpy(flat.astype(np.int64)).to(dev)
    off_d = torch.from_numpy(np.asarray(offsets, dtype=np.int64)).to(dev)
    scores = torch.empty(max(len(flat), 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ance_ip_score_rows(ctypes.c_void_p(x.data_ptr()), x.shape[0], ctypes.c_void_p(q.data_ptr()),
                                           q.shape[0], dp, ctypes.c_void_p(rows_d.data_ptr()),
                                           ctypes.c_void_p(off_d.data_ptr()), ctypes.c_void_p(scores.data_ptr()),
                                           _lib.current_stream_ptr())
    _lib.check(rc, "ance_ip_score_rows")
    torch.cuda.synchronize(dev)
    s = scores.cpu().numpy()
    out = []
    for i, r in enumerate(rows):
        sc = s[offsets[i]:offsets[i + 1]]
        out.append(r[np.lexsort((r, -sc))])
    return out


# ---------------------------------------------------------------------------------- measures
def eval_dev_query(query_embedding2id, passage_embedding2id, dev_query_positive_id, I_nearest_neighbor, topN):
    """The notebook's ``EvalDevQuery`` (cell 8).  ``I_nearest_neighbor``: [nq, >= topN] array or a list of
    per-query row arrays.  Returns a dict: ndcg (NDCG@10), queries, map (MAP@10), mrr (recip_rank),
    recall (recall@topN), hole_rate (unjudged fraction of the top 10), ahole_rate (of everything
    returned), ms_mrr (official MRR@10)."""
    p2id = np.asarray(passage_embedding2id).reshape(-1)
    total = labeled = atotal = alabeled = 0
    ranked_by_q = {}
    for row, qid in enumerate(np.asarray(query_embedding2id).reshape(-1).tolist()):
        judged = dev_query_positive_id[qid]  # KeyError like the notebook when a query has no qrels
        pids = p2id[np.asarray(I_nearest_neighbor[row])[:topN]]
        _, first = np.unique(pids, return_index=True)  # multiple vectors per document: first occurrence wins
        ranked = pids[np.sort(first)].tolist()
        unj = [p not in judged for p in ranked]
        atotal += len(ranked)
        alabeled += sum(unj)
        total += min(len(ranked), 10)
        labeled += sum(unj[:10])
        ranked_by_q[qid] = ranked  # a repeated query id keeps its last row, like the notebook's dicts
    ndcg = ap = rr = rec = 0.0
    for qid, ranked in ranked_by_q.items():
        judged = dev_query_positive_id[qid]
        gains = [max(judged.get(p, 0), 0) for p in ranked]
        ideal = sorted((g for g in judged.values() if g > 0), reverse=True)
        n_rel = len(ideal)
        dcg = sum(g / math.log2(i + 2) for i, g in enumerate(gains[:10]))
        idcg = sum(g / math.log2(i + 2) for i, g in enumerate(ideal[:10]))
        ndcg += dcg / idcg if idcg > 0 else 0.0
        hits = 0
        s = 0.0
        for i, g in enumerate(gains[:10]):
            if g > 0:
                hits += 1
                s += hits / (i + 1)
        ap += s / n_rel if n_rel else 0.0
        first_rel = next((i for i, g in enumerate(gains) if g > 0), None)
        rr += 1.0 / (first_rel + 1) if first_rel is not None else 0.0
        rec += sum(1 for g in gains[:topN] if g > 0) / n_rel if n_rel else 0.0
    n_eval = len(ranked_by_q)
    # official MS MARCO MRR@10: relevant = judged pids > 0 (pid 0 is the script's padding), mean over ALL qrel queries
    mrr10 = 0.0
    for qid, ranked in ranked_by_q.items():
        if qid not in dev_query_positive_id:
            continue
        target = {p for p in dev_query_positive_id[qid] if p > 0}
        padded = (ranked + [0] * 10)[:10]
        for i, p in enumerate(padded):
            if p in target:
                mrr10 += 1.0 / (i + 1)
                break
    n_ref = len(dev_query_positive_id)
    return dict(ndcg=ndcg / n_eval, queries=n_eval, map=ap / n_eval, mrr=rr / n_eval, recall=rec / n_eval,
                hole_rate=labeled / total if total else 0.0, ahole_rate=alabeled / atotal if atotal else 0.0,
                ms_mrr={"MRR @10": mrr10 / n_ref if n_ref else 0.0, "QueriesRanked": n_eval})


# ---------------------------------------------------------------------------------- CLI
def _paths(a):
    raw = a.raw_data_dir
    if a.data_type == 0:
        if a.test_set == 1:
            return raw + "/msmarco-test2019-queries.tsv", raw + "/msmarco-doctest2019-top100"
        return raw + "/msmarco-docdev-queries.tsv", raw + "/msmarco-docdev-top100"
    if a.test_set == 1:
        return raw + "/msmarco-test2019-queries.tsv", raw + "/msmarco-passagetest2019-top1000.tsv"
    return raw + "/queries.dev.small.tsv", raw + "/top1000.dev"


def _report(title, checkpoint, topN, r):
    print(title + " for checkpoint " + str(checkpoint))
    print("NDCG@10:" + str(r["ndcg"]))
    print("map@10:" + str(r["map"]))
    print("pytrec_mrr:" + str(r["mrr"]))
    print("recall@" + str(topN) + ":" + str(r["recall"]))
    print("hole rate@10:" + str(r["hole_rate"]))
    print("hole rate:" + str(r["ahole_rate"]))
    print("ms_mrr:" + str(r["ms_mrr"]))


def main(argv=None):
    p = argparse.ArgumentParser(description="offline metrics of evaluation/Calculate Metrics.ipynb on the MI355X")
    p.add_argument("--checkpoint_path", required=True, help="output_dir of the --inference run")
    p.add_argument("--checkpoint", required=True, type=int)
    p.add_argument("--data_type", type=int, default=1, help="0 for document, 1 for passage")
    p.add_argument("--test_set", type=int, default=0, help="0 dev set, 1 eval (TREC 2019) set")
    p.add_argument("--raw_data_dir", required=True)
    p.add_argument("--processed_data_dir", required=True)
    a = p.parse_args(argv)
    topN = 100 if a.data_type == 0 else 1000
    qrels = load_dev_qrels(a.processed_data_dir)
    q_emb, q_ids, p_emb, p_ids = load_inference_dumps(a.checkpoint_path, a.checkpoint)
    with open(os.path.join(a.processed_data_dir, "qid2offset.pickle"), "rb") as h:
        qidmap = pickle.load(h)
    with open(os.path.join(a.processed_data_dir, "pid2offset.pickle"), "rb") as h:
        pidmap = pickle.load(h)
    query_path, cand_path = _paths(a)
    results = {}
    if os.path.exists(cand_path) or glob.glob(cand_path + "*"):
        bm25 = load_bm25_candidates(query_path, cand_path, qidmap, pidmap, a.data_type)
        print("number of queries with " + str(topN) + " BM25 passages:", len(bm25))
        if bm25:
            lists = rerank(q_emb, q_ids, p_emb, p_ids, bm25)
            results["rerank"] = eval_dev_query(q_ids, p_ids, qrels, lists, topN)
            _report("Reranking Results", a.checkpoint, topN, results["rerank"])
    I = full_rank(q_emb, p_emb, topN)
    results["full"] = eval_dev_query(q_ids, p_ids, qrels, I, topN)
    _report("Results", a.checkpoint, topN, results["full"])
    return results


if __name__ == "__main__":
    main()
