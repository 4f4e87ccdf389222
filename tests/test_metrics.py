"""ance_amd.metrics (offline metrics of evaluation/"Calculate Metrics.ipynb") against
golden outputs of the notebook's own EvalDevQuery cell (tests/golden/metrics.*; generator
tests/golden/make_golden.py::golden_metrics)."""
import json
import os

import numpy as np
import pytest

from ance_amd import metrics

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    z = np.load(os.path.join(G, "metrics.npz"))
    j = json.load(open(os.path.join(G, "metrics.json")))
    qrels = {int(a): {int(b): c for b, c in d.items()} for a, d in j["qrels"].items()}
    return z, qrels, j["results"]


@pytest.mark.parametrize("topN", [100, 1000])
def test_eval_dev_query_matches_notebook(topN):
    z, qrels, want = _golden()
    got = metrics.eval_dev_query(z["q2id"], z["p2id"], qrels, z["I"], topN)
    w = want[str(topN)]
    for key in ("ndcg", "queries", "map", "mrr", "recall", "hole_rate", "ahole_rate"):
        assert got[key] == w[key], key
    assert got["ms_mrr"] == w["ms_mrr"]


def test_ragged_lists_and_missing_qrels():
    z, qrels, _ = _golden()
    lists = [row[: 5 + 3 * i] for i, row in enumerate(z["I"])]  # rerank produces ragged candidate lists
    r = metrics.eval_dev_query(z["q2id"], z["p2id"], qrels, lists, 100)
    assert 0.0 <= r["recall"] <= 1.0 and r["queries"] == len(lists)
    del qrels[3]
    with pytest.raises(KeyError):  # the notebook indexes the qrels of every ranked query
        metrics.eval_dev_query(z["q2id"], z["p2id"], qrels, z["I"], 100)


def test_dump_loader_and_candidates(tmp_path):
    out = tmp_path / "o"
    out.mkdir()
    for r in range(2):
        np.save(out / ("dev_query_7__emb_p__data_obj_%d.npy" % r), np.full((2, 4), r, np.float32))
        np.save(out / ("dev_query_7__embid_p__data_obj_%d.npy" % r), np.arange(2) + 2 * r)
        np.save(out / ("passage_7__emb_p__data_obj_%d.npy" % r), np.full((3, 4), r, np.float32))
        np.save(out / ("passage_7__embid_p__data_obj_%d.npy" % r), np.arange(3) + 3 * r)
    q, qi, p, pi = metrics.load_inference_dumps(str(out), 7)
    assert q.shape == (4, 4) and qi.tolist() == [0, 1, 2, 3] and p.shape == (6, 4) and pi.tolist() == list(range(6))
    raw = tmp_path / "raw"
    raw.mkdir()
    (raw / "queries.tsv").write_text("11\tq a\n12\tq b\n13\tq c\n")
    (raw / "top1000").write_text("11\t100\tq a\tp\n11\t101\tq a\tp\n12\t100\tq b\tp\n99\t100\tzz\tp\n13\t101\tq c\tp\n")
    bm = metrics.load_bm25_candidates(str(raw / "queries.tsv"), str(raw / "top1000"), {11: 0, 12: 1}, {100: 5, 101: 6}, 1)
    assert dict(bm) == {0: {5, 6}, 1: {5}}  # 13 has no offset (unlabelled query), 99 is not a query
    (raw / "run").write_text("11 Q0 D100 1 3.5 bm25\n12 Q0 D101 1 2.5 bm25\n")
    bm = metrics.load_bm25_candidates(str(raw / "queries.tsv"), str(raw / "run"), {11: 0, 12: 1}, {100: 5, 101: 6}, 0)
    assert dict(bm) == {0: {5}, 1: {6}}
