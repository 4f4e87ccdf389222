"""Offline-metrics searches on the MI355X (ance_amd.metrics): restricted-candidate scoring is bitwise the
full scan's arithmetic, and the whole CLI runs from --inference dumps.  Needs an MI355X."""
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rerank_is_the_restricted_exact_search():
    from ance_amd import metrics
    from oracle import search_ref, synth
    rng = np.random.default_rng(41)
    chunks = 2
    x = synth.ln_rows(rng, 1200)
    x[700] = x[13]                       # a tie between two rows of different pids
    q = synth.ln_rows(rng, 9)
    p2id = (np.arange(1200) // chunks).astype(np.int64)
    q2id = np.arange(9, dtype=np.int64)
    bm25 = {int(i): set(rng.choice(600, size=int(rng.integers(1, 80)), replace=False).tolist()) for i in range(8)}
    bm25[0] |= {6, 350}                  # pids of rows 13 and 700
    bm25[7] = set(range(600))            # every pid: must equal the full ranking
    lists = metrics.rerank(q, q2id, x, p2id, bm25)   # query 8 has no candidates
    assert len(lists) == 9 and len(lists[8]) == 0
    for i in range(8):
        rows = np.sort(np.concatenate([np.nonzero(p2id == p)[0] for p in bm25[i]]))
        D, I = search_ref.flat_ip_topk_chain(x[rows], q[i:i + 1], len(rows))
        assert np.array_equal(lists[i], rows[I[0]]), i
    full = metrics.full_rank(q, x, 1000)
    assert np.array_equal(lists[7][:1000], full[7])


def test_cli_from_inference_dumps(tmp_path, capsys):
    from ance_amd import metrics
    from oracle import synth
    rng = np.random.default_rng(42)
    n_p, n_q, d = 5000, 30, 768
    x = synth.ln_rows(rng, n_p)
    q = (x[rng.integers(0, n_p, n_q)] + 0.3 * synth.ln_rows(rng, n_q)).astype(np.float32)
    out, raw, proc = tmp_path / "out", tmp_path / "raw", tmp_path / "proc"
    for p in (out, raw, proc):
        p.mkdir()
    half = n_p // 2
    np.save(out / "passage_9__emb_p__data_obj_0.npy", x[:half])
    np.save(out / "passage_9__emb_p__data_obj_1.npy", x[half:])
    np.save(out / "passage_9__embid_p__data_obj_0.npy", np.arange(half))
    np.save(out / "passage_9__embid_p__data_obj_1.npy", np.arange(half, n_p))
    np.save(out / "dev_query_9__emb_p__data_obj_0.npy", q)
    np.save(out / "dev_query_9__embid_p__data_obj_0.npy", np.arange(n_q))
    scores = q @ x.T
    best = np.argsort(-scores, axis=1)
    with open(proc / "dev-qrel.tsv", "w") as f:
        for i in range(n_q):
            f.write("%d\t%d\t1\n" % (i, best[i, i % 5]))        # the relevant passage sits at rank (i % 5) + 1
    pickle.dump({1000 + i: i for i in range(n_q)}, open(proc / "qid2offset.pickle", "wb"))
    pickle.dump({7 * i: i for i in range(n_p)}, open(proc / "pid2offset.pickle", "wb"))
    with open(raw / "queries.dev.small.tsv", "w") as f:
        for i in range(n_q):
            f.write("%d\tquery %d\n" % (1000 + i, i))
    with open(raw / "top1000.dev", "w") as f:
        for i in range(n_q):
            for p in best[i, :50][::-1]:                          # candidates in arbitrary order
                f.write("%d\t%d\tq\tp\n" % (1000 + i, 7 * int(p)))
    res = metrics.main(["--checkpoint_path", str(out), "--checkpoint", "9", "--data_type", "1", "--test_set", "0",
                        "--raw_data_dir", str(raw), "--processed_data_dir", str(proc)])
    want_mrr = float(np.mean([1.0 / (i % 5 + 1) for i in range(n_q)]))
    for leg in ("full", "rerank"):
        assert abs(res[leg]["ms_mrr"]["MRR @10"] - want_mrr) < 1e-12
        assert abs(res[leg]["mrr"] - want_mrr) < 1e-12 and res[leg]["recall"] == 1.0
    assert res["full"]["queries"] == n_q
    assert "Reranking Results for checkpoint 9" in capsys.readouterr().out
