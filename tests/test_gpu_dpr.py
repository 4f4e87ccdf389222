"""DPR refresh job end-to-end on an MI355X: two BERT towers from one DPR checkpoint file, four
collections, answer-hit metrics and answer-filtered negatives; checked against the oracle (BERT [CLS]
restatement, chain search) and for internal consistency.  Needs an MI355X."""
import random
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_cache(path, ids, lens):
    from oracle import synth
    synth.write_cache(path, ids, lens)


@pytest.mark.parametrize("precision,tol", [("fp16", 1e-2), (None, 2e-5)])  # None: the job's default = split (fp32-grade)
def test_dpr_job_end_to_end(tmp_path, precision, tol):
    from ance_amd import ann_data_gen as adg
    from ance_amd import ann_data_gen_dpr as job
    from ance_amd import dpr
    from oracle import encoder_ref, search_ref
    rng = np.random.default_rng(3)
    L, n_p, n_q, n_t, n_v = 48, 300, 24, 10, 8
    words = ["alpha", "beta", "gamma", "delta", "paris", "tokyo", "moon", "river", "stone", "zürich", "café", "1969"]

    def make_ids(n, lo, hi):
        lens = rng.integers(lo, hi + 1, size=n)
        ids = rng.integers(1000, 30000, size=(n, L)).astype(np.int32)
        ids[:, 0] = 101
        ids[np.arange(n), lens - 1] = 102
        ids = np.where(np.arange(L)[None, :] < lens[:, None], ids, 0).astype(np.int32)
        return ids, lens.astype(np.int64)

    data = tmp_path / "data"
    data.mkdir()
    p_ids, p_len = make_ids(n_p, 6, L)
    q_ids, q_len = make_ids(n_q, 4, 16)
    t_ids, t_len = make_ids(n_t, 4, 16)
    v_ids, v_len = make_ids(n_v, 4, 16)
    _write_cache(str(data / "passages"), p_ids, p_len)
    _write_cache(str(data / "train-query"), q_ids, q_len)
    _write_cache(str(data / "test-query"), t_ids, t_len)
    _write_cache(str(data / "trivia-test-query"), v_ids, v_len)
    with open(data / "pid2offset", "w") as f:
        for pid in range(n_p):
            f.write("%d\t%d\n" % (pid + 1000, pid))
    texts = [" ".join(words[int(j)] for j in rng.integers(0, len(words), size=12)) for _ in range(n_p)]
    with open(tmp_path / "psgs_w100.tsv", "w") as f:
        f.write("id\ttext\ttitle\n")
        for pid in range(n_p):
            f.write("%d\t%s\tT%d\n" % (pid + 1000, texts[pid], pid))
    with open(data / "train-ann", "w") as f:
        for q in range(n_q):
            f.write("%d\t%d\t%r\n" % (q, int(rng.integers(0, n_p)), [words[int(rng.integers(0, len(words)))]]))
    with open(tmp_path / "nq-test.csv", "w") as f:
        for q in range(n_t):
            f.write("q%d?\t%r\n" % (q, [words[q % len(words)], "nonexistent"]))
    with open(tmp_path / "trivia-test.csv", "w") as f:
        for q in range(n_v):
            f.write("t%d?\t%r\n" % (q, ["moon river", words[(q + 3) % len(words)]]))

    sd = encoder_ref.random_state_dict(kind="bert", seed=4, n_layers=2, vocab=30522, max_pos=512, head=False,
                                       prefixes=("question_model.", "ctx_model."), ln_jitter=0.05)
    tr = tmp_path / "train"
    tr.mkdir()
    torch.save({"model_dict": sd, "optimizer_dict": {}, "scheduler_dict": {}, "offset": 0, "epoch": 0,
                "encoder_params": {}}, str(tr / "checkpoint-700"))
    out = str(tmp_path / "out")
    args = types.SimpleNamespace(
        data_dir=str(data), training_dir=str(tr), init_model_dir="/none", last_checkpoint_dir="", output_dir=out,
        cache_dir=out, model_type="dpr", end_output_num=0, max_seq_length=L, max_query_length=16, topk_training=40,
        negative_sample=12, only_keep_latest_embedding_file=False, passage_path=str(tmp_path), test_qa_path=str(tmp_path),
        trivia_test_qa_path=str(tmp_path), device=torch.device("cuda"), max_tokens=2048, encoder_precision=precision)
    random.seed(9)
    job.ann_data_gen(args)
    no, train_path, nd = adg.get_latest_ann_data(out)
    assert no == 0 and set(nd) == {"top20", "top100", "top20_trivia", "top100_trivia", "checkpoint"}
    assert nd["checkpoint"].endswith("checkpoint-700") and 0.0 <= nd["top20"] <= nd["top100"] <= 1.0

    # oracle embeddings (fp32 BERT [CLS]) agree with the towers the job used, within the stated tolerance
    with torch.no_grad():
        q_ref = encoder_ref.bert_cls(sd, torch.from_numpy(q_ids), (torch.from_numpy(q_ids) != 0).long(), "question_model.", 2).numpy()
        p_ref = encoder_ref.bert_cls(sd, torch.from_numpy(p_ids), (torch.from_numpy(p_ids) != 0).long(), "ctx_model.", 2).numpy()
    from ance_amd.encoder import load_model
    model = load_model("dpr", str(tr / "checkpoint-700"), max_seq_length=L, max_tokens=2048, precision=precision)
    q_gpu = model.query_emb(torch.from_numpy(q_ids).cuda(), (torch.from_numpy(q_ids) != 0).cuda()).cpu().numpy()
    p_gpu = model.body_emb(torch.from_numpy(p_ids).cuda(), (torch.from_numpy(p_ids) != 0).cuda()).cpu().numpy()
    assert np.abs(q_gpu - q_ref).max() <= tol and np.abs(p_gpu - p_ref).max() <= tol  # raw BERT [CLS] rows (no final LayerNorm)
    assert not np.allclose(q_gpu[:4], model.body_emb(torch.from_numpy(q_ids[:4]).cuda(), (torch.from_numpy(q_ids[:4]) != 0).cuda()).cpu().numpy(), atol=1e-3)

    # the written negatives are exactly what the DPR rules give on the exact top-k of the GPU embeddings
    _, I = search_ref.flat_ip_topk_chain(p_gpu, q_gpu, 40)
    passages, pos, answers, _, _ = job.load_data(args)
    neg = dpr.generate_negative_passage_ids(dpr.AnswerMatcher(passages), answers, np.arange(n_q), np.arange(n_p), I, pos, 12)
    lines = {int(l.split("\t")[0]): l.rstrip("\n").split("\t") for l in open(train_path)}
    assert set(lines) == set(range(n_q))
    for q in range(n_q):
        assert int(lines[q][1]) == pos[q]
        assert [int(x) for x in lines[q][2].split(",") if x] == neg[q]
