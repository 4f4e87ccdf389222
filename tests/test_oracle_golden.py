"""The oracle pinned against golden vectors produced by the REAL reference (tests/golden/
make_golden.py, run in the build container where /root/reference is importable)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import ann_ref, encoder_ref, search_ref, synth

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def _manifest(golden_dir):
    with open(os.path.join(golden_dir, "manifest.json")) as f:
        return json.load(f)


from golden_util import golden_weights as _weights  # noqa: E402


def test_encoder_firstp_matches_reference(golden_dir):
    meta = _manifest(golden_dir)["encoder"]["firstp"]
    sd = _weights(meta)
    g = np.load(os.path.join(golden_dir, "encoder_firstp.npz"))
    ids, lens = torch.from_numpy(g["ids"]), g["lens"]
    with torch.no_grad():
        emb = encoder_ref.rdot_nll_ln_emb(sd, ids, encoder_ref.mask_from_lengths(lens, ids.shape[1]), n_layers=2)
    assert np.abs(emb.numpy() - g["emb"]).max() <= 2e-5


def test_encoder_firstp_full_depth_matches_reference(golden_dir):
    """12 layers: the oracle restatement against RobertaDot_NLL_LN.body_emb itself (model/models.py:149-157) at the depth
    every headline number is quoted at."""
    meta = _manifest(golden_dir)["encoder"]["firstp12"]
    sd = _weights(meta)
    g = np.load(os.path.join(golden_dir, "encoder_firstp12.npz"))
    ids, lens = torch.from_numpy(g["ids"]), g["lens"]
    with torch.no_grad():
        emb = encoder_ref.rdot_nll_ln_emb(sd, ids, encoder_ref.mask_from_lengths(lens, ids.shape[1]), n_layers=12)
    assert np.abs(emb.numpy() - g["emb"]).max() <= 2e-5


def test_encoder_maxp_matches_reference(golden_dir):
    meta = _manifest(golden_dir)["encoder"]["maxp"]
    sd = _weights(meta)
    g = np.load(os.path.join(golden_dir, "encoder_maxp.npz"))
    ids, lens = torch.from_numpy(g["ids"]), g["lens"]
    with torch.no_grad():
        emb = encoder_ref.rdot_nll_multi_chunk_body_emb(sd, ids, encoder_ref.mask_from_lengths(lens, 2048), n_layers=1)
    assert emb.shape == (len(lens), 4, 768)
    assert np.abs(emb.numpy() - g["emb"]).max() <= 2e-5
    # all-pad chunks are one and the same vector
    assert np.array_equal(g["emb"][4, 1], g["emb"][4, 3]) and np.array_equal(g["emb"][4, 1], g["emb"][3, 2])


def test_encoder_bert_matches_reference(golden_dir):
    meta = _manifest(golden_dir)["encoder"]["bert"]
    sd = _weights(meta, kind="bert", vocab=30522, max_pos=512, head=False, prefixes=("ctx_model.",))
    g = np.load(os.path.join(golden_dir, "encoder_bert.npz"))
    ids = torch.from_numpy(g["ids"])
    with torch.no_grad():
        emb = encoder_ref.bert_cls(sd, ids, (ids != 0).long(), "ctx_model.", n_layers=2)
    assert np.abs(emb.numpy() - g["emb"]).max() <= 2e-5


def test_encoders_full_depth_other_towers_match_reference(golden_dir):
    """12 layers for the other two towers and the long FirstP case (tests/golden/make_golden.py: golden_encoder12): the oracle
    restatements against RobertaDot_CLF_ANN_NLL_MultiChunk.body_emb (model/models.py:165-199; documents straddling the chunk
    borders, all-pad chunks), HFBertEncoder (:223-244, L = 256) and RobertaDot_NLL_LN.body_emb at L = 512 (:149-157; lengths
    1, 255, 256, 257, 511, 512) -- the depth the numbers of BASELINE configs 3-5 are quoted at."""
    man = _manifest(golden_dir)["encoder12"]
    g = np.load(os.path.join(golden_dir, "encoder_maxp12.npz"))
    sd = _weights(man["maxp12"])
    with torch.no_grad():
        emb = encoder_ref.rdot_nll_multi_chunk_body_emb(sd, torch.from_numpy(g["ids"]), encoder_ref.mask_from_lengths(g["lens"], 2048),
                                                        n_layers=12)
    assert emb.shape == (6, 4, 768) and np.abs(emb.numpy() - g["emb"]).max() <= 2e-5
    e = g["emb"]  # all-pad chunks: one and the same vector, also at full depth
    assert np.array_equal(e[1, 3], e[2, 2]) and np.array_equal(e[1, 3], e[5, 1]) and not np.array_equal(e[1, 3], e[1, 2])
    g = np.load(os.path.join(golden_dir, "encoder_bert12.npz"))
    sd = _weights(man["bert12"], kind="bert", vocab=30522, max_pos=512, head=False, prefixes=("ctx_model.",))
    ids = torch.from_numpy(g["ids"])
    with torch.no_grad():
        emb = encoder_ref.bert_cls(sd, ids, (ids != 0).long(), "ctx_model.", n_layers=12)
    assert np.abs(emb.numpy() - g["emb"]).max() <= 2e-5
    g = np.load(os.path.join(golden_dir, "encoder_firstp12_L512.npz"))
    sd = _weights(man["firstp12_L512"])
    with torch.no_grad():
        emb = encoder_ref.rdot_nll_ln_emb(sd, torch.from_numpy(g["ids"]), encoder_ref.mask_from_lengths(g["lens"], 512), n_layers=12)
    assert np.abs(emb.numpy() - g["emb"]).max() <= 2e-5


def _postsearch(golden_dir):
    g = np.load(os.path.join(golden_dir, "postsearch.npz"))
    with open(os.path.join(golden_dir, "postsearch.json")) as f:
        j = json.load(f)
    train_pos = {int(k): v for k, v in j["train_pos"].items()}
    dev_pos = {int(k): {int(a): b for a, b in v.items()} for k, v in j["dev_pos"].items()}
    return g, j, train_pos, dev_pos


@pytest.mark.parametrize("topk", [False, True])
def test_negative_selection_matches_reference(golden_dir, topk):
    g, j, train_pos, _ = _postsearch(golden_dir)
    random.seed(j["seed"])
    neg, _ = ann_ref.generate_negative_passage_ids(g["q2id"], g["p2id"], train_pos, g["I"], set(g["q2id"].tolist()),
                                                  j["negative_sample"], topk)
    want = {int(k): v for k, v in j["cases"]["neg_topk%d" % int(topk)].items()}
    assert {int(k): [int(x) for x in v] for k, v in neg.items()} == want


def test_dev_ndcg_matches_reference(golden_dir):
    g, j, _, dev_pos = _postsearch(golden_dir)
    ndcg, cnt = ann_ref.eval_dev_query(np.arange(g["I"].shape[0]), g["p2id"], dev_pos, g["I"])
    assert cnt == j["ndcg_cnt"]
    assert abs(ndcg - j["ndcg"]) < 1e-12


def test_end_to_end_refresh_matches_reference(golden_dir, tmp_path):
    """Oracle pipeline (oracle encoder + BLAS flat IP + restated post-search) reproduces the files the
    reference's own generate_new_ann wrote for the same data / weights / seed."""
    with open(os.path.join(golden_dir, "e2e_toy.json")) as f:
        e = json.load(f)
    sd = _weights(e["weights"])
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, **e["data"])
    a = e["args"]
    train_pos, dev_pos = ann_ref.load_positive_ids(data)

    def enc(name, L):
        lens, ids = ann_ref.read_cache(os.path.join(data, name))
        out = []
        with torch.no_grad():
            for b0 in range(0, len(lens), 16):
                out.append(encoder_ref.rdot_nll_ln_emb(sd, torch.from_numpy(ids[b0:b0 + 16]),
                                                       encoder_ref.mask_from_lengths(lens[b0:b0 + 16], L), n_layers=2).numpy())
        return np.concatenate(out)

    dev_q, p_emb, train_q = enc("dev-query", a["max_query_length"]), enc("passages", a["max_seq_length"]), \
        enc("train-query", a["max_query_length"])
    out = str(tmp_path / "out")
    os.makedirs(out)
    random.seed(a["seed"])
    ndcg, n_dev, _, _ = ann_ref.refresh_from_embeddings(
        out, a["output_num"], a["checkpoint_path"], dev_q, np.arange(len(dev_q)), p_emb, np.arange(len(p_emb)),
        train_q, np.arange(len(train_q)), train_pos, dev_pos, a["topk_training"], a["negative_sample"],
        a["ann_chunk_factor"], a["ann_measure_topk_mrr"], search_ref.flat_ip_topk_blas)
    assert abs(ndcg - e["ann_ndcg_0"]["ndcg"]) < 1e-9
    with open(os.path.join(out, "ann_training_data_0")) as f:
        assert f.read() == e["ann_training_data_0"]
    with open(os.path.join(out, "ann_ndcg_0")) as f:
        assert json.load(f) == e["ann_ndcg_0"]


def test_end_to_end_maxp_refresh_matches_reference(golden_dir, tmp_path):
    """Config 4 (MaxP, 2048 = 4 x 512 tokens): oracle MaxP encoder + flat IP + restated post-search in the REFERENCE's row
    order -- per batch of 16 records one slab of vectors per chunk (drivers/run_ann_data_gen.py:183-186,
    oracle.ann_ref.maxp_row_order) -- reproduce the files the reference's own generate_new_ann wrote with
    RobertaDot_CLF_ANN_NLL_MultiChunk: all-pad chunks (one identical vector each) compete in the top-k lists and the
    duplicate-pid skip of GenerateNegativePassaageID (:383-384, 419-423) is what keeps the negatives distinct."""
    with open(os.path.join(golden_dir, "e2e_maxp.json")) as f:
        e = json.load(f)
    sd = _weights(e["weights"])
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, **e["data"])
    a = e["args"]
    train_pos, dev_pos = ann_ref.load_positive_ids(data)
    nl = e["weights"]["n_layers"]
    bs, chunks = a["per_gpu_eval_batch_size"], a["max_seq_length"] // 512

    def enc_q(name):
        lens, ids = ann_ref.read_cache(os.path.join(data, name))
        with torch.no_grad():
            return encoder_ref.rdot_nll_ln_emb(sd, torch.from_numpy(ids), encoder_ref.mask_from_lengths(lens, a["max_query_length"]),
                                               n_layers=nl).numpy()

    lens, ids = ann_ref.read_cache(os.path.join(data, "passages"))
    assert (lens <= 512).any() and (lens > 1536).any()  # the set exercises all-pad chunks and full documents
    slabs = []
    with torch.no_grad():
        for b0 in range(0, len(lens), bs):
            emb = encoder_ref.rdot_nll_multi_chunk_body_emb(sd, torch.from_numpy(ids[b0:b0 + bs]),
                                                            encoder_ref.mask_from_lengths(lens[b0:b0 + bs], a["max_seq_length"]),
                                                            n_layers=nl).numpy()
            slabs.extend(emb[:, c, :] for c in range(chunks))
    p_emb = np.concatenate(slabs)
    p2id = ann_ref.maxp_row_order(len(lens), 1, bs, chunks)
    assert p_emb.shape[0] == len(lens) * chunks == len(p2id)
    dev_q, train_q = enc_q("dev-query"), enc_q("train-query")
    out = str(tmp_path / "out")
    os.makedirs(out)
    random.seed(a["seed"])
    ndcg, _, _, _ = ann_ref.refresh_from_embeddings(
        out, a["output_num"], a["checkpoint_path"], dev_q, np.arange(len(dev_q)), p_emb, p2id, train_q, np.arange(len(train_q)),
        train_pos, dev_pos, a["topk_training"], a["negative_sample"], a["ann_chunk_factor"], a["ann_measure_topk_mrr"],
        search_ref.flat_ip_topk_blas)
    assert abs(ndcg - e["ann_ndcg_0"]["ndcg"]) < 1e-9
    with open(os.path.join(out, "ann_training_data_0")) as f:
        assert f.read() == e["ann_training_data_0"]


# ---- BASELINE.json configs[0] at its stated size: tests/golden/e2e_config1.* (the reference's own generate_new_ann on
# 10,000 passages / 1,000 + 200 queries, 12 layers, top-200, 20 negatives; ~11 minutes of CPU when it was generated) ----
@pytest.fixture(scope="module")
def config1(golden_dir, tmp_path_factory):
    with open(os.path.join(golden_dir, "e2e_config1.json")) as f:
        e = json.load(f)
    g = np.load(os.path.join(golden_dir, "e2e_config1.npz"))
    data = str(tmp_path_factory.mktemp("c1") / "data")
    synth.make_msmarco_like(data, **e["data"])
    return e, g, data


@pytest.mark.parametrize("run", ["topk", "shuffle"])
def test_config1_post_search_reproduces_the_reference_files(config1, tmp_path, run):
    """Given the neighbour lists the reference's index returned, the oracle's restatement of chunking, negative selection
    (both modes), dev NDCG and the writers reproduces the reference's ann_training_data / ann_ndcg files byte for byte."""
    e, g, data = config1
    r = e["runs"][run]
    a = r["args"]
    train_pos, dev_pos = ann_ref.load_positive_ids(data)
    I_dev = g["I_dev"].astype(np.int64)
    I_all = g["I_train"].astype(np.int64)
    s, t = ann_ref.query_chunk(1000, a["output_num"], a["ann_chunk_factor"])
    I = I_all[s:t] if run == "topk" else g["I_train_chunk2"].astype(np.int64)
    assert I.shape[0] == t - s
    p2id, q2id = np.arange(e["data"]["n_passages"]), np.arange(1000)[s:t]
    random.seed(a["seed"])
    ndcg, n_dev = ann_ref.eval_dev_query(np.arange(200), p2id, dev_pos, I_dev)
    assert n_dev == r["result"][1] and abs(ndcg - r["ann_ndcg"]["ndcg"]) < 1e-12
    eff = set(q2id.tolist())
    neg, _ = ann_ref.generate_negative_passage_ids(q2id, p2id, train_pos, I, eff, a["negative_sample"], a["ann_measure_topk_mrr"])
    ann_ref.write_ann_files(str(tmp_path), a["output_num"], I, q2id, eff, train_pos, neg, ndcg, a["checkpoint_path"])
    with open(os.path.join(str(tmp_path), "ann_training_data_%d" % a["output_num"])) as f:
        assert f.read() == r["ann_training_data"]
    with open(os.path.join(str(tmp_path), "ann_ndcg_%d" % a["output_num"])) as f:
        assert json.load(f) == r["ann_ndcg"]


def test_config1_embeddings_and_lists_of_the_reference(config1):
    """The oracle encoder at 12 layers against rows of the reference's own 10,000-passage run, and the stored neighbour
    lists against the scores those rows imply (first 16 queries x first 16 passages: a spot check that data, weights and
    fixture belong together; the full fp64 cross-check runs on the GPU, tests/test_gpu_config1.py)."""
    e, g, data = config1
    sd = _weights(e["weights"])

    def enc(name, L, n):
        lens, ids = ann_ref.read_cache(os.path.join(data, name))
        with torch.no_grad():
            return encoder_ref.rdot_nll_ln_emb(sd, torch.from_numpy(ids[:n]), encoder_ref.mask_from_lengths(lens[:n], L), n_layers=12).numpy()

    p, q, d = enc("passages", 128, 16), enc("train-query", 64, 16), enc("dev-query", 64, 16)
    assert np.abs(p - g["passage_emb16"]).max() <= 2e-5
    assert np.abs(q - g["query_emb16"]).max() <= 2e-5
    assert np.abs(d - g["dev_emb16"]).max() <= 2e-5
    # a stored score D[q, r] belongs to passage I[q, r]: check it wherever that passage is one of the 16 encoded here
    I, D = g["I_train"].astype(np.int64)[:16], g["D_train64"][:16]
    S = q.astype(np.float64) @ p.astype(np.float64).T
    hits = [(i, r) for i in range(16) for r in range(200) if I[i, r] < 16]
    for i, r in hits:
        assert abs(D[i, r] - S[i, I[i, r]]) <= 2e-2
    assert np.all(np.diff(g["D_train64"], axis=1) <= 0)  # lists are sorted
