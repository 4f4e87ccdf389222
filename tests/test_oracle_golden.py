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


def _checksum(sd):
    keys = sorted(sd.keys())
    return float(sum(sd[k].double().abs().sum().item() for k in keys[:: max(1, len(keys) // 16)]))


def _weights(meta, **kw):
    sd = encoder_ref.random_state_dict(seed=meta["seed"], n_layers=meta["n_layers"], ln_jitter=meta["ln_jitter"], **kw)
    if abs(_checksum(sd) - meta["checksum"]) > 1e-6 * meta["checksum"]:
        pytest.skip("torch RNG differs from the one that generated the golden vectors")
    return sd


def test_encoder_firstp_matches_reference(golden_dir):
    meta = _manifest(golden_dir)["encoder"]["firstp"]
    sd = _weights(meta)
    g = np.load(os.path.join(golden_dir, "encoder_firstp.npz"))
    ids, lens = torch.from_numpy(g["ids"]), g["lens"]
    with torch.no_grad():
        emb = encoder_ref.rdot_nll_ln_emb(sd, ids, encoder_ref.mask_from_lengths(lens, ids.shape[1]), n_layers=2)
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
