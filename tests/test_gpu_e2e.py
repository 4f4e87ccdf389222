"""End-to-end refresh on an MI355X: ance_amd.ann_data_gen.generate_new_ann on a toy MS MARCO-shaped
set, loading weights from an HF-style checkpoint dir, against (a) the oracle's post-search pipeline
run on the embeddings the GPU produced (files must be byte-identical under the same seed) and (b) the
reference's own dev NDCG within the encoder tolerance."""
import json
import os
import random
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _checksum(sd):
    keys = sorted(sd.keys())
    return float(sum(sd[k].double().abs().sum().item() for k in keys[:: max(1, len(keys) // 16)]))


def test_refresh_job_end_to_end(golden_dir, tmp_path):
    from safetensors.torch import save_file
    from ance_amd import ann_data_gen as adg
    from ance_amd import negatives
    from oracle import ann_ref, encoder_ref, search_ref, synth
    with open(os.path.join(golden_dir, "e2e_toy.json")) as f:
        e = json.load(f)
    sd = encoder_ref.random_state_dict(seed=e["weights"]["seed"], n_layers=e["weights"]["n_layers"],
                                       ln_jitter=e["weights"]["ln_jitter"])
    rng_ok = abs(_checksum(sd) - e["weights"]["checksum"]) <= 1e-6 * e["weights"]["checksum"]
    data = str(tmp_path / "data")
    synth.make_msmarco_like(data, **e["data"])
    ckpt = tmp_path / "train" / "checkpoint-100"
    ckpt.mkdir(parents=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt / "model.safetensors"))
    (ckpt / "scheduler.pt").write_text("commit marker")
    a = e["args"]
    out = str(tmp_path / "out")
    args = types.SimpleNamespace(
        data_dir=data, training_dir=str(tmp_path / "train"), init_model_dir="/nonexistent", last_checkpoint_dir="",
        output_dir=out, cache_dir=out, model_type="rdot_nll", end_output_num=0, max_seq_length=a["max_seq_length"],
        max_query_length=a["max_query_length"], ann_chunk_factor=a["ann_chunk_factor"], topk_training=a["topk_training"],
        negative_sample=a["negative_sample"], ann_measure_topk_mrr=a["ann_measure_topk_mrr"],
        only_keep_latest_embedding_file=False, inference=False, device=torch.device("cuda"), max_tokens=4096)
    random.seed(a["seed"])
    adg.ann_data_gen(args)  # poll loop: finds checkpoint-100, runs output 0, stops at end_output_num
    no, train_path, nd = adg.get_latest_ann_data(out)
    assert no == 0 and nd["checkpoint"].rstrip("/").endswith("checkpoint-100")
    assert adg.get_checkpoint_no(nd["checkpoint"]) == 100

    # (a) same embeddings -> oracle pipeline -> identical files
    from ance_amd.cache import TokenCache
    from ance_amd.encoder import load_model
    model = load_model("rdot_nll", str(ckpt), max_seq_length=a["max_seq_length"], max_tokens=4096)
    eng = adg.HipEngine()

    def emb(name, is_q):
        c = TokenCache(os.path.join(data, name))
        with c as cc:
            return eng.encode_cache(model, cc, 0, len(cc), is_q).cpu().numpy()

    dev_q, p_emb, train_q = emb("dev-query", True), emb("passages", False), emb("train-query", True)
    train_pos, dev_pos = negatives.load_positive_ids(data)
    out2 = str(tmp_path / "oracle_out")
    os.makedirs(out2)
    random.seed(a["seed"])
    ndcg_o, _, dev_I, I = ann_ref.refresh_from_embeddings(
        out2, 0, nd["checkpoint"], dev_q, np.arange(len(dev_q)), p_emb, np.arange(len(p_emb)), train_q,
        np.arange(len(train_q)), train_pos, dev_pos, a["topk_training"], a["negative_sample"], a["ann_chunk_factor"],
        a["ann_measure_topk_mrr"], search_ref.flat_ip_topk_chain)
    assert open(train_path).read() == open(os.path.join(out2, "ann_training_data_0")).read()
    assert abs(nd["ndcg"] - ndcg_o) < 1e-12

    # (b) against the reference's own run (fp32 CPU encoder): same NDCG up to encoder tolerance, and
    # the same negatives for almost every query (ann_measure_topk_mrr mode is deterministic)
    if rng_ok:
        assert abs(nd["ndcg"] - e["ann_ndcg_0"]["ndcg"]) < 0.05
        ref_lines = dict(l.split("\t", 1) for l in e["ann_training_data_0"].splitlines())
        got_lines = dict(l.split("\t", 1) for l in open(train_path).read().splitlines())
        assert set(ref_lines) == set(got_lines)
        same = sum(ref_lines[q] == got_lines[q] for q in ref_lines)
        assert same >= 0.8 * len(ref_lines), (same, len(ref_lines))

    # the reference consumer's line parser accepts the file (data/msmarco_data.py:338-343)
    for line in open(train_path):
        qid, pos, negs = line.rstrip("\n").split("\t")
        assert int(qid) >= 0 and int(pos) >= 0 and all(int(x) >= 0 for x in negs.split(","))

    # --inference dumps (seam B6)
    args.inference = True
    args.output_dir = str(tmp_path / "inf")
    adg.generate_new_ann(args, 0, str(ckpt) + "/", train_pos, dev_pos, 100)
    pe = np.load(os.path.join(args.output_dir, "passage_100__emb_p__data_obj_0.npy"))
    pi = np.load(os.path.join(args.output_dir, "passage_100__embid_p__data_obj_0.npy"))
    assert pe.shape == p_emb.shape and np.array_equal(pi, np.arange(len(p_emb))) and np.array_equal(pe, p_emb)
