"""Generates the golden vectors under tests/golden/ by running the REAL reference
(/root/reference, imported through oracle/ref_harness.py) in the build container.

    python tests/golden/make_golden.py

The reference ships no tests or fixtures of its own, so these vectors are what
pins the oracle: they are outputs of the reference's own classes / functions on seeded inputs.
Weights are not stored (a RoBERTa embedding table is 154 MB): they are regenerated from
``oracle.encoder_ref.det_state_dict(seed)`` -- a counter-based generator written out in the oracle (integer hashing +
exact float conversions), so the weights do not depend on any torch / NumPy random stream -- and the manifest records
their sha256.
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ann_ref, encoder_ref, ref_harness, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sd_checksum(sd):
    return encoder_ref.state_dict_sha256(sd)


def load_into(model, sd):
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [k for k in missing if not (k.startswith("classifier.") or "pooler" in k or "position_ids" in k)]
    assert not bad and not unexpected, (bad, unexpected)


def golden_encoder():
    rng = np.random.default_rng(2024)
    out = {}
    # FirstP / query encoder (model/models.py:149-157), 2 layers, non-trivial LN/bias parameters
    sd = encoder_ref.det_state_dict(seed=11, n_layers=2, ln_jitter=0.1)
    m = ref_harness.build_reference_model("rdot_nll", n_layers=2, seed=0)
    load_into(m, sd)
    L = 128
    lens = np.array([1, 2, 8, 31, 32, 33, 64, 70, 100, 127, 128, 128, 5, 17, 96, 77], dtype=np.int32)
    ids = synth.make_records(rng, len(lens), L, lens.astype(np.int64))
    with torch.no_grad():
        emb = m.body_emb(torch.from_numpy(ids).long(), encoder_ref.mask_from_lengths(lens, L))
    out["firstp"] = dict(gen="det", seed=11, n_layers=2, ln_jitter=0.1, checksum=sd_checksum(sd))
    np.savez_compressed(os.path.join(OUT, "encoder_firstp.npz"), ids=ids, lens=lens, emb=emb.numpy())

    # MaxP body encoder (model/models.py:165-199): 4 x 512 chunks incl. all-pad chunks
    sd2 = encoder_ref.det_state_dict(seed=12, n_layers=1, ln_jitter=0.05)
    m2 = ref_harness.build_reference_model("rdot_nll_multi_chunk", n_layers=1, seed=0)
    load_into(m2, sd2)
    lens2 = np.array([2048, 1500, 513, 512, 40, 1025], dtype=np.int32)
    ids2 = synth.make_records(rng, len(lens2), 2048, lens2.astype(np.int64))
    with torch.no_grad():
        emb2 = m2.body_emb(torch.from_numpy(ids2).long(), encoder_ref.mask_from_lengths(lens2, 2048))
    out["maxp"] = dict(gen="det", seed=12, n_layers=1, ln_jitter=0.05, checksum=sd_checksum(sd2))
    np.savez_compressed(os.path.join(OUT, "encoder_maxp.npz"), ids=ids2, lens=lens2, emb=emb2.numpy())

    # DPR / BERT tower (model/models.py:223-259): raw [CLS]
    sd3 = encoder_ref.det_state_dict(kind="bert", seed=13, n_layers=2, vocab=30522, max_pos=512, head=False,
                                        prefixes=("ctx_model.",), ln_jitter=0.1)
    m3 = ref_harness.build_reference_model("bert", n_layers=2, seed=0)
    load_into(m3, {k[len("ctx_model."):]: v for k, v in sd3.items()})
    lens3 = np.array([256, 3, 100, 255, 64, 17, 1, 200], dtype=np.int32)
    ids3 = rng.integers(1000, 30522, size=(len(lens3), 256)).astype(np.int32)
    ids3[:, 0] = 101
    ids3[np.arange(len(lens3)), lens3 - 1] = 102
    ids3[:, 0] = 101
    ids3 = np.where(np.arange(256)[None, :] < lens3[:, None], ids3, 0).astype(np.int32)
    with torch.no_grad():
        t = torch.from_numpy(ids3).long()
        emb3 = m3(t, (t != 0).long())[1]
    out["bert"] = dict(gen="det", seed=13, n_layers=2, ln_jitter=0.1, checksum=sd_checksum(sd3))
    np.savez_compressed(os.path.join(OUT, "encoder_bert.npz"), ids=ids3, lens=lens3, emb=emb3.numpy())

    # FULL DEPTH: RobertaDot_NLL_LN.body_emb itself (model/models.py:149-157) at roberta-base's 12 layers -- the depth every
    # headline number is quoted at; lengths 1, L and the 32 / 64 / 96 tile edges
    sd4 = encoder_ref.det_state_dict(seed=14, n_layers=12, ln_jitter=0.1)
    m4 = ref_harness.build_reference_model("rdot_nll", n_layers=12, seed=0)
    load_into(m4, sd4)
    lens4 = np.array([1, 2, 31, 32, 33, 63, 64, 65, 96, 97, 127, 128, 128, 70, 9, 50], dtype=np.int32)
    ids4 = synth.make_records(rng, len(lens4), L, lens4.astype(np.int64))
    with torch.no_grad():
        emb4 = m4.body_emb(torch.from_numpy(ids4).long(), encoder_ref.mask_from_lengths(lens4, L))
    out["firstp12"] = dict(gen="det", seed=14, n_layers=12, ln_jitter=0.1, checksum=sd_checksum(sd4))
    np.savez_compressed(os.path.join(OUT, "encoder_firstp12.npz"), ids=ids4, lens=lens4, emb=emb4.numpy())
    return out


def golden_encoder12():
    """FULL DEPTH (12 layers) for the other two towers and for the long-sequence FirstP case: the reference's OWN
    classes at the depth configs 3-5's numbers are quoted at.  Separate from golden_encoder so that the older fixtures stay
    byte-stable."""
    rng = np.random.default_rng(2025)
    out = {}
    # MaxP body encoder, RobertaDot_CLF_ANN_NLL_MultiChunk.body_emb (model/models.py:165-199): 6 documents straddling the chunk
    # borders (511 / 512 / 513 / 1024 / 1025 tokens), one full document, all-pad chunks behind the short ones
    sd = encoder_ref.det_state_dict(seed=31, n_layers=12, ln_jitter=0.1)
    m = ref_harness.build_reference_model("rdot_nll_multi_chunk", n_layers=12, seed=0)
    load_into(m, sd)
    lens = np.array([2048, 1025, 1024, 513, 512, 511], dtype=np.int32)
    ids = synth.make_records(rng, len(lens), 2048, lens.astype(np.int64))
    with torch.no_grad():
        emb = m.body_emb(torch.from_numpy(ids).long(), encoder_ref.mask_from_lengths(lens, 2048))
    out["maxp12"] = dict(gen="det", seed=31, n_layers=12, ln_jitter=0.1, checksum=sd_checksum(sd))
    np.savez_compressed(os.path.join(OUT, "encoder_maxp12.npz"), ids=ids, lens=lens, emb=emb.numpy())
    del m

    # DPR / BERT tower, HFBertEncoder (model/models.py:223-244): raw [CLS], L = 256
    sd3 = encoder_ref.det_state_dict(kind="bert", seed=33, n_layers=12, vocab=30522, max_pos=512, head=False,
                                        prefixes=("ctx_model.",), ln_jitter=0.1)
    m3 = ref_harness.build_reference_model("bert", n_layers=12, seed=0)
    load_into(m3, {k[len("ctx_model."):]: v for k, v in sd3.items()})
    lens3 = np.array([256, 3, 100, 255, 64, 17, 1, 200, 129, 128], dtype=np.int32)
    ids3 = rng.integers(1000, 30522, size=(len(lens3), 256)).astype(np.int32)
    ids3[np.arange(len(lens3)), lens3 - 1] = 102
    ids3[:, 0] = 101
    ids3 = np.where(np.arange(256)[None, :] < lens3[:, None], ids3, 0).astype(np.int32)
    with torch.no_grad():
        t = torch.from_numpy(ids3).long()
        emb3 = m3(t, (t != 0).long())[1]
    out["bert12"] = dict(gen="det", seed=33, n_layers=12, ln_jitter=0.1, checksum=sd_checksum(sd3))
    np.savez_compressed(os.path.join(OUT, "encoder_bert12.npz"), ids=ids3, lens=lens3, emb=emb3.numpy())
    del m3

    # FirstP at seq_len 512 (BASELINE configs[2]), RobertaDot_NLL_LN.body_emb (model/models.py:149-157): the 256-key borders of the
    # long-sequence attention path
    sd5 = encoder_ref.det_state_dict(seed=35, n_layers=12, ln_jitter=0.1)
    m5 = ref_harness.build_reference_model("rdot_nll", n_layers=12, seed=0)
    load_into(m5, sd5)
    lens5 = np.array([1, 255, 256, 257, 511, 512], dtype=np.int32)
    ids5 = synth.make_records(rng, len(lens5), 512, lens5.astype(np.int64))
    with torch.no_grad():
        emb5 = m5.body_emb(torch.from_numpy(ids5).long(), encoder_ref.mask_from_lengths(lens5, 512))
    out["firstp12_L512"] = dict(gen="det", seed=35, n_layers=12, ln_jitter=0.1, checksum=sd_checksum(sd5))
    np.savez_compressed(os.path.join(OUT, "encoder_firstp12_L512.npz"), ids=ids5, lens=lens5, emb=emb5.numpy())
    return out


def golden_postsearch():
    """GenerateNegativePassaageID / EvalDevQuery of the reference on seeded neighbour lists."""
    ref = ref_harness.load_reference()
    G = ref.driver
    rng = np.random.default_rng(7)
    n_rows, chunks, nq, k = 4000, 4, 60, 50
    p2id = (np.arange(n_rows) // chunks).astype(np.int64)  # MaxP-style: several rows per pid
    q2id = np.arange(100, 100 + nq, dtype=np.int64)
    I = np.stack([rng.choice(n_rows, size=k, replace=False) for _ in range(nq)]).astype(np.int64)
    train_pos = {int(q): int(p2id[I[i, rng.integers(0, 12)]]) for i, q in enumerate(q2id)}
    eff = set(q2id.tolist())
    cases = {}
    for topk in (False, True):
        args = types.SimpleNamespace(ann_measure_topk_mrr=topk, negative_sample=7, rank=0)
        random.seed(123)
        neg = G.GenerateNegativePassaageID(args, q2id, p2id, train_pos, I, eff)
        cases["neg_topk%d" % int(topk)] = {str(kk): [int(v) for v in vv] for kk, vv in neg.items()}
    dev_q2id = np.arange(nq, dtype=np.int64)
    dev_pos = {}
    for i in range(nq):
        rel = {}
        for j in rng.choice(60, size=int(rng.integers(1, 4)), replace=False):
            pid = int(p2id[I[i, j]]) if j < k else int(rng.integers(0, n_rows // chunks))
            rel[pid] = int(rng.integers(1, 3))
        dev_pos[i] = rel
    args = types.SimpleNamespace(rank=0)
    ndcg, cnt = G.EvalDevQuery(args, dev_q2id, p2id, dev_pos, I)
    np.savez_compressed(os.path.join(OUT, "postsearch.npz"), p2id=p2id, q2id=q2id, I=I)
    with open(os.path.join(OUT, "postsearch.json"), "w") as f:
        json.dump(dict(train_pos={str(a): b for a, b in train_pos.items()},
                       dev_pos={str(a): {str(c): d for c, d in b.items()} for a, b in dev_pos.items()},
                       cases=cases, ndcg=ndcg, ndcg_cnt=cnt, negative_sample=7, seed=123), f)
    return dict(ndcg=ndcg, cnt=cnt)


def golden_end_to_end():
    """The reference's own generate_new_ann on a toy set, CPU, 2-layer model."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="ance_golden_")
    try:
        data = os.path.join(tmp, "data")
        synth.make_msmarco_like(data, n_passages=400, n_train=60, n_dev=20, L=64, Lq=32, seed=77)
        sd = encoder_ref.det_state_dict(seed=21, n_layers=2, ln_jitter=0.1)
        m = ref_harness.build_reference_model("rdot_nll", n_layers=2, seed=0)
        load_into(m, sd)
        outd = os.path.join(tmp, "out")
        res = ref_harness.run_generate_new_ann(data, outd, m, output_num=0, checkpoint_path="/x/checkpoint-100/",
                                               step=100, seed=5, max_seq_length=64, max_query_length=32,
                                               topk_training=40, negative_sample=6, ann_chunk_factor=2,
                                               ann_measure_topk_mrr=True)
        with open(os.path.join(outd, "ann_training_data_0")) as f:
            lines = f.read()
        with open(os.path.join(outd, "ann_ndcg_0")) as f:
            nd = json.load(f)
        with open(os.path.join(OUT, "e2e_toy.json"), "w") as f:
            json.dump(dict(weights=dict(gen="det", seed=21, n_layers=2, ln_jitter=0.1, checksum=sd_checksum(sd)),
                           data=dict(n_passages=400, n_train=60, n_dev=20, L=64, Lq=32, seed=77),
                           args=dict(max_seq_length=64, max_query_length=32, topk_training=40, negative_sample=6,
                                     ann_chunk_factor=2, ann_measure_topk_mrr=True, seed=5, output_num=0,
                                     checkpoint_path="/x/checkpoint-100/"),
                           ann_training_data_0=lines, ann_ndcg_0=nd, result=[res[0], res[1]]), f)
        return dict(ndcg=nd["ndcg"], lines=lines.count("\n"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def golden_end_to_end_maxp():
    """Config 4: the reference's own generate_new_ann with RobertaDot_CLF_ANN_NLL_MultiChunk (MaxP, 2048 = 4 x 512
    tokens, drivers/run_ann_data_gen.py:183-189 one slab of vectors per chunk; :383-384,419-423 duplicate pids skipped)
    on a toy document set whose lengths straddle the 512 / 1024 / 1536 chunk boundaries, so that all-pad chunks -- one
    identical vector per such chunk -- enter the top-k lists.  CPU, 1-layer model."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="ance_golden_maxp_")
    try:
        data = os.path.join(tmp, "data")
        dargs = dict(n_passages=56, n_train=24, n_dev=8, L=2048, Lq=32, seed=79, len_median=250, len_sigma=1.1, dup_frac=0.0)
        synth.make_msmarco_like(data, **dargs)
        wargs = dict(seed=23, n_layers=1, ln_jitter=0.1)
        sd = encoder_ref.det_state_dict(**wargs)
        m = ref_harness.build_reference_model("rdot_nll_multi_chunk", n_layers=1, seed=0)
        load_into(m, sd)
        outd = os.path.join(tmp, "out")
        jargs = dict(max_seq_length=2048, max_query_length=32, topk_training=40, negative_sample=6, ann_chunk_factor=1,
                     ann_measure_topk_mrr=True, model_type="rdot_nll_multi_chunk")
        res = ref_harness.run_generate_new_ann(data, outd, m, output_num=0, checkpoint_path="/x/checkpoint-100/", step=100,
                                               seed=5, **jargs)
        with open(os.path.join(outd, "ann_training_data_0")) as f:
            lines = f.read()
        with open(os.path.join(outd, "ann_ndcg_0")) as f:
            nd = json.load(f)
        with open(os.path.join(OUT, "e2e_maxp.json"), "w") as f:
            json.dump(dict(weights=dict(gen="det", checksum=sd_checksum(sd), **wargs), data=dargs,
                           args=dict(seed=5, output_num=0, checkpoint_path="/x/checkpoint-100/", per_gpu_eval_batch_size=16, **jargs),
                           ann_training_data_0=lines, ann_ndcg_0=nd, result=[res[0], res[1]]), f)
        return dict(ndcg=nd["ndcg"], lines=lines.count("\n"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def golden_config1():
    """BASELINE.json configs[0] at its stated size: the reference's own generate_new_ann
    (drivers/run_ann_data_gen.py:231-336) on 10,000 passages / 1,000 train / 200 dev queries, L = 128, Lq = 64, roberta-base
    depth (12 layers), top-200, 20 negatives, 1 % planted duplicate passages -- twice: (a) --ann_measure_topk_mrr (the
    deterministic selection), ann_chunk_factor 1; (b) the default selection (random.shuffle under random.seed(0)),
    ann_chunk_factor 5, output_num 2 (third query chunk).  The stand-in faiss index records what the reference asked of it
    and what it returned (``I``); run (b) reuses run (a)'s embeddings through a memoising wrapper around the reference's own
    StreamInferenceDoc (same model, same caches: every line of the reference executes in run (a)).  ~10 minutes of CPU."""
    import shutil
    import tempfile
    import time
    from oracle import search_ref
    ref = ref_harness.load_reference()
    G = ref.driver
    tmp = tempfile.mkdtemp(prefix="ance_golden_c1_")
    try:
        data = os.path.join(tmp, "data")
        dargs = dict(n_passages=10000, n_train=1000, n_dev=200, L=128, Lq=64, seed=1234, dup_frac=0.01)
        synth.make_msmarco_like(data, **dargs)
        wargs = dict(seed=42, n_layers=12, ln_jitter=0.1)
        sd = encoder_ref.det_state_dict(**wargs)
        m = ref_harness.build_reference_model("rdot_nll", n_layers=12, seed=0)
        load_into(m, sd)

        calls = []

        class RecordingIndex(search_ref.OracleIndexFlatIP):
            def search(self, q, k):
                D, I = super().search(q, k)
                calls.append((np.array(q, dtype=np.float32), int(k), D.copy(), I.copy(), self._x))
                return D, I

        sys.modules["faiss"].IndexFlatIP = RecordingIndex
        real_stream = G.StreamInferenceDoc
        memo = {}

        def stream_memo(args, model, fn, prefix, f, is_query_inference=True):
            if prefix not in memo:
                memo[prefix] = real_stream(args, model, fn, prefix, f, is_query_inference=is_query_inference)
            return memo[prefix]

        G.StreamInferenceDoc = stream_memo
        runs = {}
        t0 = time.time()
        for name, jargs, seed, output_num in (
                ("topk", dict(ann_measure_topk_mrr=True, ann_chunk_factor=1), 0, 0),
                ("shuffle", dict(ann_measure_topk_mrr=False, ann_chunk_factor=5), 0, 2)):
            outd = os.path.join(tmp, "out_" + name)
            full = dict(max_seq_length=128, max_query_length=64, topk_training=200, negative_sample=20, **jargs)
            res = ref_harness.run_generate_new_ann(data, outd, m, output_num=output_num, checkpoint_path="/x/checkpoint-100/",
                                                   step=100, seed=seed, **full)
            with open(os.path.join(outd, "ann_training_data_%d" % output_num)) as f:
                lines = f.read()
            with open(os.path.join(outd, "ann_ndcg_%d" % output_num)) as f:
                nd = json.load(f)
            runs[name] = dict(args=dict(seed=seed, output_num=output_num, checkpoint_path="/x/checkpoint-100/",
                                        per_gpu_eval_batch_size=16, **full),
                              ann_training_data=lines, ann_ndcg=nd, result=[res[0], res[1]])
        G.StreamInferenceDoc = real_stream
        sys.modules["faiss"].IndexFlatIP = search_ref.OracleIndexFlatIP
        # calls: run (a) dev (k = 100), run (a) train (k = 200, all 1,000 queries), run (b) dev, run (b) train chunk
        (qd, kd, Dd, Id, X), (qt, kt, Dt, It, _) = calls[0], calls[1]
        assert kd == 100 and kt == 200 and It.shape == (1000, 200) and Id.shape == (200, 100) and X.shape == (10000, 768)
        assert np.array_equal(calls[2][3], Id) and calls[3][3].shape == (200, 200)
        # (the chunk's lists come from an sgemm of another shape: BLAS may round its scores differently, so they are stored)
        np.savez_compressed(os.path.join(OUT, "e2e_config1.npz"), I_train=It.astype(np.uint16), I_dev=Id.astype(np.uint16),
                            I_train_chunk2=calls[3][3].astype(np.uint16),
                            D_train64=Dt[:64], D_dev64=Dd[:64], passage_emb16=X[:16], query_emb16=qt[:16], dev_emb16=qd[:16])
        with open(os.path.join(OUT, "e2e_config1.json"), "w") as f:
            json.dump(dict(weights=dict(gen="det", checksum=sd_checksum(sd), **wargs), data=dargs, runs=runs,
                           cpu_seconds=round(time.time() - t0, 1), threads=torch.get_num_threads()), f)
        return dict(ndcg=runs["topk"]["ann_ndcg"]["ndcg"], lines_topk=runs["topk"]["ann_training_data"].count("\n"),
                    lines_shuffle=runs["shuffle"]["ann_training_data"].count("\n"), cpu_seconds=round(time.time() - t0, 1))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def golden_nll():
    """The reference's training forward itself -- RobertaDot_NLL_LN.forward (NLL, model/models.py:57-81) and
    RobertaDot_CLF_ANN_NLL_MultiChunk.forward (NLL_MultiChunk, :84-134) -- on seeded triplets: the loss it returns, and the
    embeddings of its own query_emb / body_emb for the logits (2 layers / 1 layer, deterministic weights)."""
    rng = np.random.default_rng(404)
    out = {}
    arrays = {}
    # FirstP: 12 triplets, Lq = 32, L = 64
    w1 = dict(seed=51, n_layers=2, ln_jitter=0.1)
    sd = encoder_ref.det_state_dict(**w1)
    m = ref_harness.build_reference_model("rdot_nll", n_layers=2, seed=0)
    load_into(m, sd)
    n = 12
    ql = synth.lognormal_lengths(rng, n, 9, 0.35, 4, 32)
    al = synth.lognormal_lengths(rng, n, 40, 0.45, 8, 64)
    bl = synth.lognormal_lengths(rng, n, 40, 0.45, 8, 64)
    qi, ai, bi = synth.make_records(rng, n, 32, ql), synth.make_records(rng, n, 64, al), synth.make_records(rng, n, 64, bl)
    ai[:6, 1:6] = qi[:6, 1:6]  # the positive shares tokens with the query for half of the triplets
    T = lambda x: torch.from_numpy(x).long()
    M = encoder_ref.mask_from_lengths
    with torch.no_grad():
        loss = m(T(qi), M(ql, 32), T(ai), M(al, 64), T(bi), M(bl, 64))[0]
        q, a, b = m.query_emb(T(qi), M(ql, 32)), m.body_emb(T(ai), M(al, 64)), m.body_emb(T(bi), M(bl, 64))
    arrays.update(f_q_ids=qi, f_q_len=ql, f_a_ids=ai, f_a_len=al, f_b_ids=bi, f_b_len=bl, f_q=q.numpy(), f_a=a.numpy(), f_b=b.numpy())
    out["firstp"] = dict(weights=dict(gen="det", checksum=sd_checksum(sd), **w1), loss=float(loss))
    # MaxP: 6 triplets, documents of 1-4 chunks (all-pad chunks are biased out by -9999)
    w2 = dict(seed=52, n_layers=1, ln_jitter=0.1)
    sd2 = encoder_ref.det_state_dict(**w2)
    m2 = ref_harness.build_reference_model("rdot_nll_multi_chunk", n_layers=1, seed=0)
    load_into(m2, sd2)
    n2 = 6
    ql2 = synth.lognormal_lengths(rng, n2, 9, 0.35, 4, 32)
    al2 = np.array([2048, 700, 512, 30, 1300, 513], dtype=np.int64)
    bl2 = np.array([100, 2048, 1025, 513, 40, 1536], dtype=np.int64)
    qi2, ai2, bi2 = synth.make_records(rng, n2, 32, ql2), synth.make_records(rng, n2, 2048, al2), synth.make_records(rng, n2, 2048, bl2)
    with torch.no_grad():
        loss2 = m2(T(qi2), M(ql2, 32), T(ai2), M(al2, 2048), T(bi2), M(bl2, 2048))[0]
        q2, a2, b2 = m2.query_emb(T(qi2), M(ql2, 32)), m2.body_emb(T(ai2), M(al2, 2048)), m2.body_emb(T(bi2), M(bl2, 2048))
    arrays.update(m_q_ids=qi2, m_q_len=ql2, m_a_ids=ai2, m_a_len=al2, m_b_ids=bi2, m_b_len=bl2, m_q=q2.numpy(), m_a=a2.numpy(),
                  m_b=b2.numpy())
    out["maxp"] = dict(weights=dict(gen="det", checksum=sd_checksum(sd2), **w2), loss=float(loss2))
    np.savez_compressed(os.path.join(OUT, "nll.npz"), **arrays)
    with open(os.path.join(OUT, "nll.json"), "w") as f:
        json.dump(out, f, indent=1)
    return {k: v["loss"] for k, v in out.items()}


def golden_dpr():
    """validate / GenerateNegativePassaageID / has_answer of the reference's DPR driver on synthetic
    passages and answers (unicode, punctuation, multi-token and empty-token answers)."""
    import importlib
    ref_harness.load_reference()
    D = importlib.import_module("run_ann_data_gen_dpr")
    rng = np.random.default_rng(31)
    vocab = ["the", "Paris", "paris", "New", "York", "new", "york", "café", "CAFÉ", "naïve", "1969", "Apollo", "11",
             "moon", "U.S.", "u.s.", "state-of-the-art", "O'Neil", "rock", "&", "roll", "Zürich", "3.14", "pi", "ﬁ",
             "Ａ", "İstanbul", "istanbul", "élan", "e\u0301lan", ",", ".", "(", ")", "dog", "cat", "blue", "red"]
    n_p = 400
    passages = {}
    for pid in range(n_p):
        words = [vocab[int(j)] for j in rng.integers(0, len(vocab), size=int(rng.integers(5, 40)))]
        passages[pid] = (" ".join(words), "title %d" % pid)
    answers_pool = [["Paris"], ["new york"], ["New  York"], ["café"], ["Apollo 11"], ["u.s."], ["rock & roll"],
                    ["state-of-the-art"], ["3.14"], ["Zürich", "zurich"], ["istanbul"], ["elan"], ["moon", "dog"],
                    ["blue cat"], ["."], [""], ["O'Neil"], ["red dog cat"], ["ﬁ"], ["A"], ["naive"]]
    nq, k = 50, 30
    answers = [answers_pool[int(j)] for j in rng.integers(0, len(answers_pool), size=nq)]
    p2id = np.arange(n_p, dtype=np.int64)
    q2id = np.arange(nq, dtype=np.int64)
    I = np.stack([rng.choice(n_p, size=k, replace=False) for _ in range(nq)]).astype(np.int64)
    pos = [int(I[i, rng.integers(0, 6)]) for i in range(nq)]
    hits = D.validate(passages, answers, I, q2id, p2id)
    args = types.SimpleNamespace(negative_sample=9)
    neg = D.GenerateNegativePassaageID(args, passages, answers, q2id, p2id, I, pos)
    from utils.dpr_utils import SimpleTokenizer, has_answer
    tok = SimpleTokenizer()
    single = [[bool(has_answer(a, passages[pid][0], tok)) for pid in range(40)] for a in answers_pool]
    with open(os.path.join(OUT, "dpr_postsearch.json"), "w") as f:
        json.dump(dict(passages={str(k_): v for k_, v in passages.items()}, answers=answers, answers_pool=answers_pool,
                       I=I.tolist(), pos=pos, negative_sample=9, hits=hits,
                       neg={str(k_): [int(x) for x in v] for k_, v in neg.items()}, single=single), f)
    return dict(top20=hits[19], n_neg=sum(len(v) for v in neg.values()))


def golden_preprocess():
    """The reference's own data/msmarco_data.py ``preprocess`` (passage and document layouts) on seeded
    raw TSVs with oracle.synth.ToyTokenizer standing in for the pretrained tokenizer: sha256 of every
    merged output (the split intermediates are covered through the merge)."""
    import hashlib
    import shutil
    import tempfile
    out = {}
    for data_type in (1, 0):
        tmp = tempfile.mkdtemp(prefix="ance_golden_pp_")
        try:
            raw, dst = os.path.join(tmp, "raw"), os.path.join(tmp, "out")
            synth.make_raw_msmarco(raw, data_type, n_passages=70, n_train=25, n_dev=9, seed=5)
            ref_harness.run_reference_preprocess(raw, dst, data_type, synth.ToyTokenizer, max_seq_length=16,
                                                 max_query_length=8)
            files = {}
            for f in sorted(os.listdir(dst)):
                if "_split" in f:
                    continue
                with open(os.path.join(dst, f), "rb") as fh:
                    files[f] = hashlib.sha256(fh.read()).hexdigest()
            out[str(data_type)] = files
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "preprocess.json"), "w") as f:
        json.dump(dict(raw=dict(n_passages=70, n_train=25, n_dev=9, seed=5), max_seq_length=16, max_query_length=8,
                       sha256=out), f, indent=1)
    return {k: len(v) for k, v in out.items()}


def golden_dpr_preprocess():
    """The reference's own data/DPR_data.py ``preprocess`` (NQ and TriviaQA layouts) on seeded raw inputs
    with oracle.synth.ToyBertTokenizer: sha256 of every merged output."""
    import hashlib
    import shutil
    import tempfile
    out = {}
    for data_type in (0, 1):
        tmp = tempfile.mkdtemp(prefix="ance_golden_dpr_")
        try:
            wiki, qd, ad = synth.make_raw_dpr(tmp, n_passages=90, n_nq=14, n_trivia=11, seed=9)
            dst = os.path.join(tmp, "out") + "/"
            ref_harness.run_reference_dpr_preprocess(wiki, qd, ad, dst, data_type, synth.ToyBertTokenizer, max_seq_length=24)
            files = {}
            for f in sorted(os.listdir(dst)):
                if "_split" in f:
                    continue
                with open(os.path.join(dst, f), "rb") as fh:
                    files[f] = hashlib.sha256(fh.read()).hexdigest()
            out[str(data_type)] = files
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "dpr_preprocess.json"), "w") as f:
        json.dump(dict(raw=dict(n_passages=90, n_nq=14, n_trivia=11, seed=9), max_seq_length=24, sha256=out), f, indent=1)
    return {k: len(v) for k, v in out.items()}


def golden_metrics():
    """evaluation/"Calculate Metrics.ipynb" cell 8 (EvalDevQuery) on seeded neighbour lists: a MaxP-style
    row -> pid map, graded judgements incl. rel 0, unjudged holes, queries without any relevant hit."""
    E = ref_harness.notebook_eval_dev_query()
    rng = np.random.default_rng(3)
    n_rows, chunks, nq = 3000, 3, 40
    p2id = (np.arange(n_rows) // chunks).astype(np.int64)
    q2id = np.arange(nq, dtype=np.int64)
    I = np.stack([rng.choice(n_rows, size=150, replace=False) for _ in range(nq)])
    qrels = {}
    for i in range(nq + 5):  # five reference queries are never ranked (they count in the MS MARCO MRR denominator)
        d = {}
        for j in rng.choice(160, size=int(rng.integers(1, 5)), replace=False):
            pid = int(p2id[I[i % nq, j]]) if j < 150 else int(rng.integers(0, n_rows // chunks))
            d[pid] = int(rng.integers(0, 4))
        qrels[i] = d
    out = {}
    for topN in (100, 1000):
        r = E(q2id, p2id, qrels, I, topN)
        out[str(topN)] = dict(ndcg=r[0], queries=r[1], map=r[2], mrr=r[3], recall=r[4], hole_rate=r[5], ms_mrr=r[6],
                              ahole_rate=r[7])
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), p2id=p2id, q2id=q2id, I=I)
    with open(os.path.join(OUT, "metrics.json"), "w") as f:
        json.dump(dict(qrels={str(a): {str(b): c for b, c in d.items()} for a, d in qrels.items()}, results=out), f)
    return {k: v["ndcg"] for k, v in out.items()}


if __name__ == "__main__":
    torch.set_num_threads(8)
    makers = dict(encoder=golden_encoder, encoder12=golden_encoder12, postsearch=golden_postsearch, e2e=golden_end_to_end, e2e_maxp=golden_end_to_end_maxp,
                  config1=golden_config1, nll=golden_nll,
                  dpr=golden_dpr, preprocess=golden_preprocess, metrics=golden_metrics, dpr_preprocess=golden_dpr_preprocess)
    which = sys.argv[1:] or list(makers)  # `make_golden.py e2e_maxp` regenerates one piece and its manifest entry
    mpath = os.path.join(OUT, "manifest.json")
    info = {}
    if sys.argv[1:] and os.path.exists(mpath):
        with open(mpath) as f:
            info = json.load(f)
    for name in which:
        info[name] = makers[name]()
    info["torch"], info["numpy"] = torch.__version__, np.__version__
    with open(mpath, "w") as f:
        json.dump(info, f, indent=1)
    print(json.dumps({k: info[k] for k in which}, indent=1))
