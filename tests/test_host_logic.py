"""Host-side product logic (ance_amd.negatives / cache / ann_data_gen helpers) against the golden
vectors of the reference and against the oracle restatement -- CPU only, no kernel calls."""
import json
import os
import random
import types

import numpy as np
import pytest

from ance_amd import ann_data_gen as adg
from ance_amd import negatives
from ance_amd.cache import TokenCache, shard_range
from oracle import ann_ref, synth


def _postsearch(golden_dir):
    g = np.load(os.path.join(golden_dir, "postsearch.npz"))
    with open(os.path.join(golden_dir, "postsearch.json")) as f:
        j = json.load(f)
    train_pos = {int(k): v for k, v in j["train_pos"].items()}
    dev_pos = {int(k): {int(a): b for a, b in v.items()} for k, v in j["dev_pos"].items()}
    return g, j, train_pos, dev_pos


@pytest.mark.parametrize("topk", [False, True])
def test_negatives_match_reference_golden(golden_dir, topk, capsys):
    g, j, train_pos, _ = _postsearch(golden_dir)
    random.seed(j["seed"])
    neg = negatives.generate_negative_passage_ids(g["q2id"], g["p2id"], train_pos, g["I"], set(g["q2id"].tolist()),
                                                  j["negative_sample"], topk)
    want = {int(k): v for k, v in j["cases"]["neg_topk%d" % int(topk)].items()}
    assert neg == want
    if topk:
        assert "ANN MRR:" in capsys.readouterr().out


def test_dev_ndcg_matches_reference_golden(golden_dir):
    g, j, _, dev_pos = _postsearch(golden_dir)
    ndcg, cnt = negatives.eval_dev_query(np.arange(g["I"].shape[0]), g["p2id"], dev_pos, g["I"])
    assert cnt == j["ndcg_cnt"] and abs(ndcg - j["ndcg"]) < 1e-12


def test_negatives_and_ndcg_match_oracle_randomised():
    rng = np.random.default_rng(99)
    for trial in range(5):
        n_rows, chunks, nq, k = 900, int(rng.integers(1, 4)), 40, 30
        p2id = np.arange(n_rows) // chunks
        q2id = rng.permutation(200)[:nq]
        I = np.stack([rng.choice(n_rows, size=k, replace=False) for _ in range(nq)])
        pos = {int(q): int(p2id[I[i, rng.integers(0, k)]]) if rng.random() < 0.7 else int(rng.integers(0, 300))
               for i, q in enumerate(q2id)}
        eff = set(q2id[: nq - 3].tolist())
        for topk in (False, True):
            random.seed(trial)
            a = negatives.generate_negative_passage_ids(q2id, p2id, pos, I, eff, 5, topk, verbose=False)
            random.seed(trial)
            b, _ = ann_ref.generate_negative_passage_ids(q2id, p2id, pos, I, eff, 5, topk)
            assert a == {int(k_): [int(x) for x in v] for k_, v in b.items()}
        dev = {int(q): {int(p2id[I[i, j]]): int(rng.integers(1, 4)) for j in rng.choice(k, 2, replace=False)}
               for i, q in enumerate(q2id) if i % 5}
        x = negatives.eval_dev_query(q2id, p2id, dev, I)
        y = ann_ref.eval_dev_query(q2id, p2id, dev, I)
        assert x[1] == y[1] and abs(x[0] - y[0]) < 1e-12


def test_query_chunk_matches_reference_rule():
    for nq in (0, 1, 7, 100, 502939):
        for cf in (1, 2, 5, 7):
            for out_num in range(0, 9):
                assert negatives.query_chunk(nq, out_num, cf) == ann_ref.query_chunk(nq, out_num, cf)
    with pytest.raises(ZeroDivisionError):
        negatives.query_chunk(10, 0, 0)
    # chunks tile the query set
    covered = []
    for out_num in range(5):
        s, e = negatives.query_chunk(103, out_num, 5)
        covered += list(range(s, e))
    assert covered == list(range(103))


def test_token_cache_roundtrip(tmp_path):
    rng = np.random.default_rng(5)
    lens = np.array([0, 1, 7, 16, 3], dtype=np.int64)
    ids = synth.make_records(rng, 5, 16, lens)
    path = str(tmp_path / "passages")
    synth.write_cache(path, ids, lens)
    c = TokenCache(path)
    assert len(c) == 5 and c.record_size == 4 + 64
    with c as cc:
        assert np.array_equal(cc.lengths(), lens)
        assert np.array_equal(cc.ids(), ids)
        pl, p = cc[2]
        assert pl == 7 and np.array_equal(p, ids[2])
        raw = cc.records(1, 3)
        assert raw.shape == (2, 68) and bytes(raw[1, :4]) == (7).to_bytes(4, "big")
    # oracle reader agrees
    l2, i2 = ann_ref.read_cache(path)
    assert np.array_equal(l2, lens) and np.array_equal(i2, ids)
    with pytest.raises(IndexError):
        with TokenCache(path) as cc:
            cc[99]


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 9, 8841823):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            for a, b in zip(parts, parts[1:]):
                assert a[1] == b[0]


def test_latest_ann_data_and_checkpoint(tmp_path):
    out = tmp_path / "ann"
    assert adg.get_latest_ann_data(str(out)) == (-1, None, None)
    out.mkdir()
    assert adg.get_latest_ann_data(str(out)) == (-1, None, None)
    for n in (0, 3, 12):
        (out / ("ann_ndcg_%d" % n)).write_text(json.dumps({"ndcg": 0.1 * n, "checkpoint": "/m/checkpoint-%d/" % (n * 100)}))
        (out / ("ann_training_data_%d" % n)).write_text("1\t2\t3,4\n")
    no, path, js = adg.get_latest_ann_data(str(out))
    assert no == 12 and path.endswith("ann_training_data_12") and js["checkpoint"].endswith("1200/")
    assert (no, path, js) == ann_ref.get_latest_ann_data(str(out))
    assert adg.get_checkpoint_no(js["checkpoint"]) == 1200 == ann_ref.get_checkpoint_no(js["checkpoint"])

    tr = tmp_path / "train"
    args = types.SimpleNamespace(training_dir=str(tr), init_model_dir="/init")
    assert adg.get_latest_checkpoint(args) == ("/init", 0)
    tr.mkdir()
    (tr / "checkpoint-100").mkdir()
    (tr / "checkpoint-100" / "scheduler.pt").write_text("x")
    (tr / "checkpoint-300").mkdir()  # no commit marker yet: must be ignored
    path, step = adg.get_latest_checkpoint(args)
    assert step == 100 and path == os.path.join(str(tr), "checkpoint-100") + "/"


def test_writers_respect_contract(tmp_path):
    q2id = np.array([5, 6, 7, 8])
    pos = {5: 50, 6: 60, 8: 80}
    neg = {5: [1, 2], 6: [3], 7: [9], 8: [4, 5, 6]}
    random.seed(3)
    train_path, ndcg_path = negatives.write_ann_files(str(tmp_path), 2, 4, q2id, {5, 6, 7, 8}, pos, neg, 0.25,
                                                      "/m/checkpoint-400/")
    lines = open(train_path).read().splitlines()
    assert sorted(lines) == sorted(["5\t50\t1,2", "6\t60\t3", "8\t80\t4,5,6"])  # 7 has no positive
    assert json.load(open(ndcg_path)) == {"ndcg": 0.25, "checkpoint": "/m/checkpoint-400/"}
    # identical to the oracle's restatement under the same seed
    d2 = tmp_path / "o"
    d2.mkdir()
    random.seed(3)
    ann_ref.write_ann_files(str(d2), 2, np.zeros((4, 1)), q2id, {5, 6, 7, 8}, pos, neg, 0.25, "/m/checkpoint-400/")
    assert open(train_path).read() == (d2 / "ann_training_data_2").read_text()
    assert adg.get_latest_ann_data(str(tmp_path))[0] == 2


def test_cli_flags_match_reference_names():
    a = adg.get_arguments(["--data_dir", "d", "--training_dir", "t", "--init_model_dir", "i", "--model_type", "rdot_nll",
                           "--output_dir", "o", "--cache_dir", "c", "--topk_training", "200", "--negative_sample", "20",
                           "--end_output_num", "0", "--ann_measure_topk_mrr", "--inference", "--local_rank", "3"])
    assert (a.topk_training, a.negative_sample, a.end_output_num, a.ann_chunk_factor) == (200, 20, 0, 5)
    assert a.max_seq_length == 128 and a.max_query_length == 64 and a.per_gpu_eval_batch_size == 128
    assert a.ann_measure_topk_mrr and a.inference and a.local_rank == 3 and not a.only_keep_latest_embedding_file


# ---- native host stage (csrc/host_postsearch.hip) ------------------------------------------------
def test_encoder_precision_flag():
    """The one added flag: --encoder_precision {fp16, split, fp32}.  Not given, the library / environment default applies
    (an explicit "split" default silently overrode ANCE_ENCODER_FP16=1 / ANCE_ENCODER_PRECISE=1); with nothing in
    the environment that is the fp32-grade split mode (the reference runs its encoder in fp32,
    drivers/run_ann_data_gen.py:158,176-180)."""
    from ance_amd import ann_data_gen as adg
    from ance_amd import ann_data_gen_dpr as dpr
    base = ["--data_dir", "d", "--training_dir", "t", "--init_model_dir", "i", "--model_type", "rdot_nll", "--output_dir", "o",
            "--cache_dir", "c"]
    assert adg.get_arguments(base).encoder_precision is None
    assert adg.get_arguments(base).max_tokens == adg.DRIVER_MAX_TOKENS == 131072
    assert adg.get_arguments(base + ["--encoder_precision", "split"]).encoder_precision == "split"
    with pytest.raises(SystemExit):
        adg.get_arguments(base + ["--encoder_precision", "bf16"])
    dbase = ["--data_dir", "d", "--training_dir", "t", "--init_model_dir", "i", "--output_dir", "o", "--cache_dir", "c",
             "--passage_path", "p", "--test_qa_path", "q", "--trivia_test_qa_path", "r"]
    try:
        assert dpr.get_arguments(dbase).encoder_precision is None
    except SystemExit:  # (the DPR parser has more required flags than this test names: the default is what matters)
        import inspect
        assert 'encoder_precision", default=None' in inspect.getsource(dpr.get_arguments)


def test_precision_is_an_argument_not_an_environment_mutation(monkeypatch):
    """ABI v5: the arithmetic travels in AnceEncoderDesc.precision; the binding no longer touches os.environ, and the size
    queries follow the descriptor (the modes differ in arena and workspace size)."""
    import ctypes
    import inspect
    from ance_amd import _lib, encoder
    assert "os.environ.update" not in inspect.getsource(encoder) and "os.environ.pop" not in inspect.getsource(encoder)
    assert _lib.PRECISION_CODES == {None: 0, "split": 1, "fp16": 2, "fp32": 3}
    for k in ("ANCE_ENCODER_PRECISE", "ANCE_ENCODER_SPLIT", "ANCE_ENCODER_FP16"):
        monkeypatch.delenv(k, raising=False)
    L = _lib.lib()

    def sizes(code):
        d = _lib.AnceEncoderDesc(arch=0, n_layers=12, hidden=768, n_heads=12, intermediate=3072, vocab_size=50265,
                                 max_position=514, pad_token_id=1, ln_eps=1e-5, has_head=1, max_seq_len=512, max_tokens=32768,
                                 precision=code)
        return L.ance_encoder_weight_bytes(ctypes.byref(d)), L.ance_encoder_workspace_bytes(ctypes.byref(d))

    default, split, fp16, fp32 = sizes(0), sizes(1), sizes(2), sizes(3)
    assert default == split                       # nothing in the environment: the default is the split mode
    assert fp16[0] < min(split[0], fp32[0]) and fp16[1] < split[1] < fp32[1] and split != fp32
    monkeypatch.setenv("ANCE_ENCODER_FP16", "1")  # the environment moves only the DEFAULT code
    assert sizes(0) == fp16 and sizes(1) == split and sizes(3) == fp32
    assert encoder.precision_from_env() == "fp16"
    monkeypatch.setenv("ANCE_ENCODER_PRECISE", "1")
    assert sizes(0) == fp32 and encoder.precision_from_env() == "fp32"
    assert sizes(4) == (0, 0) and sizes(-1) == (0, 0)  # not a precision code


def test_row_bases_of_a_refresh_are_a_pure_function_of_the_rank():
    from ance_amd import ann_data_gen as adg
    from ance_amd.cache import shard_range
    for n, w, c in ((10, 3, 1), (8841823, 8, 1), (56, 2, 4), (5, 8, 1)):
        bases = adg.shard_row_bases(n, w, c)
        assert bases == [shard_range(n, r, w)[0] * c for r in range(w)] and bases[0] == 0 and bases == sorted(bases)


def test_native_shuffle_continues_cpython_stream():
    for n in (0, 1, 2, 3, 7, 200, 1000, 4097, 100003):
        random.seed(1000 + n)
        random.random()  # move the index off the seed position
        st = random.getstate()
        ref = list(range(n))
        random.shuffle(ref)
        after_ref = random.getstate()
        random.setstate(st)
        got = negatives.py_shuffled_range(n)
        assert got.tolist() == ref
        assert random.getstate() == after_ref  # the very next draw of a caller is unchanged too


def test_native_shuffle_accepts_state_index_zero():
    """CPython's setstate takes any index in [0, 624]; index 0 is what its own regenerate leaves."""
    random.seed(77)
    ver, internal, gauss = random.getstate()
    st0 = (ver, tuple(internal[:624]) + (0,), gauss)
    random.setstate(st0)
    ref = list(range(500))
    random.shuffle(ref)
    after_ref = random.getstate()
    random.setstate(st0)
    got = negatives.py_shuffled_range(500)
    assert got.tolist() == ref and random.getstate() == after_ref


def _random_case(rng, nq, k, n_rows, chunks, dup_q=False, with_pad=False, inactive=False):
    p2id = (np.arange(n_rows) // chunks).astype(np.int64)
    q2id = np.arange(1000, 1000 + nq, dtype=np.int64)
    if dup_q:
        q2id[nq // 2] = q2id[3]
        q2id[nq - 1] = q2id[3]
    I = rng.integers(0, n_rows, size=(nq, k)).astype(np.int64)
    if with_pad:
        I[::7, -3:] = -1
    pos = {int(q): int(p2id[I[r, int(rng.integers(0, min(k, 12)))]]) for r, q in enumerate(q2id.tolist())}
    eff = set(q2id.tolist())
    if inactive:
        eff = set(q2id[::3].tolist())
    return q2id, p2id, pos, I, eff


@pytest.mark.parametrize("nq,k,chunks,flags", [(50, 20, 1, {}), (300, 64, 4, dict(dup_q=True)),
                                               (5000, 40, 1, dict(with_pad=True, inactive=True)),
                                               (4500, 200, 4, dict(dup_q=True, inactive=True))])
def test_native_selection_and_writer_match_oracle(tmp_path, nq, k, chunks, flags):
    rng = np.random.default_rng(nq + k)
    q2id, p2id, pos, I, eff = _random_case(rng, nq, k, 3000, chunks, **flags)
    for topk in (False, True):
        for ns in (0, 1, 7):
            random.seed(77)
            sel = negatives.select_negatives(q2id, p2id, pos, I, eff, ns, topk)
            a = sel.as_dict()
            d1 = tmp_path / ("a%d%d" % (topk, ns))
            d1.mkdir()
            negatives.write_ann_files(str(d1), 1, nq, q2id, eff, pos, sel, 0.5, "/m/checkpoint-7/")
            s1 = random.getstate()
            random.seed(77)
            b, mrr = ann_ref.generate_negative_passage_ids(q2id, p2id, pos, I, eff, ns, topk)
            d2 = tmp_path / ("b%d%d" % (topk, ns))
            d2.mkdir()
            ann_ref.write_ann_files(str(d2), 1, I, q2id, eff, pos, b, 0.5, "/m/checkpoint-7/")
            assert a == b
            assert mrr is None or sel.mrr == mrr
            assert (d1 / "ann_training_data_1").read_text() == (d2 / "ann_training_data_1").read_text()
            assert random.getstate() == s1
            # dict input goes through the same writer
            random.seed(5)
            d3 = tmp_path / ("c%d%d" % (topk, ns))
            d3.mkdir()
            negatives.write_ann_files(str(d3), 1, nq, q2id, eff, pos, a, 0.5, "/m/checkpoint-7/")
            random.seed(5)
            d4 = tmp_path / ("d%d%d" % (topk, ns))
            d4.mkdir()
            ann_ref.write_ann_files(str(d4), 1, I, q2id, eff, pos, b, 0.5, "/m/checkpoint-7/")
            assert (d3 / "ann_training_data_1").read_text() == (d4 / "ann_training_data_1").read_text()


def test_native_selection_errors():
    rng = np.random.default_rng(3)
    q2id, p2id, pos, I, eff = _random_case(rng, 20, 10, 100, 1)
    del pos[int(q2id[4])]
    with pytest.raises(KeyError):
        negatives.select_negatives(q2id, p2id, pos, I, eff, 3, False)
    q2id, p2id, pos, I, eff = _random_case(rng, 20, 10, 100, 1)
    I[2, 2] = 100  # out of range row id
    with pytest.raises(Exception):
        negatives.select_negatives(q2id, p2id, pos, I, eff, 3, False)
