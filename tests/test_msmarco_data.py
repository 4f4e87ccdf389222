"""ance_amd.msmarco_data (tokenised-cache producer) against golden hashes of the
reference's own data/msmarco_data.py outputs (tests/golden/preprocess.json, generator:
tests/golden/make_golden.py::golden_preprocess)."""
import hashlib
import json
import os
import pickle

import numpy as np
import pytest

from ance_amd import msmarco_data as md
from ance_amd.cache import TokenCache
from oracle import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess.json")


def _run(tmp_path, data_type, n_workers, extra=()):
    g = json.load(open(GOLD))
    raw, out = str(tmp_path / "raw"), str(tmp_path / ("out%d" % n_workers))
    if not os.path.exists(raw):
        synth.make_raw_msmarco(raw, data_type, **g["raw"])
    args = md.get_arguments(["--data_dir", raw, "--out_data_dir", out, "--model_type", "rdot_nll",
                             "--model_name_or_path", "unused", "--max_seq_length", str(g["max_seq_length"]),
                             "--max_query_length", str(g["max_query_length"]), "--data_type", str(data_type),
                             "--n_workers", str(n_workers)] + list(extra))
    args.tokenizer_factory = synth.toy_tokenizer_factory
    os.makedirs(out, exist_ok=True)
    md.preprocess(args)
    return g, out


@pytest.mark.parametrize("data_type", [1, 0])
@pytest.mark.parametrize("n_workers", [1, 3])
def test_outputs_match_reference_bytes(tmp_path, data_type, n_workers, capsys):
    g, out = _run(tmp_path, data_type, n_workers)
    want = g["sha256"][str(data_type)]
    got = {f: hashlib.sha256(open(os.path.join(out, f), "rb").read()).hexdigest()
           for f in os.listdir(out) if "_split" not in f}
    assert got == want


def test_cache_is_what_the_hot_path_reads(tmp_path):
    g, out = _run(tmp_path, 1, 2, extra=["--remove_splits"])
    assert not [f for f in os.listdir(out) if "_split" in f]
    cache = TokenCache(os.path.join(out, "passages"))
    assert len(cache) == g["raw"]["n_passages"] and cache.embedding_size == g["max_seq_length"]
    lens = cache.lengths()
    assert lens.min() >= 3 and lens.max() == g["max_seq_length"]
    rec = np.asarray(cache.records(0, 1))[0]
    ids = rec[4:].view("<i4")
    assert ids[0] == 0 and ids[lens[0] - 1] == 2 and (ids[lens[0]:] == 1).all()  # <s> ... </s> <pad>*
    pid2offset = pickle.load(open(os.path.join(out, "pid2offset.pickle"), "rb"))
    assert sorted(pid2offset.values()) == list(range(len(cache)))
    # qrels are in offset space and every labelled query survived
    qid2offset = pickle.load(open(os.path.join(out, "qid2offset.pickle"), "rb"))
    rows = [l.split("\t") for l in open(os.path.join(out, "dev-qrel.tsv")).read().splitlines()]
    assert {int(r[0]) for r in rows} == set(qid2offset.values())
    assert all(0 <= int(r[1]) < len(cache) for r in rows)
    # rerun is a no-op like the reference ("preprocessed data already exist")
    before = os.path.getmtime(os.path.join(out, "passages"))
    _run(tmp_path, 1, 2)
    assert os.path.getmtime(os.path.join(out, "passages")) == before


def test_flags_match_reference_names():
    a = md.get_arguments(["--data_dir", "d", "--out_data_dir", "o", "--model_type", "rdot_nll", "--model_name_or_path", "m"])
    assert (a.max_seq_length, a.max_query_length, a.max_doc_character, a.data_type) == (128, 64, 10000, 0)
