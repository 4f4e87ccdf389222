"""ance_amd.dpr_data (DPR tokenised-cache producer) against golden hashes of the
reference's own data/DPR_data.py outputs (tests/golden/dpr_preprocess.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

from ance_amd import dpr_data as dd
from ance_amd.cache import TokenCache
from oracle import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dpr_preprocess.json")


def _run(tmp_path, data_type, n_workers=2):
    g = json.load(open(GOLD))
    wiki, qd, ad = synth.make_raw_dpr(str(tmp_path / "raw"), **g["raw"])
    out = str(tmp_path / "out") + "/"
    args = dd.get_arguments(["--out_data_dir", out, "--max_seq_length", str(g["max_seq_length"]), "--data_type", str(data_type),
                             "--question_dir", qd, "--wiki_dir", wiki, "--answer_dir", ad, "--n_workers", str(n_workers)])
    args.tokenizer_factory = synth.toy_bert_tokenizer_factory
    os.makedirs(out, exist_ok=True)
    dd.preprocess(args)
    return g, out


@pytest.mark.parametrize("data_type", [0, 1])
def test_outputs_match_reference_bytes(tmp_path, data_type, capsys):
    g, out = _run(tmp_path, data_type)
    got = {f: hashlib.sha256(open(os.path.join(out, f), "rb").read()).hexdigest() for f in os.listdir(out) if "_split" not in f}
    assert got == g["sha256"][str(data_type)]


def test_merged_training_set_keeps_nq_first(tmp_path, capsys):
    """data_type 2 (the multi-set recipe): NQ rows, then TriviaQA rows; the reference's own run of this
    branch dereferences row 58,812 and needs the full data, so it is checked structurally."""
    g, out = _run(tmp_path, 2, n_workers=1)
    nq, tr, both = (TokenCache(out + n) for n in ("train-query-nq", "train-query-trivia", "train-query"))
    assert len(both) == len(nq) + len(tr) and len(nq) > 0 and len(tr) > 0
    assert np.array_equal(np.asarray(both.records(0, len(nq))), np.asarray(nq.records()))
    assert np.array_equal(np.asarray(both.records(len(nq), len(both))), np.asarray(tr.records()))
    ann = open(out + "train-ann").read().splitlines()
    assert ann == open(out + "train-ann-nq").read().splitlines() + open(out + "train-ann-trivia").read().splitlines()
    # passages: [CLS] title [SEP] text [SEP] pad..., mapping file round-trips
    pc = TokenCache(out + "passages")
    rec = np.asarray(pc.records(0, 1))[0]
    ids = rec[4:].view("<i4")
    assert ids[0] == 101 and 102 in ids.tolist()
    p2o, o2p = dd.load_mapping(out, "pid2offset")
    assert len(p2o) == len(pc) == g["raw"]["n_passages"] and all(o2p[v] == k for k, v in p2o.items())
