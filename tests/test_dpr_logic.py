"""DPR host logic (ance_amd.dpr: answer matching, top-k hit accuracy, answer-filtered negatives) against
golden vectors produced by the reference's own run_ann_data_gen_dpr.validate /
GenerateNegativePassaageID / utils.dpr_utils.has_answer (tests/golden/make_golden.py) -- CPU only."""
import json
import os
import types

import numpy as np
import pytest

from ance_amd import ann_data_gen_dpr as dprjob
from ance_amd import dpr


@pytest.fixture(scope="module")
def g(golden_dir):
    with open(os.path.join(golden_dir, "dpr_postsearch.json")) as f:
        j = json.load(f)
    j["passages"] = {int(k): tuple(v) for k, v in j["passages"].items()}
    j["I"] = np.asarray(j["I"], dtype=np.int64)
    return j


def test_has_answer_matches_reference(g):
    m = dpr.AnswerMatcher(g["passages"])
    for ai, ans in enumerate(g["answers_pool"]):
        got = [m.has_answer(ans, pid) for pid in range(40)]
        assert got == g["single"][ai], (ans, got, g["single"][ai])


def test_validate_matches_reference(g):
    m = dpr.AnswerMatcher(g["passages"])
    nq = g["I"].shape[0]
    hits = dpr.validate(m, g["answers"], g["I"], np.arange(nq), np.arange(len(g["passages"])))
    assert hits == g["hits"]
    assert all(b >= a for a, b in zip(hits, hits[1:]))  # cumulative


def test_negatives_match_reference(g):
    m = dpr.AnswerMatcher(g["passages"])
    nq = g["I"].shape[0]
    neg = dpr.generate_negative_passage_ids(m, g["answers"], np.arange(nq), np.arange(len(g["passages"])), g["I"],
                                            g["pos"], g["negative_sample"])
    assert {str(k): v for k, v in neg.items()} == g["neg"]
    # "examined" semantics: never more than negative_sample, positives never kept
    for q, v in neg.items():
        assert len(v) <= g["negative_sample"] and g["pos"][q] not in v


def test_tokenizer_unicode():
    assert dpr.tokenize_uncased("Zürich's CAFÉ, naïve!") == dpr.tokenize_uncased("zürich's café, naïve!")
    assert dpr.tokenize_uncased("U.S. rock&roll") == ["u", ".", "s", ".", "rock", "&", "roll"]


def test_checkpoint_discovery_and_flags(tmp_path):
    tr = tmp_path / "train"
    a = types.SimpleNamespace(training_dir=str(tr), init_model_dir="/init/dpr.cp")
    assert dprjob.get_latest_checkpoint(a) == ("/init/dpr.cp", 0)
    tr.mkdir()
    (tr / "checkpoint-500").write_text("x")
    (tr / "checkpoint-1500").write_text("x")
    (tr / "other").write_text("x")
    assert dprjob.get_latest_checkpoint(a) == (os.path.join(str(tr), "checkpoint-1500"), 1500)
    args = dprjob.get_arguments(["--data_dir", "d", "--training_dir", "t", "--init_model_dir", "i", "--model_type", "dpr",
                                 "--output_dir", "o", "--cache_dir", "c", "--passage_path", "p", "--test_qa_path", "q",
                                 "--trivia_test_qa_path", "r", "--topk_training", "200", "--negative_sample", "100"])
    assert args.topk_training == 200 and args.negative_sample == 100 and args.max_seq_length == 128


def test_load_data_formats(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    (d / "pid2offset").write_text("10\t0\n11\t1\n12\t2\n")
    (d / "train-ann").write_text("0\t1\t['Paris', 'paris france']\n1\t2\t[\"O'Neil\"]\n")
    (tmp_path / "nq-test.csv").write_text("who?\t['a', 'b']\n")
    (tmp_path / "trivia-test.csv").write_text("what?\t['c']\n")
    (tmp_path / "psgs_w100.tsv").write_text("id\ttext\ttitle\n10\t\"He said \"\"hi\"\" in Paris\"\tT0\n11\tsecond\tT1\n12\tthird\tT2\n")
    a = types.SimpleNamespace(data_dir=str(d), test_qa_path=str(tmp_path), trivia_test_qa_path=str(tmp_path),
                              passage_path=str(tmp_path))
    text, pos, ans, ta, tv = dprjob.load_data(a)
    assert pos == [1, 2] and ans == [["Paris", "paris france"], ["O'Neil"]] and ta == [["a", "b"]] and tv == [["c"]]
    assert text[0] == ('He said "hi" in Paris', "T0") and text[2] == ("third", "T2")


def test_worker_pool_equals_serial(g):
    """fork()ed has_answer workers (rows dealt in order) give exactly the serial results -- golden fixture
    tiled to a few thousand rows, with duplicate candidate ids in a row."""
    m = dpr.AnswerMatcher(g["passages"])
    nq, n_p = g["I"].shape[0], len(g["passages"])
    rng = np.random.default_rng(4)
    reps = 60
    I = np.concatenate([g["I"][rng.permutation(nq)] for _ in range(reps)], axis=0)
    I[::11, 5] = I[::11, 2]  # a repeated candidate inside a row
    answers = {q: g["answers"][q % nq] for q in range(I.shape[0])}
    pos = {q: int(I[q, int(rng.integers(0, 6))]) for q in range(I.shape[0])}
    q2id = np.arange(I.shape[0])
    p2id = np.arange(n_p)
    want_hits = dpr.validate(m, answers, I, q2id, p2id)
    want_neg = dpr.generate_negative_passage_ids(m, answers, q2id, p2id, I, pos, g["negative_sample"])
    pool = dpr.AnswerPool(g["passages"], n_workers=3)
    try:
        assert dpr.validate(m, answers, I, q2id, p2id, pool=pool) == want_hits
        assert dpr.generate_negative_passage_ids(m, answers, q2id, p2id, I, pos, g["negative_sample"], pool=pool) == want_neg
        # tiny inputs stay in-process
        assert pool.map_rows("hits", [(answers[0], p2id[I[0]].tolist())]) == [dpr._hits_row(m, answers[0], p2id[I[0]].tolist())]
    finally:
        pool.close()


def test_token_string_search_is_the_token_list_walk():
    """has_answer is computed as a substring search over space-joined token strings, with a regex-free tokenisation of
    ASCII text (ance_amd/dpr.py).  Against the definition -- the reference's token-list walk of utils/dpr_utils.py:241-262
    on the regex tokens -- for random texts over ASCII letters, digits, every ASCII punctuation and control character,
    and non-ASCII letters / marks / separators (which take the regex path)."""
    import random
    from ance_amd import dpr

    def walk(answers, text):
        t = dpr.tokenize_uncased(text)
        for a in answers:
            at = dpr.tokenize_uncased(a)
            for i in range(0, len(t) - len(at) + 1):
                if at == t[i:i + len(at)]:
                    return True
        return False

    rnd = random.Random(7)
    ascii_pool = ["a", "B", "c9", "Zq", "7", "x_y", "don't", "U.S.", "3.14", "a-b", "(", ")", ",", "&", "_", "~", "\t", "\n", "\x0b", "\x1c",
                  "\x7f", " ", "  ", "\x00", "@", "[", "`", "{", "/"]
    uni_pool = ascii_pool + ["café", "CAFÉ", "Zürich", "İstanbul", "ﬁ", " ", " ", "naïve", "Ａ", "élan", "​", "日本", "№5"]
    for pool in (ascii_pool, uni_pool):
        for _ in range(1500):
            text = "".join(rnd.choice(pool) + rnd.choice(["", " ", " "]) for _ in range(rnd.randint(0, 25)))
            answers = ["".join(rnd.choice(pool) + rnd.choice(["", " "]) for _ in range(rnd.randint(0, 3))) for _ in range(rnd.randint(1, 3))]
            m = dpr.AnswerMatcher({0: (text, "t")})
            assert m.has_answer(answers, 0) == walk(answers, text), (text, answers)
            assert dpr.token_string(text) == (" " + " ".join(dpr.tokenize_uncased(text)) + " " if dpr.tokenize_uncased(text) else " ")
