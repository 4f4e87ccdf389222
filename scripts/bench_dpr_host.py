#!/usr/bin/env python
"""Host stage of the DPR refresh at scale: `has_answer` over nq x 100 retrieved passages -- top-k hit
accuracy of the dev / trivia questions and the answer-filtered negatives of the training questions
(drivers/run_ann_data_gen_dpr.py:281-340, utils/dpr_utils.py:241-306) -- on synthetic Wikipedia-like passages.
Prints one JSON line: seconds and has_answer calls/s per phase, for the Python implementation (fork()ed workers) and,
when present, the native one."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synth(n_passages, nq, k, seed=0, words_per_passage=100):
    rng = np.random.default_rng(seed)
    base = ["w%d" % i for i in range(30000)] + ["café", "Zürich", "naïve", "İstanbul", "ﬁnal", "O'Neil", "U.S.", "3.14", "state-of-the-art",
                                               "São", "Paulo", "Ångström", "élan", "rock", "&", "roll", ",", ".", "(", ")"]
    vocab = np.array(base, dtype=object)
    zipf = rng.zipf(1.3, size=n_passages * words_per_passage) % len(vocab)
    passages = {}
    for pid in range(n_passages):
        w = vocab[zipf[pid * words_per_passage:(pid + 1) * words_per_passage]]
        passages[pid] = (" ".join(w.tolist()), "title")
    answers = []
    for q in range(nq):
        na = int(rng.integers(1, 4))
        answers.append([" ".join(vocab[rng.integers(0, 3000, size=int(rng.integers(1, 4)))].tolist()) for _ in range(na)])
    I = rng.integers(0, n_passages, size=(nq, k), dtype=np.int64)
    return passages, answers, I


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passages", type=int, default=1000000)
    ap.add_argument("--queries", type=int, default=58812)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--negative-sample", type=int, default=20)
    ap.add_argument("--workers", type=int, default=0)
    a = ap.parse_args()
    from ance_amd import dpr
    t0 = time.perf_counter()
    passages, answers, I = synth(a.passages, a.queries, a.k)
    t_synth = time.perf_counter() - t0
    p2id = np.arange(a.passages, dtype=np.int64)
    q2id = np.arange(a.queries, dtype=np.int64)
    pos = {q: int(I[q, 0]) for q in range(a.queries)}
    out = {"passages": a.passages, "queries": a.queries, "k": a.k, "cores": os.cpu_count(), "synth_s": round(t_synth, 1)}
    for impl in ("python", "native"):
        if impl == "native" and not hasattr(dpr, "NativeAnswerMatcher"):
            continue
        t0 = time.perf_counter()
        if impl == "python":
            pool = dpr.AnswerPool(passages, a.workers or None)
            matcher = dpr.AnswerMatcher(passages)
        else:
            pool = None
            matcher = dpr.NativeAnswerMatcher(passages, n_threads=a.workers or 0)
        t_init = time.perf_counter() - t0
        t0 = time.perf_counter()
        hits = dpr.validate(matcher, answers, I, q2id, p2id, pool=pool)
        t_val = time.perf_counter() - t0
        t0 = time.perf_counter()
        neg = dpr.generate_negative_passage_ids(matcher, answers, q2id, p2id, I, pos, a.negative_sample, pool=pool)
        t_neg = time.perf_counter() - t0
        if pool is not None:
            pool.close()
        out[impl] = {"init_s": round(t_init, 2), "validate_s": round(t_val, 2), "negatives_s": round(t_neg, 2),
                     "top20": hits[19], "top100": hits[-1], "n_neg": int(sum(len(v) for v in neg.values())),
                     "workers": (pool.n_workers if pool is not None else getattr(matcher, "n_threads", 1))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
