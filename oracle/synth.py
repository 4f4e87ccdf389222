"""Synthetic MS MARCO-shaped data in the reference's on-disk formats (oracle / test infra).

Formats follow the reference writers:

* tokenised cache record = 4-byte big-endian ``passage_len`` + ``L`` x int32 (native LE)
  (``data/msmarco_data.py:258,272`` minus the 8-byte id that the merge step strips,
  ``data/msmarco_data.py:160-165``), ``<name>_meta`` JSON
  ``{'type': 'int32', 'total_number': N, 'embedding_size': L}`` (``:171-176``).
* ``train-qrel.tsv`` / ``dev-qrel.tsv``: ``qid_offset \t pid_offset \t rel`` (``:116-121``).

Distributions are the ones SURVEY.md section 8(d) prescribes (config 1 / config 2).
"""
import json
import os

import numpy as np

PAD, BOS, EOS = 1, 0, 2
VOCAB = 50265


def lognormal_lengths(rng, n, median, sigma, lo, hi):
    x = rng.lognormal(mean=np.log(median), sigma=sigma, size=n)
    return np.clip(np.rint(x), lo, hi).astype(np.int64)


def make_records(rng, n, L, lengths, vocab=VOCAB, bos=BOS, eos=EOS, pad=PAD, lo_tok=3):
    """int32 [n, 1+L] array: column 0 = length (host order, byte-swapped on write)."""
    ids = rng.integers(lo_tok, vocab, size=(n, L), dtype=np.int64).astype(np.int32)
    pos = np.arange(L)[None, :]
    lens = lengths[:, None]
    ids[:, 0] = bos
    ids[np.arange(n), np.maximum(lengths - 1, 0)] = eos
    ids[:, 0] = np.where(lengths > 0, bos, pad)
    ids = np.where(pos < lens, ids, pad).astype(np.int32)
    return ids


def write_cache(path, ids, lengths):
    """Write ``ids`` int32 [n, L] + ``lengths`` in the reference cache format."""
    n, L = ids.shape
    rec = np.empty((n, 4 + 4 * L), dtype=np.uint8)
    rec[:, :4] = lengths.astype(">u4").view(np.uint8).reshape(n, 4)
    rec[:, 4:] = np.ascontiguousarray(ids.astype("<i4")).view(np.uint8).reshape(n, 4 * L)
    with open(path, "wb") as f:
        f.write(rec.tobytes())
    with open(path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": int(n), "embedding_size": int(L)}, f)


def make_msmarco_like(out_dir, n_passages=10000, n_train=1000, n_dev=200, L=128, Lq=64,
                      seed=1234, dup_frac=0.01, len_median=70, len_sigma=0.45,
                      q_median=9, q_sigma=0.35, plant_frac=0.5):
    """Config-1-style toy set.  Returns a dict of the arrays written."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    plen = lognormal_lengths(rng, n_passages, len_median, len_sigma, 8, L)
    pids = make_records(rng, n_passages, L, plen)
    # planted exact duplicates (tie-break coverage)
    n_dup = int(round(dup_frac * n_passages))
    if n_dup > 0:
        src = rng.integers(0, n_passages, size=n_dup)
        dst = rng.integers(0, n_passages, size=n_dup)
        pids[dst] = pids[src]
        plen[dst] = plen[src]

    def queries(n):
        ql = lognormal_lengths(rng, n, q_median, q_sigma, 4, Lq)
        return make_records(rng, n, Lq, ql), ql

    tq, tql = queries(n_train)
    dq, dql = queries(n_dev)
    write_cache(os.path.join(out_dir, "train-query"), tq, tql)
    write_cache(os.path.join(out_dir, "dev-query"), dq, dql)

    train_pos = rng.integers(0, n_passages, size=n_train)
    # planted hits: for a fraction of the queries the positive passage carries the query's own
    # tokens, so it is retrieved at rank 1 by any encoder (non-zero NDCG / positive-filter coverage)
    if plant_frac > 0 and Lq <= L:
        for q in range(int(plant_frac * n_train)):
            p = int(train_pos[q])
            pids[p, :] = PAD
            pids[p, :Lq] = tq[q]
            plen[p] = tql[q]
    with open(os.path.join(out_dir, "train-qrel.tsv"), "w") as f:
        for q, p in enumerate(train_pos):
            f.write("%d\t%d\t1\n" % (q, p))
    dev_rel = []
    with open(os.path.join(out_dir, "dev-qrel.tsv"), "w") as f:
        for q in range(n_dev):
            m = int(rng.integers(1, 4))
            ps = rng.choice(n_passages, size=m, replace=False)
            if plant_frac > 0 and Lq <= L and q < int(plant_frac * n_dev):
                p = int(ps[0])
                pids[p, :] = PAD
                pids[p, :Lq] = dq[q]
                plen[p] = dql[q]
            for p in ps:
                f.write("%d\t%d\t1\n" % (q, p))
                dev_rel.append((q, int(p), 1))
    write_cache(os.path.join(out_dir, "passages"), pids, plen)
    return dict(passages=pids, passage_len=plen, train_query=tq, train_query_len=tql,
                dev_query=dq, dev_query_len=dql, train_pos=train_pos, dev_rel=dev_rel)


def ln_rows(rng, n, d=768, dtype=np.float32):
    """Rows distributed like the reference head output at init: LayerNorm(N(0, I))
    (SURVEY.md section 8(d), config 2 search-only restatement)."""
    z = rng.standard_normal((n, d)).astype(np.float32)
    z -= z.mean(axis=1, keepdims=True)
    z /= np.sqrt((z * z).mean(axis=1, keepdims=True) + 1e-5)
    return z.astype(dtype)


def dyadic_rows(rng, n, d=768, levels=8, scale=1.0 / 16):
    """Rows on a coarse dyadic grid: every product and partial sum is exact in fp32,
    so inner products are order independent (test plan 3a)."""
    return (rng.integers(-levels, levels + 1, size=(n, d)).astype(np.float32) * np.float32(scale))


# ---- raw (untokenised) MS MARCO-like TSVs + a deterministic toy tokenizer -------------------------
class ToyTokenizer:
    """Whitespace tokenizer with RoBERTa's special ids (<s>=0, <pad>=1, </s>=2) and the
    transformers-2.x ``encode(max_length=...)`` truncation the reference relied on (specials kept).
    Lets the reference's data/msmarco_data.py and ance_amd.msmarco_data run on identical token ids
    without any pretrained vocabulary (there is no network)."""
    sep_token = "</s>"
    pad_token_id = 1

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        return cls()

    def encode(self, text, add_special_tokens=True, max_length=None, **kwargs):
        import zlib
        ids = [4 + zlib.crc32(w.lower().encode("utf8")) % 50000 for w in text.split()]
        if add_special_tokens:
            if max_length is not None and len(ids) > max_length - 2:
                ids = ids[:max(max_length - 2, 0)]
            return [0] + ids + [2]
        return ids[:max_length] if max_length is not None else ids


def toy_tokenizer_factory():
    return ToyTokenizer()


_WORDS = ("alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi omicron pi rho sigma tau "
          "upsilon phi chi psi omega rock river stone cloud forest engine matrix vector kernel wave").split()


def make_raw_msmarco(data_dir, data_type, n_passages=70, n_train=25, n_dev=9, seed=5):
    """Raw TSVs in the layout data/msmarco_data.py reads (data_type 1: passage; 0: document).
    Ids are sparse and shuffled, some passages exceed any small max_seq_length, some queries have no
    label (they must be dropped), one query has two labels."""
    os.makedirs(data_dir, exist_ok=True)
    rng = np.random.default_rng(seed)

    def text(lo, hi):
        return " ".join(_WORDS[int(j)] for j in rng.integers(0, len(_WORDS), size=int(rng.integers(lo, hi))))

    pids = rng.permutation(np.arange(1000, 1000 + 3 * n_passages, 3))[:n_passages].tolist()
    doc = data_type == 0
    with open(os.path.join(data_dir, "msmarco-docs.tsv" if doc else "collection.tsv"), "w", encoding="utf-8") as f:
        for p in pids:
            if doc:
                f.write("D%d\thttp://x.org/%d \t%s \t%s\n" % (p, p, text(1, 5), text(3, 60)))
            else:
                f.write("%d\t%s \n" % (p, text(3, 40)))

    def queries(fname, qrel_name, n, base, sep):
        qids = rng.permutation(np.arange(base, base + 2 * n, 2))[:n].tolist()
        with open(os.path.join(data_dir, fname), "w", encoding="utf-8") as f:
            for q in qids:
                f.write("%d\t%s\n" % (q, text(2, 12)))
        labelled = qids[: max(1, (3 * n) // 4)]
        with open(os.path.join(data_dir, qrel_name), "w", encoding="utf-8") as f:
            for j, q in enumerate(labelled):
                tgt = pids[int(rng.integers(0, len(pids)))]
                f.write(sep.join([str(q), "0", ("D%d" % tgt) if doc else str(tgt), str(1 if not doc else int(rng.integers(1, 4)))]) + "\n")
                if j == 1:
                    tgt2 = pids[int(rng.integers(0, len(pids)))]
                    f.write(sep.join([str(q), "0", ("D%d" % tgt2) if doc else str(tgt2), "1"]) + "\n")

    if doc:
        queries("msmarco-doctrain-queries.tsv", "msmarco-doctrain-qrels.tsv", n_train, 500000, " ")
        queries("msmarco-test2019-queries.tsv", "2019qrels-docs.txt", n_dev, 900000, " ")
    else:
        queries("queries.train.tsv", "qrels.train.tsv", n_train, 500000, "\t")
        queries("queries.dev.small.tsv", "qrels.dev.small.tsv", n_dev, 900000, "\t")


class ToyBertTokenizer:
    """BERT-shaped counterpart of ToyTokenizer ([CLS]=101, [SEP]=102, [PAD]=0) with pair encoding and the
    ``longest_first`` truncation transformers 2.x applied whenever ``max_length`` was given."""
    sep_token = "[SEP]"
    sep_token_id = 102
    pad_token_id = 0

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        return cls()

    @staticmethod
    def _words(text):
        import zlib
        return [1000 + zlib.crc32(w.lower().encode("utf8")) % 29000 for w in text.split()]

    def encode(self, text, text_pair=None, add_special_tokens=True, max_length=None, **kwargs):
        a = self._words(text)
        b = self._words(text_pair) if text_pair is not None else None
        if max_length is not None:
            budget = max_length - (3 if b is not None else 2)
            while len(a) + (len(b) if b is not None else 0) > max(budget, 0):
                if b is not None and len(b) > len(a):
                    b.pop()
                elif a:
                    a.pop()
                else:
                    b.pop()
        ids = [101] + a + [102]
        if b is not None:
            ids += b + [102]
        return ids


def toy_bert_tokenizer_factory():
    return ToyBertTokenizer()


def make_raw_dpr(root, n_passages=90, n_nq=14, n_trivia=11, seed=9):
    """psgs_w100.tsv (with its header row), nq/trivia train+dev json (some samples without positives or
    without hard negatives: they must be dropped), nq/trivia test csv.  Returns (wiki_dir, question_dir, answer_dir)."""
    import json as _json
    rng = np.random.default_rng(seed)
    wiki, qdir, adir = (os.path.join(root, d) for d in ("wiki", "questions", "answers"))
    for d in (wiki, qdir, adir):
        os.makedirs(d, exist_ok=True)

    def text(lo, hi):
        return " ".join(_WORDS[int(j)] for j in rng.integers(0, len(_WORDS), size=int(rng.integers(lo, hi))))

    with open(os.path.join(wiki, "psgs_w100.tsv"), "w", encoding="utf-8") as f:
        f.write("id\ttext\ttitle\n")
        for p in range(1, n_passages + 1):
            f.write('%d\t"%s"\t%s\n' % (p, text(5, 40), text(1, 4)))

    def samples(n, key):
        out = []
        for i in range(n):
            pos = [{key: str(int(rng.integers(1, n_passages + 1)))} for _ in range(int(rng.integers(0, 3)))]
            neg = [{key: str(int(rng.integers(1, n_passages + 1)))} for _ in range(int(rng.integers(0, 4)))]
            out.append({"question": text(3, 9) + ("?" if i % 2 else ""), "answers": [text(1, 3), "it's \"x\""][: 1 + i % 2],
                        "positive_ctxs": pos, "hard_negative_ctxs": neg})
        return out

    for name, n, key in (("nq-train.json", n_nq, "passage_id"), ("trivia-train.json", n_trivia, "psg_id"),
                         ("nq-dev.json", 6, "passage_id"), ("trivia-dev.json", 5, "psg_id")):
        with open(os.path.join(qdir, name), "w", encoding="utf-8") as f:
            _json.dump(samples(n, key), f)
    for name, n in (("nq-test.csv", 7), ("trivia-test.csv", 5)):
        with open(os.path.join(adir, name), "w", encoding="utf-8") as f:
            for i in range(n):
                f.write("%s%s\t['%s']\n" % (text(3, 9), "?" if i % 3 == 0 else "", text(1, 3)))
    return wiki, qdir, adir
