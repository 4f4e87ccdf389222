"""DPR (NQ / TriviaQA) host-side post-search logic of drivers/run_ann_data_gen_dpr.py: answer-string
matching, top-k hit accuracy and answer-filtered negatives.

``has_answer`` follows utils/dpr_utils.py:241-306: NFD-normalise, tokenise with the regex
``[\\p{L}\\p{N}\\p{M}]+|[^\\p{Z}\\p{C}]`` (case-insensitive, unicode), lower-case, and look for any
answer as a contiguous token sub-sequence of the passage.

How it is computed here (same answers, measured 10 x faster than the token-list walk at 1 M passages x 58,812
questions x 100 candidates, scripts/bench_dpr_host.py): no token contains a space (\\p{Z} is a separator class), so
"the answer's tokens are a contiguous run of the passage's tokens" is exactly "' ' + ' '.join(answer tokens) + ' '
is a substring of ' ' + ' '.join(passage tokens) + ' '" -- one C-speed substring search.  And for an all-ASCII passage
(most of Wikipedia) the tokenisation itself needs no regex: NFD is the identity, \\p{L}\\p{N}\\p{M} are [A-Za-z0-9],
\\p{Z} is the space, \\p{C} the control characters, so `translate` (pad every punctuation character with spaces,
controls -> space), `lower`, `split`, `join` give the same token string.  Anything non-ASCII takes the regex.
"""
import os
import unicodedata
from multiprocessing import get_context

import numpy as np
import regex

_TOKEN_RE = regex.compile(r"([\p{L}\p{N}\p{M}]+)|([^\p{Z}\p{C}])", flags=regex.IGNORECASE + regex.UNICODE + regex.MULTILINE)


def tokenize_uncased(text):
    """SimpleTokenizer(...).tokenize(text).words(uncased=True) (utils/dpr_utils.py:267-306)."""
    return [m.group().lower() for m in _TOKEN_RE.finditer(unicodedata.normalize("NFD", text))]


# ASCII fast path: every printable non-alphanumeric character is a token of its own, space and the control characters
# separate tokens (in ASCII \p{Z} = {0x20}, \p{C} = 0x00-0x1F and 0x7F)
_ASCII_TABLE = {}
for _c in range(128):
    _ch = chr(_c)
    if _ch.isalnum():
        continue
    _ASCII_TABLE[_c] = " " if (_c <= 0x20 or _c == 0x7F) else " " + _ch + " "


def token_string(text):
    """' ' + ' '.join(tokenize_uncased(text)) + ' '  (a single space for a text without tokens)."""
    toks = text.translate(_ASCII_TABLE).lower().split() if text.isascii() else tokenize_uncased(text)
    return " " + " ".join(toks) + " " if toks else " "


class AnswerMatcher:
    def __init__(self, passages):
        """``passages``: {pid_offset: (text, title)} as built by load_data (run_ann_data_gen_dpr.py:63-109)."""
        self.passages = passages
        self._ans = {}

    def answer_needle(self, answer):
        """' tok tok ' of an answer string, or None for an answer without tokens (which matches every passage: the
        reference's range(0, len(text) - 0 + 1) is never empty and [] == text[i:i])."""
        t = self._ans.get(answer, 0)
        if t == 0:
            t = token_string(answer)
            t = None if t == " " else t
            self._ans[answer] = t
        return t

    def has_answer(self, answers, doc_id):
        text = self.passages[doc_id][0]
        if text is None:
            return False
        hay = None
        for a in answers:
            needle = self.answer_needle(a)
            if needle is None:
                return True
            if hay is None:
                hay = token_string(text)
            if needle in hay:
                return True
        return False


# ---- row-parallel workers ---------------------------------------------------------------------------
# has_answer is the dominant host cost of the DPR job (about six million calls per refresh, each needing the
# retrieved passage tokenised).  Rows are independent, so they are dealt to fork()ed workers that share the
# 21 M-passage dict copy-on-write and keep their own token caches across refreshes.  The pool has to be
# created BEFORE the process touches the GPU (fork after HIP initialisation is not safe): see
# ann_data_gen_dpr.main.
_WORKER_MATCHER = None


def _hits_row(matcher, ans, doc_ids):
    for rank, doc_id in enumerate(doc_ids):
        if matcher.has_answer(ans, doc_id):
            return rank
    return -1


def _negs_row(matcher, ans, doc_ids, pos_pid, negative_sample):
    negs = []
    examined = 0
    for doc_id in doc_ids:
        if doc_id == pos_pid or doc_id in negs:
            continue
        if examined >= negative_sample:
            break
        if not matcher.has_answer(ans, doc_id):
            negs.append(doc_id)
        examined += 1
    return negs


def _work(task):
    kind, rows = task
    m = _WORKER_MATCHER
    if kind == "hits":
        return [_hits_row(m, ans, docs) for ans, docs in rows]
    return [_negs_row(m, ans, docs, pos, ns) for ans, docs, pos, ns in rows]


class AnswerPool:
    """fork()ed has_answer workers over ``passages`` ({pid_offset: (text, title)})."""

    def __init__(self, passages, n_workers=None):
        global _WORKER_MATCHER
        n = n_workers if n_workers else min(32, os.cpu_count() or 1)
        self.n_workers = max(1, int(n))
        _WORKER_MATCHER = AnswerMatcher(passages)
        self._pool = get_context("fork").Pool(self.n_workers) if self.n_workers > 1 else None

    def map_rows(self, kind, rows):
        if self._pool is None or len(rows) < 4 * self.n_workers:
            return _work((kind, rows))
        step = max(1, len(rows) // (self.n_workers * 8))
        chunks = [(kind, rows[i:i + step]) for i in range(0, len(rows), step)]
        out = []
        for part in self._pool.imap(_work, chunks):
            out.extend(part)
        return out

    def close(self):
        if self._pool is not None:
            self._pool.terminate()
            self._pool = None


def validate(matcher, answers, closest_docs, query_embedding2id, passage_embedding2id, pool=None):
    """Top-k hit accuracy list (run_ann_data_gen_dpr.py:312-340): entry i = fraction of questions whose
    first answer-bearing passage is at rank <= i."""
    p2id = np.asarray(passage_embedding2id)
    n_docs = closest_docs.shape[1]
    rows = [(answers[int(query_embedding2id[row])], p2id[closest_docs[row]].tolist())
            for row in range(closest_docs.shape[0])]
    first = pool.map_rows("hits", rows) if pool is not None else [_hits_row(matcher, a, d) for a, d in rows]
    hits_at = np.zeros(n_docs, dtype=np.int64)
    for rank in first:
        if rank >= 0:
            hits_at[rank:] += 1
    return (hits_at / closest_docs.shape[0]).tolist()


def generate_negative_passage_ids(matcher, answers, query_embedding2id, passage_embedding2id, closest_docs,
                                  training_query_positive_id, negative_sample, pool=None):
    """Negatives = examined top candidates that lack every answer string (run_ann_data_gen_dpr.py:281-309).
    As in the reference, ``neg_cnt`` counts EXAMINED candidates (kept or not), so at most
    ``negative_sample`` of the first distinct non-positive candidates are looked at."""
    p2id = np.asarray(passage_embedding2id)
    qids = [int(query_embedding2id[row]) for row in range(closest_docs.shape[0])]
    rows = [(answers[qid], p2id[closest_docs[row]].tolist(), training_query_positive_id[qid], negative_sample)
            for row, qid in enumerate(qids)]
    res = pool.map_rows("negs", rows) if pool is not None else [_negs_row(matcher, *r) for r in rows]
    return dict(zip(qids, res))
