"""DPR (NQ / TriviaQA) host-side post-search logic of drivers/run_ann_data_gen_dpr.py: answer-string
matching, top-k hit accuracy and answer-filtered negatives.

``has_answer`` follows utils/dpr_utils.py:241-306: NFD-normalise, tokenise with the regex
``[\\p{L}\\p{N}\\p{M}]+|[^\\p{Z}\\p{C}]`` (case-insensitive, unicode), lower-case, and look for any
answer as a contiguous token sub-sequence of the passage.  Passages are tokenised once and cached
(the reference re-tokenises the same passage for every query that retrieves it).
"""
import unicodedata

import numpy as np
import regex

_TOKEN_RE = regex.compile(r"([\p{L}\p{N}\p{M}]+)|([^\p{Z}\p{C}])", flags=regex.IGNORECASE + regex.UNICODE + regex.MULTILINE)


def tokenize_uncased(text):
    """SimpleTokenizer(...).tokenize(text).words(uncased=True) (utils/dpr_utils.py:267-306)."""
    return [m.group().lower() for m in _TOKEN_RE.finditer(unicodedata.normalize("NFD", text))]


class AnswerMatcher:
    def __init__(self, passages):
        """``passages``: {pid_offset: (text, title)} as built by load_data (run_ann_data_gen_dpr.py:63-109)."""
        self.passages = passages
        self._tok = {}
        self._ans = {}

    def passage_tokens(self, doc_id):
        t = self._tok.get(doc_id)
        if t is None:
            text = self.passages[doc_id][0]
            t = tokenize_uncased(text) if text is not None else None
            self._tok[doc_id] = t
        return t

    def answer_tokens(self, answer):
        t = self._ans.get(answer)
        if t is None:
            t = tokenize_uncased(answer)
            self._ans[answer] = t
        return t

    def has_answer(self, answers, doc_id):
        text = self.passage_tokens(doc_id)
        if text is None:
            return False
        n = len(text)
        for a in answers:
            at = self.answer_tokens(a)
            m = len(at)
            if m == 0:
                if n + 1 > 0:  # the reference's range(0, len(text) - 0 + 1) is non-empty: [] == text[i:i] matches
                    return True
                continue
            first = at[0]
            for i in range(0, n - m + 1):
                if text[i] == first and text[i:i + m] == at:
                    return True
        return False


def validate(matcher, answers, closest_docs, query_embedding2id, passage_embedding2id):
    """Top-k hit accuracy list (run_ann_data_gen_dpr.py:312-340): entry i = fraction of questions whose
    first answer-bearing passage is at rank <= i."""
    p2id = np.asarray(passage_embedding2id)
    n_docs = closest_docs.shape[1]
    hits_at = np.zeros(n_docs, dtype=np.int64)
    for row in range(closest_docs.shape[0]):
        qid = int(query_embedding2id[row])
        ans = answers[qid]
        for rank, pidx in enumerate(closest_docs[row]):
            if matcher.has_answer(ans, int(p2id[pidx])):
                hits_at[rank:] += 1
                break
    return (hits_at / closest_docs.shape[0]).tolist()


def generate_negative_passage_ids(matcher, answers, query_embedding2id, passage_embedding2id, closest_docs,
                                  training_query_positive_id, negative_sample):
    """Negatives = examined top candidates that lack every answer string (run_ann_data_gen_dpr.py:281-309).
    As in the reference, ``neg_cnt`` counts EXAMINED candidates (kept or not), so at most
    ``negative_sample`` of the first distinct non-positive candidates are looked at."""
    p2id = np.asarray(passage_embedding2id)
    out = {}
    for row in range(closest_docs.shape[0]):
        qid = int(query_embedding2id[row])
        pos_pid = training_query_positive_id[qid]
        negs = []
        examined = 0
        for pidx in closest_docs[row]:
            doc_id = int(p2id[pidx])
            if doc_id == pos_pid or doc_id in negs:
                continue
            if examined >= negative_sample:
                break
            if not matcher.has_answer(answers[qid], doc_id):
                negs.append(doc_id)
            examined += 1
        out[qid] = negs
    return out
