"""Post-search oracle (test infrastructure): plain-Python restatement of the reference's
host-side logic around the index, each function citing what it follows.

Pinned against the reference's own functions by tests/golden/make_golden.py (run through
oracle/ref_harness.py in the build container; vectors committed under tests/golden/).
"""
import json
import math
import os
import random
import re

import numpy as np


# --- utils/util.py:224-243 ----------------------------------------------------------------------
def get_checkpoint_no(checkpoint_path):
    nums = re.findall(r"\d+", checkpoint_path)
    return int(nums[-1]) if len(nums) > 0 else 0


def get_latest_ann_data(ann_data_path):
    prefix = "ann_ndcg_"
    if not os.path.exists(ann_data_path):
        return -1, None, None
    files = list(next(os.walk(ann_data_path))[2])
    nos = [int(s[len(prefix):]) for s in files if s[:len(prefix)] == prefix]
    if nos:
        no = max(nos)
        with open(os.path.join(ann_data_path, prefix + str(no)), "r") as f:
            ndcg_json = json.load(f)
        return no, os.path.join(ann_data_path, "ann_training_data_" + str(no)), ndcg_json
    return -1, None, None


# --- drivers/run_ann_data_gen.py:74-100 -----------------------------------------------------------
def load_positive_ids(data_dir):
    train = {}
    with open(os.path.join(data_dir, "train-qrel.tsv"), "r", encoding="utf8") as f:
        for line in f:
            topicid, docid, rel = line.rstrip("\n").split("\t")
            assert rel == "1"
            train[int(topicid)] = int(docid)
    dev = {}
    with open(os.path.join(data_dir, "dev-qrel.tsv"), "r", encoding="utf8") as f:
        for line in f:
            topicid, docid, rel = line.rstrip("\n").split("\t")
            dev.setdefault(int(topicid), {})[int(docid)] = int(rel)
    return train, dev


# --- utils/util.py:257-329 + data/msmarco_data.py:275-303 ---------------------------------------
def read_cache(path):
    """EmbeddingCache as arrays: (lengths int64[n], ids int32[n, L])."""
    with open(path + "_meta") as f:
        meta = json.load(f)
    n, L = meta["total_number"], int(meta["embedding_size"])
    assert np.dtype(meta["type"]) == np.int32
    raw = np.fromfile(path, dtype=np.uint8).reshape(n, 4 + 4 * L)
    lengths = raw[:, :4].copy().view(">u4").reshape(n).astype(np.int64)
    ids = raw[:, 4:].copy().view("<i4").reshape(n, L)
    return lengths, ids


def rank_order(n, world_size):
    """Row order produced by StreamingDataset's ``i % W == rank`` sharding
    (utils/util.py:318-329) followed by barrier_array_merge's rank-order concatenation
    (utils/util.py:129-144): [i = 0 mod W ascending, i = 1 mod W, ...]."""
    return np.concatenate([np.arange(r, n, world_size, dtype=np.int64) for r in range(world_size)])


def maxp_row_order(n, world_size, batch, chunks):
    """embedding2id for the MaxP body encoder: per rank, per batch of ``batch`` records, one slab
    of ids per chunk (drivers/run_ann_data_gen.py:183-186)."""
    out = []
    for r in range(world_size):
        mine = np.arange(r, n, world_size, dtype=np.int64)
        for b0 in range(0, len(mine), batch):
            idx = mine[b0:b0 + batch]
            for _ in range(chunks):
                out.append(idx)
    return np.concatenate(out) if out else np.zeros((0,), np.int64)


# --- drivers/run_ann_data_gen.py:281-296 ------------------------------------------------------
def query_chunk(num_queries, output_num, chunk_factor):
    effective_idx = output_num % chunk_factor  # cf == 0 raises, as in the reference
    if chunk_factor <= 0:
        chunk_factor = 1
    per = num_queries // chunk_factor
    start = per * effective_idx
    end = num_queries if effective_idx == (chunk_factor - 1) else (start + per)
    return start, end


# --- drivers/run_ann_data_gen.py:339-396 ------------------------------------------------------
def generate_negative_passage_ids(query_embedding2id, passage_embedding2id, training_query_positive_id,
                                  I, effective_q_id, negative_sample, select_topk):
    """Returns ({qid: [neg pid]}, mrr or None).  Uses the module-level ``random`` exactly as the
    reference does, so it is reproducible under ``random.seed``."""
    out = {}
    mrr = 0.0
    num_queries = 0
    for query_idx in range(I.shape[0]):
        query_id = query_embedding2id[query_idx]
        if query_id not in effective_q_id:
            continue
        num_queries += 1
        pos_pid = training_query_positive_id[query_id]
        top_ann_pid = I[query_idx, :].copy()
        if select_topk:
            selected = top_ann_pid[:negative_sample + 1]
        else:
            order = list(range(I.shape[1]))
            random.shuffle(order)
            selected = top_ann_pid[order]
        out[query_id] = []
        neg_cnt = 0
        rank = 0
        for idx in selected:
            neg_pid = passage_embedding2id[idx]
            rank += 1
            if neg_pid == pos_pid:
                if rank <= 10:
                    mrr += 1 / rank
                continue
            if neg_pid in out[query_id]:
                continue
            if neg_cnt >= negative_sample:
                break
            out[query_id].append(neg_pid)
            neg_cnt += 1
    return out, (mrr / num_queries if (select_topk and num_queries) else None)


# --- pytrec_eval stand-in (trec_eval ndcg_cut / map_cut) -----------------------------------------
_CUTS = (5, 10, 15, 20, 30, 100, 200, 500, 1000)


class RelevanceEvaluator:
    """Minimal ``pytrec_eval.RelevanceEvaluator``: trec_eval's ``ndcg_cut`` (gain = rel,
    discount log2(rank+1), ideal from the judged rels sorted descending), ``map_cut``, ``recall`` and
    ``recip_rank`` (relevant = rel > 0).  pytrec_eval itself is not installable here: these four are the
    published trec_eval definitions, unpinned at that boundary.
    Run ordering follows trec_eval: score descending, ties by doc id descending."""

    def __init__(self, qrel, measures):
        self.qrel = qrel
        self.measures = set(measures)

    def evaluate(self, run):
        res = {}
        for qid, docs in run.items():
            if qid not in self.qrel:
                continue
            rels = self.qrel[qid]
            ranked = sorted(docs.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)
            gains = [max(rels.get(d, 0), 0) for d, _ in ranked]
            ideal = sorted((r for r in rels.values() if r > 0), reverse=True)
            num_rel = len(ideal)
            out = {}
            for c in _CUTS:
                if "ndcg_cut" in self.measures:
                    dcg = sum(g / math.log2(i + 2) for i, g in enumerate(gains[:c]))
                    idcg = sum(g / math.log2(i + 2) for i, g in enumerate(ideal[:c]))
                    out["ndcg_cut_%d" % c] = dcg / idcg if idcg > 0 else 0.0
                if "map_cut" in self.measures:
                    hit = 0
                    s = 0.0
                    for i, g in enumerate(gains[:c]):
                        if g > 0:
                            hit += 1
                            s += hit / (i + 1)
                    out["map_cut_%d" % c] = s / num_rel if num_rel > 0 else 0.0
                if "recall" in self.measures:
                    out["recall_%d" % c] = sum(1 for g in gains[:c] if g > 0) / num_rel if num_rel > 0 else 0.0
            if "recip_rank" in self.measures:
                first = next((i for i, g in enumerate(gains) if g > 0), None)
                out["recip_rank"] = 1.0 / (first + 1) if first is not None else 0.0
            res[qid] = out
        return res


# --- drivers/run_ann_data_gen.py:399-440 ------------------------------------------------------
def eval_dev_query(query_embedding2id, passage_embedding2id, dev_query_positive_id, I):
    prediction = {}
    for query_idx in range(I.shape[0]):
        query_id = query_embedding2id[query_idx]
        prediction[query_id] = {}
        rank = 0
        seen = set()
        for idx in I[query_idx, :50]:
            pred_pid = passage_embedding2id[idx]
            if pred_pid not in seen:
                rank += 1
                prediction[query_id][pred_pid] = -rank
                seen.add(pred_pid)

    def to_str(d):  # utils/util.py:194-205
        return {str(k): {str(ik): iv for ik, iv in v.items()} for k, v in d.items()}

    result = RelevanceEvaluator(to_str(dev_query_positive_id), {"map_cut", "ndcg_cut"}).evaluate(to_str(prediction))
    ndcg = 0.0
    cnt = 0
    for k in result:
        cnt += 1
        ndcg += result[k]["ndcg_cut_10"]
    return ndcg / cnt, cnt


# --- drivers/run_ann_data_gen.py:314-334 ------------------------------------------------------
def write_ann_files(output_dir, output_num, I, query_embedding2id, effective_q_id,
                    training_query_positive_id, query_negative_passage, dev_ndcg, checkpoint_path):
    train_path = os.path.join(output_dir, "ann_training_data_" + str(output_num))
    with open(train_path, "w") as f:
        query_range = list(range(I.shape[0]))
        random.shuffle(query_range)
        for query_idx in query_range:
            query_id = query_embedding2id[query_idx]
            if query_id not in effective_q_id or query_id not in training_query_positive_id:
                continue
            pos_pid = training_query_positive_id[query_id]
            f.write("{}\t{}\t{}\n".format(query_id, pos_pid,
                                          ",".join(str(n) for n in query_negative_passage[query_id])))
    with open(os.path.join(output_dir, "ann_ndcg_" + str(output_num)), "w") as f:
        json.dump({"ndcg": dev_ndcg, "checkpoint": checkpoint_path}, f)
    return train_path


def refresh_from_embeddings(output_dir, output_num, checkpoint_path, dev_q, dev_q2id, p_emb, p2id,
                            train_q, train_q2id, train_pos, dev_pos, topk_training, negative_sample,
                            ann_chunk_factor, select_topk, search_fn):
    """drivers/run_ann_data_gen.py:265-336 given embeddings; ``search_fn(x, q, k) -> (D, I)``."""
    _, dev_I = search_fn(p_emb, dev_q, 100)
    dev_ndcg, n_dev = eval_dev_query(dev_q2id, p2id, dev_pos, dev_I)
    s, e = query_chunk(len(train_q), output_num, ann_chunk_factor)
    q = train_q[s:e]
    q2id = train_q2id[s:e]
    _, I = search_fn(p_emb, q, topk_training)
    eff = set(q2id.flatten())
    neg, _ = generate_negative_passage_ids(q2id, p2id, train_pos, I, eff, negative_sample, select_topk)
    write_ann_files(output_dir, output_num, I, q2id, eff, train_pos, neg, dev_ndcg, checkpoint_path)
    return dev_ndcg, n_dev, dev_I, I
