"""Host-side stages around the search: query chunking, dev NDCG@10, hard-negative selection and
the ``ann_training_data_N`` / ``ann_ndcg_N`` writers (the file contract, seam B1).

Same behaviour as the reference functions they stand in for -- cited per function -- but written
over NumPy arrays instead of per-element Python lookups.  Randomness goes through the module-level
``random`` exactly where the reference consumes it (one ``random.shuffle`` of ``range(k)`` per
query in row order, then one shuffle of the query order), so a run is reproducible under
``random.seed`` and comparable with a seeded reference run.
"""
import json
import math
import os
import random

import numpy as np


def query_chunk(num_queries, output_num, chunk_factor):
    """Slice of the train queries refreshed at ``output_num`` (drivers/run_ann_data_gen.py:281-296).
    As in the reference the modulo comes first, so ``chunk_factor == 0`` raises."""
    effective_idx = output_num % chunk_factor
    if chunk_factor <= 0:
        chunk_factor = 1
    per = num_queries // chunk_factor
    start = per * effective_idx
    end = num_queries if effective_idx == chunk_factor - 1 else start + per
    return start, end


def _first_occurrence_mask(a):
    """Boolean mask of the first occurrence of each value, per row of a 2-D int array."""
    order = np.argsort(a, axis=1, kind="stable")
    s = np.take_along_axis(a, order, axis=1)
    first_sorted = np.ones_like(s, dtype=bool)
    first_sorted[:, 1:] = s[:, 1:] != s[:, :-1]
    mask = np.zeros_like(first_sorted)
    np.put_along_axis(mask, order, first_sorted, axis=1)
    return mask


def eval_dev_query(query_embedding2id, passage_embedding2id, dev_query_positive_id, I):
    """NDCG@10 of the dev queries (drivers/run_ann_data_gen.py:399-440): first 50 neighbours,
    row -> pid, keep the first row of each pid (MaxP), score = -rank; trec_eval ``ndcg_cut_10``
    (gain = rel, discount log2(rank + 1), ideal = judged rels sorted descending) averaged over the
    queries present in both the run and the qrels.  Returns (ndcg, n_queries)."""
    I = np.asarray(I)[:, :50]
    p2id = np.asarray(passage_embedding2id)
    valid = I >= 0
    pids = np.where(valid, p2id[np.where(valid, I, 0)], -1)
    keep = _first_occurrence_mask(pids) & valid
    disc = 1.0 / np.log2(np.arange(2, 12))  # ranks 1..10
    total, cnt = 0.0, 0
    # a later row with the same query id overwrites an earlier one (dict semantics of the reference)
    last_row = {}
    for row, qid in enumerate(np.asarray(query_embedding2id).tolist()):
        last_row[qid] = row
    for qid, row in last_row.items():
        rels = dev_query_positive_id.get(qid)
        if rels is None:
            continue
        ranked = pids[row][keep[row]][:10]
        dcg = 0.0
        for r, pid in enumerate(ranked.tolist()):
            g = rels.get(pid, 0)
            if g > 0:
                dcg += g * disc[r]
        ideal = sorted((g for g in rels.values() if g > 0), reverse=True)[:10]
        idcg = sum(g * disc[r] for r, g in enumerate(ideal))
        total += dcg / idcg if idcg > 0 else 0.0
        cnt += 1
    return (total / cnt if cnt else 0.0), cnt


def generate_negative_passage_ids(query_embedding2id, passage_embedding2id, training_query_positive_id, I,
                                  effective_q_id, negative_sample, select_topk, rank=0, verbose=True):
    """{qid: [negative pid, ...]} (drivers/run_ann_data_gen.py:339-396).

    Per query row: candidates = the k neighbours in a ``random.shuffle``d order (default) or the
    first ``negative_sample + 1`` in rank order (``--ann_measure_topk_mrr``); walk them, skip the
    positive (adding 1/rank to the MRR if rank <= 10), skip repeated pids, stop at
    ``negative_sample`` negatives."""
    I = np.asarray(I)
    p2id = np.asarray(passage_embedding2id)
    q2id = np.asarray(query_embedding2id).tolist()
    k = I.shape[1]
    out = {}
    mrr = 0.0
    num_queries = 0
    base_order = list(range(k))
    for row, qid in enumerate(q2id):
        if qid not in effective_q_id:
            continue
        num_queries += 1
        pos_pid = training_query_positive_id[qid]
        if select_topk:
            sel = I[row, :negative_sample + 1]
        else:
            order = base_order[:]
            random.shuffle(order)
            sel = I[row, order]
        pids = p2id[sel]
        is_pos = pids == pos_pid
        if select_topk and is_pos[:10].any():
            mrr += float((1.0 / (np.nonzero(is_pos[:10])[0] + 1)).sum())
        cand = pids[~is_pos]
        if cand.size:
            _, first = np.unique(cand, return_index=True)
            first.sort()
            negs = cand[first[:negative_sample]]
        else:
            negs = cand
        out[qid] = negs.tolist()
    if select_topk and verbose:
        print("Rank:" + str(rank) + " --- ANN MRR:" + str(mrr / max(num_queries, 1)))
    return out


def write_ann_files(output_dir, output_num, n_rows, query_embedding2id, effective_q_id, training_query_positive_id,
                    query_negative_passage, dev_ndcg, checkpoint_path, extra_metrics=None):
    """``ann_training_data_N`` then ``ann_ndcg_N`` -- data file first, the trainer discovers a
    refresh by the ndcg file (drivers/run_ann_data_gen.py:314-334; utils/util.py:229-243).
    Lines: ``qid \\t pos_pid \\t neg,neg,...`` in a ``random.shuffle``d query order."""
    q2id = np.asarray(query_embedding2id).tolist()
    train_path = os.path.join(output_dir, "ann_training_data_" + str(output_num))
    tmp = train_path + ".tmp"
    with open(tmp, "w") as f:
        query_range = list(range(n_rows))
        random.shuffle(query_range)
        for query_idx in query_range:
            qid = q2id[query_idx]
            if qid not in effective_q_id or qid not in training_query_positive_id:
                continue
            f.write("{}\t{}\t{}\n".format(qid, training_query_positive_id[qid],
                                          ",".join(str(p) for p in query_negative_passage[qid])))
    os.replace(tmp, train_path)
    payload = {"ndcg": dev_ndcg, "checkpoint": checkpoint_path}
    if extra_metrics:
        payload.update(extra_metrics)
    ndcg_path = os.path.join(output_dir, "ann_ndcg_" + str(output_num))
    with open(ndcg_path + ".tmp", "w") as f:
        json.dump(payload, f)
    os.replace(ndcg_path + ".tmp", ndcg_path)
    return train_path, ndcg_path


def load_positive_ids(data_dir):
    """train: {qid_offset: pid_offset} (last wins, rel must be "1"); dev: {qid: {pid: rel}}
    (drivers/run_ann_data_gen.py:74-100)."""
    train = {}
    with open(os.path.join(data_dir, "train-qrel.tsv"), "r", encoding="utf8") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) != 3:
                raise ValueError("train-qrel.tsv: expected 3 tab-separated columns, got %r" % line)
            assert parts[2] == "1"
            train[int(parts[0])] = int(parts[1])
    dev = {}
    with open(os.path.join(data_dir, "dev-qrel.tsv"), "r", encoding="utf8") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) != 3:
                raise ValueError("dev-qrel.tsv: expected 3 tab-separated columns, got %r" % line)
            dev.setdefault(int(parts[0]), {})[int(parts[1])] = int(parts[2])
    return train, dev
