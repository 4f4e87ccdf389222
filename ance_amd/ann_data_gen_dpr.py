"""ANN refresh job for DPR (NQ / TriviaQA), MI355X-native drop-in for drivers/run_ann_data_gen_dpr.py.

Differences from the MS MARCO job that the reference has and this keeps (run_ann_data_gen_dpr.py):
separate question / context BERT towers loaded from one DPR checkpoint FILE (``checkpoint-N``,
:46-60,112-132); four collections (train-query, test-query, trivia-test-query, passages, :209-230);
no query chunking; dev metric = top-20/100 answer-hit accuracy on NQ and TriviaQA test questions
(:243-252,312-340); negatives filtered by answer-string match (:281-309); ``ann_ndcg_N`` carries
``top20, top100, top20_trivia, top100_trivia, checkpoint`` (:275-278).
"""
import argparse
import ast
import csv
import json
import logging
import os
import random
import time

import numpy as np

from . import ann_data_gen as adg
from . import dpr
from .cache import TokenCache

logger = logging.getLogger(__name__)


def get_latest_checkpoint(args):
    """Newest ``training_dir/checkpoint-N`` FILE, else ``init_model_dir`` (run_ann_data_gen_dpr.py:46-60)."""
    if not os.path.exists(args.training_dir):
        return args.init_model_dir, 0
    nums = [adg.get_checkpoint_no(s) for s in next(os.walk(args.training_dir))[2] if s.startswith("checkpoint-")]
    if nums:
        return os.path.join(args.training_dir, "checkpoint-" + str(max(nums))), max(nums)
    return args.init_model_dir, 0


def load_mapping(data_dir, name):
    """``pid \\t offset`` lines (data/DPR_data.py:132-144)."""
    pid2offset, offset2pid = {}, {}
    with open(os.path.join(data_dir, name), "r") as f:
        for line in f:
            a, b = line.split("\t")
            pid2offset[int(a)] = int(b)
            offset2pid[int(b)] = int(a)
    return pid2offset, offset2pid


def load_data(args):
    """(passage_text {offset: (text, title)}, train_pos_id, train_answers, test_answers, trivia_answers)
    from psgs_w100.tsv, nq-test.csv, trivia-test.csv, train-ann (run_ann_data_gen_dpr.py:63-109).  Answer
    lists are Python literals in the files; they are parsed with ast.literal_eval, not eval."""
    pid2offset, _ = load_mapping(args.data_dir, "pid2offset")
    train_pos_id, train_answers, test_answers, trivia_answers = [], [], [], []
    with open(os.path.join(args.data_dir, "train-ann"), "r", encoding="utf8") as f:
        for row in csv.reader(f, delimiter="\t"):
            train_pos_id.append(int(row[1]))
            train_answers.append(ast.literal_eval(row[2]))
    with open(os.path.join(args.test_qa_path, "nq-test.csv"), "r", encoding="utf-8") as f:
        for row in csv.reader(f, delimiter="\t"):
            test_answers.append(ast.literal_eval(row[1]))
    with open(os.path.join(args.trivia_test_qa_path, "trivia-test.csv"), "r", encoding="utf-8") as f:
        for row in csv.reader(f, delimiter="\t"):
            trivia_answers.append(ast.literal_eval(row[1]))
    passage_text = {}
    with open(os.path.join(args.passage_path, "psgs_w100.tsv"), "r", encoding="utf-8") as f:
        for row in csv.reader(f, delimiter="\t"):
            if row[0] != "id":
                passage_text[pid2offset[int(row[0])]] = (row[1], row[2])
    return passage_text, train_pos_id, train_answers, test_answers, trivia_answers


def generate_new_ann(args, output_num, checkpoint_path, preloaded_data, latest_step_num, engine=None, model=None,
                     dist=None, pool=None):
    dist = dist or adg.Dist()
    if engine is None:
        engine = adg.HipEngine(getattr(args, "device", None))
    if model is None:
        from .encoder import load_model
        model = load_model("dpr", checkpoint_path, max_seq_length=args.max_seq_length,
                           max_tokens=getattr(args, "max_tokens", None) or adg.DRIVER_MAX_TOKENS, device=getattr(args, "device", None),
                           precision=getattr(args, "encoder_precision", None))

    def enc(name, is_query):
        cache = TokenCache(os.path.join(args.data_dir, name))
        emb, row0, n = adg.encode_collection(engine, dist, model, cache, is_query)
        return emb, row0, n

    logger.info("***** inference of train query *****")
    q_local, _, n_q = enc("train-query", True)
    logger.info("***** inference of dev query *****")
    dq_local, _, n_dq = enc("test-query", True)
    tq_local, _, n_tq = enc("trivia-test-query", True)
    logger.info("***** inference of passages *****")
    p_local, p_row0, n_rows = enc("passages", False)
    logger.info("***** Done passage inference *****")

    q_all = adg.gather_queries(dist, q_local, n_q)
    dq_all = adg.gather_queries(dist, dq_local, n_dq)
    tq_all = adg.gather_queries(dist, tq_local, n_tq)
    bases = adg.shard_row_bases(n_rows, dist.world)
    _, dev_I = adg.sharded_search(engine, dist, p_local, p_row0, dq_all, 100, row_bases=bases)
    _, triv_I = adg.sharded_search(engine, dist, p_local, p_row0, tq_all, 100, row_bases=bases)
    _, I = adg.sharded_search(engine, dist, p_local, p_row0, q_all, args.topk_training, row_bases=bases)
    logger.info("***** Done ANN Index *****")

    result = None
    if dist.rank == 0:
        passage_text, train_pos_id, train_answers, test_answers, trivia_answers = preloaded_data
        dev_I, triv_I, I = engine.to_numpy(dev_I), engine.to_numpy(triv_I), engine.to_numpy(I)
        p2id = np.arange(n_rows, dtype=np.int64)
        matcher = dpr.AnswerMatcher(passage_text)
        top_k_hits = dpr.validate(matcher, test_answers, dev_I, np.arange(n_dq), p2id, pool=pool)
        top_k_hits_trivia = dpr.validate(matcher, trivia_answers, triv_I, np.arange(n_tq), p2id, pool=pool)
        q2id = np.arange(n_q, dtype=np.int64)
        neg = dpr.generate_negative_passage_ids(matcher, train_answers, q2id, p2id, I, train_pos_id, args.negative_sample,
                                                pool=pool)
        os.makedirs(args.output_dir, exist_ok=True)
        train_path = os.path.join(args.output_dir, "ann_training_data_" + str(output_num))
        with open(train_path + ".tmp", "w") as f:
            query_range = list(range(I.shape[0]))
            random.shuffle(query_range)
            for query_idx in query_range:
                qid = int(q2id[query_idx])
                f.write("{}\t{}\t{}\n".format(qid, train_pos_id[qid], ",".join(str(p) for p in neg[qid])))
        os.replace(train_path + ".tmp", train_path)
        payload = {"top20": top_k_hits[19], "top100": top_k_hits[99], "top20_trivia": top_k_hits_trivia[19],
                   "top100_trivia": top_k_hits_trivia[99], "checkpoint": checkpoint_path}
        ndcg_path = os.path.join(args.output_dir, "ann_ndcg_" + str(output_num))
        with open(ndcg_path + ".tmp", "w") as f:
            json.dump(payload, f)
        os.replace(ndcg_path + ".tmp", ndcg_path)
        result = payload
    if hasattr(engine, "release_index"):
        engine.release_index()
    dist.barrier()
    return result


def ann_data_gen(args, engine=None, dist=None, preloaded=None, pool=None):
    """Poll loop (run_ann_data_gen_dpr.py:519-553).  ``preloaded`` / ``pool``: data and has_answer workers created
    by ``main`` before the GPU was touched (otherwise loaded here, single-process matching)."""
    dist = dist or adg.Dist()
    last_checkpoint = args.last_checkpoint_dir
    ann_no, _, _ = adg.get_latest_ann_data(args.output_dir)
    output_num = ann_no + 1
    if dist.rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        os.makedirs(args.cache_dir, exist_ok=True)
        if preloaded is None:
            preloaded = load_data(args)
    while args.end_output_num == -1 or output_num <= args.end_output_num:
        # rank 0 decides, every rank follows (see ann_data_gen.ann_data_gen)
        next_checkpoint, latest_step_num = dist.broadcast_object(get_latest_checkpoint(args) if dist.rank == 0 else None)
        if args.only_keep_latest_embedding_file:
            latest_step_num = 0
        if next_checkpoint == last_checkpoint:
            time.sleep(getattr(args, "poll_seconds", 60))
        else:
            logger.info("start generate ann data number %d", output_num)
            generate_new_ann(args, output_num, next_checkpoint, preloaded, latest_step_num, engine=engine, dist=dist,
                             pool=pool)
            output_num += 1
            last_checkpoint = next_checkpoint
        dist.barrier()


def get_arguments(argv=None):
    """Flags of run_ann_data_gen_dpr.py:343-495 (same names and defaults)."""
    p = argparse.ArgumentParser()
    for name in ("data_dir", "training_dir", "init_model_dir", "model_type", "output_dir", "cache_dir"):
        p.add_argument("--" + name, required=True, type=str)
    p.add_argument("--last_checkpoint_dir", default="", type=str)
    p.add_argument("--end_output_num", default=-1, type=int)
    p.add_argument("--max_seq_length", default=128, type=int)
    p.add_argument("--max_query_length", default=64, type=int)
    p.add_argument("--max_doc_character", default=10000, type=int)
    p.add_argument("--per_gpu_eval_batch_size", default=128, type=int)
    p.add_argument("--ann_chunk_factor", default=5, type=int)
    p.add_argument("--topk_training", default=500, type=int)
    p.add_argument("--negative_sample", default=5, type=int)
    p.add_argument("--ann_measure_topk_mrr", default=False, action="store_true")
    p.add_argument("--only_keep_latest_embedding_file", default=False, action="store_true")
    p.add_argument("--no_cuda", action="store_true")
    p.add_argument("--local_rank", "--local-rank", type=int, default=-1)
    p.add_argument("--server_ip", type=str, default="")
    p.add_argument("--server_port", type=str, default="")
    p.add_argument("--passage_path", default=None, type=str, required=True)
    p.add_argument("--test_qa_path", default=None, type=str, required=True)
    p.add_argument("--trivia_test_qa_path", default=None, type=str, required=True)
    p.add_argument("--max_tokens", default=adg.DRIVER_MAX_TOKENS, type=int)
    p.add_argument("--seed", default=None, type=int)
    p.add_argument("--encoder_precision", default=None, choices=["fp16", "split", "fp32"],
                   help="see ance_amd.ann_data_gen: split (default) is fp32-grade like the reference's forward, fp16 the fast mode")
    p.add_argument("--host_workers", default=None, type=int, help="has_answer worker processes on rank 0 (default: up to 32)")
    return p.parse_args(argv)


def main(argv=None):
    args = get_arguments(argv)
    # Rank 0 loads the corpus text and fork()s the has_answer workers BEFORE the GPU / RCCL are initialised
    # (the children share the 21 M-passage dict copy-on-write; forking after HIP initialisation is not safe).
    preloaded = pool = None
    if int(os.environ.get("RANK", "0")) == 0:
        preloaded = load_data(args)
        pool = dpr.AnswerPool(preloaded[0], getattr(args, "host_workers", None))
    adg.set_env(args)
    if args.seed is not None:
        random.seed(args.seed)
    try:
        ann_data_gen(args, preloaded=preloaded, pool=pool)
    finally:
        if pool is not None:
            pool.close()


if __name__ == "__main__":
    main()
