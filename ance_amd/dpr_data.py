"""Tokenised-cache producer for the DPR (NQ / TriviaQA + Wikipedia) data: same
inputs and byte-identical outputs as the reference's ``data/DPR_data.py:23-273``:

    passages, passages_meta, pid2offset            (psgs_w100.tsv; 32-way split order, header line skipped)
    train-query[_meta], train-ann, train-data      (nq-train.json / trivia-train.json, or both merged)
    dev-query, dev-ann, dev-data, dev-query-trivia, dev-ann-trivia, dev-data-trivia
    test-query[_meta], trivia-test-query[_meta]    (qas csv: question \\t answers)

Passage record = 4-byte big-endian length + ``max_seq_length`` little-endian int32 ids of
``[CLS] title [SEP] text [SEP]``; when the tokenizer returns more than ``max_seq_length`` ids the row
is cut and its last id forced to ``[SEP]`` (data/DPR_data.py:243-246), the header keeps the
untruncated length exactly as the reference writes it.
"""
import argparse
import csv
import json
import os

import numpy as np

from . import msmarco_data as _md


def normalize_question(question):
    """data/DPR_data.py:17-20."""
    if question[-1] == "?":
        question = question[:-1]
    return question


def _fit(token_ids, seq_len, tokenizer):
    if len(token_ids) < seq_len:
        token_ids = token_ids + [tokenizer.pad_token_id] * (seq_len - len(token_ids))
    if len(token_ids) > seq_len:
        token_ids = token_ids[0:seq_len]
        token_ids[-1] = tokenizer.sep_token_id
    return token_ids


def _encode(tokenizer, text, max_length, text_pair=None):
    kw = dict(add_special_tokens=True, max_length=max_length)
    if text_pair is not None:
        kw["text_pair"] = text_pair
    try:
        return tokenizer.encode(text, truncation=True, **kw)
    except TypeError:
        return tokenizer.encode(text, **kw)


def PassagePreprocessingFn(args, line, tokenizer):
    """data/DPR_data.py:228-252 (the ``id \\t text \\t title`` header row yields nothing)."""
    line_arr = list(csv.reader([line], delimiter="\t"))[0]
    if line_arr[0] == "id":
        return bytearray()
    p_id = int(line_arr[0])
    token_ids = _encode(tokenizer, line_arr[2], args.max_seq_length, text_pair=line_arr[1])
    passage_len = len(token_ids)
    token_ids = _fit(token_ids, args.max_seq_length, tokenizer)
    return p_id.to_bytes(8, "big") + passage_len.to_bytes(4, "big") + np.array(token_ids, np.int32).tobytes()


def QueryPreprocessingFn(args, qid, text, tokenizer):
    """data/DPR_data.py:255-270."""
    token_ids = _encode(tokenizer, text, args.max_seq_length)
    passage_len = len(token_ids)
    token_ids = _fit(token_ids, args.max_seq_length, tokenizer)
    return passage_len.to_bytes(4, "big") + np.array(token_ids, np.int32).tobytes()


def _write_meta(path, n, L):
    with open(path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": n, "embedding_size": L}, f)


def write_qas_query(args, qas_file, out_query_file):
    """data/DPR_data.py:23-52: test questions in file order, ids implicit."""
    tokenizer = _md.load_tokenizer(args)
    out_query_path = os.path.join(args.out_data_dir, out_query_file)
    qid = 0
    with open(os.path.join(args.answer_dir, qas_file), "r", encoding="utf-8") as f, open(out_query_path, "wb") as out_query:
        for row in csv.reader(f, delimiter="\t"):
            out_query.write(QueryPreprocessingFn(args, qid, normalize_question(row[0]), tokenizer))
            qid += 1
    _write_meta(out_query_path, qid, args.max_seq_length)


def write_query_rel(args, pid2offset, query_file, out_query_file, out_ann_file, out_train_file, passage_id_name="passage_id"):
    """data/DPR_data.py:54-124: samples with >= 1 positive and >= 1 hard negative; ``*-ann`` keeps the
    answers (as the Python repr the reference writes), ``*-data`` the retriever's hard negatives."""
    with open(os.path.join(args.question_dir, query_file), "r", encoding="utf-8") as f:
        data = json.load(f)
    data = [r for r in data if len(r["positive_ctxs"]) > 0]
    data = [r for r in data if len(r["hard_negative_ctxs"]) > 0]
    tokenizer = _md.load_tokenizer(args)
    out_query_path = os.path.join(args.out_data_dir, out_query_file)
    qid = 0
    with open(out_query_path, "wb") as out_query, \
            open(os.path.join(args.out_data_dir, out_ann_file), "w", encoding="utf-8") as out_ann, \
            open(os.path.join(args.out_data_dir, out_train_file), "w", encoding="utf-8") as out_training:
        for sample in data:
            question = normalize_question(sample["question"])
            first_pos_pid = pid2offset[int(sample["positive_ctxs"][0][passage_id_name])]
            neg_pids = [str(pid2offset[int(c[passage_id_name])]) for c in sample["hard_negative_ctxs"]]
            out_ann.write("{}\t{}\t{}\n".format(qid, first_pos_pid, sample["answers"]))
            out_training.write("{}\t{}\t{}\n".format(qid, first_pos_pid, ",".join(neg_pids)))
            out_query.write(QueryPreprocessingFn(args, qid, question, tokenizer))
            qid += 1
    print("Total lines written: " + str(qid))
    _write_meta(out_query_path, qid, args.max_seq_length)


def write_mapping(args, id2offset, out_name):
    """data/DPR_data.py:127-134."""
    with open(os.path.join(args.out_data_dir, out_name), "w") as f:
        for k, v in id2offset.items():
            f.write("{}\t{}\n".format(k, v))


def load_mapping(data_dir, out_name):
    """data/DPR_data.py:137-149."""
    pid2offset, offset2pid = {}, {}
    with open(os.path.join(data_dir, out_name), "r") as f:
        for line in f:
            a = line.split("\t")
            pid2offset[int(a[0])] = int(a[1])
            offset2pid[int(a[1])] = int(a[0])
    return pid2offset, offset2pid


def preprocess(args):
    """data/DPR_data.py:152-226."""
    out = args.out_data_dir
    out_passage_path = os.path.join(out, "passages")
    if os.path.exists(out_passage_path):
        print("preprocessed data already exist, exit preprocessing")
        return
    print("start passage file split processing")
    _md.multi_file_process(args, _md.N_SPLITS, os.path.join(args.wiki_dir, "psgs_w100.tsv"), out_passage_path,
                           ("ance_amd.dpr_data", "PassagePreprocessingFn"))
    print("start merging splits")
    pid2offset = {}
    n = 0
    with open(out_passage_path, "wb") as f:
        for idx, record in enumerate(_md.numbered_byte_file_generator(out_passage_path, _md.N_SPLITS,
                                                                      8 + 4 + args.max_seq_length * 4)):
            f.write(record[8:])
            pid2offset[int.from_bytes(record[:8], "big")] = idx
            n += 1
    if not getattr(args, "keep_splits", True):
        _md._remove_splits(out_passage_path)
    print("Total lines written: " + str(n))
    _write_meta(out_passage_path, n, args.max_seq_length)
    write_mapping(args, pid2offset, "pid2offset")

    if args.data_type == 0:
        write_query_rel(args, pid2offset, "nq-train.json", "train-query", "train-ann", "train-data")
    elif args.data_type == 1:
        write_query_rel(args, pid2offset, "trivia-train.json", "train-query", "train-ann", "train-data", "psg_id")
    else:  # both, NQ rows first (the trainer relies on that order: data/DPR_data.py:215)
        write_query_rel(args, pid2offset, "nq-train.json", "train-query-nq", "train-ann-nq", "train-data-nq")
        write_query_rel(args, pid2offset, "trivia-train.json", "train-query-trivia", "train-ann-trivia", "train-data-trivia", "psg_id")
        with open(os.path.join(out, "train-query"), "wb") as q:
            for part in ("train-query-nq", "train-query-trivia"):
                with open(os.path.join(out, part), "rb") as src:
                    q.write(src.read())
        metas = [json.load(open(os.path.join(out, part + "_meta"), encoding="utf-8")) for part in ("train-query-nq", "train-query-trivia")]
        _write_meta(os.path.join(out, "train-query"), metas[0]["total_number"] + metas[1]["total_number"], args.max_seq_length)
        with open(os.path.join(out, "train-ann"), "w", encoding="utf-8") as a:
            for part in ("train-ann-nq", "train-ann-trivia"):
                with open(os.path.join(out, part), "r", encoding="utf-8") as src:
                    a.writelines(src.readlines())
    write_query_rel(args, pid2offset, "nq-dev.json", "dev-query", "dev-ann", "dev-data")
    write_query_rel(args, pid2offset, "trivia-dev.json", "dev-query-trivia", "dev-ann-trivia", "dev-data-trivia", "psg_id")
    write_qas_query(args, "nq-test.csv", "test-query")
    write_qas_query(args, "trivia-test.csv", "trivia-test-query")


def get_arguments(argv=None):
    """Flags of data/DPR_data.py:345-396 plus ``--n_workers`` / ``--remove_splits``."""
    p = argparse.ArgumentParser()
    p.add_argument("--out_data_dir", default="/webdata-nfs/jialliu/dpr/ann/ann_multi_data_256/", type=str)
    p.add_argument("--model_type", default="dpr", type=str)
    p.add_argument("--model_name_or_path", default="bert-base-uncased", type=str)
    p.add_argument("--max_seq_length", default=256, type=int)
    p.add_argument("--data_type", default=0, type=int, help="0 is nq, 1 is trivia, 2 is both")
    p.add_argument("--question_dir", type=str, help="location of the raw QnA question data")
    p.add_argument("--wiki_dir", type=str, help="location of the wiki corpus")
    p.add_argument("--answer_dir", type=str, help="location of the QnA answers for evaluation")
    p.add_argument("--n_workers", default=0, type=int)
    p.add_argument("--remove_splits", action="store_true")
    args = p.parse_args(argv)
    args.keep_splits = not args.remove_splits
    return args


def main(argv=None):
    args = get_arguments(argv)
    os.makedirs(args.out_data_dir, exist_ok=True)
    preprocess(args)


if __name__ == "__main__":
    main()
