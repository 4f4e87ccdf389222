"""Tokenised-cache producer for MS MARCO passage / document data: the step
right before the hot path.  Same command line, same inputs and byte-identical outputs as the
reference's ``data/msmarco_data.py:126-272``:

    passages, passages_meta, pid2offset.pickle,
    train-query, train-query_meta, train-qrel.tsv, dev-query, dev-query_meta, dev-qrel.tsv, qid2offset.pickle

Record = 4-byte big-endian ``passage_len`` + ``L`` little-endian int32 token ids (what
``ance_amd.cache.TokenCache`` memory-maps and ``ance_encode_records`` consumes verbatim).

Ordering contract kept from the reference: the input file is dealt line ``i`` -> split ``i % 32``
(utils/util.py:332-346) and the splits are concatenated in split order (:246-254), so offset ids --
and therefore every ``ann_training_data_N`` later written against them -- are identical.  The 32
splits are a property of the format; the number of worker PROCESSES is independent (``--n_workers``,
default one per core up to 32), each worker handles whole splits and reads the input once.
"""
import argparse
import csv
import gzip
import json
import os
import pickle
from multiprocessing import get_context

import numpy as np

N_SPLITS = 32  # fixed by the reference (multi_file_process(args, 32, ...)); determines the offset ids


def pad_input_ids(input_ids, max_length, pad_on_left=False, pad_token=0):
    """utils/util.py:149-163."""
    padding_length = max_length - len(input_ids)
    if padding_length <= 0:
        return input_ids[:max_length]
    padding = [pad_token] * padding_length
    return padding + input_ids if pad_on_left else input_ids + padding


def _encode(tokenizer, text, max_length):
    """``tokenizer.encode(text, add_special_tokens=True, max_length=max_length)`` with the truncation
    the reference relied on (transformers 2.x truncated whenever ``max_length`` was given; newer
    releases only do so with ``truncation=True``)."""
    try:
        return tokenizer.encode(text, add_special_tokens=True, max_length=max_length, truncation=True)
    except TypeError:
        return tokenizer.encode(text, add_special_tokens=True, max_length=max_length)


def _record(item_id, ids, max_length, pad_token_id):
    n = min(len(ids), max_length)
    body = np.array(pad_input_ids(ids, max_length, pad_token=pad_token_id), np.int32).tobytes()
    return item_id.to_bytes(8, "big") + n.to_bytes(4, "big") + body


def PassagePreprocessingFn(args, line, tokenizer):
    """data/msmarco_data.py:213-246: 8-byte id + record."""
    if args.data_type == 0:
        line_arr = line.split("\t")
        p_id = int(line_arr[0][1:])  # remove "D"
        url = line_arr[1].rstrip()
        title = line_arr[2].rstrip()
        p_text = line_arr[3].rstrip()
        full_text = url + " " + tokenizer.sep_token + " " + title + " " + tokenizer.sep_token + " " + p_text
        full_text = full_text[:args.max_doc_character]
    else:
        line_arr = line.strip().split("\t")
        p_id = int(line_arr[0])
        full_text = line_arr[1].rstrip()[:args.max_doc_character]
    return _record(p_id, _encode(tokenizer, full_text, args.max_seq_length), args.max_seq_length, tokenizer.pad_token_id)


def QueryPreprocessingFn(args, line, tokenizer):
    """data/msmarco_data.py:249-260."""
    line_arr = line.split("\t")
    q_id = int(line_arr[0])
    ids = _encode(tokenizer, line_arr[1].rstrip(), args.max_query_length)
    return _record(q_id, ids, args.max_query_length, tokenizer.pad_token_id)


def load_tokenizer(args):
    """The tokenizer of ``args.model_type`` (model/models.py MSMarcoConfigDict: RoBERTa for the
    ``rdot_nll*`` family, BERT for ``dpr``), loaded from local files only."""
    factory = getattr(args, "tokenizer_factory", None)
    if factory is not None:
        return factory()
    import transformers
    cls = transformers.BertTokenizer if args.model_type.startswith("dpr") else transformers.RobertaTokenizer
    return cls.from_pretrained(args.model_name_or_path, do_lower_case=True, cache_dir=None)


def _open_text(path):
    return gzip.open(path, "rt", encoding="utf8") if path[-2:] == "gz" else open(path, "r", encoding="utf-8")


def _tokenize_splits(job):
    """One worker: lines ``idx`` with ``idx % N_SPLITS`` in ``splits`` -> ``{out_path}_split{idx % N_SPLITS}``."""
    args, splits, in_path, out_path, fn_name = job
    tokenizer = load_tokenizer(args)
    if isinstance(fn_name, tuple):  # (module, function): other producers (ance_amd.dpr_data) reuse the split engine
        import importlib
        line_fn = getattr(importlib.import_module(fn_name[0]), fn_name[1])
    else:
        line_fn = PassagePreprocessingFn if fn_name == "passage" else QueryPreprocessingFn
    outs = {s: open("{}_split{}".format(out_path, s), "wb") for s in splits}
    try:
        with _open_text(in_path) as in_f:
            for idx, line in enumerate(in_f):
                f = outs.get(idx % N_SPLITS)
                if f is not None:
                    f.write(line_fn(args, line, tokenizer))
    finally:
        for f in outs.values():
            f.close()
    return len(splits)


def multi_file_process(args, num_process, in_path, out_path, fn_name):
    """utils/util.py:349-365 with the worker count decoupled from the split count."""
    assert num_process == N_SPLITS
    n_workers = int(getattr(args, "n_workers", 0) or 0)
    if n_workers <= 0:
        n_workers = min(N_SPLITS, os.cpu_count() or 1)
    n_workers = max(1, min(n_workers, N_SPLITS))
    jobs = [(args, list(range(w, N_SPLITS, n_workers)), in_path, out_path, fn_name) for w in range(n_workers)]
    if n_workers == 1:
        _tokenize_splits(jobs[0])
        return
    with get_context("fork").Pool(n_workers) as pool:
        list(pool.imap_unordered(_tokenize_splits, jobs))


def numbered_byte_file_generator(base_path, file_no, record_size):
    """utils/util.py:246-254."""
    for i in range(file_no):
        with open("{}_split{}".format(base_path, i), "rb") as f:
            while True:
                b = f.read(record_size)
                if not b:
                    break
                yield b


def _remove_splits(base_path):
    for i in range(N_SPLITS):
        try:
            os.remove("{}_split{}".format(base_path, i))
        except OSError:
            pass


def _qrel_reader(f, data_type):
    return csv.reader(f, delimiter=" ") if data_type == 0 else csv.reader(f, delimiter="\t")


def write_query_rel(args, pid2offset, query_file, positive_id_file, out_query_file, out_id_file):
    """data/msmarco_data.py:18-123: queries that have a label, in split order, and offset-space qrels."""
    qrel_path = os.path.join(args.data_dir, positive_id_file)
    query_positive_id = set()
    with _open_text(qrel_path) as f:
        for [topicid, _, docid, rel] in _qrel_reader(f, args.data_type):
            query_positive_id.add(int(topicid))
    out_query_path = os.path.join(args.out_data_dir, out_query_file)
    multi_file_process(args, N_SPLITS, os.path.join(args.data_dir, query_file), out_query_path, "query")
    qid2offset = {}
    idx = 0
    with open(out_query_path, "wb") as f:
        for record in numbered_byte_file_generator(out_query_path, N_SPLITS, 8 + 4 + args.max_query_length * 4):
            q_id = int.from_bytes(record[:8], "big")
            if q_id not in query_positive_id:
                continue  # not in the label set
            f.write(record[8:])
            qid2offset[q_id] = idx
            idx += 1
    if not getattr(args, "keep_splits", True):
        _remove_splits(out_query_path)
    with open(os.path.join(args.out_data_dir, "qid2offset.pickle"), "wb") as handle:
        pickle.dump(qid2offset, handle, protocol=4)
    with open(out_query_path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": idx, "embedding_size": args.max_query_length}, f)
    print("Total lines written: " + str(idx))
    out_line_count = 0
    with _open_text(qrel_path) as f, open(os.path.join(args.out_data_dir, out_id_file), "w", encoding="utf-8") as out_id:
        for [topicid, _, docid, rel] in _qrel_reader(f, args.data_type):
            topicid = int(topicid)
            docid = int(docid[1:]) if args.data_type == 0 else int(docid)
            out_id.write(str(qid2offset[topicid]) + "\t" + str(pid2offset[docid]) + "\t" + rel + "\n")
            out_line_count += 1
    print("Total lines written: " + str(out_line_count))


def preprocess(args):
    """data/msmarco_data.py:126-210."""
    in_passage_path = os.path.join(args.data_dir, "msmarco-docs.tsv" if args.data_type == 0 else "collection.tsv")
    out_passage_path = os.path.join(args.out_data_dir, "passages")
    if os.path.exists(out_passage_path):
        print("preprocessed data already exist, exit preprocessing")
        return
    print("start passage file split processing")
    multi_file_process(args, N_SPLITS, in_passage_path, out_passage_path, "passage")
    print("start merging splits")
    pid2offset = {}
    out_line_count = 0
    with open(out_passage_path, "wb") as f:
        for idx, record in enumerate(numbered_byte_file_generator(out_passage_path, N_SPLITS,
                                                                  8 + 4 + args.max_seq_length * 4)):
            f.write(record[8:])
            pid2offset[int.from_bytes(record[:8], "big")] = idx
            out_line_count += 1
    if not getattr(args, "keep_splits", True):
        _remove_splits(out_passage_path)
    print("Total lines written: " + str(out_line_count))
    with open(out_passage_path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": out_line_count, "embedding_size": args.max_seq_length}, f)
    with open(os.path.join(args.out_data_dir, "pid2offset.pickle"), "wb") as handle:
        pickle.dump(pid2offset, handle, protocol=4)
    print("done saving pid2offset")
    if args.data_type == 0:
        write_query_rel(args, pid2offset, "msmarco-doctrain-queries.tsv", "msmarco-doctrain-qrels.tsv", "train-query",
                        "train-qrel.tsv")
        write_query_rel(args, pid2offset, "msmarco-test2019-queries.tsv", "2019qrels-docs.txt", "dev-query",
                        "dev-qrel.tsv")
    else:
        write_query_rel(args, pid2offset, "queries.train.tsv", "qrels.train.tsv", "train-query", "train-qrel.tsv")
        write_query_rel(args, pid2offset, "queries.dev.small.tsv", "qrels.dev.small.tsv", "dev-query", "dev-qrel.tsv")


def get_arguments(argv=None):
    """Flags of data/msmarco_data.py:372-428 plus ``--n_workers`` / ``--remove_splits``."""
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", default=None, type=str, required=True, help="The input data dir")
    p.add_argument("--out_data_dir", default=None, type=str, required=True, help="The output data dir")
    p.add_argument("--model_type", default=None, type=str, required=True)
    p.add_argument("--model_name_or_path", default=None, type=str, required=True)
    p.add_argument("--max_seq_length", default=128, type=int)
    p.add_argument("--max_query_length", default=64, type=int)
    p.add_argument("--max_doc_character", default=10000, type=int, help="used before tokenizer to save tokenizer latency")
    p.add_argument("--data_type", default=0, type=int, help="0 for doc, 1 for passage")
    p.add_argument("--n_workers", default=0, type=int, help="worker processes (0: one per core, at most 32)")
    p.add_argument("--remove_splits", action="store_true", help="delete the *_split{i} intermediates (the reference keeps them)")
    args = p.parse_args(argv)
    args.keep_splits = not args.remove_splits
    return args


def main(argv=None):
    args = get_arguments(argv)
    os.makedirs(args.out_data_dir, exist_ok=True)
    preprocess(args)


if __name__ == "__main__":
    main()
