"""Mid-scale run of the real driver (ance_amd.ann_data_gen.ann_data_gen) on a synthetic MS MARCO-shaped set:
many encode blocks through the pinned H2D ring, chunked queries, the native host stage, the file contract.
Prints a timing breakdown.  Usage on the GPU box: python scripts/job_midscale.py [n_passages n_train]"""
import json
import logging
import os
import random
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from safetensors.torch import save_file

import bench
from ance_amd import ann_data_gen as adg
from oracle import synth  # data generator only (test infrastructure): this script is a validation run, not the product

n_p = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
tmp = tempfile.mkdtemp(prefix="ance_job_")
t0 = time.time()
data = os.path.join(tmp, "data")
synth.make_msmarco_like(data, n_passages=n_p, n_train=n_q, n_dev=2000, L=128, Lq=64, seed=7, plant_frac=0.2)
t_data = time.time() - t0
sd = bench.random_init_roberta_base(torch, 12, seed=0)
ckpt = os.path.join(tmp, "train", "checkpoint-1000")
os.makedirs(ckpt)
save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(ckpt, "model.safetensors"))
open(os.path.join(ckpt, "scheduler.pt"), "w").write("commit marker")
out = os.path.join(tmp, "out")
args = types.SimpleNamespace(
    data_dir=data, training_dir=os.path.join(tmp, "train"), init_model_dir="/nonexistent", last_checkpoint_dir="",
    output_dir=out, cache_dir=out, model_type="rdot_nll", end_output_num=0, max_seq_length=128, max_query_length=64,
    ann_chunk_factor=1, topk_training=200, negative_sample=20, ann_measure_topk_mrr=False,
    only_keep_latest_embedding_file=False, inference=False, device=torch.device("cuda"), max_tokens=65536)
logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
random.seed(0)
t1 = time.time()
adg.ann_data_gen(args)
torch.cuda.synchronize()
t_job = time.time() - t1
no, train_path, nd = adg.get_latest_ann_data(out)
lines = open(train_path).read().splitlines()
neg_counts = [len(l.split("\t")[2].split(",")) for l in lines]
ok = (no == 0 and len(lines) == n_q and min(neg_counts) == 20 and max(neg_counts) == 20 and 0.0 <= nd["ndcg"] <= 1.0)
planted = sum(1 for l in lines[:2000] if l)  # file is shuffled; just a sanity touch
print(json.dumps({"passages": n_p, "train_queries": n_q, "data_gen_s": round(t_data, 1), "job_s": round(t_job, 2),
                  "passages_per_s_incl_everything": round((n_p + n_q + 2000) / t_job), "lines": len(lines),
                  "ndcg": nd["ndcg"], "ok": bool(ok)}))
