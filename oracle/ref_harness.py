"""Runs the REAL reference (microsoft/ANCE at /root/reference) on CPU -- build container only.

Test infrastructure used to pin the oracle and to generate the golden vectors under
tests/golden/ (generator: tests/golden/make_golden.py).  /root/reference does not exist on the
GPU box, so nothing that runs there imports this module's ``load_reference``.

Only third-party leaves absent from this image are replaced (SURVEY.md section 8c recipe):
  faiss        -> oracle.search_ref.OracleIndexFlatIP (NumPy/BLAS flat IP, canonical tie-break)
  pytrec_eval  -> oracle.ann_ref.RelevanceEvaluator   (NDCG@10 / MAP restatement)
  tensorboardX -> no-op SummaryWriter
Every line of the reference's own logic (drivers/run_ann_data_gen.py, utils/util.py,
data/msmarco_data.py, model/models.py) executes unmodified.
"""
import importlib
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("ANCE_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "drivers"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_LOADED = None


def load_reference():
    """Import the reference driver module; returns a namespace with the pieces tests need."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    import torch  # noqa: F401
    import transformers  # noqa: F401

    from oracle import ann_ref, search_ref

    for p in (REF_ROOT, os.path.join(REF_ROOT, "drivers")):
        if p not in sys.path:
            sys.path.insert(0, p)

    if "faiss" not in sys.modules:
        _stub("faiss", IndexFlatIP=search_ref.OracleIndexFlatIP, omp_set_num_threads=lambda n: None)
    if "pytrec_eval" not in sys.modules:
        _stub("pytrec_eval", RelevanceEvaluator=ann_ref.RelevanceEvaluator)
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        if "tensorboardX" not in sys.modules:
            class SummaryWriter:  # pragma: no cover - trivial
                def __init__(self, *a, **k):
                    pass

                def add_scalar(self, *a, **k):
                    pass

                def close(self):
                    pass
            _stub("tensorboardX", SummaryWriter=SummaryWriter)

    models = importlib.import_module("model.models")
    util = importlib.import_module("utils.util")
    msmarco_data = importlib.import_module("data.msmarco_data")
    # transformers 5 dropped AdamW; the driver imports it at module top but never uses it here.
    tr = sys.modules["transformers"]
    if not hasattr(tr, "AdamW"):
        tr.AdamW = torch.optim.AdamW
    driver = importlib.import_module("run_ann_data_gen")
    _LOADED = types.SimpleNamespace(models=models, util=util, msmarco_data=msmarco_data, driver=driver)
    return _LOADED


def roberta_config(n_layers=12, **kw):
    from transformers import RobertaConfig
    cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1,
                        layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                        num_labels=2, num_hidden_layers=n_layers, return_dict=False, **kw)
    cfg._attn_implementation = "eager"
    return cfg


def bert_config(n_layers=12, **kw):
    from transformers import BertConfig
    cfg = BertConfig(num_hidden_layers=n_layers, return_dict=False, **kw)
    cfg._attn_implementation = "eager"
    return cfg


def build_reference_model(kind="rdot_nll", n_layers=12, seed=0):
    """Random-init reference model (reference's own class and its own _init_weights)."""
    import torch
    ref = load_reference()
    torch.manual_seed(seed)
    if kind == "rdot_nll":
        m = ref.models.RobertaDot_NLL_LN(roberta_config(n_layers))
    elif kind == "rdot_nll_multi_chunk":
        m = ref.models.RobertaDot_CLF_ANN_NLL_MultiChunk(roberta_config(n_layers))
    elif kind == "bert":
        m = ref.models.HFBertEncoder(bert_config(n_layers))
    else:
        raise ValueError(kind)
    m.eval()
    return m


def run_generate_new_ann(data_dir, output_dir, model, output_num=0, checkpoint_path="/x/checkpoint-100/",
                         step=100, seed=0, **argkw):
    """Run the reference's generate_new_ann end-to-end on CPU with ``model``."""
    import random
    import torch
    ref = load_reference()
    G = ref.driver
    os.makedirs(output_dir, exist_ok=True)
    args = types.SimpleNamespace(
        data_dir=data_dir, output_dir=output_dir, cache_dir=output_dir, local_rank=-1, rank=0,
        device=torch.device("cpu"), per_gpu_eval_batch_size=16, max_seq_length=128, max_query_length=64,
        inference=False, topk_training=200, negative_sample=20, ann_chunk_factor=5,
        ann_measure_topk_mrr=False, model_type="rdot_nll", world_size=1)
    for k, v in argkw.items():
        setattr(args, k, v)
    wrapped = types.SimpleNamespace(module=model, eval=model.eval)
    G.load_model = lambda a, ckpt: (None, None, wrapped)
    train_pos, dev_pos = G.load_positive_ids(args)
    random.seed(seed)
    return G.generate_new_ann(args, output_num, checkpoint_path, train_pos, dev_pos, step)


def run_reference_preprocess(data_dir, out_data_dir, data_type, tokenizer_cls, max_seq_length=16, max_query_length=8,
                             model_type="rdot_nll"):
    """Run the reference's own data/msmarco_data.py ``preprocess`` (32 forked tokenizer processes, split
    merge, pid2offset / qid2offset, offset-space qrels) with ``tokenizer_cls`` standing in for the
    pretrained tokenizer class (no vocabulary files offline)."""
    ref = load_reference()
    cfg = ref.models.MSMarcoConfigDict[model_type]
    old = cfg.tokenizer_class
    cfg.tokenizer_class = tokenizer_cls
    try:
        os.makedirs(out_data_dir, exist_ok=True)
        args = types.SimpleNamespace(data_dir=data_dir, out_data_dir=out_data_dir, model_type=model_type,
                                     model_name_or_path="unused", max_seq_length=max_seq_length,
                                     max_query_length=max_query_length, max_doc_character=10000, data_type=data_type)
        ref.msmarco_data.preprocess(args)
    finally:
        cfg.tokenizer_class = old


def notebook_eval_dev_query():
    """The ``EvalDevQuery`` of evaluation/"Calculate Metrics.ipynb" (cell 8) as a callable: the cell's own
    source executed with the reference's utils/msmarco_eval.compute_metrics (real code) and
    oracle.ann_ref.RelevanceEvaluator standing in for pytrec_eval."""
    import json
    ref = load_reference()  # noqa: F841  (sys.path for utils.*)
    from oracle import ann_ref
    msmarco_eval = importlib.import_module("utils.msmarco_eval")
    with open(os.path.join(REF_ROOT, "evaluation", "Calculate Metrics.ipynb")) as f:
        nb = json.load(f)
    src = None
    for c in nb["cells"]:
        if c["cell_type"] == "code" and "def EvalDevQuery" in "".join(c["source"]):
            src = "".join(c["source"])
    ns = {"pytrec_eval": types.SimpleNamespace(RelevanceEvaluator=ann_ref.RelevanceEvaluator),
          "compute_metrics": msmarco_eval.compute_metrics}
    exec(compile(src, "Calculate Metrics.ipynb#cell8", "exec"), ns)
    return ns["EvalDevQuery"]


def run_reference_dpr_preprocess(wiki_dir, question_dir, answer_dir, out_data_dir, data_type, tokenizer_cls,
                                 max_seq_length=24):
    """Run the reference's own data/DPR_data.py ``preprocess`` with ``tokenizer_cls`` standing in for
    bert-base-uncased (data_type 0 = NQ, 1 = TriviaQA; 2 dereferences row 58,812 of the merged training
    queries -- data/DPR_data.py:215 -- and cannot run on small inputs)."""
    ref = load_reference()
    dpr_data = importlib.import_module("data.DPR_data")
    cfg = ref.models.MSMarcoConfigDict["dpr"]
    old = cfg.tokenizer_class
    cfg.tokenizer_class = tokenizer_cls
    try:
        os.makedirs(out_data_dir, exist_ok=True)
        args = types.SimpleNamespace(out_data_dir=out_data_dir, model_type="dpr", model_name_or_path="unused",
                                     max_seq_length=max_seq_length, data_type=data_type, question_dir=question_dir,
                                     wiki_dir=wiki_dir, answer_dir=answer_dir)
        dpr_data.preprocess(args)
    finally:
        cfg.tokenizer_class = old
