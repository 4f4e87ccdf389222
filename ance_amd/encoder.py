"""Host-side mirror of the reference's dual-encoder interface, backed by the HIP encoder.

Reference interface mirrored (same names, argument meaning, output shapes):
  model/models.py:149-157   RobertaDot_NLL_LN.query_emb / body_emb      -> [B, 768]
  model/models.py:165-199   RobertaDot_CLF_ANN_NLL_MultiChunk.body_emb  -> [B, C, 768]
  model/models.py:254-259   BiEncoder.query_emb / body_emb (DPR, BERT)  -> [B, 768]
called as ``model.module.query_emb(input_ids=..., attention_mask=...)`` by
drivers/run_ann_data_gen.py:171-180 (seam B4).  Weights are read from a HF checkpoint directory
(``checkpoint-N/``: pytorch_model.bin or model.safetensors, keys ``roberta.*``, ``embeddingHead.*``,
``norm.*``; ``classifier.*`` / pooler ignored) or from a DPR ``model_dict`` (``question_model.*``,
``ctx_model.*``; utils/dpr_utils.py:74-78).

All arithmetic happens in csrc/ (C ABI ``ance_encoder_create`` / ``ance_encode_*``); this file is
plumbing: weight gathering, buffer ownership, call-shape adaptation.  No CPU fallback.
"""
import ctypes
import json
import os

import numpy as np

from . import _lib

ARCH_ROBERTA, ARCH_BERT = 0, 1

_LAYER_KEYS = (
    "attention.self.query.weight", "attention.self.query.bias",
    "attention.self.key.weight", "attention.self.key.bias",
    "attention.self.value.weight", "attention.self.value.bias",
    "attention.output.dense.weight", "attention.output.dense.bias",
    "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
    "intermediate.dense.weight", "intermediate.dense.bias",
    "output.dense.weight", "output.dense.bias",
    "output.LayerNorm.weight", "output.LayerNorm.bias",
)


def weight_names(prefix, n_layers, has_head):
    """State-dict keys in the order include/ance_amd.h prescribes."""
    e = prefix + "embeddings."
    names = [e + "word_embeddings.weight", e + "position_embeddings.weight", e + "token_type_embeddings.weight",
             e + "LayerNorm.weight", e + "LayerNorm.bias"]
    for i in range(n_layers):
        p = "%sencoder.layer.%d." % (prefix, i)
        names += [p + k for k in _LAYER_KEYS]
    if has_head:
        names += ["embeddingHead.weight", "embeddingHead.bias", "norm.weight", "norm.bias"]
    return names


def count_layers(state_dict, prefix):
    n = 0
    while ("%sencoder.layer.%d.output.dense.weight" % (prefix, n)) in state_dict:
        n += 1
    return n


def precision_from_env(env=None):
    """The mode a handle created with ``precision=None`` (AnceEncoderDesc.precision = ANCE_PRECISION_DEFAULT) would run: the rule of
    csrc/encoder.hip: resolve_precision."""
    env = os.environ if env is None else env
    if env.get("ANCE_ENCODER_PRECISE", "")[:1] == "1":
        return "fp32"
    s = env.get("ANCE_ENCODER_SPLIT", "")[:1]
    if s == "1":
        return "split"
    if s == "0" or env.get("ANCE_ENCODER_FP16", "")[:1] == "1":
        return "fp16"
    return "split"


def _written_through_raw_pointer(t):
    """The library wrote ``t`` through its data pointer: bump the tensor's version counter, which is what
    ``FlatIPIndex`` keys its search image on -- re-encoding into a buffer that was searched rebuilds the image."""
    import torch
    if t.is_inference():
        # tensors created under torch.inference_mode() carry no version counter: their holder must call
        # FlatIPIndex.invalidate() after re-encoding into a buffer that was searched
        return
    torch.autograd.graph.increment_version(t)


class Encoder:
    """One transformer tower + (optional) ANCE head resident in HBM."""

    PRECISIONS = ("split", "fp16", "fp32")

    def __init__(self, state_dict, arch=ARCH_ROBERTA, prefix="roberta.", has_head=True, pad_token_id=None,
                 ln_eps=None, max_seq_len=512, max_tokens=65536, device=None, precision=None):
        """precision: the arithmetic of the handle (AnceEncoderDesc.precision, include/ance_amd.h).  None = whatever the
        environment says -- with nothing set that is "split", the library's default: fp16-pair operands on the fp16 matrix cores,
        fp32-grade like the reference's own fp32 forward (2e-5; the mode in which the refresh reproduces the reference's negative
        ids).  "fp16" = the fast mode (fp16 MFMA operands, 3e-3 on the embeddings, ~2.2 x the throughput), "fp32" = fp32 operands
        (the audit path, ~4.4 x slower than split).
        max_tokens: token capacity of one micro-batch; the activation workspace scales with it -- split mode: 2.8 GB per 65,536
        tokens and lane, two lanes (the refresh drivers and bench.py pass 131,072: +1.3 % throughput for 11 GB per tower)."""
        import torch
        L = _lib.lib()
        if precision is not None and precision not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
        self._create(L, torch, state_dict, arch, prefix, has_head, pad_token_id, ln_eps, max_seq_len, max_tokens, device,
                     _lib.PRECISION_CODES[precision])

    def _create(self, L, torch, state_dict, arch, prefix, has_head, pad_token_id, ln_eps, max_seq_len, max_tokens, device, precision_code):
        self.device = torch.device(device if device is not None else "cuda")
        n_layers = count_layers(state_dict, prefix)
        if n_layers == 0:
            raise KeyError("no '%sencoder.layer.*' weights in state dict" % prefix)
        names = weight_names(prefix, n_layers, has_head)
        missing = [k for k in names if k not in state_dict]
        if missing:
            raise KeyError("missing weights: %s ..." % missing[:4])
        word = state_dict[names[0]]
        pos = state_dict[names[1]]
        inter = state_dict["%sencoder.layer.0.intermediate.dense.weight" % prefix].shape[0]
        self.arch = arch
        self.desc = _lib.AnceEncoderDesc(
            arch=arch, n_layers=n_layers, hidden=int(word.shape[1]), n_heads=12, intermediate=int(inter),
            vocab_size=int(word.shape[0]), max_position=int(pos.shape[0]),
            pad_token_id=(1 if arch == ARCH_ROBERTA else 0) if pad_token_id is None else int(pad_token_id),
            ln_eps=(1e-5 if arch == ARCH_ROBERTA else 1e-12) if ln_eps is None else float(ln_eps),
            has_head=1 if has_head else 0, max_seq_len=int(max_seq_len),
            max_tokens=max(512, int(max_tokens) // 256 * 256), precision=precision_code)
        wbytes = L.ance_encoder_weight_bytes(ctypes.byref(self.desc))
        xbytes = L.ance_encoder_workspace_bytes(ctypes.byref(self.desc))
        if wbytes == 0 or xbytes == 0:
            raise _lib.AnceLibraryError("unsupported encoder shape (hidden must be 768, heads 12)")
        self._arena = torch.empty(wbytes, dtype=torch.uint8, device=self.device)
        self._ws = torch.empty(xbytes, dtype=torch.uint8, device=self.device)
        staged = [state_dict[k].detach().to(device=self.device, dtype=torch.float32).contiguous() for k in names]
        ptrs = (ctypes.c_void_p * len(staged))(*[t.data_ptr() for t in staged])
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = L.ance_encoder_create(ctypes.byref(self.desc), ptrs, len(staged), ctypes.c_void_p(self._arena.data_ptr()),
                                       wbytes, ctypes.c_void_p(self._ws.data_ptr()), xbytes, _lib.current_stream_ptr(),
                                       ctypes.byref(handle))
            _lib.check(rc, "ance_encoder_create")
            torch.cuda.current_stream().synchronize()  # fp32 sources may now be released
        del staged
        self._h = handle
        self.precision = _lib.PRECISION_NAMES[L.ance_encoder_precision(handle)]
        self.n_layers = n_layers
        self.out_dim = 768
        # range guard (include/ance_amd.h: ance_encoder_range_faults): the counters are copied to pinned memory behind every encode
        # call and looked at without synchronising when the next call starts; check_range(sync=True) is the blocking form
        self._faults = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._faults_event = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().ance_encoder_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- range guard -------------------------------------------------------------------------------------
    def _enqueue_range_read(self):
        import torch
        rc = _lib.lib().ance_encoder_range_faults(self._h, ctypes.c_void_p(self._faults.data_ptr()), 0, _lib.current_stream_ptr())
        _lib.check(rc, "ance_encoder_range_faults")
        ev = torch.cuda.Event()
        ev.record()
        self._faults_event = ev

    def check_range(self, sync=False):
        """Raises ``AnceRangeError`` when an encode call whose counters have arrived (``sync``: of every call so far) violated the
        split mode's precondition -- a pre-LayerNorm value, Q | K | V entry or GELU output above 65,504, where the reference's fp32
        has no limit -- or produced NaN rows.  Sticky: the handle keeps failing; build one with ``precision="fp32"``
        (``--encoder_precision fp32``) for such a checkpoint."""
        ev = self._faults_event
        if ev is None:
            return
        if sync:
            ev.synchronize()
        elif not ev.query():
            return
        over, nan_rows = int(self._faults[0]), int(self._faults[1])
        if over or nan_rows:
            raise _lib.AnceRangeError(
                "encoder range guard (%s mode): %d threads stored values above 65,504 in magnitude (the hi half of an fp16 pair "
                "overflows there; the reference's fp32 arithmetic has no such limit), %d output rows are NaN -- these embeddings "
                "are not fp32-grade: run this checkpoint with --encoder_precision fp32 (Encoder(precision=\"fp32\"))"
                % (self.precision, over, nan_rows))

    # -- raw-record path: what the refresh job uses ------------------------------------------------
    def encode_records(self, records, n_chunks=1, h_lens=None, out=None):
        """records: CUDA uint8 [n, 4+4L] (or int32 [n, 1+L]) rows of the tokenised cache.
        Returns CUDA fp32 [n * n_chunks, 768], row = record * n_chunks + chunk."""
        import torch
        if records.dtype == torch.uint8:
            n, rb = records.shape
            Ltok = (rb - 4) // 4
        else:
            records = _lib.require_cuda_tensor(records, torch.int32, "records")
            n, Ltok = records.shape[0], records.shape[1] - 1
        if not records.is_cuda or not records.is_contiguous():
            raise _lib.AnceLibraryError("records must be a contiguous CUDA tensor")
        if out is None:
            out = torch.empty((n * n_chunks, self.out_dim), dtype=torch.float32, device=self.device)
        hl = None
        if h_lens is not None:
            h_lens = np.ascontiguousarray(h_lens, dtype=np.int32)
            assert h_lens.shape[0] == n
            hl = h_lens.ctypes.data_as(ctypes.c_void_p)
        self.check_range()
        with torch.cuda.device(self.device):
            rc = _lib.lib().ance_encode_records(self._h, ctypes.c_void_p(records.data_ptr()), hl, n, Ltok, n_chunks,
                                                ctypes.c_void_p(out.data_ptr()), _lib.current_stream_ptr())
            _lib.check(rc, "ance_encode_records")
            self._enqueue_range_read()
        _written_through_raw_pointer(out)
        return out

    def encode_ids(self, ids, lens, n_chunks=1, h_lens=None, out=None):
        """ids: CUDA int32 [n, L]; lens: CUDA int32 [n]."""
        import torch
        ids = _lib.require_cuda_tensor(ids, torch.int32, "ids")
        lens = _lib.require_cuda_tensor(lens, torch.int32, "lens")
        n, Ltok = ids.shape
        if out is None:
            out = torch.empty((n * n_chunks, self.out_dim), dtype=torch.float32, device=self.device)
        hl = None
        if h_lens is not None:
            h_lens = np.ascontiguousarray(h_lens, dtype=np.int32)
            hl = h_lens.ctypes.data_as(ctypes.c_void_p)
        self.check_range()
        with torch.cuda.device(self.device):
            rc = _lib.lib().ance_encode_ids(self._h, ctypes.c_void_p(ids.data_ptr()), Ltok, ctypes.c_void_p(lens.data_ptr()),
                                            hl, n, Ltok, n_chunks, ctypes.c_void_p(out.data_ptr()),
                                            _lib.current_stream_ptr())
            _lib.check(rc, "ance_encode_ids")
            self._enqueue_range_read()
        _written_through_raw_pointer(out)
        return out

    # -- tensor path (seam B4) ---------------------------------------------------------------------
    def embed(self, input_ids, attention_mask, n_chunks=1):
        """input_ids [B, L] (any int dtype), attention_mask [B, L] of the reference's form
        1 x len, 0 x pad (data/msmarco_data.py:282): only its row sums are used."""
        import torch
        ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
        lens = attention_mask.to(self.device).to(torch.int32).sum(dim=1).to(torch.int32).contiguous()
        return self.encode_ids(ids, lens, n_chunks=n_chunks)


class AnceModel:
    """Drop-in for the object ``load_model`` returns in the reference (an ``nn.Module`` under DDP):
    exposes ``.module.query_emb/body_emb`` and ``.eval()`` (drivers/run_ann_data_gen.py:158,176-178)."""

    def __init__(self, model_type, query_encoder, body_encoder=None, chunks=1):
        self.model_type = model_type
        self.q = query_encoder
        self.b = body_encoder if body_encoder is not None else query_encoder
        self.chunks = chunks
        self.module = self

    def eval(self):
        return self

    def query_emb(self, input_ids, attention_mask):
        return self.q.embed(input_ids, attention_mask)

    def forward(self, query_ids, attention_mask_q, input_ids_a=None, attention_mask_a=None, input_ids_b=None,
                attention_mask_b=None, is_query=True):
        """NLL.forward / NLL_MultiChunk.forward (model/models.py:57-81, 84-134), forward only: with one input the embedding
        (query or body tower); with a triplet ``(loss.mean(),)`` -- three encodes on the HIP encoder and the fused dot /
        max-over-chunks / 2-way log-softmax kernel ``ance_nll_forward``.  ``self.last_logits`` / ``self.last_loss_rows`` keep the
        per-triplet values of the last call (CUDA tensors)."""
        import torch
        if self.model_type == "dpr":
            # BiEncoder.forward (model/models.py:260-271) has no is_query: with no second passage it returns BOTH towers' output
            if input_ids_b is None:
                return (self.query_emb(query_ids, attention_mask_q), self.body_emb(input_ids_a, attention_mask_a))
        elif input_ids_b is None and is_query:
            return self.query_emb(query_ids, attention_mask_q)
        elif input_ids_b is None:
            return self.body_emb(query_ids, attention_mask_q)
        q = self.query_emb(query_ids, attention_mask_q).contiguous()
        a = self.body_emb(input_ids_a, attention_mask_a).contiguous()
        b = self.body_emb(input_ids_b, attention_mask_b).contiguous()
        n, d, chunks = q.shape[0], q.shape[1], self.chunks
        if chunks > 1 and (input_ids_a.shape[1] // 512 != chunks or input_ids_b.shape[1] // 512 != chunks):
            # the reference derives chunk_factor from the inputs (full_length // base_len, model/models.py:103-104)
            raise ValueError("rdot_nll_multi_chunk: inputs of %d / %d tokens do not give the %d chunks this model was loaded for"
                             % (input_ids_a.shape[1], input_ids_b.shape[1], chunks))
        ma = mb = None
        if chunks > 1:
            ma = attention_mask_a.to(q.device).reshape(n, chunks, -1)[:, :, 0].to(torch.float32).contiguous()
            mb = attention_mask_b.to(q.device).reshape(n, chunks, -1)[:, :, 0].to(torch.float32).contiguous()
        logits = torch.empty((n, 2), dtype=torch.float32, device=q.device)
        rows = torch.empty((n,), dtype=torch.float32, device=q.device)
        mean = torch.empty((1,), dtype=torch.float32, device=q.device)
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        with torch.cuda.device(q.device):
            rc = _lib.lib().ance_nll_forward(P(q), P(a), P(b), P(ma), P(mb), n, d, chunks, P(logits), P(rows), P(mean),
                                             _lib.current_stream_ptr())
        _lib.check(rc, "ance_nll_forward")
        self.last_logits, self.last_loss_rows = logits, rows
        return (mean[0],)

    __call__ = forward

    def body_emb(self, input_ids, attention_mask):
        if self.chunks == 1:
            return self.b.embed(input_ids, attention_mask)
        B = input_ids.shape[0]
        return self.b.embed(input_ids, attention_mask, n_chunks=self.chunks).reshape(B, self.chunks, -1)


# ----------------------------------------------------------------------------- checkpoint loading
def load_hf_state_dict(ckpt_dir):
    """``checkpoint-N/`` as written by the trainer's save_pretrained (drivers/run_ann.py:307-334)."""
    import torch
    st = os.path.join(ckpt_dir, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(ckpt_dir, "pytorch_model.bin")
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError("no model.safetensors / pytorch_model.bin in %s" % ckpt_dir)


def load_model(model_type, checkpoint_path, max_seq_length=128, max_tokens=65536, device=None, precision=None):
    """Registry of model/models.py:299-322 restricted to the encoders on the path.  precision: see ``Encoder``."""
    model_type = model_type.lower()
    if model_type in ("rdot_nll", "rdot_nll_multi_chunk"):
        sd = load_hf_state_dict(checkpoint_path)
        chunks = 1
        seq = max_seq_length
        if model_type == "rdot_nll_multi_chunk":
            chunks = max(1, max_seq_length // 512)  # base_len = 512 (model/models.py:163)
            seq = 512
        enc = Encoder(sd, ARCH_ROBERTA, "roberta.", True, max_seq_len=min(seq, 512), max_tokens=max_tokens, device=device,
                      precision=precision)
        return AnceModel(model_type, enc, chunks=chunks)
    if model_type == "dpr":
        import torch
        ck = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        sd = ck["model_dict"] if isinstance(ck, dict) and "model_dict" in ck else getattr(ck, "model_dict", ck)
        q = Encoder(sd, ARCH_BERT, "question_model.", False, max_seq_len=min(max_seq_length, 512), max_tokens=max_tokens,
                    device=device, precision=precision)
        c = Encoder(sd, ARCH_BERT, "ctx_model.", False, max_seq_len=min(max_seq_length, 512), max_tokens=max_tokens,
                    device=device, precision=precision)
        return AnceModel(model_type, q, c)
    raise ValueError("model_type %r is not on the MI355X path (supported: rdot_nll, rdot_nll_multi_chunk, dpr)" % model_type)


def read_config(ckpt_dir):
    p = os.path.join(ckpt_dir, "config.json")
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return {}
