"""ctypes binding of libance_amd.so (the C ABI declared in include/ance_amd.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, every
entry point raises.  ``build()`` compiles it in-tree with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# ANCE_AMD_LIB: another build of the same ABI (the measurement library of `make -C ance_amd/csrc measure`)
LIB_PATH = os.environ.get("ANCE_AMD_LIB") or os.path.join(_HERE, "libance_amd.so")
CSRC = os.path.join(_HERE, "csrc")

ANCE_OK = 0
ABI_VERSION = 5

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f32p = ctypes.POINTER(ctypes.c_float)


class AnceEncoderDesc(ctypes.Structure):
    _fields_ = [
        ("arch", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("hidden", ctypes.c_int32),
        ("n_heads", ctypes.c_int32), ("intermediate", ctypes.c_int32), ("vocab_size", ctypes.c_int32),
        ("max_position", ctypes.c_int32), ("pad_token_id", ctypes.c_int32), ("ln_eps", ctypes.c_float),
        ("has_head", ctypes.c_int32), ("max_seq_len", ctypes.c_int32), ("max_tokens", ctypes.c_int32),
        ("precision", ctypes.c_int32),
    ]


# AnceEncoderDesc.precision (include/ance_amd.h: ANCE_PRECISION_*)
PRECISION_CODES = {None: 0, "split": 1, "fp16": 2, "fp32": 3}
PRECISION_NAMES = {1: "split", 2: "fp16", 3: "fp32"}


# name -> (restype, argtypes): every symbol include/ance_amd.h declares
SYMBOLS = {
    "ance_abi_version": (ctypes.c_int, []),
    "ance_last_error": (ctypes.c_char_p, []),
    "ance_profile_enable": (None, [ctypes.c_int]),
    "ance_profile_read": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]),
    "ance_ip_topk_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "ance_ip_topk": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_size_t, ctypes.c_void_p]),
    "ance_ip_topk_scan_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "ance_ip_topk_scan": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_size_t, ctypes.c_void_p]),
    "ance_ip_index_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int]),
    "ance_ip_index_build": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p]),
    "ance_ip_topk_indexed_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "ance_ip_topk_indexed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "ance_topk_merge_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int64, ctypes.c_int]),
    "ance_topk_merge": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_void_p]),
    "ance_ip_score_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ance_encoder_weight_bytes": (ctypes.c_size_t, [ctypes.POINTER(AnceEncoderDesc)]),
    "ance_encoder_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(AnceEncoderDesc)]),
    "ance_encoder_create": (ctypes.c_int, [ctypes.POINTER(AnceEncoderDesc), ctypes.POINTER(ctypes.c_void_p),
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                           ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "ance_encoder_destroy": (None, [ctypes.c_void_p]),
    "ance_encoder_precision": (ctypes.c_int, [ctypes.c_void_p]),
    "ance_encoder_range_faults": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "ance_encode_records": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "ance_encode_ids": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    "ance_encoder_flops_per_sequence": (ctypes.c_double, [ctypes.c_int]),
    "ance_host_py_shuffle": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "ance_host_select_negatives": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "ance_host_write_ann_training": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    "ance_reload_env": (None, []),
    "ance_debug_search_stamps": (None, [ctypes.c_void_p]),
    "ance_debug_gemm": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    "ance_search_bad_image_calls": (ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong)]),
    "ance_nll_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p]),
    "ance_debug_gemm_split": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ance_pair_layout": (None, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                ctypes.POINTER(ctypes.c_float)]),
}

_lib = None


class AnceLibraryError(RuntimeError):
    pass


class AnceRangeError(AnceLibraryError):
    """The split (default) arithmetic met a value outside the fp16 range -- its stated precondition -- or produced NaN rows."""


def build(verbose=False, force=False):
    """Compile every HIP source for gfx950 into ance_amd/libance_amd.so (hipcc cross-compiles
    without a GPU).  force: rebuild every object (``make -B``) -- the tree ships its .o files to the GPU box, and a
    "does it build" check must not pass on stale objects."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise AnceLibraryError("hipcc build of libance_amd.so failed")
    return LIB_PATH


def lib():
    """The loaded C-ABI library.  Raises AnceLibraryError (never falls back) when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AnceLibraryError(
            "%s not found: the HIP extension is required (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ance_amd/csrc`); there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  -- loads the HIP runtime torch ships, so both share one runtime
    except Exception:  # pragma: no cover
        pass
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise AnceLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise AnceLibraryError("%s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if L.ance_abi_version() != ABI_VERSION:
        raise AnceLibraryError("ABI version mismatch: library %d, binding %d" % (L.ance_abi_version(), ABI_VERSION))
    _lib = L
    return L


def reload_env():
    """The library reads its ANCE_* knobs once; call this after changing one inside a running process."""
    lib().ance_reload_env()


def check(rc, what):
    if rc != ANCE_OK:
        msg = lib().ance_last_error()
        raise AnceLibraryError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


PROFILE_CATEGORIES = ("plan", "embed_ln", "gemm_qk", "gemm_vt", "attention", "gemm_attn_out", "layernorm",
                      "gemm_ffn1", "gemm_ffn2", "head", "ip_topk_scan", "topk_finalize", "ip_topk_rescore")


def profile_enable(on=True):
    lib().ance_profile_enable(1 if on else 0)


def profile_read():
    """{category: dict(ms=, work=, count=)} accumulated since profile_enable."""
    n = len(PROFILE_CATEGORIES)
    ms = (ctypes.c_double * n)()
    work = (ctypes.c_double * n)()
    cnt = (ctypes.c_longlong * n)()
    lib().ance_profile_read(ms, work, cnt, n)
    return {c: dict(ms=ms[i], work=work[i], count=int(cnt[i])) for i, c in enumerate(PROFILE_CATEGORIES)}


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda_tensor(t, dtype, what):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise AnceLibraryError("%s must be a CUDA(HIP) tensor" % what)
    if t.dtype != dtype:
        raise AnceLibraryError("%s must have dtype %s, got %s" % (what, dtype, t.dtype))
    if not t.is_contiguous():
        raise AnceLibraryError("%s must be contiguous" % what)
    return t
