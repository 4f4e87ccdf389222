"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/ance_amd.h
declares; argument validation (which never touches a device) behaves as documented."""
import ctypes
import os
import re

import pytest

from ance_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "ance_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(ance_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_and_binding_agree():
    declared = _declared_functions()
    assert declared, "no functions parsed from the header"
    assert sorted(_lib.SYMBOLS.keys()) == declared


def test_library_loads_and_exports_everything():
    L = _lib.lib()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(raw, name), name
    assert L.ance_abi_version() == _lib.ABI_VERSION == 5
    assert L.ance_last_error() is not None


def test_workspace_queries_are_pure():
    L = _lib.lib()
    assert L.ance_ip_topk_workspace_bytes(10000, 1000, 768, 200) > 0
    assert L.ance_ip_topk_workspace_bytes(10000, 1000, 768, 0) == 0          # k out of range
    assert L.ance_ip_topk_workspace_bytes(10000, 1000, 768, 5000) == 0
    assert L.ance_ip_topk_workspace_bytes(1 << 33, 10, 768, 10) == 0         # n >= 2^32
    d = _lib.AnceEncoderDesc(arch=0, n_layers=12, hidden=768, n_heads=12, intermediate=3072, vocab_size=50265,
                             max_position=514, pad_token_id=1, ln_eps=1e-5, has_head=1, max_seq_len=512, max_tokens=32768)
    assert L.ance_encoder_weight_bytes(ctypes.byref(d)) > 300e6
    assert L.ance_encoder_workspace_bytes(ctypes.byref(d)) > 100e6
    d.hidden = 1024
    assert L.ance_encoder_weight_bytes(ctypes.byref(d)) == 0
    assert abs(L.ance_encoder_flops_per_sequence(128) - 22.35e9) / 22.35e9 < 1e-3
    assert abs(L.ance_encoder_flops_per_sequence(512) - 96.64e9) / 96.64e9 < 1e-3


def test_invalid_arguments_are_rejected_before_any_launch():
    L = _lib.lib()
    # null pointers / bad k: must return ANCE_E_INVALID without touching a device
    rc = L.ance_ip_topk(None, 10, 0, None, 4, 768, 0, None, None, None, 0, None)
    assert rc == -1 and b"invalid" in L.ance_last_error()
    rc = L.ance_topk_merge(None, None, 2, 4, 10, None, None, None, 0, None)
    assert rc == -1
    rc = L.ance_encode_records(None, None, None, 4, 128, 1, None, None)
    assert rc == -1
    rc = L.ance_nll_forward(None, None, None, None, None, 4, 768, 1, None, None, None, None)
    assert rc == -1 and b"nll" in L.ance_last_error()
    rc = L.ance_debug_gemm_split(8, None, None, 256, 256, 128, None, None, None, None, 1e-5, None, None, None, None, None)
    assert rc == -1
    rc = L.ance_ip_topk_scan(None, 10, 0, None, 4, 768, 0, None, None, None, 0, None)
    assert rc == -1 and b"invalid" in L.ance_last_error()
    assert L.ance_ip_topk_scan_workspace_bytes(10000, 1000, 768, 200) > 0
    assert L.ance_ip_topk_scan_workspace_bytes(10000, 1000, 768, 0) == 0
    assert L.ance_encoder_range_faults(None, None, 0, None) == -1
    assert L.ance_encoder_precision(None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.AnceLibraryError):
        _lib.lib()
