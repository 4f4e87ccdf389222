"""The encoder's fp16 MFMA GEMM (through the C-ABI test hook) against a plain PyTorch fp32 reference of
the same op on the same fp16 inputs, for every fused epilogue.  Needs an MI355X."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(epi, M, N, K, seed=0):
    from ance_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = (torch.randn((M, K), generator=g, device="cuda") * 0.5).half()
    b = (torch.randn((N, K), generator=g, device="cuda") * 0.5).half()
    # asymmetric on purpose: a transposed or row/col-swapped result cannot pass
    a[:, 0] += torch.arange(M, device="cuda").half() * 0.01
    bias = torch.randn(N, generator=g, device="cuda")
    res = torch.randn((M, N), generator=g, device="cuda") if epi == 2 else None
    out = torch.empty((M, N), dtype=torch.float32 if epi == 2 else torch.float16, device="cuda")
    rc = L.ance_debug_gemm(0, epi, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), M, N, K,
                           ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                           ctypes.c_void_p(res.data_ptr()) if res is not None else None, _lib.current_stream_ptr())
    _lib.check(rc, "ance_debug_gemm")
    ref = a.float() @ b.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + res
    return out.float(), ref


@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("shape", [(256, 256, 128), (512, 768, 768), (768, 256, 3072), (1024, 3072, 128), (2304, 1536, 768)])
def test_gemm_matches_fp32_reference(epi, shape):
    M, N, K = shape
    out, ref = _run(epi, M, N, K, seed=epi)
    err = (out - ref).abs()
    # fp16 operands are exact in the reference too; differences = fp32 accumulation order + final
    # rounding to fp16 (epi 0/1): 2^-11 relative plus a small absolute floor
    tol = 2e-3 * ref.abs() + 2e-2 if epi != 2 else 1e-3 * ref.abs() + 2e-3
    bad = err > tol
    assert not bad.any(), "epi %d shape %s: %d bad, max err %.4g at %s" % (
        epi, shape, int(bad.sum()), float(err.max()), torch.nonzero(bad)[:3].tolist())


def test_bad_shapes_are_rejected():
    from ance_amd import _lib
    L = _lib.lib()
    t = torch.zeros(16, device="cuda")
    p = ctypes.c_void_p(t.data_ptr())
    assert L.ance_debug_gemm(0, 0, p, p, 128, 256, 64, p, p, None, _lib.current_stream_ptr()) == -1
    assert L.ance_debug_gemm(0, 5, p, p, 256, 256, 128, p, p, None, _lib.current_stream_ptr()) == -1
    assert L.ance_debug_gemm(0, 0, p, p, 256, 256, 64, p, p, None, _lib.current_stream_ptr()) == -1  # needs >= 2 K-tiles


@pytest.mark.parametrize("epi", [0, 2])
def test_full_chip_shape_repeated(epi):
    """Race screen for the ping-pong main loop (counted vmcnt, LDS-DMA in flight across barriers): the
    encoder's largest shapes (every CU busy, several rounds), five launches each, every element checked."""
    M, N, K = (16384, 3072, 768) if epi == 0 else (16384, 768, 3072)
    for rep in range(5):
        out, ref = _run(epi, M, N, K, seed=100 + rep)
        err = (out - ref).abs()
        tol = 2e-3 * ref.abs() + 2e-2 if epi != 2 else 1e-3 * ref.abs() + 4e-3
        bad = err > tol
        assert not bad.any(), "epi %d rep %d: %d bad, max err %.4g at %s" % (
            epi, rep, int(bad.sum()), float(err.max()), torch.nonzero(bad)[:3].tolist())
