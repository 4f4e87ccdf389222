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


def _pair_layout(W):
    """(hi_cols, lo_cols, lo_scale) of a W-wide pair row, from the library itself (product build: 32-column blocks [hi | lo], lo
    unscaled; the round-4 A/B build: [hi (W) | lo' (W)], lo' = (v - hi) 2^11)."""
    from ance_amd import _lib
    L = _lib.lib()
    hi, lo, sc = ctypes.c_int(), ctypes.c_int(), ctypes.c_float()
    hc, lc = [], []
    for n in range(W):
        L.ance_pair_layout(n, W, ctypes.byref(hi), ctypes.byref(lo), ctypes.byref(sc))
        hc.append(hi.value)
        lc.append(lo.value)
    return torch.tensor(hc, device="cuda"), torch.tensor(lc, device="cuda"), float(sc.value)


def _pair(v):
    """fp32 [R, W] -> (hi, lo) halves of the pair, lo = fp16((v - hi) lo_scale)"""
    _, _, sc = _pair_layout(32)
    hi = v.half()
    lo = ((v - hi.float()) * sc).half()
    return hi, lo


def _pair_rows(hi, lo):
    """(hi, lo) [R, W] -> pair rows [R, 2 W] fp16 in the library's layout"""
    R, W = hi.shape
    hc, lc, _ = _pair_layout(W)
    out = torch.zeros((R, 2 * W), dtype=torch.float16, device=hi.device)
    out[:, hc] = hi
    out[:, lc] = lo
    return out.contiguous()


def _unpair(p, n):
    hc, lc, sc = _pair_layout(n)
    return p[:, hc].double() + p[:, lc].double() / sc


def _run_split(epi, M, N, K, seed=0, zero_a_lo=False, zero_b_lo=False, wscale=1.0):
    """The split GEMM (three fp16 MFMA passes over pair operands) through its test hook with one of its three epilogues,
    against the same expression in fp64.  Returns (got, ref, scale): scale = the magnitude rounding errors are relative to."""
    from ance_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn((M, K), generator=g, device="cuda")
    b = torch.randn((N, K), generator=g, device="cuda") * 0.02
    _, _, lo_sc = _pair_layout(32)
    ah, al = _pair(a)
    bh, bl = _pair(b * wscale)  # the weight operand is stored times a power of two (undone by *d_wscale_inv in the epilogue)
    if zero_a_lo:
        al.zero_()
    if zero_b_lo:
        bl.zero_()
    ap = _pair_rows(ah, al)
    bp = _pair_rows(bh, bl)
    winv = torch.tensor([1.0 / wscale], dtype=torch.float32, device="cuda")
    bias = torch.randn(N, generator=g, device="cuda")
    vec1 = torch.randn(N, generator=g, device="cuda")
    vec2 = torch.randn(N, generator=g, device="cuda")
    # slice statistics: mean m_j per 64-column slice, M2 = 64 s^2 -> row mean mu, variance s^2 + var(m_j)
    part = torch.empty((M, 12, 2), device="cuda")
    part[:, :, 0] = torch.randn((M, 12), generator=g, device="cuda") * 0.1
    part[:, :, 1] = 64.0 * (0.5 + torch.rand((M, 12), generator=g, device="cuda"))
    eps = 1e-5
    res = torch.randn((M, N), generator=g, device="cuda")
    rp = _pair_rows(*_pair(res))
    out = torch.empty((M, N), dtype=torch.float32, device="cuda") if epi == 8 else torch.zeros((M, 2 * N), dtype=torch.float16, device="cuda")
    part_out = torch.zeros((M, N // 64, 2), device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = L.ance_debug_gemm_split(epi, P(ap), P(bp), M, N, K, P(bias), P(vec1), P(vec2), P(part), eps, P(rp), P(out), P(part_out),
                                 P(winv), _lib.current_stream_ptr())
    _lib.check(rc, "ance_debug_gemm_split")
    A = ah.double() + al.double() / lo_sc
    B = (bh.double() + bl.double() / lo_sc) / wscale
    acc = A @ B.t()
    dropped = ((al.double() / lo_sc).abs() @ (bl.double() / lo_sc / wscale).abs().t())  # the lo x lo term the kernel leaves out
    m = part[:, :, 0].double()
    mu = m.mean(1)
    var = (part[:, :, 1].double() + 64.0 * (m - mu[:, None]) ** 2).sum(1) / 768.0
    r = 1.0 / torch.sqrt(var + eps)
    mag = A.abs() @ B.abs().t()
    if epi in (8, 9):
        ref = r[:, None] * (acc - mu[:, None] * vec1.double()[None, :]) + bias.double()[None, :]
        scale = (mag + dropped + (mu[:, None] * vec1.double()[None, :]).abs()) * r[:, None] + bias.double().abs()[None, :]
        if epi == 9:
            ref = torch.nn.functional.gelu(ref)
            scale = scale + ref.abs()
        got = out.double() if epi == 8 else _unpair(out, N)
        return got, ref, scale, None, None
    R = _unpair(rp, N)
    ln = (R - mu[:, None]) * r[:, None] * vec1.double()[None, :] + vec2.double()[None, :]
    ref = acc + bias.double()[None, :] + ln
    scale = mag + dropped + bias.double().abs()[None, :] + ((R.abs() + mu.abs()[:, None]) * r[:, None] * vec1.double().abs()[None, :]) + \
        vec2.double().abs()[None, :]
    s = ref.reshape(M, N // 64, 64)
    want_part = torch.stack([s.mean(-1), ((s - s.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1)
    return _unpair(out, N), ref, scale, part_out.double(), want_part


@pytest.mark.parametrize("shape", [(256, 256, 128), (512, 2304, 768), (768, 768, 3072), (1024, 3072, 768)])
@pytest.mark.parametrize("variant", ["full", "a_lo_zero", "b_lo_zero", "both_lo_zero"])
def test_split_gemm_is_fp32_grade(shape, variant):
    """|error| relative to the magnitudes summed: every partial product is exact in fp32 (11 x 11 bits), so what is left is
    fp32 accumulation (measured unbiased, std 1.4 ulp at K = 768: scripts/mfma_rounding_probe.py) and the dropped lo x lo term
    (2^-22, part of the scale).  The variants zero one or both lo halves: a wrong K-tile of one correction segment would show
    up in exactly one of them."""
    M, N, K = shape
    got, ref, scale, _, _ = _run_split(8, M, N, K, seed=7, zero_a_lo=variant in ("a_lo_zero", "both_lo_zero"),
                                       zero_b_lo=variant in ("b_lo_zero", "both_lo_zero"), wscale=2.0 ** 17)
    rel = (got - ref).abs() / scale
    assert float(rel.max()) <= 5e-7, (variant, shape, float(rel.max()), torch.nonzero(rel > 5e-7)[:3].tolist())


@pytest.mark.parametrize("epi,shape", [(9, (512, 3072, 768)), (9, (256, 256, 128)), (10, (512, 768, 768)), (10, (768, 768, 3072)),
                                       (10, (256, 768, 128))])
def test_split_gemm_pair_epilogues(epi, shape):
    """The two pair-writing epilogues on EVERY row and column of several tiles: exact-erf GELU -> (hi, lo') pair (epi 9) and
    bias + LayerNorm(residual pair) -> pair + slice statistics (epi 10).  A pair carries 22 bits: 2^-22 of the value on top of
    the GEMM's own 5e-7."""
    M, N, K = shape
    got, ref, scale, part, want_part = _run_split(epi, M, N, K, seed=11, wscale=2.0 ** 17 if shape[0] != 256 else 1.0)
    rel = (got - ref).abs() / scale
    assert float(rel.max()) <= 1e-6, (epi, shape, float(rel.max()), torch.nonzero(rel > 1e-6)[:5].tolist())
    if part is not None:
        dm = (part[..., 0] - want_part[..., 0]).abs()
        dq = (part[..., 1] - want_part[..., 1]).abs() / want_part[..., 1]
        assert float(dm.max()) <= 2e-6 and float(dq.max()) <= 2e-5, (float(dm.max()), float(dq.max()), torch.nonzero(dq > 2e-5)[:5].tolist())


def test_mfma_keeps_f16_subnormals():
    """The split mode's pair halves are unscaled: lo = fp16(v - hi) of an element below 2^-3 is an fp16 SUBNORMAL.  The scheme
    needs the conversion to produce it and v_mfma_f32_32x32x16_f16 to multiply it (gfx90a flushed them; gfx942 / gfx950 do not).
    A with hi = 0 and lo = j 2^-24 (subnormal for j < 1024), B = 1: every output must be the exact sum of the lo halves -- a
    flushing matrix core would return 0."""
    from ance_amd import _lib
    L = _lib.lib()
    _, _, lo_sc = _pair_layout(32)
    if lo_sc != 1.0:
        pytest.skip("round-4 A/B build: lo halves are scaled into the normal range")
    M = N = 256
    K = 128
    j = (torch.arange(M * K, device="cuda").reshape(M, K) % 1023 + 1).float()
    al = (j * 2.0 ** -24).half()
    assert torch.equal(al.float(), j * 2.0 ** -24) and float(al.float().max()) < 2.0 ** -14  # exactly representable, all subnormal
    ah = torch.zeros_like(al)
    bh = torch.ones((N, K), dtype=torch.float16, device="cuda")
    bl = torch.zeros_like(bh)
    ap, bp = _pair_rows(ah, al), _pair_rows(bh, bl)
    bias = torch.zeros(N, device="cuda")
    csum = torch.zeros(N, device="cuda")
    part = torch.zeros((M, 12, 2), device="cuda")
    part[:, :, 1] = 64.0  # mean 0, variance 1 -> r = 1 / sqrt(1 + eps)
    out = torch.empty((M, N), device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = L.ance_debug_gemm_split(8, P(ap), P(bp), M, N, K, P(bias), P(csum), None, P(part), 0.0, None, P(out), None, None,
                                 _lib.current_stream_ptr())
    _lib.check(rc, "ance_debug_gemm_split")
    want = al.double().sum(1)  # exact: 128 terms of at most 10 bits at one binade spacing
    got = out.double()
    assert float(want.min()) > 0
    assert torch.equal(got, want[:, None].expand(M, N).float().double()), (float(got.abs().max()), float(want.max()))


def _split_raw(epi, M, N, K, seed, stream, monkeypatch):
    """One launch of the split GEMM through its test hook, with the persistent streaming kernel (the product's) or the
    launch-per-tile kernel (ANCE_GEMM_STREAM=0); returns the raw output buffers."""
    from ance_amd import _lib
    monkeypatch.setenv("ANCE_GEMM_STREAM", "1" if stream else "0")
    _lib.reload_env()
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(seed)
    ap = _pair_rows(*_pair(torch.randn((M, K), generator=g, device="cuda")))
    bp = _pair_rows(*_pair(torch.randn((N, K), generator=g, device="cuda") * 0.02 * 2.0 ** 17))
    winv = torch.tensor([2.0 ** -17], dtype=torch.float32, device="cuda")
    bias, vec1, vec2 = (torch.randn(N, generator=g, device="cuda") for _ in range(3))
    part = torch.empty((M, 12, 2), device="cuda")
    part[:, :, 0] = torch.randn((M, 12), generator=g, device="cuda") * 0.1
    part[:, :, 1] = 64.0 * (0.5 + torch.rand((M, 12), generator=g, device="cuda"))
    rp = _pair_rows(*_pair(torch.randn((M, N), generator=g, device="cuda")))
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda") if epi == 8 else torch.zeros((M, 2 * N), dtype=torch.float16, device="cuda")
    part_out = torch.zeros((M, max(N // 64, 1), 2), device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = L.ance_debug_gemm_split(epi, P(ap), P(bp), M, N, K, P(bias), P(vec1), P(vec2), P(part), 1e-5, P(rp), P(out), P(part_out),
                                 P(winv), _lib.current_stream_ptr())
    _lib.check(rc, "ance_debug_gemm_split")
    torch.cuda.synchronize()
    return out, part_out


@pytest.mark.parametrize("epi,shape", [(8, (256, 256, 128)), (8, (8192, 2304, 768)), (8, (16640, 2304, 768)), (9, (16384, 3072, 768)),
                                       (9, (512, 3072, 768)), (9, (33024, 3072, 768)), (8, (1024, 768, 3072))])
def test_streaming_split_gemm_equals_the_launch_per_tile_kernel(epi, shape, monkeypatch):
    """Round 6: the persistent split GEMM of the QKV and FFN1 projections (one workgroup per CU walking its output tiles, the next
    tile's first K-tiles staged under the current epilogue, 32 x 32 epilogue passes in what LDS is left) against the
    launch-per-tile kernel (ANCE_GEMM_STREAM=0; what the two RESLN GEMMs still run): same K order, same epilogue arithmetic --
    every output bit identical, on shapes with one tile per workgroup, with 2-6 tiles per workgroup, with a ragged last round,
    and with K = 128 / 768 / 3,072 (4 / 24 / 96 K-tiles)."""
    from ance_amd import _lib
    M, N, K = shape
    try:
        a, pa = _split_raw(epi, M, N, K, 5, True, monkeypatch)
        b, pb = _split_raw(epi, M, N, K, 5, False, monkeypatch)
    finally:
        monkeypatch.delenv("ANCE_GEMM_STREAM", raising=False)
        _lib.reload_env()
    assert bool(torch.isfinite(a.float()).all())
    assert torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a.view(torch.int32),
                       b.view(torch.int16) if b.dtype == torch.float16 else b.view(torch.int32)), (epi, shape)
