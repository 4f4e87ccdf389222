"""How does v_mfma_f32_32x32x16_f16 round its accumulation?  The encoder's GEMM kernel through the C-ABI test hook with the
fp32 (EPI_RES32, bias = residual = 0) epilogue, so the output IS the accumulator, against the exact result (fp16 products are
exact in fp64; the fp64 sum of <= 3072 of them is exact to 2^-53) -- signed error in units of the result's fp32 ulp:
  RNE per accumulate step  -> mean ~ 0, std ~ 0.3 sqrt(steps)
  truncation (toward zero / toward -inf) -> mean ~ -0.5 steps x (average partial sum / result) for same-sign terms
Cases: zero-mean operands (signs random), all-positive operands (partial sums grow monotonically: the LayerNorm-fold
situation with a large row mean), and a SPLIT-like chain (K = 3 x 768)."""
import ctypes
import json
import sys

import torch

sys.path.insert(0, ".")
from ance_amd import _lib  # noqa: E402

L = _lib.lib()


def run(a, b):
    M, K = a.shape
    N = b.shape[0]
    bias = torch.zeros(N, device="cuda")
    res = torch.zeros((M, N), device="cuda")
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    rc = L.ance_debug_gemm(0, 2, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), M, N, K,
                           ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(res.data_ptr()),
                           _lib.current_stream_ptr())
    _lib.check(rc, "ance_debug_gemm")
    torch.cuda.synchronize()
    return out


def ulp32(x):
    x = x.abs().clamp_min(1e-30).float()
    return torch.ldexp(torch.ones_like(x), (torch.frexp(x)[1] - 24).int()).double()


out = {}
g = torch.Generator(device="cuda").manual_seed(0)
for name, K, pos in (("zero_mean_K768", 768, False), ("positive_K768", 768, True), ("positive_K3072", 3072, True),
                     ("zero_mean_K3072", 3072, False)):
    a = torch.randn((256, K), generator=g, device="cuda")
    b = torch.randn((256, K), generator=g, device="cuda") * 0.02
    if pos:
        a, b = a.abs() + 1.0, b.abs() + 0.01
    a, b = a.half(), b.half()
    got = run(a, b).double()
    exact = a.double() @ b.double().t()
    rne = exact.float().double()  # the correctly rounded result
    err = (got - exact) / ulp32(exact)
    # a torch fp32 GEMM (fp32 accumulation in some blocked order, RNE) for scale
    t32 = (a.float() @ b.float().t()).double()
    err_t = (t32 - exact) / ulp32(exact)
    out[name] = dict(steps=K // 16, mean_ulp=float(err.mean()), std_ulp=float(err.std()), max_abs_ulp=float(err.abs().max()),
                     frac_equal_rne=float((got == rne).double().mean()), mean_ulp_signed_by_result=float((err * exact.sign()).mean()),
                     torch_fp32_mean_ulp=float(err_t.mean()), torch_fp32_std_ulp=float(err_t.std()))
    print(name, out[name])
# single-step anatomy: C = 2^24 (ulp 2), sixteen products that sum to an odd multiple of 0.5 ulp etc.
json.dump(out, open("gpurun_out/mfma_rounding_probe.json", "w"), indent=1)
