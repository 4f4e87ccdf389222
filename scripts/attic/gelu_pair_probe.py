import sys, ctypes
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_gemm as T
from ance_amd import _lib
L = _lib.lib()
M, N, K = 512, 3072, 768
g = torch.Generator(device="cuda").manual_seed(11)
a = torch.randn((M, K), generator=g, device="cuda"); b = torch.randn((N, K), generator=g, device="cuda") * 0.02
ah, al = T._pair(a); bh, bl = T._pair(b)
ap = torch.cat([ah, al], 1).contiguous(); bp = torch.cat([bh, bl], 1).contiguous()
bias = torch.randn(N, generator=g, device="cuda"); vec1 = torch.randn(N, generator=g, device="cuda"); vec2 = torch.randn(N, generator=g, device="cuda")
part = torch.empty((M, 12, 2), device="cuda")
part[:, :, 0] = torch.randn((M, 12), generator=g, device="cuda") * 0.1
part[:, :, 1] = 64.0 * (0.5 + torch.rand((M, 12), generator=g, device="cuda"))
res = torch.randn((M, N), generator=g, device="cuda"); rp = torch.cat(T._pair(res), 1).contiguous()
P = lambda t: ctypes.c_void_p(t.data_ptr())
o8 = torch.empty((M, N), dtype=torch.float32, device="cuda"); po = torch.zeros((M, N // 64, 2), device="cuda")
o9 = torch.zeros((M, 2 * N), dtype=torch.float16, device="cuda")
L.ance_debug_gemm_split(8, P(ap), P(bp), M, N, K, P(bias), P(vec1), P(vec2), P(part), 1e-5, P(rp), P(o8), P(po), _lib.current_stream_ptr())
L.ance_debug_gemm_split(9, P(ap), P(bp), M, N, K, P(bias), P(vec1), P(vec2), P(part), 1e-5, P(rp), P(o9), P(po), _lib.current_stream_ptr())
torch.cuda.synchronize()
x = o8.double()                      # the GELU's input as the same kernel computes it (epi 8 is verified)
want = torch.nn.functional.gelu(x)
hi, lo = o9[:, :N].double(), o9[:, N:].double()
got = hi + lo / 2048.0
err = (got - want).abs()
idx = torch.nonzero(err > 2e-6)
print("bad elements:", idx.shape[0], "of", M * N, " max err %.3e" % float(err.max()))
for r, c in idx[:25].tolist():
    w = float(want[r, c]); h = float(hi[r, c]); l = float(lo[r, c])
    print("x=% .6f want=% .8f hi=% .8f lo'=% .6f got=% .8f err=%.2e  fp16(want)=% .8f  (want-fp16(want))*2048=% .6f" % (
        float(x[r, c]), w, h, l, h + l / 2048.0, abs(h + l / 2048 - w), float(torch.tensor(w).half()), (w - float(torch.tensor(w).half())) * 2048))
