#!/bin/bash
# Experiment: corpus-split count vs L2 reuse in ip_topk_fast_kernel; hipBLASLt yardstick for the GEMM shapes.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for S in 1 2 4 8 16 32; do
  echo "== splits $S"; ANCE_FAST_SPLITS=$S tools/abi_probe search 8841823 32768 200 2 | tail -1
done
echo "== hipBLASLt yardstick (torch.matmul fp16)"
python - <<'PY'
import torch, time
def t(M,N,K,reps=20):
    a=torch.randn(M,K,device='cuda',dtype=torch.float16); b=torch.randn(N,K,device='cuda',dtype=torch.float16)
    for _ in range(3): c=a@b.t()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c=a@b.t()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/reps
    print("matmul NT M=%d N=%d K=%d  %.3f ms  %.0f TFLOP/s"%(M,N,K,ms,2*M*N*K/ms/1e9))
for M,N,K in [(65536,3072,768),(65536,768,3072),(65536,1536,768),(65536,768,768),(32768,32768,768),(8192,8192,8192)]:
    t(M,N,K)
PY
for epi in 0 1 2; do :; done
echo "== our gemm, same shapes"
tools/abi_probe gemm 0 1 65536 3072 768 20 | tail -1
tools/abi_probe gemm 0 2 65536 768 3072 20 | tail -1
tools/abi_probe gemm 0 0 65536 1536 768 20 | tail -1
