#!/bin/bash
# how much of the RESLN epilogue is its HBM traffic: stamps with the residual reads / the output writes redirected to cache-resident rows
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
RES_ABLATE=0,1,2,3,0 timeout 600 python scripts/gemm_res_stamps.py 2>&1 | tail -6 | tee gpurun_out/res_ablate.jsonl
