#!/bin/bash
# Round 3: time decomposition of attention_reg_kernel with the measurement library's ablations (results are WRONG in
# these runs by construction; only the attention launch time is read).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3abl
mkdir -p $O
export TMPDIR=/tmp
export ANCE_AMD_LIB=$PWD/ance_amd/libance_amd_measure.so
for a in 0 1 2 3 4 7 8 12 15; do
  ANCE_ATTN_ABLATE=$a ANCE_ENCODER_STREAMS=1 timeout 300 python bench.py --skip-search --no-cpu-baseline --steps 3 --warmup 1 > $O/abl_$a.json 2> $O/abl_$a.err
  python - $O/abl_$a.json $a <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    bk = d["roofline"]["by_kernel"]
    print("ablate=%s attention %.1f us/launch  (qk %.1f ffn1 %.1f)" % (sys.argv[2], 1e3 * bk["attention"]["ms_per_launch"], 1e3 * bk["gemm_qk"]["ms_per_launch"], 1e3 * bk["gemm_ffn1"]["ms_per_launch"]))
except Exception as e:
    print("ablate=%s (no line) %r" % (sys.argv[2], e))
PY
done
