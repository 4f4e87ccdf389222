#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for S in 2 4 8; do
  echo "S=$S default"; ANCE_FAST_SPLITS=$S tools/abi_probe search 8841823 32768 200 2 | tail -1
  echo "S=$S nt";      ANCE_FAST_NT=1 ANCE_FAST_SPLITS=$S tools/abi_probe search 8841823 32768 200 2 | tail -1
done
for S in 4 8; do
  ANCE_FAST_NT=1 ANCE_FAST_SPLITS=$S timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/var/nt_s${S} -o p -- tools/abi_probe search 8841823 32768 200 1 > gpurun_out/var/nt_s${S}.log 2>&1
done
python - <<'PY'
import csv, glob
for S in (4,8):
    vals={}
    for f in glob.glob("gpurun_out/var/nt_s%d/**/*counter_collection.csv"%S, recursive=True):
        for r in csv.DictReader(open(f)):
            if "ip_topk_fast" in r["Kernel_Name"]: vals.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    print("NT S",S,{k:[round(x/1e6,1) for x in v] for k,v in vals.items()})
PY
