#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for S in 2 8; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    tag=$(echo $c | cut -d' ' -f1)
    ANCE_FAST_SPLITS=$S timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/var/s${S}_$tag -o p -- tools/abi_probe search 8841823 32768 200 1 > gpurun_out/var/s${S}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob
for S in (2,8):
    for tag in ("FETCH_SIZE","TCC_HIT_sum","GRBM_GUI_ACTIVE"):
        vals={}
        for f in glob.glob("gpurun_out/var/s%d_%s/**/*counter_collection.csv"%(S,tag), recursive=True):
            for r in csv.DictReader(open(f)):
                if "ip_topk_fast" in r["Kernel_Name"]: vals.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
        dur=[]
        for f in glob.glob("gpurun_out/var/s%d_%s/**/*kernel_trace.csv"%(S,tag), recursive=True):
            for r in csv.DictReader(open(f)):
                if "ip_topk_fast" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
        print("S",S,tag,{k:[round(x/1e6,1) for x in v] for k,v in vals.items()},"dur_ms",[round(d,1) for d in dur])
PY
