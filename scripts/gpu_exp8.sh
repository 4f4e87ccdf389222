#!/bin/bash
# Cycle counts (GRBM_GUI_ACTIVE: clock-independent) + durations of the schedule variants.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for v in 0 8 32 33 34 35 36 37 38 39 40 41; do
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/var/v$v -o p -- tools/abi_probe gemm $v 0 8192 8192 8192 4 > gpurun_out/var/v$v.log 2>&1
done
python - <<'PY'
import csv, glob
for v in (0,8,32,33,34,35,36,37,38,39,40,41):
    cyc=[]; dur=[]
    for f in glob.glob("gpurun_out/var/v%d/**/*counter_collection.csv"%v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE": cyc.append(float(r["Counter_Value"])/8)
    for f in glob.glob("gpurun_out/var/v%d/**/*kernel_trace.csv"%v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    if cyc: print("ablate %2d cycles/XCD %s  dur_us %s  MFMA-peak-frac(cycles) %.3f" % (v, [int(c) for c in cyc], [int(d) for d in dur], (2*8192**3/(1024*1024))/ (sum(cyc)/len(cyc))))
PY
