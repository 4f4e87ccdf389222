#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
echo "== full GPU suite, one process (as the driver runs it)"
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
echo "== probes"
tools/abi_probe search 8841823 32768 200 2 | tail -1
tools/abi_probe search 8841823 4096 200 2 | tail -1
ANCE_FAST_SPLITS=32 tools/abi_probe search 8841823 4096 200 2 | tail -1
tools/abi_probe search 8841823 6980 100 2 | tail -1
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/var/keepb0 -o p -- tools/abi_probe gemm 0 0 8192 8192 8192 4 > gpurun_out/var/keepb0.log 2>&1
python - <<'PY'
import csv, glob
cyc=[]; dur=[]
for f in glob.glob("gpurun_out/var/keepb0/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE": cyc.append(float(r["Counter_Value"])/8)
for f in glob.glob("gpurun_out/var/keepb0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("keepB0 8192^3 cycles/XCD", [int(c) for c in cyc], "dur_us", [int(d) for d in dur])
PY
