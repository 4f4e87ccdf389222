#!/bin/bash
# rocBLAS yardstick under the same counters: cycles vs clock.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
tools/rocblas_probe 8192 8192 8192 10
tools/rocblas_probe 65536 3072 768 20
tools/rocblas_probe 65536 768 3072 20
tools/abi_probe gemm 0 0 8192 8192 8192 10 | tail -1
for shape in "8192 8192 8192" "65536 3072 768"; do
  tag=$(echo $shape | tr ' ' _)
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/var/rb_$tag -o p -- tools/rocblas_probe $shape 5 > gpurun_out/var/rb_$tag.log 2>&1
done
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/var/our_ffn1 -o p -- tools/abi_probe gemm 0 1 65536 3072 768 5 > gpurun_out/var/our_ffn1.log 2>&1
python - <<'PY'
import csv, glob
for d in sorted(glob.glob("gpurun_out/var/rb_*/")) + ["gpurun_out/var/our_ffn1/"]:
    cyc={}; dur={}
    for f in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]=="GRBM_GUI_ACTIVE": cyc.setdefault(r["Kernel_Name"][:60],[]).append(float(r["Counter_Value"])/8)
    for f in glob.glob(d+"**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur.setdefault(r["Kernel_Name"][:60],[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k in cyc:
        if len(cyc[k])>=3: print(d, k, "cycles/XCD", [int(c) for c in cyc[k]][:8], "dur_us", [int(x) for x in dur.get(k,[])][:8])
PY
