#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/gpmc
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TA_[A-Z_]+)\b" | sort -u > gpurun_out/gpmc/counters.txt
wc -l gpurun_out/gpmc/counters.txt
CMD="tools/abi_probe gemm 0 ${SHAPE:-0 8192 8192 8192} 3"
i=0
TAG=${TAG:-x}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/gpmc/${TAG}_s$i -o p -- $CMD > gpurun_out/gpmc/${TAG}_s$i.log 2>&1
  echo "set $i rc=$?"; grep -E "error|Error|invalid" gpurun_out/gpmc/${TAG}_s$i.log | head -3
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/gpmc/%s_s*/**/*counter_collection.csv" % __import__("os").environ.get("TAG","x"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm" not in row.get("Kernel_Name", ""): continue
        k = row["Counter_Name"]; agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
g = {k: v / n for k, (v, n) in agg.items()}
for k in sorted(g): print("%-34s per-dispatch %18.1f" % (k, g[k]))
try:
    cyc = g["GRBM_GUI_ACTIVE"] / 8.0
    print("derived: cycles/XCD %.0f  MFMA busy %.1f%%  LDS busy %.1f%%  wave parked %.1f%% issue-stall %.1f%% active %.1f%%  L2 hit %.1f%%" % (
        cyc, 100 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 100 * g["SQ_LDS_IDX_ACTIVE"] / (256 * cyc),
        100 * g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"],
        100 * g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"])))
except Exception as e:
    print("derive failed", e)
PY
