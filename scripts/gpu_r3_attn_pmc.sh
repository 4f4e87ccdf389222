#!/bin/bash
# Round 3: where the time of the attention launch goes -- SQ / TCP / TCC counters of both attention kernels on the encode
# leg (one --pmc set per pass, kernel-trace only beside it), plus the list of counters this box offers.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3attn
mkdir -p $O
export TMPDIR=/tmp
echo "== parity (hazard fix)"
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/test_default.log 2>&1; echo "rc=$?"; tail -3 $O/test_default.log
(cd /tmp && rocprofv3 --list-avail > /root/repo/$O/avail.txt 2>&1); grep -c "" $O/avail.txt
CMD="python bench.py --skip-search --no-cpu-baseline --steps 1 --warmup 1"
pass() {  # tag, env, counters...
  tag=$1; envs=$2; shift 2
  env $envs ANCE_ENCODER_STREAMS=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "attention" --output-format csv -d $O/$tag -o pmc -- $CMD > $O/$tag.log 2>&1
  echo "pass[$tag] rc=$?"
}
for k in reg lds; do
  e="A=1"; [ $k = lds ] && e="ANCE_ATTN_REG=0"
  pass ${k}_sq1 $e SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pass ${k}_sq2 $e SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
  pass ${k}_sq3 $e SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES
  pass ${k}_tcp $e TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
  pass ${k}_tcc $e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pass ${k}_grbm $e GRBM_GUI_ACTIVE
done
find $O -name "*kernel_trace.csv" -size +4M -delete
python - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r3attn"
for d in sorted(glob.glob(O + "/*/")):
    tag = os.path.basename(d.rstrip("/"))
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[(r.get("Kernel_Name", "")[:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print("%-10s %-40s %-36s per-launch %.4g  (n=%d)" % (tag, k[-40:], c, v / max(n, 1), n))
PY
