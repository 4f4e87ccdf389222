#!/bin/bash
# rocprofv3 on the torch-free ABI probe: kernel-trace stats + PMC (HBM bytes) in separate runs.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
S="tools/abi_probe search ${PMC_N:-8841823} ${PMC_NQ:-32768} 200 2"
E="tools/abi_probe encode ${PMC_NP:-16384} 128 12 2 65536"
echo "== plain runs"; timeout 300 $S; timeout 300 $E
for what in search encode; do
  cmd="$S"; [ $what = encode ] && cmd="$E"
  echo "== kernel-trace $what"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/kt_$what -o kt -- $cmd > gpurun_out/pmc/kt_$what.log 2>&1; echo "rc=$?"
  echo "== pmc cycles $what (GRBM_GUI_ACTIVE = shader clocks of the dispatch: clock-independent cost, and the clock itself)"
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc/CYCLES_$what -o pmc -- $cmd > gpurun_out/pmc/CYCLES_$what.log 2>&1; echo "rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c $what"
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/${c}_$what -o pmc -- $cmd > gpurun_out/pmc/${c}_$what.log 2>&1; echo "rc=$?"
    tail -3 gpurun_out/pmc/${c}_$what.log | cut -c1-200
  done
done
find gpurun_out/pmc -name "*.csv" | head -30
python scripts/summarize_pmc.py gpurun_out/pmc gpurun_out/pmc/pmc_traffic.json 2>&1 | head -70
