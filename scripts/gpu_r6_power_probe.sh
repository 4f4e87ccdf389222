#!/bin/bash
# Round 6: the fp16 MFMA rate the board sustains at its power cap, with and without a GEMM main loop's data movement (tools/power_probe.cpp)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 tools/power_probe 5 > gpurun_out/power_probe.txt 2>&1; echo "rc=$?"; cat gpurun_out/power_probe.txt
