#!/bin/bash
# Copies the judged summaries of the last scripts/gpu_round_end.sh (+ bench.py --full) run from gpurun_out/ to profiles/.
# usage: scripts/collect_profiles.sh r02
set -u
cd "$(dirname "$0")/.."
r=${1:?round tag, e.g. r02}
cp gpurun_out/bench.log profiles/${r}_bench_default.json
cp gpurun_out/prof_bench/kt_kernel_stats.csv profiles/${r}_rocprofv3_bench_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc/kt_search/kt_kernel_stats.csv profiles/${r}_rocprofv3_search_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc/kt_encode/kt_kernel_stats.csv profiles/${r}_rocprofv3_encode_single_stream_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc/kt_encode_split/kt_kernel_stats.csv profiles/${r}_rocprofv3_encode_split_single_stream_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc/kt_encode_fp32/kt_kernel_stats.csv profiles/${r}_rocprofv3_encode_fp32_single_stream_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
cp gpurun_out/bench_configs.jsonl profiles/${r}_bench_other_configs.jsonl 2>/dev/null
cp gpurun_out/encoder_parity.jsonl profiles/${r}_encoder_parity.jsonl 2>/dev/null
cp gpurun_out/retrieval_agreement.json profiles/${r}_retrieval_agreement.json 2>/dev/null
cp gpurun_out/faiss_boundary.json profiles/${r}_faiss_boundary.json 2>/dev/null
cp gpurun_out/config1_agreement.json profiles/${r}_config1_agreement.json 2>/dev/null
[ -f gpurun_out/bench_full.log ] && cp gpurun_out/bench_full.log profiles/${r}_full_refresh.json
[ -f gpurun_out/bench_full_split_2m.log ] && cp gpurun_out/bench_full_split_2m.log profiles/${r}_full_refresh_split_2m.json
ls -la profiles | tail -30
