#!/bin/bash
# Runs the given GPU test files (default: all), one pytest process per file; logs under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
files="$@"
if [ -z "$files" ]; then files=$(cd tests && ls test_gpu_*.py | sed 's/\.py$//'); fi
for f in $files; do
  echo "== $f"
  timeout 1500 python -m pytest tests/$f.py -m gpu -q --timeout 1200 -p no:cacheprovider > gpurun_out/$f.log 2>&1
  echo "$f rc=$?"; tail -12 gpurun_out/$f.log
done
