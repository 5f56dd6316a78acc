#!/bin/bash
# rocprofv3 kernel trace of the bench (separate from PMC passes, as gpurun requires)
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-verify > gpurun_out/prof/bench_stdout.log 2>&1
tail -2 gpurun_out/prof/bench_stdout.log
find gpurun_out/prof -name "*stats*" | head
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "== $f"; head -8 "$f"; done
# drop the big per-dispatch trace, keep the summaries
find gpurun_out/prof -name "*kernel_trace.csv" -size +2M -delete
