#!/bin/bash
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_bvh
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bvh -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 2 --tlas 1000 > gpurun_out/prof_bvh/stdout.log 2>&1
find gpurun_out/prof_bvh -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_bvh/bvh_kernel_stats.csv')))
for r in rows[:14]:
    name=r['Name'].replace('(anonymous namespace)::','')[:40]
    print(f"{name:42s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
