#!/bin/bash
# Per-kernel time of the 8.4 M-triangle BLAS build (rocprofv3 --kernel-trace --stats; 1 warm-up + REPS timed builds), per build.
#   gpurun -- 'bash tools/gpu_blas_stats.sh [REPS]'  ->  gpurun_out/blas_stats.txt
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
REPS=${1:-3}
O=gpurun_out/blas_stats; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps $REPS --blas-only > $O/stdout.log 2>&1
grep -v amdgpu.ids $O/stdout.log | tail -3
REPS=$REPS python3 - <<'PY' | tee gpurun_out/blas_stats.txt
import csv, glob, os, re
builds = int(os.environ['REPS']) + 1
f = glob.glob('gpurun_out/blas_stats/**/*kernel_stats.csv', recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name']).split('(')[0].replace('void ', '')
    if not (n.startswith('a_') or n.startswith('blas_') or n.startswith('c_') or n.startswith('b_')):
        continue
    ms = float(r['TotalDurationNs']) / builds / 1e6
    tot += ms
    print(f"{n:36s} launches/build {int(r['Calls']) / builds:7.1f}  ms/build {ms:7.3f}  avg us {float(r['AverageNs']) / 1e3:8.1f}  max us {float(r['MaxNs']) / 1e3:8.1f}")
print(f"sum of kernel time per build: {tot:.2f} ms")
PY
find $O -name "*.csv" -delete
