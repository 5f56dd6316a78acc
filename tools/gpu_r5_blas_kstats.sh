#!/bin/bash
# round 5: per-kernel times of the BLAS build for the tree and for A/B builds: tools/gpu_r5_blas_kstats.sh NAME...
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_blas_kstats; mkdir -p $O
for v in tree "$@"; do
  if [ $v = tree ]; then unset VOIDIN_HIP_LIB; else export VOIDIN_HIP_LIB=$PWD/build/ab/$v/libvoidin_hip.so; fi
  echo "== $v"; timeout 300 python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 5 --blas-only 2>&1 | grep -v amdgpu.ids | tail -1
  P=$O/prof_$v; rm -rf $P; mkdir -p $P
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 3 --blas-only > $P/stdout.log 2>&1
  V=$v python3 - <<'PY'
import csv, glob, re, os
v = os.environ["V"]
f = glob.glob(f"gpurun_out/r5_blas_kstats/prof_{v}/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name']).split('(')[0].replace('void ', '')
        if float(r['TotalDurationNs']) / 4e6 > 0.12:
            print(f"{n:44s} calls {int(r['Calls']):5d} ms/build {float(r['TotalDurationNs']) / 4e6:7.3f}  avg us {float(r['AverageNs']) / 1e3:8.1f}")
PY
  rm -rf $P
done 2>&1 | tee $O/kstats.log
