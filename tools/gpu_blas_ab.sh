#!/bin/bash
# A/B of two builds of the library on ONE box: tools/gpu_blas_ab.sh NAME  (build/ab/NAME against the tree's library), kernel stats of each
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in tree $1 tree $1; do
  if [ $v = tree ]; then unset VOIDIN_HIP_LIB; else export VOIDIN_HIP_LIB=$PWD/build/ab/$v/libvoidin_hip.so; fi
  echo "== $v"; python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 5 --blas-only 2>&1 | grep -v amdgpu.ids | tail -1
done
for v in tree $1; do
  if [ $v = tree ]; then unset VOIDIN_HIP_LIB; else export VOIDIN_HIP_LIB=$PWD/build/ab/$v/libvoidin_hip.so; fi
  O=gpurun_out/blas_ab_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 3 --blas-only > $O/stdout.log 2>&1
  echo "== $v kernel stats"; V=$v python3 - <<'PY'
import csv, glob, os, re
f = glob.glob(f"gpurun_out/blas_ab_{os.environ['V']}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name']).split('(')[0].replace('void ', '')
    if n.startswith(('a_child', 'a_eval', 'a_boundary', 'blas_mid', 'a_bits', 'a_bin', 'blas_small')):
        print(f"{n:30s} ms/build {float(r['TotalDurationNs']) / 4e6:7.3f}  min us {float(r['MinNs']) / 1e3:8.1f} max us {float(r['MaxNs']) / 1e3:8.1f}")
PY
  rm -rf $O
done
