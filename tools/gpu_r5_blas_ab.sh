#!/bin/bash
# round 5: phase-A A/Bs on one box: tools/gpu_r5_blas_ab.sh NAME... (build/ab/NAME against the tree), bit-exactness of the tree first
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_blas_ab; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_blas.py tests/test_gpu_blas_batch.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 600 python3 -m pytest tests/test_gpu_full_size.py -x -q -k "blas" 2>&1 | tail -3 | tee -a $O/pytest.log
for v in tree "$@" tree "$@"; do
  if [ $v = tree ]; then unset VOIDIN_HIP_LIB; else export VOIDIN_HIP_LIB=$PWD/build/ab/$v/libvoidin_hip.so; fi
  echo "== $v"; timeout 300 python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 5 --blas-only 2>&1 | grep -v amdgpu.ids | tail -1
done 2>&1 | tee $O/ab.log
unset VOIDIN_HIP_LIB
if [ -f build/ab/boundary_prof/libvoidin_hip.so ]; then
  VOIDIN_HIP_LIB=$PWD/build/ab/boundary_prof/libvoidin_hip.so timeout 300 python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 1 --blas-only 2>&1 | grep "a_boundary" | tail -20 | tee $O/boundary_prof.log
fi
P=$O/prof; rm -rf $P; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 3 --blas-only > $P/stdout.log 2>&1
python3 - <<'PY' | tee gpurun_out/r5_blas_ab/kernel_stats.txt
import csv, glob, re
f = glob.glob("gpurun_out/r5_blas_ab/prof/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name']).split('(')[0].replace('void ', '')
        if float(r['TotalDurationNs']) / 4e6 > 0.1:
            print(f"{n:44s} calls {int(r['Calls']):5d} ms/build {float(r['TotalDurationNs']) / 4e6:7.3f}  avg us {float(r['AverageNs']) / 1e3:8.1f} max us {float(r['MaxNs']) / 1e3:8.1f}")
PY
rm -rf $P
