#!/bin/bash
# round 6: one launch per phase-A round (a_round_kernel) against the two-launch form, same library, same box:
#   gpurun -- 'bash tools/gpu_r6_blas_fused.sh'      -> gpurun_out/r6_blas_fused/{pytest,ab,kstats}.log
# VD_BLAS_FUSED_ROUNDS=0 selects the round-5 form through the per-context option (voidin_amd/abi.py OPTION_ENV).
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_blas_fused; mkdir -p $O
if [ "${SKIP_TESTS:-0}" != 1 ]; then
export VD_BLAS_FUSED_ROUNDS=1
timeout 900 python3 -m pytest tests/test_gpu_blas.py tests/test_gpu_blas_batch.py tests/test_gpu_fuzz.py -x -q -k "blas" 2>&1 | tail -5 | tee $O/pytest.log
timeout 600 python3 -m pytest tests/test_gpu_full_size.py -x -q -k "blas" 2>&1 | tail -3 | tee -a $O/pytest.log
unset VD_BLAS_FUSED_ROUNDS
fi
for v in 1 0 1 0; do
  echo "== VD_BLAS_FUSED_ROUNDS=$v"; VD_BLAS_FUSED_ROUNDS=$v timeout 300 python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 5 --blas-only 2>&1 | grep -v amdgpu.ids | tail -1
done 2>&1 | tee $O/ab.log
for v in 1 0; do
  echo "== per kernel, VD_BLAS_FUSED_ROUNDS=$v"
  P=$O/prof_$v; rm -rf $P; mkdir -p $P
  export VD_BLAS_FUSED_ROUNDS=$v
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 3 --blas-only > $P/stdout.log 2>&1
  unset VD_BLAS_FUSED_ROUNDS
  V=$v python3 - <<'PY'
import csv, glob, re, os
v = os.environ["V"]
f = glob.glob(f"gpurun_out/r6_blas_fused/prof_{v}/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name']).split('(')[0].replace('void ', '')
        if float(r['TotalDurationNs']) / 4e6 > 0.1:
            print(f"{n:44s} calls {int(r['Calls']):5d} ms/build {float(r['TotalDurationNs']) / 4e6:7.3f}  avg us {float(r['AverageNs']) / 1e3:8.1f} max us {float(r['MaxNs']) / 1e3:8.1f}")
PY
  rm -rf $P
done 2>&1 | tee $O/kstats.log
