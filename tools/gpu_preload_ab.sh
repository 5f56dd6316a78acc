#!/bin/bash
# tree's library against build/ab/$1 on the launch-bound paths: small fused culls, the headline, TLAS build / refit, the stress-scene trace
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in tree $1 tree $1; do
  if [ $v = tree ]; then unset VOIDIN_HIP_LIB; else export VOIDIN_HIP_LIB=$PWD/build/ab/$v/libvoidin_hip.so; fi
  echo "== $v"
  for n in 1000 100000 1000000 10000000; do python3 tools/ab_cull.py --n $n --variants 0 --iters 200 2>&1 | grep "^variant.*median" | sed "s/^/cull n=$n /"; done
  python3 tools/tlas_time.py 1000 8192 32768 2>&1 | grep "^n="
  python3 tools/refit_loop.py 2>&1 | grep -v amdgpu.ids | tail -2
  python3 tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | grep "^single rays" | head -3
done
