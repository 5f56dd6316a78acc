#!/bin/bash
# round 5, mid-round check: the whole GPU suite after the scan / semaphore / tight-pad / test-tightening changes, a default bench line,
# and the mid-tier size A/B (VERDICT r4 item 2 ii: VD_MID_MAX 4096 / 8192 re-measured against the tree's 2048)
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_check; rm -rf $O; mkdir -p $O
echo "== pytest -m gpu"; timeout 2400 python3 -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -25 | tee $O/pytest_gpu.log
echo "== bench"; timeout 900 python3 bench.py 2>&1 | grep -v amdgpu.ids | grep '^{' | tail -1 > $O/bench_line.json; cut -c1-600 $O/bench_line.json
for v in tree mid4096 mid4096p8 mid8192 tree mid4096; do
  if [ $v = tree ]; then unset VOIDIN_HIP_LIB; else export VOIDIN_HIP_LIB=$PWD/build/ab/$v/libvoidin_hip.so; fi
  echo "== $v"; timeout 300 python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 5 --blas-only 2>&1 | grep -v amdgpu.ids | tail -1
done 2>&1 | tee $O/mid_ab.log
unset VOIDIN_HIP_LIB
