#!/bin/bash
# A/B of several builds of the library on ONE box: tools/gpu_blas_ab_multi.sh NAME... (build/ab/NAME; "tree" = the tree's library).
# Two passes over the list (interleaved), best of 5 builds each time.
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for pass in 1 2; do
  for v in "$@"; do
    if [ $v = tree ]; then unset VOIDIN_HIP_LIB; else export VOIDIN_HIP_LIB=$PWD/build/ab/$v/libvoidin_hip.so; fi
    printf "%-12s " $v; python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 5 --blas-only 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
