#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_blas_batch.py tests/test_gpu_blas.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -6
python tools/bench_bvh.py --u 2048 --v 2048 --reps 4 2>&1 | grep -v amdgpu.ids | head -8 | tee gpurun_out/r4/bench_bvh.log
VD_BLAS_TWO_STREAMS=0 python tools/bench_bvh.py --u 2048 --v 2048 --reps 4 2>&1 | grep -v amdgpu.ids | head -8 | grep "BLAS build"
