#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_blas_batch.py tests/test_gpu_blas.py tests/test_gpu_full_size.py tests/test_cpp_mirror.py tests/test_gltf_helmet.py -x -q -m gpu 2>&1 | tail -8 | cut -c1-300
python tools/bench_bvh.py --u 2048 --v 2048 --reps 4 2>&1 | grep -v amdgpu.ids | head -8 | tee gpurun_out/r4/bench_bvh.log
VD_BLAS_HALVES=0 python tools/bench_bvh.py --u 2048 --v 2048 --reps 4 2>&1 | grep -v amdgpu.ids | head -8 | grep "BLAS build"
