#!/bin/bash
# round 4: tight-TLAS tests, then the traversal A/B with the tuning build's counters and timeline
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_trace_tight.py tests/test_gpu_tlas_trace.py -x -q -m gpu 2>&1 | tail -15
AB_COUNTERS=1 VOIDIN_HIP_LIB=$PWD/voidin_amd/csrc/libvoidin_hip_tuning.so python tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | head -40 > gpurun_out/r4/ab_trace_tuning.log
python tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | head -12 > gpurun_out/r4/ab_trace.log
cat gpurun_out/r4/ab_trace_tuning.log | cut -c1-400 | head -30
cat gpurun_out/r4/ab_trace.log
