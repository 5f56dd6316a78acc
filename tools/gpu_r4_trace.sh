#!/bin/bash
# round 4: fan-out tests + A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_tlas_trace.py -x -q -m gpu -k "fan" 2>&1 | tail -12 | cut -c1-300
AB_FAN=1,2,3,4 timeout 300 python tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | head -12 > gpurun_out/r4/ab_fan.log
cat gpurun_out/r4/ab_fan.log
AB_FAN=1,4 AB_COUNTERS=1 VOIDIN_HIP_LIB=$PWD/voidin_amd/csrc/libvoidin_hip_tuning.so timeout 300 python tools/ab_trace.py --reps 2 2>&1 | grep -v amdgpu.ids | head -40 > gpurun_out/r4/ab_fan_tuning.log
cut -c1-420 gpurun_out/r4/ab_fan_tuning.log
