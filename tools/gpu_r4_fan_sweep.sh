#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
{
echo "## default build (grace 128, below 32)"; AB_FAN=1,2,3 timeout 300 python tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | head -8
for v in fan_g32 fan_g64 fan_g256 fan_g512 fan_b16 fan_b48 fan_b64; do
  echo "## $v"; VOIDIN_HIP_LIB=$PWD/build/ab/$v/libvoidin_hip.so AB_FAN=2 timeout 300 python tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | head -4
done
} > gpurun_out/r4/ab_fan_sweep.log 2>&1
cat gpurun_out/r4/ab_fan_sweep.log
python tools/bench_bvh.py --u 64 --v 64 --tlas 1000 2>&1 | grep -v amdgpu.ids | grep "trace" 
