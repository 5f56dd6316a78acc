#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
{
for v in "" fan_a256 fan_a512 fan_a1024 fan_a512_g32; do
  L=$PWD/voidin_amd/csrc/libvoidin_hip.so; [ -n "$v" ] && L=$PWD/build/ab/$v/libvoidin_hip.so
  echo "## ${v:-default build (age 0, grace 128, below 32)}"
  VOIDIN_HIP_LIB=$L AB_FAN=2,3 timeout 300 python tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | grep "prep"
  VOIDIN_HIP_LIB=$L timeout 300 python tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | grep "TIGHT"
done
} > gpurun_out/r4/ab_fan_age_sweep.log 2>&1
cat gpurun_out/r4/ab_fan_age_sweep.log
