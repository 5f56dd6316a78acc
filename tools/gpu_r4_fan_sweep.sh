#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
{
for v in "" fan_e1024 fan_e1536 fan_e2048; do
  L=$PWD/voidin_amd/csrc/libvoidin_hip.so; [ -n "$v" ] && L=$PWD/build/ab/$v/libvoidin_hip.so
  echo "## ${v:-default build (no early fan-out)}"
  VOIDIN_HIP_LIB=$L AB_FAN=3,4 timeout 300 python tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | grep "prep"
done
} > gpurun_out/r4/ab_fan_early_sweep.log 2>&1
cat gpurun_out/r4/ab_fan_early_sweep.log
