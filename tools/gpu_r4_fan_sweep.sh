#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
{
for V in fan_wps5 fan_wps4; do for W in 16 20 24 28; do
echo "## $V, VD_TRACE_WAVES=$W per CU"
VD_TRACE_WAVES=$W VOIDIN_HIP_LIB=$PWD/build/ab/$V/libvoidin_hip.so python tools/ab_trace.py --reps 4 2>&1 | grep -v amdgpu.ids | grep "single rays prep  "
done; done
} > gpurun_out/r4/ab_fan_wps2.log 2>&1
cat gpurun_out/r4/ab_fan_wps2.log
