#!/bin/bash
# Build A/B copies of libvoidin_hip.so with extra -D flags: tools/build_ab.sh NAME "-DFOO=1" -> build/ab/NAME/libvoidin_hip.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=build/ab/$name; mkdir -p $d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
for f in ctx cull tlas trace blas hiz dist; do
  X=""; if [ "$f" = blas ]; then X="-mllvm -amdgpu-kernarg-preload-count=16"; fi   # the Makefile's per-file flag: both legs of an A/B get it
  if [ "$f" = "${AB_FILE:-cull}" ] || [ ! -f voidin_amd/csrc/$f.o ]; then /opt/rocm/bin/hipcc $F $X "$@" -c voidin_amd/csrc/$f.hip -o $d/$f.o; else cp voidin_amd/csrc/$f.o $d/$f.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libvoidin_hip.so $d/*.o
rm -f $d/*.o
echo built $d
