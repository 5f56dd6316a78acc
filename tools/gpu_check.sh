#!/bin/bash
# One gpurun call: parity tests, smoke, bench; everything lands under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 50 --warmup 5 --extra 2>&1 | tail -5 | tee gpurun_out/bench.log
