#!/bin/bash
# the whole GPU suite, stopping at the first failure (with its message)
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5_suite
timeout 2400 python3 -m pytest tests -q -m gpu -x --durations=5 2>&1 | tail -40 | tee gpurun_out/r5_suite/pytest_gpu.log
