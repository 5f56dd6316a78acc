#!/bin/bash
# Per-launch durations of the once-per-level kernels of ONE 8.4 M-triangle build (the last of 3), from a rocprofv3 kernel trace.
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/blas_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 2 --blas-only > $O/stdout.log 2>&1
python3 - <<'PY' | tee gpurun_out/blas_trace_levels.txt
import csv, glob, re
f = glob.glob('gpurun_out/blas_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def kn(r): return re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
pre = [i for i, r in enumerate(rows) if kn(r).startswith('blas_precompute')]
last = rows[pre[-1]:]
for name in ('a_bits_kernel', 'a_bin_kernel', 'a_child_kernel', 'a_count_kernel', 'a_eval_kernel', 'a_boundary_kernel', 'blas_mid_kernel', 'blas_small_kernel'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last if kn(r).startswith(name)]
    print(f"{name:20s} us by level: " + " ".join(f"{x:.0f}" for x in d))
t0, t1 = int(last[0]['Start_Timestamp']), int(last[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last)
print(f"span {(t1 - t0) / 1e6:.2f} ms, sum of kernel time {busy / 1e6:.2f} ms, {len(last)} launches")
PY
rm -rf $O
