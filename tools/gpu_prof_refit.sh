#!/bin/bash
# Kernel trace of a queued refit loop: durations and the gaps between consecutive kernels.
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refit; rm -rf $O; mkdir -p $O
python3 tools/refit_loop.py ${1:-32768} 100
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o refit -- python3 tools/refit_loop.py ${1:-32768} 20 > $O/stdout.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/refit/kt/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'refit' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-40:]
prev = None
for r in rows[:12]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = 'prep' if 'prep' in r['Kernel_Name'] else 'up  '
    print(name, 'dur %7.1f us' % ((e - s) / 1e3), 'gap before %7.1f us' % ((s - prev) / 1e3 if prev else 0))
    prev = e
PY
