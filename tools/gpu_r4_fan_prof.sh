#!/bin/bash
# per-launch durations of the fan-out's launches (kernel trace) on the stress scene
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for F in "$@"; do
O=gpurun_out/r4/fan_kt; rm -rf $O; mkdir -p $O
VD_TRACE_FAN=$F rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/trace_stress.py fan$F 1024 1 > $O/stdout.log 2>&1
grep "closest" $O/stdout.log | tail -1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4/fan_kt/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if any(k in r['Kernel_Name'] for k in ('trace_single','fan_'))]
rows=rows[-22:]
t0=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print('   ', r['Kernel_Name'].split('(')[0][-40:], 'start +%.3f ms' % ((s-t0)/1e6), 'dur %.3f ms' % ((e-s)/1e6))
PY
rm -rf $O
done
