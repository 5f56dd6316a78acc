#!/bin/bash
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_mask; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o m -- python3 tools/ab_mask.py > $O/stdout.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_mask/**/m_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
import collections
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','')[:50]
    d[(n,r['Grid_Size'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    v=sorted(v); print(f"{k[0]:52s} grid {k[1]:>10s} n={len(v):3d} median {v[len(v)//2]:9.1f} us min {v[0]:9.1f}")
PY
find gpurun_out/prof_mask -name "*kernel_trace.csv" -delete
