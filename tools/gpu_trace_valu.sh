#!/bin/bash
# VALU-busy (and the memory unit's) of the traversal kernels on the stress scene: rocprofv3 --pmc, one counter per pass
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in VALUBusy MemUnitBusy; do
  O=gpurun_out/trace_pmc_$c; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc $c --output-format csv -d $O -o t -- python3 tools/ab_trace.py --reps 2 > $O/stdout.log 2>&1
  C=$c python3 - <<'PY'
import csv, glob, os, re
c = os.environ['C']
acc = {}
for f in glob.glob(f'gpurun_out/trace_pmc_{c}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c: continue
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
        if 'trace' in n or 'fan' in n:
            acc.setdefault(n, []).append(float(r['Counter_Value']))
for n, v in sorted(acc.items(), key=lambda kv: -len(kv[1])):
    print(f"{c:12s} {n:50s} launches {len(v):4d} mean {sum(v) / len(v):6.1f} %  max {max(v):6.1f} %")
PY
  rm -rf $O
done
