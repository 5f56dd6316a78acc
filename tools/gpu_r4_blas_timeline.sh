#!/bin/bash
# kernel timeline of one 8.4 M-triangle build: busy time, idle gaps, overlap of the two halves' streams
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for H in 1 0; do
O=gpurun_out/r4/blas_kt; rm -rf $O; mkdir -p $O
VD_BLAS_HALVES=$H rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 2 --blas-only > $O/stdout.log 2>&1
grep "BLAS build" $O/stdout.log
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/r4/blas_kt/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
# last build: take kernels after the last blas_precompute_kernel
idx=[i for i,r in enumerate(rows) if 'blas_precompute' in r['Kernel_Name']]
rows=rows[idx[-1]:]
end=[i for i,r in enumerate(rows) if 'c_permute' in r['Kernel_Name']][0]
rows=rows[:end+1]
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r.get('Queue_Id') or r.get('Stream_Id') or '?',r['Kernel_Name'].split('(')[0][-30:]) for r in rows)
t0=ev[0][0]; t1=max(e[1] for e in ev)
# busy union and overlap
pts=[]
for s,e,q,n in ev: pts+=[(s,1),(e,-1)]
pts.sort()
busy=0; over=0; cur=0; last=pts[0][0]
for t,d in pts:
    if cur>=1: busy+=t-last
    if cur>=2: over+=t-last
    cur+=d; last=t
qs=collections.Counter(q for _,_,q,_ in ev)
print('HALVES=$H: span %.2f ms, some kernel running %.2f ms, two or more %.2f ms, idle %.2f ms, kernels %d, queues %s' % ((t1-t0)/1e6,busy/1e6,over/1e6,(t1-t0-busy)/1e6,len(ev),dict(qs)))
byk=collections.defaultdict(float)
for s,e,q,n in ev: byk[n]+= (e-s)/1e6
print('   ', ', '.join('%s %.2f' % (k,v) for k,v in sorted(byk.items(), key=lambda x:-x[1])[:8]))
PY
rm -rf $O
done
