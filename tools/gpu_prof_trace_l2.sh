#!/bin/bash
# Where the node fetches of vd_trace_dev are served from: L2 hits / misses and the bytes that come over the fabric
# (one counter group per rocprofv3 --pmc pass; bench scenes of tools/bench_bvh.py).  -> gpurun_out/round/rNN_trace_l2.txt
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${1:-r02}
O=gpurun_out/prof_trace_l2; rm -rf $O; mkdir -p $O gpurun_out/round
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o t -- python3 tools/bench_bvh.py --u 64 --v 64 --reps 1 --tlas 1000 > $O/p$i.log 2>&1
done
python3 - > gpurun_out/round/${R}_trace_l2.txt <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/prof_trace_l2/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'trace_single_prep' in n:                      # plain calls de-index their leaves per call and run this kernel too (the indexed one returns at once)
            kind = 'occlusion' if 'Lb1E' in n or '<true' in n else 'closest'
            agg[(kind, r['Counter_Name'])].append(float(r['Counter_Value']))
print('# rocprofv3 --pmc, one group per pass; per launch of trace_single_prep_kernel (tools/bench_bvh.py: the plain vd_trace_dev / vd_trace_any_dev de-index')
print('# their leaves per call, so they run the same kernel as the prepared scene).  Launch order per kind: 3 x stress scene plain, 3 x stress scene prepared,')
print('# then (closest hit only) 3 x the 4 M-ray harness scene.  per ray = the stress mean / 1 048 576 rays.')
for k in sorted(agg):
    v = agg[k]
    a, b = v[:6], v[6:]
    print(k[0], k[1], 'stress mean %.5g (%.4g per ray)' % (sum(a) / len(a), sum(a) / len(a) / 1048576.0),
          ('harness mean %.5g (%.4g per ray)' % (sum(b) / len(b), sum(b) / len(b) / 4194304.0)) if b else '')
PY
cat gpurun_out/round/${R}_trace_l2.txt
for p in $O/p*.log; do grep -h "error\|Error" $p | head -2; done
