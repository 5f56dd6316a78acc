#!/bin/bash
# PMC passes over vd_trace_dev on the bench scene (one counter group per pass, as the pool requires)
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_trace; rm -rf $O; mkdir -p $O
i=0
for grp in "GRBM_GUI_ACTIVE TA_BUSY_avr" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o t -- python3 tools/bench_bvh.py --u 64 --v 64 --tlas 1000 > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/prof_trace/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'trace_kernel' in r['Kernel_Name']:
            kind='any' if 'Lb1E' in r['Kernel_Name'] or '<true>' in r['Kernel_Name'] else 'closest'
            agg[(kind,r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(agg): print(k[0], k[1], 'n', len(agg[k]), 'mean %.4g' % (sum(agg[k])/len(agg[k])))
PY
