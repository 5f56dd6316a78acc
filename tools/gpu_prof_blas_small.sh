#!/bin/bash
# What phase B (blas_small_kernel) and the mid tier wait for: SQ counters, one group per rocprofv3 --pmc pass,
# on the 8.4 M-triangle build (tools/bench_bvh.py --blas-only).  -> gpurun_out/round/rNN_blas_small_sq.txt
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${1:-r02}
O=gpurun_out/prof_blas_small; rm -rf $O; mkdir -p $O gpurun_out/round
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o t -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 1 --blas-only > $O/p$i.log 2>&1
done
python3 - > gpurun_out/round/${R}_blas_small_sq.txt <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/prof_blas_small/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for k in ('blas_small_kernel', 'blas_mid_kernel'):
            if k in r['Kernel_Name']:
                agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
print('# rocprofv3 --pmc, one group per pass; mean per launch (2 launches: warm-up build + timed build of 8 388 608 triangles)')
for k in sorted(agg):
    v = agg[k]
    print(k[0], k[1], 'mean %.5g' % (sum(v) / len(v)), 'n', len(v))
PY
cat gpurun_out/round/${R}_blas_small_sq.txt
