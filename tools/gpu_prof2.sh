#!/bin/bash
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_bvh gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bvh -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 2 --tlas 4096 > gpurun_out/prof_bvh/stdout.log 2>&1
head -30 gpurun_out/prof_bvh/bvh_kernel_stats.csv | cut -c1-200
find gpurun_out/prof_bvh -name "*kernel_trace.csv" -delete
# PMC passes for the cull kernel (separate runs, as required)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o cull -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/pmc_fetch/stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o cull -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/pmc_write/stdout.log 2>&1
ls gpurun_out/pmc_fetch gpurun_out/pmc_write
python3 - <<'PY'
import csv,glob
for d in ['pmc_fetch','pmc_write']:
    for f in glob.glob(f'gpurun_out/{d}/*counter_collection.csv'):
        rows=list(csv.DictReader(open(f)))
        ks={}
        for r in rows:
            if 'cull_compact' in r['Kernel_Name']:
                ks.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        for k,v in ks.items(): print(d,k,'n',len(v),'mean',sum(v)/len(v))
PY
