#!/bin/bash
# Full GPU pass for a round: parity tests, smoke, bench, rocprof kernel stats + PMC passes.
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round
rm -rf $O; mkdir -p $O
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.log
echo "== bench"; timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/bench.json
echo "== rocprof kernel stats (bench)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-verify > $O/kt_stdout.log 2>&1
find $O/kt -name "*kernel_trace.csv" -delete
head -4 $O/kt/bench_kernel_stats.csv | cut -c1-220
echo "== rocprof pmc"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o cull -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-extra > $O/pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o cull -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-extra > $O/pmc_write_stdout.log 2>&1
python3 - <<'PY'
import csv,glob,json
out={'note':'mean per launch; rocprofv3 --pmc, one counter per pass; units KB; gfx950 FETCH_SIZE = 1/2 of the bytes read by wide coalesced loads'}
for d,c in [('pmc_fetch','FETCH_SIZE'),('pmc_write','WRITE_SIZE')]:
    for f in glob.glob(f'gpurun_out/round/{d}/*counter_collection.csv'):
        rows=list(csv.DictReader(open(f)))
        for kern in ['cull_mask_tiled_kernel','expand_mask_u8_kernel','mask_scan_kernel','cull_compact_kernel']:
            v=[float(r['Counter_Value']) for r in rows if kern in r['Kernel_Name'] and r['Counter_Name']==c]
            if v:
                out.setdefault(kern,{})[c+'_KB']=sum(v)/len(v); out[kern][c+'_launches']=len(v)
json.dump(out,open('gpurun_out/round/cull_pmc.json','w'),indent=1); print(out)
PY
