#!/bin/bash
# Full GPU pass for a round: parity tests, smoke, bench (N = 1 and, with gloo on one device, N = 2), rocprof kernel stats +
# PMC passes.  Usage (from the repo root, through gpurun): bash tools/gpu_round.sh r02
# Everything lands in gpurun_out/round/ under the names profiles/ uses; tools/make_profiles_readme.py then writes
# profiles/README.md from those files (no number in it is typed by hand).
set -u
R=${1:-r00}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round
rm -rf $O; mkdir -p $O
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -14 | tee $O/${R}_pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/${R}_smoke.log
echo "== bench"; timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | grep '^{' | tail -1 > $O/${R}_bench_line.json; cut -c1-400 $O/${R}_bench_line.json
echo "== bench --gpus 2 (two ranks on this one GPU, gloo rendezvous: functional, the timing is not a scaling number)"
VOIDIN_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --no-extra 2>&1 | grep '^{' | tail -1 > $O/${R}_bench_line_2ranks_1gpu_gloo.json; cut -c1-300 $O/${R}_bench_line_2ranks_1gpu_gloo.json
echo "== the driver's multi-GPU form: torch.distributed.run starts the ranks (here 2 on one GPU over gloo)"
VOIDIN_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-extra --gather shard 2>&1 | grep '^{' | tail -1 > $O/${R}_bench_line_torchrun_2ranks_shard.json; cut -c1-300 $O/${R}_bench_line_torchrun_2ranks_shard.json
echo "== rocprof kernel stats (bench)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-verify > $O/kt_stdout.log 2>&1
find $O/kt -name "*kernel_trace.csv" -delete
cp $(find $O/kt -name "bench_kernel_stats.csv" | head -1) $O/${R}_bench_kernel_stats.csv
head -4 $O/${R}_bench_kernel_stats.csv | cut -c1-200
echo "== rocprof kernel stats (headline legs only: --no-extra)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kh -o head -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-verify --no-extra > $O/kh_stdout.log 2>&1
find $O/kh -name "*kernel_trace.csv" -delete
cp $(find $O/kh -name "head_kernel_stats.csv" | head -1) $O/${R}_bench_noextra_kernel_stats.csv
head -3 $O/${R}_bench_noextra_kernel_stats.csv | cut -c1-200
echo "== rocprof kernel stats (BVH side)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 2 > $O/kb_stdout.log 2>&1
find $O/kb -name "*kernel_trace.csv" -delete
cp $(find $O/kb -name "bvh_kernel_stats.csv" | head -1) $O/${R}_bvh_kernel_stats.csv
grep -v amdgpu.ids $O/kb_stdout.log | tail -22 > $O/${R}_bench_bvh.log
echo "== rocprof pmc"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o cull -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-extra > $O/pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o cull -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-extra > $O/pmc_write_stdout.log 2>&1
R=$R python3 - <<'PY'
import csv,glob,json,os,shutil
R=os.environ['R']
out={'note':'mean per launch; rocprofv3 --pmc, one counter per pass; units KB; gfx950 FETCH_SIZE = 1/2 of the bytes read by wide coalesced loads'}
for d,c in [('pmc_fetch','FETCH_SIZE'),('pmc_write','WRITE_SIZE')]:
    for f in glob.glob(f'gpurun_out/round/{d}/**/*counter_collection.csv',recursive=True):
        rows=list(csv.DictReader(open(f)))
        keep=[r for r in rows if any(k in r['Kernel_Name'] for k in ('cull_mask_tiled_kernel','expand_mask_u8_kernel','mask_scan_kernel','cull_compact_kernel'))]
        with open(f'gpurun_out/round/{R}_cull_pmc_{c.lower()}.csv','w',newline='') as g:
            w=csv.DictWriter(g,fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep[:72])
        for kern in ['cull_mask_tiled_kernel','expand_mask_u8_kernel','mask_scan_kernel','cull_compact_kernel']:
            v=[float(r['Counter_Value']) for r in rows if kern in r['Kernel_Name'] and r['Counter_Name']==c]
            if v:
                out.setdefault(kern,{})[c+'_KB']=sum(v)/len(v); out[kern][c+'_launches']=len(v)
json.dump(out,open(f'gpurun_out/round/{R}_cull_pmc.json','w'),indent=1); print(out)
PY
rm -rf $O/kt $O/kh $O/kb $O/pmc_fetch $O/pmc_write
ls -la $O
echo "== traversal A/B with the tuning build's counters (lane-steps, entries per ray, timeline): exact walk, fan-out, tight top level"
make -C voidin_amd/csrc tuning > /dev/null 2>&1
{ echo "# stress scene (2000 instances of a 131 k-triangle knot, 1 M primary rays), python tools/ab_trace.py; Mrays/s by HIP events";
  python3 tools/ab_trace.py --reps 4 2>&1 | grep -v amdgpu.ids | head -10;
  echo "## launches per call (VD_OPT_TRACE_FAN): 1 = the round-3 kernel"; AB_FAN=1,2,3,4 python3 tools/ab_trace.py --reps 3 2>&1 | grep -v amdgpu.ids | grep '^fan' | head -8;
  echo "## what the waves do (tuning build, ~2x slower; vd_debug_trace_counters / vd_debug_trace_timeline)";
  AB_COUNTERS=1 VOIDIN_HIP_LIB=$PWD/voidin_amd/csrc/libvoidin_hip_tuning.so python3 tools/ab_trace.py --reps 2 2>&1 | grep -v amdgpu.ids | head -14 | cut -c1-700;
  AB_FAN=1,3 AB_COUNTERS=1 VOIDIN_HIP_LIB=$PWD/voidin_amd/csrc/libvoidin_hip_tuning.so python3 tools/ab_trace.py --reps 2 2>&1 | grep -v amdgpu.ids | head -20 | cut -c1-700; } > $O/${R}_ab_trace.log 2>&1
tail -5 $O/${R}_ab_trace.log | cut -c1-200
ls -la $O
echo "== fuzz campaign with fresh seeds (the fixed-seed slices run inside pytest -m gpu: tests/test_gpu_fuzz.py)"
RN=${R#r}; RN=$((10#$RN))
{ echo "# round $RN fuzz campaign at the round's final tree, one MI355X: random inputs through the C ABI against the oracle, byte for byte (tools/fuzz_*.py)";
  for sd in $((RN * 100 + 11)) $((RN * 100 + 12)); do echo "## fuzz_blas.py --cases 2400 --max-tris 12000 --seed $sd"; timeout 900 python3 tools/fuzz_blas.py --cases 2400 --max-tris 12000 --seed $sd 2>&1 | grep -v amdgpu.ids | tail -4; done;
  echo "## fuzz_blas.py --cases 300 --max-tris 250000 --seed $((RN * 100 + 31)) (up to seven levels of phase A)"; timeout 900 python3 tools/fuzz_blas.py --cases 300 --max-tris 250000 --seed $((RN * 100 + 31)) 2>&1 | grep -v amdgpu.ids | tail -4;
  for sd in $((RN * 100 + 61)) $((RN * 100 + 62)); do echo "## fuzz_cull.py --cases 400 --seed $sd (fused and split form; every second case under a finite / negative / NaN far plane)"; timeout 600 python3 tools/fuzz_cull.py --cases 400 --seed $sd 2>&1 | grep -v amdgpu.ids | tail -4; done;
  for sd in $((RN * 100 + 71)); do echo "## fuzz_tlas.py --cases 400 --seed $sd (indexed build forced from 65 clusters)"; timeout 600 python3 tools/fuzz_tlas.py --cases 400 --seed $sd 2>&1 | grep -v amdgpu.ids | tail -4; done;
  for sd in $((RN * 100 + 81)) $((RN * 100 + 82)); do echo "## fuzz_trace.py --cases 250 --seed $sd (6 walks per scene; no status is accepted: deep rays take the second pass)"; timeout 900 python3 tools/fuzz_trace.py --cases 250 --seed $sd 2>&1 | grep -v amdgpu.ids | tail -4; done; } > $O/${R}_fuzz.log 2>&1
tail -3 $O/${R}_fuzz.log
