#!/bin/bash
# HBM traffic of the 8.4 M-triangle BLAS build, per kernel: rocprofv3 --pmc, one counter per pass (FETCH_SIZE, WRITE_SIZE
# in KB; gfx950 FETCH_SIZE counts half the bytes of wide coalesced loads - MI355X_MICROARCH.md, HBM section).
#   gpurun -- 'bash tools/gpu_pmc_bvh.sh r02'   ->  gpurun_out/round/rNN_bvh_pmc.json
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${1:-r02}
O=gpurun_out/round
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE VALUBusy; do       # VALUBusy: derived (% of cycles the VALUs issue); a pass that is not available leaves no file
  rocprofv3 --pmc $c --output-format csv -d $O/bvh_pmc_$c -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 1 --blas-only > $O/bvh_pmc_${c}_stdout.log 2>&1
done
# durations for the GB/s column: a kernel trace of the same command (its own run: never together with --pmc)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bvh_pmc_kt -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 1 --blas-only > $O/bvh_pmc_kt_stdout.log 2>&1
R=$R python3 - <<'PY'
import csv, glob, json, os, re
R = os.environ['R']
builds = 2                                       # bench_bvh.py --reps 1 = one warm-up build + one timed build
tot = {}
def kname(n):
    return re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')
dur = {}
for f in glob.glob('gpurun_out/round/bvh_pmc_kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[kname(r['Name'])] = float(r['TotalDurationNs']) / builds / 1e6      # ms per build
valu = {}
for f in glob.glob('gpurun_out/round/bvh_pmc_VALUBusy/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'VALUBusy':
            valu.setdefault(kname(r['Kernel_Name']), []).append(float(r['Counter_Value']))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'gpurun_out/round/bvh_pmc_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            k = kname(r['Kernel_Name'])
            if not (k.startswith('a_') or k.startswith('blas_') or k.startswith('c_')):
                continue
            d = tot.setdefault(k, {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0, 'launches': 0})
            d[c] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE':
                d['launches'] += 1
out = {'note': 'per BUILD of the 8 388 608-triangle knot mesh (totals over 2 builds / 2); MB = 1e6 B; read = 2 x FETCH_SIZE (gfx950 '
               'correction), write = WRITE_SIZE; rocprofv3 --pmc, one counter per pass; ms_per_build / GB_per_s: durations from a separate --kernel-trace --stats run of the same command; VALUBusy: its own pass', 'kernels': {}}
s_r = s_w = 0.0
for k, d in sorted(tot.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
    rd, wr = 2 * d['FETCH_SIZE'] * 1024 / builds / 1e6, d['WRITE_SIZE'] * 1024 / builds / 1e6
    out['kernels'][k] = {'launches_per_build': d['launches'] // builds, 'read_MB': round(rd, 1), 'write_MB': round(wr, 1)}
    if k in dur and dur[k] > 0:
        out['kernels'][k]['ms_per_build'] = round(dur[k], 3)
        out['kernels'][k]['GB_per_s'] = round((rd + wr) / dur[k], 1)          # MB / ms = GB/s
    if k in valu:
        out['kernels'][k]['VALUBusy_pct_mean_over_launches'] = round(sum(valu[k]) / len(valu[k]), 1)
    s_r += rd; s_w += wr
out['total'] = {'read_MB': round(s_r, 1), 'write_MB': round(s_w, 1), 'sum_MB': round(s_r + s_w, 1)}
json.dump(out, open(f'gpurun_out/round/{R}_bvh_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/bvh_pmc_FETCH_SIZE $O/bvh_pmc_WRITE_SIZE $O/bvh_pmc_VALUBusy $O/bvh_pmc_kt
