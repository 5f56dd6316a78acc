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
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/bvh_pmc_$c -o bvh -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 1 --blas-only > $O/bvh_pmc_${c}_stdout.log 2>&1
done
R=$R python3 - <<'PY'
import csv, glob, json, os, re
R = os.environ['R']
builds = 2                                       # bench_bvh.py --reps 1 = one warm-up build + one timed build
tot = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'gpurun_out/round/bvh_pmc_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
            if not (k.startswith('a_') or k.startswith('blas_') or k.startswith('c_')):
                continue
            d = tot.setdefault(k, {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0, 'launches': 0})
            d[c] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE':
                d['launches'] += 1
out = {'note': 'per BUILD of the 8 388 608-triangle knot mesh (totals over 2 builds / 2); MB = 1e6 B; read = 2 x FETCH_SIZE (gfx950 '
               'correction), write = WRITE_SIZE; rocprofv3 --pmc, one counter per pass', 'kernels': {}}
s_r = s_w = 0.0
for k, d in sorted(tot.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
    rd, wr = 2 * d['FETCH_SIZE'] * 1024 / builds / 1e6, d['WRITE_SIZE'] * 1024 / builds / 1e6
    out['kernels'][k] = {'launches_per_build': d['launches'] // builds, 'read_MB': round(rd, 1), 'write_MB': round(wr, 1)}
    s_r += rd; s_w += wr
out['total'] = {'read_MB': round(s_r, 1), 'write_MB': round(s_w, 1), 'sum_MB': round(s_r + s_w, 1)}
json.dump(out, open(f'gpurun_out/round/{R}_bvh_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/bvh_pmc_FETCH_SIZE $O/bvh_pmc_WRITE_SIZE
