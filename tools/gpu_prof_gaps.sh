#!/bin/bash
# kernel-to-kernel gaps of one BLAS build (rocprofv3 kernel trace)
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_gaps; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o g -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 1 --tlas 1000 > $O/stdout.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_gaps/**/g_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last build: find last blas_precompute
starts=[i for i,r in enumerate(rows) if 'blas_precompute' in r['Kernel_Name']]
ends=[i for i,r in enumerate(rows) if 'c_permute' in r['Kernel_Name']]
idx,end=max(zip(starts,ends),key=lambda se:int(rows[se[0]]['End_Timestamp'])-int(rows[se[0]]['Start_Timestamp']))
seq=rows[idx:end+1]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seq)
span=int(seq[-1]['End_Timestamp'])-int(seq[0]['Start_Timestamp'])
gaps=[int(seq[i+1]['Start_Timestamp'])-int(seq[i]['End_Timestamp']) for i in range(len(seq)-1)]
gs=sorted(gaps)
print(f"kernels {len(seq)} span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms gaps {sum(gaps)/1e6:.2f} ms; gap median {gs[len(gs)//2]/1e3:.1f} us p90 {gs[int(len(gs)*.9)]/1e3:.1f} us max {gs[-1]/1e3:.1f} us")
agg=collections.defaultdict(lambda:[0,0])
for r in seq:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:30]; agg[n][0]+=1; agg[n][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for n,(c,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:16]: print(f"  {n:32s} calls {c:5d} total {t/1e6:7.2f} ms avg {t/c/1e3:8.1f} us")
# per level of phase A (a level starts at a_seg_begin_kernel): span, busy, and where the busy time goes
lv=[i for i,r in enumerate(seq) if 'a_seg_begin' in r['Kernel_Name']]
ends_a=[i for i,r in enumerate(seq) if 'a_level_swap' in r['Kernel_Name']]
lv_all=[]
for k,i1 in enumerate(ends_a):
    i0=(ends_a[k-1]+1) if k else next(i for i,r in enumerate(seq) if 'c_root' in r['Kernel_Name'])+1
    lv_all.append(i0)
lv=lv_all
print("level  span_ms busy_ms  count+scan+ranks+apply  bits  bin+eval  child  big-tier(LDS)  other   (phase A)")
for k,(i0,i1) in enumerate(zip(lv,ends_a)):
    part=seq[i0:i1+1]
    span_l=(int(part[-1]['End_Timestamp'])-int(part[0]['Start_Timestamp']))/1e6
    d=collections.defaultdict(float)
    for r in part:
        n=r['Kernel_Name']; t=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
        key='big' if 'blas_mid_kernel' in n else 'round' if any(x in n for x in ('a_count','a_scan_kernel','a_ranks','a_apply')) else 'bits' if 'a_bits' in n else 'bin' if ('a_bin' in n or 'a_eval' in n) else 'child' if 'a_child' in n else 'other'
        d[key]+=t
    nxt=int(seq[lv[k+1]]['Start_Timestamp']) if k+1<len(lv) else int(seq[i1+1]['Start_Timestamp'])
    print(f"{k:5d} {span_l:8.3f} {sum(d.values()):7.3f}  {d['round']:8.3f} {d['bits']:12.3f} {d['bin']:8.3f} {d['child']:7.3f} {d['big']:9.3f} {d['other']:10.3f}   gap to next level {(nxt-int(part[-1]['End_Timestamp']))/1e3:.1f} us")
# the round kernels by round index (levels 0..11, all triangles active): where inside a level the time goes
for kn in ('a_count','a_scan_kernel','a_ranks','a_apply'):
    per=collections.defaultdict(list)
    for k,(i0,i1) in enumerate(zip(lv,ends_a)):
        if k>11: break
        calls=[r for r in seq[i0:i1+1] if kn in r['Kernel_Name']]
        for c,r in enumerate(calls): per[c].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    if per: print(f"{kn:14s} us by call within the level:", ' '.join(f"{sum(v)/len(v):.1f}" for c,v in sorted(per.items())))
big=[(g,seq[i]['Kernel_Name'][:40],seq[i+1]['Kernel_Name'][:40]) for i,g in enumerate(gaps) if g>30000]
print("gaps > 30 us:",len(big), "sum %.2f ms"%(sum(b[0] for b in big)/1e6))
for b in sorted(big, reverse=True)[:10]: print("  %.1f us after %s before %s"%(b[0]/1e3,b[1],b[2]))
PY
find $O -name "*kernel_trace.csv" -delete
