"""CPU simulation of a nearest-neighbour table for the TLAS chain (tlas.rs:56-105): hit rate against the refresh period,
every hit checked against the true answer.   python tools/tlas_nn_table_sim.py   (NN_SIM_STORE=0: batches only)"""
import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
from oracle import np_restate as npr
from voidin_amd import synth
F=np.float32
import os
STORE = os.environ.get('NN_SIM_STORE', '1') != '0'     # 0: the table is filled by the refresh batches only
def run(n, K, extent):
    meshes=synth.mesh_infos(); inst=synth.instances(n, seed=synth.SEED_BASE+6, extent=extent)
    bmin=np.zeros((2*n+1,3),F); bmax=np.zeros((2*n+1,3),F)
    bmin[1:n+1],bmax[1:n+1]=npr.tlas_leaf_bounds(inst,meshes)
    ni=np.arange(1,n+1)
    alive=np.zeros(2*n+1,bool); alive[1:n+1]=True
    nn=np.full(2*n+1,-1,np.int64); uniq=np.zeros(2*n+1,bool)   # by node id: neighbour NODE id
    def areas(t,cnt):
        sel=ni[:cnt]
        mn=np.minimum(bmin[ni[t]],bmin[sel]); mx=np.maximum(bmax[ni[t]],bmax[sel]); d=mx-mn
        a=((d[:,0]*d[:,1]+d[:,0]*d[:,2])+d[:,1]*d[:,2])*F(2)
        if t<cnt: a[t]=np.inf
        return a
    def best(cnt,t):
        a=areas(t,cnt); k=int(np.argmin(a))
        if not a[k]<F(1e30): return t,False
        return k, int((a==a[k]).sum())==1
    def batch(cnt):
        for s in range(cnt):
            x=ni[s]
            if nn[x]>=0 and alive[nn[x]] and uniq[x]: continue
            k,u=best(cnt,s); nn[x]=ni[k]; uniq[x]=u
    stats=dict(q=0,hit=0,after_merge=0,miss_dead=0,miss_none=0,miss_tie=0)
    def cached_best(cnt,t,after_merge):
        x=ni[t]
        if after_merge: stats['after_merge']+=1
        else:
            stats['q']+=1
            if nn[x]>=0 and uniq[x] and alive[nn[x]]:
                stats['hit']+=1
                # verify exactness
                k,u=best(cnt,t); assert ni[k]==nn[x], "cache wrong"
                return k
            if nn[x]<0: stats['miss_none']+=1
            elif not uniq[x]: stats['miss_tie']+=1
            else: stats['miss_dead']+=1
        k,u=best(cnt,t)
        if t<cnt and STORE: nn[x]=ni[k]; uniq[x]=u
        return k
    cnt,used,a=n,n+1,0
    batch(cnt); merges=0
    b=cached_best(cnt,a,False)
    while cnt>1:
        c=cached_best(cnt,b,False)
        if a==c:
            ia,ib=ni[a],ni[b]
            bmin[used]=np.minimum(bmin[ia],bmin[ib]); bmax[used]=np.maximum(bmax[ia],bmax[ib])
            alive[ia]=alive[ib]=False; alive[used]=True
            ni[a]=used; used+=1; ni[b]=ni[cnt-1]; cnt-=1; merges+=1
            if K and merges%K==0: batch(cnt)
            b=cached_best(cnt,a,True)
        else: a,b=b,c
    return stats
for n,K in ((4096,0),(4096,256),(4096,64),(4096,16),(4096,4)):
    t=time.time(); s=run(n,K,300.0)
    print(n,K,s,"hit rate of non-post-merge queries %.2f"%(s['hit']/max(s['q'],1)), "%.1fs"%(time.time()-t),flush=True)
