"""Do two independent BLAS builds overlap on the device?  (Would running the two halves of a level's segments on two streams
recover the ramp-up / tail of the 8 k-workgroup round kernels?)  One 4.2 M-triangle build alone, two in sequence on one
context, two at once on two contexts (= two streams) from two host threads.
    python tools/blas_concurrent_probe.py"""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import abi, synth
from voidin_amd.runtime import Context

u, v = 2048, 1024
verts, idx = synth.knot_mesh(u, v)
n_tri = len(idx) // 3
ctxs = [Context(0, use_torch_stream=False), Context(0, use_torch_stream=False)]
bufs = []
for c in ctxs:
    d_v = c.upload(np.ascontiguousarray(verts, dtype=np.float32))
    d_i0 = c.upload(np.ascontiguousarray(idx, dtype=np.uint32))
    d_i = d_i0.clone()
    d_n = c.empty(2 * n_tri * 32)
    bufs.append((d_v, d_i0, d_i, d_n))
torch.cuda.synchronize()


def build(k):
    c = ctxs[k]; d_v, d_i0, d_i, d_n = bufs[k]
    d_i.copy_(d_i0); torch.cuda.synchronize()
    t = time.perf_counter()
    c.bvh_build_dev(d_v, len(verts), d_i, n_tri, d_n, 2 * n_tri)
    return time.perf_counter() - t

for _ in range(2): build(0); build(1)
alone = min(build(0) for _ in range(3))
seq = []
for _ in range(3):
    bufs[0][2].copy_(bufs[0][1]); bufs[1][2].copy_(bufs[1][1]); torch.cuda.synchronize()
    t = time.perf_counter()
    ctxs[0].bvh_build_dev(bufs[0][0], len(verts), bufs[0][2], n_tri, bufs[0][3], 2 * n_tri)
    ctxs[1].bvh_build_dev(bufs[1][0], len(verts), bufs[1][2], n_tri, bufs[1][3], 2 * n_tri)
    seq.append(time.perf_counter() - t)
par = []
for _ in range(3):
    bufs[0][2].copy_(bufs[0][1]); bufs[1][2].copy_(bufs[1][1]); torch.cuda.synchronize()
    th = [threading.Thread(target=lambda k=k: ctxs[k].bvh_build_dev(bufs[k][0], len(verts), bufs[k][2], n_tri, bufs[k][3], 2 * n_tri)) for k in (0, 1)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    par.append(time.perf_counter() - t)
print(f"{n_tri} triangles per build: alone {alone * 1e3:.2f} ms; two in sequence {min(seq) * 1e3:.2f} ms; two at once on two streams {min(par) * 1e3:.2f} ms "
      f"= {2 * n_tri / min(par) / 1e6:.1f} Mprims/s aggregate (sequence: {2 * n_tri / min(seq) / 1e6:.1f})")
