"""Wall time of BLAS builds of small meshes (the sizes the reference's scenes load): one 15 k-triangle mesh alone, device arrays;
64 of them in one batch.  python tools/blas_small_build_ab.py   (VOIDIN_HIP_LIB selects the library)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import abi, synth
from voidin_amd.runtime import Context
ctx = Context(0)
for (u, v) in ((88, 88), (256, 256), (724, 724)):
    mv, mi = synth.knot_mesh(u, v)
    nt = len(mi) // 3
    d_v, d_n = ctx.upload(mv), ctx.empty(2 * nt * 32)
    ts = []
    for r in range(12):
        d_i = ctx.upload(mi); torch.cuda.synchronize()
        t = time.perf_counter(); ctx.bvh_build_dev(d_v, len(mv), d_i, nt, d_n, 2 * nt); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    st = ctx.bvh_last_build_stats()
    print(f"{nt} triangles alone: best {min(ts[2:]) * 1e3:.3f} ms, median {np.median(ts[2:]) * 1e3:.3f} ms; last build: {st}", flush=True)
