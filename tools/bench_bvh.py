"""Timing of the BVH side of the path on one GPU: BLAS build (Mprims/s), TLAS build / refit, trace.
Usage: python tools/bench_bvh.py [--tris-u 2048 --tris-v 2048] [--verify]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import abi, synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--u", type=int, default=2048)
ap.add_argument("--v", type=int, default=2048)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--verify", action="store_true")
ap.add_argument("--tlas", type=int, default=32768)
args = ap.parse_args()

ctx = Context(0)
t0 = time.time()
v, i = synth.knot_mesh(args.u, args.v)
n_tri = len(i) // 3
print(f"mesh: {n_tri} tris, {len(v)} verts (gen {time.time()-t0:.1f}s)", flush=True)
d_v = ctx.upload(v)
d_n = ctx.empty(2 * n_tri * 32)
times = []
for r in range(args.reps + 1):
    d_i = ctx.upload(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n_nodes = ctx.bvh_build_dev(d_v, len(v), d_i, n_tri, d_n, 2 * n_tri)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    if r:
        times.append(dt)
    print(f"  build {r}: {dt*1e3:.1f} ms  ({n_tri/dt/1e6:.1f} Mprims/s), nodes {n_nodes}, gpu_ms {ctx.last_gpu_ms():.1f}", flush=True)
best = min(times)
print(f"BLAS build: {n_tri} prims, best {best*1e3:.1f} ms = {n_tri/best/1e6:.1f} Mprims/s")
if args.verify:
    from oracle import ref
    t = time.time()
    wn, wi = ref.bvh_build(v, i)
    dt = time.time() - t
    nodes = d_n.cpu().numpy()[: n_nodes * 32].view(abi.BVH_NODE)
    idx = d_i.cpu().numpy().view(np.uint32)[: 3 * n_tri]
    ok = len(nodes) == len(wn) and all(np.array_equal(nodes[f], wn[f]) for f in wn.dtype.names) and np.array_equal(idx, wi)
    print(f"oracle: {dt:.1f}s = {n_tri/dt/1e6:.3f} Mprims/s (1 core); bit-exact: {ok}")

# TLAS
meshes = synth.mesh_infos()
for n in [1000, args.tlas]:
    inst = synth.instances(n, seed=synth.SEED_BASE + 6, extent=300.0)
    d_inst, d_m = ctx.upload(inst), ctx.upload(meshes)
    d_t = ctx.empty((2 * n + 1) * 32)
    for r in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.tlas_build_dev(d_inst, n, d_m, len(meshes), d_t)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"TLAS build n={n}: {dt*1e3:.2f} ms", flush=True)
    for r in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.tlas_refit_dev(d_inst, n, d_m, len(meshes), d_t)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"TLAS refit n={n}: {dt*1e3:.3f} ms (gpu {ctx.last_gpu_ms():.3f} ms)", flush=True)
    if args.verify and n <= 4096:
        from oracle import ref
        t = time.time(); w = ref.tlas_build(inst, meshes); dt = time.time() - t
        got = d_t.cpu().numpy()[: (2 * n + 1) * 32].view(abi.TLAS_NODE)
        print(f"  oracle TLAS build {dt*1e3:.1f} ms; match {got.tobytes()==w.tobytes()}")
