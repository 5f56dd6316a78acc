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
ap.add_argument("--tlas", type=int, default=32768)
ap.add_argument("--blas-only", action="store_true", help="stop after the BLAS builds (tools/gpu_pmc_bvh.sh)")
args = ap.parse_args()

ctx = Context(0)
ctx.set_timing(True)
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
# (parity against the oracle lives in tests/test_gpu_blas.py and tests/test_gpu_full_size.py; this tool only times)
if args.blas_only:
    sys.exit(0)

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

# ---- traversal: bvh_gpu.rs-shaped scene (one big mesh + many instances), primary rays ----
tv, ti = synth.knot_mesh(512, 128)                        # 131k triangles
nodes_b, idx_b = ctx.bvh_build(tv, ti)
infos = np.zeros(1, dtype=abi.MESH_INFO)
infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(tv)
infos[0]["index_count"] = len(idx_b)
inst_t = synth.instances(2000, n_mesh=1, seed=synth.SEED_BASE + 8, extent=120.0, scale_range=(0.5, 2.0))
tl = ctx.tlas_build(inst_t, infos)
cam_t = synth.camera_uniform(eye=(0, 2.5, 90), pitch_deg=0)
W = 1024
rays = synth.primary_rays(cam_t, W, W)
scene_np = (tl, inst_t, infos, nodes_b, tv, idx_b)
d_arrs = [ctx.upload(np.ascontiguousarray(a)) for a in (tl, inst_t, infos, nodes_b, tv.reshape(-1), idx_b)]
s = abi.TraceScene()
s.tlas_nodes, s.n_tlas_nodes = d_arrs[0].data_ptr(), len(tl)
s.instances, s.n_instances = d_arrs[1].data_ptr(), len(inst_t)
s.meshes, s.n_meshes = d_arrs[2].data_ptr(), 1
s.bvh_nodes, s.n_bvh_nodes = d_arrs[3].data_ptr(), len(nodes_b)
s.vertices, s.n_vertices = d_arrs[4].data_ptr(), len(tv)
s.indices, s.n_indices = d_arrs[5].data_ptr(), len(idx_b)
import ctypes as C  # noqa: E402
d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
for r in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = ctx.lib.vd_trace_dev(ctx.h, C.byref(s), d_rays.data_ptr(), len(rays), d_hits.data_ptr())
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    assert rc == 0, ctx.lib.vd_last_error(ctx.h)
hits = d_hits.cpu().numpy()[: len(rays) * 16].view(abi.HIT)
print(f"trace: {len(rays)} rays, {dt*1e3:.2f} ms = {len(rays)/dt/1e6:.1f} Mrays/s, hit fraction {hits['hit'].mean():.3f}", flush=True)
d_any = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
for r in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = ctx.lib.vd_trace_any_dev(ctx.h, C.byref(s), d_rays.data_ptr(), len(rays), d_any.data_ptr())
    torch.cuda.synchronize(); dta = time.perf_counter() - t
assert rc == 0 and np.array_equal(d_any.cpu().numpy().astype(np.uint32), hits["hit"])
print(f"trace_any (occlusion): {dta*1e3:.2f} ms = {len(rays)/dta/1e6:.1f} Mrays/s (flags equal vd_trace's)", flush=True)

# the same scene behind vd_trace_prepare_dev (de-indexed leaf triangles; results must not change)
ds_t = ctx.device_scene(scene_np)
acc = ctx.trace_prepare(ds_t)
d_hits_p = ctx.empty(len(rays) * 16)
for r in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.trace_prepared_dev(acc, d_rays, len(rays), d_hits_p)
    torch.cuda.synchronize(); dtp = time.perf_counter() - t
assert np.array_equal(d_hits_p.cpu().numpy()[: len(rays) * 16], d_hits.cpu().numpy()[: len(rays) * 16])
d_any_p = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
for r in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.trace_any_prepared_dev(acc, d_rays, len(rays), d_any_p)
    torch.cuda.synchronize(); dtap = time.perf_counter() - t
assert torch.equal(d_any_p, d_any)
print(f"trace, prepared leaves: {dtp*1e3:.2f} ms = {len(rays)/dtp/1e6:.1f} Mrays/s (hits bit-equal); "
      f"occlusion {dtap*1e3:.2f} ms = {len(rays)/dtap/1e6:.1f} Mrays/s (flags equal)", flush=True)
acc.close()

# ---- the reference's own harness shape (src/bin/bvh_gpu.rs:107-131, camera at (0, 2.5, 15): bvh_gpu.rs:221) ----
inst2, infos2, B, V, I = synth.harness_scene(ctx.bvh_build)
tl2 = ctx.tlas_build(inst2, infos2)
rays2 = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 15), pitch_deg=0), 2048, 2048)
ds2 = ctx.device_scene((tl2, inst2, infos2, B, V, I))
d_r2, d_h2 = ctx.upload(rays2), ctx.empty(len(rays2) * 16)
ctx.set_timing(True)
best = 1e9
for r in range(3):
    ctx.trace_dev(ds2, d_r2, len(rays2), d_h2); best = min(best, ctx.last_gpu_ms())
h2 = d_h2.cpu().numpy()[: len(rays2) * 16].view(abi.HIT)
print(f"trace, bvh_gpu.rs-shaped scene ({len(I)//3} triangles, 5 instances, {len(rays2)} primary rays): {best:.2f} ms = "
      f"{len(rays2)/best/1e3:.0f} Mrays/s, hit fraction {h2['hit'].mean():.3f}", flush=True)
