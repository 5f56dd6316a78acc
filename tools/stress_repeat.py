"""Repeat the fence-free / arrival-counter paths many times and compare every run with the first one bit for bit
(vd_cull_compact split form, vd_expand_mask_dev multi-shard, vd_tlas_refit_dev, vd_bvh_build_dev).  A race shows up as a
run that differs.  Usage (GPU box): python tools/stress_repeat.py [--iters 300]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth
from voidin_amd import dist as vdist
from voidin_amd.runtime import Context

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=300); args = ap.parse_args()
ctx = Context(0)
cam, meshes = synth.camera_uniform(), synth.mesh_infos()
d_m = ctx.upload(meshes)
bad = 0
for n in (1_048_576, 3_000_001, 10_000_000):
    inst = synth.instances(n, seed=synth.SEED_BASE + 3, with_inverse=False)
    d_i = ctx.upload(inst)
    d_o, d_c = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
    ref_o = ref_c = None
    for it in range(args.iters):
        d_o.zero_()
        ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_o, d_c)
        c = int(d_c[0].item())
        if ref_o is None: ref_o, ref_c = d_o[: c * 20].clone(), c
        elif c != ref_c or not torch.equal(d_o[: c * 20], ref_o): bad += 1; print(f"cull_compact n={n}: run {it} differs (count {c} vs {ref_c})")
    print(f"cull_compact n={n}: {args.iters} runs, count {ref_c}", flush=True)
    if n == 3_000_001:      # multi-shard expansion, fast (shard % 4 == 0) and general paths
        for shards in (4, 3):
            S = vdist.shard_size(n, shards); wps = vdist.mask_words(S)
            d_mask = torch.zeros(wps * shards, dtype=torch.int64, device="cuda")
            ids = np.zeros(S * shards, np.uint8); ids[:n] = inst["mesh"]
            for r in range(shards):
                lo, hi = vdist.shard_range(n, r, shards)
                ctx.cull_mask_dev(cam, d_m, len(meshes), ctx.upload(inst[lo:hi]), hi - lo, d_mask[r * wps:])
            d_ids = ctx.upload(ids)
            for it in range(args.iters):
                d_o.zero_()
                ctx.expand_mask_dev(d_mask, n, S, d_ids, d_m, len(meshes), d_o, d_c, id_bytes=1)
                c = int(d_c[0].item())
                if c != ref_c or not torch.equal(d_o[: c * 20], ref_o): bad += 1; print(f"expand shards={shards}: run {it} differs")
            print(f"expand_mask shards={shards}: {args.iters} runs equal to cull_compact", flush=True)
    del d_i, d_o
# TLAS refit
n = 32768
tinst = synth.instances(n, seed=synth.SEED_BASE + 6, extent=300.0)
d_ti, d_t = ctx.upload(tinst), ctx.empty((2 * n + 1) * 32)
ctx.tlas_build_dev(d_ti, n, d_m, len(meshes), d_t); torch.cuda.synchronize()
built = d_t.clone()
for it in range(args.iters):
    ctx.tlas_refit_dev(d_ti, n, d_m, len(meshes), d_t)
    if not torch.equal(d_t, built): bad += 1; print(f"tlas_refit: run {it} differs from build")
print(f"tlas_refit n={n}: {args.iters} runs == build", flush=True)
# BLAS build (phase A arrival patterns, mid tier, phase B)
v, i = synth.knot_mesh(512, 256)
n_tri = len(i) // 3
d_v, d_n = ctx.upload(v), ctx.empty(2 * n_tri * 32)
ref_n = ref_i = None
for it in range(max(10, args.iters // 10)):
    d_idx = ctx.upload(i)
    nn = ctx.bvh_build_dev(d_v, len(v), d_idx, n_tri, d_n, 2 * n_tri); torch.cuda.synchronize()
    if ref_n is None: ref_n, ref_i, ref_nn = d_n[: nn * 32].clone(), d_idx.clone(), nn
    elif nn != ref_nn or not torch.equal(d_n[: nn * 32], ref_n) or not torch.equal(d_idx, ref_i): bad += 1; print(f"bvh_build: run {it} differs")
print(f"bvh_build {n_tri} tris: {max(10, args.iters // 10)} runs identical", flush=True)
# traversal: persistent waves draw rays from a counter; results must not depend on which lane got which ray
from voidin_amd import abi
inst2, infos2, B2, V2, I2 = synth.harness_scene(ctx.bvh_build, big=(256, 64), small_res=24)
tl2 = ctx.tlas_build(inst2, infos2)
rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 15), pitch_deg=0), 509, 383)      # ragged ray count
ds = ctx.device_scene((tl2, inst2, infos2, B2, V2, I2))
d_r, d_h = ctx.upload(rays), ctx.empty(len(rays) * 16)
d_a = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
ref_h = ref_a = None
for it in range(args.iters):
    d_h.zero_(); d_a.zero_()
    ctx.trace_dev(ds, d_r, len(rays), d_h); ctx.trace_any_dev(ds, d_r, len(rays), d_a)
    if ref_h is None: ref_h, ref_a = d_h.clone(), d_a.clone()
    elif not torch.equal(d_h, ref_h) or not torch.equal(d_a, ref_a): bad += 1; print(f"trace: run {it} differs")
hits = ref_h.cpu().numpy()[: len(rays) * 16].view(abi.HIT)
assert np.array_equal(ref_a.cpu().numpy().astype(np.uint32), hits["hit"]) and 0 < hits["hit"].sum() < len(rays)
print(f"trace + trace_any, {len(rays)} rays: {args.iters} runs identical, occlusion flags == closest-hit flags", flush=True)
print("STRESS", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
