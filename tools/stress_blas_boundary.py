"""Repeat BLAS builds whose level boundaries run on several workgroups (the last-workgroup hand-over of a_boundary_kernel) and whose
levels are queued one ahead of the host, and compare every run with the first one bit for bit: one 2.1 M-triangle mesh (levels of up to
~500 segments = 4 workgroups), and a batch of 3000 meshes of 2.5-6 k triangles (a first level of 3000 segments = 24 workgroups).
Usage (GPU box): python tools/stress_blas_boundary.py [--iters 40]"""
import argparse, os, sys, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth
from voidin_amd.runtime import Context

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=40); args = ap.parse_args()
ctx = Context(0)
bad = 0
v, i = synth.knot_mesh(1024, 1024)
nt = len(i) // 3
d_v, d_n = ctx.upload(v), ctx.empty(2 * nt * 32)
ref = None
for it in range(args.iters):
    d_i = ctx.upload(i)
    nn = ctx.bvh_build_dev(d_v, len(v), d_i, nt, d_n, 2 * nt)
    key = (nn, zlib.crc32(d_n[: nn * 32].cpu().numpy().tobytes()), zlib.crc32(d_i.cpu().numpy().tobytes()))
    if ref is None: ref = key
    elif key != ref: bad += 1; print(f"single mesh: run {it} differs: {key} vs {ref}")
print(f"single {nt}-triangle mesh: {args.iters} builds, nodes {ref[0]}", flush=True)
# a batch whose FIRST level already has thousands of segments
rng = np.random.default_rng(11)
meshes = [synth.knot_mesh(int(u), 36) for u in rng.integers(36, 84, size=3000)]
ref = None
for it in range(max(4, args.iters // 4)):
    nodes, per = ctx.bvh_build_batch(meshes)
    key = (zlib.crc32(nodes.tobytes()),) + tuple((f, c, zlib.crc32(ix.tobytes())) for (f, c, ix) in per)
    if ref is None: ref = key
    elif key != ref: bad += 1; print(f"batch: run {it} differs in {sum(a != b for a, b in zip(key, ref))} meshes")
print(f"batch of {len(meshes)} meshes ({sum(len(m[1]) // 3 for m in meshes)} triangles): {max(4, args.iters // 4)} builds", flush=True)
print("ALL EQUAL" if bad == 0 else f"{bad} RUNS DIFFER")
sys.exit(1 if bad else 0)
