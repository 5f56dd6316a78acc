"""Random scenes through vd_trace / vd_trace_any (plain, indexed and prepared leaves; one launch and fanned out) against the oracle's
vd_ref_trace (shaders/utils/bvh.wgsl:35-123 restated): hit flags, distance BITS, instance and triangle ids equal; rays with zero / NaN /
axis-parallel directions, origins inside boxes and on faces, instances with negative / anisotropic scale, overlapping instances.
    python tools/fuzz_trace.py [--cases 60] [--seed 1]
`run(cases, seed, ctx)` is what tests/test_gpu_fuzz.py calls with a fixed seed."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref  # noqa: E402
from voidin_amd import abi, synth  # noqa: E402
from voidin_amd.runtime import Context, VoidinError  # noqa: E402


def run(cases, seed, ctx=None, log=print, sides=(48, 160, 700), only=None):
    """-> (mismatches, cases whose rays needed the second pass: the oracle's deepest stack is > 64 there).  `only`: the cases
    to trace (the others are generated, so that the random stream stays the campaign's, and skipped)."""
    rng = np.random.default_rng(seed)
    ctx = ctx or Context(0)
    bad = overflows = 0
    for case in range(cases):
        n_mesh = int(rng.integers(1, 5))
        srcs = []
        for _ in range(n_mesh):
            k = int(rng.integers(0, 3))
            if k == 0:
                srcs.append(synth.uv_sphere(float(rng.choice([0.5, 1.0, 3.0])), int(rng.integers(1, 8))))
            elif k == 1:
                srcs.append(synth.knot_mesh(int(rng.integers(8, 64)), int(rng.integers(4, 16)), seed=int(rng.integers(1 << 30))))
            else:
                srcs.append(synth.triangle_soup(int(rng.integers(1, 200)), seed=int(rng.integers(1 << 30))))
        V, I, B = [], [], []
        infos = np.zeros(n_mesh, dtype=abi.MESH_INFO)
        vo = bo = no = 0
        for k, (v, i) in enumerate(srcs):
            v = np.asarray(v, np.float32).reshape(-1, 3)
            nodes, idx = ref.bvh_build(v, i)
            infos[k]["min"], infos[k]["max"] = synth.mesh_bounds(v)
            infos[k]["index_count"], infos[k]["base_index"] = len(idx), bo
            infos[k]["vertex_offset"], infos[k]["bvh_index"] = vo, no
            V.append(v); I.append(idx); B.append(nodes)
            vo += len(v); bo += len(idx); no += len(nodes)
        V, I, B = np.concatenate(V), np.concatenate(I), np.concatenate(B)
        n_inst = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500]))
        ext = float(rng.choice([4.0, 40.0, 200.0]))
        inst = synth.instances(n_inst, n_mesh=n_mesh, seed=int(rng.integers(1 << 30)), extent=ext, scale_range=(0.2, float(rng.choice([1.0, 4.0]))))
        if rng.random() < 0.3:                          # mirror a few instances (negative determinant: backface culling flips)
            m = rng.integers(0, n_inst, max(1, n_inst // 10))
            T = inst["transform"].reshape(-1, 4, 4).copy()
            T[m, 0, :3] *= np.float32(-1)
            inst["transform"] = T.reshape(-1, 16)
            Td = T.astype(np.float64).transpose(0, 2, 1)
            inst["inv_transform"] = np.linalg.inv(Td).transpose(0, 2, 1).reshape(-1, 16).astype(np.float32)
        tl = ref.tlas_build(inst, infos)
        side = int(rng.choice(sides))          # 700 x 700 = 490 k rays: enough for a call to fan out
        cam = synth.camera_uniform(eye=(float(rng.random() - 0.5) * ext, float(rng.random() - 0.5) * ext, ext * float(rng.choice([0.0, 0.6, 1.5]))),
                                   yaw_deg=float(rng.random() * 360), pitch_deg=float(rng.random() * 60 - 30))
        rays = synth.primary_rays(cam, side, side)
        k = max(8, len(rays) // 200)
        sel = rng.integers(0, len(rays), k)
        rays["dir"][sel[: k // 4]] = rng.choice(np.array([0.0, 1.0, -1.0], np.float32), (k // 4, 3))                 # axis-parallel, zero
        rays["dir"][sel[k // 4: k // 2], rng.integers(0, 3, k // 2 - k // 4)] = np.float32(np.nan)
        rays["eye"][sel[k // 2:]] = inst["transform"][rng.integers(0, n_inst, k - k // 2)][:, 12:15]              # from instance centres
        scene = (tl, inst, infos, B, V, I)
        if only is not None and case not in only:
            continue
        want, want_stack = ref.trace(scene, rays, threads=16)
        if want_stack > 64:
            overflows += 1
        import torch
        ds = ctx.device_scene(scene)
        d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
        d_any = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
        acc = ctx.trace_prepare(ds)
        for fan in (1, 3):
            for mode in ("plain", "indexed", "prepared"):
                ctx.set_option("trace.fan", fan)
                ctx.set_option("trace.auto_prepare", 0 if mode == "indexed" else None)
                try:
                    if mode == "prepared":
                        ctx.trace_prepared_dev(acc, d_rays, len(rays), d_hits); ctx.trace_any_prepared_dev(acc, d_rays, len(rays), d_any)
                    else:
                        ctx.trace_dev(ds, d_rays, len(rays), d_hits); ctx.trace_any_dev(ds, d_rays, len(rays), d_any)
                except VoidinError as e:
                    # no status is acceptable: a ray that runs out of the 128 entries a lane holds is walked again with its stack in
                    # global memory (trace.hip, launch_trace) - round 5 still reported VD_ERR_STACK_OVERFLOW for such scenes
                    bad += 1
                    log(f"case {case}: {n_inst} instances, {len(rays)} rays, fan {fan} {mode}: {e} (oracle's deepest stack {want_stack})")
                    continue
                got = d_hits.cpu().numpy()[: len(rays) * 16].view(abi.HIT)
                ok = got.tobytes() == np.ascontiguousarray(want).tobytes() and np.array_equal(d_any.cpu().numpy().astype(np.uint32), want["hit"])
                if not ok:
                    bad += 1
                    log(f"case {case}: {n_inst} instances, {len(rays)} rays, fan {fan} {mode}: DIFFERS "
                          f"(hit flags {int((got['hit'] != want['hit']).sum())}, dist bits {int((got['dist'].view(np.uint32) != want['dist'].view(np.uint32)).sum())}, "
                          f"ids {int(((got['instance'] != want['instance']) | (got['triangle'] != want['triangle'])).sum())})")
        ctx.set_option("trace.fan", None); ctx.set_option("trace.auto_prepare", None)
        acc.close()
    return bad, overflows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    bad, overflows = run(args.cases, args.seed, log=lambda m: print(m, flush=True))
    print(f"{args.cases} cases x 6 walks, {bad} mismatches; {overflows} cases with a ray deeper than 64 entries on the oracle's side")
    sys.exit(1 if bad else 0)
