"""Random scenes through vd_cull_emit / vd_cull_compact (fused and split form) against the oracle (shaders/emit_draws.wgsl:13-64
restated): transforms with NaN / inf / zero / negative scale, mesh ids past the table, cameras with jitter, tiny and huge
instances, and - half of the cases - a FINITE far plane and a moved near plane (the third return of is_visible, emit_draws.wgsl:28-30,
which the reference's own zfar = +inf never takes) - command buffers and survivor lists byte for byte.
    python tools/fuzz_cull.py [--cases 200] [--seed 1]
`run(cases, seed, ctx)` is what tests/test_gpu_fuzz.py calls with a fixed seed."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref  # noqa: E402
from voidin_amd import synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402


def run(cases, seed, ctx=None, log=print, sizes=(1, 63, 64, 65, 255, 256, 257, 1023, 1025, 4097, 50_000, 200_003)):
    """-> (mismatches, cases in which the far-plane test changed the survivor set)"""
    rng = np.random.default_rng(seed)
    ctx = ctx or Context(0)
    bad = far_cases = 0
    for case in range(cases):
        n = int(rng.choice(sizes))
        n_mesh = int(rng.choice([1, 3, 16, 255, 256, 257, 700]))
        meshes = synth.mesh_infos(n_mesh, seed=int(rng.integers(1 << 30)))
        inst = synth.instances(n, n_mesh=n_mesh, seed=int(rng.integers(1 << 30)), extent=float(rng.choice([10.0, 600.0, 2000.0])),
                               scale_range=(0.01, float(rng.choice([0.3, 4.0, 50.0]))), with_inverse=False)
        k = max(1, n // 40)
        t = inst["transform"]
        t[rng.integers(0, n, k), rng.integers(0, 16, k)] = rng.choice(np.array([np.nan, np.inf, -np.inf, 0.0, -1.0, 1e30, 1e-40], np.float32), k)
        inst["mesh"][rng.integers(0, n, k)] = rng.integers(n_mesh, n_mesh + 1000, k).astype(np.uint32)       # past the table: clamped (unsigned)
        inst["mesh"][rng.integers(0, n, 1)] = np.uint32(0xffffffff)
        cam = synth.camera_uniform(eye=tuple(float(v) for v in (rng.random(3) - 0.5) * 40), yaw_deg=float(rng.random() * 360), pitch_deg=float(rng.random() * 120 - 60),
                                   jitter=(float(rng.random() - 0.5) * 0.01, float(rng.random() - 0.5) * 0.01))
        if case & 1:        # the uniform is an input: a finite (even negative or NaN) far plane, a near plane anywhere
            cam = cam.copy()
            cam["zfar"] = np.float32(rng.choice([0.5, 5.0, 50.0, 500.0, -10.0, 0.0, np.nan, -np.inf]))
            cam["znear"] = np.float32(rng.choice([0.001, 1.0, 100.0, -5.0, np.inf]))
            inf_cam = cam.copy()
            inf_cam["zfar"] = np.float32(np.inf)
        want = ref.cull_emit(cam, meshes, inst, threads=8)
        if case & 1 and int(ref.cull_emit(inf_cam, meshes, inst, threads=8)["instance_count"].sum()) != int(want["instance_count"].sum()):
            far_cases += 1
        wc, wn = ref.compact(want)
        for split in (None, 1):
            ctx.set_option("cull.split_min", split)
            got = ctx.cull_emit(cam, meshes, inst)
            gc, gn = ctx.cull_compact(cam, meshes, inst)
            if got.tobytes() != want.tobytes() or gn != wn or gc[:gn].tobytes() != wc[:wn].tobytes():
                bad += 1
                log(f"case {case} n {n} meshes {n_mesh} split {split}: DIFFERS (count {gn} vs {wn})")
        ctx.set_option("cull.split_min", None)
    return bad, far_cases


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    bad, far = run(args.cases, args.seed, log=lambda m: print(m, flush=True))
    print(f"{args.cases} cases ({far} in which the far plane changed the survivor set), {bad} mismatches")
    sys.exit(1 if bad else 0)
