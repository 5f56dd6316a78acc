"""Random scenes through the indexed TLAS build (forced on from 65 clusters) against the oracle: clouds of boxes with
duplicates, nesting and zero extents mixed in, real instances (rotations, anisotropic scale), random sizes.
    python tools/fuzz_tlas.py [--cases 300] [--seed 1]
`run(cases, seed, ctx)` is what tests/test_gpu_fuzz.py calls with a fixed seed."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref  # noqa: E402
from voidin_amd import abi, synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402


def run(cases, seed, ctx=None, log=print, max_n=2500):
    rng = np.random.default_rng(seed)
    ctx = ctx or Context(0)
    ctx.set_option("tlas.index_min", 65)
    ctx.set_option("tlas.phase2", 64)
    bad = 0
    try:
        for case in range(cases):
            kind = case % 3
            n = int(rng.integers(65, max_n))
            ctx.set_option("tlas.refresh", int(rng.choice([0, 7, 64, 1024])))
            if kind == 2:                                           # real instances of the bench family
                meshes = synth.mesh_infos(int(rng.integers(1, 40)))
                inst = synth.instances(n, n_mesh=len(meshes), seed=int(rng.integers(1 << 30)), extent=float(rng.choice([5.0, 60.0, 900.0])),
                                       scale_range=(0.05, 4.0))
            else:
                c = (rng.random((n, 3)).astype(np.float32) - np.float32(0.5)) * np.float32(rng.choice([1.0, 30.0, 400.0]))
                h = rng.random((n, 3)).astype(np.float32) * np.float32(rng.choice([0.0, 0.5, 8.0]))
                if kind == 1:                                       # ties: snap to a lattice, duplicate a third, nest a few giants
                    c = np.round(c)
                    h = np.round(h * 2) / 2
                    dup = rng.integers(0, n, n // 3)
                    c[: len(dup)] = c[dup]; h[: len(dup)] = h[dup]
                    g = rng.integers(0, n, 3)
                    h[g] += np.float32(300.0)
                boxes = np.concatenate([c - h, c + h], axis=1).astype(np.float32)
                boxes = boxes[rng.permutation(n)]
                meshes = np.zeros(n, dtype=abi.MESH_INFO)
                meshes["min"], meshes["max"] = boxes[:, :3], boxes[:, 3:]
                inst = np.zeros(n, dtype=abi.INSTANCE)
                eye = np.eye(4, dtype=np.float32).reshape(16)
                inst["transform"], inst["inv_transform"] = eye, eye
                inst["mesh"] = np.arange(n, dtype=np.uint32)
            want = ref.tlas_build(inst, meshes)
            got = ctx.tlas_build(inst, meshes)
            if got.tobytes() != want.tobytes():
                bad += 1
                log(f"case {case} kind {kind} n {n}: DIFFERS")
    finally:
        for o in ("tlas.index_min", "tlas.phase2", "tlas.refresh"):
            ctx.set_option(o, None)
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    bad = run(args.cases, args.seed, log=lambda m: print(m, flush=True))
    print(f"{args.cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)
