"""Random meshes through the GPU BLAS builder against the oracle (crates/bvh/src/blas.rs:51-204 restated): sizes that land in every
tier (lane groups, wave-wide, block-wide root, mid tier, phase A), coordinates with ties (lattice-snapped), duplicated triangles,
slivers, huge / tiny magnitudes, NaN / inf vertices, shared vertices - nodes and permuted indices byte for byte, error codes equal.
    python tools/fuzz_blas.py [--cases 400] [--seed 1] [--max-tris 6000]
`run(cases, seed, ctx)` is what tests/test_gpu_fuzz.py calls with a fixed seed."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref  # noqa: E402
from voidin_amd import abi  # noqa: E402
from voidin_amd.runtime import Context, VoidinError  # noqa: E402

SIZES = [4, 5, 8, 9, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4097]


def run(cases, seed, ctx=None, log=print, max_tris=6000):
    """-> (mismatches, cases both sides rejected as degenerate)"""
    rng = np.random.default_rng(seed)
    ctx = ctx or Context(0)
    bad = degenerate = 0
    sizes = SIZES
    for case in range(cases):
        kind = case % 8
        n = int(rng.choice(sizes)) if rng.random() < 0.5 else int(rng.integers(4, max_tris))
        ext = np.float32(rng.choice([1.0, 30.0, 1e4]))
        base = (rng.random((n, 3)).astype(np.float32) - np.float32(0.5)) * ext
        size = np.float32(rng.choice([1e-3, 0.3, 5.0]))
        tri = base[:, None, :] + (rng.random((n, 3, 3)).astype(np.float32) - np.float32(0.5)) * size
        if kind == 1:                                   # ties: everything on a coarse lattice
            tri = np.round(tri / ext * np.float32(8)) * ext / np.float32(8)
        elif kind == 2:                                 # a third of the triangles are copies of others (equal centroids, equal boxes)
            dup = rng.integers(0, n, n // 3)
            tri[: len(dup)] = tri[dup]
        elif kind == 3:                                 # all centroids on a line / a plane: two axes without extent
            tri[:, :, 1] = tri[:, :1, 1] * 0 + np.float32(2.5)
            if rng.random() < 0.5:
                tri[:, :, 2] = np.float32(-1.0)
        elif kind == 4:                                 # magnitudes up to the format's 1e30 and down to denormals
            tri = tri * np.float32(rng.choice([1e25, 1e-30, 1e-42]))
        elif kind == 5:                                 # NaN / inf vertices sprinkled in
            k = max(1, n // 50)
            tri.reshape(-1)[rng.integers(0, tri.size, k)] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), k)
        elif kind == 6:                                 # clusters: most triangles in a few tight clumps
            c = (rng.random((6, 3)).astype(np.float32) - np.float32(0.5)) * ext
            tri = c[rng.integers(0, 6, n)][:, None, :] + (rng.random((n, 3, 3)).astype(np.float32) - np.float32(0.5)) * np.float32(1e-2)
        verts = tri.reshape(-1, 3).astype(np.float32)
        idx = np.arange(3 * n, dtype=np.uint32)
        if kind == 7:                                   # an indexed mesh with shared vertices, shuffled triangle order
            nv = max(4, n // 2)
            verts = ((rng.random((nv, 3)).astype(np.float32) - np.float32(0.5)) * ext).astype(np.float32)
            idx = rng.integers(0, nv, 3 * n).astype(np.uint32)
        want_rc, got_rc = 0, 0
        try:
            wn, wi = ref.bvh_build(verts, idx)
        except ref.OracleError as e:
            want_rc = e.code
        try:
            gn, gi = ctx.bvh_build(verts, idx)
        except VoidinError as e:
            got_rc = e.code
        if want_rc or got_rc:
            degenerate += 1
            if want_rc != got_rc:
                bad += 1
                log(f"case {case} kind {kind} n {n}: status differs: oracle {want_rc}, gpu {got_rc}")
            continue
        if gn.tobytes() != wn.tobytes() or gi.tobytes() != wi.tobytes():
            bad += 1
            k = next((i for i in range(min(len(gn), len(wn))) if gn[i].tobytes() != wn[i].tobytes()), -1)
            log(f"case {case} kind {kind} n {n}: DIFFERS (nodes {len(gn)} vs {len(wn)}, first differing node {k})")
    return bad, degenerate


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-tris", type=int, default=6000)
    args = ap.parse_args()
    bad, degenerate = run(args.cases, args.seed, log=lambda m: print(m, flush=True), max_tris=args.max_tris)
    print(f"{args.cases} cases ({degenerate} rejected as degenerate by both sides), {bad} mismatches")
    sys.exit(1 if bad else 0)
