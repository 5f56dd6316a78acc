"""Repeat the multi-workgroup TLAS build and compare every result with the first (bytes): a race in the
exchange would show as a different chain.  python tools/stress_tlas.py [n] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = Context(0)
meshes = synth.mesh_infos()
inst = synth.instances(n, seed=synth.SEED_BASE + 15, extent=500.0)
wide = n > 32768
d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
d_t = ctx.empty((2 * n + 1) * (48 if wide else 32))
first = None
bad = 0
for r in range(reps):
    d_t.zero_()
    ctx.tlas_build_dev(d_i, n, d_m, len(meshes), d_t, wide=wide)
    torch.cuda.synchronize()
    got = d_t.cpu().numpy().tobytes()
    if first is None:
        first = got
    elif got != first:
        bad += 1
        print(f"rep {r}: DIFFERENT", flush=True)
print(f"n={n}: {reps} builds, {bad} different from the first")
sys.exit(1 if bad else 0)
