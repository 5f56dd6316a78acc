"""Fused single-launch form vs split form of vd_cull_compact around the switch-over size.
VD_SPLIT_MIN=<n> selects the form; run once per setting: python tools/ab_split_min.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

ctx = Context(0)
cam, meshes = synth.camera_uniform(), synth.mesh_infos()
N = 12 << 20
inst = synth.instances(N, seed=synth.SEED_BASE + 3, with_inverse=False, scale_range=(0.25, 4.0))
d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
d_out, d_cnt = ctx.empty(N * 20), torch.zeros(4, dtype=torch.int32, device="cuda")
print("VD_SPLIT_MIN =", os.environ.get("VD_SPLIT_MIN", "(default)"))
for n in (1 << 16, 1 << 18, 1 << 20, 1 << 21, 3 << 20, 1 << 22, 5 << 20, 6 << 20, 1 << 23, 10_000_000, 12 << 20):
    f = lambda: ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt)
    for _ in range(5):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    t_c = (time.perf_counter() - t0) / 200
    g = lambda: ctx.cull_emit_dev(cam, d_m, len(meshes), d_i, n, d_out)
    for _ in range(5):
        g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        g()
    torch.cuda.synchronize()
    print(f"  n = {n:8d}: compact {t_c * 1e6:8.2f} us/call   emit {(time.perf_counter() - t0) / 200 * 1e6:8.2f} us/call")
