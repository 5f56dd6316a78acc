import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from voidin_amd import abi, synth
from voidin_amd.runtime import Context
ctx = Context(0); meshes = synth.mesh_infos(); d_m = ctx.upload(meshes)
for n in (5700, 6000, 6500, 7000, 7400, 7800, 8192):
    inst = synth.instances(n, seed=synth.SEED_BASE + 6, extent=300.0); d_i = ctx.upload(inst); d_t = ctx.empty((2 * n + 1) * 32)
    out = {}
    for idx in (1, 0):
        ctx.set_option("tlas.index", idx); ctx.set_option("tlas.index_min", 2048); ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); ctx.tlas_build_dev(d_i, n, d_m, len(meshes), d_t); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        out[idx] = (min(ts) * 1e3, d_t.cpu().numpy().tobytes())
    print(f"n={n}: indexed {out[1][0]:.2f} ms, chain {out[0][0]:.2f} ms, same bytes {out[1][1] == out[0][1]}", flush=True)
