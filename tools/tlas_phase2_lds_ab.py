"""The indexed TLAS build with the slot arrays of its final plain scans (<= 2048 clusters) in LDS against memory
(VD_OPT_TLAS_CHAIN_LDS 1 / 0).  python tools/tlas_phase2_lds_ab.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import abi, synth
from voidin_amd.runtime import Context
ctx = Context(0); meshes = synth.mesh_infos(); d_m = ctx.upload(meshes)
for n in (7000, 8192, 12000, 32768):
    inst = synth.instances(n, seed=synth.SEED_BASE + 6, extent=300.0); d_i = ctx.upload(inst); d_t = ctx.empty((2 * n + 1) * 32)
    out = {}
    for lds in (1, 0, 1, 0):
        ctx.set_option("tlas.chain_lds", lds); ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); ctx.tlas_build_dev(d_i, n, d_m, len(meshes), d_t); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        out.setdefault(lds, []).append(min(ts) * 1e3); out[("b", lds)] = d_t.cpu().numpy().tobytes()
    print(f"n={n}: final scans from LDS {min(out[1]):.2f} ms, from memory {min(out[0]):.2f} ms, same bytes {out[('b', 1)] == out[('b', 0)]}", flush=True)
