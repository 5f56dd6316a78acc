"""Time the multi-GPU wire-format kernels on one GPU: cull_mask (bitmask) and expand_mask."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import synth
from voidin_amd import dist as vdist
from voidin_amd.runtime import Context
ctx = Context(0)
ctx.set_timing(True)
n = 10_000_000
cam, meshes = synth.camera_uniform(), synth.mesh_infos()
inst = synth.instances(n, seed=synth.SEED_BASE + 3, with_inverse=False)
d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
import sys as _s
SH = tuple(int(x) for x in _s.argv[1].split(",")) if len(_s.argv) > 1 else (1, 8)
for shards in SH:
    N = n * shards
    S = n
    wps = vdist.mask_words(S)
    d_mask = torch.zeros(wps * shards, dtype=torch.int64, device="cuda")
    ids = torch.from_numpy(np.tile(inst["mesh"].astype(np.uint8), shards)).cuda()
    d_out = ctx.empty(N * 20); d_cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    tm, te = [], []
    for it in range(12):
        ctx.cull_mask_dev(cam, d_m, len(meshes), d_i, n, d_mask)
        tm.append(ctx.last_gpu_ms())
        if shards > 1 and it == 0:
            for r in range(1, shards): d_mask[r * wps:(r + 1) * wps] = d_mask[:wps]
        ctx.expand_mask_dev(d_mask, N, S, ids, d_m, len(meshes), d_out, d_cnt)
        te.append(ctx.last_gpu_ms())
    cnt = int(d_cnt[0].item())
    print(f"shards {shards}: cull_mask {np.median(tm[2:])*1e3:.1f} us ({n*144/np.median(tm[2:])/1e6:.0f} GB/s); "
          f"expand {N} inst -> {cnt} draws: {np.median(te[2:])*1e3:.1f} us ({(N*4+cnt*20)/np.median(te[2:])/1e6:.0f} GB/s)")
