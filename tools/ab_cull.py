"""A/B the forms of vd_cull_compact in ONE process (interleaved rounds, median + min).
Variants: 0 = library default (split form from 2^20 instances on), 4/8/16/32 = fused single-pass
kernel with that many rounds per wave per tile, m70 (-70) = split form with per-round id stores in pass 1,
m71 (-71) / m74 (-74) = pass 2 forced to its LDS-staged / direct-store form (default: direct up to 12 Mi instances),
m81 (-81) = staged form without the same-XCD prefetch, m80 (-80) = vd_cull_emit as one kernel (no split).
(The r01 logs under profiles/ use the numbering of the variants that were pruned afterwards:
LDS-DMA, strided loads, compact staging, persistent, wave-tile, ablations, stream probes.)
Usage (on a GPU box): python tools/ab_cull.py [--variants 0,32,16,m70] [--n 10000000] [--dist baseline|small]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import abi, synth  # noqa: E402
from voidin_amd.runtime import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,32,16,m70")
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--dist", default="baseline")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()

ctx = Context(0)
lib = ctx.lib
cam, meshes = synth.camera_uniform(), synth.mesh_infos()
kw = dict(scale_range=(0.25, 4.0)) if args.dist == "baseline" else dict(scale_range=(0.02, 0.6), extent=600.0)
inst = synth.instances(args.n, seed=synth.SEED_BASE + 3, with_inverse=False, **kw)
n = args.n
d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
d_out = ctx.empty(n * 20)
d_cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
variants = [int(v) for v in args.variants.replace('m', '-').split(",")]
ref_bytes = None
times = {v: [] for v in variants}
for rnd in range(args.rounds):
    for v in variants:
        lib.vd_ctx_set_option(ctx.h, 2, v)      # VD_OPT_CULL_VARIANT: signed ids, taken as is
        for _ in range(2):
            ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt)
        torch.cuda.synchronize()
        if rnd == 0:
            cnt = int(d_cnt[0].item())
            b = d_out[: cnt * 20].cpu().numpy().tobytes()
            if ref_bytes is None:
                ref_bytes, ref_cnt = b, cnt
            ok = (cnt == ref_cnt and b == ref_bytes)
            print(f"variant {v}: count {cnt} matches variant {variants[0]}: {ok}", flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / args.iters)
vis = ref_cnt / n
alg = n * (144 + 20 * vis)
print(f"n={n} visible={vis:.4f} algorithmic bytes={alg/1e9:.3f} GB")
for v in variants:
    t = np.array(times[v])
    print(f"variant {v:2d}: median {np.median(t)*1e3:8.1f} us  min {t.min()*1e3:8.1f} us  -> {alg/np.median(t)/1e6:7.1f} GB/s ({alg/np.median(t)/1e6/8000*100:.1f}% of 8 TB/s)")
