"""Timing of the traversal on the stress scene of bench.py (2000 overlapping instances of a 131 k-triangle mesh).
    python tools/trace_stress.py [tag] [side=1024] [prepared=0]      (per-context options through VD_TRACE_* variables)"""
import os, sys, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import abi, synth
from voidin_amd.runtime import Context
tag = sys.argv[1] if len(sys.argv) > 1 else ""
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prepared = len(sys.argv) > 3 and sys.argv[3] == "1"
ctx = Context(0)
tv, ti = synth.knot_mesh(512, 128)
nodes_b, idx_b = ctx.bvh_build(tv, ti)
infos = np.zeros(1, dtype=abi.MESH_INFO)
infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(tv)
infos[0]["index_count"] = len(idx_b)
inst_t = synth.instances(2000, n_mesh=1, seed=synth.SEED_BASE + 8, extent=120.0, scale_range=(0.5, 2.0))
tl = ctx.tlas_build(inst_t, infos)
rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 90), pitch_deg=0), side, side)
ds = ctx.device_scene((tl, inst_t, infos, nodes_b, tv, idx_b))
acc = ctx.trace_prepare(ds) if prepared else None
d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
d_any = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
print(tag, "rays", len(rays), "prepared", prepared, flush=True)
ctx.set_timing(True)
for rep in range(2):
    t_cl, t_any = [], []
    for _ in range(3):
        if prepared:
            ctx.trace_prepared_dev(acc, d_rays, len(rays), d_hits); t_cl.append(ctx.last_gpu_ms())
            ctx.trace_any_prepared_dev(acc, d_rays, len(rays), d_any); t_any.append(ctx.last_gpu_ms())
        else:
            ctx.trace_dev(ds, d_rays, len(rays), d_hits); t_cl.append(ctx.last_gpu_ms())
            ctx.trace_any_dev(ds, d_rays, len(rays), d_any); t_any.append(ctx.last_gpu_ms())
    print(f"{tag} closest {len(rays) / min(t_cl) / 1e3:7.1f} Mrays/s  occlusion {len(rays) / min(t_any) / 1e3:7.1f} Mrays/s", flush=True)
print("crc", zlib.crc32(d_hits.cpu().numpy().tobytes()))
