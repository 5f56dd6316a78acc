#!/usr/bin/env python3
"""A/B of the traversal on the bench scenes: supply (round-2 single-ray counter / chunks per workgroup), ray binning,
chunk size, de-indexed leaf triangles.  Every variant's output is compared byte for byte with the first one's.
    python tools/ab_trace.py [--rays 1024] [--harness]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voidin_amd import abi, synth
from voidin_amd.runtime import Context

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=1024)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--harness", action="store_true", help="the bvh_gpu.rs-shaped scene instead of the stress scene")
args = ap.parse_args()
ctx = Context(0)
if args.harness:
    import bench
    sys.exit("use bench.py for the harness scene")
tv, ti = synth.knot_mesh(512, 128)
nodes_b, idx_b = ctx.bvh_build(tv, ti)
infos = np.zeros(1, dtype=abi.MESH_INFO)
infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(tv)
infos[0]["index_count"] = len(idx_b)
inst_t = synth.instances(2000, n_mesh=1, seed=synth.SEED_BASE + 8, extent=120.0, scale_range=(0.5, 2.0))
tl = ctx.tlas_build(inst_t, infos)
rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 90), pitch_deg=0), args.rays, args.rays)
ds = ctx.device_scene((tl, inst_t, infos, nodes_b, tv, idx_b))
acc = ctx.trace_prepare(ds)
ctx.set_option("trace.tight_tlas", 1)          # opt-in: private top level over tight world boxes (not the reference's visit order)
acc_tight = ctx.trace_prepare(ds)
ctx.set_option("trace.tight_tlas", 2)          # ... the same boxes under an LBVH (built on all CUs)
acc_lbvh = ctx.trace_prepare(ds)
ctx.set_option("trace.tight_tlas", None)
ctx.synchronize()
for _acc, _nm in ((acc_tight, "agglomerative"), (acc_lbvh, "LBVH")):
    _t = []
    for _ in range(5):
        t0 = time.perf_counter(); _acc.update(); _t.append(time.perf_counter() - t0)
    print(f"private top level, {_nm}: vd_trace_accel_update_dev (rebuild from the instance buffer, blocking) {min(_t) * 1e3:.3f} ms")
d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
d_any = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
variants = [("single rays (default)", dict(sort=0, chunk=1), False), ("single rays prep", dict(sort=0, chunk=1), True),
            ("single rays prep TIGHT TLAS", dict(sort=0, chunk=1), "tight"), ("single rays prep TIGHT TLAS (LBVH)", dict(sort=0, chunk=1), "lbvh"),
            ("chunk64 nosort", dict(sort=0, chunk=64), False), ("chunk64 nosort prep", dict(sort=0, chunk=64), True),
             ("chunk64 sort", dict(sort=1, chunk=64), False),
            ("chunk64 sort prep", dict(sort=1, chunk=64), True), ("chunk256 sort prep", dict(sort=1, chunk=256), True),
            ("chunk128 sort prep", dict(sort=1, chunk=128), True)]
if os.environ.get("AB_YIELD"):        # sweep of VD_OPT_TRACE_YIELD on the default supply
    variants = [("yield %2d%s" % (y, " prep" if pr else ""), dict(sort=0, chunk=1, **{"yield": y}), pr)
                for y in [int(v) for v in os.environ["AB_YIELD"].split(",")] for pr in (False, True)]
if os.environ.get("AB_FAN"):          # sweep of VD_OPT_TRACE_FAN (launches per call) on the default supply
    variants = [("fan %d%s" % (r, " prep" if pr else ""), dict(sort=0, chunk=1, fan=r), pr)
                for r in [int(v) for v in os.environ["AB_FAN"].split(",")] for pr in (False, True)]
if os.environ.get("AB_WAVES"):        # sweep of VD_OPT_TRACE_WAVES on the default supply
    variants = [("waves/CU %2d%s" % (w, " prep" if pr else ""), dict(sort=0, chunk=1, waves=w), pr)
                for w in [int(v) for v in os.environ["AB_WAVES"].split(",")] for pr in (False, True)]
ref_bytes = ref_any = None
ctx.set_timing(True)
for name, opts, prep in variants:
    for k in ("sort", "chunk", "yield", "waves", "fan"):
        ctx.set_option("trace." + k, opts.get(k, -1))
    t_cl, t_any = [], []
    acc_v = acc_tight if prep == "tight" else (acc_lbvh if prep == "lbvh" else acc)
    for _ in range(args.reps):
        if prep:
            ctx.trace_prepared_dev(acc_v, d_rays, len(rays), d_hits); t_cl.append(ctx.last_gpu_ms())
            ctx.trace_any_prepared_dev(acc_v, d_rays, len(rays), d_any); t_any.append(ctx.last_gpu_ms())
        else:
            ctx.trace_dev(ds, d_rays, len(rays), d_hits); t_cl.append(ctx.last_gpu_ms())
            ctx.trace_any_dev(ds, d_rays, len(rays), d_any); t_any.append(ctx.last_gpu_ms())
    if hasattr(ctx.lib, "vd_debug_trace_counters") and os.environ.get("AB_COUNTERS"):      # tuning build: what the waves did (last call = occlusion)
        import ctypes as C
        for kind in ("closest", "occlusion"):
            if kind == "closest":
                (ctx.trace_prepared_dev(acc_v, d_rays, len(rays), d_hits) if prep else ctx.trace_dev(ds, d_rays, len(rays), d_hits))
            else:
                (ctx.trace_any_prepared_dev(acc_v, d_rays, len(rays), d_any) if prep else ctx.trace_any_dev(ds, d_rays, len(rays), d_any))
            c = (C.c_uint64 * 11)()
            ctx.lib.vd_debug_trace_counters.argtypes = [C.c_void_p, C.c_void_p]
            ctx.lib.vd_debug_trace_counters(ctx.h, c)
            n = len(rays)
            print(f"    {kind}: outer iterations {c[0]}, stepping iterations {c[1]} ({c[1] / (256 * 28):.0f} per wave), lanes per iteration {c[2] / max(1, c[1]):.1f}, "
                  f"lane-steps per ray {c[2] / n:.0f} (leaf {c[3] / n:.1f}, entry {c[4] / n:.1f}, TLAS interior {c[5] / n:.1f}); longest ray {c[6]} steps; "
                  f"iterations after the last ray was handed out: {c[7] / (256 * 28):.0f} per wave; the longest wave: {c[8]} iterations; most iterations of a wave with one busy lane: {c[9]}, with two to four: {c[10]}")
            if hasattr(ctx.lib, "vd_debug_trace_timeline"):
                tl_ = (C.c_uint32 * 46)()
                ctx.lib.vd_debug_trace_timeline.argtypes = [C.c_void_p, C.c_void_p]
                ctx.lib.vd_debug_trace_timeline(ctx.h, tl_)
                ended, iters = list(tl_)[:23], list(tl_)[23:]
                last = max([k for k in range(23) if ended[k] or iters[k]] + [0])
                print("    timeline, 2 ms slots: waves ended " + " ".join(str(v) for v in ended[: last + 1]))
                print("                 k wave-iterations " + " ".join(str(v // 1000) for v in iters[: last + 1]))
    b, a = d_hits.cpu().numpy().tobytes(), d_any.cpu().numpy().tobytes()
    if ref_bytes is None:
        ref_bytes, ref_any = b, a
    if prep in ("tight", "lbvh"):       # same hits and distances; the instance / triangle reported for two hits at one distance may differ
        h, r = np.frombuffer(b, dtype=abi.HIT), np.frombuffer(ref_bytes, dtype=abi.HIT)
        m = r["hit"] == 1
        print(f"    tight: hit flags equal {np.array_equal(h['hit'], r['hit'])}, distances bit-equal {int((h['dist'][m].view(np.uint32) == r['dist'][m].view(np.uint32)).sum())} of {int(m.sum())}")
    print(f"{name:32s} closest {len(rays) / min(t_cl) / 1e3:7.1f} Mrays/s ({min(t_cl):7.2f} ms)  occlusion {len(rays) / min(t_any) / 1e3:7.1f} Mrays/s"
          f"  same bytes: {b == ref_bytes} {a == ref_any}", flush=True)
