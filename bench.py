#!/usr/bin/env python3
"""bench.py — headline benchmark of the visibility hot path on MI355X.

A "step" = one pass of cull + emit + ordered compaction (vd_cull_compact_shard_dev) over one
batch of synthetic instances already resident in HBM.  Workload at N=1 = BASELINE.json
configs[2]: 10M synthetic AABB instances (BASELINE.md §3 distribution), 16 MeshInfo, the
model.rs camera.  N>1: weak scaling — every rank owns a 10M-instance shard of an N*10M scene,
culls + compacts it with global base_instance values, then the compacted draw lists are
exchanged (counts all-gather + one-shot direct all-gather over RCCL/xGMI) inside the step.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel: cull_compact_kernel, HBM-bound) and `cpu_baseline` (the oracle's restatement of the
cull timed on this box's host cores; a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--instances", type=int, default=10_000_000, help="instances per GPU")
    ap.add_argument("--dist", choices=["baseline", "small"], default="baseline",
                    help="baseline = BASELINE.md §3 (S in [0.25,4]); small = S in [0.02,0.6] (more culled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (emit_draws, BVH build, TLAS)")
    ap.add_argument("--bvh-u", type=int, default=2048, help="knot mesh resolution: 2*u*v triangles (default 8.4M)")
    ap.add_argument("--bvh-v", type=int, default=2048)
    args = ap.parse_args()

    import numpy as np
    import torch

    from voidin_amd import abi, synth
    from voidin_amd import dist as vdist
    from voidin_amd.runtime import Context

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL ("nccl" on ROCm).  VOIDIN_DIST_BACKEND=gloo lets the same code path be
        # exercised with several ranks on ONE GPU (tests / debugging): ranks then share device 0.
        backend = os.environ.get("VOIDIN_DIST_BACKEND", "nccl")
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    ctx = Context(dev.index)  # raises if the HIP extension or a gfx950 GPU is missing

    n = args.instances
    n_total = n * world
    first = rank * n
    kw = dict(scale_range=(0.25, 4.0)) if args.dist == "baseline" else dict(scale_range=(0.02, 0.6), extent=600.0)
    cam = synth.camera_uniform()
    meshes = synth.mesh_infos()
    t0 = time.time()
    inst = synth.instances(n, seed=synth.SEED_BASE + 3, offset=first, with_inverse=False, **kw)
    t_gen = time.time() - t0

    d_m, d_i = ctx.upload(meshes), ctx.upload(inst)
    d_out = ctx.empty(n * 20)
    d_cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    # N > 1: every rank ends the step with the ordered draw list of the WHOLE scene.  The exchange is
    # one bit per instance (bitmask all-gather) + local expansion (voidin_amd/dist.py).
    d_all = ctx.empty(n_total * 20) if distributed else None
    d_cnt_all = torch.zeros(4, dtype=torch.int32, device=dev) if distributed else None
    sv = vdist.ShardedVisibility(ctx, n_total, d_m, len(meshes), d_i) if distributed else None

    # N = 1: consecutive steps see DIFFERENT instance buffers - the second is the first after one compute_update
    # animation step (every transform changed, mesh assignment unchanged: what the reference's dynamic scene does,
    # shaders/compute_update.wgsl:10-28) - so nothing but the per-scene instance->mesh table can carry over between
    # steps (pass 1 re-derives that table every step and rewrites only rows that changed).
    d_i_b = None
    if not distributed:
        d_i_b = d_i.clone()
        d_all_idx = torch.arange(n, dtype=torch.int32, device=dev)
        ctx.compute_update_dev(d_all_idx, n, d_i_b, n, 1.0, 0.016)
        torch.cuda.synchronize()
        del d_all_idx
    step_no = [0]

    def step():
        if distributed:
            sv.step(cam, d_all, d_cnt_all)
        else:
            src = d_i if (step_no[0] & 1) == 0 else d_i_b
            step_no[0] += 1
            ctx.cull_compact_dev(cam, d_m, len(meshes), src, n, d_out, d_cnt, False, first)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    if distributed:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    ms_per_step = wall * 1e3 / args.steps
    breakdown = None
    if distributed:
        # where a step goes at this N (every rank runs every leg; max over ranks, like the headline): the local cull to a
        # bitmask, the RCCL all-gather of the masks alone, and the expansion of ALL shards to the full draw list
        def leg(fn):
            fn(); barrier(); t = time.perf_counter()
            for _ in range(args.steps):
                fn()
            barrier()
            tw = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            return round(float(tw.item()) * 1e3 / args.steps, 4)
        breakdown = {
            "cull_to_mask_ms": leg(lambda: ctx.cull_mask_dev(cam, d_m, len(meshes), d_i, n, sv.d_mask)),
            "mask_allgather_ms": leg(lambda: dist.all_gather_into_tensor(sv.d_mask_all, sv.d_mask)),
            "expand_all_shards_ms": leg(lambda: ctx.expand_mask_dev(sv.d_mask_all, n_total, sv.S, sv.d_mesh_ids, d_m, len(meshes), d_all, d_cnt_all)),
            "mask_bytes_per_rank": int(sv.wps * 8), "draw_list_bytes_written_per_rank": int(d_cnt_all[0].item()) * 20,
            "note": "every GPU materialises the whole list: the expansion leg writes N x the single-GPU output and is bound by the HBM write ceiling (DESIGN.md 6)"}
        sv.step(cam, d_all, d_cnt_all)       # leave d_all / d_cnt_all as a full step leaves them
        torch.cuda.synchronize()
    if distributed:   # also run the local fused kernel once so the roofline / verification legs have its output
        ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt, False, first)
        torch.cuda.synchronize()
    count = int(d_cnt[0].item())

    # Kernel-level timing: HIP events recorded by the library on the launch stream around each pass of
    # vd_cull_compact (large inputs run two passes: cull -> bitmask + mesh id, then expansion).
    barrier()
    ctx.set_timing(True)
    k_all, k_cull, k_expand = [], [], []
    for _ in range(args.steps):
        ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt, False, first)
        k_all.append(ctx.last_gpu_ms())
        k_cull.append(ctx.last_gpu_ms_stage(0))
        k_expand.append(ctx.last_gpu_ms_stage(1))
    ctx.set_timing(False)
    step_kernel_ms = sum(k_all) / len(k_all)
    split = min(k_cull) > 0
    vis = count / n
    step_bytes = n * (144.0 + 20.0 * vis)          # 144 B read + 20 B per survivor
    if split:
        id_bytes = 1 if len(meshes) <= 256 else (2 if len(meshes) <= 65536 else 4)
        kernel_name = "cull_mask_tiled_kernel"
        kernel_ms = sum(k_cull) / len(k_cull)
        alg_bytes = n * (144.0 + 0.125 + id_bytes)  # 144 B instance read + 1 bit written + the compact mesh id (compared; rewritten when it changed)
        expand_ms = sum(k_expand) / len(k_expand)
        expand_bytes = n * (0.125 + id_bytes) + count * 20.0
    else:
        kernel_name, kernel_ms, alg_bytes, expand_ms, expand_bytes = "cull_compact_kernel", step_kernel_ms, step_bytes, None, None
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    verified = None
    cpu = None
    near = None
    if rank == 0:
        from oracle import ref  # checker + cpu_baseline leg only
        if not args.no_verify:
            # how many instances sit so close to a frustum plane that an implementation with different rounding (WGSL leaves
            # length / sqrt precision and FMA contraction to the driver) could decide them differently: the test is
            # `lhs < -radius`; distance of lhs + radius from 0, relative to the radius (hard part ii)
            m = min(n, 2_000_000)
            mx_, my_, r_ = ref.cull_margins(cam, meshes, inst[:m])
            with np.errstate(divide="ignore", invalid="ignore"):
                rel = np.minimum(np.abs(mx_), np.abs(my_)) / np.abs(r_)
            near = {"sample": int(m), "within_1e-6_of_a_plane": int((rel < 1e-6).sum()), "within_1e-4": int((rel < 1e-4).sum()),
                    "within_1e-2": int((rel < 1e-2).sum())}
            want = ref.cull_emit(cam, meshes, inst, threads=os.cpu_count() or 1)
            want["base_instance"] += np.uint32(first)
            wc, wn = ref.compact(want)
            got = d_out.cpu().numpy()[: count * 20]
            verified = bool(wn == count and got.tobytes() == wc[:wn].tobytes())
            if distributed:   # rank 0 owns the first shard: the head of the gathered list must be its compaction
                head = d_all.cpu().numpy()[: count * 20]
                total = int(d_cnt_all[0].item())
                verified = bool(verified and head.tobytes() == wc[:wn].tobytes() and total >= count)
        if not args.no_cpu_baseline and not distributed:      # reported at N = 1 only
            cores = os.cpu_count() or 1
            m = min(n, 10_000_000)
            reps1 = 3
            t = time.perf_counter()
            for _ in range(reps1):
                ref.cull_emit(cam, meshes, inst[:m], threads=1)
            t1 = (time.perf_counter() - t) / reps1
            repsN = 10
            t = time.perf_counter()
            for _ in range(repsN):
                d = ref.cull_emit(cam, meshes, inst[:m], threads=cores)
            tN = (time.perf_counter() - t) / repsN
            t = time.perf_counter()
            ref.compact(d)
            tc = time.perf_counter() - t
            cpu = {"value": round(m / (tN + tc) / 1e6, 2), "unit": "M instances culled+compacted/s", "cores": cores,
                   "kind": "port",
                   "sample": f"{m} instances of the same workload; cull on {cores} threads x{repsN} + serial compaction; "
                             f"1-thread cull: {m / t1 / 1e6:.2f} M inst/s"}

    extra = {}

    def pipelined(fn, reps=50):          # per-call time when calls are queued back to back (a frame loop does not sync per call)
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    if not args.no_extra and rank == 0 and not distributed:   # single-GPU extras (they use no collective; N > 1 runs skip them)
        # the smaller BASELINE configs (configs[0] 1 k, configs[1] 100 k instances; 1 M = the largest input of the fused
        # single-launch form): cull + compaction per call, calls queued back to back; bit-exactness of these sizes is in tests/
        small = {}
        for m_ in (1000, 100_000, 1_000_000):
            if m_ <= n:
                t_ = pipelined(lambda: ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, m_, d_out, d_cnt, False, first), reps=200)
                small[str(m_)] = {"us_per_call": round(t_ * 1e6, 2), "M_inst_per_s": round(m_ / t_ / 1e6, 1)}
        extra["cull_compact_small_inputs"] = small
        d_emit = ctx.empty(n * 20)
        for _ in range(3):
            ctx.cull_emit_dev(cam, d_m, len(meshes), d_i, n, d_emit)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            ctx.cull_emit_dev(cam, d_m, len(meshes), d_i, n, d_emit)
        e1.record()
        torch.cuda.synchronize()
        ems = e0.elapsed_time(e1) / args.steps
        extra["emit_draws_uncompacted"] = {"ms": round(ems, 4), "M_inst_per_s": round(n / ems / 1e3, 1),
                                           "GBps": round(n * 164.0 / ems / 1e6, 1),
                                           "frac_of_8TBps": round(n * 164.0 / ems / 1e6 / HBM_PEAK_GBS, 4)}
        del d_emit
        # the multi-GPU wire-format path on one GPU: cull -> bitmask, bitmask -> ordered draw list
        sv1 = vdist.ShardedVisibility(ctx, n, d_m, len(meshes), d_i)
        d_o2, d_c2 = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device=dev)
        for _ in range(3):
            sv1.step(cam, d_o2, d_c2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            sv1.step(cam, d_o2, d_c2)
        e1.record()
        torch.cuda.synchronize()
        same = bool(int(d_c2[0].item()) == count and torch.equal(d_o2[: count * 20], d_out[: count * 20]))
        extra["mask_then_expand_1gpu"] = {"ms": round(e0.elapsed_time(e1) / args.steps, 4), "equals_fused": same}
        # worst case for pass 1's id table: every instance changes its mesh between consecutive steps (all rows rewritten)
        inst_c = inst.copy()
        inst_c["mesh"] = (inst_c["mesh"] + 1) % len(meshes)
        d_i_c = ctx.upload(inst_c)
        del inst_c
        for k in range(4):
            ctx.cull_compact_dev(cam, d_m, len(meshes), d_i if k & 1 else d_i_c, n, d_out, d_cnt, False, first)
        torch.cuda.synchronize()
        e0.record()
        for k in range(args.steps):
            ctx.cull_compact_dev(cam, d_m, len(meshes), d_i if k & 1 else d_i_c, n, d_out, d_cnt, False, first)
        e1.record()
        torch.cuda.synchronize()
        extra["cull_compact_all_mesh_ids_changing"] = {"ms": round(e0.elapsed_time(e1) / args.steps, 4),
                                                       "note": "every step rewrites the whole instance->mesh table; the headline steps alternate two buffers with equal mesh ids and different transforms"}
        del d_i_c
        ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt, False, first)   # leave d_out / the table as the later legs expect
        torch.cuda.synchronize()
        del d_o2, sv1
        # --- BASELINE config 5: SAH BVH build of a dragon-like 8M-tri mesh, TLAS build/refit ---
        from oracle import ref
        v, idx = synth.knot_mesh(args.bvh_u, args.bvh_v)
        n_tri = len(idx) // 3
        d_v = ctx.upload(v)
        d_n = ctx.empty(2 * n_tri * 32)
        best = None
        for r in range(3):
            d_idx = ctx.upload(idx)
            torch.cuda.synchronize()
            t = time.perf_counter()
            n_nodes = ctx.bvh_build_dev(d_v, len(v), d_idx, n_tri, d_n, 2 * n_tri)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            best = dt if best is None or (r and dt < best) else best
        # CPU baseline + parity on a bounded sample of the same mesh family
        sv, sidx = synth.knot_mesh(512, 256)            # 262k triangles
        t = time.perf_counter()
        wn, wi = ref.bvh_build(sv, sidx)
        t_cpu = time.perf_counter() - t
        gn, gi = ctx.bvh_build(sv, sidx)
        bvh_ok = bool(len(gn) == len(wn) and gn.tobytes() == wn.tobytes() and np.array_equal(gi, wi))
        extra["bvh_build"] = {"metric": "SAH BVH build Mprims/s", "value": round(n_tri / best / 1e6, 1), "n_tris": n_tri,
                              "ms": round(best * 1e3, 2), "nodes": int(n_nodes),
                              "cpu_baseline": {"value": round(len(sidx) // 3 / t_cpu / 1e6, 3), "unit": "Mprims/s", "cores": 1,
                                               "kind": "port", "sample": f"{len(sidx)//3}-tri knot mesh, oracle vd_ref_bvh_build"},
                              "sample_bit_exact_vs_oracle": bvh_ok}
        del d_v, d_n, d_idx
        n_tl = 32768
        tinst = synth.instances(n_tl, seed=synth.SEED_BASE + 6, extent=300.0)
        d_ti = ctx.upload(tinst)
        d_t = ctx.empty((2 * n_tl + 1) * 32)
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.tlas_build_dev(d_ti, n_tl, d_m, len(meshes), d_t)
        torch.cuda.synchronize(); t_build = time.perf_counter() - t
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            ctx.tlas_refit_dev(d_ti, n_tl, d_m, len(meshes), d_t)
            torch.cuda.synchronize(); t_refit = time.perf_counter() - t
        t_pipe = pipelined(lambda: ctx.tlas_refit_dev(d_ti, n_tl, d_m, len(meshes), d_t))
        extra["tlas"] = {"n_instances": n_tl, "build_ms": round(t_build * 1e3, 1), "refit_ms": round(t_refit * 1e3, 3),
                         "refit_queued_ms": round(t_pipe * 1e3, 4)}
        # the dynamic-scene frame: animate 10 % of the instances (compute_update, inverse kept in step), refit
        # the TLAS of the first 32768, cull + compact all of them - queued back to back as a frame loop does
        if n >= n_tl:
            d_dyn = d_i.clone()
            d_mov = torch.arange(0, n, 10, dtype=torch.int32, device=dev)
            d_t2 = ctx.empty((2 * n_tl + 1) * 32)
            ctx.tlas_build_dev(d_dyn, n_tl, d_m, len(meshes), d_t2)

            def frame():
                ctx.compute_update_dev(d_mov, d_mov.numel(), d_dyn, n, 1.0, 0.016, True)
                ctx.tlas_refit_dev(d_dyn, n_tl, d_m, len(meshes), d_t2)
                ctx.cull_compact_dev(cam, d_m, len(meshes), d_dyn, n, d_out, d_cnt, False, first)
            t_frame = pipelined(frame, reps=30)
            extra["dynamic_frame"] = {"instances": n, "moving": int(d_mov.numel()), "tlas_instances": n_tl,
                                      "ms_per_frame": round(t_frame * 1e3, 4), "fps_equivalent": round(1.0 / t_frame, 1)}
            del d_dyn, d_mov, d_t2
            ctx.cull_compact_dev(cam, d_m, len(meshes), d_i, n, d_out, d_cnt, False, first)   # leave d_out / the id table as the later legs expect
        # BASELINE config 5 names a 64k-instance refit: beyond the reference's 16-bit child ids (tlas.rs:71), so in
        # the wide layout; timed with HIP events by the library (wall clock of a 0.1 ms call is mostly launch latency)
        n_w = 65536
        winst = synth.instances(n_w, seed=synth.SEED_BASE + 7, extent=400.0)
        d_wi = ctx.upload(winst)
        d_w = ctx.empty((2 * n_w + 1) * 48)
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.tlas_build_dev(d_wi, n_w, d_m, len(meshes), d_w, wide=True)
        torch.cuda.synchronize(); t_wbuild = time.perf_counter() - t
        ctx.set_timing(True)
        g = []
        for _ in range(5):
            ctx.tlas_refit_dev(d_wi, n_w, d_m, len(meshes), d_w, wide=True)
            g.append(ctx.last_gpu_ms())
        ctx.set_timing(False)
        t_wpipe = pipelined(lambda: ctx.tlas_refit_dev(d_wi, n_w, d_m, len(meshes), d_w, wide=True))
        extra["tlas_wide_64k"] = {"n_instances": n_w, "build_ms": round(t_wbuild * 1e3, 1), "refit_gpu_ms": round(min(g), 4),
                                  "refit_queued_ms": round(t_wpipe * 1e3, 4)}
        del d_wi, d_w
        # traversal (no roofline claim: latency/L1-bound): bvh_gpu.rs-shaped scene, 1 M primary rays
        tv, ti = synth.knot_mesh(512, 128)                        # 131k triangles
        nodes_b, idx_b = ctx.bvh_build(tv, ti)
        infos = np.zeros(1, dtype=abi.MESH_INFO)
        infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(tv)
        infos[0]["index_count"] = len(idx_b)
        inst_t = synth.instances(2000, n_mesh=1, seed=synth.SEED_BASE + 8, extent=120.0, scale_range=(0.5, 2.0))
        tl = ctx.tlas_build(inst_t, infos)
        rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 90), pitch_deg=0), 1024, 1024)
        ds = ctx.device_scene((tl, inst_t, infos, nodes_b, tv, idx_b))
        d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
        d_any = torch.zeros(len(rays), dtype=torch.int32, device=dev)
        ctx.set_timing(True)
        t_cl, t_any = [], []
        for _ in range(3):
            ctx.trace_dev(ds, d_rays, len(rays), d_hits); t_cl.append(ctx.last_gpu_ms())
            ctx.trace_any_dev(ds, d_rays, len(rays), d_any); t_any.append(ctx.last_gpu_ms())
        ctx.set_timing(False)
        hits = d_hits.cpu().numpy()[: len(rays) * 16].view(abi.HIT)
        extra["trace"] = {"n_rays": len(rays), "scene": "2000 instances x 131k-triangle mesh",
                          "closest_hit_Mrays_per_s": round(len(rays) / min(t_cl) / 1e3, 1),
                          "occlusion_Mrays_per_s": round(len(rays) / min(t_any) / 1e3, 1),
                          "hit_fraction": round(float(hits["hit"].mean()), 3),
                          "occlusion_flags_equal_closest_hit": bool(np.array_equal(d_any.cpu().numpy().astype(np.uint32), hits["hit"]))}
        del ds, d_rays, d_hits, d_any
        # the reference's own harness shape (src/bin/bvh_gpu.rs:107-131): one large mesh + four small ones, 4 M primary rays
        inst2, infos2, B2, V2, I2 = synth.harness_scene(ctx.bvh_build)
        tl2 = ctx.tlas_build(inst2, infos2)
        rays2 = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 15), pitch_deg=0), 2048, 2048)
        ds2 = ctx.device_scene((tl2, inst2, infos2, B2, V2, I2))
        d_r2, d_h2 = ctx.upload(rays2), ctx.empty(len(rays2) * 16)
        ctx.set_timing(True)
        t2 = []
        for _ in range(3):
            ctx.trace_dev(ds2, d_r2, len(rays2), d_h2); t2.append(ctx.last_gpu_ms())
        ctx.set_timing(False)
        h2 = d_h2.cpu().numpy()[: len(rays2) * 16].view(abi.HIT)
        extra["trace_harness_scene"] = {"n_rays": len(rays2), "scene": f"bvh_gpu.rs shape: {len(I2)//3} triangles, 5 instances",
                                        "closest_hit_Mrays_per_s": round(len(rays2) / min(t2) / 1e3, 1),
                                        "hit_fraction": round(float(h2["hit"].mean()), 3)}
        # the CPU harness (src/bin/bvh_cpu.rs:39-96): per-pixel rays + Bvh::traverse_iter against ONE mesh, on the device;
        # the harness's own 64-triangle soup at 640 x 640, and the large mesh of the scene above at 2048 x 2048
        cam_h = synth.camera_uniform(eye=(0, 0, 15), pitch_deg=0)
        sv, si = synth.triangle_soup(64)
        sn, si = ctx.bvh_build(sv, si)
        big_v, big_i = synth.knot_mesh(1024, 256)
        big_v = np.ascontiguousarray(big_v * np.float32(3.0))
        bn, bi = ctx.bvh_build(big_v, big_i)
        res = {}
        for tag, (nn, vv, ii, w) in {"soup64_640x640": (sn, sv, si, 640), "knot_524k_2048x2048": (bn, big_v, bi, 2048)}.items():
            d_pr = ctx.empty(w * w * 32)
            d_t = torch.zeros(w * w, dtype=torch.float32, device=dev)
            d_n, d_v, d_ix = ctx.upload(nn), ctx.upload(np.ascontiguousarray(vv, dtype=np.float32)), ctx.upload(ii)
            ctx.primary_rays_dev(cam_h, w, w, d_pr)
            ctx.set_timing(True)
            tt = []
            for _ in range(3):
                ctx.traverse_iter_dev(d_n, len(nn), d_v, d_ix, d_pr, w * w, d_t); tt.append(ctx.last_gpu_ms())
            ctx.set_timing(False)
            res[tag] = {"Mrays_per_s": round(w * w / min(tt) / 1e3, 1), "hit_fraction": round(float((d_t >= 0).float().mean()), 3)}
        extra["traverse_iter_cpu_harness"] = res
        # EXTENSION, no reference counterpart: depth pyramid of a 1920 x 1080 buffer + occlusion refinement
        # of the frustum mask of the headline's 10 M instances (a synthetic depth buffer: half the screen covered)
        W, H = 1920, 1080
        depth = np.zeros((H, W), dtype=np.float32)
        depth[:, : W // 2] = np.float32(0.001 / 40.0)
        Lz = ctx.hiz_layout(W, H)
        d_depth, d_pyr = ctx.upload(depth), torch.zeros(Lz.total_texels, dtype=torch.float32, device=dev)
        d_mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
        d_mask2 = torch.zeros_like(d_mask)
        ctx.cull_mask_dev(cam, d_m, len(meshes), d_i, n, d_mask)
        ctx.set_timing(True)
        t_p, t_o = [], []
        for _ in range(5):
            ctx.hiz_build_dev(d_depth, W, H, d_pyr); t_p.append(ctx.last_gpu_ms())
            ctx.occlusion_mask_dev(cam, d_m, len(meshes), d_i, n, d_pyr, W, H, d_mask, d_mask2); t_o.append(ctx.last_gpu_ms())
        ctx.set_timing(False)
        pop = lambda t: int(sum(bin(int(x) & 0xFFFFFFFFFFFFFFFF).count("1") for x in t.cpu().numpy()[:4096]))
        extra["occlusion_extension"] = {"note": "extension, no reference counterpart", "depth": f"{W}x{H}",
                                        "pyramid_build_ms": round(min(t_p), 4), "occlusion_mask_ms": round(min(t_o), 4),
                                        "instances": n, "GBps_instances_read": round(n * 144 / min(t_o) / 1e6, 1),
                                        "kept_of_first_262144_frustum_visible": [pop(d_mask2), pop(d_mask)]}
        del d_depth, d_pyr, d_mask, d_mask2

    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_cull_pmc.json")
    if rank == 0 and os.path.exists(pmc_path) and n == 10_000_000 and args.dist == "baseline":
        # HBM bytes per launch of the dominant kernel from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
        # passes of this same command (profiles/README.md); gfx950: FETCH_SIZE counts 64 B per 128-B request
        pmc = json.load(open(pmc_path)).get(kernel_name)
        if pmc:
            traffic = int((2.0 * pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024)
    if rank == 0:
        value = n_total / (ms_per_step * 1e-3) / 1e6
        line = {
            "metric": "M instances culled+compacted/sec",
            "value": round(value, 1),
            "unit": "M instances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "fps_equivalent": round(1e3 / ms_per_step, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("configs[2]: 10M synthetic AABB instances, cull + prefix-sum compaction "
                                    "(BASELINE.md §3 distribution, model.rs camera)") if not distributed else
                                   ("configs[3] shape, weak-scaled: 10M synthetic AABB instances PER GPU, sharded by instance; "
                                    "every GPU ends the step with the ordered compacted draw list of the whole scene"),
                       "instances_per_gpu": n, "instances_total": n_total, "n_meshes": int(len(meshes)),
                       "visible_fraction": round(vis, 4), "distribution": args.dist,
                       "parallelism": f"instance-shard x{world}" + (" + visibility-bitmask all-gather (RCCL) + local expansion to the full draw list" if distributed else ""),
                       "verified_bit_exact_vs_oracle": verified, "near_frustum_plane": near, "input_gen_s": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "second_kernel": None if not split else {
                             "kernel": "mask_scan_kernel + expand_mask_u8_kernel", "kernel_ms": round(expand_ms, 4),
                             "algorithmic_bytes_per_launch": int(expand_bytes),
                             "achieved": round(expand_bytes / (expand_ms * 1e-3) / 1e9, 1)},
                         "step": {"kernels_ms": round(step_kernel_ms, 4), "algorithmic_bytes": int(step_bytes),
                                  "achieved": round(step_bytes / (step_kernel_ms * 1e-3) / 1e9, 1),
                                  "frac": round(step_bytes / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
            "cpu_baseline": cpu,
        }
        if extra:
            line["extra"] = extra
        if breakdown:
            line["step_breakdown"] = breakdown
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
