#!/usr/bin/env python3
"""bench.py — headline benchmark of the visibility hot path on MI355X.

A "step" = one pass of cull + emit + ordered compaction over one batch of synthetic instances already
resident in HBM.

N = 1 (BASELINE.json configs[2]): 10 M synthetic AABB instances (BASELINE.md §3 distribution), 16 MeshInfo,
the model.rs camera; `vd_cull_compact_shard_dev` = `cull_mask_tiled_kernel` (dominant, HBM-bound) +
`mask_scan_kernel` + `expand_mask_u8_kernel`.

N > 1 (BASELINE.json configs[3]): the SAME 10 M instances sharded by instance over N GPUs, one process per GPU
(`--scaling strong`, the default; `--scaling weak` = 10 M per GPU).  `--gather` picks what every GPU holds at the
end of a step: `full` (default, the north-star exchange: every GPU ends with the ordered draw list of the whole
scene; wire = 1 bit per instance over RCCL + local expansion), `draws` (the literal all-gather of the 20-byte
commands), `indices` (4 B per survivor), `shard` (own shard only, no exchange).  The other modes and the weak-scaled
run are timed in the same process and reported under `extra`.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF (child
processes, before anything touches the GPU); under `python -m torch.distributed.run ... bench.py --gpus N` it is
one of the ranks.  Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP events on the launch stream)
and `cpu_baseline` (the oracle's restatement of the cull on this box's host cores: a reported baseline).
"""
import argparse
import concurrent.futures
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--instances", type=int, default=10_000_000,
                    help="instances of the scene (strong scaling: in total; weak scaling: per GPU)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = --instances in total (BASELINE configs[3]); weak = --instances per GPU")
    ap.add_argument("--gather", choices=["full", "draws", "indices", "shard"], default="full",
                    help="N > 1: what every GPU holds after a step (see the module docstring)")
    ap.add_argument("--exchange", choices=["rccl", "torch"], default=None,
                    help="N > 1: rccl (default) = the exchange behind the C ABI (vd_dist_*: RCCL on the ctx stream, bound by the "
                         "library); torch = torch.distributed collectives between the two kernels (default when "
                         "VOIDIN_DIST_BACKEND is set; gloo runs several ranks on one GPU)")
    ap.add_argument("--timeout", type=float, default=0.0, help="launcher: kill the ranks and fail after this many seconds (0 = none)")
    ap.add_argument("--dist", choices=["baseline", "small"], default="baseline",
                    help="baseline = BASELINE.md §3 (S in [0.25,4]); small = S in [0.02,0.6] (more culled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (emit_draws, BVH build, TLAS, ...)")
    ap.add_argument("--bvh-u", type=int, default=2048, help="knot mesh resolution: 2*u*v triangles (default 8.4M)")
    ap.add_argument("--bvh-v", type=int, default=2048)
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins the process group and rank 0 prints the world it sees")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` -> N child ranks.  Nothing here imports torch or touches HIP.
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, timeout=0.0):
    import signal
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))

    def stop_all(grace=5.0):
        """terminate() exactly the children started above, kill() what is still there after the grace period"""
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + grace
        while time.time() < t_end and any(p.poll() is None for p in procs):
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()

    def on_signal(signum, _frame):      # the launcher itself is being stopped (e.g. `timeout 600 python bench.py --gpus 2`)
        print(f"bench.py: launcher got signal {signum}; stopping the ranks", file=sys.stderr, flush=True)
        stop_all()
        sys.exit(128 + signum)

    old_handlers = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    deadline = time.time() + timeout if timeout and timeout > 0 else None
    rc = 0
    alive = set(range(n))
    try:
        while alive:
            for r in sorted(alive):
                c = procs[r].poll()
                if c is None:
                    continue
                alive.discard(r)
                if c != 0 and rc == 0:
                    rc = c
                    print(f"bench.py: rank {r} exited with {c}; stopping the other ranks", file=sys.stderr, flush=True)
                    stop_all()
            if deadline and alive and time.time() > deadline:
                print(f"bench.py: ranks {sorted(alive)} still running after {timeout:.0f} s; stopping them", file=sys.stderr, flush=True)
                stop_all()
                rc = rc or 124
                break
            time.sleep(0.05)
    finally:
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    return rc


def launch_check():
    """`--launch-check`: the rendezvous of the N ranks without any GPU work (CPU test of the launcher)."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    backend = os.environ.get("VOIDIN_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            li = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(li)
            dist.init_process_group("nccl", device_id=torch.device("cuda", li))
        else:
            dist.init_process_group(backend)
        dev = "cuda" if backend == "nccl" else "cpu"
        seen = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(seen, torch.tensor([rank], dtype=torch.int64, device=dev))
        n_gpus, ranks = dist.get_world_size(), [int(x) for x in seen.cpu()]
        dist.barrier()
        dist.destroy_process_group()
    else:
        n_gpus, ranks = 1, [0]
    hold = float(os.environ.get("VOIDIN_LAUNCH_CHECK_HOLD_S", "0"))     # tests of the launcher's signal / deadline handling
    if hold > 0:
        time.sleep(hold)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": n_gpus, "ranks": ranks, "backend": backend if world > 1 else None}), flush=True)


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def tree_shape(nodes):
    """depth, interior nodes and sum over interior nodes of their primitive count (= Σ_l A_l of SURVEY 8d: every
    interior node of n prims was partitioned once) of a VdBvhNode array, level by level in numpy."""
    import numpy as np
    left, cnt = nodes["left_first"].astype(np.int64), nodes["count"].astype(np.int64)
    levels = [np.zeros(1, dtype=np.int64)]
    while True:
        f = levels[-1]
        inner = f[cnt[f] == 0]
        if inner.size == 0:
            break
        levels.append(np.concatenate([left[inner], left[inner] + 1]))
    prims = cnt.copy()
    for f in reversed(levels[:-1]):
        inner = f[cnt[f] == 0]
        prims[inner] = prims[left[inner]] + prims[left[inner] + 1]
    per_level = [int(prims[f[cnt[f] == 0]].sum()) for f in levels]
    return {"depth": len(levels) - 1, "interior_nodes": int((cnt == 0).sum()) - 1,   # node 1 is the unused all-zero slot
            "sum_active_prims": int(sum(per_level)), "levels_with_work": int(sum(1 for a in per_level if a))}


def scaling_ceiling(n_total, world, vis, weak):
    """DESIGN.md 6's model of one step at N GPUs, from the one-GPU kernel rates measured in round 2 (pass 1 reads at
    6.5 TB/s, launch floor 8 us; the expansion writes at 5.5 TB/s after a 6 us scan) and an ASSUMED 25 us for the
    latency-bound all-gather of the bitmasks (unmeasured until an 8-GPU run exists).  `full`: every GPU expands the
    whole list, so that leg does not shrink with N; `shard`: no exchange."""
    shard = (n_total + world - 1) // world
    cull = max(8e-3, shard * 145.125 / 6.5e12 * 1e3)
    expand_all = 6e-3 + n_total * (1.125 + 20.0 * vis) / 5.5e12 * 1e3
    expand_own = 6e-3 + shard * (1.125 + 20.0 * vis) / 5.5e12 * 1e3
    full = cull + (25e-3 if world > 1 else 0.0) + expand_all
    own = cull + expand_own
    return {"full_ms_per_step": round(full, 4), "shard_ms_per_step": round(own, 4),
            "full_M_inst_per_s": round(n_total / full / 1e3, 1), "shard_M_inst_per_s": round(n_total / own / 1e3, 1),
            "model": "max(8us, shard*145.1B/6.5TB/s) + [25us all-gather, assumed] + 6us + list bytes/5.5TB/s (DESIGN.md 6); "
                     "full = every GPU writes the whole list, shard = its own part only"}


def latest_pmc(kind="cull"):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{kind}_pmc.json")))
    return files[-1] if files else None


def promote_second_metric(line, extra):
    """BASELINE.json's metric names TWO numbers ("M instances culled+compacted/sec; SAH BVH build Mprims/sec") and config 5 a TLAS
    refit.  `value` carries the first; the second and the refit are measured by the N = 1 extras - this copies a compact summary of
    them into `roofline` / `cpu_baseline`, the objects a reader of the line keeps, next to the consumer's padded form of the headline
    (`roofline.pad_tail`).  Nothing is measured here."""
    b = extra.get("bvh_build")
    if b:
        rf = b["roofline"]
        out = {"metric": "SAH BVH build Mprims/s", "n_tris": b["n_tris"], "ms": b["ms"], "Mprims_per_s": b["value"],
               "ms_phase_a": (b.get("phases_ms") or {}).get("ms_phase_a"), "ms_phase_b": (b.get("phases_ms") or {}).get("ms_phase_b"),
               "kernel_launches": (b.get("phases_ms") or {}).get("kernel_launches"),
               "frac_770B": rf["emulating_770B_per_prim_level"]["frac"], "frac_44B": rf["binned_44B_per_prim_level"]["frac"],
               "bit_exact_vs_oracle": b.get("bit_exact_vs_oracle")}
        pmc_path = latest_pmc("bvh")
        if pmc_path and b["n_tris"] == 8_388_608:
            tot = json.load(open(pmc_path)).get("total") or {}
            if tot.get("sum_MB"):
                traffic = int(tot["sum_MB"] * 1e6)
                out["traffic_bytes"] = traffic
                out["traffic_source"] = (os.path.join("profiles", os.path.basename(pmc_path)) +
                                         " (2 x FETCH_SIZE + WRITE_SIZE over every kernel of one build of this mesh, separate rocprofv3 --pmc "
                                         "passes, committed; not re-measured by this run)")
                out["frac_real"] = round(traffic / (b["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)      # real bytes / this run's time / 8 TB/s
        line["roofline"]["bvh_build"] = out
        if line.get("cpu_baseline") is not None and b.get("cpu_baseline"):
            c = b["cpu_baseline"]
            line["cpu_baseline"]["bvh_build"] = {"Mprims_per_s": c["value"], "cores": c["cores"], "kind": c["kind"], "sample": c["sample"]}
    t, w = extra.get("tlas"), extra.get("tlas_wide_64k")
    if t:
        line["roofline"]["tlas_refit_ms"] = {"32768": t["refit_queued_ms"], "65536_wide": (w or {}).get("refit_gpu_ms"),
                                             "build_ms_32768": t["build_ms"], "build_bit_exact_vs_oracle": t.get("bit_exact_vs_oracle"),
                                             "refit_after_motion_bit_exact_vs_oracle_65536": (w or {}).get("refit_after_motion_bit_exact_vs_oracle")}
        if line.get("cpu_baseline") is not None and t.get("cpu_baseline"):
            line["cpu_baseline"]["tlas_build_ms_32768"] = t["cpu_baseline"]["value"]
    p = extra.get("cull_compact_pad_tail")
    if p:
        line["roofline"]["pad_tail"] = {k: {"ms": p[k]["ms"], "M_inst_per_s": p[k]["M_inst_per_s"], "visible_fraction": p[k]["visible_fraction"],
                                            "bit_exact_vs_oracle": p[k]["whole_padded_buffer_bit_exact_vs_oracle"]} for k in ("baseline", "dist_small")}


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def run_rank(args):
    import numpy as np
    import torch

    from voidin_amd import abi, synth
    from voidin_amd import dist as vdist
    from voidin_amd.runtime import Context

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world_env > 1
    exchange = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # One rank per GPU; the data path is RCCL over xGMI either way:
        #   --exchange rccl (default): the exchange lives behind the C ABI (vd_dist_*, csrc/dist.hip): cull_mask ->
        #       ncclAllGather ON THE CTX STREAM -> expand_mask in one call; torch.distributed is only the control plane
        #       (the 128-byte communicator id, the barrier and the max over ranks of the timed region) and runs on gloo;
        #   --exchange torch: torch.distributed collectives between the two kernels, backend VOIDIN_DIST_BACKEND (nccl =
        #       RCCL; gloo lets several ranks share ONE GPU - functional check only, RCCL refuses two ranks per device).
        env_backend = os.environ.get("VOIDIN_DIST_BACKEND")
        exchange = args.exchange or ("torch" if env_backend else "rccl")
        backend = env_backend or ("gloo" if exchange == "rccl" else "nccl")
        n_dev = torch.cuda.device_count()
        # VOIDIN_RANKS_SHARE_GPUS=1: the ranks may share devices (rank -> device local_rank % n_dev).  Real RCCL refuses two ranks
        # per device; a host that binds another implementation through $VD_RCCL_LIB (the tests' double) can run the whole N-rank
        # command - launcher, input generation, C-ABI exchange, rank-0 verification - on one GPU: functional check, not a timing.
        share = os.environ.get("VOIDIN_RANKS_SHARE_GPUS") == "1"
        if (backend == "nccl" or exchange == "rccl") and n_dev < world_env and not share:
            raise SystemExit(f"bench.py: {world_env} ranks over RCCL need {world_env} GPUs, {n_dev} visible "
                             "(VOIDIN_DIST_BACKEND=gloo runs the ranks on fewer devices: functional check only)")
        dev_index = local_rank % max(n_dev, 1)
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()          # what the process group reports, not what the flag asked for
    else:
        dist, backend, world, dev_index = None, None, 1, 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    ctx = Context(dev.index)  # raises if the HIP extension or a gfx950 GPU is missing

    weak = distributed and args.scaling == "weak"
    n_total = args.instances * world if weak else args.instances
    lo, hi = vdist.shard_range(n_total, rank, world)
    n = hi - lo                                    # this rank's shard
    kw = dict(scale_range=(0.25, 4.0)) if args.dist == "baseline" else dict(scale_range=(0.02, 0.6), extent=600.0)
    cam = synth.camera_uniform()
    meshes = synth.mesh_infos()
    n_mesh = len(meshes)
    t0 = time.time()
    inst = synth.instances(n, seed=synth.SEED_BASE + 3, offset=lo, with_inverse=False, **kw)
    t_gen = time.time() - t0

    d_m = ctx.upload(meshes)
    d_cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    # the very first call on a fresh context: scratch allocation + first-touch of the id table, nothing warm
    d_i = ctx.upload(inst)
    d_out = ctx.empty(max(n, 1) * 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.cull_compact_dev(cam, d_m, n_mesh, d_i, n, d_out, d_cnt, False, lo)
    torch.cuda.synchronize()
    cold_first_step_ms = (time.perf_counter() - t0) * 1e3

    # Consecutive steps see DIFFERENT instance buffers - the second is the first after one compute_update animation
    # step (every transform changed, mesh assignment unchanged: what the reference's dynamic scene does,
    # shaders/compute_update.wgsl:10-28) - so nothing but the per-scene instance->mesh table can carry over between
    # steps (pass 1 re-derives that table every step and rewrites only rows that changed).
    d_i_b = d_i.clone()
    d_all_idx = torch.arange(n, dtype=torch.int32, device=dev)
    ctx.compute_update_dev(d_all_idx, n, d_i_b, n, 1.0, 0.016)
    torch.cuda.synchronize()
    del d_all_idx

    # N > 1: the sharded scene.  d_all = the whole scene's list (full / draws / indices), on every rank.
    sv, rccl_info, data_group = None, None, None
    if distributed and exchange == "rccl":
        try:
            sv = vdist.RcclVisibility(ctx, n_total, d_m, n_mesh, d_i)
            ok = 1
        except Exception as e:      # VD_ERR_COMM: RCCL not loadable / communicator refused -> every rank falls back together
            ok, why = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # The data path stays RCCL or the run fails: the control-plane group here is gloo, and a host-staged number
            # under an "RCCL" headline would be worse than no number.  Fall back to torch.distributed's own RCCL group.
            if sv is not None:
                sv.close()
            sv = None
            if rank == 0:
                print(f"bench.py: C-ABI RCCL exchange unavailable ({why if not ok else 'another rank failed'}); "
                      "trying torch.distributed over a new nccl (= RCCL) group", file=sys.stderr, flush=True)
            try:
                data_group = dist.new_group(backend="nccl", device_id=dev) if backend != "nccl" else None
                probe = torch.zeros(world, dtype=torch.int32, device=dev)
                dist.all_gather_into_tensor(probe, torch.ones(1, dtype=torch.int32, device=dev), group=data_group)
                torch.cuda.synchronize()
                good = int(probe.sum().item()) == world
            except Exception as e:
                good, why = False, repr(e)
            flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                raise SystemExit(f"bench.py: rank {rank}: no RCCL data path (C-ABI exchange and torch.distributed/nccl both failed: "
                                 f"{why if not good else 'another rank failed'}); refusing to time a gloo exchange under --exchange rccl")
            exchange = "torch (fallback, nccl group)"
        else:
            rccl_info = {"version": int(sv.info.rccl_version), "library": sv.info.rccl_library.decode()}
    if distributed and sv is None:
        sv = vdist.ShardedVisibility(ctx, n_total, d_m, n_mesh, d_i, group=data_group)
    via_c = distributed and exchange == "rccl"
    d_all = ctx.empty(n_total * 20) if distributed else None
    d_cnt_all = torch.zeros(4, dtype=torch.int32, device=dev) if distributed else None
    step_no = [0]

    def step_fn(mode):
        def f():
            src = d_i if (step_no[0] & 1) == 0 else d_i_b
            step_no[0] += 1
            if not distributed:
                ctx.cull_compact_dev(cam, d_m, n_mesh, src, n, d_out, d_cnt, False, lo)
                return
            sv.d_inst = src
            if mode == "full":
                sv.step(cam, d_all, d_cnt_all)
            elif mode == "draws":
                sv.step_draws(cam, d_all, d_cnt_all)
            elif mode == "indices":
                sv.step_indices(cam, d_all, d_cnt_all)
            else:
                sv.step_shard(cam, d_out, d_cnt)
        return f

    def barrier():
        torch.cuda.synchronize()          # this rank's GPU work is done (every stream of the device) ...
        if distributed:
            dist.barrier()                # ... and so is every other rank's
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        tw = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        return float(tw.item())

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        return max_over_ranks(time.perf_counter() - t0) * 1e3 / steps

    mode = args.gather if distributed else "full"
    step = step_fn(mode)
    ms_per_step = timed(step, args.steps, args.warmup)

    breakdown, gather_modes, weak_line, bvh_replicas = None, None, None, None
    if distributed:
        # where a step goes at this N (every rank runs every leg; max over ranks, like the headline): the local cull to a
        # bitmask, the RCCL all-gather of the masks alone, and the expansion of ALL shards to the full draw list
        leg = lambda fn: round(timed(fn, args.steps, 1), 4)
        sv.d_inst = d_i
        if via_c:
            legs = (lambda: sv.cull_to_mask(cam), sv.allgather_masks, lambda: sv.expand_all(d_all, d_cnt_all))
        else:
            legs = (lambda: ctx.cull_mask_dev(cam, d_m, n_mesh, d_i, n, sv.d_mask),
                    lambda: dist.all_gather_into_tensor(sv.d_mask_all, sv.d_mask, group=data_group),
                    lambda: ctx.expand_mask_dev(sv.d_mask_all, n_total, sv.S, sv.d_mesh_ids, d_m, n_mesh, d_all, d_cnt_all, id_bytes=sv.id_bytes))
        breakdown = {
            "cull_to_mask_ms": leg(legs[0]),
            "mask_allgather_ms": leg(legs[1]),
            "expand_all_shards_ms": leg(legs[2]),
            "mask_bytes_per_rank": int(sv.wps * 8),
            "note": "gather=full: every GPU materialises the whole list, so the expansion leg does not shrink with N (DESIGN.md 6)"}
        gather_modes = {}
        for m_ in ("full", "draws", "indices", "shard"):
            t_ = ms_per_step if m_ == mode else timed(step_fn(m_), args.steps, 2)
            gather_modes[m_] = {"ms_per_step": round(t_, 4), "M_inst_per_s": round(n_total / t_ / 1e3, 1)}
        gather_modes["note"] = ("full/draws/indices: every GPU ends with the whole ordered list (wire: 1 bit per instance / 20 B per survivor / "
                                "4 B per survivor; draws and indices read the counts back on the host); shard: each GPU keeps its own "
                                "shard's list, no exchange")
        if not weak and not args.no_extra:
            # the weak-scaled run next to the strong headline: --instances PER GPU, same mode
            n_w = args.instances
            inst_w = synth.instances(n_w, seed=synth.SEED_BASE + 3, offset=rank * n_w, with_inverse=False, **kw)
            d_iw = ctx.upload(inst_w)
            del inst_w
            sv_w = (vdist.RcclVisibility(ctx, n_w * world, d_m, n_mesh, d_iw) if via_c else
                    vdist.ShardedVisibility(ctx, n_w * world, d_m, n_mesh, d_iw, group=data_group))
            d_all_w, d_cnt_w = ctx.empty(n_w * world * 20), torch.zeros(4, dtype=torch.int32, device=dev)
            t_full = timed(lambda: sv_w.step(cam, d_all_w, d_cnt_w), args.steps, 2)
            t_shard = timed(lambda: sv_w.step_shard(cam, d_all_w, d_cnt_w), args.steps, 2)
            weak_line = {"instances_per_gpu": n_w, "instances_total": n_w * world,
                         "full": {"ms_per_step": round(t_full, 4), "M_inst_per_s": round(n_w * world / t_full / 1e3, 1)},
                         "shard": {"ms_per_step": round(t_shard, 4), "M_inst_per_s": round(n_w * world / t_shard / 1e3, 1)}}
            if via_c:
                sv_w.close()
            del sv_w, d_all_w, d_iw
        if not args.no_extra:
            # BASELINE's second metric at N GPUs.  The BLAS build of one mesh does not shard (every split is a
            # global reduction over the node's segment, and the arrangement is order-dependent): REPLICAS ONLY - every rank
            # builds a mesh of its own, no collective; aggregate = all ranks' triangles / the slowest rank's time.
            import zlib
            bu, bv_ = min(args.bvh_u, 1024), min(args.bvh_v, 1024)           # 2.1 M triangles per rank: bounded host time per rank
            mv, mi = synth.knot_mesh(bu, bv_)
            nt = len(mi) // 3
            d_v, d_n = ctx.upload(mv), ctx.empty(2 * nt * 32)
            t_b = []
            for r_ in range(3):
                d_ix = ctx.upload(mi)                                      # the build permutes the index buffer in place
                barrier()
                t0 = time.perf_counter()
                nn = ctx.bvh_build_dev(d_v, len(mv), d_ix, nt, d_n, 2 * nt)
                torch.cuda.synchronize()
                t_b.append(max_over_ranks(time.perf_counter() - t0))
            my_crc = zlib.crc32(d_n[: nn * 32].cpu().numpy().tobytes())
            same = max_over_ranks(float(my_crc)) == float(my_crc) and max_over_ranks(-float(my_crc)) == -float(my_crc)   # max == min == mine
            bvh_replicas = {"metric": "SAH BVH build Mprims/s, aggregate over ranks", "value": round(world * nt / min(t_b) / 1e6, 1),
                            "tris_per_rank": nt, "ms_slowest_rank": round(min(t_b) * 1e3, 2), "ranks": world,
                            "same_nodes_on_every_rank": bool(same), "nodes_crc32": my_crc,
                            "note": "replicas only (SURVEY 8e): one mesh per rank, no collective; best of 3, max over ranks per build"}
            del d_v, d_n, d_ix
        # leave d_all / d_cnt_all as a step of the headline mode on d_i leaves them
        step_no[0] = 0
        step()
        torch.cuda.synchronize()

    # the local kernels once more on d_i (whatever buffer the timed loop ended on): count, roofline, verification
    ctx.cull_compact_dev(cam, d_m, n_mesh, d_i, n, d_out, d_cnt, False, lo)
    torch.cuda.synchronize()
    count = int(d_cnt[0].item())

    # Kernel-level timing: HIP events recorded by the library on the launch stream around each pass of
    # vd_cull_compact (large inputs run two passes: cull -> bitmask + mesh id, then expansion).
    barrier()
    ctx.set_timing(True)
    k_all, k_cull, k_expand = [], [], []
    for _ in range(args.steps):
        ctx.cull_compact_dev(cam, d_m, n_mesh, d_i, n, d_out, d_cnt, False, lo)
        k_all.append(ctx.last_gpu_ms())
        k_cull.append(ctx.last_gpu_ms_stage(0))
        k_expand.append(ctx.last_gpu_ms_stage(1))
    ctx.set_timing(False)
    step_kernel_ms = sum(k_all) / len(k_all)
    split = min(k_cull) > 0
    vis = count / max(n, 1)
    step_bytes = n * (144.0 + 20.0 * vis)          # 144 B read + 20 B per survivor
    if split:
        id_bytes = vdist.id_width(n_mesh)
        kernel_name = "cull_mask_tiled_kernel"
        kernel_ms = sum(k_cull) / len(k_cull)
        alg_bytes = n * (144.0 + 0.125 + id_bytes)  # 144 B instance read + 1 bit written + the compact mesh id (compared; rewritten when it changed)
        expand_ms = sum(k_expand) / len(k_expand)
        expand_bytes = n * (0.125 + id_bytes) + count * 20.0
    else:
        kernel_name, kernel_ms, alg_bytes, expand_ms, expand_bytes = "cull_compact_kernel", step_kernel_ms, step_bytes, None, None
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    verified, cpu, near, list_crc = None, None, None, None
    if rank == 0:
        from oracle import ref  # checker + cpu_baseline leg only
        threads = os.cpu_count() or 1
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
            want = ref.cull_emit(cam, meshes, inst, threads=threads)
            want["base_instance"] += np.uint32(lo)
            wc, wn = ref.compact(want)
            got = d_out.cpu().numpy()[: count * 20]
            verified = bool(wn == count and got.tobytes() == wc[:wn].tobytes())
            if distributed and mode != "shard":
                total = int(d_cnt_all[0].item())
                if n_total <= 20_000_000:    # the WHOLE gathered list against the oracle on the whole scene
                    inst_all = synth.instances(n_total, seed=synth.SEED_BASE + 3, with_inverse=False, **kw)
                    wa, wan = ref.compact(ref.cull_emit(cam, meshes, inst_all, threads=threads))
                    del inst_all
                    verified = bool(verified and total == wan and d_all.cpu().numpy()[: total * 20].tobytes() == wa[:wan].tobytes())
                else:                        # rank 0 owns the first shard: the head of the list must be its compaction
                    verified = bool(verified and total >= count and d_all.cpu().numpy()[: count * 20].tobytes() == wc[:wn].tobytes())
        # CRC-32 of the ordered draw list of the WHOLE scene as this run left it (equal at every N and in every gather
        # mode that materialises the whole list)
        import zlib
        if not distributed:
            list_crc = zlib.crc32(d_out.cpu().numpy()[: count * 20].tobytes())
        elif mode != "shard":
            list_crc = zlib.crc32(d_all.cpu().numpy()[: int(d_cnt_all[0].item()) * 20].tobytes())
        if not args.no_cpu_baseline and not distributed:      # reported at N = 1 only
            m = min(n, 10_000_000)
            reps1 = 3
            t = time.perf_counter()
            for _ in range(reps1):
                ref.cull_emit(cam, meshes, inst[:m], threads=1)
            t1 = (time.perf_counter() - t) / reps1
            repsN = 10
            t = time.perf_counter()
            for _ in range(repsN):
                d = ref.cull_emit(cam, meshes, inst[:m], threads=threads)
            tN = (time.perf_counter() - t) / repsN
            t = time.perf_counter()
            ref.compact(d)
            tc = time.perf_counter() - t
            cpu = {"value": round(m / (tN + tc) / 1e6, 2), "unit": "M instances culled+compacted/s", "cores": threads,
                   "kind": "port",
                   "sample": f"{m} instances of the same workload; cull on {threads} threads x{repsN} + serial compaction; "
                             f"1-thread cull: {m / t1 / 1e6:.2f} M inst/s"}

    extra = {"cold_first_step_ms": round(cold_first_step_ms, 3)}
    if bvh_replicas:
        extra["bvh_build_replicas"] = bvh_replicas
    if gather_modes:
        extra["gather_modes"] = gather_modes
    if weak_line:
        extra["weak_scaling"] = weak_line
    if not args.no_extra and rank == 0 and not distributed:   # single-GPU extras (no collective; N > 1 runs skip them)
        extra.update(single_gpu_extras(args, ctx, dev, cam, meshes, d_m, inst, d_i, d_out, d_cnt, n, lo, count))

    traffic = None
    pmc_path = latest_pmc()
    if rank == 0 and pmc_path and n == 10_000_000 and args.dist == "baseline":
        # HBM bytes per launch of the dominant kernel from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
        # passes of this same command (profiles/README.md); gfx950: FETCH_SIZE counts 64 B per 128-B request
        pmc = json.load(open(pmc_path)).get(kernel_name)
        if pmc:
            traffic = int((2.0 * pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024)
    if rank == 0:
        value = n_total / (ms_per_step * 1e-3) / 1e6
        if not distributed:
            workload = ("configs[2]: 10M synthetic AABB instances, cull + prefix-sum compaction "
                        "(BASELINE.md §3 distribution, model.rs camera)")
            par = "single GPU"
        else:
            workload = (("configs[3]: 10M synthetic AABB instances sharded by instance over the GPUs" if not weak else
                         "configs[3] shape, weak-scaled: 10M synthetic AABB instances PER GPU, sharded by instance") + "; " +
                        {"full": "every GPU ends the step with the ordered compacted draw list of the whole scene",
                         "draws": "every GPU ends the step with the whole list (literal all-gather of the 20-byte commands)",
                         "indices": "every GPU ends the step with the whole list (survivor indices exchanged, commands rebuilt locally)",
                         "shard": "every GPU keeps the ordered compacted list of its own shard (no exchange)"}[mode])
            par = (f"instance-shard x{world}, one process per GPU; exchange: " +
                   (f"C ABI vd_dist_* over RCCL {rccl_info['version']} on the ctx stream (control plane: torch.distributed/{backend})" if via_c
                    else f"torch.distributed/{backend} ({exchange})") + f"; gather={mode}" +
                   {"full": ": visibility-bitmask all-gather (RCCL) + local expansion to the full draw list",
                    "draws": ": counts all-gather + exact-size direct all-gather of the commands (RCCL send/recv)",
                    "indices": ": counts all-gather + exact-size direct all-gather of u32 survivor indices + local rebuild",
                    "shard": ""}[mode])
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "instances_per_gpu": n, "instances_total": n_total, "n_meshes": int(n_mesh),
                       "visible_fraction": round(vis, 4), "distribution": args.dist, "parallelism": par,
                       "verified_bit_exact_vs_oracle": verified, "draw_list_crc32": list_crc, "near_frustum_plane": near,
                       "input_gen_s": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": (os.path.join("profiles", os.path.basename(pmc_path)) +
                                                                " (separate rocprofv3 --pmc passes of this command, committed; not re-measured by this run)") if traffic else None,
                         "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "second_kernel": None if not split else {
                             "kernel": "mask_scan_kernel + expand_mask_u8_kernel", "kernel_ms": round(expand_ms, 4),
                             "algorithmic_bytes_per_launch": int(expand_bytes),
                             "achieved": round(expand_bytes / (expand_ms * 1e-3) / 1e9, 1)},
                         "step": {"kernels_ms": round(step_kernel_ms, 4), "algorithmic_bytes": int(step_bytes),
                                  "achieved": round(step_bytes / (step_kernel_ms * 1e-3) / 1e9, 1),
                                  "frac": round(step_bytes / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
            "cpu_baseline": cpu,
            "extra": extra,
        }
        promote_second_metric(line, extra)
        line["scaling_ceiling"] = scaling_ceiling(n_total, world, vis if not distributed else count / max(n, 1), weak)
        if rccl_info:
            line["config"]["rccl"] = rccl_info
        if breakdown:
            line["step_breakdown"] = breakdown
            line["scaling_ceiling"]["mask_allgather_ms_measured"] = breakdown["mask_allgather_ms"]
            line["scaling_ceiling"]["mask_allgather_ms_assumed"] = 0.025
        if gather_modes:
            # both ends of the design space in every N > 1 line, whatever --gather selected for `value`: `full` = every GPU
            # holds the whole list (cannot scale past the list write), `shard` = each GPU holds its own part (can)
            line["value_full"] = gather_modes["full"]["M_inst_per_s"]
            line["value_shard"] = gather_modes["shard"]["M_inst_per_s"]
        print(json.dumps(line), flush=True)
    if via_c:
        sv.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


# ------------------------------------------------------------------------------------------------
# N = 1 secondary legs: the other BASELINE configs and the rows of SURVEY.md 8a beside the headline
# ------------------------------------------------------------------------------------------------
def single_gpu_extras(args, ctx, dev, cam, meshes, d_m, inst, d_i, d_out, d_cnt, n, first, count):
    import numpy as np
    import torch

    from oracle import ref   # checker + cpu_baseline legs only
    from voidin_amd import abi, synth
    from voidin_amd import dist as vdist

    n_mesh = len(meshes)
    extra = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def pipelined(fn, reps=50):          # per-call time when calls are queued back to back (a frame loop does not sync per call)
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def event_ms(fn, reps):              # torch's current stream IS the ctx stream (Context binds it)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # CPU oracles that take tens of seconds run on host threads from here on, while the GPU legs below proceed
    # (ctypes releases the GIL): the BLAS of the timed mesh and the 32768-instance TLAS, both 1 thread each.
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=3)
    v, idx = synth.knot_mesh(args.bvh_u, args.bvh_v)
    n_tri = len(idx) // 3
    n_tl = 32768
    tinst = synth.instances(n_tl, seed=synth.SEED_BASE + 6, extent=300.0)

    def timed_call(fn, *a):
        t = time.perf_counter()
        r = fn(*a)
        return r, time.perf_counter() - t
    verify = not args.no_verify
    fut_bvh = pool.submit(timed_call, ref.bvh_build, v, idx) if verify else None
    fut_tlas = pool.submit(timed_call, ref.tlas_build, tinst, meshes) if verify else None

    # the smaller BASELINE configs (configs[0] 1 k, configs[1] 100 k instances; 1 M = the largest input of the fused
    # single-launch form): cull + compaction per call, calls queued back to back; bit-exactness of these sizes is in tests/
    small = {}
    for m_ in (1000, 100_000, 1_000_000):
        if m_ <= n:
            t_ = pipelined(lambda: ctx.cull_compact_dev(cam, d_m, n_mesh, d_i, m_, d_out, d_cnt, False, first), reps=200)
            small[str(m_)] = {"us_per_call": round(t_ * 1e6, 2), "M_inst_per_s": round(m_ / t_ / 1e6, 1)}
    extra["cull_compact_small_inputs"] = small
    d_emit = ctx.empty(n * 20)
    ems = event_ms(lambda: ctx.cull_emit_dev(cam, d_m, n_mesh, d_i, n, d_emit), args.steps)
    extra["emit_draws_uncompacted"] = {"ms": round(ems, 4), "M_inst_per_s": round(n / ems / 1e3, 1),
                                       "GBps": round(n * 164.0 / ems / 1e6, 1),
                                       "frac_of_8TBps": round(n * 164.0 / ems / 1e6 / HBM_PEAK_GBS, 4)}
    del d_emit
    # the multi-GPU wire formats on one GPU: cull -> bitmask -> ordered draw list; cull -> indices -> draw list
    sv1 = vdist.ShardedVisibility(ctx, n, d_m, n_mesh, d_i)
    d_o2, d_c2 = ctx.empty(n * 20), torch.zeros(4, dtype=torch.int32, device=dev)
    t_mask = event_ms(lambda: sv1.step(cam, d_o2, d_c2), args.steps)
    same = bool(int(d_c2[0].item()) == count and torch.equal(d_o2[: count * 20], d_out[: count * 20]))
    extra["mask_then_expand_1gpu"] = {"ms": round(t_mask, 4), "equals_fused": same}
    d_o2.zero_()
    t_idx = pipelined(lambda: sv1.step_indices(cam, d_o2, d_c2), reps=10)
    same = bool(int(d_c2[0].item()) == count and torch.equal(d_o2[: count * 20], d_out[: count * 20]))
    extra["indices_then_rebuild_1gpu"] = {"ms": round(t_idx * 1e3, 4), "equals_fused": same}
    del d_o2, sv1
    # worst case for pass 1's id table: every instance changes its mesh between consecutive steps (all rows rewritten)
    inst_c = inst.copy()
    inst_c["mesh"] = (inst_c["mesh"] + 1) % n_mesh
    d_i_c = ctx.upload(inst_c)
    del inst_c
    k_ = [0]

    def flip():
        k_[0] += 1
        ctx.cull_compact_dev(cam, d_m, n_mesh, d_i if k_[0] & 1 else d_i_c, n, d_out, d_cnt, False, first)
    extra["cull_compact_all_mesh_ids_changing"] = {"ms": round(event_ms(flip, args.steps), 4),
                                                   "note": "every step rewrites the whole instance->mesh table; the headline steps alternate two buffers with equal mesh ids and different transforms"}
    del d_i_c
    # the write-light regime: the small-scale distribution (most instances culled), same size
    inst_s = synth.instances(n, seed=synth.SEED_BASE + 3, offset=first, with_inverse=False, scale_range=(0.02, 0.6), extent=600.0)
    d_i_s = ctx.upload(inst_s)
    t_s = event_ms(lambda: ctx.cull_compact_dev(cam, d_m, n_mesh, d_i_s, n, d_out, d_cnt, False, first), args.steps)
    cnt_s = int(d_cnt[0].item())
    ok_s = None
    if verify:
        ws, wsn = ref.compact(ref.cull_emit(cam, meshes, inst_s, threads=os.cpu_count() or 1))
        ok_s = bool(wsn == cnt_s and d_out.cpu().numpy()[: cnt_s * 20].tobytes() == ws[:wsn].tobytes())
    extra["cull_compact_dist_small"] = {"ms": round(t_s, 4), "M_inst_per_s": round(n / t_s / 1e3, 1), "visible_fraction": round(cnt_s / n, 4),
                                        "step_GBps": round(n * (144.0 + 20.0 * cnt_s / n) / t_s / 1e6, 1), "verified_bit_exact_vs_oracle": ok_s}
    # The form the UNCHANGED consumer needs - multi_draw_indexed_indirect(buf, 0, N), crates/app/src/pass/visibility.rs:188-192:
    # pad_tail = 1 zero-fills out[count..N) behind the compacted list (INTEGRATION.md 5's per-frame call).  Both clouds: the
    # baseline one (95.6 % visible: a short tail) and `dist small` (37 % visible: 126 MB of tail stores per step).
    def pad_leg(d_src, host_inst, want_list=None):
        t_ = event_ms(lambda: ctx.cull_compact_dev(cam, d_m, n_mesh, d_src, n, d_out, d_cnt, True, first), args.steps)
        c_ = int(d_cnt[0].item())
        ok_ = None
        if verify:
            if want_list is None:
                w_ = ref.cull_emit(cam, meshes, host_inst, threads=os.cpu_count() or 1)
                w_["base_instance"] += np.uint32(first)
                want_list = ref.compact(w_, pad_tail=True)
            wl, wln = want_list
            ok_ = bool(wln == c_ and d_out.cpu().numpy()[: n * 20].tobytes() == wl[:n].tobytes())
        return {"ms": round(t_, 4), "M_inst_per_s": round(n / t_ / 1e3, 1), "visible_fraction": round(c_ / n, 4),
                "tail_bytes": int((n - c_) * 20), "whole_padded_buffer_bit_exact_vs_oracle": ok_}
    pad_small = pad_leg(d_i_s, inst_s)
    del d_i_s, inst_s
    extra["cull_compact_pad_tail"] = {"baseline": pad_leg(d_i, inst), "dist_small": pad_small,
                                      "note": "vd_cull_compact_shard_dev(.., pad_tail = 1): the compacted list + instance_count = 0 commands up to slot N, "
                                              "what multi_draw_indexed_indirect(buf, 0, N) consumes unchanged (visibility.rs:188-192)"}
    ctx.cull_compact_dev(cam, d_m, n_mesh, d_i, n, d_out, d_cnt, False, first)   # leave d_out / the table as the later legs expect
    torch.cuda.synchronize()

    # --- BASELINE config 5: SAH BVH build of a dragon-like 8M-tri mesh, TLAS build/refit ---
    d_v = ctx.upload(v)
    d_n = ctx.empty(2 * n_tri * 32)
    best, best_stats = None, None
    for r in range(3):
        d_idx = ctx.upload(idx)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n_nodes = ctx.bvh_build_dev(d_v, len(v), d_idx, n_tri, d_n, 2 * n_tri)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if best is None or (r and dt < best):
            best, best_stats = dt, ctx.bvh_last_build_stats()
    g_nodes = d_n.cpu().numpy()[: n_nodes * 32].view(abi.BVH_NODE)
    g_idx = d_idx.cpu().numpy().view(np.uint32)[: 3 * n_tri]
    shape = tree_shape(g_nodes)
    # bytes of a faithful emulation (770 B / prim / level: 22 shuffles + 21 bounds) and of a non-emulating
    # binned builder (44 B / prim / level), + precompute 84 B / prim, nodes 32 B, final index permute 24 B / prim
    fixed = n_tri * 84.0 + n_nodes * 32.0 + n_tri * 24.0
    b770, b44 = fixed + shape["sum_active_prims"] * 770.0, fixed + shape["sum_active_prims"] * 44.0
    extra["bvh_build"] = {"metric": "SAH BVH build Mprims/s", "value": round(n_tri / best / 1e6, 1), "n_tris": n_tri,
                          "ms": round(best * 1e3, 2), "timing": "best of 3 builds, host wall clock around vd_bvh_build_dev (synchronous)",
                          "nodes": int(n_nodes), "phases_ms": best_stats,
                          "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", **shape,
                                       "emulating_770B_per_prim_level": {"algorithmic_bytes": int(b770), "achieved": round(b770 / best / 1e9, 1),
                                                                         "frac": round(b770 / best / 1e9 / HBM_PEAK_GBS, 4)},
                                       "binned_44B_per_prim_level": {"algorithmic_bytes": int(b44), "achieved": round(b44 / best / 1e9, 1),
                                                                     "frac": round(b44 / best / 1e9 / HBM_PEAK_GBS, 4),
                                                                     "floor_ms_at_peak": round(b44 / HBM_PEAK_GBS / 1e6, 3)},
                                       "note": "the builder moves ~21 B per active prim per shuffle round in phase A (DESIGN 3.3), not 770 B per level: the "
                                               "770 B figure prices the reference's 21 bounds passes, which one binning pass replaces"}}
    del d_v, d_n, d_idx

    d_ti = ctx.upload(tinst)
    d_t = ctx.empty((2 * n_tl + 1) * 32)
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.tlas_build_dev(d_ti, n_tl, d_m, n_mesh, d_t)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t
    g_tlas = d_t.cpu().numpy()[: (2 * n_tl + 1) * 32].view(abi.TLAS_NODE).copy()
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.tlas_refit_dev(d_ti, n_tl, d_m, n_mesh, d_t)
        torch.cuda.synchronize(); t_refit = time.perf_counter() - t
    t_pipe = pipelined(lambda: ctx.tlas_refit_dev(d_ti, n_tl, d_m, n_mesh, d_t))
    extra["tlas"] = {"n_instances": n_tl, "build_ms": round(t_build * 1e3, 1), "refit_ms": round(t_refit * 1e3, 3),
                     "refit_queued_ms": round(t_pipe * 1e3, 4)}
    # the dynamic-scene frame: animate 10 % of the instances (compute_update, inverse kept in step), refit
    # the TLAS of the first 32768, cull + compact all of them - queued back to back as a frame loop does
    if n >= n_tl:
        d_dyn = d_i.clone()
        d_mov = torch.arange(0, n, 10, dtype=torch.int32, device=dev)
        d_t2 = ctx.empty((2 * n_tl + 1) * 32)
        ctx.tlas_build_dev(d_dyn, n_tl, d_m, n_mesh, d_t2)

        def frame():
            ctx.compute_update_dev(d_mov, d_mov.numel(), d_dyn, n, 1.0, 0.016, True)
            ctx.tlas_refit_dev(d_dyn, n_tl, d_m, n_mesh, d_t2)
            ctx.cull_compact_dev(cam, d_m, n_mesh, d_dyn, n, d_out, d_cnt, False, first)
        t_frame = pipelined(frame, reps=30)
        extra["dynamic_frame"] = {"instances": n, "moving": int(d_mov.numel()), "tlas_instances": n_tl,
                                  "ms_per_frame": round(t_frame * 1e3, 4), "fps_equivalent": round(1.0 / t_frame, 1)}
        # The same frame on TWO streams: the refit (a latency-bound climb over 65 k nodes on a few CUs) does not depend on the cull
        # (HBM-bound, all CUs) - both only read what compute_update wrote - so a second context on its own stream runs it beside the
        # cull; events order update -> {refit | cull} -> next update.  Same bytes out (checked against the one-stream frame).
        from voidin_amd.runtime import Context
        ctx2 = Context(dev.index, use_torch_stream=False)
        s1, s2 = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
        ctx2.set_stream(s2.cuda_stream)
        ev_upd, ev_ref = torch.cuda.Event(), torch.cuda.Event()

        def frame2():
            ctx.compute_update_dev(d_mov, d_mov.numel(), d_dyn, n, 1.0, 0.016, True)
            ev_upd.record(s1)
            s2.wait_event(ev_upd)
            ctx2.tlas_refit_dev(d_dyn, n_tl, d_m, n_mesh, d_t2)
            ev_ref.record(s2)
            ctx.cull_compact_dev(cam, d_m, n_mesh, d_dyn, n, d_out, d_cnt, False, first)
            s1.wait_event(ev_ref)                                 # the next frame's update must not overtake the refit's reads
        # one frame of each form from the same state: same top level, same survivor count
        d_dyn.copy_(d_i); ctx.tlas_build_dev(d_dyn, n_tl, d_m, n_mesh, d_t2); torch.cuda.synchronize()
        frame(); torch.cuda.synchronize()
        a_t, a_c = d_t2.clone(), int(d_cnt[0].item())
        d_dyn.copy_(d_i); ctx.tlas_build_dev(d_dyn, n_tl, d_m, n_mesh, d_t2); torch.cuda.synchronize()
        frame2(); torch.cuda.synchronize()
        d_dyn2_ok = bool(torch.equal(a_t, d_t2) and a_c == int(d_cnt[0].item()))
        del a_t
        t_frame2 = pipelined(frame2, reps=30)
        extra["dynamic_frame"]["two_streams"] = {"ms_per_frame": round(t_frame2 * 1e3, 4), "fps_equivalent": round(1.0 / t_frame2, 1),
                                                 "same_tlas_and_count_as_one_stream": d_dyn2_ok,
                                                 "note": "refit on a second context / stream beside the cull (events: update -> {refit | cull} -> next update)"}
        ctx2.close()
        del d_dyn, d_mov, d_t2
        ctx.cull_compact_dev(cam, d_m, n_mesh, d_i, n, d_out, d_cnt, False, first)   # leave d_out / the id table as the later legs expect
    # BASELINE config 5 names a 64k-instance refit: beyond the reference's 16-bit child ids (tlas.rs:71), so in
    # the wide layout; timed with HIP events by the library (wall clock of a 0.1 ms call is mostly launch latency)
    n_w = 65536
    winst = synth.instances(n_w, seed=synth.SEED_BASE + 7, extent=400.0)
    d_wi = ctx.upload(winst)
    d_w = ctx.empty((2 * n_w + 1) * 48)
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.tlas_build_dev(d_wi, n_w, d_m, n_mesh, d_w, wide=True)
    torch.cuda.synchronize(); t_wbuild = time.perf_counter() - t
    ctx.set_timing(True)
    g = []
    for _ in range(5):
        ctx.tlas_refit_dev(d_wi, n_w, d_m, n_mesh, d_w, wide=True)
        g.append(ctx.last_gpu_ms())
    ctx.set_timing(False)
    t_wpipe = pipelined(lambda: ctx.tlas_refit_dev(d_wi, n_w, d_m, n_mesh, d_w, wide=True))
    # refit after motion against the oracle's refit on the GPU-built topology (O(N) on the CPU)
    wide_ok = None
    if verify:
        moved = ref.compute_update(np.arange(0, n_w, 10, dtype=np.uint32), winst, 1.0, 0.016)
        topo = d_w.cpu().numpy()[: (2 * n_w + 1) * 48].view(abi.TLAS_NODE_WIDE).copy()
        d_wm = ctx.upload(moved)
        ctx.tlas_refit_dev(d_wm, n_w, d_m, n_mesh, d_w, wide=True)
        torch.cuda.synchronize()
        got_w = d_w.cpu().numpy()[: (2 * n_w + 1) * 48].view(abi.TLAS_NODE_WIDE)
        wide_ok = bool(got_w.tobytes() == ref.tlas_refit(moved, meshes, topo).tobytes())
        del d_wm
    extra["tlas_wide_64k"] = {"n_instances": n_w, "build_ms": round(t_wbuild * 1e3, 1), "refit_gpu_ms": round(min(g), 4),
                              "refit_queued_ms": round(t_wpipe * 1e3, 4), "refit_after_motion_bit_exact_vs_oracle": wide_ok}
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
    scene_host = (tl, inst_t, infos, nodes_b, tv, idx_b)
    ds = ctx.device_scene(scene_host)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
    d_any = torch.zeros(len(rays), dtype=torch.int32, device=dev)
    # per-scene preparation (vd_trace_prepare_dev: de-indexed leaf triangles); the timed calls are the prepared ones, the
    # plain entry points are timed beside them and must give the same bytes
    acc = ctx.trace_prepare(ds)
    ctx.set_option("trace.fan", 3)                # the default's value, PINNED: left to itself the fan-out skips itself for 15 calls after a call
                                                  # that found no long rays (per-context state), and timings would depend on the calls before
    ctx.set_timing(True)
    t_cl, t_any, t_cl0, t_any0 = [], [], [], []
    for _ in range(3):
        ctx.trace_dev(ds, d_rays, len(rays), d_hits); t_cl0.append(ctx.last_gpu_ms())
        ctx.trace_any_dev(ds, d_rays, len(rays), d_any); t_any0.append(ctx.last_gpu_ms())
    plain_bytes = (d_hits.cpu().numpy().tobytes(), d_any.cpu().numpy().tobytes())
    t_cli, t_anyi = [], []                        # the indexed leaves (indices[] -> vertices[]): what a plain call walks when it cannot de-index
    ctx.set_option("trace.auto_prepare", 0)
    for _ in range(2):
        ctx.trace_dev(ds, d_rays, len(rays), d_hits); t_cli.append(ctx.last_gpu_ms())
        ctx.trace_any_dev(ds, d_rays, len(rays), d_any); t_anyi.append(ctx.last_gpu_ms())
    ctx.set_option("trace.auto_prepare", None)
    indexed_bytes = (d_hits.cpu().numpy().tobytes(), d_any.cpu().numpy().tobytes())
    for _ in range(3):
        ctx.trace_prepared_dev(acc, d_rays, len(rays), d_hits); t_cl.append(ctx.last_gpu_ms())
        ctx.trace_any_prepared_dev(acc, d_rays, len(rays), d_any); t_any.append(ctx.last_gpu_ms())
    ctx.set_timing(False)
    hits = d_hits.cpu().numpy()[: len(rays) * 16].view(abi.HIT)
    extra["trace"] = {"n_rays": len(rays), "scene": "2000 instances x 131k-triangle mesh",
                      "closest_hit_Mrays_per_s": round(len(rays) / min(t_cl) / 1e3, 1),
                      "occlusion_Mrays_per_s": round(len(rays) / min(t_any) / 1e3, 1),
                      "without_vd_trace_prepare": {"closest_hit_Mrays_per_s": round(len(rays) / min(t_cl0) / 1e3, 1),
                                                   "occlusion_Mrays_per_s": round(len(rays) / min(t_any0) / 1e3, 1),
                                                   "note": "the plain vd_trace_dev / vd_trace_any_dev: they de-index the leaves themselves, per call",
                                                   "same_bytes_as_prepared": bool(plain_bytes == (d_hits.cpu().numpy().tobytes(), d_any.cpu().numpy().tobytes()))},
                      "indexed_leaves": {"closest_hit_Mrays_per_s": round(len(rays) / min(t_cli) / 1e3, 1),
                                         "occlusion_Mrays_per_s": round(len(rays) / min(t_anyi) / 1e3, 1),
                                         "same_bytes_as_prepared": bool(indexed_bytes == (d_hits.cpu().numpy().tobytes(), d_any.cpu().numpy().tobytes()))},
                      "hit_fraction": round(float(hits["hit"].mean()), 3),
                      "occlusion_flags_equal_closest_hit": bool(np.array_equal(d_any.cpu().numpy().astype(np.uint32), hits["hit"]))}
    # OPT-IN, not the reference's visit order (VD_OPT_TRACE_TIGHT_TLAS): the same scene prepared with a private top level over
    # tight world boxes; acceptance = hit flags equal, distances within 1e-5 of the exact walk's (in practice the same bits)
    ctx.set_option("trace.fan", 1)                # under the tight top level no ray is long enough to fan out: the one-launch kernels are what the
                                                  # default converges to there (the fan-out skips itself); pinned for the same reason as above
    ctx.set_option("trace.tight_tlas", 1)
    t_prep = time.perf_counter()
    acc_t = ctx.trace_prepare(ds)
    t_prep = time.perf_counter() - t_prep
    ctx.set_option("trace.tight_tlas", None)
    d_hits_t = ctx.empty(len(rays) * 16)
    d_any_t = torch.zeros(len(rays), dtype=torch.int32, device=dev)
    ctx.set_timing(True)
    t_clt, t_anyt = [], []
    for _ in range(3):
        ctx.trace_prepared_dev(acc_t, d_rays, len(rays), d_hits_t); t_clt.append(ctx.last_gpu_ms())
        ctx.trace_any_prepared_dev(acc_t, d_rays, len(rays), d_any_t); t_anyt.append(ctx.last_gpu_ms())
    ctx.set_timing(False)
    ht = d_hits_t.cpu().numpy()[: len(rays) * 16].view(abi.HIT)
    hm = hits["hit"] == 1
    abserr = np.abs(ht["dist"][hm].astype(np.float64) - hits["dist"][hm])
    extra["trace"]["tight_tlas_option"] = {
        "closest_hit_Mrays_per_s": round(len(rays) / min(t_clt) / 1e3, 1), "occlusion_Mrays_per_s": round(len(rays) / min(t_anyt) / 1e3, 1),
        "prepare_ms": round(t_prep * 1e3, 2), "fallback_instances": acc_t.info()["tight_fallback_instances"],
        "hit_flags_equal_exact_walk": bool(np.array_equal(ht["hit"], hits["hit"]) and np.array_equal(d_any_t.cpu().numpy().astype(np.uint32), hits["hit"])),
        "max_abs_distance_error": float(abserr.max()) if abserr.size else 0.0,       # vs the exact walk; north_star allows 1e-5
        "distances_bit_equal": int((ht["dist"][hm].view(np.uint32) == hits["dist"][hm].view(np.uint32)).sum()), "hits": int(hm.sum()),
        "note": "default off; the exact (reference visit order) numbers above are the parity path"}
    acc_t.close()
    # the same boxes under an LBVH built on all CUs (VD_OPT_TRACE_TIGHT_TLAS = 2): the form for scenes that move - the top level is
    # rebuilt from the instance buffer per frame (vd_trace_accel_update_dev)
    ctx.set_option("trace.tight_tlas", 2)
    acc_l = ctx.trace_prepare(ds)
    ctx.set_option("trace.tight_tlas", None)
    t_upd = []
    for _ in range(5):
        t = time.perf_counter(); acc_l.update(); t_upd.append(time.perf_counter() - t)
    ctx.set_timing(True)
    t_cll, t_anyl = [], []
    for _ in range(3):
        ctx.trace_prepared_dev(acc_l, d_rays, len(rays), d_hits_t); t_cll.append(ctx.last_gpu_ms())
        ctx.trace_any_prepared_dev(acc_l, d_rays, len(rays), d_any_t); t_anyl.append(ctx.last_gpu_ms())
    ctx.set_timing(False)
    hl = d_hits_t.cpu().numpy()[: len(rays) * 16].view(abi.HIT)
    extra["trace"]["tight_tlas_option"]["lbvh"] = {
        "closest_hit_Mrays_per_s": round(len(rays) / min(t_cll) / 1e3, 1), "occlusion_Mrays_per_s": round(len(rays) / min(t_anyl) / 1e3, 1),
        "rebuild_ms_blocking": round(min(t_upd) * 1e3, 3),
        "hit_flags_equal_exact_walk": bool(np.array_equal(hl["hit"], hits["hit"]) and np.array_equal(d_any_t.cpu().numpy().astype(np.uint32), hits["hit"])),
        "max_abs_distance_error": float(np.abs(hl["dist"][hm].astype(np.float64) - hits["dist"][hm]).max()) if hm.any() else 0.0,
        "distances_bit_equal": int((hl["dist"][hm].view(np.uint32) == hits["dist"][hm].view(np.uint32)).sum())}
    acc_l.close()
    ctx.set_option("trace.fan", 3)
    del d_hits_t, d_any_t
    if not args.no_cpu_baseline:
        # vd_ref_trace on a bounded sample of the same rays: every 4th ray on all threads, every 64th on one thread
        threads = os.cpu_count() or 1
        sN, s1 = np.ascontiguousarray(rays[::4]), np.ascontiguousarray(rays[::64])
        t = time.perf_counter(); h1, _ = ref.trace(scene_host, s1, threads=1); t1 = time.perf_counter() - t
        t = time.perf_counter(); hN, _ = ref.trace(scene_host, sN, threads=threads); tN = time.perf_counter() - t
        sub = hits[::4]
        hit = hN["hit"] == 1
        ok = bool(np.array_equal(hN["hit"], sub["hit"]) and hN["dist"].tobytes() == sub["dist"].tobytes())
        max_abs = float(np.abs(hN["dist"][hit].astype(np.float64) - sub["dist"][hit]).max()) if hit.any() else 0.0
        extra["trace"]["cpu_baseline"] = {"value": round(len(sN) / tN / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                                          "one_thread_Mrays_per_s": round(len(s1) / t1 / 1e6, 4),
                                          "sample": f"every 4th ray of the same batch on {threads} threads ({len(sN)} rays), every 64th on 1 thread "
                                                    f"({len(s1)} rays), oracle vd_ref_trace",
                                          "gpu_equals_oracle_on_sample_hit_flags_and_distance_bits": ok,
                                          "max_abs_distance_error": max_abs}       # exact walk vs oracle: 0.0 (bit-equal); north_star allows 1e-5
    acc.close()
    del ds, d_rays, d_hits, d_any
    # the reference's own harness shape (src/bin/bvh_gpu.rs:107-131): one large mesh + four small ones, 4 M primary rays
    inst2, infos2, B2, V2, I2 = synth.harness_scene(ctx.bvh_build)
    tl2 = ctx.tlas_build(inst2, infos2)
    rays2 = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 15), pitch_deg=0), 2048, 2048)
    ds2 = ctx.device_scene((tl2, inst2, infos2, B2, V2, I2))
    d_r2, d_h2 = ctx.upload(rays2), ctx.empty(len(rays2) * 16)
    acc2 = ctx.trace_prepare(ds2)
    ctx.set_timing(True)
    t2 = []
    for _ in range(3):
        ctx.trace_prepared_dev(acc2, d_r2, len(rays2), d_h2); t2.append(ctx.last_gpu_ms())
    ctx.set_timing(False)
    h2 = d_h2.cpu().numpy()[: len(rays2) * 16].view(abi.HIT)
    extra["trace_harness_scene"] = {"n_rays": len(rays2), "scene": f"bvh_gpu.rs shape: {len(I2)//3} triangles, 5 instances",
                                    "closest_hit_Mrays_per_s": round(len(rays2) / min(t2) / 1e3, 1),
                                    "hit_fraction": round(float(h2["hit"].mean()), 3)}
    acc2.close()
    ctx.set_option("trace.fan", None)
    del ds2, d_r2, d_h2
    # the one real mesh the reference checkout carries, as its default demo places and views it (src/bin/model.rs:100-106, :235):
    # tests/golden/helmet.npz = the arrays GltfDocument::import hands to MeshPool::add + the oracle's tree checksums and hits
    hp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "helmet.npz")
    if os.path.exists(hp):
        import zlib
        g = np.load(hp)
        hv, hi = np.ascontiguousarray(g["vertices"]), np.ascontiguousarray(g["indices"])
        tb = []
        for _ in range(3):
            t = time.perf_counter(); hn, hidx = ctx.bvh_build(hv, hi); tb.append(time.perf_counter() - t)
        htl = ctx.tlas_build(g["instances"], g["meshes"])
        hrays = synth.primary_rays(g["camera"], 1024, 1024)
        hds = ctx.device_scene((htl, g["instances"], g["meshes"], hn, hv, hidx))
        d_hr, d_hh = ctx.upload(hrays), ctx.empty(len(hrays) * 16)
        ctx.set_timing(True)
        th = []
        for _ in range(3):
            ctx.trace_dev(hds, d_hr, len(hrays), d_hh); th.append(ctx.last_gpu_ms())
        ctx.set_timing(False)
        small = synth.primary_rays(g["camera"], int(g["width"]), int(g["height"]))
        d_sr, d_sh = ctx.upload(small), ctx.empty(len(small) * 16)
        ctx.trace_dev(hds, d_sr, len(small), d_sh)
        sh = d_sh.cpu().numpy()[: len(small) * 16].view(abi.HIT)
        hit = g["hit"] == 1
        extra["reference_helmet"] = {"mesh": "DamagedHelmet.glb of the reference's assets: 14556 vertices, 15452 triangles",
                                     "blas_build_ms_host_arrays": round(min(tb) * 1e3, 3),
                                     "blas_equals_oracle_checksums": bool(zlib.crc32(hn.tobytes()) == int(g["nodes_crc"]) and zlib.crc32(hidx.tobytes()) == int(g["indices_out_crc"])),
                                     "demo_view_1024x1024_Mrays_per_s": round(len(hrays) / min(th) / 1e3, 1),
                                     "demo_view_hits_equal_oracle_fixture": bool(np.array_equal(sh["hit"], g["hit"]) and sh["dist"][hit].tobytes() == g["dist"][hit].tobytes())}
        del hds, d_hr, d_hh, d_sr, d_sh
        # K meshes in ONE build (vd_bvh_build_batch_dev; MeshPool::add for a whole scene, mesh/mod.rs:309-351): 64 copies of the
        # helmet, and a Sponza-class load of 400 meshes of 1 k - 50 k triangles; device arrays, packed node buffer; every mesh's
        # nodes are compared (CRC) with the single build's
        def batch_leg(meshes, reps=3):
            K = len(meshes)
            items = (abi.BvhBatchItem * K)()
            dev_v = [ctx.upload(np.ascontiguousarray(v, dtype=np.float32)) for v, _ in meshes]
            host_i = [np.ascontiguousarray(i, dtype=np.uint32).reshape(-1) for _, i in meshes]
            dev_i = [ctx.upload(i) for i in host_i]
            n_tri = sum(len(i) // 3 for i in host_i)
            d_nodes = ctx.empty(2 * n_tri * 32 + 64 * K)
            best = None
            for _ in range(reps):
                for m in range(K):
                    dev_i[m].copy_(torch.from_numpy(host_i[m].view(np.uint8)))       # the build permutes the indices in place
                    items[m].verts_xyz, items[m].indices_inout, items[m].out_nodes = abi.ptr(dev_v[m]), abi.ptr(dev_i[m]), None
                    items[m].n_vert, items[m].n_tri, items[m].node_cap = len(meshes[m][0]), len(host_i[m]) // 3, 0
                torch.cuda.synchronize()
                t = time.perf_counter()
                ctx.bvh_build_batch_dev(items, K, d_nodes, 2 * n_tri + 2 * K, 0)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            nodes = d_nodes.cpu().numpy()
            crcs = [zlib.crc32(nodes[items[m].out_first_node * 32: (items[m].out_first_node + items[m].out_n_nodes) * 32].tobytes()) for m in range(K)]
            return best, n_tri, crcs, ctx.bvh_last_build_stats()
        t64, n64, crc64, st64 = batch_leg([(hv, hi)] * 64)
        rng = np.random.default_rng(5)
        many = []
        for k in range(400):
            t_ = int(np.exp(rng.uniform(np.log(1000), np.log(50_000))))
            u_ = max(8, int(np.sqrt(2 * t_)))
            many.append(synth.knot_mesh(u_, max(4, t_ // (2 * u_)), seed=synth.SEED_BASE + 100 + k))
        t400, n400, crc400, st400 = batch_leg(many, reps=2)
        t_single = time.perf_counter()
        ok400 = True
        for k in range(0, 400, 8):                                                   # every 8th mesh against its single build
            sn_, _ = ctx.bvh_build(*many[k])
            ok400 = ok400 and zlib.crc32(sn_.tobytes()) == crc400[k]
        t_single = (time.perf_counter() - t_single) / 50
        extra["bvh_build_batch"] = {
            "helmet_x64": {"ms_total": round(t64 * 1e3, 3), "ms_per_mesh_amortised": round(t64 * 1e3 / 64, 4), "Mprims_per_s": round(n64 / t64 / 1e6, 1),
                           "levels_phase_a": st64["levels_phase_a"], "kernel_launches": st64["kernel_launches"],
                           "every_mesh_equals_the_single_build": bool(all(c == int(g["nodes_crc"]) for c in crc64))},
            "meshes_400_of_1k_to_50k": {"triangles": n400, "ms_total": round(t400 * 1e3, 2), "Mprims_per_s": round(n400 / t400 / 1e6, 1),
                                        "levels_phase_a": st400["levels_phase_a"], "kernel_launches": st400["kernel_launches"],
                                        "single_build_ms_per_mesh_host_arrays": round(t_single * 1e3, 3),
                                        "sampled_meshes_equal_their_single_builds": bool(ok400)},
            "note": "device arrays, packed node buffer, wall clock around the blocking call; single helmet build for comparison: reference_helmet.blas_build_ms_host_arrays"}
    # the CPU harness (src/bin/bvh_cpu.rs:39-96): per-pixel rays + Bvh::traverse_iter against ONE mesh, on the device;
    # the harness's own 64-triangle soup at 640 x 640, and the large mesh of the scene above at 2048 x 2048
    cam_h = synth.camera_uniform(eye=(0, 0, 15), pitch_deg=0)
    sv_, si = synth.triangle_soup(64)
    sn, si = ctx.bvh_build(sv_, si)
    big_v, big_i = synth.knot_mesh(1024, 256)
    big_v = np.ascontiguousarray(big_v * np.float32(3.0))
    bn, bi = ctx.bvh_build(big_v, big_i)
    res = {}
    for tag, (nn, vv, ii, w) in {"soup64_640x640": (sn, sv_, si, 640), "knot_524k_2048x2048": (bn, big_v, bi, 2048)}.items():
        d_pr = ctx.empty(w * w * 32)
        d_td = torch.zeros(w * w, dtype=torch.float32, device=dev)
        d_nn, d_vv, d_ix = ctx.upload(nn), ctx.upload(np.ascontiguousarray(vv, dtype=np.float32)), ctx.upload(ii)
        ctx.primary_rays_dev(cam_h, w, w, d_pr)
        ctx.set_timing(True)
        tt = []
        for _ in range(3):
            ctx.traverse_iter_dev(d_nn, len(nn), d_vv, d_ix, d_pr, w * w, d_td); tt.append(ctx.last_gpu_ms())
        ctx.set_timing(False)
        res[tag] = {"Mrays_per_s": round(w * w / min(tt) / 1e3, 1), "hit_fraction": round(float((d_td >= 0).float().mean()), 3)}
        # the recursive Bvh::traverse (blas.rs:211-245) on the same rays; where traverse_iter hits, the same distance
        d_tr = torch.zeros(w * w, dtype=torch.float32, device=dev)
        ctx.set_timing(True)
        tr = []
        for _ in range(3):
            ctx.traverse_dev(d_nn, len(nn), d_vv, d_ix, d_pr, w * w, d_tr); tr.append(ctx.last_gpu_ms())
        ctx.set_timing(False)
        hit_it = d_td >= 0
        res[tag]["recursive_traverse_Mrays_per_s"] = round(w * w / min(tr) / 1e3, 1)
        res[tag]["recursive_equals_iter_where_it_hits"] = bool(torch.equal(d_tr[hit_it], d_td[hit_it]))
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
    ctx.cull_mask_dev(cam, d_m, n_mesh, d_i, n, d_mask)
    ctx.set_timing(True)
    t_p, t_o = [], []
    for _ in range(5):
        ctx.hiz_build_dev(d_depth, W, H, d_pyr); t_p.append(ctx.last_gpu_ms())
        ctx.occlusion_mask_dev(cam, d_m, n_mesh, d_i, n, d_pyr, W, H, d_mask, d_mask2); t_o.append(ctx.last_gpu_ms())
    ctx.set_timing(False)
    pop = lambda t: int(sum(bin(int(x) & 0xFFFFFFFFFFFFFFFF).count("1") for x in t.cpu().numpy()[:4096]))
    extra["occlusion_extension"] = {"note": "extension, no reference counterpart", "depth": f"{W}x{H}",
                                    "pyramid_build_ms": round(min(t_p), 4), "occlusion_mask_ms": round(min(t_o), 4),
                                    "instances": n, "GBps_instances_read": round(n * 144 / min(t_o) / 1e6, 1),
                                    "kept_of_first_262144_frustum_visible": [pop(d_mask2), pop(d_mask)]}
    del d_depth, d_pyr, d_mask, d_mask2

    # join the CPU oracles: parity of the TIMED inputs + the CPU baselines of SURVEY 8d (i), (iii)
    if fut_bvh is not None:
        (wn, wi), t_cpu = fut_bvh.result()
        extra["bvh_build"]["bit_exact_vs_oracle"] = bool(len(g_nodes) == len(wn) and g_nodes.tobytes() == wn.tobytes() and np.array_equal(g_idx, wi))
        extra["bvh_build"]["cpu_baseline"] = {"value": round(n_tri / t_cpu / 1e6, 3), "unit": "Mprims/s", "cores": 1, "kind": "port",
                                              "sample": f"the timed {n_tri}-tri knot mesh, oracle vd_ref_bvh_build ({t_cpu:.1f} s; ran beside the GPU legs)"}
        del wn, wi
    if fut_tlas is not None:
        want_t, t_cpu_t = fut_tlas.result()
        t = time.perf_counter(); ref.tlas_build(tinst[:1000], meshes); t_1k = time.perf_counter() - t
        extra["tlas"]["bit_exact_vs_oracle"] = bool(g_tlas.tobytes() == want_t.tobytes())
        extra["tlas"]["cpu_baseline"] = {"value": round(t_cpu_t * 1e3, 1), "unit": "ms per build", "cores": 1, "kind": "port",
                                         "sample": f"the timed {n_tl}-instance scene, oracle vd_ref_tlas_build ({t_cpu_t:.1f} s; ran beside the GPU legs); "
                                                   f"first 1000 instances: {t_1k * 1e3:.1f} ms", "ms_1000_instances": round(t_1k * 1e3, 2)}
    pool.shutdown()
    return extra


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.timeout))
    if args.launch_check:
        launch_check()
        return
    run_rank(args)


if __name__ == "__main__":
    main()
