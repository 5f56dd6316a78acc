"""Parity of the HIP TLAS build / refit and traversal against the CPU oracle, through the C ABI.
Node layouts are bit-exact.  Hit distances: the default walk visits the nodes in the reference's order on the oracle's
evaluation model, so its distances are asserted BIT-EQUAL to the oracle's (north_star allows 1e-5; the order-changing
options are checked against that absolute tolerance in tests/test_gpu_trace_tight.py)."""
import numpy as np
import pytest

from conftest import fields_equal, golden
from voidin_amd import abi, synth
from voidin_amd.runtime import VoidinError

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 5, 40, 300, "nan_60", "model_scene"])     # model_scene: the reference's own demo scene (make_golden.model_scene_cases)
def test_tlas_build_matches_golden(ctx, n):
    g = golden(f"tlas_{n}.npz")       # nan_60: NaN / inf - inf transforms; f32::min/max ignore the NaN corners (tlas.rs:39-44)
    nodes = ctx.tlas_build(g["instances"], g["meshes"])
    assert fields_equal(nodes, g["nodes"])
    # T3: refit(build(x), x) == build(x), bit for bit
    assert ctx.tlas_refit(g["instances"], g["meshes"], nodes).tobytes() == nodes.tobytes()


@pytest.mark.parametrize("n", [1000, 4097, 5600, 6003, 6900])
def test_tlas_build_seeded_vs_oracle(ctx, oracle, n):
    """Sizes on every default path of the build: the single-workgroup chain over LDS-resident slot arrays (up to 5600
    instances), the same chain reading them from memory (up to VD_OPT_TLAS_INDEX_MIN = 6800), the indexed build above."""
    meshes = synth.mesh_infos()
    inst = synth.instances(n, seed=synth.SEED_BASE + 6, extent=300.0)
    want = oracle.tlas_build(inst, meshes)
    got = ctx.tlas_build(inst, meshes)
    assert fields_equal(got, want)
    wide = ctx.tlas_build(inst, meshes, wide=True)
    assert np.array_equal(wide["left"] + (wide["right"] << 16), want["left_right"])
    assert np.array_equal(wide["min"], want["min"]) and np.array_equal(wide["instance_idx"], want["instance_idx"])


def test_tlas_build_on_several_workgroups_vs_oracle(ctx, oracle):
    """From 12288 instances on the chain runs on 16 workgroups that exchange their scan results through memory
    (tlas.hip, "build, several workgroups"): same chain, same nodes - NaN-free fast arithmetic and, with a poisoned
    instance, the total-order arithmetic; and the same bytes every time."""
    meshes = synth.mesh_infos()
    n = 16384 + 77
    inst = synth.instances(n, seed=synth.SEED_BASE + 14, extent=700.0)
    want = oracle.tlas_build(inst, meshes)
    got = ctx.tlas_build(inst, meshes)
    assert fields_equal(got, want)
    wide = ctx.tlas_build(inst, meshes, wide=True)
    assert np.array_equal(wide["left"] + (wide["right"] << 16), want["left_right"]) and np.array_equal(wide["max"], want["max"])
    bad = inst.copy()
    bad["transform"][4321][13] = np.float32("nan")    # NaN corners drop out of the leaf box (f32::min/max, tlas.rs:43)
    bad["transform"][77][12] = np.float32("inf")      # an infinite leaf box: the plain (not indexed) chain runs
    got_b, want_b = ctx.tlas_build(bad, meshes), oracle.tlas_build(bad, meshes)
    assert fields_equal(got_b, want_b)
    # workgroups that do not hear from each other in time give up and the build is redone on one workgroup
    ctx.set_option("tlas.spin_limit", 0)
    try:
        assert fields_equal(ctx.tlas_build(inst, meshes), want)
    finally:
        ctx.set_option("tlas.spin_limit", None)
    big = synth.instances(32768, seed=synth.SEED_BASE + 6, extent=300.0)
    first = ctx.tlas_build(big, meshes).tobytes()
    for _ in range(3):
        assert ctx.tlas_build(big, meshes).tobytes() == first


def test_cpu_harness_rays_and_traverse_iter(ctx, oracle):
    """The reference's CPU harness on the device: per-pixel rays (bvh_cpu.rs:71-83) bit-exact,
    Bvh::traverse_iter (blas.rs:247-295) bit-exact - same visit order, same arithmetic."""
    import torch
    cam = synth.camera_uniform(eye=(0, 0, 15), pitch_deg=0)                     # bvh_cpu.rs:134
    for w, h in ((64, 64), (640, 640), (1, 1)):
        want = oracle.primary_rays(cam, w, h)
        d_rays = torch.zeros(w * h * abi.RAY.itemsize, dtype=torch.uint8, device="cuda")
        ctx.primary_rays_dev(cam, w, h, d_rays)
        ctx.synchronize()
        assert d_rays.cpu().numpy().tobytes() == want.tobytes()
    ctx.primary_rays_dev(cam, 0, 0, None)                                       # nothing to do
    assert ctx.lib.vd_primary_rays_dev(ctx.h, None, 4, 4, None) == abi.VD_ERR_INVALID_ARG
    g = golden("harness_soup64.npz")                                            # fixture from the numpy restatement
    d_gr = torch.zeros(len(g["rays"]) * abi.RAY.itemsize, dtype=torch.uint8, device="cuda")
    ctx.primary_rays_dev(g["camera"], int(g["width"]), int(g["height"]), d_gr)
    assert d_gr.cpu().numpy().tobytes() == g["rays"].tobytes()
    d_gd = torch.zeros(len(g["rays"]), dtype=torch.float32, device="cuda")
    ctx.traverse_iter_dev(ctx.upload(g["nodes"]), len(g["nodes"]), ctx.upload(g["vertices"].astype(np.float32)),
                          ctx.upload(g["indices"].astype(np.uint32)), d_gr, len(g["rays"]), d_gd)
    assert d_gd.cpu().numpy().tobytes() == g["dist"].tobytes()
    rays = oracle.primary_rays(cam, 160, 160)
    for name in ("blas_soup64.npz", "blas_sphere_1_10.npz", "blas_knot_2k.npz", "blas_plane.npz"):
        g = golden(name)
        want = oracle.traverse_iter(g["nodes"], g["vertices"], g["indices_out"], rays)
        d_out = torch.full((len(rays),), 7.0, dtype=torch.float32, device="cuda")
        ctx.traverse_iter_dev(ctx.upload(g["nodes"]), len(g["nodes"]), ctx.upload(g["vertices"].astype(np.float32)),
                              ctx.upload(g["indices_out"].astype(np.uint32)), ctx.upload(rays), len(rays), d_out)
        got = d_out.cpu().numpy()
        assert got.tobytes() == want.tobytes(), name
        if name != "blas_plane.npz":
            assert (got >= 0).any() and (got < 0).any()
    # a freshly built big mesh, rays from all around it (two-sided test: back faces count)
    v, i = synth.knot_mesh(256, 64)
    nodes, idx = ctx.bvh_build(v, i)
    n = 20000
    u = synth.uniform01(synth.SEED_BASE + 31, 0, 6 * n).reshape(n, 6).astype(np.float64)
    o = u[:, :3] * 2 - 1
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * 6.0
    t = (u[:, 3:] * 2 - 1) * 1.5
    rs = np.zeros(n, dtype=abi.RAY)
    rs["eye"], rs["dir"] = o, (t - o) / np.linalg.norm(t - o, axis=1, keepdims=True)
    want = oracle.traverse_iter(nodes, v, idx, rs)
    d_out = torch.zeros(len(rs), dtype=torch.float32, device="cuda")
    ctx.traverse_iter_dev(ctx.upload(nodes), len(nodes), ctx.upload(v.astype(np.float32)), ctx.upload(idx), ctx.upload(rs), len(rs), d_out)
    assert d_out.cpu().numpy().tobytes() == want.tobytes()
    assert ctx.lib.vd_traverse_iter_dev(ctx.h, None, 0, None, None, None, 4, None) == abi.VD_ERR_INVALID_ARG
    # axis-aligned rays (zero direction components: the slab test divides by them - inf and NaN must flow through min / max
    # exactly as in the restated loop), rays that start inside the mesh, and degenerate rays
    g = golden("blas_sphere_1_10.npz")
    k = 3000
    u = synth.uniform01(synth.SEED_BASE + 32, 0, 3 * k).reshape(k, 3).astype(np.float64)
    ar = np.zeros(6 * k + 3, dtype=abi.RAY)
    for a in range(3):
        for sgn in (0, 1):
            blk = ar[(2 * a + sgn) * k: (2 * a + sgn + 1) * k]
            e = (u * 2 - 1) * 1.2
            e[:, a] = -3.0 if sgn == 0 else 0.0          # outside along the axis / inside the sphere
            blk["eye"] = e
            blk["dir"][:, a] = 1.0 if sgn == 0 else -1.0
    ar["dir"][-3] = (0, 0, 0)                            # no direction at all
    ar["dir"][-2] = (np.nan, 1, 0)
    ar["eye"][-1] = (np.inf, 0, 0); ar["dir"][-1] = (-1, 0, 0)
    want = oracle.traverse_iter(g["nodes"], g["vertices"], g["indices_out"], ar)
    d_out = torch.zeros(len(ar), dtype=torch.float32, device="cuda")
    ctx.traverse_iter_dev(ctx.upload(g["nodes"]), len(g["nodes"]), ctx.upload(g["vertices"].astype(np.float32)),
                          ctx.upload(g["indices_out"].astype(np.uint32)), ctx.upload(ar), len(ar), d_out)
    got = d_out.cpu().numpy()
    assert got.view(np.uint32).tobytes() == want.view(np.uint32).tobytes()
    assert (got[: 6 * k] >= 0).sum() > k


def test_recursive_traverse_vs_oracle(ctx, oracle):
    """`Bvh::traverse` (crates/bvh/src/blas.rs:211-245), the reference's recursive walk - left then right,
    no near / far ordering, Hit(t0) when the root box is entered and nothing is hit, Miss only when the root box is
    missed - on the device with an explicit stack, bit for bit against the oracle's literal recursion: the harness soup,
    a knot mesh (deep tree), axis-aligned / degenerate rays, a finite t0, and the host-pointer form."""
    import ctypes as C
    import torch
    g = golden("harness_soup64.npz")
    cases = [(g["nodes"], g["vertices"].astype(np.float32), g["indices"].astype(np.uint32), g["rays"])]
    v, i = synth.knot_mesh(96, 24)
    nodes, idx = oracle.bvh_build(v, i)
    cases.append((nodes, v, idx, synth.primary_rays(synth.camera_uniform(eye=(0, 0, 6), pitch_deg=0), 64, 64)))
    for nodes, v, idx, rays in cases:
        odd = rays[:64].copy()
        odd["dir"][:16] = (0, 0, -1); odd["dir"][16:32] = (1, 0, 0); odd["dir"][32] = (0, 0, 0); odd["dir"][33] = (np.nan, 1, 0)
        odd["eye"][34] = (np.inf, 0, 0)
        rays = np.concatenate([rays, odd])
        d_n, d_v, d_i, d_r = ctx.upload(nodes), ctx.upload(v), ctx.upload(idx), ctx.upload(rays)
        for t0 in (1e30, 14.0):
            want = oracle.traverse_recursive(nodes, v, idx, rays, t0=t0)
            d_out = torch.zeros(len(rays), dtype=torch.float32, device="cuda")
            ctx.traverse_dev(d_n, len(nodes), d_v, d_i, d_r, len(rays), d_out, t0=t0)
            got = d_out.cpu().numpy()
            assert got.view(np.uint32).tobytes() == want.view(np.uint32).tobytes()
        hits = (want >= 0) & (want < np.float32(14.0))
        assert hits.sum() > 20 and (want == np.float32(14.0)).sum() > 0 and (want < 0).sum() > 0   # real hits, Hit(t0) quirk, Miss
        # where traverse_iter hits, the recursive walk returns the same distance
        it = oracle.traverse_iter(nodes, v, idx, rays)
        full = oracle.traverse_recursive(nodes, v, idx, rays)
        assert np.array_equal(full[it >= 0], it[it >= 0])
    # host-pointer form + argument checks
    nodes, v, idx, rays = cases[0]
    out = np.zeros(len(rays), dtype=np.float32)
    rc = ctx.lib.vd_traverse(ctx.h, nodes.ctypes.data, len(nodes), v.ctypes.data, len(v), idx.ctypes.data, len(idx) // 3,
                             np.ascontiguousarray(rays).ctypes.data, len(rays), C.c_float(1e30), out.ctypes.data)
    assert rc == 0 and out.tobytes() == oracle.traverse_recursive(nodes, v, idx, rays).tobytes()
    assert ctx.lib.vd_traverse_dev(ctx.h, None, 0, None, None, None, 4, C.c_float(1e30), None) == abi.VD_ERR_INVALID_ARG


def test_tlas_build_with_nan_and_inf_boxes(ctx, oracle):
    """A NaN transform entry makes NaN corners, and Rust's f32::min/max (tlas.rs:43) ignore them: the leaf box stays the
    object-space seed (tlas.rs:39) joined with the finite corners - never a NaN.  inf entries make infinite boxes
    (kept), and inf columns of opposite sign make NaNs the arithmetic GENERATES (inf - inf: sign bit set on x86, clear
    on gfx950 - nothing may depend on it).  Leaves, chain and refit must reproduce the oracle bit for bit."""
    meshes = synth.mesh_infos()
    for case in range(3):
        inst = synth.instances(700, seed=synth.SEED_BASE + 12, extent=100.0)
        if case == 0:
            inst["transform"][13, 12] = np.nan            # translation x of instance 13
            inst["transform"][400, 5] = np.nan
        elif case == 1:
            inst["transform"][13, 12] = np.inf
            inst["transform"][400, 5] = -np.inf
        else:
            inst["transform"][13, 0] = np.inf; inst["transform"][13, 4] = -np.inf      # X.x = inf, Y.x = -inf
            inst["transform"][400, 2] = -np.inf; inst["transform"][400, 10] = np.inf
            inst["transform"][555, 12] = np.nan
        got = ctx.tlas_build(inst, meshes)
        want = oracle.tlas_build(inst, meshes)
        assert fields_equal(got, want)
        assert not np.isnan(got["min"]).any() and not np.isnan(got["max"]).any()
        if case != 1:
            m = meshes[inst["mesh"][13]] if case == 0 else None
            if m is not None:                             # every corner's x is NaN: the x range is the mesh's own
                assert got["min"][14][0] == m["min"][0] and got["max"][14][0] == m["max"][0]
        assert ctx.tlas_refit(inst, meshes, got).tobytes() == got.tobytes()


def test_tlas_refit_after_motion(ctx, oracle):
    """compute_update-style motion (rotz(theta) * transform on 10 % of the instances,
    shaders/compute_update.wgsl:12-28), then refit == oracle refit == same-topology recompute."""
    meshes = synth.mesh_infos()
    n = 2000
    inst = synth.instances(n, seed=synth.SEED_BASE + 7, extent=200.0)
    nodes = ctx.tlas_build(inst, meshes)
    th = np.float32(0.3)
    c, s = np.cos(th), np.sin(th)
    rot = np.array([[c, s, 0, 0], [-s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)  # column-major [col][row]
    moved = inst.copy()
    for i in range(0, n, 10):
        T = moved["transform"][i].reshape(4, 4)
        moved["transform"][i] = (T @ rot.T.T).reshape(16) if False else np.einsum("kr,ck->cr", rot, T).reshape(16)
    got = ctx.tlas_refit(moved, meshes, nodes)
    want = oracle.tlas_refit(moved, meshes, nodes)
    assert got.tobytes() == want.tobytes()
    assert np.array_equal(got["left_right"], nodes["left_right"])          # topology untouched
    assert not np.array_equal(got["min"], nodes["min"])                    # boxes did move
    # every interior box is the union of its children
    for k in range(n + 1, 2 * n + 1):
        l, r = int(got["left_right"][k] & 0xFFFF), int(got["left_right"][k] >> 16)
        assert np.array_equal(got["min"][k], np.minimum(got["min"][l], got["min"][r]))
        assert np.array_equal(got["max"][k], np.maximum(got["max"][l], got["max"][r]))


def test_tlas_refit_arena_survives_a_change_of_instance_count(oracle):
    """The refit arena (epoch-tagged links, then the arrival words) is laid out by n.  Refits at n1 followed by refits
    at a slightly larger n2 on the SAME context (capacity unchanged) must not read n1's arrival words as n2's link
    records: six refits at each size, every one compared with the oracle, both orders, on a context of its own."""
    from voidin_amd.runtime import Context
    meshes = synth.mesh_infos()
    own = Context(0)
    try:
        for sizes in ((1000, 1150, 1000, 1249), (4096, 4097, 3500)):
            for n in sizes:
                inst = synth.instances(n, seed=synth.SEED_BASE + 70 + n % 7, extent=150.0)
                nodes = own.tlas_build(inst, meshes)
                for k in range(6):
                    moved = inst.copy()
                    moved["transform"][:: 3 + k, 12:15] += np.float32(0.37 * (k + 1))
                    got = own.tlas_refit(moved, meshes, nodes)
                    want = oracle.tlas_refit(moved, meshes, nodes)
                    assert got.tobytes() == want.tobytes(), (sizes, n, k)
    finally:
        own.close()


def test_tlas_overflow_and_bad_args(ctx):
    meshes = synth.mesh_infos()
    inst = np.zeros(abi.TLAS_MAX_INSTANCES + 1, abi.INSTANCE)
    with pytest.raises(VoidinError) as e:
        ctx.tlas_build(inst, meshes)
    assert e.value.code == abi.VD_ERR_TLAS_OVERFLOW
    assert ctx.lib.vd_tlas_build(ctx.h, None, 0, None, 0, None) == abi.VD_ERR_INVALID_ARG
    # the other entry points of this file: incomplete scene, null buffers, bad external-buffer arguments
    import ctypes as C
    empty = abi.TraceScene()
    assert ctx.lib.vd_trace_dev(ctx.h, C.byref(empty), None, 8, None) == abi.VD_ERR_INVALID_ARG
    assert ctx.lib.vd_trace_any_dev(ctx.h, C.byref(empty), None, 8, None) == abi.VD_ERR_INVALID_ARG
    assert b"incomplete scene" in ctx.lib.vd_last_error(ctx.h)
    assert ctx.lib.vd_shadow_rays_dev(ctx.h, None, None, 8, None, None) == abi.VD_ERR_INVALID_ARG
    assert ctx.lib.vd_shadow_rays_dev(ctx.h, None, None, 0, None, None) == abi.VD_OK          # nothing to do
    hnd, ptr = C.c_void_p(), C.c_void_p()
    assert ctx.lib.vd_import_external_buffer(ctx.h, -1, 4096, C.byref(hnd), C.byref(ptr)) == abi.VD_ERR_INVALID_ARG
    assert ctx.lib.vd_import_external_buffer(ctx.h, 0, 0, C.byref(hnd), C.byref(ptr)) == abi.VD_ERR_INVALID_ARG
    assert ctx.lib.vd_release_external_buffer(ctx.h, None) == abi.VD_ERR_INVALID_ARG
    # external semaphores: argument validation, and a descriptor that is not a semaphore is an error code,
    # not a crash.  (A functional wait / signal round trip needs a Vulkan device on the box: not claimed, not faked.)
    sem = C.c_void_p()
    assert ctx.lib.vd_import_external_semaphore(ctx.h, -1, 0, C.byref(sem)) == abi.VD_ERR_INVALID_ARG
    assert ctx.lib.vd_import_external_semaphore(ctx.h, 0, 1, None) == abi.VD_ERR_INVALID_ARG
    assert ctx.lib.vd_wait_external_semaphore_async(ctx.h, None, 1) == abi.VD_ERR_INVALID_ARG
    assert ctx.lib.vd_signal_external_semaphore_async(ctx.h, None, 1) == abi.VD_ERR_INVALID_ARG
    assert ctx.lib.vd_release_external_semaphore(ctx.h, None) == abi.VD_ERR_INVALID_ARG
    import os
    for timeline in (0, 1):
        r, w = os.pipe()                     # an open descriptor, but no semaphore behind it
        sem = C.c_void_p()
        rc = ctx.lib.vd_import_external_semaphore(ctx.h, r, timeline, C.byref(sem))
        if rc == abi.VD_OK:                  # (a driver that accepts any fd at import must at least hand out a handle that can be released)
            assert sem.value and ctx.lib.vd_release_external_semaphore(ctx.h, sem) in (abi.VD_OK, abi.VD_ERR_HIP)
        else:
            assert rc == abi.VD_ERR_HIP and not sem.value and b"hipImportExternalSemaphore" in ctx.lib.vd_last_error(ctx.h)
            os.close(r)
        os.close(w)
    # the context still works afterwards
    ctx.synchronize()


def test_trace_matches_golden(ctx):
    g = golden("trace_40.npz")
    scene = (g["tlas"], g["instances"], g["meshes"], g["bvh_nodes"], g["vertices"], g["indices"])
    h = ctx.trace(scene, g["rays"])
    assert np.array_equal(h["hit"], g["hit"])
    hit = g["hit"] == 1
    assert h["dist"].tobytes() == g["dist"].tobytes()                  # bit for bit, misses (1e30) included
    assert (h["dist"][~hit] == np.float32(1e30)).all()


def test_trace_seeded_scene_vs_oracle(ctx, oracle):
    """bvh_gpu.rs-shaped scene: a few meshes, many instances, 256x256 primary rays."""
    meshes_src = [synth.uv_sphere(1.0, 4), synth.knot_mesh(96, 24), synth.triangle_soup(64)]
    V, I, B = [], [], []
    infos = np.zeros(len(meshes_src), dtype=abi.MESH_INFO)
    vo = bo = no = 0
    for k, (v, i) in enumerate(meshes_src):
        nodes, idx = oracle.bvh_build(v, i)
        infos[k]["min"], infos[k]["max"] = synth.mesh_bounds(v)
        infos[k]["index_count"], infos[k]["base_index"] = len(idx), bo
        infos[k]["vertex_offset"], infos[k]["bvh_index"] = vo, no
        V.append(v); I.append(idx); B.append(nodes)
        vo += len(v); bo += len(idx); no += len(nodes)
    V, I, B = np.concatenate(V), np.concatenate(I), np.concatenate(B)
    inst = synth.instances(500, n_mesh=3, seed=synth.SEED_BASE + 8, extent=60.0, scale_range=(0.5, 3.0))
    tl = ctx.tlas_build(inst, infos)
    assert fields_equal(tl, oracle.tlas_build(inst, infos))
    cam = synth.camera_uniform(eye=(0, 2.5, 45), pitch_deg=0)
    rays = synth.primary_rays(cam, 256, 256)
    scene = (tl, inst, infos, B, V, I)
    want, max_stack = oracle.trace(scene, rays, threads=8)
    got = ctx.trace(scene, rays)
    assert np.array_equal(got["hit"], want["hit"]) and want["hit"].sum() > 1000
    hit = want["hit"] == 1
    assert got["dist"].tobytes() == want["dist"].tobytes()             # bit for bit, misses (1e30) included
    assert max_stack <= 64
    # device-pointer entry point + the occlusion query of the shadow pass (raytraced_shadows.wgsl:97-102): its flag
    # is the closest-hit traversal's `hit`
    import torch
    ds = ctx.device_scene(scene)
    d_rays = ctx.upload(rays)
    d_hits = ctx.empty(len(rays) * 16)
    d_any = torch.full((len(rays),), 7, dtype=torch.int32, device="cuda")
    ctx.trace_dev(ds, d_rays, len(rays), d_hits)
    ctx.trace_any_dev(ds, d_rays, len(rays), d_any)
    torch.cuda.synchronize()
    assert d_hits.cpu().numpy().view(abi.HIT).tobytes() == got.tobytes()
    assert np.array_equal(d_any.cpu().numpy().astype(np.uint32), want["hit"])
    # shadow rays from the hit points towards a point light: generator bit-exact, occlusion flags equal the oracle's
    pts = (rays["eye"] + rays["dir"] * want["dist"][:, None])[hit][:20000]
    nor = -rays["dir"][hit][:20000]
    light = np.array([3.0, 40.0, 20.0], np.float32)
    want_rays = oracle.shadow_rays(pts, nor, light)
    d_sr = ctx.empty(len(pts) * 32)
    ctx.shadow_rays_dev(ctx.upload(pts.astype(np.float32)), ctx.upload(nor.astype(np.float32)), len(pts), light, d_sr)
    d_occ = torch.zeros(len(pts), dtype=torch.int32, device="cuda")
    ctx.trace_any_dev(ds, d_sr, len(pts), d_occ)
    torch.cuda.synchronize()
    assert d_sr.cpu().numpy()[: len(pts) * 32].tobytes() == want_rays.tobytes()
    occ_want, _ = oracle.trace(scene, want_rays, threads=8)
    assert np.array_equal(d_occ.cpu().numpy().astype(np.uint32), occ_want["hit"])
    assert 0 < occ_want["hit"].sum() < len(pts)


@pytest.mark.parametrize("side", [96, 320])
def test_trace_supply_binning_and_prepared_triangles_are_order_only(ctx, oracle, ctx_options, side):
    """Round 3: rays are binned (sorted by origin cell and direction), handed to workgroups in chunks of the sorted
    order, and leaf triangles may come de-indexed from vd_trace_prepare_dev.  All of that changes the ORDER in which rays
    are walked and where a triangle's vertices are fetched from - never a result: every variant must reproduce the
    oracle's hit flags / distances, and all variants the same bytes, on a multi-mesh scene
    (vertex_offset / base_index / bvh_index all non-zero) with degenerate rays mixed in."""
    import torch
    meshes_src = [synth.uv_sphere(1.0, 4), synth.knot_mesh(96, 24), synth.triangle_soup(64)]
    V, I, B = [], [], []
    infos = np.zeros(len(meshes_src), dtype=abi.MESH_INFO)
    vo = bo = no = 0
    for k, (v, i) in enumerate(meshes_src):
        nodes, idx = oracle.bvh_build(v, i)
        infos[k]["min"], infos[k]["max"] = synth.mesh_bounds(v)
        infos[k]["index_count"], infos[k]["base_index"] = len(idx), bo
        infos[k]["vertex_offset"], infos[k]["bvh_index"] = vo, no
        V.append(v); I.append(idx); B.append(nodes)
        vo += len(v); bo += len(idx); no += len(nodes)
    V, I, B = np.concatenate(V), np.concatenate(I), np.concatenate(B)
    inst = synth.instances(300, n_mesh=3, seed=synth.SEED_BASE + 18, extent=50.0, scale_range=(0.5, 3.0))
    tl = ctx.tlas_build(inst, infos)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 40), pitch_deg=0), side, side)
    rays = np.concatenate([rays, rays[::7]])                    # duplicates: equal keys in the sort
    rays["dir"][5] = (0, 0, 0); rays["dir"][6] = (np.nan, 1, 0); rays["eye"][7] = (np.inf, 0, 0); rays["dir"][8] = (0, 0, -1)
    rays["eye"][9] = (1e30, -1e30, 0)
    scene = (tl, inst, infos, B, V, I)
    want, _ = oracle.trace(scene, rays, threads=8)
    ds = ctx.device_scene(scene)
    acc = ctx.trace_prepare(ds)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
    d_any = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
    first = None
    variants = [dict(), dict(chunk=64), dict(sort=1, sort_min=1), dict(sort=1, sort_min=1, chunk=64), dict(sort=1, sort_min=1, chunk=4096),
                dict(sort=1, sort_min=1, chunk=100), dict(chunk=1000),     # default: single rays from one counter, no binning
                # when a wave leaves its stepping loop to serve waiting lanes, and how many waves a CU runs
                {"yield": 4}, {"yield": 64}, {"yield": 64, "chunk": 64}, dict(waves=3),
                # the plain calls de-index the leaves themselves by default; without that they walk the indexed leaves
                dict(auto_prepare=0), {"auto_prepare": 0, "yield": 4}]
    for opts in variants:
        for k in ("sort", "sort_min", "chunk", "yield", "waves", "auto_prepare"):
            ctx_options("trace." + k, opts.get(k))
        for prep in (False, True):
            d_hits.zero_(); d_any.fill_(9)
            if prep:
                ctx.trace_prepared_dev(acc, d_rays, len(rays), d_hits); ctx.trace_any_prepared_dev(acc, d_rays, len(rays), d_any)
            else:
                ctx.trace_dev(ds, d_rays, len(rays), d_hits); ctx.trace_any_dev(ds, d_rays, len(rays), d_any)
            torch.cuda.synchronize()
            got = d_hits.cpu().numpy().view(abi.HIT)[: len(rays)]
            if first is None:
                first = got.tobytes()
                assert np.array_equal(got["hit"], want["hit"]) and want["hit"].sum() > 300
                hit = want["hit"] == 1
                assert got["dist"][hit].tobytes() == want["dist"][hit].tobytes()          # bit for bit
                # which instance and which triangle: the walk carries the TLAS leaf and resolves the instance when the ray retires
                assert np.array_equal(got["instance"][hit], want["instance"][hit])
                assert np.array_equal(got["triangle"][hit], want["triangle"][hit])
            assert got.tobytes() == first, (opts, prep)
            assert np.array_equal(d_any.cpu().numpy().astype(np.uint32), want["hit"]), (opts, prep)
    acc.close()
    # argument checks of the prepared form
    assert ctx.lib.vd_trace_prepared_dev(ctx.h, None, abi.ptr(d_rays), 4, abi.ptr(d_hits)) == abi.VD_ERR_INVALID_ARG
    bad = ctx.device_scene((tl, inst, infos, B, V, I[: len(I) - 30]))         # the last mesh's index range runs past the buffer
    with pytest.raises(VoidinError) as e:
        ctx.trace_prepare(bad)
    assert e.value.code == abi.VD_ERR_INVALID_ARG


def _fan_scene(oracle, n_inst, extent, seed):
    inst = synth.instances(n_inst, n_mesh=2, seed=seed, extent=extent, scale_range=(0.5, 2.0))
    meshes_src = [synth.knot_mesh(64, 16), synth.uv_sphere(1.0, 6)]
    V, I, B = [], [], []
    infos = np.zeros(2, dtype=abi.MESH_INFO)
    vo = bo = no = 0
    for k, (v, i) in enumerate(meshes_src):
        nodes, idx = oracle.bvh_build(v, i)
        infos[k]["min"], infos[k]["max"] = synth.mesh_bounds(v)
        infos[k]["index_count"], infos[k]["base_index"], infos[k]["vertex_offset"], infos[k]["bvh_index"] = len(idx), bo, vo, no
        V.append(np.asarray(v, dtype=np.float32).reshape(-1, 3)); I.append(idx); B.append(nodes)
        vo += len(V[-1]); bo += len(idx); no += len(nodes)
    return (oracle.tlas_build(inst, infos), inst, infos, np.concatenate(B), np.concatenate(V), np.concatenate(I))


@pytest.mark.parametrize("n_inst,extent", [(300, 100.0), (1500, 60.0)])
def test_trace_fan_out_gives_the_same_bytes(ctx, oracle, ctx_options, n_inst, extent):
    """VD_OPT_TRACE_FAN: a call with at least one ray per lane of the persistent grid runs as up to four launches; a draining
    wave turns its last rays into one job per TLAS subtree on their stacks, and the ray's record is the minimum over its
    jobs of (t, visit order).  The output must not change by a bit: closest-hit records (distance, instance AND triangle:
    the tie-break by visit order) and occlusion flags of 1 (no fan-out), 2, 3 and 4 launches are the same bytes, through
    the prepared, the plain and the indexed walk, and a sample of the rays equals the oracle.  The second scene is dense
    (1500 overlapping instances: rays of thousands of steps, jobs that fan out again)."""
    import torch
    scene = _fan_scene(oracle, n_inst, extent, synth.SEED_BASE + 8)
    side = 720                                         # 518 400 rays >= 256 CUs x 24 waves x 64 lanes
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 0.75 * extent), pitch_deg=0), side, side)
    n = len(rays)
    ds = ctx.device_scene(scene)
    acc = ctx.trace_prepare(ds)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(n * 16)
    d_any = torch.zeros(n, dtype=torch.int32, device="cuda")
    ref_bytes = None
    for mode in ("prepared", "plain", "indexed"):
        ctx_options("trace.auto_prepare", 0 if mode == "indexed" else None)
        for fan in (1, 2, 3, 4):
            ctx_options("trace.fan", fan)
            d_hits.fill_(0xEE); d_any.fill_(7)
            if mode == "prepared":
                ctx.trace_prepared_dev(acc, d_rays, n, d_hits); ctx.trace_any_prepared_dev(acc, d_rays, n, d_any)
            else:
                ctx.trace_dev(ds, d_rays, n, d_hits); ctx.trace_any_dev(ds, d_rays, n, d_any)
            got = (d_hits.cpu().numpy().tobytes(), d_any.cpu().numpy().tobytes())
            if ref_bytes is None:
                ref_bytes = got
                sub = np.ascontiguousarray(rays[::16])
                want, _ = oracle.trace(scene, sub, threads=8)
                h = np.frombuffer(got[0], dtype=abi.HIT)[::16]
                hit = want["hit"] == 1
                assert np.array_equal(h["hit"], want["hit"]) and hit.sum() > 3000
                for f in ("dist", "instance", "triangle"):
                    assert h[f][hit].tobytes() == want[f][hit].tobytes(), f
                assert np.array_equal(np.frombuffer(got[1], dtype=np.uint32)[::16], want["hit"])
            if got[0] != ref_bytes[0]:
                a, b = np.frombuffer(got[0], dtype=abi.HIT), np.frombuffer(ref_bytes[0], dtype=abi.HIT)
                bad = np.nonzero((a["hit"] != b["hit"]) | (a["dist"].view(np.uint32) != b["dist"].view(np.uint32)) |
                                 (a["instance"] != b["instance"]) | (a["triangle"] != b["triangle"]))[0]
                assert False, (mode, fan, len(bad), bad[:5].tolist(), a[bad[:3]].tolist(), b[bad[:3]].tolist())
            assert got[1] == ref_bytes[1], (mode, fan, "occlusion", int((np.frombuffer(got[1], np.uint32) != np.frombuffer(ref_bytes[1], np.uint32)).sum()))
    acc.close()


@pytest.mark.parametrize("slots", [0, 1000, 100_000])
def test_trace_fan_out_with_a_job_list_that_fills_up(ctx, oracle, ctx_options, slots):
    """The job list is full (VD_OPT_TRACE_FAN_SLOTS: 0, a few, some of what the call wants): waves that cannot reserve keep
    their rays and run them to the end, the reserved-but-unusable slots read as empty - same bytes as the call without fan-out."""
    import torch
    scene = _fan_scene(oracle, 1500, 60.0, synth.SEED_BASE + 9)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 45.0), pitch_deg=0), 720, 720)
    n = len(rays)
    ds = ctx.device_scene(scene)
    acc = ctx.trace_prepare(ds)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(n * 16)
    d_any = torch.zeros(n, dtype=torch.int32, device="cuda")
    out = {}
    for fan, cap in ((1, None), (3, slots), (4, slots)):
        ctx_options("trace.fan", fan)
        ctx_options("trace.fan_slots", cap)
        d_hits.fill_(0xEE); d_any.fill_(7)
        ctx.trace_prepared_dev(acc, d_rays, n, d_hits); ctx.trace_any_prepared_dev(acc, d_rays, n, d_any)
        out[fan] = (d_hits.cpu().numpy().tobytes(), d_any.cpu().numpy().tobytes())
    assert out[3] == out[1] and out[4] == out[1]
    acc.close()


def test_trace_records_survive_stale_slots_and_report_bad_leaves(ctx, oracle):
    """The per-call records of the walk (trace.hip, records_kernel): the two children of a TLAS node sit side by side at
    the LEFT child's index, tagged with the right child's.  An unreachable slot of the TLAS array may name the same left
    child with another right child (the reference's array keeps such slots: tlas.rs:62-84) - whichever of the two wrote
    the slot, the walk must notice a foreign tag and read the nodes themselves.  And a leaf whose instance index or whose
    mesh's BLAS root lies outside the scene's buffers is an error when a ray enters it, not a wild read."""
    import torch
    v, i = synth.knot_mesh(64, 16)
    nodes, idx = oracle.bvh_build(v, i)
    infos = np.zeros(1, dtype=abi.MESH_INFO)
    infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(v)
    infos[0]["index_count"] = len(idx)
    inst = synth.instances(200, n_mesh=1, seed=synth.SEED_BASE + 33, extent=40.0, scale_range=(0.5, 2.0))
    tl = oracle.tlas_build(inst, infos)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 45), pitch_deg=0), 160, 160)
    want, _ = oracle.trace((tl, inst, infos, nodes, v, idx), rays, threads=8)
    assert want["hit"].sum() > 500
    # reachable nodes, and slots nobody reaches
    reach = np.zeros(len(tl), dtype=bool); todo = [0]
    while todo:
        k = todo.pop(); reach[k] = True
        lr = int(tl["left_right"][k])
        if lr: todo += [lr & 0xffff, lr >> 16]
    free = np.nonzero(~reach)[0]
    interior = np.nonzero(reach & (tl["left_right"] != 0))[0]
    assert len(free) >= 1 and len(interior) > 50
    stale = tl.copy()
    for n, k in enumerate(free[:64]):                     # every free slot claims some real node's left child, with another partner
        lr = int(tl["left_right"][interior[(7 * n) % len(interior)]])
        stale["left_right"][k] = (lr & 0xffff) | (int(interior[(3 * n + 1) % len(interior)]) << 16)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
    d_any = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
    for tlas_nodes in (tl, stale):
        ds = ctx.device_scene((tlas_nodes, inst, infos, nodes, v, idx))
        for rep in range(3):                              # which writer wins a contested slot may differ from launch to launch
            ctx.trace_dev(ds, d_rays, len(rays), d_hits); ctx.trace_any_dev(ds, d_rays, len(rays), d_any)
            got = d_hits.cpu().numpy().view(abi.HIT)[: len(rays)]
            hit = want["hit"] == 1
            assert np.array_equal(got["hit"], want["hit"])
            for f in ("dist", "instance", "triangle"):
                assert got[f][hit].tobytes() == want[f][hit].tobytes(), f
            assert np.array_equal(d_any.cpu().numpy().astype(np.uint32), want["hit"])
    # a leaf that names an instance past the buffer / a mesh whose BLAS root is past the node buffer: VD_ERR_INVALID_ARG
    leaf = int(np.nonzero(reach & (tl["left_right"] == 0))[0][0])
    bad_leaf = tl.copy(); bad_leaf["instance_idx"][leaf] = len(inst) + 5
    bad_mesh = infos.copy(); bad_mesh[0]["bvh_index"] = len(nodes) + 7
    # rays aimed at every instance: some ray enters the poisoned leaf
    dirs = np.array([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0)], dtype=np.float32)
    aim = np.zeros(len(inst) * 6, dtype=abi.RAY)          # from six sides: something in front may hide the leaf from one of them
    for k, d in enumerate(dirs):
        aim["eye"][k::6] = inst["transform"][:, 12:15] - 30.0 * d; aim["dir"][k::6] = d
    d_aim, d_h2 = ctx.upload(aim), ctx.empty(len(aim) * 16)
    for scene in ((bad_leaf, inst, infos, nodes, v, idx), (tl, inst, bad_mesh, nodes, v, idx)):
        with pytest.raises(VoidinError) as e:
            ctx.trace_dev(ctx.device_scene(scene), d_aim, len(aim), d_h2)
        assert e.value.code == abi.VD_ERR_INVALID_ARG
    ctx.trace_dev(ctx.device_scene((tl, inst, infos, nodes, v, idx)), d_aim, len(aim), d_h2)      # the context is fine afterwards


def test_trace_falls_back_to_indexed_leaves_when_a_mesh_cannot_be_deindexed(ctx, oracle):
    """A plain vd_trace_dev de-indexes the leaf triangles itself (36 B per triangle, index-buffer order) - which needs every
    mesh's base_index to be a multiple of 3.  The reference's fetch_vertex (bvh.wgsl:30-33) has no such need: with an index
    buffer that starts one word late the call must notice on the device and walk the indexed leaves instead - same hits."""
    v, i = synth.knot_mesh(48, 12)
    nodes, idx = oracle.bvh_build(v, i)
    infos = np.zeros(1, dtype=abi.MESH_INFO)
    infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(v)
    infos[0]["index_count"], infos[0]["base_index"] = len(idx), 1
    idx1 = np.concatenate([np.zeros(1, dtype=np.uint32), idx])
    inst = synth.instances(60, n_mesh=1, seed=synth.SEED_BASE + 34, extent=25.0, scale_range=(0.5, 2.0))
    tl = oracle.tlas_build(inst, infos)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 30), pitch_deg=0), 128, 128)
    scene = (tl, inst, infos, nodes, v, idx1)
    want, _ = oracle.trace(scene, rays, threads=8)
    assert want["hit"].sum() > 300
    ds = ctx.device_scene(scene)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
    ctx.trace_dev(ds, d_rays, len(rays), d_hits)
    got = d_hits.cpu().numpy().view(abi.HIT)[: len(rays)]
    hit = want["hit"] == 1
    assert np.array_equal(got["hit"], want["hit"])
    for f in ("dist", "instance", "triangle"):
        assert got[f][hit].tobytes() == want[f][hit].tobytes(), f
    with pytest.raises(VoidinError) as e:                    # the explicit per-scene call says why it cannot
        ctx.trace_prepare(ds)
    assert e.value.code == abi.VD_ERR_INVALID_ARG


def test_trace_many_small_scenes_vs_oracle(ctx, oracle, ctx_options):
    """Thirty small scenes of every shape the walk's state machine has a special case for: meshes of one, two and three
    triangles (the BLAS root is a leaf), a single instance (the TLAS root is a leaf), several meshes side by side (non-zero
    bvh_index / base_index / vertex_offset), rays from everywhere in all directions (from inside boxes too).  Hit flag,
    distance, instance and triangle against the oracle, the plain call (de-indexing its leaves itself), the indexed
    leaves and the prepared scene; the occlusion query's flags on each."""
    import torch
    rng = np.random.default_rng(20261003)
    pool = [synth.triangle_soup(1, seed=5), synth.triangle_soup(2, seed=6), synth.triangle_soup(3, seed=7), synth.triangle_soup(10, seed=8),
            synth.uv_sphere(1.0, 3), synth.knot_mesh(24, 8), synth.plane_mesh(2.0, 2.0)]
    built = [(v, *oracle.bvh_build(v, i)) for v, i in pool]            # (verts, nodes, reordered indices)
    total_hits = 0
    for scene_no in range(30):
        picks = rng.choice(len(built), size=int(rng.integers(1, 4)), replace=False)
        V, I, B = [], [], []
        infos = np.zeros(len(picks), dtype=abi.MESH_INFO)
        vo = bo = no = 0
        for k, m in enumerate(picks):
            v, nodes, idx = built[m]
            infos[k]["min"], infos[k]["max"] = synth.mesh_bounds(v)
            infos[k]["index_count"], infos[k]["base_index"] = len(idx), bo
            infos[k]["vertex_offset"], infos[k]["bvh_index"] = vo, no
            V.append(v); I.append(idx); B.append(nodes)
            vo += len(v); bo += len(idx); no += len(nodes)
        V, I, B = np.concatenate(V), np.concatenate(I), np.concatenate(B)
        n_inst = int(rng.choice([1, 1, 2, 3, 7, 40]))
        inst = synth.instances(n_inst, n_mesh=len(picks), seed=synth.SEED_BASE + 100 + scene_no, extent=12.0, scale_range=(0.5, 3.0))
        tl = oracle.tlas_build(inst, infos)
        n_rays = 1500
        rays = np.zeros(n_rays, dtype=abi.RAY)
        eye = rng.normal(size=(n_rays, 3)).astype(np.float32) * rng.choice([0.5, 6.0, 25.0], size=(n_rays, 1)).astype(np.float32)
        target = rng.normal(size=(n_rays, 3)).astype(np.float32) * 5.0
        d = target - eye
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6).astype(np.float32)
        axis = rng.integers(0, n_rays, size=60)
        d[axis] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, size=60)] * rng.choice([-1.0, 1.0], size=(60, 1)).astype(np.float32)
        rays["eye"], rays["dir"] = eye, d.astype(np.float32)
        scene = (tl, inst, infos, B, V, I)
        want, _ = oracle.trace(scene, rays, threads=8)
        hit = want["hit"] == 1
        total_hits += int(hit.sum())
        ds = ctx.device_scene(scene)
        acc = ctx.trace_prepare(ds)
        d_rays, d_hits = ctx.upload(rays), ctx.empty(n_rays * 16)
        d_any = torch.zeros(n_rays, dtype=torch.int32, device="cuda")
        for mode in ("plain", "indexed", "prepared"):
            ctx_options("trace.auto_prepare", 0 if mode == "indexed" else None)
            d_hits.zero_(); d_any.fill_(7)
            if mode == "prepared":
                ctx.trace_prepared_dev(acc, d_rays, n_rays, d_hits); ctx.trace_any_prepared_dev(acc, d_rays, n_rays, d_any)
            else:
                ctx.trace_dev(ds, d_rays, n_rays, d_hits); ctx.trace_any_dev(ds, d_rays, n_rays, d_any)
            got = d_hits.cpu().numpy().view(abi.HIT)[:n_rays]
            assert np.array_equal(got["hit"], want["hit"]), (scene_no, mode)
            for f in ("dist", "instance", "triangle"):
                assert got[f][hit].tobytes() == want[f][hit].tobytes(), (scene_no, mode, f)
            assert np.array_equal(d_any.cpu().numpy().astype(np.uint32), want["hit"]), (scene_no, mode)
        acc.close()
    assert total_hits > 2000


def test_trace_axis_aligned_and_degenerate_rays(ctx, oracle):
    """The WGSL walk multiplies by inv_dir = 1 / dir: rays with zero direction components give inf and 0 * inf = NaN
    in the slab test, which must flow through min / max as in the restated shader (WGSL min / max = IEEE minNum /
    maxNum, SURVEY 8a C2'); plus rays that start inside boxes, a ray with no direction, NaN and inf origins."""
    v, i = synth.uv_sphere(1.0, 6)
    nodes, idx = oracle.bvh_build(v, i)
    infos = np.zeros(1, dtype=abi.MESH_INFO)
    infos["min"], infos["max"], infos["index_count"] = *synth.mesh_bounds(v), len(idx)
    inst = synth.instances(60, n_mesh=1, seed=synth.SEED_BASE + 33, extent=12.0, scale_range=(0.8, 2.5))
    tl = ctx.tlas_build(inst, infos)
    k = 2000
    u = synth.uniform01(synth.SEED_BASE + 34, 0, 3 * k).reshape(k, 3).astype(np.float64)
    rays = np.zeros(6 * k + 4, dtype=abi.RAY)
    for a in range(3):
        for sgn in (0, 1):
            blk = rays[(2 * a + sgn) * k: (2 * a + sgn + 1) * k]
            e = (u * 2 - 1) * 7.0
            e[:, a] = -20.0 if sgn == 0 else 20.0
            blk["eye"] = e
            blk["dir"][:, a] = 1.0 if sgn == 0 else -1.0
    rays["dir"][-4] = (0, 0, 0)
    rays["dir"][-3] = (np.nan, 1, 0)
    rays["eye"][-2] = (np.inf, 0, 0); rays["dir"][-2] = (-1, 0, 0)
    rays["eye"][-1] = inst["transform"][0][12:15]; rays["dir"][-1] = (0, 1, 0)      # from the centre of an instance
    scene = (tl, inst, infos, nodes, v, idx)
    want, _ = oracle.trace(scene, rays, threads=8)
    got = ctx.trace(scene, rays)
    assert np.array_equal(got["hit"], want["hit"]) and want["hit"].sum() > 500
    hit = want["hit"] == 1
    assert got["dist"].view(np.uint32).tobytes() == want["dist"].view(np.uint32).tobytes()      # bit for bit, NaN-free or not


def _boxes_as_scene(boxes):
    """Identity transforms + one MeshInfo per box: the leaf boxes are exactly `boxes` (tests/test_tlas_index_model.py)."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 6)
    n = len(boxes)
    meshes = np.zeros(n, dtype=abi.MESH_INFO)
    meshes["min"], meshes["max"] = boxes[:, :3], boxes[:, 3:]
    inst = np.zeros(n, dtype=abi.INSTANCE)
    eye = np.eye(4, dtype=np.float32).reshape(16)
    inst["transform"], inst["inv_transform"] = eye, eye
    inst["mesh"] = np.arange(n, dtype=np.uint32)
    return inst, meshes


@pytest.mark.parametrize("helpers", ["1", "0", "2"])
def test_indexed_build_on_ties_nesting_and_stale_slots(ctx, oracle, ctx_options, helpers):
    """The indexed build (tlas.hip, "build, indexed": pruned queries, two at a time - the one the chain needs and the one a
    merge would need) forced onto small inputs made of what can break it: identical and nested boxes (every union area
    ties: the slot index decides), lattices, zero extents, chains that keep ending on the last slot (the stale index of
    tlas.rs:72-75), and ordinary clouds; and inputs it must hand to the plain chain (NaN, inf, huge)."""
    ctx_options("tlas.index_min", 65)
    ctx_options("tlas.phase2", 64)
    ctx_options("tlas.refresh", 37)
    ctx_options("tlas.spec", int(helpers))                                      # with / without the four helper waves; 2: a helper WORKGROUP

    def cloud(n, seed, extent=60.0, size=4.0):
        u = synth.uniform01(seed, 0, 6 * n).reshape(n, 6).astype(np.float32)
        c = (u[:, :3] - 0.5) * np.float32(extent)
        h = u[:, 3:] * np.float32(size) * np.float32(0.5)
        return np.concatenate([c - h, c + h], axis=1)
    rng = np.random.default_rng(5)
    cases = [cloud(n, 100 + n) for n in (65, 100, 257, 700, 1500, 3000)]
    cases.append(np.tile(np.array([[0, 0, 0, 1, 1, 1]], np.float32), (90, 1)))
    big = np.array([[-50, -50, -50, 50, 50, 50]], np.float32)
    cases.append(np.concatenate([cloud(160, 7, 40.0, 2.0), big, cloud(160, 8, 40.0, 2.0), big * np.float32(0.5)]))
    g = np.stack(np.meshgrid(np.arange(8), np.arange(7), np.arange(6), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float32) * 3
    cases.append(np.concatenate([g, g + 1], axis=1))
    cases.append(np.concatenate([g, g + 1], axis=1)[rng.permutation(len(g))])
    line = np.zeros((297, 6), np.float32); line[:, 0] = np.arange(297); line[:, 3] = np.arange(297) + 0.5; line[:, 4:] = 0.5
    cases.append(line)
    cases.append(np.concatenate([cloud(200, 9), cloud(200, 9)]))
    flat = cloud(350, 11); flat[:, 2] = 0; flat[:, 5] = 0
    cases.append(flat)
    n = 400
    x = np.cumsum(1.0 + 0.01 * np.arange(n))[::-1].astype(np.float32)          # the closest pair is always the last two slots
    stale = np.zeros((n, 6), np.float32); stale[:, 0], stale[:, 3] = x, x + np.float32(0.25); stale[:, 4:] = 0.25
    cases.append(stale)
    # zeros of both signs on the box faces: the unions must pick -0 as the minimum and +0 as the maximum like the oracle's
    # total order (the indexed build uses single v_min_f32 / v_max_f32 instructions, which order the zeros the same way)
    zed = cloud(600, 13, 30.0, 20.0)
    sign = np.where(rng.random((600, 6)) < 0.5, np.float32(-0.0), np.float32(0.0)).astype(np.float32)
    snap = rng.random((600, 6)) < 0.4
    snap[:, :3] &= zed[:, :3] <= 0                                                # keeps mn <= 0 <= mx where a face is moved onto zero
    snap[:, 3:] &= zed[:, 3:] >= 0
    zed[snap] = sign[snap]
    cases.append(zed)
    for k, boxes in enumerate(cases):
        inst, meshes = _boxes_as_scene(boxes)
        want = oracle.tlas_build(inst, meshes)
        got = ctx.tlas_build(inst, meshes)
        assert fields_equal(got, want), f"case {k} ({len(boxes)} boxes)"
        wide = ctx.tlas_build(inst, meshes, wide=True)
        assert np.array_equal(wide["left"] + (wide["right"] << 16), want["left_right"]) and wide["min"].tobytes() == want["min"].tobytes()
    for bad in (np.nan, np.inf, 1e19):                                          # precondition fails -> plain chain, same answer
        boxes = cloud(300, 3)
        boxes[41, 3] = bad
        inst, meshes = _boxes_as_scene(boxes)
        want, got = oracle.tlas_build(inst, meshes), ctx.tlas_build(inst, meshes)
        assert got["left_right"].tobytes() == want["left_right"].tobytes()
        assert got["min"].view(np.uint32).tobytes() == want["min"].view(np.uint32).tobytes() and got["max"].view(np.uint32).tobytes() == want["max"].view(np.uint32).tobytes()


@pytest.mark.parametrize("chain_lds", [0, 1])
def test_tlas_chain_slot_arrays_in_lds_or_in_memory(ctx, oracle, ctx_options, chain_lds):
    """VD_OPT_TLAS_CHAIN_LDS: the plain scans - the whole build of a small scene, and the end of an indexed build - read their
    slot arrays from LDS (default) or from memory.  Same nodes either way, on a small scene, on one that fills the LDS
    allocation to the last slots, and on an indexed build forced onto 3000 instances with the hand-over at 700 clusters."""
    ctx_options("tlas.chain_lds", chain_lds)
    meshes = synth.mesh_infos()
    for n, forced in ((1500, False), (5600, False), (3000, True)):
        if forced:
            ctx_options("tlas.index_min", 65); ctx_options("tlas.phase2", 700); ctx_options("tlas.refresh", 300)
        inst = synth.instances(n, seed=synth.SEED_BASE + 40 + n % 5, extent=200.0)
        assert fields_equal(ctx.tlas_build(inst, meshes), oracle.tlas_build(inst, meshes)), (n, forced, chain_lds)


@pytest.mark.parametrize("spin_limit", [None, 0, 1, 40])
def test_indexed_build_with_a_helper_workgroup_that_is_late_or_absent(ctx, oracle, ctx_options, spin_limit):
    """VD_OPT_TLAS_SPEC = 2 (tlas.hip, "the two-workgroup form"): the speculative query runs on a second workgroup of the
    same XCC and comes back through a mailbox.  Bounded waits decide what happens when the partner is not there: with a
    spin limit of 0 the chain's workgroup closes the role at once (nobody can claim it: the launch leaves everything
    untouched and the one-workgroup form queued behind it builds the tree); with 1 or 40 polls the first answers are
    late, the chain stops asking and finishes on its own.  The same nodes every time."""
    ctx_options("tlas.index_min", 65)
    ctx_options("tlas.phase2", 64)
    ctx_options("tlas.refresh", 200)
    ctx_options("tlas.spec", 2)
    if spin_limit is not None:
        ctx_options("tlas.spin_limit", spin_limit)
    for n, seed in ((3000, 21), (9000, 22)):
        inst = synth.instances(n, seed=synth.SEED_BASE + seed, extent=150.0)
        meshes = synth.mesh_infos()
        want = oracle.tlas_build(inst, meshes)
        for _ in range(2):                                                      # the mailbox is cleared per build
            assert fields_equal(ctx.tlas_build(inst, meshes), want), (n, spin_limit)


@pytest.mark.parametrize("spec", [1, 2])
def test_indexed_build_declines_when_every_area_ties(ctx, oracle, ctx_options, spec):
    """Boxes that defeat the pruning (thousands of identical instances; a nest of boxes that all contain the origin cube):
    every union area ties, a query through the index would look into most slices, so the indexed build declines on the
    device and the plain chain queued behind it (16 workgroups at this size) starts over.  Same nodes as the oracle."""
    import time
    ctx_options("tlas.spec", spec)                                              # 2: the chain's workgroup also tells its helper workgroup to leave
    n = 13000
    same = np.tile(np.array([[-1, -2, -3, 1, 2, 3]], np.float32), (n, 1))
    rng = np.random.default_rng(3)
    r = (1.0 + rng.random((n, 1)) * 50).astype(np.float32)
    nest = np.concatenate([-r, -r, -r, r, r, r], axis=1).astype(np.float32)          # cubes around the origin: totally ordered by inclusion
    for boxes in (same, nest):
        inst, meshes = _boxes_as_scene(boxes)
        want = oracle.tlas_build(inst, meshes)
        t = time.perf_counter()
        got = ctx.tlas_build(inst, meshes)
        dt = time.perf_counter() - t
        assert fields_equal(got, want)
        assert dt < 3.0, f"{dt:.1f} s: the build did not fall back to the plain chain"


def _chain_scene(oracle, n_leaves=200):
    """A TLAS that is one long chain - interior node k = {interior k - 1, leaf k}, spheres one unit apart along +x - in the
    reference's node layout (leaves at 1..N, interior nodes behind them, node 0 = a copy of the root).  A ray from x = -5
    along +x finds the chain as its NEAR child at every level and pushes the leaf: N - 1 pending entries when it reaches
    sphere 0 - more than the 128 a lane holds (trace.hip), fewer than the oracle's 256."""
    v, i = synth.uv_sphere(0.4, 2)
    v = np.asarray(v, np.float32).reshape(-1, 3)
    nodes, idx = oracle.bvh_build(v, i)
    infos = np.zeros(1, dtype=abi.MESH_INFO)
    infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(v)
    infos[0]["index_count"] = len(idx)
    N = n_leaves
    inst = np.zeros(N, dtype=abi.INSTANCE)
    T = np.tile(np.eye(4, dtype=np.float32), (N, 1, 1))
    T[:, 3, 0] = np.arange(N, dtype=np.float32)            # column-major storage: translation in elements 12..14
    inst["transform"] = T.reshape(N, 16)
    Ti = T.copy(); Ti[:, 3, 0] *= np.float32(-1)
    inst["inv_transform"] = Ti.reshape(N, 16)
    tl = np.zeros(2 * N, dtype=abi.TLAS_NODE)
    mn, mx = infos[0]["min"], infos[0]["max"]
    for k in range(N):
        tl[1 + k]["min"] = mn + np.array([k, 0, 0], np.float32)
        tl[1 + k]["max"] = mx + np.array([k, 0, 0], np.float32)
        tl[1 + k]["left_right"], tl[1 + k]["instance_idx"] = 0, k
    prev = 1
    for k in range(1, N):
        me = N + k
        tl[me]["min"] = np.minimum(tl[prev]["min"], tl[1 + k]["min"])
        tl[me]["max"] = np.maximum(tl[prev]["max"], tl[1 + k]["max"])
        tl[me]["left_right"], tl[me]["instance_idx"] = prev | ((1 + k) << 16), 0xFFFFFFFF
        prev = me
    tl[0] = tl[prev]
    return (tl, inst, infos, nodes, v, idx)


def _chain_blas_scene(oracle, n_tris=190):
    """ONE instance whose BLAS is a hand-made chain in the reference's node layout (node 0 the root, node 1 unused, child pairs
    behind): every interior node splits off the FARTHEST triangle as a one-triangle leaf and keeps the rest.  A ray from
    x = -1 along +x takes the rest as its near child at every level and pushes the leaf (bvh.wgsl:56-74 pushes the far child
    while nothing is hit): n - 3 pending BLAS entries before the first triangle test."""
    n = n_tris
    tri = np.zeros((n, 3, 3), np.float32)
    tri[:, :, 0] = np.arange(n, dtype=np.float32)[:, None]
    tri[:, 0, 1:] = [-0.5, -0.5]; tri[:, 1, 1:] = [0.0, 0.6]; tri[:, 2, 1:] = [0.5, -0.5]
    verts, idx = tri.reshape(-1, 3).copy(), np.arange(3 * n, dtype=np.uint32)
    levels = n - 3
    nodes = np.zeros(2 + 2 * levels, dtype=abi.BVH_NODE)
    box = lambda a, b: (tri[a:b].reshape(-1, 3).min(axis=0), tri[a:b].reshape(-1, 3).max(axis=0))
    cur, hi = 0, n
    for j in range(levels):
        pair = 2 + 2 * j
        nodes[cur]["min"], nodes[cur]["max"] = box(0, hi)
        nodes[cur]["left_first"], nodes[cur]["count"] = pair, 0
        nodes[pair + 1]["min"], nodes[pair + 1]["max"] = box(hi - 1, hi)
        nodes[pair + 1]["left_first"], nodes[pair + 1]["count"] = hi - 1, 1
        cur, hi = pair, hi - 1
    nodes[cur]["min"], nodes[cur]["max"] = box(0, hi)
    nodes[cur]["left_first"], nodes[cur]["count"] = 0, hi
    infos = np.zeros(1, dtype=abi.MESH_INFO)
    infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(verts)
    infos[0]["index_count"] = len(idx)
    inst = np.zeros(1, dtype=abi.INSTANCE)
    inst["transform"] = inst["inv_transform"] = np.eye(4, dtype=np.float32).reshape(16)
    return (oracle.tlas_build(inst, infos), inst, infos, nodes, verts, idx)


@pytest.mark.parametrize("kind", ["tlas chain", "blas chain"])
@pytest.mark.parametrize("n_rays", [2_000, 420_000])       # one launch / a call large enough to fan out (>= one ray per lane of the grid)
def test_rays_deeper_than_a_lanes_stack_are_walked_again(ctx, ctx_options, oracle, n_rays, kind):
    """The call is TOTAL: a ray that needs more than the 128 stack entries a lane holds used to end the call
    with VD_ERR_STACK_OVERFLOW; now it is walked again with its stack in global memory and the record is the oracle's, bit for
    bit - closest hit and occlusion, plain / indexed / prepared leaves, one launch and fanned out, deep rays mixed with cheap ones."""
    import torch
    scene = _chain_scene(oracle) if kind == "tlas chain" else _chain_blas_scene(oracle)      # depth on the TLAS side / inside one instance
    rng = np.random.default_rng(128)
    rays = np.zeros(n_rays, dtype=abi.RAY)
    rays["eye"] = (rng.random((n_rays, 3)).astype(np.float32) - np.float32(0.5)) * np.array([0.0, 0.6, 0.6], np.float32) + np.array([-5.0, 0, 0], np.float32)
    rays["dir"] = np.array([1.0, 0.0, 0.0], np.float32)
    far_side = rng.random(n_rays) < 0.5                      # from the far end the leaf is the near child: no depth at all
    rays["eye"][far_side, 0] = np.float32(250.0)
    rays["dir"][far_side] = np.array([-1.0, 0.0, 0.0], np.float32)
    cheap = rng.random(n_rays) < (0.3 if n_rays < 10_000 else 0.97)   # the large call: most rays leave the scene at once
    rays["dir"][cheap] = np.array([0.0, 1.0, 0.0], np.float32)
    want, deepest = oracle.trace(scene, rays, threads=16)
    assert 128 < deepest <= 256                              # deeper than a lane's stack, within the oracle's
    assert want["hit"].sum() > n_rays // 100
    ds = ctx.device_scene(scene)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(n_rays * 16)
    d_any = torch.zeros(n_rays, dtype=torch.int32, device="cuda")
    acc = ctx.trace_prepare(ds)
    for fan in (1, 3):
        ctx_options("trace.fan", fan)
        for mode in ("plain", "indexed", "prepared"):
            ctx_options("trace.auto_prepare", 0 if mode == "indexed" else None)
            d_hits.zero_(); d_any.zero_()
            if mode == "prepared":
                ctx.trace_prepared_dev(acc, d_rays, n_rays, d_hits); ctx.trace_any_prepared_dev(acc, d_rays, n_rays, d_any)
            else:
                ctx.trace_dev(ds, d_rays, n_rays, d_hits); ctx.trace_any_dev(ds, d_rays, n_rays, d_any)
            got = d_hits.cpu().numpy()[: n_rays * 16].view(abi.HIT)
            assert got.tobytes() == np.ascontiguousarray(want).tobytes(), (fan, mode)
            assert np.array_equal(d_any.cpu().numpy().astype(np.uint32), want["hit"]), (fan, mode)
    acc.close()
    # a call that overflows nowhere, right after: the bitmap was left clean
    easy = rays[cheap | far_side]
    w2, d2 = oracle.trace(scene, easy, threads=16)
    assert d2 <= 64 and ctx.trace(scene, easy).tobytes() == np.ascontiguousarray(w2).tobytes()


def test_a_fuzz_scene_that_used_to_overflow(ctx, oracle):
    """profiles/r05_fuzz.log: seed 22, case 85 of tools/fuzz_trace.py - 1 500 instances, oracle's deepest stack 132 - ended
    every one of its six walks with VD_ERR_STACK_OVERFLOW in round 5."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_trace
    out = []
    bad, deep = fuzz_trace.run(86, seed=22, ctx=ctx, log=out.append, only={85})
    assert bad == 0, "\n".join(out[:10])
    assert deep == 1
