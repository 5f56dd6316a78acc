"""VD_OPT_TRACE_TIGHT_TLAS: a prepared scene with its own top level over tight world boxes (default off; never part of a
bit-exact parity run).  Acceptance is north_star's own: hit flags equal to the reference walk's, distances within 1e-5
(absolute) - checked against the oracle's vd_ref_trace on the stress-scene shape, a bvh_gpu.rs-shaped scene, the
reference's helmet as its demo views it, thirty small random scenes, grazing rays, and instances the option must NOT
tighten (inv_transform that does not invert transform)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from voidin_amd import abi, synth

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-5          # north_star: "traversal hit distances match within 1e-5" - absolute, whatever the distance


def make_scene(oracle, meshes_src, inst):
    V, I, B = [], [], []
    infos = np.zeros(len(meshes_src), dtype=abi.MESH_INFO)
    vo = bo = no = 0
    for k, (v, i) in enumerate(meshes_src):
        nodes, idx = oracle.bvh_build(v, i)
        infos[k]["min"], infos[k]["max"] = synth.mesh_bounds(v)
        infos[k]["index_count"], infos[k]["base_index"] = len(idx), bo
        infos[k]["vertex_offset"], infos[k]["bvh_index"] = vo, no
        V.append(np.asarray(v, dtype=np.float32).reshape(-1, 3)); I.append(idx); B.append(nodes)
        vo += len(V[-1]); bo += len(idx); no += len(nodes)
    V, I, B = np.concatenate(V), np.concatenate(I), np.concatenate(B)
    tl = oracle.tlas_build(inst, infos)
    return (tl, inst, infos, B, V, I)


MODE = 1          # VD_OPT_TRACE_TIGHT_TLAS value the helpers use: 1 = agglomerative top level, 2 = LBVH (set by the fixture below)


@pytest.fixture(params=[1, 2], ids=["agglomerative", "lbvh"], autouse=True)
def tight_mode(request):
    global MODE
    MODE = request.param
    yield request.param


def check_against_oracle(ctx, oracle, ctx_options, scene, rays, expect_tight=True, min_hits=1):
    """hit flags equal, distances within 1e-5, occlusion flags equal; returns (hits, bit-equal distances, accel info)."""
    import torch
    want, _ = oracle.trace(scene, rays, threads=8)
    ds = ctx.device_scene(scene)
    ctx_options("trace.tight_tlas", MODE)
    acc = ctx.trace_prepare(ds)
    ctx_options("trace.tight_tlas", 0)              # the option acts when the scene is prepared, not when it is traced
    info = acc.info()
    assert info["tight_tlas"] == (MODE if expect_tight else 0)
    n = len(rays)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(n * 16)
    d_any = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    ctx.trace_prepared_dev(acc, d_rays, n, d_hits)
    ctx.trace_any_prepared_dev(acc, d_rays, n, d_any)
    got = d_hits.cpu().numpy().view(abi.HIT)[:n]
    hit = want["hit"] == 1
    assert np.array_equal(got["hit"], want["hit"]), f"{int((got['hit'] != want['hit']).sum())} hit flags differ"
    assert hit.sum() >= min_hits
    err = np.abs(got["dist"][hit].astype(np.float64) - want["dist"][hit])
    assert np.all(err <= ABS_TOL), float(err.max())
    assert np.array_equal(d_any.cpu().numpy().astype(np.uint32), want["hit"])
    # instance ids are scene instance indices in either top level; on equal distance the walk may meet another candidate first
    same = got["dist"][hit].view(np.uint32) == want["dist"][hit].view(np.uint32)
    agree = (got["instance"][hit] == want["instance"][hit]) & (got["triangle"][hit] == want["triangle"][hit])
    assert not same.any() or agree[same].mean() > 0.999
    acc.close()
    return int(hit.sum()), int(same.sum()), info


def test_stress_shape_and_harness_shape(ctx, oracle, ctx_options):
    # the bench's stress scene in small: overlapping instances of one knot mesh (object boxes all contain the origin)
    inst = synth.instances(400, n_mesh=1, seed=synth.SEED_BASE + 8, extent=120.0, scale_range=(0.5, 2.0))
    scene = make_scene(oracle, [synth.knot_mesh(128, 32)], inst)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 90), pitch_deg=0), 192, 192)
    hits, same, info = check_against_oracle(ctx, oracle, ctx_options, scene, rays, min_hits=2000)
    assert same == hits and info["tight_fallback_instances"] == 0 and info["n_tlas_nodes"] == (801 if MODE == 1 else 800)
    # bvh_gpu.rs shape: a few meshes, many instances
    inst = synth.instances(500, n_mesh=3, seed=synth.SEED_BASE + 8, extent=60.0, scale_range=(0.5, 3.0))
    scene = make_scene(oracle, [synth.uv_sphere(1.0, 4), synth.knot_mesh(96, 24), synth.triangle_soup(64)], inst)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 45), pitch_deg=0), 256, 256)
    hits, same, _ = check_against_oracle(ctx, oracle, ctx_options, scene, rays, min_hits=1000)
    assert same == hits


def test_reference_helmet_demo_view(ctx, oracle, ctx_options):
    g = np.load(os.path.join(ROOT, "tests", "golden", "helmet.npz"))
    nodes, idx = oracle.bvh_build(g["vertices"], g["indices"])
    scene = (g["tlas"], g["instances"], g["meshes"], nodes, g["vertices"], idx)
    rays = synth.primary_rays(g["camera"], int(g["width"]), int(g["height"]))
    # one instance: nothing to cluster - the option leaves the scene's own top level in place
    check_against_oracle(ctx, oracle, ctx_options, scene, rays, expect_tight=len(g["instances"]) >= 2, min_hits=100)


def test_thirty_small_scenes(ctx, oracle, ctx_options):
    rng = np.random.default_rng(20261004)
    pool = [synth.triangle_soup(1, seed=5), synth.triangle_soup(2, seed=6), synth.triangle_soup(3, seed=7), synth.triangle_soup(10, seed=8),
            synth.uv_sphere(1.0, 3), synth.knot_mesh(24, 8), synth.plane_mesh(2.0, 2.0)]
    total = 0
    for scene_no in range(30):
        picks = rng.choice(len(pool), size=int(rng.integers(1, 4)), replace=False)
        n_inst = int(rng.choice([2, 3, 7, 40, 200]))
        inst = synth.instances(n_inst, n_mesh=len(picks), seed=synth.SEED_BASE + 300 + scene_no, extent=12.0, scale_range=(0.5, 3.0))
        scene = make_scene(oracle, [pool[m] for m in picks], inst)
        n_rays = 1500
        rays = np.zeros(n_rays, dtype=abi.RAY)
        eye = rng.normal(size=(n_rays, 3)).astype(np.float32) * rng.choice([0.5, 6.0, 25.0], size=(n_rays, 1)).astype(np.float32)
        d = rng.normal(size=(n_rays, 3)).astype(np.float32) * 5.0 - eye
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6).astype(np.float32)
        axis = rng.integers(0, n_rays, size=60)
        d[axis] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, size=60)] * rng.choice([-1.0, 1.0], size=(60, 1)).astype(np.float32)
        rays["eye"], rays["dir"] = eye, d.astype(np.float32)
        hits, same, _ = check_against_oracle(ctx, oracle, ctx_options, scene, rays, min_hits=0)
        assert same == hits, scene_no
        total += hits
    assert total > 2000


def test_grazing_rays(ctx, oracle, ctx_options):
    """Rays aimed exactly at what the tight boxes are made of and at what lies on their faces: the eight world corners of
    every instance's box, the mesh's own extreme vertices (they lie ON the box), silhouette points of the sphere, and
    rays running in a box's face planes."""
    v, i = synth.uv_sphere(1.0, 8)
    inst = synth.instances(60, n_mesh=1, seed=synth.SEED_BASE + 41, extent=14.0, scale_range=(0.6, 2.5))
    scene = make_scene(oracle, [(v, i)], inst)
    mn, mx = synth.mesh_bounds(v)
    T = inst["transform"].reshape(-1, 4, 4).astype(np.float64)            # [col][row]
    corners = np.array([[(mn, mx)[(c >> k) & 1][k] for k in range(3)] for c in range(8)], dtype=np.float64)
    world_c = np.einsum("icr,pc->ipr", T[:, :3, :3], corners) + T[:, None, 3, :3]      # instance, corner, xyz
    vv = np.asarray(v, dtype=np.float64).reshape(-1, 3)
    ext = np.concatenate([vv[vv[:, k].argmax()][None] for k in range(3)] + [vv[vv[:, k].argmin()][None] for k in range(3)])
    world_v = np.einsum("icr,pc->ipr", T[:, :3, :3], ext) + T[:, None, 3, :3]
    targets = np.concatenate([world_c.reshape(-1, 3), world_v.reshape(-1, 3)])
    eyes = np.array([[0, 0, 40.0], [35.0, 3.0, -2.0], [-20.0, 30.0, 10.0]])
    rays = []
    for e in eyes:
        d = targets - e
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = np.zeros(len(targets), dtype=abi.RAY)
        r["eye"], r["dir"] = e.astype(np.float32), d.astype(np.float32)
        rays.append(r)
    # rays inside the face planes of the world boxes: origin on the plane x = box.max.x, direction within the plane
    wmin, wmax = world_c.min(axis=1), world_c.max(axis=1)
    r = np.zeros(len(inst) * 2, dtype=abi.RAY)
    r["eye"][0::2] = np.stack([wmax[:, 0], wmin[:, 1] - 5.0, 0.5 * (wmin[:, 2] + wmax[:, 2])], axis=1).astype(np.float32)
    r["dir"][0::2] = (0, 1, 0)
    r["eye"][1::2] = np.stack([0.5 * (wmin[:, 0] + wmax[:, 0]), wmax[:, 1], wmin[:, 2] - 5.0], axis=1).astype(np.float32)
    r["dir"][1::2] = (0, 0, 1)
    rays.append(r)
    rays = np.concatenate(rays)
    check_against_oracle(ctx, oracle, ctx_options, scene, rays, min_hits=500)


def test_scenes_that_must_not_be_tightened(ctx, oracle, ctx_options):
    """inv_transform is what the walk sends the ray through, transform is what bounds the instance (tlas.rs:40): when one
    is not the inverse of the other (the reference's compute_update leaves inv_transform stale, compute_update.wgsl:10-28)
    the instance's hits lie outside its box and whether the reference finds them depends on ITS visit order; and a top
    level over non-finite boxes may have dropped clusters only the reference's build reproduces.  One such instance and
    the option declines: the prepared scene walks the scene's own top level - bit for bit the exact walk."""
    import torch
    for case in ("stale", "nonfinite"):
        inst = synth.instances(80, n_mesh=2, seed=synth.SEED_BASE + 43, extent=20.0, scale_range=(0.6, 2.5))
        if case == "stale":
            bad = np.arange(0, 80, 5)
            inst["transform"][bad, 12:15] += np.float32(3.5)               # moved, inverse not updated
        else:
            bad = np.array([7, 11])
            inst["transform"][7, 0] = np.nan
            inst["transform"][11, 13] = np.inf
        scene = make_scene(oracle, [synth.uv_sphere(1.0, 6), synth.knot_mesh(48, 12)], inst)
        rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 35), pitch_deg=0), 200, 200)
        want, _ = oracle.trace(scene, rays, threads=8)
        ds = ctx.device_scene(scene)
        ctx_options("trace.tight_tlas", MODE)
        acc = ctx.trace_prepare(ds)
        ctx_options("trace.tight_tlas", 0)
        info = acc.info()
        assert not info["tight_tlas"] and info["tight_fallback_instances"] == len(bad) and info["n_tlas_nodes"] == 161, (case, info)
        d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
        ctx.trace_prepared_dev(acc, d_rays, len(rays), d_hits)
        got = d_hits.cpu().numpy().view(abi.HIT)[: len(rays)]
        hit = want["hit"] == 1
        assert np.array_equal(got["hit"], want["hit"]) and hit.sum() > (1000 if case == "stale" else 10), case
        for f in ("dist", "instance", "triangle"):
            assert got[f][hit].tobytes() == want[f][hit].tobytes(), (case, f)
        acc.close()


def test_option_off_keeps_the_scene_top_level(ctx, oracle, ctx_options):
    inst = synth.instances(50, n_mesh=1, seed=synth.SEED_BASE + 44, extent=20.0)
    scene = make_scene(oracle, [synth.uv_sphere(1.0, 6)], inst)
    ds = ctx.device_scene(scene)
    acc = ctx.trace_prepare(ds)
    info = acc.info()
    assert not info["tight_tlas"] and info["n_tlas_nodes"] == 101 and info["tight_fallback_instances"] == 0
    acc.close()


def test_update_follows_moving_instances(ctx, oracle, ctx_options):
    """vd_trace_accel_update_dev: the instances move (compute_update with the inverse kept in step), the private top level is
    rebuilt from the instance buffer in place, the triangles are not touched; the walk equals the oracle's on the moved scene.
    Then one instance's inverse goes stale: the update declines (the walk goes back to the scene's own top level, which the
    host refits) - and takes the private one again once the inverse is repaired."""
    import torch
    inst = synth.instances(300, n_mesh=2, seed=synth.SEED_BASE + 47, extent=40.0, scale_range=(0.6, 2.0))
    scene = make_scene(oracle, [synth.uv_sphere(1.0, 6), synth.knot_mesh(48, 12)], inst)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 2.5, 60), pitch_deg=0), 160, 160)
    n = len(rays)
    ds = ctx.device_scene(scene)
    ctx_options("trace.tight_tlas", MODE)
    acc = ctx.trace_prepare(ds)
    ctx_options("trace.tight_tlas", 0)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(n * 16)

    def put(m):                                                  # the host animates the scene's instance buffer in place and refits ITS top level
        ds.tensors[1].copy_(torch.from_numpy(np.ascontiguousarray(m).view(np.uint8).reshape(-1)))
        ctx.tlas_refit_dev(ds.tensors[1], len(m), ds.tensors[2], len(scene[2]), ds.tensors[0])

    def moved(step):
        m = inst.copy()
        T = m["transform"].reshape(-1, 4, 4).astype(np.float64)
        T[:, 3, 0] += 1.5 * step * np.cos(np.arange(len(m)))
        T[:, 3, 2] += 1.0 * step * np.sin(np.arange(len(m)))
        m["transform"] = T.reshape(-1, 16).astype(np.float32)
        m["inv_transform"] = np.linalg.inv(T.transpose(0, 2, 1)).transpose(0, 2, 1).reshape(-1, 16).astype(np.float32)
        return m

    for step in (1, 2):
        m = moved(step)
        put(m)
        acc.update()
        assert acc.info()["tight_tlas"] == MODE
        want, _ = oracle.trace((oracle.tlas_build(m, scene[2]), m) + scene[2:], rays, threads=8)
        ctx.trace_prepared_dev(acc, d_rays, n, d_hits)
        got = d_hits.cpu().numpy().view(abi.HIT)[:n]
        hit = want["hit"] == 1
        assert np.array_equal(got["hit"], want["hit"]) and hit.sum() > 500
        assert np.all(np.abs(got["dist"][hit].astype(np.float64) - want["dist"][hit]) <= ABS_TOL)
    stale = moved(2)
    stale["transform"][5, 12] += np.float32(4.0)                  # moved without its inverse
    put(stale)
    acc.update()
    assert acc.info()["tight_tlas"] == 0 and acc.info()["tight_fallback_instances"] == 1
    want, _ = oracle.trace((oracle.tlas_refit(stale, scene[2], scene[0]), stale) + scene[2:], rays, threads=8)
    ctx.trace_prepared_dev(acc, d_rays, n, d_hits)
    got = d_hits.cpu().numpy().view(abi.HIT)[:n]
    assert got.tobytes() == np.ascontiguousarray(want).tobytes() or (np.array_equal(got["hit"], want["hit"]) and got["dist"][want["hit"] == 1].tobytes() == want["dist"][want["hit"] == 1].tobytes())
    put(moved(2))
    acc.update()
    assert acc.info()["tight_tlas"] == MODE
    acc.close()


def test_slightly_stale_inverse_stays_inside_its_box(ctx, oracle, ctx_options):
    """an instance whose transform was nudged by 2e-4 .. 6e-4 without refreshing inv_transform passes the
    |T * Tinv - I| <= 1e-3 qualification, but the walk sees its geometry where inv_transform puts it.  The private leaf box is
    padded by exactly that displacement (per axis, from T * Tinv - I), so the option keeps the reference walk's hit flags -
    here on a scene near the origin with small instances, where 2e-5 of the largest coordinate alone would not cover it."""
    import torch
    inst = synth.instances(200, n_mesh=2, seed=synth.SEED_BASE + 71, extent=6.0, scale_range=(0.05, 0.2))
    rng = np.random.default_rng(71)
    nudged = inst.copy()
    d = rng.uniform(2e-4, 6e-4, size=(len(inst), 3)).astype(np.float32) * rng.choice([-1.0, 1.0], size=(len(inst), 3)).astype(np.float32)
    nudged["transform"][:, 12:15] += d                           # translation moved, inverse left as it was
    nudged["transform"][::3, 0] *= np.float32(1.0 + 5e-5)       # and a third of them scaled a little along x (|T * Tinv - I| stays below 1e-3:
                                                                 # the inverse's translation column multiplies this one)
    scene = make_scene(oracle, [synth.uv_sphere(1.0, 6), synth.knot_mesh(48, 12)], nudged)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 0.5, 9), pitch_deg=0), 512, 512)
    # what must come out: the hits of the geometry where inv_transform puts it.  The reference walk on the nudged scene is not
    # that yardstick - its own leaf boxes come from `transform` too, unpadded, so rays that graze an instance within the nudge
    # are lost or found by its visit order - the reference walk on the CONSISTENT scene (transform := inv_transform^-1) is.
    consistent = nudged.copy()
    Vm = nudged["inv_transform"].reshape(-1, 4, 4).astype(np.float64).transpose(0, 2, 1)
    consistent["transform"] = np.linalg.inv(Vm).transpose(0, 2, 1).reshape(-1, 16).astype(np.float32)
    want, _ = oracle.trace((oracle.tlas_build(consistent, scene[2]), consistent) + scene[2:], rays, threads=8)
    ds = ctx.device_scene(scene)
    ctx_options("trace.tight_tlas", MODE)
    acc = ctx.trace_prepare(ds)
    ctx_options("trace.tight_tlas", 0)
    info = acc.info()
    n = len(rays)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(n * 16)
    d_any = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    ctx.trace_prepared_dev(acc, d_rays, n, d_hits)
    ctx.trace_any_prepared_dev(acc, d_rays, n, d_any)
    got = d_hits.cpu().numpy().view(abi.HIT)[:n]
    hit = want["hit"] == 1
    assert hit.sum() > 2000
    assert info["tight_tlas"] == MODE and info["tight_fallback_instances"] == 0      # they qualify: the pad has to do the work
    assert np.array_equal(got["hit"], want["hit"]), f"{int((got['hit'] != want['hit']).sum())} hit flags differ"
    assert np.array_equal(d_any.cpu().numpy().astype(np.uint32), want["hit"])
    assert np.all(np.abs(got["dist"][hit].astype(np.float64) - want["dist"][hit]) <= ABS_TOL)
    acc.close()
