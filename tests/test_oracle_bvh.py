"""Oracle pinning for BLAS / TLAS / traversal: C oracle == numpy restatement == golden fixtures,
plus the structural properties SURVEY.md §8a lists (B1-B8, T1-T4, R1-R2)."""
import numpy as np
import pytest

from conftest import fields_equal, golden
from oracle import np_restate as npr
from voidin_amd import abi, synth

BLAS = ["blas_plane.npz", "blas_sphere_1_1.npz", "blas_sphere_1_10.npz", "blas_soup64.npz", "blas_knot_2k.npz",
        "blas_plane_rot.npz",    # the X-rotated built-in plane, crates/pools/src/mesh/mod.rs:269-272
        "blas_cube_obj.npz",     # the reference's own asset assets/cube/cube.obj through ObjModel::load
        "blas_soup_nan.npz"]     # NaN vertices: f32::min/max ignore them (blas.rs:190-198), the reference builds a tree


@pytest.mark.parametrize("name", BLAS)
def test_blas_c_oracle_matches_golden(oracle, name):
    g = golden(name)
    nodes, idx = oracle.bvh_build(g["vertices"], g["indices"])
    assert fields_equal(nodes, g["nodes"]) and np.array_equal(idx, g["indices_out"])


@pytest.mark.parametrize("name", ["blas_sphere_1_1.npz", "blas_soup64.npz", "blas_soup_nan.npz"])
def test_blas_numpy_matches_golden(name):
    g = golden(name)
    nodes, idx = npr.bvh_build(g["vertices"], g["indices"])
    assert fields_equal(nodes, g["nodes"]) and np.array_equal(idx, g["indices_out"])


def check_tree(nodes, verts, idx):
    """B1/B2/B6 invariants: node 1 unused, DFS pre-order pair allocation, leaves <= 3 tris
    covering [0,T) exactly once in order, bounds = vertex bounds of the subtree."""
    tri = verts[idx.reshape(-1, 3)]
    with np.errstate(invalid="ignore"):
        tmin, tmax = np.fmin.reduce(tri, axis=1), np.fmax.reduce(tri, axis=1)      # f32::min/max ignore a NaN vertex
    seed = np.float32(1e30)                                                       # blas.rs:185-186
    assert not nodes[1:2].view(np.uint8).any()
    pool, covered = [2], [0]

    def rec(k):
        n = nodes[k]
        if n["count"] > 0:
            assert n["count"] <= 3 and n["left_first"] == covered[0]
            lo, hi = int(n["left_first"]), int(n["left_first"] + n["count"])
            covered[0] = hi
            return lo, hi
        assert n["left_first"] == pool[0]
        pool[0] += 2
        l = int(n["left_first"])
        a = rec(l)
        b = rec(l + 1)
        assert a[1] == b[0]
        for c, (lo, hi) in ((l, a), (l + 1, b)):
            assert np.array_equal(nodes[c]["min"], np.fmin(seed, np.fmin.reduce(tmin[lo:hi], axis=0)))
            assert np.array_equal(nodes[c]["max"], np.fmax(-seed, np.fmax.reduce(tmax[lo:hi], axis=0)))
        return a[0], b[1]

    import sys
    sys.setrecursionlimit(100000)
    lo, hi = rec(0)
    assert (lo, hi) == (0, len(tri)) and pool[0] == len(nodes)


@pytest.mark.parametrize("name", BLAS)
def test_blas_structure(name):
    g = golden(name)
    check_tree(g["nodes"], g["vertices"], g["indices_out"])
    # the permuted index buffer is a permutation of the input triangles
    a = np.sort(g["indices"].reshape(-1, 3).view([("", np.uint32)] * 3).ravel())
    b = np.sort(g["indices_out"].reshape(-1, 3).view([("", np.uint32)] * 3).ravel())
    assert np.array_equal(a, b)


def test_partition_shuffle_unexamined_element(oracle):
    # blas.rs:168-182: the element where i == e meet is never tested and lands on the right
    keys = np.array([0.1, 0.2, 0.3, 0.4], np.float32)
    piv, ids = oracle.partition_shuffle(keys, np.arange(4), 0, 4, 1.0)  # all "true"
    assert piv == 3 and list(ids) == [0, 1, 2, 3]
    piv, ids = oracle.partition_shuffle(keys, np.arange(4), 0, 4, 0.0)  # all "false"
    assert piv == 0


def test_blas_degenerate_input_is_an_error(oracle):
    # >= 4 triangles with identical centroids: the reference crashes
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    idx = np.tile(np.array([0, 1, 2], np.uint32), 5)
    with pytest.raises(oracle.OracleError) as e:
        oracle.bvh_build(v, idx)
    assert e.value.code == abi.VD_ERR_DEGENERATE


@pytest.mark.parametrize("n", [1, 2, 5, 40, 300, "model_scene"])
def test_tlas_c_oracle_matches_golden(oracle, n):
    g = golden(f"tlas_{n}.npz")
    nodes = oracle.tlas_build(g["instances"], g["meshes"])
    assert fields_equal(nodes, g["nodes"])
    n = len(g["instances"])
    # T2: all 2N+1 slots used; node 2N (copied to 0) merges the true root 2N-1 with itself
    assert nodes[0]["left_right"] == (2 * n - 1) | ((2 * n - 1) << 16)
    assert fields_equal(nodes[0:1], nodes[2 * n:2 * n + 1])
    # T3: refit(build(x), x) == build(x)
    assert oracle.tlas_refit(g["instances"], g["meshes"], nodes).tobytes() == nodes.tobytes()
    wide = oracle.tlas_build(g["instances"], g["meshes"], wide=True)
    assert np.array_equal(wide["left"] + (wide["right"] << 16), nodes["left_right"])
    assert np.array_equal(wide["min"], nodes["min"]) and np.array_equal(wide["max"], nodes["max"])


@pytest.mark.parametrize("name", ["tlas_40.npz", "tlas_nan_60.npz", "tlas_model_scene.npz"])
def test_tlas_numpy_matches_golden(name):
    g = golden(name)
    with np.errstate(invalid="ignore", over="ignore"):
        assert fields_equal(npr.tlas_nodes(g["instances"], g["meshes"]), g["nodes"])


def test_min_max_are_rusts(oracle):
    """Rust's f32::min / f32::max - what glam 0.24's scalar Vec3::min/max call (blas.rs:190-198, tlas.rs:43,69-70,96-97) -
    return the OTHER operand when one is a NaN, so a NaN never enters a box: a mesh with NaN vertices builds (the NaN
    centroid fails every `<` and goes right), a NaN / inf - inf transform leaves the seeded mesh box plus the finite
    corners.  Both restatements agree on the fixtures (make_golden.py asserts it); here: the C oracle reproduces them
    and no box of either holds a NaN."""
    g = golden("blas_soup_nan.npz")
    assert np.isnan(g["vertices"]).sum() >= 5
    nodes, idx = oracle.bvh_build(g["vertices"], g["indices"])
    assert fields_equal(nodes, g["nodes"]) and np.array_equal(idx, g["indices_out"])
    assert not np.isnan(nodes["min"]).any() and not np.isnan(nodes["max"]).any()
    # the triangle whose x is NaN in all three vertices keeps the +-1e30 seeds in x wherever it sits alone
    t = golden("tlas_nan_60.npz")
    tn = oracle.tlas_build(t["instances"], t["meshes"])
    assert fields_equal(tn, t["nodes"])
    assert not np.isnan(tn["min"]).any() and not np.isnan(tn["max"]).any()
    m = t["meshes"][t["instances"]["mesh"][7]]
    # instance 7 (NaN translation x): every corner's x is NaN -> the leaf's x range is the object-space seed (tlas.rs:39)
    assert tn["min"][8][0] == m["min"][0] and tn["max"][8][0] == m["max"][0]
    assert np.isinf(tn["min"]).any()          # the infinite (not NaN) boxes are kept as they are
    assert oracle.tlas_refit(t["instances"], t["meshes"], tn).tobytes() == tn.tobytes()


def test_tlas_leaf_seeded_with_object_space_box():
    # tlas.rs:39: the fold is seeded with [mesh.min, mesh.max] (object space) — bug-compatible
    g = golden("tlas_5.npz")
    inst, meshes, nodes = g["instances"], g["meshes"], g["nodes"]
    for i in range(5):
        m = meshes[inst["mesh"][i]]
        assert (nodes["min"][i + 1] <= m["min"]).all() and (nodes["max"][i + 1] >= m["max"]).all()


def test_tlas_overflow_guard(oracle):
    # tlas.rs:71 packs 16-bit ids: n > 32768 cannot be represented
    inst = np.zeros(abi.TLAS_MAX_INSTANCES + 1, abi.INSTANCE)
    with pytest.raises(oracle.OracleError) as e:
        oracle.tlas_build(inst, synth.mesh_infos())
    assert e.value.code == abi.VD_ERR_TLAS_OVERFLOW


def test_trace_c_oracle_matches_golden(oracle):
    g = golden("trace_40.npz")
    h, max_stack = oracle.trace((g["tlas"], g["instances"], g["meshes"], g["bvh_nodes"], g["vertices"], g["indices"]), g["rays"])
    assert np.array_equal(h["hit"], g["hit"]) and np.array_equal(h["dist"], g["dist"])
    assert max_stack <= 24  # the reference's unchecked stack (stack.wgsl:1) suffices here
    assert (h["dist"][h["hit"] == 0] == np.float32(1e30)).all()


def test_trace_brute_force_agrees(oracle):
    """Closest hit over ALL triangles (no BVH) equals the traversal result: no truly-hit leaf
    is pruned."""
    g = golden("trace_40.npz")
    inst, meshes, V, I = g["instances"], g["meshes"], g["vertices"].reshape(-1, 3), g["indices"]
    rays = g["rays"][::7]
    want = g["dist"][::7]
    best = np.full(len(rays), 1e30)
    for i in range(len(inst)):
        m = meshes[inst["mesh"][i]]
        M = inst["inv_transform"][i].reshape(4, 4).astype(np.float64).T
        tri = V[int(m["vertex_offset"]) + I[int(m["base_index"]):int(m["base_index"]) + int(m["index_count"])].reshape(-1, 3)].astype(np.float64)
        e = (M @ np.concatenate([rays["eye"].astype(np.float64), np.ones((len(rays), 1))], 1).T).T[:, :3]
        d = (M @ np.concatenate([rays["dir"].astype(np.float64), np.zeros((len(rays), 1))], 1).T).T[:, :3]
        e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
        for r in range(len(rays)):
            u_ = np.cross(d[r], e2)
            det = (e1 * u_).sum(1)
            ok = det >= 1e-10
            inv = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
            o = e[r] - tri[:, 0]
            u = inv * (o * u_).sum(1)
            vv = np.cross(o, e1)
            v = inv * (d[r] * vv).sum(1)
            t = inv * (e2 * vv).sum(1)
            ok &= (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0)
            if ok.any():
                best[r] = min(best[r], t[ok].min())
    hit = want < 1e29
    assert np.array_equal(hit, best < 1e29) or (np.abs(best[hit] - want[hit]) / want[hit]).max() < 1e-4
    assert np.allclose(best[hit], want[hit], rtol=1e-4)


def test_primary_rays_follow_the_cpu_harness(oracle):
    """vd_ref_primary_rays (bvh_cpu.rs:71-83) against the float64 numpy restatement: fp32 rounding apart."""
    cam = synth.camera_uniform(eye=(0, 0, 15), pitch_deg=0)
    for w, h in ((48, 48), (64, 64), (1, 1)):
        got = oracle.primary_rays(cam, w, h)
        want = synth.primary_rays(cam, w, h)
        assert np.abs(got["eye"] - want["eye"]).max() < 1e-5 and np.abs(got["dir"] - want["dir"]).max() < 1e-6
        assert np.abs(np.linalg.norm(got["dir"].astype(np.float64), axis=1) - 1).max() < 1e-6
        assert not got["_pad0"].any() and not got["_pad1"].any()
    # centre pixel of an even grid looks straight down -z from the eye (camera yaw 0 pitch 0)
    r = oracle.primary_rays(cam, 64, 64)[32 * 64 + 32]
    assert np.allclose(r["dir"], (0, 0, -1), atol=1e-6) and np.allclose(r["eye"][:2], (0, 0), atol=1e-4)
    assert len(oracle.primary_rays(cam, 0, 0)) == 0


def test_cpu_harness_matches_golden(oracle):
    """bvh_cpu.rs per-pixel rays + Bvh::traverse_iter: the C oracle against the fixture written from the independent
    numpy restatement (tests/golden/make_golden.py: harness_case)."""
    g = golden("harness_soup64.npz")
    rays = oracle.primary_rays(g["camera"], int(g["width"]), int(g["height"]))
    assert rays.tobytes() == g["rays"].tobytes()
    d = oracle.traverse_iter(g["nodes"], g["vertices"], g["indices"], g["rays"])
    assert d.tobytes() == g["dist"].tobytes()


def test_rust_cpu_traversal_variant(oracle):
    # R2 (blas.rs:247-295): two-sided, divides by dir; agrees with R1 on front-facing hits
    g = golden("blas_soup64.npz")
    cam = synth.camera_uniform(eye=(0, 0, 15), pitch_deg=0)  # bvh_cpu.rs:134
    rays = synth.primary_rays(cam, 48, 48)
    d = oracle.traverse_iter(g["nodes"], g["vertices"], g["indices_out"], rays)
    assert (d >= 0).sum() > 20 and d[d >= 0].min() > 5.0


def test_reference_cube_asset_through_both_obj_readers():
    """tests/golden/cube.obj = the reference's assets/cube/cube.obj (216 v, 218 mixed quad + triangle faces): the Python
    and the C++ ObjModel::load agree, the counts match an independent parse, and the fixture's inputs are what they read."""
    import os
    from conftest import GOLDEN
    from voidin_amd.obj import ObjModel
    path = os.path.join(GOLDEN, "cube.obj")
    (m,) = ObjModel.load(path)
    faces = [l.split()[1:] for l in open(path) if l.startswith("f ")]
    assert len(faces) == 218 and sum(1 for l in open(path) if l.startswith("v ")) == 216
    assert len(m.indices) // 3 == sum(len(f) - 2 for f in faces) == 428          # fan triangulation of quads
    assert len(m.positions) == len({t for f in faces for t in f}) == 277          # single_index: one vertex per distinct v/vt/vn
    v, i = m.arrays()
    g = golden("blas_cube_obj.npz")
    assert v.tobytes() == g["vertices"].tobytes() and np.array_equal(i, g["indices"])
    assert m.name == "Cube_Finished_Cube.001" and m.material_id == 0 and len(m.normals) == len(m.texcoords) // 2 == 277


def test_builtin_pool_bookkeeping(oracle):
    """MeshPool::new's four built-in meshes (mesh/mod.rs:266-274) through MeshPool::add's bookkeeping (mesh/mod.rs:309-351):
    the fixture's MeshInfo offsets are running sums and its BLAS blocks are the oracle's."""
    g = golden("pool_builtin.npz")
    meshes = [synth.plane_mesh(), synth.plane_mesh_rot_x(), synth.uv_sphere(1.0, 1), synth.uv_sphere(1.0, 10)]
    vo = bo = no = 0
    for k, (v, i) in enumerate(meshes):
        nodes, idx = oracle.bvh_build(v, i)
        info = g["meshes"][k]
        assert (info["vertex_offset"], info["base_index"], info["bvh_index"], info["index_count"]) == (vo, bo, no, len(i))
        assert fields_equal(g["bvh_nodes"][no: no + len(nodes)], nodes) and np.array_equal(g["indices"][bo: bo + len(idx)], idx)
        assert np.array_equal(info["min"], v.min(axis=0)) and np.array_equal(info["max"], v.max(axis=0))
        vo += len(v); bo += len(idx); no += len(nodes)
    # the rotated plane stands up: y spans [-0.5, 0.5], z is +-2.2e-8 (cos(-PI/2 as f32) != 0)
    r = synth.plane_mesh_rot_x()[0]
    assert np.array_equal(np.abs(r[:, 1]), np.full(4, 0.5, np.float32)) and np.all(np.abs(r[:, 2]) == np.float32(2.1855694e-08))


def test_recursive_traverse_agrees_with_the_iterative_walk(oracle):
    """R3 (blas.rs:211-245), dead code in the reference: restated for completeness.  Where a triangle is hit it reports
    the same distance as Bvh::traverse_iter (both take the minimum over the triangles their box tests let through, and a
    box test only ever prunes with the current best); where none is hit it returns Hit(t0) if the root box is entered
    (the quirk of blas.rs:244) and Miss otherwise."""
    g = golden("harness_soup64.npz")
    it = oracle.traverse_iter(g["nodes"], g["vertices"], g["indices"], g["rays"])
    rec = oracle.traverse_recursive(g["nodes"], g["vertices"], g["indices"], g["rays"], 1e30)
    hit = it >= 0
    assert hit.sum() > 20 and np.array_equal(rec[hit], it[hit])
    assert set(np.unique(rec[~hit])) <= {np.float32(-1.0), np.float32(1e30)}
    # the root box test decides between Miss and Hit(t0)
    root_min, root_max = g["nodes"]["min"][0], g["nodes"]["max"][0]
    o, d = g["rays"]["eye"][~hit].astype(np.float32), g["rays"]["dir"][~hit].astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1, t2 = (root_min - o) / d, (root_max - o) / d
    tmax, tmin = np.maximum(t1, t2).min(axis=1), np.minimum(t1, t2).max(axis=1)
    enters = (tmax >= tmin) & (tmin < np.float32(1e30)) & (tmax > 0)
    assert np.array_equal(rec[~hit] == np.float32(1e30), enters)
    v, i = synth.knot_mesh(48, 12)
    nodes, idx = oracle.bvh_build(v, i)
    rays = synth.primary_rays(synth.camera_uniform(eye=(0, 0, 6), pitch_deg=0), 40, 40)
    it, rec = oracle.traverse_iter(nodes, v, idx, rays), oracle.traverse_recursive(nodes, v, idx, rays)
    assert (it >= 0).sum() > 100 and np.array_equal(rec[it >= 0], it[it >= 0])
