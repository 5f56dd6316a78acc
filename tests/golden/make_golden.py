"""Write the golden vectors under tests/golden/ (run HERE, in the build container).

The reference (Rust + WGSL) cannot be built or imported in this image, so the vectors are
produced by the independent numpy restatement (oracle/np_restate.py) and must equal what the C
oracle produces — the script asserts that before writing.  Each .npz holds inputs and expected
outputs only (no reference source text).  Parity stays "unpinned" in the sense of SURVEY.md §8c:
these fixtures freeze OUR restatement, they are not outputs of the reference binary.

Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_restate as npr  # noqa: E402
from oracle import ref  # noqa: E402
from voidin_amd import abi, synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def same(a, b):
    return all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


def cull_cases():
    meshes = synth.mesh_infos()
    cams = {
        "model": synth.camera_uniform(),                                        # src/bin/model.rs:235
        "jitter": synth.camera_uniform(eye=(-30, 12, 7), yaw_deg=135, pitch_deg=10, jitter=(0.003, -0.002)),
    }
    for name, cam in cams.items():
        for tag, kw in {"wide": dict(scale_range=(0.25, 4.0)), "small": dict(scale_range=(0.01, 0.3), extent=400.0)}.items():
            inst = synth.instances(1000, seed=synth.SEED_BASE + 1, **kw)          # BASELINE config 1 size
            d_np = npr.cull_emit(cam, meshes, inst)
            d_c = ref.cull_emit(cam, meshes, inst)
            assert d_np.tobytes() == d_c.tobytes(), (name, tag)
            comp, cnt = npr.compact(d_np)
            c2, cnt2 = ref.compact(d_c)
            assert cnt == cnt2 and comp.tobytes() == c2[:cnt2].tobytes()
            np.savez_compressed(os.path.join(OUT, f"cull_{name}_{tag}.npz"), camera=cam, meshes=meshes,
                                instances=inst, draws=d_np, compact=comp, count=np.uint32(cnt))
            print("cull", name, tag, "visible", cnt, "/ 1000")


def cube_obj():
    """tests/golden/cube.obj is a byte copy of /root/reference/assets/cube/cube.obj - the only OBJ asset the reference
    still ships (216 v / 218 mixed quad + triangle faces): ObjModel::load's fan triangulation + single-index vertices."""
    from voidin_amd.obj import ObjModel
    src = "/root/reference/assets/cube/cube.obj"
    if os.path.exists(src):
        assert open(src, "rb").read() == open(os.path.join(OUT, "cube.obj"), "rb").read(), "fixture differs from the reference asset"
    (m,) = ObjModel.load(os.path.join(OUT, "cube.obj"))
    return m.arrays()


def builtin_pool_case():
    """MeshPool::new (mesh/mod.rs:266-274): plane, rotated plane, uv_sphere(1, 1), uv_sphere(1, 10) added in that order;
    MeshPool::add (mesh/mod.rs:309-351) bookkeeping: vertex_offset / base_index / bvh_index are running sums."""
    infos, V, I, B = pool([synth.plane_mesh(), synth.plane_mesh_rot_x(), synth.uv_sphere(1.0, 1), synth.uv_sphere(1.0, 10)])
    np.savez_compressed(os.path.join(OUT, "pool_builtin.npz"), meshes=infos, vertices=V, indices=I, bvh_nodes=B)
    print("builtin pool", infos["bvh_index"], infos["base_index"], infos["vertex_offset"])


def blas_cases():
    cases = {
        "plane": synth.plane_mesh(),                   # crates/pools/src/mesh/plane.rs:5-38
        "plane_rot": synth.plane_mesh_rot_x(),         # mesh/mod.rs:269-272: the plane rotated by Mat3::from_rotation_x(-PI/2)
        "cube_obj": cube_obj(),                        # the reference's own asset, assets/cube/cube.obj (copied: data, not source)
        "sphere_1_1": synth.uv_sphere(1.0, 1),         # mesh/mod.rs:273
        "sphere_1_10": synth.uv_sphere(1.0, 10),       # mesh/mod.rs:274 (6320 tris)
        "soup64": synth.triangle_soup(64),             # src/bin/bvh_cpu.rs:39-52 distribution
        "knot_2k": synth.knot_mesh(64, 16),
    }
    for name, (v, i) in cases.items():
        n_np, i_np = npr.bvh_build(v, i)
        n_c, i_c = ref.bvh_build(v, i)
        assert same(n_np, n_c) and np.array_equal(i_np, i_c), name
        np.savez_compressed(os.path.join(OUT, f"blas_{name}.npz"), vertices=v, indices=i, nodes=n_np,
                            indices_out=i_np)
        print("blas", name, "tris", len(i) // 3, "nodes", len(n_np))


def nan_soup(n_tri=200):
    """A soup with NaN vertices: Rust's f32::min/max (glam scalar Vec3::min/max, blas.rs:190-198) IGNORE a NaN operand,
    so the reference builds a tree - the NaN vertex drops out of every box, its triangle's centroid fails every `<`
    (blas.rs:173) and goes right.  One vertex has a single NaN coordinate, one triangle is NaN in all three vertices."""
    v, i = synth.triangle_soup(n_tri, seed=synth.SEED_BASE + 77)
    v = v.copy()
    v[5, 1] = np.nan                       # one coordinate of one vertex
    v[3 * 40: 3 * 40 + 3, 0] = np.nan      # triangle 40: x of all three vertices -> its x box stays at the +-1e30 seeds
    v[3 * 90 + 1] = np.nan                 # a whole vertex
    return v, i


def nan_tlas_instances(n=60):
    """Transforms that poison leaf corners: a NaN entry, and inf / -inf columns whose products cancel to a NaN the
    arithmetic GENERATES (inf - inf: 0xFFC00000 on x86, 0x7FC00000 on gfx950 - the sign must not matter).  The fold of
    tlas.rs:39-44 starts from the mesh box and f32::min/max ignore the NaN corners."""
    inst = synth.instances(n, n_mesh=3, seed=synth.SEED_BASE + 31, extent=30.0, scale_range=(0.5, 3.0))
    inst["transform"][7, 12] = np.nan                      # translation x
    inst["transform"][19, 5] = np.nan                      # a rotation / scale entry
    inst["transform"][23, 0] = np.inf; inst["transform"][23, 4] = -np.inf    # X.x = inf, Y.x = -inf: inf*px + (-inf)*py
    inst["transform"][41, 13] = -np.inf                    # translation y = -inf: an infinite (not NaN) leaf box
    inst["transform"][52, 2] = np.inf; inst["transform"][52, 10] = np.inf; inst["transform"][52, 6] = -np.inf
    return inst


def nan_cases():
    v, i = nan_soup()
    n_np, i_np = npr.bvh_build(v, i)
    n_c, i_c = ref.bvh_build(v, i)
    assert same_bits(n_np, n_c) and np.array_equal(i_np, i_c)
    assert not np.isnan(n_np["min"]).any() and not np.isnan(n_np["max"]).any()
    np.savez_compressed(os.path.join(OUT, "blas_soup_nan.npz"), vertices=v, indices=i, nodes=n_np, indices_out=i_np)
    print("blas soup_nan tris", len(i) // 3, "nodes", len(n_np))
    infos, V, I, B = pool([synth.uv_sphere(1.0, 2), synth.knot_mesh(32, 8), synth.triangle_soup(64)])
    inst = nan_tlas_instances()
    with np.errstate(invalid="ignore", over="ignore"):
        t_np = npr.tlas_nodes(inst, infos)
    t_c = ref.tlas_build(inst, infos)
    assert same_bits(t_np, t_c)
    assert not np.isnan(t_np["min"]).any() and not np.isnan(t_np["max"]).any()
    np.savez_compressed(os.path.join(OUT, "tlas_nan_60.npz"), instances=inst, meshes=infos, nodes=t_np)
    print("tlas nan_60")


def same_bits(a, b):
    return all(np.ascontiguousarray(a[f]).view(np.uint8).tobytes() == np.ascontiguousarray(b[f]).view(np.uint8).tobytes() for f in a.dtype.names)


def pool(mesh_list):
    V, I, B = [], [], []
    infos = np.zeros(len(mesh_list), dtype=abi.MESH_INFO)
    vo = bo = no = 0
    for k, (v, i) in enumerate(mesh_list):
        nodes, idx = npr.bvh_build(v, i)
        mn, mx = synth.mesh_bounds(v)
        infos[k]["min"], infos[k]["max"] = mn, mx
        infos[k]["index_count"], infos[k]["base_index"] = len(idx), bo
        infos[k]["vertex_offset"], infos[k]["bvh_index"] = vo, no
        V.append(v); I.append(idx); B.append(nodes)
        vo += len(v); bo += len(idx); no += len(nodes)
    return infos, np.concatenate(V), np.concatenate(I), np.concatenate(B)


def tlas_trace_cases():
    infos, V, I, B = pool([synth.uv_sphere(1.0, 2), synth.knot_mesh(32, 8), synth.triangle_soup(64)])
    for n in (1, 2, 5, 40, 300):
        inst = synth.instances(n, n_mesh=3, seed=synth.SEED_BASE + 5, extent=30.0, scale_range=(0.5, 3.0))
        t_np = npr.tlas_nodes(inst, infos)
        t_c = ref.tlas_build(inst, infos)
        assert same(t_np, t_c), n
        np.savez_compressed(os.path.join(OUT, f"tlas_{n}.npz"), instances=inst, meshes=infos, nodes=t_np)
        print("tlas", n)
    inst = synth.instances(40, n_mesh=3, seed=synth.SEED_BASE + 5, extent=30.0, scale_range=(0.5, 3.0))
    tl = npr.tlas_nodes(inst, infos)
    cam = synth.camera_uniform(eye=(0, 2.5, 30), pitch_deg=0)   # src/bin/bvh_gpu.rs:221 shape
    rays = synth.primary_rays(cam, 32, 32)
    h_np = npr.trace((tl, inst, infos, B, V, I), rays)
    h_c, _ = ref.trace((tl, inst, infos, B, V, I), rays)
    assert np.array_equal(h_np["hit"], h_c["hit"]) and np.array_equal(h_np["dist"], h_c["dist"])
    np.savez_compressed(os.path.join(OUT, "trace_40.npz"), tlas=tl, instances=inst, meshes=infos, bvh_nodes=B,
                        vertices=V, indices=I, rays=rays, dist=h_np["dist"], hit=h_np["hit"])
    print("trace hits", int(h_np["hit"].sum()), "/", len(rays))


def harness_case():
    """src/bin/bvh_cpu.rs: camera (0, 0, 15), per-pixel rays, Bvh::traverse_iter over the 64-triangle soup."""
    cam = synth.camera_uniform(eye=(0, 0, 15), pitch_deg=0)                 # bvh_cpu.rs:134
    v, i = synth.triangle_soup(64)
    nodes, idx = npr.bvh_build(v, i)
    r_np, r_c = npr.primary_rays(cam, 48, 48), ref.primary_rays(cam, 48, 48)
    assert r_np.tobytes() == r_c.tobytes()
    d_np, d_c = npr.traverse_iter(nodes, v, idx, r_np), ref.traverse_iter(nodes, v, idx, r_c)
    assert d_np.tobytes() == d_c.tobytes() and (d_np >= 0).sum() > 20
    np.savez_compressed(os.path.join(OUT, "harness_soup64.npz"), camera=cam, width=np.uint32(48), height=np.uint32(48), rays=r_np,
                        nodes=nodes, vertices=v, indices=idx, dist=d_np)
    print("harness hits", int((d_np >= 0).sum()), "/", len(r_np))


def occlusion_case():
    """The occlusion extension (no reference counterpart): depth -> min pyramid -> refined mask."""
    cam = synth.camera_uniform(eye=(0, 0, 50), pitch_deg=0, jitter=(0.003, -0.002))
    meshes = synth.mesh_infos()
    inst = synth.instances(1500, seed=synth.SEED_BASE + 42, extent=300.0, centre=(0.0, 0.0, -150.0), scale_range=(0.25, 4.0))
    w, h = 100, 60
    u = synth.uniform01(synth.SEED_BASE + 43, 0, 64 * 5).reshape(64, 5).astype(np.float64)
    depth = np.zeros((h, w), dtype=np.float32)
    for x, y, sx, sy, z in u:
        x0, y0 = int(x * w), int(y * h)
        depth[y0: y0 + 1 + int(sy * h / 3), x0: x0 + 1 + int(sx * w / 3)] = np.float32(0.001 / (20.0 + 300.0 * z))
    p_np, _ = npr.hiz_build(depth)
    p_c = ref.hiz_build(depth)
    assert p_np.tobytes() == p_c.tobytes()
    frustum = np.zeros((len(inst) + 63) // 64, dtype=np.uint64)
    vis = ref.cull_emit(cam, meshes, inst)["instance_count"] == 1
    padded = np.zeros(len(frustum) * 64, dtype=np.uint8)
    padded[: len(inst)] = vis
    frustum = np.packbits(padded, bitorder="little").view(np.uint64)
    m_np = npr.occlusion_mask(cam, meshes, inst, depth, frustum)
    m_c = ref.occlusion_mask(cam, meshes, inst, p_c, w, h, frustum)
    assert np.array_equal(m_np, m_c)
    np.savez_compressed(os.path.join(OUT, "occlusion_1500.npz"), camera=cam, meshes=meshes, instances=inst, depth=depth, pyramid=p_np,
                        mask_in=frustum, mask_out=m_np)
    pop = lambda m: int(np.unpackbits(m.view(np.uint8)).sum())
    print("occlusion: frustum", pop(frustum), "-> kept", pop(m_np))


def helmet_case():
    """The one real mesh the reference checkout carries: assets/glTF-Sample-Models/2.0/DamagedHelmet/glTF-Binary/
    DamagedHelmet.glb (14 556 vertices, 15 452 triangles), which the default demo loads and places at
    translation(0, 0, 9) * scale(3) (src/bin/model.rs:100-106; camera at (2, 5, 12), pitch -20 deg: model.rs:235).
    The fixture holds what GltfDocument::import hands to MeshPool::add - the POSITION accessor's Vec3s and the indices as
    u32 (gltf_model/mod.rs:118-150) - checksums of the C oracle's BLAS over it (15 k triangles are too many for the numpy
    restatement's Python loops: this one fixture freezes the C oracle alone), the instance, its one-leaf TLAS and the
    oracle's hits for 160 x 160 primary rays from that camera.  No reference file is copied: geometry arrays only."""
    import zlib
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from gltf_reader import GltfDocument
    src = "/root/reference/assets/glTF-Sample-Models/2.0/DamagedHelmet/glTF-Binary/DamagedHelmet.glb"
    if not os.path.exists(src):
        print("helmet_case: reference asset not present, fixture left as it is")
        return
    doc = GltfDocument.load(src)
    (prim,) = doc.primitives()
    v, i = prim.arrays()
    assert v.shape == (14556, 3) and i.shape == (46356,)
    nodes, idx = ref.bvh_build(v, i)
    T = np.eye(4); T[:3, 3] = (0.0, 0.0, 9.0)
    ((M, _, _),) = doc.scene_instances(T @ np.diag([3.0, 3.0, 3.0, 1.0]))
    inst = synth.instance_from_matrix(M.T.reshape(16), mesh=0).reshape(1)
    infos = np.zeros(1, dtype=abi.MESH_INFO)
    infos[0]["min"], infos[0]["max"] = synth.mesh_bounds(v)
    infos[0]["index_count"] = len(idx)
    tl = ref.tlas_build(inst, infos)
    cam = synth.camera_uniform(eye=(2.0, 5.0, 12.0), pitch_deg=-20.0)
    rays = synth.primary_rays(cam, 160, 160)
    hits, _ = ref.trace((tl, inst, infos, nodes, v, idx), rays, threads=8)
    assert 2000 < hits["hit"].sum() < len(rays)
    np.savez_compressed(os.path.join(OUT, "helmet.npz"), vertices=v, indices=i, n_nodes=np.uint32(len(nodes)),
                        nodes_crc=np.uint32(zlib.crc32(nodes.tobytes())), indices_out_crc=np.uint32(zlib.crc32(idx.tobytes())),
                        instances=inst, meshes=infos, tlas=tl, camera=cam, width=np.uint32(160), height=np.uint32(160),
                        hit=hits["hit"].astype(np.uint8), dist=hits["dist"], triangle=hits["triangle"])
    print("helmet: nodes", len(nodes), "hits", int(hits["hit"].sum()))



# ---- the reference's own demo scene, for the cull (src/bin/model.rs) -----------------------------------------------
def _f32(x):
    return np.float32(x)


def _mul(A, B):
    """glam 0.24 Mat4 * Mat4 in f32: column j of the product = ((A.x*b.x + A.y*b.y) + A.z*b.z) + A.w*b.w (mul_vec4 per column).
    Matrices here are (4, 4) arrays of COLUMNS: M[j] = column j."""
    A, B = np.asarray(A, np.float32), np.asarray(B, np.float32)
    out = np.zeros((4, 4), np.float32)
    for j in range(4):
        out[j] = ((A[0] * B[j][0] + A[1] * B[j][1]) + A[2] * B[j][2]) + A[3] * B[j][3]
    return out


def _translation(x, y, z):
    M = np.eye(4, dtype=np.float32); M[3][:3] = (_f32(x), _f32(y), _f32(z)); return M


def _scale(x, y, z):
    return np.diag(np.array([x, y, z, 1.0], np.float32)).astype(np.float32)


def _rot(axis, angle):
    a = _f32(angle)
    s, c = np.sin(a, dtype=np.float32), np.cos(a, dtype=np.float32)
    M = np.eye(4, dtype=np.float32)
    if axis == "x": M[1][:3] = (0, c, s); M[2][:3] = (0, -s, c)
    if axis == "y": M[0][:3] = (c, 0, -s); M[2][:3] = (s, 0, c)
    if axis == "z": M[0][:3] = (c, s, 0); M[1][:3] = (-s, c, 0)
    return M


def _node_matrix_f32(n):
    """gltf Node::transform().matrix() of a decomposed node, in f32: T * R * S."""
    if "matrix" in n:
        return np.asarray(n["matrix"], np.float32).reshape(4, 4)
    x, y, z, w = (np.float32(v) for v in n.get("rotation", (0, 0, 0, 1)))
    one, two = np.float32(1), np.float32(2)
    R = np.eye(4, dtype=np.float32)
    R[0][:3] = (one - two * (y * y + z * z), two * (x * y + z * w), two * (x * z - y * w))
    R[1][:3] = (two * (x * y - z * w), one - two * (x * x + z * z), two * (y * z + x * w))
    R[2][:3] = (two * (x * z + y * w), two * (y * z - x * w), one - two * (x * x + y * y))
    t, sc = n.get("translation", (0, 0, 0)), n.get("scale", (1, 1, 1))
    return _mul(_mul(_translation(*t), R), _scale(*sc))


def model_scene_cases():
    """The ONE place the reference runs emit_draws on a real scene: src/bin/model.rs:62-147 under the camera of :235.
    What the cull reads of a mesh is MeshInfo.{min, max, index_count, base_index, vertex_offset} (emit_draws.wgsl:13-63),
    and MeshPool::add (crates/pools/src/mesh/mod.rs:309-351) fills those from the POSITION bounds, the index count and the
    running sums - all of which a glTF document states in its JSON (accessor min / max are mandatory for POSITION, exact f32).
    So the table is restated for every mesh the demo loads whose document is in the checkout:
      ids 0..3    MeshPool::new's four built-in meshes (mesh/mod.rs:266-274)
      ids 4..106  Sponza's 103 primitives (Sponza.gltf is there; Sponza.bin - the triangles - is a missing blob, and not needed)
      id  107     DamagedHelmet.glb
      id  108     make_uv_sphere(1, 10), added again by model.rs:119-120
    assets/ferris3d_v1.0.glb is a missing blob (.MISSING_LARGE_BLOBS): its meshes and the instances model.rs makes of them
    (:108-117, :137-141) are left out, so ids after 107 are one document short of the reference's.  bvh_index is 0 throughout:
    emit_draws never reads it, and Sponza's node counts cannot be known without its triangles.
    Instances, in the reference's order: the two area-light planes (app.rs:220-235), Sponza's node walk under
    rot_y(pi/2) * T(7,-5,1) * S(3), the helmet under T(0,0,9) * S(3), the ten spheres of the moving ring (:122-134; their
    materials are thread_rng there - irrelevant to the cull - and 1 here).  Matrix products follow glam's f32 evaluation order.
    Three fixtures: the demo camera (everything passes: with the reference's object-space radius - SURVEY 8a C2 - Sponza's
    primitives, hundreds of units large before the node's 0.008 scale, are never culled from there); a camera at the end of
    the nave that leaves about half the scene behind it; and the scene three times over (327 meshes: past the 256 rows an
    8-bit mesh id can name, 348 instances) from inside, half of it culled."""
    import json
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from gltf_reader import GltfDocument
    base = "/root/reference/assets/glTF-Sample-Models/2.0/"
    sponza_p, helmet_p = base + "Sponza/glTF/Sponza.gltf", base + "DamagedHelmet/glTF-Binary/DamagedHelmet.glb"
    if not (os.path.exists(sponza_p) and os.path.exists(helmet_p)):
        print("model_scene_cases: reference assets not present, fixtures left as they are")
        return
    rows = []                                            # (min, max, index_count, vertex_count)
    for v, i in (synth.plane_mesh(), synth.plane_mesh_rot_x(), synth.uv_sphere(1.0, 1), synth.uv_sphere(1.0, 10)):
        mn, mx = synth.mesh_bounds(v)
        rows.append((mn, mx, len(i), len(v)))
    sponza = json.load(open(sponza_p))
    acc = sponza["accessors"]
    sponza_prims = []
    for mi, m in enumerate(sponza["meshes"]):
        for pi, pr in enumerate(m["primitives"]):
            at = pr["attributes"]
            if "POSITION" not in at or "NORMAL" not in at:                      # gltf_model/mod.rs:118-123
                continue
            a = acc[at["POSITION"]]
            n_idx = acc[pr["indices"]]["count"] if "indices" in pr else a["count"]
            rows.append((np.asarray(a["min"], np.float32), np.asarray(a["max"], np.float32), n_idx, a["count"]))
            sponza_prims.append((mi, pi))
    assert len(sponza_prims) == 103
    helmet = GltfDocument.load(helmet_p)
    (hp,) = helmet.primitives()
    mn, mx = synth.mesh_bounds(hp.positions)
    rows.append((mn, mx, len(hp.indices), len(hp.positions)))
    sv, si = synth.uv_sphere(1.0, 10)
    mn, mx = synth.mesh_bounds(sv)
    rows.append((mn, mx, len(si), len(sv)))
    meshes = np.zeros(len(rows), dtype=abi.MESH_INFO)
    base_index = vertex_offset = 0
    for k, (mn, mx, n_idx, n_v) in enumerate(rows):                             # mesh/mod.rs:310-345
        meshes[k]["min"], meshes[k]["max"] = mn, mx
        meshes[k]["index_count"], meshes[k]["base_index"], meshes[k]["vertex_offset"] = n_idx, base_index, vertex_offset
        base_index += n_idx; vertex_offset += n_v
    SPONZA0, HELMET, SPHERE = 4, 4 + 103, 4 + 103 + 1

    inst = []
    PI = np.float32(np.pi)

    def add(M, mesh):
        inst.append(synth.instance_from_matrix(np.asarray(M, np.float32).reshape(16), mesh=mesh))

    for T in (_mul(_translation(0, 10, 15), _rot("x", -PI / _f32(4))), _mul(_translation(0, 10, -25), _rot("x", _f32(-3) * PI / _f32(4)))):
        add(_mul(T, _scale(2.5, 4.0, 1.0)), 1)                                   # app.rs:230-234: transform * scale((wh / 2).extend(1)), VERTICAL_PLANE_MESH

    def walk(doc, ni, parent, mesh_of):
        n = doc["nodes"][ni]
        M = _mul(parent, _node_matrix_f32(n))
        for c in n.get("children", []):                                         # gltf_model/mod.rs:189-191: children first
            walk(doc, c, M, mesh_of)
        if "mesh" in n:
            for pi in range(len(doc["meshes"][n["mesh"]]["primitives"])):
                if (n["mesh"], pi) in mesh_of:
                    add(M, mesh_of[(n["mesh"], pi)])

    root = _mul(_mul(_rot("y", PI / _f32(2)), _translation(7, -5, 1)), _scale(3, 3, 3))          # model.rs:94-98
    for ni in sponza["scenes"][0]["nodes"]:
        walk(sponza, ni, root, {k: SPONZA0 + j for j, k in enumerate(sponza_prims)})
    root = _mul(_translation(0, 0, 9), _scale(3, 3, 3))                                              # model.rs:104-106
    for ni in helmet.doc["scenes"][0]["nodes"]:
        walk(helmet.doc, ni, root, {(hp.mesh, hp.primitive): HELMET})
    for i in range(10):                                                                              # model.rs:122-134
        angle = _f32(2) * PI * _f32(i) / _f32(10)
        add(_translation(_f32(3.5) * np.cos(angle, dtype=np.float32), _f32(3.5) * np.sin(angle, dtype=np.float32), -17), SPHERE)
    inst = np.array(inst, dtype=abi.INSTANCE)
    assert len(inst) == 2 + 103 + 1 + 10 and len(meshes) == 109

    # the scene three times over: mesh tables appended (running sums go on), copies moved aside
    m3 = np.concatenate([meshes] * 3)
    m3["base_index"] = np.concatenate([[0], np.cumsum(m3["index_count"].astype(np.uint64))[:-1]]).astype(np.uint32)
    vcount = np.array([r[3] for r in rows] * 3, dtype=np.int64)
    m3["vertex_offset"] = np.concatenate([[0], np.cumsum(vcount)[:-1]]).astype(np.int32)
    i3 = []
    for k, dx in enumerate((0.0, 55.0, -55.0)):
        for r in inst:
            M = _mul(_translation(dx, 0, 0), r["transform"].reshape(4, 4))
            i3.append(synth.instance_from_matrix(M.reshape(16), mesh=int(r["mesh"]) + 109 * k))
    i3 = np.array(i3, dtype=abi.INSTANCE)

    cases = {"cull_model_scene": (synth.camera_uniform(), meshes, inst),                                      # model.rs:235
             "cull_model_scene_nave": (synth.camera_uniform(eye=(-20.0, 3.0, 0.0), yaw_deg=90.0, pitch_deg=0.0), meshes, inst),
             "cull_model_scene_x3": (synth.camera_uniform(eye=(0.0, 3.0, 0.0), yaw_deg=180.0, pitch_deg=0.0), m3, i3)}
    for name, (cam, ms, ins) in cases.items():
        d_np = npr.cull_emit(cam, ms, ins)
        d_c = ref.cull_emit(cam, ms, ins)
        assert d_np.tobytes() == d_c.tobytes(), name
        comp, cnt = npr.compact(d_np)
        c2, cnt2 = ref.compact(d_c)
        assert cnt == cnt2 and comp.tobytes() == c2[:cnt2].tobytes()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), camera=cam, meshes=ms, instances=ins, draws=d_np, compact=comp, count=np.uint32(cnt))
        print(name, "visible", cnt, "/", len(ins), "meshes", len(ms))
    # ... and the top level App::setup_scene builds over exactly these instances (app.rs:252-253 -> MeshPool::generate_tlas ->
    # Tlas::build, tlas.rs:31-105): 116 leaves, most of them Sponza primitives whose boxes nest and overlap
    t_np = npr.tlas_nodes(inst, meshes)
    t_c = ref.tlas_build(inst, meshes)
    assert same(t_np, t_c)
    np.savez_compressed(os.path.join(OUT, "tlas_model_scene.npz"), instances=inst, meshes=meshes, nodes=t_np)
    print("tlas_model_scene", len(t_np), "nodes")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    helmet_case()
    model_scene_cases()
    cull_cases()
    blas_cases()
    builtin_pool_case()
    tlas_trace_cases()
    nan_cases()
    harness_case()
    occlusion_case()
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes", sz)
