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


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    helmet_case()
    cull_cases()
    blas_cases()
    builtin_pool_case()
    tlas_trace_cases()
    nan_cases()
    harness_case()
    occlusion_case()
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes", sz)
