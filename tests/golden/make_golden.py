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


def blas_cases():
    cases = {
        "plane": synth.plane_mesh(),                   # crates/pools/src/mesh/plane.rs:5-38
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


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    cull_cases()
    blas_cases()
    tlas_trace_cases()
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes", sz)
