"""vd_bvh_build_batch*: K meshes in ONE build (MeshPool::add for a whole scene, crates/pools/src/mesh/mod.rs:309-351).
Every mesh's node array and permuted index buffer must be what vd_bvh_build gives for that mesh alone and what the
oracle's literal builder gives - bit for bit - whatever its neighbours in the batch are and whichever tier it starts in."""
import os
import zlib

import numpy as np
import pytest

from conftest import ROOT, fields_equal, golden
from voidin_amd import abi, synth
from voidin_amd.runtime import VoidinError

pytestmark = pytest.mark.gpu


def scene_meshes():
    """The reference's own meshes (plane pair + sphere pair of MeshPool::new, mesh/mod.rs:267-274; cube.obj; the DamagedHelmet)
    and meshes for every tier: 1..4 triangles (leaf roots), <= 512 (small), <= 2048 (mid), levels of phase A above."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "helmet.npz"))
    cube = golden("blas_cube_obj.npz")
    out = [synth.plane_mesh(), synth.plane_mesh_rot_x(), synth.uv_sphere(1.0, 1), synth.uv_sphere(1.0, 10),
           (cube["vertices"], cube["indices"]), (g["vertices"], g["indices"]),
           synth.knot_mesh(40, 16, seed=5), synth.knot_mesh(96, 24), synth.knot_mesh(256, 64),
           synth.triangle_soup(64), synth.triangle_soup(1), synth.triangle_soup(3, seed=7), synth.triangle_soup(4, seed=8),
           synth.triangle_soup(513, seed=9), synth.triangle_soup(2048, seed=10), synth.triangle_soup(2049, seed=11)]
    return [(np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3), np.asarray(i, dtype=np.uint32).reshape(-1)) for v, i in out]


def test_batch_equals_single_builds_equals_oracle(ctx, oracle):
    meshes = scene_meshes()
    want = [oracle.bvh_build(v, i) for v, i in meshes]
    single = [ctx.bvh_build(v, i) for v, i in meshes]
    for (sn, si), (wn, wi) in zip(single, want):
        assert fields_equal(sn, wn) and np.array_equal(si, wi)
    # packed: one shared node array, MeshPool's bvh_index bookkeeping
    shared, parts = ctx.bvh_build_batch(meshes, packed=True)
    at = 0
    for m, ((first, n_nodes, idx), (wn, wi)) in enumerate(zip(parts, want)):
        assert first == at and n_nodes == len(wn), f"mesh {m}: nodes [{first}, +{n_nodes}) want [{at}, +{len(wn)})"
        assert fields_equal(shared[first: first + n_nodes], wn), f"mesh {m}: nodes"
        assert np.array_equal(idx, wi), f"mesh {m}: index permutation"
        at += n_nodes
    assert at == len(shared)
    # own arrays per mesh, and another order of the same meshes (a mesh's result does not depend on its neighbours)
    order = list(range(len(meshes)))[::-1]
    own = ctx.bvh_build_batch([meshes[k] for k in order], packed=False)
    for (nodes, idx), k in zip(own, order):
        assert fields_equal(nodes, want[k][0]) and np.array_equal(idx, want[k][1]), f"mesh {k}"
    st = ctx.bvh_last_build_stats()
    assert st["levels_phase_a"] >= 2 and st["n_mid_roots"] >= 2 and st["n_small_roots"] > len(meshes)


def test_batch_of_one_and_of_only_small_meshes(ctx, oracle):
    for meshes in ([synth.knot_mesh(128, 32)], [synth.triangle_soup(n, seed=40 + n) for n in (2, 5, 17, 64, 300, 512)]):
        want = [oracle.bvh_build(v, i) for v, i in meshes]
        shared, parts = ctx.bvh_build_batch(meshes)
        for (first, n_nodes, idx), (wn, wi) in zip(parts, want):
            assert fields_equal(shared[first: first + n_nodes], wn) and np.array_equal(idx, wi)


def test_device_form_packed_behind_existing_nodes(ctx, oracle):
    """vd_bvh_build_batch_dev into a node buffer that already holds other meshes' nodes (packed_first > 0): the batch's
    nodes go behind them, out_first_node = what MeshInfo.bvh_index gets, the nodes in front are untouched."""
    import ctypes as C

    import torch
    meshes = [synth.uv_sphere(1.0, 10), synth.knot_mesh(96, 24), synth.triangle_soup(3, seed=3), synth.knot_mesh(256, 64)]
    want = [oracle.bvh_build(v, i) for v, i in meshes]
    first = 1000
    cap = first + sum(2 * (len(i) // 3) for _, i in meshes)
    d_nodes = torch.full((cap * 32,), 0x5A, dtype=torch.uint8, device="cuda")
    items = (abi.BvhBatchItem * len(meshes))()
    keep = []
    for m, (v, i) in enumerate(meshes):
        d_v, d_i = ctx.upload(np.ascontiguousarray(v, dtype=np.float32)), ctx.upload(np.ascontiguousarray(i, dtype=np.uint32))
        keep.append((d_v, d_i))
        items[m].verts_xyz, items[m].indices_inout, items[m].out_nodes = abi.ptr(d_v), abi.ptr(d_i), None
        items[m].n_vert, items[m].n_tri, items[m].node_cap = len(v), len(i) // 3, 0
    end = ctx.bvh_build_batch_dev(items, len(meshes), d_nodes, cap, first)
    host = d_nodes.cpu().numpy()
    assert (host[: first * 32] == 0x5A).all()
    at = first
    for m, (wn, wi) in enumerate(want):
        assert items[m].status == 0 and items[m].out_first_node == at and items[m].out_n_nodes == len(wn)
        assert fields_equal(host[at * 32: (at + len(wn)) * 32].view(abi.BVH_NODE), wn), f"mesh {m}"
        assert np.array_equal(keep[m][1].cpu().numpy().view(np.uint32), wi), f"mesh {m}"
        at += len(wn)
    assert end == at
    # a packed buffer that is too small fails the batch and names the mesh
    with pytest.raises(VoidinError) as e:
        ctx.bvh_build_batch_dev(items, len(meshes), d_nodes, first + len(want[0][0]) + 10, first)
    assert e.value.code == abi.VD_ERR_INVALID_ARG and items[1].status == abi.VD_ERR_INVALID_ARG and items[0].status == 0


def test_batch_errors_name_the_mesh(ctx):
    good = synth.uv_sphere(1.0, 10)
    bad_v, bad_i = synth.triangle_soup(700, seed=12)
    bad_i = np.array(bad_i, dtype=np.uint32)
    bad_i[301] = len(bad_v) + 5                                   # an index beyond the mesh's vertices
    K = 3
    items = (abi.BvhBatchItem * K)()
    arrs = []
    for m, (v, i) in enumerate([good, (bad_v, bad_i), good]):
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
        i = np.array(i, dtype=np.uint32).reshape(-1).copy()
        nodes = np.zeros(2 * (len(i) // 3), dtype=abi.BVH_NODE)
        arrs.append((v, i, nodes))
        items[m].verts_xyz, items[m].indices_inout, items[m].out_nodes = v.ctypes.data, i.ctypes.data, nodes.ctypes.data
        items[m].n_vert, items[m].n_tri, items[m].node_cap = len(v), len(i) // 3, len(nodes)
    import ctypes as C
    rc = ctx.lib.vd_bvh_build_batch(ctx.h, C.addressof(items), K, None, 0, 0, None)
    assert rc == abi.VD_ERR_INVALID_ARG and b"mesh 1" in ctx.lib.vd_last_error(ctx.h)
    assert [items[m].status for m in range(K)] == [0, abi.VD_ERR_INVALID_ARG, 0]
    # degenerate input (>= 4 triangles with one centroid: the reference crashes) fails the batch
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    deg = (np.tile(tri, (8, 1)), np.arange(24, dtype=np.uint32))
    with pytest.raises(VoidinError) as e:
        ctx.bvh_build_batch([good, deg])
    assert e.value.code == abi.VD_ERR_DEGENERATE
    with pytest.raises(VoidinError):
        ctx.bvh_build_batch([])


def test_many_meshes_batch_vs_single_builds(ctx):
    """400 meshes of 1 k - 50 k triangles (a Sponza-class load): the batch == the 400 single builds (CRC of nodes and
    indices per mesh)."""
    rng = np.random.default_rng(5)
    meshes = []
    for k in range(400):
        t = int(np.exp(rng.uniform(np.log(1000), np.log(50_000))))
        u = max(8, int(np.sqrt(t / 2 * 4)))
        vv = max(4, t // (2 * u))
        meshes.append(synth.knot_mesh(u, vv, seed=synth.SEED_BASE + 100 + k))
    shared, parts = ctx.bvh_build_batch(meshes)
    for m, (v, i) in enumerate(meshes):
        sn, si = ctx.bvh_build(v, i)
        first, n_nodes, idx = parts[m]
        assert n_nodes == len(sn) and zlib.crc32(shared[first: first + n_nodes].tobytes()) == zlib.crc32(sn.tobytes()), f"mesh {m}"
        assert np.array_equal(idx, si), f"mesh {m}"
