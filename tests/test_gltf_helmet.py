"""The reference's own mesh asset as a parity input (tests/golden/helmet.npz, made by tests/golden/make_golden.py from
assets/glTF-Sample-Models/2.0/DamagedHelmet/glTF-Binary/DamagedHelmet.glb: the arrays GltfDocument::import hands to
MeshPool::add) and the glTF reader that extracted it (tests/gltf_reader.py - test infrastructure: asset IO is out of the path's scope)."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import ROOT, fields_equal
from voidin_amd import abi, synth
from gltf_reader import GltfDocument

GOLD = os.path.join(ROOT, "tests", "golden", "helmet.npz")


def _glb(doc: dict, binary: bytes) -> bytes:
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    binary += b"\0" * (-len(binary) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binary), 0x004E4942) + binary
    return struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body


def test_gltf_reader_reads_what_the_reference_import_reads(tmp_path):
    """POSITION bytes as Vec3s, indices widened to u32 (u8 / u16 / u32 / none), primitives without NORMAL skipped
    (gltf_model/mod.rs:118-150), node transforms composed parent-first with the children listed before the node itself
    (mod.rs:180-207); a self-contained .glb and a .gltf with an interleaved external buffer."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], dtype=np.float32)
    nor = np.tile(np.array([0, 0, 1], dtype=np.float32), (4, 1))
    idx16 = np.array([0, 1, 2, 2, 1, 3], dtype=np.uint16)
    blob = pos.tobytes() + nor.tobytes() + idx16.tobytes()
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(blob)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 48},
                           {"buffer": 0, "byteOffset": 96, "byteLength": 12}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                         {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2},
                                      {"attributes": {"POSITION": 0}, "indices": 2},              # no normals: skipped
                                      {"attributes": {"POSITION": 0, "NORMAL": 1}}]}],            # no indices: 0..n
           "nodes": [{"children": [1], "translation": [1, 2, 3]}, {"mesh": 0, "rotation": [0.7071067811865476, 0, 0, 0.7071067811865476], "scale": [2, 2, 2]}],
           "scenes": [{"nodes": [0]}]}
    path = tmp_path / "quad.glb"
    path.write_bytes(_glb(doc, blob))
    d = GltfDocument.load(str(path))
    prims = d.primitives()
    assert [(p.mesh, p.primitive) for p in prims] == [(0, 0), (0, 2)]
    assert prims[0].positions.dtype == np.float32 and np.array_equal(prims[0].positions, pos)
    assert prims[0].indices.dtype == np.uint32 and prims[0].indices.tolist() == [0, 1, 2, 2, 1, 3]
    assert prims[1].indices.tolist() == [0, 1, 2, 3]
    inst = d.scene_instances(np.diag([1.0, 1.0, 1.0, 1.0]))
    assert [(m, p) for _, m, p in inst] == [(0, 0), (0, 2)]
    M = inst[0][0]
    assert np.allclose(M @ np.array([0, 1, 0, 1.0]), [1, 2, 3 + 2, 1])          # +y -> scaled by 2, rotated about x onto +z, then translated
    # .gltf + external interleaved buffer (position and normal share a view with a 24-byte stride)
    inter = np.concatenate([pos, nor], axis=1).astype(np.float32).tobytes() + np.array([0, 1, 2, 2, 1, 3], dtype=np.uint32).tobytes()
    (tmp_path / "quad.bin").write_bytes(inter)
    doc2 = {"asset": {"version": "2.0"}, "buffers": [{"uri": "quad.bin", "byteLength": len(inter)}],
            "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 96, "byteStride": 24}, {"buffer": 0, "byteOffset": 96, "byteLength": 24}],
            "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                          {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": 4, "type": "VEC3"},
                          {"bufferView": 1, "componentType": 5125, "count": 6, "type": "SCALAR"}],
            "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2}]}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}]}
    (tmp_path / "quad.gltf").write_text(json.dumps(doc2))
    (p2,) = GltfDocument.load(str(tmp_path / "quad.gltf")).primitives()
    assert np.array_equal(p2.positions, pos) and p2.indices.tolist() == [0, 1, 2, 2, 1, 3]


def test_fixture_is_the_reference_asset_when_it_is_here():
    src = "/root/reference/assets/glTF-Sample-Models/2.0/DamagedHelmet/glTF-Binary/DamagedHelmet.glb"
    if not os.path.exists(src):
        pytest.skip("the reference checkout is not on this machine")
    g = np.load(GOLD)
    (p,) = GltfDocument.load(src).primitives()
    assert np.array_equal(p.positions, g["vertices"]) and np.array_equal(p.indices, g["indices"])


def test_oracle_on_the_reference_helmet(oracle):
    """The C oracle's BLAS over the helmet is the one the fixture froze (node count + CRCs), every leaf holds <= 3 triangles,
    the permuted index buffer is a permutation of the triangles, and its trace of the demo's view reproduces the stored hits."""
    g = np.load(GOLD)
    v, i = g["vertices"], g["indices"]
    nodes, idx = oracle.bvh_build(v, i)
    assert len(nodes) == int(g["n_nodes"]) and zlib.crc32(nodes.tobytes()) == int(g["nodes_crc"]) and zlib.crc32(idx.tobytes()) == int(g["indices_out_crc"])
    assert nodes["count"].max() <= 3
    tri_in = np.sort(i.reshape(-1, 3).view([("a", "<u4"), ("b", "<u4"), ("c", "<u4")]).reshape(-1), order=("a", "b", "c"))
    tri_out = np.sort(idx.reshape(-1, 3).view([("a", "<u4"), ("b", "<u4"), ("c", "<u4")]).reshape(-1), order=("a", "b", "c"))
    assert np.array_equal(tri_in, tri_out)
    tl = oracle.tlas_build(g["instances"], g["meshes"])
    assert fields_equal(tl, g["tlas"])
    rays = synth.primary_rays(g["camera"], int(g["width"]), int(g["height"]))
    hits, _ = oracle.trace((tl, g["instances"], g["meshes"], nodes, v, idx), rays, threads=8)
    assert np.array_equal(hits["hit"], g["hit"])
    h = g["hit"] == 1
    assert hits["dist"][h].tobytes() == g["dist"][h].tobytes() and np.array_equal(hits["triangle"][h], g["triangle"][h])


@pytest.mark.gpu
def test_gpu_builds_and_traces_the_reference_helmet(ctx, oracle):
    """The helmet through the C ABI: vd_bvh_build == the oracle's tree bit for bit (and the frozen CRCs), and the demo's view
    traced through the plain, the indexed and the prepared walk == the stored hits."""
    import torch
    g = np.load(GOLD)
    v, i = g["vertices"], g["indices"]
    want_nodes, want_idx = oracle.bvh_build(v, i)
    nodes, idx = ctx.bvh_build(v, i)
    assert fields_equal(nodes, want_nodes) and np.array_equal(idx, want_idx)
    assert zlib.crc32(nodes.tobytes()) == int(g["nodes_crc"]) and zlib.crc32(idx.tobytes()) == int(g["indices_out_crc"])
    tl = ctx.tlas_build(g["instances"], g["meshes"])
    assert fields_equal(tl, g["tlas"])
    rays = synth.primary_rays(g["camera"], int(g["width"]), int(g["height"]))
    ds = ctx.device_scene((tl, g["instances"], g["meshes"], nodes, v, idx))
    acc = ctx.trace_prepare(ds)
    d_rays, d_hits = ctx.upload(rays), ctx.empty(len(rays) * 16)
    d_any = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
    h = g["hit"] == 1
    for mode in ("plain", "indexed", "prepared"):
        ctx.set_option("trace.auto_prepare", 0 if mode == "indexed" else None)
        d_hits.zero_(); d_any.fill_(5)
        if mode == "prepared":
            ctx.trace_prepared_dev(acc, d_rays, len(rays), d_hits); ctx.trace_any_prepared_dev(acc, d_rays, len(rays), d_any)
        else:
            ctx.trace_dev(ds, d_rays, len(rays), d_hits); ctx.trace_any_dev(ds, d_rays, len(rays), d_any)
        got = d_hits.cpu().numpy().view(abi.HIT)[: len(rays)]
        assert np.array_equal(got["hit"], g["hit"]), mode
        assert got["dist"][h].tobytes() == g["dist"][h].tobytes() and np.array_equal(got["triangle"][h], g["triangle"][h]), mode
        assert np.array_equal(d_any.cpu().numpy().astype(np.uint8), g["hit"]), mode
    ctx.set_option("trace.auto_prepare", None)
    acc.close()
