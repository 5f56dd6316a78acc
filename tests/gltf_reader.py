"""glTF mesh ingest: the geometry half of the reference's `GltfDocument::import`
(crates/app/src/models/gltf_model/mod.rs:100-155) - for every primitive that has POSITION (and NORMAL) the position
accessor's bytes taken as `[Vec3]` and `read_indices().into_u32()` (0..n when the primitive has none), handed to
`app.add_mesh(MeshRef)` -> `MeshPool::add` -> `BvhBuilder` (crates/pools/src/mesh/mod.rs:309-351) - and the node walk of
`gather_instances_recursive` (mod.rs:180-207: transform = parent * node.transform().matrix(), children first, one
instance per primitive of the node's mesh).  Only what the BVH path consumes: no materials, textures, tangents.
Self-contained `.glb` files and `.gltf` files whose buffers sit beside them; triangle lists only (mode 4, the glTF
default - the reference does not look at the mode either).
"""
from __future__ import annotations

import json
import os
import struct
from dataclasses import dataclass

import numpy as np

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


@dataclass
class GltfPrimitive:
    mesh: int                  # (gltf mesh index, primitive index): the key of the reference's mesh map
    primitive: int
    positions: np.ndarray      # (V, 3) f32
    indices: np.ndarray        # (3T,) u32

    def arrays(self):
        return self.positions, self.indices


class GltfDocument:
    def __init__(self, doc: dict, buffers: list):
        self.doc, self.buffers = doc, buffers

    @staticmethod
    def load(path: str) -> "GltfDocument":
        raw = open(path, "rb").read()
        if raw[:4] == b"glTF":
            magic, version, length = struct.unpack_from("<III", raw, 0)
            if version != 2 or length > len(raw):
                raise ValueError("not a glTF 2.0 binary")
            off, doc, binary = 12, None, None
            while off + 8 <= length:
                clen, ctype = struct.unpack_from("<II", raw, off)
                chunk = raw[off + 8: off + 8 + clen]
                if ctype == 0x4E4F534A:
                    doc = json.loads(chunk)
                elif ctype == 0x004E4942 and binary is None:
                    binary = chunk
                off += 8 + clen + (-clen % 4)
            if doc is None:
                raise ValueError("glb without a JSON chunk")
        else:
            doc, binary = json.loads(raw), None
        buffers = []
        for b in doc.get("buffers", []):
            if "uri" not in b:
                buffers.append(binary)
            elif b["uri"].startswith("data:"):
                import base64
                buffers.append(base64.b64decode(b["uri"].split(",", 1)[1]))
            else:
                buffers.append(open(os.path.join(os.path.dirname(path), b["uri"]), "rb").read())
        return GltfDocument(doc, buffers)

    def accessor(self, index: int) -> np.ndarray:
        a = self.doc["accessors"][index]
        if "sparse" in a or "bufferView" not in a:
            raise ValueError("sparse / view-less accessors are not supported")
        bv = self.doc["bufferViews"][a["bufferView"]]
        ct, w = np.dtype(_COMPONENT[a["componentType"]]), _WIDTH[a["type"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or ct.itemsize * w
        buf = self.buffers[bv["buffer"]]
        if stride == ct.itemsize * w:
            out = np.frombuffer(buf, dtype=ct, count=a["count"] * w, offset=off)
        else:
            rows = np.lib.stride_tricks.as_strided(np.frombuffer(buf, dtype=np.uint8, offset=off), shape=(a["count"], ct.itemsize * w),
                                                   strides=(stride, 1))
            out = np.ascontiguousarray(rows).view(ct).reshape(-1)
        return out.reshape(a["count"], w) if w > 1 else out

    def primitives(self) -> list:
        out = []
        for mi, m in enumerate(self.doc.get("meshes", [])):
            for pi, p in enumerate(m["primitives"]):
                at = p.get("attributes", {})
                if "POSITION" not in at or "NORMAL" not in at:          # mod.rs:118-123: skipped
                    continue
                pos = np.ascontiguousarray(self.accessor(at["POSITION"]), dtype=np.float32)
                idx = (self.accessor(p["indices"]).astype(np.uint32) if "indices" in p else np.arange(len(pos), dtype=np.uint32))
                out.append(GltfPrimitive(mi, pi, pos, np.ascontiguousarray(idx)))
        return out

    @staticmethod
    def _node_matrix(n: dict) -> np.ndarray:
        """node.transform().matrix() as a row-major 4x4 acting on column vectors (float64)."""
        if "matrix" in n:
            return np.asarray(n["matrix"], dtype=np.float64).reshape(4, 4).T
        t = np.asarray(n.get("translation", (0, 0, 0)), dtype=np.float64)
        x, y, z, w = np.asarray(n.get("rotation", (0, 0, 0, 1)), dtype=np.float64)
        s = np.asarray(n.get("scale", (1, 1, 1)), dtype=np.float64)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        M = np.eye(4)
        M[:3, :3] = R * s[None, :]
        M[:3, 3] = t
        return M

    def scene_instances(self, transform=None) -> list:
        """[(row-major 4x4 world matrix, gltf mesh index, primitive index)] in the reference's order (children first)."""
        out = []
        prim_ok = {(p.mesh, p.primitive) for p in self.primitives()}

        def walk(ni, parent):
            n = self.doc["nodes"][ni]
            M = parent @ self._node_matrix(n)
            for c in n.get("children", []):
                walk(c, M)
            if "mesh" in n:
                for pi in range(len(self.doc["meshes"][n["mesh"]]["primitives"])):
                    if (n["mesh"], pi) in prim_ok:
                        out.append((M, n["mesh"], pi))
        root = np.eye(4) if transform is None else np.asarray(transform, dtype=np.float64)
        for sc in self.doc.get("scenes", []):
            for ni in sc.get("nodes", []):
                walk(ni, root)
        return out
