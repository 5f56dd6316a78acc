"""OBJ ingest: the Python mirror of `ObjModel::load` in include/voidin.hpp, i.e. of the reference's
`ObjModel::import` (crates/app/src/models/mod.rs:19-57: tobj 4.0.0 `load_obj(path, &GPU_LOAD_OPTIONS)` =
triangulate + single_index, points and lines ignored, then one `MeshRef{positions, indices}` per model into
`MeshPool::add`).

tobj is not on disk; the reader restates its documented behaviour (parity unpinned): a new model starts at every
`o` / `g` statement and when `usemtl` switches material after faces were read; polygons are fan-triangulated
(v0, vi, vi+1); each distinct v/vt/vn triple becomes one vertex, numbered per model in order of first use; indices
may be negative (relative to the vertices declared so far).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class ObjMesh:
    name: str = "unnamed_object"
    positions: list = field(default_factory=list)     # (x, y, z) f32 triples
    normals: list = field(default_factory=list)
    texcoords: list = field(default_factory=list)     # u, v pairs, flat
    indices: list = field(default_factory=list)
    material_id: int = -1                             # order of the `usemtl` names' first appearance (no .mtl parsing)

    def arrays(self):
        """(vertices (V,3) f32, indices (3T,) u32) as BvhBuilder::new takes them."""
        return (np.asarray(self.positions, dtype=np.float32).reshape(-1, 3), np.asarray(self.indices, dtype=np.uint32))


class ObjModel:
    @staticmethod
    def load(path: str) -> list[ObjMesh]:
        v, vn, vt = [], [], []
        out: list[ObjMesh] = []
        cur = ObjMesh()
        seen: dict = {}
        materials: dict = {}

        def flush(next_name):
            nonlocal cur, seen
            mat = cur.material_id
            if cur.indices:
                out.append(cur)
            cur = ObjMesh(name=next_name, material_id=mat)
            seen = {}

        f32 = lambda t: float(np.float32(t))
        with open(path, "r", errors="replace") as f:
            for raw in f:
                line = raw.lstrip(" \t")
                tok = line.split()
                if not tok:
                    continue
                if tok[0] == "v":
                    v.append((f32(tok[1]), f32(tok[2]), f32(tok[3])))
                elif tok[0] == "vn":
                    vn.append((f32(tok[1]), f32(tok[2]), f32(tok[3])))
                elif tok[0] == "vt":
                    vt.append((f32(tok[1]), f32(tok[2])))
                elif tok[0] == "f":
                    poly = []
                    for t in tok[1:]:
                        if t.startswith("#"):
                            break
                        parts = (t.split("/") + ["", ""])[:3]
                        idx = [int(p) if p not in ("", "+", "-") else 0 for p in parts]
                        nv, nt, nn = len(v), len(vt), len(vn)
                        iv = nv + idx[0] if idx[0] < 0 else idx[0] - 1
                        it = -1 if idx[1] == 0 else (nt + idx[1] if idx[1] < 0 else idx[1] - 1)
                        inn = -1 if idx[2] == 0 else (nn + idx[2] if idx[2] < 0 else idx[2] - 1)
                        if iv < 0 or iv >= nv or it >= nt or inn >= nn:
                            raise ValueError(f"ObjModel: face index out of range in {path}")
                        key = (iv, it, inn)
                        k = seen.get(key)
                        if k is None:
                            k = seen[key] = len(cur.positions)
                            cur.positions.append(v[iv])
                            if it >= 0:
                                cur.texcoords.extend(vt[it])
                            if inn >= 0:
                                cur.normals.append(vn[inn])
                        poly.append(k)
                    for k in range(1, len(poly) - 1):      # fan; points and lines (< 3 vertices) are dropped
                        cur.indices.extend((poly[0], poly[k], poly[k + 1]))
                elif tok[0] in ("o", "g") and line[1:2] in (" ", "\t"):
                    flush(line[2:].rstrip("\n\r "))
                elif line.startswith("usemtl"):
                    name = line[7:].rstrip("\n\r ")
                    mid = materials.setdefault(name, len(materials))
                    if mid != cur.material_id and cur.indices:
                        flush(cur.name)
                    cur.material_id = mid
        if cur.indices:
            out.append(cur)
        return out
