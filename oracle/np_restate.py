"""Independent numpy-float32 restatement of the visibility path — TEST INFRASTRUCTURE ONLY.

Second, separately written restatement of the same reference lines as the C oracle
(oracle/vd_oracle_*.c).  It exists to cross-check the C oracle HERE (tests/test_oracle_*.py)
and to write the golden vectors under tests/golden/ (tests/golden/make_golden.py).  It is slow
(pure-Python loops for the order-dependent parts) and only used on small cases.
numpy float32 element-wise add/mul/div/sqrt are IEEE-754 single operations without FMA
contraction, which is the evaluation model SURVEY.md §8a C2' fixes.

Reference lines restated: shaders/emit_draws.wgsl:13-64, shaders/utils/math.wgsl:67-73,
crates/bvh/src/blas.rs:51-204, crates/bvh/src/tlas.rs:31-105,
crates/bvh/src/intersection.rs:16-19, shaders/utils/bvh.wgsl:35-123,
shaders/utils/intersections.wgsl:13-45.
"""
from __future__ import annotations

import numpy as np

from voidin_amd import abi

F = np.float32
MAX_DIST = F(1e30)


# ------------------------------------------------------------------ cull / emit ----------
def _len3(x, y, z):
    return np.sqrt((x * x + y * y) + z * z)


def cull_emit(cam, meshes, inst):
    V = cam["view"].reshape(4, 4)                       # [col][row]
    T = inst["transform"].reshape(-1, 4, 4)             # [i][col][row]
    mid = np.minimum(inst["mesh"], np.uint32(len(meshes) - 1))
    m = meshes[mid]
    c0 = (m["max"] + m["min"]) / F(2.0)
    n = len(inst)
    VT = np.empty((n, 4, 3), dtype=F)
    for j in range(4):
        for r in range(3):
            VT[:, j, r] = ((V[0, r] * T[:, j, 0] + V[1, r] * T[:, j, 1]) + V[2, r] * T[:, j, 2]) + V[3, r] * T[:, j, 3]
    c = np.empty((n, 3), dtype=F)
    for r in range(3):
        c[:, r] = ((VT[:, 0, r] * c0[:, 0] + VT[:, 1, r] * c0[:, 1]) + VT[:, 2, r] * c0[:, 2]) + VT[:, 3, r] * F(1.0)
    s = [_len3(T[:, j, 0], T[:, j, 1], T[:, j, 2]) for j in range(3)]
    max_scale = np.maximum(np.maximum(np.abs(s[0]), np.abs(s[1])), np.abs(s[2]))
    dmin = m["min"] - c
    dmax = m["max"] - c
    radius = np.maximum(_len3(dmin[:, 0], dmin[:, 1], dmin[:, 2]),
                        _len3(dmax[:, 0], dmax[:, 1], dmax[:, 2])) * max_scale
    fr = cam["frustum"]
    vis = np.ones(n, dtype=bool)
    with np.errstate(invalid="ignore"):
        vis &= ~(c[:, 2] * fr[1] - np.abs(c[:, 0]) * fr[0] < -radius)
        vis &= ~(c[:, 2] * fr[3] - np.abs(c[:, 1]) * fr[2] < -radius)
        vis &= ~((c[:, 2] + radius > cam["znear"]) & (c[:, 2] - radius > cam["zfar"]))
    out = np.zeros(n, dtype=abi.DRAW)
    out["vertex_count"] = m["index_count"]
    out["instance_count"] = vis.astype(np.uint32)
    out["base_index"] = m["base_index"]
    out["vertex_offset"] = m["vertex_offset"]
    out["base_instance"] = np.arange(n, dtype=np.uint32)
    return out


def compact(draws):
    keep = draws["instance_count"] == 1
    return draws[keep].copy(), int(keep.sum())


# ------------------------------------------------------------------ BLAS ------------------
def _area(mn, mx):
    d = (mx - mn).astype(F)
    return F(F(F(F(d[0] * d[1]) + F(d[0] * d[2])) + F(d[1] * d[2])) * F(2.0))


class _Blas:
    def __init__(self, verts, indices):
        self.v = np.asarray(verts, dtype=F).reshape(-1, 3)
        self.idx = np.asarray(indices, dtype=np.uint32).reshape(-1, 3)
        t = self.v[self.idx]                                      # [T][3][3]
        with np.errstate(over="ignore"):
            self.cent = (((t[:, 0] + t[:, 1]) + t[:, 2]) / F(3.0)).astype(F)
        self.tmin = t.min(axis=1)
        self.tmax = t.max(axis=1)
        self.ids = list(range(len(self.idx)))
        self.nodes = np.zeros(2 * len(self.idx), dtype=abi.BVH_NODE)
        self.cent_cols = [self.cent[:, a].tolist() for a in range(3)]

    def bounds(self, first, amount, centroids):
        if amount == 0:
            return np.full(3, MAX_DIST, F), np.full(3, -MAX_DIST, F)
        sel = np.asarray(self.ids[first:first + amount], dtype=np.int64)
        if centroids:
            c = self.cent[sel]
            mn, mx = c.min(axis=0), c.max(axis=0)
        else:
            mn, mx = self.tmin[sel].min(axis=0), self.tmax[sel].max(axis=0)
        return np.minimum(mn, MAX_DIST).astype(F), np.maximum(mx, -MAX_DIST).astype(F)

    def shuffle(self, axis, pos, start, count):
        ids, key = self.ids, self.cent_cols[axis]
        pos = float(pos)
        i, e = start, start + count - 1
        while i < e:
            if key[ids[i]] < pos:
                i += 1
            else:
                ids[i], ids[e] = ids[e], ids[i]
                e -= 1
        return i

    def partition(self, start, count):
        best = (0, F(0), 0, np.finfo(F).max)
        ok = False
        cmn, cmx = self.bounds(start, count, True)
        with np.errstate(over="ignore", invalid="ignore"):
            for axis in range(3):
                for k in range(1, 8):
                    scale = F(k) / F(8)
                    pos = F(cmn[axis] + F(F(cmx[axis] - cmn[axis]) * scale))
                    piv = self.shuffle(axis, pos, start, count)
                    n1 = piv - start
                    n2 = count - n1
                    a1 = _area(*self.bounds(start, n1, False))
                    a2 = _area(*self.bounds(piv, n2, False))
                    cost = F(F(a1 * F(n1)) + F(a2 * F(n2)))
                    if cost < best[3]:
                        best = (axis, pos, piv, cost)
                        ok = True
        self.shuffle(best[0], best[1], start, count)
        if not ok:
            raise ValueError("degenerate")
        return best[2]

    def build(self):
        nd = self.nodes
        n = len(self.idx)
        nd[0]["left_first"], nd[0]["count"] = 0, n
        nd[0]["min"], nd[0]["max"] = self.bounds(0, n, False)
        pool = 2
        stack = [(0, 0)]
        while stack:
            cur, start = stack.pop()
            cnt = int(nd[cur]["count"])
            if cnt <= 3:
                nd[cur]["left_first"] = start
                continue
            index = pool
            pool += 2
            nd[cur]["left_first"] = index
            piv = self.partition(start, cnt)
            lc = piv - start
            nd[index]["count"] = lc
            nd[index]["min"], nd[index]["max"] = self.bounds(start, lc, False)
            nd[index + 1]["count"] = cnt - lc
            nd[index + 1]["min"], nd[index + 1]["max"] = self.bounds(piv, cnt - lc, False)
            nd[cur]["count"] = 0
            stack.append((index + 1, piv))
            stack.append((index, start))
        return nd[:pool].copy(), self.idx[np.asarray(self.ids)].reshape(-1).copy()


def bvh_build(verts, indices):
    return _Blas(verts, indices).build()


# ------------------------------------------------------------------ TLAS ------------------
def tlas_leaf_bounds(inst, meshes):
    m = meshes[np.minimum(inst["mesh"], np.uint32(len(meshes) - 1))]
    T = inst["transform"].reshape(-1, 4, 4)
    b = np.stack([m["min"], m["max"]], axis=1)            # [i][2][3]
    mn, mx = m["min"].copy(), m["max"].copy()
    for i in range(8):
        ix, iy, iz = int((i & 1) == 0), int((i & 2) == 0), int((i & 4) == 0)
        px, py, pz = b[:, ix, 0], b[:, iy, 1], b[:, iz, 2]
        p = np.stack([((T[:, 0, r] * px + T[:, 1, r] * py) + T[:, 2, r] * pz) + T[:, 3, r] for r in range(3)], axis=1)
        mn, mx = np.minimum(mn, p), np.maximum(mx, p)
    return mn.astype(F), mx.astype(F)


def _areas(mn, mx):
    d = mx - mn
    return ((d[:, 0] * d[:, 1] + d[:, 0] * d[:, 2]) + d[:, 1] * d[:, 2]) * F(2.0)


def tlas_build(inst, meshes):
    n = len(inst)
    total = 2 * n + 1
    bmin, bmax = np.zeros((total, 3), F), np.zeros((total, 3), F)
    left, right = np.zeros(total, np.uint32), np.zeros(total, np.uint32)
    iidx = np.zeros(total, np.uint32)
    bmin[1:n + 1], bmax[1:n + 1] = tlas_leaf_bounds(inst, meshes)
    iidx[1:n + 1] = np.arange(n)
    ni = np.arange(1, n + 1)

    def best(cnt, t):
        if cnt == 0:
            return t
        sel = ni[:cnt]
        a = _areas(np.minimum(bmin[ni[t]], bmin[sel]), np.maximum(bmax[ni[t]], bmax[sel]))
        if t < cnt:
            a[t] = np.inf
        k = int(np.argmin(a))
        return k if a[k] < F(1e30) else t

    cnt, used, a = n, n + 1, 0
    b = best(cnt, a)
    while cnt > 0:
        c = best(cnt, b)
        if a == c:
            ia, ib = ni[a], ni[b]
            bmin[used], bmax[used] = np.minimum(bmin[ia], bmin[ib]), np.maximum(bmax[ia], bmax[ib])
            left[used], right[used], iidx[used] = ia, ib, 0xFFFFFFFF
            ni[a] = used
            used += 1
            ni[b] = ni[cnt - 1]
            cnt -= 1
            b = best(cnt, a)
        else:
            a, b = b, c
    r = ni[a]
    bmin[0], bmax[0], left[0], right[0], iidx[0] = bmin[r], bmax[r], left[r], right[r], iidx[r]
    return bmin, bmax, left, right, iidx


def tlas_nodes(inst, meshes):
    bmin, bmax, left, right, iidx = tlas_build(inst, meshes)
    out = np.zeros(len(left), dtype=abi.TLAS_NODE)
    out["min"], out["max"] = bmin, bmax
    out["left_right"] = left + (right << np.uint32(16))
    out["instance_idx"] = iidx
    return out


# ------------------------------------------------------------------ traversal -------------
def _aabb_hit(eye, inv, bmin, bmax, t):
    with np.errstate(invalid="ignore", over="ignore"):
        tx1 = (bmin - eye) * inv
        tx2 = (bmax - eye) * inv
        tmax = np.fmin.reduce(np.fmax(tx1, tx2))
        tmin = np.fmax.reduce(np.fmin(tx1, tx2))
    if tmax >= tmin and tmin < t and tmax > 0:
        return F(tmin)
    return MAX_DIST


def _dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def _cross(a, b):
    return np.array([F(a[1] * b[2]) - F(b[1] * a[2]), F(a[2] * b[0]) - F(b[2] * a[0]),
                     F(a[0] * b[1]) - F(b[0] * a[1])], dtype=F)


def _tri(eye, d, v0, v1, v2, hit):
    e1, e2 = v1 - v0, v2 - v0
    uvec = _cross(d, e2)
    det = _dot(e1, uvec)
    if det < F(1e-10):
        return None
    inv_det = F(1.0) / det
    o = eye - v0
    u = F(inv_det * _dot(o, uvec))
    if u < 0 or u > 1:
        return None
    vvec = _cross(o, e1)
    v = F(inv_det * _dot(d, vvec))
    if v < 0 or F(u + v) > 1:
        return None
    t = F(inv_det * _dot(e2, vvec))
    return t if (t > 0 and t < hit) else None


def trace(scene, rays):
    """scene = (tlas_nodes, instances, meshes, bvh_nodes, vertices[*,3], indices)."""
    tl, inst, meshes, bn, verts, idx = scene
    verts = np.asarray(verts, F).reshape(-1, 3)
    out = np.zeros(len(rays), dtype=abi.HIT)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for ri, ray in enumerate(rays):
            eye, d = ray["eye"].astype(F), ray["dir"].astype(F)
            inv = F(1.0) / d
            dist, hit = MAX_DIST, 0
            st = [0]
            while st:
                node = tl[st.pop()]
                if node["left_right"] == 0:
                    I = inst[node["instance_idx"]]
                    mesh = meshes[min(int(I["mesh"]), len(meshes) - 1)]
                    M = I["inv_transform"].reshape(4, 4)
                    e2 = np.array([((M[0, r] * eye[0] + M[1, r] * eye[1]) + M[2, r] * eye[2]) + M[3, r] * F(1) for r in range(3)], F)
                    d2 = np.array([((M[0, r] * d[0] + M[1, r] * d[1]) + M[2, r] * d[2]) + M[3, r] * F(0) for r in range(3)], F)
                    inv2 = F(1.0) / d2
                    bs = [int(mesh["bvh_index"])]
                    h = dist
                    while bs:
                        nd = bn[bs.pop()]
                        if nd["count"] > 0:
                            for i in range(int(nd["count"])):
                                t3 = int(nd["left_first"]) + i
                                vv = [verts[int(np.uint32(mesh["vertex_offset"])) + int(idx[int(mesh["base_index"]) + 3 * t3 + k])] for k in range(3)]
                                t = _tri(e2, d2, vv[0], vv[1], vv[2], h)
                                if t is not None:
                                    h = t
                                    dist, hit = t, 1
                        else:
                            a = int(mesh["bvh_index"]) + int(nd["left_first"])
                            b = a + 1
                            da = _aabb_hit(e2, inv2, bn[a]["min"], bn[a]["max"], h)
                            db = _aabb_hit(e2, inv2, bn[b]["min"], bn[b]["max"], h)
                            if da > db:
                                a, b, da, db = b, a, db, da
                            if da >= h:
                                continue
                            if db <= h:
                                bs.append(b)
                            bs.append(a)
                else:
                    a, b = int(node["left_right"] & 0xFFFF), int(node["left_right"] >> 16)
                    da = _aabb_hit(eye, inv, tl[a]["min"], tl[a]["max"], dist)
                    db = _aabb_hit(eye, inv, tl[b]["min"], tl[b]["max"], dist)
                    if da > db:
                        a, b, da, db = b, a, db, da
                    if da >= dist:
                        continue
                    if db < dist:
                        st.append(b)
                    st.append(a)
            out[ri]["dist"], out[ri]["hit"] = dist, hit
    return out
