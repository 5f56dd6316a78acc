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



# ------------------------------------------------------------------ Rust f32::min / max ----
def _key(x):
    i = np.asarray(x, dtype=F).view(np.int32)
    return i ^ ((i >> 31) & np.int32(0x7FFFFFFF))


def rmin(a, b):
    """Rust f32::min, component-wise (glam 0.24 scalar Vec3::min): a NaN operand is ignored;
    -0 < +0 (spec decision)."""
    a, b = np.broadcast_arrays(np.asarray(a, dtype=F), np.asarray(b, dtype=F))
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.where(_key(b) < _key(a), b, a))).astype(F)


def rmax(a, b):
    a, b = np.broadcast_arrays(np.asarray(a, dtype=F), np.asarray(b, dtype=F))
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.where(_key(b) > _key(a), b, a))).astype(F)


def _unkey(k):
    k = np.asarray(k, dtype=np.int32)
    return (k ^ ((k >> 31) & np.int32(0x7FFFFFFF))).view(F)


def _fold_min(arr, seed):
    """fold of rmin over axis 0 from `seed` (the reference folds from +1e30: blas.rs:185-186);
    order-independent: the minimum of the non-NaN keys."""
    k = np.where(np.isnan(arr), np.int32(0x7FFFFFFF), _key(arr)).min(axis=0)
    return np.where(k < _key(seed), _unkey(k), seed).astype(F)


def _fold_max(arr, seed):
    k = np.where(np.isnan(arr), np.int32(-0x80000000), _key(arr)).max(axis=0)
    return np.where(k > _key(seed), _unkey(k), seed).astype(F)


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
        self.tmin = rmin(rmin(t[:, 0], t[:, 1]), t[:, 2])      # NaN vertices drop out (f32::min ignores a NaN)
        self.tmax = rmax(rmax(t[:, 0], t[:, 1]), t[:, 2])
        self.ids = list(range(len(self.idx)))
        self.nodes = np.zeros(2 * len(self.idx), dtype=abi.BVH_NODE)
        self.cent_cols = [self.cent[:, a].tolist() for a in range(3)]

    def bounds(self, first, amount, centroids):
        if amount == 0:
            return np.full(3, MAX_DIST, F), np.full(3, -MAX_DIST, F)
        sel = np.asarray(self.ids[first:first + amount], dtype=np.int64)
        lo, hi = (self.cent[sel], self.cent[sel]) if centroids else (self.tmin[sel], self.tmax[sel])
        return _fold_min(lo, np.full(3, MAX_DIST, F)), _fold_max(hi, np.full(3, -MAX_DIST, F))

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
        mn, mx = rmin(mn, p), rmax(mx, p)
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
        a = _areas(rmin(bmin[ni[t]], bmin[sel]), rmax(bmax[ni[t]], bmax[sel]))
        if t < cnt:
            a[t] = np.inf
        k = int(np.argmin(np.where(np.isnan(a), np.inf, a)))     # a NaN area never passes `<` (tlas.rs:99)
        return k if a[k] < F(1e30) else t

    cnt, used, a = n, n + 1, 0
    b = best(cnt, a)
    while cnt > 0:
        c = best(cnt, b)
        if a == c:
            ia, ib = ni[a], ni[b]
            bmin[used], bmax[used] = rmin(bmin[ia], bmin[ib]), rmax(bmax[ia], bmax[ib])
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


# ---- the CPU harness: per-pixel rays + Bvh::traverse_iter (src/bin/bvh_cpu.rs:71-96, blas.rs:247-295) ----
def primary_rays(cam, width, height):
    """bvh_cpu.rs:71-83 in float32, glam operation order (Mat4 * Vec4 = ((c0*x + c1*y) + c2*z) + c3*w;
    Vec3::normalize = v * (1 / sqrt((x*x + y*y) + z*z)))."""
    M = np.asarray(cam["clip_to_world"], F).reshape(16)
    n = width * height
    i = np.arange(n, dtype=np.int64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        x = ((i % width).astype(F) / F(width))
        y = ((i // height).astype(F) / F(height)) if n else x
        x = ((x - F(0.5)) * F(2.0)).astype(F)
        y = ((y - F(0.5)) * F(-2.0)).astype(F)
        p = [(((M[r] * x + M[4 + r] * y).astype(F) + M[8 + r] * F(1.0)).astype(F) + M[12 + r] * F(1.0)).astype(F) for r in range(4)]
        t = [(((M[r] * x + M[4 + r] * y).astype(F) + M[8 + r] * F(0.0)).astype(F) + M[12 + r] * F(1.0)).astype(F) for r in range(4)]
        rl = (F(1.0) / np.sqrt(((t[0] * t[0] + t[1] * t[1]).astype(F) + t[2] * t[2]).astype(F)).astype(F)).astype(F)
        rays = np.zeros(n, dtype=abi.RAY)
        for k in range(3):
            rays["eye"][:, k] = (p[k] / p[3]).astype(F)
            rays["dir"][:, k] = (t[k] * rl).astype(F)
    return rays


def _aabb_rs(orig, d, bmin, bmax, t):
    """intersection.rs:47-55 -> (hit, tmin)."""
    tx1 = ((bmin - orig) / d).astype(F)
    tx2 = ((bmax - orig) / d).astype(F)
    hi, lo = np.fmax(tx1, tx2), np.fmin(tx1, tx2)
    tmax = np.fmin(hi[0], np.fmin(hi[1], hi[2]))
    tmin = np.fmax(lo[0], np.fmax(lo[1], lo[2]))
    return bool(tmax >= tmin and tmin < t and tmax > 0), F(tmin)


def _tri_rs(orig, d, v0, v1, v2):
    """intersection.rs:68-92 -> t or None."""
    eps = F(0.0001)
    e1, e2 = v1 - v0, v2 - v0
    h = _cross(d, e2)
    a = _dot(e1, h)
    if -eps < a < eps:
        return None
    f = F(1.0) / a
    s = orig - v0
    u = F(f * _dot(s, h))
    if not (F(0) <= u <= F(1)):
        return None
    q = _cross(s, e1)
    v = F(f * _dot(d, q))
    if v < 0 or F(u + v) > 1:
        return None
    t = F(f * _dot(e2, q))
    return t if t > eps else None


def traverse_iter(nodes, verts, indices, rays):
    """Bvh::traverse_iter (blas.rs:247-295): -1 = Dist::Miss."""
    verts = np.asarray(verts, F).reshape(-1, 3)
    idx = np.asarray(indices, np.uint32).reshape(-1, 3)
    out = np.zeros(len(rays), dtype=F)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for ri, ray in enumerate(rays):
            orig, d = ray["eye"].astype(F), ray["dir"].astype(F)
            st, hit = [0], None
            while st:
                nd = nodes[st.pop()]
                if nd["count"] > 0:
                    for i in range(int(nd["count"])):
                        tri = idx[int(nd["left_first"]) + i]
                        t = _tri_rs(orig, d, verts[tri[0]], verts[tri[1]], verts[tri[2]])
                        if t is not None:
                            hit = t if hit is None else F(min(hit, t))
                else:
                    a, b = int(nd["left_first"]), int(nd["left_first"]) + 1
                    lim = hit if hit is not None else MAX_DIST
                    ha, ta = _aabb_rs(orig, d, nodes[a]["min"], nodes[a]["max"], lim)
                    hb, tb = _aabb_rs(orig, d, nodes[b]["min"], nodes[b]["max"], lim)
                    # derive(PartialOrd) on enum Dist { Hit(f32), Miss }: Hit(x) < Miss
                    gt = (not ha) if ha != hb else (ha and ta > tb)
                    if gt:
                        a, b, ha, hb = b, a, hb, ha
                    if not ha:
                        continue
                    st.append(a)
                    if hb:
                        st.append(b)
            out[ri] = F(-1.0) if hit is None else hit
    return out


# ---- occlusion extension (no reference counterpart; definition in include/voidin_abi.h) ----
def hiz_build(depth):
    depth = np.asarray(depth, F)
    levels = [depth]
    while levels[-1].shape != (1, 1):
        s = levels[-1]
        h, w = s.shape
        dh, dw = (h + 1) // 2, (w + 1) // 2
        ys = np.minimum(np.arange(dh)[:, None] * 2 + np.array([0, 1])[None, :], h - 1)     # clamped child rows
        xs = np.minimum(np.arange(dw)[:, None] * 2 + np.array([0, 1])[None, :], w - 1)
        levels.append(np.minimum.reduce([s[ys[:, a]][:, xs[:, b]] for a in (0, 1) for b in (0, 1)]))
    return np.concatenate([l.reshape(-1) for l in levels]).astype(F), [l.shape for l in levels]


def occlusion_mask(cam, meshes, inst, depth, mask_in):
    """Per instance, in float32 and the operation order the header fixes."""
    pyr, shapes = hiz_build(depth)
    offs = np.concatenate([[0], np.cumsum([h * w for h, w in shapes])])
    H, W = shapes[0]
    V = np.asarray(cam["view"], F).reshape(16)
    P = np.asarray(cam["projection"], F).reshape(16)
    znear = F(cam["znear"])
    n = len(inst)
    bits = np.unpackbits(np.asarray(mask_in, np.uint64).view(np.uint8), bitorder="little")[:n].astype(bool)
    keep = bits.copy()
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for i in np.flatnonzero(bits):
            m = meshes[min(int(inst["mesh"][i]), len(meshes) - 1)]
            T = inst["transform"][i].astype(F)
            mn, mx = m["min"].astype(F), m["max"].astype(F)
            c0 = ((mx + mn) / F(2.0)).astype(F)
            c = []
            for r in range(3):
                col = [F(F(F(V[r] * T[4 * j] + V[4 + r] * T[4 * j + 1]) + V[8 + r] * T[4 * j + 2]) + V[12 + r] * T[4 * j + 3]) for j in range(4)]
                c.append(F(F(F(col[0] * c0[0] + col[1] * c0[1]) + col[2] * c0[2]) + col[3] * F(1.0)))
            ms = max(abs(_len3(T[0], T[1], T[2])), abs(_len3(T[4], T[5], T[6])), abs(_len3(T[8], T[9], T[10])))
            e = (mx - mn).astype(F)
            r_ = F(F(_len3(e[0], e[1], e[2]) * F(0.5)) * ms)
            d = F(-c[2])
            dn = F(d - r_)
            if not (dn > znear):
                continue
            rr, dd, rd = F(r_ * r_), F(d * d), F(r_ * d)
            tx = F(np.sqrt(F(F(c[0] * c[0] + dd) - rr)))
            ty = F(np.sqrt(F(F(c[1] * c[1] + dd) - rr)))
            dxm, dxp = F(d * tx + c[0] * r_), F(d * tx - c[0] * r_)
            dym, dyp = F(d * ty + c[1] * r_), F(d * ty - c[1] * r_)
            if not (dxm > 0 and dxp > 0 and dym > 0 and dyp > 0):
                continue
            sx0, sx1 = F(F(c[0] * tx - rd) / dxm), F(F(c[0] * tx + rd) / dxp)
            sy0, sy1 = F(F(c[1] * ty - rd) / dym), F(F(c[1] * ty + rd) / dyp)
            nxa, nxb = F(P[0] * sx0 - P[8]), F(P[0] * sx1 - P[8])
            nya, nyb = F(P[5] * sy0 - P[9]), F(P[5] * sy1 - P[9])
            nx_lo, nx_hi, ny_lo, ny_hi = np.fmin(nxa, nxb), np.fmax(nxa, nxb), np.fmin(nya, nyb), np.fmax(nya, nyb)
            Wf, Hf = F(W), F(H)
            u0 = F(F(F(nx_lo * F(0.5)) + F(0.5)) * Wf - F(0.5)); u1 = F(F(F(nx_hi * F(0.5)) + F(0.5)) * Wf + F(0.5))
            v0 = F(F(F(0.5) - ny_hi * F(0.5)) * Hf - F(0.5)); v1 = F(F(F(0.5) - ny_lo * F(0.5)) * Hf + F(0.5))
            if not (u1 >= 0 and v1 >= 0 and u0 < Wf and v0 < Hf):
                continue
            x0, x1 = int(np.floor(np.fmax(u0, F(0)))), int(np.floor(np.fmin(u1, F(Wf - F(1)))))
            y0, y1 = int(np.floor(np.fmax(v0, F(0)))), int(np.floor(np.fmin(v1, F(Hf - F(1)))))
            lvl = min(max(x1 - x0, y1 - y0).bit_length(), len(shapes) - 1)
            lh, lw = shapes[lvl]
            t = pyr[offs[lvl]: offs[lvl] + lh * lw].reshape(lh, lw)
            hmin = min(t[y0 >> lvl, x0 >> lvl], t[y0 >> lvl, x1 >> lvl], t[y1 >> lvl, x0 >> lvl], t[y1 >> lvl, x1 >> lvl])
            depth_s = F(F(P[14] - F(P[10] * dn)) / dn)
            if depth_s < hmin:
                keep[i] = False
    out = np.zeros((n + 63) // 64 * 64, dtype=np.uint8)
    out[:n] = keep
    return np.packbits(out, bitorder="little").view(np.uint64)
