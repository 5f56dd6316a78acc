"""Deterministic synthetic inputs for the visibility path.

Everything here produces INPUTS (the bytes handed to the path): camera uniform, MeshInfo
tables, instance clouds, triangle meshes, rays.  The reference builds these with glam / dolly
on the host (crates/components/src/camera.rs:128-169, crates/components/src/shared.rs:90-98,
crates/pools/src/mesh/{plane,sphere}.rs, src/bin/bvh_cpu.rs:39-52); they are restated in
float64 and rounded to f32 once, so they are plain data for both the oracle and the HIP path.

RNG: counter-based splitmix64 (the reference uses unseeded `thread_rng`: bvh_cpu.rs:39).
"""
from __future__ import annotations

import math

import numpy as np

from . import abi

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
SEED_BASE = 0x5EED0000


def splitmix64(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 uniforms in [0,1) with 24 random bits: element i of stream `stream`."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    s = (seed * 0x100000001B3 + stream * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF
    u = splitmix64(s, idx)
    return (u >> np.uint64(40)).astype(np.float64) * (2.0 ** -24)


# --- camera (camera.rs:128-169) ----------------------------------------------------------

def _look_at_rh(eye, center, up):
    f = center - eye
    f = f / np.linalg.norm(f)
    s = np.cross(f, up)
    s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.zeros((4, 4))  # m[col][row]
    m[0] = [s[0], u[0], -f[0], 0]
    m[1] = [s[1], u[1], -f[1], 0]
    m[2] = [s[2], u[2], -f[2], 0]
    m[3] = [-s.dot(eye), -u.dot(eye), f.dot(eye), 1]
    return m


def _perspective_infinite_reverse_rh(fovy, aspect, znear):
    f = 1.0 / math.tan(0.5 * fovy)
    m = np.zeros((4, 4))
    m[0] = [f / aspect, 0, 0, 0]
    m[1] = [0, f, 0, 0]
    m[2] = [0, 0, 0, -1]
    m[3] = [0, 0, znear, 0]
    return m


def camera_uniform(eye=(2.0, 5.0, 12.0), yaw_deg=0.0, pitch_deg=-20.0, aspect=1.25,
                   jitter=(0.0, 0.0), fovy=math.pi / 2, znear=0.001) -> np.ndarray:
    """The 320-byte CameraUniform of Camera::get_uniform (camera.rs:135-169) for a settled rig
    (dolly YawPitch: rotation = Ry(yaw)·Rx(pitch); forward = rot·(-Z); up = rot·Y).
    Defaults = src/bin/model.rs:235 + camera.rs:110-111,121."""
    eye = np.asarray(eye, dtype=np.float64)
    yaw, pitch = math.radians(yaw_deg), math.radians(pitch_deg)
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    rot = ry @ rx
    fwd = rot @ np.array([0.0, 0.0, -1.0])
    up = rot @ np.array([0.0, 1.0, 0.0])
    view = _look_at_rh(eye, eye + fwd, up)
    proj = _perspective_infinite_reverse_rh(fovy, aspect, znear)
    proj[2][0] += jitter[0]
    proj[2][1] += jitter[1]
    proj32 = proj.astype(np.float32)
    # column-major storage m[col][row] -> math matrix M[row][col] = m[col][row]
    P = proj.T
    V = view.T
    pv = P @ V
    cam = np.zeros((), dtype=abi.CAMERA)
    cam["view_position"] = np.array([*eye, 1.0], dtype=np.float32)
    cam["projection"] = proj32.reshape(16)
    cam["view"] = view.astype(np.float32).reshape(16)
    cam["clip_to_world"] = np.linalg.inv(pv).T.astype(np.float32).reshape(16)
    cam["prev_world_to_clip"] = pv.T.astype(np.float32).reshape(16)
    # camera.rs:143-148, in f32 as glam does: rows 3+0 and 3+1 of the projection, normalised
    row = lambda r: np.array([proj32[c][r] for c in range(4)], dtype=np.float32)
    fx = row(3) + row(0)
    fy = row(3) + row(1)

    def _norm(v):
        d = np.float32(0)
        for k in range(4):
            d = np.float32(d + np.float32(v[k] * v[k]))
        return (v * np.float32(np.float32(1.0) / np.float32(np.sqrt(d)))).astype(np.float32)

    fx, fy = _norm(fx), _norm(fy)
    cam["frustum"] = np.array([fx[0], fx[2], fy[1], fy[2]], dtype=np.float32)
    cam["zfar"] = np.float32(np.inf)
    cam["znear"] = np.float32(znear)
    cam["jitter"] = np.asarray(jitter, dtype=np.float32)
    return cam


# --- mesh table + instance cloud -------------------------------

def mesh_infos(n_mesh: int = 16, seed: int = SEED_BASE) -> np.ndarray:
    u = uniform01(seed, 100, n_mesh * 7).reshape(n_mesh, 7)
    half = 0.25 + 1.75 * u[:, 0:3]
    ctr = u[:, 3:6] - 0.5
    m = np.zeros(n_mesh, dtype=abi.MESH_INFO)
    m["min"] = (ctr - half).astype(np.float32)
    m["max"] = (ctr + half).astype(np.float32)
    # index_count in {36 .. 3e5}, multiple of 3, log-uniform
    ic = (36.0 * (300000.0 / 36.0) ** u[:, 6]).astype(np.int64) // 3 * 3
    m["index_count"] = ic.astype(np.uint32)
    base = np.concatenate([[0], np.cumsum(ic)[:-1]])
    m["base_index"] = base.astype(np.uint32)
    m["vertex_offset"] = (base // 3).astype(np.int32)
    m["bvh_index"] = (2 * base // 3).astype(np.uint32)
    return m


def _quat_to_mat(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(x), 3, 3))  # R[i][row][col]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _instances_chunk(out, c0, m, n_mesh, seed, extent, centre, lo, hi, offset, with_inverse):
    u = np.stack([uniform01(seed, s, m, offset + c0) for s in range(10)], axis=1)
    t = (u[:, 0:3] - 0.5) * extent + np.asarray(centre)
    # Shoemake uniform quaternion
    r1, r2 = np.sqrt(1 - u[:, 3]), np.sqrt(u[:, 3])
    a1, a2 = 2 * math.pi * u[:, 4], 2 * math.pi * u[:, 5]
    q = np.stack([r1 * np.sin(a1), r1 * np.cos(a1), r2 * np.sin(a2), r2 * np.cos(a2)], axis=1)
    R = _quat_to_mat(q)
    S = np.exp(lo + (hi - lo) * u[:, 6:9])
    M = np.zeros((m, 4, 4))  # M[i][col][row]
    for j in range(3):
        M[:, j, 0:3] = R[:, :, j] * S[:, j:j + 1]
    M[:, 3, 0:3] = t
    M[:, 3, 3] = 1.0
    rec = np.zeros(m, dtype=abi.INSTANCE)
    rec["transform"] = M.reshape(m, 16).astype(np.float32)
    if with_inverse:
        # Instance::new: inv_transform = transform.inverse() (shared.rs:93) — an input
        Mi = np.linalg.inv(np.transpose(rec["transform"].reshape(m, 4, 4).astype(np.float64), (0, 2, 1)))
        rec["inv_transform"] = np.transpose(Mi, (0, 2, 1)).reshape(m, 16).astype(np.float32)
    rec["mesh"] = np.minimum((u[:, 9] * n_mesh).astype(np.uint32), n_mesh - 1)
    rec["material"] = 1  # MaterialId::default (shared.rs:55-58)
    out[c0:c0 + m] = rec


def instances(n: int, n_mesh: int = 16, seed: int = SEED_BASE + 2, extent: float = 2000.0,
              centre=(0.0, 0.0, 0.0), scale_range=(0.25, 4.0), offset: int = 0,
              with_inverse: bool = True, chunk: int = 1 << 17, workers: int = 8) -> np.ndarray:
    """n instances `transform = T·R·S` (T uniform in an extent³ cube, R from a uniform unit
    quaternion, S log-uniform per axis), mesh id uniform.  Element i depends only on
    (seed, offset+i), so shards of one cloud can be generated independently."""
    out = np.zeros(n, dtype=abi.INSTANCE)
    lo, hi = math.log(scale_range[0]), math.log(scale_range[1])
    jobs = [(c0, min(chunk, n - c0)) for c0 in range(0, n, chunk)]
    args = (n_mesh, seed, extent, centre, lo, hi, offset, with_inverse)
    if len(jobs) <= 1 or workers <= 1:
        for c0, m in jobs:
            _instances_chunk(out, c0, m, *args)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(lambda j: _instances_chunk(out, j[0], j[1], *args), jobs))
    return out


def instance_from_matrix(M_colmajor: np.ndarray, mesh: int, material: int = 1) -> np.ndarray:
    """Instance::new (shared.rs:90-98) for one column-major 4x4 (float64 in, f32 out)."""
    inst = np.zeros((), dtype=abi.INSTANCE)
    t32 = np.asarray(M_colmajor, dtype=np.float32).reshape(16)
    inst["transform"] = t32
    Mi = np.linalg.inv(t32.reshape(4, 4).astype(np.float64).T).T
    inst["inv_transform"] = Mi.astype(np.float32).reshape(16)
    inst["mesh"] = mesh
    inst["material"] = material
    return inst


# --- triangle meshes for the BLAS builder ---------------------------------------------------

def plane_mesh(width=1.0, height=1.0):
    """make_plane_mesh (crates/pools/src/mesh/plane.rs:5-38)."""
    w, h = width / 2, height / 2
    v = np.array([[-w, 0, -h], [-w, 0, h], [w, 0, h], [w, 0, -h]], dtype=np.float32)
    i = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    return v, i


def plane_mesh_rot_x(width=1.0, height=1.0):
    """The second built-in mesh of MeshPool::new (crates/pools/src/mesh/mod.rs:269-272): the plane with every vertex
    multiplied by glam Mat3::from_rotation_x(-PI/2) = columns X, (0, cos, sin), (0, -sin, cos); Mat3 * v =
    (X*v.x + Y*v.y) + Z*v.z in f32 (glam 0.24 scalar/SSE2 order, from memory - unpinned like the other glam choices).
    sin/cos of the f32 angle are taken in f64 and rounded (cos(-PI/2 as f32) = -4.371139e-08, sin = -1)."""
    v, i = plane_mesh(width, height)
    a = np.float32(-np.pi / 2)
    c, s = np.float32(np.cos(np.float64(a))), np.float32(np.sin(np.float64(a)))
    X = np.array([1, 0, 0], dtype=np.float32)
    Y = np.array([0, c, s], dtype=np.float32)
    Z = np.array([0, -s, c], dtype=np.float32)
    out = np.empty_like(v)
    for k in range(len(v)):
        out[k] = (X * v[k, 0] + Y * v[k, 1]) + Z * v[k, 2]
    return out, i


def uv_sphere(radius=1.0, resolution=10):
    """make_uv_sphere (crates/pools/src/mesh/sphere.rs:6-66).  Vertices use libm sin/cos in
    f32 there; here they are computed in f64 and rounded, and the arrays are committed as
    fixtures rather than regenerated on the GPU box."""
    vside = 4 * resolution
    uside = vside * 2
    verts = []
    for vi in range(vside + 1):
        v = np.float32(vi) / np.float32(vside)
        for ui in range(uside + 1):
            u = np.float32(ui) / np.float32(uside)
            theta = 2.0 * math.pi * float(u) + math.pi
            phi = math.pi * float(v)
            verts.append([math.cos(theta) * math.sin(phi) * radius, -math.cos(phi) * radius,
                          math.sin(theta) * math.sin(phi) * radius])
    idx = []
    sc = uside
    for i in range(vside):
        k1r = i * (sc + 1)
        for j in range(sc):
            k1, k2 = j + k1r, j + k1r + sc + 1
            if i != 0:
                idx += [k1, k2, k1 + 1]
            if i != vside:  # always true, as in the reference (sphere.rs:52)
                idx += [k1 + 1, k2, k2 + 1]
    return np.asarray(verts, dtype=np.float32), np.asarray(idx, dtype=np.uint32)


def triangle_soup(n_tri=64, seed=SEED_BASE + 10):
    """The 64-triangle soup of src/bin/bvh_cpu.rs:39-52 (same distribution, seeded)."""
    u = uniform01(seed, 7, n_tri * 9).reshape(n_tri, 3, 3)
    base = u[:, 0] * 9.0 - np.array([5.0, 5.0, 0.0])
    v = np.stack([base, base + u[:, 1], base + u[:, 2]], axis=1).reshape(-1, 3)
    return v.astype(np.float32), np.arange(3 * n_tri, dtype=np.uint32)


def knot_mesh(n_u: int, n_v: int, seed=SEED_BASE + 20, displace=0.08):
    """Closed 'dragon-like' surface: a (2,3) torus-knot tube, n_u x n_v quads -> 2·n_u·n_v
    triangles, radially displaced by a smooth pseudo-noise so that no two triangles share a
    centroid (SURVEY.md §8a B7 keeps degenerate input out of parity runs).  Vertices are shared
    (indexed mesh, as tobj GPU_LOAD_OPTIONS produces: crates/app/src/models/mod.rs:24)."""
    uu = np.arange(n_u, dtype=np.float64) / n_u * 2 * math.pi
    vv = np.arange(n_v, dtype=np.float64) / n_v * 2 * math.pi
    p, q, Rk, rk = 2.0, 3.0, 2.0, 0.8

    def centre(t):
        r = Rk + rk * np.cos(q * t)
        return np.stack([r * np.cos(p * t), r * np.sin(p * t), -rk * np.sin(q * t)], axis=-1)

    c = centre(uu)
    d = centre(uu + 1e-4) - c
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ref = np.array([0.0, 0.0, 1.0])
    nrm = np.cross(d, ref)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    bi = np.cross(d, nrm)
    U, V = np.meshgrid(uu, vv, indexing="ij")
    ph = uniform01(seed, 3, 6) * 2 * math.pi
    noise = (np.sin(7 * U + 3 * V + ph[0]) * np.sin(5 * V - 2 * U + ph[1]) +
             0.5 * np.sin(23 * U + ph[2]) * np.sin(17 * V + ph[3]) +
             0.25 * np.sin(61 * U + 41 * V + ph[4]) + 0.125 * np.sin(131 * V - 97 * U + ph[5]))
    tube = 0.35 * (1.0 + displace * noise)
    P = (c[:, None, :] + tube[..., None] * (np.cos(V)[..., None] * nrm[:, None, :] +
                                            np.sin(V)[..., None] * bi[:, None, :]))
    verts = P.reshape(-1, 3).astype(np.float32)
    iu = np.arange(n_u)[:, None]
    iv = np.arange(n_v)[None, :]
    a = (iu * n_v + iv)
    b = (((iu + 1) % n_u) * n_v + iv)
    cc = (((iu + 1) % n_u) * n_v + (iv + 1) % n_v)
    dd = (iu * n_v + (iv + 1) % n_v)
    tris = np.stack([np.stack([a, b, cc], -1), np.stack([a, cc, dd], -1)], axis=2)
    return verts, tris.reshape(-1).astype(np.uint32)


def mesh_bounds(verts: np.ndarray):
    """calculate_bounds (crates/pools/src/mesh/mod.rs:22-27)."""
    return verts.min(axis=0), verts.max(axis=0)


def primary_rays(cam: np.ndarray, width: int, height: int) -> np.ndarray:
    """Per-pixel rays as src/bin/bvh_cpu.rs:71-83 builds them from clip_to_world."""
    M = cam["clip_to_world"].reshape(4, 4).astype(np.float64).T  # math matrix
    i = np.arange(width * height)
    x = (i % width) / width
    y = (i // height) / height
    x = (x - 0.5) * 2.0
    y = (y - 0.5) * -2.0
    ones = np.ones_like(x)
    vp = (M @ np.stack([x, y, ones, ones])).T
    vt = (M @ np.stack([x, y, 0 * ones, ones])).T
    eye = vp[:, :3] / vp[:, 3:4]
    d = vt[:, :3]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(width * height, dtype=abi.RAY)
    rays["eye"] = eye.astype(np.float32)
    rays["dir"] = d.astype(np.float32)
    return rays


def harness_scene(build_fn, big=(1024, 256), small_res=100):
    """The shape of the reference's GPU harness (src/bin/bvh_gpu.rs:107-131): one large mesh
    `rotY(pi/2) * T(0,2,0) * S(..)` and four small ones `T(+-8,+-8,0) * S(3)`.  bunny.obj / dragon.obj are
    not in the checkout: a knot mesh stands in for the dragon, uv-spheres for the bunnies.
    build_fn(vertices, indices) -> (nodes, permuted indices).  Returns (instances, mesh_infos,
    bvh_nodes, vertices, indices)."""
    def mat(tr, scale, roty=0.0):
        c, s_ = math.cos(roty), math.sin(roty)
        R = np.array([[c, 0, s_, 0], [0, 1, 0, 0], [-s_, 0, c, 0], [0, 0, 0, 1]], np.float64)
        T = np.eye(4)
        T[:3, 3] = tr
        return (R @ T @ np.diag([scale, scale, scale, 1.0])).T.reshape(16)      # column-major
    mesh_src = [knot_mesh(*big), uv_sphere(1.0, small_res)]
    V, I, B = [], [], []
    infos = np.zeros(len(mesh_src), dtype=abi.MESH_INFO)
    vo = bo = no = 0
    for k, (mv, mi) in enumerate(mesh_src):
        nodes_k, idx_k = build_fn(mv, mi)
        infos[k]["min"], infos[k]["max"] = mesh_bounds(mv)
        infos[k]["index_count"], infos[k]["base_index"] = len(idx_k), bo
        infos[k]["vertex_offset"], infos[k]["bvh_index"] = vo, no
        V.append(mv); I.append(idx_k); B.append(nodes_k)
        vo += len(mv); bo += len(idx_k); no += len(nodes_k)
    ext = float(max(np.abs(infos[0]["min"]).max(), np.abs(infos[0]["max"]).max()))
    insts = [instance_from_matrix(mat((0, 2, 0), 5.0 / ext, math.pi / 2), 0)]
    insts += [instance_from_matrix(mat((x, y, 0), 3.0), 1) for x, y in ((8, 8), (-8, 8), (8, -8), (-8, -8))]
    return np.array(insts, dtype=abi.INSTANCE), infos, np.concatenate(B), np.concatenate(V), np.concatenate(I)
