"""ctypes bindings of oracle/libvd_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker (see oracle/vd_oracle.h; parity unpinned).  The product package
voidin_amd never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from voidin_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VD_ORACLE_LIB") or os.path.join(_HERE, "libvd_oracle.so")   # VD_ORACLE_LIB: e.g. the `make asan` build
_lib = None
_P, _U, _I = C.c_void_p, C.c_uint32, C.c_int


def build() -> None:
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        protos = {
            "vd_ref_cull_emit": (_I, [_P, _P, _U, _P, _U, _P, _I]),
            "vd_ref_compact": (_I, [_P, _U, _P, _P, _I]),
            "vd_ref_cull_margins": (_I, [_P, _P, _U, _P, _U, _P, _P, _P]),
            "vd_ref_bvh_build": (_I, [_P, _U, _P, _U, _P, _U, _P]),
            "vd_ref_partition_shuffle": (_U, [_P, _P, _U, _U, C.c_float]),
            "vd_ref_tlas_build": (_I, [_P, _U, _P, _U, _P]),
            "vd_ref_tlas_build_wide": (_I, [_P, _U, _P, _U, _P]),
            "vd_ref_tlas_refit": (_I, [_P, _U, _P, _U, _P]),
            "vd_ref_tlas_refit_wide": (_I, [_P, _U, _P, _U, _P]),
            "vd_ref_trace": (_I, [C.POINTER(abi.TraceScene), _P, _U, _P, _P, _I]),
            "vd_ref_traverse_iter": (_I, [_P, _U, _P, _P, _P, _U, _P]),
            "vd_ref_traverse": (_I, [_P, _U, _P, _P, _P, _U, C.c_float, _P]),
            "vd_ref_shadow_rays": (_I, [_P, _P, _U, _P, _P]),
            "vd_ref_primary_rays": (_I, [_P, _U, _U, _P]),
            "vd_ref_hiz_layout": (_I, [_U, _U, _P]),
            "vd_ref_hiz_build": (_I, [_P, _U, _U, _P]),
            "vd_ref_occlusion_mask": (_I, [_P, _P, _U, _P, _U, _P, _U, _U, _P, _P]),
            "vd_ref_compute_update": (_I, [_P, _U, _P, _U, C.c_float, C.c_float, _I]),
            "vd_ref_version": (C.c_char_p, []),
        }
        for name, (res, args) in protos.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle returned {abi.STATUS_NAMES.get(code, code)}")
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def cull_emit(camera, meshes, instances, threads=1):
    camera, meshes, instances = _c(camera, abi.CAMERA), _c(meshes, abi.MESH_INFO), _c(instances, abi.INSTANCE)
    out = np.zeros(len(instances), dtype=abi.DRAW)
    _chk(load().vd_ref_cull_emit(camera.ctypes.data, meshes.ctypes.data, len(meshes),
                                 instances.ctypes.data, len(instances), out.ctypes.data, threads))
    return out


def cull_margins(camera, meshes, instances):
    camera, meshes, instances = _c(camera, abi.CAMERA), _c(meshes, abi.MESH_INFO), _c(instances, abi.INSTANCE)
    n = len(instances)
    mx, my, r = (np.zeros(n, dtype=np.float32) for _ in range(3))
    _chk(load().vd_ref_cull_margins(camera.ctypes.data, meshes.ctypes.data, len(meshes),
                                    instances.ctypes.data, n, mx.ctypes.data, my.ctypes.data, r.ctypes.data))
    return mx, my, r


def compute_update(indices, instances, time, dt, fix_inverse=False):
    indices = _c(indices, np.uint32)
    inst = np.array(instances, dtype=abi.INSTANCE, copy=True)
    _chk(load().vd_ref_compute_update(indices.ctypes.data, len(indices), inst.ctypes.data, len(inst), float(time), float(dt), int(fix_inverse)))
    return inst


def compact(draws, pad_tail=False):
    draws = _c(draws, abi.DRAW)
    out = np.zeros(len(draws), dtype=abi.DRAW)
    cnt = C.c_uint32(0)
    _chk(load().vd_ref_compact(draws.ctypes.data, len(draws), out.ctypes.data, C.addressof(cnt), int(pad_tail)))
    return out, cnt.value


def bvh_build(verts, indices):
    """Returns (nodes[:n_nodes], permuted indices). Raises OracleError(VD_ERR_DEGENERATE)."""
    verts = _c(verts, np.float32).reshape(-1, 3)
    idx = np.array(indices, dtype=np.uint32).reshape(-1).copy()
    n_tri = len(idx) // 3
    nodes = np.zeros(2 * n_tri, dtype=abi.BVH_NODE)
    n_nodes = C.c_uint32(0)
    _chk(load().vd_ref_bvh_build(verts.ctypes.data, len(verts), idx.ctypes.data, n_tri,
                                 nodes.ctypes.data, len(nodes), C.addressof(n_nodes)))
    return nodes[:n_nodes.value].copy(), idx


def partition_shuffle(keys_by_id, ids, start, count, pos):
    keys = _c(keys_by_id, np.float32)
    ids = np.array(ids, dtype=np.uint32).copy()
    piv = load().vd_ref_partition_shuffle(keys.ctypes.data, ids.ctypes.data, start, count, float(pos))
    return piv, ids


def tlas_build(instances, meshes, wide=False):
    instances, meshes = _c(instances, abi.INSTANCE), _c(meshes, abi.MESH_INFO)
    n = len(instances)
    out = np.zeros(2 * n + 1, dtype=abi.TLAS_NODE_WIDE if wide else abi.TLAS_NODE)
    fn = load().vd_ref_tlas_build_wide if wide else load().vd_ref_tlas_build
    _chk(fn(instances.ctypes.data, n, meshes.ctypes.data, len(meshes), out.ctypes.data))
    return out


def tlas_refit(instances, meshes, nodes):
    instances, meshes = _c(instances, abi.INSTANCE), _c(meshes, abi.MESH_INFO)
    wide = nodes.dtype == abi.TLAS_NODE_WIDE
    nodes = np.array(nodes, copy=True)
    fn = load().vd_ref_tlas_refit_wide if wide else load().vd_ref_tlas_refit
    _chk(fn(instances.ctypes.data, len(instances), meshes.ctypes.data, len(meshes), nodes.ctypes.data))
    return nodes


def make_scene(tlas_nodes, instances, meshes, bvh_nodes, vertices, indices):
    """Host-pointer VdTraceScene; returns (scene, keepalive)."""
    arrs = [_c(tlas_nodes, abi.TLAS_NODE), _c(instances, abi.INSTANCE), _c(meshes, abi.MESH_INFO),
            _c(bvh_nodes, abi.BVH_NODE), _c(vertices, np.float32).reshape(-1), _c(indices, np.uint32).reshape(-1)]
    s = abi.TraceScene()
    s.tlas_nodes, s.n_tlas_nodes = arrs[0].ctypes.data, len(arrs[0])
    s.instances, s.n_instances = arrs[1].ctypes.data, len(arrs[1])
    s.meshes, s.n_meshes = arrs[2].ctypes.data, len(arrs[2])
    s.bvh_nodes, s.n_bvh_nodes = arrs[3].ctypes.data, len(arrs[3])
    s.vertices, s.n_vertices = arrs[4].ctypes.data, len(arrs[4]) // 3
    s.indices, s.n_indices = arrs[5].ctypes.data, len(arrs[5])
    return s, arrs


def trace(scene_arrays, rays, threads=1):
    """scene_arrays = (tlas_nodes, instances, meshes, bvh_nodes, vertices, indices)."""
    s, keep = make_scene(*scene_arrays)
    rays = _c(rays, abi.RAY)
    out = np.zeros(len(rays), dtype=abi.HIT)
    ms = C.c_uint32(0)
    _chk(load().vd_ref_trace(C.byref(s), rays.ctypes.data, len(rays), out.ctypes.data, C.addressof(ms), threads))
    return out, ms.value


def shadow_rays(positions, normals, light_position):
    pos, nor = _c(positions, np.float32).reshape(-1, 3), _c(normals, np.float32).reshape(-1, 3)
    lp = _c(light_position, np.float32).reshape(3)
    out = np.zeros(len(pos), dtype=abi.RAY)
    _chk(load().vd_ref_shadow_rays(pos.ctypes.data, nor.ctypes.data, len(pos), lp.ctypes.data, out.ctypes.data))
    return out


def primary_rays(camera, width, height):
    cam = _c(camera, abi.CAMERA).reshape(1)
    out = np.zeros(width * height, dtype=abi.RAY)
    _chk(load().vd_ref_primary_rays(cam.ctypes.data, width, height, out.ctypes.data))
    return out


def hiz_layout(width, height):
    L = abi.HizLayout()
    _chk(load().vd_ref_hiz_layout(width, height, C.byref(L)))
    return L


def hiz_build(depth):
    depth = _c(depth, np.float32)
    h, w = depth.shape
    out = np.zeros(hiz_layout(w, h).total_texels, dtype=np.float32)
    _chk(load().vd_ref_hiz_build(depth.ctypes.data, w, h, out.ctypes.data))
    return out


def occlusion_mask(camera, meshes, instances, pyramid, width, height, mask_in):
    cam, meshes, instances = _c(camera, abi.CAMERA).reshape(1), _c(meshes, abi.MESH_INFO), _c(instances, abi.INSTANCE)
    pyramid, mask_in = _c(pyramid, np.float32), _c(mask_in, np.uint64)
    out = np.zeros_like(mask_in)
    _chk(load().vd_ref_occlusion_mask(cam.ctypes.data, meshes.ctypes.data, len(meshes), instances.ctypes.data, len(instances),
                                      pyramid.ctypes.data, width, height, mask_in.ctypes.data, out.ctypes.data))
    return out


def traverse_iter(nodes, verts, indices, rays):
    nodes, verts = _c(nodes, abi.BVH_NODE), _c(verts, np.float32).reshape(-1)
    indices, rays = _c(indices, np.uint32).reshape(-1), _c(rays, abi.RAY)
    out = np.zeros(len(rays), dtype=np.float32)
    _chk(load().vd_ref_traverse_iter(nodes.ctypes.data, len(nodes), verts.ctypes.data,
                                     indices.ctypes.data, rays.ctypes.data, len(rays), out.ctypes.data))
    return out


def traverse_recursive(nodes, verts, indices, rays, t0=1e30):
    """Bvh::traverse (blas.rs:211-245, R3; dead code in the reference): Hit(t) -> t (t0 itself if nothing nearer), Miss -> -1."""
    nodes, verts = _c(nodes, abi.BVH_NODE), _c(verts, np.float32).reshape(-1)
    indices, rays = _c(indices, np.uint32).reshape(-1), _c(rays, abi.RAY)
    out = np.zeros(len(rays), dtype=np.float32)
    _chk(load().vd_ref_traverse(nodes.ctypes.data, len(nodes), verts.ctypes.data, indices.ctypes.data, rays.ctypes.data, len(rays),
                                float(t0), out.ctypes.data))
    return out
