"""ctypes view of include/voidin_abi.h: numpy dtypes for the POD wire structs and the loader
for libvoidin_hip.so.

The struct layouts restate SURVEY.md §8a D1-D6 (reference: crates/components/src/shared.rs:29-75,
crates/components/src/lib.rs:99-107, crates/components/src/camera.rs:13-27,
crates/bvh/src/blas.rs:10-17, crates/bvh/src/tlas.rs:7-14).

There is NO CPU fallback: `load()` raises if the HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOIDIN_HIP_LIB") or os.path.join(_HERE, "csrc", "libvoidin_hip.so")

# --- wire structs ---------------------------------------------------------------------
INSTANCE = np.dtype([("transform", "<f4", (16,)), ("inv_transform", "<f4", (16,)),
                     ("mesh", "<u4"), ("material", "<u4"), ("junk", "<u4", (2,))])
MESH_INFO = np.dtype([("min", "<f4", (3,)), ("index_count", "<u4"), ("max", "<f4", (3,)),
                      ("base_index", "<u4"), ("vertex_offset", "<i4"), ("bvh_index", "<u4"),
                      ("junk", "<u4", (2,))])
DRAW = np.dtype([("vertex_count", "<u4"), ("instance_count", "<u4"), ("base_index", "<u4"),
                 ("vertex_offset", "<i4"), ("base_instance", "<u4")])
CAMERA = np.dtype([("view_position", "<f4", (4,)), ("projection", "<f4", (16,)),
                   ("view", "<f4", (16,)), ("clip_to_world", "<f4", (16,)),
                   ("prev_world_to_clip", "<f4", (16,)), ("frustum", "<f4", (4,)),
                   ("zfar", "<f4"), ("znear", "<f4"), ("jitter", "<f4", (2,)),
                   ("prev_jitter", "<f4", (2,)), ("_padding", "<f4", (2,))])
BVH_NODE = np.dtype([("min", "<f4", (3,)), ("left_first", "<u4"), ("max", "<f4", (3,)),
                     ("count", "<u4")])
TLAS_NODE = np.dtype([("min", "<f4", (3,)), ("left_right", "<u4"), ("max", "<f4", (3,)),
                      ("instance_idx", "<u4")])
TLAS_NODE_WIDE = np.dtype([("min", "<f4", (3,)), ("left", "<u4"), ("max", "<f4", (3,)),
                           ("right", "<u4"), ("instance_idx", "<u4"), ("_pad", "<u4", (3,))])
RAY = np.dtype([("eye", "<f4", (3,)), ("_pad0", "<f4"), ("dir", "<f4", (3,)), ("_pad1", "<f4")])
HIT = np.dtype([("dist", "<f4"), ("hit", "<u4"), ("instance", "<u4"), ("triangle", "<u4")])

assert INSTANCE.itemsize == 144 and MESH_INFO.itemsize == 48 and DRAW.itemsize == 20
assert CAMERA.itemsize == 320 and BVH_NODE.itemsize == 32 and TLAS_NODE.itemsize == 32
assert TLAS_NODE_WIDE.itemsize == 48 and RAY.itemsize == 32 and HIT.itemsize == 16

MAX_DIST = np.float32(1e30)
CULL_SPLIT_MIN = 2 << 20   # VdCtx default: vd_cull_compact / vd_cull_emit run their split form from this many instances
TLAS_MAX_INSTANCES = 32768

VD_OK = 0
VD_ERR_INVALID_ARG = -1
VD_ERR_HIP = -2
VD_ERR_DEGENERATE = -3
VD_ERR_TLAS_OVERFLOW = -4
VD_ERR_NO_DEVICE = -5
VD_ERR_STACK_OVERFLOW = -6
VD_ERR_OOM = -7
VD_ERR_COMM = -8
VD_DIST_ID_BYTES = 128
# VdOption (include/voidin_abi.h); the environment names are read by the PYTHON harness only (Context.apply_env_options),
# never by the library
OPTIONS = {"cull.split_min": 1, "cull.variant": 2, "tlas.index": 10, "tlas.index_min": 11, "tlas.phase2": 12, "tlas.refresh": 13,
           "tlas.groups": 14, "tlas.spin_limit": 15, "tlas.spec": 16, "tlas.profile": 17, "tlas.chain_lds": 18,
           "blas.wide_payload": 30, "trace.sort": 21, "trace.sort_min": 22, "trace.chunk": 23, "trace.yield": 24, "trace.waves": 25, "trace.auto_prepare": 27, "trace.tight_tlas": 28, "trace.fan": 29, "trace.fan_slots": 26}
OPTION_ENV = {"VD_SPLIT_MIN": "cull.split_min", "VD_CULL_VARIANT": "cull.variant", "VD_TLAS_INDEX": "tlas.index",
              "VD_TLAS_INDEX_MIN": "tlas.index_min", "VD_TLAS_PHASE2": "tlas.phase2", "VD_TLAS_REFRESH": "tlas.refresh",
              "VD_TLAS_GROUPS": "tlas.groups", "VD_TLAS_SPIN_LIMIT": "tlas.spin_limit", "VD_TLAS_SPEC": "tlas.spec",
              "VD_TLAS_PROFILE": "tlas.profile", "VD_TLAS_CHAIN_LDS": "tlas.chain_lds", "VD_TRACE_SORT": "trace.sort",
              "VD_TRACE_SORT_MIN": "trace.sort_min", "VD_TRACE_CHUNK": "trace.chunk", "VD_TRACE_YIELD": "trace.yield", "VD_TRACE_WAVES": "trace.waves", "VD_TRACE_AUTO_PREPARE": "trace.auto_prepare", "VD_TRACE_TIGHT_TLAS": "trace.tight_tlas", "VD_TRACE_FAN": "trace.fan"}

STATUS_NAMES = {0: "VD_OK", -1: "VD_ERR_INVALID_ARG", -2: "VD_ERR_HIP", -3: "VD_ERR_DEGENERATE",
                -4: "VD_ERR_TLAS_OVERFLOW", -5: "VD_ERR_NO_DEVICE", -6: "VD_ERR_STACK_OVERFLOW",
                -7: "VD_ERR_OOM", -8: "VD_ERR_COMM"}


class TraceScene(C.Structure):
    """VdTraceScene (include/voidin_abi.h) — the six storage buffers of the trace bind group
    (reference: crates/app/src/app.rs:255-287)."""
    _fields_ = [("tlas_nodes", C.c_void_p), ("n_tlas_nodes", C.c_uint32),
                ("instances", C.c_void_p), ("n_instances", C.c_uint32),
                ("meshes", C.c_void_p), ("n_meshes", C.c_uint32),
                ("bvh_nodes", C.c_void_p), ("n_bvh_nodes", C.c_uint32),
                ("vertices", C.c_void_p), ("n_vertices", C.c_uint32),
                ("indices", C.c_void_p), ("n_indices", C.c_uint32)]


class HizLayout(C.Structure):
    """VdHizLayout (include/voidin_abi.h, occlusion extension): sizes and texel offsets of the depth pyramid."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("n_levels", C.c_uint32), ("total_texels", C.c_uint32),
                ("level_offset", C.c_uint32 * 17), ("level_width", C.c_uint32 * 17), ("level_height", C.c_uint32 * 17)]


class DistInfo(C.Structure):
    """VdDistInfo (include/voidin_abi.h, multi-GPU exchange)."""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("n_total", C.c_uint32), ("shard_size", C.c_uint32),
                ("first_instance", C.c_uint32), ("n_local", C.c_uint32), ("mask_words_per_shard", C.c_uint32), ("id_bytes", C.c_uint32),
                ("d_mask", C.c_void_p), ("d_mask_all", C.c_void_p), ("d_mesh_ids", C.c_void_p), ("rccl_version", C.c_int32),
                ("_pad", C.c_int32), ("rccl_library", C.c_char * 128)]


class TraceAccelInfo(C.Structure):
    """VdTraceAccelInfo (include/voidin_abi.h)."""
    _fields_ = [("tight_tlas", C.c_uint32), ("n_tlas_nodes", C.c_uint32), ("tight_fallback_instances", C.c_uint32), ("_pad", C.c_uint32),
                ("triangle_bytes", C.c_uint64), ("d_tlas_nodes", C.c_void_p)]


class BvhBatchItem(C.Structure):
    """VdBvhBatchItem (include/voidin_abi.h, "Batched BLAS build")."""
    _fields_ = [("verts_xyz", C.c_void_p), ("indices_inout", C.c_void_p), ("out_nodes", C.c_void_p),
                ("n_vert", C.c_uint32), ("n_tri", C.c_uint32), ("node_cap", C.c_uint32),
                ("out_n_nodes", C.c_uint32), ("out_first_node", C.c_uint32), ("status", C.c_int32)]


class BvhBuildStats(C.Structure):
    """VdBvhBuildStats (include/voidin_abi.h, instrumentation)."""
    _fields_ = [("ms_precompute", C.c_float), ("ms_phase_a", C.c_float), ("ms_mid", C.c_float), ("ms_phase_b", C.c_float),
                ("ms_phase_c", C.c_float), ("levels_phase_a", C.c_uint32), ("n_top_nodes", C.c_uint32),
                ("n_mid_roots", C.c_uint32), ("n_small_roots", C.c_uint32), ("kernel_launches", C.c_uint32)]


_P = C.c_void_p
_U = C.c_uint32
_I = C.c_int

# name -> (restype, argtypes); every symbol include/voidin_abi.h declares
PROTOTYPES = {
    "vd_ctx_create": (_I, [_I, C.POINTER(_P)]),
    "vd_ctx_destroy": (_I, [_P]),
    "vd_ctx_set_stream": (_I, [_P, _P]),
    "vd_ctx_reset_stream": (_I, [_P]),
    "vd_ctx_synchronize": (_I, [_P]),
    "vd_ctx_set_option": (_I, [_P, _I, C.c_int64]),
    "vd_last_error": (C.c_char_p, [_P]),
    "vd_version": (C.c_char_p, []),
    "vd_cull_emit": (_I, [_P, _P, _P, _U, _P, _U, _P]),
    "vd_cull_emit_dev": (_I, [_P, _P, _P, _U, _P, _U, _P]),
    "vd_cull_compact": (_I, [_P, _P, _P, _U, _P, _U, _P, _P, _I]),
    "vd_cull_compact_dev": (_I, [_P, _P, _P, _U, _P, _U, _P, _P, _I]),
    "vd_cull_emit_shard_dev": (_I, [_P, _P, _P, _U, _P, _U, _U, _P]),
    "vd_cull_compact_shard_dev": (_I, [_P, _P, _P, _U, _P, _U, _U, _P, _P, _I]),
    "vd_cull_mask_dev": (_I, [_P, _P, _P, _U, _P, _U, _P]),
    "vd_expand_mask_dev": (_I, [_P, _P, _U, _U, _P, _U, _P, _U, _P, _P]),
    "vd_mask_to_indices_dev": (_I, [_P, _P, _U, _U, _P, _P]),
    "vd_indices_to_draws_dev": (_I, [_P, _P, _U, _P, _U, _U, _P, _U, _P]),
    "vd_compact_draws_dev": (_I, [_P, _P, _U, _P, _P]),
    "vd_bvh_build": (_I, [_P, _P, _U, _P, _U, _P, _U, _P]),
    "vd_bvh_build_dev": (_I, [_P, _P, _U, _P, _U, _P, _U, _P]),
    "vd_bvh_build_batch": (_I, [_P, _P, _U, _P, C.c_uint64, _U, _P]),
    "vd_bvh_build_batch_dev": (_I, [_P, _P, _U, _P, C.c_uint64, _U, _P]),
    "vd_tlas_build": (_I, [_P, _P, _U, _P, _U, _P]),
    "vd_tlas_build_dev": (_I, [_P, _P, _U, _P, _U, _P]),
    "vd_tlas_build_wide": (_I, [_P, _P, _U, _P, _U, _P]),
    "vd_tlas_build_wide_dev": (_I, [_P, _P, _U, _P, _U, _P]),
    "vd_tlas_refit": (_I, [_P, _P, _U, _P, _U, _P]),
    "vd_tlas_refit_dev": (_I, [_P, _P, _U, _P, _U, _P]),
    "vd_tlas_refit_wide_dev": (_I, [_P, _P, _U, _P, _U, _P]),
    "vd_trace": (_I, [_P, C.POINTER(TraceScene), _P, _U, _P]),
    "vd_trace_dev": (_I, [_P, C.POINTER(TraceScene), _P, _U, _P]),
    "vd_import_external_buffer": (_I, [_P, C.c_int, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "vd_release_external_buffer": (_I, [_P, _P]),
    "vd_import_external_semaphore": (_I, [_P, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "vd_wait_external_semaphore_async": (_I, [_P, _P, C.c_uint64]),
    "vd_signal_external_semaphore_async": (_I, [_P, _P, C.c_uint64]),
    "vd_release_external_semaphore": (_I, [_P, _P]),
    "vd_wait_value32_async": (_I, [_P, _P, _U]),
    "vd_write_value32_async": (_I, [_P, _P, _U]),
    "vd_host_callback_async": (_I, [_P, C.CFUNCTYPE(None, C.c_void_p), _P]),
    "vd_trace_any_dev": (_I, [_P, C.POINTER(TraceScene), _P, _U, _P]),
    "vd_trace_prepare_dev": (_I, [_P, C.POINTER(TraceScene), C.POINTER(_P)]),
    "vd_trace_release": (_I, [_P, _P]),
    "vd_trace_accel_info": (_I, [_P, _P]),
    "vd_trace_accel_update_dev": (_I, [_P, _P]),
    "vd_trace_prepared_dev": (_I, [_P, _P, _P, _U, _P]),
    "vd_trace_any_prepared_dev": (_I, [_P, _P, _P, _U, _P]),
    "vd_shadow_rays_dev": (_I, [_P, _P, _P, _U, C.POINTER(C.c_float), _P]),
    "vd_primary_rays_dev": (_I, [_P, _P, _U, _U, _P]),
    "vd_traverse_iter_dev": (_I, [_P, _P, _U, _P, _P, _P, _U, _P]),
    "vd_traverse_dev": (_I, [_P, _P, _U, _P, _P, _P, _U, C.c_float, _P]),
    "vd_traverse": (_I, [_P, _P, _U, _P, _U, _P, _U, _P, _U, C.c_float, _P]),
    "vd_primary_rays": (_I, [_P, _P, _U, _U, _P]),
    "vd_traverse_iter": (_I, [_P, _P, _U, _P, _U, _P, _U, _P, _U, _P]),
    "vd_hiz_layout": (_I, [_U, _U, _P]),
    "vd_hiz_build_dev": (_I, [_P, _P, _U, _U, _P]),
    "vd_occlusion_mask_dev": (_I, [_P, _P, _P, _U, _P, _U, _P, _U, _U, _P, _P]),
    "vd_compute_update_dev": (_I, [_P, _P, _U, _P, _U, C.c_float, C.c_float, _I]),
    "vd_ctx_set_timing": (_I, [_P, _I]),
    "vd_last_gpu_ms": (C.c_float, [_P]),
    "vd_last_gpu_ms_stage": (C.c_float, [_P, _I]),
    "vd_bvh_last_build_stats": (_I, [_P, C.POINTER(BvhBuildStats)]),
    "vd_dist_unique_id": (_I, [_P]),
    "vd_dist_create": (_I, [_P, _P, _I, _I, C.POINTER(_P)]),
    "vd_dist_destroy": (_I, [_P]),
    "vd_dist_info": (_I, [_P, C.POINTER(DistInfo)]),
    "vd_dist_set_scene_dev": (_I, [_P, _P, _U, _U, _U]),
    "vd_dist_step_full_dev": (_I, [_P, _P, _P, _U, _P, _P, _P]),
    "vd_dist_step_draws_dev": (_I, [_P, _P, _P, _U, _P, _P, _P]),
    "vd_dist_step_indices_dev": (_I, [_P, _P, _P, _U, _P, _P, _P]),
    "vd_dist_allgather_dev": (_I, [_P, _P, _P, C.c_uint64]),
}

_lib = None


class VoidinHipMissing(RuntimeError):
    pass


def load(path: str | None = None) -> C.CDLL:
    """dlopen libvoidin_hip.so and bind every prototype.  Raises (never falls back) when the
    library or any declared symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise VoidinHipMissing(
            f"{p} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C voidin_amd/csrc`; there is no CPU fallback")
    # A process must hold ONE HIP runtime.  PyTorch ships its own libamdhip64 and loads it by path; if the system copy
    # this library links against is loaded first, torch later finds "No HIP GPUs".  Loading torch first lets the
    # dynamic loader satisfy our DT_NEEDED with the copy that is already in the process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def ptr(a) -> int:
    """Raw address of a numpy array / torch tensor / int."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError(type(a))
