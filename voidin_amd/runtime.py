"""Host-side mirror of the reference's pass-graph / builder API over the C ABI.

Names follow the reference:
  EmitDraws.record      <- crates/app/src/pass/visibility.rs:233-254 (`impl Pass for EmitDraws`)
  BvhBuilder(...).build <- crates/bvh/src/blas.rs:51-103
  Tlas.build            <- crates/bvh/src/tlas.rs:31-85
  traverse_tlas         <- shaders/utils/bvh.wgsl:89-123

torch is plumbing only (device memory + streams); all compute is in libvoidin_hip.so.
There is no CPU fallback: constructing a Context without the HIP library or a gfx950 GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi


class VoidinError(RuntimeError):
    def __init__(self, code: int, msg: str = ""):
        super().__init__(f"{abi.STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


def _as_u8_tensor(arr: np.ndarray, device):
    import torch
    a = np.ascontiguousarray(arr)
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(device)


def to_numpy(t, dtype: np.dtype) -> np.ndarray:
    """uint8 device tensor -> structured numpy array."""
    return t.detach().cpu().numpy().view(dtype)


class Context:
    """One VdCtx bound to a device and (by default) torch's current HIP stream."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        self.lib = abi.load()
        h = C.c_void_p()
        rc = self.lib.vd_ctx_create(device, C.byref(h))
        if rc != abi.VD_OK:
            raise VoidinError(rc, "vd_ctx_create failed (need a gfx950 GPU; there is no CPU fallback)")
        self.h = h
        self.device = device
        import torch
        self.torch_device = torch.device("cuda", device)
        if use_torch_stream:
            self.set_stream(torch.cuda.current_stream(self.torch_device).cuda_stream)
        self.apply_env_options()

    def set_option(self, name: str, value: int | None):
        """vd_ctx_set_option (include/voidin_abi.h VdOption) by name, e.g. "tlas.spin_limit"; None = default."""
        dflt = 0 if name == "cull.variant" else -1            # the variant is a signed id taken as is (0 = default)
        self._chk(self.lib.vd_ctx_set_option(self.h, abi.OPTIONS[name], dflt if value is None else int(value)))

    def apply_env_options(self):
        """A/B scripts under tools/ select variants with VD_* environment variables; the LIBRARY reads none of them -
        this harness forwards them as per-context options when a Context is made."""
        import os
        for env, name in abi.OPTION_ENV.items():
            if os.environ.get(env) not in (None, ""):
                self.set_option(name, int(os.environ[env]))

    # -- plumbing -------------------------------------------------------------------------
    def _chk(self, rc: int):
        if rc != abi.VD_OK:
            raise VoidinError(rc, self.lib.vd_last_error(self.h).decode())

    def set_stream(self, hip_stream: int | None):
        self._chk(self.lib.vd_ctx_set_stream(self.h, hip_stream))

    def synchronize(self):
        self._chk(self.lib.vd_ctx_synchronize(self.h))

    def set_timing(self, enabled: bool):
        self._chk(self.lib.vd_ctx_set_timing(self.h, int(enabled)))

    def last_gpu_ms(self) -> float:
        return float(self.lib.vd_last_gpu_ms(self.h))

    def last_gpu_ms_stage(self, stage: int) -> float:
        return float(self.lib.vd_last_gpu_ms_stage(self.h, stage))

    def close(self):
        if getattr(self, "h", None):
            self.lib.vd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, arr: np.ndarray):
        return _as_u8_tensor(arr, self.torch_device)

    def empty(self, nbytes: int):
        import torch
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.torch_device)

    # -- cull / emit (device pointers) ------------------------------------------------------
    def cull_emit_dev(self, camera: np.ndarray, d_meshes, n_mesh, d_inst, n_inst, d_out, first_instance=0):
        cam = np.ascontiguousarray(camera, dtype=abi.CAMERA)
        self._chk(self.lib.vd_cull_emit_shard_dev(self.h, cam.ctypes.data, abi.ptr(d_meshes), n_mesh,
                                                  abi.ptr(d_inst), n_inst, first_instance, abi.ptr(d_out)))

    def cull_compact_dev(self, camera: np.ndarray, d_meshes, n_mesh, d_inst, n_inst, d_out, d_count,
                         pad_tail: bool = False, first_instance=0):
        cam = np.ascontiguousarray(camera, dtype=abi.CAMERA)
        self._chk(self.lib.vd_cull_compact_shard_dev(self.h, cam.ctypes.data, abi.ptr(d_meshes), n_mesh,
                                                     abi.ptr(d_inst), n_inst, first_instance, abi.ptr(d_out),
                                                     abi.ptr(d_count), int(pad_tail)))

    def cull_mask_dev(self, camera: np.ndarray, d_meshes, n_mesh, d_inst, n_inst, d_mask):
        cam = np.ascontiguousarray(camera, dtype=abi.CAMERA)
        self._chk(self.lib.vd_cull_mask_dev(self.h, cam.ctypes.data, abi.ptr(d_meshes), n_mesh, abi.ptr(d_inst), n_inst,
                                            abi.ptr(d_mask)))

    def expand_mask_dev(self, d_mask, n_total, shard_size, d_mesh_ids, d_meshes, n_mesh, d_out, d_count, id_bytes=None):
        if id_bytes is None:
            id_bytes = d_mesh_ids.element_size() if hasattr(d_mesh_ids, "element_size") else 4
        self._chk(self.lib.vd_expand_mask_dev(self.h, abi.ptr(d_mask), n_total, shard_size, abi.ptr(d_mesh_ids), id_bytes,
                                              abi.ptr(d_meshes), n_mesh, abi.ptr(d_out), abi.ptr(d_count)))

    def mask_to_indices_dev(self, d_mask, n_inst, first_instance, d_out_indices, d_count):
        self._chk(self.lib.vd_mask_to_indices_dev(self.h, abi.ptr(d_mask), n_inst, first_instance, abi.ptr(d_out_indices), abi.ptr(d_count)))

    def indices_to_draws_dev(self, d_indices, n_indices, d_mesh_ids, n_total, d_meshes, n_mesh, d_out, id_bytes=None):
        if id_bytes is None:
            id_bytes = d_mesh_ids.element_size() if hasattr(d_mesh_ids, "element_size") else 4
        self._chk(self.lib.vd_indices_to_draws_dev(self.h, abi.ptr(d_indices), n_indices, abi.ptr(d_mesh_ids), id_bytes, n_total,
                                                   abi.ptr(d_meshes), n_mesh, abi.ptr(d_out)))

    def compact_draws_dev(self, d_in, n, d_out, d_count):
        self._chk(self.lib.vd_compact_draws_dev(self.h, abi.ptr(d_in), n, abi.ptr(d_out), abi.ptr(d_count)))

    def compute_update_dev(self, d_indices, n_indices, d_instances, n_instances, time, dt, fix_inverse=False):
        self._chk(self.lib.vd_compute_update_dev(self.h, abi.ptr(d_indices), n_indices, abi.ptr(d_instances), n_instances,
                                                 float(time), float(dt), int(fix_inverse)))

    # -- ordering through a shared word / the host (voidin_abi.h: "Ordering that WORKS on this platform") ------
    def wait_value32(self, d_word_ptr: int, value: int):
        self._chk(self.lib.vd_wait_value32_async(self.h, d_word_ptr, value))

    def write_value32(self, d_word_ptr: int, value: int):
        self._chk(self.lib.vd_write_value32_async(self.h, d_word_ptr, value))

    def host_callback(self, fn, user=None):
        """fn: a ctypes CFUNCTYPE(None, c_void_p) object the CALLER keeps alive until it has run."""
        self._chk(self.lib.vd_host_callback_async(self.h, fn, user))

    # -- cull / emit (host arrays) ----------------------------------------------------------
    def cull_emit(self, camera, meshes, instances) -> np.ndarray:
        cam = np.ascontiguousarray(camera, dtype=abi.CAMERA)
        meshes = np.ascontiguousarray(meshes, dtype=abi.MESH_INFO)
        instances = np.ascontiguousarray(instances, dtype=abi.INSTANCE)
        out = np.zeros(len(instances), dtype=abi.DRAW)
        self._chk(self.lib.vd_cull_emit(self.h, cam.ctypes.data, meshes.ctypes.data, len(meshes),
                                        instances.ctypes.data, len(instances), out.ctypes.data))
        return out

    def cull_compact(self, camera, meshes, instances, pad_tail=False):
        cam = np.ascontiguousarray(camera, dtype=abi.CAMERA)
        meshes = np.ascontiguousarray(meshes, dtype=abi.MESH_INFO)
        instances = np.ascontiguousarray(instances, dtype=abi.INSTANCE)
        out = np.zeros(len(instances), dtype=abi.DRAW)
        out.view(np.uint8)[:] = 0xAB  # poison: only [0,count) (or everything with pad_tail) is defined
        cnt = C.c_uint32(0)
        self._chk(self.lib.vd_cull_compact(self.h, cam.ctypes.data, meshes.ctypes.data, len(meshes),
                                           instances.ctypes.data, len(instances), out.ctypes.data,
                                           C.addressof(cnt), int(pad_tail)))
        return out, cnt.value

    # -- BLAS -------------------------------------------------------------------------------
    def bvh_build(self, verts, indices):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        idx = np.array(indices, dtype=np.uint32).reshape(-1).copy()
        n_tri = len(idx) // 3
        nodes = np.zeros(max(2 * n_tri, 2), dtype=abi.BVH_NODE)
        n_nodes = C.c_uint32(0)
        self._chk(self.lib.vd_bvh_build(self.h, verts.ctypes.data, len(verts), idx.ctypes.data, n_tri,
                                        nodes.ctypes.data, len(nodes), C.addressof(n_nodes)))
        return nodes[:n_nodes.value].copy(), idx

    def bvh_build_dev(self, d_verts, n_vert, d_indices, n_tri, d_nodes, node_cap) -> int:
        n_nodes = C.c_uint32(0)
        self._chk(self.lib.vd_bvh_build_dev(self.h, abi.ptr(d_verts), n_vert, abi.ptr(d_indices), n_tri,
                                            abi.ptr(d_nodes), node_cap, C.addressof(n_nodes)))
        return n_nodes.value

    def bvh_build_batch(self, meshes, packed=True):
        """K meshes in ONE build (vd_bvh_build_batch; MeshPool::add for a whole scene, mesh/mod.rs:309-351).
        meshes: [(vertices (V,3) f32, indices (3T,) u32)].  packed: one shared node array, each mesh's nodes behind the
        previous mesh's -> (nodes, [(first_node, n_nodes, permuted indices)]); else [(nodes, permuted indices)]."""
        K = len(meshes)
        items = (abi.BvhBatchItem * K)()
        keep = []
        for m, (v, i) in enumerate(meshes):
            v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
            idx = np.array(i, dtype=np.uint32).reshape(-1).copy()
            n_tri = len(idx) // 3
            own = None if packed else np.zeros(max(2 * n_tri, 2), dtype=abi.BVH_NODE)
            keep.append((v, idx, own))
            items[m].verts_xyz, items[m].indices_inout = v.ctypes.data, idx.ctypes.data
            items[m].out_nodes = None if packed else own.ctypes.data
            items[m].n_vert, items[m].n_tri, items[m].node_cap = len(v), n_tri, 0 if packed else len(own)
        cap = sum(max(2 * (len(k[1]) // 3), 2) for k in keep) if packed else 0
        shared = np.zeros(max(cap, 1), dtype=abi.BVH_NODE)
        end = C.c_uint32(0)
        self._chk(self.lib.vd_bvh_build_batch(self.h, C.addressof(items), K, shared.ctypes.data if packed else None, cap, 0, C.addressof(end)))
        if packed:
            return shared[: end.value].copy(), [(int(items[m].out_first_node), int(items[m].out_n_nodes), keep[m][1]) for m in range(K)]
        return [(keep[m][2][: items[m].out_n_nodes].copy(), keep[m][1]) for m in range(K)]

    def bvh_build_batch_dev(self, items, n_items, d_packed=None, packed_cap=0, packed_first=0) -> int:
        """items: (abi.BvhBatchItem * K) with device pointers; returns one past the last packed node."""
        end = C.c_uint32(0)
        self._chk(self.lib.vd_bvh_build_batch_dev(self.h, C.addressof(items), n_items, abi.ptr(d_packed) if d_packed is not None else None,
                                                  packed_cap, packed_first, C.addressof(end)))
        return end.value

    def bvh_last_build_stats(self) -> dict:
        st = abi.BvhBuildStats()
        self._chk(self.lib.vd_bvh_last_build_stats(self.h, C.byref(st)))
        return {k: (round(getattr(st, k), 3) if t is C.c_float else int(getattr(st, k))) for k, t in st._fields_}

    # -- TLAS -------------------------------------------------------------------------------
    def tlas_build(self, instances, meshes, wide=False) -> np.ndarray:
        instances = np.ascontiguousarray(instances, dtype=abi.INSTANCE)
        meshes = np.ascontiguousarray(meshes, dtype=abi.MESH_INFO)
        out = np.zeros(2 * len(instances) + 1, dtype=abi.TLAS_NODE_WIDE if wide else abi.TLAS_NODE)
        fn = self.lib.vd_tlas_build_wide if wide else self.lib.vd_tlas_build
        self._chk(fn(self.h, instances.ctypes.data, len(instances), meshes.ctypes.data, len(meshes),
                     out.ctypes.data))
        return out

    def tlas_refit(self, instances, meshes, nodes) -> np.ndarray:
        instances = np.ascontiguousarray(instances, dtype=abi.INSTANCE)
        meshes = np.ascontiguousarray(meshes, dtype=abi.MESH_INFO)
        nodes = np.array(nodes, dtype=abi.TLAS_NODE, copy=True)
        self._chk(self.lib.vd_tlas_refit(self.h, instances.ctypes.data, len(instances), meshes.ctypes.data,
                                         len(meshes), nodes.ctypes.data))
        return nodes

    def tlas_build_dev(self, d_inst, n, d_meshes, n_mesh, d_nodes, wide=False):
        fn = self.lib.vd_tlas_build_wide_dev if wide else self.lib.vd_tlas_build_dev
        self._chk(fn(self.h, abi.ptr(d_inst), n, abi.ptr(d_meshes), n_mesh, abi.ptr(d_nodes)))

    def tlas_refit_dev(self, d_inst, n, d_meshes, n_mesh, d_nodes, wide=False):
        fn = self.lib.vd_tlas_refit_wide_dev if wide else self.lib.vd_tlas_refit_dev
        self._chk(fn(self.h, abi.ptr(d_inst), n, abi.ptr(d_meshes), n_mesh, abi.ptr(d_nodes)))

    # -- traversal ----------------------------------------------------------------------------
    def trace(self, scene_arrays, rays) -> np.ndarray:
        """scene_arrays = (tlas_nodes, instances, meshes, bvh_nodes, vertices, indices), host."""
        dts = [abi.TLAS_NODE, abi.INSTANCE, abi.MESH_INFO, abi.BVH_NODE, np.float32, np.uint32]
        arrs = [np.ascontiguousarray(a, dtype=d).reshape(-1) for a, d in zip(scene_arrays, dts)]
        s = abi.TraceScene()
        s.tlas_nodes, s.n_tlas_nodes = arrs[0].ctypes.data, len(arrs[0])
        s.instances, s.n_instances = arrs[1].ctypes.data, len(arrs[1])
        s.meshes, s.n_meshes = arrs[2].ctypes.data, len(arrs[2])
        s.bvh_nodes, s.n_bvh_nodes = arrs[3].ctypes.data, len(arrs[3])
        s.vertices, s.n_vertices = arrs[4].ctypes.data, len(arrs[4]) // 3
        s.indices, s.n_indices = arrs[5].ctypes.data, len(arrs[5])
        rays = np.ascontiguousarray(rays, dtype=abi.RAY)
        out = np.zeros(len(rays), dtype=abi.HIT)
        self._chk(self.lib.vd_trace(self.h, C.byref(s), rays.ctypes.data, len(rays), out.ctypes.data))
        return out


    class DeviceScene:
        """Device-resident copy of a trace scene; keeps the tensors alive next to the VdTraceScene."""

        def __init__(self, ctx, scene_arrays):
            dts = [abi.TLAS_NODE, abi.INSTANCE, abi.MESH_INFO, abi.BVH_NODE, np.float32, np.uint32]
            arrs = [np.ascontiguousarray(a, dtype=d).reshape(-1) for a, d in zip(scene_arrays, dts)]
            self.tensors = [ctx.upload(a) for a in arrs]
            s = abi.TraceScene()
            s.tlas_nodes, s.n_tlas_nodes = self.tensors[0].data_ptr(), len(arrs[0])
            s.instances, s.n_instances = self.tensors[1].data_ptr(), len(arrs[1])
            s.meshes, s.n_meshes = self.tensors[2].data_ptr(), len(arrs[2])
            s.bvh_nodes, s.n_bvh_nodes = self.tensors[3].data_ptr(), len(arrs[3])
            s.vertices, s.n_vertices = self.tensors[4].data_ptr(), len(arrs[4]) // 3
            s.indices, s.n_indices = self.tensors[5].data_ptr(), len(arrs[5])
            self.struct = s

    def device_scene(self, scene_arrays) -> "Context.DeviceScene":
        return Context.DeviceScene(self, scene_arrays)

    def trace_dev(self, scene: "Context.DeviceScene", d_rays, n_rays, d_out):
        self._chk(self.lib.vd_trace_dev(self.h, C.byref(scene.struct), abi.ptr(d_rays), n_rays, abi.ptr(d_out)))

    def trace_any_dev(self, scene: "Context.DeviceScene", d_rays, n_rays, d_out_hit):
        """Occlusion query: d_out_hit[i] (u32) = 1 iff ray i hits anything (raytraced_shadows.wgsl:97-102)."""
        self._chk(self.lib.vd_trace_any_dev(self.h, C.byref(scene.struct), abi.ptr(d_rays), n_rays, abi.ptr(d_out_hit)))

    class TraceAccel:
        """vd_trace_prepare_dev: per-scene de-indexed leaf triangles (include/voidin_abi.h); keeps the DeviceScene alive."""

        def __init__(self, ctx: "Context", scene: "Context.DeviceScene"):
            self.ctx, self.scene = ctx, scene
            h = C.c_void_p()
            ctx._chk(ctx.lib.vd_trace_prepare_dev(ctx.h, C.byref(scene.struct), C.byref(h)))
            self.h = h

        def info(self) -> dict:
            st = abi.TraceAccelInfo()
            self.ctx._chk(self.ctx.lib.vd_trace_accel_info(self.h, C.byref(st)))
            return {"tight_tlas": int(st.tight_tlas), "n_tlas_nodes": int(st.n_tlas_nodes), "tight_fallback_instances": int(st.tight_fallback_instances),
                    "triangle_bytes": int(st.triangle_bytes), "d_tlas_nodes": st.d_tlas_nodes}

        def update(self):
            """vd_trace_accel_update_dev: rebuild the private top level from the instance buffer as it is now."""
            self.ctx._chk(self.ctx.lib.vd_trace_accel_update_dev(self.ctx.h, self.h))

        def close(self):
            if getattr(self, "h", None):
                self.ctx.lib.vd_trace_release(self.ctx.h, self.h)
                self.h = None

        def __del__(self):
            try:
                self.close()
            except Exception:
                pass

    def trace_prepare(self, scene: "Context.DeviceScene") -> "Context.TraceAccel":
        return Context.TraceAccel(self, scene)

    def trace_prepared_dev(self, accel: "Context.TraceAccel", d_rays, n_rays, d_out):
        self._chk(self.lib.vd_trace_prepared_dev(self.h, accel.h, abi.ptr(d_rays), n_rays, abi.ptr(d_out)))

    def trace_any_prepared_dev(self, accel: "Context.TraceAccel", d_rays, n_rays, d_out_hit):
        self._chk(self.lib.vd_trace_any_prepared_dev(self.h, accel.h, abi.ptr(d_rays), n_rays, abi.ptr(d_out_hit)))

    def shadow_rays_dev(self, d_positions, d_normals, n_points, light_position, d_rays):
        lp = (C.c_float * 3)(*[float(x) for x in light_position])
        self._chk(self.lib.vd_shadow_rays_dev(self.h, abi.ptr(d_positions), abi.ptr(d_normals), n_points, lp, abi.ptr(d_rays)))

    def primary_rays_dev(self, camera, width, height, d_rays):
        """One ray per pixel from camera.clip_to_world (src/bin/bvh_cpu.rs:71-83); d_rays: width*height rays."""
        cam = np.ascontiguousarray(camera, dtype=abi.CAMERA).reshape(1)
        self._chk(self.lib.vd_primary_rays_dev(self.h, cam.ctypes.data, width, height, abi.ptr(d_rays)))

    def traverse_iter_dev(self, d_nodes, n_nodes, d_verts, d_indices, d_rays, n_rays, d_out_dist):
        """Bvh::traverse_iter (crates/bvh/src/blas.rs:247-295) for a batch of rays against one mesh; -1 = Miss."""
        self._chk(self.lib.vd_traverse_iter_dev(self.h, abi.ptr(d_nodes), n_nodes, abi.ptr(d_verts), abi.ptr(d_indices),
                                                abi.ptr(d_rays), n_rays, abi.ptr(d_out_dist)))

    def traverse_dev(self, d_nodes, n_nodes, d_verts, d_indices, d_rays, n_rays, d_out_dist, t0=1e30):
        """Bvh::traverse (crates/bvh/src/blas.rs:211-245), the recursive walk, started as bvh_cpu.rs:86 would
        (`traverse(.., ray, 0, 1e30)`); Hit(t0) when the root box is entered and nothing is hit, -1 = Miss."""
        self._chk(self.lib.vd_traverse_dev(self.h, abi.ptr(d_nodes), n_nodes, abi.ptr(d_verts), abi.ptr(d_indices),
                                           abi.ptr(d_rays), n_rays, C.c_float(t0), abi.ptr(d_out_dist)))

    # -- occlusion extension (no reference counterpart; include/voidin_abi.h "Occlusion culling") ----------
    def hiz_layout(self, width, height) -> "abi.HizLayout":
        L = abi.HizLayout()
        self._chk(self.lib.vd_hiz_layout(width, height, C.byref(L)))
        return L

    def hiz_build_dev(self, d_depth, width, height, d_pyramid):
        self._chk(self.lib.vd_hiz_build_dev(self.h, abi.ptr(d_depth), width, height, abi.ptr(d_pyramid)))

    def occlusion_mask_dev(self, camera, d_meshes, n_mesh, d_inst, n, d_pyramid, width, height, d_mask_in, d_mask_out):
        cam = np.ascontiguousarray(camera, dtype=abi.CAMERA).reshape(1)
        self._chk(self.lib.vd_occlusion_mask_dev(self.h, cam.ctypes.data, abi.ptr(d_meshes), n_mesh, abi.ptr(d_inst), n,
                                                 abi.ptr(d_pyramid), width, height, abi.ptr(d_mask_in), abi.ptr(d_mask_out)))


# ------------------------------------------------------------------------------------------
# Reference-shaped façade
# ------------------------------------------------------------------------------------------
class EmitDraws:
    """`impl Pass for EmitDraws` (visibility.rs:230-255): record() leaves draw_cmd_buffer[0..N)
    valid on the ctx's stream before the consumer (Geometry::record) runs on the same stream."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def record(self, camera, mesh_info_buf, n_mesh, instances_buf, n_inst, draw_cmd_buffer):
        self.ctx.cull_emit_dev(camera, mesh_info_buf, n_mesh, instances_buf, n_inst, draw_cmd_buffer)

    def record_compacted(self, camera, mesh_info_buf, n_mesh, instances_buf, n_inst, draw_cmd_buffer,
                         draw_count_buf, pad_tail=False):
        self.ctx.cull_compact_dev(camera, mesh_info_buf, n_mesh, instances_buf, n_inst, draw_cmd_buffer,
                                  draw_count_buf, pad_tail)


class Bvh:
    def __init__(self, nodes: np.ndarray):
        self.nodes = nodes


class BvhBuilder:
    """BvhBuilder::new(vertices, indices).build() (blas.rs:51-103).  `indices` is permuted in
    place, as the reference does to the caller's slice."""

    def __init__(self, ctx: Context, vertices: np.ndarray, indices: np.ndarray):
        self.ctx, self.vertices, self.indices = ctx, vertices, indices
        self.num_bins = 8

    def set_bin_number(self, num_bins: int):
        self.num_bins = num_bins  # stored and ignored, exactly like blas.rs:64-67 vs :136
        return self

    def build(self) -> Bvh:
        nodes, idx = self.ctx.bvh_build(self.vertices, self.indices)
        self.indices.reshape(-1)[:] = idx
        return Bvh(nodes)


class Tlas:
    """Tlas::empty() / Tlas::build(&mut self, instances, meshes) (tlas.rs:27-85)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.nodes = np.zeros(0, dtype=abi.TLAS_NODE)

    @classmethod
    def empty(cls, ctx: Context):
        return cls(ctx)

    def build(self, instances, meshes):
        self.nodes = self.ctx.tlas_build(instances, meshes)

    def refit(self, instances, meshes):
        self.nodes = self.ctx.tlas_refit(instances, meshes, self.nodes)
