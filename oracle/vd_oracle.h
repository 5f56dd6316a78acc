/*
 * vd_oracle.h — CPU oracle for the voidin visibility path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (libvoidin_hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference (pudnax/voidin v0.69.0) ships no tests, golden vectors or
 * fixtures for this path and cannot be built here (no Rust toolchain; glam 0.24.1 /
 * naga 0.13 / wgpu 0.17.1 sources are not on disk; WGSL needs a Vulkan device).  Each function
 * below is a strict-fp32 restatement of the cited reference lines; an independent numpy
 * restatement (oracle/np_restate.py) cross-checks it and the golden fixtures in tests/golden/
 * freeze the result.  Where WGSL or glam leave a choice open, the choice is recorded as a
 * "spec decision" next to the code.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fno-fast-math)
 */
#ifndef VD_ORACLE_H
#define VD_ORACLE_H

#include "../include/voidin_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* shaders/emit_draws.wgsl:13-64 (+ utils/math.wgsl:67-73). threads <= 1: scalar loop;
 * threads > 1: static contiguous ranges over pthreads (the "CPU re-run of the cull").   */
int vd_ref_cull_emit(const VdCameraUniform* camera, const VdMeshInfo* meshes, uint32_t n_mesh,
                     const VdInstance* instances, uint32_t n_inst, VdDrawIndexedIndirect* out,
                     int threads);
/* SURVEY.md §8a C3 — pure function of the emit_draws output.                            */
int vd_ref_compact(const VdDrawIndexedIndirect* in, uint32_t n, VdDrawIndexedIndirect* out,
                   uint32_t* out_count, int pad_tail);
/* Signed margins of the two frustum-plane tests (lhs + radius), for the near-boundary
 * histogram (SURVEY.md §7 hard part ii).  margin < 0 => culled by that plane.            */
int vd_ref_cull_margins(const VdCameraUniform* camera, const VdMeshInfo* meshes, uint32_t n_mesh,
                        const VdInstance* instances, uint32_t n_inst, float* out_margin_x,
                        float* out_margin_y, float* out_radius);

/* crates/bvh/src/blas.rs:51-204 — literal sequential builder, quirks included.          */
int vd_ref_bvh_build(const float* verts_xyz, uint32_t n_vert, uint32_t* indices_inout,
                     uint32_t n_tri, VdBvhNode* out_nodes, uint32_t node_cap,
                     uint32_t* out_n_nodes);
/* blas.rs:168-182 on a bare key array (exposed so the closed form can be tested).       */
uint32_t vd_ref_partition_shuffle(const float* keys_by_id, uint32_t* ids, uint32_t start,
                                  uint32_t count, float pos);

/* crates/bvh/src/tlas.rs:31-105                                                          */
int vd_ref_tlas_build(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                      uint32_t n_mesh, VdTlasNode* out_nodes);
int vd_ref_tlas_build_wide(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                           uint32_t n_mesh, VdTlasNodeWide* out_nodes);
/* SURVEY.md §8a T3                                                                       */
int vd_ref_tlas_refit(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                      uint32_t n_mesh, VdTlasNode* nodes_inout);
int vd_ref_tlas_refit_wide(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                           uint32_t n_mesh, VdTlasNodeWide* nodes_inout);

/* shaders/utils/bvh.wgsl:35-123 + intersections.wgsl:13-45 (R1). out_max_stack (optional)
 * receives the deepest stack use seen (the reference's 24-entry stack is unchecked).    */
int vd_ref_trace(const VdTraceScene* scene, const VdRay* rays, uint32_t n_rays, VdHit* out,
                 uint32_t* out_max_stack, int threads);
/* crates/bvh/src/blas.rs:247-295 + intersection.rs:47-92 (R2, BLAS only, Rust CPU harness).
 * out_dist[i] = t, or -1 for Dist::Miss.                                                 */
/* Shadow rays of src/bin/raytraced_shadows.wgsl:90-97: eye = pos + nor * 0.0001, dir = light - pos. */
int vd_ref_shadow_rays(const float* positions, const float* normals, uint32_t n_points, const float* light_position,
                       VdRay* out);

/* src/bin/bvh_cpu.rs:71-83: one ray per pixel from camera.clip_to_world */
int vd_ref_primary_rays(const VdCameraUniform* camera, uint32_t width, uint32_t height, VdRay* out);

int vd_ref_traverse_iter(const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz,
                         const uint32_t* indices, const VdRay* rays, uint32_t n_rays,
                         float* out_dist);

/* crates/bvh/src/blas.rs:211-245 (R3): the recursive Bvh::traverse, unused in the reference (bvh_cpu.rs:86 is commented
 * out).  out_dist[i] = t of Hit(t) - which is t0 itself when the root box is entered and no triangle is nearer - or -1 for Miss. */
int vd_ref_traverse(const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz,
                    const uint32_t* indices, const VdRay* rays, uint32_t n_rays, float t0, float* out_dist);

/* CPU twin of the HiZ occlusion extension (no reference code exists: SURVEY.md 8a C4); definitions in voidin_abi.h */
int vd_ref_hiz_layout(uint32_t width, uint32_t height, VdHizLayout* out);
int vd_ref_hiz_build(const float* depth, uint32_t width, uint32_t height, float* pyramid);
int vd_ref_occlusion_mask(const VdCameraUniform* camera, const VdMeshInfo* meshes, uint32_t n_mesh,
                          const VdInstance* instances, uint32_t n_inst, const float* pyramid, uint32_t width,
                          uint32_t height, const uint64_t* mask_in, uint64_t* mask_out);

/* shaders/compute_update.wgsl:10-28 (+ fix-forward of inv_transform) */
int vd_ref_compute_update(const uint32_t* indices, uint32_t n_indices, VdInstance* instances,
                          uint32_t n_instances, float time, float dt, int fix_inverse);

const char* vd_ref_version(void);

#ifdef __cplusplus
}
#endif
#endif
