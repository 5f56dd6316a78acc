/*
 * voidin_abi.h — C ABI of libvoidin_hip.so, the MI355X (gfx950) implementation of
 * voidin's GPU-driven visibility path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch
 * types.  Every entry point names the reference interface it replaces; paths are relative
 * to the reference checkout (pudnax/voidin v0.69.0).
 *
 * Conventions
 *   - every function returns VD_OK (0) or a negative VdStatus; nothing aborts or throws
 *     across the boundary (the reference panics — crates/bvh/src/blas.rs:114-116 — we do not);
 *   - `*_dev` variants take DEVICE pointers (hipMalloc'd on the ctx's device) and enqueue
 *     on the ctx's stream without synchronising (exceptions, stated at their declarations:
 *     vd_bvh_build_dev, vd_dist_step_draws_dev and the traversal entry points - which read
 *     a status word back - block); the plain variants take HOST pointers,
 *     stage through ctx-owned device buffers and return after the result is in host memory;
 *   - the caller owns every in/out buffer; device scratch belongs to the VdCtx;
 *   - a VdCtx is thread-compatible, not thread-safe (the reference's `World` is
 *     RefCell-based and single-threaded: crates/components/src/world.rs:81-84).
 */
#ifndef VOIDIN_ABI_H
#define VOIDIN_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------ */
/* Data contract. All little-endian, f32/u32/i32, #[repr(C)].      */
/* ------------------------------------------------------------------------------------ */

/* D1  crates/components/src/shared.rs:67-75, shaders/shared.wgsl:53-59 — 144 B */
typedef struct VdInstance {
    float    transform[16];      /* mat4, column-major: transform[4*c + r]          */
    float    inv_transform[16];  /* mat4, column-major                              */
    uint32_t mesh;               /* MeshId                                          */
    uint32_t material;           /* MaterialId                                      */
    uint32_t junk[2];
} VdInstance;

/* D2  crates/components/src/shared.rs:29-39, shaders/shared.wgsl:43-51 — 48 B */
typedef struct VdMeshInfo {
    float    min[3];
    uint32_t index_count;
    float    max[3];
    uint32_t base_index;
    int32_t  vertex_offset;
    uint32_t bvh_index;
    uint32_t junk[2];
} VdMeshInfo;

/* D3  crates/components/src/lib.rs:99-107, shaders/shared.wgsl:69-75 — 20 B */
typedef struct VdDrawIndexedIndirect {
    uint32_t vertex_count;       /* = MeshInfo.index_count                          */
    uint32_t instance_count;     /* 1 visible, 0 culled                             */
    uint32_t base_index;
    int32_t  vertex_offset;
    uint32_t base_instance;      /* = instance index (visibility.wgsl:33 reads it)  */
} VdDrawIndexedIndirect;

/* D4  crates/components/src/camera.rs:13-27, shaders/shared.wgsl:13-24 — 320 B */
typedef struct VdCameraUniform {
    float view_position[4];
    float projection[16];
    float view[16];
    float clip_to_world[16];
    float prev_world_to_clip[16];
    float frustum[4];
    float zfar;
    float znear;
    float jitter[2];
    float prev_jitter[2];
    float _padding[2];
} VdCameraUniform;

/* D5  crates/bvh/src/blas.rs:10-17, shaders/utils/bvh.wgsl:11-16 — 32 B.
 * count > 0: leaf over triangles [left_first, left_first+count) of the reordered index
 * buffer; count == 0: interior, children at node ids left_first and left_first+1.      */
typedef struct VdBvhNode {
    float    min[3];
    uint32_t left_first;
    float    max[3];
    uint32_t count;
} VdBvhNode;

/* D6  crates/bvh/src/tlas.rs:7-14, shaders/utils/bvh.wgsl:4-9 — 32 B.
 * left_right = left | right << 16 (0 => leaf); instance_idx = u32::MAX for interior.    */
typedef struct VdTlasNode {
    float    min[3];
    uint32_t left_right;
    float    max[3];
    uint32_t instance_idx;
} VdTlasNode;

/* Extended TLAS node for more than VD_TLAS_MAX_INSTANCES instances:
 * the reference packs two 16-bit ids into left_right, so it cannot address n > 32768.
 * Same build rule, children kept as two full u32. NOT a reference layout.               */
typedef struct VdTlasNodeWide {
    float    min[3];
    uint32_t left;               /* 0 and right==0 => leaf                           */
    float    max[3];
    uint32_t right;
    uint32_t instance_idx;
    uint32_t _pad[3];
} VdTlasNodeWide;

/* Ray + hit records of the batch traversal entry point (the reference traces one ray per
 * fragment: shaders/utils/intersections.wgsl:3-11, shaders/utils/bvh.wgsl:18-28).        */
typedef struct VdRay {
    float eye[3];
    float _pad0;
    float dir[3];
    float _pad1;
} VdRay;

typedef struct VdHit {
    float    dist;               /* MAX_DIST (1e30) on miss                          */
    uint32_t hit;                /* 0 / 1                                            */
    uint32_t instance;           /* instance whose BLAS produced the closest hit     */
    uint32_t triangle;           /* mesh-local triangle slot in the reordered buffer */
} VdHit;

#define VD_MAX_DIST 1e30f
#define VD_TLAS_MAX_INSTANCES 32768u

#if defined(__cplusplus)
static_assert(sizeof(VdInstance) == 144, "Instance is 144 B (shared.rs:67-75)");
static_assert(sizeof(VdMeshInfo) == 48, "MeshInfo is 48 B (shared.rs:29-39)");
static_assert(sizeof(VdDrawIndexedIndirect) == 20, "DrawIndexedIndirect is 20 B (lib.rs:99-107)");
static_assert(sizeof(VdCameraUniform) == 320, "CameraUniform is 320 B (camera.rs:13-27)");
static_assert(sizeof(VdBvhNode) == 32, "BvhNode is 32 B (blas.rs:10-17)");
static_assert(sizeof(VdTlasNode) == 32, "TlasNode is 32 B (tlas.rs:7-14)");
static_assert(sizeof(VdTlasNodeWide) == 48, "wide TLAS node is 48 B");
static_assert(sizeof(VdRay) == 32 && sizeof(VdHit) == 16, "ray/hit records");
#else
_Static_assert(sizeof(VdInstance) == 144, "Instance is 144 B");
_Static_assert(sizeof(VdMeshInfo) == 48, "MeshInfo is 48 B");
_Static_assert(sizeof(VdDrawIndexedIndirect) == 20, "DrawIndexedIndirect is 20 B");
_Static_assert(sizeof(VdCameraUniform) == 320, "CameraUniform is 320 B");
_Static_assert(sizeof(VdBvhNode) == 32, "BvhNode is 32 B");
_Static_assert(sizeof(VdTlasNode) == 32, "TlasNode is 32 B");
_Static_assert(sizeof(VdTlasNodeWide) == 48, "wide TLAS node is 48 B");
_Static_assert(sizeof(VdRay) == 32 && sizeof(VdHit) == 16, "ray/hit records");
#endif

/* ------------------------------------------------------------------------------------ */
/* Status codes                                                                          */
/* ------------------------------------------------------------------------------------ */
typedef enum VdStatus {
    VD_OK = 0,
    VD_ERR_INVALID_ARG = -1,     /* null pointer, zero/oversized count, bad capacity     */
    VD_ERR_HIP = -2,             /* HIP runtime error; text in vd_last_error             */
    VD_ERR_DEGENERATE = -3,      /* BVH input the reference builder crashes on
                                    (all 21 split candidates rejected: blas.rs:137-140)  */
    VD_ERR_TLAS_OVERFLOW = -4,   /* n > 32768 in the 16-bit reference TLAS layout
                                    (tlas.rs:71)                                         */
    VD_ERR_NO_DEVICE = -5,       /* no gfx950 device / extension built for another arch  */
    VD_ERR_STACK_OVERFLOW = -6,  /* traversal stack exceeded (reference has no check:
                                    shaders/utils/stack.wgsl:1-20).  vd_trace*: only for a
                                    stack of more than 8 Mi entries (see there);
                                    vd_traverse_iter / vd_traverse: beyond 128           */
    VD_ERR_OOM = -7,
    VD_ERR_COMM = -8             /* RCCL: library not found, communicator or collective failed;
                                    text in vd_last_error                                  */
} VdStatus;

typedef struct VdCtx VdCtx;

/* ------------------------------------------------------------------------------------ */
/* Context                                                                               */
/* ------------------------------------------------------------------------------------ */
/* Replaces the wgpu device/queue pair the passes pull from `World`
 * (crates/app/src/app.rs:108-118). One HIP stream per ctx.                              */
int         vd_ctx_create(int device, VdCtx** out_ctx);
int         vd_ctx_destroy(VdCtx* ctx);
/* Run on a caller-owned hipStream_t (e.g. the host framework's current stream); NULL is
 * the HIP default stream.  vd_ctx_reset_stream goes back to the ctx-owned stream.       */
int         vd_ctx_set_stream(VdCtx* ctx, void* hip_stream);
int         vd_ctx_reset_stream(VdCtx* ctx);
int         vd_ctx_synchronize(VdCtx* ctx);
const char* vd_last_error(const VdCtx* ctx);
const char* vd_version(void);

/* Per-context options.  The library reads NO environment variable for these (the one it reads
 * at all is VD_RCCL_LIB, the path of the RCCL shared object: "Multi-GPU exchange" below); defaults
 * are the measured optima.  They exist for A/B measurements (tools/) and so that the
 * tests can force paths that otherwise depend on timing or size (the single-workgroup redo of the
 * TLAS chain, the indexed build on small inputs).  value < 0 restores the default.            */
typedef enum VdOption {
    VD_OPT_CULL_SPLIT_MIN = 1,    /* vd_cull_*: inputs of at least this many instances run the split form
                                     (pass 1 -> scan -> pass 2); default 2 Mi                            */
    VD_OPT_CULL_VARIANT = 2,      /* cull kernel variant (A/B; a small SIGNED id taken as is); default 0  */
    VD_OPT_TLAS_INDEX = 10,       /* 0: never use the indexed TLAS build; default 1                       */
    VD_OPT_TLAS_INDEX_MIN = 11,   /* smallest n the indexed build takes; default 6800                     */
    VD_OPT_TLAS_PHASE2 = 12,      /* clusters left at which the indexed build hands over to plain scans;
                                     default 4096                                                        */
    VD_OPT_TLAS_REFRESH = 13,     /* merges between two re-tightenings of the index corners; default 512  */
    VD_OPT_TLAS_GROUPS = 14,      /* workgroups of the plain chain (1 = single workgroup); default by n   */
    VD_OPT_TLAS_SPIN_LIMIT = 15,  /* polls before the several-workgroup chain gives up and the build is
                                     redone on one workgroup (0 forces the redo: tests)                   */
    VD_OPT_TLAS_SPEC = 16,        /* 0: indexed build without the speculative helper waves; default 1;
                                     2: the speculative query runs on a second workgroup of the same XCC
                                     (measured slower: 205 vs 188 ms at 32 768)    */
    VD_OPT_TLAS_PROFILE = 17,     /* 1: the indexed build prints its in-kernel cycle counters             */
    VD_OPT_TLAS_CHAIN_LDS = 18,   /* 0: the single-workgroup chain reads its slot arrays from memory even when
                                     they would fit LDS (<= 5600 instances); default 1 (A/B)               */
    VD_OPT_BLAS_WIDE_PAYLOAD = 30,/* 1: vd_bvh_build moves the 8-byte payload (what meshes above 2^25 triangles use) at
                                     any size (tests); default 0                                                  */
    VD_OPT_TRACE_SORT = 21,       /* 1: vd_trace* bin the rays first (sorted by origin cell + direction) and hand them
                                     out in that order; results are per ray, so only the order changes.  Default 0:
                                     measured slower on this part                                */
    VD_OPT_TRACE_SORT_MIN = 22,   /* fewest rays a call bins when VD_OPT_TRACE_SORT is on; default 65536          */
    VD_OPT_TRACE_CHUNK = 23,      /* 1 (default): idle lanes draw single rays from one counter; >= 64: consecutive
                                     rays per workgroup in chunks of this size (measured slower: imbalance)      */
    VD_OPT_TRACE_YIELD = 24,      /* lanes of a wave that wait (at a BLAS leaf, or with a finished ray) before the wave
                                     leaves its stepping loop to serve them; default 16                          */
    VD_OPT_TRACE_WAVES = 25,      /* persistent waves per CU of the single-ray supply (1..24); default 24          */
    VD_OPT_TRACE_AUTO_PREPARE = 27,/* 1 (default): a vd_trace_dev / vd_trace_any_dev call de-indexes the leaf triangles
                                     itself (what vd_trace_prepare_dev does once per scene) when that is cheap next to
                                     the walk: n_rays * 8 >= triangles <= 2 Mi, at most 65 535 meshes.  The 36 B per
                                     triangle live in the context's grow-only scratch (<= 72 MB, kept until
                                     vd_ctx_destroy); 2: up to 16 Mi triangles (576 MB); 0: never                */
    VD_OPT_TRACE_FAN = 29,        /* launches one vd_trace* call runs as (1..4; default 3, and 1 for 15 calls after a call whose rays were all too short to fan out).  Once the rays are handed out, a wave that
                                     is down to a few live rays turns each into jobs - one per TLAS subtree on its stack - for
                                     the next launch's waves: the rays that take thousands of steps stop being what the call
                                     waits for.  Same bytes out (the minimum over (t, visit order) is kept through order keys;
                                     DESIGN.md 3.5).  1 = one launch (A/B).  Calls with at least one ray per lane of the grid
                                     and scenes of >= 64 instances only; 8 B per ray + up to 96 MB of job slots in the context's
                                     grow-only scratch                                                             */
    VD_OPT_TRACE_FAN_SLOTS = 26,  /* upper limit of the fan-out's job slots (default: 2 per ray, at most 2 Mi).  A wave that finds the
                                     list full keeps its rays and runs them to the end; the tests use a small value to get there */
    VD_OPT_TRACE_TIGHT_TLAS = 28, /* 1: vd_trace_prepare_dev builds a PRIVATE top level for the prepared scene over tight
                                     world boxes (the 8 transformed corners of each instance's BLAS root box, WITHOUT the
                                     object-space seed of tlas.rs:39 that makes the reference's leaves overlap at the
                                     origin; padded by 2e-5 of the largest coordinate) and starts rays at its true root.
                                     Every BLAS and the walk inside an instance are unchanged, so hit flags are the
                                     reference's and distances are within north_star's 1e-5 (bit-equal in practice; which
                                     of two triangles at the SAME distance is reported may differ).  NOT the reference's
                                     visit order: default 0, and off in every bit-exact parity run.  Needs
                                     inv_transform = transform^-1 and finite boxes for EVERY instance (checked; one
                                     that fails and the scene keeps its own top level - VdTraceAccelInfo says so) and
                                     2 .. 32 768 instances (else ignored).  1: the top level is clustered by the
                                     agglomerative builder of tlas.rs:56-105 (best tree; the sequential chain: 35 ms for
                                     2 000 instances); 2: an LBVH built on all CUs (~0.1 ms, <= 32 767 instances) - the form
                                     for scenes that move.  The private top level is a snapshot of the instances:
                                     vd_trace_accel_update_dev rebuilds it after they moved.                       */
    VD_OPT_COUNT_ = 32
} VdOption;
int         vd_ctx_set_option(VdCtx* ctx, int option /* VdOption */, int64_t value);

/* HIP -> wgpu hand-off.  The renderer owns `draw_cmd_buffer`
 * (ResizableBuffer<DrawIndexedIndirect>, crates/components/src/buffer.rs:42-47, created in
 * app.rs:167-171); to let Geometry::record consume the commands without a PCIe round trip the
 * Vulkan allocation behind it is exported as an opaque fd (VK_KHR_external_memory_fd) and mapped
 * here; the returned device pointer is then passed as `d_out` of vd_cull_emit_dev /
 * vd_cull_compact_dev.  The fd is consumed on success (Vulkan/HIP convention).  Ordering
 * against the consumer: the external semaphores below (no CPU wait), or vd_ctx_synchronize.  */
typedef struct VdExternalBuffer VdExternalBuffer;
int vd_import_external_buffer(VdCtx* ctx, int opaque_fd, uint64_t size_bytes, VdExternalBuffer** out_handle,
                              void** out_device_ptr);
int vd_release_external_buffer(VdCtx* ctx, VdExternalBuffer* handle);

/* ... and its ordering.  The renderer's frame is ONE queue.submit followed by present
 * (crates/app/src/app.rs:334-348); a CPU wait on each side of a 0.27 ms cull would be most of the frame.  A Vulkan
 * semaphore exported as an opaque fd (VK_KHR_external_semaphore_fd; binary, or timeline with is_timeline != 0) is
 * imported once, and then
 *     vd_wait_external_semaphore_async(ctx, frame_ready, f)      the submit that wrote instances / camera has finished
 *     vd_cull_compact_dev(ctx, ..., d_out = imported buffer, ...)
 *     vd_signal_external_semaphore_async(ctx, draws_ready, f)    Geometry::record's submit waits for this one
 * are three operations ENQUEUED on the context's stream: no host round trip.  `value` is the timeline point (ignored
 * for a binary semaphore).  The fd is consumed on success.  vd_release_external_semaphore synchronises the stream
 * first (queued waits / signals refer to the semaphore).
 * What this image can test: argument validation and the error path of a descriptor that is not a semaphore
 * (tests/test_gpu_tlas_trace.py), and - round 6 - an import of a REAL kernel sync object (what a Vulkan binary semaphore
 * exported as an opaque fd is on amdgpu; tests/cpp/external_semaphore_test.cpp makes one with DRM_IOCTL_SYNCOBJ_CREATE):
 * the HIP runtime of this image answers hipImportExternalSemaphore with "operation not supported" (binary) and
 * "invalid argument" (timeline), so VD_ERR_HIP is what these calls return on this platform today and no functional
 * round trip is claimed (profiles/r06_external_semaphore_probe.log; the test runs the signal and the wait round trip
 * on a runtime that accepts the import).  The calls are hipImportExternalSemaphore /
 * hip{Wait,Signal}ExternalSemaphoresAsync / hipDestroyExternalSemaphore and nothing else.  Until a runtime implements
 * them, order the frame through the host without blocking the render thread: INTEGRATION.md 5.                       */
typedef struct VdExternalSemaphore VdExternalSemaphore;
int vd_import_external_semaphore(VdCtx* ctx, int opaque_fd, int is_timeline, VdExternalSemaphore** out_handle);
int vd_wait_external_semaphore_async(VdCtx* ctx, VdExternalSemaphore* handle, uint64_t value);
int vd_signal_external_semaphore_async(VdCtx* ctx, VdExternalSemaphore* handle, uint64_t value);
int vd_release_external_semaphore(VdCtx* ctx, VdExternalSemaphore* handle);

/* Ordering that WORKS on this platform (NEW, round 6): a word of shared memory and the host.
 * The external-semaphore import above is refused by this image's HIP runtime (profiles/r06_external_semaphore_probe.log),
 * so the frame's two hand-overs have a second form:
 *   renderer -> HIP, GPU side only: the renderer's upload submit ends by writing the frame number f into a 32-bit word of a
 *     buffer both APIs see (vkCmdFillBuffer into an allocation imported with vd_import_external_buffer - the draw buffer's
 *     own tail will do); vd_wait_value32_async(ctx, d_word, f) holds the context's stream until *d_word >= f (unsigned,
 *     hipStreamWaitValue32 / GTE), then the cull runs.  No host thread waits, none is woken.
 *   HIP -> renderer: Vulkan cannot wait on a memory word; vd_host_callback_async(ctx, callback, user) runs callback(user) on a thread
 *     of the HIP runtime once the stream reaches it (hipLaunchHostFunc) - it calls vkSignalSemaphore on a timeline semaphore
 *     the frame's queue.submit waits for.  The callback must not call into HIP or this library.
 *   vd_write_value32_async(ctx, d_word, v) is the stream-side store (hipStreamWriteValue32): a second HIP context, or a
 *     test, stands in for the renderer's queue with it.
 * Enqueue-only, like the per-frame entry points; d_word must be 4-byte aligned device-accessible memory of the context's
 * device (ordinary allocations and imported external buffers both work here: tests/cpp/frame_ordering_test.cpp).  Stream
 * memory operations are not captured into HIP graphs: keep them outside a captured frame.  A wait nobody ever satisfies
 * holds the stream for good (vd_ctx_synchronize and everything that synchronises would not return): the protocol on the
 * word - monotone frame numbers, written once per frame by the other queue - is the caller's.                          */
typedef void (*VdHostFn)(void* user);
int vd_wait_value32_async(VdCtx* ctx, const uint32_t* d_word, uint32_t value);
int vd_write_value32_async(VdCtx* ctx, uint32_t* d_word, uint32_t value);
int vd_host_callback_async(VdCtx* ctx, VdHostFn callback, void* user);

/* ------------------------------------------------------------------------------------ */
/* Cull + emit                                                      */
/* ------------------------------------------------------------------------------------ */
/* C1/C2 — replaces the `emit_draws` compute dispatch recorded by EmitDraws::record
 * (crates/app/src/pass/visibility.rs:233-254; shaders/emit_draws.wgsl:13-64).
 * Writes out[0..n_inst): every slot, culled ones with instance_count = 0.
 * mesh ids >= n_mesh are clamped to n_mesh-1 (the reference leaves this to Vulkan robust
 * buffer access, i.e. undefined).                                                       */
int vd_cull_emit(VdCtx* ctx, const VdCameraUniform* camera,
                 const VdMeshInfo* meshes, uint32_t n_mesh,
                 const VdInstance* instances, uint32_t n_inst,
                 VdDrawIndexedIndirect* out);
int vd_cull_emit_dev(VdCtx* ctx, const VdCameraUniform* camera /* host */,
                     const VdMeshInfo* d_meshes, uint32_t n_mesh,
                     const VdInstance* d_instances, uint32_t n_inst,
                     VdDrawIndexedIndirect* d_out);

/* C1+C3 fused — cull and emit only the survivors, in ascending instance order:
 *   S = [i : emit_draws(i).instance_count == 1];  out[k] = emit_draws(S[k]);  *count = |S|.
 * base_instance keeps the original instance index, so shaders/visibility.wgsl:33 is
 * unchanged.  If pad_tail != 0, out[count..n_inst) is filled with zeroed commands
 * (instance_count = 0) so that the unchanged
 * `multi_draw_indexed_indirect(buf, 0, N)` consumer (visibility.rs:188-192) stays valid;
 * with pad_tail == 0 only out[0..count) is written (for multi_draw_indexed_indirect_count).
 * `out` must hold n_inst commands either way.
 * When a launch fails: the ordered scans behind the compaction wait for each other across workgroups, and every
 * such wait is bounded by the wall clock (two seconds).  If one times out (a workgroup of the launch was lost or
 * stalled) no list is written, *count becomes 0 - with pad_tail the whole buffer is zeroed, so neither consumer
 * draws anything, least of all a mix of this frame's and the last frame's commands - and the context remembers:
 * vd_cull_compact returns VD_ERR_HIP for that very call; after a *_dev call the NEXT vd_cull_compact* /
 * vd_compact_draws* call on the context returns VD_ERR_HIP once (and resets the scan state) instead of launching.
 * One case cannot be turned into a count of 0: a launch that loses the workgroup that writes the count leaves the
 * value its first workgroup pre-stored, 0xffffffff - larger than any n_inst.  vd_cull_compact returns VD_ERR_HIP
 * for it as well; pad_tail zeroes the whole buffer for it; a caller that hands the count straight to
 * multi_draw_indexed_indirect_count should clamp it to 0 when it exceeds n_inst.                              */
int vd_cull_compact(VdCtx* ctx, const VdCameraUniform* camera,
                    const VdMeshInfo* meshes, uint32_t n_mesh,
                    const VdInstance* instances, uint32_t n_inst,
                    VdDrawIndexedIndirect* out, uint32_t* out_count, int pad_tail);
int vd_cull_compact_dev(VdCtx* ctx, const VdCameraUniform* camera /* host */,
                        const VdMeshInfo* d_meshes, uint32_t n_mesh,
                        const VdInstance* d_instances, uint32_t n_inst,
                        VdDrawIndexedIndirect* d_out, uint32_t* d_out_count, int pad_tail);

/* Shard variants (NEW): the instance array is a contiguous shard
 * [first_instance, first_instance + n_inst) of a larger scene; base_instance is written as the
 * GLOBAL index so that shards concatenated in rank order equal the single-GPU result.      */
int vd_cull_emit_shard_dev(VdCtx* ctx, const VdCameraUniform* camera /* host */,
                           const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           const VdInstance* d_instances, uint32_t n_inst, uint32_t first_instance,
                           VdDrawIndexedIndirect* d_out);
int vd_cull_compact_shard_dev(VdCtx* ctx, const VdCameraUniform* camera /* host */,
                              const VdMeshInfo* d_meshes, uint32_t n_mesh,
                              const VdInstance* d_instances, uint32_t n_inst, uint32_t first_instance,
                              VdDrawIndexedIndirect* d_out, uint32_t* d_out_count, int pad_tail);

/* Multi-GPU wire format (NEW): the exchange between GPUs carries ONE BIT per
 * instance instead of a 20-byte command.
 *   vd_cull_mask_dev    runs the cull and writes bit i%64 of d_mask[i/64] = emit_draws(i).instance_count
 *                       (ceil(n_inst/64) words; padding bits are 0).
 *   vd_expand_mask_dev  rebuilds the ordered compacted draw list of a whole scene from the
 *                       concatenated shard masks: shard r covers instances [r*shard_size,
 *                       min(n_total, (r+1)*shard_size)) and owns ceil(shard_size/64) words; the
 *                       per-instance mesh ids (d_mesh_ids[n_total], id_bytes = 1, 2 or 4 bytes each;
 *                       static per scene, replicated once) supply the command fields.  Output ==
 *                       vd_cull_compact on the whole scene, bit for bit.                          */
int vd_cull_mask_dev(VdCtx* ctx, const VdCameraUniform* camera /* host */,
                     const VdMeshInfo* d_meshes, uint32_t n_mesh,
                     const VdInstance* d_instances, uint32_t n_inst, uint64_t* d_mask);
int vd_expand_mask_dev(VdCtx* ctx, const uint64_t* d_mask, uint32_t n_total, uint32_t shard_size,
                       const void* d_mesh_ids, uint32_t id_bytes, const VdMeshInfo* d_meshes,
                       uint32_t n_mesh, VdDrawIndexedIndirect* d_out, uint32_t* d_out_count);

/* Indices-only wire format (NEW; SURVEY.md §8e "gather only survivor indices (4 B each) and rebuild
 * commands locally from replicated mesh_ids"): pays off against the bitmask when fewer than 1 instance
 * in 32 survives.
 *   vd_mask_to_indices_dev  ascending list of the set bits of a shard mask (ceil(n_inst/64) words) as
 *                           GLOBAL instance indices first_instance + i; *d_out_count = their number.
 *   vd_indices_to_draws_dev d_out[k] = the command emit_draws writes for instance d_indices[k] with
 *                           instance_count = 1 (mesh fields through d_mesh_ids, as vd_expand_mask_dev).
 * vd_indices_to_draws(concat over shards of vd_mask_to_indices(vd_cull_mask(shard))) == vd_cull_compact
 * on the whole scene, bit for bit.                                                                  */
int vd_mask_to_indices_dev(VdCtx* ctx, const uint64_t* d_mask, uint32_t n_inst, uint32_t first_instance,
                           uint32_t* d_out_indices, uint32_t* d_out_count);
int vd_indices_to_draws_dev(VdCtx* ctx, const uint32_t* d_indices, uint32_t n_indices, const void* d_mesh_ids,
                            uint32_t id_bytes, uint32_t n_total, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                            VdDrawIndexedIndirect* d_out);

/* C3 alone — ordered compaction of an existing emit_draws output (same definition).     */
int vd_compact_draws_dev(VdCtx* ctx, const VdDrawIndexedIndirect* d_in, uint32_t n,
                         VdDrawIndexedIndirect* d_out, uint32_t* d_out_count);

/* ------------------------------------------------------------------------------------ */
/* BLAS build                                                       */
/* ------------------------------------------------------------------------------------ */
/* Replaces `BvhBuilder::new(vertices, indices).build()` (crates/bvh/src/blas.rs:51-103),
 * called from MeshPool::add (crates/pools/src/mesh/mod.rs:320-321).
 *   verts_xyz      n_vert * 3 floats
 *   indices_inout  n_tri * 3 u32; permuted in place exactly as blas.rs:95-100 does
 *   out_nodes      capacity node_cap (>= 2*n_tri as the reference allocates, blas.rs:52)
 *   out_n_nodes    number of nodes used (`nodes.truncate(pool)`, blas.rs:93); node 1 is
 *                  the reference's never-used all-zero slot
 * VD_ERR_DEGENERATE where the reference would crash.
 *
 * vd_bvh_build_dev takes device pointers but - unlike the per-frame `*_dev` entry points - it
 * BLOCKS: it returns when the build is complete on the ctx stream, out_n_nodes is a HOST pointer,
 * and errors (degenerate input, bad index) are reported by its return value.  The builder is a
 * load-time call in the reference (MeshPool::add, mesh/mod.rs:320-345, before the first frame), the
 * level loop reads one 32-byte control word per level back to size the next level's launches, and
 * the DFS pre-order numbering of the top tree (blas.rs:110-112) is a host pass that overlaps the
 * device's phase B; it cannot be captured into a HIP graph.  Builds of different meshes overlap by
 * using one VdCtx (= one stream) per concurrent build.                                       */
int vd_bvh_build(VdCtx* ctx, const float* verts_xyz, uint32_t n_vert,
                 uint32_t* indices_inout, uint32_t n_tri,
                 VdBvhNode* out_nodes, uint32_t node_cap, uint32_t* out_n_nodes);
int vd_bvh_build_dev(VdCtx* ctx, const float* d_verts_xyz, uint32_t n_vert,
                     uint32_t* d_indices_inout, uint32_t n_tri,
                     VdBvhNode* d_out_nodes, uint32_t node_cap, uint32_t* out_n_nodes /* host */);

/* Batched BLAS build (NEW): K meshes in ONE build.  MeshPool::add builds one BLAS per mesh at load
 * (crates/pools/src/mesh/mod.rs:309-351: Sponza-class scenes are hundreds of small meshes) and a
 * single build has a fixed cost of launches and host round trips that dwarfs a 15 k-triangle mesh.
 * Here the meshes' triangles lie side by side in one position space and every pass of the builder -
 * the level loop (one host round trip per level for the WHOLE batch), the mid tier, the small
 * subtrees - runs once for all of them; meshes of <= 2048 / <= 512 triangles start in those tiers
 * directly.  Each mesh's nodes and permuted indices are exactly what vd_bvh_build gives for that
 * mesh alone (mesh-local node ids and leaf positions, node 1 all-zero).
 *   items          host array; per mesh: vertices, mesh-local indices (permuted in place), and either
 *                  its own node array (out_nodes, node_cap) or out_nodes = NULL = PACKED: the mesh's
 *                  nodes follow the previous packed mesh's in `packed_nodes`, from `packed_first` on -
 *                  MeshPool's `bvh_index = bvh_nodes.len()` bookkeeping (mesh/mod.rs:320-345);
 *                  written back: out_n_nodes, out_first_node (= MeshInfo.bvh_index when packed), status
 *   packed_nodes   shared node buffer of `packed_cap` nodes (may be NULL when no item is packed)
 *   out_packed_end one past the last packed node written (host pointer, may be NULL)
 * _dev: every pointer inside the items and packed_nodes are device pointers; blocks like
 * vd_bvh_build_dev.  An error fails the whole batch (nothing is to be used); items[m].status names the
 * mesh when the error has one (bad index, capacity).                                              */
typedef struct VdBvhBatchItem {
    const float* verts_xyz;        /* n_vert * 3 floats                                     */
    uint32_t*    indices_inout;    /* n_tri * 3 u32, mesh-local                             */
    VdBvhNode*   out_nodes;        /* NULL = packed                                         */
    uint32_t     n_vert, n_tri, node_cap;
    uint32_t     out_n_nodes;      /* written                                               */
    uint32_t     out_first_node;   /* written                                               */
    int32_t      status;           /* written                                               */
} VdBvhBatchItem;
int vd_bvh_build_batch(VdCtx* ctx, VdBvhBatchItem* items, uint32_t n_items, VdBvhNode* packed_nodes,
                       uint64_t packed_cap, uint32_t packed_first, uint32_t* out_packed_end);
int vd_bvh_build_batch_dev(VdCtx* ctx, VdBvhBatchItem* items, uint32_t n_items, VdBvhNode* d_packed_nodes,
                           uint64_t packed_cap, uint32_t packed_first, uint32_t* out_packed_end);

/* ------------------------------------------------------------------------------------ */
/* TLAS build / refit                                               */
/* ------------------------------------------------------------------------------------ */
/* Replaces `Tlas::build(&mut self, &[Instance], &[MeshInfo])`
 * (crates/bvh/src/tlas.rs:31-105), called from MeshPool::generate_tlas
 * (crates/pools/src/mesh/mod.rs:279-286). out_nodes holds 2*n+1 nodes.
 * n > 32768 => VD_ERR_TLAS_OVERFLOW (use the *_wide variant).  The *_dev forms only enqueue
 * work (from 12288 instances on the chain is shared by 16 workgroups; whether they stayed in
 * step, and the redo on one workgroup if not, is decided on the device).                  */
int vd_tlas_build(VdCtx* ctx, const VdInstance* instances, uint32_t n,
                  const VdMeshInfo* meshes, uint32_t n_mesh, VdTlasNode* out_nodes);
int vd_tlas_build_dev(VdCtx* ctx, const VdInstance* d_instances, uint32_t n,
                      const VdMeshInfo* d_meshes, uint32_t n_mesh, VdTlasNode* d_out_nodes);
int vd_tlas_build_wide(VdCtx* ctx, const VdInstance* instances, uint32_t n,
                       const VdMeshInfo* meshes, uint32_t n_mesh, VdTlasNodeWide* out_nodes);
int vd_tlas_build_wide_dev(VdCtx* ctx, const VdInstance* d_instances, uint32_t n,
                           const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           VdTlasNodeWide* d_out_nodes);

/* T3 refit (NEW, not in the reference): keep the topology in nodes_inout, recompute leaf
 * boxes (tlas.rs:34-54) and interior boxes bottom-up (ascending k = n+1..2n, then
 * nodes[0] = nodes[2n]).  refit(build(x), x) == build(x) bit for bit.                   */
int vd_tlas_refit(VdCtx* ctx, const VdInstance* instances, uint32_t n,
                  const VdMeshInfo* meshes, uint32_t n_mesh, VdTlasNode* nodes_inout);
int vd_tlas_refit_dev(VdCtx* ctx, const VdInstance* d_instances, uint32_t n,
                      const VdMeshInfo* d_meshes, uint32_t n_mesh, VdTlasNode* d_nodes_inout);
int vd_tlas_refit_wide_dev(VdCtx* ctx, const VdInstance* d_instances, uint32_t n,
                           const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           VdTlasNodeWide* d_nodes_inout);

/* ------------------------------------------------------------------------------------ */
/* Traversal                                                           */
/* ------------------------------------------------------------------------------------ */
/* The scene the trace bind group exposes (crates/app/src/app.rs:255-287): six storage
 * buffers.  All device pointers for *_dev, all host pointers otherwise.                 */
typedef struct VdTraceScene {
    const VdTlasNode* tlas_nodes;   uint32_t n_tlas_nodes;
    const VdInstance* instances;    uint32_t n_instances;
    const VdMeshInfo* meshes;       uint32_t n_meshes;
    const VdBvhNode*  bvh_nodes;    uint32_t n_bvh_nodes;
    const float*      vertices;     uint32_t n_vertices;   /* xyz triples            */
    const uint32_t*   indices;      uint32_t n_indices;    /* reordered, all meshes  */
} VdTraceScene;

/* Replaces `traverse_tlas(ray)` (shaders/utils/bvh.wgsl:89-123) for a batch of rays.
 * out[i].dist matches the WGSL result within 1e-5 relative; a ray has 128 stack entries
 * for its TLAS + BLAS walk in registers / LDS, pushing far children only (reference: 24 per
 * walk, unchecked).  The call is TOTAL: a ray that needs more is walked again, from its start,
 * by a second pass whose stack goes on in global memory (grow-only scratch of the context,
 * allocated when first needed: 1 Ki entries per lane, then 4 Ki, ... under a 256 MB budget) -
 * same visits, same arithmetic, same record as an unbounded stack gives; calls without such a
 * ray pay nothing.  VD_ERR_STACK_OVERFLOW is left for a stack deeper than 8 Mi entries or than
 * the scene has nodes (cyclic node arrays).  BLAS leaves hold at most 3 triangles
 * (what BvhBuilder makes, blas.rs:108); anything else is VD_ERR_INVALID_ARG, and so is a TLAS
 * leaf a ray ENTERS whose instance index, whose mesh's BLAS root or whose root's children lie
 * outside the scene's buffers (unreachable slots of the TLAS array may hold anything).  Every
 * call first re-lays what the walk reads at a TLAS step and at an instance entry into one
 * cache line each (<= 65 536 TLAS nodes: a few microseconds, in the context's scratch), from
 * the scene's own buffers - nothing is cached across calls.                                   */
int vd_trace(VdCtx* ctx, const VdTraceScene* scene, const VdRay* rays, uint32_t n_rays,
             VdHit* out);
int vd_trace_dev(VdCtx* ctx, const VdTraceScene* d_scene /* struct on host, pointers on device */,
                 const VdRay* d_rays, uint32_t n_rays, VdHit* d_out);

/* Per-scene preparation for many trace calls over static geometry (the reference binds the six
 * buffers once per scene: app.rs:255-287): the leaf triangles are written out de-indexed, 36 bytes
 * per triangle in index-buffer (= leaf) order, so that a BLAS leaf is ONE contiguous fetch instead
 * of indices[] -> vertices[] (bvh.wgsl:30-33, 49-53).  Same vertices, same arithmetic: results are
 * bit-identical to vd_trace_dev / vd_trace_any_dev.  The accel holds a copy of the scene struct
 * (the pointers, not the data): rebuild it when vertices / indices / meshes change; instances
 * and TLAS nodes may change freely (they are read through the scene's pointers).  vd_trace_prepare_dev
 * blocks (it validates the index ranges); release with vd_trace_release.                      */
typedef struct VdTraceAccel VdTraceAccel;
int vd_trace_prepare_dev(VdCtx* ctx, const VdTraceScene* d_scene, VdTraceAccel** out);
int vd_trace_release(VdCtx* ctx, VdTraceAccel* accel);
/* What a prepared scene holds: whether it walks a private top level (VD_OPT_TRACE_TIGHT_TLAS was set when it was
 * prepared and every instance qualified), the nodes of the top level it walks (device pointer: the scene's own or
 * the private one), how many instances did NOT qualify (inv_transform not the inverse of transform, or non-finite
 * corners: any makes the option decline), and the bytes of de-indexed triangles it owns.                          */
typedef struct VdTraceAccelInfo {
    uint32_t tight_tlas /* 0: the scene's own top level is walked; 1 / 2: the private one, by builder */, n_tlas_nodes, tight_fallback_instances, _pad;
    uint64_t triangle_bytes;
    const VdTlasNode* d_tlas_nodes;
} VdTraceAccelInfo;
int vd_trace_accel_info(const VdTraceAccel* accel, VdTraceAccelInfo* out);
/* Rebuilds the private top level of a prepared scene from the scene's instance buffer as it is NOW (the instances moved:
 * shaders/compute_update.wgsl:10-28).  The triangles are not touched.  Blocks (it reads back whether every instance still
 * qualifies; if one does not, the walk goes back to the scene's own top level until an update finds all qualifying again).
 * A no-op for a scene prepared without VD_OPT_TRACE_TIGHT_TLAS: that one walks the host's top level, which the host refits.  */
int vd_trace_accel_update_dev(VdCtx* ctx, VdTraceAccel* accel);
int vd_trace_prepared_dev(VdCtx* ctx, const VdTraceAccel* accel, const VdRay* d_rays, uint32_t n_rays, VdHit* d_out);
int vd_trace_any_prepared_dev(VdCtx* ctx, const VdTraceAccel* accel, const VdRay* d_rays, uint32_t n_rays,
                              uint32_t* d_out_hit);

/* Occlusion query: d_out_hit[i] = traverse_tlas(ray i).hit, which is all the
 * reference's shadow pass reads (src/bin/raytraced_shadows.wgsl:97-102).  Same walk as vd_trace
 * up to the first accepted triangle, where the lane stops; the flag equals vd_trace's `hit`.  */
int vd_trace_any_dev(VdCtx* ctx, const VdTraceScene* d_scene, const VdRay* d_rays, uint32_t n_rays,
                     uint32_t* d_out_hit);
/* The shadow rays of that pass for one point light: eye = pos + nor * 0.0001, dir = light - pos
 * (not normalised; raytraced_shadows.wgsl:90-97).  positions / normals: n_points x 3 floats,
 * device; light_position: 3 floats, host.                                                  */
int vd_shadow_rays_dev(VdCtx* ctx, const float* d_positions, const float* d_normals, uint32_t n_points,
                       const float* light_position, VdRay* d_rays);

/* The CPU harness of the reference, batched on the device.
 * vd_primary_rays_dev: one ray per pixel from camera->clip_to_world, as src/bin/bvh_cpu.rs:71-83
 * builds them (pixel i: x = (i % width) / width, y = (i / height) / height - the source's own
 * row formula; eye = unprojected (x, y, 1), dir = normalised unprojected (x, y, 0)).  camera on
 * the host, d_rays: width * height rays, device.  The fragment-shader harness
 * (src/bin/bvh_trace.wgsl:225-234) makes the same rays from interpolated uv.
 * vd_traverse_iter_dev: `Bvh::traverse_iter` (crates/bvh/src/blas.rs:247-295) for every ray
 * against ONE mesh (mesh-local node ids, no TLAS): slab test dividing by dir
 * (intersection.rs:47-55), two-sided triangle test with EPS 1e-4 (intersection.rs:68-92),
 * near child pushed first.  d_out_dist[i] = closest t, or -1 for Dist::Miss.  The reference's
 * 32-entry stack panics when it overflows (blas.rs:298-324); here 128 entries and
 * VD_ERR_STACK_OVERFLOW.                                                                  */
int vd_primary_rays_dev(VdCtx* ctx, const VdCameraUniform* camera /* host */, uint32_t width,
                        uint32_t height, VdRay* d_rays);
int vd_traverse_iter_dev(VdCtx* ctx, const VdBvhNode* d_nodes, uint32_t n_nodes,
                         const float* d_verts_xyz, const uint32_t* d_indices,
                         const VdRay* d_rays, uint32_t n_rays, float* d_out_dist);
/* vd_traverse_dev: `Bvh::traverse` (crates/bvh/src/blas.rs:211-245), the reference's RECURSIVE
 * walk (its one call, `self.bvh.traverse(.., ray, 0, 1e30)`, is commented out at
 * src/bin/bvh_cpu.rs:86), for every ray against one mesh, started at node 0 with `t = t0`: same
 * slab and triangle tests as traverse_iter, children visited left then right without ordering,
 * the running t handed from call to call.  d_out_dist[i] = the returned Hit(t) - which is t0
 * itself when the ray enters the root box and hits nothing (the reference's quirk, kept) - or
 * -1 for Dist::Miss (root box missed).  The recursion is run with an explicit stack of pending
 * right children: 128 entries, VD_ERR_STACK_OVERFLOW beyond.  (The reference passes Vec4 / UVec4
 * arrays there and uses xyz; here the Vec3 / UVec3 arrays of traverse_iter.)                 */
int vd_traverse_dev(VdCtx* ctx, const VdBvhNode* d_nodes, uint32_t n_nodes,
                    const float* d_verts_xyz, const uint32_t* d_indices,
                    const VdRay* d_rays, uint32_t n_rays, float t0, float* d_out_dist);
/* host-pointer forms (staged through the context, synchronous)                            */
int vd_primary_rays(VdCtx* ctx, const VdCameraUniform* camera, uint32_t width, uint32_t height,
                    VdRay* rays);
int vd_traverse_iter(VdCtx* ctx, const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz,
                     uint32_t n_vert, const uint32_t* indices /* 3 * n_tri, as vd_bvh_build left them */,
                     uint32_t n_tri, const VdRay* rays, uint32_t n_rays, float* out_dist);
int vd_traverse(VdCtx* ctx, const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz,
                uint32_t n_vert, const uint32_t* indices, uint32_t n_tri, const VdRay* rays,
                uint32_t n_rays, float t0, float* out_dist);

/* ------------------------------------------------------------------------------------ */
/* Multi-GPU exchange over RCCL  (NEW; SURVEY.md §8e, BASELINE.json configs[3])          */
/* ------------------------------------------------------------------------------------ */
/* The reference is single-GPU (one wgpu::Device, crates/app/src/app.rs:108-118); this is the
 * north-star extension: instances shard by contiguous ranges - rank r of `world` owns
 * [r*S, min(N, (r+1)*S)), S = ceil(N / world) - and every rank ends a step with the ordered
 * compacted draw list of the WHOLE scene, bit-identical to vd_cull_compact on one GPU.  One
 * process (or thread) per GPU, one VdDist per VdCtx; the collectives are enqueued on the
 * context's stream between the two kernels, so a step is three enqueues on one stream.
 *
 *   vd_dist_unique_id       rank 0 makes the 128-byte communicator id (ncclGetUniqueId) and hands it to
 *                           the other ranks by whatever channel the host has (the Rust host: its own
 *                           IPC; the tests: a gloo broadcast).
 *   vd_dist_create          ncclCommInitRank on ctx's device; collective over all ranks.  world = 1 is a
 *                           valid communicator (a one-GPU functional check of the whole path).
 *   vd_dist_set_scene_dev   per scene: sizes the shard, builds this rank's rows of the replicated
 *                           instance -> mesh table (min(instance.mesh, n_mesh - 1); 1, 2 or 4 bytes by
 *                           table size) and all-gathers the table.  n_local must be this rank's shard
 *                           size.  Mesh assignment is static in the reference's scenes (only transforms
 *                           animate: shaders/compute_update.wgsl:10-28); call again when it changes.
 *   vd_dist_step_full_dev   vd_cull_mask_dev(own shard) -> ncclAllGather(1 bit per instance) ->
 *                           vd_expand_mask_dev(all shards): d_out[0..*d_out_count) = the whole scene's
 *                           list (d_out holds n_total commands).  No host round trip, no allocation.
 *   vd_dist_step_draws_dev  the literal exchange of the 20-byte commands: compact the own shard, all-gather
 *                           the counts (read back on the host: the sizes are data dependent), exact-size
 *                           grouped ncclSend / ncclRecv straight into every peer's final buffer.
 *   vd_dist_step_indices_dev  the same exchange with 4-byte survivor indices on the wire (SURVEY.md §8e's option;
 *                           5x fewer bytes than the commands, fewer than the bitmask when < 1 in 32 survives):
 *                           vd_cull_mask_dev -> vd_mask_to_indices_dev (global indices) -> counts (host read
 *                           back) -> grouped ncclSend / ncclRecv -> vd_indices_to_draws_dev.  Its two index
 *                           buffers (4 B per shard slot + 4 B per scene slot) are allocated on first use.
 *   vd_dist_allgather_dev   plain byte all-gather on the ctx stream (d_recv holds world * bytes_per_rank).
 * All three steps leave the same bytes in d_out[0..*d_out_count) on every rank: vd_cull_compact of the
 * whole scene.  A failed ncclSend / ncclRecv closes its group before VD_ERR_COMM is returned, so the
 * communicator's thread is never left with an open group.
 * RCCL is bound at run time: $VD_RCCL_LIB if the host sets it (an explicit choice wins - the tests bind
 * their test double this way), else the copy already loaded in the process, else librccl.so.1 /
 * /opt/rocm/lib; failure is VD_ERR_COMM, never an abort.                                             */
#define VD_DIST_ID_BYTES 128
typedef struct VdDist VdDist;
typedef struct VdDistInfo {
    int32_t  rank, world;
    uint32_t n_total, shard_size, first_instance, n_local;
    uint32_t mask_words_per_shard, id_bytes;
    uint64_t* d_mask;        /* this rank's mask (mask_words_per_shard words)                  */
    uint64_t* d_mask_all;    /* all shards' masks after a full step                            */
    void*     d_mesh_ids;    /* replicated instance -> mesh table (shard_size * world rows)    */
    int32_t  rccl_version;   /* ncclGetVersion, e.g. 22606                                     */
    int32_t  _pad;
    char     rccl_library[128];
} VdDistInfo;
int vd_dist_unique_id(void* out_id /* VD_DIST_ID_BYTES */);
int vd_dist_create(VdCtx* ctx, const void* unique_id /* VD_DIST_ID_BYTES */, int rank, int world, VdDist** out);
int vd_dist_destroy(VdDist* dist);
int vd_dist_info(const VdDist* dist, VdDistInfo* out);
int vd_dist_set_scene_dev(VdDist* dist, const VdInstance* d_shard_instances, uint32_t n_local, uint32_t n_total,
                          uint32_t n_mesh);
int vd_dist_step_full_dev(VdDist* dist, const VdCameraUniform* camera /* host */, const VdMeshInfo* d_meshes,
                          uint32_t n_mesh, const VdInstance* d_shard_instances, VdDrawIndexedIndirect* d_out,
                          uint32_t* d_out_count);
int vd_dist_step_draws_dev(VdDist* dist, const VdCameraUniform* camera /* host */, const VdMeshInfo* d_meshes,
                           uint32_t n_mesh, const VdInstance* d_shard_instances, VdDrawIndexedIndirect* d_out,
                           uint32_t* d_out_count);
int vd_dist_step_indices_dev(VdDist* dist, const VdCameraUniform* camera /* host */, const VdMeshInfo* d_meshes,
                             uint32_t n_mesh, const VdInstance* d_shard_instances, VdDrawIndexedIndirect* d_out,
                             uint32_t* d_out_count);
int vd_dist_allgather_dev(VdDist* dist, const void* d_send, void* d_recv, uint64_t bytes_per_rank);

/* ------------------------------------------------------------------------------------ */
/* Occlusion culling  (SURVEY.md §8a C4 / §8f N4 — EXTENSION, no reference counterpart)     */
/* ------------------------------------------------------------------------------------ */
/* voidin has no occlusion culling: its README (README.md:33) only links "Two-Pass Occlusion
 * Culling".  These entry points are the building blocks of that scheme on top of the cull
 * path, defined here and nowhere else; they are OFF in every parity run (nothing in
 * vd_cull_* calls them).  All device pointers.
 *
 * Depth convention: the reference's camera (perspective_infinite_reverse_rh, camera.rs:130-148):
 * right-handed view space looking down -z, reverse Z — depth = ndc z, 1 at the near plane, 0 at
 * infinity, cleared to 0; GREATER is nearer.  projection[11] must be -1 and projection[15] 0.
 *
 * Pyramid: level 0 = the depth buffer (width x height, row-major, row 0 = top = ndc y +1);
 * level k texel (x, y) = MIN (= farthest) of the level k-1 texels (2x..2x+1, 2y..2y+1) that
 * exist; level dims halve rounding up until 1 x 1.  vd_hiz_layout gives sizes and offsets
 * (host only, no context); the pyramid buffer holds total_texels floats.
 *
 * vd_occlusion_mask_dev: mask_out = mask_in with the bit of every OCCLUDED instance cleared
 * (bit i of word i/64 = instance i, as vd_cull_mask_dev writes them; in-place allowed).
 * Instance i is occluded when all of this holds, in fp32, in this order, no FMA:
 *   c  = ((view * transform) * vec4((mesh.min + mesh.max) / 2, 1)).xyz      (as emit_draws.wgsl:14-15)
 *   r  = length(mesh.max - mesh.min) * 0.5 * max_scale(transform)  — a TRUE bounding radius
 *        (emit_draws.wgsl:19's radius is kept bug-compatible in the frustum test; it is always
 *        >= this one but usually reaches the camera, which would disable the test)
 *   d = -c.z, dn = d - r;  dn > camera.znear            (whole sphere beyond the near plane)
 *   tangent slopes of the sphere seen from the eye, per axis a in {x, y}, t = sqrt(c.a^2 + d^2 - r^2):
 *     lo = (c.a t - r d) / (d t + c.a r),  hi = (c.a t + r d) / (d t - c.a r),  both divisors > 0
 *   ndc = P[0] * slope_x - P[8],  P[5] * slope_y - P[9];  texel range = ndc -> pixels, widened by
 *     half a texel each side, clipped to the screen (entirely off screen: not occluded)
 *   level = number of bits of max(x1 - x0, y1 - y0) (so the range is <= 2 x 2 texels there)
 *   hmin = min of the four corner texels of the range at that level
 *   (P[14] - P[10] * dn) / dn  <  hmin                  (nearest point of the sphere is farther)
 * Anything NaN fails a comparison and leaves the instance visible.                          */
typedef struct VdHizLayout {
    uint32_t width, height, n_levels, total_texels;
    uint32_t level_offset[17], level_width[17], level_height[17];   /* in texels                */
} VdHizLayout;
int vd_hiz_layout(uint32_t width, uint32_t height, VdHizLayout* out);   /* width * height <= 2^30 */
int vd_hiz_build_dev(VdCtx* ctx, const float* d_depth, uint32_t width, uint32_t height,
                     float* d_pyramid);
int vd_occlusion_mask_dev(VdCtx* ctx, const VdCameraUniform* camera /* host */,
                          const VdMeshInfo* d_meshes, uint32_t n_mesh,
                          const VdInstance* d_instances, uint32_t n_inst,
                          const float* d_pyramid, uint32_t width, uint32_t height,
                          const uint64_t* d_mask_in, uint64_t* d_mask_out);

/* ------------------------------------------------------------------------------------ */
/* Instance animation  (SURVEY.md §8f N2 — the upstream mutator of the cull / TLAS input)  */
/* ------------------------------------------------------------------------------------ */
/* Replaces the `update` compute pass (shaders/compute_update.wgsl:10-28, recorded by
 * ComputeUpdate::record, crates/app/src/pass/compute_update.rs:51-73): for every listed
 * instance id, transform = rotz(speed * dt) * transform with speed = 2*sin(time*0.5), negated
 * when transform[3][2] <= -15.  sin/cos are evaluated once on the host (libm, f32) for the
 * two possible angles, so the device work is plain mul/add and matches the oracle bit for bit
 * (WGSL leaves sin/cos precision to the driver).
 * fix_inverse != 0 additionally keeps inv_transform consistent (inv' = inv * rotz(-angle));
 * the reference leaves it stale.                                   */
int vd_compute_update_dev(VdCtx* ctx, const uint32_t* d_indices, uint32_t n_indices,
                          VdInstance* d_instances, uint32_t n_instances, float time, float dt,
                          int fix_inverse);

/* ------------------------------------------------------------------------------------ */
/* Instrumentation (replaces the wgpu_profiler scopes: visibility.rs:50,243-245)         */
/* ------------------------------------------------------------------------------------ */
/* Opt-in kernel timing: with timing enabled every call brackets its kernels with a HIP event
 * pair on the ctx's stream (costs a few microseconds of GPU idle per call, so it is off by
 * default).  vd_last_gpu_ms returns the milliseconds of the most recent call's kernels
 * (synchronises); negative if timing is off or nothing was recorded.                     */
int   vd_ctx_set_timing(VdCtx* ctx, int enabled);
float vd_last_gpu_ms(VdCtx* ctx);
/* For calls that run two passes (vd_cull_compact* on large inputs: cull-to-bitmask, then
 * expansion): milliseconds of pass `stage` (0 or 1); negative when the call had one pass.  */
float vd_last_gpu_ms_stage(VdCtx* ctx, int stage);
/* Where the most recent vd_bvh_build[_dev] on this ctx spent its time: host wall clock between the
 * synchronisation points the builder has anyway (after the precompute, after the level loop of the
 * large segments, after the mid tier, after the small subtrees, after copy-out + index permute).   */
typedef struct VdBvhBuildStats {
    float    ms_precompute, ms_phase_a, ms_mid, ms_phase_b, ms_phase_c;
    uint32_t levels_phase_a;      /* level-synchronous rounds over segments > 2048 prims */
    uint32_t n_top_nodes, n_mid_roots, n_small_roots;
    uint32_t kernel_launches;     /* kernels enqueued by the build */
} VdBvhBuildStats;
int vd_bvh_last_build_stats(const VdCtx* ctx, VdBvhBuildStats* out);

#ifdef __cplusplus
}
#endif
#endif /* VOIDIN_ABI_H */
