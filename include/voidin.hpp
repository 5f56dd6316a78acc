// voidin.hpp — C++ host-side mirror of the reference's builder / pass API over the C ABI
// (include/voidin_abi.h).  The reference's host code is Rust; this image has no Rust toolchain,
// so the host side above the boundary is written in C++ with the reference's names, argument
// meaning and error behaviour.  Header-only; link with libvoidin_hip.so.
//
//   voidin::BvhBuilder(vertices, indices).build() -> Bvh{nodes}      crates/bvh/src/blas.rs:51-103
//   voidin::Tlas::empty(); tlas.build(instances, meshes); tlas.nodes   crates/bvh/src/tlas.rs:27-85
//   voidin::pass::EmitDraws::record(world, encoder, resources)         crates/app/src/pass/visibility.rs:230-255
//   voidin::MeshPool::add(mesh) / generate_tlas(instances)             crates/pools/src/mesh/mod.rs:279-351
//   voidin::InstancePool::add(instances)                               crates/pools/src/instance.rs:67-79
//
// Error behaviour: the Rust constructors return color_eyre::Result and the builder panics on
// degenerate input; here every failure is a voidin::Error (exception) carrying the VdStatus and
// vd_last_error text — nothing aborts.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <array>
#include <vector>

#include "voidin_abi.h"

namespace voidin {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// glam types as the wire structs use them
struct Vec3 { float x, y, z; };
struct UVec3 { uint32_t x, y, z; };
using Instance = VdInstance;
using MeshInfo = VdMeshInfo;
using DrawIndexedIndirect = VdDrawIndexedIndirect;
using CameraUniform = VdCameraUniform;
using BvhNode = VdBvhNode;
using TlasNode = VdTlasNode;

// Owns the VdCtx (the reference's wgpu device/queue pair inside `World`: app.rs:108-118).
class Gpu {
   public:
    explicit Gpu(int device = 0) {
        int rc = vd_ctx_create(device, &ctx_);
        if (rc != VD_OK) throw Error(rc, "vd_ctx_create failed: no gfx950 device (there is no CPU fallback)");
    }
    ~Gpu() { if (ctx_) vd_ctx_destroy(ctx_); }
    Gpu(const Gpu&) = delete;
    Gpu& operator=(const Gpu&) = delete;
    VdCtx* ctx() const { return ctx_; }
    void check(int rc) const { if (rc != VD_OK) throw Error(rc, vd_last_error(ctx_)); }
    void set_stream(void* hip_stream) { check(vd_ctx_set_stream(ctx_, hip_stream)); }
    void synchronize() { check(vd_ctx_synchronize(ctx_)); }
    // per-context tuning (VdOption in voidin_abi.h); a negative value restores the default
    void set_option(VdOption option, int64_t value) { check(vd_ctx_set_option(ctx_, (int)option, value)); }

   private:
    VdCtx* ctx_ = nullptr;
};

// HIP view of a buffer the renderer owns (ResizableBuffer<T>, crates/components/src/buffer.rs:42-47) that was exported
// as an opaque fd on the Vulkan side: `ptr()` is what the *_dev entry points take as `d_out` / `d_instances`.
class ExternalBuffer {
   public:
    ExternalBuffer(const Gpu& gpu, int opaque_fd, uint64_t size_bytes) : gpu_(gpu) {
        gpu.check(vd_import_external_buffer(gpu.ctx(), opaque_fd, size_bytes, &handle_, &ptr_));   // consumes the fd
    }
    ~ExternalBuffer() { if (handle_) vd_release_external_buffer(gpu_.ctx(), handle_); }
    ExternalBuffer(const ExternalBuffer&) = delete;
    ExternalBuffer& operator=(const ExternalBuffer&) = delete;
    void* ptr() const { return ptr_; }

   private:
    const Gpu& gpu_;
    VdExternalBuffer* handle_ = nullptr;
    void* ptr_ = nullptr;
};

// A semaphore the renderer exported (VK_KHR_external_semaphore_fd) - what orders the frame's queue.submit
// (crates/app/src/app.rs:334-348) against the HIP stream without a CPU wait: wait(frame) ... cull ... signal(frame), all
// enqueued on the context's stream.  `value` is the timeline point; a binary semaphore ignores it.
class ExternalSemaphore {
   public:
    ExternalSemaphore(const Gpu& gpu, int opaque_fd, bool timeline) : gpu_(gpu) {
        gpu.check(vd_import_external_semaphore(gpu.ctx(), opaque_fd, timeline ? 1 : 0, &handle_));   // consumes the fd
    }
    ~ExternalSemaphore() { if (handle_) vd_release_external_semaphore(gpu_.ctx(), handle_); }
    ExternalSemaphore(const ExternalSemaphore&) = delete;
    ExternalSemaphore& operator=(const ExternalSemaphore&) = delete;
    void wait_async(uint64_t value = 0) const { gpu_.check(vd_wait_external_semaphore_async(gpu_.ctx(), handle_, value)); }
    void signal_async(uint64_t value = 0) const { gpu_.check(vd_signal_external_semaphore_async(gpu_.ctx(), handle_, value)); }

   private:
    const Gpu& gpu_;
    VdExternalSemaphore* handle_ = nullptr;
};

// The ordering that works on a runtime without external-semaphore import (voidin_abi.h, "Ordering that WORKS on this
// platform"): a frame word in memory both APIs see, and a host function for the way back.
class FrameWord {
   public:
    FrameWord(const Gpu& gpu, uint32_t* device_word) : gpu_(gpu), word_(device_word) {}
    void wait_async(uint32_t frame) const { gpu_.check(vd_wait_value32_async(gpu_.ctx(), word_, frame)); }       // holds the stream until *word >= frame
    void write_async(uint32_t frame) const { gpu_.check(vd_write_value32_async(gpu_.ctx(), word_, frame)); }
    void then_on_host(VdHostFn callback, void* user) const { gpu_.check(vd_host_callback_async(gpu_.ctx(), callback, user)); }

   private:
    const Gpu& gpu_;
    uint32_t* word_;
};

// ---- crates/bvh ---------------------------------------------------------------------------
enum class DistKind { Hit, Miss };   // enum Dist { Hit(f32), Miss } (crates/bvh/src/intersection.rs:22-26)
struct Dist {
    DistKind kind = DistKind::Miss;
    float t = 0.0f;
    bool is_hit() const { return kind == DistKind::Hit; }
};

struct Bvh {
    std::vector<BvhNode> nodes;   // `pub nodes: Vec<BvhNode>` (blas.rs:206-208)
    // `Bvh::traverse_iter(&self, &[Vec3], &[UVec3], Ray) -> Dist` (blas.rs:247-295), for a batch of rays
    std::vector<Dist> traverse_iter(const Gpu& gpu, const Vec3* vertices, size_t n_vertices, const UVec3* indices, size_t n_triangles,
                                    const std::vector<VdRay>& rays) const {
        std::vector<float> t(rays.size());
        gpu.check(vd_traverse_iter(gpu.ctx(), nodes.data(), (uint32_t)nodes.size(), &vertices->x, (uint32_t)n_vertices, &indices->x,
                                   (uint32_t)n_triangles, rays.data(), (uint32_t)rays.size(), t.data()));
        std::vector<Dist> out(rays.size());
        for (size_t i = 0; i < t.size(); ++i)
            if (t[i] >= 0.0f) out[i] = Dist{DistKind::Hit, t[i]};
        return out;
    }
    // `Bvh::traverse(&self, &[Vec4], &[UVec4], Ray, node_idx, t) -> Dist` (blas.rs:211-245), the recursive walk, from node 0,
    // for a batch of rays; Hit(t0) when the root box is entered and nothing is hit (the reference's quirk)
    std::vector<Dist> traverse(const Gpu& gpu, const Vec3* vertices, size_t n_vertices, const UVec3* indices, size_t n_triangles,
                               const std::vector<VdRay>& rays, float t0 = 1e30f) const {
        std::vector<float> t(rays.size());
        gpu.check(vd_traverse(gpu.ctx(), nodes.data(), (uint32_t)nodes.size(), &vertices->x, (uint32_t)n_vertices, &indices->x,
                              (uint32_t)n_triangles, rays.data(), (uint32_t)rays.size(), t0, t.data()));
        std::vector<Dist> out(rays.size());
        for (size_t i = 0; i < t.size(); ++i)
            if (t[i] >= 0.0f) out[i] = Dist{DistKind::Hit, t[i]};
        return out;
    }
};

// One ray per pixel from camera.clip_to_world, as the CPU harness makes them (src/bin/bvh_cpu.rs:71-83)
inline std::vector<VdRay> primary_rays(const Gpu& gpu, const CameraUniform& camera, uint32_t width, uint32_t height) {
    std::vector<VdRay> rays((size_t)width * height);
    gpu.check(vd_primary_rays(gpu.ctx(), &camera, width, height, rays.data()));
    return rays;
}

// BvhBuilder::new(&[Vec3], &mut [UVec3]) — borrows both slices; build() permutes `indices` in
// place exactly as blas.rs:95-100 does.
class BvhBuilder {
   public:
    BvhBuilder(const Gpu& gpu, const Vec3* vertices, size_t n_vertices, UVec3* indices, size_t n_triangles)
        : gpu_(gpu), vertices_(vertices), n_vertices_(n_vertices), indices_(indices), n_triangles_(n_triangles) {}
    // stored and ignored, like the reference (blas.rs:64-67 vs the hard-coded `let bins = 8` at :136)
    BvhBuilder& set_bin_number(size_t num_bins) { num_bins_ = num_bins; return *this; }
    Bvh build() {
        Bvh bvh;
        bvh.nodes.resize(2 * n_triangles_ > 2 ? 2 * n_triangles_ : 2);   // `vec![BvhNode::default(); indices.len() * 2]`
        uint32_t n_nodes = 0;
        gpu_.check(vd_bvh_build(gpu_.ctx(), &vertices_->x, (uint32_t)n_vertices_, &indices_->x, (uint32_t)n_triangles_,
                                bvh.nodes.data(), (uint32_t)bvh.nodes.size(), &n_nodes));
        bvh.nodes.resize(n_nodes);                                       // `self.nodes.truncate(new_node_index)`
        return bvh;
    }

   private:
    const Gpu& gpu_;
    const Vec3* vertices_; size_t n_vertices_;
    UVec3* indices_; size_t n_triangles_;
    size_t num_bins_ = 8;
};

class Tlas {
   public:
    std::vector<TlasNode> nodes;   // `pub nodes: Vec<TlasNode>` (tlas.rs:22-24)
    static Tlas empty() { return Tlas(); }
    // Tlas::build(&mut self, &[Instance], &[MeshInfo]) (tlas.rs:31)
    void build(const Gpu& gpu, const Instance* instances, size_t n, const MeshInfo* meshes, size_t n_mesh) {
        nodes.assign(2 * n + 1, TlasNode{});
        gpu.check(vd_tlas_build(gpu.ctx(), instances, (uint32_t)n, meshes, (uint32_t)n_mesh, nodes.data()));
    }
    // NEW: same topology, boxes recomputed
    void refit(const Gpu& gpu, const Instance* instances, size_t n, const MeshInfo* meshes, size_t n_mesh) {
        if (nodes.size() != 2 * n + 1) throw Error(VD_ERR_INVALID_ARG, "Tlas::refit: node count does not match instance count");
        gpu.check(vd_tlas_refit(gpu.ctx(), instances, (uint32_t)n, meshes, (uint32_t)n_mesh, nodes.data()));
    }
};

// ---- crates/pools (CPU-side bookkeeping only; GPU buffers stay with the renderer) --------------
struct MeshRef {
    const Vec3* vertices; size_t n_vertices;
    uint32_t* indices; size_t n_indices;          // permuted by add(), like `mesh.indices` in mesh/mod.rs:320-321
};

class MeshPool {
   public:
    std::vector<Vec3> vertices;
    std::vector<uint32_t> indices;
    std::vector<BvhNode> bvh_nodes;
    std::vector<MeshInfo> mesh_info_cpu;
    Tlas tlas = Tlas::empty();

    explicit MeshPool(const Gpu& gpu) : gpu_(gpu) {}

    // MeshPool::add (mesh/mod.rs:309-351): build the BLAS, append to the concatenated buffers,
    // record {min,max,index_count,base_index,vertex_offset,bvh_index}
    uint32_t add(MeshRef mesh) {
        const uint32_t vertex_offset = (uint32_t)vertices.size();
        vertices.insert(vertices.end(), mesh.vertices, mesh.vertices + mesh.n_vertices);
        Bvh bvh = BvhBuilder(gpu_, mesh.vertices, mesh.n_vertices, reinterpret_cast<UVec3*>(mesh.indices), mesh.n_indices / 3).build();
        const uint32_t bvh_index = (uint32_t)bvh_nodes.size();
        bvh_nodes.insert(bvh_nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
        const uint32_t base_index = (uint32_t)indices.size();
        indices.insert(indices.end(), mesh.indices, mesh.indices + mesh.n_indices);
        MeshInfo info{};
        // calculate_bounds (mesh/mod.rs:22-27): fold from (+inf, -inf)
        float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        for (size_t i = 0; i < mesh.n_vertices; ++i) {
            const float p[3] = {mesh.vertices[i].x, mesh.vertices[i].y, mesh.vertices[i].z};
            for (int k = 0; k < 3; ++k) { mn[k] = p[k] < mn[k] ? p[k] : mn[k]; mx[k] = p[k] > mx[k] ? p[k] : mx[k]; }
        }
        std::memcpy(info.min, mn, 12); std::memcpy(info.max, mx, 12);
        info.index_count = (uint32_t)mesh.n_indices;
        info.base_index = base_index;
        info.vertex_offset = (int32_t)vertex_offset;
        info.bvh_index = bvh_index;
        mesh_info_cpu.push_back(info);
        return (uint32_t)mesh_info_cpu.size() - 1;     // MeshId
    }

    // MeshPool::add for a whole scene: the same bookkeeping for every mesh, ONE batched BLAS build (vd_bvh_build_batch)
    // instead of a build per mesh - a load of hundreds of small meshes pays the builder's fixed cost once.  The meshes'
    // index slices are permuted in place, as `add` does.  Returns the MeshId of the first mesh added (ids are consecutive).
    uint32_t add_many(const MeshRef* meshes, size_t n_meshes) {
        const uint32_t first_id = (uint32_t)mesh_info_cpu.size();
        if (n_meshes == 0) return first_id;
        std::vector<VdBvhBatchItem> items(n_meshes);
        size_t node_room = 0;
        for (size_t m = 0; m < n_meshes; ++m) {
            items[m] = VdBvhBatchItem{};
            items[m].verts_xyz = &meshes[m].vertices->x; items[m].n_vert = (uint32_t)meshes[m].n_vertices;
            items[m].indices_inout = meshes[m].indices; items[m].n_tri = (uint32_t)(meshes[m].n_indices / 3);
            items[m].out_nodes = nullptr;                       // packed: bvh_index = bvh_nodes.len() (mesh/mod.rs:320-345)
            node_room += 2 * (meshes[m].n_indices / 3) + 2;     // the reference allocates 2 * T nodes (blas.rs:52)
        }
        const uint32_t first_node = (uint32_t)bvh_nodes.size();
        bvh_nodes.resize(first_node + node_room);
        uint32_t end = first_node;
        const int rc = vd_bvh_build_batch(gpu_.ctx(), items.data(), (uint32_t)n_meshes, bvh_nodes.data(), bvh_nodes.size(), first_node, &end);
        bvh_nodes.resize(rc == VD_OK ? end : first_node);       // nodes.truncate(pool) of every mesh (blas.rs:93); a failed batch leaves the pool as it was
        gpu_.check(rc);
        for (size_t m = 0; m < n_meshes; ++m) {
            const MeshRef& mesh = meshes[m];
            MeshInfo info{};
            float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            for (size_t i = 0; i < mesh.n_vertices; ++i) {
                const float p[3] = {mesh.vertices[i].x, mesh.vertices[i].y, mesh.vertices[i].z};
                for (int k = 0; k < 3; ++k) { mn[k] = p[k] < mn[k] ? p[k] : mn[k]; mx[k] = p[k] > mx[k] ? p[k] : mx[k]; }
            }
            std::memcpy(info.min, mn, 12); std::memcpy(info.max, mx, 12);
            info.index_count = (uint32_t)mesh.n_indices;
            info.base_index = (uint32_t)indices.size();
            info.vertex_offset = (int32_t)vertices.size();
            info.bvh_index = items[m].out_first_node;
            vertices.insert(vertices.end(), mesh.vertices, mesh.vertices + mesh.n_vertices);
            indices.insert(indices.end(), mesh.indices, mesh.indices + mesh.n_indices);
            mesh_info_cpu.push_back(info);
        }
        return first_id;
    }

    // MeshPool::generate_tlas (mesh/mod.rs:279-286)
    void generate_tlas(const std::vector<Instance>& instances) {
        if (instances.empty()) return;
        tlas.build(gpu_, instances.data(), instances.size(), mesh_info_cpu.data(), mesh_info_cpu.size());
    }

    VdTraceScene trace_scene(const std::vector<Instance>& instances) const {
        VdTraceScene s{};
        s.tlas_nodes = tlas.nodes.data(); s.n_tlas_nodes = (uint32_t)tlas.nodes.size();
        s.instances = instances.data(); s.n_instances = (uint32_t)instances.size();
        s.meshes = mesh_info_cpu.data(); s.n_meshes = (uint32_t)mesh_info_cpu.size();
        s.bvh_nodes = bvh_nodes.data(); s.n_bvh_nodes = (uint32_t)bvh_nodes.size();
        s.vertices = &vertices.data()->x; s.n_vertices = (uint32_t)vertices.size();
        s.indices = indices.data(); s.n_indices = (uint32_t)indices.size();
        return s;
    }

   private:
    const Gpu& gpu_;
};

// ---- OBJ ingest --------------------------------------------------------------
// ObjModel::import (crates/app/src/models/mod.rs:19-57) loads with tobj 4.0.0's GPU_LOAD_OPTIONS
// (triangulate + single_index, points and lines ignored) and hands every model's positions / indices
// to MeshPool::add.  tobj is not on disk; this reader restates its documented behaviour (parity
// unpinned): a new model starts at every `o` / `g` statement and when `usemtl` switches material
// after faces were read; polygons are fan-triangulated (v0, vi, vi+1); each distinct v/vt/vn triple
// becomes one vertex, numbered per model in order of first use; indices may be negative (relative).
struct ObjMesh {
    std::string name;
    std::vector<Vec3> positions, normals;
    std::vector<float> texcoords;                    // u, v pairs
    std::vector<uint32_t> indices;
    int material_id = -1;                            // order of `usemtl` names' first appearance (no .mtl parsing)
};

class ObjModel {
   public:
    static std::vector<ObjMesh> load(const std::string& path) {
        std::FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw Error(VD_ERR_INVALID_ARG, "ObjModel: failed to open file: " + path);
        std::vector<Vec3> v, vn;
        std::vector<float> vt;
        std::vector<ObjMesh> out;
        ObjMesh cur; cur.name = "unnamed_object";
        std::map<std::tuple<long, long, long>, uint32_t> seen;
        std::map<std::string, int> materials;
        auto flush = [&](std::string next_name) {
            const int mat = cur.material_id;
            if (!cur.indices.empty()) out.push_back(std::move(cur));
            cur = ObjMesh{}; cur.name = next_name; cur.material_id = mat;
            seen.clear();
        };
        auto rest = [](const char* p) { std::string r(p); while (!r.empty() && (r.back() == '\n' || r.back() == '\r' || r.back() == ' ')) r.pop_back(); return r; };
        char line[4096];
        while (std::fgets(line, sizeof line, f)) {
            const char* p = line;
            while (*p == ' ' || *p == '\t') ++p;
            if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
                char* e; Vec3 a; a.x = std::strtof(p + 2, &e); a.y = std::strtof(e, &e); a.z = std::strtof(e, &e); v.push_back(a);
            } else if (p[0] == 'v' && p[1] == 'n') {
                char* e; Vec3 a; a.x = std::strtof(p + 2, &e); a.y = std::strtof(e, &e); a.z = std::strtof(e, &e); vn.push_back(a);
            } else if (p[0] == 'v' && p[1] == 't') {
                char* e; const float a = std::strtof(p + 2, &e), b = std::strtof(e, &e); vt.push_back(a); vt.push_back(b);
            } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
                std::vector<uint32_t> poly;
                const char* q = p + 1;
                for (;;) {
                    while (*q == ' ' || *q == '\t') ++q;
                    if (*q == '\0' || *q == '\n' || *q == '\r' || *q == '#') break;
                    long idx[3] = {0, 0, 0};           // 1-based as written; 0 = absent
                    for (int k = 0; k < 3; ++k) {
                        char* e; const long t = std::strtol(q, &e, 10);
                        if (e != q) idx[k] = t;
                        q = e;
                        if (*q != '/') break;
                        ++q;
                    }
                    while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;
                    const long nv = (long)v.size(), nt = (long)vt.size() / 2, nn = (long)vn.size();
                    const long iv = idx[0] < 0 ? nv + idx[0] : idx[0] - 1;
                    const long it = idx[1] == 0 ? -1 : (idx[1] < 0 ? nt + idx[1] : idx[1] - 1);
                    const long in = idx[2] == 0 ? -1 : (idx[2] < 0 ? nn + idx[2] : idx[2] - 1);
                    if (iv < 0 || iv >= nv || it >= nt || in >= nn) { std::fclose(f); throw Error(VD_ERR_INVALID_ARG, "ObjModel: face index out of range in " + path); }
                    const auto key = std::make_tuple(iv, it, in);
                    auto hit = seen.find(key);
                    if (hit == seen.end()) {
                        hit = seen.emplace(key, (uint32_t)cur.positions.size()).first;
                        cur.positions.push_back(v[(size_t)iv]);
                        if (it >= 0) { cur.texcoords.push_back(vt[2 * (size_t)it]); cur.texcoords.push_back(vt[2 * (size_t)it + 1]); }
                        if (in >= 0) cur.normals.push_back(vn[(size_t)in]);
                    }
                    poly.push_back(hit->second);
                }
                for (size_t k = 1; k + 1 < poly.size(); ++k) {   // fan; points and lines (< 3 vertices) are dropped
                    cur.indices.push_back(poly[0]); cur.indices.push_back(poly[k]); cur.indices.push_back(poly[k + 1]);
                }
            } else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t')) {
                flush(rest(p + 2));
            } else if (std::strncmp(p, "usemtl", 6) == 0) {
                const std::string name = rest(p + 7);
                const int id = materials.emplace(name, (int)materials.size()).first->second;
                if (id != cur.material_id && !cur.indices.empty()) flush(cur.name);
                cur.material_id = id;
            }
        }
        std::fclose(f);
        if (!cur.indices.empty()) out.push_back(std::move(cur));
        return out;
    }

    // one MeshPool entry (and BLAS) per model, as `app.add_mesh(MeshRef{..})` does in models/mod.rs:40-53
    static std::vector<uint32_t> import(MeshPool& pool, const std::string& path) {
        std::vector<uint32_t> ids;
        for (ObjMesh& m : load(path))
            ids.push_back(pool.add(MeshRef{m.positions.data(), m.positions.size(), m.indices.data(), m.indices.size()}));
        return ids;
    }
};

class InstancePool {
   public:
    std::vector<Instance> instances_data;            // instance.rs:9
    // InstancePool::add (instance.rs:67-79): returns the ids of the appended instances
    std::vector<uint32_t> add(const Instance* instances, size_t n) {
        const uint32_t first = (uint32_t)instances_data.size();
        instances_data.insert(instances_data.end(), instances, instances + n);
        std::vector<uint32_t> ids(n);
        for (size_t i = 0; i < n; ++i) ids[i] = first + (uint32_t)i;
        return ids;
    }
    uint32_t count() const { return (uint32_t)instances_data.size(); }
    void clear() { instances_data.clear(); }
};

// ---- crates/app/src/pass ------------------------------------------------------------------------
// The reference's `World` is a TypeId -> resource map (components/src/world.rs:81-162); the pass
// pulls CameraUniformBinding, MeshPool, InstancePool from it.  Here `World` carries exactly those
// device-side resources.
struct World {
    const CameraUniform* camera;                      // host copy of the 320-byte uniform (camera.rs:13-27)
    const MeshInfo* d_mesh_info; uint32_t n_meshes;   // device buffer (MeshPool::mesh_info)
    const Instance* d_instances; uint32_t n_instances;// device buffer (InstancePool::instances)
};

struct ProfilerCommandEncoder {                       // app.rs:660-703: here just the stream the pass records on
    void* hip_stream = nullptr;
};

namespace pass {

// trait Pass { type Resources<'a>; fn record(&self, &World, &mut ProfilerCommandEncoder, Self::Resources<'_>); }
// (crates/app/src/pass/mod.rs:9-18)
struct EmitDrawsResource {                            // visibility.rs:225-228
    DrawIndexedIndirect* draw_cmd_buffer;             // device buffer, draw_cmd_buffer.len() == n_instances (app.rs:242-246)
    uint32_t* draw_count = nullptr;                   // device u32; non-null => compacted emission
    bool pad_tail = false;
};

class EmitDraws {
   public:
    explicit EmitDraws(Gpu& gpu) : gpu_(gpu) {}       // EmitDraws::new(&World) -> Result<Self> (visibility.rs:200-222)
    // Leaves draw_cmd_buffer[0..N) valid on the encoder's stream before the consumer
    // (Geometry::record, visibility.rs:151-193) runs on the same stream.
    void record(const World& world, ProfilerCommandEncoder& encoder, EmitDrawsResource resources) const {
        gpu_.set_stream(encoder.hip_stream);
        if (resources.draw_count)
            gpu_.check(vd_cull_compact_dev(gpu_.ctx(), world.camera, world.d_mesh_info, world.n_meshes, world.d_instances,
                                           world.n_instances, resources.draw_cmd_buffer, resources.draw_count,
                                           resources.pad_tail ? 1 : 0));
        else
            gpu_.check(vd_cull_emit_dev(gpu_.ctx(), world.camera, world.d_mesh_info, world.n_meshes, world.d_instances,
                                        world.n_instances, resources.draw_cmd_buffer));
    }

   private:
    Gpu& gpu_;
};

}  // namespace pass

// `traverse_tlas(ray)` for a batch (shaders/utils/bvh.wgsl:89-123)
inline std::vector<VdHit> traverse_tlas(const Gpu& gpu, const VdTraceScene& scene, const std::vector<VdRay>& rays) {
    std::vector<VdHit> out(rays.size());
    gpu.check(vd_trace(gpu.ctx(), &scene, rays.data(), (uint32_t)rays.size(), out.data()));
    return out;
}

// The trace bind group the reference builds once per scene (app.rs:255-287): the six device buffers of a scene plus the
// de-indexed leaf triangles vd_trace_prepare_dev derives from them.  Rebuild it where that bind group is rebuilt (meshes,
// vertices or indices changed); instances and TLAS nodes may change between calls.  All pointers are device memory.
class TraceScene {
   public:
    TraceScene(const Gpu& gpu, const VdTraceScene& d_scene) : gpu_(gpu) { gpu.check(vd_trace_prepare_dev(gpu.ctx(), &d_scene, &accel_)); }
    ~TraceScene() { if (accel_) vd_trace_release(gpu_.ctx(), accel_); }
    TraceScene(const TraceScene&) = delete;
    TraceScene& operator=(const TraceScene&) = delete;
    // traverse_tlas(ray) for a batch (bvh.wgsl:89-123); d_out: n_rays hits
    void trace(const VdRay* d_rays, uint32_t n_rays, VdHit* d_out) const { gpu_.check(vd_trace_prepared_dev(gpu_.ctx(), accel_, d_rays, n_rays, d_out)); }
    // the shadow pass's test (raytraced_shadows.wgsl:97-102): d_hit[i] = traverse_tlas(ray i).hit
    void trace_any(const VdRay* d_rays, uint32_t n_rays, uint32_t* d_hit) const {
        gpu_.check(vd_trace_any_prepared_dev(gpu_.ctx(), accel_, d_rays, n_rays, d_hit));
    }
    // Opt-in (Gpu::set_option(VD_OPT_TRACE_TIGHT_TLAS, 1 or 2) before construction): the scene walks a private top level over
    // tight world boxes - same hits, distances within 1e-5 (bit-equal in every test), not the reference's visit order.
    // `info().tight_tlas` says whether it does (every instance must have inv_transform = transform^-1); `update()` rebuilds it
    // from the instance buffer after the instances moved (compute_update.wgsl:10-28) - 2 = LBVH is the builder for that.
    VdTraceAccelInfo info() const { VdTraceAccelInfo i{}; gpu_.check(vd_trace_accel_info(accel_, &i)); return i; }
    void update() { gpu_.check(vd_trace_accel_update_dev(gpu_.ctx(), accel_)); }

   private:
    const Gpu& gpu_;
    VdTraceAccel* accel_ = nullptr;
};

// EmitDraws over an instance-sharded scene, one process per GPU (INTEGRATION.md 7 has the Rust form).  Rank 0
// makes the communicator id and hands it to the others through the host's own channel; every rank then constructs this
// with its rank.  `record` = cull the own shard -> RCCL all-gather (1 bit per instance) on the context's stream -> expand:
// every GPU ends with the ordered draw list of the whole scene, no host round trip.
class DistEmitDraws {
   public:
    using Id = std::array<uint8_t, VD_DIST_ID_BYTES>;
    static Id unique_id() {
        Id id{};
        const int rc = vd_dist_unique_id(id.data());
        if (rc != VD_OK) throw Error(rc, "vd_dist_unique_id: RCCL could not be loaded");
        return id;
    }
    DistEmitDraws(Gpu& gpu, const Id& id, int rank, int world) : gpu_(gpu) { gpu.check(vd_dist_create(gpu.ctx(), id.data(), rank, world, &dist_)); }
    ~DistEmitDraws() { if (dist_) vd_dist_destroy(dist_); }
    DistEmitDraws(const DistEmitDraws&) = delete;
    DistEmitDraws& operator=(const DistEmitDraws&) = delete;
    VdDistInfo info() const { VdDistInfo i{}; gpu_.check(vd_dist_info(dist_, &i)); return i; }
    // per scene (mesh assignment is static in the reference's scenes): this rank's shard of n_total instances
    void set_scene(const Instance* d_shard_instances, uint32_t n_local, uint32_t n_total, uint32_t n_mesh) {
        gpu_.check(vd_dist_set_scene_dev(dist_, d_shard_instances, n_local, n_total, n_mesh));
    }
    // per frame: d_out holds n_total commands, *d_out_count the survivors of the whole scene
    void record(const CameraUniform& camera, const MeshInfo* d_meshes, uint32_t n_mesh, const Instance* d_shard_instances,
                DrawIndexedIndirect* d_out, uint32_t* d_out_count) {
        gpu_.check(vd_dist_step_full_dev(dist_, &camera, d_meshes, n_mesh, d_shard_instances, d_out, d_out_count));
    }
    // the literal exchange of the 20-byte commands (blocks: the sizes are data dependent)
    void record_draws(const CameraUniform& camera, const MeshInfo* d_meshes, uint32_t n_mesh, const Instance* d_shard_instances,
                      DrawIndexedIndirect* d_out, uint32_t* d_out_count) {
        gpu_.check(vd_dist_step_draws_dev(dist_, &camera, d_meshes, n_mesh, d_shard_instances, d_out, d_out_count));
    }
    // the same exchange with 4-byte survivor indices on the wire (SURVEY.md 8e's option; blocks like record_draws)
    void record_indices(const CameraUniform& camera, const MeshInfo* d_meshes, uint32_t n_mesh, const Instance* d_shard_instances,
                        DrawIndexedIndirect* d_out, uint32_t* d_out_count) {
        gpu_.check(vd_dist_step_indices_dev(dist_, &camera, d_meshes, n_mesh, d_shard_instances, d_out, d_out_count));
    }

   private:
    Gpu& gpu_;
    VdDist* dist_ = nullptr;
};

// EXTENSION (no reference counterpart: README.md:33 only links "Two-Pass Occlusion Culling"): the second pass of that
// scheme.  `build` turns the depth buffer the first pass rendered into a min pyramid; `refine` clears, in a frustum
// mask made by vd_cull_mask_dev, the bits of instances hidden behind it; vd_expand_mask_dev then makes the draw list.
class HizPyramid {
   public:
    HizPyramid(const Gpu& gpu, uint32_t width, uint32_t height) : gpu_(gpu) { gpu.check(vd_hiz_layout(width, height, &layout_)); }
    const VdHizLayout& layout() const { return layout_; }
    size_t bytes() const { return (size_t)layout_.total_texels * sizeof(float); }
    // d_pyramid: bytes() of device memory owned by the caller
    void build(const float* d_depth, float* d_pyramid) const {
        gpu_.check(vd_hiz_build_dev(gpu_.ctx(), d_depth, layout_.width, layout_.height, d_pyramid));
    }
    void refine(const CameraUniform& camera, const MeshInfo* d_meshes, uint32_t n_meshes, const Instance* d_instances, uint32_t n_instances,
                const float* d_pyramid, const uint64_t* d_mask_in, uint64_t* d_mask_out) const {
        gpu_.check(vd_occlusion_mask_dev(gpu_.ctx(), &camera, d_meshes, n_meshes, d_instances, n_instances, d_pyramid, layout_.width,
                                         layout_.height, d_mask_in, d_mask_out));
    }

   private:
    const Gpu& gpu_;
    VdHizLayout layout_{};
};

// The shadow pass's occlusion test (src/bin/raytraced_shadows.wgsl:90-102) for one point light: one ray per G-buffer
// point, `occluded[i]` = traverse_tlas(ray).hit.  Scene pointers, positions and normals are device memory.
inline void shadow_occlusion(const Gpu& gpu, const VdTraceScene& d_scene, const float* d_positions, const float* d_normals,
                             uint32_t n_points, const float light_position[3], VdRay* d_rays_scratch, uint32_t* d_occluded) {
    gpu.check(vd_shadow_rays_dev(gpu.ctx(), d_positions, d_normals, n_points, light_position, d_rays_scratch));
    gpu.check(vd_trace_any_dev(gpu.ctx(), &d_scene, d_rays_scratch, n_points, d_occluded));
}

}  // namespace voidin
