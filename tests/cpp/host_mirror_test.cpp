// host_mirror_test.cpp — drives the path through the C++ mirror of the reference API
// (include/voidin.hpp) the way the reference's own call sites do, and checks every result
// against the CPU oracle (tests may link it; the product does not).
//   MeshPool::add -> BvhBuilder::new(..).build()         crates/pools/src/mesh/mod.rs:309-351
//   MeshPool::generate_tlas -> Tlas::build               crates/pools/src/mesh/mod.rs:279-286
//   EmitDraws::record                                    crates/app/src/pass/visibility.rs:233-254
//   traverse_tlas                                        shaders/utils/bvh.wgsl:89-123
// Build: hipcc --offload-arch=gfx950 -I include tests/cpp/host_mirror_test.cpp -L... -lvoidin_hip -lvd_oracle
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/voidin.hpp"
#include "../../oracle/vd_oracle.h"

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static uint64_t rng_state = 0x5EED0042ull;
static float frand() {   // splitmix64 -> [0,1)
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

int main() {
    voidin::Gpu gpu(0);
    voidin::MeshPool pool(gpu);

    // make_plane_mesh(1,1) (crates/pools/src/mesh/plane.rs:5-38) and a bvh_cpu.rs-style soup
    std::vector<voidin::Vec3> plane_v = {{-.5f, 0, -.5f}, {-.5f, 0, .5f}, {.5f, 0, .5f}, {.5f, 0, -.5f}};
    std::vector<uint32_t> plane_i = {0, 1, 2, 0, 2, 3};
    std::vector<voidin::Vec3> soup_v; std::vector<uint32_t> soup_i;
    for (int t = 0; t < 900; ++t) {
        voidin::Vec3 b{frand() * 9 - 5, frand() * 9 - 5, frand() * 9};
        soup_v.push_back(b);
        soup_v.push_back({b.x + frand(), b.y + frand(), b.z + frand()});
        soup_v.push_back({b.x + frand(), b.y + frand(), b.z + frand()});
        for (int k = 0; k < 3; ++k) soup_i.push_back(3 * t + k);
    }
    // oracle BLAS on copies (the builder permutes the caller's indices)
    std::vector<uint32_t> ref_i = soup_i;
    std::vector<VdBvhNode> ref_nodes(2 * 900);
    uint32_t ref_n = 0;
    REQUIRE(vd_ref_bvh_build(&soup_v[0].x, (uint32_t)soup_v.size(), ref_i.data(), 900, ref_nodes.data(), 1800, &ref_n) == VD_OK);

    const uint32_t plane_id = pool.add({plane_v.data(), plane_v.size(), plane_i.data(), plane_i.size()});
    const uint32_t soup_id = pool.add({soup_v.data(), soup_v.size(), soup_i.data(), soup_i.size()});
    REQUIRE(plane_id == 0 && soup_id == 1);
    REQUIRE(pool.mesh_info_cpu[1].bvh_index == 2 && pool.mesh_info_cpu[1].base_index == 6 && pool.mesh_info_cpu[1].vertex_offset == 4);
    REQUIRE(pool.bvh_nodes.size() == 2 + ref_n);
    REQUIRE(std::memcmp(pool.bvh_nodes.data() + 2, ref_nodes.data(), ref_n * sizeof(VdBvhNode)) == 0);
    REQUIRE(soup_i == ref_i);
    {
        // MeshPool::add_many: the same two meshes (and the plane once more) through ONE batched build == add() three times
        voidin::MeshPool pool2(gpu), pool3(gpu);
        std::vector<uint32_t> pi_a = {0, 1, 2, 0, 2, 3}, pi_b = pi_a, pi_c = pi_a, pi_d = pi_a, si_a, si_b;
        for (int t = 0; t < 900; ++t) for (int k = 0; k < 3; ++k) { si_a.push_back(3 * t + k); si_b.push_back(3 * t + k); }
        pool2.add({plane_v.data(), plane_v.size(), pi_a.data(), pi_a.size()});
        pool2.add({soup_v.data(), soup_v.size(), si_a.data(), si_a.size()});
        pool2.add({plane_v.data(), plane_v.size(), pi_b.data(), pi_b.size()});
        const voidin::MeshRef many[3] = {{plane_v.data(), plane_v.size(), pi_c.data(), pi_c.size()}, {soup_v.data(), soup_v.size(), si_b.data(), si_b.size()},
                                         {plane_v.data(), plane_v.size(), pi_d.data(), pi_d.size()}};
        REQUIRE(pool3.add_many(many, 3) == 0);
        REQUIRE(pool3.mesh_info_cpu.size() == 3 && pool3.bvh_nodes.size() == pool2.bvh_nodes.size() && pool3.indices == pool2.indices);
        REQUIRE(std::memcmp(pool3.bvh_nodes.data(), pool2.bvh_nodes.data(), pool2.bvh_nodes.size() * sizeof(VdBvhNode)) == 0);
        REQUIRE(std::memcmp(pool3.mesh_info_cpu.data(), pool2.mesh_info_cpu.data(), 3 * sizeof(voidin::MeshInfo)) == 0);
        REQUIRE(si_b == ref_i);
    }

    // instances: translate + uniform scale, Instance::new computes inv_transform (shared.rs:90-98)
    voidin::InstancePool ipool;
    std::vector<voidin::Instance> inst(300);
    for (size_t i = 0; i < inst.size(); ++i) {
        std::memset(&inst[i], 0, sizeof(inst[i]));
        const float s = 0.5f + 2.0f * frand(), tx = frand() * 80 - 40, ty = frand() * 80 - 40, tz = frand() * 80 - 40;
        float* T = inst[i].transform; float* I = inst[i].inv_transform;
        T[0] = T[5] = T[10] = s; T[15] = 1; T[12] = tx; T[13] = ty; T[14] = tz;
        I[0] = I[5] = I[10] = 1.0f / s; I[15] = 1; I[12] = -tx / s; I[13] = -ty / s; I[14] = -tz / s;
        inst[i].mesh = (i % 3) ? 1 : 0; inst[i].material = 1;
    }
    ipool.add(inst.data(), inst.size());
    pool.generate_tlas(ipool.instances_data);
    std::vector<VdTlasNode> ref_tlas(2 * inst.size() + 1);
    REQUIRE(vd_ref_tlas_build(inst.data(), (uint32_t)inst.size(), pool.mesh_info_cpu.data(), 2, ref_tlas.data()) == VD_OK);
    REQUIRE(std::memcmp(ref_tlas.data(), pool.tlas.nodes.data(), ref_tlas.size() * sizeof(VdTlasNode)) == 0);

    // rays through the scene
    std::vector<VdRay> rays(4096);
    for (auto& r : rays) {
        std::memset(&r, 0, sizeof(r));
        r.eye[0] = frand() * 20 - 10; r.eye[1] = frand() * 20 - 10; r.eye[2] = 90;
        float d[3] = {frand() - .5f, frand() - .5f, -1.0f - frand()};
        const float l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int k = 0; k < 3; ++k) r.dir[k] = d[k] / l;
    }
    const VdTraceScene scene = pool.trace_scene(ipool.instances_data);
    std::vector<VdHit> hits = voidin::traverse_tlas(gpu, scene, rays);
    std::vector<VdHit> ref_hits(rays.size());
    uint32_t max_stack = 0;
    REQUIRE(vd_ref_trace(&scene, rays.data(), (uint32_t)rays.size(), ref_hits.data(), &max_stack, 1) == VD_OK);
    size_t n_hit = 0;
    for (size_t i = 0; i < rays.size(); ++i) {
        REQUIRE(hits[i].hit == ref_hits[i].hit);
        if (hits[i].hit) { ++n_hit; REQUIRE(std::fabs(hits[i].dist - ref_hits[i].dist) <= 1e-5f * std::fabs(ref_hits[i].dist)); }
    }
    REQUIRE(n_hit > 10);

    // EmitDraws::record on device buffers, then the consumer-side contract: draw_cmd_buffer[0..N) valid
    VdCameraUniform cam; std::memset(&cam, 0, sizeof(cam));
    for (int k = 0; k < 4; ++k) cam.view[5 * k] = 1.0f;       // identity view
    cam.view[14] = -60.0f;                                    // camera at z = +60 looking down -Z
    cam.frustum[0] = 0.7808688f; cam.frustum[1] = -0.6246950f; cam.frustum[2] = 0.7071068f; cam.frustum[3] = -0.7071068f;
    cam.zfar = INFINITY; cam.znear = 0.001f;
    voidin::Instance* d_inst; voidin::MeshInfo* d_mesh; voidin::DrawIndexedIndirect* d_draw; uint32_t* d_count;
    REQUIRE(hipMalloc(&d_inst, inst.size() * sizeof(voidin::Instance)) == hipSuccess);
    REQUIRE(hipMalloc(&d_mesh, 2 * sizeof(voidin::MeshInfo)) == hipSuccess);
    REQUIRE(hipMalloc(&d_draw, inst.size() * sizeof(voidin::DrawIndexedIndirect)) == hipSuccess);
    REQUIRE(hipMalloc(&d_count, 16) == hipSuccess);
    REQUIRE(hipMemcpy(d_inst, inst.data(), inst.size() * sizeof(voidin::Instance), hipMemcpyHostToDevice) == hipSuccess);
    REQUIRE(hipMemcpy(d_mesh, pool.mesh_info_cpu.data(), 2 * sizeof(voidin::MeshInfo), hipMemcpyHostToDevice) == hipSuccess);
    hipStream_t stream; REQUIRE(hipStreamCreate(&stream) == hipSuccess);
    voidin::World world{&cam, d_mesh, 2, d_inst, (uint32_t)inst.size()};
    voidin::ProfilerCommandEncoder encoder{stream};
    voidin::pass::EmitDraws emit_draws(gpu);
    emit_draws.record(world, encoder, {d_draw});
    std::vector<voidin::DrawIndexedIndirect> draws(inst.size()), ref_draws(inst.size());
    REQUIRE(hipMemcpyAsync(draws.data(), d_draw, draws.size() * sizeof(draws[0]), hipMemcpyDeviceToHost, stream) == hipSuccess);
    REQUIRE(hipStreamSynchronize(stream) == hipSuccess);
    REQUIRE(vd_ref_cull_emit(&cam, pool.mesh_info_cpu.data(), 2, inst.data(), (uint32_t)inst.size(), ref_draws.data(), 1) == VD_OK);
    REQUIRE(std::memcmp(draws.data(), ref_draws.data(), draws.size() * sizeof(draws[0])) == 0);
    // compacted emission
    emit_draws.record(world, encoder, {d_draw, d_count, true});
    uint32_t count = 0, ref_count = 0;
    REQUIRE(hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, stream) == hipSuccess);
    REQUIRE(hipMemcpyAsync(draws.data(), d_draw, draws.size() * sizeof(draws[0]), hipMemcpyDeviceToHost, stream) == hipSuccess);
    REQUIRE(hipStreamSynchronize(stream) == hipSuccess);
    std::vector<voidin::DrawIndexedIndirect> ref_comp(inst.size());
    REQUIRE(vd_ref_compact(ref_draws.data(), (uint32_t)inst.size(), ref_comp.data(), &ref_count, 1) == VD_OK);
    REQUIRE(count == ref_count && std::memcmp(draws.data(), ref_comp.data(), draws.size() * sizeof(draws[0])) == 0);

    // occlusion extension through the mirror: frustum mask -> pyramid of a near wall -> refined mask, against the twin
    {
        VdCameraUniform oc = cam;
        oc.projection[0] = 0.8f; oc.projection[5] = 1.0f; oc.projection[11] = -1.0f; oc.projection[14] = 0.001f;
        const uint32_t W = 320, H = 200;
        voidin::HizPyramid hiz(gpu, W, H);
        std::vector<float> depth((size_t)W * H, 0.0f);
        for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W / 2; ++x) depth[(size_t)y * W + x] = 0.001f / 30.0f;   // wall 30 units away, left half
        float *d_depth, *d_pyr; uint64_t* d_mask;
        const size_t words = (inst.size() + 63) / 64;
        REQUIRE(hipMalloc(&d_depth, depth.size() * 4) == hipSuccess && hipMalloc(&d_pyr, hiz.bytes()) == hipSuccess);
        REQUIRE(hipMalloc(&d_mask, words * 8) == hipSuccess);
        REQUIRE(hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(vd_cull_mask_dev(gpu.ctx(), &oc, d_mesh, 2, d_inst, (uint32_t)inst.size(), d_mask) == VD_OK);
        std::vector<uint64_t> frustum(words), refined(words), ref_refined(words);
        REQUIRE(hipMemcpy(frustum.data(), d_mask, words * 8, hipMemcpyDeviceToHost) == hipSuccess);
        hiz.build(d_depth, d_pyr);
        hiz.refine(oc, d_mesh, 2, d_inst, (uint32_t)inst.size(), d_pyr, d_mask, d_mask);
        REQUIRE(hipMemcpy(refined.data(), d_mask, words * 8, hipMemcpyDeviceToHost) == hipSuccess);
        std::vector<float> ref_pyr(hiz.layout().total_texels);
        REQUIRE(vd_ref_hiz_build(depth.data(), W, H, ref_pyr.data()) == VD_OK);
        REQUIRE(vd_ref_occlusion_mask(&oc, pool.mesh_info_cpu.data(), 2, inst.data(), (uint32_t)inst.size(), ref_pyr.data(), W, H,
                                      frustum.data(), ref_refined.data()) == VD_OK);
        REQUIRE(refined == ref_refined);
        size_t a = 0, b = 0;
        for (size_t w = 0; w < words; ++w) { a += __builtin_popcountll(frustum[w]); b += __builtin_popcountll(refined[w]); }
        REQUIRE(b < a && b > 0);
        hipFree(d_depth); hipFree(d_pyr); hipFree(d_mask);
    }

    // the CPU harness (src/bin/bvh_cpu.rs:71-96): per-pixel rays from clip_to_world, Bvh::traverse_iter per ray.
    // clip_to_world here: eye = (x, y, 15), dir = normalize(x, y, -1)
    {
        VdCameraUniform hc; std::memset(&hc, 0, sizeof(hc));
        hc.clip_to_world[0] = 1.0f; hc.clip_to_world[5] = 1.0f;                 // columns 0, 1
        hc.clip_to_world[10] = 16.0f; hc.clip_to_world[11] = 1.0f;              // column 2 = (0, 0, 16, 1)
        hc.clip_to_world[14] = -1.0f;                                           // column 3 = (0, 0, -1, 0)
        const uint32_t W = 96, H = 96;
        std::vector<VdRay> prim = voidin::primary_rays(gpu, hc, W, H), ref_prim(W * H);
        REQUIRE(vd_ref_primary_rays(&hc, W, H, ref_prim.data()) == VD_OK);
        REQUIRE(std::memcmp(prim.data(), ref_prim.data(), prim.size() * sizeof(VdRay)) == 0);
        voidin::Bvh soup_bvh;
        soup_bvh.nodes.assign(ref_nodes.begin(), ref_nodes.begin() + ref_n);
        std::vector<voidin::Dist> dist = soup_bvh.traverse_iter(gpu, soup_v.data(), soup_v.size(),
                                                                reinterpret_cast<const voidin::UVec3*>(soup_i.data()), 900, prim);
        std::vector<float> ref_dist(prim.size());
        REQUIRE(vd_ref_traverse_iter(ref_nodes.data(), ref_n, &soup_v[0].x, ref_i.data(), ref_prim.data(), W * H, ref_dist.data()) == VD_OK);
        size_t n_h = 0;
        for (size_t i = 0; i < dist.size(); ++i) {
            REQUIRE(dist[i].is_hit() == (ref_dist[i] >= 0.0f));
            if (dist[i].is_hit()) { REQUIRE(std::memcmp(&dist[i].t, &ref_dist[i], 4) == 0); ++n_h; }
        }
        REQUIRE(n_h > 100 && n_h < dist.size());
    }

    // round 3: the prepared scene (TraceScene = the trace bind group built once per scene), a per-context option, and the
    // sharded EmitDraws with one rank - everything on device copies of the scene's buffers
    {
        auto up = [&](const void* h, size_t bytes) { void* d = nullptr; if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) std::abort();
                                                      if (h && bytes && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) std::abort(); return d; };
        VdTraceScene ds = scene;
        ds.tlas_nodes = (const VdTlasNode*)up(scene.tlas_nodes, sizeof(VdTlasNode) * scene.n_tlas_nodes);
        ds.instances = (const VdInstance*)up(scene.instances, sizeof(VdInstance) * scene.n_instances);
        ds.meshes = (const VdMeshInfo*)up(scene.meshes, sizeof(VdMeshInfo) * scene.n_meshes);
        ds.bvh_nodes = (const VdBvhNode*)up(scene.bvh_nodes, sizeof(VdBvhNode) * scene.n_bvh_nodes);
        ds.vertices = (const float*)up(scene.vertices, 12 * (size_t)scene.n_vertices);
        ds.indices = (const uint32_t*)up(scene.indices, 4 * (size_t)scene.n_indices);
        VdRay* d_rays = (VdRay*)up(rays.data(), sizeof(VdRay) * rays.size());
        VdHit* d_hits = (VdHit*)up(nullptr, sizeof(VdHit) * rays.size());
        uint32_t* d_any = (uint32_t*)up(nullptr, 4 * rays.size());
        std::vector<VdHit> got(rays.size());
        std::vector<uint32_t> any(rays.size());
        {
            voidin::TraceScene ts(gpu, ds);
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) gpu.set_option(VD_OPT_TRACE_YIELD, 8);              // when a wave serves its waiting lanes: order only
                ts.trace(d_rays, (uint32_t)rays.size(), d_hits);
                ts.trace_any(d_rays, (uint32_t)rays.size(), d_any);
                gpu.synchronize();
                REQUIRE(hipMemcpy(got.data(), d_hits, sizeof(VdHit) * rays.size(), hipMemcpyDeviceToHost) == hipSuccess);
                REQUIRE(hipMemcpy(any.data(), d_any, 4 * rays.size(), hipMemcpyDeviceToHost) == hipSuccess);
                for (size_t i = 0; i < rays.size(); ++i) {
                    REQUIRE(got[i].hit == hits[i].hit && any[i] == hits[i].hit);
                    if (got[i].hit) REQUIRE(std::memcmp(&got[i], &hits[i], sizeof(VdHit)) == 0);      // the host-pointer call's bytes
                }
            }
            gpu.set_option(VD_OPT_TRACE_YIELD, -1);
        }
        // round 4: the opt-in private top level (agglomerative, then LBVH + an update): the instances here are translate + uniform scale
        // with exact inverses, so every instance qualifies; hit flags and distances must be the exact walk's
        for (int mode = 1; mode <= 2; ++mode) {
            gpu.set_option(VD_OPT_TRACE_TIGHT_TLAS, mode);
            voidin::TraceScene ts(gpu, ds);
            gpu.set_option(VD_OPT_TRACE_TIGHT_TLAS, -1);
            REQUIRE(ts.info().tight_tlas == (uint32_t)mode && ts.info().tight_fallback_instances == 0);
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) ts.update();                                         // same instances: the same top level again
                ts.trace(d_rays, (uint32_t)rays.size(), d_hits);
                ts.trace_any(d_rays, (uint32_t)rays.size(), d_any);
                gpu.synchronize();
                REQUIRE(hipMemcpy(got.data(), d_hits, sizeof(VdHit) * rays.size(), hipMemcpyDeviceToHost) == hipSuccess);
                REQUIRE(hipMemcpy(any.data(), d_any, 4 * rays.size(), hipMemcpyDeviceToHost) == hipSuccess);
                for (size_t i = 0; i < rays.size(); ++i) {
                    REQUIRE(got[i].hit == hits[i].hit && any[i] == hits[i].hit);
                    if (got[i].hit) REQUIRE(std::fabs(got[i].dist - hits[i].dist) <= 1e-5f * std::fabs(hits[i].dist));
                }
            }
        }
        // one-rank DistEmitDraws == EmitDraws' compacted list (the RCCL library is bound at run time; skipped when it is not there)
        bool have_rccl = true;
        voidin::DistEmitDraws::Id id{};
        try { id = voidin::DistEmitDraws::unique_id(); } catch (const voidin::Error& e) { have_rccl = false; REQUIRE(e.code == VD_ERR_COMM); }
        if (have_rccl) {
            voidin::DistEmitDraws dd(gpu, id, 0, 1);
            REQUIRE(dd.info().world == 1);
            const uint32_t n = (uint32_t)inst.size();
            VdDrawIndexedIndirect* d_a = (VdDrawIndexedIndirect*)up(nullptr, 20 * (size_t)n);
            VdDrawIndexedIndirect* d_b = (VdDrawIndexedIndirect*)up(nullptr, 20 * (size_t)n);
            uint32_t* d_cnt = (uint32_t*)up(nullptr, 16);
            dd.set_scene(ds.instances, n, n, scene.n_meshes);
            dd.record(cam, ds.meshes, scene.n_meshes, ds.instances, d_a, d_cnt);
            gpu.check(vd_cull_compact_dev(gpu.ctx(), &cam, ds.meshes, scene.n_meshes, ds.instances, n, d_b, d_cnt + 1, 0));
            gpu.synchronize();
            uint32_t c2[2] = {0, 0};
            REQUIRE(hipMemcpy(c2, d_cnt, 8, hipMemcpyDeviceToHost) == hipSuccess);
            REQUIRE(c2[0] == c2[1] && c2[0] == count);
            std::vector<char> la(20 * (size_t)c2[0]), lb(20 * (size_t)c2[0]);
            REQUIRE(hipMemcpy(la.data(), d_a, la.size(), hipMemcpyDeviceToHost) == hipSuccess);
            REQUIRE(hipMemcpy(lb.data(), d_b, lb.size(), hipMemcpyDeviceToHost) == hipSuccess);
            REQUIRE(la == lb);
            hipFree(d_a); hipFree(d_b); hipFree(d_cnt);
        }
        std::printf("round-3 mirror: prepared trace == vd_trace (two yields), DistEmitDraws(world 1) %s\n", have_rccl ? "== EmitDraws" : "skipped (no RCCL)");
    }

    // error behaviour: degenerate input is an error code, not a crash (blas.rs:137-140 would panic)
    std::vector<voidin::Vec3> tv = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}};
    std::vector<uint32_t> ti; for (int k = 0; k < 5; ++k) { ti.push_back(0); ti.push_back(1); ti.push_back(2); }
    bool threw = false;
    try { voidin::BvhBuilder(gpu, tv.data(), 3, reinterpret_cast<voidin::UVec3*>(ti.data()), 5).build(); }
    catch (const voidin::Error& e) { threw = e.code == VD_ERR_DEGENERATE; }
    REQUIRE(threw);
    std::printf("host_mirror_test OK: %zu hits, %u of %zu instances visible\n", n_hit, count, inst.size());
    return 0;
}
