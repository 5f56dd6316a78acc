// GPU check of the RCCL exchange behind the C ABI (include/voidin_abi.h, "Multi-GPU exchange over RCCL") from a plain
// C++ host - no Python, no torch: a ONE-rank communicator runs the whole step (vd_cull_mask_dev -> ncclAllGather on the
// context's stream -> vd_expand_mask_dev) and the result must equal vd_cull_compact_dev on the same instances, byte for
// byte; likewise the literal 20-byte exchange (vd_dist_step_draws_dev).  What a Rust host does per rank, with world = 1.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "voidin_abi.h"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s  [%s]\n", __FILE__, __LINE__, #c, ctx ? vd_last_error(ctx) : ""); return 1; } } while (0)

static uint64_t rng_state = 0x5EED0000C0FFEEull;
static float frand() {   // splitmix64 -> [0, 1)
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)((z >> 40) * (1.0 / 16777216.0));
}

int main(int argc, char** argv) {
    VdCtx* ctx = nullptr;
    CHECK(vd_ctx_create(0, &ctx) == VD_OK);
    const uint32_t n_mesh = 16;
    const uint32_t sizes[2] = {100000u, (2u << 20) + 4321u};     // the fused and the split form of the cull (switch at 2 Mi)
    // camera: looks down -z from (0, 5, 12); only view, frustum, znear, zfar are read by the cull (emit_draws.wgsl:13-33)
    VdCameraUniform cam;
    std::memset(&cam, 0, sizeof(cam));
    cam.view[0] = cam.view[5] = cam.view[10] = cam.view[15] = 1.0f; cam.view[13] = -5.0f; cam.view[14] = -12.0f;
    cam.frustum[0] = 0.6247f; cam.frustum[1] = 0.7809f; cam.frustum[2] = 0.7071f; cam.frustum[3] = 0.7071f;
    cam.zfar = INFINITY; cam.znear = 0.001f;
    std::vector<VdMeshInfo> meshes(n_mesh);
    std::memset(meshes.data(), 0, sizeof(VdMeshInfo) * n_mesh);
    for (uint32_t m = 0; m < n_mesh; ++m) {
        for (int k = 0; k < 3; ++k) { const float c = frand() - 0.5f, h = 0.25f + 1.75f * frand(); meshes[m].min[k] = c - h; meshes[m].max[k] = c + h; }
        meshes[m].index_count = 36u + 3u * m; meshes[m].base_index = 1000u * m; meshes[m].vertex_offset = (int32_t)(77u * m); meshes[m].bvh_index = 10u * m;
    }
    VdMeshInfo* d_meshes = nullptr;
    CHECK(hipMalloc(&d_meshes, sizeof(VdMeshInfo) * n_mesh) == hipSuccess);
    CHECK(hipMemcpy(d_meshes, meshes.data(), sizeof(VdMeshInfo) * n_mesh, hipMemcpyHostToDevice) == hipSuccess);

    unsigned char id[VD_DIST_ID_BYTES];
    const int rc_id = vd_dist_unique_id(id);
    if (rc_id != VD_OK) { std::fprintf(stderr, "FAILED: vd_dist_unique_id -> %d (RCCL not loadable?)\n", rc_id); return 1; }
    VdDist* dist = nullptr;
    VdDist* bad = nullptr;
    CHECK(vd_dist_create(ctx, id, 1, 1, &bad) == VD_ERR_INVALID_ARG && bad == nullptr);     // rank out of range: an error code, not a hang
    CHECK(vd_dist_create(ctx, id, 0, 1, &dist) == VD_OK);

    for (const uint32_t n : sizes) {
        std::vector<VdInstance> inst(n);
        std::memset(inst.data(), 0, sizeof(VdInstance) * (size_t)n);
        for (uint32_t i = 0; i < n; ++i) {
            float* t = inst[i].transform;
            const float s = 0.05f + 0.6f * frand();
            t[0] = s; t[5] = s * (0.5f + frand()); t[10] = s; t[15] = 1.0f;
            t[1] = 0.1f * (frand() - 0.5f); t[6] = 0.1f * (frand() - 0.5f);
            t[12] = 600.0f * (frand() - 0.5f); t[13] = 600.0f * (frand() - 0.5f); t[14] = 600.0f * (frand() - 0.5f);
            inst[i].mesh = (uint32_t)(frand() * 20.0f);                 // some ids beyond the table: the clamp is part of the contract
        }
        VdInstance* d_inst = nullptr; VdDrawIndexedIndirect *d_a = nullptr, *d_b = nullptr; uint32_t* d_cnt = nullptr;
        CHECK(hipMalloc(&d_inst, sizeof(VdInstance) * (size_t)n) == hipSuccess);
        CHECK(hipMalloc(&d_a, 20 * (size_t)n) == hipSuccess && hipMalloc(&d_b, 20 * (size_t)n) == hipSuccess && hipMalloc(&d_cnt, 64) == hipSuccess);
        CHECK(hipMemcpy(d_inst, inst.data(), sizeof(VdInstance) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemset(d_cnt, 0, 64) == hipSuccess);
        CHECK(vd_cull_compact_dev(ctx, &cam, d_meshes, n_mesh, d_inst, n, d_a, d_cnt, 0) == VD_OK);
        CHECK(vd_dist_set_scene_dev(dist, d_inst, n - 1, n, n_mesh) == VD_ERR_INVALID_ARG);   // not this rank's shard size
        CHECK(vd_dist_set_scene_dev(dist, d_inst, n, n, n_mesh) == VD_OK);
        VdDistInfo info;
        CHECK(vd_dist_info(dist, &info) == VD_OK && info.world == 1 && info.shard_size == n && info.n_local == n && info.id_bytes == 1);
        for (int mode = 0; mode < 2; ++mode) {
            CHECK(hipMemsetAsync(d_b, 0xEE, 20 * (size_t)n, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess);
            for (int rep = 0; rep < 2; ++rep)
                CHECK((mode == 0 ? vd_dist_step_full_dev : vd_dist_step_draws_dev)(dist, &cam, d_meshes, n_mesh, d_inst, d_b, d_cnt + 4) == VD_OK);
            CHECK(vd_ctx_synchronize(ctx) == VD_OK);
            uint32_t c[8];
            CHECK(hipMemcpy(c, d_cnt, 32, hipMemcpyDeviceToHost) == hipSuccess);
            CHECK(c[0] == c[4] && c[0] > n / 50 && c[0] < n);
            std::vector<char> a(20 * (size_t)c[0]), b(20 * (size_t)c[0]);
            CHECK(hipMemcpy(a.data(), d_a, a.size(), hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(b.data(), d_b, b.size(), hipMemcpyDeviceToHost) == hipSuccess);
            CHECK(std::memcmp(a.data(), b.data(), a.size()) == 0);
            std::printf("n = %u, %s: %u survivors, identical to vd_cull_compact_dev\n", n, mode == 0 ? "bitmask all-gather + expansion" : "20-byte exchange", c[0]);
        }
        if (n == sizes[0]) std::printf("RCCL %d from %s\n", info.rccl_version, info.rccl_library);
        (void)hipFree(d_inst); (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_cnt);
    }
    CHECK(vd_dist_destroy(dist) == VD_OK);
    CHECK(vd_ctx_destroy(ctx) == VD_OK);
    std::printf("dist_world1_test OK\n");
    (void)argc; (void)argv;
    return 0;
}
