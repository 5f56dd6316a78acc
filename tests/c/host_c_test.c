/* The drop-in boundary from a plain C11 host, compiled by gcc (no hipcc, no C++): what a cgo / Rust FFI / C caller sees.
 * Reads a scene + the oracle's expected bytes from a file written by tests/test_c_host.py, runs it through the C ABI -
 * vd_cull_emit, vd_cull_compact (with and without pad_tail), vd_bvh_build, vd_tlas_build - and compares byte for byte.
 *   file: u32 n_mesh, n_inst, n_vert, n_tri, n_nodes, count; camera[320]; meshes; instances; draws[n_inst]; compact[count];
 *         verts[3 n_vert] f32; indices_in[3 n_tri]; nodes[n_nodes]; indices_out[3 n_tri]; tlas[2 n_inst + 1]                */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "voidin_abi.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, ctx ? vd_last_error(ctx) : ""); return 1; } } while (0)

static void* rd(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read (%zu bytes)\n", bytes); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    VdCtx* ctx = NULL;
    if (argc < 2) { fprintf(stderr, "usage: host_c_test scene.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    uint32_t h[6];
    if (fread(h, 4, 6, f) != 6) return 2;
    const uint32_t n_mesh = h[0], n_inst = h[1], n_vert = h[2], n_tri = h[3], n_nodes = h[4], count = h[5];
    VdCameraUniform* cam = rd(f, sizeof(VdCameraUniform));
    VdMeshInfo* meshes = rd(f, sizeof(VdMeshInfo) * n_mesh);
    VdInstance* inst = rd(f, sizeof(VdInstance) * n_inst);
    VdDrawIndexedIndirect* want_draws = rd(f, sizeof(VdDrawIndexedIndirect) * n_inst);
    VdDrawIndexedIndirect* want_compact = rd(f, sizeof(VdDrawIndexedIndirect) * count);
    float* verts = rd(f, 12u * (size_t)n_vert);
    uint32_t* idx = rd(f, 12u * (size_t)n_tri);
    VdBvhNode* want_nodes = rd(f, sizeof(VdBvhNode) * n_nodes);
    uint32_t* want_idx = rd(f, 12u * (size_t)n_tri);
    VdTlasNode* want_tlas = rd(f, sizeof(VdTlasNode) * (2u * (size_t)n_inst + 1u));
    fclose(f);

    CHECK(vd_ctx_create(0, &ctx) == VD_OK);
    CHECK(strstr(vd_version(), "gfx950") != NULL);
    /* C1 / C2: every slot */
    VdDrawIndexedIndirect* out = malloc(sizeof(VdDrawIndexedIndirect) * n_inst);
    CHECK(vd_cull_emit(ctx, cam, meshes, n_mesh, inst, n_inst, out) == VD_OK);
    CHECK(memcmp(out, want_draws, sizeof(VdDrawIndexedIndirect) * n_inst) == 0);
    /* C3: ordered compaction, then the padded form the unchanged consumer reads */
    uint32_t got = 0;
    memset(out, 0xab, sizeof(VdDrawIndexedIndirect) * n_inst);
    CHECK(vd_cull_compact(ctx, cam, meshes, n_mesh, inst, n_inst, out, &got, 0) == VD_OK);
    CHECK(got == count && memcmp(out, want_compact, sizeof(VdDrawIndexedIndirect) * count) == 0);
    CHECK(vd_cull_compact(ctx, cam, meshes, n_mesh, inst, n_inst, out, &got, 1) == VD_OK);
    CHECK(got == count && memcmp(out, want_compact, sizeof(VdDrawIndexedIndirect) * count) == 0);
    for (uint32_t i = count; i < n_inst; ++i) CHECK(out[i].instance_count == 0u && out[i].vertex_count == 0u && out[i].base_instance == 0u);
    /* B1-B8: the BLAS of one mesh; the caller's index buffer is permuted in place */
    VdBvhNode* nodes = malloc(sizeof(VdBvhNode) * 2u * (size_t)n_tri);
    uint32_t nn = 0;
    CHECK(vd_bvh_build(ctx, verts, n_vert, idx, n_tri, nodes, 2u * n_tri, &nn) == VD_OK);
    CHECK(nn == n_nodes && memcmp(nodes, want_nodes, sizeof(VdBvhNode) * n_nodes) == 0 && memcmp(idx, want_idx, 12u * (size_t)n_tri) == 0);
    /* T1 / T2: the top level over the same instances */
    VdTlasNode* tlas = malloc(sizeof(VdTlasNode) * (2u * (size_t)n_inst + 1u));
    CHECK(vd_tlas_build(ctx, inst, n_inst, meshes, n_mesh, tlas) == VD_OK);
    CHECK(memcmp(tlas, want_tlas, sizeof(VdTlasNode) * (2u * (size_t)n_inst + 1u)) == 0);
    /* errors come back as codes with a message, never as an abort */
    CHECK(vd_cull_emit(ctx, cam, meshes, 0, inst, n_inst, out) == VD_ERR_INVALID_ARG && strlen(vd_last_error(ctx)) > 0);
    CHECK(vd_ctx_destroy(ctx) == VD_OK);
    printf("host_c_test OK (%u instances, %u survivors, %u triangles -> %u nodes, %u TLAS nodes; plain C11 host)\n", n_inst, count, n_tri, n_nodes,
           2u * n_inst + 1u);
    return 0;
}
