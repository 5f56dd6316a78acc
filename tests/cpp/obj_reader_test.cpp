// CPU-only check of voidin::ObjModel::load (include/voidin.hpp) on tests/golden/two_objects.obj.
#include <cstdio>
#include <cstdlib>
#include <string>
#include "voidin.hpp"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

// `dump <obj> <out>`: positions (f32 xyz) + indices (u32) of every model, length-prefixed (host only);
// `pool <obj> <out>`: ObjModel::import into a MeshPool (BLAS built on the GPU through the C ABI), then
//                     MeshInfo records, concatenated BLAS nodes and permuted indices.
static void put(std::FILE* f, const void* p, size_t bytes) { const uint64_t n = bytes; std::fwrite(&n, 8, 1, f); std::fwrite(p, 1, bytes, f); }

static int dump_mode(const char* mode, const char* path, const char* out_path) {
    std::FILE* f = std::fopen(out_path, "wb");
    if (!f) return 3;
    if (std::string(mode) == "dump") {
        const auto meshes = voidin::ObjModel::load(path);
        const uint64_t n = meshes.size();
        std::fwrite(&n, 8, 1, f);
        for (const auto& m : meshes) {
            put(f, m.positions.data(), m.positions.size() * sizeof(voidin::Vec3));
            put(f, m.indices.data(), m.indices.size() * 4);
        }
    } else {
        voidin::Gpu gpu(0);
        voidin::MeshPool pool(gpu);
        const auto ids = voidin::ObjModel::import(pool, path);
        const uint64_t n = ids.size();
        std::fwrite(&n, 8, 1, f);
        put(f, pool.mesh_info_cpu.data(), pool.mesh_info_cpu.size() * sizeof(voidin::MeshInfo));
        put(f, pool.bvh_nodes.data(), pool.bvh_nodes.size() * sizeof(voidin::BvhNode));
        put(f, pool.indices.data(), pool.indices.size() * 4);
    }
    std::fclose(f);
    std::printf("obj_reader_test %s OK\n", mode);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    if (argc == 4) return dump_mode(argv[1], argv[2], argv[3]);
    const auto meshes = voidin::ObjModel::load(argv[1]);
    CHECK(meshes.size() == 3);
    // "plate": quad fan-triangulated + one triangle reusing the same v/vt/vn triples; the `l` line is ignored
    const auto& a = meshes[0];
    CHECK(a.name == "plate" && a.material_id == 0);
    CHECK(a.positions.size() == 4 && a.normals.size() == 4 && a.texcoords.size() == 8);
    const uint32_t want_a[9] = {0, 1, 2, 0, 2, 3, 0, 2, 3};
    CHECK(a.indices.size() == 9);
    for (int i = 0; i < 9; ++i) CHECK(a.indices[i] == want_a[i]);
    CHECK(a.positions[2].x == 1.0f && a.positions[2].y == 1.0f && a.texcoords[5] == 1.0f && a.normals[3].z == 1.0f);
    // "spike", material red: relative index -1 is the vertex just declared; positions re-indexed per model
    const auto& b = meshes[1];
    CHECK(b.name == "spike" && b.material_id == 0);
    CHECK(b.positions.size() == 3 && b.normals.empty() && b.texcoords.empty());
    CHECK(b.indices.size() == 3 && b.indices[0] == 0 && b.indices[1] == 1 && b.indices[2] == 2);
    CHECK(b.positions[0].z == 2.25f && b.positions[1].x == 0.0f && b.positions[2].x == 1.0f);
    // material switch after faces: new model, same name; v//vn triples are distinct from the bare v of the previous face
    const auto& c = meshes[2];
    CHECK(c.name == "spike" && c.material_id == 1);
    CHECK(c.positions.size() == 3 && c.normals.size() == 3 && c.texcoords.empty());
    CHECK(c.positions[2].z == 2.25f);
    std::printf("obj_reader_test OK\n");
    return 0;
}
