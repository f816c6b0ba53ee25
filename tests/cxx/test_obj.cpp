// Loads an OBJ file through the C++ shim's Scene::loadFromObj (the reference's signature,
// src/scene.cpp:159) and dumps what it holds:  test_obj <folder> <file> <out.bin>
// out.bin: u64 counts (meshes, vertices, triangles, spheres, materials), then per mesh its
// vertices, normals, mesh-local triangles and material number, then spheres, then materials.
#include <cstdio>
#include "../../yart_amd/cxx/yart_shim.hpp"

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    std::unique_ptr<yart_amd::Scene> scene;
    try {
        scene = yart_amd::Scene::loadFromObj(argv[1], argv[2]);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 4;
    }
    FILE* f = fopen(argv[3], "wb");
    if (!f) return 2;
    uint64_t nv = 0, nt = 0;
    for (const auto& m : scene->getTriangleMeshes()) { nv += m.vertices.size(); nt += m.triangles.size(); }
    const uint64_t counts[5] = { scene->getTriangleMeshes().size(), nv, nt, scene->getShapes().size(), scene->getMaterials().size() };
    fwrite(counts, 8, 5, f);
    for (const auto& m : scene->getTriangleMeshes()) {
        const uint64_t c[2] = { m.vertices.size(), m.triangles.size() };
        fwrite(c, 8, 2, f);
        fwrite(m.vertices.data(), 12, m.vertices.size(), f);
        fwrite(m.normals.data(), 12, m.normals.size(), f);
        fwrite(m.triangles.data(), 12, m.triangles.size(), f);
        fwrite(&m.material, 4, 1, f);
    }
    for (const auto& s : scene->getShapes()) { fwrite(&s.position, 12, 1, f); fwrite(&s.radius, 4, 1, f); }
    fwrite(scene->getMaterials().data(), sizeof(YartObjMaterial), scene->getMaterials().size(), f);
    fclose(f);
    return 0;
}
