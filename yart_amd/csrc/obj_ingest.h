// OBJ / MTL ingestion: Wavefront OBJ -> the flat scene yart_accel_create takes,
// with the result the reference gets from Scene::loadFromObj (src/scene.cpp:159-298) through its
// bundled tinyobjloader (external/tinyobjloader/tiny_obj_loader.h:838-1133): one mesh per
// `usemtl` / `g` / `o` run of faces, vertices renumbered per mesh in order of first use, polygons
// fanned, the mesh's material = the one in force for its faces, vertex normals taken from the file
// when every vertex of the mesh carries one and generated otherwise (include/triangle.h:76-111),
// plus the light sphere the reference always appends (src/scene.cpp:280-283).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace yart {

struct ObjMaterial {            // the fields of tinyobj::material_t that src/scene.cpp:172-208 reads
    std::string name;
    float diffuse[3] = { 0, 0, 0 };
    float specular[3] = { 0, 0, 0 };
    float shininess = 1.0f;
    float ior = 1.0f;
    int32_t illum = 0;
};

struct ObjMesh {
    std::string name;
    std::vector<float> positions;   // xyz per vertex, mesh-local numbering
    std::vector<float> normals;     // xyz per vertex (from the file, or generated)
    std::vector<int32_t> indices;   // 3 per triangle, mesh-local
    int32_t material = -1;          // index into materials, -1: none in force (undefined in the reference)
    bool normalsFromFile = false;
};

struct ObjScene {
    std::vector<ObjMesh> meshes;
    std::vector<ObjMaterial> materials;
    std::vector<float> spheres;     // cx, cy, cz, r
    std::string warnings;
};

// false + err when the OBJ file cannot be opened or holds an index the reference would read out of
// bounds with.  `folder` is prepended to `file` and to every mtllib name, as the reference does.
bool loadObjScene(const std::string& folder, const std::string& file, ObjScene& out, std::string& err);

// Which Bsdf src/scene.cpp:183-207 makes of a material: 0 Lambertian(diffuse), 1 Torrance-Sparrow
// conductor (shininess), 2 Fresnel dielectric (ior), 3 black Lambertian (every other illum).
int objBsdfKind(const ObjMaterial& m);

} // namespace yart
