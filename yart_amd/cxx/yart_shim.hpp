// yart_shim.hpp -- host-side mirror of the reference's Scene / BvhAccel interface on top of the
// C-ABI (include/yart_amd.h).  Header-only C++17; link against libyart_amd.so.
//
// Same names, argument meaning and return conventions as the reference for this path:
//   Scene::preprocess()                          include/scene.h:37-74
//   bool Scene::intersect(const Ray&, RayHitInfo*) const      include/scene.h:136-168
//   bool Scene::intersectShadow(const Ray&) const             include/scene.h:170-194
//   BvhAccel(const Scene&), intersect, intersectShadow        include/bvhaccel.h:30-50
// plus the batched forms a wavefront caller uses (the reference has none: its renderer
// issues one ray per call, src/renderer.cpp:59,104).
//
// Hit post-processing happens here on the host, from (prim, u, v), with the
// reference's formulas: geometric normal = normal(cross(e1, e2)) (include/triangle.h:158-164),
// shading normal = n0 (1-u-v) + n1 u + n2 v (include/triangle.h:166-171, not normalised),
// sphere normal = normal(orig + dir t - position) (include/sphere.h:80-81), areaLight = nullptr
// for triangles (src/bvhaccel.cpp:291).  Compile with -ffp-contract=off to get the reference's bits.
//
// Failures of the engine (no device, HIP error) throw std::runtime_error: the reference has no
// error convention at all (bool = hit/miss), so hit/miss stays the return value.
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/yart_amd.h"

namespace yart_amd {

struct Vector3f {                       // include/vector.h:36-180 (the members this path needs)
    float x, y, z;
    Vector3f() = default;
    Vector3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vector3f operator+(const Vector3f& r) const { return Vector3f(x + r.x, y + r.y, z + r.z); }
    Vector3f operator-(const Vector3f& r) const { return Vector3f(x - r.x, y - r.y, z - r.z); }
    Vector3f operator*(float r) const { return Vector3f(x * r, y * r, z * r); }
    float length() const { return std::sqrt(x * x + y * y + z * z); }
    Vector3f normal() const { return *this * (1.0f / length()); }           // vector.h:165-168
    Vector3f cross(const Vector3f& r) const { return Vector3f(y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x); }
};

struct Ray {                            // include/vector.h:238-250, same 32-byte layout as YartRay
    Vector3f orig;
    float minT;
    Vector3f dir;
    float maxT;
    Ray(Vector3f o, Vector3f d) : orig(o), minT(0.0f), dir(d), maxT(std::numeric_limits<float>::infinity()) {}
};
static_assert(sizeof(Ray) == sizeof(YartRay), "Ray must match the 32-byte C-ABI record");

class Bsdf;                             // opaque to this path: borrowed pointers, as in the reference
class AreaLight;

struct RayHitInfo {                     // include/vector.h:255-263 (+ prim, which the reference hides)
    Vector3f normal;
    float t;
    Vector3f shadingNormal;
    float u;
    float v;
    Bsdf* bsdf;
    AreaLight* areaLight;
    int32_t prim;
};

struct Triangle { int32_t idx0, idx1, idx2; };          // include/triangle.h:19-33

struct TriangleMesh {                   // the data of include/triangle.h:71-200 that the hot path reads
    std::vector<Vector3f> vertices;
    std::vector<Vector3f> normals;      // per vertex (the reference generates them at :76-111 when absent)
    std::vector<Triangle> triangles;
    Bsdf* bsdf = nullptr;
    int32_t material = -1;              // loadFromObj: index into Scene::getMaterials()

    Vector3f getNormal(int32_t tri) const               // include/triangle.h:158-164
    {
        const Triangle& t = triangles[tri];
        const Vector3f e1 = vertices[t.idx1] - vertices[t.idx0];
        const Vector3f e2 = vertices[t.idx2] - vertices[t.idx0];
        return e1.cross(e2).normal();
    }
    Vector3f getShadingNormal(int32_t tri, float u, float v) const      // include/triangle.h:166-171
    {
        const Triangle& t = triangles[tri];
        return normals[t.idx0] * (1.0f - u - v) + normals[t.idx1] * u + normals[t.idx2] * v;
    }
};

struct Sphere {                         // include/sphere.h:21-117 (geometry + borrowed pointers)
    float radius;
    Vector3f position;
    Bsdf* bsdf = nullptr;
    AreaLight* light = nullptr;         // Shape::getLight(), include/shape.h:24-27
};

class Scene;

// include/bvhaccel.h:30-60.  Owns the device-resident accel (triangle tree + sphere tree).
class BvhAccel {
public:
    explicit BvhAccel(const Scene& scene, int device = 0);
    ~BvhAccel() { yart_accel_destroy(accel_); }
    BvhAccel(const BvhAccel&) = delete;
    BvhAccel& operator=(const BvhAccel&) = delete;

    YartAccel* handle() const { return accel_; }
    const std::vector<YartTriAccel>& triaccel() const { return triaccel_; }
    bool hasSurfaceData() const { return surface_; }    // normals are derived on the device for batches

private:
    YartAccel* accel_ = nullptr;
    std::vector<YartTriAccel> triaccel_;    // leaf order: prim -> (meshIdx, triIdx)
    bool surface_ = false;
};

class Scene {
public:
    Scene(std::vector<TriangleMesh> meshes, std::vector<Sphere> shapes)
        : meshes_(std::move(meshes)), shapes_(std::move(shapes)) {}
    ~Scene() { delete accel_; }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    // src/scene.cpp:159-298: OBJ + MTL -> meshes (vertices, normals from the file or generated,
    // triangles, material number) + the light sphere the reference appends.  The caller makes its
    // Bsdf objects from getMaterials() (YartObjMaterial::bsdf names the class src/scene.cpp:183-207
    // picks) and hangs them on the meshes; a Scene is not copyable here, hence the pointer.
    static std::unique_ptr<Scene> loadFromObj(const std::string& folder, const std::string& file)
    {
        YartObjScene* h = nullptr;
        check(yart_obj_load(folder.c_str(), file.c_str(), &h));
        std::vector<float> v(yart_obj_num_vertices(h) * 3), n(v.size()), sph(yart_obj_num_spheres(h) * 4);
        std::vector<int32_t> t(yart_obj_num_triangles(h) * 5);
        std::vector<YartObjMesh> ms(yart_obj_num_meshes(h));
        std::vector<YartObjMaterial> mats(yart_obj_num_materials(h));
        const int st = yart_obj_get(h, v.data(), n.data(), t.data(), sph.data(), ms.data(), mats.data());
        yart_obj_free(h);
        check(st);
        std::vector<TriangleMesh> meshes(ms.size());
        for (size_t m = 0; m < ms.size(); ++m) {
            TriangleMesh& mesh = meshes[m];
            for (int32_t i = ms[m].first_vertex; i < ms[m].first_vertex + ms[m].num_vertices; ++i) {
                mesh.vertices.emplace_back(v[i * 3], v[i * 3 + 1], v[i * 3 + 2]);
                mesh.normals.emplace_back(n[i * 3], n[i * 3 + 1], n[i * 3 + 2]);
            }
            for (int32_t i = ms[m].first_triangle; i < ms[m].first_triangle + ms[m].num_triangles; ++i)
                mesh.triangles.push_back(Triangle{ t[i * 5] - ms[m].first_vertex, t[i * 5 + 1] - ms[m].first_vertex,
                                                   t[i * 5 + 2] - ms[m].first_vertex });
            mesh.material = ms[m].material;
        }
        std::vector<Sphere> shapes;
        for (size_t i = 0; i + 3 < sph.size(); i += 4) {
            Sphere sp;
            sp.radius = sph[i + 3];
            sp.position = Vector3f(sph[i], sph[i + 1], sph[i + 2]);
            shapes.push_back(sp);
        }
        std::unique_ptr<Scene> scene(new Scene(std::move(meshes), std::move(shapes)));
        scene->materials_ = std::move(mats);
        return scene;
    }
    const std::vector<YartObjMaterial>& getMaterials() const { return materials_; }

    // include/scene.h:37-74: build the acceleration structure (here: build + upload to `device`)
    void preprocess(int device = 0)
    {
        delete accel_;
        accel_ = nullptr;
        accel_ = new BvhAccel(*this, device);
    }

    // include/scene.h:136-168
    bool intersect(const Ray& ray, RayHitInfo* const isect) const
    {
        YartHit h;
        check(yart_intersect_closest(handle(), reinterpret_cast<const YartRay*>(&ray), 1, &h, flags_));
        fill(ray, h, isect);
        return h.prim != -1;
    }

    // include/scene.h:170-194
    bool intersectShadow(const Ray& ray) const
    {
        uint8_t occ = 0;
        check(yart_intersect_any(handle(), reinterpret_cast<const YartRay*>(&ray), 1, &occ, flags_));
        return occ != 0;
    }

    // Batched forms (host buffers).  hits[i] is filled exactly as intersect() would.
    void intersectBatch(const Ray* rays, size_t n, RayHitInfo* hits, uint8_t* hitFlags = nullptr) const
    {
        std::vector<YartHit> h(n);
        check(yart_intersect_closest(handle(), reinterpret_cast<const YartRay*>(rays), n, h.data(), flags_));
        if (accel_->hasSurfaceData() && n >= 256) {     // row K as a device gather (yart_surface)
            std::vector<YartSurface> s(n);
            check(yart_surface(handle(), reinterpret_cast<const YartRay*>(rays), h.data(), n, s.data()));
            for (size_t i = 0; i < n; ++i) {
                RayHitInfo& o = hits[i];
                o.normal = Vector3f(s[i].normal[0], s[i].normal[1], s[i].normal[2]);
                o.shadingNormal = Vector3f(s[i].shading_normal[0], s[i].shading_normal[1], s[i].shading_normal[2]);
                o.t = s[i].t; o.u = s[i].u; o.v = s[i].v; o.prim = s[i].prim;
                o.bsdf = nullptr;
                o.areaLight = nullptr;
                if (s[i].prim >= 0) o.bsdf = meshes_[s[i].mesh_idx].bsdf;
                else if (s[i].prim <= -2) { o.bsdf = shapes_[s[i].tri_idx].bsdf; o.areaLight = shapes_[s[i].tri_idx].light; }
                if (hitFlags) hitFlags[i] = s[i].prim != -1;
            }
            return;
        }
        for (size_t i = 0; i < n; ++i) {
            fill(rays[i], h[i], &hits[i]);
            if (hitFlags) hitFlags[i] = h[i].prim != -1;
        }
    }
    void intersectShadowBatch(const Ray* rays, size_t n, uint8_t* occluded) const
    {
        check(yart_intersect_any(handle(), reinterpret_cast<const YartRay*>(rays), n, occluded, flags_));
    }

    // YART_REFERENCE_RULE etc. (include/yart_amd.h)
    void setFlags(unsigned flags) { flags_ = flags; }

    const std::vector<TriangleMesh>& getTriangleMeshes() const { return meshes_; }     // include/scene.h:237-240
    const std::vector<Sphere>& getShapes() const { return shapes_; }
    const BvhAccel* accel() const { return accel_; }

private:
    YartAccel* handle() const
    {
        if (!accel_) throw std::runtime_error("yart_amd::Scene: preprocess() was not called");
        return accel_->handle();
    }
    static void check(int st)
    {
        if (st != YART_OK)
            throw std::runtime_error(std::string("yart_amd: ") + yart_status_string(st) + ": " + yart_last_error());
    }
    // what the reference derives after a hit (src/bvhaccel.cpp:283-292, sphere.h:79-83)
    void fill(const Ray& ray, const YartHit& h, RayHitInfo* isect) const
    {
        isect->t = h.t;
        isect->u = h.u;
        isect->v = h.v;
        isect->prim = h.prim;
        isect->areaLight = nullptr;
        isect->bsdf = nullptr;
        if (h.prim >= 0) {
            const YartTriAccel& ta = accel_->triaccel()[h.prim];
            const TriangleMesh& m = meshes_[ta.meshIdx];
            isect->normal = m.getNormal(ta.triIdx);
            isect->shadingNormal = m.getShadingNormal(ta.triIdx, h.u, h.v);
            isect->bsdf = m.bsdf;
        } else if (h.prim <= -2) {
            const Sphere& s = shapes_[-(h.prim + 2)];
            isect->normal = (ray.orig + ray.dir * h.t - s.position).normal();
            isect->shadingNormal = isect->normal;
            isect->bsdf = s.bsdf;
            isect->areaLight = s.light;
        }
    }

    std::vector<TriangleMesh> meshes_;
    std::vector<Sphere> shapes_;
    std::vector<YartObjMaterial> materials_;
    BvhAccel* accel_ = nullptr;
    unsigned flags_ = YART_DEFAULT;
};

inline BvhAccel::BvhAccel(const Scene& scene, int device)
{
    // Flatten to the C-ABI's exchange format in the order the reference enumerates triangles
    // (src/bvhaccel.cpp:328-343): mesh by mesh, triangle by triangle.
    std::vector<float> verts;
    std::vector<int32_t> tris;
    int32_t base = 0, mid = 0;
    for (const TriangleMesh& m : scene.getTriangleMeshes()) {
        for (const Vector3f& v : m.vertices) { verts.push_back(v.x); verts.push_back(v.y); verts.push_back(v.z); }
        int32_t tid = 0;
        for (const Triangle& t : m.triangles) {
            const int32_t rec[5] = { t.idx0 + base, t.idx1 + base, t.idx2 + base, mid, tid++ };
            tris.insert(tris.end(), rec, rec + 5);
        }
        base += static_cast<int32_t>(m.vertices.size());
        ++mid;
    }
    std::vector<float> sph;
    for (const Sphere& s : scene.getShapes()) {
        sph.push_back(s.position.x); sph.push_back(s.position.y); sph.push_back(s.position.z); sph.push_back(s.radius);
    }
    const int st = yart_accel_create(verts.data(), verts.size() / 3, tris.data(), tris.size() / 5,
                                     sph.data(), sph.size() / 4, device, &accel_);
    if (st != YART_OK)
        throw std::runtime_error(std::string("yart_amd::BvhAccel: ") + yart_status_string(st) + ": " + yart_last_error());
    triaccel_.resize(yart_accel_num_triangles(accel_));
    if (!triaccel_.empty()) yart_accel_get_triaccel(accel_, triaccel_.data());
    // vertex normals of every mesh -> the device, so that batches get their normals there
    std::vector<float> nrm;
    bool complete = true;
    for (const TriangleMesh& m : scene.getTriangleMeshes()) {
        complete = complete && m.normals.size() == m.vertices.size();
        for (const Vector3f& n : m.normals) { nrm.push_back(n.x); nrm.push_back(n.y); nrm.push_back(n.z); }
    }
    if (complete)
        surface_ = yart_accel_set_surface_data(accel_, verts.data(), verts.size() / 3, tris.data(), tris.size() / 5, nrm.data(),
                                               sph.data(), sph.size() / 4) == YART_OK;
}

} // namespace yart_amd
