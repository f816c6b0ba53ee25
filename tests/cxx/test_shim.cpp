// Drives yart_amd/cxx/yart_shim.hpp the way yart's renderer drives Scene: one ray per call
// through Scene::intersect / Scene::intersectShadow, then the batched forms.  Reads a scene +
// rays from a flat binary file written by tests/test_cxx_shim.py, writes the RayHitInfo fields
// back for comparison with the golden vectors of the real reference.
//   test_shim <in.bin> <out.bin> [reference_rule]     exit 0 ok, 3 = engine reported "no device"
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../yart_amd/cxx/yart_shim.hpp"
#include "../../yart_amd/cxx/wavefront.hpp"

using namespace yart_amd;

template <class T> static std::vector<T> readArr(FILE* f)
{
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) { fprintf(stderr, "short read\n"); exit(2); }
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

// One closest-hit ray and one shadow ray per slot through the batched driver: must give what the
// direct batch calls give.
struct OneShot : yart_amd::PathIntegrator {
    const YartRay* rays; const YartRay* shadow; size_t nShadow;
    std::vector<uint8_t> started; std::vector<int32_t> prim; std::vector<uint8_t> occ;
    bool nextPath(size_t s, YartRay* r) override { if (started[s]) return false; started[s] = 1; *r = rays[s]; return true; }
    bool onClosest(size_t s, const YartRay&, const YartHit& h, YartRay* sh) override
    {
        prim[s] = h.prim;
        if (s >= nShadow) return false;
        *sh = shadow[s];
        return true;
    }
    bool onShadow(size_t s, bool o, YartRay*) override { occ[s] = o ? 1 : 0; return false; }
};

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    auto verts = readArr<float>(f);       // nV*3
    auto normals = readArr<float>(f);     // nV*3
    auto tris = readArr<int32_t>(f);      // nT*5
    auto sph = readArr<float>(f);         // nS*4
    auto rays = readArr<float>(f);        // n*8
    auto shadow = readArr<float>(f);      // m*8
    fclose(f);

    // rebuild meshes (contiguous vertex blocks per mesh, as the fixtures are laid out)
    std::vector<TriangleMesh> meshes;
    static int bsdfTags[1 << 20];
    const size_t nT = tris.size() / 5;
    for (size_t i = 0; i < nT;) {
        const int32_t mid = tris[i * 5 + 3];
        size_t j = i;
        int32_t vmin = 1 << 30, vmax = -1;
        for (; j < nT && tris[j * 5 + 3] == mid; ++j)
            for (int c = 0; c < 3; ++c) { vmin = std::min(vmin, tris[j * 5 + c]); vmax = std::max(vmax, tris[j * 5 + c]); }
        TriangleMesh m;
        for (int32_t k = vmin; k <= vmax; ++k) {
            m.vertices.emplace_back(verts[k * 3], verts[k * 3 + 1], verts[k * 3 + 2]);
            m.normals.emplace_back(normals[k * 3], normals[k * 3 + 1], normals[k * 3 + 2]);
        }
        for (size_t k = i; k < j; ++k) m.triangles.push_back(Triangle{ tris[k * 5] - vmin, tris[k * 5 + 1] - vmin, tris[k * 5 + 2] - vmin });
        m.bsdf = reinterpret_cast<Bsdf*>(&bsdfTags[mid % (1 << 20)]);
        meshes.push_back(std::move(m));
        i = j;
    }
    std::vector<Sphere> shapes;
    for (size_t s = 0; s < sph.size() / 4; ++s) {
        Sphere sp;
        sp.radius = sph[s * 4 + 3];
        sp.position = Vector3f(sph[s * 4], sph[s * 4 + 1], sph[s * 4 + 2]);
        shapes.push_back(sp);
    }
    Scene scene(std::move(meshes), std::move(shapes));
    try {
        scene.preprocess(0);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return strstr(e.what(), "no HIP device") ? 3 : 4;
    }
    if (argc > 3) scene.setFlags(YART_REFERENCE_RULE);

    const size_t n = rays.size() / 8, m = shadow.size() / 8;
    std::vector<float> out(n * 10);       // hit, t, u, v, normal xyz, shadingNormal xyz
    std::vector<int32_t> prim(n);
    const Ray* R = reinterpret_cast<const Ray*>(rays.data());
    const size_t single = n < 512 ? n : 512;                 // one ray per call, like the reference's caller
    auto store = [&](size_t i, bool hit, const RayHitInfo& h) {
        float* o = &out[i * 10];
        o[0] = hit ? 1.0f : 0.0f; o[1] = h.t; o[2] = h.u; o[3] = h.v;
        o[4] = h.normal.x; o[5] = h.normal.y; o[6] = h.normal.z;
        o[7] = h.shadingNormal.x; o[8] = h.shadingNormal.y; o[9] = h.shadingNormal.z;
        prim[i] = h.prim;
    };
    for (size_t i = 0; i < single; ++i) {
        RayHitInfo h;
        std::memset(&h, 0, sizeof h);
        const bool hit = scene.intersect(R[i], &h);
        store(i, hit, h);
    }
    if (n > single) {
        std::vector<RayHitInfo> hs(n - single);
        std::memset(hs.data(), 0, hs.size() * sizeof(RayHitInfo));
        std::vector<uint8_t> flags(n - single);
        scene.intersectBatch(R + single, n - single, hs.data(), flags.data());
        for (size_t i = single; i < n; ++i) store(i, flags[i - single] != 0, hs[i - single]);
    }
    std::vector<uint8_t> occ(m);
    const Ray* S = reinterpret_cast<const Ray*>(shadow.data());
    const size_t singleS = m < 256 ? m : 256;
    for (size_t i = 0; i < singleS; ++i) occ[i] = scene.intersectShadow(S[i]) ? 1 : 0;
    if (m > singleS) scene.intersectShadowBatch(S + singleS, m - singleS, occ.data() + singleS);

    {
        OneShot integ;
        integ.rays = reinterpret_cast<const YartRay*>(rays.data());
        integ.shadow = reinterpret_cast<const YartRay*>(shadow.data());
        integ.nShadow = m;
        integ.started.assign(n, 0); integ.prim.assign(n, -7); integ.occ.assign(n, 9);
        yart_amd::WavefrontDriver driver(yart_amd::linkedEngine(scene.accel()->handle(), argc > 3 ? YART_REFERENCE_RULE : YART_DEFAULT), n, 4);
        const yart_amd::WavefrontStats st = driver.run(integ);
        if (st.closestRays != n || st.shadowRays != (m < n ? m : n)) { fprintf(stderr, "driver ray counts\n"); return 5; }
        for (size_t i = 0; i < n; ++i) if (integ.prim[i] != prim[i]) { fprintf(stderr, "driver prim differs at %zu\n", i); return 5; }
        for (size_t i = 0; i < m && i < n; ++i) if (integ.occ[i] != occ[i]) { fprintf(stderr, "driver occlusion differs at %zu\n", i); return 5; }
    }

    FILE* g = fopen(argv[2], "wb");
    if (!g) return 2;
    fwrite(out.data(), 4, out.size(), g);
    fwrite(prim.data(), 4, prim.size(), g);
    fwrite(occ.data(), 1, occ.size(), g);
    fclose(g);
    return 0;
}
