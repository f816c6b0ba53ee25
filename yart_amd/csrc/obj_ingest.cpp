// OBJ / MTL ingestion.  See obj_ingest.h.
//
// Compile with -ffp-contract=off.  Bit parity with the reference's loader rests on three things:
//   * numbers are converted the way its tinyobjloader converts them (tiny_obj_loader.h:266-383):
//     decimal digits accumulated in a double, scaled by pow(5, e) and ldexp(., e), then narrowed --
//     NOT strtod, whose correctly rounded result differs in the last bit for many inputs;
//   * lines are read with the same iostream calls (peek / getline into 8192 bytes), so over-long
//     lines and embedded NULs end the file where they end it for the reference;
//   * generated vertex normals add the face normals in the iteration order of a
//     std::unordered_set<int32_t> filled triangle by triangle (include/triangle.h:83-102) -- the
//     same libstdc++ container, filled the same way.
#include "obj_ingest.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <tuple>
#include <unordered_set>

namespace yart {
namespace {

constexpr std::streamsize kLineBytes = 8192;        // tiny_obj_loader.h:883, :594

inline bool blank(char c) { return c == ' ' || c == '\t'; }
inline bool digit(char c) { return c >= '0' && c <= '9'; }

// One text line and a read position in it.
struct Cursor {
    const char* p;

    void skipBlanks() { p += std::strspn(p, " \t"); }
    bool atEnd() const { return *p == '\0' || *p == '\r' || *p == '\n'; }
    bool keyword(const char* kw)        // kw followed by a blank; does not consume
    {
        const size_t n = std::strlen(kw);
        return std::strncmp(p, kw, n) == 0 && blank(p[n]);
    }

    // tiny_obj_loader.h:266-383.  Grammar: [sign] digits [. digits] [(e|E) [sign] digits]; anything
    // else (".5", "nan", "1e", "1e+") yields 0.  The value is  sign * ldexp(m * pow(5,e), e).
    float number()
    {
        skipBlanks();
        const char* s = p;
        const char* const end = p + std::strcspn(p, " \t\r");
        p = end;
        if (s >= end) return 0.0f;
        bool negative = false;
        if (*s == '+' || *s == '-') { negative = *s == '-'; ++s; }
        else if (!digit(*s)) return 0.0f;
        double m = 0.0;
        int nDigits = 0;
        for (; s != end && digit(*s); ++s, ++nDigits) { m *= 10; m += static_cast<int>(*s - '0'); }
        if (nDigits == 0) return 0.0f;
        int e = 0;
        if (s != end && *s == '.') {
            ++s;
            for (int place = 1; s != end && digit(*s); ++s, ++place)
                m += static_cast<int>(*s - '0') * std::pow(10.0, -place);
        }
        if (s != end && (*s == 'e' || *s == 'E')) {
            ++s;
            bool eNegative = false;
            if (s != end && (*s == '+' || *s == '-')) { eNegative = *s == '-'; ++s; }
            else if (!digit(*s)) return 0.0f;           // *end is a blank, '\r' or NUL: readable
            int eDigits = 0;
            for (; s != end && digit(*s); ++s, ++eDigits)
                if (e < 100000) { e *= 10; e += static_cast<int>(*s - '0'); }      // saturates: pow(5, e) is 0 or inf long before (no signed overflow on hostile input)
            if (eDigits == 0) return 0.0f;
            if (eNegative) e = -e;
        }
        const double value = (negative ? -1 : 1) * std::ldexp(m * std::pow(5.0, e), e);
        return static_cast<float>(value);
    }

    int integer()                       // tiny_obj_loader.h:232-237
    {
        skipBlanks();
        const int i = std::atoi(p);
        p += std::strcspn(p, " \t\r");
        return i;
    }

    std::string word()                  // what sscanf("%s") extracts: the next whitespace-delimited word
    {
        const char* s = p + std::strspn(p, " \t\n\v\f\r");
        const size_t n = std::strcspn(s, " \t\n\v\f\r");
        return std::string(s, n);
    }
};

// A face corner: zero-based position / texcoord / normal numbers, -1 = absent.
struct Corner {
    int v = -1, vt = -1, vn = -1;
    bool operator<(const Corner& o) const { return std::tie(v, vn, vt) < std::tie(o.v, o.vn, o.vt); }
};

// one-based -> zero-based; negative = relative to the count so far (tiny_obj_loader.h:215-221)
inline int zeroBased(int idx, int count) { return idx > 0 ? idx - 1 : (idx == 0 ? 0 : count + idx); }

// "i", "i/j", "i//k", "i/j/k" (tiny_obj_loader.h:421-452)
Corner parseCorner(Cursor& c, int nV, int nVn, int nVt)
{
    auto skipNumber = [&]() { c.p += std::strcspn(c.p, "/ \t\r"); };
    Corner k;
    k.v = zeroBased(std::atoi(c.p), nV);
    skipNumber();
    if (*c.p != '/') return k;
    ++c.p;
    if (*c.p != '/') {
        k.vt = zeroBased(std::atoi(c.p), nVt);
        skipNumber();
        if (*c.p != '/') return k;
    }
    ++c.p;
    k.vn = zeroBased(std::atoi(c.p), nVn);
    skipNumber();
    return k;
}

// Reads lines the way the reference's loader does; false at end of input.
bool nextLine(std::istream& in, std::vector<char>& buf, std::string& line)
{
    if (in.peek() == -1) return false;
    in.getline(buf.data(), kLineBytes);
    line.assign(buf.data());
    if (!line.empty() && line.back() == '\n') line.pop_back();
    if (!line.empty() && line.back() == '\r') line.pop_back();
    return true;
}

// tiny_obj_loader.h:587-813.  A missing file still contributes one unnamed default material.
void readMaterials(std::istream& in, std::vector<ObjMaterial>& materials, std::map<std::string, int>& byName)
{
    ObjMaterial cur;
    auto commit = [&]() {
        byName.insert({ cur.name, static_cast<int>(materials.size()) });   // first of a name wins
        materials.push_back(cur);
    };
    std::vector<char> buf(static_cast<size_t>(kLineBytes));
    std::string line;
    while (nextLine(in, buf, line)) {
        Cursor c{ line.c_str() };
        c.skipBlanks();
        if (*c.p == '\0' || *c.p == '#') continue;
        auto triple = [&](float* dst) { c.p += 2; dst[0] = c.number(); dst[1] = c.number(); dst[2] = c.number(); };
        if (c.keyword("newmtl")) {
            if (!cur.name.empty()) commit();
            cur = ObjMaterial{};
            c.p += 7;
            cur.name = c.word();
        } else if (c.keyword("Kd")) {
            triple(cur.diffuse);
        } else if (c.keyword("Ks")) {
            triple(cur.specular);
        } else if (c.keyword("Ni")) {
            c.p += 2;
            cur.ior = c.number();
        } else if (c.keyword("Ns")) {
            c.p += 2;
            cur.shininess = c.number();
        } else if (c.keyword("illum")) {
            c.p += 6;
            cur.illum = c.integer();
        }
        // every other statement (Ka, Kt, Ke, d, Tr, map_*, bump, disp, unknown) sets fields that
        // src/scene.cpp never reads
    }
    commit();
}

struct MeshBuilder {
    ObjMesh mesh;
    std::map<Corner, int32_t> seen;     // corner -> mesh-local vertex
};

} // namespace

int objBsdfKind(const ObjMaterial& m)
{
    switch (m.illum) {
    case 2: return 0;
    case 5: return 1;
    case 7: return 2;
    default: return 3;
    }
}

bool loadObjScene(const std::string& folder, const std::string& file, ObjScene& out, std::string& err)
{
    out = ObjScene{};
    const std::string path = folder + file;
    std::ifstream in(path.c_str());
    if (!in) {
        err = "cannot open OBJ file [" + path + "]";
        return false;
    }

    std::vector<float> v, vn;
    size_t nVt = 0;
    std::vector<std::vector<Corner>> faces;         // the faces since the last usemtl / g / o
    std::map<std::string, int> materialByName;
    int material = -1;
    std::string name;
    bool badIndex = false;

    // faces collected so far -> one mesh (tiny_obj_loader.h:516-585 + src/scene.cpp:216-277)
    auto flush = [&]() {
        if (faces.empty()) return;
        MeshBuilder b;
        auto vertexOf = [&](const Corner& k) -> int32_t {
            auto it = b.seen.find(k);
            if (it != b.seen.end()) return it->second;
            if (k.v < 0 || static_cast<size_t>(k.v) * 3 + 2 >= v.size()) { badIndex = true; return 0; }
            for (int a = 0; a < 3; ++a) b.mesh.positions.push_back(v[static_cast<size_t>(k.v) * 3 + a]);
            if (k.vn >= 0 && static_cast<size_t>(k.vn) * 3 + 2 < vn.size()) {
                for (int a = 0; a < 3; ++a) b.mesh.normals.push_back(vn[static_cast<size_t>(k.vn) * 3 + a]);
            }
            const int32_t idx = static_cast<int32_t>(b.mesh.positions.size() / 3 - 1);
            b.seen[k] = idx;
            return idx;
        };
        for (const auto& f : faces) {
            for (size_t k = 2; k < f.size(); ++k) {         // polygon -> fan around its first corner
                const int32_t i0 = vertexOf(f[0]), i1 = vertexOf(f[k - 1]), i2 = vertexOf(f[k]);
                b.mesh.indices.push_back(i0);
                b.mesh.indices.push_back(i1);
                b.mesh.indices.push_back(i2);
            }
        }
        faces.clear();
        if (b.mesh.indices.empty()) return;     // only faces of < 3 corners: the reference makes an empty
                                                // shape of them and then reads material_ids[0] of nothing
        b.mesh.name = name;
        b.mesh.material = material;
        b.mesh.normalsFromFile = b.mesh.normals.size() == b.mesh.positions.size();
        out.meshes.push_back(std::move(b.mesh));
    };

    std::vector<char> buf(static_cast<size_t>(kLineBytes));
    std::string line;
    while (nextLine(in, buf, line)) {
        Cursor c{ line.c_str() };
        c.skipBlanks();
        if (*c.p == '\0' || *c.p == '#') continue;
        if (c.p[0] == 'v' && blank(c.p[1])) {
            c.p += 2;
            for (int a = 0; a < 3; ++a) v.push_back(c.number());
        } else if (c.p[0] == 'v' && c.p[1] == 'n' && blank(c.p[2])) {
            c.p += 3;
            for (int a = 0; a < 3; ++a) vn.push_back(c.number());
        } else if (c.p[0] == 'v' && c.p[1] == 't' && blank(c.p[2])) {
            ++nVt;                                          // texture coordinates are not used by yart
        } else if (c.p[0] == 'f' && blank(c.p[1])) {
            c.p += 2;
            c.skipBlanks();
            std::vector<Corner> f;
            while (!c.atEnd()) {
                f.push_back(parseCorner(c, static_cast<int>(v.size() / 3), static_cast<int>(vn.size() / 3),
                                        static_cast<int>(nVt)));
                c.p += std::strspn(c.p, " \t\r");
            }
            faces.push_back(std::move(f));
        } else if (c.keyword("usemtl")) {
            c.p += 7;
            const std::string m = c.word();
            flush();
            auto it = materialByName.find(m);
            material = it == materialByName.end() ? -1 : it->second;
        } else if (c.keyword("mtllib")) {
            c.p += 7;
            const std::string mtlPath = folder + c.word();
            std::ifstream mtl(mtlPath.c_str());
            if (!mtl) out.warnings += "material file [" + mtlPath + "] not found, default material created\n";
            readMaterials(mtl, out.materials, materialByName);
        } else if (c.p[0] == 'g' && blank(c.p[1])) {
            flush();
            c.p += 1;                                       // the statement's own "g" is word 0 there
            Cursor rest{ c.p };
            rest.skipBlanks();
            const size_t n = std::strcspn(rest.p, " \t\r");
            name.assign(rest.p, n);
        } else if (c.p[0] == 'o' && blank(c.p[1])) {
            flush();
            c.p += 2;
            name = c.word();
        }
        // tags ("t") and unknown statements do not touch geometry
    }
    flush();
    if (badIndex) {
        err = "OBJ file [" + path + "]: a face refers to a vertex that does not exist";
        return false;
    }

    // vertex normals where the file has none for the mesh (include/triangle.h:76-111)
    for (ObjMesh& m : out.meshes) {
        if (m.normalsFromFile) continue;
        const size_t nv = m.positions.size() / 3, nt = m.indices.size() / 3;
        std::vector<std::unordered_set<int32_t>> trianglesOf(nv);
        for (size_t t = 0; t < nt; ++t)
            for (int k = 0; k < 3; ++k) trianglesOf[m.indices[t * 3 + k]].emplace(static_cast<int32_t>(t));
        auto faceNormal = [&](int32_t t, float* n) {                // include/triangle.h:158-164
            const float* a = &m.positions[static_cast<size_t>(m.indices[t * 3 + 0]) * 3];
            const float* b = &m.positions[static_cast<size_t>(m.indices[t * 3 + 1]) * 3];
            const float* c = &m.positions[static_cast<size_t>(m.indices[t * 3 + 2]) * 3];
            const float e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
            const float e2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
            const float x = e1[1] * e2[2] - e1[2] * e2[1];
            const float y = e1[2] * e2[0] - e1[0] * e2[2];
            const float z = e1[0] * e2[1] - e1[1] * e2[0];
            const float inv = 1.0f / std::sqrt(x * x + y * y + z * z);          // include/vector.h:165-168
            n[0] = x * inv; n[1] = y * inv; n[2] = z * inv;
        };
        m.normals.assign(nv * 3, 0.0f);
        for (size_t i = 0; i < nv; ++i) {
            float s[3] = { 0.0f, 0.0f, 0.0f };
            for (const int32_t t : trianglesOf[i]) {
                float n[3];
                faceNormal(t, n);
                s[0] += n[0]; s[1] += n[1]; s[2] += n[2];
            }
            const float inv = 1.0f / std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
            m.normals[i * 3 + 0] = s[0] * inv;
            m.normals[i * 3 + 1] = s[1] * inv;
            m.normals[i * 3 + 2] = s[2] * inv;
        }
    }

    // the light the reference hard-codes for every OBJ scene (src/scene.cpp:280-283)
    out.spheres = { 0.0f, 1.0f, -0.5f, 0.05f };
    return true;
}

} // namespace yart
