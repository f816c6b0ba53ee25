// cover.h -- what makes the default (4-wide tree) rule agree with the reference on ILL-CONDITIONED primitives too.
//
// The reference's primitive tests are not exact.  Wald's projected triangle test (include/triaccel.h:169-201) accepts
// points a rounding band OUTSIDE the triangle's edges -- a band that for a sliver is wider than the sliver -- and the
// sphere test (include/sphere.h:63-97) forms b*b - |op|^2 + r^2 from fp32 b and |op|^2, so a ray that passes within
// sqrt(r^2 + ~2^-20 |op|^2) of the centre can be accepted.  Which of those false positives a traversal returns
// depends on the boxes it culls with: the reference meets a triangle iff the ray passes the box of the triangle's LEAF
// in its tree (include/bbox.h:30-50 against [minT, maxT]; a leaf's box lies inside every ancestor's and the slab test
// is monotone, so the leaf box alone decides), and it meets every sphere (linear loop, include/scene.h:142-144).  The
// engine's tree culls with its own boxes.  Two measures make the outcomes equal:
//
//   * every primitive's box in the engine's trees CONTAINS every point its test can accept: a triangle's box is grown
//     by its error band (triBand, a bound derived from the record's own coefficients), a sphere's box is that of radius
//     r_eff = sqrt(r^2 + 2^-20 D^2), D = the longest |op| a covered ray can have;
//   * a triangle accepted within its band of an edge counts only if the ray also passes the reference's leaf box
//     (checked in the kernel, traverse_wide.hip); away from the edges the hit point lies inside the triangle, hence
//     inside that box.
//
// Both need a bound on the rays: their origins (|coordinate| <= M) and how far they can travel to a primitive (D).
// `Cover` states it: the box of origins the bounds hold for.  A ray that starts outside it is not taken by the wide
// kernel; it goes to the fix-up launches on the reference's own tree (traverse.hip), whose sphere boxes get a per-ray
// pad instead.  yart's rays start inside the scene (camera, surface points), so this never happens there.
//
// Host and device use the same expressions (double arithmetic, + - * / only, -ffp-contract=off).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define YART_HD __host__ __device__
#else
#define YART_HD
#endif

namespace yart {

struct Cover {
    float lo[3], hi[3];     // the ray origins the primitives' boxes are padded for
    float M;                // >= |coordinate| of any covered origin
    float padMax;           // how far a triangle's error-band box (triBand) reaches beyond the triangle's own box, at most:
                            // set by the builders; the pruned walk of the reference's tree culls that much later (traverse.hip)
    float reserved;
    uint32_t valid;         // 0: nothing to bound (empty scene): every origin is covered
};

// Square of the longest distance from a covered origin to a point of the box [plo, phi].
YART_HD inline double coverFar2(const Cover& c, const float plo[3], const float phi[3])
{
    double d2 = 0.0;
    for (int a = 0; a < 3; ++a) {
        const double x = (double)phi[a] - (double)c.lo[a], y = (double)c.hi[a] - (double)plo[a];
        const double f = x > y ? x : y;
        d2 += f * f;
    }
    return d2;
}

constexpr double kHalfUlp = 5.9604644775390625e-08;      // 2^-24, unit roundoff of fp32
constexpr double kSphereNoise = 9.5367431640625e-07;     // 2^-20 = 16 * 2^-24: see sphereReff2

YART_HD inline float coverNextUp(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return f;                     // inf / NaN
    if (u == 0x80000000u || u == 0u) u = 1u;                            // +-0 -> smallest positive
    else if (u & 0x80000000u) --u;
    else ++u;
    memcpy(&f, &u, 4);
    return f;
}
YART_HD inline float coverNextDown(float f) { return -coverNextUp(-f); }
YART_HD inline float coverFloatUp(double x) { float f = (float)x; return ((double)f < x) ? coverNextUp(f) : f; }
YART_HD inline float coverFloatDown(double x) { float f = (float)x; return ((double)f > x) ? coverNextDown(f) : f; }
YART_HD inline double coverAbs(double x) { return x < 0.0 ? -x : x; }
YART_HD inline bool coverFinite(double x) { return x - x == 0.0; }

YART_HD inline bool coverHolds(const Cover& c, float ox, float oy, float oz)
{
    return !c.valid || (ox >= c.lo[0] && ox <= c.hi[0] && oy >= c.lo[1] && oy <= c.hi[1] && oz >= c.lo[2] && oz <= c.hi[2]);
}

// How far the band box [lo, hi] of triBand reaches beyond the bounding box of the triangle a, b, c.
YART_HD inline float bandExcess(const float lo[3], const float hi[3], const float* a, const float* b, const float* c)
{
    float e = 0.0f;
    for (int i = 0; i < 3; ++i) {
        const float l = a[i] < b[i] ? (a[i] < c[i] ? a[i] : c[i]) : (b[i] < c[i] ? b[i] : c[i]);
        const float h = a[i] > b[i] ? (a[i] > c[i] ? a[i] : c[i]) : (b[i] > c[i] ? b[i] : c[i]);
        if (!(lo[i] <= hi[i])) return 0.0f;          // the empty box of a triangle that is never accepted
        const float x = l - lo[i], y = hi[i] - h;
        e = x > e ? x : e;
        e = y > e ? y : e;
    }
    return e;
}

// The cover of a scene whose primitives lie (outliers aside, wide_build.cpp: sceneCover) in [slo, shi] and whose smallest
// sphere radius is rMin (nSpheres == 0: none).  The margin around that box is as wide as it can be while the sphere boxes
// stay within 5 % of the spheres' radii (2^-20 D^2 <= 0.1 rMin^2), between 1/16 of a diagonal and one diagonal; for
// triangles the bounds enter the band linearly (a margin of eight diagonals cost the architectural scenes 4-7 %).
inline Cover makeCover(const float slo[3], const float shi[3], bool haveBounds, size_t nSpheres, float rMin);

// include/sphere.h:66-70.  With op = fl(c - o) and unit roundoff h = 2^-24: |fl(b) - b| <= 3h |op| so b*b is off by
// <= 6h |op|^2, fl(|op|^2) by <= 3h |op|^2, a direction normalised in fp32 (v / sqrt(fl(v.v)), include/vector.h) has
// | |d|^2 - 1 | <= 7h: the computed determinant exceeds r^2 - (distance of the centre from the ray)^2 by at most
// 16h |op|^2 = 2^-20 |op|^2.
// |op| <= the distance from the centre to the far corner of the cover.
YART_HD inline double sphereReff2(const float s[4], const Cover& c)
{
    const double r = (double)s[3];
    return r * r + (c.valid ? kSphereNoise * coverFar2(c, s, s) : 0.0);
}

// Box of a sphere in the engine's trees: centre +- r_eff, outward-rounded with a relative hair for the fp32 slab test.
// sqrtReff = sqrt(sphereReff2) supplied by the caller (double sqrt, correctly rounded on host and device).
// A sphere with an inf / NaN centre or radius is never accepted (its determinant is NaN, or both roots are infinite).
YART_HD inline bool sphereFinite(const float s[4])
{
    return coverFinite((double)s[0]) && coverFinite((double)s[1]) && coverFinite((double)s[2]) && coverFinite((double)s[3]);
}

YART_HD inline void sphereBox(const float s[4], double sqrtReff, float lo[3], float hi[3])
{
    for (int a = 0; a < 3; ++a) {
        const double m = 4.0 * kHalfUlp * (coverAbs((double)s[a]) + sqrtReff) + 1e-30;
        lo[a] = coverFloatDown((double)s[a] - sqrtReff - m);
        hi[a] = coverFloatUp((double)s[a] + sqrtReff + m);
    }
}

// include/triaccel.h:54-108 in fp32, operation for operation (bvh_build.cpp projectTriangle is the host builder's copy):
// out = n_u, n_v, n_d, k | b_u, b_v, b_d, triIdx | c_u, c_v, c_d, meshIdx.
YART_HD inline void projectRecord(float out[12], const float* a, const float* b, const float* c, int32_t triIdx, int32_t meshIdx)
{
    const float ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
    const float ac[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
    const float nrm[3] = { ab[1] * ac[2] - ab[2] * ac[1],
                           ab[2] * ac[0] - ab[0] * ac[2],
                           ab[0] * ac[1] - ab[1] * ac[0] };
    int k = 1;
    const float ax = nrm[0] < 0.0f ? -nrm[0] : nrm[0], ay = nrm[1] < 0.0f ? -nrm[1] : nrm[1], az = nrm[2] < 0.0f ? -nrm[2] : nrm[2];
    if (ax > ay) k = (ax > az) ? 0 : 2;
    else if (az > ay) k = 2;
    const int u = (k + 1) % 3, v = (k + 2) % 3;
    const float proj[3] = { nrm[0] / nrm[k], nrm[1] / nrm[k], nrm[2] / nrm[k] };
    out[0] = proj[u];
    out[1] = proj[v];
    out[2] = a[0] * proj[0] + a[1] * proj[1] + a[2] * proj[2];
    memcpy(&out[3], &k, 4);
    const float det = ab[u] * ac[v] - ab[v] * ac[u];
    out[4] = -ab[v] / det;
    out[5] = ab[u] / det;
    out[6] = (ab[v] * a[u] - ab[u] * a[v]) / det;
    memcpy(&out[7], &triIdx, 4);
    out[8] = ac[v] / det;
    out[9] = -ac[u] / det;
    out[10] = -(ac[v] * a[u] - ac[u] * a[v]) / det;
    memcpy(&out[11], &meshIdx, 4);
}

// Error band of one triangle.  rec = the TriAccel record (n_u, n_v, n_d, k | b_u, b_v, b_d, - | c_u, c_v, c_d, -), a, b,
// c its vertices.  Out: lo / hi = a box that contains every point of a covered ray the record's test can accept (and the
// triangle itself); cls = the band's class, stored in bits 29..31 of the leaf row's fourth word: a hit whose smallest
// barycentric weight is below bandOfClass(cls) is "near an edge".
//
// lambda = fl(fl(fl(hu b_u) + fl(hv b_v)) + b_d), hu = fl(o_u + fl(t d_u)) (include/triaccel.h:183-189).  With B = |b_u| +
// |b_v|, coordinates of the hit point bounded by Mt (the triangle's own) and of the origin by M:
//   evaluation   h B (|t d_u| + 3 |hu|) + h            <= h (B (M + 4 Mt) + 1)
//   record       b_u, b_v, b_d rounded, the numerator of b_d (two products of an edge and a vertex) and the edges
//                fl(b - a) themselves                   <= h * 8 B Mt
//                the common divisor det = fl(fl(ab_u ac_v) - fl(ab_v ac_u)) scales both rows by 1 + 2h kappa,
//                kappa = (|ab_u ac_v| + |ab_v ac_u|) / |det| <= 2 Mt max(B, C)
// so E = h (max(B, C) (M + 16 Mt) + 2) bounds the error of lambda, of mue and (twice it) of their sum.  A point the test
// accepts therefore has exact weights lambda, mue >= -E, lambda + mue <= 1 + 3E w.r.t. the record's own triangle
// (the solution of the record's two affine forms), whose corners follow in double; its third coordinate lies within
// rho = h (5 (|n_d| + 3M) + 6D) of the record's plane (the residual of t = fl(fl(num) * fl(1 / fl(den)))), D = the
// longest way from a covered origin to the triangle.
YART_HD inline float bandOfClass(uint32_t cls)
{
    const uint32_t bits = (106u + 3u * cls) << 23;                     // 2^(3 cls - 21); class 7: 1.0 = every hit
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

YART_HD inline void triBand(const float rec[12], const float* a, const float* b, const float* c, const Cover& cv,
                            float lo[3], float hi[3], uint32_t& cls)
{
    double Mt = 0.0;
    for (int i = 0; i < 3; ++i) {
        const float l = a[i] < b[i] ? (a[i] < c[i] ? a[i] : c[i]) : (b[i] < c[i] ? b[i] : c[i]);
        const float h = a[i] > b[i] ? (a[i] > c[i] ? a[i] : c[i]) : (b[i] > c[i] ? b[i] : c[i]);
        lo[i] = l;
        hi[i] = h;
        const double m = coverAbs((double)l) > coverAbs((double)h) ? coverAbs((double)l) : coverAbs((double)h);
        Mt = m > Mt ? m : Mt;
    }
    cls = 0;
    if (!coverFinite(Mt)) {
        // a vertex with an inf / NaN coordinate: the edges, hence the normal, hence n_u, n_v, n_d hold inf or NaN, and t is
        // NaN, +inf or -inf for every ray -- the test never accepts.  An empty box: the builders leave the triangle out.
        for (int i = 0; i < 3; ++i) { lo[i] = 3.0e38f; hi[i] = -3.0e38f; }
        return;
    }
    bool finite = true;
    for (int i = 0; i < 11; ++i)
        if ((i & 3) != 3 && !coverFinite((double)rec[i])) finite = false;
    if (!finite) return;                            // finite vertices, a degenerate record (0 / 0): never accepts either; its own box
    uint32_t kbits;
    memcpy(&kbits, &rec[3], 4);
    const int k = (int)(kbits & 3u), u = (k + 1) % 3, v = (k + 2) % 3;
    const double nu = rec[0], nv = rec[1], nd = rec[2];
    const double bu = rec[4], bv = rec[5], bd = rec[6], cu = rec[8], cvv = rec[9], cd = rec[10];
    const double M = cv.valid ? (double)cv.M : Mt;
    double D = 0.0;
    if (cv.valid) {                                 // sqrt of coverFar2 without a sqrt: the 1-norm bounds the 2-norm
        for (int i = 0; i < 3; ++i) {
            const double x = (double)hi[i] - (double)cv.lo[i], y = (double)cv.hi[i] - (double)lo[i];
            D += x > y ? x : y;
        }
    }
    const double B = coverAbs(bu) + coverAbs(bv), C = coverAbs(cu) + coverAbs(cvv);
    const double E = kHalfUlp * ((B > C ? B : C) * (M + 16.0 * Mt) + 2.0);
    cls = 7;
    for (uint32_t q = 1; q < 7; ++q)
        if (E <= (double)bandOfClass(q)) { cls = q; break; }
    // corners of the record's triangle grown by the band: (lambda, mue) = (-E, -E), (-E, 1 + 3E'), (1 + 3E', -E)
    const double det = bu * cvv - bv * cu;
    const double big = 3.0e38;
    bool ok = coverFinite(det) && det != 0.0 && coverFinite(E);
    double blo[3] = { big, big, big }, bhi[3] = { -big, -big, -big };
    if (ok) {
        const double far = 1.0 + 3.0 * E + 2.0 * kHalfUlp;
        const double lam[3] = { -E, -E, far }, mue[3] = { -E, far, -E };
        for (int i = 0; i < 3 && ok; ++i) {
            const double rl = lam[i] - bd, rm = mue[i] - cd;
            const double pu = (cvv * rl - bv * rm) / det, pv = (bu * rm - cu * rl) / det;
            const double pk = nd - nu * pu - nv * pv;
            if (!(coverFinite(pu) && coverFinite(pv) && coverFinite(pk))) { ok = false; break; }
            const double p[3] = { pu, pv, pk };
            const int ax[3] = { u, v, k };
            for (int j = 0; j < 3; ++j) {
                blo[ax[j]] = p[j] < blo[ax[j]] ? p[j] : blo[ax[j]];
                bhi[ax[j]] = p[j] > bhi[ax[j]] ? p[j] : bhi[ax[j]];
            }
        }
    }
    if (!ok) {                                      // no usable bound: the primitive is met by every ray, every hit is checked
        cls = 7;
        for (int i = 0; i < 3; ++i) { lo[i] = -3.0e38f; hi[i] = 3.0e38f; }
        return;
    }
    const double rho = 2.0 * kHalfUlp * (5.0 * (coverAbs(nd) + 3.0 * M) + 6.0 * D);
    const double hair = 4.0 * kHalfUlp * (M > Mt ? M : Mt);
    double pad[3];
    pad[u] = hair;
    pad[v] = hair;
    pad[k] = rho + hair;
    for (int i = 0; i < 3; ++i) {
        const float l = coverFloatDown(blo[i] - pad[i]), h = coverFloatUp(bhi[i] + pad[i]);
        lo[i] = l < lo[i] ? l : lo[i];
        hi[i] = h > hi[i] ? h : hi[i];
    }
}

} // namespace yart

#include <cmath>
#include <cstddef>

namespace yart {

inline Cover makeCover(const float slo[3], const float shi[3], bool haveBounds, size_t nSpheres, float rMin)
{
    Cover c;
    std::memset(&c, 0, sizeof c);
    if (!haveBounds) return c;
    double ext[3], diag2 = 0.0, m = 0.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = (double)shi[a] - (double)slo[a];
        diag2 += ext[a] * ext[a];
        m = std::fmax(m, std::fmax(std::fabs((double)slo[a]), std::fabs((double)shi[a])));
    }
    const double diag = std::sqrt(diag2);
    double E = diag;
    if (nSpheres) {
        const double allow = 323.8 * (double)rMin;               // 2^-20 D^2 <= 0.1 r^2
        const double Es = (allow - diag) / 1.7320508075688772;
        E = std::fmin(E, std::fmax(Es, diag / 16.0));
    }
    if (!(E > 0.0)) E = std::fmax(1e-3 * m, 1e-30);              // a scene without extent
    for (int a = 0; a < 3; ++a) {
        c.lo[a] = coverFloatDown((double)slo[a] - E);
        c.hi[a] = coverFloatUp((double)shi[a] + E);
    }
    c.M = coverFloatUp(1.01 * (m + E));
    c.valid = 1;
    return c;
}

} // namespace yart
