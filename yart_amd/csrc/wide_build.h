// The engine's own acceleration structure for the default (distance-pruned) rule: ONE 4-wide
// BVH over triangles AND spheres, laid out for how MI355X's memory system charges a scattered
// fetch (measured, tools/calib_gather.hip + profiles/): what a one-ray-per-lane traversal pays for is
// the NUMBER of scattered line requests and 16-byte lane requests it issues (~1 lane request per
// clock per CU, ~50 G random lines/s beyond L2), not their bytes.  So a node is ONE 64-byte line
// holding four quantised child boxes (four 16-byte lane requests per visit instead of the sixteen a
// binary tree with full-precision boxes needs for the same four boxes), and the
// leaf data every primitive's first rejection test needs is packed densely, apart from the rest.
//
// The reference has no counterpart (its tree is the binary one of src/bvhaccel.cpp, which the
// library also keeps, bit for bit, for YART_REFERENCE_RULE / YART_REFERENCE_TREE).  A closest hit
// does not depend on the tree it is found with (except exact-t ties between different primitives),
// so parity with the reference is kept at the bar of BASELINE.json: same primitive for >= 99.99 % of
// rays, t within 1e-5 relative.  Triangle rows are the reference's TriAccel bits (include/triaccel.h:16-32).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "cover.h"

namespace yart {

// The BUILDERS' form of a node (the kernel reads the packed form further down; yart_accel_get_wide hands this one
// out): four child boxes quantised to 8 bits per plane on the node's own grid (origin + q * scale per axis;
// lower planes rounded down, upper planes up, so a quantised box always contains the exact one; scale is a step
// the packed form can hold, packedStep()), SoA by plane so a plane of all four children is one dword.
//   dwords 0-2  origin x, y, z (float)          dwords 3-5   scale x, y, z (float)
//   dword  6-8  lower planes x, y, z, byte c = child c      dwords 9-11  upper planes x, y, z
//   dwords 12-15 child refs
struct alignas(64) WideNode {
    float origin[3];
    float scale[3];
    uint8_t qlo[3][4];
    uint8_t qhi[3][4];
    uint32_t ref[4];
};
static_assert(sizeof(WideNode) == 64, "wide node is one 64-byte line");

// ref encoding
//   interior : bit31 = 0, bits30..0 = node index
//   leaf     : bit31 = 1, bit30 = kind (0 triangles, 1 spheres), bits29..27 = count - 1,
//              bits26..0 = index of the leaf's first primitive in the row arrays
//   kWideNone: empty child slot / "lane has no node"
constexpr uint32_t kWideLeafBit = 0x80000000u;
constexpr uint32_t kWideSphereBit = 0x40000000u;
constexpr uint32_t kWideNone = 0xFFFFFFFFu;
constexpr uint32_t kWideMaxPrims = (1u << 27) - 1;
constexpr int kWideMaxLeaf = 8;
constexpr int kWideStack = 64;          // traversal stack entries per ray (16 in LDS + 48 in scratch)

// Leaf storage is split by WHEN a traversal needs it.  rowsA holds, densely (16 bytes per
// primitive, a leaf's primitives consecutive, leaves in depth-first order so neighbours in space
// are neighbours in memory), the row every test starts with:
//   triangle  n_u, n_v, n_d, bits(k | slot << 2)   -- the plane row of the reference's TriAccel
//             (include/triaccel.h:16-20); slot = index into the reference's leaf-ordered array = YartHit.prim
//   sphere    cx, cy, cz, radius
// rowsBC holds, 32 bytes per primitive at the same index, what only a ray that passed the first
// test reads: triangle rows b_u, b_v, b_d, - | c_u, c_v, c_d, - ; sphere: its number in Scene::shapes_.
// Most tests end at the plane distance, so the working set of a batch is the nodes plus rowsA.
struct WideBvh {
    std::vector<WideNode> nodes;        // nodes[0] = root; 64 bytes each
    std::vector<float> rowsA;           // 4 floats per primitive
    std::vector<float> rowsBC;          // 8 floats per primitive
    int depth = 0;                      // wide levels (root = 1)
    size_t numLeaves = 0;
    float rootLo[3] = { 0, 0, 0 }, rootHi[3] = { 0, 0, 0 };     // bounds of all (padded) primitive boxes: every node's grid lies inside
    float padMax = 0.0f;                // Cover::padMax of this scene (cover.h)
};

// ---------------------------------------------------------------------------------------------------------
// What the traversal kernel reads: the same tree PACKED into one buffer of 16-byte units, so that a node visit
// is THREE 16-byte lane requests instead of four (the request count, not the byte count, is what the kernel
// pays for: profiles/r02_ab_extraload.txt).  The children of a node form one block of consecutive units --
// its interior children first (64 bytes = 4 units each, 64-byte aligned), then its leaf children's rows --
// so the four child references shrink to one block address and four one-byte offsets; the grid steps are
// kept to 10 bits each (8 exponent + 2 mantissa bits, rounded up).
//
//   node, 4 units (the fourth unused):
//     unit 0   origin x, y, z (float) | steps: bits 30..21 x, 20..11 y, 10..1 z = bits 30..21 of the float step
//     unit 1   lower planes x, y, z (byte c = child c) | upper planes x
//     unit 2   upper planes y, z | block: (unit index of the children block) << 2 | offsets: byte c = (unit offset of
//              child c inside the block) << 2 | class of child c
//   reference = (unit index << 2) | class;  class bit 0 = leaf, bit 1 = sphere;  the root node is reference 0
//   triangle leaf, 1 unit:  n_u, n_v, n_d, bits(k | slot << 2 | band << 29)   -- second stage: rows b and c of TriAccel
//                           record `slot` of the reference's leaf-ordered array (the accel's dTriAccel), nothing is
//                           duplicated; band = class of the triangle's error band (cover.h: triBand, bandOfClass)
//   sphere leaf, 2 units:   cx, cy, cz, radius | sphere number, 0, 0, 0
// Leaves hold one primitive.  An unused child slot keeps its inverted box (never entered); its offset byte is 0.
constexpr uint32_t kPackLeaf = 1u, kPackSphere = 2u;
constexpr uint32_t kPackBandShift = 29;                  // fourth word of a triangle row: k | slot << 2 | band class << 29
constexpr uint32_t kPackSlotMask = 0x1FFFFFFCu;
constexpr size_t kPackMaxUnits = (size_t(1) << 30) - 2;

struct PackedWide {
    std::vector<uint32_t> words;        // 4 per unit
    size_t numUnits = 0;
};

// The 10-bit step the packed node stores for a float step `s` (> 0, finite): the smallest representable value >= s,
// as the float it decodes to.  The builders quantise child boxes on THIS step, so WideNode::scale holds it too.
float packedStep(float s);

// WideBvh (single-primitive leaves) -> packed buffer.  Nodes keep their order: the block of node i's children
// follows the block of node i - 1's.  false: the tree does not fit the format (too many units).
bool packWide(const WideBvh& bvh, PackedWide& out);
// The inverse, for inspection and tests (yart_accel_get_wide): nodes numbered in the order a breadth-first walk
// from the root meets them, rows in the order it meets their leaves.  `triaccel`: the reference's leaf-ordered records.
bool unpackWide(const uint32_t* words, size_t numUnits, const YartTriAccel* triaccel, size_t nTriangles, WideBvh& out);

// `tri` supplies the reference-order TriAccel records and, per input triangle, its leaf slot.
// maxLeaf is kept for the callers that pass it; leaves hold one primitive (the packed format has no count, and
// larger leaves measured slower on MI355X: tools/exp_wide.py).  Returns YART_OK or an error status with `err` set.
// `cover` (cover.h): the rays the primitives' boxes are padded for.
int buildWideBvh(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                 const TriangleBvh& tri, const float* spheres, size_t nSpheres,
                 int maxLeaf, const Cover& cover, WideBvh& out, std::string& err);

// The cover of a scene: bounds of its finite vertex coordinates and sphere extents, widened (cover.h: makeCover).
// padMax is left 0: the builders fill it in (WideBvh::padMax, DeviceWide::padMax), or trianglePadMax for a scene that
// gets no wide tree.  The triangles' vertex indices must have been checked.
Cover sceneCover(const float* verts, size_t nVerts, const float* spheres, size_t nSpheres);
float trianglePadMax(const float* verts, const int32_t* tris, size_t nTris, const Cover& cover);

} // namespace yart
