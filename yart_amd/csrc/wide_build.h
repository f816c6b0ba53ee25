// The engine's own acceleration structure for the default (distance-pruned) rule: ONE 4-wide
// BVH over triangles AND spheres, laid out for how MI355X's memory system charges a scattered
// fetch (measured, tools/calib_gather.hip): beyond L2 a random aligned 128-byte record costs the
// same as a 64-byte or a 16-byte one (~45-50 G records/s), so a node is one 128-byte line holding
// four child records, and a leaf is a 64-byte-aligned block whose first rows are everything the
// first rejection test of each primitive needs.
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

namespace yart {

// 32-byte child record; four of them make a 128-byte node.
struct WideChild {
    float lo[3];
    float hi[3];
    uint32_t ref;       // see below
    uint32_t pad;
};
struct alignas(128) WideNode {
    WideChild child[4];
};
static_assert(sizeof(WideNode) == 128, "wide node is one 128-byte line");

// ref encoding
//   interior : bit31 = 0, bits30..0 = node index
//   leaf     : bit31 = 1, bit30 = kind (0 triangles, 1 spheres), bits29..27 = count - 1,
//              bits26..0 = leaf block offset in units of 4 rows (64 bytes) into the row array
//   kWideNone: empty child slot / "lane has no node"
constexpr uint32_t kWideLeafBit = 0x80000000u;
constexpr uint32_t kWideSphereBit = 0x40000000u;
constexpr uint32_t kWideNone = 0xFFFFFFFFu;
constexpr uint32_t kWideMaxBlocks = (1u << 27) - 1;
constexpr int kWideMaxLeaf = 8;
constexpr int kWideStack = 64;          // traversal stack entries per ray (16 in LDS + 48 in scratch)

// Leaf blocks, rows of 16 bytes:
//   triangles (count c): c rows A = n_u, n_v, n_d, bits(k | slot << 2)   -- slot = index into the
//                        reference's leaf-ordered TriAccel array = YartHit.prim
//                        c rows B = b_u, b_v, b_d, -      c rows C = c_u, c_v, c_d, -
//   spheres   (count c): c rows   = cx, cy, cz, radius;   then ceil(c/4) rows of uint32 sphere numbers
//   padded to a multiple of 4 rows.
struct WideBvh {
    std::vector<WideNode> nodes;        // nodes[0] = root
    std::vector<float> rows;            // 4 floats per row
    int depth = 0;                      // wide levels (root = 1)
    size_t numLeaves = 0;
};

// `tri` supplies the reference-order TriAccel records and, per input triangle, its leaf slot.
// maxLeaf in 1..8.  Returns YART_OK or an error status with `err` set.
int buildWideBvh(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                 const TriangleBvh& tri, const float* spheres, size_t nSpheres,
                 int maxLeaf, WideBvh& out, std::string& err);

} // namespace yart
