// Host-side acceleration-structure build for the MI355X intersection engine.
//
// Produces, from the flat scene description of include/yart_amd.h:
//   * the reference's flattened 32-byte node array and leaf-ordered 48-byte
//     TriAccel array, bit-for-bit what elaic/yart's BvhAccel constructor builds
//     (src/bvhaccel.cpp:100-151, :215-234, :309-363; include/triaccel.h:54-108),
//   * the same 32-byte node records re-laid as SIBLING PAIRS (64-byte, 64-byte
//     aligned) for the device: one fetch brings both children of a node,
//   * a sphere BVH in the same pair layout (the reference loops linearly over
//     its spheres, include/scene.h:142-144; the engine reproduces that loop's
//     result with an order-independent acceptance rule, see traverse.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/yart_amd.h"
#include "cover.h"

namespace yart {

// One child record of the device layout: the reference's 32-byte node record
// with `offset` replaced by an encoded reference the kernel can follow without
// further decoding:
//   interior : bit31 = 0, bits30..29 = the child's own split axis,
//              bits28..0 = index of the pair holding ITS two children
//   leaf     : bit31 = 1, bits30..3 = first primitive slot, bits2..0 = count (0..7)
struct DevNodeRec {
    float lo[3];
    float hi[3];
    uint32_t ref;
    uint8_t num_prims;
    uint8_t split_axis;
    uint8_t pad[2];
};
struct alignas(64) DevNodePair {
    DevNodeRec child[2];    // [0] = left (index+1 in the reference order), [1] = right
};
static_assert(sizeof(DevNodeRec) == 32, "device node record is 32 bytes");
static_assert(sizeof(DevNodePair) == 64, "device node pair is 64 bytes");

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kEmptyLeaf = kLeafBit;            // leaf with zero primitives at slot 0: doubles as the
                                                     // kernel's "lane has no node" marker
constexpr uint32_t kPadLeaf = kLeafBit | (1u << 3);  // the empty leaf next to the root in pair 0 (slot 1,
                                                     // zero primitives): distinct from kEmptyLeaf
constexpr uint32_t kMaxPairs = 1u << 29;
constexpr uint32_t kMaxLeafSlot = 1u << 28;
constexpr int kMaxDepth = 64;                        // src/bvhaccel.cpp:244

inline uint32_t interiorRef(uint32_t pair, uint32_t axis) { return (axis << 29) | pair; }
inline uint32_t leafRef(uint32_t slot, uint32_t count) { return kLeafBit | (slot << 3) | count; }

struct TriangleBvh {
    std::vector<YartBvhNode> nodes;       // reference layout and order
    std::vector<YartTriAccel> triaccel;   // leaf order
    std::vector<uint32_t> slotOfInput;    // caller's triangle number -> leaf slot (index into triaccel)
    std::vector<DevNodePair> pairs;       // pairs[0] = { root, empty leaf }
    int depth = 0;
};

struct SphereBvh {
    std::vector<DevNodePair> pairs;       // pairs[0] = { root, empty leaf }
    std::vector<float> records;           // 4 floats per sphere, leaf-slot order
    std::vector<uint32_t> index;          // leaf slot -> sphere number in the caller's array
    int depth = 0;
};

// Returns YART_OK or an error status; `err` receives a message on failure.
int buildTriangleBvh(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                     TriangleBvh& out, std::string& err);
int buildSphereBvh(const float* spheres, size_t nSpheres, const Cover& cover, SphereBvh& out, std::string& err);

// One TriAccel record (include/triaccel.h:54-108) with the host's arithmetic; the device builder
// (ref_build_dev.hip) uses it for the degenerate triangles whose records hold NaNs.
void projectTriangleHost(YartTriAccel& ta, const float* a, const float* b, const float* c, int32_t triIdx, int32_t meshIdx);

} // namespace yart
