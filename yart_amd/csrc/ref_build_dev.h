// The REFERENCE's triangle tree (bvh_build.h: flattened 32-byte nodes in the reference's order, TriAccel
// records in the reference's leaf order) built on the GPU (ref_build_dev.hip), bit for bit the arrays the
// host builder and elaic/yart's BvhAccel constructor produce.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "cover.h"

namespace yart {

struct DeviceRefTree {
    // device arrays, owned by the caller after build(): hipFree
    void* nodes = nullptr;          // YartBvhNode[numNodes], reference order
    void* pairs = nullptr;          // DevNodePair[numPairs], pairs[0] = { root, empty leaf }
    void* triaccel = nullptr;       // YartTriAccel[numTriangles], leaf order
    void* slotOfInput = nullptr;    // uint32[numTriangles]: caller's triangle number -> leaf slot
    size_t numNodes = 0, numPairs = 0, numTriangles = 0;
    int depth = 0;
    float rootLo[3] = { 0, 0, 0 }, rootHi[3] = { 0, 0, 0 };
    double ms = 0.0;
    size_t medianSplitsOnHost = 0;  // ranges whose midpoint split was one-sided: std::nth_element ran on the host
    size_t projectionsOnHost = 0;   // degenerate triangles whose NaN-bearing TriAccel was recomputed on the host
    void release();                 // frees the four arrays
};

// dVerts / dTris: the scene on the device (3 floats per vertex, 5 int32 per triangle, as yart_accel_create
// takes them); verts / tris: the same arrays on the host (read only for the rare host fix-ups above).
// Returns YART_OK, or an error status with `err` set.  *useHost is set when the scene is one this builder
// does not take (NaN coordinates, too many one-sided splits): nothing is wrong with the scene, the caller
// builds on the host instead.
int buildReferenceTreeDevice(const float* dVerts, const float* verts, size_t nVerts, const int32_t* dTris, const int32_t* tris,
                             size_t nTris, DeviceRefTree& out, bool* useHost, std::string& err);

// The engine's sphere tree of the reference-tree path (bvh_build.h: SphereBvh) from the same level-synchronous build:
// leaves of at most 4 spheres, numbered left to right; identical to the host builder's arrays.
struct DeviceSphereTree {
    void* pairs = nullptr;          // DevNodePair[numPairs]
    void* records = nullptr;        // 4 floats per sphere, leaf-slot order
    void* index = nullptr;          // uint32: leaf slot -> sphere number
    size_t numPairs = 0, numSpheres = 0;
    int depth = 0;
    float rootLo[3] = { 0, 0, 0 }, rootHi[3] = { 0, 0, 0 };
    double ms = 0.0;
    void release();
};
// dSpheres: 4 floats per sphere on the device.  *useHost as above (NaN, too many one-sided splits).
int buildSphereTreeDevice(const float* dSpheres, size_t nSpheres, const Cover& cover, DeviceSphereTree& out, bool* useHost, std::string& err);

} // namespace yart
