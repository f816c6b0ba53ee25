// The 4-wide tree of wide_build.h built on the GPU (wide_build_dev.hip).  Two phases, so that the
// first -- everything that needs only the scene -- can run while the host still builds the reference's
// tree, whose leaf order (the primitive ids of YartHit) the leaf rows of the second phase carry.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "cover.h"

namespace yart {

struct DeviceWide {
    void* units = nullptr;          // the packed tree (wide_build.h: PackedWide) on the device, owned by the caller after finish(): hipFree
    size_t numUnits = 0;
    size_t numNodes = 0, numPrims = 0;
    int depth = 0;
    float rootLo[3] = { 0, 0, 0 }, rootHi[3] = { 0, 0, 0 };     // bounds of all (padded) primitive boxes
    float padMax = 0.0f;            // Cover::padMax of this scene (cover.h)
    double msBuild = 0.0;           // wall time from the start of topology() to the end of finish()
    double msTopology = 0.0;        // of it: phase 1 (upload of the scene, sort, topology, fit, plan, wide nodes)
    double msFinish = 0.0;          // and phase 2 (leaf rows, exact-size arrays)
};

class DeviceWideBuilder {
public:
    DeviceWideBuilder();
    ~DeviceWideBuilder();
    DeviceWideBuilder(const DeviceWideBuilder&) = delete;
    DeviceWideBuilder& operator=(const DeviceWideBuilder&) = delete;
    // Host arrays as yart_accel_create takes them; the accel's device must be current in the calling thread.
    // YART_ERR_TREE_TOO_DEEP: the linear BVH of this scene does not fit the traversal stack (use the host builder).
    // cover (cover.h): the rays the primitives' boxes are padded for, as the host builder takes it.
    int topology(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris, const float* spheres, size_t nSpheres,
                 const Cover& cover, std::string& err);
    // dTriAccel: the reference's TriAccel records in leaf order on the device; slotOfInput: leaf slot per input triangle,
    // a host array or (slotsOnDevice) a device array.
    int finish(const void* dTriAccel, const uint32_t* slotOfInput, bool slotsOnDevice, DeviceWide& out, std::string& err);
    // The scene as topology() uploaded it; valid until finish() or destruction (ref_build_dev.hip reads it).
    const float* deviceVerts() const;
    const int32_t* deviceTris() const;
    const float* deviceSpheres() const;

private:
    struct State;
    State* state_ = nullptr;
};

} // namespace yart
