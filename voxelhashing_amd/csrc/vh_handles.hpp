// vh_handles.hpp -- the opaque handles of the C ABI (include/vh_api.h): each holds one C++ host class of
// include/vh.hpp.  Shared by the translation units that implement handle-level entry points.
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "../../include/vh.hpp"

struct VhSceneRep { CUDASceneRepHashSDF impl; VhSceneRep(const HashParams& p, const VhSceneOptions& o, vhStream_t s) : impl(p, o, s) {} };
struct VhRayCast { CUDARayCastSDF impl; VhRayCast(const RayCastParams& p, vhStream_t s) : impl(p, s) {} };
struct VhMarchingCubes { CUDAMarchingCubesHashSDF impl; VhMarchingCubes(const MarchingCubesParams& p, vhStream_t s) : impl(p, s) {} };
struct VhRGBDSensor { CUDARGBDSensor impl; VhRGBDSensor(const CUDARGBDSensor::Config& c, vhStream_t s) : impl(c, s) {} };
struct VhSensorData { vh::SensorData impl; };
struct VhSensorDataReader { vh::SensorDataReader impl; };
struct VhCameraTracking { CUDACameraTrackingMultiRes impl; VhCameraTracking(unsigned int w, unsigned int h, unsigned int l, vhStream_t s) : impl(w, h, l, s) {} };
struct VhChunkGrid {
    CUDASceneRepChunkGrid impl;
    VhChunkGrid(CUDASceneRepHashSDF* s, const vh::vec3f& e, const vh::vec3i& d, const vh::vec3i& m, unsigned int l, bool en, unsigned int parts)
        : impl(s, e, d, m, l, en, parts) {}
};
struct VhReconstruction {
    Reconstruction impl;
    VhReconstruction(CUDASceneRepHashSDF* s, CUDARayCastSDF* r, CUDASceneRepChunkGrid* g, const DepthCameraParams& cp, const ReconstructionOptions& o)
        : impl(s, r, g, cp, o) {}
};

// text of the last error raised by a handle-level call on this thread (vh_last_error_message)
char* vh_last_error_buffer(); // 512 bytes, vh_c_api.cpp

// exceptions become error codes at the C boundary
template <class F> int vh_guarded(F&& f)
{
    char* buf = vh_last_error_buffer();
    try {
        f();
        return VH_OK;
    } catch (const vh::Error& e) {
        std::strncpy(buf, e.what(), 511);
        return e.code ? e.code : VH_ERR_BAD_ARGUMENT;
    } catch (const std::bad_alloc&) {
        std::strncpy(buf, "out of host memory", 511);
        return -(int)hipErrorOutOfMemory;
    } catch (const std::exception& e) {
        std::strncpy(buf, e.what(), 511);
        return VH_ERR_BAD_ARGUMENT;
    }
}
