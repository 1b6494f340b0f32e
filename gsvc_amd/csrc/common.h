// gsvc_amd/csrc/common.h — shared host-side helpers of libgsvc_hip.so (error slot, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/gsvc_hip.h"

namespace gsvc {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return GSVC_E_LAUNCH;
    }
    return GSVC_OK;
}

// bench.py's per-kernel timer: brackets a launch with events on its stream when profiling is enabled
bool profile_enabled();
bool bwd_probe_enabled();      // gsvc_profile_enable bit 1: k_blend_bwd_tile counts its replays (diagnostic instantiation)
bool deterministic();           // gsvc_set_deterministic: launches whose float sums depend on workgroup order take their fixed-order form
void profile_begin(const char *name, hipStream_t s);
void profile_end(hipStream_t s);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(const char *name, hipStream_t stream) : s(stream), on(profile_enabled())
    {
        if (on) profile_begin(name, s);
    }
    ~ProfScope()
    {
        if (on) profile_end(s);
    }
};

inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

}  // namespace gsvc

#define GSVC_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            gsvc::set_error(__VA_ARGS__);  \
            return GSVC_E_INVALID;         \
        }                                  \
    } while (0)
