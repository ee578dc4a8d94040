// Shared host-side helpers for libralf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ralf_hip.h"

namespace ralf {
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return RALF_ERR_LAUNCH;
    }
    return RALF_OK;
}
}  // namespace ralf

#define RALF_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            ralf::set_error(__VA_ARGS__);    \
            return RALF_ERR_INVALID;         \
        }                                    \
    } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
