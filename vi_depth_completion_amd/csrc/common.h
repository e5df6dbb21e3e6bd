// Shared helpers for libvidc.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/vidc.h"

namespace vidc {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(vidc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define VIDC_REQUIRE(cond, code, ...)                  \
    do {                                               \
        if (!(cond)) {                                 \
            ::vidc::set_error(__VA_ARGS__);            \
            return (code);                             \
        }                                              \
    } while (0)

#define VIDC_CHECK_LAUNCH(name)                                                          \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            ::vidc::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return VIDC_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define VIDC_HIP(call)                                                                        \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ::vidc::set_error("%s failed: %s", #call, hipGetErrorString(e__));                \
            return VIDC_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace vidc
