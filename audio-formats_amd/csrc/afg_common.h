// afg_common.h -- shared plumbing of the C-ABI library (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/afg.h"

namespace afg {

// thread-local detail string behind afg_last_error()
void set_error(const char *fmt, ...);

// Verifies a gfx950 device is present and selected; loud failure otherwise.
int require_device();

#define AFG_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::afg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                             __FILE__, __LINE__);                                        \
            return AFG_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// Owns a device buffer filled from a host array at plan creation.
struct DeviceArray {
    void *ptr = nullptr;
    size_t bytes = 0;
    int upload(const void *host, size_t nbytes);
    void release();
};

}  // namespace afg
