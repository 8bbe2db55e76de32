// pf_common.hpp -- error plumbing shared by the translation units of libprefhetch_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/prefhetch_hip.h"

namespace pf {

std::string &last_error_ref();
inline pf_status fail(pf_status s, const std::string &msg) { last_error_ref() = msg; return s; }

#define PF_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return ::pf::fail(e_ == hipErrorOutOfMemory ? PF_ERR_OOM : PF_ERR_HIP,                     \
                              std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

// Makes `device` current for the scope of a call and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) { err = hipSetDevice(device); switched = (err == hipSuccess); }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

#define PF_GUARD(device)                                                                               \
    ::pf::DeviceGuard guard_(device);                                                                  \
    if (guard_.err != hipSuccess) return ::pf::fail(PF_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err))

inline hipStream_t as_stream(pf_stream s) { return reinterpret_cast<hipStream_t>(s); }

// base matrix of a flat index, for the translation units that read it (defined in pf_flat.hip; not part of the ABI)
const float *flat_base_device(const pf_flat *f, size_t *nb, uint32_t *d, int *device);

}  // namespace pf
