// pf_common.hpp -- error plumbing shared by the translation units of libprefhetch_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/prefhetch_hip.h"


// Experiment switches.  Every PF_ABL_* macro builds a library that returns WRONG RESULTS (timing-only ablations), PF_FLAT_STAMPS one that
// writes phase stamps into a debug buffer: such a build must say so (-DPF_EXPERIMENT_BUILD), goes to its own object directory (the
// Makefile hashes $(EXTRA) into BUILD), reports itself through pf_build_flags() and is refused by the Python loader unless it was
// asked for by path (PREFHETCH_HIP_LIB).  A default `make` can therefore never link an ablation object into lib/libprefhetch_hip.so.
#if defined(PF_ABL_NOEMIT) || defined(PF_ABL_NODMA) || defined(PF_ABL_NOSURV) || defined(PF_ABL_NOBAR) || defined(PF_ABL_NODRAIN) || \
    defined(PF_ABL_EXACTFLUSH) || defined(PF_ABL_I8ONLY) || defined(PF_FLAT_STAMPS) || defined(PF_DEV_ONLY_D128) || defined(PF_ABL_W8_HOTB) || defined(PF_ABL_SEL_STAGE)
#define PF_HAS_EXPERIMENT_SWITCH 1
#ifndef PF_EXPERIMENT_BUILD
#error "PF_ABL_* / PF_FLAT_STAMPS are experiment switches (wrong results / debug buffers): add -DPF_EXPERIMENT_BUILD and build into a directory of its own"
#endif
#endif

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
