// Shared helpers for the gfx950 iS-DQN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/isdqn_hip.h"

namespace isdqn {

void set_last_error(const char* fmt, ...);

#define ISDQN_HIP_CHECK(expr)                                                                         \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            ::isdqn::set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                                    __LINE__);                                                        \
            return ISDQN_ERR_HIP;                                                                     \
        }                                                                                             \
    } while (0)

#define ISDQN_REQUIRE(cond, code, msg)                                         \
    do {                                                                       \
        if (!(cond)) {                                                         \
            ::isdqn::set_last_error("%s (%s:%d)", msg, __FILE__, __LINE__);    \
            return code;                                                       \
        }                                                                      \
    } while (0)

// Unsigned division by a runtime constant, valid for 0 <= n < 2^31 (all index spaces here are < 2^31).
struct FastDiv {
    uint32_t d, m, sh;
    __host__ __device__ FastDiv() : d(1), m(0), sh(0) {}
    __host__ __device__ explicit FastDiv(uint32_t div) : d(div), m(0), sh(0) {
        if (div > 1) {
            uint32_t lg = 0;
            while ((1u << lg) < div) ++lg;  // ceil(log2(div))
            uint32_t p = 31 + lg;
            m = (uint32_t)(((1ull << p) + div - 1) / div);
            sh = p - 32;
        }
    }
    // Branch-free on the device: a `d == 1 ? n : ...` expression becomes a scalar branch, which splits the K-loop
    // bodies that use it into several basic blocks and stops the scheduler from interleaving their pieces.
    __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t q = __umulhi(n, m) >> sh;
        return d == 1 ? n : q;
#else
        return d == 1 ? n : (uint32_t)((((uint64_t)n * m) >> 32) >> sh);
#endif
    }
    __host__ __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
        q = div(n);
        r = n - q * d;
    }
};

// Development switches (phase stamps, ablation, environment overrides) exist only in -DISDQN_DEV builds
// (`python is-dqn_amd/build.py --variant dev -DISDQN_DEV`): the product library reads no environment variable and has
// no code path that skips work.
#if defined(ISDQN_DEV)
#define ISDQN_DEV_ENV(name) (getenv(name) != nullptr)
#else
#define ISDQN_DEV_ENV(name) (false)
#endif

// Development builds: with ISDQN_DEBUG_OCCUPANCY set, print once per kernel how many workgroups per CU the runtime co-schedules
// and the grid it was launched with.
#if defined(ISDQN_DEV)
#define ISDQN_REPORT_OCCUPANCY(kernel, threads, lds_bytes, grid)                                                              \
    do {                                                                                                                      \
        static bool reported_ = false;                                                                                        \
        if (!reported_ && getenv("ISDQN_DEBUG_OCCUPANCY")) {                                                                  \
            reported_ = true;                                                                                                 \
            int nb_ = 0;                                                                                                      \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, reinterpret_cast<const void*>(kernel), (threads),          \
                                                             (lds_bytes)) == hipSuccess)                                      \
                fprintf(stderr, "[isdqn] %s: grid %d x %d threads, %d B LDS -> %d workgroups per CU\n", __PRETTY_FUNCTION__,  \
                        (int)(grid), (int)(threads), (int)(lds_bytes), nb_);                                                  \
        }                                                                                                                     \
    } while (0)
#else
#define ISDQN_REPORT_OCCUPANCY(kernel, threads, lds_bytes, grid) do {} while (0)
#endif

// Process-level state of the library is keyed by HIP device: the ">64 KB of LDS" function attribute is per device, and so
// are the weight-gradient side stream and its events.
constexpr int ISDQN_MAX_DEVICES = 16;
static inline int current_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= ISDQN_MAX_DEVICES) d = 0;
    return d;
}
// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device), raised when a larger request comes
struct LdsConfigured {
    int bytes[ISDQN_MAX_DEVICES] = {0};
};
template <class F>
static inline int ensure_dynamic_lds(F kernel, int lds_bytes, LdsConfigured& state) {
    if (lds_bytes <= 65536) return ISDQN_OK;
    const int d = current_device_slot();
    if (lds_bytes > state.bytes[d]) {
        ISDQN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        state.bytes[d] = lds_bytes;
    }
    return ISDQN_OK;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace isdqn

// -DISDQN_EMPTY (measurement only, results are wrong by construction): every kernel of the learn step returns at once -- what is left
// of the step is its launches, its graph edges and its kernel boundaries (scripts/r3/empty_step.sh)
#if defined(ISDQN_EMPTY)
#define ISDQN_EMPTY_KERNEL_RETURN return;
#else
#define ISDQN_EMPTY_KERNEL_RETURN
#endif
