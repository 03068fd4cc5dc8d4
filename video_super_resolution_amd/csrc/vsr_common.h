// vsr_common.h -- shared host-side helpers for the gfx950 kernels behind include/vsr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/vsr_hip.h"

namespace vsr {

inline char* err_buf() {
    static thread_local char buf[256] = "";
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 256, fmt, ap);
    va_end(ap);
    return code;
}

// Which kernel a launcher routed its last call to (thread-local; tests log it per layer: vsr_last_route()).
inline char* route_buf() {
    static thread_local char buf[96] = "";
    return buf;
}
inline void route(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(route_buf(), 96, fmt, ap);
    va_end(ap);
}

// The batch the convolution launchers DECIDE by (kernel, tile width, split-K) when it is not the batch they run:
// vsr_conv2d_route_batch(num, den), both > 0, makes every following launch of this thread choose as if its batch N were N * num / den -- a
// trunk evaluated on a part of its usual batch (VSR's streaming mode: the frames a window shares with the previous one are cached; the
// hourglass then sees 2 of 4 frames, FlowNet2 1 of 2 pairs -- its per-frame layers 2 of 4 frames) runs the kernels of the full batch, and
// a frame's result does not depend on the batch it travelled in.  den = 0 (default): decide by the actual batch.
struct RouteScale {
    int num, den;
};
inline RouteScale& route_scale_ref() {
    static thread_local RouteScale r = {0, 0};
    return r;
}
inline int route_batch(int N) {
    const RouteScale& r = route_scale_ref();
    if (r.den <= 0 || r.num <= 0) return N;
    const long long v = ((long long)N * r.num + r.den - 1) / r.den;
    return v > 65535 ? 65535 : (v < 1 ? 1 : (int)v);
}

// Called right after a kernel launch: reports launch-configuration errors without synchronising.
inline int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VSR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return VSR_OK;
}

inline hipStream_t S(vsr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// hipFuncSetAttribute (dynamic LDS beyond 64 KB) holds per DEVICE: a launcher keeps one bit per device ordinal and
// sets the attribute again the first time it runs with another device current.
inline int current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 63;
}
inline bool device_marked(unsigned long long mask) { return (mask >> current_device()) & 1ull; }
inline void mark_device(unsigned long long& mask) { mask |= 1ull << current_device(); }

}  // namespace vsr

// Two libraries are built from these sources (Makefile): libvsr_hip.so -- the shipping kernels behind include/vsr_hip.h -- and
// libvsr_hip_xcheck.so (-DVSR_BUILD_XCHECK: include/vsr_hip_xcheck.h on top), which ALSO holds the superseded builds kept as
// bit-identity cross-checks (the first gather kernel, the two-waves-per-SIMD fused stage, the LDS-ring tail, the transposing 1x1,
// the five-set ring, the LDS warp ...), the stamped diagnostic builds and the run-time switches that select them.  In the shipping
// library the switches are compile-time constants at their defaults and the superseded kernels do not exist.
#ifdef VSR_BUILD_XCHECK
#define VSR_X 1
#define VSR_TUNABLE static int
#else
#define VSR_X 0
#define VSR_TUNABLE static constexpr int
#endif

#define VSR_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) return vsr::fail(VSR_E_ARG, __VA_ARGS__); \
    } while (0)
