// Shared helpers for the gfx950 kernels of libeigspec_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "eigspec.h"

namespace egs {

constexpr int kWave = 64;   // CDNA wavefront

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define EGS_HIP(call)                                            \
    do {                                                         \
        int _st = ::egs::check_hip((call), #call);               \
        if (_st != 0) return _st;                                \
    } while (0)

#define EGS_ARG(cond, idx)                                       \
    do {                                                         \
        if (!(cond)) {                                           \
            ::egs::set_error("argument %d invalid: %s", (idx), #cond); \
            return -(idx);                                       \
        }                                                        \
    } while (0)

#define EGS_LAUNCH_CHECK() EGS_HIP(hipGetLastError())

inline hipStream_t as_stream(egs_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Compute units of the CURRENT device (queried per call: a process may drive several GPUs / partition modes, and the
// resident grids of the work-queue kernels are sized from it).  hipDeviceGetAttribute is a table lookup, no device sync.
inline int current_device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    return cus;
}

// ---- device-side helpers ---------------------------------------------------------------------------
// LDS traffic of one wavefront is executed in order, so cross-lane hand-offs through LDS inside a wave
// only need the compiler not to reorder the accesses: a wavefront-scope fence.
__device__ __forceinline__ void wave_lds_sync() {
#ifdef EGS_WAVE_SYNC_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    // compiler-only ordering: DS instructions of one wave are issued and executed in program order, so no
    // s_waitcnt is needed between a lane's LDS store and another lane's later load of it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#endif
}

// Broadcast lane `src` (wave-uniform) of a double to every lane through two v_readlane_b32.
__device__ __forceinline__ double bcast_lane(double x, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), s);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), s);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float bcast_lane(float x, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), __builtin_amdgcn_readfirstlane(src)));
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// 1/x to ~1 ulp: hardware seed + two Newton steps (cheaper than the IEEE division expansion; used where
// the consumer is a Householder reflector, which is insensitive to the last bit).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
#ifdef EGS_ONE_NEWTON
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    // one cubically convergent step: r (1 + e + e^2), e = 1 - x r  (seed error 5e-8 -> < 1 ulp)
    const double e = __builtin_fma(-x, r, 1.0);
    const double p = __builtin_fma(e, e, e);
    return __builtin_fma(r, p, r);
#endif
}

// 1/sqrt(x) for normal positive x: hardware seed + one cubically convergent step.
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
#ifdef EGS_ONE_NEWTON
    const double e = __builtin_fma(-0.5 * x * r, r, 0.5);
    return __builtin_fma(r, e, r);
#else
    // y (1 + e/2 + 3 e^2 / 8), e = 1 - x y^2
    const double e = __builtin_fma(-x * r, r, 1.0);
    const double p = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(r * e, p, r);
#endif
}

}  // namespace egs
