// vpf_common.h -- shared host/device helpers for libvipformer_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vipformer_hip.h"

#define VPF_WAVE 64

#define VPF_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return VPF_ERR_HIP;           \
    } while (0)

static inline int vpf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ bf16 helpers
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; NaN stays NaN (quiet)
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ------------------------------------------------------------------ wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

// ------------------------------------------------------------------ counter-based dropout RNG
// rng_state (device memory, 4 x uint32): {seed_lo, seed_hi, step, reserved}.  A keep decision is a
// pure function of (state, site, element index) so backward regenerates forward's mask and a
// captured hipGraph gets fresh masks every replay by bumping `step` on the device.
__device__ __forceinline__ uint32_t vpf_hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct VpfRng { uint32_t k0, k1; uint32_t thresh; float scale; };
__device__ __forceinline__ VpfRng vpf_rng_init(const uint32_t* rng_state, uint32_t site, float p)
{
    VpfRng r;
    uint32_t s0 = rng_state[0], s1 = rng_state[1], st = rng_state[2];
    r.k0 = vpf_hash32(s0 ^ vpf_hash32(site * 0x9E3779B9u + 0x85ebca6bu));
    r.k1 = vpf_hash32(s1 + st * 0x9E3779B9u + 0xc2b2ae35u);
    // drop iff rand32 < thresh
    double t = (double)p * 4294967296.0;
    r.thresh = p <= 0.f ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)t);
    r.scale = p < 1.f ? 1.0f / (1.0f - p) : 0.f;
    return r;
}
__device__ __forceinline__ uint32_t vpf_rand32(const VpfRng& r, uint64_t idx)
{
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    return vpf_hash32(vpf_hash32(lo ^ r.k0) + r.k1 + hi * 0x27d4eb2fu);
}
__device__ __forceinline__ bool vpf_keep(const VpfRng& r, uint64_t idx) { return vpf_rand32(r, idx) >= r.thresh; }
