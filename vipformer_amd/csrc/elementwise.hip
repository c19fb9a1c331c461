// elementwise.hip -- normalisation, dropout/residual, reductions, pooling, BatchNorm pieces.
// Replaces the aten layer_norm / batch_norm / dropout / add / max / mean / cat calls (and their
// autograd backward) of vipformer/model/pointcloud/partseg.py:100-116,191-213,519-525,547 and
// utils.py:156,163,179-189.  All kernels are HBM-bound streaming kernels: coalesced rows,
// wavefront (64-lane) reductions, fp32 statistics.
#include "vpf_common.h"

template <typename T> __device__ __forceinline__ float ld_f(const T* p, size_t i);
template <> __device__ __forceinline__ float ld_f<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_f<h16_t>(const h16_t* p, size_t i) { return h16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void st_f(T* p, size_t i, float v);
template <> __device__ __forceinline__ void st_f<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_f<h16_t>(h16_t* p, size_t i, float v) { p[i] = f32_to_h16(v); }

static inline int grid_for(long n, int per_block, int cap = 4096)
{
    long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// =============================================================================== cast
__global__ void cast_f32_h16_kernel(const float* __restrict__ x, h16_t* __restrict__ y, long n)
{
    const long n4 = n / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        uint2 o; o.x = pack_h16x2(v.x, v.y); o.y = pack_h16x2(v.z, v.w);
        reinterpret_cast<uint2*>(y)[i] = o;
    }
    for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = f32_to_h16(x[i]);
}
extern "C" int vpf_cast_f32_h16(const float* x, void* y, long n, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !y) return VPF_ERR_NULL;
    if (n <= 0) return n == 0 ? VPF_OK : VPF_ERR_BADSHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return VPF_ERR_BADALIGN;
    hipLaunchKernelGGL(cast_f32_h16_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, (h16_t*)y, n);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

__global__ void cast_h16_f32_kernel(const h16_t* __restrict__ x, float* __restrict__ y, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = h16_to_f32(x[i]);
}
extern "C" int vpf_cast_h16_f32(const void* x, float* y, long n, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !y) return VPF_ERR_NULL;
    if (n <= 0) return n == 0 ? VPF_OK : VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(cast_h16_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, y, n);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== LayerNorm
// One wavefront per row, D <= 512 values in registers (lane owns columns lane + 64 i).
#define LN_MAXI 8
template <typename TIN>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const TIN* __restrict__ x, const float* __restrict__ pos, int pos_rows,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          h16_t* __restrict__ y, float* __restrict__ xsum,
                                                          float* __restrict__ mean, float* __restrict__ rstd, long rows, int D, float eps)
{
    const int lane = threadIdx.x & 63;
    const long wave0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = wave0; r < rows; r += nw) {
        float v[LN_MAXI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXI; ++i) {
            const int c = lane + 64 * i;
            v[i] = 0.f;
            if (c < D) {
                v[i] = ld_f<TIN>(x, (size_t)r * D + c);
                if (pos) v[i] += pos[(size_t)(r % pos_rows) * D + c];
                if (xsum) xsum[(size_t)r * D + c] = v[i];
                s += v[i];
            }
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXI; ++i) { const int c = lane + 64 * i; if (c < D) { const float d = v[i] - mu; q += d * d; } }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXI; ++i) {
            const int c = lane + 64 * i;
            if (c < D) y[(size_t)r * D + c] = f32_to_h16((v[i] - mu) * rs * gamma[c] + beta[c]);
        }
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}
extern "C" int vpf_layernorm_fwd(const void* x, int x_is_h16, const float* pos, int pos_rows, const float* gamma,
                                 const float* beta, void* y_h16, float* xsum, float* mean, float* rstd, long rows, int D,
                                 float eps, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !gamma || !beta || !y_h16 || !mean || !rstd) return VPF_ERR_NULL;
    if (rows < 0 || D <= 0 || D > 64 * LN_MAXI || (pos && pos_rows <= 0)) return VPF_ERR_BADSHAPE;
    if (rows == 0) return VPF_OK;
    const int grid = grid_for(rows, 4);
    if (x_is_h16)
        hipLaunchKernelGGL(layernorm_fwd_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, pos, pos_rows,
                           gamma, beta, (h16_t*)y_h16, xsum, mean, rstd, rows, D, eps);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, pos, pos_rows,
                           gamma, beta, (h16_t*)y_h16, xsum, mean, rstd, rows, D, eps);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// dx = (dres ? dres : 0) + rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)); dgamma += dy*xhat; dbeta += dy
template <typename TX, typename TDX>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const h16_t* __restrict__ dy, const TX* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ dres,
                                                          TDX* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ ws, long rows, int D)
{
    __shared__ float red[2][4][64 * LN_MAXI];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long wave0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    float ag[LN_MAXI], ab[LN_MAXI], gm[LN_MAXI];
#pragma unroll
    for (int i = 0; i < LN_MAXI; ++i) { ag[i] = ab[i] = 0.f; const int c = lane + 64 * i; gm[i] = c < D ? gamma[c] : 0.f; }
    for (long r = wave0; r < rows; r += nw) {
        const float mu = mean[r], rs = rstd[r];
        float xh[LN_MAXI], g[LN_MAXI];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXI; ++i) {
            const int c = lane + 64 * i;
            xh[i] = g[i] = 0.f;
            if (c < D) {
                const float d = h16_to_f32(dy[(size_t)r * D + c]);
                xh[i] = (ld_f<TX>(x, (size_t)r * D + c) - mu) * rs;
                ag[i] += d * xh[i]; ab[i] += d;
                g[i] = d * gm[i];
                s1 += g[i]; s2 += g[i] * xh[i];
            }
        }
        s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < LN_MAXI; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                float v = rs * (g[i] - s1 - xh[i] * s2);
                if (dres) v += dres[(size_t)r * D + c];
                st_f<TDX>(dx, (size_t)r * D + c, v);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAXI; ++i) { red[0][wv][lane + 64 * i] = ag[i]; red[1][wv][lane + 64 * i] = ab[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (ws) {   // per-block partials, folded by layernorm_bwd_finish_kernel (no contended atomics)
            ws[(size_t)blockIdx.x * D + c] = a;
            ws[((size_t)gridDim.x + blockIdx.x) * D + c] = b;
        } else { atomicAdd(dgamma + c, a); atomicAdd(dbeta + c, b); }
    }
}
__global__ void layernorm_bwd_finish_kernel(const float* __restrict__ ws, int nblk, int D, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;      // column in [0, 2D)
    if (c >= 2 * D) return;
    const int which = c / D, cc = c % D;
    const int per = (nblk + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(nblk, r0 + per);
    // four loads in flight (a single running sum pays one load latency per row: 25 rows took 9 us)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += ws[((size_t)which * nblk + r) * D + cc]; s1 += ws[((size_t)which * nblk + r + 1) * D + cc];
        s2 += ws[((size_t)which * nblk + r + 2) * D + cc]; s3 += ws[((size_t)which * nblk + r + 3) * D + cc];
    }
    for (; r < r1; ++r) s0 += ws[((size_t)which * nblk + r) * D + cc];
    atomicAdd((which ? dbeta : dgamma) + cc, (s0 + s1) + (s2 + s3));
}
extern "C" int vpf_layernorm_bwd(const void* dy_h16, const void* x, int x_is_h16, const float* mean, const float* rstd,
                                 const float* gamma, const float* dres, void* dx, int dx_is_h16, float* dgamma, float* dbeta,
                                 float* ws, long ws_floats, long rows, int D, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!dy_h16 || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta) return VPF_ERR_NULL;
    if (rows < 0 || D <= 0 || D > 64 * LN_MAXI) return VPF_ERR_BADSHAPE;
    if (rows == 0) return VPF_OK;
    int grid = grid_for(rows, rows >= 65536 ? 16 : 4, 1024);      // a wave per row for the encoder-sized inputs (4 k .. 12 k rows): the row loop is a latency chain
    hipStream_t st = (hipStream_t)stream;
    // workspace of 2 * grid * D floats -> per-block partial sums + a finishing pass; without it: fp32 atomics
    float* wsp = (ws && ws_floats >= 2L * grid * D && grid > 8) ? ws : nullptr;
    if (!wsp && grid > 64) grid = 64;            // atomic fallback: keep the contention on the 2D addresses low
#define LNB(TX, TDX) hipLaunchKernelGGL((layernorm_bwd_kernel<TX, TDX>), dim3(grid), dim3(256), 0, st, (const h16_t*)dy_h16, (const TX*)x, \
                                        mean, rstd, gamma, dres, (TDX*)dx, dgamma, dbeta, wsp, rows, D)
    if (x_is_h16 && dx_is_h16) LNB(h16_t, h16_t);
    else if (x_is_h16) LNB(h16_t, float);
    else if (dx_is_h16) LNB(float, h16_t);
    else LNB(float, float);
#undef LNB
    if (wsp) hipLaunchKernelGGL(layernorm_bwd_finish_kernel, dim3(vpf_cdiv(2 * D, 256), 32), dim3(256), 0, st, wsp, grid, D, dgamma, dbeta);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== dropout (+ residual)
__global__ void dropout_add_fwd_kernel(const h16_t* __restrict__ y, const float* __restrict__ res, float* __restrict__ out,
                                       long n, const uint32_t* __restrict__ rng_state, uint32_t site, float p)
{
    const VpfRng rng = vpf_rng_init(rng_state, site, p);
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g * 4 < n; g += (long)gridDim.x * blockDim.x) {
        const uint32_t keep = vpf_keep4(rng, (uint64_t)g);
        for (int e = 0; e < 4 && g * 4 + e < n; ++e) {
            const long i = g * 4 + e;
            const float v = h16_to_f32(y[i]);
            out[i] = (res ? res[i] : 0.f) + (((keep >> e) & 1u) ? v * rng.scale : 0.f);
        }
    }
}
extern "C" int vpf_dropout_add_fwd(const void* y_h16, const float* res, float* out, long n, const uint32_t* rng_state,
                                   uint32_t site, float p, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!y_h16 || !out || !rng_state) return VPF_ERR_NULL;
    if (n <= 0) return n == 0 ? VPF_OK : VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(dropout_add_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)y_h16, res, out, n,
                       rng_state, site, p);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// dy(h16) = keep ? dout * scale : 0
__global__ void dropout_bwd_kernel(const float* __restrict__ dout, h16_t* __restrict__ dy, long n,
                                   const uint32_t* __restrict__ rng_state, uint32_t site, float p)
{
    const VpfRng rng = vpf_rng_init(rng_state, site, p);
    const bool vec = (n % 4 == 0) && (((uintptr_t)dout & 15) == 0) && (((uintptr_t)dy & 7) == 0);
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g * 4 < n; g += (long)gridDim.x * blockDim.x) {
        const uint32_t keep = vpf_keep4(rng, (uint64_t)g);
        if (vec) {
            const float4 d = *reinterpret_cast<const float4*>(dout + g * 4);
            uint2 w;
            w.x = pack_h16x2((keep & 1u) ? d.x * rng.scale : 0.f, (keep & 2u) ? d.y * rng.scale : 0.f);
            w.y = pack_h16x2((keep & 4u) ? d.z * rng.scale : 0.f, (keep & 8u) ? d.w * rng.scale : 0.f);
            *reinterpret_cast<uint2*>(dy + g * 4) = w;
        } else {
            for (int e = 0; e < 4 && g * 4 + e < n; ++e)
                dy[g * 4 + e] = f32_to_h16(((keep >> e) & 1u) ? dout[g * 4 + e] * rng.scale : 0.f);
        }
    }
}
extern "C" int vpf_dropout_bwd(const float* dout, void* dy_h16, long n, const uint32_t* rng_state, uint32_t site, float p,
                               void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!dout || !dy_h16 || !rng_state) return VPF_ERR_NULL;
    if (n <= 0) return n == 0 ? VPF_OK : VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dout, (h16_t*)dy_h16, n, rng_state, site, p);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// keep-mask export (tests feed it to the oracle): out[i] = 1 if element i of `site` is kept
__global__ void dropout_mask_kernel(uint8_t* __restrict__ out, long n, const uint32_t* __restrict__ rng_state, uint32_t site, float p)
{
    const VpfRng rng = vpf_rng_init(rng_state, site, p);
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g * 4 < n; g += (long)gridDim.x * blockDim.x) {
        const uint32_t keep = vpf_keep4(rng, (uint64_t)g);
        for (int e = 0; e < 4 && g * 4 + e < n; ++e) out[g * 4 + e] = (keep >> e) & 1u;
    }
}
extern "C" int vpf_dropout_mask(uint8_t* out, long n, const uint32_t* rng_state, uint32_t site, float p, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!out || !rng_state) return VPF_ERR_NULL;
    if (n <= 0) return n == 0 ? VPF_OK : VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, out, n, rng_state, site, p);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// Timeline diagnostics: one lane writes the device's constant-rate clock.  Captured into the step's hipGraph at branch boundaries it
// shows when the REPLAYED graph really runs each branch (a kernel trace serialises differently from the unprofiled replay).
__global__ void stamp_kernel(unsigned long long* slot) { if (threadIdx.x == 0 && blockIdx.x == 0) *slot = wall_clock64(); }
extern "C" int vpf_stamp(unsigned long long* slots, int slot, void* stream)
{
    (void)hipGetLastError();
    if (!slots) return VPF_ERR_NULL;
    if (slot < 0) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slots + slot);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
extern "C" int vpf_wall_clock_khz(void)
{
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
    return khz;
}
__global__ void rng_advance_kernel(uint32_t* st) { if (threadIdx.x == 0 && blockIdx.x == 0) st[2] += 1u; }
extern "C" int vpf_rng_advance(uint32_t* rng_state, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!rng_state) return VPF_ERR_NULL;
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rng_state);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== column sums (bias grads, BN stats)
// x [M,C] -> acc[c] += sum_m x ; acc2[c] += sum_m x^2 (optional).  A block owns a slab of rows; every thread
// streams 8 consecutive columns (16-byte loads for h16) of one row per step; the block's row-lanes meet in
// LDS and ONE fp32 atomic per column and block leaves the CU.
// small M: block = 64 columns (8 groups of 8) x 32 row lanes; fixed-order fold
template <typename T>
__global__ void __launch_bounds__(256) colsum_small_kernel(const T* __restrict__ x, long M, int C, float* __restrict__ acc, float* __restrict__ acc2)
{
    __shared__ float red[2][32][65];
    const int t = threadIdx.x, g = t & 7, rl = t >> 3;
    const int c0 = blockIdx.x * 64 + g * 8;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    for (long r = rl; r < M; r += 32) {
        float v[8];
        if (sizeof(T) == 2) {
            const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const h16_t*>(x) + (size_t)r * C + c0);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = h16_lo(w[j]); v[2 * j + 1] = h16_hi(w[j]); }
        } else {
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + (size_t)r * C + c0);
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + (size_t)r * C + c0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][rl][g * 8 + j] = s[j]; red[1][rl][g * 8 + j] = q[j]; }
    __syncthreads();
    if (t < 128) {
        const int which = t >> 6, c = t & 63;
        float a = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) a += red[which][k][c];
        if (which == 0) acc[blockIdx.x * 64 + c] += a;
        else if (acc2) acc2[blockIdx.x * 64 + c] += a;
    }
}
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ x, long M, int C, float* __restrict__ acc, float* __restrict__ acc2,
                                                   int rows_per_block)
{
    __shared__ float red[2][256][9];
    const int c8 = C / 8;                       // column groups (C % 8 == 0 on this path)
    const int gpr = c8 < 256 ? c8 : 256;        // groups handled per row by the block at a time
    const int rlanes = 256 / gpr;               // rows in flight
    const int t = threadIdx.x, cg0 = t % gpr, rl = t / gpr;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int cgb = 0; cgb < c8; cgb += gpr) {
        const int cg = cgb + cg0;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
        if (cg < c8 && rl < rlanes) {
            for (long r = r0 + rl; r < r1; r += rlanes) {
                float v[8];
                if (sizeof(T) == 2) {
                    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const h16_t*>(x) + (size_t)r * C + cg * 8);
                    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[2 * j] = h16_lo(w[j]); v[2 * j + 1] = h16_hi(w[j]); }
                } else {
                    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + (size_t)r * C + cg * 8);
                    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + (size_t)r * C + cg * 8 + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[0][t][j] = s[j]; red[1][t][j] = q[j]; }
        __syncthreads();
        // thread (cg0, rl==0 .. ) : fold the row lanes; spread the 8 columns over the row-lane threads
        for (int e = t; e < gpr * 8; e += 256) {
            const int g = e / 8, j = e % 8;
            if (cgb + g < c8) {
                float a = 0.f, b = 0.f;
                for (int k = 0; k < rlanes; ++k) { a += red[0][k * gpr + g][j]; b += red[1][k * gpr + g][j]; }
                atomicAdd(acc + (cgb + g) * 8 + j, a);
                if (acc2) atomicAdd(acc2 + (cgb + g) * 8 + j, b);
            }
        }
        __syncthreads();
    }
}
// generic fallback (C % 8 != 0): thread = column
template <typename T>
__global__ void colsum_generic_kernel(const T* __restrict__ x, long M, int C, float* __restrict__ acc, float* __restrict__ acc2, int rows_per_block)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f, q = 0.f;
    for (long r = r0; r < r1; ++r) { const float v = ld_f<T>(x, (size_t)r * C + c); s += v; q += v * v; }
    atomicAdd(acc + c, s);
    if (acc2) atomicAdd(acc2 + c, q);
}
extern "C" int vpf_colsum(const void* x, int x_is_h16, long M, int C, float* acc, float* acc2, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !acc) return VPF_ERR_NULL;
    if (M < 0 || C <= 0) return VPF_ERR_BADSHAPE;
    if (M == 0) return VPF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (C % 8 == 0 && (((uintptr_t)x & 15) == 0)) {
        int rpb = 32;
        if (M <= 2048 && C % 64 == 0) {
            // a batch of samples (BatchNorm of the projection heads): every column is summed by ONE block in a fixed order
            // (deterministic), but the columns are spread over C/64 blocks and the rows over 32 lanes, so that no thread walks
            // more than M/32 rows
            if (x_is_h16) hipLaunchKernelGGL(colsum_small_kernel<h16_t>, dim3(C / 64), dim3(256), 0, st, (const h16_t*)x, M, C, acc, acc2);
            else hipLaunchKernelGGL(colsum_small_kernel<float>, dim3(C / 64), dim3(256), 0, st, (const float*)x, M, C, acc, acc2);
            VPF_CHECK_LAUNCH();
            return VPF_OK;
        }
        if (M <= 2048) rpb = (int)M;                        // one block: a deterministic sum
        while ((M + rpb - 1) / rpb > 512) rpb *= 2;      // <= 512 blocks -> <= 512 atomics per column
        const int grid = (int)((M + rpb - 1) / rpb);
        if (x_is_h16) hipLaunchKernelGGL(colsum_kernel<h16_t>, dim3(grid), dim3(256), 0, st, (const h16_t*)x, M, C, acc, acc2, rpb);
        else hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, M, C, acc, acc2, rpb);
    } else {
        int rpb = 128;
        while ((M + rpb - 1) / rpb > 16384) rpb *= 2;
        dim3 grid(vpf_cdiv(C, 256), (unsigned)((M + rpb - 1) / rpb));
        if (x_is_h16) hipLaunchKernelGGL(colsum_generic_kernel<h16_t>, grid, dim3(256), 0, st, (const h16_t*)x, M, C, acc, acc2, rpb);
        else hipLaunchKernelGGL(colsum_generic_kernel<float>, grid, dim3(256), 0, st, (const float*)x, M, C, acc, acc2, rpb);
    }
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// out[c] = sum_r x[r, c] in a FIXED order (r ascending): deterministic fold of per-workgroup partial statistics
__global__ void __launch_bounds__(1024) sum_rows_kernel(const float* __restrict__ x, int R, int C, float* __restrict__ out)
{
    // 64 columns x 16 row groups per block; every thread keeps 4 independent partial sums so that its loads are in flight
    // together (a single running sum serialises R load latencies); the fold order is fixed, so the result is deterministic
    __shared__ float fold[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int r = rg;
        for (; r + 48 < R; r += 64) {
            s0 += x[(size_t)r * C + c]; s1 += x[(size_t)(r + 16) * C + c];
            s2 += x[(size_t)(r + 32) * C + c]; s3 += x[(size_t)(r + 48) * C + c];
        }
        for (; r < R; r += 16) s0 += x[(size_t)r * C + c];
    }
    fold[rg][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += fold[k][threadIdx.x];
        out[c] = t;
    }
}
extern "C" int vpf_sum_rows_f32(const float* x, int R, int C, float* out, void* stream)
{
    (void)hipGetLastError();
    if (!x || !out) return VPF_ERR_NULL;
    if (R <= 0 || C <= 0) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(sum_rows_kernel, dim3(vpf_cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, x, R, C, out);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== BatchNorm (channels-last [M,C])
// stat layout: [mean(C) | rstd(C)].  training: from batch sums, updates running stats
// (momentum 0.1, unbiased variance) like nn.BatchNorm1d; eval: from running stats.
__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ sumsq, long M, int C, float eps, float momentum,
                                   int training, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ num_batches, float* __restrict__ stat)
{
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        if (training) {
            const float mu = sums[c] / (float)M;
            float var = sumsq[c] / (float)M - mu * mu;
            var = var < 0.f ? 0.f : var;
            stat[c] = mu; stat[C + c] = rsqrtf(var + eps);
            if (running_mean) {
                const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
            }
        } else {
            stat[c] = running_mean[c]; stat[C + c] = rsqrtf(running_var[c] + eps);
        }
    }
    if (training && num_batches && blockIdx.x == 0 && threadIdx.x == 0) *num_batches += 1;
}
extern "C" int vpf_bn_finalize(const float* sums, const float* sumsq, long M, int C, float eps, float momentum, int training,
                               float* running_mean, float* running_var, long long* num_batches, float* stat, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!stat || (training && (!sums || !sumsq)) || (!training && (!running_mean || !running_var))) return VPF_ERR_NULL;
    if (C <= 0 || M <= 0) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(vpf_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, sumsq, M, C, eps, momentum,
                       training, running_mean, running_var, num_batches, stat);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// fold BatchNorm into y = a[c]*x + b[c]:  a = rstd*gamma, b = beta - mean*rstd*gamma   (ab = [a(C) | b(C)])
__global__ void bn_affine_kernel(const float* __restrict__ stat, const float* __restrict__ gamma, const float* __restrict__ beta, int C, float* __restrict__ ab)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { const float a = stat[C + c] * gamma[c]; ab[c] = a; ab[C + c] = beta[c] - stat[c] * a; }
}
extern "C" int vpf_bn_affine(const float* stat, const float* gamma, const float* beta, int C, float* ab, void* stream)
{
    (void)hipGetLastError();
    if (!stat || !gamma || !beta || !ab) return VPF_ERR_NULL;
    if (C <= 0) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(bn_affine_kernel, dim3(vpf_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, stat, gamma, beta, C, ab);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

template <typename TIN, typename TOUT>
__global__ void bn_act_fwd_kernel(const TIN* __restrict__ x, const float* __restrict__ stat, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, TOUT* __restrict__ y, long total, int C, int relu)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = (ld_f<TIN>(x, i) - stat[c]) * stat[C + c] * gamma[c] + beta[c];
        if (relu == 1) v = fmaxf(v, 0.f); else if (relu == 2) v = v > 0.f ? v : 0.2f * v;   // 2: LeakyReLU(0.2), partseg.py:393
        st_f<TOUT>(y, i, v);
    }
}
extern "C" int vpf_bn_act_fwd(const void* x, int x_is_h16, const float* stat, const float* gamma, const float* beta, void* y,
                              int y_is_h16, long M, int C, int relu, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !stat || !gamma || !beta || !y) return VPF_ERR_NULL;
    if (M < 0 || C <= 0) return VPF_ERR_BADSHAPE;
    const long total = M * C;
    if (total == 0) return VPF_OK;
    const int grid = grid_for(total, 256);
    hipStream_t st = (hipStream_t)stream;
#define BNF(TI, TO) hipLaunchKernelGGL((bn_act_fwd_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const TI*)x, stat, gamma, beta, (TO*)y, total, C, relu)
    if (x_is_h16 && y_is_h16) BNF(h16_t, h16_t);
    else if (x_is_h16) BNF(h16_t, float);
    else if (y_is_h16) BNF(float, h16_t);
    else BNF(float, float);
#undef BNF
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// backward pass 1: tmp[c] += sum g ; tmp[C+c] += sum g*xhat   with g = dy * relu'(y)
template <typename TX, typename TDY>
__global__ void bn_bwd_reduce_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ stat,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, long M, int C, int relu,
                                     float* __restrict__ tmp, int rows_per_block)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mu = stat[c], rs = stat[C + c], ga = gamma[c], be = beta[c];
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f, q = 0.f;
    long r = r0;
    for (; r + 4 <= r1; r += 4) {          // 4 rows in flight: a single running pair would serialise the load latencies
        float xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { xv[u] = ld_f<TX>(x, (size_t)(r + u) * C + c); gv[u] = ld_f<TDY>(dy, (size_t)(r + u) * C + c); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xh = (xv[u] - mu) * rs;
            float g = gv[u];
            if (relu && (xh * ga + be) <= 0.f) g = relu == 2 ? 0.2f * g : 0.f;
            s += g; q += g * xh;
        }
    }
    for (; r < r1; ++r) {
        const float xh = (ld_f<TX>(x, (size_t)r * C + c) - mu) * rs;
        float g = ld_f<TDY>(dy, (size_t)r * C + c);
        if (relu && (xh * ga + be) <= 0.f) g = relu == 2 ? 0.2f * g : 0.f;
        s += g; q += g * xh;
    }
    atomicAdd(tmp + c, s); atomicAdd(tmp + C + c, q);
}
// backward pass 2: dx = gamma*rstd*(g - sum_g/M - xhat*sum_gxh/M); block 0 also does dgamma += sum_gxh, dbeta += sum_g
template <typename TX, typename TDY, typename TDX>
__global__ void bn_bwd_apply_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ stat,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ tmp,
                                    long M, int C, int relu, int training, TDX* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const long total = M * C;
    const float invM = 1.f / (float)M;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float rs = stat[C + c], ga = gamma[c];
        const float xh = (ld_f<TX>(x, i) - stat[c]) * rs;
        float g = ld_f<TDY>(dy, i);
        if (relu && (xh * ga + beta[c]) <= 0.f) g = relu == 2 ? 0.2f * g : 0.f;
        const float v = training ? ga * rs * (g - tmp[c] * invM - xh * tmp[C + c] * invM) : ga * rs * g;
        if (dx) st_f<TDX>(dx, i, v);
    }
    if (blockIdx.x == 0 && dgamma)
        for (int c = threadIdx.x; c < C; c += blockDim.x) { atomicAdd(dgamma + c, tmp[C + c]); atomicAdd(dbeta + c, tmp[c]); }
}
extern "C" int vpf_bn_bwd(const void* dy, int dy_is_h16, const void* x, int x_is_h16, const float* stat, const float* gamma,
                          const float* beta, long M, int C, int relu, int training, float* tmp2C_zeroed, void* dx, int dx_is_h16,
                          float* dgamma, float* dbeta, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!dy || !x || !stat || !gamma || !beta || !tmp2C_zeroed) return VPF_ERR_NULL;
    if (M <= 0 || C <= 0) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    int rpb = (int)((M + 63) / 64);          // ~64 row slabs: small batches (the projection heads: 64 .. 128 rows) must not be one serial loop
    if (rpb < 8) rpb = 8;
    if (rpb > 128) rpb = 128;
    while ((M + rpb - 1) / rpb > 16384) rpb *= 2;
    dim3 g1(vpf_cdiv(C, 256), (unsigned)((M + rpb - 1) / rpb));
    const int g2 = grid_for(M * C, 256);
#define BNR(TX, TDY) hipLaunchKernelGGL((bn_bwd_reduce_kernel<TX, TDY>), g1, dim3(256), 0, st, (const TDY*)dy, (const TX*)x, stat, gamma, beta, M, C, relu, tmp2C_zeroed, rpb)
#define BNA(TX, TDY, TDX) hipLaunchKernelGGL((bn_bwd_apply_kernel<TX, TDY, TDX>), dim3(g2), dim3(256), 0, st, (const TDY*)dy, (const TX*)x, stat, gamma, beta, tmp2C_zeroed, M, C, relu, training, (TDX*)dx, dgamma, dbeta)
    if (x_is_h16 && dy_is_h16) { BNR(h16_t, h16_t); if (dx_is_h16) BNA(h16_t, h16_t, h16_t); else BNA(h16_t, h16_t, float); }
    else if (x_is_h16 && !dy_is_h16) { BNR(h16_t, float); if (dx_is_h16) BNA(h16_t, float, h16_t); else BNA(h16_t, float, float); }
    else if (!x_is_h16 && dy_is_h16) { BNR(float, h16_t); if (dx_is_h16) BNA(float, h16_t, h16_t); else BNA(float, h16_t, float); }
    else { BNR(float, float); if (dx_is_h16) BNA(float, float, h16_t); else BNA(float, float, float); }
#undef BNR
#undef BNA
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== max over group members
// h [NG, K, C] h16 -> out [NG, C] (f32 or h16), arg [NG, C] (uint8, first max)
template <typename TOUT>
__global__ void group_max_fwd_kernel(const h16_t* __restrict__ h, long NG, int K, int C, TOUT* __restrict__ out, uint8_t* __restrict__ arg)
{
    const long total = NG * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i / C; const int c = (int)(i % C);
        float best = -INFINITY; int bi = 0;
        for (int k = 0; k < K; ++k) { const float v = h16_to_f32(h[((size_t)g * K + k) * C + c]); if (v > best) { best = v; bi = k; } }
        st_f<TOUT>(out, i, best);
        if (arg) arg[i] = (uint8_t)bi;
    }
}
extern "C" int vpf_group_max_fwd(const void* h_h16, long NG, int K, int C, void* out, int out_is_h16, uint8_t* arg, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!h_h16 || !out) return VPF_ERR_NULL;
    if (NG < 0 || K <= 0 || K > 255 || C <= 0) return VPF_ERR_BADSHAPE;
    if (NG == 0) return VPF_OK;
    const int grid = grid_for(NG * C, 256);
    if (out_is_h16) hipLaunchKernelGGL(group_max_fwd_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const h16_t*)h_h16, NG, K, C, (h16_t*)out, arg);
    else hipLaunchKernelGGL(group_max_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const h16_t*)h_h16, NG, K, C, (float*)out, arg);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// dh [NG,K,C] h16 = (k == arg) ? dout : 0
template <typename TIN>
__global__ void group_max_bwd_kernel(const TIN* __restrict__ dout, const uint8_t* __restrict__ arg, long NG, int K, int C, h16_t* __restrict__ dh)
{
    const long total = NG * K * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long gk = i / C; const int k = (int)(gk % K); const long g = gk / K;
        dh[i] = (arg[g * C + c] == k) ? f32_to_h16(ld_f<TIN>(dout, (size_t)g * C + c)) : (h16_t)0;
    }
}
extern "C" int vpf_group_max_bwd(const void* dout, int dout_is_h16, const uint8_t* arg, long NG, int K, int C, void* dh_h16, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!dout || !arg || !dh_h16) return VPF_ERR_NULL;
    if (NG < 0 || K <= 0 || C <= 0) return VPF_ERR_BADSHAPE;
    if (NG == 0) return VPF_OK;
    const int grid = grid_for(NG * K * C, 256);
    if (dout_is_h16) hipLaunchKernelGGL(group_max_bwd_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const h16_t*)dout, arg, NG, K, C, (h16_t*)dh_h16);
    else hipLaunchKernelGGL(group_max_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dout, arg, NG, K, C, (h16_t*)dh_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// Group2Emb concat (utils.py:183): feat[m, 0:C] = gmax[m / K, :], feat[m, C:2C] = h[m, :]
__global__ void g2e_concat_fwd_kernel(const h16_t* __restrict__ gmax, const h16_t* __restrict__ h, long M, int K, int C, h16_t* __restrict__ feat)
{
    const long total = M * 2 * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (2 * C)); const long m = i / (2 * C);
        feat[i] = c < C ? gmax[(m / K) * C + c] : h[m * C + (c - C)];
    }
}
extern "C" int vpf_g2e_concat_fwd(const void* gmax_h16, const void* h_h16, long M, int K, int C, void* feat_h16, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!gmax_h16 || !h_h16 || !feat_h16) return VPF_ERR_NULL;
    if (M < 0 || K <= 0 || C <= 0 || (M % K)) return VPF_ERR_BADSHAPE;
    if (M == 0) return VPF_OK;
    hipLaunchKernelGGL(g2e_concat_fwd_kernel, dim3(grid_for(M * 2 * C, 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)gmax_h16,
                       (const h16_t*)h_h16, M, K, C, (h16_t*)feat_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// dh[m,c] = dfeat[m, C+c] + (k == arg[g,c] ? sum_k' dfeat[(g,k'), c] : 0)
__global__ void g2e_concat_bwd_kernel(const h16_t* __restrict__ dfeat, const uint8_t* __restrict__ arg, long NG, int K, int C, h16_t* __restrict__ dh)
{
    const long total = NG * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i / C; const int c = (int)(i % C);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += h16_to_f32(dfeat[((size_t)g * K + k) * 2 * C + c]);
        const int a = arg[i];
        for (int k = 0; k < K; ++k) {
            const float v = h16_to_f32(dfeat[((size_t)g * K + k) * 2 * C + C + c]) + (k == a ? s : 0.f);
            dh[((size_t)g * K + k) * C + c] = f32_to_h16(v);
        }
    }
}
extern "C" int vpf_g2e_concat_bwd(const void* dfeat_h16, const uint8_t* arg, long NG, int K, int C, void* dh_h16, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!dfeat_h16 || !arg || !dh_h16) return VPF_ERR_NULL;
    if (NG < 0 || K <= 0 || C <= 0) return VPF_ERR_BADSHAPE;
    if (NG == 0) return VPF_OK;
    hipLaunchKernelGGL(g2e_concat_bwd_kernel, dim3(grid_for(NG * C, 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)dfeat_h16, arg, NG, K, C, (h16_t*)dh_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// sum over the K group members: x h16 [NG,K,C] -> out f32 [NG,C]   (gradient of the broadcast "global" feature, utils.py:183)
__global__ void group_sum_kernel(const h16_t* __restrict__ x, long NG, int K, int C, float* __restrict__ out)
{
    const long total = NG * (C / 2);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i / (C / 2); const int c2 = (int)(i % (C / 2));
        float s0 = 0.f, s1 = 0.f;
        for (int k = 0; k < K; ++k) {
            const uint32_t u = *reinterpret_cast<const uint32_t*>(x + ((size_t)g * K + k) * C + 2 * c2);
            s0 += h16_lo(u); s1 += h16_hi(u);
        }
        out[g * C + 2 * c2] = s0; out[g * C + 2 * c2 + 1] = s1;
    }
}
extern "C" int vpf_group_sum(const void* x_h16, long NG, int K, int C, float* out, void* stream)
{
    (void)hipGetLastError();
    if (!x_h16 || !out) return VPF_ERR_NULL;
    if (NG < 0 || K <= 0 || C <= 0 || (C & 1)) return VPF_ERR_BADSHAPE;
    if (NG == 0) return VPF_OK;
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid_for(NG * (C / 2), 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x_h16, NG, K, C, out);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// dh[g, arg[g,c], c] += dg[g,c]   (max-pool backward added onto an existing gradient; one writer per element)
__global__ void group_max_scatter_add_kernel(const h16_t* __restrict__ dg, const uint8_t* __restrict__ arg, long NG, int K, int C, h16_t* __restrict__ dh)
{
    const long total = NG * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i / C; const int c = (int)(i % C);
        const size_t o = ((size_t)g * K + arg[i]) * C + c;
        dh[o] = f32_to_h16(h16_to_f32(dh[o]) + h16_to_f32(dg[i]));
    }
}
extern "C" int vpf_group_max_scatter_add(const void* dg_h16, const uint8_t* arg, long NG, int K, int C, void* dh_h16, void* stream)
{
    (void)hipGetLastError();
    if (!dg_h16 || !arg || !dh_h16) return VPF_ERR_NULL;
    if (NG < 0 || K <= 0 || C <= 0) return VPF_ERR_BADSHAPE;
    if (NG == 0) return VPF_OK;
    hipLaunchKernelGGL(group_max_scatter_add_kernel, dim3(grid_for(NG * C, 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)dg_h16, arg, NG, K, C, (h16_t*)dh_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== token pooling  partseg.py:547
// x f32 [B,L,D] -> out f32 [B,2D] = [max over L | mean over L], arg int32 [B,D]
__global__ void __launch_bounds__(512) pool_fwd_kernel(const float* __restrict__ x, int B, int L, int D, float* __restrict__ out, int* __restrict__ arg)
{
    // block = (sample, 64 channels); 8 row groups walk the tokens with 4 loads in flight each; first maximum wins ties
    __shared__ float sm[8][64], ss[8][64];
    __shared__ int si[8][64];
    const int b = blockIdx.y, d = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float best = -INFINITY, s = 0.f; int bi = 0;
    if (d < D) {
        const float* xp = x + (size_t)b * L * D + d;
        int l = rg;
        for (; l + 24 < L; l += 32) {
            const float v0 = xp[(size_t)l * D], v1 = xp[(size_t)(l + 8) * D], v2 = xp[(size_t)(l + 16) * D], v3 = xp[(size_t)(l + 24) * D];
            s += (v0 + v1) + (v2 + v3);
            if (v0 > best) { best = v0; bi = l; }
            if (v1 > best) { best = v1; bi = l + 8; }
            if (v2 > best) { best = v2; bi = l + 16; }
            if (v3 > best) { best = v3; bi = l + 24; }
        }
        for (; l < L; l += 8) { const float v = xp[(size_t)l * D]; s += v; if (v > best) { best = v; bi = l; } }
    }
    sm[rg][threadIdx.x & 63] = best; ss[rg][threadIdx.x & 63] = s; si[rg][threadIdx.x & 63] = bi;
    __syncthreads();
    if (rg == 0 && d < D) {
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float v = sm[k][threadIdx.x]; const int vi = si[k][threadIdx.x];
            s += ss[k][threadIdx.x];
            if (v > best || (v == best && vi < bi)) { best = v; bi = vi; }
        }
        out[(size_t)b * 2 * D + d] = best; out[(size_t)b * 2 * D + D + d] = s / (float)L;
        if (arg) arg[(size_t)b * D + d] = bi;
    }
}
extern "C" int vpf_pool_fwd(const float* x, int B, int L, int D, float* out, int* arg, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !out) return VPF_ERR_NULL;
    if (B < 0 || L <= 0 || D <= 0) return VPF_ERR_BADSHAPE;
    if (B == 0) return VPF_OK;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(vpf_cdiv(D, 64), B), dim3(512), 0, (hipStream_t)stream, x, B, L, D, out, arg);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
__global__ void pool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ arg, int B, int L, int D, float* __restrict__ dx)
{
    const long total = (long)B * L * D;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D); const long bl = i / D; const int l = (int)(bl % L); const int b = (int)(bl / L);
        float v = dout[(size_t)b * 2 * D + D + d] / (float)L;
        if (arg[(size_t)b * D + d] == l) v += dout[(size_t)b * 2 * D + d];
        dx[i] = v;
    }
}
// D % 4 == 0 and fewer than 2^31 elements: a thread writes four channels of one token row (16-byte stores, 32-bit index arithmetic):
// 10.8 -> 6.2 us at 128 x 96 x 256 (round 4)
__global__ void __launch_bounds__(256) pool_bwd_vec_kernel(const float* __restrict__ dout, const int* __restrict__ arg, int rows, int L, int D4,
                                                           float* __restrict__ dx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * D4) return;
    const int row = i / D4, c4 = i - row * D4, b = row / L, l = row - b * L, D = 4 * D4;
    const float4 mx = *reinterpret_cast<const float4*>(dout + (size_t)b * 2 * D + 4 * c4);
    const float4 mn = *reinterpret_cast<const float4*>(dout + (size_t)b * 2 * D + D + 4 * c4);
    const int4 a = *reinterpret_cast<const int4*>(arg + (size_t)b * D + 4 * c4);
    const float fl = (float)L;
    float4 v = make_float4(mn.x / fl, mn.y / fl, mn.z / fl, mn.w / fl);
    if (a.x == l) v.x += mx.x;
    if (a.y == l) v.y += mx.y;
    if (a.z == l) v.z += mx.z;
    if (a.w == l) v.w += mx.w;
    *reinterpret_cast<float4*>(dx + (size_t)i * 4) = v;
}
extern "C" int vpf_pool_bwd(const float* dout, const int* arg, int B, int L, int D, float* dx, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!dout || !arg || !dx) return VPF_ERR_NULL;
    if (B < 0 || L <= 0 || D <= 0) return VPF_ERR_BADSHAPE;
    if (B == 0) return VPF_OK;
    if (D % 4 == 0 && (long)B * L * D < (1l << 31) && !((uintptr_t)dout & 15) && !((uintptr_t)arg & 15) && !((uintptr_t)dx & 15))
        hipLaunchKernelGGL(pool_bwd_vec_kernel, dim3((unsigned)(((long)B * L * (D / 4) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, arg, B * L, L, D / 4, dx);
    else
        hipLaunchKernelGGL(pool_bwd_kernel, dim3(grid_for((long)B * L * D, 256)), dim3(256), 0, (hipStream_t)stream, dout, arg, B, L, D, dx);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== misc
// y += a * x   (fp32)
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float a)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}
extern "C" int vpf_axpy_f32(const float* x, float* y, long n, float a, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !y) return VPF_ERR_NULL;
    if (n <= 0) return n == 0 ? VPF_OK : VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// acc[r % period, :] += x[r, :]  (positional-embedding gradient: sum over the batch), fp32
__global__ void rowsum_mod_kernel(const float* __restrict__ x, long rows, int D, int period, float* __restrict__ acc)
{
    const long total = (long)period * D;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D); const long t = i / D;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        long r = t;
        for (; r + 3 * (long)period < rows; r += 4 * (long)period) {        // 4 loads in flight
            s0 += x[(size_t)r * D + d]; s1 += x[(size_t)(r + period) * D + d];
            s2 += x[(size_t)(r + 2 * (long)period) * D + d]; s3 += x[(size_t)(r + 3 * (long)period) * D + d];
        }
        for (; r < rows; r += period) s0 += x[(size_t)r * D + d];
        acc[i] += (s0 + s1) + (s2 + s3);
    }
}
extern "C" int vpf_rowsum_mod_f32(const float* x, long rows, int D, int period, float* acc, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !acc) return VPF_ERR_NULL;
    if (rows < 0 || D <= 0 || period <= 0) return VPF_ERR_BADSHAPE;
    if (rows == 0) return VPF_OK;
    hipLaunchKernelGGL(rowsum_mod_kernel, dim3(grid_for((long)period * D, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, D, period, acc);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== BatchNorm over a small batch, one kernel
// nn.BatchNorm1d (+ ReLU) of the projection heads (partseg.py:519-525) in training mode: the batch is 64 .. 256 rows, so a
// block owning 64 channels can do everything -- batch statistics, running-statistics update, normalisation + ReLU -- by
// itself (BatchNorm is per channel), instead of a zero-fill, a column-sum, a finalize and an apply launch.
// Block = 64 channels x 8 row lanes (512 threads), fixed-order folds (deterministic).
__global__ void __launch_bounds__(512) bn_small_fwd_kernel(const float* __restrict__ x, int M, int C, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          long long* __restrict__ num_batches, float* __restrict__ stat,
                                                          h16_t* __restrict__ y, int relu)
{
    __shared__ float fs[8][64], fq[8][64];
    __shared__ float smu[64], srs[64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    int r = rl;
#pragma unroll 4                                     // 8 independent loads in flight per thread (the batch is 64 .. 256 rows: 8 dependent round trips otherwise)
    for (; r + 8 < M; r += 16) {
        const float a = x[(size_t)r * C + c], b = x[(size_t)(r + 8) * C + c];
        s0 += a; q0 += a * a; s1 += b; q1 += b * b;
    }
    for (; r < M; r += 8) { const float a = x[(size_t)r * C + c]; s0 += a; q0 += a * a; }
    fs[rl][cl] = s0 + s1; fq[rl][cl] = q0 + q1;
    __syncthreads();
    if (rl == 0) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s += fs[k][cl]; q += fq[k][cl]; }
        const float mu = s / (float)M;
        float var = q / (float)M - mu * mu;
        var = var < 0.f ? 0.f : var;
        const float rs = rsqrtf(var + eps);
        smu[cl] = mu; srs[cl] = rs;
        stat[c] = mu; stat[C + c] = rs;
        if (running_mean) {
            const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
        }
        if (num_batches && blockIdx.x == 0 && cl == 0) *num_batches += 1;
    }
    __syncthreads();
    const float mu = smu[cl], rs = srs[cl], ga = gamma[c], be = beta[c];
#pragma unroll 8
    for (r = rl; r < M; r += 8) {
        float v = (x[(size_t)r * C + c] - mu) * rs * ga + be;
        if (relu == 1) v = fmaxf(v, 0.f); else if (relu == 2) v = v > 0.f ? v : 0.2f * v;   // 2: LeakyReLU(0.2), partseg.py:393
        y[(size_t)r * C + c] = f32_to_h16(v);
    }
}
extern "C" int vpf_bn_small_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var, long long* num_batches, float* stat, void* y_h16, int relu,
                                void* stream)
{
    (void)hipGetLastError();
    if (!x || !gamma || !beta || !stat || !y_h16) return VPF_ERR_NULL;
    if (M <= 0 || M > 4096 || C <= 0 || (C % 64)) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(C / 64), dim3(512), 0, (hipStream_t)stream, x, M, C, gamma, beta, eps, momentum, running_mean,
                       running_var, num_batches, stat, (h16_t*)y_h16, relu);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// backward of the same (training mode): g = dy * relu'(y); dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat));
// dgamma += sum(g * xhat), dbeta += sum(g)   (one block per 64 channels: the single writer of its columns)
template <typename TDX>
__global__ void __launch_bounds__(512) bn_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stat,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, int M, int C, int relu,
                                                          TDX* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    __shared__ float fs[8][64], fq[8][64];
    __shared__ float ssum[64], sqsum[64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const float mu = stat[c], rs = stat[C + c], ga = gamma[c], be = beta[c];
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    int r = rl;
#pragma unroll 4
    for (; r + 8 < M; r += 16) {
        const float xa = (x[(size_t)r * C + c] - mu) * rs, xb = (x[(size_t)(r + 8) * C + c] - mu) * rs;
        float ga_ = dy[(size_t)r * C + c], gb_ = dy[(size_t)(r + 8) * C + c];
        if (relu && (xa * ga + be) <= 0.f) ga_ = relu == 2 ? 0.2f * ga_ : 0.f;
        if (relu && (xb * ga + be) <= 0.f) gb_ = relu == 2 ? 0.2f * gb_ : 0.f;
        s0 += ga_; q0 += ga_ * xa; s1 += gb_; q1 += gb_ * xb;
    }
    for (; r < M; r += 8) {
        const float xa = (x[(size_t)r * C + c] - mu) * rs;
        float ga_ = dy[(size_t)r * C + c];
        if (relu && (xa * ga + be) <= 0.f) ga_ = relu == 2 ? 0.2f * ga_ : 0.f;
        s0 += ga_; q0 += ga_ * xa;
    }
    fs[rl][cl] = s0 + s1; fq[rl][cl] = q0 + q1;
    __syncthreads();
    if (rl == 0) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s += fs[k][cl]; q += fq[k][cl]; }
        ssum[cl] = s; sqsum[cl] = q;
        if (dgamma) { dgamma[c] += q; dbeta[c] += s; }
    }
    __syncthreads();
    if (!dx) return;
    const float sm = ssum[cl] / (float)M, qm = sqsum[cl] / (float)M;
#pragma unroll 8
    for (r = rl; r < M; r += 8) {
        const float xh = (x[(size_t)r * C + c] - mu) * rs;
        float g = dy[(size_t)r * C + c];
        if (relu && (xh * ga + be) <= 0.f) g = relu == 2 ? 0.2f * g : 0.f;
        st_f<TDX>(dx, (size_t)r * C + c, ga * rs * (g - sm - xh * qm));
    }
}
extern "C" int vpf_bn_small_bwd(const float* dy, const float* x, const float* stat, const float* gamma, const float* beta, int M, int C,
                                int relu, void* dx, int dx_is_h16, float* dgamma, float* dbeta, void* stream)
{
    (void)hipGetLastError();
    if (!dy || !x || !stat || !gamma || !beta) return VPF_ERR_NULL;
    if (M <= 0 || M > 4096 || C <= 0 || (C % 64)) return VPF_ERR_BADSHAPE;
    if (dx_is_h16) hipLaunchKernelGGL(bn_small_bwd_kernel<h16_t>, dim3(C / 64), dim3(512), 0, (hipStream_t)stream, dy, x, stat, gamma, beta, M, C, relu, (h16_t*)dx, dgamma, dbeta);
    else hipLaunchKernelGGL(bn_small_bwd_kernel<float>, dim3(C / 64), dim3(512), 0, (hipStream_t)stream, dy, x, stat, gamma, beta, M, C, relu, (float*)dx, dgamma, dbeta);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
