// small.hip -- the K=3 front-ends (per-point adapter MLP, centre position MLP, Group2Emb's first
// conv + BatchNorm), image patchify, NT-Xent and the fused AdamW step.
// Replaces: classifier.py:31-36 (PointCloudInputAdapter.point_mlp[0:3]); partseg.py:498-501
// (position_emb[0:2]); utils.py:153-157 (first_conv[0:3]); partseg.py:632 (Rearrange);
// lightly NTXentLoss (pretrain.py:155,196,202); torch.optim.AdamW (pretrain.py:121-124,210).
#include "vpf_common.h"

static inline int grid_for(long n, int per_block, int cap = 4096)
{
    long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x)
{
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    return cdf + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

// =============================================================================== adapter front: Linear(C,64) -> LN(64) -> ReLU
// wave per row (grid-stride), lane = channel.  out h16 [M,64] feeds the 64->D MFMA GEMM.
#define AD_MAXC 8
__global__ void __launch_bounds__(256) adapter_front_fwd_kernel(const float* __restrict__ x, long M, int C, const float* __restrict__ W,
                                                              const float* __restrict__ b, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, h16_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long wave0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    float w[AD_MAXC];
#pragma unroll
    for (int j = 0; j < AD_MAXC; ++j) w[j] = j < C ? W[lane * C + j] : 0.f;
    const float bb = b[lane], ga = gamma[lane], be = beta[lane];
    constexpr int U = 4;                 // rows in flight per wave (the loop is latency-bound otherwise)
    for (long r0 = wave0 * U; r0 < M; r0 += nw * U) {
        float h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + u < M ? r0 + u : M - 1;
            h[u] = bb;
#pragma unroll
            for (int j = 0; j < AD_MAXC; ++j) if (j < C) h[u] += w[j] * x[(size_t)r * C + j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float mu = wave_sum(h[u]) * (1.f / 64.f);
            const float d = h[u] - mu;
            const float rs = rsqrtf(wave_sum(d * d) * (1.f / 64.f) + 1e-5f);
            if (r0 + u < M) out[(size_t)(r0 + u) * 64 + lane] = f32_to_h16(fmaxf(d * rs * ga + be, 0.f));
        }
    }
}
extern "C" int vpf_adapter_front_fwd(const float* x, long M, int C, const float* W, const float* b, const float* gamma,
                                     const float* beta, void* out_h16, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !W || !b || !gamma || !beta || !out_h16) return VPF_ERR_NULL;
    if (M < 0 || C <= 0 || C > AD_MAXC) return VPF_ERR_BADSHAPE;
    if (M == 0) return VPF_OK;
    hipLaunchKernelGGL(adapter_front_fwd_kernel, dim3(grid_for(M, 64, 2048)), dim3(256), 0, (hipStream_t)stream, x, M, C, W, b, gamma, beta, (h16_t*)out_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// da (h16 [M,64]) -> dW[64,C] += , db[64] +=, dgamma[64] +=, dbeta[64] +=   (input needs no grad)
// CT = the compile-time bound of the input-channel loops: 3 for xyz clouds (the loops over AD_MAXC = 8 channels with a run-time C spent
// a fifth of the kernel's instructions on five channels that do not exist), AD_MAXC otherwise
template <int CT>
__global__ void __launch_bounds__(256) adapter_front_bwd_kernel(const float* __restrict__ x, const h16_t* __restrict__ da, long M, int C,
                                                              const float* __restrict__ W, const float* __restrict__ b,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ partial)
{
    const int lane = threadIdx.x & 63;
    const long wave0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    float w[CT], aw[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) { w[j] = j < C ? W[lane * C + j] : 0.f; aw[j] = 0.f; }
    const float bb = b[lane], ga = gamma[lane], be = beta[lane];
    float adb = 0.f, adg = 0.f, adbe = 0.f;
    constexpr int U = 4;
    for (long r0 = wave0 * U; r0 < M; r0 += nw * U) {
        float xv[U][CT], h[U], gin[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = r0 + u < M;
            const long r = ok ? r0 + u : M - 1;
            h[u] = bb;
#pragma unroll
            for (int j = 0; j < CT; ++j) { xv[u][j] = ((CT < AD_MAXC || j < C) && ok) ? x[(size_t)r * C + j] : 0.f; h[u] += w[j] * xv[u][j]; }
            gin[u] = ok ? h16_to_f32(da[(size_t)r * 64 + lane]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float mu = wave_sum(h[u]) * (1.f / 64.f);
            const float d = h[u] - mu;
            const float rs = rsqrtf(wave_sum(d * d) * (1.f / 64.f) + 1e-5f);
            const float xh = d * rs;
            float g = gin[u];
            if (xh * ga + be <= 0.f) g = 0.f;
            adg += g * xh; adbe += g;
            const float gx = g * ga;
            const float m1 = wave_sum(gx) * (1.f / 64.f), m2 = wave_sum(gx * xh) * (1.f / 64.f);
            const float dh = (r0 + u < M) ? rs * (gx - m1 - xh * m2) : 0.f;
            adb += dh;
#pragma unroll
            for (int j = 0; j < CT; ++j) aw[j] += dh * xv[u][j];
        }
    }
    // Thousands of waves adding into the same ~400 addresses serialise in L2 (this tail used to be most of the kernel):
    // fold the 4 waves of the block in LDS and leave ONE partial row per block for adapter_front_fold_kernel.
    __shared__ float fold[4][64 * (3 + AD_MAXC)];
    const int wv = threadIdx.x >> 6;
    fold[wv][lane] = adb; fold[wv][64 + lane] = adg; fold[wv][128 + lane] = adbe;
#pragma unroll
    for (int j = 0; j < AD_MAXC; ++j) fold[wv][192 + j * 64 + lane] = j < CT ? aw[j < CT ? j : 0] : 0.f;
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * (3 + AD_MAXC); e += 256)
        partial[(size_t)blockIdx.x * 64 * (3 + AD_MAXC) + e] = (fold[0][e] + fold[1][e]) + (fold[2][e] + fold[3][e]);
}
// db | dgamma | dbeta | dW[:, j] += sum over blocks of the partial rows (fixed order)
__global__ void __launch_bounds__(1024) adapter_front_fold_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ dW,
                                                                 float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    __shared__ float fold[16][64];
    constexpr int ROW = 64 * (3 + AD_MAXC);
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;       // column of the partial row
    // gridDim.y row slices (a handful of atomics per output instead of one block per column walking every row)
    const int per = (nblk + gridDim.y - 1) / gridDim.y, rbeg = blockIdx.y * per, rend = min(nblk, rbeg + per);
    float s0 = 0.f, s1 = 0.f;
    int r = rbeg + rg;
    for (; r + 16 < rend; r += 32) { s0 += partial[(size_t)r * ROW + e]; s1 += partial[(size_t)(r + 16) * ROW + e]; }
    for (; r < rend; r += 16) s0 += partial[(size_t)r * ROW + e];
    fold[rg][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += fold[k][threadIdx.x];
        const int lane = e & 63, grp = e >> 6;
        if (grp == 0) atomicAdd(db + lane, t);
        else if (grp == 1) atomicAdd(dgamma + lane, t);
        else if (grp == 2) atomicAdd(dbeta + lane, t);
        else if (grp - 3 < C) atomicAdd(dW + lane * C + (grp - 3), t);
    }
}
// dW1 / db1 / dgamma / dbeta += the fold of nblk partial rows of 704 floats (used by vpf_adapter_front_bwd and vpf_adapter_kv_bwd)
int vpf_adapter_front_fold(const float* partial, int nblk, int C, float* dW, float* db, float* dgamma, float* dbeta, void* stream)
{
    hipLaunchKernelGGL(adapter_front_fold_kernel, dim3(3 + AD_MAXC, 8), dim3(1024), 0, (hipStream_t)stream, partial, nblk, C, dW, db, dgamma, dbeta);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
extern "C" int vpf_adapter_front_bwd(const float* x, const void* da_h16, long M, int C, const float* W, const float* b,
                                     const float* gamma, const float* beta, float* dW, float* db, float* dgamma, float* dbeta,
                                     float* ws, long ws_floats, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !da_h16 || !W || !b || !gamma || !beta || !dW || !db || !dgamma || !dbeta || !ws) return VPF_ERR_NULL;
    if (M < 0 || C <= 0 || C > AD_MAXC) return VPF_ERR_BADSHAPE;
    if (M == 0) return VPF_OK;
    constexpr int ROW = 64 * (3 + AD_MAXC);
    int nblk = grid_for(M, 64, 2048);
    if ((long)nblk * ROW > ws_floats) nblk = (int)(ws_floats / ROW);
    if (nblk < 1) return VPF_ERR_BADSHAPE;
    if (C == 3) hipLaunchKernelGGL(adapter_front_bwd_kernel<3>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, (const h16_t*)da_h16, M, C, W, b, gamma, beta, ws);
    else hipLaunchKernelGGL(adapter_front_bwd_kernel<AD_MAXC>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, (const h16_t*)da_h16, M, C, W, b, gamma, beta, ws);
    hipLaunchKernelGGL(adapter_front_fold_kernel, dim3(3 + AD_MAXC, 8), dim3(1024), 0, (hipStream_t)stream, (const float*)ws, nblk, C, dW, db, dgamma, dbeta);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== generic tiny-K front: y = act(x W^T + b), x f32 [M,C<=8], W [N,C]
// act 0 = none, 1 = GELU.  out h16 [M,N].  (position_emb[0:2]: C=3, N=128, GELU)
__global__ void smallk_fwd_kernel(const float* __restrict__ x, long M, int C, const float* __restrict__ W, const float* __restrict__ b,
                                  int N, int act, h16_t* __restrict__ out)
{
    const long total = M * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N); const long m = i / N;
        float u = b[n];
        for (int j = 0; j < C; ++j) u += W[n * C + j] * x[(size_t)m * C + j];
        out[i] = f32_to_h16(act == 1 ? gelu_f(u) : u);
    }
}
extern "C" int vpf_smallk_fwd(const float* x, long M, int C, const float* W, const float* b, int N, int act, void* out_h16, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !W || !b || !out_h16) return VPF_ERR_NULL;
    if (M < 0 || C <= 0 || C > 8 || N <= 0) return VPF_ERR_BADSHAPE;
    if (M == 0) return VPF_OK;
    hipLaunchKernelGGL(smallk_fwd_kernel, dim3(grid_for(M * N, 256)), dim3(256), 0, (hipStream_t)stream, x, M, C, W, b, N, act, (h16_t*)out_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// dy h16 [M,N] -> dW[N,C] +=, db[N] +=.   thread = output channel n, block = row slab
// block = bx output channels x 8 row lanes (blockDim = (bx, 8)); a block owns `rows_per_block` rows, row lane rl walks rows
// r0 + rl, r0 + rl + 8, ... four at a time; the 8 row lanes meet in LDS and ONE atomic per value and block leaves the CU
// (one thread per channel walking the whole slab, and hundreds of blocks adding into the same addresses, was 10x slower)
__global__ void __launch_bounds__(1024) smallk_bwd_kernel(const float* __restrict__ x, const h16_t* __restrict__ dy, long M, int C,
                                                         const float* __restrict__ W, const float* __restrict__ b, int N, int act,
                                                         float* __restrict__ dW, float* __restrict__ db, int rows_per_block)
{
    __shared__ float fold[8][9][128];
    const int tn = threadIdx.x, rl = threadIdx.y;
    const int n = blockIdx.x * blockDim.x + tn;
    const bool nok = n < N;
    float w[8], aw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w[j] = (nok && j < C) ? W[n * C + j] : 0.f; aw[j] = 0.f; }
    const float bb = nok ? b[n] : 0.f;
    float ab = 0.f;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (long r = r0 + rl; r < r1; r += 32) {
        float gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q] = (nok && r + 8 * q < r1) ? h16_to_f32(dy[(size_t)(r + 8 * q) * N + n]) : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long rr = r + 8 * q < r1 ? r + 8 * q : r1 - 1;
            float xv[8];
            float u = bb;
#pragma unroll
            for (int j = 0; j < 8; ++j) { xv[j] = j < C ? x[(size_t)rr * C + j] : 0.f; u += w[j] * xv[j]; }
            float g = gv[q];
            if (act == 1) g *= gelu_grad_f(u);
            ab += g;
#pragma unroll
            for (int j = 0; j < 8; ++j) aw[j] += g * xv[j];
        }
    }
    fold[rl][0][tn] = ab;
#pragma unroll
    for (int j = 0; j < 8; ++j) fold[rl][1 + j][tn] = aw[j];
    __syncthreads();
    if (rl == 0 && nok) {
        float t0 = 0.f, tw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tw[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            t0 += fold[k][0][tn];
#pragma unroll
            for (int j = 0; j < 8; ++j) tw[j] += fold[k][1 + j][tn];
        }
        atomicAdd(db + n, t0);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < C) atomicAdd(dW + n * C + j, tw[j]);
    }
}
extern "C" int vpf_smallk_bwd(const float* x, const void* dy_h16, long M, int C, const float* W, const float* b, int N, int act,
                              float* dW, float* db, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !dy_h16 || !W || !b || !dW || !db) return VPF_ERR_NULL;
    if (M < 0 || C <= 0 || C > 8 || N <= 0) return VPF_ERR_BADSHAPE;
    if (M == 0) return VPF_OK;
    // ~100 row blocks: with 24 (512 rows each) the kernel ran on 24 CUs and took 51 us for 12 k rows; with many more the
    // atomics of the blocks on the same N x (C + 1) addresses serialise in L2
    const int rpb0 = vpf_debug().smallk_rpb;
    int rpb = rpb0 > 0 ? rpb0 : (int)(((M + 95) / 96 + 31) / 32 * 32);
    if (rpb < 128) rpb = 128;                        // 8 row lanes x rpb / 8 rows each
    while ((M + rpb - 1) / rpb > 4096) rpb *= 2;
    const int bx = N >= 128 ? 128 : 64;
    dim3 grid(vpf_cdiv(N, bx), (unsigned)((M + rpb - 1) / rpb));
    hipLaunchKernelGGL(smallk_bwd_kernel, grid, dim3(bx, 8), 0, (hipStream_t)stream, x, (const h16_t*)dy_h16, M, C, W, b, N, act, dW, db, rpb);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== Group2Emb first conv (C -> 64) + BatchNorm(64) + ReLU
// h1 = x W^T + b is never stored: statistics pass, then normalise pass (both recompute it from the 12-byte row).
__global__ void __launch_bounds__(256) g2e_conv1_stats_kernel(const float* __restrict__ x, long M, int C, const float* __restrict__ W,
                                                            const float* __restrict__ b, float* __restrict__ sums, float* __restrict__ sumsq)
{
    const int lane = threadIdx.x & 63;
    const long wave0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    float w[AD_MAXC];
#pragma unroll
    for (int j = 0; j < AD_MAXC; ++j) w[j] = j < C ? W[lane * C + j] : 0.f;
    const float bb = b[lane];
    float s = 0.f, q = 0.f;
    for (long r = wave0; r < M; r += nw) {
        float h = bb;
#pragma unroll
        for (int j = 0; j < AD_MAXC; ++j) if (j < C) h += w[j] * x[(size_t)r * C + j];
        s += h; q += h * h;
    }
    atomicAdd(sums + lane, s); atomicAdd(sumsq + lane, q);
}
extern "C" int vpf_g2e_conv1_stats(const float* x, long M, int C, const float* W, const float* b, float* sums, float* sumsq, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !W || !b || !sums || !sumsq) return VPF_ERR_NULL;
    if (M <= 0 || C <= 0 || C > AD_MAXC) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(g2e_conv1_stats_kernel, dim3(grid_for(M, 512, 1024)), dim3(256), 0, (hipStream_t)stream, x, M, C, W, b, sums, sumsq);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// Batch statistics of h1 = x W^T + b WITHOUT touching the M x 64 matrix: h1 is affine in the C<=8 inputs, so
//   sum_m h1_c   = W_c . S1 + M b_c                     S1 = sum_m x        (C)
//   sum_m h1_c^2 = W_c S2 W_c^T + 2 b_c W_c . S1 + M b_c^2   S2 = sum_m x x^T   (C x C)
// mom (f32, zeroed): [S1 (8) | S2 (64)]
__global__ void __launch_bounds__(256) g2e_moments_kernel(const float* __restrict__ x, long M, int C, float* __restrict__ mom)
{
    __shared__ float red[4][72];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s1[AD_MAXC], s2[AD_MAXC][AD_MAXC];
#pragma unroll
    for (int i = 0; i < AD_MAXC; ++i) { s1[i] = 0.f;
#pragma unroll
        for (int j = 0; j < AD_MAXC; ++j) s2[i][j] = 0.f; }
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < M; r += (long)gridDim.x * blockDim.x) {
        float v[AD_MAXC];
#pragma unroll
        for (int i = 0; i < AD_MAXC; ++i) v[i] = i < C ? x[(size_t)r * C + i] : 0.f;
#pragma unroll
        for (int i = 0; i < AD_MAXC; ++i) if (i < C) { s1[i] += v[i];
#pragma unroll
            for (int j = 0; j < AD_MAXC; ++j) if (j <= i) s2[i][j] += v[i] * v[j]; }
    }
#pragma unroll
    for (int i = 0; i < AD_MAXC; ++i) if (i < C) {
        const float a = wave_sum(s1[i]);
        if (lane == 0) red[wv][i] = a;
#pragma unroll
        for (int j = 0; j < AD_MAXC; ++j) if (j <= i) { const float b = wave_sum(s2[i][j]); if (lane == 0) red[wv][8 + i * 8 + j] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 72) {
        const int e = threadIdx.x, i = e < 8 ? e : (e - 8) / 8, j = e < 8 ? 0 : (e - 8) % 8;
        mom[(size_t)blockIdx.x * 72 + e] = (i < C && (e < 8 || j <= i)) ? red[0][e] + red[1][e] + red[2][e] + red[3][e] : 0.f;
    }
}
__global__ void g2e_moments_to_sums_kernel(const float* __restrict__ mom, long M, int C, const float* __restrict__ W, const float* __restrict__ b,
                                           float* __restrict__ sums, float* __restrict__ sumsq)
{
    const int c = threadIdx.x;          // 64 channels
    if (c >= 64) return;
    float ws1 = 0.f, q = 0.f;
    for (int i = 0; i < C; ++i) {
        ws1 += W[c * C + i] * mom[i];
        for (int j = 0; j < C; ++j) { const float s2 = j <= i ? mom[8 + i * 8 + j] : mom[8 + j * 8 + i]; q += W[c * C + i] * W[c * C + j] * s2; }
    }
    const float bb = b[c];
    sums[c] = ws1 + (float)M * bb;
    sumsq[c] = q + 2.f * bb * ws1 + (float)M * bb * bb;
}
__global__ void __launch_bounds__(1024) g2e_moments_fold_kernel(const float* __restrict__ part, int nblk, float* __restrict__ mom)
{
    // 72 moments x 14 row groups, 4 loads in flight per thread, fixed fold order (deterministic statistics)
    __shared__ float fold[14][72];
    const int e = threadIdx.x % 72, rg = threadIdx.x / 72;
    if (rg >= 14) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rg;
    for (; r + 42 < nblk; r += 56) {
        s0 += part[(size_t)r * 72 + e]; s1 += part[(size_t)(r + 14) * 72 + e];
        s2 += part[(size_t)(r + 28) * 72 + e]; s3 += part[(size_t)(r + 42) * 72 + e];
    }
    for (; r < nblk; r += 14) s0 += part[(size_t)r * 72 + e];
    fold[rg][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 14; ++k) t += fold[k][e];
        mom[e] = t;
    }
}
// scratch: f32 [72 + 512*72]
extern "C" int vpf_g2e_conv1_stats_moments(const float* x, long M, int C, const float* W, const float* b, float* scratch,
                                           float* sums, float* sumsq, void* stream)
{
    (void)hipGetLastError();
    if (!x || !W || !b || !scratch || !sums || !sumsq) return VPF_ERR_NULL;
    if (M <= 0 || C <= 0 || C > AD_MAXC) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = grid_for(M, 256 * 4, 512);
    hipLaunchKernelGGL(g2e_moments_kernel, dim3(nblk), dim3(256), 0, st, x, M, C, scratch + 72);
    hipLaunchKernelGGL(g2e_moments_fold_kernel, dim3(1), dim3(1024), 0, st, (const float*)(scratch + 72), nblk, scratch);
    hipLaunchKernelGGL(g2e_moments_to_sums_kernel, dim3(1), dim3(64), 0, st, (const float*)scratch, M, C, W, b, sums, sumsq);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// Everything between the input moments and the first persistent kernel in ONE single-block launch (training mode): fold of the
// per-block moment partials (fixed order) -> per-channel sum / sum^2 of the first conv's output -> BatchNorm-1 batch statistics
// + running-statistics update -> the BatchNorm as an affine (a | b) -> the affine folded into the conv (w1e, b1e).  Five
// dependent 64-thread launches of ~5 us each sat on the point-cloud branch's critical path for this.
__global__ void __launch_bounds__(1024) g2e_bn1_prepare_kernel(const float* __restrict__ part, int nblk, long M, int C, const float* __restrict__ W,
                                                              const float* __restrict__ b, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float momentum,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              long long* __restrict__ num_batches, float* __restrict__ stat,
                                                              float* __restrict__ ab, float* __restrict__ w1e, float* __restrict__ b1e,
                                                              float* __restrict__ mom_out)
{
    __shared__ float fold[14][72];
    __shared__ float mom[72];
    const int e = threadIdx.x % 72, rg = threadIdx.x / 72;
    if (rg < 14) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = rg;
        for (; r + 42 < nblk; r += 56) {
            s0 += part[(size_t)r * 72 + e]; s1 += part[(size_t)(r + 14) * 72 + e];
            s2 += part[(size_t)(r + 28) * 72 + e]; s3 += part[(size_t)(r + 42) * 72 + e];
        }
        for (; r < nblk; r += 14) s0 += part[(size_t)r * 72 + e];
        fold[rg][e] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 14; ++k) t += fold[k][e];
        mom[e] = t;
        mom_out[e] = t;                      // kept for the backward pass (vpf_g2e_conv1_bwd)
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c >= 64) return;
    float ws1 = 0.f, q = 0.f;
    for (int i = 0; i < C; ++i) {
        ws1 += W[c * C + i] * mom[i];
        for (int j = 0; j < C; ++j) { const float s2 = j <= i ? mom[8 + i * 8 + j] : mom[8 + j * 8 + i]; q += W[c * C + i] * W[c * C + j] * s2; }
    }
    const float bb = b[c];
    const float sum = ws1 + (float)M * bb, sumsq = q + 2.f * bb * ws1 + (float)M * bb * bb;
    const float mu = sum / (float)M;
    float var = sumsq / (float)M - mu * mu;
    var = var < 0.f ? 0.f : var;
    const float rs = rsqrtf(var + eps);
    stat[c] = mu; stat[64 + c] = rs;
    if (running_mean) {
        const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
    if (num_batches && c == 0) *num_batches += 1;
    const float a = rs * gamma[c], bo = beta[c] - mu * a;
    ab[c] = a; ab[64 + c] = bo;
    for (int i = 0; i < C; ++i) w1e[c * C + i] = a * W[c * C + i];
    b1e[c] = a * bb + bo;
}
// scratch: f32 [72 + 512*72]
extern "C" int vpf_g2e_bn1_prepare(const float* x, long M, int C, const float* W, const float* b, float* scratch, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                   long long* num_batches, float* stat, float* ab, float* w1e, float* b1e, void* stream)
{
    (void)hipGetLastError();
    if (!x || !W || !b || !scratch || !gamma || !beta || !stat || !ab || !w1e || !b1e) return VPF_ERR_NULL;
    if (M <= 0 || C <= 0 || C > AD_MAXC) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = grid_for(M, 256 * 4, 512);
    hipLaunchKernelGGL(g2e_moments_kernel, dim3(nblk), dim3(256), 0, st, x, M, C, scratch + 72);
    hipLaunchKernelGGL(g2e_bn1_prepare_kernel, dim3(1), dim3(1024), 0, st, (const float*)(scratch + 72), nblk, M, C, W, b, gamma, beta, eps, momentum,
                       running_mean, running_var, num_batches, stat, ab, w1e, b1e, scratch);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// BatchNorm (training mode) from per-workgroup partial rows [nrows][2C] = (sum | sum^2) in one launch: fixed-order fold -> batch
// statistics + running-statistics update -> the BatchNorm as an affine (ab = a | b).  C % 64 == 0; one block per 64 channels
// (their 64 sums and 64 sums of squares x 8 row groups = 1024 threads, 4 loads in flight per thread).
__global__ void __launch_bounds__(1024) bn_partials_finalize_kernel(const float* __restrict__ part, int nrows, int C, long M, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float eps, float momentum,
                                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                    long long* __restrict__ num_batches, float* __restrict__ stat, float* __restrict__ ab)
{
    __shared__ float fold[8][128];
    const int W2 = 2 * C;
    const int e = threadIdx.x & 127, rg = threadIdx.x >> 7;
    const int col = e < 64 ? blockIdx.x * 64 + e : C + blockIdx.x * 64 + (e - 64);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rg;
    for (; r + 24 < nrows; r += 32) {
        s0 += part[(size_t)r * W2 + col]; s1 += part[(size_t)(r + 8) * W2 + col];
        s2 += part[(size_t)(r + 16) * W2 + col]; s3 += part[(size_t)(r + 24) * W2 + col];
    }
    for (; r < nrows; r += 8) s0 += part[(size_t)r * W2 + col];
    fold[rg][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int c = blockIdx.x * 64 + threadIdx.x;
    float sum = 0.f, sumsq = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sum += fold[k][threadIdx.x]; sumsq += fold[k][64 + threadIdx.x]; }
    const float mu = sum / (float)M;
    float var = sumsq / (float)M - mu * mu;
    var = var < 0.f ? 0.f : var;
    const float rs = rsqrtf(var + eps);
    stat[c] = mu; stat[C + c] = rs;
    if (running_mean) {
        const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
    if (num_batches && c == 0) *num_batches += 1;
    const float a = rs * gamma[c];
    ab[c] = a; ab[C + c] = beta[c] - mu * a;
}
extern "C" int vpf_bn_partials_finalize(const float* partials, int nrows, int C, long M, const float* gamma, const float* beta, float eps,
                                        float momentum, float* running_mean, float* running_var, long long* num_batches, float* stat,
                                        float* ab, void* stream)
{
    (void)hipGetLastError();
    if (!partials || !gamma || !beta || !stat || !ab) return VPF_ERR_NULL;
    if (nrows <= 0 || C <= 0 || (C % 64) || M <= 0) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(bn_partials_finalize_kernel, dim3(C / 64), dim3(1024), 0, (hipStream_t)stream, partials, nrows, C, M, gamma, beta, eps, momentum,
                       running_mean, running_var, num_batches, stat, ab);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

__global__ void __launch_bounds__(256) g2e_conv1_apply_kernel(const float* __restrict__ x, long M, int C, const float* __restrict__ W,
                                                            const float* __restrict__ b, const float* __restrict__ stat,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, h16_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long wave0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    float w[AD_MAXC];
#pragma unroll
    for (int j = 0; j < AD_MAXC; ++j) w[j] = j < C ? W[lane * C + j] : 0.f;
    const float bb = b[lane], mu = stat[lane], rs = stat[64 + lane], ga = gamma[lane], be = beta[lane];
    constexpr int U = 8;
    for (long r0 = wave0 * U; r0 < M; r0 += nw * U) {
        float h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + u < M ? r0 + u : M - 1;
            h[u] = bb;
#pragma unroll
            for (int j = 0; j < AD_MAXC; ++j) if (j < C) h[u] += w[j] * x[(size_t)r * C + j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r0 + u < M) out[(size_t)(r0 + u) * 64 + lane] = f32_to_h16(fmaxf((h[u] - mu) * rs * ga + be, 0.f));
    }
}
extern "C" int vpf_g2e_conv1_apply(const float* x, long M, int C, const float* W, const float* b, const float* stat, const float* gamma,
                                   const float* beta, void* out_h16, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !W || !b || !stat || !gamma || !beta || !out_h16) return VPF_ERR_NULL;
    if (M <= 0 || C <= 0 || C > AD_MAXC) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(g2e_conv1_apply_kernel, dim3(grid_for(M, 64, 2048)), dim3(256), 0, (hipStream_t)stream, x, M, C, W, b, stat, gamma, beta, (h16_t*)out_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// Backward of Conv1d(C, 64) + BatchNorm-1 + ReLU in ONE pass over the incoming gradient.  With g = da masked by the ReLU and
// xh = the normalised activation, BatchNorm's backward is dh = gamma rs (g - mean(g) - xh mean(g xh)); every gradient is a combination of
//   S_g = sum g,  S_gxh = sum g xh,  S_gx[i] = sum g x_i                      (this kernel: 5 values per channel)
// and of sums over the INPUT alone, which the forward pass already has (g2e_moments_kernel: S1 = sum x, S2 = sum x x^T):
//   sum_r xh_c = 0,   sum_r xh_c x_i = rs_c (sum_j W_cj S2[j][i] + (b_c - mu_c) S1[i])
//   dW_ci = gamma_c rs_c (S_gx[i] - S_g / M S1[i] - S_gxh / M sum_r xh_c x_i),   db_c = 0,   dgamma_c = S_gxh,   dbeta_c = S_g.
// (The two-pass form -- statistics, then dh -- read the 50 MB gradient twice: 2 x 117 us of the point-cloud branch's tail.)
// thread = (row, 8 channels): 16-byte loads of the incoming gradient, four rows in flight, per-thread partial sums, LDS fold per block.
__global__ void __launch_bounds__(256) g2e_conv1_bwd_kernel(const float* __restrict__ x, const h16_t* __restrict__ da, long M, int C,
                                                          const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ stat,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ partial)
{
    __shared__ float redf[8 * 320];              // [4 waves x 2 halves][5 values][64 channels]
    const int t = threadIdx.x, rl = t >> 3, cg = (t & 7) * 8;
    float wr[8][3], br[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg + j;
        const float rs = stat[64 + c];
        br[j] = (b[c] - stat[c]) * rs; ga[j] = gamma[c]; be[j] = beta[c];
#pragma unroll
        for (int i = 0; i < 3; ++i) wr[j][i] = i < C ? W[c * C + i] * rs : 0.f;
    }
    float a0[8], a1[8], aw[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a0[j] = a1[j] = 0.f; aw[j][0] = aw[j][1] = aw[j][2] = 0.f; }
    const long rstep = (long)gridDim.x * 32;
    auto row_math = [&](const uint4 dv, const float x0, const float x1, const float x2) {
        const uint32_t uw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = wr[j][0] * x0 + wr[j][1] * x1 + wr[j][2] * x2 + br[j];
            float g = (j & 1) ? h16_hi(uw[j >> 1]) : h16_lo(uw[j >> 1]);
            if (xh * ga[j] + be[j] <= 0.f) g = 0.f;
            a0[j] += g; a1[j] += g * xh; aw[j][0] += g * x0; aw[j][1] += g * x1; aw[j][2] += g * x2;
        }
    };
    for (long r = (long)blockIdx.x * 32 + rl; r < M; r += 4 * rstep) {
        uint4 dv[4]; float xv[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long ru = r + u * rstep;
            const bool ok = ru < M;
            dv[u] = ok ? *reinterpret_cast<const uint4*>(da + (size_t)ru * 64 + cg) : make_uint4(0, 0, 0, 0);     // (a zero gradient adds nothing)
#pragma unroll
            for (int i = 0; i < 3; ++i) xv[u][i] = (ok && i < C) ? x[(size_t)ru * C + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) row_math(dv[u], xv[u][0], xv[u][1], xv[u][2]);
    }
    // Fold the 8 row-lanes of every channel group WITHOUT atomics: lanes that share a channel group differ in lane bits 3..5
    // (lane ^ 8 = a rotate by 8 inside the 16-lane DPP row, lane ^ 16 = a ds_swizzle; the two 32-lane halves go to LDS separately),
    // the 4 waves meet in LDS, and the block leaves one partial row [5][64] for the fold kernel
#define VPF_FOLD2(v) do { v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));   \
                          v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); } while (0)
#pragma unroll
    for (int j = 0; j < 8; ++j) { VPF_FOLD2(a0[j]); VPF_FOLD2(a1[j]); VPF_FOLD2(aw[j][0]); VPF_FOLD2(aw[j][1]); VPF_FOLD2(aw[j][2]); }
#undef VPF_FOLD2
    const int wv = t >> 6, lane = t & 63;
    if ((lane & 31) < 8) {
        float* dst = redf + (wv * 2 + (lane >> 5)) * 320 + (lane & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { dst[j] = a0[j]; dst[64 + j] = a1[j]; dst[128 + j] = aw[j][0]; dst[192 + j] = aw[j][1]; dst[256 + j] = aw[j][2]; }
    }
    __syncthreads();
    for (int e = t; e < 320; e += 256) {
        float sacc = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) sacc += redf[w8 * 320 + e];
        partial[(size_t)blockIdx.x * 320 + e] = sacc;
    }
}
// The same sums with the second conv's input gradient folded in: da = dh2 . W2 (Conv1d(64, 128) backward, K = 128) is produced on the
// matrix cores from dh2 rows read straight from HBM in A-fragment layout and consumed in the accumulator registers -- the 50 MB h16
// da tensor is neither written nor read (a 42 us GEMM + a 55 us sums pass -> one pass over dh2).  A wave owns 32 rows x 64 channels per
// iteration; in the accumulator layout a lane owns ONE channel per 32-channel tile (column = lane & 31) and 16 of the 32 rows, so the
// five per-channel sums are plain per-lane accumulations; the rows' inputs x sit in a wave-private LDS slot.
typedef h16x8_t g2e_h16x8_t;
typedef __attribute__((ext_vector_type(16))) float g2e_f32x16_t;
__global__ void __launch_bounds__(256) g2e_conv1_bwd_fused_kernel(const float* __restrict__ x, const h16_t* __restrict__ dh2, long M, int C,
                                                                const float* __restrict__ W, const float* __restrict__ b,
                                                                const float* __restrict__ stat, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const h16_t* __restrict__ W2,
                                                                float* __restrict__ partial)
{
    __shared__ float redf[4 * 320];              // [4 waves][5 values][64 channels]
    __shared__ __attribute__((aligned(16))) h16_t sW2[128 * 72];     // W2 [128 out][64 in], rows padded to 72
    __shared__ float sAccAll[4 * 16 * 64];                            // per wave: the accumulator tile [16][64] f32
    __shared__ float4 sXrow[4][32];                                   // per wave: the current tile's input rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cl = lane & 31, hl = lane >> 5;
    for (int e = threadIdx.x; e < 128 * 8; e += 256)
        *reinterpret_cast<uint4*>(sW2 + (e >> 3) * 72 + (e & 7) * 8) = *reinterpret_cast<const uint4*>(W2 + (size_t)e * 8);
    __syncthreads();
    // W2 as B fragments: column n = input channel (lane & 31, + 32 j), 8 consecutive k = output channels 16 ks + 8 hl ..  The matrix is
    // k-strided for that ([k][n] rows): read from LDS with the transposing ds_read_b64_tr_b16 (as gemm.hip's frag_read) in front of
    // every MFMA -- holding the 16 fragments in registers instead costs 64 VGPRs, i.e. a wave per SIMD and the room for the next tile's loads
    typedef __attribute__((ext_vector_type(4))) short s16x4_t;
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const int g16 = lane >> 4, i16 = lane & 15;
    const h16_t* wfrag = sW2 + (8 * (g16 >> 1) + (i16 >> 2)) * 72 + 16 * (g16 & 1) + 4 * (i16 & 3);
    auto bfrag = [&](int j, int ks) {
        const h16_t* a0p = wfrag + ks * 16 * 72 + 32 * j;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0p));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0p + 4 * 72));
        s16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(g2e_h16x8_t, v);
    };
    float* sAcc = sAccAll + wave * 16 * 64;
    float wr[2][3], br[2], ga[2], be[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = 32 * j + cl;
        const float rs = stat[64 + c];
        br[j] = (b[c] - stat[c]) * rs; ga[j] = gamma[c]; be[j] = beta[c];
#pragma unroll
        for (int i = 0; i < 3; ++i) wr[j][i] = i < C ? W[c * C + i] * rs : 0.f;
    }
    float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f}, aw[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const long ntiles = (M + 31) / 32, tstep = (long)gridDim.x * 4;
    uint4 af[8]; float4 xr0;
    auto load_tile = [&](long tile) {                             // A fragments straight from HBM: row = lane & 31, 8 values at k = 16 ks + 8 hl
        const long rr = min(min(tile, ntiles - 1) * 32 + cl, M - 1);      // (rows / tiles past the end: clamped loads, their gradient is zeroed below)
        const h16_t* src = dh2 + (size_t)rr * 128 + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const uint4*>(src + ks * 16);
        const float* xp = x + (size_t)rr * C;
        xr0 = make_float4(xp[0], C > 1 ? xp[1] : 0.f, C > 2 ? xp[2] : 0.f, 0.f);
    };
    long tile = (long)blockIdx.x * 4 + wave;
    load_tile(tile);
    for (; tile < ntiles; tile += tstep) {
        const int nrows = (int)min(32L, M - tile * 32);
        sXrow[wave][cl] = xr0;                                     // (wave-private: the wave's own LDS accesses are ordered)
        g2e_f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const g2e_h16x8_t a = __builtin_bit_cast(g2e_h16x8_t, af[ks]);
            acc[0] = vpf_mfma32(a, bfrag(0, ks), acc[0]);
            acc[1] = vpf_mfma32(a, bfrag(1, ks), acc[1]);
        }
        load_tile(tile + tstep);                                  // the next tile's rows travel under this tile's per-row arithmetic
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // the 16 accumulator registers go through a wave-private LDS slot so that the per-row arithmetic can be a ROLLED loop: fully
            // unrolled, the scheduler keeps ~180 registers of row products alive (280 in all, one wave per SIMD, 100 us)
#pragma unroll
            for (int r = 0; r < 16; ++r) sAcc[r * 64 + lane] = acc[j][r];
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hl;    // C / D layout of the 32 x 32 MFMA
                const float4 xr = sXrow[wave][row];
                const float xh = wr[j][0] * xr.x + wr[j][1] * xr.y + wr[j][2] * xr.z + br[j];
                float g = sAcc[r * 64 + lane];
                const bool dead = (xh * ga[j] + be[j] <= 0.f) | (row >= nrows);      // (one select; `a || b` is a branch per row here)
                g = dead ? 0.f : g;
                a0[j] += g; a1[j] += g * xh; aw[j][0] += g * xr.x; aw[j][1] += g * xr.y; aw[j][2] += g * xr.z;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        a0[j] += __shfl_xor(a0[j], 32); a1[j] += __shfl_xor(a1[j], 32);
        aw[j][0] += __shfl_xor(aw[j][0], 32); aw[j][1] += __shfl_xor(aw[j][1], 32); aw[j][2] += __shfl_xor(aw[j][2], 32);
        if (hl == 0) {
            float* dst = redf + wave * 320 + 32 * j + cl;
            dst[0] = a0[j]; dst[64] = a1[j]; dst[128] = aw[j][0]; dst[192] = aw[j][1]; dst[256] = aw[j][2];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 320; e += 256)
        partial[(size_t)blockIdx.x * 320 + e] = (redf[e] + redf[320 + e]) + (redf[640 + e] + redf[960 + e]);
}
// fold of the per-block partial rows [nblk][320] in a fixed order: block = 32 of the 320 columns x 32 row groups -> sums[320]
__global__ void __launch_bounds__(1024) g2e_conv1_fold_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ sums)
{
    __shared__ float fold[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5, col = blockIdx.x * 32 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rg;
    for (; r + 96 < nblk; r += 128) {
        s0 += partial[(size_t)r * 320 + col]; s1 += partial[(size_t)(r + 32) * 320 + col];
        s2 += partial[(size_t)(r + 64) * 320 + col]; s3 += partial[(size_t)(r + 96) * 320 + col];
    }
    for (; r < nblk; r += 32) s0 += partial[(size_t)r * 320 + col];
    fold[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0) {
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tsum += fold[k][cl];
        sums[col] = tsum;
    }
}
// the algebra of the header comment, in double (the terms of dW cancel to a few percent of their size)
__global__ void g2e_conv1_grads_kernel(const float* __restrict__ sums, const float* __restrict__ mom, long M, int C, const float* __restrict__ W,
                                       const float* __restrict__ b, const float* __restrict__ stat, const float* __restrict__ gamma, int training,
                                       float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const int c = threadIdx.x;
    if (c >= 64) return;
    const double Sg = sums[c], Sgxh = sums[64 + c], rs = stat[64 + c], ga = gamma[c];
    dgamma[c] += (float)Sgxh;
    dbeta[c] += (float)Sg;
    if (!training) {
        db[c] += (float)(ga * rs * Sg);
        for (int i = 0; i < C; ++i) dW[c * C + i] += (float)(ga * rs * (double)sums[128 + 64 * i + c]);
        return;
    }
    // (db = gamma rs (S_g - M mean(g) - mean(g xh) sum xh) = 0: a bias in front of a training-mode BatchNorm has no gradient)
    for (int i = 0; i < C; ++i) {
        double sxhx = ((double)b[c] - (double)stat[c]) * (double)mom[i];
        for (int j = 0; j < C; ++j) sxhx += (double)W[c * C + j] * (double)(j <= i ? mom[8 + i * 8 + j] : mom[8 + j * 8 + i]);
        sxhx *= rs;
        const double v = (double)sums[128 + 64 * i + c] - Sg / (double)M * (double)mom[i] - Sgxh / (double)M * sxhx;
        dW[c * C + i] += (float)(ga * rs * v);
    }
}
extern "C" int vpf_g2e_conv1_bwd(const float* x, const void* da_h16, long M, int C, const float* W, const float* b, const float* stat,
                                 const float* gamma, const float* beta, int training, const float* mom, float* dW, float* db,
                                 float* dgamma, float* dbeta, float* ws, long ws_floats, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!x || !da_h16 || !W || !b || !stat || !gamma || !beta || !dW || !db || !dgamma || !dbeta || !ws) return VPF_ERR_NULL;
    if (training && !mom) return VPF_ERR_NULL;
    if (M <= 0 || C <= 0 || C > 3) return VPF_ERR_BADSHAPE;
    int grid = grid_for(M, 32 * 4, 1024);
    if ((long)(grid + 1) * 320 > ws_floats) grid = (int)(ws_floats / 320) - 1;
    if (grid < 1) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    float* sums = ws + (size_t)grid * 320;
    hipLaunchKernelGGL(g2e_conv1_bwd_kernel, dim3(grid), dim3(256), 0, st, x, (const h16_t*)da_h16, M, C, W, b, stat, gamma, beta, ws);
    hipLaunchKernelGGL(g2e_conv1_fold_kernel, dim3(10), dim3(1024), 0, st, (const float*)ws, grid, sums);
    hipLaunchKernelGGL(g2e_conv1_grads_kernel, dim3(1), dim3(64), 0, st, (const float*)sums, mom, M, C, W, b, stat, gamma, training, dW, db, dgamma, dbeta);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

extern "C" int vpf_g2e_conv1_bwd_fused(const float* x, const void* dh2_h16, long M, int C, const float* W, const float* b, const float* stat,
                                       const float* gamma, const float* beta, int training, const float* mom, const void* W2_h16,
                                       float* dW, float* db, float* dgamma, float* dbeta, float* ws, long ws_floats, void* stream)
{
    (void)hipGetLastError();
    if (!x || !dh2_h16 || !W || !b || !stat || !gamma || !beta || !W2_h16 || !dW || !db || !dgamma || !dbeta || !ws) return VPF_ERR_NULL;
    if (training && !mom) return VPF_ERR_NULL;
    if (M <= 0 || C <= 0 || C > 3) return VPF_ERR_BADSHAPE;
    if ((uintptr_t)dh2_h16 & 15) return VPF_ERR_BADALIGN;
    int grid = grid_for(M, 32 * 4 * 3, 1024);
    if ((long)(grid + 1) * 320 > ws_floats) grid = (int)(ws_floats / 320) - 1;
    if (grid < 1) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    float* sums = ws + (size_t)grid * 320;
    hipLaunchKernelGGL(g2e_conv1_bwd_fused_kernel, dim3(grid), dim3(256), 0, st, x, (const h16_t*)dh2_h16, M, C, W, b, stat, gamma, beta,
                       (const h16_t*)W2_h16, ws);
    hipLaunchKernelGGL(g2e_conv1_fold_kernel, dim3(10), dim3(1024), 0, st, (const float*)ws, grid, sums);
    hipLaunchKernelGGL(g2e_conv1_grads_kernel, dim3(1), dim3(64), 0, st, (const float*)sums, mom, M, C, W, b, stat, gamma, training, dW, db, dgamma, dbeta);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== patchify  'b (h p1) (w p2) c -> b (h w) (p1 p2 c)'
// imgs is an arbitrary-stride [B,H,W,3] view (pretrain.py:179 hands a permuted NCHW tensor); out h16 [B*T, p*p*3]
__global__ void patchify_kernel(const float* __restrict__ img, long sb, long sh, long sw, long sc, int B, int Hh, int Ww, int Cc, int p,
                                h16_t* __restrict__ out)
{
    const int wp = Ww / p, hp = Hh / p, pd = p * p * Cc;
    const long total = (long)B * hp * wp * pd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % pd); const long tok = i / pd;
        const int c = e % Cc, p2 = (e / Cc) % p, p1 = e / (Cc * p);
        const int tw = (int)(tok % wp), th = (int)((tok / wp) % hp); const long b = tok / ((long)wp * hp);
        out[i] = f32_to_h16(img[b * sb + (long)(th * p + p1) * sh + (long)(tw * p + p2) * sw + c * sc]);
    }
}
// One block per (image, row of patches): the p image rows x W x C slab is read with the thread on the w axis (contiguous
// in the NCHW memory the loader hands over), re-ordered in LDS into the token-major output order and written back with
// 16-byte stores (the slab's tokens are one contiguous piece of the output).  8 loads in flight per thread.
__global__ void __launch_bounds__(256) patchify_rows_kernel(const float* __restrict__ img, long sb, long sh, long sw, long sc, int Hh, int Ww,
                                                           int Cc, int p, h16_t* __restrict__ out)
{
    extern __shared__ h16_t tile[];                 // [wp][p*p*Cc]
    const int wp = Ww / p, hp = Hh / p, pd = p * p * Cc;
    const int b = blockIdx.x / hp, th = blockIdx.x % hp;
    const float* base = img + (long)b * sb + (long)(th * p) * sh;
    const int npl = Cc * p;                           // (channel, image row) planes of the slab
    for (int w = threadIdx.x; w < wp * p; w += blockDim.x) {
        const int tw = w / p, p2 = w % p;
        for (int pl0 = 0; pl0 < npl; pl0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pl = pl0 + u < npl ? pl0 + u : npl - 1, c = pl / p, p1 = pl % p;
                v[u] = base[(long)p1 * sh + (long)w * sw + (long)c * sc];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pl = pl0 + u;
                if (pl < npl) { const int c = pl / p, p1 = pl % p; tile[tw * pd + (p1 * p + p2) * Cc + c] = f32_to_h16(v[u]); }
            }
        }
    }
    __syncthreads();
    const int nchunk = wp * pd / 8;                   // 16-byte chunks (pd % 8 == 0 checked by the launcher)
    uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)b * hp + th) * wp * pd);
    const uint4* src = reinterpret_cast<const uint4*>(tile);
    for (int e = threadIdx.x; e < nchunk; e += blockDim.x) dst[e] = src[e];
}
extern "C" int vpf_patchify(const float* img, long sb, long sh, long sw, long sc, int B, int H, int W, int C, int p, void* out_h16, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!img || !out_h16) return VPF_ERR_NULL;
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || p <= 0 || (H % p) || (W % p)) return VPF_ERR_BADSHAPE;
    if (B == 0) return VPF_OK;
    const size_t lds = sizeof(h16_t) * (size_t)(W / p) * p * p * C;
    if ((p * p * C) % 8 == 0 && lds <= 64 * 1024 && (((uintptr_t)out_h16) & 15) == 0)
        hipLaunchKernelGGL(patchify_rows_kernel, dim3(B * (H / p)), dim3(256), lds, (hipStream_t)stream, img, sb, sh, sw, sc, H, W, C, p, (h16_t*)out_h16);
    else
        hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((long)B * H * W * C, 256)), dim3(256), 0, (hipStream_t)stream, img, sb, sh, sw, sc, B, H, W, C, p, (h16_t*)out_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== NT-Xent (lightly 1.1.21 formulation, memory bank 0)
// z0,z1 f32 [b,D].  n = 2b rows.  fwd: zn (normalised rows) [n,D], inv norms [n], P [n,n] softmax over j != i of zn_i.zn_j/T,
// loss = mean_i -log P[i, pos(i)], pos(i) = (i + b) mod n.  bwd: dz from dloss.
__global__ void __launch_bounds__(256) ntxent_norm_kernel(const float* __restrict__ z0, const float* __restrict__ z1, int b, int D,
                                                        float* __restrict__ zn, float* __restrict__ inv)
{
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= 2 * b) return;
    const float* src = r < b ? z0 + (size_t)r * D : z1 + (size_t)(r - b) * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += src[c] * src[c];
    s = wave_sum(s);
    const float iv = 1.f / fmaxf(sqrtf(s), 1e-12f);     // F.normalize eps
    for (int c = lane; c < D; c += 64) zn[(size_t)r * D + c] = src[c] * iv;
    if (lane == 0) inv[r] = iv;
}
// one workgroup per row i
__global__ void __launch_bounds__(256) ntxent_row_kernel(const float* __restrict__ zn, int b, int D, float invT, float* __restrict__ P,
                                                       float* __restrict__ loss_rows)
{
    extern __shared__ float sm[];      // zi[D] + logits[n]
    const int n = 2 * b, i = blockIdx.x;
    zn += (size_t)blockIdx.y * n * D; P += (size_t)blockIdx.y * n * n; loss_rows += (size_t)blockIdx.y * n;     // problem index (fused losses)
    float* zi = sm; float* lg = sm + D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) zi[c] = zn[(size_t)i * D + c];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    // 8 rows per wave at a time: their loads are issued together (one running dot product per row would pay the load
    // latency n / nwv times in sequence, and this kernel sits alone between the forward and the backward pass)
    for (int j0 = wv * 8; j0 < n; j0 += nwv * 8) {
        float s[8];
        if (D == 256) {
            // fixed trip counts: the 32 loads of the 8 rows are really in flight together
            float zr[4], v[8][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) zr[k] = zi[lane + 64 * k];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u < n ? j0 + u : n - 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = zn[(size_t)j * 256 + lane + 64 * k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = (zr[0] * v[u][0] + zr[1] * v[u][1]) + (zr[2] * v[u][2] + zr[3] * v[u][3]);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s[u] = 0.f;
                const int j = j0 + u < n ? j0 + u : n - 1;
                for (int c = lane; c < D; c += 64) s[u] += zi[c] * zn[(size_t)j * D + c];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float t = wave_sum(s[u]);
            const int j = j0 + u;
            if (lane == 0 && j < n) lg[j] = (j == i) ? -INFINITY : t * invT;
        }
    }
    __syncthreads();
    __shared__ float red[8];
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < n; j += blockDim.x) mx = fmaxf(mx, lg[j]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < nwv; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();
    float se = 0.f;
    for (int j = threadIdx.x; j < n; j += blockDim.x) se += __expf(lg[j] - mx);
    se = wave_sum(se);
    if (lane == 0) red[wv] = se;
    __syncthreads();
    se = 0.f;
    for (int w = 0; w < nwv; ++w) se += red[w];
    const float lse = mx + logf(se);
    for (int j = threadIdx.x; j < n; j += blockDim.x) P[(size_t)i * n + j] = (j == i) ? 0.f : __expf(lg[j] - lse);
    if (threadIdx.x == 0) loss_rows[i] = lse - lg[(i + b) % n];
}
__global__ void ntxent_mean_kernel(const float* __restrict__ loss_rows, int n, float* __restrict__ loss)
{
    float s = 0.f;
    for (int j = threadIdx.x; j < n; j += 64) s += loss_rows[j];
    s = wave_sum(s);
    if (threadIdx.x == 0) *loss = s / (float)n;
}
extern "C" int vpf_ntxent_fwd(const float* z0, const float* z1, int b, int D, float temperature, float* zn, float* inv_norm, float* P,
                              float* loss_rows, float* loss, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!z0 || !z1 || !zn || !inv_norm || !P || !loss_rows || !loss) return VPF_ERR_NULL;
    if (b <= 0 || D <= 0 || 2 * b > 8192) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int n = 2 * b;
    hipLaunchKernelGGL(ntxent_norm_kernel, dim3(vpf_cdiv(n, 4)), dim3(256), 0, st, z0, z1, b, D, zn, inv_norm);
    hipLaunchKernelGGL(ntxent_row_kernel, dim3(n), dim3(256), sizeof(float) * (D + n), st, zn, b, D, 1.f / temperature, P, loss_rows);
    hipLaunchKernelGGL(ntxent_mean_kernel, dim3(1), dim3(64), 0, st, loss_rows, n, loss);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// dzn_i = (dL/n/T) * sum_j (G[i,j] + G[j,i]) zn_j,  G = P - onehot(pos);  dz = inv * (dzn - zn (zn . dzn))
__global__ void __launch_bounds__(256) ntxent_bwd_kernel(const float* __restrict__ zn, const float* __restrict__ inv, const float* __restrict__ P,
                                                       int b, int D, float invT, const float* __restrict__ dloss, float* __restrict__ dz0,
                                                       float* __restrict__ dz1, float w1)
{
    extern __shared__ float sm[];      // coef[n] + dzn[D]
    const int n = 2 * b, i = blockIdx.x;
    // problem index (fused losses): problem 1 is weighted by w1 and writes the second [n, D] slab of dz0 (dz1 = dz0 + b rows)
    zn += (size_t)blockIdx.y * n * D; inv += (size_t)blockIdx.y * n; P += (size_t)blockIdx.y * n * n;
    dz0 += (size_t)blockIdx.y * n * D; dz1 += (size_t)blockIdx.y * n * D;
    float* coef = sm; float* dzn = sm + n;
    const float gs = dloss[0] * (blockIdx.y ? w1 : 1.f) * invT / (float)n;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        float g = P[(size_t)i * n + j] + P[(size_t)j * n + i];
        if (j == (i + b) % n) g -= 1.f;
        if (i == (j + b) % n) g -= 1.f;
        coef[j] = (j == i) ? 0.f : g * gs;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < n; ++j) s += coef[j] * zn[(size_t)j * D + c];
        dzn[c] = s;
    }
    __syncthreads();
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    float dot = 0.f;
    for (int c = threadIdx.x; c < D; c += blockDim.x) dot += dzn[c] * zn[(size_t)i * D + c];
    dot = wave_sum(dot);
    if (lane == 0) red[wv] = dot;
    __syncthreads();
    dot = 0.f;
    for (int w = 0; w < nwv; ++w) dot += red[w];
    float* dst = i < b ? dz0 + (size_t)i * D : dz1 + (size_t)(i - b) * D;
    const float iv = inv[i];
    for (int c = threadIdx.x; c < D; c += blockDim.x) dst[c] = iv * (dzn[c] - zn[(size_t)i * D + c] * dot);
}
extern "C" int vpf_ntxent_bwd(const float* zn, const float* inv_norm, const float* P, int b, int D, float temperature, const float* dloss,
                              float* dz0, float* dz1, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!zn || !inv_norm || !P || !dloss || !dz0 || !dz1) return VPF_ERR_NULL;
    if (b <= 0 || D <= 0 || 2 * b > 8192) return VPF_ERR_BADSHAPE;
    const int n = 2 * b;
    hipLaunchKernelGGL(ntxent_bwd_kernel, dim3(n), dim3(256), sizeof(float) * (n + D), (hipStream_t)stream, zn, inv_norm, P, b, D, 1.f / temperature, dloss, dz0, dz1, 1.f);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== fused AdamW over a flat buffer (+ h16 shadow)
// torch.optim.AdamW semantics (decoupled weight decay, bias correction) + torch.cuda.amp.GradScaler's step / update
// (pretrain.py:154,209-211: scaler.scale(loss).backward(); scaler.step(opt); scaler.update()).  hyper (device, 16 floats):
//   [0] lr  [1] beta1  [2] beta2  [3] eps  [4] weight_decay  [5] grad_scale (1 / world size)  [6] step (float, incremented here)
//   [7] skip flag (caller-owned: capture warm-ups)
//   [8] loss scale S (0: no loss scaling -- gradients are taken as they are)   [9] growth tracker (good steps since the last change)
//   [10] growth interval (2000)   [11] found_inf (set by vpf_grad_check, cleared here)   [12] growth factor (2)   [13] backoff (0.5)
//   [14] number of skipped (overflowed) steps so far   [15] 1: g has been unscaled already (GradScaler.unscale_; cleared here)
// The gradients in `g` carry the factor S; AdamW divides it out.  A step whose gradients hold an inf / NaN is skipped (parameters,
// moments and the step counter stay put) and S is halved; after `growth interval` good steps in a row S doubles.
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             h16_t* __restrict__ shadow, long n, const float* __restrict__ hyper, int zero_grad)
{
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], step = hyper[6] + 1.f;
    const float S = hyper[8];
    const float gs = (S > 0.f && hyper[15] == 0.f) ? hyper[5] / S : hyper[5];
    const bool skip = hyper[7] != 0.f || hyper[11] != 0.f;
    const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float pv = p[i];
        if (!skip) {
            const float gv = g[i] * gs;
            const float mv = b1 * m[i] + (1.f - b1) * gv;
            const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
            m[i] = mv; v[i] = vv;
            pv = pv * (1.f - lr * wd);
            pv -= step_size * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps);
            p[i] = pv;
        }
        if (shadow) shadow[i] = f32_to_h16(pv);
        if (zero_grad) g[i] = 0.f;          // optimizer.zero_grad() of the NEXT step (pretrain.py:174) folded in: no 33 MB fill launch
    }
}
// behind the last AdamW launch of a step: the bias-correction counter and GradScaler.update()
__global__ void adamw_step_kernel(float* hyper)
{
    if (threadIdx.x != 0 || hyper[7] != 0.f) return;                 // (a caller-skipped step changes nothing)
    const bool inf = hyper[11] != 0.f;
    if (!inf) hyper[6] += 1.f;
    if (hyper[8] > 0.f) {
        if (inf) { hyper[8] *= hyper[13]; hyper[9] = 0.f; hyper[14] += 1.f; }
        else {
            hyper[9] += 1.f;
            if (hyper[9] >= hyper[10]) { hyper[8] *= hyper[12]; hyper[9] = 0.f; }
        }
    }
    hyper[11] = 0.f;
    hyper[15] = 0.f;
}
extern "C" int vpf_adamw_step(float* p, float* g, float* m, float* v, void* shadow_h16, long n, float* hyper_dev, int advance_step,
                              void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!p || !g || !m || !v || !hyper_dev) return VPF_ERR_NULL;
    if (n <= 0) return n == 0 ? VPF_OK : VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, p, g, m, v, (h16_t*)shadow_h16, n, hyper_dev, (advance_step >> 1) & 1);
    if (advance_step & 1) hipLaunchKernelGGL(adamw_step_kernel, dim3(1), dim3(64), 0, st, hyper_dev);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// GradScaler's overflow check (torch._amp_foreach_non_finite_check_and_unscale_, pretrain.py:210) over the flat gradient: hyper[11] = 1
// if any element is inf / NaN.  One streaming read (33 MB at c2: ~8 us); every thread that sees one writes the same 1.0f.
__global__ void grad_check_kernel(const float* __restrict__ g, long n, float* __restrict__ hyper)
{
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    bool bad = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 x = g4[i];
        // |x| < inf is false for inf and NaN alike
        bad = bad || !(fabsf(x.x) < INFINITY) || !(fabsf(x.y) < INFINITY) || !(fabsf(x.z) < INFINITY) || !(fabsf(x.w) < INFINITY);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad = bad || !(fabsf(g[n4 * 4 + threadIdx.x]) < INFINITY);
    if (bad) hyper[11] = 1.f;
}
extern "C" int vpf_grad_check(const float* g, long n, float* hyper_dev, void* stream)
{
    (void)hipGetLastError();
    if (!g || !hyper_dev) return VPF_ERR_NULL;
    if (n <= 0) return n == 0 ? VPF_OK : VPF_ERR_BADSHAPE;
    if ((uintptr_t)g & 15) return VPF_ERR_BADALIGN;
    hipLaunchKernelGGL(grad_check_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, g, n, hyper_dev);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}


// =============================================================================== both pre-training losses at once
// pretrain.py:196-204: loss_imid = NTXent(f1, f2), loss_cmid = NTXent((f1 + f2) / 2, g), total = imid + w * cmid, with
// f = [f1; f2] (the two point-cloud views, [2b, D]) and g the image features [b, D].  Three launches forward and two
// backward instead of ~25 tiny kernels that sit alone between the forward and the backward pass.
__global__ void __launch_bounds__(256) loss2_norm_kernel(const float* __restrict__ f, const float* __restrict__ g, int b, int D,
                                                       float* __restrict__ zn, float* __restrict__ inv)
{
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n = 2 * b;
    if (r >= 2 * n) return;
    const int p = r / n, rr = r % n;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float v = p == 0 ? f[(size_t)rr * D + c] : (rr < b ? 0.5f * (f[(size_t)rr * D + c] + f[(size_t)(rr + b) * D + c]) : g[(size_t)(rr - b) * D + c]);
        s += v * v;
    }
    s = wave_sum(s);
    const float iv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < D; c += 64) {
        const float v = p == 0 ? f[(size_t)rr * D + c] : (rr < b ? 0.5f * (f[(size_t)rr * D + c] + f[(size_t)(rr + b) * D + c]) : g[(size_t)(rr - b) * D + c]);
        zn[(size_t)r * D + c] = v * iv;
    }
    if (lane == 0) inv[r] = iv;
}
__global__ void loss2_mean_kernel(const float* __restrict__ rows, int n, float w, float* __restrict__ total, float* __restrict__ parts)
{
    float s0 = 0.f, s1 = 0.f;
    for (int j = threadIdx.x; j < n; j += 64) { s0 += rows[j]; s1 += rows[n + j]; }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (threadIdx.x == 0) { const float a = s0 / (float)n, c = s1 / (float)n; total[0] = a + w * c; parts[0] = a; parts[1] = c; }
}
// df[r] = dzA[r] + dzC[r mod b] / 2   (r < 2b: the views feed problem A directly and problem B through their mean); dg = dzB[b + r]
__global__ void loss2_combine_kernel(const float* __restrict__ dz, int b, int D, float* __restrict__ df, float* __restrict__ dg)
{
    const long nD = (long)2 * b * D;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nD + (long)b * D; i += (long)gridDim.x * blockDim.x) {
        if (i < nD) {
            const long r = i / D; const int c = (int)(i % D);
            df[i] = dz[i] + 0.5f * dz[nD + (r % b) * D + c];
        } else {
            dg[i - nD] = dz[nD + (long)b * D + (i - nD)];
        }
    }
}
extern "C" int vpf_pretrain_loss_fwd(const float* f, const float* g, int b, int D, float temperature, float cmid_weight, float* zn,
                                     float* inv_norm, float* P, float* loss_rows, float* total, float* parts, void* stream)
{
    (void)hipGetLastError();
    if (!f || !g || !zn || !inv_norm || !P || !loss_rows || !total || !parts) return VPF_ERR_NULL;
    if (b <= 0 || D <= 0 || 2 * b > 8192) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int n = 2 * b;
    hipLaunchKernelGGL(loss2_norm_kernel, dim3(vpf_cdiv(2 * n, 4)), dim3(256), 0, st, f, g, b, D, zn, inv_norm);
    hipLaunchKernelGGL(ntxent_row_kernel, dim3(n, 2), dim3(256), sizeof(float) * (D + n), st, (const float*)zn, b, D, 1.f / temperature, P, loss_rows);
    hipLaunchKernelGGL(loss2_mean_kernel, dim3(1), dim3(64), 0, st, (const float*)loss_rows, n, cmid_weight, total, parts);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
extern "C" int vpf_pretrain_loss_bwd(const float* zn, const float* inv_norm, const float* P, int b, int D, float temperature,
                                     float cmid_weight, const float* dtotal, float* ws_dz, float* df, float* dg, void* stream)
{
    (void)hipGetLastError();
    if (!zn || !inv_norm || !P || !dtotal || !ws_dz || !df || !dg) return VPF_ERR_NULL;
    if (b <= 0 || D <= 0 || 2 * b > 8192) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int n = 2 * b;
    hipLaunchKernelGGL(ntxent_bwd_kernel, dim3(n, 2), dim3(256), sizeof(float) * (n + D), st, zn, inv_norm, P, b, D, 1.f / temperature, dtotal,
                       ws_dz, ws_dz + (size_t)b * D, cmid_weight);
    hipLaunchKernelGGL(loss2_combine_kernel, dim3(grid_for((long)3 * b * D, 256)), dim3(256), 0, st, (const float*)ws_dz, b, D, df, dg);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
