// g2e_fused.hip -- Group2Emb forward (vipformer/model/pointcloud/utils.py:168-189) as two persistent,
// weight-stationary kernels for group_size == 32.
//
// The block is HBM-bound when run as separate GEMMs (every conv reads and writes an [M, 64..256] activation,
// M = batch*groups*32 = 393 216 rows at the benchmark size).  Here one workgroup (8 waves) walks pairs of
// groups (64 rows); the conv weights live in REGISTERS as MFMA B-fragments (each wave owns fixed output
// columns), activations of the pair stay in LDS, and only what BatchNorm's batch statistics force out
// (the pre-BN2 activation h3, h16) plus what backward needs (a1, h2, max-pool winners) goes to HBM.
//
//   g2e_fwd_a:  x --conv1+BN1+ReLU (BN folded into the 3-tap weights)--> a1 --conv2 (MFMA)--> h2
//               --max over the 32 members--> gmax --conv3 on [gmax | h2] (MFMA, gmax read as an LDS broadcast,
//               the concat never exists)--> h3 ; per-column sum / sum^2 of h3 for BatchNorm-2
//   g2e_fwd_b:  h3 --BN2+ReLU while staging--> conv4 (MFMA) --max over the 32 members in registers--> out, arg
#include "vpf_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define A1LD 72      // a1 tile row stride (h16): 64 + 8
#define H2LD 136     // h2 tile row stride: 128 + 8
#define H3LD 264     // h3 / a3 tile row stride: 256 + 8

__device__ __forceinline__ h16x8_t ldfrag(const h16_t* p) { return __builtin_bit_cast(h16x8_t, *reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ uint32_t h16_sortable(h16_t v) { return (v & 0x8000u) ? (uint32_t)(uint16_t)~v : (uint32_t)(v | 0x8000u); }
__device__ __forceinline__ h16_t h16_unsortable(uint32_t s) { return (s & 0x8000u) ? (h16_t)(s & 0x7fffu) : (h16_t)~s; }

struct G2eA {
    const float* x; long NG; int C;                 // x [NG*32, C] fp32
    const float* w1e; const float* b1e;              // BN1 folded into conv1: [64,C], [64]
    const h16_t* w2; const float* b2;               // [128,64] h16, [128]
    const h16_t* w3; const float* b3;               // [256,256] h16 ([global | local] columns), [256]
    h16_t* a1; h16_t* h2; h16_t* gmax; uint8_t* arg2; h16_t* h3;   // outputs
    float* sums;                                     // [gridDim.x][512] = per-workgroup sum | sumsq of h3 per column
};

__global__ void __launch_bounds__(512) g2e_fwd_a_kernel(G2eA p)
{
    __shared__ __attribute__((aligned(16))) h16_t sA1[64 * A1LD];
    __shared__ __attribute__((aligned(16))) h16_t sH2[64 * H2LD];
    __shared__ __attribute__((aligned(16))) h16_t sG[2 * 128];
    __shared__ __attribute__((aligned(16))) h16_t sH3[64 * H3LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hl = lane >> 5, l31 = lane & 31;

    // ---- per-thread constants: conv1 (BN folded) for 8 channels of one row
    const int c1row = t >> 3, c1ch = (t & 7) * 8;
    // (kept in LDS as [channel][w0 w1 w2 b]: 32 registers per thread for them made the kernel spill next to the two weight-stationary
    // fragment sets)
    __shared__ float4 sW1[64];
    if (t < 64) sW1[t] = make_float4(p.C > 0 ? p.w1e[t * p.C + 0] : 0.f, p.C > 1 ? p.w1e[t * p.C + 1] : 0.f, p.C > 2 ? p.w1e[t * p.C + 2] : 0.f, p.b1e[t]);
    __syncthreads();
    // ---- weight-stationary MFMA B fragments
    // conv2: wave w -> row tile rt2 = w >> 2, column tile ct2 = w & 3 ; B[k][n] = W2[n][k], lane holds n = ct2*32 + l31, k = ks*16 + 8h + j
    const int rt2 = w >> 2, ct2 = w & 3;
    h16x8_t w2f[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w2f[ks] = ldfrag(p.w2 + (size_t)(ct2 * 32 + l31) * 64 + ks * 16 + 8 * hl);
    const float b2v = p.b2[ct2 * 32 + l31];
    // conv3: wave w -> column tile w (32 of 256 columns), both row tiles
    h16x8_t w3f[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w3f[ks] = ldfrag(p.w3 + (size_t)(w * 32 + l31) * 256 + ks * 16 + 8 * hl);
    const float b3v = p.b3[w * 32 + l31];
    float ssum = 0.f, ssq = 0.f;

    const long npairs = (p.NG + 1) / 2;
    // this thread's input row of the NEXT pair is requested while the current pair is processed (a load consumed at the top of
    // the iteration that issued it exposes one HBM latency per 64 rows)
    float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f;
    auto request = [&](long pq) {
        const long r0 = pq * 64, nr = min((long)64, p.NG * 32 - r0);
        if (pq < npairs && c1row < nr) {
            const float* xr = p.x + (size_t)(r0 + c1row) * p.C;
            nx0 = xr[0]; nx1 = xr[1]; nx2 = xr[2];
        }
    };
    request(blockIdx.x);
    for (long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const long row0 = pr * 64;                                   // first of 64 rows
        const int nrows = (int)min((long)64, p.NG * 32 - row0);          // 64, or 32 for an odd tail
        // ---- conv1 + BN1 + ReLU -> a1 (LDS + HBM)
        {
            uint4 o = make_uint4(0, 0, 0, 0);
            const float x0 = nx0, x1 = nx1, x2 = nx2;
            request(pr + gridDim.x);
            if (c1row < nrows) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float4 wv = sW1[c1ch + j]; v[j] = fmaxf(wv.x * x0 + wv.y * x1 + wv.z * x2 + wv.w, 0.f); }
                o.x = pack_h16x2(v[0], v[1]); o.y = pack_h16x2(v[2], v[3]); o.z = pack_h16x2(v[4], v[5]); o.w = pack_h16x2(v[6], v[7]);
                *reinterpret_cast<uint4*>(p.a1 + (size_t)(row0 + c1row) * 64 + c1ch) = o;
            }
            *reinterpret_cast<uint4*>(sA1 + c1row * A1LD + c1ch) = o;
        }
        __syncthreads();
        // ---- conv2 (K = 64) -> h2 tile [32 x 32] of this wave, + bias, h16 ; group max over the 32 rows
        {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = vpf_mfma32(ldfrag(sA1 + (rt2 * 32 + l31) * A1LD + ks * 16 + 8 * hl), w2f[ks], acc);
            uint32_t best = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hl;
                const h16_t hb = f32_to_h16(acc[r] + b2v);
                sH2[(rt2 * 32 + row) * H2LD + ct2 * 32 + l31] = hb;
                const uint32_t key = (h16_sortable(hb) << 8) | (uint32_t)(31 - row);      // max value, ties -> first row
                best = key > best ? key : best;
            }
            const uint32_t other = __shfl_xor(best, 32, 64);
            best = other > best ? other : best;
            if (hl == 0) {
                const h16_t gv = h16_unsortable(best >> 8);
                sG[rt2 * 128 + ct2 * 32 + l31] = gv;
                const long g = pr * 2 + rt2;
                if (g < p.NG) {
                    p.gmax[(size_t)g * 128 + ct2 * 32 + l31] = gv;
                    p.arg2[(size_t)g * 128 + ct2 * 32 + l31] = (uint8_t)(31 - (best & 0xff));
                }
            }
        }
        __syncthreads();
        // ---- h2 tile -> HBM (16-byte rows), conv3 on [gmax | h2] (K = 256)
        for (int c = t; c < 64 * 16; c += 512) {
            const int row = c >> 4, ch = c & 15;
            if (row < nrows) *reinterpret_cast<uint4*>(p.h2 + (size_t)(row0 + row) * 128 + ch * 8) = *reinterpret_cast<const uint4*>(sH2 + row * H2LD + ch * 8);
        }
        {
            f32x16_t acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const h16_t* ap = ks < 8 ? (sG + rt * 128 + ks * 16 + 8 * hl)                                   // same global feature for all rows
                                              : (sH2 + (rt * 32 + l31) * H2LD + (ks - 8) * 16 + 8 * hl);
                    acc[rt] = vpf_mfma32(ldfrag(ap), w3f[ks], acc[rt]);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                // a pair holds 32 or 64 valid rows (whole groups): the second row tile is valid as a whole or not at all -- one uniform
                // test per tile instead of a compare + select per element (round 4)
                h16_t hbv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                    hbv[r] = f32_to_h16(acc[rt][r] + b3v);
                    sH3[row * H3LD + w * 32 + l31] = hbv[r];
                }
                if (rt == 0 || nrows == 64) {                 // (same additions in the same order as before)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float hv = h16_to_f32(hbv[r]); ssum += hv; ssq += hv * hv; }
                }
            }
        }
        __syncthreads();
        for (int c = t; c < 64 * 32; c += 512) {
            const int row = c >> 5, ch = c & 31;
            if (row < nrows) *reinterpret_cast<uint4*>(p.h3 + (size_t)(row0 + row) * 256 + ch * 8) = *reinterpret_cast<const uint4*>(sH3 + row * H3LD + ch * 8);
        }
        // sA1 / sH2 / sG are rewritten only after the next iteration's first barrier pair; sH3 after its third: safe
    }
    ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
    if (hl == 0) { p.sums[(size_t)blockIdx.x * 512 + w * 32 + l31] = ssum; p.sums[(size_t)blockIdx.x * 512 + 256 + w * 32 + l31] = ssq; }
}

struct G2eB {
    const h16_t* h3; long NG;                       // [NG*32, 256]
    const float* ab2;                                // BN2 as an affine: a[256] | b[256]
    const h16_t* w4; const float* b4; int Dm;       // [Dm,256] h16, [Dm]
    float* out; uint8_t* arg4;                       // [NG,Dm]
};

// NT = column tiles of 32 handled per wave (Dm = 256 * NT... = 32 * 8 * NT): Dm in {256, 512}; Dm = 384 -> NT 2 with masking
template <int NT>
__global__ void __launch_bounds__(512) g2e_fwd_b_kernel(G2eB p)
{
    __shared__ __attribute__((aligned(16))) h16_t sA3[64 * H3LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hl = lane >> 5, l31 = lane & 31;
    h16x8_t w4f[NT][16];
    float b4v[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int n = (w + 8 * q) * 32 + l31;
        b4v[q] = n < p.Dm ? p.b4[n] : 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            w4f[q][ks] = n < p.Dm ? ldfrag(p.w4 + (size_t)n * 256 + ks * 16 + 8 * hl) : __builtin_bit_cast(h16x8_t, make_uint4(0, 0, 0, 0));
    }
    // staging: thread owns 16-byte chunk column (t & 31) -> BN2 affine for those 8 channels
    float aa[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { aa[j] = p.ab2[(t & 31) * 8 + j]; bb[j] = p.ab2[256 + (t & 31) * 8 + j]; }

    const long npairs = (p.NG + 1) / 2;
    // the next pair's h3 rows are requested before this pair's MFMAs (consumed after them): a load issued at the top of an
    // iteration and used at once exposes the whole HBM latency every 64 rows
    uint4 nx[4];
    auto request = [&](long pq) {
        const long r0 = pq * 64, nr = min((long)64, p.NG * 32 - r0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + i * 512, row = c >> 5, ch = c & 31;
            nx[i] = (pq < npairs && row < nr) ? *reinterpret_cast<const uint4*>(p.h3 + (size_t)(r0 + row) * 256 + ch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    request(blockIdx.x);
    for (long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const long row0 = pr * 64;
        const int nrows = (int)min((long)64, p.NG * 32 - row0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + i * 512, row = c >> 5, ch = c & 31;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < nrows) {
                v = nx[i];
                uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    u[j] = pack_h16x2(fmaxf(fmaf(aa[2 * j], h16_lo(u[j]), bb[2 * j]), 0.f),
                                       fmaxf(fmaf(aa[2 * j + 1], h16_hi(u[j]), bb[2 * j + 1]), 0.f));
                v = make_uint4(u[0], u[1], u[2], u[3]);
            }
            *reinterpret_cast<uint4*>(sA3 + row * H3LD + ch * 8) = v;
        }
        __syncthreads();
        request(pr + gridDim.x);
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int n = (w + 8 * q) * 32 + l31;
            f32x16_t acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    acc[rt] = vpf_mfma32(ldfrag(sA3 + (rt * 32 + l31) * H3LD + ks * 16 + 8 * hl), w4f[q][ks], acc[rt]);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                uint32_t best = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hl;
                    const uint32_t key = (h16_sortable(f32_to_h16(acc[rt][r] + b4v[q])) << 8) | (uint32_t)(31 - row);
                    best = key > best ? key : best;
                }
                const uint32_t other = __shfl_xor(best, 32, 64);
                best = other > best ? other : best;
                const long g = pr * 2 + rt;
                if (hl == 0 && g < p.NG && n < p.Dm) {
                    p.out[(size_t)g * p.Dm + n] = h16_to_f32(h16_unsortable(best >> 8));
                    p.arg4[(size_t)g * p.Dm + n] = (uint8_t)(31 - (best & 0xff));
                }
            }
        }
        __syncthreads();
    }
}

// fold BatchNorm-1 (ab1 = a | b from vpf_bn_affine) into the first conv: w1e[c,:] = a[c] W1[c,:], b1e[c] = a[c] b1[c] + b[c]
__global__ void g2e_fold_bn1_kernel(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ ab1, int C,
                                    float* __restrict__ w1e, float* __restrict__ b1e)
{
    const int c = threadIdx.x;
    if (c >= 64) return;
    const float a = ab1[c];
    for (int i = 0; i < C; ++i) w1e[c * C + i] = a * W1[c * C + i];
    b1e[c] = a * b1[c] + ab1[64 + c];
}
// persistent workgroups per launch: one per CU by default (VPF_G2E_GRID overrides, e.g. to leave CUs to a concurrent stream)
static int g2e_max_grid()
{
    const int g = vpf_debug().g2e_grid;
    return g < 1 ? 256 : g;
}
extern "C" int vpf_g2e_fold_bn1(const float* W1, const float* b1, const float* ab1, int C, float* w1e, float* b1e, void* stream)
{
    (void)hipGetLastError();
    if (!W1 || !b1 || !ab1 || !w1e || !b1e) return VPF_ERR_NULL;
    if (C <= 0 || C > 8) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(g2e_fold_bn1_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, W1, b1, ab1, C, w1e, b1e);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// x [NG*32, C] -> a1 [M,64], h2 [M,128], gmax [NG,128], arg2 [NG,128], h3 [M,256] (all h16 / u8); partials [256][512] receives one
// row of column sum | sum^2 of h3 per workgroup (rows >= *nrows_out are not written): fold with vpf_sum_rows_f32 (deterministic).  w1e/b1e: first conv with BatchNorm-1 folded in (fp32); w2 [128,64], w3 [256,256] h16.
extern "C" int vpf_g2e_fwd_a(const float* x, long NG, int C, const float* w1e, const float* b1e, const void* w2_h16, const float* b2,
                             const void* w3_h16, const float* b3, void* a1, void* h2, void* gmax, uint8_t* arg2, void* h3,
                             float* partials_256x512, int* nrows_out, void* stream)
{
    (void)hipGetLastError();
    if (!x || !w1e || !b1e || !w2_h16 || !b2 || !w3_h16 || !b3 || !a1 || !h2 || !gmax || !arg2 || !h3 || !partials_256x512 || !nrows_out) return VPF_ERR_NULL;
    if (NG <= 0 || C < 3 || C > 3) return VPF_ERR_BADSHAPE;       // xyz groups only (the pre-training path)
    if (((uintptr_t)w2_h16 & 15) || ((uintptr_t)w3_h16 & 15) || ((uintptr_t)a1 & 15) || ((uintptr_t)h2 & 15) || ((uintptr_t)h3 & 15)) return VPF_ERR_BADALIGN;
    G2eA p = {x, NG, C, w1e, b1e, (const h16_t*)w2_h16, b2, (const h16_t*)w3_h16, b3, (h16_t*)a1, (h16_t*)h2, (h16_t*)gmax, arg2,
              (h16_t*)h3, partials_256x512};
    long grid = (NG + 1) / 2; if (grid > g2e_max_grid()) grid = g2e_max_grid();
    *nrows_out = (int)grid;
    hipLaunchKernelGGL(g2e_fwd_a_kernel, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// h3 [NG*32,256] h16, ab2 = BN2 affine (a | b) -> out f32 [NG,Dm] = max over the 32 members of conv4(relu(bn(h3))), arg4 u8
extern "C" int vpf_g2e_fwd_b(const void* h3_h16, long NG, const float* ab2, const void* w4_h16, const float* b4, int Dm, float* out,
                             uint8_t* arg4, void* stream)
{
    (void)hipGetLastError();
    if (!h3_h16 || !ab2 || !w4_h16 || !b4 || !out || !arg4) return VPF_ERR_NULL;
    if (NG <= 0 || Dm <= 0 || Dm > 512 || (Dm % 32)) return VPF_ERR_BADSHAPE;
    if (((uintptr_t)h3_h16 & 15) || ((uintptr_t)w4_h16 & 15)) return VPF_ERR_BADALIGN;
    G2eB p = {(const h16_t*)h3_h16, NG, ab2, (const h16_t*)w4_h16, b4, Dm, out, arg4};
    long grid = (NG + 1) / 2; if (grid > g2e_max_grid()) grid = g2e_max_grid();
    if (Dm <= 256) hipLaunchKernelGGL(g2e_fwd_b_kernel<1>, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(g2e_fwd_b_kernel<2>, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== backward (group_size == 32)
// dW4[n,:] += sum_g dout[g,n] * a3[(g, arg4[g,n]), :]   and   db4[n] += sum_g dout[g,n]
// The gradient of the last conv's output is the max-pool gradient: ONE non-zero per (group, column).  A dense wgrad
// over all 32*NG rows does 32x the necessary work; here each thread owns 128 entries of one row of dW4 in registers
// and walks the groups, fetching the winning row of a3 = relu(bn(h3)) from an LDS tile.
struct G2eW4 {
    const h16_t* h3; long NG; const float* ab2;
    const float* dout; const uint8_t* arg4; int Dm;
    float* dW4; float* db4;                     // [Dm,256], [Dm]
};
__global__ void __launch_bounds__(512) g2e_wgrad4_kernel(G2eW4 p)
{
    __shared__ __attribute__((aligned(16))) h16_t sA3[2][32 * H3LD];
    __shared__ float sD[2][256];
    __shared__ uint8_t sR[2][256];
    const int t = threadIdx.x, nl = t & 255, kh = t >> 8;
    const int n = blockIdx.y * 256 + nl;
    float aa[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { aa[j] = p.ab2[(t & 31) * 8 + j]; bb[j] = p.ab2[256 + (t & 31) * 8 + j]; }
    float acc[128];
#pragma unroll
    for (int j = 0; j < 128; ++j) acc[j] = 0.f;
    float accb = 0.f;

    // request = the global loads of a group (registers), commit = BatchNorm affine + ReLU into the LDS buffer: the loads of the
    // NEXT group are in flight while this group's rows are accumulated (issued and consumed back to back they cost one HBM
    // latency per group of 32 rows)
    uint4 hv[2];
    float dn = 0.f;
    uint8_t rn = 0;
    auto request = [&](long g) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + i * 512, row = c >> 5, ch = c & 31;
            hv[i] = *reinterpret_cast<const uint4*>(p.h3 + ((size_t)g * 32 + row) * 256 + ch * 8);
        }
        if (t < 256) {
            const bool ok = n < p.Dm;
            dn = ok ? p.dout[(size_t)g * p.Dm + n] : 0.f;
            rn = ok ? p.arg4[(size_t)g * p.Dm + n] : 0;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + i * 512, row = c >> 5, ch = c & 31;
            uint32_t u[4] = {hv[i].x, hv[i].y, hv[i].z, hv[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                u[j] = pack_h16x2(fmaxf(fmaf(aa[2 * j], h16_lo(u[j]), bb[2 * j]), 0.f),
                                   fmaxf(fmaf(aa[2 * j + 1], h16_hi(u[j]), bb[2 * j + 1]), 0.f));
            *reinterpret_cast<uint4*>(&sA3[buf][row * H3LD + ch * 8]) = make_uint4(u[0], u[1], u[2], u[3]);
        }
        if (t < 256) { sD[buf][t] = dn; sR[buf][t] = rn; }
    };
    long g = blockIdx.x;
    int buf = 0;
    if (g < p.NG) { request(g); commit(0); }
    __syncthreads();
    for (; g < p.NG; g += gridDim.x) {
        const long gn = g + gridDim.x;
        if (gn < p.NG) request(gn);
        const float d = sD[buf][nl];
        const h16_t* row = &sA3[buf][(int)sR[buf][nl] * H3LD + kh * 128];
        if (kh == 0) accb += d;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(row + q * 8);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[q * 8 + 2 * j] = fmaf(d, h16_lo(u[j]), acc[q * 8 + 2 * j]);
                acc[q * 8 + 2 * j + 1] = fmaf(d, h16_hi(u[j]), acc[q * 8 + 2 * j + 1]);
            }
        }
        if (gn < p.NG) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // flush: a wave-wide atomic must cover contiguous bytes (64 lanes in 64 different rows run ~17x slower), so the
    // per-thread rows are transposed through LDS 64 rows of dW4 at a time and added as 256-byte row segments
    float* tile = reinterpret_cast<float*>(&sA3[0][0]);            // 32 x 256 fp32 = 32 KB <= 2 x 32 x H3LD x 2 B
    static_assert(sizeof(h16_t) * 2 * 32 * H3LD >= 32 * 256 * 4, "flush tile must fit in the staging buffers");
    if (kh == 0 && n < p.Dm) atomicAdd(p.db4 + n, accb);
    for (int rr = 0; rr < 8; ++rr) {
        __syncthreads();
        if ((nl >> 5) == rr) {
#pragma unroll
            for (int j = 0; j < 128; ++j) tile[(nl & 31) * 256 + kh * 128 + j] = acc[j];
        }
        __syncthreads();
        for (int e = t; e < 32 * 256; e += 512) {
            const int nn = blockIdx.y * 256 + rr * 32 + (e >> 8);
            if (nn < p.Dm) atomicAdd(p.dW4 + (size_t)nn * 256 + (e & 255), tile[e]);
        }
    }
}
// Round 3: the same walk with the 256 input channels (k) split over FOUR workgroups per walker instead of the rows over 4 x as many
// walkers: a workgroup owns [256 columns n] x [64 k] of dW4 (32 accumulators per thread) and stages the k quarter of FOUR groups per
// step (the same 16 KB tile per step, the same number of steps), so the 256 workgroups of the launch flush 64 KB each instead of
// 256 KB: 16.7 MB of fp32 atomics instead of 67 MB (at the atomic units' ~1.3 TB/s that flush was ~50 us of the kernel's 134).
#define W4_G 4
#define W4_KLD 72
__global__ void __launch_bounds__(512) g2e_wgrad4_kq_kernel(G2eW4 p)
{
    __shared__ __attribute__((aligned(16))) h16_t sA3[2][W4_G * 32 * W4_KLD];
    __shared__ float sD[2][W4_G][256];
    __shared__ uint8_t sR[2][W4_G][256];
    const int t = threadIdx.x, nl = t & 255, kh = t >> 8;
    const int nblk = blockIdx.y >> 2, kq = blockIdx.y & 3;
    const int n = nblk * 256 + nl;
    float aa[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { aa[j] = p.ab2[kq * 64 + (t & 7) * 8 + j]; bb[j] = p.ab2[256 + kq * 64 + (t & 7) * 8 + j]; }
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    float accb = 0.f;
    uint4 hv[2];
    float dn[2];
    uint8_t rn[2];
    auto request = [&](long g0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + i * 512, j = c >> 8, row = (c >> 3) & 31, ch = c & 7;
            const long gg = g0 + j;
            hv[i] = gg < p.NG ? *reinterpret_cast<const uint4*>(p.h3 + ((size_t)gg * 32 + row) * 256 + kq * 64 + ch * 8) : make_uint4(0, 0, 0, 0);
            const int col = c & 255, nn = nblk * 256 + col;
            const bool ok = nn < p.Dm && gg < p.NG;
            dn[i] = ok ? p.dout[(size_t)gg * p.Dm + nn] : 0.f;
            rn[i] = ok ? p.arg4[(size_t)gg * p.Dm + nn] : 0;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = t + i * 512, j = c >> 8, row = (c >> 3) & 31, ch = c & 7;
            uint32_t u[4] = {hv[i].x, hv[i].y, hv[i].z, hv[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                u[q] = pack_h16x2(fmaxf(fmaf(aa[2 * q], h16_lo(u[q]), bb[2 * q]), 0.f),
                                   fmaxf(fmaf(aa[2 * q + 1], h16_hi(u[q]), bb[2 * q + 1]), 0.f));
            *reinterpret_cast<uint4*>(&sA3[buf][(j * 32 + row) * W4_KLD + ch * 8]) = make_uint4(u[0], u[1], u[2], u[3]);
            sD[buf][j][c & 255] = dn[i]; sR[buf][j][c & 255] = rn[i];
        }
    };
    long g0 = (long)blockIdx.x * W4_G;
    const long gstep = (long)gridDim.x * W4_G;
    int buf = 0;
    if (g0 < p.NG) { request(g0); commit(0); }
    __syncthreads();
    for (; g0 < p.NG; g0 += gstep) {
        const long gn = g0 + gstep;
        if (gn < p.NG) request(gn);
#pragma unroll
        for (int j = 0; j < W4_G; ++j) {
            const float d = sD[buf][j][nl];
            const h16_t* row = &sA3[buf][(j * 32 + (int)sR[buf][j][nl]) * W4_KLD + kh * 32];
            if (kh == 0) accb += d;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + q * 8);
                const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[q * 8 + 2 * e] = fmaf(d, h16_lo(u[e]), acc[q * 8 + 2 * e]);
                    acc[q * 8 + 2 * e + 1] = fmaf(d, h16_hi(u[e]), acc[q * 8 + 2 * e + 1]);
                }
            }
        }
        if (gn < p.NG) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // flush: [256 n] x [64 k] through LDS in two halves of 128 rows, added as 256-byte row segments (64 lanes = 64 consecutive k)
    float* tile = reinterpret_cast<float*>(&sA3[0][0]);            // 128 x 64 f32 = 32 KB <= 2 x 128 x 72 x 2 B
    static_assert(sizeof(h16_t) * 2 * W4_G * 32 * W4_KLD >= 128 * 64 * 4, "flush tile must fit in the staging buffers");
    if (kq == 0 && kh == 0 && n < p.Dm) atomicAdd(p.db4 + n, accb);
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((nl >> 7) == half) {
#pragma unroll
            for (int j = 0; j < 32; ++j) tile[(nl & 127) * 64 + kh * 32 + j] = acc[j];
        }
        __syncthreads();
        for (int e = t; e < 128 * 64; e += 512) {
            const int nn = nblk * 256 + half * 128 + (e >> 6);
            if (nn < p.Dm) atomicAdd(p.dW4 + (size_t)nn * 256 + kq * 64 + (e & 63), tile[e]);
        }
    }
}
extern "C" int vpf_g2e_wgrad4(const void* h3_h16, long NG, const float* ab2, const float* dout, const uint8_t* arg4, int Dm,
                              float* dW4, float* db4, void* stream)
{
    (void)hipGetLastError();
    if (!h3_h16 || !ab2 || !dout || !arg4 || !dW4 || !db4) return VPF_ERR_NULL;
    if (NG <= 0 || Dm <= 0) return VPF_ERR_BADSHAPE;
    G2eW4 p = {(const h16_t*)h3_h16, NG, ab2, dout, arg4, Dm, dW4, db4};
    if (vpf_debug().g2e_w4_grid >= 0) {       // (VPF_G2E_W4_GRID < 0: the round-1 kernel below with -grid workgroups)
        const int walkers = vpf_debug().g2e_w4_grid > 0 ? vpf_debug().g2e_w4_grid : 64;
        long gx = vpf_cdiv(NG, (long)W4_G) < walkers ? vpf_cdiv(NG, (long)W4_G) : walkers;
        hipLaunchKernelGGL(g2e_wgrad4_kq_kernel, dim3((unsigned)gx, 4 * vpf_cdiv(Dm, 256)), dim3(512), 0, (hipStream_t)stream, p);
        VPF_CHECK_LAUNCH();
        return VPF_OK;
    }
    const int cap = vpf_debug().g2e_w4_grid < -1 ? -vpf_debug().g2e_w4_grid : 256;      // one workgroup per CU: twice the flush atomics of 128 workgroups, half the walk (168 -> ~110 us)
    long gx = NG < cap ? NG : cap;
    hipLaunchKernelGGL(g2e_wgrad4_kernel, dim3((unsigned)gx, vpf_cdiv(Dm, 256)), dim3(512), 0, (hipStream_t)stream, p);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// dst[c][r] = src[r][c]  (h16; k-strided weight operands of the persistent backward kernels are read from a transposed shadow)
__global__ void transpose_h16_kernel(const h16_t* __restrict__ src, long lds_, int R, int C, h16_t* __restrict__ dst)
{
    __shared__ h16_t tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) { const int r = r0 + i, c = c0 + threadIdx.x; tile[i][threadIdx.x] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : (h16_t)0; }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) { const int c = c0 + i, r = r0 + threadIdx.x; if (c < C && r < R) dst[(size_t)c * R + r] = tile[threadIdx.x][i]; }
}
extern "C" int vpf_transpose_h16(const void* src, long ld, int R, int C, void* dst, void* stream)
{
    (void)hipGetLastError();
    if (!src || !dst) return VPF_ERR_NULL;
    if (R <= 0 || C <= 0) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(transpose_h16_kernel, dim3(vpf_cdiv(C, 32), vpf_cdiv(R, 32)), dim3(32, 8), 0, (hipStream_t)stream, (const h16_t*)src, ld, R, C, (h16_t*)dst);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// g2e_bwd kernels (Dm <= 256): per pair of groups the max-pool gradient tile dh4 [64 x Dm] is rebuilt in LDS from
// (dout, arg4), da3 = dh4 . W4 runs on MFMA with W4^T fragments held in registers, and BatchNorm-2's backward is
// applied in the accumulator layout (column = lane, so the per-channel statistics are per-lane scalars):
//   PASS 0: tmp[c] += sum g, tmp[256+c] += sum g*xhat          (g = da3 * relu'(bn(h3)))
//   PASS 1: dh3 = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M) -> HBM (h16), dgb[g,:] = sum over the 32 members of dh3,
//           dh2 = dh3 . W3[:,128:]  (W3b^T fragments in registers) -> HBM (h16), without the max-pool term of the
//           global feature (added afterwards by vpf_group_max_scatter_add)
struct G2eBwd {
    const float* dout; const uint8_t* arg4; int Dm; long NG;
    const h16_t* h3; const float* stat2; const float* gamma2; const float* beta2;
    const h16_t* w4t;            // [256][Dm]  (W4 transposed)
    const h16_t* w3bt;           // [128][256] (W3[:,128:] transposed)
    float* tmp;                   // [512]
    float invM; int training;
    h16_t* dh3; float* dgb; h16_t* dh2;
    long long* dbg;               // diagnostic: per-phase cycle sums of wave 0 of every workgroup (nullable)
};

// KSC (round 4): the number of 16-wide k-steps of the dh4 . W4 product as a template parameter (16 at Dm = 256; 0 = read Dm at run time).
// With `ks < Dm / 16` tested at run time every k-step was its own basic block -- two LDS reads, a wait for them, two MFMAs, a branch --
// and nothing of step ks + 1 could be issued under step ks (tools/inst_mix.py: 16 x "LLWMWM s_cbranch").
template <int PASS, int KSC>
__global__ void __launch_bounds__(512) g2e_bwd_kernel(G2eBwd p)
{
    extern __shared__ __attribute__((aligned(16))) h16_t smem[];
    h16_t* sD4 = smem;                         // [64][H3LD]  dh4 tile (Dm <= 256), later dh2 staging
    h16_t* sH3 = smem + 64 * H3LD;             // [64][H3LD]  h3, overwritten by dh3
    h16_t* sW3 = sH3 + 64 * H3LD;              // PASS 1: W3b^T in MFMA fragment order, [4 column tiles][16 k-steps][64 lanes] x 16 B = 64 KB
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hl = lane >> 5, l31 = lane & 31;
    if (PASS == 1) {
        // (64 more live registers would spill, and fetched from L2 inside the loop every pair paid that latency in front of
        // its 16 MFMAs: ~4.5 us of a 12 us iteration)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = t + i * 512, f = e >> 6, ln = e & 63, ct = f >> 4, ks = f & 15;
            *reinterpret_cast<uint4*>(sW3 + (size_t)e * 8) =
                *reinterpret_cast<const uint4*>(p.w3bt + (size_t)(ct * 32 + (ln & 31)) * 256 + ks * 16 + 8 * (ln >> 5));
        }
        // (visible to every wave after the first barrier of the loop)
    }
    const int KS4 = KSC ? KSC : p.Dm / 16;      // k-steps of the dh4 . W4 product (<= 16)
    h16x8_t w4f[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
        w4f[ks] = (KSC ? ks < KSC : ks < KS4) ? ldfrag(p.w4t + (size_t)(w * 32 + l31) * p.Dm + ks * 16 + 8 * hl) : __builtin_bit_cast(h16x8_t, make_uint4(0, 0, 0, 0));
    const int col = w * 32 + l31;
    const float mean = p.stat2[col], rstd = p.stat2[256 + col], ga = p.gamma2[col], be = p.beta2[col];
    float sg = 0.f, sgx = 0.f;
    if (PASS == 1 && p.training) { sg = p.tmp[col] * p.invM; sgx = p.tmp[256 + col] * p.invM; }
    // dh2 = dh3 . W3b : wave -> row tile rt2 = w >> 2, column tile ct2 = w & 3 of [64 x 128]
    const int rt2 = w >> 2, ct2 = w & 3;
    float a0 = 0.f, a1 = 0.f;

    const long npairs = (p.NG + 1) / 2;
    uint4 rh[4];
    auto load_h3 = [&](long pr) {
        const long r0 = pr * 64, nr = min((long)64, p.NG * 32 - r0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + i * 512, row = c >> 5, ch = c & 31;
            rh[i] = (pr < npairs && row < nr) ? *reinterpret_cast<const uint4*>(p.h3 + (size_t)(r0 + row) * 256 + ch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    load_h3(blockIdx.x);
    uint2 nar = make_uint2(0xffffffffu, 0xffffffffu);
    float4 nd0 = make_float4(0.f, 0.f, 0.f, 0.f), nd1 = nd0;
    auto load_d4 = [&](long pq) {
        const int gi = t >> 8, ch = (t >> 3) & 31;
        const long g = pq * 2 + gi;
        nar = make_uint2(0xffffffffu, 0xffffffffu);
        nd0 = make_float4(0.f, 0.f, 0.f, 0.f); nd1 = nd0;
        if (ch * 8 < p.Dm && pq < npairs && g < p.NG) {
            const size_t o = (size_t)g * p.Dm + ch * 8;
            nar = *reinterpret_cast<const uint2*>(p.arg4 + o);
            nd0 = *reinterpret_cast<const float4*>(p.dout + o);
            nd1 = *reinterpret_cast<const float4*>(p.dout + o + 4);
        }
    };
    load_d4(blockIdx.x);
    long long ph[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
#define STAMP(i) do { if (p.dbg) { t1 = clock64(); ph[i] += t1 - t0; t0 = t1; } } while (0)
    if (p.dbg) t0 = clock64();
    for (long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const long row0 = pr * 64;
        const int nrows = (int)min((long)64, p.NG * 32 - row0);
        const int nrows_i = nrows;
        // ---- stage dh4 (virtual) : thread = (group gi, 8-channel chunk ch, slice of 4 members)
        {
            const int gi = t >> 8, ch = (t >> 3) & 31, sl = t & 7;
            const long g = pr * 2 + gi;
            (void)g;
            if (ch * 8 < p.Dm) {
                // (winner indices and gradients of this pair were requested during the previous pair's BatchNorm epilogue: read
                // and used at once they cost one HBM latency per pair)
                uint32_t aw[2] = {nar.x, nar.y};
                float dd[8] = {nd0.x, nd0.y, nd0.z, nd0.w, nd1.x, nd1.y, nd1.z, nd1.w};
                // the 8 gradients as h16 pairs once; every member row then keeps a pair element iff it won the max
                uint32_t dp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dp[j] = pack_h16x2(dd[2 * j], dd[2 * j + 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t k = (uint32_t)(sl * 4 + i);
                    uint32_t u[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t e0 = (aw[(2 * j) >> 2] >> (8 * ((2 * j) & 3))) & 0xffu, e1 = (aw[(2 * j + 1) >> 2] >> (8 * ((2 * j + 1) & 3))) & 0xffu;
                        u[j] = (e0 == k ? (dp[j] & 0x0000ffffu) : 0u) | (e1 == k ? (dp[j] & 0xffff0000u) : 0u);
                    }
                    *reinterpret_cast<uint4*>(sD4 + (gi * 32 + k) * H3LD + ch * 8) = make_uint4(u[0], u[1], u[2], u[3]);
                }
            }
        }
        // ---- stage h3 (requested at the end of the previous iteration, ahead of that iteration's stores)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + i * 512, row = c >> 5, ch = c & 31;
            *reinterpret_cast<uint4*>(sH3 + row * H3LD + ch * 8) = rh[i];
        }
        __syncthreads();
        STAMP(0);
        // ---- da3 tile (columns w*32.., both row tiles)
        f32x16_t acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            if (KSC ? ks < KSC : ks < KS4) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    acc[rt] = vpf_mfma32(ldfrag(sD4 + (rt * 32 + l31) * H3LD + ks * 16 + 8 * hl), w4f[ks], acc[rt]);
            }
        STAMP(1);
        load_d4(pr + gridDim.x);
        if (PASS == 0) load_h3(pr + gridDim.x);        // (requested in front of the MFMAs instead: 198 k -> 214 k cycles per workgroup, measured again in round 3)
        float gsum[2] = {0.f, 0.f};
        // one row tile at a time: its 16 h3 values of this lane first (independent LDS reads in flight together), then the
        // math, then the stores (both row tiles at once held 64 registers here and the kernel spilled)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            h16_t hraw[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) hraw[r] = sH3[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl) * H3LD + col];
            h16_t dres[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                const float xh = (h16_to_f32(hraw[r]) - mean) * rstd;
                float g = h16_to_f32(f32_to_h16(acc[rt][r]));            // da3 is a h16 tensor in the unfused path
                // (one select on a 32-bit predicate: written as `a || b` with the 64-bit row count this was a branch, two exec-mask
                // sequences, a 64-bit compare and a scratch reload of the count PER ELEMENT -- 862 of the kernel's 1 511 VALU instructions
                // were moves)
                // Rows beyond the tile's valid rows (the second group of the last pair when NG is odd) need no test: their dh4 rows are
                // all zero, so da3 and g are exactly 0 there (pass 0 adds nothing), and what pass 1 computes for them never reaches HBM
                // (dh3 / dh2 rows and dgb entries of an invalid group are not stored).  Round 4: four VALU instructions per element fewer.
                const bool dead = xh * ga + be <= 0.f;
                g = dead ? 0.f : g;
                if (PASS == 0) { a0 += g; a1 += g * xh; }
                else {
                    const float dv = ga * rstd * (g - sg - xh * sgx);        // (eval: sg = sgx = 0 above -- the same value as ga * rstd * g, without a select per element)
                    const h16_t db = f32_to_h16(dv);
                    dres[r] = db;
                    gsum[rt] += h16_to_f32(db);
                }
            }
            if (PASS == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sH3[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl) * H3LD + col] = dres[r];
            }
        }
        if (PASS == 1) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float s = gsum[rt] + __shfl_xor(gsum[rt], 32, 64);
                const long g = pr * 2 + rt;
                if (hl == 0 && g < p.NG) p.dgb[(size_t)g * 256 + col] = s;
            }
            __syncthreads();                                       // dh3 tile complete; dh4 tile no longer needed
            STAMP(2);
            load_h3(pr + gridDim.x);
            for (int c = t; c < 64 * 32; c += 512) {
                const int row = c >> 5, ch = c & 31;
                if (row < nrows) *reinterpret_cast<uint4*>(p.dh3 + (size_t)(row0 + row) * 256 + ch * 8) = *reinterpret_cast<const uint4*>(sH3 + row * H3LD + ch * 8);
            }
            STAMP(3);
            f32x16_t a2;
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[r] = 0.f;
            // W3b^T fragments from the LDS copy (conflict-free 16-byte reads, lane-contiguous)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                a2 = vpf_mfma32(ldfrag(sH3 + (rt2 * 32 + l31) * H3LD + ks * 16 + 8 * hl),
                                ldfrag(sW3 + ((size_t)(ct2 * 16 + ks) * 64 + lane) * 8), a2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                sD4[row * H2LD + ct2 * 32 + l31] = f32_to_h16(a2[r]);
            }
            __syncthreads();
            STAMP(4);
            for (int c = t; c < 64 * 16; c += 512) {
                const int row = c >> 4, ch = c & 15;
                if (row < nrows) *reinterpret_cast<uint4*>(p.dh2 + (size_t)(row0 + row) * 128 + ch * 8) = *reinterpret_cast<const uint4*>(sD4 + row * H2LD + ch * 8);
            }
        }
        __syncthreads();
        STAMP(5);
    }
    if (p.dbg && t == 0) { for (int i = 0; i < 6; ++i) p.dbg[(size_t)(blockIdx.x * 2 + PASS) * 6 + i] = ph[i]; }
#undef STAMP
    if (PASS == 0) {
        a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64);
        if (hl == 0) { atomicAdd(p.tmp + col, a0); atomicAdd(p.tmp + 256 + col, a1); }
    }
}

// Group2Emb backward through conv4 / BatchNorm-2 / conv3's per-point half for group_size 32, Dm <= 256 (see above).
// tmp512_zeroed: f32 scratch.  Also accumulates dgamma2 / dbeta2.
__global__ void g2e_bn2_param_grad_kernel(const float* __restrict__ tmp, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const int c = threadIdx.x;
    if (c < 256) { atomicAdd(dgamma + c, tmp[256 + c]); atomicAdd(dbeta + c, tmp[c]); }
}
extern "C" int vpf_g2e_bwd(const float* dout, const uint8_t* arg4, int Dm, long NG, const void* h3_h16, const float* stat2,
                           const float* gamma2, const float* beta2, const void* w4t_h16, const void* w3bt_h16, int training,
                           float* tmp512_zeroed, void* dh3_h16, float* dgb, void* dh2_h16, float* dgamma2, float* dbeta2, long long* dbg,
                           void* stream)
{
    (void)hipGetLastError();
    if (!dout || !arg4 || !h3_h16 || !stat2 || !gamma2 || !beta2 || !w4t_h16 || !w3bt_h16 || !tmp512_zeroed || !dh3_h16 || !dgb || !dh2_h16 ||
        !dgamma2 || !dbeta2) return VPF_ERR_NULL;
    if (NG <= 0 || Dm <= 0 || Dm > 256 || (Dm % 16)) return VPF_ERR_BADSHAPE;
    G2eBwd p = {dout, arg4, Dm, NG, (const h16_t*)h3_h16, stat2, gamma2, beta2, (const h16_t*)w4t_h16, (const h16_t*)w3bt_h16, tmp512_zeroed,
                1.0f / (float)(NG * 32), training, (h16_t*)dh3_h16, dgb, (h16_t*)dh2_h16, dbg};
    const size_t lds = sizeof(h16_t) * 2 * 64 * H3LD, lds1 = lds + 64 * 1024;      // pass 1 also keeps W3b^T (64 KB) in LDS
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        for (const void* f : {(const void*)g2e_bwd_kernel<0, 0>, (const void*)g2e_bwd_kernel<0, 16>})
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VPF_ERR_HIP;
        for (const void* f : {(const void*)g2e_bwd_kernel<1, 0>, (const void*)g2e_bwd_kernel<1, 16>})
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    long grid = (NG + 1) / 2; if (grid > g2e_max_grid()) grid = g2e_max_grid();
    hipStream_t st = (hipStream_t)stream;
    if (Dm == 256) {
        if (training) hipLaunchKernelGGL((g2e_bwd_kernel<0, 16>), dim3((unsigned)grid), dim3(512), lds, st, p);
        hipLaunchKernelGGL((g2e_bwd_kernel<1, 16>), dim3((unsigned)grid), dim3(512), lds1, st, p);
    } else {
        if (training) hipLaunchKernelGGL((g2e_bwd_kernel<0, 0>), dim3((unsigned)grid), dim3(512), lds, st, p);
        hipLaunchKernelGGL((g2e_bwd_kernel<1, 0>), dim3((unsigned)grid), dim3(512), lds1, st, p);
    }
    if (training) hipLaunchKernelGGL(g2e_bn2_param_grad_kernel, dim3(1), dim3(256), 0, st, (const float*)tmp512_zeroed, dgamma2, dbeta2);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
