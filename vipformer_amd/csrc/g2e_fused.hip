// g2e_fused.hip -- Group2Emb forward (vipformer/model/pointcloud/utils.py:168-189) as two persistent,
// weight-stationary kernels for group_size == 32.
//
// The block is HBM-bound when run as separate GEMMs (every conv reads and writes an [M, 64..256] activation,
// M = batch*groups*32 = 393 216 rows at the benchmark size).  Here one workgroup (8 waves) walks pairs of
// groups (64 rows); the conv weights live in REGISTERS as MFMA B-fragments (each wave owns fixed output
// columns), activations of the pair stay in LDS, and only what BatchNorm's batch statistics force out
// (the pre-BN2 activation h3, bf16) plus what backward needs (a1, h2, max-pool winners) goes to HBM.
//
//   g2e_fwd_a:  x --conv1+BN1+ReLU (BN folded into the 3-tap weights)--> a1 --conv2 (MFMA)--> h2
//               --max over the 32 members--> gmax --conv3 on [gmax | h2] (MFMA, gmax read as an LDS broadcast,
//               the concat never exists)--> h3 ; per-column sum / sum^2 of h3 for BatchNorm-2
//   g2e_fwd_b:  h3 --BN2+ReLU while staging--> conv4 (MFMA) --max over the 32 members in registers--> out, arg
#include "vpf_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define A1LD 72      // a1 tile row stride (bf16): 64 + 8
#define H2LD 136     // h2 tile row stride: 128 + 8
#define H3LD 264     // h3 / a3 tile row stride: 256 + 8

__device__ __forceinline__ bf16x8_t ldfrag(const bf16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ uint32_t bf16_sortable(bf16_t v) { return (v & 0x8000u) ? (uint32_t)(uint16_t)~v : (uint32_t)(v | 0x8000u); }
__device__ __forceinline__ bf16_t bf16_unsortable(uint32_t s) { return (s & 0x8000u) ? (bf16_t)(s & 0x7fffu) : (bf16_t)~s; }

struct G2eA {
    const float* x; long NG; int C;                 // x [NG*32, C] fp32
    const float* w1e; const float* b1e;              // BN1 folded into conv1: [64,C], [64]
    const bf16_t* w2; const float* b2;               // [128,64] bf16, [128]
    const bf16_t* w3; const float* b3;               // [256,256] bf16 ([global | local] columns), [256]
    bf16_t* a1; bf16_t* h2; bf16_t* gmax; uint8_t* arg2; bf16_t* h3;   // outputs
    float* sums;                                     // [512] = sum | sumsq of h3 per column (atomics)
};

__global__ void __launch_bounds__(512) g2e_fwd_a_kernel(G2eA p)
{
    __shared__ __attribute__((aligned(16))) bf16_t sA1[64 * A1LD];
    __shared__ __attribute__((aligned(16))) bf16_t sH2[64 * H2LD];
    __shared__ __attribute__((aligned(16))) bf16_t sG[2 * 128];
    __shared__ __attribute__((aligned(16))) bf16_t sH3[64 * H3LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hl = lane >> 5, l31 = lane & 31;

    // ---- per-thread constants: conv1 (BN folded) for 8 channels of one row
    const int c1row = t >> 3, c1ch = (t & 7) * 8;
    float w1[8][3], b1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        b1[j] = p.b1e[c1ch + j];
#pragma unroll
        for (int i = 0; i < 3; ++i) w1[j][i] = i < p.C ? p.w1e[(c1ch + j) * p.C + i] : 0.f;
    }
    // ---- weight-stationary MFMA B fragments
    // conv2: wave w -> row tile rt2 = w >> 2, column tile ct2 = w & 3 ; B[k][n] = W2[n][k], lane holds n = ct2*32 + l31, k = ks*16 + 8h + j
    const int rt2 = w >> 2, ct2 = w & 3;
    bf16x8_t w2f[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w2f[ks] = ldfrag(p.w2 + (size_t)(ct2 * 32 + l31) * 64 + ks * 16 + 8 * hl);
    const float b2v = p.b2[ct2 * 32 + l31];
    // conv3: wave w -> column tile w (32 of 256 columns), both row tiles
    bf16x8_t w3f[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w3f[ks] = ldfrag(p.w3 + (size_t)(w * 32 + l31) * 256 + ks * 16 + 8 * hl);
    const float b3v = p.b3[w * 32 + l31];
    float ssum = 0.f, ssq = 0.f;

    const long npairs = (p.NG + 1) / 2;
    for (long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const long row0 = pr * 64;                                   // first of 64 rows
        const long nrows = min((long)64, p.NG * 32 - row0);          // 64, or 32 for an odd tail
        // ---- conv1 + BN1 + ReLU -> a1 (LDS + HBM)
        {
            uint4 o = make_uint4(0, 0, 0, 0);
            if (c1row < nrows) {
                const float* xr = p.x + (size_t)(row0 + c1row) * p.C;
                const float x0 = xr[0], x1 = xr[1], x2 = xr[2];
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(w1[j][0] * x0 + w1[j][1] * x1 + w1[j][2] * x2 + b1[j], 0.f);
                o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
                *reinterpret_cast<uint4*>(p.a1 + (size_t)(row0 + c1row) * 64 + c1ch) = o;
            }
            *reinterpret_cast<uint4*>(sA1 + c1row * A1LD + c1ch) = o;
        }
        __syncthreads();
        // ---- conv2 (K = 64) -> h2 tile [32 x 32] of this wave, + bias, bf16 ; group max over the 32 rows
        {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldfrag(sA1 + (rt2 * 32 + l31) * A1LD + ks * 16 + 8 * hl), w2f[ks], acc, 0, 0, 0);
            uint32_t best = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hl;
                const bf16_t hb = f32_to_bf16(acc[r] + b2v);
                sH2[(rt2 * 32 + row) * H2LD + ct2 * 32 + l31] = hb;
                const uint32_t key = (bf16_sortable(hb) << 8) | (uint32_t)(31 - row);      // max value, ties -> first row
                best = key > best ? key : best;
            }
            const uint32_t other = __shfl_xor(best, 32, 64);
            best = other > best ? other : best;
            if (hl == 0) {
                const bf16_t gv = bf16_unsortable(best >> 8);
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
                    const bf16_t* ap = ks < 8 ? (sG + rt * 128 + ks * 16 + 8 * hl)                                   // same global feature for all rows
                                              : (sH2 + (rt * 32 + l31) * H2LD + (ks - 8) * 16 + 8 * hl);
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldfrag(ap), w3f[ks], acc[rt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                    const bf16_t hb = f32_to_bf16(acc[rt][r] + b3v);
                    sH3[row * H3LD + w * 32 + l31] = hb;
                    if (row < nrows) { const float hv = bf16_to_f32(hb); ssum += hv; ssq += hv * hv; }
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
    if (hl == 0) { atomicAdd(p.sums + w * 32 + l31, ssum); atomicAdd(p.sums + 256 + w * 32 + l31, ssq); }
}

struct G2eB {
    const bf16_t* h3; long NG;                       // [NG*32, 256]
    const float* ab2;                                // BN2 as an affine: a[256] | b[256]
    const bf16_t* w4; const float* b4; int Dm;       // [Dm,256] bf16, [Dm]
    float* out; uint8_t* arg4;                       // [NG,Dm]
};

// NT = column tiles of 32 handled per wave (Dm = 256 * NT... = 32 * 8 * NT): Dm in {256, 512}; Dm = 384 -> NT 2 with masking
template <int NT>
__global__ void __launch_bounds__(512) g2e_fwd_b_kernel(G2eB p)
{
    __shared__ __attribute__((aligned(16))) bf16_t sA3[64 * H3LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hl = lane >> 5, l31 = lane & 31;
    bf16x8_t w4f[NT][16];
    float b4v[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int n = (w + 8 * q) * 32 + l31;
        b4v[q] = n < p.Dm ? p.b4[n] : 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            w4f[q][ks] = n < p.Dm ? ldfrag(p.w4 + (size_t)n * 256 + ks * 16 + 8 * hl) : __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
    }
    // staging: thread owns 16-byte chunk column (t & 31) -> BN2 affine for those 8 channels
    float aa[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { aa[j] = p.ab2[(t & 31) * 8 + j]; bb[j] = p.ab2[256 + (t & 31) * 8 + j]; }

    const long npairs = (p.NG + 1) / 2;
    for (long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const long row0 = pr * 64;
        const long nrows = min((long)64, p.NG * 32 - row0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + i * 512, row = c >> 5, ch = c & 31;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < nrows) {
                v = *reinterpret_cast<const uint4*>(p.h3 + (size_t)(row0 + row) * 256 + ch * 8);
                uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    u[j] = pack_bf16x2(fmaxf(fmaf(aa[2 * j], __uint_as_float(u[j] << 16), bb[2 * j]), 0.f),
                                       fmaxf(fmaf(aa[2 * j + 1], __uint_as_float(u[j] & 0xffff0000u), bb[2 * j + 1]), 0.f));
                v = make_uint4(u[0], u[1], u[2], u[3]);
            }
            *reinterpret_cast<uint4*>(sA3 + row * H3LD + ch * 8) = v;
        }
        __syncthreads();
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
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldfrag(sA3 + (rt * 32 + l31) * H3LD + ks * 16 + 8 * hl), w4f[q][ks], acc[rt], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                uint32_t best = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hl;
                    const uint32_t key = (bf16_sortable(f32_to_bf16(acc[rt][r] + b4v[q])) << 8) | (uint32_t)(31 - row);
                    best = key > best ? key : best;
                }
                const uint32_t other = __shfl_xor(best, 32, 64);
                best = other > best ? other : best;
                const long g = pr * 2 + rt;
                if (hl == 0 && g < p.NG && n < p.Dm) {
                    p.out[(size_t)g * p.Dm + n] = bf16_to_f32(bf16_unsortable(best >> 8));
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
extern "C" int vpf_g2e_fold_bn1(const float* W1, const float* b1, const float* ab1, int C, float* w1e, float* b1e, void* stream)
{
    (void)hipGetLastError();
    if (!W1 || !b1 || !ab1 || !w1e || !b1e) return VPF_ERR_NULL;
    if (C <= 0 || C > 8) return VPF_ERR_BADSHAPE;
    hipLaunchKernelGGL(g2e_fold_bn1_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, W1, b1, ab1, C, w1e, b1e);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// x [NG*32, C] -> a1 [M,64], h2 [M,128], gmax [NG,128], arg2 [NG,128], h3 [M,256] (all bf16 / u8), sums512 (zeroed) += column
// sum | sum^2 of h3.  w1e/b1e: first conv with BatchNorm-1 folded in (fp32); w2 [128,64], w3 [256,256] bf16.
extern "C" int vpf_g2e_fwd_a(const float* x, long NG, int C, const float* w1e, const float* b1e, const void* w2_bf16, const float* b2,
                             const void* w3_bf16, const float* b3, void* a1, void* h2, void* gmax, uint8_t* arg2, void* h3,
                             float* sums512_zeroed, void* stream)
{
    (void)hipGetLastError();
    if (!x || !w1e || !b1e || !w2_bf16 || !b2 || !w3_bf16 || !b3 || !a1 || !h2 || !gmax || !arg2 || !h3 || !sums512_zeroed) return VPF_ERR_NULL;
    if (NG <= 0 || C < 3 || C > 3) return VPF_ERR_BADSHAPE;       // xyz groups only (the pre-training path)
    if (((uintptr_t)w2_bf16 & 15) || ((uintptr_t)w3_bf16 & 15) || ((uintptr_t)a1 & 15) || ((uintptr_t)h2 & 15) || ((uintptr_t)h3 & 15)) return VPF_ERR_BADALIGN;
    G2eA p = {x, NG, C, w1e, b1e, (const bf16_t*)w2_bf16, b2, (const bf16_t*)w3_bf16, b3, (bf16_t*)a1, (bf16_t*)h2, (bf16_t*)gmax, arg2,
              (bf16_t*)h3, sums512_zeroed};
    long grid = (NG + 1) / 2; if (grid > 256) grid = 256;
    hipLaunchKernelGGL(g2e_fwd_a_kernel, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// h3 [NG*32,256] bf16, ab2 = BN2 affine (a | b) -> out f32 [NG,Dm] = max over the 32 members of conv4(relu(bn(h3))), arg4 u8
extern "C" int vpf_g2e_fwd_b(const void* h3_bf16, long NG, const float* ab2, const void* w4_bf16, const float* b4, int Dm, float* out,
                             uint8_t* arg4, void* stream)
{
    (void)hipGetLastError();
    if (!h3_bf16 || !ab2 || !w4_bf16 || !b4 || !out || !arg4) return VPF_ERR_NULL;
    if (NG <= 0 || Dm <= 0 || Dm > 512 || (Dm % 32)) return VPF_ERR_BADSHAPE;
    if (((uintptr_t)h3_bf16 & 15) || ((uintptr_t)w4_bf16 & 15)) return VPF_ERR_BADALIGN;
    G2eB p = {(const bf16_t*)h3_bf16, NG, ab2, (const bf16_t*)w4_bf16, b4, Dm, out, arg4};
    long grid = (NG + 1) / 2; if (grid > 256) grid = 256;
    if (Dm <= 256) hipLaunchKernelGGL(g2e_fwd_b_kernel<1>, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(g2e_fwd_b_kernel<2>, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
