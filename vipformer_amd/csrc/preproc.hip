// preproc.hip -- farthest-point sampling, kNN grouping, index_points, square_distance.
// Replaces the aten compositions of vipformer/model/pointcloud/utils.py:6-141 (reference).
// Bit-exact contract: fp32, fixed evaluation order, the ONLY fused multiply-adds are the
// explicit fmaf() calls (this file is compiled with -ffp-contract=off, and the pragma below
// keeps that true if the flag is ever dropped).
#pragma clang fp contract(off)
#include "vpf_common.h"
#include <stdlib.h>

// =============================================================================== FPS
// One workgroup per cloud.  The cloud's xyz is staged once into LDS (SoA, coalesced HBM read:
// the whole cloud is 12 KB at N=1024) for the centroid broadcast; every thread keeps PPT points
// and their running min-distance in registers.  Per iteration: PPT distance updates, a
// wavefront arg-max (64-bit key = distance bits : ~index, so max == farthest, ties -> lowest
// index), one LDS slot per wave, ONE barrier (slots are double-buffered by iteration parity).
#ifndef FPS_STAMP
#define FPS_STAMP(i)
#define FPS_STAMP_INIT
#define FPS_STAMP_FINI
#endif
template <int NT, int PPT>
__global__ void __launch_bounds__(NT) fps_kernel(const float* __restrict__ pts, int N, int C,
                                                const int64_t* __restrict__ start_idx, int G,
                                                int64_t* __restrict__ out_idx)
{
    extern __shared__ float4 smem4[];
    constexpr int NW = NT / 64;
    float4* sp = smem4;                                         // [N] xyz_ : one ds_read_b128 fetches the new centroid
    uint2* slot2 = reinterpret_cast<uint2*>(sp + N);            // [2][NW] per-wave (max distance bits, lowest index)
    int* sel = reinterpret_cast<int*>(slot2 + 2 * NW);          // [G] the selection, written out once at the end

    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* p = pts + (size_t)b * N * C;

    float px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = t * PPT + j;      // a thread owns PPT consecutive points: index order == (wave, lane, j) order
        if (i < N) {
            px[j] = p[(size_t)i * C + 0]; py[j] = p[(size_t)i * C + 1]; pz[j] = p[(size_t)i * C + 2];
            sp[i] = make_float4(px[j], py[j], pz[j], 0.f);
            dist[j] = 1e10f;
        } else {
            px[j] = py[j] = pz[j] = 0.f;
            dist[j] = -1.f;   // padding: min(-1, d >= 0) stays -1 and never beats a real point
        }
    }
    int far = (int)start_idx[b];
    int zero = 0;
    asm volatile("" : "+v"(zero));      // an opaque 0: keeps the slot reads below vector reads (no readfirstlane + scalar branch chain)
    __syncthreads();

    // No global store inside the loop (a barrier would wait for its acknowledgement) and no divergent branch:
    // the loop-carried chain is centroid read -> PPT distance updates -> two DPP reductions -> slot -> barrier -> slot reads.
    FPS_STAMP_INIT
    for (int g = 0; g < G; ++g) {
        if (t == 0) sel[g] = far;
        if (g + 1 == G) break;
        FPS_STAMP(0);
        const float4 c = sp[far];
        float bd = -1.f; uint32_t bi = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float dx = px[j] - c.x, dy = py[j] - c.y, dz = pz[j] - c.z;
            float d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
            const float nd = d < dist[j] ? d : dist[j];
            dist[j] = nd;
            const bool gt = nd > bd;                              // strict >: the lowest index of this thread wins ties
            bd = gt ? nd : bd;
            bi = gt ? (uint32_t)(t * PPT + j) : bi;
        }
        FPS_STAMP(1);
        // wavefront arg-max: the maximum distance, then the lowest lane attaining it (== the lowest index, see the layout)
        const float wm = wave_max_f32_asm(bd);
        const unsigned long long tied = __ballot(bd == wm);
        const uint32_t wi = (uint32_t)__builtin_amdgcn_readlane((int)bi, __builtin_ctzll(tied));
        FPS_STAMP(2);
        uint2* s = slot2 + (g & 1) * NW;
        if (lane == 0) s[wave] = wm < 0.f ? make_uint2(0u, 0xffffffffu) : make_uint2(__float_as_uint(wm), wi);   // wave of padding only
        __syncthreads();
        FPS_STAMP(3);
        // cross-wave merge without a serial chain: lane l takes slot l mod NW, a DPP max over aligned groups of NW lanes,
        // then the lowest tied lane (slots are in wave == index order) hands over its index
        const uint2 v = s[(lane & (NW - 1)) + zero];
        const float gm = row_max_f32_asm<NW>(__uint_as_float(v.x));     // distances are >= +0
        const unsigned long long tied2 = __ballot(__uint_as_float(v.x) == gm);
        far = __builtin_amdgcn_readlane((int)v.y, __builtin_ctzll(tied2));
        FPS_STAMP(4);
    }
    FPS_STAMP_FINI
    __syncthreads();
    for (int g = t; g < G; g += NT) out_idx[(size_t)b * G + g] = sel[g];
}

template <int NT, int PPT>
static int launch_fps(const float* pts, int B, int N, int C, const int64_t* start, int G, int64_t* out, hipStream_t st)
{
    size_t lds = sizeof(float4) * (size_t)N + sizeof(uint2) * 2 * (NT / 64) + sizeof(int) * (size_t)G;
    if (lds > 160 * 1024) return VPF_ERR_BADSHAPE;
    // VPF_FPS_EXCLUSIVE_CU=1 pads the request to the whole 160 KB of LDS (a CU of its own per cloud): the containment used while the
    // mis-sampling beside gemm_kernel workgroups was not understood.  The trigger were the packed-fp32
    // instructions the SLP vectoriser formed in the distance update; this file is now compiled without them (build.py) and the sampler
    // shares its CU again (1 000 of 1 000 launches bit-identical beside dgrad GEMMs either way).
    const int exclusive = vpf_debug().fps_exclusive_cu;
    if (exclusive) lds = 160 * 1024;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)fps_kernel<NT, PPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VPF_ERR_HIP;
    hipLaunchKernelGGL((fps_kernel<NT, PPT>), dim3(B), dim3(NT), lds, st, pts, N, C, start, G, out);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

extern "C" int vpf_fps_f32(const float* pts, int B, int N, int C, const int64_t* start_idx, int G,
                           int64_t* out_idx, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!pts || !start_idx || !out_idx) return VPF_ERR_NULL;
    if (B < 0 || N <= 0 || C < 3 || G < 0 || N > 4096) return VPF_ERR_BADSHAPE;
    if (B == 0 || G == 0) return VPF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 256) return launch_fps<256, 1>(pts, B, N, C, start_idx, G, out_idx, st);
    if (N <= 512) return launch_fps<256, 2>(pts, B, N, C, start_idx, G, out_idx, st);
    // measured (tools/microbench.py preproc): 4 points per thread beats more waves per cloud, the barrier grows with the wave count
    if (N <= 1024) return launch_fps<256, 4>(pts, B, N, C, start_idx, G, out_idx, st);
    if (N <= 2048) return launch_fps<512, 4>(pts, B, N, C, start_idx, G, out_idx, st);
    return launch_fps<1024, 4>(pts, B, N, C, start_idx, G, out_idx, st);
}

// =============================================================================== index_points
__global__ void index_points_kernel(const float* __restrict__ points, int N, int C, const int64_t* __restrict__ idx,
                                    int S, float* __restrict__ out, long total)
{
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const long r = e / C;            // b*S + s
        const long b = r / S;
        const long j = idx[r];
        out[e] = points[((size_t)b * N + j) * C + c];
    }
}
extern "C" int vpf_index_points_f32(const float* points, int B, int N, int C, const int64_t* idx, int S,
                                    float* out, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!points || !idx || !out) return VPF_ERR_NULL;
    if (B < 0 || N <= 0 || C <= 0 || S < 0) return VPF_ERR_BADSHAPE;
    const long total = (long)B * S * C;
    if (total == 0) return VPF_OK;
    int grid = vpf_cdiv(total, 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(index_points_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, points, N, C, idx, S, out, total);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== square_distance
__device__ __forceinline__ float sq3(float x, float y, float z)
{
    float s = x * x;
    s = s + y * y;
    s = s + z * z;
    return s;
}
__device__ __forceinline__ float sqdist3(float s0, float s1, float s2, float sn, float d0, float d1, float d2, float dn)
{
    float m = s0 * d0;
    m = fmaf(s1, d1, m);
    m = fmaf(s2, d2, m);
    float r = -2.0f * m;
    r = r + sn;
    r = r + dn;
    return r;
}
__global__ void square_distance_kernel(const float* __restrict__ src, int Cs, const float* __restrict__ dst, int Cd,
                                       int Ns, int Nd, float* __restrict__ out)
{
    const int b = blockIdx.z, i = blockIdx.y;
    const float* s = src + ((size_t)b * Ns + i) * Cs;
    const float s0 = s[0], s1 = s[1], s2 = s[2];
    const float sn = sq3(s0, s1, s2);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < Nd; j += gridDim.x * blockDim.x) {
        const float* d = dst + ((size_t)b * Nd + j) * Cd;
        const float d0 = d[0], d1 = d[1], d2 = d[2];
        out[((size_t)b * Ns + i) * Nd + j] = sqdist3(s0, s1, s2, sn, d0, d1, d2, sq3(d0, d1, d2));
    }
}
extern "C" int vpf_square_distance_f32(const float* src, int Cs, const float* dst, int Cd, int B, int Ns, int Nd,
                                       float* out, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!src || !dst || !out) return VPF_ERR_NULL;
    if (B < 0 || Ns < 0 || Nd < 0 || Cs < 3 || Cd < 3 || B > 65535 || Ns > 65535) return VPF_ERR_BADSHAPE;
    if (B == 0 || Ns == 0 || Nd == 0) return VPF_OK;
    int gx = vpf_cdiv(Nd, 256); if (gx > 64) gx = 64;
    hipLaunchKernelGGL(square_distance_kernel, dim3(gx, Ns, B), dim3(256), 0, (hipStream_t)stream, src, Cs, dst, Cd, Ns, Nd, out);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== kNN + group
// One workgroup (4 waves) per (cloud, slice of centres).  The cloud is staged once into LDS as
// SoA x/y/z/|p|^2 (coalesced HBM read, conflict-free LDS reads: lane l owns points l, l+64, ...).
// Each wave owns one centre at a time: PPL exact-recipe distances per lane as 64-bit keys
// (order-preserving distance bits : index), then K wave-wide min extractions -> ascending
// distance, ties -> lower index.  Lane k keeps the k-th winner and writes idx / dist / gathered
// neighbour row (with the utils.py:36 member-axis centre subtraction).
__device__ __forceinline__ uint32_t f32_sortable(float f)
{
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float sortable_f32(uint32_t s)
{
    const uint32_t u = s ^ ((s >> 31) ? 0x80000000u : 0xffffffffu);
    return __uint_as_float(u);
}

template <int PPL>
__global__ void __launch_bounds__(256) knn_group_kernel(const float* __restrict__ xyz, int N, int C,
                                                       const float* __restrict__ centers, int Cc, int G, int K,
                                                       int quirk, int centres_per_wg, int64_t* __restrict__ knn_idx,
                                                       float* __restrict__ knn_dist, float* __restrict__ neighbors)
{
    extern __shared__ float smem[];
    float* sx = smem;
    float* sy = sx + N;
    float* sz = sy + N;
    float* sn = sz + N;
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so the
    // parts of ONE cloud are given to block ids that are congruent mod 8 -> one L2 fetches the cloud once.
    // (placement is a speed matter only: any mapping is correct.)
    const int parts = gridDim.x, nclouds = gridDim.y;
    int b, part;
    {
        const int L = blockIdx.y * parts + blockIdx.x;                 // linear id
        const int full = (nclouds / 8) * 8;                            // clouds covered by complete groups of 8
        const int slot = L / 8, xcd = L % 8;
        if (L < full * parts) { b = (slot / parts) * 8 + xcd; part = slot % parts; }
        else { const int r = L - full * parts; b = full + r / parts; part = r % parts; }
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* p = xyz + (size_t)b * N * C;
    for (int i = t; i < N; i += 256) {
        const float x = p[(size_t)i * C + 0], y = p[(size_t)i * C + 1], z = p[(size_t)i * C + 2];
        sx[i] = x; sy[i] = y; sz[i] = z; sn[i] = sq3(x, y, z);
    }
    __syncthreads();

    const int g0 = part * centres_per_wg;
    const int g1 = min(G, g0 + centres_per_wg);
    for (int g = g0 + wave; g < g1; g += 4) {
        const float* c = centers + ((size_t)b * G + g) * Cc;
        const float c0 = c[0], c1 = c[1], c2 = c[2];
        const float cn = sq3(c0, c1, c2);
        unsigned long long key[PPL];
        unsigned long long lmin = ~0ull;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int i = lane + j * 64;
            if (i < N) {
                const float d = sqdist3(c0, c1, c2, cn, sx[i], sy[i], sz[i], sn[i]);
                key[j] = ((unsigned long long)f32_sortable(d) << 32) | (unsigned)i;
            } else {
                key[j] = ~0ull;
            }
            lmin = key[j] < lmin ? key[j] : lmin;
        }
        unsigned long long mine = ~0ull;
        for (int k = 0; k < K; ++k) {
            const unsigned long long w = wave_min_u64(lmin);
            if (lane == k) mine = w;
            // the owner retires the winner and refreshes its local minimum
            unsigned long long nm = ~0ull;
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                key[j] = (key[j] == w) ? ~0ull : key[j];
                nm = key[j] < nm ? key[j] : nm;
            }
            lmin = nm;
        }
        if (lane < K) {
            const int j = (int)(mine & 0xffffffffull);
            const size_t o = ((size_t)b * G + g) * K + lane;
            if (knn_idx) knn_idx[o] = (int64_t)j;
            if (knn_dist) knn_dist[o] = sortable_f32((uint32_t)(mine >> 32));
            if (neighbors) {
                const bool sub = quirk && lane < 3;
                float* dstp = neighbors + o * C;
                const float* srcp = p + (size_t)j * C;
                for (int ch = 0; ch < C; ++ch) {
                    const float v = srcp[ch];
                    dstp[ch] = sub ? (v - c[ch]) : v;
                }
            }
        }
    }
}

// wave-wide sum of a small per-lane count (single-instruction DPP adds, result broadcast through lane 63)
__device__ __forceinline__ uint32_t wave_sum_u32_asm(uint32_t v)
{
    asm volatile("s_nop 4\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// kNN by SELECTION instead of K successive wave-wide minimum extractions (the extraction loop costs ~170 VALU instructions
// per neighbour): the K-th smallest distance T is found by a most-significant-bit-first bisection over the sortable
// distance bits (one wave-wide count per bit, starting below the bits all candidates share), ties at T are resolved
// towards the lowest indices by a second, 12-bit bisection (only when there ARE more ties than open slots), the selected
// keys are compacted through LDS and a 64-lane bitonic sort puts them in the canonical order (ascending distance, ties ->
// lower index).  Same results as knn_group_kernel bit for bit; ~3.5x fewer instructions.
template <int PPL>
__global__ void __launch_bounds__(256) knn_group_select_kernel(const float* __restrict__ xyz, int N, int C,
                                                              const float* __restrict__ centers, int Cc, int G, int K,
                                                              int quirk, int centres_per_wg, int64_t* __restrict__ knn_idx,
                                                              float* __restrict__ knn_dist, float* __restrict__ neighbors)
{
    extern __shared__ float smem[];
    float* sx = smem;
    float* sy = sx + N;
    float* sz = sy + N;
    float* sn = sz + N;
    unsigned long long* sSel = reinterpret_cast<unsigned long long*>(sn + N + (N & 1));     // [4 waves][64] selected keys
    uint32_t* sCnt = reinterpret_cast<uint32_t*>(sSel + 4 * 64);                             // [4] compaction cursors
    const int parts = gridDim.x, nclouds = gridDim.y;
    int b, part;
    {
        const int L = blockIdx.y * parts + blockIdx.x;                 // linear id (XCD-aware mapping, see knn_group_kernel)
        const int full = (nclouds / 8) * 8;
        const int slot = L / 8, xcd = L % 8;
        if (L < full * parts) { b = (slot / parts) * 8 + xcd; part = slot % parts; }
        else { const int r = L - full * parts; b = full + r / parts; part = r % parts; }
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* p = xyz + (size_t)b * N * C;
    for (int i = t; i < N; i += 256) {
        const float x = p[(size_t)i * C + 0], y = p[(size_t)i * C + 1], z = p[(size_t)i * C + 2];
        sx[i] = x; sy[i] = y; sz[i] = z; sn[i] = sq3(x, y, z);
    }
    __syncthreads();

    unsigned long long* mySel = sSel + wave * 64;
    const int g0 = part * centres_per_wg;
    const int g1 = min(G, g0 + centres_per_wg);
    for (int g = g0 + wave; g < g1; g += 4) {
        const float* c = centers + ((size_t)b * G + g) * Cc;
        const float c0 = c[0], c1 = c[1], c2 = c[2];
        const float cn = sq3(c0, c1, c2);
        uint32_t dk[PPL];                       // sortable distance bits; padding = 0xffffffff
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int i = lane + j * 64;
            dk[j] = i < N ? f32_sortable(sqdist3(c0, c1, c2, cn, sx[i], sy[i], sz[i], sn[i])) : 0xffffffffu;
            lo = min(lo, dk[j]);
            if (i < N) hi = max(hi, dk[j]);
        }
        // bits above the highest bit in which min and max differ are shared by every candidate
        const uint32_t wlo = (uint32_t)wave_min_u32_asm(lo);
        uint32_t whi;
        { uint32_t nh = ~hi; nh = wave_min_u32_asm(nh); whi = ~nh; }
        const uint32_t diff = wlo ^ whi;
        int top = diff ? 31 - __builtin_clz(diff) : -1;
        uint32_t T = top >= 0 ? (wlo & ~((2u << top) - 1u)) : wlo;      // common prefix
        if (top == 31) T = 0u;
        for (int bit = top; bit >= 0; --bit) {
            const uint32_t cand = T | (1u << bit);
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < PPL; ++j) cnt += dk[j] < cand ? 1u : 0u;
            if (wave_sum_u32_asm(cnt) < (uint32_t)K) T = cand;
        }
        // T = K-th smallest distance.  c_lt candidates are strictly closer; the remaining slots go to ties, lowest index first
        uint32_t clt = 0, ceq = 0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) { clt += dk[j] < T ? 1u : 0u; ceq += dk[j] == T ? 1u : 0u; }
        clt = wave_sum_u32_asm(clt); ceq = wave_sum_u32_asm(ceq);
        const uint32_t slots = (uint32_t)K - clt;                   // >= 1
        uint32_t ilim = 0xffffffffu;                                // ties with index <= ilim are taken
        if (ceq > slots) {
            uint32_t I = 0;                                         // slots-th smallest index among the ties
            for (int bit = 12; bit >= 0; --bit) {
                const uint32_t cand = I | (1u << bit);
                uint32_t cnt = 0;
#pragma unroll
                for (int j = 0; j < PPL; ++j) cnt += (dk[j] == T && (uint32_t)(lane + j * 64) < cand) ? 1u : 0u;
                if (wave_sum_u32_asm(cnt) < slots) I = cand;
            }
            ilim = I;
        }
        // compaction (order irrelevant: sorted below)
        if (lane == 0) sCnt[wave] = 0;
        uint32_t mycnt = 0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) mycnt += (dk[j] < T || (dk[j] == T && (uint32_t)(lane + j * 64) <= ilim)) ? 1u : 0u;
        uint32_t pos = mycnt ? atomicAdd(&sCnt[wave], mycnt) : 0u;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const uint32_t i = (uint32_t)(lane + j * 64);
            if (dk[j] < T || (dk[j] == T && i <= ilim)) { mySel[pos & 63] = ((unsigned long long)dk[j] << 32) | i; ++pos; }
        }
        unsigned long long key = lane < K ? mySel[lane] : ~0ull;
        // 64-lane bitonic sort, ascending
#pragma unroll
        for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
            for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)key, j2, 64);
                const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), j2, 64);
                const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
                const bool up = (lane & k2) == 0;                    // this block sorts ascending
                const bool lower = (lane & j2) == 0;                 // this lane keeps the smaller of the pair when ascending
                const bool take_min = up == lower;
                const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
                key = take_min ? mn : mx;
            }
        }
        if (lane < K) {
            const int j = (int)(key & 0xffffffffull);
            const size_t o = ((size_t)b * G + g) * K + lane;
            if (knn_idx) knn_idx[o] = (int64_t)j;
            if (knn_dist) knn_dist[o] = sortable_f32((uint32_t)(key >> 32));
            if (neighbors) {
                const bool sub = quirk && lane < 3;
                float* dstp = neighbors + o * C;
                const float* srcp = p + (size_t)j * C;
                for (int ch = 0; ch < C; ++ch) {
                    const float v = srcp[ch];
                    dstp[ch] = sub ? (v - c[ch]) : v;
                }
            }
        }
    }
}

extern "C" int vpf_knn_group_f32(const float* xyz, int B, int N, int C, const float* centers, int Cc, int G, int K,
                                 int apply_ref_axis_quirk, int64_t* knn_idx, float* knn_dist, float* neighbors,
                                 void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (!xyz || !centers) return VPF_ERR_NULL;
    if (B < 0 || N <= 0 || C < 3 || Cc < 3 || G < 0 || K <= 0 || K > 64 || K > N || N > 4096 || B > 65535)
        return VPF_ERR_BADSHAPE;
    if (neighbors && Cc != C) return VPF_ERR_BADSHAPE;
    if (B == 0 || G == 0) return VPF_OK;
    hipStream_t st = (hipStream_t)stream;
    // aim for >= ~1024 workgroups, each re-staging the cloud at most G/4 times
    int cpw = 4;
    while ((long)B * vpf_cdiv(G, cpw) > 2048 && cpw < G) cpw *= 2;
    dim3 grid(vpf_cdiv(G, cpw), B);
    const int sel = vpf_debug().knn_select;
    const size_t lds = sizeof(float) * (4 * (size_t)N + 2) + (sel ? sizeof(unsigned long long) * 4 * 64 + 16 : 0);
#define VPF_KNN_LAUNCH(PPL)                                                                                      \
    if (sel) hipLaunchKernelGGL((knn_group_select_kernel<PPL>), grid, dim3(256), lds, st, xyz, N, C, centers, Cc, G, K,  \
                                apply_ref_axis_quirk, cpw, knn_idx, knn_dist, neighbors);                           \
    else hipLaunchKernelGGL((knn_group_kernel<PPL>), grid, dim3(256), lds, st, xyz, N, C, centers, Cc, G, K,        \
                            apply_ref_axis_quirk, cpw, knn_idx, knn_dist, neighbors)
    if (N <= 256) { VPF_KNN_LAUNCH(4); }
    else if (N <= 512) { VPF_KNN_LAUNCH(8); }
    else if (N <= 1024) { VPF_KNN_LAUNCH(16); }
    else if (N <= 2048) { VPF_KNN_LAUNCH(32); }
    else { VPF_KNN_LAUNCH(64); }
#undef VPF_KNN_LAUNCH
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// =============================================================================== 3-NN inverse-distance weights
// PointNetFeaturePropagation.forward utils.py:219-230: dists = square_distance(xyz1, xyz2); (dists, idx) = sort(dists)[:, :, :3];
// recip = 1 / (dists + 1e-8); weight = recip / sum(recip).  The reference sorts all S distances per point to keep three; here one
// thread per point keeps a running 3-minimum over the centres staged in LDS.  Distances follow the exact square_distance recipe
// (sqdist3), ties -> lower centre index (what a stable sort gives), the weights are the same IEEE operations in the same order
// ((r0 + r1) + r2), so indices and weights are bit-exact against the fp32 reference on tie-free inputs.
__global__ void __launch_bounds__(256) three_nn_kernel(const float* __restrict__ xyz, int C, const float* __restrict__ ctr, int Cc,
                                                       int N, int S, int* __restrict__ idx, float* __restrict__ w)
{
    extern __shared__ float cs[];      // [4][S]: x, y, z, |c|^2
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const float* c = ctr + ((size_t)b * S + j) * Cc;
        const float c0 = c[0], c1 = c[1], c2 = c[2];
        cs[j] = c0; cs[S + j] = c1; cs[2 * S + j] = c2; cs[3 * S + j] = sq3(c0, c1, c2);
    }
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* p = xyz + ((size_t)b * N + n) * C;
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    const float pn = sq3(p0, p1, p2);
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    for (int j = 0; j < S; ++j) {
        const float d = sqdist3(p0, p1, p2, pn, cs[j], cs[S + j], cs[2 * S + j], cs[3 * S + j]);
        if (d < d2) {
            if (d < d1) {
                d2 = d1; i2 = i1;
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            } else { d2 = d; i2 = j; }
        }
    }
    const size_t o = ((size_t)b * N + n) * 3;
    if (S == 1) {       // utils.py:216-217: a single centre is broadcast (points2.repeat): weight exactly 1 on it, nothing else
        idx[o] = 0; idx[o + 1] = 0; idx[o + 2] = 0;
        w[o] = 1.0f; w[o + 1] = 0.0f; w[o + 2] = 0.0f;
        return;
    }
    if (S < 3) { d2 = d1; i2 = i1; }       // S == 2: the reference cannot slice three neighbours (the caller raises); keep the kernel total
    const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);
    float norm = r0 + r1;
    norm = norm + r2;
    idx[o] = i0; idx[o + 1] = i1; idx[o + 2] = i2;
    w[o] = r0 / norm; w[o + 1] = r1 / norm; w[o + 2] = r2 / norm;
}
extern "C" int vpf_three_nn_f32(const float* xyz, int B, int N, int C, const float* centers, int Cc, int S, int* idx, float* weight,
                                void* stream)
{
    (void)hipGetLastError();
    if (!xyz || !centers || !idx || !weight) return VPF_ERR_NULL;
    if (B < 0 || N < 0 || S <= 0 || C < 3 || Cc < 3 || B > 65535) return VPF_ERR_BADSHAPE;
    if (S > 4096) return VPF_ERR_UNSUPPORTED;       // 16 S bytes of LDS for the staged centres: 64 KB without an opt-in (the path uses S <= 128)
    if (B == 0 || N == 0) return VPF_OK;
    hipLaunchKernelGGL(three_nn_kernel, dim3(vpf_cdiv(N, 256), B), dim3(256), sizeof(float) * 4 * S, (hipStream_t)stream, xyz, C, centers, Cc,
                       N, S, idx, weight);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
