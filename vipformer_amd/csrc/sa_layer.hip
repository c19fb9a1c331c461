// sa_layer.hip -- one self-attention encoder layer (forward) as ONE kernel for gfx950.
//
// Replaces, for SelfAttentionLayer.forward (vipformer/model/pointcloud/partseg.py:170-188 with the
// Residual / MultiHeadAttention / MLP pieces of :14-141,191-213), the chain
//     attention -> o_proj + dropout + residual -> LayerNorm -> fc1 + GELU -> fc2 + dropout + residual
// and, when a next layer follows, that layer's  (+pos) -> LayerNorm -> q/k/v projection  head, so that a stack of
// self-attention layers is one launch per layer instead of nine.  At the reference's sizes (96 or 196 tokens x 256
// channels per sequence, ~12 k tokens per batch) every separate kernel is dominated by its ramp-up and by writing
// and re-reading small activations; here one workgroup owns one sequence chunk and keeps the activations in LDS.
//
// Workgroup = 256 threads (4 waves) = one chunk of a sequence (pc: the 96 tokens; img: 98 of the 196), RB blocks of
// 32 tokens.  All matrix products are "swapped":  C^T[channel, token] = W[channel, :] . X[token, :]
//   A operand = weight fragments, read from HBM/L2 in a PRE-PACKED fragment order (vpf_pack_wfrag: one coalesced
//               1 KB load per 32x16 fragment),
//   B operand = activation rows from LDS (ds_read_b128),
// so a lane owns ONE token (lane & 31) and its registers run over channels: bias / LayerNorm / GELU / dropout /
// residual epilogues are per-lane, LayerNorm statistics are in-register sums + one cross-wave LDS exchange.
// Wave w owns channels [64 w, 64 w + 64) of every 256-channel output chunk.
// Everything the existing backward needs is written exactly as the unfused path writes it (o, lse, x1, LN stats,
// n2, u, h, next base / n1 / qkv), so forward and backward can be fused independently.
#include "vpf_common.h"
#include <stdlib.h>
#include "vipformer_hip.h"
#include "sa_rows.h"

typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define SA_D 256
#define SA_HID 512
#define SA_H 4
#define SA_DH 64
#define ALD 264         // LDS row stride (h16) of a [tokens][256] activation tile (256 + 8 pad)
#define XLD 260         // LDS row stride (f32) of the [tokens][256] residual tile (256 + 4 pad)
#define QLD 776         // LDS row stride (h16) of the [tokens][768] q | k | v staging tile (768 + 8 pad)
#define KLD 72          // LDS row stride (h16) of a [tokens][64] K / V tile
#define LOG2E 1.4426950408889634f
#define LN2F 0.6931471805599453f

__device__ __forceinline__ s16x4_t sa_lds_tr16(const h16_t* p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
}
// see attention.hip: A-operand fragment of V^T with the k slots in accumulator order
__device__ __forceinline__ h16x8_t sa_frag_tr_perm(const h16_t* S, int ld, int kbase, int c0)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, coff = 16 * (g & 1);
    const h16_t* a = S + (kbase + 4 * h + q) * ld + c0 + coff + 4 * p;
    const s16x4_t lo = sa_lds_tr16(a), hi = sa_lds_tr16(a + 8 * ld);
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(h16x8_t, v);
}
__device__ __forceinline__ h16x8_t sa_frag_row(const h16_t* S, int ld, int row0, int kbase)
{
    const int lane = threadIdx.x & 63;
    const uint4 v = *reinterpret_cast<const uint4*>(S + (row0 + (lane & 31)) * ld + kbase + 8 * (lane >> 5));
    return __builtin_bit_cast(h16x8_t, v);
}
__device__ __forceinline__ h16x8_t sa_pack8(const float* f)
{
    uint4 u;
    u.x = pack_h16x2(f[0], f[1]); u.y = pack_h16x2(f[2], f[3]); u.z = pack_h16x2(f[4], f[5]); u.w = pack_h16x2(f[6], f[7]);
    return __builtin_bit_cast(h16x8_t, u);
}
__device__ __forceinline__ float sa_gelu(float x) { return vpf_gelu(x); }

// ------------------------------------------------------------------------------------------------ weight packing
// natural W[N][K] (h16, row-major)  ->  fragment order: frag(cb, ks) = 64 lanes x 8 values,
//   lane l holds W[cb*32 + (l & 31)][ks*16 + 8*(l >> 5) + 0..7];   offset ((cb * (K/16) + ks) * 64 + l) * 8
struct PackJobs { VpfPackJob job[VPF_PACK_MAX_JOBS]; int n; };
__global__ void pack_wfrag_kernel(PackJobs jobs)
{
    const VpfPackJob j = jobs.job[blockIdx.y];
    const long nfrag = (long)(j.N / 32) * (j.K / 16) * 64;      // 16-byte units
    const uint4* src = reinterpret_cast<const uint4*>(j.src);
    uint4* dst = reinterpret_cast<uint4*>(j.dst);
    const int ksn = j.K / 16;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < nfrag; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        const long f = e >> 6;
        const int ks = (int)(f % ksn), cb = (int)(f / ksn);
        const long row = cb * 32 + (l & 31), col = ks * 16 + 8 * (l >> 5);
        if (!j.transposed) {
            dst[e] = src[(row * j.K + col) >> 3];
        } else {
            // logical A[row][col] = src[col * N + row]  (src stored [K][N]: the transposed view of a natural weight)
            const h16_t* s16 = reinterpret_cast<const h16_t*>(j.src);
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[q] = (uint32_t)s16[(col + 2 * q) * (long)j.N + row] | ((uint32_t)s16[(col + 2 * q + 1) * (long)j.N + row] << 16);
            dst[e] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
extern "C" int vpf_pack_wfrag(const VpfPackJob* jobs, int njobs, void* stream)
{
    (void)hipGetLastError();
    if (!jobs) return VPF_ERR_NULL;
    if (njobs <= 0 || njobs > VPF_PACK_MAX_JOBS) return VPF_ERR_BADSHAPE;
    PackJobs pj;
    pj.n = njobs;
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].src || !jobs[i].dst) return VPF_ERR_NULL;
        if (jobs[i].N <= 0 || jobs[i].K <= 0 || (jobs[i].N % 32) || (jobs[i].K % 16)) return VPF_ERR_BADSHAPE;
        if (((uintptr_t)jobs[i].src & 15) || ((uintptr_t)jobs[i].dst & 15)) return VPF_ERR_BADALIGN;
        pj.job[i] = jobs[i];
    }
    hipLaunchKernelGGL(pack_wfrag_kernel, dim3(32, njobs), dim3(256), 0, (hipStream_t)stream, pj);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// ------------------------------------------------------------------------------------------------ building blocks
// The first group (4 k-steps x 2 channel blocks) of a unit's weight fragments.  It is fetched BEFORE the stores of the
// previous phase are issued: vmcnt retires in order, so a load queued behind 48 scattered stores would wait for all of them.
template <int NJ>
struct SaWPre { uint4 w[NJ][8]; };            // groups 0 and 1 (k-steps 0..7) of the wave's NJ channel blocks
template <int NJ>
__device__ __forceinline__ void sa_wprefetch(const h16_t* __restrict__ Wp, int ksn, int ks0, int cb0, SaWPre<NJ>& w)
{
    const int lane = threadIdx.x & 63;
    const uint4* w0 = reinterpret_cast<const uint4*>(Wp) + ((size_t)cb0 * ksn + ks0) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) w.w[j][kk] = w0[((size_t)j * ksn + kk) * 64];
}
// acc[j][i] += Wfrag(cb0 + j, ks0 + ks) . X[i*32.., ks*16..]   for ks = 0..15 (a 256-deep slice of the contraction).
// Weight fragments stream from L2 two groups of 4 k-steps ahead of the MFMAs (the first two groups come from the caller's
// prefetch); the compiler barriers keep the loads from being hoisted further (register pressure next to two live
// accumulator sets).
template <int RB, int NJ>
__device__ __forceinline__ void sa_gemm_unit(const h16_t* __restrict__ Wp, int ksn, int ks0, int cb0, const h16_t* act,
                                             f32x16_t (&acc)[NJ][RB], const SaWPre<NJ>& pre)
{
    const int lane = threadIdx.x & 63;
    const uint4* w0 = reinterpret_cast<const uint4*>(Wp) + ((size_t)cb0 * ksn + ks0) * 64 + lane;
    uint4 wl[NJ][8];                    // groups 2 and 3
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wl[j][kk] = w0[((size_t)j * ksn + 8 + kk) * 64];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        h16x8_t af[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) af[j] = __builtin_bit_cast(h16x8_t, ks < 8 ? pre.w[j][ks & 7] : wl[j][ks & 7]);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const h16x8_t x = sa_frag_row(act, ALD, i * 32, ks * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j][i] = vpf_mfma32(af[j], x, acc[j][i]);
        }
    }
}
template <int RB, int NJ>
__device__ __forceinline__ void sa_zero(f32x16_t (&acc)[NJ][RB])
{
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
}
// Accumulator element (j, i, r) of wave w <-> token i*32 + (lane & 31), channel 64 w + 32 j + 8 (r >> 2) + 4 (lane >> 5) + (r & 3):
// a lane holds groups of 4 consecutive channels (j, g = r >> 2).

// per-token sum over the 256 channels of v(j,i,r): in-lane over registers, lane ^ 32, then the 4 waves through LDS
template <int RB, int NJ>
__device__ __forceinline__ void sa_token_sum(float (&part)[RB], float* sStat, float (&tot)[RB])
{
    constexpr int NWV = 8 / NJ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        part[i] += __shfl_xor(part[i], 32, 64);
        if (lane < 32) sStat[(i * 32 + lane) * NWV + wave] = part[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float t = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < NWV; w4 += 4) {
            const float4 v = *reinterpret_cast<const float4*>(sStat + (i * 32 + (lane & 31)) * NWV + w4);
            t += (v.x + v.y) + (v.z + v.w);
        }
        tot[i] = t;
    }
}

// LayerNorm over the channel axis of the values in acc, in place: acc <- (acc - mean) * rstd * gamma + beta.
// Every wave first reduces ITS channels of a token to (mean, centred second moment) -- two lane^32 shuffles -- and the
// waves' pairs are merged exactly (Chan et al.): one LDS exchange and one barrier, and no E[x^2] - mean^2 cancellation.
// mean / rstd of every token are returned (all lanes of the token agree).
template <int RB, int NJ>
__device__ __forceinline__ void sa_layernorm(f32x16_t (&acc)[NJ][RB], const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* sStatA, float* sStatB, float (&mean)[RB], float (&rstd)[RB])
{
    constexpr int NWV = 8 / NJ, CPW = 32 * NJ;          // waves, channels per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5;
    float2* sPair = reinterpret_cast<float2*>(sStatA);   // [TOK][NWV] (mean_w, M2_w); sStatA and sStatB are contiguous
    (void)sStatB;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[j][i][r];
        s += __shfl_xor(s, 32, 64);
        const float mw = s * (1.0f / CPW);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[j][i][r] - mw; q += d * d; }
        q += __shfl_xor(q, 32, 64);
        if (lane < 32) sPair[(i * 32 + lane) * NWV + wave] = make_float2(mw, q);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float2 pw[NWV];
        float ms = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { pw[w] = sPair[(i * 32 + (lane & 31)) * NWV + w]; ms += pw[w].x; }
        const float mu = ms * (1.0f / NWV);
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { const float d = pw[w].x - mu; m2 += pw[w].y + (float)CPW * d * d; }
        mean[i] = mu;
        rstd[i] = rsqrtf(m2 * (1.0f / SA_D) + 1e-5f);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
            const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][i][4 * g + q] = (acc[j][i][4 * g + q] - mean[i]) * rstd[i] * gg[q] + bb[q];
        }
}

// store the accumulator tile as h16 into an LDS activation tile [tokens][ALD] (column offset col0) and, for valid tokens,
// to a global [M][ld] matrix (column offset gcol0)
template <int RB, int NJ>
__device__ __forceinline__ void sa_store_h16(const f32x16_t (&acc)[NJ][RB], h16_t* sAct, int col0, h16_t* __restrict__ G, long ld,
                                              int gcol0, long m0, int nvalid)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = pack_h16x2(acc[j][i][4 * g + 0], acc[j][i][4 * g + 1]);
                u.y = pack_h16x2(acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
                const int c = 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                const int tok = i * 32 + t;
                if (sAct) *reinterpret_cast<uint2*>(sAct + tok * ALD + col0 + c) = u;
                if (G && tok < nvalid) *reinterpret_cast<uint2*>(G + (size_t)(m0 + tok) * ld + gcol0 + c) = u;
            }
}

// The same tile, row-coalesced: every thread moves 16 B (8 channels) of a row, a wave-instruction writes two whole 512-byte rows.
// (The accumulator layout above makes a wave-instruction touch 32 rows with 16 bytes each: ~5x the cycles in the address unit.)
template <int TOK, int NT>
__device__ __forceinline__ void sa_tile_store_rows(const h16_t* sAct, h16_t* __restrict__ G, long ld, int gcol0, long m0, int nvalid)
{
#pragma unroll
    for (int it = 0; it < TOK * 32 / NT; ++it) {
        const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
        const uint4 v = *reinterpret_cast<const uint4*>(sAct + row * ALD + ch * 8);
        if (row < nvalid) *reinterpret_cast<uint4*>(G + (size_t)(m0 + row) * ld + gcol0 + ch * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------ the layer kernel
// one (head, 32-query block) attention unit of a wave; K / V tiles of the head are in LDS, LPT = padded sequence length
template <int LPT>
__device__ __forceinline__ void sa_attn_unit(const h16_t* sK, const h16_t* sV, const h16x8_t (&qf)[4], int L, float c, const VpfRng& rng,
                                             bool drop, uint64_t rbase, f32x16_t (&o)[2], float& m, float& l)
{
    const int hl = (threadIdx.x & 63) >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    m = -INFINITY; l = 0.f;
#pragma unroll
    for (int kv0 = 0; kv0 < LPT; kv0 += 32) {
        if (kv0 >= L) break;
        f32x16_t s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s = vpf_mfma32(sa_frag_row(sK, KLD, kv0, ks * 16), qf[ks], s);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
            s[r] = kv < L ? s[r] * c : -INFINITY;
            tmax = fmaxf(tmax, s[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float alpha = vpf_exp2(m - mn);
        m = mn;
        float ps = 0.f;
        float pv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const uint32_t keep = drop ? vpf_keep4_at(rng, rbase + (uint64_t)(kv0 + 8 * g4 + 4 * hl)) : 15u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                const float pr = vpf_exp2(s[r] - mn);
                ps += pr;
                pv[r] = drop ? (((keep >> e) & 1u) ? pr * rng.scale : 0.f) : pr;
            }
        }
        l = l * alpha + ps;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const h16x8_t pf = sa_pack8(pv + 8 * s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                o[dt] = vpf_mfma32(sa_frag_tr_perm(sV, KLD, kv0 + 16 * s2, dt * 32), pf, o[dt]);
        }
    }
}

// ATT = false: the attention output o is an INPUT (vpf_attention_fwd ran before); a workgroup then owns any RB*32
// consecutive rows of the [B*L, D] token matrix, needs no K / V tiles and two workgroups fit on a CU.
template <int RB, int HPR, int LPT, bool ATT, int NJ>
__global__ void __launch_bounds__(64 * (8 / NJ)) sa_layer_fwd_kernel(VpfSaLayerFwd a)
{
    constexpr int NWV = 8 / NJ, NT = 64 * NWV;          // waves per workgroup: each owns NJ blocks of 32 channels per 256-wide chunk
    static_assert(!ATT || NJ == 2, "the in-kernel attention distributes its units over 4 waves");
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    constexpr int TOK = RB * 32;
    constexpr int UPW = HPR * RB / 4;                 // attention units per wave and round
    static_assert(HPR * RB % 4 == 0, "units must divide over the 4 waves");
    h16_t* actA = lds;                               // [TOK][ALD]   o -> n2 -> next n1
    h16_t* reg1 = lds + TOK * ALD;                   // K/V tiles during attention, then actH + LayerNorm exchange
    h16_t* actH = reg1;                              // [TOK][ALD]   one 256-wide chunk of the hidden activation
    float* sStatA = reinterpret_cast<float*>(reg1 + TOK * ALD);   // [TOK][NWV]
    float* sStatB = sStatA + TOK * NWV;
    // XR (no attention inside, one channel block per wave): the f32 residual stream of the workgroup's rows lives in LDS
    // (base -> x1 -> x1 + pos -> out) and crosses HBM only in row-coalesced passes: in the accumulator layout a wave-instruction
    // touches 32 rows with 16 .. 32 bytes each, which costs the address unit far more than the bytes do.
    constexpr bool XR = !ATT && NJ == 1;
    float* xres = sStatB + TOK * NWV;                 // [TOK][XLD] f32 (XR only)
    // XR: the per-channel vectors of the epilogues (bo | ln2 gamma | ln2 beta | b1 [512] | b2 | next ln1 gamma | beta = 2048 floats)
    // are fetched once at the start (one 16-byte load per thread) and read from LDS: fetched where they are used, every
    // epilogue and LayerNorm starts with an L2 round trip that nothing overlaps
    float* sPar = xres + TOK * XLD;
    const float* bo_p = XR ? sPar : a.bo;
    const float* g2_p = XR ? sPar + 256 : a.ln2_g;
    const float* be2_p = XR ? sPar + 512 : a.ln2_b;
    const float* b1_p = XR ? sPar + 768 : a.b1;
    const float* b2_p = XR ? sPar + 1280 : a.b2;
    const float* g1n_p = XR ? sPar + 1536 : a.ln1n_g;
    const float* be1n_p = XR ? sPar + 1792 : a.ln1n_b;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int L = a.L;
    const long mseq = (long)b * L;                    // first row of the sequence
    const long m0 = ATT ? mseq + (long)chunk * a.chunk_rows : (long)blockIdx.x * TOK;   // first row of this workgroup
    // row of the positional table for block row r: (m0 + r) % pos_rows.  The 64-bit modulo is taken ONCE per workgroup (m0 is uniform);
    // per row it is an add and -- the table has at least as many rows as a block in every shipped configuration -- one conditional
    // subtraction (as `(m0 + row) % pos_rows` it was a 64-bit software division in front of each of a thread's 10 positional loads)
    const int pos_m0 = a.pos ? (int)(m0 % (long)a.pos_rows) : 0;
    auto pos_row = [&](int r) -> int {
        int pr = pos_m0 + r;
        if (a.pos_rows >= TOK) return pr >= a.pos_rows ? pr - a.pos_rows : pr;
        return pr % a.pos_rows;
    };
    const int nvalid = ATT ? min(a.chunk_rows, L - chunk * a.chunk_rows) : (int)min((long)TOK, (long)a.B * L - m0);
    const h16_t* qkv = (const h16_t*)a.qkv;
    long long t0_ = 0, t1_;
    int ph_ = 0;
#define SA_STAMP() do { if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { t1_ = clock64(); a.dbg[ph_++] = t1_ - t0_; t0_ = t1_; } } while (0)
    if (a.dbg) t0_ = clock64();

    SaWPre<NJ> wpre;
    if constexpr (!ATT) {
        // stage this workgroup's rows of the attention output (coalesced 16-byte loads)
        sa_wprefetch((const h16_t*)a.Wo, SA_D / 16, 0, NJ * wave, wpre);
        constexpr int CPT = TOK * 32 / NT;
        uint4 r[CPT];
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
            r[it] = row < nvalid ? *reinterpret_cast<const uint4*>((const h16_t*)a.o + (size_t)(m0 + row) * SA_D + ch * 8) : make_uint4(0, 0, 0, 0);
        }
        if constexpr (XR) {                                        // the residual base, row-coalesced, into xres
            constexpr int XPT = TOK * 64 / NT;
            {
                const float* src = wave == 0 ? a.bo : wave == 1 ? a.ln2_g : wave == 2 ? a.ln2_b : wave == 3 ? a.b1 : wave == 4 ? a.b1 + 256
                                 : wave == 5 ? a.b2 : wave == 6 ? a.ln1n_g : a.ln1n_b;
                const float4 pz = src ? *reinterpret_cast<const float4*>(src + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(sPar + wave * 256 + lane * 4) = pz;
            }
            float4 rb[XPT];
#pragma unroll
            for (int it = 0; it < XPT; ++it) {
                const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
                rb[it] = row < nvalid ? *reinterpret_cast<const float4*>(a.base + (size_t)(m0 + row) * SA_D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int it = 0; it < XPT; ++it) {
                const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
                *reinterpret_cast<float4*>(xres + row * XLD + c4 * 4) = rb[it];
            }
        }
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
            *reinterpret_cast<uint4*>(actA + row * ALD + ch * 8) = r[it];
        }
    }
    // ============================================================ attention (heads in rounds of HPR)
    if constexpr (ATT) {
        const float c = a.scale * LOG2E;
        const VpfRng rng = vpf_rng_init(a.rng, a.site_att, a.p_att);
        const bool drop = a.p_att > 0.f;
        constexpr int NCH = HPR * 2 * LPT * 8;                       // 16-byte chunks of a round's K / V tiles
        constexpr int CPT = (NCH + 255) / 256;
#pragma unroll 1
        for (int round = 0; round < SA_H / HPR; ++round) {
            if (round) __syncthreads();                       // every wave is done with the previous round's K / V
            // stage K_h, V_h [L][64] of the round's heads (zero rows up to LPT): all loads in flight at once
            uint4 kvr[CPT];
#pragma unroll
            for (int it = 0; it < CPT; ++it) {
                const int e = threadIdx.x + it * 256;
                const int ch = e & 7, row = (e >> 3) % LPT, kv = ((e >> 3) / LPT) & 1, slot = (e >> 3) / (2 * LPT);
                const int hd = round * HPR + slot;
                kvr[it] = make_uint4(0, 0, 0, 0);
                if (e < NCH && row < L) kvr[it] = *reinterpret_cast<const uint4*>(qkv + (size_t)(mseq + row) * (3 * SA_D) + (1 + kv) * SA_D + hd * SA_DH + ch * 8);
            }
            // the query fragments of this wave's units
            h16x8_t qf[UPW][4];
#pragma unroll
            for (int uu = 0; uu < UPW; ++uu) {
                const int u = wave + 4 * uu, slot = u / RB, rb = u % RB, hd = round * HPR + slot;
                const int tok = rb * 32 + t;
                const bool qok = tok < nvalid;
                const h16_t* qp = qkv + (size_t)(m0 + (qok ? tok : 0)) * (3 * SA_D) + hd * SA_DH + 8 * hl;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    qf[uu][ks] = __builtin_bit_cast(h16x8_t, qok ? *reinterpret_cast<const uint4*>(qp + ks * 16) : make_uint4(0, 0, 0, 0));
            }
#pragma unroll
            for (int it = 0; it < CPT; ++it) {
                const int e = threadIdx.x + it * 256;
                const int ch = e & 7, row = (e >> 3) % LPT, kv = ((e >> 3) / LPT) & 1, slot = (e >> 3) / (2 * LPT);
                if (e < NCH) *reinterpret_cast<uint4*>(reg1 + ((slot * 2 + kv) * LPT + row) * KLD + ch * 8) = kvr[it];
            }
            __syncthreads();
            if (round + 1 == SA_H / HPR) sa_wprefetch((const h16_t*)a.Wo, SA_D / 16, 0, NJ * wave, wpre);   // ahead of the o stores
#pragma unroll
            for (int uu = 0; uu < UPW; ++uu) {
                const int u = wave + 4 * uu, slot = u / RB, rb = u % RB, hd = round * HPR + slot;
                const h16_t* sK = reg1 + (slot * 2) * LPT * KLD;
                const h16_t* sV = sK + LPT * KLD;
                const int tok = rb * 32 + t;
                const bool qok = tok < nvalid;
                const int q = chunk * a.chunk_rows + tok;             // query index inside the sequence
                const int bh = b * SA_H + hd;
                const uint64_t rbase = ((uint64_t)bh * L + (uint64_t)(qok ? q : 0)) * (uint64_t)L;
                f32x16_t o[2];
                float m, l;
                sa_attn_unit<LPT>(sK, sV, qf[uu], L, c, rng, drop, rbase, o, m, l);
                const float lt = l + __shfl_xor(l, 32, 64);
                const float inv = qok ? 1.f / lt : 0.f;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        uint2 w;
                        w.x = pack_h16x2(o[dt][4 * gq + 0] * inv, o[dt][4 * gq + 1] * inv);
                        w.y = pack_h16x2(o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv);
                        const int cc = hd * SA_DH + dt * 32 + 8 * gq + 4 * hl;
                        *reinterpret_cast<uint2*>(actA + tok * ALD + cc) = w;
                        if (qok) *reinterpret_cast<uint2*>((h16_t*)a.o + (size_t)(m0 + tok) * SA_D + cc) = w;
                    }
                if (qok && hl == 0) a.lse[(size_t)bh * L + q] = (m + log2f(lt)) * LN2F;
            }
        }
    }
    __syncthreads();                                           // o complete in actA; K / V dead
    SA_STAMP();     // 0: attention

    f32x16_t acc[NJ][RB];
    // ============================================================ x1 = base + dropout(o . Wo^T + bo);  n2 = LN2(x1)
    {
        // the residual base of every element this lane owns: issued before the GEMM, consumed after it
        float4 res[XR ? 1 : NJ][XR ? 1 : 4][XR ? 1 : RB];
        if constexpr (!XR) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int tok = i * 32 + t;
                    const bool ok = tok < nvalid;
                    const size_t off = (size_t)(m0 + (ok ? tok : 0)) * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                    res[j][g][i] = ok ? *reinterpret_cast<const float4*>(a.base + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
        sa_zero<RB, NJ>(acc);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.Wo, SA_D / 16, 0, NJ * wave, actA, acc, wpre);
        sa_wprefetch((const h16_t*)a.W1, SA_D / 16, 0, NJ * wave, wpre);           // fc1 chunk 0, ahead of the x1 / n2 stores
        SA_STAMP();     // 1: o_proj MFMA
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res1, a.p_res1);
        const bool drop = a.p_res1 > 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cch = 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                const float4 bo = *reinterpret_cast<const float4*>(bo_p + cch);
                const float bb[4] = {bo.x, bo.y, bo.z, bo.w};
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from interleaving all the unrolled hash chains (spills)
                    const int tok = i * 32 + t;
                    const bool ok = tok < nvalid;
                    const size_t off = (size_t)(m0 + (ok ? tok : 0)) * SA_D + cch;
                    float4 rv;
                    if constexpr (XR) rv = *reinterpret_cast<const float4*>(xres + tok * XLD + cch);
                    else rv = res[j][g][i];
                    const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
                    float v[4];
                    const uint32_t keep = drop ? vpf_keep4(rng, (uint64_t)off >> 2) : 15u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float y = acc[j][i][4 * g + q] + bb[q];
                        if (drop) y = ((keep >> q) & 1u) ? y * rng.scale : 0.f;
                        v[q] = rr[q] + y;
                        acc[j][i][4 * g + q] = v[q];
                    }
                    if constexpr (XR) *reinterpret_cast<float4*>(xres + tok * XLD + cch) = make_float4(v[0], v[1], v[2], v[3]);
                    else if (ok) *reinterpret_cast<float4*>(a.x1 + off) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        SA_STAMP();     // 2: dropout + residual epilogue
        float mean[RB], rstd[RB];
        sa_layernorm<RB, NJ>(acc, g2_p, be2_p, sStatA, sStatB, mean, rstd);
        SA_STAMP();     // 3a: LayerNorm 2 maths
        if (wave == 0 && hl == 0) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (i * 32 + t < nvalid) { a.mean2[m0 + i * 32 + t] = mean[i]; a.rstd2[m0 + i * 32 + t] = rstd[i]; }
        }
        // (the LayerNorm exchange barriers guarantee every wave has finished reading o from actA)
        sa_store_h16<RB, NJ>(acc, actA, 0, XR ? nullptr : (h16_t*)a.n2, SA_D, 0, m0, nvalid);
        SA_STAMP();     // 3b: n2 stores
    }
    __syncthreads();                                           // n2 complete in actA
    if constexpr (XR) {
        // x1 leaves for HBM row-coalesced; the positional term of the next base is added on the way: xres = x1 + pos
        constexpr int XPT = TOK * 64 / NT;
        float4 pv[XPT];
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            pv[it] = (a.pos && row < nvalid) ? *reinterpret_cast<const float4*>(a.pos + (size_t)pos_row(row) * SA_D + c4 * 4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        sa_tile_store_rows<TOK, NT>(actA, (h16_t*)a.n2, SA_D, 0, m0, nvalid);
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            float4 v = *reinterpret_cast<const float4*>(xres + row * XLD + c4 * 4);
            if (row < nvalid) *reinterpret_cast<float4*>(a.x1 + (size_t)(m0 + row) * SA_D + c4 * 4) = v;
            v.x += pv[it].x; v.y += pv[it].y; v.z += pv[it].z; v.w += pv[it].w;
            *reinterpret_cast<float4*>(xres + row * XLD + c4 * 4) = v;
        }
    }
    SA_STAMP();     // 3c: barrier

    // ============================================================ MLP: two 256-wide chunks of the hidden layer
    f32x16_t acc2[NJ][RB];
    sa_zero<RB, NJ>(acc2);
    // the final epilogue's side inputs (x1, written by this very thread above, and pos): with one channel block per wave there
    // are registers to fetch them BEHIND the last GEMM unit instead of in front of the epilogue
    float4 resx[XR ? 1 : NJ][XR ? 1 : 4][XR ? 1 : RB], resp[XR ? 1 : NJ][XR ? 1 : 4][XR ? 1 : RB];
    int prow[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) prow[i] = a.pos ? pos_row(i * 32 + t) : 0;
    auto load_final = [&]() {
        if constexpr (XR) return;
        else {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int tok = i * 32 + t;
                    const bool ok = tok < nvalid;
                    const int cch = 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                    resx[j][g][i] = ok ? *reinterpret_cast<const float4*>(a.x1 + (size_t)(m0 + tok) * SA_D + cch) : make_float4(0.f, 0.f, 0.f, 0.f);
                    resp[j][g][i] = (ok && a.pos) ? *reinterpret_cast<const float4*>(a.pos + (size_t)prow[i] * SA_D + cch)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
    };
#pragma unroll
    for (int hc = 0; hc < SA_HID / SA_D; ++hc) {
        sa_zero<RB, NJ>(acc);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.W1, SA_D / 16, 0, hc * 8 + NJ * wave, actA, acc, wpre);
        sa_wprefetch((const h16_t*)a.W2, SA_HID / 16, hc * 16, NJ * wave, wpre);   // this chunk's fc2 slice, ahead of the u / h stores
        if (hc == 1) SA_STAMP();    // 4a: (chunk 0 and) fc1 of chunk 1
        if constexpr (XR) {
            // u = h16(acc + b1) goes to actH in the accumulator layout; a row-coalesced pass then sends u to HBM, turns it into
            // h = gelu(u) in place and sends h to HBM (16 bytes per lane, whole rows per wave-instruction)
            if (hc) __syncthreads();                           // every wave is done reading the previous chunk from actH
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = 32 * wave + 8 * g + 4 * hl;
                const float4 b1 = *reinterpret_cast<const float4*>(b1_p + hc * SA_D + cl);
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    uint2 w;
                    w.x = pack_h16x2(acc[0][i][4 * g + 0] + b1.x, acc[0][i][4 * g + 1] + b1.y);
                    w.y = pack_h16x2(acc[0][i][4 * g + 2] + b1.z, acc[0][i][4 * g + 3] + b1.w);
                    *reinterpret_cast<uint2*>(actH + (i * 32 + t) * ALD + cl) = w;
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < TOK * 32 / NT; ++it) {
                const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
                const uint4 v = *reinterpret_cast<const uint4*>(actH + row * ALD + ch * 8);
                const size_t go = (size_t)(m0 + row) * SA_HID + hc * SA_D + ch * 8;
                if (row < nvalid) *reinterpret_cast<uint4*>((h16_t*)a.u + go) = v;
                const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
                uint32_t hh[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    hh[q] = pack_h16x2(sa_gelu(h16_lo(vv[q])), sa_gelu(h16_hi(vv[q])));
                const uint4 hv = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                *reinterpret_cast<uint4*>(actH + row * ALD + ch * 8) = hv;
                if (row < nvalid) *reinterpret_cast<uint4*>((h16_t*)a.h + go) = hv;
            }
            __syncthreads();
        } else {
        // u = h16(acc + b1) (saved), h = gelu(u)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cch = hc * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                const float4 b1 = *reinterpret_cast<const float4*>(b1_p + cch);
                const float bb[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int tok = i * 32 + t;
                    float uu[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) uu[q] = acc[j][i][4 * g + q] + bb[q];
                    uint2 w;
                    w.x = pack_h16x2(uu[0], uu[1]); w.y = pack_h16x2(uu[2], uu[3]);
                    if (tok < nvalid) *reinterpret_cast<uint2*>((h16_t*)a.u + (size_t)(m0 + tok) * SA_HID + cch) = w;
                    acc[j][i][4 * g + 0] = sa_gelu(h16_lo(w.x));
                    acc[j][i][4 * g + 1] = sa_gelu(h16_hi(w.x));
                    acc[j][i][4 * g + 2] = sa_gelu(h16_lo(w.y));
                    acc[j][i][4 * g + 3] = sa_gelu(h16_hi(w.y));
                }
            }
        if (hc == 1) SA_STAMP();    // 4b: bias + u store + GELU
        if (hc) __syncthreads();                               // every wave is done reading the previous chunk from actH
        if (hc == 1) SA_STAMP();    // 4c: barrier
        sa_store_h16<RB, NJ>(acc, actH, 0, (h16_t*)a.h, SA_HID, hc * SA_D, m0, nvalid);
        if (hc == 1) SA_STAMP();    // 4d: h stores
        __syncthreads();
        }
        if (hc == 1) SA_STAMP();    // 4e: barrier
        if (NJ == 1 && hc + 1 == SA_HID / SA_D) load_final();
        sa_gemm_unit<RB, NJ>((const h16_t*)a.W2, SA_HID / 16, hc * 16, NJ * wave, actH, acc2, wpre);
        if (hc + 1 < SA_HID / SA_D) sa_wprefetch((const h16_t*)a.W1, SA_D / 16, 0, (hc + 1) * 8 + NJ * wave, wpre);
        if (hc == 1) SA_STAMP();    // 4f: fc2 of chunk 1
    }
    if (NJ != 1) load_final();
    const bool nxt = a.qkv_next != nullptr;
    if (nxt) sa_wprefetch((const h16_t*)a.Wqkv_next, SA_D / 16, 0, NJ * wave, wpre);
    SA_STAMP();     // 4: MLP (fc1 + GELU + fc2)

    // ============================================================ x2 = x1 + dropout(h . W2^T + b2)  [+ pos -> next base, LN1, qkv]
    {
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res2, a.p_res2);
        const bool drop = a.p_res2 > 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cch = 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                const float4 b2 = *reinterpret_cast<const float4*>(b2_p + cch);
                const float bb[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int tok = i * 32 + t;
                    const bool ok = tok < nvalid;
                    const size_t off = (size_t)(m0 + (ok ? tok : 0)) * SA_D + cch;
                    float rr[4];
                    if constexpr (XR) {
                        const float4 rv = *reinterpret_cast<const float4*>(xres + tok * XLD + cch);
                        rr[0] = rv.x; rr[1] = rv.y; rr[2] = rv.z; rr[3] = rv.w;
                    } else {
                        rr[0] = resx[j][g][i].x + resp[j][g][i].x; rr[1] = resx[j][g][i].y + resp[j][g][i].y;
                        rr[2] = resx[j][g][i].z + resp[j][g][i].z; rr[3] = resx[j][g][i].w + resp[j][g][i].w;
                    }
                    const uint32_t keep = drop ? vpf_keep4(rng, (uint64_t)off >> 2) : 15u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float y = acc2[j][i][4 * g + q] + bb[q];
                        if (drop) y = ((keep >> q) & 1u) ? y * rng.scale : 0.f;
                        acc2[j][i][4 * g + q] = rr[q] + y;
                    }
                    if constexpr (XR) *reinterpret_cast<float4*>(xres + tok * XLD + cch) = make_float4(acc2[j][i][4 * g + 0], acc2[j][i][4 * g + 1], acc2[j][i][4 * g + 2], acc2[j][i][4 * g + 3]);
                    else if (ok) *reinterpret_cast<float4*>(a.out + off) = make_float4(acc2[j][i][4 * g + 0], acc2[j][i][4 * g + 1], acc2[j][i][4 * g + 2], acc2[j][i][4 * g + 3]);
                }
            }
        SA_STAMP();     // 5: final dropout + residual epilogue
        auto store_out = [&]() {
            if constexpr (XR) {
#pragma unroll
                for (int it = 0; it < TOK * 64 / NT; ++it) {
                    const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
                    const float4 v = *reinterpret_cast<const float4*>(xres + row * XLD + c4 * 4);
                    if (row < nvalid) *reinterpret_cast<float4*>(a.out + (size_t)(m0 + row) * SA_D + c4 * 4) = v;
                }
            }
        };
        if (!nxt) {
            if constexpr (XR) { __syncthreads(); store_out(); }
            return;
        }
        float mean[RB], rstd[RB];
        sa_layernorm<RB, NJ>(acc2, g1n_p, be1n_p, sStatA, sStatB, mean, rstd);
        if (wave == 0 && hl == 0) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (i * 32 + t < nvalid) { a.mean1n[m0 + i * 32 + t] = mean[i]; a.rstd1n[m0 + i * 32 + t] = rstd[i]; }
        }
        store_out();                  // (the LayerNorm exchange barrier: every wave's part of out is in xres)
        sa_store_h16<RB, NJ>(acc2, actA, 0, XR ? nullptr : (h16_t*)a.n1n, SA_D, 0, m0, nvalid);     // n2 is dead: every wave passed the last fc1 barrier
    }
    __syncthreads();                  // n1n complete in actA; actH, the LayerNorm exchange and xres are dead
    if constexpr (XR) sa_tile_store_rows<TOK, NT>(actA, (h16_t*)a.n1n, SA_D, 0, m0, nvalid);
    SA_STAMP();     // 6: next LayerNorm 1
    // q | k | v of the next layer: the results are held (packed h16) and stored after the last unit, so that no weight load
    // ever queues behind a batch of stores
    if constexpr (XR) {
        h16_t* qst = reg1;                                   // [TOK][QLD] over actH | exchange | xres
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            sa_zero<RB, NJ>(acc);
            sa_gemm_unit<RB, NJ>((const h16_t*)a.Wqkv_next, SA_D / 16, 0, part * 8 + NJ * wave, actA, acc, wpre);
            if (part + 1 < 3) sa_wprefetch((const h16_t*)a.Wqkv_next, SA_D / 16, 0, (part + 1) * 8 + NJ * wave, wpre);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 w;
                    w.x = pack_h16x2(acc[0][i][4 * g + 0], acc[0][i][4 * g + 1]);
                    w.y = pack_h16x2(acc[0][i][4 * g + 2], acc[0][i][4 * g + 3]);
                    *reinterpret_cast<uint2*>(qst + (i * 32 + t) * QLD + part * SA_D + 32 * wave + 8 * g + 4 * hl) = w;
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < TOK * 96 / NT; ++it) {
            const int e = threadIdx.x + it * NT, row = e / 96, ch = e - row * 96;
            const uint4 v = *reinterpret_cast<const uint4*>(qst + row * QLD + ch * 8);
            if (row < nvalid) *reinterpret_cast<uint4*>((h16_t*)a.qkv_next + (size_t)(m0 + row) * (3 * SA_D) + ch * 8) = v;
        }
        SA_STAMP();     // 7: next q/k/v projection
        return;
    }
    uint2 held[3][NJ][RB][4];
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        sa_zero<RB, NJ>(acc);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.Wqkv_next, SA_D / 16, 0, part * 8 + NJ * wave, actA, acc, wpre);
        if (part + 1 < 3) sa_wprefetch((const h16_t*)a.Wqkv_next, SA_D / 16, 0, (part + 1) * 8 + NJ * wave, wpre);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    held[part][j][i][g].x = pack_h16x2(acc[j][i][4 * g + 0], acc[j][i][4 * g + 1]);
                    held[part][j][i][g].y = pack_h16x2(acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
                }
    }
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int tok = i * 32 + t;
                    if (tok < nvalid)
                        *reinterpret_cast<uint2*>((h16_t*)a.qkv_next + (size_t)(m0 + tok) * (3 * SA_D) + part * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl) = held[part][j][i][g];
                }
    SA_STAMP();     // 7: next q/k/v projection
#undef SA_STAMP
}

template <int RB, int HPR, int LPT, bool ATT, int NJ>
static int sa_launch(const VpfSaLayerFwd& a, int chunks, hipStream_t st)
{
    const int LP = LPT, TOK = RB * 32;
    const size_t kv = ATT ? (size_t)HPR * 2 * LP * KLD * 2 : 0, mlp = (size_t)TOK * ALD * 2 + (size_t)2 * TOK * (8 / NJ) * 4;
    const size_t lds = (size_t)TOK * ALD * 2 + (kv > mlp ? kv : mlp) + ((!ATT && NJ == 1) ? (size_t)TOK * XLD * 4 + 2048 * 4 : 0);
    if (lds > 160 * 1024) return VPF_ERR_UNSUPPORTED;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)sa_layer_fwd_kernel<RB, HPR, LPT, ATT, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return VPF_ERR_HIP;
        attr = true;
    }
    const dim3 grid = ATT ? dim3(a.B, chunks) : dim3(vpf_cdiv((long)a.B * a.L, TOK), 1);
    hipLaunchKernelGGL((sa_layer_fwd_kernel<RB, HPR, LPT, ATT, NJ>), grid, dim3(64 * (8 / NJ)), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

extern "C" int vpf_sa_layer_fwd(const VpfSaLayerFwd* args, void* stream)
{
    (void)hipGetLastError();
    if (!args) return VPF_ERR_NULL;
    const VpfSaLayerFwd& a = *args;
    if (!a.qkv || !a.base || !a.rng || !a.Wo || !a.bo || !a.ln2_g || !a.ln2_b || !a.W1 || !a.b1 || !a.W2 || !a.b2 || !a.o || !a.lse ||
        !a.x1 || !a.mean2 || !a.rstd2 || !a.n2 || !a.u || !a.h || !a.out) return VPF_ERR_NULL;
    if (a.qkv_next && (!a.ln1n_g || !a.ln1n_b || !a.Wqkv_next || !a.mean1n || !a.rstd1n || !a.n1n)) return VPF_ERR_NULL;
    if (a.pos && a.pos_rows <= 0) return VPF_ERR_BADSHAPE;
    if (a.B <= 0 || a.L <= 0 || a.chunk_rows <= 0) return VPF_ERR_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    // round 3: the row-block kernels built for two workgroups per CU (sa_rows.hip); D = 384 / hidden 1536 exists only there
    if (a.attention_done && sa_rows_supported(a.D, a.hidden) && ((vpf_debug().sa_wg2 & 1) || a.D != SA_D || a.hidden != SA_HID)) {
        if (a.D != a.H * SA_DH) return VPF_ERR_UNSUPPORTED;
        return sa_rows_fwd_launch(a, st);
    }
    if (a.D != SA_D || a.H != SA_H || a.hidden != SA_HID) return VPF_ERR_UNSUPPORTED;
    const int chunks = vpf_cdiv(a.L, a.chunk_rows);
    const int nj = vpf_debug().sa_nj;
    if (a.attention_done) {                                                   // o is an input: 64-row blocks, any sequence length
        if (nj == 2) return sa_launch<2, 2, 32, false, 2>(a, 1, st);          // 4 waves x 64 channels
        return sa_launch<2, 2, 32, false, 1>(a, 1, st);                       // 8 waves x 32 channels: two waves per SIMD overlap MFMA, VALU and memory waits
    }
    // one chunk = up to 4 blocks of 32 tokens; all heads' K / V resident when the sequence is short, else one head per round
    if (a.chunk_rows <= 96 && a.L <= 96) return sa_launch<3, 4, 96, true, 2>(a, chunks, st);
    if (a.chunk_rows <= 128 && a.L <= 224) return sa_launch<4, 1, 224, true, 2>(a, chunks, st);
    return VPF_ERR_UNSUPPORTED;
}

extern "C" int vpf_ca_front_fwd(const VpfCaFront* args, void* stream)
{
    (void)hipGetLastError();
    if (!args) return VPF_ERR_NULL;
    const VpfCaFront& a = *args;
    if (!a.centers || !a.W0 || !a.b0 || !a.W1 || !a.b1 || !a.x || !a.lnq_g || !a.lnq_b || !a.Wq || !a.hpos || !a.pos || !a.base || !a.mean ||
        !a.rstd || !a.nq || !a.q) return VPF_ERR_NULL;
    if (a.M <= 0) return VPF_ERR_BADSHAPE;
    return sa_rows_ca_front_launch(a, (hipStream_t)stream);
}

// ================================================================================================ backward
// The dgrad chain of a self-attention layer in two row-block kernels around the attention backward:
//   vpf_sa_layer_bwd_mlp : d(x2) -> dropout' -> [dz2] -> . W2 * gelu'(u) -> [du] -> . W1 -> LayerNorm-2' (+ d) -> [dx1]
//                          -> dropout' -> [dz1] -> . Wo -> [do]
//   vpf_sa_layer_bwd_qkv : [dqkv] . Wqkv -> LayerNorm-1' (+ dx1) -> [dbase]
// ([..] = written to HBM: the h16 ones are the operands of the weight-gradient GEMMs / the attention backward.)
// Same layout as the forward: swapped products, a lane owns a token, weights in (transposed) fragment order.

// LayerNorm backward in place on the accumulator tile: acc = dL/dy -> dL/dx;  x (the forward input) from HBM.
// The per-channel parameter gradients of this workgroup's tokens go to pgrad[0..255] (dgamma) / pgrad[256..511] (dbeta).
// XLDS: x points at an f32 LDS tile [tokens][XLD] of the workgroup's rows (staged row-coalesced) instead of at the HBM matrix.
template <int RB, int NJ, bool XBF = false, bool XLDS = false>
__device__ __forceinline__ void sa_layernorm_bwd(f32x16_t (&acc)[NJ][RB], const float* __restrict__ x, const float* __restrict__ mean,
                                                 const float* __restrict__ rstd, const float* __restrict__ gamma, float* sStat2,
                                                 float* __restrict__ pgrad, long m0, int nvalid)
{
    constexpr int NWV = 8 / NJ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
    float mu[RB], rs[RB];
    float4 xh[NJ][4][RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const bool ok = i * 32 + t < nvalid;
        mu[i] = ok ? mean[m0 + i * 32 + t] : 0.f;
        rs[i] = ok ? rstd[m0 + i * 32 + t] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int tok = i * 32 + t;
                const size_t off = (size_t)(m0 + (tok < nvalid ? tok : 0)) * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                if constexpr (XLDS && XBF) {  // a h16 LDS tile [tokens][ALD]
                    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const h16_t*>(x) + tok * ALD + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl);
                    xh[j][g][i] = make_float4(h16_lo(u.x), h16_hi(u.x), h16_lo(u.y), h16_hi(u.y));
                } else if constexpr (XLDS) {
                    xh[j][g][i] = *reinterpret_cast<const float4*>(x + tok * XLD + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl);
                } else if constexpr (XBF) {   // the LayerNorm input was stored as h16 (x points at h16 data)
                    const uint2 u = tok < nvalid ? *reinterpret_cast<const uint2*>(reinterpret_cast<const h16_t*>(x) + off) : make_uint2(0u, 0u);
                    xh[j][g][i] = make_float4(h16_lo(u.x), h16_hi(u.x), h16_lo(u.y), h16_hi(u.y));
                } else {
                    xh[j][g][i] = tok < nvalid ? *reinterpret_cast<const float4*>(x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
    float s1[RB], s2[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    float dgam[NJ][4][4], dbet[NJ][4][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 ga = *reinterpret_cast<const float4*>(gamma + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl);
            const float gg[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { dgam[j][g][q] = 0.f; dbet[j][g][q] = 0.f; }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                float xv[4] = {xh[j][g][i].x, xh[j][g][i].y, xh[j][g][i].z, xh[j][g][i].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xn = (xv[q] - mu[i]) * rs[i];
                    const float dy = acc[j][i][4 * g + q];
                    dgam[j][g][q] += dy * xn;
                    dbet[j][g][q] += dy;
                    const float gy = dy * gg[q];
                    s1[i] += gy;
                    s2[i] += gy * xn;
                    acc[j][i][4 * g + q] = gy;
                    xv[q] = xn;
                }
                xh[j][g][i] = make_float4(xv[0], xv[1], xv[2], xv[3]);
            }
        }
    // per-token sums over the channels: lane ^ 32, then the 4 waves
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        s1[i] += __shfl_xor(s1[i], 32, 64);
        s2[i] += __shfl_xor(s2[i], 32, 64);
        if (lane < 32) *reinterpret_cast<float2*>(sStat2 + ((i * 32 + lane) * NWV + wave) * 2) = make_float2(s1[i], s2[i]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NWV; w2 += 2) {
            const float4 v0 = *reinterpret_cast<const float4*>(sStat2 + ((i * 32 + t) * NWV + w2) * 2);
            t1 += v0.x + v0.z; t2 += v0.y + v0.w;
        }
        s1[i] = t1 * (1.0f / SA_D);
        s2[i] = t2 * (1.0f / SA_D);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const float xv[4] = {xh[j][g][i].x, xh[j][g][i].y, xh[j][g][i].z, xh[j][g][i].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][i][4 * g + q] = rs[i] * (acc[j][i][4 * g + q] - s1[i] - xv[q] * s2[i]);
            }
    // parameter gradients: sum over this workgroup's tokens = over the 32 lanes of each half
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a = dgam[j][g][q], b = dbet[j][g][q];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (t == 0) {
                    const int c = 32 * NJ * wave + 32 * j + 8 * g + 4 * hl + q;
                    pgrad[c] = a;
                    pgrad[SA_D + c] = b;
                }
            }
}

// dropout backward of the f32 rows `src` in accumulator layout -> acc (f32, scaled / zeroed)
template <int RB, int NJ>
__device__ __forceinline__ void sa_load_dropout_bwd(f32x16_t (&acc)[NJ][RB], const float* __restrict__ src, const VpfRng& rng, bool drop,
                                                    long m0, int nvalid)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
    float4 v[NJ][4][RB];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int tok = i * 32 + t;
                v[j][g][i] = tok < nvalid ? *reinterpret_cast<const float4*>(src + (size_t)(m0 + tok) * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int tok = i * 32 + t;
                const size_t off = (size_t)(m0 + (tok < nvalid ? tok : 0)) * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                const uint32_t keep = drop ? vpf_keep4(rng, (uint64_t)off >> 2) : 15u;
                const float s = drop ? rng.scale : 1.f;
                acc[j][i][4 * g + 0] = (keep & 1u) ? v[j][g][i].x * s : 0.f;
                acc[j][i][4 * g + 1] = (keep & 2u) ? v[j][g][i].y * s : 0.f;
                acc[j][i][4 * g + 2] = (keep & 4u) ? v[j][g][i].z * s : 0.f;
                acc[j][i][4 * g + 3] = (keep & 8u) ? v[j][g][i].w * s : 0.f;
            }
}

template <int RB, int NJ>
__global__ void __launch_bounds__(64 * (8 / NJ)) sa_bwd_mlp_kernel(VpfSaLayerBwd a)
{
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    constexpr int TOK = RB * 32;
    h16_t* actA = lds;                               // dz2 -> dz1
    h16_t* actH = lds + TOK * ALD;                   // one 256-wide chunk of du
    float* sStat2 = reinterpret_cast<float*>(actH + TOK * ALD);   // [TOK][4] float2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
    const long M = (long)a.M;
    const long m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);

    SaWPre<NJ> wpre;
    sa_wprefetch((const h16_t*)a.W2T, SA_D / 16, 0, NJ * wave, wpre);
    f32x16_t acc[NJ][RB], acc2[NJ][RB];
    // ---- dz2 = dropout'(d)
    {
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res2, a.p_res2);
        sa_load_dropout_bwd<RB, NJ>(acc, a.d, rng, a.p_res2 > 0.f, m0, nvalid);
        sa_store_h16<RB, NJ>(acc, actA, 0, (h16_t*)a.dz2, SA_D, 0, m0, nvalid);
    }
    __syncthreads();
    // ---- du = (dz2 . W2) * gelu'(u) ;  dn = du . W1
    sa_zero<RB, NJ>(acc2);
#pragma unroll
    for (int hc = 0; hc < SA_HID / SA_D; ++hc) {
        uint2 uu[NJ][4][RB];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int tok = i * 32 + t;
                    uu[j][g][i] = tok < nvalid ? *reinterpret_cast<const uint2*>((const h16_t*)a.u + (size_t)(m0 + tok) * SA_HID + hc * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl)
                                               : make_uint2(0u, 0u);
                }
        sa_zero<RB, NJ>(acc);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.W2T, SA_D / 16, 0, hc * 8 + NJ * wave, actA, acc, wpre);
        sa_wprefetch((const h16_t*)a.W1T, SA_HID / 16, hc * 16, NJ * wave, wpre);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    acc[j][i][4 * g + 0] *= vpf_gelu_grad(h16_lo(uu[j][g][i].x));
                    acc[j][i][4 * g + 1] *= vpf_gelu_grad(h16_hi(uu[j][g][i].x));
                    acc[j][i][4 * g + 2] *= vpf_gelu_grad(h16_lo(uu[j][g][i].y));
                    acc[j][i][4 * g + 3] *= vpf_gelu_grad(h16_hi(uu[j][g][i].y));
                }
        if (hc) __syncthreads();
        sa_store_h16<RB, NJ>(acc, actH, 0, (h16_t*)a.du, SA_HID, hc * SA_D, m0, nvalid);
        __syncthreads();
        sa_gemm_unit<RB, NJ>((const h16_t*)a.W1T, SA_HID / 16, hc * 16, NJ * wave, actH, acc2, wpre);
        if (hc + 1 < SA_HID / SA_D) sa_wprefetch((const h16_t*)a.W2T, SA_D / 16, 0, (hc + 1) * 8 + NJ * wave, wpre);
    }
    sa_wprefetch((const h16_t*)a.WoT, SA_D / 16, 0, NJ * wave, wpre);
    // ---- dx1 = LayerNorm-2'(dn) + d
    sa_layernorm_bwd<RB, NJ>(acc2, a.x1, a.mean2, a.rstd2, a.ln2_g, sStat2, a.pgrad2 + (size_t)blockIdx.x * 2 * SA_D, m0, nvalid);
    {
        float4 dv[NJ][4][RB];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int tok = i * 32 + t;
                    dv[j][g][i] = tok < nvalid ? *reinterpret_cast<const float4*>(a.d + (size_t)(m0 + tok) * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res1, a.p_res1);
        const bool drop = a.p_res1 > 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int tok = i * 32 + t;
                    const size_t off = (size_t)(m0 + (tok < nvalid ? tok : 0)) * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                    float v[4] = {acc2[j][i][4 * g + 0] + dv[j][g][i].x, acc2[j][i][4 * g + 1] + dv[j][g][i].y,
                                  acc2[j][i][4 * g + 2] + dv[j][g][i].z, acc2[j][i][4 * g + 3] + dv[j][g][i].w};
                    if (tok < nvalid) *reinterpret_cast<float4*>(a.dx1 + off) = make_float4(v[0], v[1], v[2], v[3]);
                    const uint32_t keep = drop ? vpf_keep4(rng, (uint64_t)off >> 2) : 15u;
                    const float s = drop ? rng.scale : 1.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc2[j][i][4 * g + q] = ((keep >> q) & 1u) ? v[q] * s : 0.f;
                }
    }
    // (the barriers of the hidden-chunk loop guarantee every wave has finished reading dz2 from actA)
    sa_store_h16<RB, NJ>(acc2, actA, 0, (h16_t*)a.dz1, SA_D, 0, m0, nvalid);
    __syncthreads();
    // ---- do = dz1 . Wo
    sa_zero<RB, NJ>(acc);
    sa_gemm_unit<RB, NJ>((const h16_t*)a.WoT, SA_D / 16, 0, NJ * wave, actA, acc, wpre);
    sa_store_h16<RB, NJ>(acc, nullptr, 0, (h16_t*)a.dout_attn, SA_D, 0, m0, nvalid);
}

// The same with every HBM access but the u loads row-coalesced (8 waves x 32 channels).  d is only ever needed element-wise
// (dropout', + d), so it is handled in the row layout; x1 (the LayerNorm-2 input) is staged into an f32 LDS tile at the start,
// LayerNorm-2' leaves its result in that tile and a row pass turns it into dx1 (HBM) and dz1 (LDS operand tile + HBM).
// STAGED: dz2 (operand tile + HBM) and the x1 rows (xt) were put in place by the kernel body that ran in front of this one in the
// same workgroup (sa_bwd_qkv_rows_body<.., true> of the layer above: one launch per layer boundary instead of two, and the
// gradient rows cross HBM once instead of twice)
template <int RB, bool STAGED>
__device__ __forceinline__ void sa_bwd_mlp_rows_body(const VpfSaLayerBwd& a)
{
    constexpr int NJ = 1, NT = 512, TOK = RB * 32, XPT = TOK * 64 / NT;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* actA = lds;                               // dz2 -> dz1
    h16_t* actH = lds + TOK * ALD;                   // one 256-wide chunk of du, then do
    float* sStat2 = reinterpret_cast<float*>(actH + TOK * ALD);   // [TOK][8] float2
    float* xt = sStat2 + TOK * 8 * 2;                 // [TOK][XLD] f32: x1, then LayerNorm-2'(dn)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
    const long M = (long)a.M;
    const long m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);

    SaWPre<NJ> wpre;
    sa_wprefetch((const h16_t*)a.W2T, SA_D / 16, 0, NJ * wave, wpre);
    f32x16_t acc[NJ][RB], acc2[NJ][RB];
    // ---- dz2 = dropout'(d)  (row layout: operand tile + HBM);  x1 -> xt
    if constexpr (!STAGED) {
        float4 dr[XPT], xr[XPT];
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            const size_t off = (size_t)(m0 + (row < nvalid ? row : 0)) * SA_D + c4 * 4;
            dr[it] = *reinterpret_cast<const float4*>(a.d + off);
            xr[it] = *reinterpret_cast<const float4*>(a.x1 + off);
        }
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res2, a.p_res2);
        const bool drop = a.p_res2 > 0.f;
        const float sc = drop ? rng.scale : 1.f;
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            const bool ok = row < nvalid;
            const size_t off = (size_t)(m0 + (ok ? row : 0)) * SA_D + c4 * 4;
            const uint32_t keep = !ok ? 0u : (drop ? vpf_keep4(rng, (uint64_t)off >> 2) : 15u);
            uint2 w;
            w.x = pack_h16x2((keep & 1u) ? dr[it].x * sc : 0.f, (keep & 2u) ? dr[it].y * sc : 0.f);
            w.y = pack_h16x2((keep & 4u) ? dr[it].z * sc : 0.f, (keep & 8u) ? dr[it].w * sc : 0.f);
            *reinterpret_cast<uint2*>(actA + row * ALD + c4 * 4) = w;
            if (ok) *reinterpret_cast<uint2*>((h16_t*)a.dz2 + off) = w;
            *reinterpret_cast<float4*>(xt + row * XLD + c4 * 4) = ok ? xr[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    // ---- du = (dz2 . W2) * gelu'(u) ;  dn = du . W1
    sa_zero<RB, NJ>(acc2);
#pragma unroll
    for (int hc = 0; hc < SA_HID / SA_D; ++hc) {
        uint2 uu[4][RB];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int tok = i * 32 + t;
                uu[g][i] = tok < nvalid ? *reinterpret_cast<const uint2*>((const h16_t*)a.u + (size_t)(m0 + tok) * SA_HID + hc * SA_D + 32 * wave + 8 * g + 4 * hl)
                                        : make_uint2(0u, 0u);
            }
        sa_zero<RB, NJ>(acc);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.W2T, SA_D / 16, 0, hc * 8 + NJ * wave, actA, acc, wpre);
        sa_wprefetch((const h16_t*)a.W1T, SA_HID / 16, hc * 16, NJ * wave, wpre);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                acc[0][i][4 * g + 0] *= vpf_gelu_grad(h16_lo(uu[g][i].x));
                acc[0][i][4 * g + 1] *= vpf_gelu_grad(h16_hi(uu[g][i].x));
                acc[0][i][4 * g + 2] *= vpf_gelu_grad(h16_lo(uu[g][i].y));
                acc[0][i][4 * g + 3] *= vpf_gelu_grad(h16_hi(uu[g][i].y));
            }
        if (hc) __syncthreads();
        sa_store_h16<RB, NJ>(acc, actH, 0, nullptr, SA_HID, hc * SA_D, m0, nvalid);
        __syncthreads();
        sa_tile_store_rows<TOK, NT>(actH, (h16_t*)a.du, SA_HID, hc * SA_D, m0, nvalid);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.W1T, SA_HID / 16, hc * 16, NJ * wave, actH, acc2, wpre);
        if (hc + 1 < SA_HID / SA_D) sa_wprefetch((const h16_t*)a.W2T, SA_D / 16, 0, (hc + 1) * 8 + NJ * wave, wpre);
    }
    sa_wprefetch((const h16_t*)a.WoT, SA_D / 16, 0, NJ * wave, wpre);
    // ---- LayerNorm-2'(dn) -> xt (the slots this lane read its x1 from)
    sa_layernorm_bwd<RB, NJ, false, true>(acc2, xt, a.mean2, a.rstd2, a.ln2_g, sStat2, a.pgrad2 + (size_t)blockIdx.x * 2 * SA_D, m0, nvalid);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < RB; ++i)
            *reinterpret_cast<float4*>(xt + (i * 32 + t) * XLD + 32 * wave + 8 * g + 4 * hl) =
                make_float4(acc2[0][i][4 * g + 0], acc2[0][i][4 * g + 1], acc2[0][i][4 * g + 2], acc2[0][i][4 * g + 3]);
    // ---- dx1 = . + d;  dz1 = dropout'(dx1)   (row layout; every wave has left the hidden-chunk loop: dz2 in actA is dead)
    {
        float4 dr[XPT];
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            dr[it] = *reinterpret_cast<const float4*>(a.d + (size_t)(m0 + (row < nvalid ? row : 0)) * SA_D + c4 * 4);
        }
        __syncthreads();
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res1, a.p_res1);
        const bool drop = a.p_res1 > 0.f;
        const float sc = drop ? rng.scale : 1.f;
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            const bool ok = row < nvalid;
            const size_t off = (size_t)(m0 + (ok ? row : 0)) * SA_D + c4 * 4;
            float4 v = *reinterpret_cast<const float4*>(xt + row * XLD + c4 * 4);
            v.x += dr[it].x; v.y += dr[it].y; v.z += dr[it].z; v.w += dr[it].w;
            if (ok) *reinterpret_cast<float4*>(a.dx1 + off) = v;
            const uint32_t keep = !ok ? 0u : (drop ? vpf_keep4(rng, (uint64_t)off >> 2) : 15u);
            uint2 w;
            w.x = pack_h16x2((keep & 1u) ? v.x * sc : 0.f, (keep & 2u) ? v.y * sc : 0.f);
            w.y = pack_h16x2((keep & 4u) ? v.z * sc : 0.f, (keep & 8u) ? v.w * sc : 0.f);
            *reinterpret_cast<uint2*>(actA + row * ALD + c4 * 4) = w;
            if (ok) *reinterpret_cast<uint2*>((h16_t*)a.dz1 + off) = w;
        }
    }
    __syncthreads();
    // ---- do = dz1 . Wo   (staged through actH: the last du chunk is dead)
    sa_zero<RB, NJ>(acc);
    sa_gemm_unit<RB, NJ>((const h16_t*)a.WoT, SA_D / 16, 0, NJ * wave, actA, acc, wpre);
    sa_store_h16<RB, NJ>(acc, actH, 0, nullptr, SA_D, 0, m0, nvalid);
    __syncthreads();
    sa_tile_store_rows<TOK, NT>(actH, (h16_t*)a.dout_attn, SA_D, 0, m0, nvalid);
}
template <int RB>
__global__ void __launch_bounds__(512) sa_bwd_mlp_rows_kernel(VpfSaLayerBwd a)
{
    sa_bwd_mlp_rows_body<RB, false>(a);
}

template <int RB, int NJ>
__global__ void __launch_bounds__(64 * (8 / NJ)) sa_bwd_qkv_kernel(VpfSaLayerBwd a)
{
    constexpr int NT = 64 * (8 / NJ);
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    constexpr int TOK = RB * 32;
    h16_t* actA = lds;                                // two buffers of [TOK][ALD]: the q | k | v slices of dqkv
    float* sStat2 = reinterpret_cast<float*>(lds + 2 * TOK * ALD);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
    const long M = (long)a.M;
    const long m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);
    constexpr int CPT = TOK * 32 / NT;

    SaWPre<NJ> wpre;
    sa_wprefetch((const h16_t*)a.WqkvT, 3 * SA_D / 16, 0, NJ * wave, wpre);
    uint4 r[CPT];
    auto load_part = [&](int part) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
            r[it] = row < nvalid ? *reinterpret_cast<const uint4*>((const h16_t*)a.dqkv + (size_t)(m0 + row) * (3 * SA_D) + part * SA_D + ch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_part = [&](int buf) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
            *reinterpret_cast<uint4*>(actA + buf * TOK * ALD + row * ALD + ch * 8) = r[it];
        }
    };
    load_part(0);
    store_part(0);
    __syncthreads();
    f32x16_t acc[NJ][RB];
    sa_zero<RB, NJ>(acc);
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        if (part + 1 < 3) load_part(part + 1);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.WqkvT, 3 * SA_D / 16, part * 16, NJ * wave, actA + (part & 1) * TOK * ALD, acc, wpre);
        if (part + 1 < 3) {
            sa_wprefetch((const h16_t*)a.WqkvT, 3 * SA_D / 16, (part + 1) * 16, NJ * wave, wpre);
            store_part((part + 1) & 1);
            __syncthreads();
        }
    }
    // ---- dbase = LayerNorm-1'(dn1) + dx1
    sa_layernorm_bwd<RB, NJ>(acc, a.base, a.mean1, a.rstd1, a.ln1_g, sStat2, a.pgrad1 + (size_t)blockIdx.x * 2 * SA_D, m0, nvalid);
    float4 dv[NJ][4][RB];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int tok = i * 32 + t;
                dv[j][g][i] = tok < nvalid ? *reinterpret_cast<const float4*>(a.dx1 + (size_t)(m0 + tok) * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int tok = i * 32 + t;
                if (tok >= nvalid) continue;
                const size_t off = (size_t)(m0 + tok) * SA_D + 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
                const float4 v = make_float4(acc[j][i][4 * g + 0] + dv[j][g][i].x, acc[j][i][4 * g + 1] + dv[j][g][i].y,
                                             acc[j][i][4 * g + 2] + dv[j][g][i].z, acc[j][i][4 * g + 3] + dv[j][g][i].w);
                *reinterpret_cast<float4*>(a.dbase + off) = v;
                if (a.dsum) {
                    float4 s = a.dsum_init ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(a.dsum + off);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                    *reinterpret_cast<float4*>(a.dsum + off) = s;
                }
            }
}

// The same with every HBM access row-coalesced (8 waves x 32 channels): the LayerNorm input is staged into an f32 LDS tile at
// the start, LayerNorm' leaves its result in that tile, and a row pass adds dx1 and writes dbase (and dsum).
// HANDOFF: the rows of dbase are also turned into the MLP backward's first operand of the layer BELOW (b: dz2 = dropout'(dbase) into
// the operand tile and HBM, b's x1 rows into xt) -- sa_bwd_mlp_rows_body<.., true>(b) follows in the same workgroup.
// NP: 256-wide parts of the incoming gradient (3: dq | dk | dv of a self-attention layer against Wqkv^T; 1: dq of a cross-attention
// layer against Wq^T -- the same chain, vpf_ca_front_bwd)
template <int RB, bool HANDOFF, int NP = 3>
__device__ __forceinline__ void sa_bwd_qkv_rows_body(const VpfSaLayerBwd& a, const VpfSaLayerBwd& b)
{
    constexpr int NJ = 1, NT = 512, TOK = RB * 32, CPT = TOK * 32 / NT, XPT = TOK * 64 / NT;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* actA = lds;                                // two buffers of [TOK][ALD]: the q | k | v slices of dqkv
    float* sStat2 = reinterpret_cast<float*>(lds + 2 * TOK * ALD);
    float* xt = sStat2 + TOK * 8 * 2;                  // [TOK][XLD] f32: base, then LayerNorm-1'(dn1)
    const int wave = threadIdx.x >> 6;
    const long M = (long)a.M;
    const long m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);

    SaWPre<NJ> wpre;
    sa_wprefetch((const h16_t*)a.WqkvT, NP * SA_D / 16, 0, NJ * wave, wpre);
    uint4 r[CPT];
    auto load_part = [&](int part) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
            r[it] = row < nvalid ? *reinterpret_cast<const uint4*>((const h16_t*)a.dqkv + (size_t)(m0 + row) * (NP * SA_D) + part * SA_D + ch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_part = [&](int buf) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
            *reinterpret_cast<uint4*>(actA + buf * TOK * ALD + row * ALD + ch * 8) = r[it];
        }
    };
    load_part(0);
    {
        float4 xb[XPT];
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            xb[it] = row < nvalid ? *reinterpret_cast<const float4*>(a.base + (size_t)(m0 + row) * SA_D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        store_part(0);
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            *reinterpret_cast<float4*>(xt + row * XLD + c4 * 4) = xb[it];
        }
    }
    __syncthreads();
    f32x16_t acc[NJ][RB];
    sa_zero<RB, NJ>(acc);
#pragma unroll
    for (int part = 0; part < NP; ++part) {
        if (part + 1 < NP) load_part(part + 1);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.WqkvT, NP * SA_D / 16, part * 16, NJ * wave, actA + (part & 1) * TOK * ALD, acc, wpre);
        if (part + 1 < NP) {
            sa_wprefetch((const h16_t*)a.WqkvT, NP * SA_D / 16, (part + 1) * 16, NJ * wave, wpre);
            store_part((part + 1) & 1);
            __syncthreads();
        }
    }
    // ---- dbase = LayerNorm-1'(dn1) + dx1
    sa_layernorm_bwd<RB, NJ, false, true>(acc, xt, a.mean1, a.rstd1, a.ln1_g, sStat2, a.pgrad1 + (size_t)blockIdx.x * 2 * SA_D, m0, nvalid);
    {
        const int lane = threadIdx.x & 63, hl = lane >> 5, t = lane & 31;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i)      // (the slots this lane read its x from: no other lane touches them)
                *reinterpret_cast<float4*>(xt + (i * 32 + t) * XLD + 32 * wave + 8 * g + 4 * hl) =
                    make_float4(acc[0][i][4 * g + 0], acc[0][i][4 * g + 1], acc[0][i][4 * g + 2], acc[0][i][4 * g + 3]);
    }
    float4 dv[XPT], sv[XPT];
#pragma unroll
    for (int it = 0; it < XPT; ++it) {
        const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
        const size_t off = (size_t)(m0 + (row < nvalid ? row : 0)) * SA_D + c4 * 4;
        dv[it] = a.dx1 ? *reinterpret_cast<const float4*>(a.dx1 + off) : make_float4(0.f, 0.f, 0.f, 0.f);      // (no residual: vpf_ca_kv_bwd)
        if (a.dsum) sv[it] = a.dsum_init ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(a.dsum + off);
    }
    __syncthreads();
    if constexpr (!HANDOFF) {
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            if (row >= nvalid) continue;
            const size_t off = (size_t)(m0 + row) * SA_D + c4 * 4;
            float4 v = *reinterpret_cast<const float4*>(xt + row * XLD + c4 * 4);
            v.x += dv[it].x; v.y += dv[it].y; v.z += dv[it].z; v.w += dv[it].w;
            *reinterpret_cast<float4*>(a.dbase + off) = v;
            if (a.dsum) *reinterpret_cast<float4*>(a.dsum + off) = make_float4(sv[it].x + v.x, sv[it].y + v.y, sv[it].z + v.z, sv[it].w + v.w);
        }
    } else {
        // the same row pass, continued into the first pass of sa_bwd_mlp_rows_body of the layer below (identical arithmetic)
        float4 xr[XPT];
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            xr[it] = *reinterpret_cast<const float4*>(b.x1 + (size_t)(m0 + (row < nvalid ? row : 0)) * SA_D + c4 * 4);
        }
        const VpfRng rng = vpf_rng_init(b.rng, b.site_res2, b.p_res2);
        const bool drop = b.p_res2 > 0.f;
        const float sc = drop ? rng.scale : 1.f;
#pragma unroll
        for (int it = 0; it < XPT; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 6, c4 = e & 63;
            const bool ok = row < nvalid;
            const size_t off = (size_t)(m0 + (ok ? row : 0)) * SA_D + c4 * 4;
            float4 v = *reinterpret_cast<const float4*>(xt + row * XLD + c4 * 4);
            v.x += dv[it].x; v.y += dv[it].y; v.z += dv[it].z; v.w += dv[it].w;
            if (ok) {
                *reinterpret_cast<float4*>(a.dbase + off) = v;
                if (a.dsum) *reinterpret_cast<float4*>(a.dsum + off) = make_float4(sv[it].x + v.x, sv[it].y + v.y, sv[it].z + v.z, sv[it].w + v.w);
            }
            const uint32_t keep = !ok ? 0u : (drop ? vpf_keep4(rng, (uint64_t)off >> 2) : 15u);
            uint2 w;
            w.x = pack_h16x2((keep & 1u) ? v.x * sc : 0.f, (keep & 2u) ? v.y * sc : 0.f);
            w.y = pack_h16x2((keep & 4u) ? v.z * sc : 0.f, (keep & 8u) ? v.w * sc : 0.f);
            *reinterpret_cast<uint2*>(actA + row * ALD + c4 * 4) = w;
            if (ok) *reinterpret_cast<uint2*>((h16_t*)b.dz2 + off) = w;
            *reinterpret_cast<float4*>(xt + row * XLD + c4 * 4) = ok ? xr[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
template <int RB>
__global__ void __launch_bounds__(512) sa_bwd_qkv_rows_kernel(VpfSaLayerBwd a)
{
    sa_bwd_qkv_rows_body<RB, false>(a, a);
}
// the front of a cross-attention layer, backward: dq . Wq -> q LayerNorm' -> + dx1 -> dx (and the running positional-gradient sum)
template <int RB>
__global__ void __launch_bounds__(512) ca_front_bwd_rows_kernel(VpfSaLayerBwd a)
{
    sa_bwd_qkv_rows_body<RB, false, 1>(a, a);
}
// the key / value side of a cross-attention layer, backward: dk | dv . (Wk | Wv) -> kv LayerNorm' -> dxkv (no residual)
template <int RB>
__global__ void __launch_bounds__(512) ca_kv_bwd_rows_kernel(VpfSaLayerBwd a)
{
    sa_bwd_qkv_rows_body<RB, false, 2>(a, a);
}
// qkv backward of layer i and MLP backward of layer i - 1 (the layer below) in one workgroup: b.d == a.dbase
template <int RB>
__global__ void __launch_bounds__(512) sa_bwd_qkv_mlp_rows_kernel(VpfSaLayerBwd a, VpfSaLayerBwd b)
{
    sa_bwd_qkv_rows_body<RB, true>(a, b);
    __syncthreads();                      // the operand tile and xt are complete; every wave's dbase rows are on their way to L2
    sa_bwd_mlp_rows_body<RB, true>(b);
}

// dgamma[c] += sum_r partials[r][c], dbeta[c] += sum_r partials[r][256 + c]  (fixed order: deterministic) for a list of
// LayerNorms in one launch (all the LayerNorms of a layer stack at the end of its backward)
struct PgradJobs { VpfPgradJob job[VPF_PGRAD_MAX_JOBS]; };
__global__ void __launch_bounds__(1024) sa_pgrad_reduce_kernel(PgradJobs jobs)
{
    // block = 16 of the 2 D columns x 64 row groups (a job with 2 048 partial rows -- the K / V producer's LayerNorm -- is 32 dependent
    // loads per thread this way; with 64 columns x 16 row groups it was 128 and took 21 us at the very end of the backward pass)
    __shared__ float fold[64][17];
    const VpfPgradJob j = jobs.job[blockIdx.y];
    const int D = j.D > 0 ? j.D : SA_D;
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    if (blockIdx.x * 16 >= 2 * D) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rg;
    for (; r + 192 < j.rows; r += 256) {
        s0 += j.partials[(size_t)r * 2 * D + c];
        s1 += j.partials[(size_t)(r + 64) * 2 * D + c];
        s2 += j.partials[(size_t)(r + 128) * 2 * D + c];
        s3 += j.partials[(size_t)(r + 192) * 2 * D + c];
    }
    for (; r < j.rows; r += 64) s0 += j.partials[(size_t)r * 2 * D + c];
    fold[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) tot += fold[k][cl];
        if (c < D) j.dgamma[c] += tot; else j.dbeta[c - D] += tot;
    }
}
extern "C" int vpf_ln_pgrad_reduce(const VpfPgradJob* jobs, int njobs, void* stream)
{
    (void)hipGetLastError();
    if (!jobs) return VPF_ERR_NULL;
    if (njobs <= 0 || njobs > VPF_PGRAD_MAX_JOBS) return VPF_ERR_BADSHAPE;
    PgradJobs pj;
    int dmax = SA_D;
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].partials || !jobs[i].dgamma || !jobs[i].dbeta) return VPF_ERR_NULL;
        if (jobs[i].rows <= 0 || jobs[i].D < 0 || (jobs[i].D % 8)) return VPF_ERR_BADSHAPE;
        pj.job[i] = jobs[i];
        if (jobs[i].D > dmax) dmax = jobs[i].D;
    }
    hipLaunchKernelGGL(sa_pgrad_reduce_kernel, dim3(2 * dmax / 16, njobs), dim3(1024), 0, (hipStream_t)stream, pj);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// rows of LayerNorm parameter-gradient partials (2 D floats each) that vpf_sa_layer_bwd_mlp / _qkv write for M tokens: the caller
// sizes pgrad1 / pgrad2 and the reduce job with it (one row per 64 tokens for the D = 256 kernels of rounds 1 - 2, per 32 tokens for
// the round-3 kernels)
extern "C" int vpf_sa_layer_pgrad_rows_h(long M, int D, int hidden)
{
    if (M <= 0) return 0;
    if (D == SA_D && hidden == SA_HID && !(vpf_debug().sa_wg2 & 2)) return vpf_cdiv(M, 64);
    const int t = sa_rows_bwd_pgrad_tokens(D);
    return D == SA_D ? vpf_cdiv(M, 64) * (64 / t) : vpf_cdiv(M, t);
}
extern "C" int vpf_sa_layer_pgrad_rows(long M, int D) { return vpf_sa_layer_pgrad_rows_h(M, D, D == SA_D ? SA_HID : 4 * D); }

static int sa_bwd_nj()
{
    return vpf_debug().sa_nj;
}
static bool sa_bwd_rows3(const VpfSaLayerBwd& a)      // the round-3 kernels (sa_rows.hip): any supported width when asked for, the only ones beyond D = 256
{
    return sa_rows_supported(a.D, a.hidden) && ((vpf_debug().sa_wg2 & 2) || a.D != SA_D || a.hidden != SA_HID);
}
static int sa_bwd_check(const VpfSaLayerBwd& a)
{
    if (a.M <= 0) return VPF_ERR_BADSHAPE;
    if (sa_bwd_rows3(a)) return VPF_OK;
    if (a.D != SA_D || a.hidden != SA_HID) return VPF_ERR_UNSUPPORTED;
    return VPF_OK;
}
extern "C" int vpf_sa_layer_bwd_mlp(const VpfSaLayerBwd* args, void* stream)
{
    (void)hipGetLastError();
    if (!args) return VPF_ERR_NULL;
    const VpfSaLayerBwd& a = *args;
    int rc = sa_bwd_check(a);
    if (rc) return rc;
    if (!a.d || !a.rng || !a.u || !a.x1 || !a.mean2 || !a.rstd2 || !a.ln2_g || !a.W2T || !a.W1T || !a.WoT || !a.dz2 || !a.du || !a.dx1 ||
        !a.dz1 || !a.dout_attn || !a.pgrad2) return VPF_ERR_NULL;
    if (sa_bwd_rows3(a)) return sa_rows_bwd_mlp_launch(a, (hipStream_t)stream);
    constexpr int RB = 2, TOK = RB * 32;
    const size_t lds = (size_t)2 * TOK * ALD * 2 + (size_t)TOK * 8 * 2 * 4;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)sa_bwd_mlp_kernel<RB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        if (hipFuncSetAttribute((const void*)sa_bwd_mlp_kernel<RB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        if (hipFuncSetAttribute((const void*)sa_bwd_mlp_rows_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    const int nwg = vpf_cdiv((long)a.M, TOK);
    const int rows = vpf_debug().sa_bwd_rows;
    if (sa_bwd_nj() == 2) hipLaunchKernelGGL((sa_bwd_mlp_kernel<RB, 2>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    else if (rows) hipLaunchKernelGGL((sa_bwd_mlp_rows_kernel<RB>), dim3(nwg), dim3(512), lds + (size_t)TOK * XLD * 4, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((sa_bwd_mlp_kernel<RB, 1>), dim3(nwg), dim3(512), lds, (hipStream_t)stream, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
extern "C" int vpf_sa_layer_bwd_qkv(const VpfSaLayerBwd* args, void* stream)
{
    (void)hipGetLastError();
    if (!args) return VPF_ERR_NULL;
    const VpfSaLayerBwd& a = *args;
    int rc = sa_bwd_check(a);
    if (rc) return rc;
    if (!a.dqkv || !a.WqkvT || !a.base || !a.mean1 || !a.rstd1 || !a.ln1_g || !a.dx1 || !a.dbase || !a.pgrad1) return VPF_ERR_NULL;
    if (sa_bwd_rows3(a)) return sa_rows_bwd_qkv_launch(a, (hipStream_t)stream);
    constexpr int RB = 2, TOK = RB * 32;
    const size_t lds = (size_t)2 * TOK * ALD * 2 + (size_t)TOK * 8 * 2 * 4;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)sa_bwd_qkv_kernel<RB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        if (hipFuncSetAttribute((const void*)sa_bwd_qkv_kernel<RB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        if (hipFuncSetAttribute((const void*)sa_bwd_qkv_rows_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    const int nwg = vpf_cdiv((long)a.M, TOK);
    const int rows = vpf_debug().sa_bwd_rows;
    if (sa_bwd_nj() == 2) hipLaunchKernelGGL((sa_bwd_qkv_kernel<RB, 2>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    else if (rows) hipLaunchKernelGGL((sa_bwd_qkv_rows_kernel<RB>), dim3(nwg), dim3(512), lds + (size_t)TOK * XLD * 4, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((sa_bwd_qkv_kernel<RB, 1>), dim3(nwg), dim3(512), lds, (hipStream_t)stream, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// Backward of a cross-attention layer's query side (partseg.py:100-116,48-51 + the residual of :201-213) as one row-block kernel:
// dq h16 [M, D] (a->dqkv) . Wq (a->WqkvT = vpf_pack_wfrag(transposed) of the h16 [D, D] q weight) -> q LayerNorm' (a->base = its
// input, mean1 / rstd1 / ln1_g) -> + a->dx1 -> a->dbase (f32 [M, D]); a->dsum (nullable) accumulates it; the LayerNorm's parameter
// gradients leave as partial rows (a->pgrad1).  D = 256 (else VPF_ERR_UNSUPPORTED: the caller keeps its GEMM + LayerNorm kernels).
extern "C" int vpf_ca_front_bwd(const VpfSaLayerBwd* args, void* stream)
{
    (void)hipGetLastError();
    if (!args) return VPF_ERR_NULL;
    const VpfSaLayerBwd& a = *args;
    if (a.M <= 0) return VPF_ERR_BADSHAPE;
    if (!a.dqkv || !a.WqkvT || !a.base || !a.mean1 || !a.rstd1 || !a.ln1_g || !a.dx1 || !a.dbase || !a.pgrad1) return VPF_ERR_NULL;
    if (a.D != SA_D) return sa_rows_ca_front_bwd_launch(a, (hipStream_t)stream);      // D = 384; anything else: VPF_ERR_UNSUPPORTED
    constexpr int RB = 2, TOK = RB * 32;
    const size_t lds = (size_t)2 * TOK * ALD * 2 + (size_t)TOK * 8 * 2 * 4 + (size_t)TOK * XLD * 4;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)ca_front_bwd_rows_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((ca_front_bwd_rows_kernel<RB>), dim3(vpf_cdiv((long)a.M, TOK)), dim3(512), lds, (hipStream_t)stream, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// The key / value side of the same layer when its input is an f32 [M, D] tensor (the image branch's patch embeddings; the point-cloud
// branch's K / V producer has vpf_adapter_kv_bwd): dqkv = dk | dv h16 [M, 2D], WqkvT = vpf_pack_wfrag(transposed) of the h16 [2D, D]
// k | v weights, base / mean1 / rstd1 / ln1_g = the kv LayerNorm's input, statistics, scale; dx1 may be NULL (no residual); dbase = dxkv f32.
extern "C" int vpf_ca_kv_bwd(const VpfSaLayerBwd* args, void* stream)
{
    (void)hipGetLastError();
    if (!args) return VPF_ERR_NULL;
    const VpfSaLayerBwd& a = *args;
    if (a.M <= 0) return VPF_ERR_BADSHAPE;
    if (a.D != SA_D) return VPF_ERR_UNSUPPORTED;
    if (!a.dqkv || !a.WqkvT || !a.base || !a.mean1 || !a.rstd1 || !a.ln1_g || !a.dbase || !a.pgrad1) return VPF_ERR_NULL;
    constexpr int RB = 2, TOK = RB * 32;
    const size_t lds = (size_t)2 * TOK * ALD * 2 + (size_t)TOK * 8 * 2 * 4 + (size_t)TOK * XLD * 4;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)ca_kv_bwd_rows_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((ca_kv_bwd_rows_kernel<RB>), dim3(vpf_cdiv((long)a.M, TOK)), dim3(512), lds, (hipStream_t)stream, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// qkv backward of one layer and the MLP backward of the layer BELOW it (mlp->d must be qkv->dbase: the gradient that leaves the
// upper layer is the one that enters the lower one) as ONE launch where the fused row-block kernel exists (D = 256, the round-2
// kernels); otherwise the two launches of vpf_sa_layer_bwd_qkv / vpf_sa_layer_bwd_mlp.  Same results either way.
extern "C" int vpf_sa_layer_bwd_qkv_mlp(const VpfSaLayerBwd* qkv, const VpfSaLayerBwd* mlp, void* stream)
{
    (void)hipGetLastError();
    if (!qkv || !mlp) return VPF_ERR_NULL;
    const VpfSaLayerBwd& a = *qkv;
    const VpfSaLayerBwd& b = *mlp;
    const bool fusable = a.D == SA_D && a.hidden == SA_HID && b.D == SA_D && b.hidden == SA_HID && a.M == b.M && !sa_bwd_rows3(a) && !sa_bwd_rows3(b) &&
                         sa_bwd_nj() != 2 && vpf_debug().sa_bwd_rows && vpf_debug().sa_bwd_fuse && (const void*)b.d == (const void*)a.dbase;
    if (!fusable) {
        const int rc = vpf_sa_layer_bwd_qkv(qkv, stream);
        return rc ? rc : vpf_sa_layer_bwd_mlp(mlp, stream);
    }
    if (!a.dqkv || !a.WqkvT || !a.base || !a.mean1 || !a.rstd1 || !a.ln1_g || !a.dx1 || !a.dbase || !a.pgrad1) return VPF_ERR_NULL;
    if (!b.d || !b.rng || !b.u || !b.x1 || !b.mean2 || !b.rstd2 || !b.ln2_g || !b.W2T || !b.W1T || !b.WoT || !b.dz2 || !b.du || !b.dx1 ||
        !b.dz1 || !b.dout_attn || !b.pgrad2) return VPF_ERR_NULL;
    constexpr int RB = 2, TOK = RB * 32;
    const size_t lds = (size_t)2 * TOK * ALD * 2 + (size_t)TOK * 8 * 2 * 4 + (size_t)TOK * XLD * 4;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)sa_bwd_qkv_mlp_rows_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((sa_bwd_qkv_mlp_rows_kernel<RB>), dim3(vpf_cdiv((long)a.M, TOK)), dim3(512), lds, (hipStream_t)stream, a, b);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// ================================================================================================ cross-attention K / V producer
// PointCloudInputAdapter.point_mlp (classifier.py:31-36: Linear(C,64) -> LayerNorm(64) -> ReLU -> Linear(64,D)), the
// cross-attention kv LayerNorm (partseg.py:100-116) and the bias-free K / V projections (partseg.py:48-51) for 64 points
// per workgroup in ONE kernel: the per-point embedding [B*N, D] -- the largest activation of the step -- goes from the
// 64-wide hidden layer to K / V through LDS and registers.  It is still written once (h16, with the hidden layer and the
// normalised rows) because the backward pass reads it, but it is never read back in the forward pass.
template <int NJ>
__global__ void __launch_bounds__(64 * (8 / NJ), 4) adapter_kv_fwd_kernel(VpfAdapterKv a)
{
    constexpr int RB = 2, TOK = RB * 32, NT = 64 * (8 / NJ), NWV = 8 / NJ, A1LD = 72;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* sA1 = lds;                                  // [TOK][A1LD]  hidden layer (h16)
    h16_t* actA = lds + TOK * A1LD;                    // [TOK][ALD]   normalised per-point embedding
    float* sStat = reinterpret_cast<float*>(actA + TOK * ALD);      // [TOK][NWV] float2
    h16_t* sX = reinterpret_cast<h16_t*>(sStat + TOK * NWV * 2);  // [TOK][ALD]   per-point embedding before the LayerNorm, then the K rows, then
                                                                    //              the V rows (staging for HBM).  79 KB in all: two workgroups share a
                                                                    //              CU and one's MFMA phases fill the other's LayerNorm / store phases
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
    const long M = a.M, m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);
    const int C = a.C;

    SaWPre<NJ> wpre;
    sa_wprefetch((const h16_t*)a.Wkv, SA_D / 16, 0, NJ * wave, wpre);
    // ---- hidden layer: thread = (token, 8 of the 64 channels); LayerNorm over the token's 8 threads (lanes ^1 ^2 ^4)
    for (int e = threadIdx.x; e < TOK * 8; e += NT) {
        const int tok = e >> 3, cg = (e & 7) * 8;
        const bool ok = tok < nvalid;
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = (ok && j < C) ? a.x[(size_t)(m0 + tok) * C + j] : 0.f;
        float h[8], s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = a.b1[cg + k];
            for (int j = 0; j < C; ++j) v += a.W1[(cg + k) * C + j] * xv[j];
            h[k] = v; s += v;
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mu = s * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { h[k] -= mu; q += h[k] * h[k]; }
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        const float rs = rsqrtf(q * (1.f / 64.f) + 1e-5f);
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = fmaxf(h[2 * k] * rs * a.ln_g[cg + 2 * k] + a.ln_b[cg + 2 * k], 0.f);
            const float hi = fmaxf(h[2 * k + 1] * rs * a.ln_g[cg + 2 * k + 1] + a.ln_b[cg + 2 * k + 1], 0.f);
            w[k] = pack_h16x2(lo, hi);
        }
        const uint4 v4 = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(sA1 + tok * A1LD + cg) = v4;
        if (ok) *reinterpret_cast<uint4*>((h16_t*)a.a1 + (size_t)(m0 + tok) * 64 + cg) = v4;
    }
    __syncthreads();
    // ---- per-point embedding = hidden . W2^T + b2 (K = 64), rounded to h16 as the unfused path stores it, then kv LayerNorm
    f32x16_t acc[NJ][RB];
    sa_zero<RB, NJ>(acc);
    {
        const uint4* w0 = reinterpret_cast<const uint4*>(a.W2) + ((size_t)(NJ * wave) * 4) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            h16x8_t af[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) af[j] = __builtin_bit_cast(h16x8_t, w0[((size_t)j * 4 + ks) * 64]);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const h16x8_t x = sa_frag_row(sA1, A1LD, i * 32, ks * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j][i] = vpf_mfma32(af[j], x, acc[j][i]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * NJ * wave + 32 * j + 8 * g + 4 * hl;
            const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + c);
            const float bb[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                uint2 u;
                u.x = pack_h16x2(acc[j][i][4 * g + 0] + bb[0], acc[j][i][4 * g + 1] + bb[1]);
                u.y = pack_h16x2(acc[j][i][4 * g + 2] + bb[2], acc[j][i][4 * g + 3] + bb[3]);
                const int tok = i * 32 + t;
                *reinterpret_cast<uint2*>(sX + tok * ALD + c) = u;       // to HBM in the row pass below (whole rows per wave-instruction)
                acc[j][i][4 * g + 0] = h16_lo(u.x); acc[j][i][4 * g + 1] = h16_hi(u.x);
                acc[j][i][4 * g + 2] = h16_lo(u.y); acc[j][i][4 * g + 3] = h16_hi(u.y);
            }
        }
    float mean[RB], rstd[RB];
    sa_layernorm<RB, NJ>(acc, a.lnkv_g, a.lnkv_b, sStat, sStat, mean, rstd);
    if (wave == 0 && hl == 0) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
            if (i * 32 + t < nvalid) { a.mean[m0 + i * 32 + t] = mean[i]; a.rstd[m0 + i * 32 + t] = rstd[i]; }
    }
    sa_store_h16<RB, NJ>(acc, actA, 0, nullptr, SA_D, 0, m0, nvalid);
    __syncthreads();
    sa_tile_store_rows<TOK, NT>(sX, (h16_t*)a.xkv, SA_D, 0, m0, nvalid);
    sa_tile_store_rows<TOK, NT>(actA, (h16_t*)a.nk, SA_D, 0, m0, nvalid);
    // ---- K | V = normalised . Wkv^T   (two 256-channel halves, each staged in LDS and stored as whole 512-byte half rows)
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        sa_zero<RB, NJ>(acc);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.Wkv, SA_D / 16, 0, part * 8 + NJ * wave, actA, acc, wpre);
        if (part == 0) sa_wprefetch((const h16_t*)a.Wkv, SA_D / 16, 0, 8 + NJ * wave, wpre);
        __syncthreads();                                             // the row pass that read sX last is done
        sa_store_h16<RB, NJ>(acc, sX, 0, nullptr, SA_D, 0, m0, nvalid);
        __syncthreads();
        sa_tile_store_rows<TOK, NT>(sX, (h16_t*)a.kv, 2 * SA_D, part * SA_D, m0, nvalid);
    }
}

extern "C" int vpf_adapter_kv_fwd(const VpfAdapterKv* args, void* stream)
{
    (void)hipGetLastError();
    if (!args) return VPF_ERR_NULL;
    const VpfAdapterKv& a = *args;
    if (!a.x || !a.W1 || !a.b1 || !a.ln_g || !a.ln_b || !a.W2 || !a.b2 || !a.lnkv_g || !a.lnkv_b || !a.Wkv || !a.a1 || !a.xkv || !a.mean ||
        !a.rstd || !a.nk || !a.kv) return VPF_ERR_NULL;
    if (a.M <= 0 || a.C <= 0 || a.C > 8) return VPF_ERR_BADSHAPE;
    if (a.D != SA_D || (vpf_debug().sa_wg2 & 8)) return sa_rows_adapter_kv_fwd_launch(a, (hipStream_t)stream);       // D = 384 (bit 3: D = 256 too); anything else: VPF_ERR_UNSUPPORTED
    constexpr int TOK = 64;
    const size_t lds = (size_t)TOK * 72 * 2 + (size_t)TOK * ALD * 2 + (size_t)TOK * 8 * 2 * 4 + (size_t)TOK * ALD * 2;
    static_assert(TOK * 72 * 2 + TOK * ALD * 2 + TOK * 8 * 2 * 4 + TOK * ALD * 2 <= 80 * 1024, "two workgroups per CU");
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)adapter_kv_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((adapter_kv_fwd_kernel<1>), dim3(vpf_cdiv(a.M, (long)TOK)), dim3(512), lds, (hipStream_t)stream, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}


// ================================================================================================ K / V producer, backward
// dkv [M, 2D] -> (. Wk | Wv) -> kv LayerNorm' -> dxkv (h16, also the operand of the adapter's weight gradient) -> . W2 -> da1
// (h16 [M, 64]), 64 points per workgroup.  The kv LayerNorm's parameter gradients leave as per-workgroup partial rows
// (folded by vpf_ln_pgrad_reduce); the two weight-gradient GEMMs (dkv x nk, dxkv x a1) stay GEMMs and the 3 -> 64 front's
// backward stays vpf_adapter_front_bwd (a per-workgroup fold of its 704 parameter-gradient sums costs more than it saves).
__global__ void __launch_bounds__(512) adapter_kv_bwd_kernel(VpfAdapterKvBwd a)
{
    constexpr int RB = 2, NJ = 1, TOK = 64, NT = 512;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* actA = lds;                                 // two [TOK][ALD] buffers: dk | dv rows, then dxkv in buffer 0
    float* sStat2 = reinterpret_cast<float*>(lds + 2 * TOK * ALD);  // [TOK][8] float2
    h16_t* sX = reinterpret_cast<h16_t*>(sStat2 + TOK * 8 * 2);   // [TOK][ALD] the kv LayerNorm's input rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, t = lane & 31;
    const long M = a.M, m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);

    SaWPre<NJ> wpre;
    sa_wprefetch((const h16_t*)a.WkvT, 2 * SA_D / 16, 0, wave, wpre);
    {   // stage both halves of the dkv rows and the LayerNorm input rows (coalesced 16-byte loads, all in flight together)
        uint4 rx[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
            rx[it] = row < nvalid ? *reinterpret_cast<const uint4*>((const h16_t*)a.xkv + (size_t)(m0 + row) * SA_D + ch * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = threadIdx.x + it * NT, row = e >> 5, ch = e & 31;
            *reinterpret_cast<uint4*>(sX + row * ALD + ch * 8) = rx[it];
        }
        uint4 r[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = threadIdx.x + it * NT, part = e >> 11, row = (e >> 5) & 63, ch = e & 31;
            r[it] = row < nvalid ? *reinterpret_cast<const uint4*>((const h16_t*)a.dkv + (size_t)(m0 + row) * (2 * SA_D) + part * SA_D + ch * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = threadIdx.x + it * NT, part = e >> 11, row = (e >> 5) & 63, ch = e & 31;
            *reinterpret_cast<uint4*>(actA + part * TOK * ALD + row * ALD + ch * 8) = r[it];
        }
    }
    __syncthreads();
    f32x16_t acc[NJ][RB];
    sa_zero<RB, NJ>(acc);
    sa_gemm_unit<RB, NJ>((const h16_t*)a.WkvT, 2 * SA_D / 16, 0, wave, actA, acc, wpre);
    sa_wprefetch((const h16_t*)a.WkvT, 2 * SA_D / 16, 16, wave, wpre);
    sa_gemm_unit<RB, NJ>((const h16_t*)a.WkvT, 2 * SA_D / 16, 16, wave, actA + TOK * ALD, acc, wpre);
    if (wave < 2) sa_wprefetch((const h16_t*)a.W2T, SA_D / 16, 0, wave, wpre);
    // the unfused path stores dnk as h16 before the LayerNorm backward: round the same way
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][i][r] = h16_to_f32(f32_to_h16(acc[0][i][r]));
    sa_layernorm_bwd<RB, NJ, true, true>(acc, reinterpret_cast<const float*>(sX), a.mean, a.rstd, a.lnkv_g, sStat2, a.pgrad_kv + (size_t)blockIdx.x * 2 * SA_D, m0, nvalid);
    // (the exchange barrier inside guarantees every wave has finished reading the dk / dv rows)
    sa_store_h16<RB, NJ>(acc, actA, 0, nullptr, SA_D, 0, m0, nvalid);
    __syncthreads();
    sa_tile_store_rows<TOK, NT>(actA, (h16_t*)a.dxkv, SA_D, 0, m0, nvalid);
    // ---- da1 = dxkv . W2   (64 hidden channels: waves 0 and 1), stored as h16 like the unfused dgrad output
    if (wave < 2) {
        sa_zero<RB, NJ>(acc);
        sa_gemm_unit<RB, NJ>((const h16_t*)a.W2T, SA_D / 16, 0, wave, actA, acc, wpre);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = pack_h16x2(acc[0][i][4 * g + 0], acc[0][i][4 * g + 1]);
                u.y = pack_h16x2(acc[0][i][4 * g + 2], acc[0][i][4 * g + 3]);
                if (i * 32 + t < nvalid) *reinterpret_cast<uint2*>((h16_t*)a.da1 + (size_t)(m0 + i * 32 + t) * 64 + 32 * wave + 8 * g + 4 * hl) = u;
            }
    }
}

// rows of kv-LayerNorm parameter-gradient partials (2 D floats each) vpf_adapter_kv_bwd writes for M points
extern "C" int vpf_adapter_kv_pgrad_rows(long M, int D)
{
    const int tok = D == SA_D ? 64 : sa_rows_adapter_kv_tokens(D);
    return (int)((M + tok - 1) / tok);
}
extern "C" int vpf_adapter_kv_bwd(const VpfAdapterKvBwd* args, void* stream)
{
    (void)hipGetLastError();
    if (!args) return VPF_ERR_NULL;
    const VpfAdapterKvBwd& a = *args;
    if (!a.dkv || !a.WkvT || !a.xkv || !a.mean || !a.rstd || !a.lnkv_g || !a.W2T || !a.dxkv || !a.da1 || !a.pgrad_kv) return VPF_ERR_NULL;
    if (a.M <= 0) return VPF_ERR_BADSHAPE;
    if (a.D != SA_D) return sa_rows_adapter_kv_bwd_launch(a, (hipStream_t)stream);
    if (!(vpf_debug().sa_wg2 & 4)) return sa_rows_adapter_kv_bwd_launch(a, (hipStream_t)stream);      // bit 2 set: the round-2 kernel below
    constexpr int TOK = 64;
    const size_t lds = (size_t)2 * TOK * ALD * 2 + (size_t)TOK * 8 * 2 * 4 + (size_t)TOK * ALD * 2;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)adapter_kv_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL(adapter_kv_bwd_kernel, dim3((int)vpf_cdiv(a.M, (long)TOK)), dim3(512), lds, (hipStream_t)stream, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
