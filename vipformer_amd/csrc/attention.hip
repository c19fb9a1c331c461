// attention.hip -- fused multi-head attention forward / backward for gfx950 (head dim 64).
//
// Replaces MultiHeadAttention.forward's einsum -> scale -> softmax -> dropout -> einsum chain
// (vipformer/model/pointcloud/partseg.py:67-86) and its autograd backward.  Flash-style: the
// [b*h, Lq, Lkv] score matrix never reaches HBM; forward keeps log-sum-exp per query row and
// backward recomputes the probabilities.  Dropout on the probabilities (partseg.py:80) is a
// counter-based function of (rng state, site, (bh, q, kv)) so backward regenerates the mask.
//
// Layout ("swapped" products, key/value index in registers, query on the lane):
//   S^T[kv,q] = K[kv,:] . Q[q,:]      v_mfma_f32_32x32x16_f16, A = K tile rows (LDS, ds_read_b128),
//                                     B = Q fragment (registers, loaded once)
//   softmax statistics are per-lane scalars (query = lane & 31; partner lane ^ 32 holds the other
//   16 keys of a 32-key sub-tile), P^T accumulators are re-used in place as the B operand of
//   O^T[d,q] += V^T[d,kv] . P^T[kv,q]  with V^T fragments fetched from the natural [kv][d] LDS tile
//   by the gfx950 transposing read ds_read_b64_tr_b16 (no transposed copy of V anywhere).
// One workgroup = one (batch, head); one 64-lane wave = 32 query rows; K/V tiles are shared by
// the workgroup's waves and double-buffered in LDS (global loads of tile t+1 in flight during
// the matrix work on tile t).
#include "vpf_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define DH 64
#define KLD 72          // LDS row stride (h16) of the K / V / Q / dO tiles: 64 + 8 pad
#define TLD 40          // LDS row stride (h16) of the per-wave P / dS scratch: 32 + 8 pad
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

struct AttnArgs {
    const h16_t* Q; const h16_t* K; const h16_t* V;
    long ldq, ldk, ldv;           // row strides (elements); head h starts at column h*64
    h16_t* O; long ldo;
    float* LSE;                   // [B,H,Lq]
    int B, H, Lq, Lkv;
    float scale;
    const uint32_t* rng; uint32_t site; float p;
    // backward
    const h16_t* dO; long lddo;
    h16_t* dQ; h16_t* dK; h16_t* dV; long lddq, lddk, lddv;
    // optional key padding mask [B, Lkv], non-zero = padding key (partseg.py:73-76).  Only the tiled kernels read it.
    const uint8_t* pad;
    // 1: the score tensor has fewer than 2^32 elements and Lkv % 4 == 0 -- dropout groups are indexed in 32 bits and always aligned
    // (vpf_keep4_32: the same masks as the 64-bit path, ~7 VALU instructions fewer per group of four scores)
    int rng_fast;
};
// A padded key's score is the most negative finite float, as masked_fill_(pad_mask, -finfo.max) leaves it: exp() of it is 0 beside any
// real key, and a row whose keys are ALL padded comes out uniform over its Lkv keys.  Such a row's log-sum-exp (~ -2.4e38) cannot carry
// log(Lkv) any more: the backward kernels recognise it by PAD_ROW_LSE2 and take p = 1 / Lkv.  No gradient reaches a padded score
// (masked_fill_ overwrote it): dS = 0 there, dV still receives p * dO.
#define PAD_SCORE (-3.402823466e+38f)
#define PAD_ROW_LSE2 (-1.0e30f)

__device__ __forceinline__ s16x4_t lds_tr16(const h16_t* p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
}
// A-operand fragment (rows = 32 consecutive columns c0.. of a natural [k][col] LDS tile, k-slots in the
// PERMUTED order of an accumulator-as-operand chain): slot (h,j) <-> k = kbase + 8*(j>>2) + 4h + (j&3)
__device__ __forceinline__ h16x8_t frag_tr_perm(const h16_t* S, int ld, int kbase, int c0)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, coff = 16 * (g & 1);
    const h16_t* a = S + (kbase + 4 * h + q) * ld + c0 + coff + 4 * p;
    const s16x4_t lo = lds_tr16(a), hi = lds_tr16(a + 8 * ld);
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(h16x8_t, v);
}
// natural k order: slot (h,j) <-> k = kbase + 8h + j   (operand element (k, c0 + (lane&31)))
__device__ __forceinline__ h16x8_t frag_tr_nat(const h16_t* S, int ld, int kbase, int c0)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, coff = 16 * (g & 1);
    const h16_t* a = S + (kbase + 8 * h + q) * ld + c0 + coff + 4 * p;
    const s16x4_t lo = lds_tr16(a), hi = lds_tr16(a + 4 * ld);
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(h16x8_t, v);
}
// row fragment: element (row0 + (lane&31), kbase + 8h + j) of a [row][k] LDS tile
__device__ __forceinline__ h16x8_t frag_row(const h16_t* S, int ld, int row0, int kbase)
{
    const int lane = threadIdx.x & 63;
    const uint4 v = *reinterpret_cast<const uint4*>(S + (row0 + (lane & 31)) * ld + kbase + 8 * (lane >> 5));
    return __builtin_bit_cast(h16x8_t, v);
}
__device__ __forceinline__ h16x8_t pack8(const float* f)
{
    uint4 u;
    u.x = pack_h16x2(f[0], f[1]); u.y = pack_h16x2(f[2], f[3]); u.z = pack_h16x2(f[4], f[5]); u.w = pack_h16x2(f[6], f[7]);
    return __builtin_bit_cast(h16x8_t, u);
}
__device__ __forceinline__ uint4 ld16_or_zero(const h16_t* p, bool ok)
{
    return ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
}

// every lane reads quad lane E's value (DPP quad_perm broadcast)
template <int E>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, E | (E << 2) | (E << 4) | (E << 6), 0xF, 0xF, true);
}

// cooperative stage of a [ROWS x 64] h16 tile (rows r0.. of a [L, ld] matrix, head column offset applied by caller)
template <int ROWS, int MAXC>
__device__ __forceinline__ void kv_load(const h16_t* __restrict__ G, long ld, int L, int r0, uint4 (&regs)[MAXC], int nthreads)
{
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * nthreads;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < ROWS * 8) { const int row = c >> 3, ch = c & 7; if (r0 + row < L) v = *reinterpret_cast<const uint4*>(G + (size_t)(r0 + row) * ld + ch * 8); }
        regs[i] = v;
    }
}
template <int ROWS, int MAXC>
__device__ __forceinline__ void kv_store(h16_t* __restrict__ S, const uint4 (&regs)[MAXC], int nthreads)
{
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * nthreads;
        if (c < ROWS * 8) { const int row = c >> 3, ch = c & 7; *reinterpret_cast<uint4*>(S + row * KLD + ch * 8) = regs[i]; }
    }
}

// MODE (round 4): the dropout variant is a TEMPLATE parameter, so the loops over the resident blocks are straight-line code the
// compiler schedules across blocks -- the run-time tests cost a scalar branch per group of four scores (phase B of the backward: two
// per score) that cut every block into pieces.  RES_GENERAL keeps every run-time test (64-bit group indices, L % 4 != 0).
enum { RES_GENERAL = 0, RES_DROP32 = 1, RES_NODROP = 2 };
static inline int res_mode(const AttnArgs& a) { return a.p > 0.f ? (a.rng_fast ? RES_DROP32 : RES_GENERAL) : RES_NODROP; }
// =============================================================================== forward
#define FWD_MAXC 3  // ceil(64*8 / 192) chunks per thread at the smallest block (3 waves); 1-wave blocks loop 8x below
// FWD_KT = kv rows per LDS stage (64 = two 32-row sub-tiles; 128 when there are many keys)
// MODE: RES_DROP32 (training: dropout on, 32-bit group indices, no padding mask) or RES_GENERAL (every run-time test)
// KS (round 4): the keys of a stage are split over KS waves per query block (QB * KS waves in all; wave = ks * QB + qb takes the
// 32-key sub-tiles ks, ks + KS, .. of every stage) and the KS partial (max, sum, output) triples are merged through LDS at the end.
// A few-queries-many-keys attention (96 latents against 1024 points) otherwise runs three waves per (cloud, head), each one serial
// chain S -> softmax -> P V over 32 blocks with nothing to hide its latencies behind.
template <int QB, int FWD_KT, int MODE, int KS>
__global__ void __launch_bounds__(QB * KS * 64) attn_fwd_kernel(AttnArgs a)
{
    constexpr int NW = QB * KS, NT = NW * 64;
    constexpr int MAXC = (FWD_KT * 8 + NT - 1) / NT;
    static_assert((FWD_KT / 32) % KS == 0 && (KS & (KS - 1)) == 0, "the sub-tiles of a stage divide over the key splits");
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];               // [2 buf][K,V][FWD_KT][KLD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5;
    const int qb = KS == 1 ? wave : wave % QB, ks = KS == 1 ? 0 : wave / QB;
    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H;
    const int q = (blockIdx.y * QB + qb) * 32 + (lane & 31);
    const bool qok = q < a.Lq;
    const h16_t* Kg = a.K + (size_t)b * a.Lkv * a.ldk + hd * DH;
    const h16_t* Vg = a.V + (size_t)b * a.Lkv * a.ldv + hd * DH;

    h16x8_t qf[4];
    {
        const h16_t* qp = a.Q + ((size_t)b * a.Lq + (qok ? q : 0)) * a.ldq + hd * DH + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(qp + ks * 16, qok));
    }
    f32x16_t o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = MODE == RES_DROP32 ? true : a.p > 0.f;
    const uint64_t rbase = ((uint64_t)bh * a.Lq + (uint64_t)(qok ? q : 0)) * (uint64_t)a.Lkv;
    const uint8_t* padrow = MODE == RES_GENERAL && a.pad ? a.pad + (size_t)b * a.Lkv : nullptr;

    uint4 rk[MAXC], rv[MAXC];
    const int nt = (a.Lkv + FWD_KT - 1) / FWD_KT;
    kv_load<FWD_KT, MAXC>(Kg, a.ldk, a.Lkv, 0, rk, NT);
    kv_load<FWD_KT, MAXC>(Vg, a.ldv, a.Lkv, 0, rv, NT);
    kv_store<FWD_KT, MAXC>(lds, rk, NT);
    kv_store<FWD_KT, MAXC>(lds + FWD_KT * KLD, rv, NT);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const h16_t* sK = lds + (t & 1) * 2 * FWD_KT * KLD;
        const h16_t* sV = sK + FWD_KT * KLD;
        if (t + 1 < nt) {
            kv_load<FWD_KT, MAXC>(Kg, a.ldk, a.Lkv, (t + 1) * FWD_KT, rk, NT);
            kv_load<FWD_KT, MAXC>(Vg, a.ldv, a.Lkv, (t + 1) * FWD_KT, rv, NT);
        }
#pragma unroll
        for (int si = 0; si < FWD_KT / 32 / KS; ++si) {
            const int sub = ks + si * KS;
            const int kv0 = t * FWD_KT + sub * 32;
            if (kv0 >= a.Lkv) break;
            f32x16_t s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s = vpf_mfma32(frag_row(sK, KLD, sub * 32, ks * 16), qf[ks], s);
            float tmax = -INFINITY, mn, alpha, ps = 0.f;
            float pv[16];
            if (MODE == RES_DROP32) {
                // raw scores: the softmax scale rides in the exponent's fused multiply-add; the key bound is tested in the last block only
                if (kv0 + 32 > a.Lkv) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = (kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl) < a.Lkv ? s[r] : -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                mn = fmaxf(m, tmax * c);                   // finite: every sub-tile has >= 1 valid key (c > 0)
                alpha = vpf_exp2(m - mn);
                m = mn;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const uint2 w = vpf_rand4x16_32(rng, ((uint32_t)rbase + (uint32_t)(kv0 + 8 * g4 + 4 * hl)) >> 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        const float pr = vpf_exp2(fmaf(s[r], c, -mn));
                        ps += pr;
                        const uint32_t word = (e & 2) ? w.y : w.x;
                        const uint32_t fld = (e & 1) ? (word >> 16) : (word & 0xffffu);
                        pv[r] = pr * (fld >= rng.thresh ? rng.scale : 0.f);
                    }
                }
            } else {
                // the same arithmetic with every run-time test (equal masks give bit-identical results); a padded key's SCALED score is
                // PAD_SCORE: it enters the row maximum as that constant and its exponent is PAD_SCORE - max, exactly 0 when every key
                // so far was padded
                uint32_t padbits = 0u;
                if (kv0 + 32 > a.Lkv) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = (kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl) < a.Lkv ? s[r] : -INFINITY;
                }
                if (padrow) {                              // (block-uniform; the unmasked path is untouched)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                        if (kv < a.Lkv && padrow[kv]) { padbits |= 1u << r; s[r] = -INFINITY; }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax *= c;                                 // (-inf stays -inf: c > 0)
                if (padbits) tmax = fmaxf(tmax, PAD_SCORE);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                mn = fmaxf(m, tmax);                       // finite: every sub-tile has >= 1 valid key
                alpha = vpf_exp2(m - mn);
                m = mn;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const uint32_t keep = drop ? (a.rng_fast ? vpf_keep4_32(rng, ((uint32_t)rbase + (uint32_t)(kv0 + 8 * g4 + 4 * hl)) >> 2) : vpf_keep4_at(rng, rbase + (uint64_t)(kv0 + 8 * g4 + 4 * hl))) : 15u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        float arg = fmaf(s[r], c, -mn);
                        if (padrow) arg = ((padbits >> r) & 1u) ? PAD_SCORE - mn : arg;
                        const float pr = vpf_exp2(arg);
                        ps += pr;
                        pv[r] = pr * (((keep >> e) & 1u) ? rng.scale : 0.f);      // p = 0: keep = 15, scale = 1
                    }
                }
            }
            l = l * alpha + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h16x8_t pf = pack8(pv + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = vpf_mfma32(frag_tr_perm(sV, KLD, sub * 32 + 16 * s2, dt * 32), pf, o[dt]);
            }
        }
        if (t + 1 < nt) {
            h16_t* nK = lds + ((t + 1) & 1) * 2 * FWD_KT * KLD;
            kv_store<FWD_KT, MAXC>(nK, rk, NT);
            kv_store<FWD_KT, MAXC>(nK + FWD_KT * KLD, rv, NT);
        }
        __syncthreads();
    }
    if (KS > 1) {
        // merge tree over the key splits (the loop above ended on a barrier: the stage buffers are free).  Slot layout [34][64]: element-
        // major, one float per lane -- conflict-free.  A split that saw no key at all (Lkv <= 32 * ks) carries m = -inf, l = 0, o = 0.
        float* sM = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int step = KS / 2; step >= 1; step >>= 1) {
            if (ks >= step && ks < 2 * step) {
                float* slot = sM + (size_t)((ks - step) * QB + qb) * 34 * 64 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) { slot[r * 64] = o[0][r]; slot[(16 + r) * 64] = o[1][r]; }
                slot[32 * 64] = m; slot[33 * 64] = l;
            }
            __syncthreads();
            if (ks < step) {
                const float* slot = sM + (size_t)(ks * QB + qb) * 34 * 64 + lane;
                const float m2 = slot[32 * 64], l2 = slot[33 * 64];
                const float mn = fmaxf(m, m2);
                const float a1 = m == -INFINITY ? 0.f : vpf_exp2(m - mn), a2 = m2 == -INFINITY ? 0.f : vpf_exp2(m2 - mn);
                l = l * a1 + l2 * a2;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[0][r] = o[0][r] * a1 + slot[r * 64] * a2;
                    o[1][r] = o[1][r] * a1 + slot[(16 + r) * 64] * a2;
                }
                m = mn;
            }
            if (step > 1) __syncthreads();
        }
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = 1.f / lt;
    if (qok && ks == 0) {
        h16_t* op = a.O + ((size_t)b * a.Lq + q) * a.ldo + hd * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u;
                u.x = pack_h16x2(o[dt][4 * gq + 0] * inv, o[dt][4 * gq + 1] * inv);
                u.y = pack_h16x2(o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv);
                *reinterpret_cast<uint2*>(op + dt * 32 + 8 * gq + 4 * hl) = u;
            }
        if (hl == 0 && a.LSE) a.LSE[(size_t)bh * a.Lq + q] = (m + log2f(lt)) * LN2;
    }
}

template <int QB, int KT, int KS>
static int launch_fwd_ks(const AttnArgs& a, hipStream_t st)
{
    dim3 grid(a.B * a.H, vpf_cdiv(a.Lq, 32 * QB));
    const bool d32 = res_mode(a) == RES_DROP32 && !a.pad;
    constexpr size_t lds = sizeof(h16_t) * 2 * 2 * KT * KLD;
    static_assert(KS == 1 || sizeof(float) * (KS / 2) * QB * 34 * 64 <= lds, "the merge slots reuse the stage buffers");
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (lds > 65536)
            for (const void* f : {(const void*)attn_fwd_kernel<QB, KT, RES_GENERAL, KS>, (const void*)attn_fwd_kernel<QB, KT, RES_DROP32, KS>})
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    if (d32) hipLaunchKernelGGL((attn_fwd_kernel<QB, KT, RES_DROP32, KS>), grid, dim3(QB * KS * 64), lds, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<QB, KT, RES_GENERAL, KS>), grid, dim3(QB * KS * 64), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
template <int NW>
static int launch_fwd(const AttnArgs& a, hipStream_t st)
{
    if (a.Lkv >= 256) {
        if constexpr (NW <= 4) {          // few queries, many keys: split the keys over the waves too (VPF_ATTN_KSPLIT: 1 / 2 / 4)
            const int ksplit = a.Lkv >= 512 ? vpf_debug().attn_ksplit : 1;
            if constexpr (NW <= 3) if (ksplit >= 4) return launch_fwd_ks<NW, 128, 4>(a, st);      // (4 x 4 waves would leave 128 registers a lane: spills)
            if (ksplit >= 2) return launch_fwd_ks<NW, 128, 2>(a, st);
        }
        return launch_fwd_ks<NW, 128, 1>(a, st);
    }
    return launch_fwd_ks<NW, 64, 1>(a, st);
}

// =============================================================================== resident self-attention, forward
// Lq == Lkv <= 224 (the encoder's self-attention: 96 latents / 196 patches): the head's whole K and V tiles are staged in
// ONE round of loads (all of them in flight together) and every wave walks them without further barriers.  The tiled
// kernel above pays one global-load latency plus a barrier per 64 keys, which is all there is at these sizes.
template <int NW, int MODE>
__global__ void __launch_bounds__(NW * 64) attn_res_fwd_kernel(AttnArgs a)
{
    constexpr int NT = NW * 64, LPT = NW * 32;
    constexpr int NCH = 2 * LPT * 8, CPT = (NCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) h16_t rlds[];
    h16_t* sK = rlds;
    h16_t* sV = rlds + LPT * KLD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H, L = a.Lq;
    const int q = wave * 32 + (lane & 31);
    const bool qok = q < L;
    uint4 kvr[CPT];
#pragma unroll
    for (int it = 0; it < CPT; ++it) {
        const int e = threadIdx.x + it * NT, ch = e & 7, row = (e >> 3) % LPT, kv = (e >> 3) / LPT;
        kvr[it] = make_uint4(0, 0, 0, 0);
        if (e < NCH && row < L) {
            const h16_t* src = kv ? a.V + ((size_t)b * L + row) * a.ldv : a.K + ((size_t)b * L + row) * a.ldk;
            kvr[it] = *reinterpret_cast<const uint4*>(src + hd * DH + ch * 8);
        }
    }
    h16x8_t qf[4];
    {
        const h16_t* qp = a.Q + ((size_t)b * L + (qok ? q : 0)) * a.ldq + hd * DH + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(qp + ks * 16, qok));
    }
#pragma unroll
    for (int it = 0; it < CPT; ++it) {
        const int e = threadIdx.x + it * NT, ch = e & 7, row = (e >> 3) % LPT, kv = (e >> 3) / LPT;
        if (e < NCH) *reinterpret_cast<uint4*>(rlds + (kv * LPT + row) * KLD + ch * 8) = kvr[it];
    }
    __syncthreads();

    f32x16_t o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = MODE == RES_DROP32 ? true : MODE == RES_NODROP ? false : a.p > 0.f;
    const uint64_t rbase = ((uint64_t)bh * L + (uint64_t)(qok ? q : 0)) * (uint64_t)L;
    // Round 4: the whole score row of a query is resident (<= 7 blocks of 32 keys = 112 registers per lane), so the softmax is TWO-PASS
    // instead of online: pass 1 forms every S block and the row maximum; pass 2 exponentiates against that one maximum (the softmax
    // scale rides in the same fused multiply-add), draws the dropout decisions and runs P V.  Gone per block of 32 keys: the running
    // maximum's exponential, the rescaling of the 32 output accumulators and of the running sum, the per-score scale multiply and --
    // in every block but the last -- the key-bound select; the dropout decision compares the hash's 16-bit fields directly instead of
    // assembling a 4-bit mask and taking it apart again (NOTES.md round 4: 27 -> 17 VALU instructions per score).
    f32x16_t sc[NW];
    float tmax = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < NW; ++blk) {
        const int kv0 = blk * 32;
        if (kv0 >= L) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[blk][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            sc[blk] = vpf_mfma32(frag_row(sK, KLD, kv0, ks * 16), qf[ks], sc[blk]);
        if (kv0 + 32 > L) {                                   // (only the last block can hold keys beyond L: a uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                sc[blk][r] = kv < L ? sc[blk][r] : -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[blk][r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m = tmax * c;                                  // the row maximum of the scaled scores, in log2 units (c > 0)
    float l = 0.f;
#pragma unroll
    for (int blk = 0; blk < NW; ++blk) {
        const int kv0 = blk * 32;
        if (kv0 >= L) break;
        float pv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const uint32_t i0 = (uint32_t)rbase + (uint32_t)(kv0 + 8 * g4 + 4 * hl);
            if (MODE == RES_NODROP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    pv[r] = vpf_exp2(fmaf(sc[blk][r], c, -m));
                    l += pv[r];
                }
            } else if (MODE == RES_DROP32 || !drop || a.rng_fast) {
                // aligned groups of four scores: the hash's four 16-bit uniforms against the threshold (p = 0: no hash, everything kept)
                uint2 w = make_uint2(0xffffffffu, 0xffffffffu);
                if (drop) w = vpf_rand4x16_32(rng, i0 >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const float pr = vpf_exp2(fmaf(sc[blk][r], c, -m));
                    l += pr;
                    const uint32_t word = (e & 2) ? w.y : w.x;
                    const uint32_t fld = (e & 1) ? (word >> 16) : (word & 0xffffu);
                    pv[r] = pr * (fld >= rng.thresh ? rng.scale : 0.f);
                }
            } else {
                const uint32_t keep = vpf_keep4_at(rng, rbase + (uint64_t)(kv0 + 8 * g4 + 4 * hl));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const float pr = vpf_exp2(fmaf(sc[blk][r], c, -m));
                    l += pr;
                    pv[r] = ((keep >> e) & 1u) ? pr * rng.scale : 0.f;
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const h16x8_t pf = pack8(pv + 8 * s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                o[dt] = vpf_mfma32(frag_tr_perm(sV, KLD, kv0 + 16 * s2, dt * 32), pf, o[dt]);
        }
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = 1.f / lt;
    {
        // the wave's [32 x 64] output tile goes through its own LDS staging rows and leaves as 128-byte rows (16 bytes per lane,
        // 8 rows per wave-instruction) instead of 8 bytes per lane into 32 different rows
        // (the staging rows take over the K tile once every wave is past its last Q K^T: 2 instead of 3 tiles of LDS -- 64 KB at 196
        // patches, so two workgroups share a CU)
        __syncthreads();
        h16_t* sO = rlds + wave * 32 * KLD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u;
                u.x = pack_h16x2(o[dt][4 * gq + 0] * inv, o[dt][4 * gq + 1] * inv);
                u.y = pack_h16x2(o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv);
                *reinterpret_cast<uint2*>(sO + (lane & 31) * KLD + dt * 32 + 8 * gq + 4 * hl) = u;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave: LDS operations complete in order
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + (lane >> 3), row = wave * 32 + rl, ch = (lane & 7) * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(sO + rl * KLD + ch);
            if (row < L) *reinterpret_cast<uint4*>(a.O + ((size_t)b * L + row) * a.ldo + hd * DH + ch) = v;
        }
    }
    if (qok) {
        if (hl == 0 && a.LSE) a.LSE[(size_t)bh * L + q] = (m + log2f(lt)) * LN2;
    }
}
template <int NW>
static int launch_res_fwd(const AttnArgs& a, hipStream_t st)
{
    constexpr size_t lds = (size_t)2 * NW * 32 * KLD * sizeof(h16_t);      // K and V (the per-wave output staging rows reuse K)
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (lds > 65536)
            for (const void* f : {(const void*)attn_res_fwd_kernel<NW, RES_GENERAL>, (const void*)attn_res_fwd_kernel<NW, RES_DROP32>, (const void*)attn_res_fwd_kernel<NW, RES_NODROP>})
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    switch (res_mode(a)) {
    case RES_DROP32: hipLaunchKernelGGL((attn_res_fwd_kernel<NW, RES_DROP32>), dim3(a.B * a.H), dim3(NW * 64), lds, st, a); break;
    case RES_NODROP: hipLaunchKernelGGL((attn_res_fwd_kernel<NW, RES_NODROP>), dim3(a.B * a.H), dim3(NW * 64), lds, st, a); break;
    default: hipLaunchKernelGGL((attn_res_fwd_kernel<NW, RES_GENERAL>), dim3(a.B * a.H), dim3(NW * 64), lds, st, a);
    }
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

static int check_common(const AttnArgs& a)
{
    if (!a.Q || !a.K || !a.V || !a.rng) return VPF_ERR_NULL;
    if (a.B <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lkv <= 0) return VPF_ERR_BADSHAPE;
    if ((a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) || ((uintptr_t)a.Q & 15) || ((uintptr_t)a.K & 15) || ((uintptr_t)a.V & 15))
        return VPF_ERR_BADALIGN;
    return VPF_OK;
}

static int attention_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, int B, int H,
                         int Lq, int Lkv, int head_dim, float scale, float dropout_p, const uint32_t* rng_state,
                         uint32_t site, void* out, long ldo, float* lse, const uint8_t* pad, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (head_dim != DH) return VPF_ERR_UNSUPPORTED;
    AttnArgs a = {};
    a.pad = pad;
    a.rng_fast = vpf_debug().attn_rng32 && ((unsigned long long)B * (unsigned long long)H * (unsigned long long)Lq * (unsigned long long)Lkv < (1ull << 32)) && (Lkv % 4 == 0);
    a.Q = (const h16_t*)q; a.K = (const h16_t*)k; a.V = (const h16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
    a.O = (h16_t*)out; a.ldo = ldo; a.LSE = lse; a.B = B; a.H = H; a.Lq = Lq; a.Lkv = Lkv; a.scale = scale;
    a.rng = rng_state; a.site = site; a.p = dropout_p;
    int rc = check_common(a);
    if (rc) return rc;
    if (!out) return VPF_ERR_NULL;
    if ((ldo % 4) || ((uintptr_t)out & 7)) return VPF_ERR_BADALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int nqb = vpf_cdiv(Lq, 32);
    const int res = vpf_debug().attn_resident;
    if (res && Lq == Lkv && k != q && !pad) {     // self-attention with the whole head resident in LDS
        if (nqb == 3) return launch_res_fwd<3>(a, st);
        if (nqb == 4) return launch_res_fwd<4>(a, st);       // 128 latents: BASELINE configs 3 and 4
        if (nqb == 5) return launch_res_fwd<5>(a, st);       // 144 tokens: the reference's shipped 144 x 144 / patch 12 image geometry
        if (nqb == 7) return launch_res_fwd<7>(a, st);
    }
    if (nqb <= 1) return launch_fwd<1>(a, st);
    if (nqb <= 2) return launch_fwd<2>(a, st);
    if (nqb <= 3) return launch_fwd<3>(a, st);
    if (nqb <= 4) return launch_fwd<4>(a, st);
    if (nqb == 5) return launch_fwd<5>(a, st);
    if (nqb == 6) return launch_fwd<6>(a, st);
    if (nqb == 7) return launch_fwd<7>(a, st);
    return launch_fwd<8>(a, st);
}
extern "C" int vpf_attention_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, int B, int H,
                                 int Lq, int Lkv, int head_dim, float scale, float dropout_p, const uint32_t* rng_state,
                                 uint32_t site, void* out, long ldo, float* lse, void* stream)
{
    return attention_fwd(q, ldq, k, ldk, v, ldv, B, H, Lq, Lkv, head_dim, scale, dropout_p, rng_state, site, out, ldo, lse, nullptr, stream);
}
extern "C" int vpf_attention_fwd_pad(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, int B, int H,
                                     int Lq, int Lkv, int head_dim, float scale, float dropout_p, const uint32_t* rng_state,
                                     uint32_t site, void* out, long ldo, float* lse, const uint8_t* pad_mask, void* stream)
{
    if (!pad_mask) return VPF_ERR_NULL;
    return attention_fwd(q, ldq, k, ldk, v, ldv, B, H, Lq, Lkv, head_dim, scale, dropout_p, rng_state, site, out, ldo, lse, pad_mask, stream);
}

// =============================================================================== backward
// Two kernels, neither of which reduces anything across waves (no LDS / global atomics, no flush phase):
//   attn_bwd_dq_kernel   wave = 32 query rows, loops over K/V tiles exactly like forward:
//                        S^T, dP^T (swapped products), dS^T accumulators feed dQ^T += K^T . dS^T directly.
//                        Also writes delta[q] = rowsum(dO * O) for the second kernel.
//   attn_bwd_dkv_kernel  wave = 32 key rows (K and V fragments live in registers), loops over Q/dO tiles:
//                        S[q,kv], dP[q,kv] with the KEY on the lane, so the P and dS accumulators are already
//                        the B operands of dV^T += dO^T . P and dK^T += Q^T . dS (Q^T / dO^T by transposing reads).
// Recomputing S and dP in both costs 8 extra MFMAs per 32x32 (q,kv) pair; it buys the absence of any
// cross-wave reduction, which dominated an earlier single-kernel version (LDS float atomics).
#define BWD_KT 32
// KT = keys per LDS stage (32 for short sequences; 128 when there are many keys: a barrier and a load latency per stage)
template <int NW, int KT>
__global__ void __launch_bounds__(NW * 64) attn_bwd_dq_kernel(AttnArgs a, float* __restrict__ delta_out)
{
    constexpr int NT = NW * 64;
    constexpr int MAXC = (KT * 8 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) h16_t sKV[];                  // [2 buf][K,V][KT][KLD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, ql = lane & 31;
    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H;
    const int q = (blockIdx.y * NW + wave) * 32 + ql;
    const bool qok = q < a.Lq;
    const h16_t* Kg = a.K + (size_t)b * a.Lkv * a.ldk + hd * DH;
    const h16_t* Vg = a.V + (size_t)b * a.Lkv * a.ldv + hd * DH;

    h16x8_t qf[4], dof[4];
    float delta = 0.f;
    {
        const size_t row = (size_t)b * a.Lq + (qok ? q : 0);
        const h16_t* qp = a.Q + row * a.ldq + hd * DH + 8 * hl;
        const h16_t* dp = a.dO + row * a.lddo + hd * DH + 8 * hl;
        const h16_t* op = a.O + row * a.ldo + hd * DH + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 uq = ld16_or_zero(qp + ks * 16, qok), ud = ld16_or_zero(dp + ks * 16, qok), uo = ld16_or_zero(op + ks * 16, qok);
            qf[ks] = __builtin_bit_cast(h16x8_t, uq); dof[ks] = __builtin_bit_cast(h16x8_t, ud);
            const uint32_t dw[4] = {ud.x, ud.y, ud.z, ud.w}, ow[4] = {uo.x, uo.y, uo.z, uo.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                delta += h16_lo(dw[j]) * h16_lo(ow[j]);
                delta += h16_hi(dw[j]) * h16_hi(ow[j]);
            }
        }
    }
    delta += __shfl_xor(delta, 32, 64);
    if (qok && hl == 0) delta_out[(size_t)bh * a.Lq + q] = delta;
    const float lse2 = qok ? a.LSE[(size_t)bh * a.Lq + q] * LOG2E : 0.f;
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = a.p > 0.f;
    const uint64_t rbase = ((uint64_t)bh * a.Lq + (uint64_t)(qok ? q : 0)) * (uint64_t)a.Lkv;

    const uint8_t* padrow = a.pad ? a.pad + (size_t)b * a.Lkv : nullptr;

    f32x16_t dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }

    uint4 rk[MAXC], rv[MAXC];
    const int nt = (a.Lkv + KT - 1) / KT;
    kv_load<KT, MAXC>(Kg, a.ldk, a.Lkv, 0, rk, NT);
    kv_load<KT, MAXC>(Vg, a.ldv, a.Lkv, 0, rv, NT);
    kv_store<KT, MAXC>(sKV, rk, NT);
    kv_store<KT, MAXC>(sKV + KT * KLD, rv, NT);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const h16_t* sK = sKV + (t & 1) * 2 * KT * KLD;
        const h16_t* sV = sK + KT * KLD;
        const int kv0 = t * KT;
        if (t + 1 < nt) {
            kv_load<KT, MAXC>(Kg, a.ldk, a.Lkv, (t + 1) * KT, rk, NT);
            kv_load<KT, MAXC>(Vg, a.ldv, a.Lkv, (t + 1) * KT, rv, NT);
        }
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int kvs = kv0 + sub * 32;
            if (kvs >= a.Lkv) break;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = vpf_mfma32(frag_row(sK, KLD, sub * 32, ks * 16), qf[ks], s);
                dp = vpf_mfma32(frag_row(sV, KLD, sub * 32, ks * 16), dof[ks], dp);
            }
            float ds[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint32_t kbits = drop ? (a.rng_fast ? vpf_keep4_32(rng, ((uint32_t)rbase + (uint32_t)(kvs + 8 * g4 + 4 * hl)) >> 2) : vpf_keep4_at(rng, rbase + (uint64_t)(kvs + 8 * g4 + 4 * hl))) : 15u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const int kv = kvs + e + 8 * g4 + 4 * hl;
                    const bool ok = qok && kv < a.Lkv;
                    const float ex = vpf_exp2(s[r] * c - lse2);                 // unconditional: a select, not an exec-mask branch per score
                    const float pr = ok ? ex : 0.f;
                    const float keep = ((kbits >> e) & 1u) ? rng.scale : 0.f;      // p = 0: kbits = 15, scale = 1
                    ds[r] = pr * (dp[r] * keep - delta) * a.scale;
                }
            }
            if (padrow) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kvs + (r & 3) + 8 * (r >> 2) + 4 * hl;
                    if (kv < a.Lkv && padrow[kv]) ds[r] = 0.f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h16x8_t dsf = pack8(ds + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = vpf_mfma32(frag_tr_perm(sK, KLD, sub * 32 + 16 * s2, dt * 32), dsf, dq[dt]);
            }
        }
        if (t + 1 < nt) {
            h16_t* nK = sKV + ((t + 1) & 1) * 2 * KT * KLD;
            kv_store<KT, MAXC>(nK, rk, NT);
            kv_store<KT, MAXC>(nK + KT * KLD, rv, NT);
        }
        __syncthreads();
    }
    if (qok) {
        h16_t* op = a.dQ + ((size_t)b * a.Lq + q) * a.lddq + hd * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u;
                u.x = pack_h16x2(dq[dt][4 * gq + 0], dq[dt][4 * gq + 1]);
                u.y = pack_h16x2(dq[dt][4 * gq + 2], dq[dt][4 * gq + 3]);
                *reinterpret_cast<uint2*>(op + dt * 32 + 8 * gq + 4 * hl) = u;
            }
    }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) attn_bwd_dkv_kernel(AttnArgs a, const float* __restrict__ delta_in)
{
    constexpr int NT = NW * 64;
    constexpr int MAXC = (BWD_KT * 8 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) h16_t sQD[2 * 2 * BWD_KT * KLD];     // [2 buf][Q,dO][32][KLD]
    __shared__ float sStat[2 * 2 * BWD_KT];                                        // [2 buf][lse*log2e, delta][32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, kl = lane & 31;
    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H;
    const int kv = (blockIdx.y * NW + wave) * 32 + kl;
    const bool kvok = kv < a.Lkv;
    const h16_t* Qg = a.Q + (size_t)b * a.Lq * a.ldq + hd * DH;
    const h16_t* Dg = a.dO + (size_t)b * a.Lq * a.lddo + hd * DH;
    const float* lseg = a.LSE + (size_t)bh * a.Lq;
    const float* delg = delta_in + (size_t)bh * a.Lq;

    h16x8_t kf[4], vf[4];
    {
        const size_t row = (size_t)b * a.Lkv + (kvok ? kv : 0);
        const h16_t* kp = a.K + row * a.ldk + hd * DH + 8 * hl;
        const h16_t* vp = a.V + row * a.ldv + hd * DH + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(kp + ks * 16, kvok));
            vf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(vp + ks * 16, kvok));
        }
    }
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = a.p > 0.f;
    const bool padded = a.pad && kvok && a.pad[(size_t)b * a.Lkv + kv];

    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = dk[1][r] = dv[0][r] = dv[1][r] = 0.f; }

    uint4 rq[MAXC], rd[MAXC];
    float st0 = 0.f;
    const int nt = (a.Lq + BWD_KT - 1) / BWD_KT;
    auto stat_load = [&](int q0) {
        // threads 0..31: lse * log2e of row q0 + t ; threads 32..63: delta
        float v = 0.f;
        if (threadIdx.x < 64) {
            const int qq = q0 + (threadIdx.x & 31);
            if (qq < a.Lq) v = threadIdx.x < 32 ? lseg[qq] * LOG2E : delg[qq];
        }
        return v;
    };
    kv_load<BWD_KT, MAXC>(Qg, a.ldq, a.Lq, 0, rq, NT);
    kv_load<BWD_KT, MAXC>(Dg, a.lddo, a.Lq, 0, rd, NT);
    st0 = stat_load(0);
    kv_store<BWD_KT, MAXC>(sQD, rq, NT);
    kv_store<BWD_KT, MAXC>(sQD + BWD_KT * KLD, rd, NT);
    if (threadIdx.x < 64) sStat[threadIdx.x] = st0;
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const h16_t* sQ = sQD + (t & 1) * 2 * BWD_KT * KLD;
        const h16_t* sD = sQ + BWD_KT * KLD;
        const float* sL = sStat + (t & 1) * 2 * BWD_KT;
        const int q0 = t * BWD_KT;
        if (t + 1 < nt) {
            kv_load<BWD_KT, MAXC>(Qg, a.ldq, a.Lq, (t + 1) * BWD_KT, rq, NT);
            kv_load<BWD_KT, MAXC>(Dg, a.lddo, a.Lq, (t + 1) * BWD_KT, rd, NT);
            st0 = stat_load((t + 1) * BWD_KT);
        }
        // S[q,kv] and dP[q,kv]: A = Q / dO rows (LDS), B = K / V fragments (registers); key on the lane
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = vpf_mfma32(frag_row(sQ, KLD, 0, ks * 16), kf[ks], s);
            dp = vpf_mfma32(frag_row(sD, KLD, 0, ks * 16), vf[ks], dp);
        }
        float pd[16], ds[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            // Dropout groups run along the key axis, which is the LANE axis here: the 4 lanes of a quad share one group per
            // query.  Each quad lane hashes the group of a different query of this register quad and the results are
            // exchanged with DPP quad broadcasts: one hash per 4 elements instead of one per element.
            uint2 grp = make_uint2(0u, 0u);
            const bool quad_ok = (a.Lkv & 3) == 0;
            const bool slow_keep = drop && !quad_ok;
            if (drop && quad_ok) {
                const int qh = q0 + 8 * g4 + 4 * hl + (lane & 3);
                grp = a.rng_fast ? vpf_rand4x16_32(rng, (((uint32_t)bh * a.Lq + (uint32_t)qh) * (uint32_t)a.Lkv + (uint32_t)kv) >> 2) : vpf_rand4x16(rng, (((uint64_t)bh * a.Lq + (uint64_t)qh) * (uint64_t)a.Lkv + (uint64_t)kv) >> 2);
            }
            uint32_t gw[4];
            {
                const uint32_t mine = (lane & 2) ? 1u : 0u;      // which 32-bit word holds this lane's key (kv & 3)
                const uint32_t x0 = quad_bcast<0>(grp.x), x1 = quad_bcast<1>(grp.x), x2 = quad_bcast<2>(grp.x), x3 = quad_bcast<3>(grp.x);
                const uint32_t y0 = quad_bcast<0>(grp.y), y1 = quad_bcast<1>(grp.y), y2 = quad_bcast<2>(grp.y), y3 = quad_bcast<3>(grp.y);
                gw[0] = mine ? y0 : x0; gw[1] = mine ? y1 : x1; gw[2] = mine ? y2 : x2; gw[3] = mine ? y3 : x3;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                const int qr = e + 8 * g4 + 4 * hl;
                const int q = q0 + qr;
                const bool ok = kvok && q < a.Lq;
                const float ex = vpf_exp2(s[r] * c - sL[qr]);                 // unconditional: a select, not an exec-mask branch per score
                    const float pr = ok ? ex : 0.f;
                const uint32_t word = gw[e];                                   // this lane's half of quad lane e's group (p = 0: thresh 0, scale 1)
                float keep = (((lane & 1) ? (word >> 16) : (word & 0xffffu)) >= rng.thresh) ? rng.scale : 0.f;
                if (slow_keep) keep = vpf_keep(rng, ((uint64_t)bh * a.Lq + (uint64_t)q) * (uint64_t)a.Lkv + (uint64_t)kv) ? rng.scale : 0.f;
                pd[r] = pr * keep;
                ds[r] = pr * (dp[r] * keep - sL[BWD_KT + qr]) * a.scale;
                if (padded) {             // this lane's key is padding: p = 0, or 1 / Lkv in a row whose keys are all padded; dS = 0
                    pd[r] = (ok && sL[qr] < PAD_ROW_LSE2) ? keep / (float)a.Lkv : 0.f;
                    ds[r] = 0.f;
                }
            }
        }
        // dV^T[d,kv] += dO^T[d,q] . P[q,kv] ; dK^T[d,kv] += Q^T[d,q] . dS[q,kv]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const h16x8_t pf = pack8(pd + 8 * s2), sf = pack8(ds + 8 * s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = vpf_mfma32(frag_tr_perm(sD, KLD, 16 * s2, dt * 32), pf, dv[dt]);
                dk[dt] = vpf_mfma32(frag_tr_perm(sQ, KLD, 16 * s2, dt * 32), sf, dk[dt]);
            }
        }
        if (t + 1 < nt) {
            h16_t* nQ = sQD + ((t + 1) & 1) * 2 * BWD_KT * KLD;
            kv_store<BWD_KT, MAXC>(nQ, rq, NT);
            kv_store<BWD_KT, MAXC>(nQ + BWD_KT * KLD, rd, NT);
            if (threadIdx.x < 64) sStat[((t + 1) & 1) * 2 * BWD_KT + threadIdx.x] = st0;
        }
        __syncthreads();
    }
    if (kvok) {
        h16_t* kp = a.dK + ((size_t)b * a.Lkv + kv) * a.lddk + hd * DH;
        h16_t* vp = a.dV + ((size_t)b * a.Lkv + kv) * a.lddv + hd * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u, w;
                u.x = pack_h16x2(dk[dt][4 * gq + 0], dk[dt][4 * gq + 1]); u.y = pack_h16x2(dk[dt][4 * gq + 2], dk[dt][4 * gq + 3]);
                w.x = pack_h16x2(dv[dt][4 * gq + 0], dv[dt][4 * gq + 1]); w.y = pack_h16x2(dv[dt][4 * gq + 2], dv[dt][4 * gq + 3]);
                *reinterpret_cast<uint2*>(kp + dt * 32 + 8 * gq + 4 * hl) = u;
                *reinterpret_cast<uint2*>(vp + dt * 32 + 8 * gq + 4 * hl) = w;
            }
    }
}


// =============================================================================== resident self-attention, backward
// One workgroup per (batch, head) holds Q, K, V and dO of the head in LDS (one round of loads).  Phase A is the dq kernel
// above (a wave = 32 queries, delta computed on the way), phase B the dk / dv kernel (a wave = 32 keys), both walking
// the resident tiles without barriers: one launch and one load latency instead of two kernels that each re-stage
// their operands tile by tile.
template <int NW, int MODE>
__global__ void __launch_bounds__(NW * 64) attn_res_bwd_kernel(AttnArgs a, float* __restrict__ delta_out)
{
    constexpr int NT = NW * 64, LPT = NW * 32;
    constexpr int NCH = 4 * LPT * 8, CPT = (NCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) h16_t rlds[];
    h16_t* sQ = rlds;
    h16_t* sK = sQ + LPT * KLD;
    h16_t* sV = sK + LPT * KLD;
    h16_t* sD = sV + LPT * KLD;
    float* sL = reinterpret_cast<float*>(sD + LPT * KLD);        // [LPT] lse * log2e
    float* sDel = sL + LPT;                                      // [LPT] delta
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, ql = lane & 31;
    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H, L = a.Lq;
    const int q = wave * 32 + ql;                                 // phase A: this lane's query; phase B: this lane's key
    const bool qok = q < L;
#ifdef VPF_EXP_STAMP
    uint64_t stamp[6]; stamp[0] = __builtin_amdgcn_s_memtime();
#define VPF_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VPF_STAMP(i) do {} while (0)
#endif
    {
        uint4 rr[CPT];
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, ch = e & 7, row = (e >> 3) % LPT, which = (e >> 3) / LPT;
            rr[it] = make_uint4(0, 0, 0, 0);
            if (e < NCH && row < L) {
                const size_t gr = (size_t)b * L + row;
                const h16_t* src = which == 0 ? a.Q + gr * a.ldq : which == 1 ? a.K + gr * a.ldk : which == 2 ? a.V + gr * a.ldv : a.dO + gr * a.lddo;
                rr[it] = *reinterpret_cast<const uint4*>(src + hd * DH + ch * 8);
            }
        }
        // O fragments of this lane's query (only needed for delta) and its log-sum-exp
        uint4 of[4];
        {
            const h16_t* op = a.O + ((size_t)b * L + (qok ? q : 0)) * a.ldo + hd * DH + 8 * hl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) of[ks] = ld16_or_zero(op + ks * 16, qok);
        }
        const float lse2 = qok ? a.LSE[(size_t)bh * L + q] * LOG2E : 0.f;
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, ch = e & 7, row = (e >> 3) % LPT, which = (e >> 3) / LPT;
            if (e < NCH) *reinterpret_cast<uint4*>(rlds + (which * LPT + row) * KLD + ch * 8) = rr[it];
        }
        __syncthreads();
        VPF_STAMP(1);
        // delta[q] = sum_d dO[q,d] * O[q,d]
        float delta = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 ud = *reinterpret_cast<const uint4*>(sD + q * KLD + ks * 16 + 8 * hl);
            const uint32_t dw[4] = {ud.x, ud.y, ud.z, ud.w}, ow[4] = {of[ks].x, of[ks].y, of[ks].z, of[ks].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                delta += h16_lo(dw[j]) * h16_lo(ow[j]);
                delta += h16_hi(dw[j]) * h16_hi(ow[j]);
            }
        }
        delta += __shfl_xor(delta, 32, 64);
        if (hl == 0) { sL[q] = lse2; sDel[q] = delta; }
        if (qok && hl == 0) delta_out[(size_t)bh * L + q] = delta;
    }
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = MODE == RES_DROP32 ? true : MODE == RES_NODROP ? false : a.p > 0.f;
    // Round 4 (NOTES.md): the key / query bound is only tested in the LAST block of 32 (the only one that can be partial; rows of invalid
    // lanes are never stored, so their own validity is not tested at all), phase A compares the hash's 16-bit fields directly instead
    // of assembling a 4-bit mask, and carries the softmax scale inside the keep factor and delta.
    // A wave's [32 x 64] result tiles (dQ, dK, dV) leave through ITS OWN 32 rows of the K / V tiles, which only it reads once
    // phase B has its fragments in registers: the accumulator layout would store 8 bytes per lane into 32 different rows,
    // from LDS the same tile goes out as 128-byte rows (16 bytes per lane, 8 rows per wave-instruction).
    uint2 dqp[2][4];
    auto tile_out = [&](h16_t* S, const uint2 (&tp)[2][4], h16_t* G, long ld) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<uint2*>(S + (wave * 32 + ql) * KLD + dt * 32 + 8 * gq + 4 * hl) = tp[dt][gq];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave: LDS operations complete in order
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = wave * 32 + it * 8 + (lane >> 3), ch = (lane & 7) * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(S + row * KLD + ch);
            if (row < L) *reinterpret_cast<uint4*>(G + ((size_t)b * L + row) * ld + hd * DH + ch) = v;
        }
    };
    // ------------------------------------------------------------------ phase A: dQ (lane = query)
    {
        h16x8_t qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = frag_row(sQ, KLD, wave * 32, ks * 16); dof[ks] = frag_row(sD, KLD, wave * 32, ks * 16); }
        const float lse2 = sL[q], delta = sDel[q];       // written by this very lane (hl == 0) or its partner: same wave, in order
        const float delta_s = delta * a.scale, keep_s = rng.scale * a.scale;      // dS = p (dP keep - delta) scale, the scale inside the operands
        const uint64_t rbase = ((uint64_t)bh * L + (uint64_t)(qok ? q : 0)) * (uint64_t)L;
        f32x16_t dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
#pragma unroll
        for (int kv0 = 0; kv0 < LPT; kv0 += 32) {
            if (kv0 >= L) break;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = vpf_mfma32(frag_row(sK, KLD, kv0, ks * 16), qf[ks], s);
                dp = vpf_mfma32(frag_row(sV, KLD, kv0, ks * 16), dof[ks], dp);
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = vpf_exp2(fmaf(s[r], c, -lse2));      // p (unconditional: no exec-mask branch per score)
            if (kv0 + 32 > L) {                                                         // keys beyond L: only in the last block
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = (kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl) < L ? ds[r] : 0.f;
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if (MODE == RES_NODROP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        ds[r] = ds[r] * fmaf(dp[r], a.scale, -delta_s);
                    }
                } else if (MODE == RES_DROP32 || !drop || a.rng_fast) {
                    uint2 w = make_uint2(0xffffffffu, 0xffffffffu);                     // p = 0: thresh 0, scale 1 -- everything kept
                    if (drop) w = vpf_rand4x16_32(rng, ((uint32_t)rbase + (uint32_t)(kv0 + 8 * g4 + 4 * hl)) >> 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        const uint32_t word = (e & 2) ? w.y : w.x;
                        const uint32_t fld = (e & 1) ? (word >> 16) : (word & 0xffffu);
                        ds[r] = ds[r] * fmaf(dp[r], fld >= rng.thresh ? keep_s : 0.f, -delta_s);
                    }
                } else {
                    const uint32_t kbits = vpf_keep4_at(rng, rbase + (uint64_t)(kv0 + 8 * g4 + 4 * hl));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        ds[r] = ds[r] * fmaf(dp[r], ((kbits >> e) & 1u) ? keep_s : 0.f, -delta_s);
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h16x8_t dsf = pack8(ds + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = vpf_mfma32(frag_tr_perm(sK, KLD, kv0 + 16 * s2, dt * 32), dsf, dq[dt]);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                dqp[dt][gq].x = pack_h16x2(dq[dt][4 * gq + 0], dq[dt][4 * gq + 1]);
                dqp[dt][gq].y = pack_h16x2(dq[dt][4 * gq + 2], dq[dt][4 * gq + 3]);
            }
    }
    __syncthreads();                      // sL / sDel of every query are in LDS; no wave reads K / V rows of another wave any more
    // ------------------------------------------------------------------ phase B: dK, dV (lane = key)
    {
        const int kv = q;
        const bool kvok = qok;
        h16x8_t kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = frag_row(sK, KLD, wave * 32, ks * 16); vf[ks] = frag_row(sV, KLD, wave * 32, ks * 16); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments are in registers before the rows are reused
        VPF_STAMP(2);
        tile_out(sK, dqp, a.dQ, a.lddq);
        f32x16_t dk[2], dv[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[0][r] = dk[1][r] = dv[0][r] = dv[1][r] = 0.f; }
        const bool quad_ok = MODE == RES_DROP32 || (L & 3) == 0;
        const bool slow_keep = MODE == RES_GENERAL && drop && !quad_ok;
        const bool rfast = MODE == RES_DROP32 || a.rng_fast;
#pragma unroll
        for (int q0 = 0; q0 < LPT; q0 += 32) {
            if (q0 >= L) break;
            const bool tailq = q0 + 32 > L;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = vpf_mfma32(frag_row(sQ, KLD, q0, ks * 16), kf[ks], s);
                dp = vpf_mfma32(frag_row(sD, KLD, q0, ks * 16), vf[ks], dp);
            }
            float pd[16], ds[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                                        // p, with the four queries' statistics in one 16-byte read each
                const float4 l4 = *reinterpret_cast<const float4*>(sL + q0 + 8 * g4 + 4 * hl);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) pd[4 * g4 + e] = vpf_exp2(fmaf(s[4 * g4 + e], c, -lq[e]));   // unconditional: no exec-mask branch per score
            }
            if (tailq) {                                                            // queries beyond L: only in the last block (uniform branch)
#pragma unroll
                for (int r = 0; r < 16; ++r) pd[r] = (q0 + (r & 3) + 8 * (r >> 2) + 4 * hl) < L ? pd[r] : 0.f;
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 d4 = *reinterpret_cast<const float4*>(sDel + q0 + 8 * g4 + 4 * hl);
                const float dq4[4] = {d4.x, d4.y, d4.z, d4.w};
                if (MODE == RES_NODROP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        ds[r] = pd[r] * (dp[r] - dq4[e]) * a.scale;
                    }
                    continue;
                }
                uint2 grp = make_uint2(0u, 0u);
                if (drop && quad_ok) {
                    const int qh = q0 + 8 * g4 + 4 * hl + (lane & 3);
                    grp = rfast ? vpf_rand4x16_32(rng, (((uint32_t)bh * L + (uint32_t)qh) * (uint32_t)L + (uint32_t)kv) >> 2) : vpf_rand4x16(rng, (((uint64_t)bh * L + (uint64_t)qh) * (uint64_t)L + (uint64_t)kv) >> 2);
                }
                uint32_t gw[4];
                {
                    const uint32_t mine = (lane & 2) ? 1u : 0u;
                    const uint32_t x0 = quad_bcast<0>(grp.x), x1 = quad_bcast<1>(grp.x), x2 = quad_bcast<2>(grp.x), x3 = quad_bcast<3>(grp.x);
                    const uint32_t y0 = quad_bcast<0>(grp.y), y1 = quad_bcast<1>(grp.y), y2 = quad_bcast<2>(grp.y), y3 = quad_bcast<3>(grp.y);
                    gw[0] = mine ? y0 : x0; gw[1] = mine ? y1 : x1; gw[2] = mine ? y2 : x2; gw[3] = mine ? y3 : x3;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const int qq = q0 + e + 8 * g4 + 4 * hl;
                    const float pr = pd[r];
                    const uint32_t word = gw[e];                                   // this lane's half of quad lane e's group (p = 0: thresh 0, scale 1)
                    float keep = (((lane & 1) ? (word >> 16) : (word & 0xffffu)) >= rng.thresh) ? rng.scale : 0.f;
                    if (MODE == RES_GENERAL && slow_keep) keep = vpf_keep(rng, ((uint64_t)bh * L + (uint64_t)qq) * (uint64_t)L + (uint64_t)kv) ? rng.scale : 0.f;
                    pd[r] = pr * keep;
                    ds[r] = pr * (dp[r] * keep - dq4[e]) * a.scale;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h16x8_t pf = pack8(pd + 8 * s2), sf = pack8(ds + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = vpf_mfma32(frag_tr_perm(sD, KLD, q0 + 16 * s2, dt * 32), pf, dv[dt]);
                    dk[dt] = vpf_mfma32(frag_tr_perm(sQ, KLD, q0 + 16 * s2, dt * 32), sf, dk[dt]);
                }
            }
        }
        {
            (void)kvok; (void)kv;
            uint2 tk[2][4], tv[2][4];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    tk[dt][gq].x = pack_h16x2(dk[dt][4 * gq + 0], dk[dt][4 * gq + 1]); tk[dt][gq].y = pack_h16x2(dk[dt][4 * gq + 2], dk[dt][4 * gq + 3]);
                    tv[dt][gq].x = pack_h16x2(dv[dt][4 * gq + 0], dv[dt][4 * gq + 1]); tv[dt][gq].y = pack_h16x2(dv[dt][4 * gq + 2], dv[dt][4 * gq + 3]);
                }
            VPF_STAMP(3);
            tile_out(sK, tk, a.dK, a.lddk);
            tile_out(sV, tv, a.dV, a.lddv);
            VPF_STAMP(4);
#ifdef VPF_EXP_STAMP
            if (lane == 0) for (int i = 1; i < 5; ++i) delta_out[(size_t)bh * L + wave * 8 + i] = (float)(stamp[i] - stamp[0]);
            if (lane == 0) delta_out[(size_t)bh * L + wave * 8] = (float)(stamp[0] & 0xffffff);
#endif
        }
    }
}
template <int NW>
static int launch_res_bwd(const AttnArgs& a, float* delta, hipStream_t st)
{
    constexpr size_t lds = (size_t)4 * NW * 32 * KLD * sizeof(h16_t) + (size_t)2 * NW * 32 * sizeof(float);
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (lds > 65536)
            for (const void* f : {(const void*)attn_res_bwd_kernel<NW, RES_GENERAL>, (const void*)attn_res_bwd_kernel<NW, RES_DROP32>, (const void*)attn_res_bwd_kernel<NW, RES_NODROP>})
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    switch (res_mode(a)) {
    case RES_DROP32: hipLaunchKernelGGL((attn_res_bwd_kernel<NW, RES_DROP32>), dim3(a.B * a.H), dim3(NW * 64), lds, st, a, delta); break;
    case RES_NODROP: hipLaunchKernelGGL((attn_res_bwd_kernel<NW, RES_NODROP>), dim3(a.B * a.H), dim3(NW * 64), lds, st, a, delta); break;
    default: hipLaunchKernelGGL((attn_res_bwd_kernel<NW, RES_GENERAL>), dim3(a.B * a.H), dim3(NW * 64), lds, st, a, delta);
    }
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// dK / dV when the QUERY side is short (cross-attention: 96 latents against 1024 points): the head's whole Q and dO tiles,
// lse and delta are staged in one round of loads; a wave owns 32 keys (K / V fragments in registers) and walks the
// resident queries without barriers.  The tiled kernel above re-stages 32 queries per step behind a barrier.
template <int NW, int QB>
__global__ void __launch_bounds__(NW * 64) attn_bwd_dkv_resq_kernel(AttnArgs a, const float* __restrict__ delta_in)
{
    constexpr int NT = NW * 64, QPT = QB * 32;
    constexpr int NCH = 2 * QPT * 8, CPT = (NCH + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) h16_t sQD[2 * QPT * KLD];
    __shared__ float sL[QPT], sDel[QPT];
    h16_t* sQ = sQD;
    h16_t* sD = sQD + QPT * KLD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, kl = lane & 31;
    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H;
    const int kv = (blockIdx.y * NW + wave) * 32 + kl;
    const bool kvok = kv < a.Lkv;
    {
        uint4 rr[CPT];
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, ch = e & 7, row = (e >> 3) % QPT, which = (e >> 3) / QPT;
            rr[it] = make_uint4(0, 0, 0, 0);
            if (e < NCH && row < a.Lq) {
                const size_t gr = (size_t)b * a.Lq + row;
                const h16_t* src = which == 0 ? a.Q + gr * a.ldq : a.dO + gr * a.lddo;
                rr[it] = *reinterpret_cast<const uint4*>(src + hd * DH + ch * 8);
            }
        }
        for (int qq = threadIdx.x; qq < QPT; qq += NT) {
            sL[qq] = qq < a.Lq ? a.LSE[(size_t)bh * a.Lq + qq] * LOG2E : 0.f;
            sDel[qq] = qq < a.Lq ? delta_in[(size_t)bh * a.Lq + qq] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, ch = e & 7, row = (e >> 3) % QPT, which = (e >> 3) / QPT;
            if (e < NCH) *reinterpret_cast<uint4*>(sQD + (which * QPT + row) * KLD + ch * 8) = rr[it];
        }
    }
    h16x8_t kf[4], vf[4];
    {
        const size_t row = (size_t)b * a.Lkv + (kvok ? kv : 0);
        const h16_t* kp = a.K + row * a.ldk + hd * DH + 8 * hl;
        const h16_t* vp = a.V + row * a.ldv + hd * DH + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(kp + ks * 16, kvok));
            vf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(vp + ks * 16, kvok));
        }
    }
    __syncthreads();
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = a.p > 0.f;
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = dk[1][r] = dv[0][r] = dv[1][r] = 0.f; }
    const bool quad_ok = (a.Lkv & 3) == 0;
    const bool slow_keep = drop && !quad_ok;
#pragma unroll
    for (int q0 = 0; q0 < QPT; q0 += 32) {
        if (q0 >= a.Lq) break;
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = vpf_mfma32(frag_row(sQ, KLD, q0, ks * 16), kf[ks], s);
            dp = vpf_mfma32(frag_row(sD, KLD, q0, ks * 16), vf[ks], dp);
        }
        float pd[16], ds[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            uint2 grp = make_uint2(0u, 0u);
            if (drop && quad_ok) {
                const int qh = q0 + 8 * g4 + 4 * hl + (lane & 3);
                grp = a.rng_fast ? vpf_rand4x16_32(rng, (((uint32_t)bh * a.Lq + (uint32_t)qh) * (uint32_t)a.Lkv + (uint32_t)kv) >> 2) : vpf_rand4x16(rng, (((uint64_t)bh * a.Lq + (uint64_t)qh) * (uint64_t)a.Lkv + (uint64_t)kv) >> 2);
            }
            uint32_t gw[4];
            {
                const uint32_t mine = (lane & 2) ? 1u : 0u;
                const uint32_t x0 = quad_bcast<0>(grp.x), x1 = quad_bcast<1>(grp.x), x2 = quad_bcast<2>(grp.x), x3 = quad_bcast<3>(grp.x);
                const uint32_t y0 = quad_bcast<0>(grp.y), y1 = quad_bcast<1>(grp.y), y2 = quad_bcast<2>(grp.y), y3 = quad_bcast<3>(grp.y);
                gw[0] = mine ? y0 : x0; gw[1] = mine ? y1 : x1; gw[2] = mine ? y2 : x2; gw[3] = mine ? y3 : x3;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                const int qq = q0 + e + 8 * g4 + 4 * hl;
                const bool ok = kvok && qq < a.Lq;
                const float ex = vpf_exp2(s[r] * c - sL[qq]);                 // unconditional: a select, not an exec-mask branch per score
                    const float pr = ok ? ex : 0.f;
                const uint32_t word = gw[e];                                   // this lane's half of quad lane e's group (p = 0: thresh 0, scale 1)
                float keep = (((lane & 1) ? (word >> 16) : (word & 0xffffu)) >= rng.thresh) ? rng.scale : 0.f;
                if (slow_keep) keep = vpf_keep(rng, ((uint64_t)bh * a.Lq + (uint64_t)qq) * (uint64_t)a.Lkv + (uint64_t)kv) ? rng.scale : 0.f;
                pd[r] = pr * keep;
                ds[r] = pr * (dp[r] * keep - sDel[qq]) * a.scale;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const h16x8_t pf = pack8(pd + 8 * s2), sf = pack8(ds + 8 * s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = vpf_mfma32(frag_tr_perm(sD, KLD, q0 + 16 * s2, dt * 32), pf, dv[dt]);
                dk[dt] = vpf_mfma32(frag_tr_perm(sQ, KLD, q0 + 16 * s2, dt * 32), sf, dk[dt]);
            }
        }
    }
    if (kvok) {
        h16_t* kp = a.dK + ((size_t)b * a.Lkv + kv) * a.lddk + hd * DH;
        h16_t* vp = a.dV + ((size_t)b * a.Lkv + kv) * a.lddv + hd * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u, w;
                u.x = pack_h16x2(dk[dt][4 * gq + 0], dk[dt][4 * gq + 1]); u.y = pack_h16x2(dk[dt][4 * gq + 2], dk[dt][4 * gq + 3]);
                w.x = pack_h16x2(dv[dt][4 * gq + 0], dv[dt][4 * gq + 1]); w.y = pack_h16x2(dv[dt][4 * gq + 2], dv[dt][4 * gq + 3]);
                *reinterpret_cast<uint2*>(kp + dt * 32 + 8 * gq + 4 * hl) = u;
                *reinterpret_cast<uint2*>(vp + dt * 32 + 8 * gq + 4 * hl) = w;
            }
    }
}

// dQ, dK and dV of a cross-attention with FEW queries and MANY keys in ONE kernel (96 / 128 latents against 1024 points): a
// workgroup owns one (cloud, head), keeps its Q and dO tiles, lse and delta (computed here from dO and O) resident and walks the
// keys in tiles of 128.  Phase A of a tile is attn_bwd_dkv_resq_kernel's body (a wave owns 32 keys; S and dP with the key on the
// lane; dV^T += dO^T . P, dK^T += Q^T . dS), and leaves the tile's h16 dS in LDS as [key][query] beside the K tile.  Phase B: wave w
// owns query block w and adds K^T . dS^T over the tile's 128 keys into its dQ^T accumulators (both operands by transposing LDS reads,
// in the k order attn_bwd_dq_kernel uses) -- no cross-wave reduction, no atomics; S, dP, the exponentials and the dropout hash are
// computed once instead of twice and K / V cross HBM once.  Same arithmetic and rounding points as the two kernels it replaces.
template <int QB, int MODE>
__global__ void __launch_bounds__(256, 2) attn_bwd_ca_kernel(AttnArgs a)
{
    constexpr int NW = 4, NT = NW * 64, QPT = QB * 32, KT = NW * 32, DLD = QPT + 8;
    constexpr int NCH = 2 * QPT * 8, CPT = (NCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) h16_t sm[];
    h16_t* sQ = sm;                                  // [QPT][KLD]
    h16_t* sD = sQ + QPT * KLD;                      // [QPT][KLD]  dO
    h16_t* sK = sD + QPT * KLD;                      // [KT][KLD]   this tile's keys
    h16_t* sT = sK + KT * KLD;                       // [KT][DLD]   this tile's dS, [key][query]
    float* sL = reinterpret_cast<float*>(sT + KT * DLD);
    float* sDel = sL + QPT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, kl = lane & 31;
    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H;
    {
        uint4 rr[CPT];
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, ch = e & 7, row = (e >> 3) % QPT, which = (e >> 3) / QPT;
            rr[it] = make_uint4(0, 0, 0, 0);
            if (e < NCH && row < a.Lq) {
                const size_t gr = (size_t)b * a.Lq + row;
                const h16_t* src = which == 0 ? a.Q + gr * a.ldq : a.dO + gr * a.lddo;
                rr[it] = *reinterpret_cast<const uint4*>(src + hd * DH + ch * 8);
            }
        }
        if (wave < QB) {                               // delta[q] = rowsum(dO * O), summed as attn_bwd_dq_kernel sums it
            const int q = wave * 32 + kl;
            const bool qok = q < a.Lq;
            const size_t row = (size_t)b * a.Lq + (qok ? q : 0);
            const h16_t* dp = a.dO + row * a.lddo + hd * DH + 8 * hl;
            const h16_t* op = a.O + row * a.ldo + hd * DH + 8 * hl;
            float delta = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 ud = ld16_or_zero(dp + ks * 16, qok), uo = ld16_or_zero(op + ks * 16, qok);
                const uint32_t dw[4] = {ud.x, ud.y, ud.z, ud.w}, ow[4] = {uo.x, uo.y, uo.z, uo.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    delta += h16_lo(dw[j]) * h16_lo(ow[j]);
                    delta += h16_hi(dw[j]) * h16_hi(ow[j]);
                }
            }
            delta += __shfl_xor(delta, 32, 64);
            if (hl == 0) {
                sDel[q] = qok ? delta : 0.f;
                sL[q] = qok ? a.LSE[(size_t)bh * a.Lq + q] * LOG2E : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int e = threadIdx.x + it * NT, ch = e & 7, row = (e >> 3) % QPT, which = (e >> 3) / QPT;
            if (e < NCH) *reinterpret_cast<uint4*>(sQ + (which * QPT + row) * KLD + ch * 8) = rr[it];
        }
    }
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = MODE == RES_DROP32 ? true : MODE == RES_NODROP ? false : a.p > 0.f;
    const bool quad_ok = MODE == RES_DROP32 || (a.Lkv & 3) == 0;
    const bool slow_keep = MODE == RES_GENERAL && drop && !quad_ok;
    const bool rfast = MODE == RES_DROP32 || a.rng_fast;
    const int nt = (a.Lkv + KT - 1) / KT;
    const bool bwave = wave < QB && wave * 32 < a.Lq;  // this wave owns a query block in phase B
    f32x16_t dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    h16x8_t kf[4], vf[4];
    {
        const int kv = wave * 32 + kl;
        const bool kvok = kv < a.Lkv;
        const size_t row = (size_t)b * a.Lkv + (kvok ? kv : 0);
        const h16_t* kp = a.K + row * a.ldk + hd * DH + 8 * hl;
        const h16_t* vp = a.V + row * a.ldv + hd * DH + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(kp + ks * 16, kvok));
            vf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(vp + ks * 16, kvok));
        }
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int kv = t * KT + wave * 32 + kl;
        const bool kvok = kv < a.Lkv;
        const bool tail_t = t * KT + KT > a.Lkv;          // (uniform) only the last tile can hold keys beyond Lkv
        h16_t* myK = sK + (wave * 32 + kl) * KLD + 8 * hl;
        h16_t* myT = sT + (wave * 32 + kl) * DLD + 4 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) *reinterpret_cast<uint4*>(myK + ks * 16) = __builtin_bit_cast(uint4, kf[ks]);
        f32x16_t dk[2], dv[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[0][r] = dk[1][r] = dv[0][r] = dv[1][r] = 0.f; }
        // ---- phase A  (rolled: unrolled over the query blocks the live set passes 256 registers)
#pragma unroll 1
        for (int q0 = 0; q0 < QPT; q0 += 32) {
            if (q0 >= a.Lq) break;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = vpf_mfma32(frag_row(sQ, KLD, q0, ks * 16), kf[ks], s);
                dp = vpf_mfma32(frag_row(sD, KLD, q0, ks * 16), vf[ks], dp);
            }
            float pd[16], ds[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                                        // p, the four queries' statistics in one 16-byte read each
                const float4 l4 = *reinterpret_cast<const float4*>(sL + q0 + 8 * g4 + 4 * hl);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) pd[4 * g4 + e] = vpf_exp2(fmaf(s[4 * g4 + e], c, -lq[e]));   // unconditional: no exec-mask branch per score
            }
            if (tail_t || q0 + 32 > a.Lq) {                                         // keys / queries beyond the bounds: last tile, last block only
#pragma unroll
                for (int r = 0; r < 16; ++r) pd[r] = (kvok && (q0 + (r & 3) + 8 * (r >> 2) + 4 * hl) < a.Lq) ? pd[r] : 0.f;
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 d4 = *reinterpret_cast<const float4*>(sDel + q0 + 8 * g4 + 4 * hl);
                const float dq4[4] = {d4.x, d4.y, d4.z, d4.w};
                if (MODE == RES_NODROP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        ds[r] = pd[r] * (dp[r] - dq4[e]) * a.scale;
                    }
                } else {
                    uint2 grp = make_uint2(0u, 0u);
                    if (drop && quad_ok) {
                        const int qh = q0 + 8 * g4 + 4 * hl + (lane & 3);
                        grp = rfast ? vpf_rand4x16_32(rng, (((uint32_t)bh * a.Lq + (uint32_t)qh) * (uint32_t)a.Lkv + (uint32_t)kv) >> 2) : vpf_rand4x16(rng, (((uint64_t)bh * a.Lq + (uint64_t)qh) * (uint64_t)a.Lkv + (uint64_t)kv) >> 2);
                    }
                    uint32_t gw[4];
                    {
                        const uint32_t mine = (lane & 2) ? 1u : 0u;
                        const uint32_t x0 = quad_bcast<0>(grp.x), x1 = quad_bcast<1>(grp.x), x2 = quad_bcast<2>(grp.x), x3 = quad_bcast<3>(grp.x);
                        const uint32_t y0 = quad_bcast<0>(grp.y), y1 = quad_bcast<1>(grp.y), y2 = quad_bcast<2>(grp.y), y3 = quad_bcast<3>(grp.y);
                        gw[0] = mine ? y0 : x0; gw[1] = mine ? y1 : x1; gw[2] = mine ? y2 : x2; gw[3] = mine ? y3 : x3;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        const int qq = q0 + e + 8 * g4 + 4 * hl;
                        const float pr = pd[r];
                        const uint32_t word = gw[e];                               // this lane's half of quad lane e's group (p = 0: thresh 0, scale 1)
                        float keep = (((lane & 1) ? (word >> 16) : (word & 0xffffu)) >= rng.thresh) ? rng.scale : 0.f;
                        if (MODE == RES_GENERAL && slow_keep) keep = vpf_keep(rng, ((uint64_t)bh * a.Lq + (uint64_t)qq) * (uint64_t)a.Lkv + (uint64_t)kv) ? rng.scale : 0.f;
                        pd[r] = pr * keep;
                        ds[r] = pr * (dp[r] * keep - dq4[e]) * a.scale;
                    }
                }
                uint2 w;
                w.x = pack_h16x2(ds[4 * g4 + 0], ds[4 * g4 + 1]); w.y = pack_h16x2(ds[4 * g4 + 2], ds[4 * g4 + 3]);
                *reinterpret_cast<uint2*>(myT + q0 + 8 * g4) = w;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h16x8_t pf = pack8(pd + 8 * s2), sf = pack8(ds + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = vpf_mfma32(frag_tr_perm(sD, KLD, q0 + 16 * s2, dt * 32), pf, dv[dt]);
                    dk[dt] = vpf_mfma32(frag_tr_perm(sQ, KLD, q0 + 16 * s2, dt * 32), sf, dk[dt]);
                }
            }
        }
        if (t + 1 < nt) {                               // the next tile's keys / values travel behind the stores and phase B
            const int kv2 = kv + KT;
            const bool ok2 = kv2 < a.Lkv;
            const size_t row = (size_t)b * a.Lkv + (ok2 ? kv2 : 0);
            const h16_t* kp = a.K + row * a.ldk + hd * DH + 8 * hl;
            const h16_t* vp = a.V + row * a.ldv + hd * DH + 8 * hl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(kp + ks * 16, ok2));
                vf[ks] = __builtin_bit_cast(h16x8_t, ld16_or_zero(vp + ks * 16, ok2));
            }
        }
        __syncthreads();
        // ---- phase B: dQ^T[dh][q] += K^T[dh][key] . dS^T[key][q] over the tile's keys
        if (bwave) {
#pragma unroll
            for (int k16 = 0; k16 < KT / 16; ++k16) {
                if (t * KT + k16 * 16 >= a.Lkv) break;
                const h16x8_t dsf = frag_tr_perm(sT, DLD, k16 * 16, wave * 32);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = vpf_mfma32(frag_tr_perm(sK, KLD, k16 * 16, dt * 32), dsf, dq[dt]);
            }
        }
        __syncthreads();
        // ---- dK / dV of the tile leave through this wave's OWN 32 rows of the K tile (nobody reads it any more; the next iteration
        // rewrites exactly these rows): in the accumulator layout a store instruction puts 16 bytes into each of 32 rows -- an ablation
        // without these stores ran 75 instead of 118 us (round 4) -- from LDS the same tile goes out as whole 128-byte rows.
        {
            h16_t* S = sK + (wave * 32) * KLD;
            auto tile_out = [&](const f32x16_t (&acc)[2], h16_t* G, long ld) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        uint2 u;
                        u.x = pack_h16x2(acc[dt][4 * gq + 0], acc[dt][4 * gq + 1]); u.y = pack_h16x2(acc[dt][4 * gq + 2], acc[dt][4 * gq + 3]);
                        *reinterpret_cast<uint2*>(S + kl * KLD + dt * 32 + 8 * gq + 4 * hl) = u;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave: LDS operations complete in order
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rl = it * 8 + (lane >> 3), ch = (lane & 7) * 8;
                    const int key = t * KT + wave * 32 + rl;
                    const uint4 v = *reinterpret_cast<const uint4*>(S + rl * KLD + ch);
                    if (key < a.Lkv) *reinterpret_cast<uint4*>(G + ((size_t)b * a.Lkv + key) * ld + hd * DH + ch) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the reads are done before the rows are written again)
            };
            tile_out(dk, a.dK, a.lddk);
            tile_out(dv, a.dV, a.lddv);
        }
    }
    if (bwave) {
        const int q = wave * 32 + kl;
        if (q < a.Lq) {
            h16_t* op = a.dQ + ((size_t)b * a.Lq + q) * a.lddq + hd * DH;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    uint2 u;
                    u.x = pack_h16x2(dq[dt][4 * gq + 0], dq[dt][4 * gq + 1]);
                    u.y = pack_h16x2(dq[dt][4 * gq + 2], dq[dt][4 * gq + 3]);
                    *reinterpret_cast<uint2*>(op + dt * 32 + 8 * gq + 4 * hl) = u;
                }
        }
    }
}
template <int QB>
static int launch_bwd_ca(const AttnArgs& a, hipStream_t st)
{
    constexpr int QPT = QB * 32, KT = 128;
    constexpr size_t lds = sizeof(h16_t) * (2 * QPT * KLD + KT * KLD + KT * (QPT + 8)) + sizeof(float) * 2 * QPT;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        for (const void* f : {(const void*)attn_bwd_ca_kernel<QB, RES_GENERAL>, (const void*)attn_bwd_ca_kernel<QB, RES_DROP32>, (const void*)attn_bwd_ca_kernel<QB, RES_NODROP>})
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    switch (res_mode(a)) {
    case RES_DROP32: hipLaunchKernelGGL((attn_bwd_ca_kernel<QB, RES_DROP32>), dim3(a.B * a.H), dim3(256), lds, st, a); break;
    case RES_NODROP: hipLaunchKernelGGL((attn_bwd_ca_kernel<QB, RES_NODROP>), dim3(a.B * a.H), dim3(256), lds, st, a); break;
    default: hipLaunchKernelGGL((attn_bwd_ca_kernel<QB, RES_GENERAL>), dim3(a.B * a.H), dim3(256), lds, st, a);
    }
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

template <int NWQ, int NWK>
static int launch_bwd(const AttnArgs& a, float* delta, hipStream_t st)
{
    // few queries, many keys, enough (cloud, head) pairs to fill the chip twice: one kernel for all three gradients
    if (NWK == 4 && vpf_debug().attn_resident && vpf_debug().attn_ca_merged && !a.pad && a.Lq <= 128 && a.Lkv >= 512 &&
        (long)a.B * a.H >= vpf_debug().attn_ca_merged)
        return a.Lq <= 96 ? launch_bwd_ca<3>(a, st) : launch_bwd_ca<4>(a, st);
    if (a.Lkv >= 256) {
        constexpr int KT = 128;
        constexpr size_t lds = sizeof(h16_t) * 2 * 2 * KT * KLD;
        static VpfPerDevice attr_dev; bool& attr = attr_dev();
        if (!attr) {
            if (hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<NWQ, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VPF_ERR_HIP;
            attr = true;
        }
        hipLaunchKernelGGL((attn_bwd_dq_kernel<NWQ, KT>), dim3(a.B * a.H, vpf_cdiv(a.Lq, 32 * NWQ)), dim3(NWQ * 64), lds, st, a, delta);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<NWQ, 32>), dim3(a.B * a.H, vpf_cdiv(a.Lq, 32 * NWQ)), dim3(NWQ * 64), sizeof(h16_t) * 2 * 2 * 32 * KLD, st, a, delta);
    }
    const int res = vpf_debug().attn_resident;
    if (res && !a.pad && a.Lq <= 96 && NWK == 4)
        hipLaunchKernelGGL((attn_bwd_dkv_resq_kernel<NWK, 3>), dim3(a.B * a.H, vpf_cdiv(a.Lkv, 32 * NWK)), dim3(NWK * 64), 0, st, a, (const float*)delta);
    else if (res && !a.pad && a.Lq <= 128 && NWK == 4)
        hipLaunchKernelGGL((attn_bwd_dkv_resq_kernel<NWK, 4>), dim3(a.B * a.H, vpf_cdiv(a.Lkv, 32 * NWK)), dim3(NWK * 64), 0, st, a, (const float*)delta);
    else
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<NWK>), dim3(a.B * a.H, vpf_cdiv(a.Lkv, 32 * NWK)), dim3(NWK * 64), 0, st, a, (const float*)delta);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
template <int NWQ>
static int launch_bwd_k(const AttnArgs& a, float* delta, hipStream_t st)
{
    const int nkb = vpf_cdiv(a.Lkv, 32);
    if (nkb <= 1) return launch_bwd<NWQ, 1>(a, delta, st);
    if (nkb <= 2) return launch_bwd<NWQ, 2>(a, delta, st);
    if (nkb == 3 || nkb == 6 || nkb == 9) return launch_bwd<NWQ, 3>(a, delta, st);
    if (nkb == 7) return launch_bwd<NWQ, 7>(a, delta, st);
    return launch_bwd<NWQ, 4>(a, delta, st);
}

static int attention_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* out,
                         long ldo, const void* dout, long lddo, const float* lse, int B, int H, int Lq, int Lkv,
                         int head_dim, float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                         void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, float* delta_ws, const uint8_t* pad, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    if (head_dim != DH) return VPF_ERR_UNSUPPORTED;
    AttnArgs a = {};
    a.pad = pad;
    a.rng_fast = vpf_debug().attn_rng32 && ((unsigned long long)B * (unsigned long long)H * (unsigned long long)Lq * (unsigned long long)Lkv < (1ull << 32)) && (Lkv % 4 == 0);
    a.Q = (const h16_t*)q; a.K = (const h16_t*)k; a.V = (const h16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
    a.O = (h16_t*)out; a.ldo = ldo; a.LSE = (float*)lse; a.B = B; a.H = H; a.Lq = Lq; a.Lkv = Lkv; a.scale = scale;
    a.rng = rng_state; a.site = site; a.p = dropout_p;
    a.dO = (const h16_t*)dout; a.lddo = lddo; a.dQ = (h16_t*)dq; a.dK = (h16_t*)dk; a.dV = (h16_t*)dv;
    a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    int rc = check_common(a);
    if (rc) return rc;
    if (!out || !dout || !lse || !dq || !dk || !dv || !delta_ws) return VPF_ERR_NULL;
    if ((ldo % 8) || (lddo % 8) || (lddq % 4) || (lddk % 4) || (lddv % 4)) return VPF_ERR_BADALIGN;
    if (((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dq & 7) || ((uintptr_t)dk & 7) || ((uintptr_t)dv & 7)) return VPF_ERR_BADALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int nqb = vpf_cdiv(Lq, 32);
    const int res = vpf_debug().attn_resident;
    if (res && Lq == Lkv && k != q && !pad) {
        if (nqb == 3) return launch_res_bwd<3>(a, delta_ws, st);
        if (nqb == 4) return launch_res_bwd<4>(a, delta_ws, st);
        if (nqb == 5) return launch_res_bwd<5>(a, delta_ws, st);
        if (nqb == 7) return launch_res_bwd<7>(a, delta_ws, st);
    }
    if (nqb <= 1) return launch_bwd_k<1>(a, delta_ws, st);
    if (nqb <= 2) return launch_bwd_k<2>(a, delta_ws, st);
    if (nqb == 3 || nqb == 6 || nqb == 9) return launch_bwd_k<3>(a, delta_ws, st);
    if (nqb == 7) return launch_bwd_k<7>(a, delta_ws, st);
    return launch_bwd_k<4>(a, delta_ws, st);
}
extern "C" int vpf_attention_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* out,
                                 long ldo, const void* dout, long lddo, const float* lse, int B, int H, int Lq, int Lkv,
                                 int head_dim, float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                                 void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, float* delta_ws, void* stream)
{
    return attention_bwd(q, ldq, k, ldk, v, ldv, out, ldo, dout, lddo, lse, B, H, Lq, Lkv, head_dim, scale, dropout_p, rng_state, site,
                         dq, lddq, dk, lddk, dv, lddv, delta_ws, nullptr, stream);
}
extern "C" int vpf_attention_bwd_pad(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* out,
                                     long ldo, const void* dout, long lddo, const float* lse, int B, int H, int Lq, int Lkv,
                                     int head_dim, float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                                     void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, float* delta_ws,
                                     const uint8_t* pad_mask, void* stream)
{
    if (!pad_mask) return VPF_ERR_NULL;
    return attention_bwd(q, ldq, k, ldk, v, ldv, out, ldo, dout, lddo, lse, B, H, Lq, Lkv, head_dim, scale, dropout_p, rng_state, site,
                         dq, lddq, dk, lddk, dv, lddv, delta_ws, pad_mask, stream);
}
