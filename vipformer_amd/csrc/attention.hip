// attention.hip -- fused multi-head attention forward / backward for gfx950 (head dim 64).
//
// Replaces MultiHeadAttention.forward's einsum -> scale -> softmax -> dropout -> einsum chain
// (vipformer/model/pointcloud/partseg.py:67-86) and its autograd backward.  Flash-style: the
// [b*h, Lq, Lkv] score matrix never reaches HBM; forward keeps log-sum-exp per query row and
// backward recomputes the probabilities.  Dropout on the probabilities (partseg.py:80) is a
// counter-based function of (rng state, site, (bh, q, kv)) so backward regenerates the mask.
//
// Layout ("swapped" products, key/value index in registers, query on the lane):
//   S^T[kv,q] = K[kv,:] . Q[q,:]      v_mfma_f32_32x32x16_bf16, A = K tile rows (LDS, ds_read_b128),
//                                     B = Q fragment (registers, loaded once)
//   softmax statistics are per-lane scalars (query = lane & 31; partner lane ^ 32 holds the other
//   16 keys of a 32-key sub-tile), P^T accumulators are re-used in place as the B operand of
//   O^T[d,q] += V^T[d,kv] . P^T[kv,q]  with V^T fragments fetched from the natural [kv][d] LDS tile
//   by the gfx950 transposing read ds_read_b64_tr_b16 (no transposed copy of V anywhere).
// One workgroup = one (batch, head); one 64-lane wave = 32 query rows; K/V tiles are shared by
// the workgroup's waves and double-buffered in LDS (global loads of tile t+1 in flight during
// the matrix work on tile t).
#include "vpf_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define DH 64
#define KLD 72          // LDS row stride (bf16) of the K / V / Q / dO tiles: 64 + 8 pad
#define TLD 40          // LDS row stride (bf16) of the per-wave P / dS scratch: 32 + 8 pad
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V;
    long ldq, ldk, ldv;           // row strides (elements); head h starts at column h*64
    bf16_t* O; long ldo;
    float* LSE;                   // [B,H,Lq]
    int B, H, Lq, Lkv;
    float scale;
    const uint32_t* rng; uint32_t site; float p;
    // backward
    const bf16_t* dO; long lddo;
    bf16_t* dQ; bf16_t* dK; bf16_t* dV; long lddq, lddk, lddv;
};

__device__ __forceinline__ s16x4_t lds_tr16(const bf16_t* p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
}
// A-operand fragment (rows = 32 consecutive columns c0.. of a natural [k][col] LDS tile, k-slots in the
// PERMUTED order of an accumulator-as-operand chain): slot (h,j) <-> k = kbase + 8*(j>>2) + 4h + (j&3)
__device__ __forceinline__ bf16x8_t frag_tr_perm(const bf16_t* S, int ld, int kbase, int c0)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, coff = 16 * (g & 1);
    const bf16_t* a = S + (kbase + 4 * h + q) * ld + c0 + coff + 4 * p;
    const s16x4_t lo = lds_tr16(a), hi = lds_tr16(a + 8 * ld);
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}
// natural k order: slot (h,j) <-> k = kbase + 8h + j   (operand element (k, c0 + (lane&31)))
__device__ __forceinline__ bf16x8_t frag_tr_nat(const bf16_t* S, int ld, int kbase, int c0)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, coff = 16 * (g & 1);
    const bf16_t* a = S + (kbase + 8 * h + q) * ld + c0 + coff + 4 * p;
    const s16x4_t lo = lds_tr16(a), hi = lds_tr16(a + 4 * ld);
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}
// row fragment: element (row0 + (lane&31), kbase + 8h + j) of a [row][k] LDS tile
__device__ __forceinline__ bf16x8_t frag_row(const bf16_t* S, int ld, int row0, int kbase)
{
    const int lane = threadIdx.x & 63;
    const uint4 v = *reinterpret_cast<const uint4*>(S + (row0 + (lane & 31)) * ld + kbase + 8 * (lane >> 5));
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t pack8(const float* f)
{
    uint4 u;
    u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]); u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
    return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ uint4 ld16_or_zero(const bf16_t* p, bool ok)
{
    return ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
}

// cooperative stage of a [ROWS x 64] bf16 tile (rows r0.. of a [L, ld] matrix, head column offset applied by caller)
template <int ROWS, int MAXC>
__device__ __forceinline__ void kv_load(const bf16_t* __restrict__ G, long ld, int L, int r0, uint4 (&regs)[MAXC], int nthreads)
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
__device__ __forceinline__ void kv_store(bf16_t* __restrict__ S, const uint4 (&regs)[MAXC], int nthreads)
{
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * nthreads;
        if (c < ROWS * 8) { const int row = c >> 3, ch = c & 7; *reinterpret_cast<uint4*>(S + row * KLD + ch * 8) = regs[i]; }
    }
}

// =============================================================================== forward
#define FWD_KT 64   // kv rows per LDS tile (two 32-row sub-tiles)
#define FWD_MAXC 3  // ceil(64*8 / 192) chunks per thread at the smallest block (3 waves); 1-wave blocks loop 8x below
template <int NW>
__global__ void __launch_bounds__(NW * 64) attn_fwd_kernel(AttnArgs a)
{
    constexpr int NT = NW * 64;
    constexpr int MAXC = (FWD_KT * 8 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 2 * FWD_KT * KLD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H;
    const int q = (blockIdx.y * NW + wave) * 32 + (lane & 31);
    const bool qok = q < a.Lq;
    const bf16_t* Kg = a.K + (size_t)b * a.Lkv * a.ldk + hd * DH;
    const bf16_t* Vg = a.V + (size_t)b * a.Lkv * a.ldv + hd * DH;

    bf16x8_t qf[4];
    {
        const bf16_t* qp = a.Q + ((size_t)b * a.Lq + (qok ? q : 0)) * a.ldq + hd * DH + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(bf16x8_t, ld16_or_zero(qp + ks * 16, qok));
    }
    f32x16_t o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = a.p > 0.f;
    const uint64_t rbase = ((uint64_t)bh * a.Lq + (uint64_t)(qok ? q : 0)) * (uint64_t)a.Lkv;

    uint4 rk[MAXC], rv[MAXC];
    const int nt = (a.Lkv + FWD_KT - 1) / FWD_KT;
    kv_load<FWD_KT, MAXC>(Kg, a.ldk, a.Lkv, 0, rk, NT);
    kv_load<FWD_KT, MAXC>(Vg, a.ldv, a.Lkv, 0, rv, NT);
    kv_store<FWD_KT, MAXC>(lds, rk, NT);
    kv_store<FWD_KT, MAXC>(lds + FWD_KT * KLD, rv, NT);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const bf16_t* sK = lds + (t & 1) * 2 * FWD_KT * KLD;
        const bf16_t* sV = sK + FWD_KT * KLD;
        if (t + 1 < nt) {
            kv_load<FWD_KT, MAXC>(Kg, a.ldk, a.Lkv, (t + 1) * FWD_KT, rk, NT);
            kv_load<FWD_KT, MAXC>(Vg, a.ldv, a.Lkv, (t + 1) * FWD_KT, rv, NT);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kv0 = t * FWD_KT + sub * 32;
            if (kv0 >= a.Lkv) break;
            f32x16_t s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(sK, KLD, sub * 32, ks * 16), qf[ks], s, 0, 0, 0);
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                s[r] = kv < a.Lkv ? s[r] * c : -INFINITY;
                tmax = fmaxf(tmax, s[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mn = fmaxf(m, tmax);           // finite: every sub-tile has >= 1 valid key
            const float alpha = exp2f(m - mn);
            m = mn;
            float ps = 0.f;
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = exp2f(s[r] - mn);
                ps += pr;
                float pd = pr;
                if (drop) {
                    const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                    pd = vpf_keep(rng, rbase + (uint64_t)kv) ? pr * rng.scale : 0.f;
                }
                pv[r] = pd;
            }
            l = l * alpha + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t pf = pack8(pv + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_perm(sV, KLD, sub * 32 + 16 * s2, dt * 32), pf, o[dt], 0, 0, 0);
            }
        }
        if (t + 1 < nt) {
            bf16_t* nK = lds + ((t + 1) & 1) * 2 * FWD_KT * KLD;
            kv_store<FWD_KT, MAXC>(nK, rk, NT);
            kv_store<FWD_KT, MAXC>(nK + FWD_KT * KLD, rv, NT);
        }
        __syncthreads();
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = 1.f / lt;
    if (qok) {
        bf16_t* op = a.O + ((size_t)b * a.Lq + q) * a.ldo + hd * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u;
                u.x = pack_bf16x2(o[dt][4 * gq + 0] * inv, o[dt][4 * gq + 1] * inv);
                u.y = pack_bf16x2(o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv);
                *reinterpret_cast<uint2*>(op + dt * 32 + 8 * gq + 4 * hl) = u;
            }
        if (hl == 0 && a.LSE) a.LSE[(size_t)bh * a.Lq + q] = (m + log2f(lt)) * LN2;
    }
}

template <int NW>
static int launch_fwd(const AttnArgs& a, hipStream_t st)
{
    dim3 grid(a.B * a.H, vpf_cdiv(a.Lq, 32 * NW));
    hipLaunchKernelGGL((attn_fwd_kernel<NW>), grid, dim3(NW * 64), 0, st, a);
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

extern "C" int vpf_attention_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, int B, int H,
                                 int Lq, int Lkv, int head_dim, float scale, float dropout_p, const uint32_t* rng_state,
                                 uint32_t site, void* out, long ldo, float* lse, void* stream)
{
    if (head_dim != DH) return VPF_ERR_UNSUPPORTED;
    AttnArgs a = {};
    a.Q = (const bf16_t*)q; a.K = (const bf16_t*)k; a.V = (const bf16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
    a.O = (bf16_t*)out; a.ldo = ldo; a.LSE = lse; a.B = B; a.H = H; a.Lq = Lq; a.Lkv = Lkv; a.scale = scale;
    a.rng = rng_state; a.site = site; a.p = dropout_p;
    int rc = check_common(a);
    if (rc) return rc;
    if (!out) return VPF_ERR_NULL;
    if ((ldo % 4) || ((uintptr_t)out & 7)) return VPF_ERR_BADALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int nqb = vpf_cdiv(Lq, 32);
    if (nqb <= 1) return launch_fwd<1>(a, st);
    if (nqb <= 2) return launch_fwd<2>(a, st);
    if (nqb <= 3) return launch_fwd<3>(a, st);
    if (nqb <= 4) return launch_fwd<4>(a, st);
    if (nqb == 5) return launch_fwd<5>(a, st);
    if (nqb == 6) return launch_fwd<6>(a, st);
    if (nqb == 7) return launch_fwd<7>(a, st);
    return launch_fwd<8>(a, st);
}

// =============================================================================== backward
// One workgroup = one (batch, head) with ALL its query rows (one wave per 32 rows), looping over
// 32-row K/V tiles.  Per tile and wave: S^T and dP^T as in forward; dS^T accumulators feed
// dQ^T += K^T . dS^T directly; P and dS cross LDS once ([q][kv] scratch) so that
// dV[kv,d] = sum_q P[q,kv] dO[q,d] and dK[kv,d] = sum_q dS[q,kv] Q[q,d] contract over the query index with
// transposing reads.  The waves' dK/dV partial tiles meet in an fp32 LDS accumulator (ds_add_f32), which
// is flushed to HBM as bf16 once per tile -- no global atomics, no cross-workgroup reduction.
#define BWD_KT 32
template <int NW>
__global__ void __launch_bounds__(NW * 64) attn_bwd_kernel(AttnArgs a)
{
    constexpr int NT = NW * 64;
    constexpr int MAXC = (BWD_KT * 8 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sKV = reinterpret_cast<bf16_t*>(smem_raw);                          // [2 buf][K,V][32][KLD]
    float* sAcc = reinterpret_cast<float*>(sKV + 2 * 2 * BWD_KT * KLD);          // [2 buf][dK,dV][32][64]
    bf16_t* sW = reinterpret_cast<bf16_t*>(sAcc + 2 * 2 * BWD_KT * DH);          // per wave: Q[32][KLD], dO[32][KLD], P[32][TLD], dS[32][TLD]
    constexpr int WSZ = 2 * 32 * KLD + 2 * 32 * TLD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, ql = lane & 31;
    bf16_t* sQ = sW + wave * WSZ;
    bf16_t* sdO = sQ + 32 * KLD;
    bf16_t* sP = sdO + 32 * KLD;
    bf16_t* sdS = sP + 32 * TLD;

    const int bh = blockIdx.x, b = bh / a.H, hd = bh % a.H;
    const int q = wave * 32 + ql;
    const bool qok = q < a.Lq;
    const bf16_t* Kg = a.K + (size_t)b * a.Lkv * a.ldk + hd * DH;
    const bf16_t* Vg = a.V + (size_t)b * a.Lkv * a.ldv + hd * DH;

    // ---- per-wave prologue: Q / dO fragments (registers + natural LDS image), delta = rowsum(dO * O)
    bf16x8_t qf[4], dof[4];
    float delta = 0.f;
    {
        const size_t row = (size_t)b * a.Lq + (qok ? q : 0);
        const bf16_t* qp = a.Q + row * a.ldq + hd * DH + 8 * hl;
        const bf16_t* dp = a.dO + row * a.lddo + hd * DH + 8 * hl;
        const bf16_t* op = a.O + row * a.ldo + hd * DH + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 uq = ld16_or_zero(qp + ks * 16, qok), ud = ld16_or_zero(dp + ks * 16, qok), uo = ld16_or_zero(op + ks * 16, qok);
            qf[ks] = __builtin_bit_cast(bf16x8_t, uq); dof[ks] = __builtin_bit_cast(bf16x8_t, ud);
            *reinterpret_cast<uint4*>(sQ + ql * KLD + ks * 16 + 8 * hl) = uq;
            *reinterpret_cast<uint4*>(sdO + ql * KLD + ks * 16 + 8 * hl) = ud;
            const uint32_t dw[4] = {ud.x, ud.y, ud.z, ud.w}, ow[4] = {uo.x, uo.y, uo.z, uo.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                delta += __uint_as_float(dw[j] << 16) * __uint_as_float(ow[j] << 16);
                delta += __uint_as_float(dw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
            }
        }
    }
    delta += __shfl_xor(delta, 32, 64);
    const float lse2 = qok ? a.LSE[(size_t)bh * a.Lq + q] * LOG2E : 0.f;
    const float c = a.scale * LOG2E;
    const VpfRng rng = vpf_rng_init(a.rng, a.site, a.p);
    const bool drop = a.p > 0.f;
    const uint64_t rbase = ((uint64_t)bh * a.Lq + (uint64_t)(qok ? q : 0)) * (uint64_t)a.Lkv;

    f32x16_t dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }

    for (int i = threadIdx.x; i < 2 * 2 * BWD_KT * DH; i += NT) sAcc[i] = 0.f;
    uint4 rk[MAXC], rv[MAXC];
    const int nt = (a.Lkv + BWD_KT - 1) / BWD_KT;
    kv_load<BWD_KT, MAXC>(Kg, a.ldk, a.Lkv, 0, rk, NT);
    kv_load<BWD_KT, MAXC>(Vg, a.ldv, a.Lkv, 0, rv, NT);
    kv_store<BWD_KT, MAXC>(sKV, rk, NT);
    kv_store<BWD_KT, MAXC>(sKV + BWD_KT * KLD, rv, NT);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const bf16_t* sK = sKV + (t & 1) * 2 * BWD_KT * KLD;
        const bf16_t* sV = sK + BWD_KT * KLD;
        float* accK = sAcc + (t & 1) * 2 * BWD_KT * DH;
        float* accV = accK + BWD_KT * DH;
        const int kv0 = t * BWD_KT;
        if (t + 1 < nt) {
            kv_load<BWD_KT, MAXC>(Kg, a.ldk, a.Lkv, (t + 1) * BWD_KT, rk, NT);
            kv_load<BWD_KT, MAXC>(Vg, a.ldv, a.Lkv, (t + 1) * BWD_KT, rv, NT);
        }
        // S^T and dP^T
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(sK, KLD, 0, ks * 16), qf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(sV, KLD, 0, ks * 16), dof[ks], dp, 0, 0, 0);
        }
        float pd[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
            const bool ok = qok && kv < a.Lkv;
            const float pr = ok ? exp2f(s[r] * c - lse2) : 0.f;
            float keep = 1.f;
            if (drop) keep = vpf_keep(rng, rbase + (uint64_t)kv) ? rng.scale : 0.f;
            pd[r] = pr * keep;
            ds[r] = pr * (dp[r] * keep - delta) * a.scale;
        }
        // dQ^T += K^T . dS^T   (dS accumulators as the B operand, K^T by transposing reads)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t dsf = pack8(ds + 8 * s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_perm(sK, KLD, 16 * s2, dt * 32), dsf, dq[dt], 0, 0, 0);
        }
        // P, dS -> per-wave LDS scratch [q][kv]
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            uint2 u, w;
            u.x = pack_bf16x2(pd[4 * gq + 0], pd[4 * gq + 1]); u.y = pack_bf16x2(pd[4 * gq + 2], pd[4 * gq + 3]);
            w.x = pack_bf16x2(ds[4 * gq + 0], ds[4 * gq + 1]); w.y = pack_bf16x2(ds[4 * gq + 2], ds[4 * gq + 3]);
            *reinterpret_cast<uint2*>(sP + ql * TLD + 8 * gq + 4 * hl) = u;
            *reinterpret_cast<uint2*>(sdS + ql * TLD + 8 * gq + 4 * hl) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // dV[kv,d] += P^T dO ; dK[kv,d] += dS^T Q   (contract over this wave's 32 query rows)
        f32x16_t dv[2], dk[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dv[0][r] = dv[1][r] = dk[0][r] = dk[1][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t pf = frag_tr_nat(sP, TLD, 16 * ks, 0);
            const bf16x8_t sf = frag_tr_nat(sdS, TLD, 16 * ks, 0);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_tr_nat(sdO, KLD, 16 * ks, dt * 32), dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sf, frag_tr_nat(sQ, KLD, 16 * ks, dt * 32), dk[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kvl = (r & 3) + 8 * (r >> 2) + 4 * hl;
                atomicAdd(accV + kvl * DH + dt * 32 + ql, dv[dt][r]);
                atomicAdd(accK + kvl * DH + dt * 32 + ql, dk[dt][r]);
            }
        if (t + 1 < nt) {
            bf16_t* nK = sKV + ((t + 1) & 1) * 2 * BWD_KT * KLD;
            kv_store<BWD_KT, MAXC>(nK, rk, NT);
            kv_store<BWD_KT, MAXC>(nK + BWD_KT * KLD, rv, NT);
        }
        __syncthreads();
        // flush this tile's dK / dV (bf16) and clear the accumulator for tile t+2
        for (int e = threadIdx.x; e < 2 * BWD_KT * DH / 4; e += NT) {
            const int which = e / (BWD_KT * DH / 4), r4 = e % (BWD_KT * DH / 4);
            const int kvl = r4 / (DH / 4), d4 = (r4 % (DH / 4)) * 4;
            float* src = accK + which * BWD_KT * DH + kvl * DH + d4;
            const float4 v = *reinterpret_cast<float4*>(src);
            *reinterpret_cast<float4*>(src) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kv0 + kvl < a.Lkv) {
                uint2 u; u.x = pack_bf16x2(v.x, v.y); u.y = pack_bf16x2(v.z, v.w);
                bf16_t* dst = which ? a.dV + ((size_t)b * a.Lkv + kv0 + kvl) * a.lddv + hd * DH + d4
                                    : a.dK + ((size_t)b * a.Lkv + kv0 + kvl) * a.lddk + hd * DH + d4;
                *reinterpret_cast<uint2*>(dst) = u;
            }
        }
    }
    if (qok) {
        bf16_t* op = a.dQ + ((size_t)b * a.Lq + q) * a.lddq + hd * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u;
                u.x = pack_bf16x2(dq[dt][4 * gq + 0], dq[dt][4 * gq + 1]);
                u.y = pack_bf16x2(dq[dt][4 * gq + 2], dq[dt][4 * gq + 3]);
                *reinterpret_cast<uint2*>(op + dt * 32 + 8 * gq + 4 * hl) = u;
            }
    }
}

template <int NW>
static int launch_bwd(const AttnArgs& a, hipStream_t st)
{
    const size_t lds = sizeof(bf16_t) * (2 * 2 * BWD_KT * KLD) + sizeof(float) * (2 * 2 * BWD_KT * DH) +
                       sizeof(bf16_t) * (size_t)NW * (2 * 32 * KLD + 2 * 32 * TLD);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_bwd_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VPF_ERR_HIP;
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_kernel<NW>), dim3(a.B * a.H), dim3(NW * 64), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

extern "C" int vpf_attention_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* out,
                                 long ldo, const void* dout, long lddo, const float* lse, int B, int H, int Lq, int Lkv,
                                 int head_dim, float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                                 void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, void* stream)
{
    if (head_dim != DH) return VPF_ERR_UNSUPPORTED;
    AttnArgs a = {};
    a.Q = (const bf16_t*)q; a.K = (const bf16_t*)k; a.V = (const bf16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
    a.O = (bf16_t*)out; a.ldo = ldo; a.LSE = (float*)lse; a.B = B; a.H = H; a.Lq = Lq; a.Lkv = Lkv; a.scale = scale;
    a.rng = rng_state; a.site = site; a.p = dropout_p;
    a.dO = (const bf16_t*)dout; a.lddo = lddo; a.dQ = (bf16_t*)dq; a.dK = (bf16_t*)dk; a.dV = (bf16_t*)dv;
    a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    int rc = check_common(a);
    if (rc) return rc;
    if (!out || !dout || !lse || !dq || !dk || !dv) return VPF_ERR_NULL;
    if ((ldo % 8) || (lddo % 8) || (lddq % 4) || (lddk % 4) || (lddv % 4)) return VPF_ERR_BADALIGN;
    if (((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dq & 7) || ((uintptr_t)dk & 7) || ((uintptr_t)dv & 7)) return VPF_ERR_BADALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int nqb = vpf_cdiv(Lq, 32);
    switch (nqb) {
        case 1: return launch_bwd<1>(a, st);
        case 2: return launch_bwd<2>(a, st);
        case 3: return launch_bwd<3>(a, st);
        case 4: return launch_bwd<4>(a, st);
        case 5: return launch_bwd<5>(a, st);
        case 6: return launch_bwd<6>(a, st);
        case 7: return launch_bwd<7>(a, st);
        default: return VPF_ERR_UNSUPPORTED;   // Lq > 224: not on the pre-training path (G <= 128, T = 196)
    }
}
