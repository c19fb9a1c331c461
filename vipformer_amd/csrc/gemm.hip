// gemm.hip -- h16 MFMA GEMM family for gfx950:  C[m,n] (+)= sum_k A(m,k) * B(n,k)
//
// Replaces the aten linear / conv1d(k=1) / mm / bmm calls of the reference's encoder, MLP,
// projections and Group2Emb (vipformer/model/pointcloud/partseg.py:48-51,67-86,191-198;
// utils.py:153-165) and their autograd backward (dgrad / wgrad).
//
// One kernel template covers forward (A=[M,K], B=W[N,K]), dgrad (B = W read "k-strided") and
// wgrad (both operands k-strided, split-K with fp32 atomics into the gradient buffer):
//   * operands are h16 in HBM; an operand is either K-MAJOR (contraction index contiguous)
//     or K-STRIDED (stored [K][rows]).  Tiles are staged in LDS in their natural HBM
//     orientation with 16-byte coalesced loads; K-major fragments are read with ds_read_b128,
//     K-strided fragments with the gfx950 transposing read ds_read_b64_tr_b16, so no operand
//     is ever transposed in memory.
//   * 64-lane wavefronts, v_mfma_f32_32x32x16_f16, fp32 accumulate; 256 threads = 4 waves per
//     workgroup, wave tile = TM x TN MFMA tiles; LDS double-buffered with register prefetch
//     (global loads of tile t+1 are in flight while tile t is on the matrix cores).
//   * fused epilogues: bias, GELU (+ pre-activation for backward), dropout + residual add,
//     GELU' multiply (dgrad), ReLU, per-row-group bias, fp32 atomic accumulate.
#include "vpf_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LDS_PAD 8   // h16 elements (16 B) of row padding of a K-major tile: 16-byte fragment reads of 32 rows then touch every bank once
// K-strided tiles ([BK][ROWS], read with ds_read_b64_tr_b16) carry NO padding: the transposing read addresses 4 consecutive k rows x
// (2 x 16 rows) per 32-lane half, and a row pitch of ROWS * 2 bytes = a multiple of 256 B puts those 4 k rows on the same banks; a
// pitch of ROWS + 8 elements (rounds 1 - 2) spread them by 4 banks where each needs 16: every transposing read took 8 LDS cycles
// instead of 2 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.60 on the grouped weight gradient, profiles/r03_step_issue.json; the
// layout search is tools/lds_banks.py).  Instead the 32-row column blocks of k row k are XOR-permuted by k's low bits (tr_swz):
// 16-byte staging stores stay contiguous, the four k rows of a read land on four different 64-byte bank groups, nothing is padded.

enum {
    EPI_STORE = 0,       // C = acc (+bias)            -> h16 or f32
    EPI_GELU = 1,        // u = acc + bias; C2 = u (h16, pre-activation), C = gelu(u) (h16)
    EPI_DROP_RES = 2,    // C(f32) = res(f32) + dropout(acc + bias)
    EPI_GELU_BWD = 3,    // C(h16) = acc * gelu'(aux_u(h16))
    EPI_ATOMIC = 4,      // C(f32) += acc   (split-K)
    EPI_RELU = 5,        // C = relu(acc + bias)
    EPI_GROUPBIAS = 6,   // C = acc + gbias[(m / group) * N + n]   (f32 row-group bias), h16/f32 out
    EPI_GROUPMAX = 7,    // C(f32)[m / group, n] = max over the group's rows of h16(acc + bias); C2(u8) = first arg-max
    EPI_PARTIAL = 8,     // split-K without atomics: partial tiles to a workspace, the last-arriving slice of a tile adds them up (grouped wgrad)
};

// Operand prologue: what a 16-byte chunk (8 values along the operand's contiguous dimension = channel axis) becomes
// while it is staged into LDS.  kind 1: y = relu(a[c] * x + b[c]) (BatchNorm + ReLU folded into an affine, the
// normalised activation is never materialised).  kind 2: the operand is VIRTUAL -- the gradient of a max over `group`
// consecutive rows: value(token, c) = (arg[(token/group)*ncols + c] == token % group) ? dout[(token/group)*ncols + c] : 0.
struct OpXform {
    int kind;
    const float* a; const float* b;          // kind 1
    const float* dout; const uint8_t* arg;   // kind 2
    int group; long ncols;
};

struct GemmArgs {
    const h16_t* A; const h16_t* B;
    long lda, ldb;              // leading dimension (elements) of the stored matrices
    long sAb, sBb, sCb;         // batch strides (elements); 0 = shared
    int M, N, K;
    int splitk;                 // >1: blockIdx.z = k-slice (EPI_ATOMIC), else blockIdx.z = batch
    int mode;
    void* C; long ldc; int c_f32;
    void* C2; long ldc2;        // EPI_GELU: pre-activation (h16)
    const float* bias;          // [N] or null
    const float* res; long ldres;      // EPI_DROP_RES
    const h16_t* aux; long ldaux;     // EPI_GELU_BWD: u
    const float* gbias; int group;     // EPI_GROUPBIAS
    const uint32_t* rng; uint32_t site; float p;   // dropout
    float* dbias;                                   // EPI_ATOMIC with a k-strided A: dbias[m] += sum_k A(m,k)
    int uneven;                                     // split-K slices of alternating length (4/3, 2/3 of the mean): see vpf_wgrad_group
    int dbg;                                        // timing experiments only (VPF_WGROUP_DBG): 1 = no flush; in -DVPF_GEMM_DBG_LOOP builds 2 = no MFMA, 4 = no LDS fragment reads
    float* part; int* cnt; int ntx;                 // EPI_PARTIAL: partial tiles [tile][slice][BM*BN] (accumulator order), arrival counters [tile], tiles per row
    OpXform xa, xb;                                 // operand prologues (kind 0 = none)
};

__device__ __forceinline__ float gelu_f(float x) { return vpf_gelu(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return vpf_gelu_grad(x); }

// thread index inside the 256-thread GROUP that works on one K slice: the whole workgroup everywhere except the pair-merged
// grouped weight gradient (gemm_tile<.., PAIR = true>: 512 threads = two groups on two K slices of the same output tile)
__device__ __forceinline__ int gtid() { return threadIdx.x & 255; }

// ------------------------------------------------------------------ tile staging
// K-major operand tile: LDS [ROWS][BK + pad]; K-strided operand tile: LDS [BK][ROWS + pad].
template <int ROWS, bool TR, int BK>
struct TileCfg {
    static_assert(!TR || ROWS == 32 || ROWS == 64 || ROWS % 128 == 0, "tr_swz permutes whole 32-row blocks inside 64 or 128 rows");
    static constexpr int LD = TR ? ROWS : (BK + LDS_PAD);
    static constexpr int ELEMS = TR ? BK * LD : ROWS * LD;
    // column of element (k, col) inside k row k of a K-strided tile
    static __device__ __forceinline__ int tr_swz(int k, int col)
    {
        if constexpr (ROWS >= 128) return col ^ ((k & 3) << 5);
        else if constexpr (ROWS == 64) return col ^ ((k & 2) << 4);
        else return col;
    }
    static constexpr int CHUNKS = ROWS * BK / 8;          // 16-byte chunks per tile
    static constexpr int PER_THREAD = (CHUNKS + 255) / 256;
};

template <int XK>
__device__ __forceinline__ uint4 xform_chunk(const OpXform& xf, uint4 v, int token, int ch)
{
    if constexpr (XK == 1) {
        const float4 a0 = *reinterpret_cast<const float4*>(xf.a + ch), a1 = *reinterpret_cast<const float4*>(xf.a + ch + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(xf.b + ch), b1 = *reinterpret_cast<const float4*>(xf.b + ch + 4);
        const float aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = fmaxf(fmaf(aa[2 * j], h16_lo(w[j]), bb[2 * j]), 0.f);
            const float hi = fmaxf(fmaf(aa[2 * j + 1], h16_hi(w[j]), bb[2 * j + 1]), 0.f);
            w[j] = pack_h16x2(lo, hi);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    // kind 2 (group is a power of two: it divides 32)
    const int sh = 31 - __clz(xf.group);
    const int g = token >> sh, k = token & (xf.group - 1);
    const size_t o = (size_t)g * xf.ncols + ch;
    const uint2 ar = *reinterpret_cast<const uint2*>(xf.arg + o);
    const float4 d0 = *reinterpret_cast<const float4*>(xf.dout + o), d1 = *reinterpret_cast<const float4*>(xf.dout + o + 4);
    const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const uint32_t aw[2] = {ar.x, ar.y};
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e0 = (aw[(2 * j) >> 2] >> (8 * ((2 * j) & 3))) & 0xff, e1 = (aw[(2 * j + 1) >> 2] >> (8 * ((2 * j + 1) & 3))) & 0xff;
        w[j] = pack_h16x2(e0 == k ? dd[2 * j] : 0.f, e1 == k ? dd[2 * j + 1] : 0.f);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int ROWS, bool TR, int BK, int XK>
__device__ __forceinline__ void tile_load(const h16_t* __restrict__ G, long ld, int R, int K, int r0, int k0, int kend,
                                          uint4 (&regs)[TileCfg<ROWS, TR, BK>::PER_THREAD], const OpXform& xf)
{
    using Cfg = TileCfg<ROWS, TR, BK>;
#pragma unroll
    for (int i = 0; i < Cfg::PER_THREAD; ++i) {
        const int c = gtid() + i * 256;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < Cfg::CHUNKS) {
            if (!TR) {
                const int row = c / (BK / 8), kc = c % (BK / 8);
                const int gr = r0 + row, gk = k0 + kc * 8;
                if (gr < R && gk < kend) {
                    if constexpr (XK != 2) v = *reinterpret_cast<const uint4*>(G + (size_t)gr * ld + gk);
                    if constexpr (XK != 0) v = xform_chunk<XK>(xf, v, gr, gk);
                }
            } else {
                const int krow = c / (ROWS / 8), rc = c % (ROWS / 8);
                const int gk = k0 + krow, gr = r0 + rc * 8;
                if (gk < kend && gr < R) {
                    if constexpr (XK != 2) v = *reinterpret_cast<const uint4*>(G + (size_t)gk * ld + gr);
                    if constexpr (XK != 0) v = xform_chunk<XK>(xf, v, gk, gr);
                }
            }
        }
        regs[i] = v;
    }
}

template <int ROWS, bool TR, int BK>
__device__ __forceinline__ void tile_store(h16_t* __restrict__ S, const uint4 (&regs)[TileCfg<ROWS, TR, BK>::PER_THREAD])
{
    using Cfg = TileCfg<ROWS, TR, BK>;
#pragma unroll
    for (int i = 0; i < Cfg::PER_THREAD; ++i) {
        const int c = gtid() + i * 256;
        if (c < Cfg::CHUNKS) {
            int off;
            if (!TR) { const int row = c / (BK / 8), kc = c % (BK / 8); off = row * Cfg::LD + kc * 8; }
            else { const int krow = c / (ROWS / 8), rc = c % (ROWS / 8); off = krow * Cfg::LD + Cfg::tr_swz(krow, rc * 8); }
            *reinterpret_cast<uint4*>(S + off) = regs[i];
        }
    }
}

#ifdef VPF_GEMM_DBG_LOOP
#define GEMM_DBG_S(s) ((g.dbg & 4) ? 0 : (s))
#else
#define GEMM_DBG_S(s) (s)
#endif
// fragment for one 32x32x16 MFMA: rows [row0, row0+32), k-step s (16 wide) of the BK-deep tile
template <int ROWS, bool TR, int BK>
__device__ __forceinline__ h16x8_t frag_read(const h16_t* __restrict__ S, int row0, int s)
{
    using Cfg = TileCfg<ROWS, TR, BK>;
    const int lane = threadIdx.x & 63;
    if (!TR) {
        const int r = lane & 31, h = lane >> 5;
        const uint4 v = *reinterpret_cast<const uint4*>(S + (row0 + r) * Cfg::LD + s * 16 + 8 * h);
        return __builtin_bit_cast(h16x8_t, v);
    } else {
        // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(row) block; lane 4q+p addresses row q,
        // columns 4p..4p+3; lane i receives column i of the 4 rows.
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int h = g >> 1, roff = 16 * (g & 1);
        const h16_t* a0 = S + (s * 16 + 8 * h + q) * Cfg::LD + Cfg::tr_swz(q, row0 + roff + 4 * p);      // (k & 3 = q; k + 4 swizzles alike)
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * Cfg::LD));
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        s16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(h16x8_t, v);
    }
}

// ------------------------------------------------------------------ kernel
template <int TM, int TN, int WM, int WN, int BK, bool ATR, bool BTR, int AX, int BX, int PF = 1, bool PAIR = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, const int bz_)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    using ACfg = TileCfg<BM, ATR, BK>;
    using BCfg = TileCfg<BN, BTR, BK>;
    extern __shared__ __attribute__((aligned(16))) h16_t lds_all[];  // 2 * (ACfg::ELEMS + BCfg::ELEMS) per 256-thread group
    constexpr int STAGE = ACfg::ELEMS + BCfg::ELEMS;
    // PAIR (EPI_ATOMIC split-K only): the workgroup is TWO 256-thread groups that accumulate K slices 2 bz_ and 2 bz_ + 1 of the SAME
    // output tile side by side (own staging buffers, common barriers); at the end they exchange half a tile through LDS and each
    // flushes ONE half with atomics: half the flushed bytes per slice at the same number of resident waves per CU.
    const int half = PAIR ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    h16_t* lds = lds_all + half * 2 * STAGE;
    const int bz = PAIR ? 2 * bz_ + half : bz_;

    const int wave = gtid() >> 6, lane = threadIdx.x & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = by * BM, n0 = bx * BN;

    int kbeg = 0, kend = g.K;
    const h16_t* A = g.A; const h16_t* B = g.B;
    long cb = 0;
    if (g.splitk > 1) {
        const int per = ((g.K + g.splitk - 1) / g.splitk + BK - 1) / BK * BK;
        kbeg = bz * per; kend = min(g.K, kbeg + per);
        if (g.uneven && (g.splitk & 1) == 0) {
            const int base = (bz >> 1) * 2 * per, cut = base + (((6 + g.uneven) * per) / 6 + BK - 1) / BK * BK;      // uneven = 2: 4/3 and 2/3
            if (bz & 1) { kbeg = cut; kend = min(g.K, base + 2 * per); } else { kbeg = base; kend = min(g.K, cut); }
        }
        if (kbeg >= kend) {
            if (g.mode != EPI_PARTIAL && !PAIR) return;
            kend = kbeg;                            // an empty slice still takes part in the arrival count / the pair's barriers (with a zero tile)
        }
    } else {
        A += (size_t)bz * g.sAb; B += (size_t)bz * g.sBb; cb = (long)bz * g.sCb;
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // PF register sets = PF K-stages of global loads in flight per thread (Little's law: at ~2 us of load latency one stage in
    // flight per workgroup cannot keep the memory system busy; PF is chosen so that the VGPR count keeps the occupancy)
    uint4 ra[PF][ACfg::PER_THREAD], rb[PF][BCfg::PER_THREAD];
    const int nk = (kend - kbeg + BK - 1) / BK;
    int nk_loop = nk;                               // PAIR: both groups run the longer slice's number of stages (common barriers)
    if (PAIR) {
        const int per = ((g.K + g.splitk - 1) / g.splitk + BK - 1) / BK * BK;
        const int base = bz_ * 2 * per;
        const int cut = (g.uneven && (g.splitk & 1) == 0) ? base + (((6 + g.uneven) * per) / 6 + BK - 1) / BK * BK : base + per;
        const int n0k = (max(0, min(g.K, cut) - base) + BK - 1) / BK, n1k = (max(0, min(g.K, base + 2 * per) - cut) + BK - 1) / BK;
        nk_loop = max(n0k, n1k);
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < nk) {
            tile_load<BM, ATR, BK, AX>(A, g.lda, g.M, g.K, m0, kbeg + p * BK, kend, ra[p], g.xa);
            tile_load<BN, BTR, BK, BX>(B, g.ldb, g.N, g.K, n0, kbeg + p * BK, kend, rb[p], g.xb);
        }
    tile_store<BM, ATR, BK>(lds, ra[0]);
    tile_store<BN, BTR, BK>(lds + ACfg::ELEMS, rb[0]);
    __syncthreads();

    // fused bias gradient (wgrad): the waves that own the first column of wave tiles in the first column of workgroups also sum their
    // dY fragments over the tokens.  A lane's fragment holds 8 tokens of ONE output row: four v_dot2c_f32_f16 against (1, 1) add them
    // to a per-lane partial; lanes r and r + 32 (the two token halves of row r) meet in the epilogue.  The scalar LDS column sums this
    // replaces (32 two-byte LDS reads per thread and stage) cost 11 us of the 59 us grouped launch.
    const bool do_bias = ATR && g.dbias != nullptr && bx == 0 && wn == 0;
    float bsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bsum[i] = 0.f;
    for (int kt0 = 0; kt0 < nk_loop; kt0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kt = kt0 + p;
            if (kt >= nk_loop) break;
            if (PAIR && kt >= nk) { __syncthreads(); continue; }     // this group's slice is done: keep the other group's barriers company
            const int cur = kt & 1;
            const h16_t* cA = lds + cur * STAGE;
            const h16_t* cB = cA + ACfg::ELEMS;
            h16_t* nA = lds + (cur ^ 1) * STAGE;
            h16_t* nB = nA + ACfg::ELEMS;
            // register set p held stage kt, which is in LDS by now: refill it with stage kt + PF
            if (kt + PF < nk) {
                tile_load<BM, ATR, BK, AX>(A, g.lda, g.M, g.K, m0, kbeg + (kt + PF) * BK, kend, ra[p], g.xa);
                tile_load<BN, BTR, BK, BX>(B, g.ldb, g.N, g.K, n0, kbeg + (kt + PF) * BK, kend, rb[p], g.xb);
            }
#ifdef VPF_GEMM_DBG_LOOP
            if (!(g.dbg & 2))       // (experiment builds only: run-time tests in this loop turn the fragment offsets into VALU arithmetic)
#endif
            {
#pragma unroll
            for (int s = 0; s < BK / 16; ++s) {
                h16x8_t fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = frag_read<BM, ATR, BK>(cA, (wm * TM + i) * 32, GEMM_DBG_S(s));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = frag_read<BN, BTR, BK>(cB, (wn * TN + j) * 32, GEMM_DBG_S(s));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = vpf_mfma32(fa[i], fb[j], acc[i][j]);
                if (ATR && do_bias) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const uint4 w = __builtin_bit_cast(uint4, fa[i]);
                        bsum[i] = h16_dot2(w.x, VPF_H16_ONES2, bsum[i]);
                        bsum[i] = h16_dot2(w.y, VPF_H16_ONES2, bsum[i]);
                        bsum[i] = h16_dot2(w.z, VPF_H16_ONES2, bsum[i]);
                        bsum[i] = h16_dot2(w.w, VPF_H16_ONES2, bsum[i]);
                    }
                }
            }
            }
            if (kt + 1 < nk) {
                tile_store<BM, ATR, BK>(nA, ra[(p + 1) % PF]);
                tile_store<BN, BTR, BK>(nB, rb[(p + 1) % PF]);
            }
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- epilogue
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const int col_l = lane & 31, rsub = 4 * (lane >> 5);
    if (ATR && do_bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32);             // the fragment's two k halves (frag_read: lanes r and r + 32)
            const int row = m0 + (wm * TM + i) * 32 + col_l;
            if (lane < 32 && row < g.M) atomicAdd(g.dbias + row, tot);
        }
    }
    if (g.mode == EPI_PARTIAL) {
        // Split-K without atomics (OPTIONAL: the caller hands over a workspace; measured SLOWER than the atomics on MI355X, see
        // DESIGN.md section 4 -- kept because it is deterministic: dW no longer depends on the order in which atomics land).
        // Every slice parks its 64 KB accumulator tile in the workspace with 16-byte stores IN ACCUMULATOR ORDER (lane-contiguous: one
        // wave instruction = 1 KB, no LDS staging, and the reader owns the same positions), takes a ticket, and the slice that
        // arrives last adds the others' tiles to its registers in slice order and updates dW with ordinary loads and stores -- it
        // is the only writer of that tile.
        const int tile = by * g.ntx + bx;
        // Hand-off without cache maintenance: an agent-scope release / acquire pair would write back and INVALIDATE the XCD's L2 for
        // every workgroup (measured: 2.4x slower, the operand reuse of the slices still running is gone).  Instead every store of the
        // partial tile is an sc1 (device-coherent, write-through) store, drained (s_waitcnt) before the workgroup's barrier and the
        // ticket, and every load of another slice's tile is an sc1 load -- MI355X_MICROARCH.md, "Inter-workgroup visibility".
        float* P = g.part + ((size_t)tile * g.splitk + bz) * (size_t)(BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    float* dst = P + ((size_t)(((wave * TM + i) * TN + j) * 4 + q) * 64 + lane) * 4;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tile has left for the coherence point before the ticket is taken
        int* flag = reinterpret_cast<int*>(lds);
        __syncthreads();
        if (threadIdx.x == 0) *flag = (atomicAdd(g.cnt + tile, 1) == g.splitk - 1);
        __syncthreads();
        if (!*flag) return;
        for (int z = 0; z < g.splitk; ++z) {
            if (z == bz) continue;
            const float* Q = g.part + ((size_t)tile * g.splitk + z) * (size_t)(BM * BN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4_t v0, v1, v2, v3;
                    const float* src = Q + ((size_t)(((wave * TM + i) * TN + j) * 4) * 64 + lane) * 4;
                    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                                 "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                                 "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
                                 "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
                                 "s_waitcnt vmcnt(0)"
                                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(src) : "memory");
                    acc[i][j][0] += v0.x; acc[i][j][1] += v0.y; acc[i][j][2] += v0.z; acc[i][j][3] += v0.w;
                    acc[i][j][4] += v1.x; acc[i][j][5] += v1.y; acc[i][j][6] += v1.z; acc[i][j][7] += v1.w;
                    acc[i][j][8] += v2.x; acc[i][j][9] += v2.y; acc[i][j][10] += v2.z; acc[i][j][11] += v2.w;
                    acc[i][j][12] += v3.x; acc[i][j][13] += v3.y; acc[i][j][14] += v3.z; acc[i][j][15] += v3.w;
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + col_l;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + rsub;
                    if (n < g.N && m < g.M) {
                        float* o = reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + n;
                        *o += acc[i][j][r];                // dW += : the gradient buffer may already hold other contributions
                    }
                }
            }
        if (threadIdx.x == 0) g.cnt[tile] = 0;            // ready for the next launch (ordered by the kernel boundary)
        return;
    }
    if (g.mode == EPI_ATOMIC) {
        if (g.dbg & 1) { if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(g.C)[0] = 1.f; return; }
        if constexpr (PAIR) {
            // group h keeps the wave tiles i == h of the 128 x 128 tile: it parks the OTHER half of its accumulators in its (dead) staging
            // buffers in accumulator order (lane-contiguous: conflict-free), and after the barrier adds the other group's parked half
            static_assert(TM == 2, "the pair exchange splits the tile by its two 32-row blocks per wave");
            float* mine = reinterpret_cast<float*>(lds);
            float* theirs = reinterpret_cast<float*>(lds_all + (half ^ 1) * 2 * STAGE);
            __syncthreads();                        // (both groups are past their last fragment reads)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((wave * TN + j) * 16 + r) * 64 + lane] = half ? acc[0][j][r] : acc[1][j][r];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float o = theirs[((wave * TN + j) * 16 + r) * 64 + lane];
                    if (half) acc[1][j][r] += o; else acc[0][j][r] += o;
                }
        }
        // split-K partial sums: fp32 atomics straight from the accumulators (128-byte row segments)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (PAIR && i != half) continue;
                const int n = n0 + (wn * TN + j) * 32 + col_l;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + rsub;
                    if (n < g.N && m < g.M) atomicAdd(reinterpret_cast<float*>(g.C) + (size_t)cb + (size_t)m * g.ldc + n, acc[i][j][r]);
                }
            }
        return;
    }
    // Every other mode: the whole BM x BN accumulator tile is parked in LDS as fp32 (+bias; the operand buffers are dead),
    // ONE barrier, then each thread finishes 4 consecutive columns of a row: GELU / GELU' / dropout+residual / group bias /
    // group max with 16-byte loads of the side inputs and 8- or 16-byte stores.
    constexpr int SLD = BN + 4;
    if constexpr (BM * SLD * 4 > 2 * STAGE * 2) return;     // tile shapes instantiated for EPI_ATOMIC only (grouped wgrad)
    float* sf = reinterpret_cast<float*>(lds);
    VpfRng rng;
    if (g.mode == EPI_DROP_RES) rng = vpf_rng_init(g.rng, g.site, g.p);
    const bool vec = (g.N % 4 == 0) && (g.ldc % 4 == 0) && (((uintptr_t)g.C & 15) == 0) && (g.sCb % 4 == 0) &&
                     (g.mode != EPI_GELU || ((g.ldc2 % 4 == 0) && (((uintptr_t)g.C2 & 7) == 0))) &&
                     (g.mode != EPI_DROP_RES || ((g.ldres % 4 == 0) && (((uintptr_t)g.res & 15) == 0))) &&
                     (g.mode != EPI_GELU_BWD || ((g.ldaux % 4 == 0) && (((uintptr_t)g.aux & 7) == 0))) &&
                     (g.mode != EPI_GROUPBIAS || (((uintptr_t)g.gbias & 15) == 0));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wn * TN + j) * 32 + col_l;
            const float bv = (g.bias && n0 + nl < g.N) ? g.bias[n0 + nl] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sf[((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + rsub) * SLD + nl] = acc[i][j][r] + bv;
        }
    __syncthreads();
    if (g.mode == EPI_GROUPMAX) {
        // max over the `group` consecutive rows (group divides 32, tiles are 32-row aligned)
        const int gpb = BM / g.group;
        for (int e = threadIdx.x; e < gpb * (BN / 4); e += 256) {
            const int gi = e / (BN / 4), sc = (e % (BN / 4)) * 4;
            const int m = m0 + gi * g.group, n = n0 + sc;
            if (m >= g.M || n >= g.N) continue;
            float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int bi[4] = {0, 0, 0, 0};
            for (int k = 0; k < g.group; ++k) {
                const float4 a4 = *reinterpret_cast<const float4*>(sf + (gi * g.group + k) * SLD + sc);
                const float vv[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = h16_to_f32(f32_to_h16(vv[q]));
                    if (t > best[q]) { best[q] = t; bi[q] = k; }
                }
            }
            const size_t go = (size_t)(m / g.group) * g.ldc + n;
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + go) = make_float4(best[0], best[1], best[2], best[3]);
            *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(g.C2) + (size_t)(m / g.group) * g.ldc2 + n) =
                (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
        }
        return;
    }
    for (int e = threadIdx.x; e < BM * (BN / 4); e += 256) {
        const int sr = e / (BN / 4), sc = (e % (BN / 4)) * 4;
        const int m = m0 + sr, n = n0 + sc;
        if (m >= g.M || n >= g.N) continue;
        const float4 a4 = *reinterpret_cast<const float4*>(sf + sr * SLD + sc);
        float v[4] = {a4.x, a4.y, a4.z, a4.w};
        const int nv = min(4, g.N - n);
        const size_t co = (size_t)cb + (size_t)m * g.ldc + n;
        if (g.mode == EPI_GELU) {
            h16_t ub[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { ub[q] = f32_to_h16(v[q]); v[q] = gelu_f(h16_to_f32(ub[q])); }
            h16_t* u = reinterpret_cast<h16_t*>(g.C2) + (size_t)m * g.ldc2 + n;
            if (vec) { uint2 w; w.x = ub[0] | ((uint32_t)ub[1] << 16); w.y = ub[2] | ((uint32_t)ub[3] << 16); *reinterpret_cast<uint2*>(u) = w; }
            else { for (int q = 0; q < nv; ++q) u[q] = ub[q]; }
        } else if (g.mode == EPI_DROP_RES) {
            const float* rp = g.res + (size_t)m * g.ldres + n;
            float rr[4] = {0.f, 0.f, 0.f, 0.f};
            if (vec) { const float4 t = *reinterpret_cast<const float4*>(rp); rr[0] = t.x; rr[1] = t.y; rr[2] = t.z; rr[3] = t.w; }
            else { for (int q = 0; q < nv; ++q) rr[q] = rp[q]; }
            const uint64_t base = (uint64_t)m * (uint64_t)g.N + (uint64_t)n;
            const uint32_t keep = vpf_keep4_at(rng, base);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = rr[q] + (((keep >> q) & 1u) ? v[q] * rng.scale : 0.f);
        } else if (g.mode == EPI_GELU_BWD) {
            const h16_t* ap = g.aux + (size_t)m * g.ldaux + n;
            h16_t ab[4] = {0, 0, 0, 0};
            if (vec) { const uint2 t = *reinterpret_cast<const uint2*>(ap); ab[0] = t.x & 0xffff; ab[1] = t.x >> 16; ab[2] = t.y & 0xffff; ab[3] = t.y >> 16; }
            else { for (int q = 0; q < nv; ++q) ab[q] = ap[q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] *= gelu_grad_f(h16_to_f32(ab[q]));
        } else if (g.mode == EPI_RELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        } else if (g.mode == EPI_GROUPBIAS) {
            const float* gp = g.gbias + (size_t)(m / g.group) * g.N + n;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q < nv) v[q] += gp[q];
        }
        if (g.c_f32) {
            float* o = reinterpret_cast<float*>(g.C) + co;
            if (vec) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            else { for (int q = 0; q < nv; ++q) o[q] = v[q]; }
        } else {
            h16_t* o = reinterpret_cast<h16_t*>(g.C) + co;
            if (vec) { uint2 w; w.x = pack_h16x2(v[0], v[1]); w.y = pack_h16x2(v[2], v[3]); *reinterpret_cast<uint2*>(o) = w; }
            else { for (int q = 0; q < nv; ++q) o[q] = f32_to_h16(v[q]); }
        }
    }
}

template <int TM, int TN, int WM, int WN, int BK, bool ATR, bool BTR, int AX, int BX>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g)
{
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.splitk > 1 && (g.splitk & 7) == 0) {
        // split-K: give the tiles of one K slice ids that are congruent mod 8 (same XCD, same L2), see gemm_wgrad_group_kernel
        const int nx = gridDim.x, ny = gridDim.y;
        const int id = bx + nx * (by + ny * bz);
        const int x = id & 7, u = id >> 3, tile = u % (nx * ny), zg = u / (nx * ny);
        bz = zg * 8 + x; bx = tile % nx; by = tile / nx;
    } else if (g.splitk <= 1 && gridDim.z == 1 && gridDim.x > 1 && (gridDim.y & 7) == 0) {
        // the column tiles of one row tile read the same A rows: keep them on one XCD (ids congruent mod 8)
        const int nx = gridDim.x;
        const int id = bx + nx * by;
        by = (id & 7) + 8 * (id / (8 * nx)); bx = (id >> 3) % nx;
    }
    gemm_tile<TM, TN, WM, WN, BK, ATR, BTR, AX, BX>(g, bx, by, bz);
}

// Several independent split-K weight-gradient GEMMs in ONE launch (the four of a transformer layer): each of them alone
// is a few hundred short workgroups whose ramp-up and tail dominate; together they fill the chip.
// Round 3: up to 32 problems -- the weight gradients of a whole encoder stack in one launch (ops.WgradBatch): with 7 x 32 output tiles
// two K slices per tile already give ~450 workgroups, so the split-K flush (fp32 atomics at ~1.3 TB/s, 33 MB = 23 us of a 43 us
// one-layer launch) is paid once per stack instead of once per layer.  The kernel argument holds compact descriptors (a GemmArgs
// per problem would not fit the 4 KB argument segment); the workgroup builds the GemmArgs of its problem in registers.
#define GEMM_GROUP_MAX 32
struct WgDesc {
    const h16_t* A; const h16_t* B; float* C; float* dbias; float* part; int* cnt;
    int lda, ldb, ldc, M, N, K, splitk, ntx, mode;
};
#define WGROUP_XLIST_MAX 96
struct GemmGroup {
    WgDesc d[GEMM_GROUP_MAX]; int start[GEMM_GROUP_MAX + 1]; short nx[GEMM_GROUP_MAX], ny[GEMM_GROUP_MAX]; int n, uneven, dbg;
    // XCD lists (few K slices per tile -- a whole stack in one launch): every (problem, K slice) = "slab" is dealt to ONE XCD -- all its
    // output tiles read the same dY / X rows, so its operands cross HBM once instead of once per XCD that holds one of its tiles
    // (1 126 -> ~600 MB per stack launch).  Workgroup id -> XCD id % 8 (the dispatcher's round-robin), slot id / 8 -> walk the XCD's list.
    int xmode;                                              // 1: the lists below are in use; grid = 8 x the longest list's tile count
    unsigned char xp[WGROUP_XLIST_MAX], xz[WGROUP_XLIST_MAX], xoff[9];     // slab i: problem xp[i], slice xz[i]; XCD x owns slabs xoff[x] .. xoff[x + 1] - 1
};
template <int TM, int TN, int WM, int WN, int BK, int PF, bool PAIR = false>
__global__ void __launch_bounds__(PAIR ? 512 : 256) gemm_wgrad_group_kernel(GemmGroup grp)
{
    int p = 0, local = 0, xbz = 0;
    if (grp.xmode) {
        const int x = blockIdx.x & 7;
        int slot = blockIdx.x >> 3, i = grp.xoff[x];
        const int end = grp.xoff[x + 1];
        for (; i < end; ++i) {
            const int nt = grp.nx[grp.xp[i]] * grp.ny[grp.xp[i]];
            if (slot < nt) break;
            slot -= nt;
        }
        if (i >= end) return;                               // this XCD's list is shorter than the longest one
        p = grp.xp[i]; xbz = grp.xz[i]; local = slot;
    } else {
        while (p + 1 < grp.n && (int)blockIdx.x >= grp.start[p + 1]) ++p;
        local = blockIdx.x - grp.start[p];
    }
    const WgDesc& d = grp.d[p];
    GemmArgs g = {};
    g.A = d.A; g.B = d.B; g.lda = d.lda; g.ldb = d.ldb; g.M = d.M; g.N = d.N; g.K = d.K; g.splitk = d.splitk; g.mode = d.mode;
    g.C = d.C; g.ldc = d.ldc; g.c_f32 = 1; g.dbias = d.dbias; g.group = 1; g.uneven = grp.uneven; g.dbg = grp.dbg;
    g.part = d.part; g.cnt = d.cnt; g.ntx = d.ntx;
    const int nx = grp.nx[p], ny = grp.ny[p], sk = PAIR ? g.splitk / 2 : g.splitk;      // PAIR: a workgroup = two K slices
    int bx, by, bz;
    if (grp.xmode) {
        bz = xbz; bx = local % nx; by = local / nx;
    } else if ((sk & 7) == 0 && (grp.start[p] & 7) == 0) {
        // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  All output tiles
        // of one token slice read the same dY / X rows, so a slice's tiles are given ids that are congruent mod 8: the
        // slice's operands are fetched from HBM once per XCD instead of once per tile.
        const int x = local & 7, u = local >> 3, tile = u % (nx * ny), zg = u / (nx * ny);
        bz = zg * 8 + x; bx = tile % nx; by = tile / nx;
    } else {
        bx = local % nx; by = (local / nx) % ny; bz = local / (nx * ny);
    }
    gemm_tile<TM, TN, WM, WN, BK, true, true, 0, 0, PF, PAIR>(g, bx, by, bz);
}
// ------------------------------------------------------------------ grouped weight gradient, LDS-DMA staging (round 5)
// What the anatomy of the STACK launch says (tools/microbench.py wstack under a kernel trace, NOTES.md round 5): 208 us, of which the
// staging loop alone -- global -> registers -> LDS + one barrier per 64-token stage, no MFMA, no fragment reads, no flush -- is 158 us.
// It moves 1.4 GB at 18 B/clk per CU where 55 B/clk is available: every stage exposes one L2 / HBM round trip (the registers that carry
// a stage are refilled one iteration ahead of their store), and a deeper register pipeline costs the occupancy it needs.
// Here the stages travel by LDS-DMA (global_load_lds_dwordx4: memory -> LDS, no registers, no ds_write): three 48 KB stages per
// workgroup, two in flight behind the one on the matrix cores.  256 x 128 output tiles on 8 waves (a quarter fewer staged bytes than
// 128 x 128: every B row block serves 256 output rows), ONE workgroup per CU (144 KB of LDS), <= 256 workgroups per launch.
//   * The K-strided tile layout of gemm_tile ([BK][ROWS], 32-row column blocks of k row k XOR-permuted by k & 3: conflict-free
//     transposing reads) is produced on the GLOBAL side: a DMA wave-instruction writes 1 KB of LDS contiguously (lane i -> +16 i), so
//     lane i fetches the 16-byte chunk whose swizzled position is i -- chunk (pos ^ ((k & 3) << 2)) of its k row; all within the
//     row's contiguous ROWS x 2 bytes, coalescing is unchanged.
//   * The DMA is issued from inline asm: through the builtin the compiler (correctly, not knowing better) waits for vmcnt(0) in front
//     of every LDS read that might alias a pending DMA and in front of every __syncthreads, which is the pipeline.  Protocol per stage:
//     every wave waits for ITS OWN copies of stage kt (s_waitcnt vmcnt(PER): the next stage's PER copies may stay in flight; vmcnt
//     retires in order), one s_barrier (all waves' copies have landed, all waves are done with stage kt - 1), the copies of stage
//     kt + 2 go into stage kt - 1's buffer, then the products of stage kt.
// Conforming problems only (wgrad_dma_conforms; in THIS kernel's names M = N_out rows of dW, N = K_in columns, K = tokens):
// N_out % 128 == 0, K_in % 128 == 0, tokens % 64 == 0, 16-byte aligned operands, atomics flush.  Tile rows: 256 when every problem of the
// launch has N_out % 256 == 0, else 128 for the WHOLE launch (TM = 1: one N_out that is a multiple of 128 only moves all of them).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"          // (m0 is a reserved register: it is exactly what this instruction takes its LDS address from)
__device__ __forceinline__ void wg_dma16(const void* gptr, unsigned lds_byte)
{
    // s_nop 0: a SALU write of M0 followed by an LDS-DMA read of M0 is a GFX9-family hazard that needs one wait state; the compiler's
    // hazard recogniser does not look inside inline asm, so the wait state is written out (as in sa_rows.hip st16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_byte), "v"(gptr) : "memory", "m0");
}
#pragma clang diagnostic pop
#define WGDMA_BM 256           // tile rows of every configuration; tile columns: 128 (configuration 0) or 256 (configuration 1)
#define WGDMA_BN 128
#define WGDMA_BK 64            // conforming problems have a multiple of 64 tokens (both configurations' stages divide it)
// Configurations <TN, BK, NS> on 8 waves (4 x 2), wave tile 64 x 32 TN:
//   <2, 64, 3>  256 x 128 tiles, three 48 KB stages of 64 tokens                 (any conforming problem)
//   <4, 32, 4>  256 x 256 tiles, four 32 KB stages of 32 tokens: a third fewer staged bytes (every A row block serves 256 output columns
//               too) and 0.75 instead of 1 fragment read per MFMA, for twice the flushed tile  (problems whose K_in is a multiple of 256).
//               Measured: 151.8 vs 148.2 us for the stack launch and +0.06 ms per step (3.404 vs 3.344 ms, three alternating pairs) --
//               200 registers, half the slices' length, twice the barriers per token.  Kept behind VPF_WGROUP_DMA_TN=0.
//   <TM 1: 2, 64, 3>  128 x 128 tiles (three 32 KB stages) for problems whose N_out is a multiple of 128 only (D = 384: config 4)
template <int TN, int BK, int NS, int TM = 2>
__global__ void __launch_bounds__(512, 2) gemm_wgrad_dma_kernel(GemmGroup grp)
{
    constexpr int BM = 128 * TM, BN = 64 * TN;
    using ACfg = TileCfg<BM, true, BK>;
    using BCfg = TileCfg<BN, true, BK>;
    constexpr int STAGE = ACfg::ELEMS + BCfg::ELEMS;                 // h16 elements per stage
    constexpr int RA = 1024 / (BM * 2), RB = 1024 / (BN * 2);        // k rows per 1 KB DMA wave-instruction
    constexpr int NA = BK / RA / 8, NB = BK / RB / 8, PER = NA + NB;  // DMA instructions per wave and stage
    constexpr int KS = BK / 16;                                      // k-steps per stage
    static_assert(NA * RA * 8 == BK && NB * RB * 8 == BK && PER >= KS, "the copies divide over the waves and the k-steps");
    extern __shared__ __attribute__((aligned(16))) h16_t lds_dma[];
    // ---- which problem, tile and K slice (as gemm_wgrad_group_kernel)
    int p = 0, local = 0, xbz = 0;
    if (grp.xmode) {
        const int x = blockIdx.x & 7;
        int slot = blockIdx.x >> 3, i = grp.xoff[x];
        const int end = grp.xoff[x + 1];
        for (; i < end; ++i) {
            const int nt = grp.nx[grp.xp[i]] * grp.ny[grp.xp[i]];
            if (slot < nt) break;
            slot -= nt;
        }
        if (i >= end) return;
        p = grp.xp[i]; xbz = grp.xz[i]; local = slot;
    } else {
        while (p + 1 < grp.n && (int)blockIdx.x >= grp.start[p + 1]) ++p;
        local = blockIdx.x - grp.start[p];
    }
    const WgDesc& d = grp.d[p];
    const int nx = grp.nx[p], ny = grp.ny[p];
    int bx, by, bz;
    if (grp.xmode) { bz = xbz; bx = local % nx; by = local / nx; }
    else { bx = local % nx; by = (local / nx) % ny; bz = local / (nx * ny); }
    const int m0 = by * BM, n0 = bx * BN;
    int kbeg, kend;
    if (grp.uneven > 0 && d.splitk >= 16) {
        // Many slices of ONE or a few tiles (a single weight gradient over 10^5 tokens and more): equal slices end together and their
        // 128 KB flushes -- 33 MB of fp32 atomics at ~1.3 TB/s, all onto the same tile -- queue up behind the last stage (26 us of a
        // 70 us launch).  A linear RAMP of slice lengths, (1 - r) .. (1 + r) x the mean with r = uneven / 100, lets the early finishers'
        // flushes drain under the others' stages.  Boundaries in whole 64-token stages: f(z) = T z ((1 - r) + r (z - 1) / (S - 1)) / S.
        const long T = d.K / WGDMA_BK, S = d.splitk;
        auto f = [&](long z) -> long {
            const double r = 0.01 * grp.uneven;
            const double v = (double)T * (double)z * ((1.0 - r) + r * (double)(z - 1) / (double)(S - 1)) / (double)S;
            long q = (long)(v + 0.5);
            return q < 0 ? 0 : (q > T ? T : q);
        };
        kbeg = (int)(f(bz) * WGDMA_BK); kend = bz + 1 == S ? d.K : (int)(f(bz + 1) * WGDMA_BK);
    } else {
        const int per = ((d.K + d.splitk - 1) / d.splitk + WGDMA_BK - 1) / WGDMA_BK * WGDMA_BK;
        kbeg = bz * per; kend = min(d.K, kbeg + per);
    }
    if (kbeg >= kend) return;
    const int nk = (kend - kbeg) / BK;

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    // ---- this lane's part of a stage: element offsets inside the stage's BK k rows
    unsigned offA[NA], offB[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int q = wave + 8 * j, kk = RA * q + lane / (64 / RA), pos = lane % (64 / RA);
        offA[j] = (unsigned)kk * (unsigned)d.lda + (unsigned)((pos ^ ((kk & 3) << 2)) * 8);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = wave + 8 * j, kk = RB * q + lane / (64 / RB), pos = lane % (64 / RB);
        offB[j] = (unsigned)kk * (unsigned)d.ldb + (unsigned)((pos ^ ((kk & 3) << 2)) * 8);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) h16_t*)lds_dma;
    const h16_t* Ag = d.A + (size_t)kbeg * d.lda + m0;
    const h16_t* Bg = d.B + (size_t)kbeg * d.ldb + n0;
    // copy `pc` (0 .. NA - 1: the A tile, NA .. PER - 1: the B tile) of stage st
    auto issue_piece = [&](int st, int pc) {
        const unsigned buf = lds0 + (unsigned)(st % NS) * (STAGE * 2);
        if (pc < NA) wg_dma16(Ag + (size_t)st * BK * d.lda + offA[pc < NA ? pc : 0], buf + (unsigned)(wave + 8 * pc) * 1024u);
        else wg_dma16(Bg + (size_t)st * BK * d.ldb + offB[pc >= NA ? pc - NA : 0], buf + (unsigned)(ACfg::ELEMS * 2) + (unsigned)(wave + 8 * (pc - NA)) * 1024u);
    };
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_bias = d.dbias != nullptr && bx == 0 && wn == 0;
    float bsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bsum[i] = 0.f;

    // ---- this lane's fragment addresses inside a stage (frag_read's transposing pattern: per 16-lane group a 4 (k) x 16 (row) block;
    //      the swizzle depends on k & 3 = q only, so ONE base per fragment serves all k-steps of a stage through immediate offsets)
    int aoff[TM], boff[TN];
    {
        const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, hh = g4 >> 1, ro = 16 * (g4 & 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) aoff[i] = (8 * hh + q) * ACfg::LD + ACfg::tr_swz(q, (wm * TM + i) * 32 + ro + 4 * pp);
#pragma unroll
        for (int j = 0; j < TN; ++j) boff[j] = ACfg::ELEMS + (8 * hh + q) * BCfg::LD + BCfg::tr_swz(q, (wn * TN + j) * 32 + ro + 4 * pp);
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"      // (LDS pointers are 32 bits wide)
    auto frag = [&](unsigned base, int ld, int ks) -> h16x8_t {                 // k-step ks of the fragment whose lane's LDS byte address is `base`
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(base + (unsigned)(ks * 16 * ld * 2)));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(base + (unsigned)((ks * 16 + 4) * ld * 2)));
        s16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(h16x8_t, v);
    };
#pragma clang diagnostic pop
    // ---- NS - 1 stages in flight at the start; per stage: wait for OUR copies of stage kt (the stages behind it may stay in flight:
    //      vmcnt retires in order), barrier, then stage kt + NS - 1 goes into stage kt - 1's buffer between the k-steps' products
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) {
#pragma unroll
            for (int pc = 0; pc < PER; ++pc) issue_piece(st, pc);
        }
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NS - 2, nk - 1 - kt);               // stages behind kt whose copies are in flight
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NS > 3 ? 2 * PER : 0) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = kt + NS - 1 < nk;
        // ONE address register per fragment and stage (opaque to the optimiser: otherwise it keeps 26 per-read lane offsets in registers
        // and adds the stage base to each with two VALU instructions per read); the k-steps are immediate offsets of the reads
        const unsigned cS = lds0 + (unsigned)(kt % NS) * (STAGE * 2);
        unsigned pa[TM], pb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { pa[i] = cS + (unsigned)aoff[i] * 2u; asm volatile("" : "+v"(pa[i])); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { pb[j] = cS + (unsigned)boff[j] * 2u; asm volatile("" : "+v"(pb[j])); }
        // the fragments of k-step s + 1 are requested in front of the products of k-step s
        h16x8_t fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = frag(pa[i], ACfg::LD, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = frag(pb[j], BCfg::LD, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[(s + 1) & 1][i] = frag(pa[i], ACfg::LD, s + 1);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[(s + 1) & 1][j] = frag(pb[j], BCfg::LD, s + 1);
            }
            if (more) {                      // the PER copies spread over the KS k-steps: the matrix cores start right behind the barrier
#pragma unroll
                for (int pc = s * PER / KS; pc < (s + 1) * PER / KS; ++pc) issue_piece(kt + NS - 1, pc);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = vpf_mfma32(fa[s & 1][i], fb[s & 1][j], acc[i][j]);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const uint4 w = __builtin_bit_cast(uint4, fa[s & 1][i]);
                    bsum[i] = h16_dot2(w.x, VPF_H16_ONES2, bsum[i]);
                    bsum[i] = h16_dot2(w.y, VPF_H16_ONES2, bsum[i]);
                    bsum[i] = h16_dot2(w.z, VPF_H16_ONES2, bsum[i]);
                    bsum[i] = h16_dot2(w.w, VPF_H16_ONES2, bsum[i]);
                }
            }
        }
        // (the LDS reads of this stage are complete before a wave's MFMAs consume them, i.e. before it reaches the next barrier)
    }
    // ---- flush: bias sums, then fp32 atomics straight from the accumulators (128-byte row segments), as gemm_tile
    const int col_l = lane & 31, rsub = 4 * (lane >> 5);
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
            const int row = m0 + (wm * TM + i) * 32 + col_l;
            if (lane < 32 && row < d.M) atomicAdd(d.dbias + row, tot);
        }
    }
    if (grp.dbg & 1) { if (acc[0][0][0] == 12345.678f) d.C[0] = 1.f; return; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + col_l;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + rsub;
                atomicAdd(d.C + (size_t)m * d.ldc + n, acc[i][j][r]);
            }
        }
}
template <int TN, int BK, int NS, int TM = 2>
static int launch_wgrad_dma_cfg(const GemmGroup& grp, int nblocks, hipStream_t st)
{
    constexpr size_t lds = sizeof(h16_t) * NS * (TileCfg<128 * TM, true, BK>::ELEMS + TileCfg<64 * TN, true, BK>::ELEMS);
    static_assert(lds <= 160 * 1024, "the stages must fit the CU's LDS");
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)gemm_wgrad_dma_kernel<TN, BK, NS, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((gemm_wgrad_dma_kernel<TN, BK, NS, TM>), dim3(nblocks), dim3(512), lds, st, grp);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
static int launch_wgrad_dma(const GemmGroup& grp, int nblocks, int tm, int tn, hipStream_t st)
{
    if (tm == 128) return launch_wgrad_dma_cfg<2, 64, 3, 1>(grp, nblocks, st);
    if (tn == 256) return launch_wgrad_dma_cfg<4, 32, 4>(grp, nblocks, st);
    return launch_wgrad_dma_cfg<2, 64, 3>(grp, nblocks, st);
}

template <int TM, int TN, int WM, int WN, int BK, int PF = 1, bool PAIR = false>
static int launch_wgrad_group(const GemmGroup& grp, int nblocks, hipStream_t st)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t lds = sizeof(h16_t) * 2 * (TileCfg<BM, true, BK>::ELEMS + TileCfg<BN, true, BK>::ELEMS) * (PAIR ? 2 : 1);
    static_assert(!PAIR || sizeof(h16_t) * 2 * (TileCfg<BM, true, BK>::ELEMS + TileCfg<BN, true, BK>::ELEMS) >= (size_t)BM * BN * 4 / 2,
                  "a group's staging buffers must hold half an accumulator tile");
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (lds > 65536 && hipFuncSetAttribute((const void*)gemm_wgrad_group_kernel<TM, TN, WM, WN, BK, PF, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((gemm_wgrad_group_kernel<TM, TN, WM, WN, BK, PF, PAIR>), dim3(nblocks), dim3(PAIR ? 512 : 256), lds, st, grp);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

template <int TM, int TN, int WM, int WN, int BK, bool ATR, bool BTR, int AX, int BX>
static int launch_one(const GemmArgs& g, dim3 grid, hipStream_t st)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t lds = sizeof(h16_t) * 2 * (TileCfg<BM, ATR, BK>::ELEMS + TileCfg<BN, BTR, BK>::ELEMS);
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (lds > 65536 && hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, BK, ATR, BTR, AX, BX>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, BK, ATR, BTR, AX, BX>), grid, dim3(256), lds, st, g);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
template <int TM, int TN, int WM, int WN, int BK>
static int launch_cfg(const GemmArgs& g, int a_tr, int b_tr, int batch, hipStream_t st)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    dim3 grid(vpf_cdiv(g.N, BN), vpf_cdiv(g.M, BM), g.splitk > 1 ? g.splitk : batch);
    const int ax = g.xa.kind, bx = g.xb.kind;
    if (ax == 0 && bx == 0) {
        if (!a_tr && !b_tr) return launch_one<TM, TN, WM, WN, BK, false, false, 0, 0>(g, grid, st);
        if (!a_tr && b_tr) return launch_one<TM, TN, WM, WN, BK, false, true, 0, 0>(g, grid, st);
        if (a_tr && !b_tr) return launch_one<TM, TN, WM, WN, BK, true, false, 0, 0>(g, grid, st);
        return launch_one<TM, TN, WM, WN, BK, true, true, 0, 0>(g, grid, st);
    }
    // operand prologues: only the combinations the Group2Emb tail uses are instantiated
    if (ax == 1 && bx == 0 && !a_tr && !b_tr) return launch_one<TM, TN, WM, WN, BK, false, false, 1, 0>(g, grid, st);   // fwd: relu(bn(h3)) . W^T
    if (ax == 2 && bx == 0 && !a_tr && b_tr) return launch_one<TM, TN, WM, WN, BK, false, true, 2, 0>(g, grid, st);     // dgrad: dmax . W
    if (ax == 2 && bx == 1 && a_tr && b_tr) return launch_one<TM, TN, WM, WN, BK, true, true, 2, 1>(g, grid, st);       // wgrad: dmax^T . relu(bn(h3))
    return VPF_ERR_UNSUPPORTED;
}

static bool wgrad_dma_conforms(const VpfWgradJob& j);
static int wgrad_single_dma(const VpfWgradJob& j, int ldc, hipStream_t st);
static int gemm_dispatch(GemmArgs& g, int a_tr, int b_tr, int batch, hipStream_t st)
{
    if (!g.A || !g.B || !g.C) return VPF_ERR_NULL;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || batch <= 0) return VPF_ERR_BADSHAPE;
    // 16-byte vector loads along the contiguous dimension of each operand
    const int acont = a_tr ? g.M : g.K, bcont = b_tr ? g.N : g.K;
    if ((acont % 8) || (bcont % 8) || (g.lda % 8) || (g.ldb % 8) || (g.sAb % 8) || (g.sBb % 8)) return VPF_ERR_BADALIGN;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15)) return VPF_ERR_BADALIGN;
    if (g.mode == EPI_ATOMIC) {
        if (!g.c_f32) return VPF_ERR_UNSUPPORTED;
        // round 5: a single weight gradient that fits the LDS-DMA kernel and can fill the chip with slices of >= 1 024 tokens goes there
        // (Group2Emb's dW3[:, 128:] = dh3^T h2: 393 216 tokens, one 256 x 128 tile, 256 slices)
        if (a_tr && b_tr && batch == 1 && g.splitk <= 0 && g.xa.kind == 0 && g.xb.kind == 0 && g.ldc <= 0x7fffffff) {
            VpfWgradJob j = {g.A, g.B, g.K, g.M, g.N, reinterpret_cast<float*>(g.C), g.dbias};
            const long tiles = (long)(g.M / WGDMA_BM) * (g.N / WGDMA_BN);
            if (wgrad_dma_conforms(j) && !(g.M % WGDMA_BM) && g.lda == g.M && g.ldb == g.N && tiles > 0 && tiles * (256 / tiles < g.K / 1024 ? 256 / tiles : g.K / 1024) >= 128)
                return wgrad_single_dma(j, (int)g.ldc, st);
        }
        const int wcfg = vpf_debug().wgrad_cfg, wtarget = vpf_debug().wgrad_wgs > 0 ? vpf_debug().wgrad_wgs : 512;
        const int tm = wcfg == 0 ? 64 : 128, tn = wcfg == 2 ? 128 : 64;
        if (g.splitk <= 0) {
            // fill ~512 workgroups of 64x64 tiles
            const long tiles = (long)vpf_cdiv(g.M, tm) * vpf_cdiv(g.N, tn);
            long s = wtarget / (tiles > 0 ? tiles : 1);
            const long maxs = vpf_cdiv(g.K, 256);
            if (s > maxs) s = maxs;
            if (s > 8) s &= ~7L;                  // a multiple of 8: the kernel's per-slice XCD ordering applies (a slice's tiles share one L2)
            if (s < 1) s = 1;
            g.splitk = (int)s;
        }
        if (g.splitk == 1) g.splitk = 0, batch = 1;
        if (batch != 1 && g.splitk > 1) return VPF_ERR_UNSUPPORTED;
        if (wcfg == 2) return launch_cfg<2, 2, 2, 2, 64>(g, a_tr, b_tr, batch, st);
        if (wcfg == 1) return launch_cfg<1, 2, 4, 1, 64>(g, a_tr, b_tr, batch, st);
        return launch_cfg<1, 1, 2, 2, 128>(g, a_tr, b_tr, batch, st);
    }
    g.splitk = 0;
    // Tile choice.  These GEMMs are small (M ~ 12 k tokens, N, K in 256..768) and bound by the bytes each CU has to pull
    // (A re-read once per column tile, W once per row tile), so the largest tile that still gives every CU a workgroup wins.
    const int forced = vpf_debug().gemm_cfg;
    const long wg128 = (long)vpf_cdiv(g.M, 128) * vpf_cdiv(g.N, 128) * batch;
    const long wg_128x64 = (long)vpf_cdiv(g.M, 128) * vpf_cdiv(g.N, 64) * batch;
    int cfg = (wg128 >= 160 && g.N >= 128) ? 2 : (wg_128x64 >= 192 ? 1 : 0);
    if (forced >= 0) cfg = forced;
    if (cfg == 2) return launch_cfg<2, 2, 2, 2, 64>(g, a_tr, b_tr, batch, st);    // 128x128
    if (cfg == 1) return launch_cfg<1, 2, 4, 1, 64>(g, a_tr, b_tr, batch, st);    // 128x64
    return launch_cfg<1, 1, 2, 2, 128>(g, a_tr, b_tr, batch, st);                 // 64x64
}

// ------------------------------------------------------------------ C ABI
extern "C" int vpf_gemm_h16(const void* A, int a_kstrided, long lda, const void* B, int b_kstrided, long ldb,
                             int M, int N, int K, int batch, long sAb, long sBb, long sCb,
                             void* C, long ldc, int c_is_f32, int mode, const float* bias,
                             void* C2, long ldc2, const float* res, long ldres, const void* aux, long ldaux,
                             const float* gbias, int group, const uint32_t* rng_state, uint32_t site, float p,
                             int splitk, float* dbias, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    GemmArgs g;
    g.A = (const h16_t*)A; g.B = (const h16_t*)B; g.lda = lda; g.ldb = ldb;
    g.sAb = sAb; g.sBb = sBb; g.sCb = sCb; g.M = M; g.N = N; g.K = K; g.splitk = splitk; g.mode = mode;
    g.C = C; g.ldc = ldc; g.c_f32 = c_is_f32; g.C2 = C2; g.ldc2 = ldc2; g.bias = bias; g.res = res; g.ldres = ldres;
    g.aux = (const h16_t*)aux; g.ldaux = ldaux; g.gbias = gbias; g.group = group > 0 ? group : 1;
    g.rng = rng_state; g.site = site; g.p = p; g.dbias = dbias;
    g.xa.kind = 0; g.xb.kind = 0; g.uneven = 0; g.dbg = 0;
    if (dbias && !(mode == EPI_ATOMIC && a_kstrided)) return VPF_ERR_UNSUPPORTED;
    if (mode < 0 || mode > EPI_GROUPBIAS) return VPF_ERR_UNSUPPORTED;   // EPI_GROUPMAX: vpf_gemm_h16_fused
    if (mode == EPI_GELU && !C2) return VPF_ERR_NULL;
    if (mode == EPI_DROP_RES && (!res || !rng_state || !c_is_f32)) return VPF_ERR_NULL;
    if (mode == EPI_GELU_BWD && !aux) return VPF_ERR_NULL;
    if (mode == EPI_GROUPBIAS && !gbias) return VPF_ERR_NULL;
    return gemm_dispatch(g, a_kstrided, b_kstrided, batch, (hipStream_t)stream);
}

// GEMM with operand prologues and the max-pool epilogue (Group2Emb's last conv and its backward):
//   a_kind / b_kind: 0 none | 1 relu(a[c]*x + b[c]) on the fly (xa = a, xb = b: f32 [channels]) |
//                    2 (A only) virtual max-pool gradient from (dout f32 [tokens/group, ncols], arg u8), A pointer ignored
//   mode: VPF_EPI_STORE / VPF_EPI_ATOMIC (+dbias) / 7 = group max: C f32 [M/group, N] (ldc), C2 u8 arg (ldc2)
extern "C" int vpf_gemm_h16_fused(const void* A, int a_kstrided, long lda, int a_kind, const float* a_scale, const float* a_shift,
                                   const float* a_dout, const uint8_t* a_arg, int a_group, long a_ncols,
                                   const void* B, int b_kstrided, long ldb, int b_kind, const float* b_scale, const float* b_shift,
                                   int M, int N, int K, void* C, long ldc, int c_is_f32, int mode, const float* bias,
                                   void* C2, long ldc2, int group, int splitk, float* dbias, void* stream)
{
    (void)hipGetLastError();   // drop any stale (non-sticky) error left by an earlier runtime call of this thread
    GemmArgs g = {};
    g.A = (const h16_t*)(a_kind == 2 ? (const void*)a_dout : A); g.B = (const h16_t*)B; g.lda = lda; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.splitk = splitk; g.mode = mode; g.C = C; g.ldc = ldc; g.c_f32 = c_is_f32; g.C2 = C2; g.ldc2 = ldc2;
    g.bias = bias; g.group = group > 0 ? group : 1; g.dbias = dbias;
    g.xa.kind = a_kind; g.xa.a = a_scale; g.xa.b = a_shift; g.xa.dout = a_dout; g.xa.arg = a_arg; g.xa.group = a_group > 0 ? a_group : 1; g.xa.ncols = a_ncols;
    g.xb.kind = b_kind; g.xb.a = b_scale; g.xb.b = b_shift; g.xb.group = 1;
    if (!(mode == EPI_STORE || mode == EPI_ATOMIC || mode == EPI_GROUPMAX)) return VPF_ERR_UNSUPPORTED;
    if (a_kind < 0 || a_kind > 2 || b_kind < 0 || b_kind > 1) return VPF_ERR_UNSUPPORTED;
    if ((a_kind == 1 && (!a_scale || !a_shift)) || (b_kind == 1 && (!b_scale || !b_shift)) || (a_kind == 2 && (!a_dout || !a_arg))) return VPF_ERR_NULL;
    if (a_kind == 2 && ((a_ncols % 8) || ((uintptr_t)a_dout & 15) || ((uintptr_t)a_arg & 7))) return VPF_ERR_BADALIGN;
    if (dbias && !(mode == EPI_ATOMIC && a_kstrided)) return VPF_ERR_UNSUPPORTED;
    if (mode == EPI_GROUPMAX) {
        if (!C2 || !c_is_f32) return VPF_ERR_NULL;
        if (g.group > 32 || (32 % g.group) || (M % g.group) || (N % 4) || (ldc % 4) || (ldc2 % 4) || ((uintptr_t)C & 15) || ((uintptr_t)C2 & 3))
            return VPF_ERR_UNSUPPORTED;
    }
    return gemm_dispatch(g, a_kstrided, b_kstrided, 1, (hipStream_t)stream);
}


// dW_i[N_i,K_i] += dY_i[M_i,N_i]^T . X_i[M_i,K_i]  (+ dbias_i[N_i] += column sums of dY_i) for up to 32 problems in one launch
#define WGROUP_CNT_INTS 1024        // arrival counters at the head of the workspace (zeroed ONCE by the caller; the kernel re-zeroes what it used)
// does job j fit the LDS-DMA kernel?  (multiples of its 256 x 128 tile and of its 64-token stages, enough tokens to stream)
static bool wgrad_dma_conforms(const VpfWgradJob& j)
{
    const int min_tokens = vpf_debug().wgroup_dma;
    return min_tokens > 0 && j.M >= min_tokens && j.M > 0 && !(j.N % 128) && !(j.K % WGDMA_BN) && !(j.M % WGDMA_BK) && j.N > 0 && j.K > 0;
}
static int wgrad_group_launch(const VpfWgradJob* jobs, int njobs, void* ws, long ws_bytes, void* stream, bool dma_ok, int ldc_override = 0);
extern "C" int vpf_wgrad_group(const VpfWgradJob* jobs, int njobs, void* ws, long ws_bytes, void* stream)
{
    (void)hipGetLastError();
    if (!jobs) return VPF_ERR_NULL;
    if (njobs <= 0 || njobs > GEMM_GROUP_MAX) return VPF_ERR_BADSHAPE;
    // round 5: conforming problems go to the LDS-DMA kernel, the others to the register-staged one -- two launches when a group has both
    if (vpf_debug().wgroup_dma > 0 && ws == nullptr) {
        VpfWgradJob a[GEMM_GROUP_MAX], b[GEMM_GROUP_MAX];
        int na = 0, nb = 0;
        for (int i = 0; i < njobs; ++i) { if (wgrad_dma_conforms(jobs[i])) a[na++] = jobs[i]; else b[nb++] = jobs[i]; }
        if (na && nb) {
            const int rc = wgrad_group_launch(a, na, nullptr, 0, stream, true);
            return rc != VPF_OK ? rc : wgrad_group_launch(b, nb, nullptr, 0, stream, false);
        }
        return wgrad_group_launch(jobs, njobs, ws, ws_bytes, stream, na > 0);
    }
    return wgrad_group_launch(jobs, njobs, ws, ws_bytes, stream, false);
}
static int wgrad_group_launch(const VpfWgradJob* jobs, int njobs, void* ws, long ws_bytes, void* stream, bool dma_ok, int ldc_override)
{
    GemmGroup grp = {};
    grp.n = njobs;
    const int partial = ws != nullptr;
    // round 5: the LDS-DMA kernel (256 x 128 tiles, one workgroup per CU) when every problem conforms -- multiples of its tile, 64-token
    // stages, the atomics flush -- and the launch is big enough to stream (VPF_WGROUP_DMA: 0 = never, N > 0 = from N tokens per problem on)
    const bool dma = dma_ok && !partial;
    const int cfg = dma ? 2 : vpf_debug().wgroup_cfg;
    const int target = vpf_debug().wgroup_wgs > 0 ? vpf_debug().wgroup_wgs : (dma ? 256 : 512);      // measured on the c2 step: 128x128 tiles, ~512 workgroups per group
    // DMA kernel: 256 x 256 tiles when every problem's K_in is a multiple of 256 (VPF_WGROUP_DMA_TN = 128 / 256 forces one), else 256 x 128
    int dma_tn = WGDMA_BN, dma_tm = WGDMA_BM;
    if (dma) {
        bool all256 = true, rows256 = true;
        for (int i = 0; i < njobs; ++i) { if (jobs[i].K % 256) all256 = false; if (jobs[i].N % 256) rows256 = false; }
        const int want = vpf_debug().wgroup_dma_tn;
        dma_tm = rows256 ? 256 : 128;                      // 128 x 128 tiles for a group with an N_out that is a multiple of 128 only (D = 384)
        dma_tn = (all256 && rows256 && want != 128) ? 256 : 128;
    }
    const int tm = dma ? dma_tm : (cfg == 0 ? 64 : 128), tn = dma ? dma_tn : ((cfg == 3 || cfg == 4) ? 256 : ((cfg == 2 || cfg >= 5) ? 128 : 64));
    const bool pair = cfg == 8 && !partial && !dma;          // two K slices per 8-wave workgroup: half the flush atomics (round 3)
    long total_tiles = 0;
    for (int i = 0; i < njobs; ++i) total_tiles += (long)vpf_cdiv(jobs[i].N, tm) * vpf_cdiv(jobs[i].K, tn);
    int at = 0;
    for (int i = 0; i < njobs; ++i) {
        const VpfWgradJob& j = jobs[i];
        if (!j.dy || !j.x || !j.dW) return VPF_ERR_NULL;
        if (j.M <= 0 || j.N <= 0 || j.K <= 0) return VPF_ERR_BADSHAPE;
        if ((j.N % 8) || (j.K % 8) || ((uintptr_t)j.dy & 15) || ((uintptr_t)j.x & 15)) return VPF_ERR_BADALIGN;
        WgDesc& g = grp.d[i];
        // C[m = n_out, n = k_in] += sum over tokens: A = dY read k-strided (rows = N), B = X read k-strided (rows = K)
        g.A = (const h16_t*)j.dy; g.B = (const h16_t*)j.x; g.lda = j.N; g.ldb = j.K;
        g.M = j.N; g.N = j.K; g.K = j.M; g.mode = EPI_ATOMIC; g.C = j.dW; g.ldc = ldc_override > 0 ? ldc_override : j.K; g.dbias = j.dbias;
        const int nx = vpf_cdiv(g.N, tn), ny = vpf_cdiv(g.M, tm);
        // ~`target` workgroups over the whole group, every K slice at least 256 tokens deep
        long sp = target / (total_tiles > 0 ? total_tiles : 1);
        const long maxs = vpf_cdiv(g.K, dma ? 1024 : 256);      // (DMA kernel: a slice flushes a 128 KB tile -- at least 16 stages of work in front of it)
        if (sp > maxs) sp = maxs;
        if (sp < 2 && !dma) sp = 2;               // gemm_tile reads splitk > 1 as "blockIdx.z is a K slice"
        if (sp < 1) sp = 1;
        if (sp > 8 && vpf_debug().wgroup_xlist) sp &= ~7L;      // many slices: a multiple of 8, so that the kernel's XCD ordering applies (a slice's tiles on one XCD)
        if (pair) sp &= ~1L;                      // whole pairs
        g.splitk = (int)sp;
        grp.nx[i] = (short)nx; grp.ny[i] = (short)ny; grp.start[i] = at;
        at += nx * ny * (int)(pair ? sp / 2 : sp);
    }
    grp.start[njobs] = at;
    {   // XCD lists: when no problem's slices can use the sk % 8 == 0 ordering of the kernel and the slabs fit the table
        int nslab = 0; bool ok = vpf_debug().wgroup_xlist != 0;
        for (int i = 0; i < njobs && ok; ++i) {
            const int sk = pair ? grp.d[i].splitk / 2 : grp.d[i].splitk;
            if ((sk & 7) == 0) ok = false;
            if (grp.d[i].K / (sk > 0 ? sk : 1) < 4096) ok = false;      // short slabs (config 4: 2 048 tokens per slice) gain nothing: 4.40 vs 4.43 ms; c2 (6 144): 4.20 -> 4.10
            nslab += sk;
        }
        if (ok && nslab <= WGROUP_XLIST_MAX && njobs > 1) {
            // greedy: largest slabs first, each to the XCD with the fewest tiles so far
            int order[WGROUP_XLIST_MAX], sp_[WGROUP_XLIST_MAX], sz_[WGROUP_XLIST_MAX], n = 0;
            for (int i = 0; i < njobs; ++i) {
                const int sk = pair ? grp.d[i].splitk / 2 : grp.d[i].splitk;
                for (int z = 0; z < sk; ++z) { sp_[n] = i; sz_[n] = z; order[n] = n; ++n; }
            }
            auto tiles = [&](int k) { return (int)grp.nx[sp_[k]] * (int)grp.ny[sp_[k]]; };
            for (int a = 1; a < n; ++a) {                   // insertion sort by tile count, descending (stable)
                const int k = order[a]; int b = a - 1;
                while (b >= 0 && tiles(order[b]) < tiles(k)) { order[b + 1] = order[b]; --b; }
                order[b + 1] = k;
            }
            int load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, owner[WGROUP_XLIST_MAX];
            for (int a = 0; a < n; ++a) {
                int best = 0;
                for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
                owner[order[a]] = best; load[best] += tiles(order[a]);
            }
            int w = 0, longest = 0;
            for (int x = 0; x < 8; ++x) {
                grp.xoff[x] = (unsigned char)w;
                for (int k = 0; k < n; ++k) if (owner[k] == x) { grp.xp[w] = (unsigned char)sp_[k]; grp.xz[w] = (unsigned char)sz_[k]; ++w; }
                if (load[x] > longest) longest = load[x];
            }
            grp.xoff[8] = (unsigned char)w;
            grp.xmode = 1;
            at = 8 * longest;
        }
    }
    // K slices of alternating length (4/3 and 2/3 of the mean): the two workgroups a CU holds then leave their staging loops at
    // different times, and one's atomic flush (L2 atomic units) overlaps the other's staging (L2 read bandwidth) instead of all
    // 512 workgroups flushing together at the end (-0.035 ms/step, 15 launches)
    grp.uneven = dma ? vpf_debug().wgroup_dma_ramp : vpf_debug().wgroup_uneven;      // (DMA kernel: a ramp of slice lengths, in percent, for launches of >= 16 slices)
    grp.dbg = vpf_debug().wgroup_dbg;
    hipStream_t st = (hipStream_t)stream;
    if (dma) return launch_wgrad_dma(grp, at, tm, tn, st);
    // workspace split-K (EPI_PARTIAL) when the caller handed over enough scratch: [counters | one tm x tn f32 tile per workgroup]
    if (partial && ws && total_tiles <= WGROUP_CNT_INTS && ws_bytes >= (long)(WGROUP_CNT_INTS * 4 + (size_t)at * tm * tn * 4) && !((uintptr_t)ws & 15)) {
        float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + WGROUP_CNT_INTS * 4);
        int* cnt = reinterpret_cast<int*>(ws);
        for (int i = 0; i < njobs; ++i) {
            WgDesc& g = grp.d[i];
            g.mode = EPI_PARTIAL; g.part = part + (size_t)grp.start[i] * tm * tn; g.cnt = cnt; g.ntx = grp.nx[i];
            cnt += grp.nx[i] * grp.ny[i];
        }
    }
    // (measured and removed: 256 x 256 tiles with the accumulators in all 512 registers of a lane -- half the operand bytes staged
    //  per output element, but 154 us against 59: 67 MB of flush atomics and one wave per SIMD with nothing to hide behind)
    if (pair) return launch_wgrad_group<2, 2, 2, 2, 64, 1, true>(grp, at, st); // 128x128, two K slices per 8-wave workgroup
    if (cfg == 7) return launch_wgrad_group<2, 2, 2, 2, 64, 3>(grp, at, st);  // 3 stages of loads in flight
    if (cfg == 6) return launch_wgrad_group<2, 2, 2, 2, 64, 2>(grp, at, st);  // 2 stages of loads in flight
    if (cfg == 5) return launch_wgrad_group<2, 2, 2, 2, 32>(grp, at, st);     // 128x128, shallow stages: 4 workgroups per CU
    if (cfg == 4) return launch_wgrad_group<2, 4, 2, 2, 64>(grp, at, st);     // wave tile 64x128: half the LDS fragment reads per MFMA
    if (cfg == 3) return launch_wgrad_group<2, 4, 2, 2, 32>(grp, at, st);
    if (cfg == 2) return launch_wgrad_group<2, 2, 2, 2, 64>(grp, at, st);
    if (cfg == 1) return launch_wgrad_group<1, 2, 4, 1, 64>(grp, at, st);
    return launch_wgrad_group<1, 1, 2, 2, 128>(grp, at, st);
}

// one conforming weight gradient whose dW is a column block of a wider matrix (row pitch ldc): the group launcher with one job
static int wgrad_single_dma(const VpfWgradJob& j, int ldc, hipStream_t st)
{
    return wgrad_group_launch(&j, 1, nullptr, 0, st, true, ldc);
}
