// sa_rows.hip -- the encoder's row-block kernels, round 3.
//
// Same arithmetic, in the same order, as sa_layer_fwd_kernel<.., false, 1> / sa_bwd_mlp_rows_kernel / sa_bwd_qkv_rows_kernel of
// sa_layer.hip (SelfAttentionLayer / CrossAttentionLayer tail, vipformer/model/pointcloud/partseg.py:144-213, and its dgrad chain):
// all products are "swapped" (C^T[channel, token] = W . X^T), a wave owns 32 channels, a lane owns one token (lane & 31) and its
// registers run over channels.  Results are bit-identical to those kernels (tests/test_kernels_gpu.py).
//
// What is different.  Those kernels keep the f32 residual rows of a 64-token block in a 64 KB LDS tile.  Here the residual stream
// lives in REGISTERS between the points where it crosses HBM, and crosses in whole 128-byte row segments through a wave-PRIVATE
// 4 KB LDS slice (rows <-> accumulator layout, XOR-swizzled: conflict-free on both sides, tools/lds_banks.py) -- no barrier, the LDS
// operations of one wave execute in order.  Addresses are workgroup-uniform bases + 32-bit lane offsets recomputed per phase
// (fresh_tid), weights stream through a register ring; nothing spills at 128 registers.
//
// Geometry <RB, TH>: a wave owns RB blocks of 32 tokens x 32 channels; TH groups of D / 32 waves split the workgroup's tokens
// (TOK = 32 RB TH).
//   <2, 1>  8 waves, 64 tokens, 78 KB of LDS, 127 VGPRs: two workgroups fit a CU.  Measured: no gain -- a
//           grid of 192 - 196 workgroups never puts two on one CU, and where the two branches' kernels do meet they run in lockstep.
//   <1, 2>  (default at D = 256) the SAME 64 tokens on 16 waves.  In-kernel stamps (tools/microbench.py satail) show a workgroup's
//           80 k cycles as 37 k of VALU epilogues bound by what ONE wave can issue (two waves per SIMD), 27 k for the eight MFMA
//           units (weight-latency bound at a 4-deep ring) and 14 k of exposed memory waits: so the second workgroup's waves are
//           brought INSIDE the workgroup -- four waves per SIMD, half the epilogue work per wave, the residual rows stay in
//           registers across the MLP (no x1 re-read), 8-deep weight ring, slices in a region of their own (144 KB of LDS).
//   <1, 1>  D = 384 (BASELINE config 4: 6 heads, hidden 1536): 12 waves, 32-token blocks.
#include "sa_rows.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

template <int D, int RB, int TH>
struct Cfg {
    static constexpr int NWV = D / 32, NW = NWV * TH, NT = 64 * NW, TOK = 32 * RB * TH, ALD = D + 8, KS = D / 16, C8 = D / 8;
    static constexpr int TILE = TOK * ALD;                               // h16 elements of one operand tile
    static constexpr bool XDED = (NW * 4096 > TILE * 2);                 // the transposition slices need a region of their own
    static constexpr int PD = RB == 1 ? 8 : 4;                           // weight ring depth (k-steps in flight per wave)
    static constexpr bool KEEPX = RB == 1;                               // the residual rows stay in registers across the MLP
};

// The thread index through an opaque move.  Every helper derives its lane / token / address values from a copy of its own, so the
// compiler cannot keep the ~50 address registers of all phases alive from the top of the kernel to their last use (common
// subexpression elimination across phases + no rematerialisation = spills at a 128-register budget); a handful of VALU operations
// per phase recompute them instead.
__device__ __forceinline__ int fresh_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
// which channel block (cw) and which first 32-token block (tb0) the wave owns -- scalar registers
template <int D, int RB>
__device__ __forceinline__ void who(int tid_, int& cw, int& tb0)
{
    const int wv = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    cw = wv % (D / 32);
    tb0 = (wv / (D / 32)) * RB;
}

// ------------------------------------------------------------------ 16-byte row stores with a cache policy (round 5)
// ST = 0 plain; 1 `sc1` (write-through: the line does not stay dirty in the XCD's L2, so the end-of-kernel write-back of the
// launch's last ~30 MB of outputs is spread over the kernel instead -- MI355X_MICROARCH.md, "stores of each flavour"); 2 `nt`;
// 3 `sc0 sc1`.  The inline-asm stores are invisible to the compiler's vmcnt bookkeeping, which is safe: vmcnt retires in order, so
// extra operations in the queue only make a counted wait cover more than the compiler asked for.  The trailing s_nop is the wait
// state the ISA asks for between a store of more than 8 bytes and a write to its data registers (the compiler inserts it for its
// own stores; it does not see these).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
template <int ST, class V>
__device__ __forceinline__ void st16(V* p, const V& v)
{
    static_assert(sizeof(V) == 16, "16-byte stores");
    if constexpr (ST == 0) *p = v;
    else {
        const u32x4_t d = __builtin_bit_cast(u32x4_t, v);
        if constexpr (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(p), "v"(d) : "memory");
        else if constexpr (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" :: "v"(p), "v"(d) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 0" :: "v"(p), "v"(d) : "memory");
    }
}

// ------------------------------------------------------------------ wave groups (round 5)
// The D / 32 waves that share the token blocks [tb0, tb0 + RB) are a GROUP; a workgroup has TH of them.  Nothing but the per-channel
// parameter vectors is shared between groups: operand-tile rows, LayerNorm exchange entries and slices are per token / per wave.
//   DEC = false  the groups run in lockstep: workgroup barriers, whole-tile cooperative row passes (rounds 3 - 4).
//   DEC = true   every group is a pipeline of its own: its barriers are a counter in LDS that only ITS waves arrive at and poll
//                (gfx950 has one hardware barrier per workgroup and no named barriers), its row passes cover its own rows.  The two
//                groups of a 16-wave workgroup then drift apart like two workgroups sharing a CU -- one group's MFMA units run under
//                the other's epilogues and stores -- at the co-residency a 192-workgroup grid never gets from the dispatcher.
// The counter protocol: a wave drains its LDS operations (s_waitcnt lgkmcnt(0): its tile / exchange writes have landed), lane 0 adds 1,
// every wave polls until the count reaches NWV x (barriers so far).  LDS operations of a CU execute in issue order, so a wave that
// has seen the count sees the data written before the adds.  Outstanding global loads (the weight ring) stay in flight.
template <class C, bool DEC>
struct Grp {
    static constexpr int NT = DEC ? 64 * C::NWV : C::NT;               // threads that share a row pass
    static constexpr int TOK = DEC ? C::TOK / (C::NW / C::NWV) : C::TOK;   // rows of a row pass
    unsigned* ctr;      // DEC: this group's arrival counter (LDS, zeroed before the workgroup's first barrier)
    unsigned want;      // arrivals expected once the next barrier is complete
    int th;             // group index (scalar)
    __device__ __forceinline__ int tid() const { return DEC ? fresh_tid() - th * NT : fresh_tid(); }
    __device__ __forceinline__ int row0() const { return DEC ? th * TOK : 0; }
    __device__ __forceinline__ void sync()
    {
        if constexpr (!DEC) { __syncthreads(); }
        else {
            want += C::NWV;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((fresh_tid() & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (;;) {
                const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if ((int)(v - want) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        }
    }
};

template <int PD> struct WRing { uint4 w[PD]; };

template <int PD>
__device__ __forceinline__ void ring_fill(const h16_t* __restrict__ Wp, int ksn, int ks0, int cb, WRing<PD>& r)
{
    const unsigned lane = fresh_tid() & 63;
    const uint4* w0 = reinterpret_cast<const uint4*>(Wp) + ((size_t)cb * ksn + ks0) * 64;      // wave-uniform (cb is): scalar base + lane
#pragma unroll
    for (int kk = 0; kk < PD; ++kk) r.w[kk] = w0[kk * 64 + lane];
}
// acc[i] += Wfrag(cb, ks0 + ks) . X[(tb0 + i)*32.., ks*16..] for ks = 0 .. KSU-1; the ring holds k-steps 0 .. PD - 1 on entry (ring_fill).
// Explicit software pipeline: the activation fragments of k-step ks + 1 and the weight fragment of k-step ks + PD are requested in
// front of the MFMAs of k-step ks, and a scheduling fence per k-step keeps the compiler from hoisting all RB KSU LDS reads of the
// unrolled loop to the top (that is what it does on its own: 128 registers of fragments, spills everywhere else).
template <int RB, int KSU, int PD>
__device__ __forceinline__ void gemm_unit(const h16_t* __restrict__ Wp, int ksn, int ks0, int cb, const h16_t* act, int ald, int tb0,
                                          f32x16_t (&acc)[RB], WRing<PD>& r)
{
    constexpr int XD = 2;                                  // activation fragments: k-steps in flight
    const unsigned lane = fresh_tid() & 63;
    const uint4* w0 = reinterpret_cast<const uint4*>(Wp) + ((size_t)cb * ksn + ks0) * 64;
    const h16_t* xrow = act + (tb0 * 32 + (lane & 31)) * ald + 8 * (lane >> 5);
    uint4 xb[XD][RB];
#pragma unroll
    for (int p = 0; p < XD - 1; ++p)
#pragma unroll
        for (int i = 0; i < RB; ++i) xb[p][i] = *reinterpret_cast<const uint4*>(xrow + i * 32 * ald + p * 16);
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) {
        if (ks + XD - 1 < KSU) {
#pragma unroll
            for (int i = 0; i < RB; ++i) xb[(ks + XD - 1) % XD][i] = *reinterpret_cast<const uint4*>(xrow + i * 32 * ald + (ks + XD - 1) * 16);
        }
        const h16x8_t af = __builtin_bit_cast(h16x8_t, r.w[ks % PD]);
        if (ks + PD < KSU) r.w[ks % PD] = w0[(ks + PD) * 64 + lane];
#pragma unroll
        for (int i = 0; i < RB; ++i)
            acc[i] = vpf_mfma32(af, __builtin_bit_cast(h16x8_t, xb[ks % XD][i]), acc[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int RB>
__device__ __forceinline__ void zero(f32x16_t (&acc)[RB])
{
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
}

// ------------------------------------------------------------------ wave-private transposition slices
// f32: [32 tokens][8 chunks of 16 B] = 4 KB; logical chunk c of token t sits at physical chunk c ^ swz(t).
__device__ __forceinline__ int swz(int t) { return ((t >> 1) & 7) ^ ((t & 1) << 2); }

// Row side: lane -> (token 8 it + (lane >> 3), chunk lane & 7); a wave-instruction moves 8 tokens x 128 B.
// Addresses are a workgroup-uniform base (scalar registers) + a 32-bit element offset per lane: one VGPR per row instead of a 64-bit
// pointer, and the offsets are shared by every matrix that is walked in the same pattern (base, x1, out, d, dx1 ...).
struct RowOff { unsigned o[4]; unsigned ok; };   // element offset of this lane's 16 bytes in rows 8 it + (lane >> 3) of token block tb; bit it of ok: the row exists
template <int D>
__device__ __forceinline__ RowOff row_offsets(int tb, int cw, int nvalid)
{
    const unsigned lane = fresh_tid() & 63, tl = lane >> 3, ch = lane & 7;
    RowOff r;
    r.ok = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const unsigned row = tb * 32 + 8 * it + tl;
        r.o[it] = row * D + 32 * cw + 4 * ch;
        r.ok |= ((int)row < nvalid ? 1u : 0u) << it;
    }
    return r;
}
// row offsets into the positional term: row m uses pos[m % pos_rows]
template <int D>
__device__ __forceinline__ RowOff pos_offsets(int tb, int cw, long m0, int pos_rows, const RowOff& ro, bool have)
{
    const unsigned lane = fresh_tid() & 63;
    RowOff po;
    // one modulo per call on the (uniform) block start; per row an add and a conditional subtraction when the table has at least a
    // block's rows (every shipped configuration), a 32-bit modulo otherwise
    const uint32_t base = (uint32_t)((m0 + tb * 32) % (long)pos_rows), pr = (uint32_t)pos_rows;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        uint32_t r = base + 8 * it + (lane >> 3);
        r = pr >= 32u ? (r >= pr ? r - pr : r) : r % pr;
        po.o[it] = r * D + 32 * cw + 4 * (lane & 7);
    }
    po.ok = have ? ro.ok : 0u;
    return po;
}
__device__ __forceinline__ void rows_load(const float* __restrict__ ubase, const RowOff& ro, float4 (&r)[4])
{
#pragma unroll
    for (int it = 0; it < 4; ++it)
        r[it] = ((ro.ok >> it) & 1u) ? *reinterpret_cast<const float4*>(ubase + ro.o[it]) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void rows_to_slice(float* slice, const float4 (&r)[4])
{
    const int lane = fresh_tid() & 63, tl = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int tok = 8 * it + tl;
        *reinterpret_cast<float4*>(slice + tok * 32 + ((ch ^ swz(tok)) << 2)) = r[it];
    }
}
// slice rows -> HBM
template <int ST = 0>
__device__ __forceinline__ void slice_store_rows(const float* slice, float* __restrict__ ubase, const RowOff& ro)
{
    const int lane = fresh_tid() & 63, tl = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int tok = 8 * it + tl;
        const float4 v = *reinterpret_cast<const float4*>(slice + tok * 32 + ((ch ^ swz(tok)) << 2));
        if ((ro.ok >> it) & 1u) st16<ST>(reinterpret_cast<float4*>(ubase + ro.o[it]), v);
    }
}
// accumulator side: lane (t = lane & 31, hl = lane >> 5) owns channels 8 g + 4 hl + q of token t = registers 4 g + q
template <bool ADD>
__device__ __forceinline__ void slice_to_acc(const float* slice, f32x16_t& v)
{
    const int lane = fresh_tid() & 63, t = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 x = *reinterpret_cast<const float4*>(slice + t * 32 + (((2 * g + hl) ^ swz(t)) << 2));
        if (ADD) { v[4 * g + 0] += x.x; v[4 * g + 1] += x.y; v[4 * g + 2] += x.z; v[4 * g + 3] += x.w; }
        else { v[4 * g + 0] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w; }
    }
}
__device__ __forceinline__ void acc_to_slice(float* slice, const f32x16_t& v)
{
    const int lane = fresh_tid() & 63, t = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(slice + t * 32 + (((2 * g + hl) ^ swz(t)) << 2)) = make_float4(v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
// h16: [32 RB tokens][4 chunks of 16 B] (the wave's 32 channels of one GEMM result): accumulator side writes 8 B, row side moves
// 16 tokens x 64 B per wave-instruction
__device__ __forceinline__ int swz2(int t) { return (t >> 1) & 3; }
template <int RB>
__device__ __forceinline__ void acc_to_slice_h16(h16_t* slice, const f32x16_t (&v)[RB])
{
    const int lane = fresh_tid() & 63, t = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_h16x2(v[i][4 * g + 0], v[i][4 * g + 1]);
            w.y = pack_h16x2(v[i][4 * g + 2], v[i][4 * g + 3]);
            const int tok = i * 32 + t;
            *reinterpret_cast<uint2*>(slice + tok * 32 + ((g ^ swz2(tok)) << 3) + 4 * hl) = w;
        }
}
// rows of the slice -> Gu[tok * ld + 8 chunk ..] (Gu = the matrix at the wave's first token and first column) for tok < nv
template <int RB, int ST = 0>
__device__ __forceinline__ void slice_h16_store_rows(const h16_t* slice, h16_t* __restrict__ Gu, int ld, int nv)
{
    const int lane = fresh_tid() & 63, tl = lane >> 2, ch = lane & 3;
#pragma unroll
    for (int it = 0; it < 2 * RB; ++it) {
        const int tok = 16 * it + tl;
        const uint4 v = *reinterpret_cast<const uint4*>(slice + tok * 32 + ((ch ^ swz2(tok)) << 3));
        if (tok < nv) st16<ST>(reinterpret_cast<uint4*>(Gu + (unsigned)(tok * ld + ch * 8)), v);
    }
}

// ------------------------------------------------------------------ operand tiles
// accumulator tile -> h16 operand tile [tokens][ALD] (this wave's 32 columns of its token blocks)
template <int D, int RB>
__device__ __forceinline__ void acc_to_tile(const f32x16_t (&acc)[RB], h16_t* sAct, int ald)
{
    const int tid_ = fresh_tid();
    int cw, tb0;
    who<D, RB>(tid_, cw, tb0);
    const int lane = tid_ & 63, hl = lane >> 5, t = lane & 31;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 u;
            u.x = pack_h16x2(acc[i][4 * g + 0], acc[i][4 * g + 1]);
            u.y = pack_h16x2(acc[i][4 * g + 2], acc[i][4 * g + 3]);
            *reinterpret_cast<uint2*>(sAct + ((tb0 + i) * 32 + t) * ald + 32 * cw + 8 * g + 4 * hl) = u;
        }
}
// u = h16(acc + bias) -> operand tile in the accumulator layout
template <int D, int RB>
__device__ __forceinline__ void bias_to_tile(const f32x16_t (&acc)[RB], const float* bias, h16_t* sAct, int ald)
{
    const int tid_ = fresh_tid();
    int cw, tb0;
    who<D, RB>(tid_, cw, tb0);
    const int lane = tid_ & 63, hl = lane >> 5, t = lane & 31;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int cl = 32 * cw + 8 * g + 4 * hl;
        const float4 b1 = *reinterpret_cast<const float4*>(bias + cl);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            uint2 w;
            w.x = pack_h16x2(acc[i][4 * g + 0] + b1.x, acc[i][4 * g + 1] + b1.y);
            w.y = pack_h16x2(acc[i][4 * g + 2] + b1.z, acc[i][4 * g + 3] + b1.w);
            *reinterpret_cast<uint2*>(sAct + ((tb0 + i) * 32 + t) * ald + cl) = w;
        }
    }
}
// the whole tile, row-coalesced: every thread moves 16 B of a row
template <class C>
__device__ __forceinline__ void tile_store_rows(const h16_t* sAct, h16_t* __restrict__ Gu /* row 0, column 0 of the block */, int ld, int nvalid)
{
    const int tid_ = fresh_tid();
#pragma unroll
    for (int it = 0; it < C::TOK * C::C8 / C::NT; ++it) {
        const int e = tid_ + it * C::NT, row = e / C::C8, ch = e - row * C::C8;
        const uint4 v = *reinterpret_cast<const uint4*>(sAct + row * C::ALD + ch * 8);
        if (row < nvalid) *reinterpret_cast<uint4*>(Gu + (unsigned)(row * ld + ch * 8)) = v;
    }
}
// the rows of one group (Grp: all rows when the groups run in lockstep)
template <class C, int ST = 0, class G>
__device__ __forceinline__ void tile_store_rows(const G& gr, const h16_t* sAct, h16_t* __restrict__ Gu, int ld, int nvalid)
{
    const int tid_ = gr.tid(), r0 = gr.row0();
#pragma unroll
    for (int it = 0; it < G::TOK * C::C8 / G::NT; ++it) {
        const int e = tid_ + it * G::NT, row = r0 + e / C::C8, ch = e % C::C8;
        const uint4 v = *reinterpret_cast<const uint4*>(sAct + row * C::ALD + ch * 8);
        if (row < nvalid) st16<ST>(reinterpret_cast<uint4*>(Gu + (unsigned)(row * ld + ch * 8)), v);
    }
}
template <class C>
__device__ __forceinline__ void tile_load_rows(h16_t* sAct, const h16_t* __restrict__ Gu, int ld, int nvalid)
{
    const int tid_ = fresh_tid();
    uint4 r[C::TOK * C::C8 / C::NT];
#pragma unroll
    for (int it = 0; it < C::TOK * C::C8 / C::NT; ++it) {
        const int e = tid_ + it * C::NT, row = e / C::C8, ch = e - row * C::C8;
        r[it] = row < nvalid ? *reinterpret_cast<const uint4*>(Gu + (unsigned)(row * ld + ch * 8)) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < C::TOK * C::C8 / C::NT; ++it) {
        const int e = tid_ + it * C::NT, row = e / C::C8, ch = e - row * C::C8;
        *reinterpret_cast<uint4*>(sAct + row * C::ALD + ch * 8) = r[it];
    }
}
// row pass over the hidden chunk in sAct: u -> HBM, h = gelu(u) in place and -> HBM (16 B per lane, whole rows per wave-instruction)
template <class C>
__device__ __forceinline__ void gelu_rows_pass(h16_t* sAct, h16_t* __restrict__ u_u, h16_t* __restrict__ h_u, int ldh, int nvalid)
{
    const int tid_ = fresh_tid();
#pragma unroll
    for (int it = 0; it < C::TOK * C::C8 / C::NT; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        const int e = tid_ + it * C::NT, row = e / C::C8, ch = e - row * C::C8;
        const uint4 v = *reinterpret_cast<const uint4*>(sAct + row * C::ALD + ch * 8);
        const unsigned go = row * ldh + ch * 8;
        if (row < nvalid) *reinterpret_cast<uint4*>(u_u + go) = v;
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
        uint32_t hh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            hh[q] = pack_h16x2(vpf_gelu(h16_lo(vv[q])), vpf_gelu(h16_hi(vv[q])));
        const uint4 hv = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        *reinterpret_cast<uint4*>(sAct + row * C::ALD + ch * 8) = hv;
        if (row < nvalid) *reinterpret_cast<uint4*>(h_u + go) = hv;
    }
}
template <class C, int ST = 0, class G>
__device__ __forceinline__ void gelu_rows_pass(const G& gr, h16_t* sAct, h16_t* __restrict__ u_u, h16_t* __restrict__ h_u, int ldh, int nvalid)
{
    const int tid_ = gr.tid(), r0 = gr.row0();
#pragma unroll
    for (int it = 0; it < G::TOK * C::C8 / G::NT; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        const int e = tid_ + it * G::NT, row = r0 + e / C::C8, ch = e % C::C8;
        const uint4 v = *reinterpret_cast<const uint4*>(sAct + row * C::ALD + ch * 8);
        const unsigned go = row * ldh + ch * 8;
        if (row < nvalid) st16<ST>(reinterpret_cast<uint4*>(u_u + go), v);
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
        uint32_t hh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            hh[q] = pack_h16x2(vpf_gelu(h16_lo(vv[q])), vpf_gelu(h16_hi(vv[q])));
        const uint4 hv = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        *reinterpret_cast<uint4*>(sAct + row * C::ALD + ch * 8) = hv;
        if (row < nvalid) st16<ST>(reinterpret_cast<uint4*>(h_u + go), hv);
    }
}
// per-channel vectors -> sPar in pieces of 128 floats (128 divides 256, 384, 512 and 1536: a piece never straddles two vectors), one
// wave per piece with a wave-uniform source (scalar selects, no lane branches); half the lanes carry a float4 each
template <int D, int HID, int NW>
__device__ __forceinline__ void stage_params(float* sPar, const float* bo, const float* g2, const float* be2, const float* b2, const float* g1n,
                                             const float* be1n, const float* b1)
{
    static_assert(D % 128 == 0 && HID % 128 == 0, "pieces of 128 floats");
    const int tid_ = fresh_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    for (int pc = wave; pc < (6 * D + HID) / 128; pc += NW) {
        const int f = pc * 128;
        const float* src = f < D ? bo : f < 2 * D ? g2 : f < 3 * D ? be2 : f < 4 * D ? b2 : f < 5 * D ? g1n : f < 6 * D ? be1n : b1;
        const int o = (f < 6 * D ? f % D : f - 6 * D) + lane * 4;
        if (lane < 32) *reinterpret_cast<float4*>(sPar + f + lane * 4) = src ? *reinterpret_cast<const float4*>(src + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------ LayerNorm over the channel axis (forward), in place on v
// Every wave reduces ITS 32 channels of a token to (mean, centred second moment); the pairs of the D / 32 waves that share the token
// are merged exactly (Chan et al.) through one LDS exchange and one barrier.  Identical arithmetic to sa_layernorm of sa_layer.hip.
// The statistics go to HBM (gm, gr: the block's first token) from the wave that owns channel block 0.
template <int D, int RB, class SYNC>
__device__ __forceinline__ void layernorm(f32x16_t (&v)[RB], const float* gamma, const float* beta, float2* sPair, float* __restrict__ gm,
                                          float* __restrict__ gr, int nvalid, SYNC&& sync)
{
    constexpr int NWV = D / 32;
    const int tid_ = fresh_tid();
    int cw, tb0;
    who<D, RB>(tid_, cw, tb0);
    const int lane = tid_ & 63, hl = lane >> 5;
    float mean[RB], rstd[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[i][r];
        s += __shfl_xor(s, 32, 64);
        const float mw = s * (1.0f / 32);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = v[i][r] - mw; q += d * d; }
        q += __shfl_xor(q, 32, 64);
        if (lane < 32) sPair[((tb0 + i) * 32 + lane) * NWV + cw] = make_float2(mw, q);
    }
    sync();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float2 pw[NWV];
        float ms = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { pw[w] = sPair[((tb0 + i) * 32 + (lane & 31)) * NWV + w]; ms += pw[w].x; }
        const float mu = ms * (1.0f / NWV);
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { const float d = pw[w].x - mu; m2 += pw[w].y + 32.0f * d * d; }
        mean[i] = mu;
        rstd[i] = rsqrtf(m2 * (1.0f / D) + 1e-5f);
        const int tok = (tb0 + i) * 32 + lane;
        if (cw == 0 && lane < 32 && tok < nvalid) { gm[tok] = mu; gr[tok] = rstd[i]; }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = 32 * cw + 8 * g + 4 * hl;
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
        const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[i][4 * g + q] = (v[i][4 * g + q] - mean[i]) * rstd[i] * gg[q] + bb[q];
    }
}

template <int D, int RB>
__device__ __forceinline__ void layernorm(f32x16_t (&v)[RB], const float* gamma, const float* beta, float2* sPair, float* __restrict__ gm,
                                          float* __restrict__ gr, int nvalid)
{
    layernorm<D, RB>(v, gamma, beta, sPair, gm, gr, nvalid, [] { __syncthreads(); });
}

// v += dropout(acc + bias) for the 32-token block tb: the Residual epilogue (partseg.py:201-213) on the accumulator tile; the mask is
// a function of the element's offset in the [M, D] matrix, exactly as every other kernel of the library draws it
template <int D>
__device__ __forceinline__ void residual_epilogue(f32x16_t& v, const f32x16_t& acc, int tb, int cw, const float* bias, const VpfRng& rng, bool drop,
                                                  long m0, int nvalid)
{
    const int lane = fresh_tid() & 63, hl = lane >> 5, t = lane & 31;
    const int tok = tb * 32 + t;
    const size_t rowoff = (size_t)(m0 + (tok < nvalid ? tok : 0)) * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from interleaving all the unrolled hash chains (spills)
        const int cch = 32 * cw + 8 * g + 4 * hl;
        const float4 b4 = *reinterpret_cast<const float4*>(bias + cch);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
        const uint32_t keep = drop ? vpf_keep4(rng, (uint64_t)(rowoff + cch) >> 2) : 15u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y = acc[4 * g + q] + bb[q];
            if (drop) y = ((keep >> q) & 1u) ? y * rng.scale : 0.f;
            v[4 * g + q] += y;
        }
    }
}

// ================================================================================================ forward
template <int D, int HID, int RB, int TH, int MINW, bool FULL, bool DBG = false, bool DEC = false, int ST = 0>
__global__ void __launch_bounds__(2 * D * TH, MINW) sa_rows_fwd_kernel(VpfSaLayerFwd a, int tpw, int stagger)
{
    using C = Cfg<D, RB, TH>;
    using G = Grp<C, DEC>;
    constexpr int NWV = C::NWV, TOK = C::TOK, ALD = C::ALD, KS = C::KS, HC = HID / D, PD = C::PD;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* actA = lds;                                               // [TOK][ALD]  o -> n2 -> (slices) out, q|k|v staging
    h16_t* actH = lds + C::TILE;                                     // [TOK][ALD]  (slices) base, x1, pos -> hidden chunk -> next n1
    float2* sPair = reinterpret_cast<float2*>(actH + C::TILE);        // [TOK][NWV]  LayerNorm exchange
    float* sPar = reinterpret_cast<float*>(sPair + TOK * NWV);        // bo | ln2 g | ln2 b | b2 | next ln1 g | b | b1[HID]
    float* xded = sPar + 6 * D + HID;                                 // XDED: the slices' own region
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % NWV, tb0 = (wave / NWV) * RB;               // this wave: channels [32 cw, +32) of token blocks tb0 .. tb0 + RB - 1
    G grp;
    grp.th = wave / NWV; grp.want = 0;
    grp.ctr = reinterpret_cast<unsigned*>(xded + (C::XDED ? C::NW * 1024 : 0)) + grp.th;      // DEC: one arrival counter per group behind everything else
    if (DEC && threadIdx.x < TH) reinterpret_cast<unsigned*>(xded + (C::XDED ? C::NW * 1024 : 0))[threadIdx.x] = 0u;
    // tpw = tokens per workgroup (<= TOK): a grid of ~2 workgroups per CU is cut to fit the machine (sa_rows_tokens_per_wg)
    const long m0 = FULL ? (long)blockIdx.x * TOK : (long)blockIdx.x * tpw;
    const int nvalid = FULL ? TOK : (int)min((long)tpw, (long)a.B * a.L - m0);     // FULL: every block is whole, the bounds checks fold away
    float* sliceH = (C::XDED ? xded : reinterpret_cast<float*>(actH)) + wave * 1024;
    float* sliceA = (C::XDED ? xded : reinterpret_cast<float*>(actA)) + wave * 1024;
    const float* bo_p = sPar, *g2_p = sPar + D, *be2_p = sPar + 2 * D, *b2_p = sPar + 3 * D, *g1n_p = sPar + 4 * D, *be1n_p = sPar + 5 * D,
               *b1_p = sPar + 6 * D;
    const bool nxt = a.qkv_next != nullptr;

    long long t0_ = 0, t1_;
    int ph_ = 0;
#define R_STAMP() do { if (DBG && a.dbg && blockIdx.x == 0 && threadIdx.x == 0) { t1_ = clock64(); a.dbg[ph_++] = t1_ - t0_; t0_ = t1_; } } while (0)
    long long wg_t0 = 0;
    if (DBG && a.dbg) { t0_ = clock64(); wg_t0 = (long long)__builtin_amdgcn_s_memrealtime(); }
    auto wg_record = [&]() {         // DBG: per workgroup {start, end} in 100 MHz ticks, XCC id, HW_ID (which CU it ran on): dbg[32 + 4 b ..]
        if (DBG && a.dbg && threadIdx.x == 0) {
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            long long* d = a.dbg + 32 + 4 * (long)blockIdx.x;
            d[0] = wg_t0; d[1] = (long long)__builtin_amdgcn_s_memrealtime(); d[2] = xcc & 0xf; d[3] = hw;
        }
        // the LAST group's end (decoupled groups finish at different times): dbg[32 + 4 (1024 + b) + 1]
        if (DBG && a.dbg && TH > 1 && (int)threadIdx.x == (TH - 1) * 64 * NWV && blockIdx.x < 1024)
            a.dbg[32 + 4 * (1024 + (long)blockIdx.x) + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    };
    if (stagger < 0) {                                                // experiment: workgroup b starts (b mod 4) x |stagger| x 64 cycles late -- de-phases
        const int n = -stagger * (int)(blockIdx.x & 3);                   // the HBM bursts of a launch whose workgroups all run the same program in step
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
    }
    WRing<PD> ring;
    ring_fill<PD>((const h16_t*)a.Wo, KS, 0, cw, ring);
    // ---- residual base rows (consumed behind the first product), o rows -> actA, per-channel vectors -> sPar
    float4 rb[RB][4];
#pragma unroll
    for (int i = 0; i < RB; ++i) rows_load(a.base + m0 * D, row_offsets<D>(tb0 + i, cw, nvalid), rb[i]);
    stage_params<D, HID, C::NW>(sPar, a.bo, a.ln2_g, a.ln2_b, a.b2, a.ln1n_g, a.ln1n_b, a.b1);
    tile_load_rows<C>(actA, (const h16_t*)a.o + m0 * D, D, nvalid);
    __syncthreads();                                                  // o in actA, parameters in sPar (DEC: the workgroup's only common barrier)
    if (DEC && stagger > 0 && grp.th > 0) {                           // DEC: the later groups start `stagger` x 64 cycles late
        for (int i = 0; i < stagger * grp.th; ++i) __builtin_amdgcn_s_sleep(1);
    }
    R_STAMP();      // 0: loads (o, parameters) + barrier
    // ============================================================ x1 = base + dropout(o . Wo^T + bo);  n2 = LN2(x1)
    f32x16_t acc[RB];
    zero<RB>(acc);
    gemm_unit<RB, KS, PD>((const h16_t*)a.Wo, KS, 0, cw, actA, ALD, tb0, acc, ring);
    ring_fill<PD>((const h16_t*)a.W1, KS, 0, cw, ring);              // fc1 chunk 0
    R_STAMP();      // 1: o_proj MFMA
    f32x16_t xr[RB];                                                  // the residual rows: rows -> slice -> accumulator layout
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        rows_to_slice(sliceH, rb[i]);
        slice_to_acc<false>(sliceH, xr[i]);
    }
    {
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res1, a.p_res1);
#pragma unroll
        for (int i = 0; i < RB; ++i) residual_epilogue<D>(xr[i], acc[i], tb0 + i, cw, bo_p, rng, a.p_res1 > 0.f, m0, nvalid);
    }
    // x1 leaves for HBM in whole row segments (actH is not in use yet)
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        acc_to_slice(sliceH, xr[i]);
        slice_store_rows<ST>(sliceH, a.x1 + m0 * D, row_offsets<D>(tb0 + i, cw, nvalid));
    }
    R_STAMP();      // 2: base rows in, residual epilogue, x1 rows out
#pragma unroll
    for (int i = 0; i < RB; ++i) acc[i] = xr[i];
    layernorm<D, RB>(acc, g2_p, be2_p, sPair, a.mean2 + m0, a.rstd2 + m0, nvalid, [&] { grp.sync(); });
    acc_to_tile<D, RB>(acc, actA, ALD);                                // (the LayerNorm barrier: every wave is done reading o)
    grp.sync();                                                       // n2 complete in actA; every wave's x1 slices are drained
    R_STAMP();      // 3: LayerNorm 2 + n2 tile + barrier
    tile_store_rows<C, ST>(grp, actA, (h16_t*)a.n2 + m0 * D, D, nvalid);
    R_STAMP();      // 4: n2 rows out

    // ============================================================ MLP: HC chunks of D hidden channels
    f32x16_t acc2[RB];
    zero<RB>(acc2);
    float4 rp[RB][4];                                                 // KEEPX: the positional rows, requested in front of the last product
#pragma unroll
    for (int hc = 0; hc < HC; ++hc) {
        zero<RB>(acc);
        gemm_unit<RB, KS, PD>((const h16_t*)a.W1, KS, 0, hc * NWV + cw, actA, ALD, tb0, acc, ring);
        ring_fill<PD>((const h16_t*)a.W2, HID / 16, hc * KS, cw, ring);   // this chunk's fc2 slice
        if (hc == 0) R_STAMP();      // 5: fc1 MFMA (chunk 0)
        if (hc) grp.sync();                                           // every wave is done reading the previous chunk from actH
        bias_to_tile<D, RB>(acc, b1_p + hc * D, actH, ALD);             // u = h16(acc + b1), accumulator layout
        grp.sync();
        gelu_rows_pass<C, ST>(grp, actH, (h16_t*)a.u + m0 * HID + hc * D, (h16_t*)a.h + m0 * HID + hc * D, HID, nvalid);
        if (hc == 0) R_STAMP();      // 6: u tile + barrier + gelu row pass (chunk 0)
        grp.sync();
        if (hc == 0) R_STAMP();          // 7: barrier
        if (hc + 1 == HC) {
            if constexpr (C::KEEPX) {
                // x1 is still in xr; the positional rows are requested here and land under the last fc2 unit
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const RowOff ro = row_offsets<D>(tb0 + i, cw, nvalid);
                    rows_load(a.pos, pos_offsets<D>(tb0 + i, cw, m0, a.pos_rows, ro, a.pos != nullptr), rp[i]);
                }
            } else {
                // x1 (+ pos) comes back in row segments through the slices in actA (n2 is dead: every wave has passed the barriers behind
                // the last fc1 unit) IN FRONT of the last fc2 unit: behind it the row loads, the accumulators of both products and the
                // next unit's weights would all be live at once (> 128 registers)
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    float4 rx[4], rq[4];
                    const RowOff ro = row_offsets<D>(tb0 + i, cw, nvalid);
                    rows_load(a.x1 + m0 * D, ro, rx);
                    rows_load(a.pos, pos_offsets<D>(tb0 + i, cw, m0, a.pos_rows, ro, a.pos != nullptr), rq);
#pragma unroll
                    for (int it = 0; it < 4; ++it) { rx[it].x += rq[it].x; rx[it].y += rq[it].y; rx[it].z += rq[it].z; rx[it].w += rq[it].w; }
                    rows_to_slice(sliceA, rx);
                    slice_to_acc<false>(sliceA, xr[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            R_STAMP();     // 8 (last chunk): fc1 .. gelu of the last chunk + the x1 / pos row requests
        }
        gemm_unit<RB, KS, PD>((const h16_t*)a.W2, HID / 16, hc * KS, cw, actH, ALD, tb0, acc2, ring);
        if (hc + 1 < HC) ring_fill<PD>((const h16_t*)a.W1, KS, 0, (hc + 1) * NWV + cw, ring);
        if (hc == 0 && HC > 1) R_STAMP();          // 8 (first chunk): fc2 MFMA (chunk 0)
    }
    if (nxt) ring_fill<PD>((const h16_t*)a.Wqkv_next, KS, 0, cw, ring);
    R_STAMP();      // fc2 MFMA (last chunk)

    // ============================================================ x2 = x1 + dropout(h . W2^T + b2)  [+ pos -> next base, LN1, q|k|v]
    {
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res2, a.p_res2);
        const bool drop = a.p_res2 > 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (C::KEEPX) {                                    // xr = x1 + pos (the slices in actA: n2 is dead)
                rows_to_slice(sliceA, rp[i]);
                slice_to_acc<true>(sliceA, xr[i]);
            }
            residual_epilogue<D>(xr[i], acc2[i], tb0 + i, cw, b2_p, rng, drop, m0, nvalid);
            acc_to_slice(sliceA, xr[i]);
            slice_store_rows<ST>(sliceA, a.out + m0 * D, row_offsets<D>(tb0 + i, cw, nvalid));
        }
    }
    R_STAMP();      // final residual epilogue + out rows
    if (!nxt) { wg_record(); return; }
    layernorm<D, RB>(xr, g1n_p, be1n_p, sPair, a.mean1n + m0, a.rstd1n + m0, nvalid, [&] { grp.sync(); });
    acc_to_tile<D, RB>(xr, actH, ALD);                                 // (the LayerNorm barrier: every wave is done with the hidden chunk)
    grp.sync();                                                       // next n1 complete in actH
    R_STAMP();      // next LayerNorm 1 + tile + barrier
    tile_store_rows<C, ST>(grp, actH, (h16_t*)a.n1n + m0 * D, D, nvalid);
    R_STAMP();      // n1 rows out
    // q | k | v of the next layer: each wave's [32 RB x 32] result leaves through its slice as 64-byte row pieces
    h16_t* qslice = reinterpret_cast<h16_t*>(sliceA);
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        zero<RB>(acc);
        gemm_unit<RB, KS, PD>((const h16_t*)a.Wqkv_next, KS, 0, part * NWV + cw, actH, ALD, tb0, acc, ring);
        if (part + 1 < 3) ring_fill<PD>((const h16_t*)a.Wqkv_next, KS, 0, (part + 1) * NWV + cw, ring);
        acc_to_slice_h16<RB>(qslice, acc);
        slice_h16_store_rows<RB, ST>(qslice, (h16_t*)a.qkv_next + (m0 + tb0 * 32) * (3 * D) + part * D + 32 * cw, 3 * D, nvalid - tb0 * 32);
    }
    R_STAMP();      // q | k | v
    wg_record();
#undef R_STAMP
}

template <int D, int HID, int RB, int TH>
size_t fwd_lds()
{
    using C = Cfg<D, RB, TH>;
    return (size_t)2 * C::TILE * 2 + (size_t)C::TOK * C::NWV * 8 + (size_t)(6 * D + HID) * 4 + (C::XDED ? (size_t)C::NW * 4096 : 0) + 16;   // + the groups' arrival counters
}

// Tokens per workgroup of the 8-wave / 32-token geometry: two such workgroups share a CU at 7 % cost each (measured: 26.3 us alone,
// 28.2 us sharing), so the grid is cut to at most 2 x 256 workgroups -- M / 512 tokens each when that is at least 20, whole 32-token
// blocks otherwise.
static int tokens_per_wg(long M, int cus = 256)
{
    const long t = (M + 2 * cus - 1) / (2 * cus);
    return (t >= 20 && t <= 32) ? (int)t : 32;
}

template <int D, int HID, int RB, int TH, int MINW, bool DEC = false, int ST = 0>
int fwd_launch(const VpfSaLayerFwd& a, hipStream_t st, bool cut = false)
{
    using C = Cfg<D, RB, TH>;
    const long M = (long)a.B * a.L;
    const size_t lds = fwd_lds<D, HID, RB, TH>();
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)sa_rows_fwd_kernel<D, HID, RB, TH, MINW, true, false, DEC, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        if (hipFuncSetAttribute((const void*)sa_rows_fwd_kernel<D, HID, RB, TH, MINW, false, false, DEC, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    int tpw = (cut && C::TOK == 32) ? tokens_per_wg(M) : C::TOK;
    // VPF_SA_TPW (experiment): tokens per workgroup, N > 0: min(N, TOK); N < 0: ceil(M / (256 * -N)) -- the grid cut to -N workgroups per CU
    if (const int t = vpf_debug().sa_tpw) {
        const long want = t > 0 ? t : (M + 256L * -t - 1) / (256L * -t);
        tpw = (int)(want < C::TOK ? (want < 1 ? 1 : want) : C::TOK);
    }
    const int stagger = (DEC || vpf_debug().sa_stagger < 0) ? vpf_debug().sa_stagger : 0;
    if (a.dbg) {                                                     // phase stamps of workgroup 0 (tools/microbench.py): a build of its own
        if (hipFuncSetAttribute((const void*)sa_rows_fwd_kernel<D, HID, RB, TH, MINW, false, true, DEC, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        hipLaunchKernelGGL((sa_rows_fwd_kernel<D, HID, RB, TH, MINW, false, true, DEC, ST>), dim3(vpf_cdiv(M, tpw)), dim3(C::NT), lds, st, a, tpw, stagger);
    }
    else if (tpw == C::TOK && M % C::TOK == 0) hipLaunchKernelGGL((sa_rows_fwd_kernel<D, HID, RB, TH, MINW, true, false, DEC, ST>), dim3((unsigned)(M / C::TOK)), dim3(C::NT), lds, st, a, tpw, stagger);
    else hipLaunchKernelGGL((sa_rows_fwd_kernel<D, HID, RB, TH, MINW, false, false, DEC, ST>), dim3(vpf_cdiv(M, tpw)), dim3(C::NT), lds, st, a, tpw, stagger);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// ================================================================================================ cross-attention front
// What sits between Group2Emb and the cross-attention of the point-cloud branch -- on the step's critical path, as five launches of
// 20 - 36 us each (smallk_fwd, gemm, pack, layernorm_fwd, gemm: 112 us) -- as ONE row-block kernel:
//   hpos = gelu(centres . W0^T + b0)          position_emb[0:2], partseg.py:498-501 (K = 3: VALU)
//   pos  = hpos . W1^T + b1                   position_emb[2]
//   base = x + pos                            Encoder.forward, partseg.py:326 (the residual base of the cross-attention layer)
//   nq   = LayerNorm_q(base);  q = nq . Wq^T  CrossAttention / MultiHeadAttention, partseg.py:100-116, 48-51
// Everything the backward passes of PosMLPFn / EncoderFusedFn read is written as the separate kernels write it.
template <int D, int RB, int TH, int MINW>
__global__ void __launch_bounds__(2 * D * TH, MINW) ca_front_fwd_kernel(VpfCaFront a)
{
    using C = Cfg<D, RB, TH>;
    constexpr int NWV = C::NWV, TOK = C::TOK, ALD = C::ALD, KS = C::KS, PD = C::PD, HP = 128, HLD = HP + 8;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* actH = lds;                                               // [TOK][HLD]  hidden layer of the position MLP
    h16_t* actA = lds + TOK * HLD;                                   // [TOK][ALD]  nq
    float2* sPair = reinterpret_cast<float2*>(actA + C::TILE);        // [TOK][NWV]
    float* xded = reinterpret_cast<float*>(sPair + TOK * NWV);        // the transposition slices (a region of their own)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % NWV, tb0 = (wave / NWV) * RB;
    const long m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, a.M - m0);
    float* slice = xded + wave * 1024;

    WRing<PD> ring;
    ring_fill<PD>((const h16_t*)a.W1, HP / 16, 0, cw, ring);
    float4 rb[RB][4];                                                 // the tokens' rows: consumed behind the first product
#pragma unroll
    for (int i = 0; i < RB; ++i) rows_load(a.x + m0 * D, row_offsets<D>(tb0 + i, cw, nvalid), rb[i]);
    // ---- hidden layer: thread -> (token, 16 consecutive hidden channels); whole 256-byte rows leave for HBM
    {
        const int tid_ = fresh_tid();
        for (int e = tid_; e < TOK * (HP / 16); e += C::NT) {
            const int row = e / (HP / 16), c0 = (e % (HP / 16)) * 16;
            const bool ok = row < nvalid;
            float cx[3] = {0.f, 0.f, 0.f};
            if (ok) { const float* cp = a.centers + (size_t)(m0 + row) * a.C; cx[0] = cp[0]; cx[1] = cp[1]; cx[2] = cp[2]; }
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float u[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ch = c0 + 2 * j + h;
                    const float* wr = a.W0 + ch * a.C;
                    // the order smallk_fwd_kernel adds in: bias, then the taps
                    float t = a.b0[ch];
                    t += wr[0] * cx[0]; t += wr[1] * cx[1]; t += wr[2] * cx[2];
                    u[h] = vpf_gelu(t);
                }
                w[j] = pack_h16x2(u[0], u[1]);
            }
            *reinterpret_cast<uint4*>(actH + row * HLD + c0) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(actH + row * HLD + c0 + 8) = make_uint4(w[4], w[5], w[6], w[7]);
            if (ok) {
                h16_t* hp = (h16_t*)a.hpos + (size_t)(m0 + row) * HP + c0;
                *reinterpret_cast<uint4*>(hp) = make_uint4(w[0], w[1], w[2], w[3]);
                *reinterpret_cast<uint4*>(hp + 8) = make_uint4(w[4], w[5], w[6], w[7]);
            }
        }
    }
    __syncthreads();
    // ---- pos = hpos . W1^T + b1;  base = x + pos
    f32x16_t acc[RB], xr[RB];
    zero<RB>(acc);
    gemm_unit<RB, HP / 16, PD>((const h16_t*)a.W1, HP / 16, 0, cw, actH, HLD, tb0, acc, ring);
    ring_fill<PD>((const h16_t*)a.Wq, KS, 0, cw, ring);
    {
        const int lane = fresh_tid() & 63, hl = lane >> 5;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.b1 + 32 * cw + 8 * g + 4 * hl);
#pragma unroll
            for (int i = 0; i < RB; ++i) { acc[i][4 * g + 0] += b4.x; acc[i][4 * g + 1] += b4.y; acc[i][4 * g + 2] += b4.z; acc[i][4 * g + 3] += b4.w; }
        }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const RowOff ro = row_offsets<D>(tb0 + i, cw, nvalid);
        acc_to_slice(slice, acc[i]);
        slice_store_rows(slice, a.pos + m0 * D, ro);                   // pos rows out
        rows_to_slice(slice, rb[i]);
        slice_to_acc<false>(slice, xr[i]);
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[i][r] += acc[i][r];             // base = x + pos
        acc_to_slice(slice, xr[i]);
        slice_store_rows(slice, a.base + m0 * D, ro);
    }
    layernorm<D, RB>(xr, a.lnq_g, a.lnq_b, sPair, a.mean + m0, a.rstd + m0, nvalid);
    acc_to_tile<D, RB>(xr, actA, ALD);
    __syncthreads();
    tile_store_rows<C>(actA, (h16_t*)a.nq + m0 * D, D, nvalid);
    // ---- q = nq . Wq^T
    zero<RB>(acc);
    gemm_unit<RB, KS, PD>((const h16_t*)a.Wq, KS, 0, cw, actA, ALD, tb0, acc, ring);
    h16_t* qslice = reinterpret_cast<h16_t*>(slice);
    acc_to_slice_h16<RB>(qslice, acc);
    slice_h16_store_rows<RB>(qslice, (h16_t*)a.q + (m0 + tb0 * 32) * D + 32 * cw, D, nvalid - tb0 * 32);
}

// ================================================================================================ backward
// LayerNorm backward in place on the accumulator tile: acc = dL/dy -> dL/dx; x = the forward input of the LayerNorm in the same
// layout (becomes x-hat).  The per-channel parameter gradients of this wave's tokens go to pgrad[0 .. D) (dgamma) / pgrad[D .. 2D)
// (dbeta).  Identical arithmetic to sa_layernorm_bwd of sa_layer.hip.
// EARLY: the parameter-gradient sums of each channel group are folded over the lanes and written as soon as they exist (8 live
// registers instead of 32 across the exchange barrier; same sums, same order).
template <int D, int RB, bool EARLY = false>
__device__ __forceinline__ void layernorm_bwd(f32x16_t (&acc)[RB], f32x16_t (&x)[RB], const float* __restrict__ mean, const float* __restrict__ rstd,
                                              const float* __restrict__ gamma, float* sStat2, float* __restrict__ pgrad, int nvalid)
{
    constexpr int NWV = D / 32;
    const int tid_ = fresh_tid();
    int cw, tb0;
    who<D, RB>(tid_, cw, tb0);
    const int lane = tid_ & 63, hl = lane >> 5, t = lane & 31;
    float mu[RB], rs[RB], s1[RB], s2[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int tok = (tb0 + i) * 32 + t;
        const bool ok = tok < nvalid;
        mu[i] = ok ? mean[tok] : 0.f;
        rs[i] = ok ? rstd[tok] : 0.f;
        s1[i] = 0.f; s2[i] = 0.f;
    }
    float dgam[4][4], dbet[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 ga = *reinterpret_cast<const float4*>(gamma + 32 * cw + 8 * g + 4 * hl);
        const float gg[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { dgam[g][q] = 0.f; dbet[g][q] = 0.f; }
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xn = (x[i][4 * g + q] - mu[i]) * rs[i];
                const float dy = acc[i][4 * g + q];
                dgam[g][q] += dy * xn;
                dbet[g][q] += dy;
                const float gy = dy * gg[q];
                s1[i] += gy;
                s2[i] += gy * xn;
                acc[i][4 * g + q] = gy;
                x[i][4 * g + q] = xn;
            }
        if constexpr (EARLY) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a = dgam[g][q], b = dbet[g][q];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (t == 0) {
                    const int c = 32 * cw + 8 * g + 4 * hl + q;
                    pgrad[c] = a;
                    pgrad[D + c] = b;
                }
            }
        }
    }
    // per-token sums over the channels: lane ^ 32, then the waves that share the token
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        s1[i] += __shfl_xor(s1[i], 32, 64);
        s2[i] += __shfl_xor(s2[i], 32, 64);
        if (lane < 32) *reinterpret_cast<float2*>(sStat2 + (((tb0 + i) * 32 + lane) * NWV + cw) * 2) = make_float2(s1[i], s2[i]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NWV; w2 += 2) {
            const float4 v0 = *reinterpret_cast<const float4*>(sStat2 + (((tb0 + i) * 32 + t) * NWV + w2) * 2);
            t1 += v0.x + v0.z; t2 += v0.y + v0.w;
        }
        s1[i] = t1 * (1.0f / D);
        s2[i] = t2 * (1.0f / D);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][4 * g + q] = rs[i] * (acc[i][4 * g + q] - s1[i] - x[i][4 * g + q] * s2[i]);
    // parameter gradients: sum over this wave's tokens = over the 32 lanes of each half
    if constexpr (!EARLY)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = dgam[g][q], b = dbet[g][q];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if (t == 0) {
                const int c = 32 * cw + 8 * g + 4 * hl + q;
                pgrad[c] = a;
                pgrad[D + c] = b;
            }
        }
}

// cooperative row pass: dst tile (h16, [TOK][ALD]) and HBM (h16 rows, ld = D) = dropout'(src f32 rows) -- the operand of the next
// product and of the weight-gradient GEMM
template <class C, int D>
__device__ __forceinline__ void dropout_bwd_rows(const float* __restrict__ src_u, h16_t* sAct, h16_t* __restrict__ dst_u, const VpfRng& rng, bool drop,
                                                 long m0, int nvalid)
{
    constexpr int XPT = C::TOK * (D / 4) / C::NT;
    const int tid_ = fresh_tid();
    float4 dr[XPT];
#pragma unroll
    for (int it = 0; it < XPT; ++it) {
        const int e = tid_ + it * C::NT, row = e / (D / 4), c4 = e - row * (D / 4);
        dr[it] = *reinterpret_cast<const float4*>(src_u + (unsigned)((row < nvalid ? row : 0) * D + c4 * 4));
    }
    const float sc = drop ? rng.scale : 1.f;
#pragma unroll
    for (int it = 0; it < XPT; ++it) {
        const int e = tid_ + it * C::NT, row = e / (D / 4), c4 = e - row * (D / 4);
        const bool ok = row < nvalid;
        const size_t off = (size_t)(m0 + (ok ? row : 0)) * D + c4 * 4;
        const uint32_t keep = !ok ? 0u : (drop ? vpf_keep4(rng, (uint64_t)off >> 2) : 15u);
        uint2 w;
        w.x = pack_h16x2((keep & 1u) ? dr[it].x * sc : 0.f, (keep & 2u) ? dr[it].y * sc : 0.f);
        w.y = pack_h16x2((keep & 4u) ? dr[it].z * sc : 0.f, (keep & 8u) ? dr[it].w * sc : 0.f);
        *reinterpret_cast<uint2*>(sAct + row * C::ALD + c4 * 4) = w;
        if (ok) *reinterpret_cast<uint2*>(dst_u + (unsigned)(row * D + c4 * 4)) = w;
    }
}

//   d(x2) -> dropout' -> [dz2] -> . W2 * gelu'(u) -> [du] -> . W1 -> LayerNorm-2' (+ d) -> [dx1] -> dropout' -> [dz1] -> . Wo -> [do]
template <int D, int HID, int RB, int TH, int MINW>
__global__ void __launch_bounds__(2 * D * TH, MINW) sa_rows_bwd_mlp_kernel(VpfSaLayerBwd a)
{
    using C = Cfg<D, RB, TH>;
    constexpr int NWV = C::NWV, TOK = C::TOK, ALD = C::ALD, KS = C::KS, HC = HID / D, PD = C::PD;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* actA = lds;                                               // dz2 -> (slices: x1, LayerNorm-2' result, do)
    h16_t* actH = lds + C::TILE;                                     // one D-wide chunk of du, then dz1
    float* sStat2 = reinterpret_cast<float*>(actH + C::TILE);         // [TOK][NWV] float2
    float* xded = sStat2 + TOK * NWV * 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % NWV, tb0 = (wave / NWV) * RB, th = wave / NWV;
    const long M = (long)a.M, m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);
    float* slice = (C::XDED ? xded : reinterpret_cast<float*>(actA)) + wave * 1024;

    WRing<PD> ring;
    ring_fill<PD>((const h16_t*)a.W2T, KS, 0, cw, ring);
    // ---- dz2 = dropout'(d): operand tile + HBM, in the row layout
    {
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res2, a.p_res2);
        dropout_bwd_rows<C, D>(a.d + m0 * D, actA, (h16_t*)a.dz2 + m0 * D, rng, a.p_res2 > 0.f, m0, nvalid);
    }
    __syncthreads();
    // ---- du = (dz2 . W2) * gelu'(u) ;  dn = du . W1
    f32x16_t acc[RB], acc2[RB];
    zero<RB>(acc2);
#pragma unroll
    for (int hc = 0; hc < HC; ++hc) {
        uint2 uu[4][RB];                                               // the pre-GELU values of this lane's elements (accumulator layout)
        {
            const int lane = fresh_tid() & 63, hl = lane >> 5, t = lane & 31;
            const h16_t* u_u = (const h16_t*)a.u + m0 * HID + hc * D + 32 * cw;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int tok = (tb0 + i) * 32 + t;
                    uu[g][i] = tok < nvalid ? *reinterpret_cast<const uint2*>(u_u + (unsigned)(tok * HID + 8 * g + 4 * hl)) : make_uint2(0u, 0u);
                }
        }
        zero<RB>(acc);
        gemm_unit<RB, KS, PD>((const h16_t*)a.W2T, KS, 0, hc * NWV + cw, actA, ALD, tb0, acc, ring);
        ring_fill<PD>((const h16_t*)a.W1T, HID / 16, hc * KS, cw, ring);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                acc[i][4 * g + 0] *= vpf_gelu_grad(h16_lo(uu[g][i].x));
                acc[i][4 * g + 1] *= vpf_gelu_grad(h16_hi(uu[g][i].x));
                acc[i][4 * g + 2] *= vpf_gelu_grad(h16_lo(uu[g][i].y));
                acc[i][4 * g + 3] *= vpf_gelu_grad(h16_hi(uu[g][i].y));
            }
        if (hc) __syncthreads();                                      // every wave is done reading the previous chunk from actH
        acc_to_tile<D, RB>(acc, actH, ALD);
        __syncthreads();
        tile_store_rows<C>(actH, (h16_t*)a.du + m0 * HID + hc * D, HID, nvalid);
        gemm_unit<RB, KS, PD>((const h16_t*)a.W1T, HID / 16, hc * KS, cw, actH, ALD, tb0, acc2, ring);
        if (hc + 1 < HC) ring_fill<PD>((const h16_t*)a.W2T, KS, 0, (hc + 1) * NWV + cw, ring);
    }
    ring_fill<PD>((const h16_t*)a.WoT, KS, 0, cw, ring);
    // ---- LayerNorm-2'(dn): x1 comes in row segments through the slices (dz2 in actA is dead: every wave has left the chunk loop's
    //      last W2 product behind a barrier)
    f32x16_t xr[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float4 rx[4];
        rows_load(a.x1 + m0 * D, row_offsets<D>(tb0 + i, cw, nvalid), rx);
        rows_to_slice(slice, rx);
        slice_to_acc<false>(slice, xr[i]);
    }
    layernorm_bwd<D, RB>(acc2, xr, a.mean2 + m0, a.rstd2 + m0, a.ln2_g, sStat2, a.pgrad2 + ((size_t)blockIdx.x * TH + th) * 2 * D, nvalid);
    // ---- dx1 = . + d;  dz1 = dropout'(dx1): row layout through the slices; dz1 -> actH (the last du chunk is dead: every wave passed
    //      the LayerNorm barrier behind its last W1 product) + HBM
    {
        const VpfRng rng = vpf_rng_init(a.rng, a.site_res1, a.p_res1);
        const bool drop = a.p_res1 > 0.f;
        const float sc = drop ? rng.scale : 1.f;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            const RowOff ro = row_offsets<D>(tb0 + i, cw, nvalid);
            float4 rd[4];
            rows_load(a.d + m0 * D, ro, rd);
            acc_to_slice(slice, acc2[i]);
            const int lane = fresh_tid() & 63, tl = lane >> 3, ch = lane & 7;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int tok = 8 * it + tl, row = (tb0 + i) * 32 + tok;
                float4 v = *reinterpret_cast<const float4*>(slice + tok * 32 + ((ch ^ swz(tok)) << 2));
                v.x += rd[it].x; v.y += rd[it].y; v.z += rd[it].z; v.w += rd[it].w;
                const bool ok = (ro.ok >> it) & 1u;
                if (ok) *reinterpret_cast<float4*>(a.dx1 + m0 * D + ro.o[it]) = v;
                const uint32_t keep = !ok ? 0u : (drop ? vpf_keep4(rng, ((uint64_t)m0 * D + ro.o[it]) >> 2) : 15u);
                uint2 w;
                w.x = pack_h16x2((keep & 1u) ? v.x * sc : 0.f, (keep & 2u) ? v.y * sc : 0.f);
                w.y = pack_h16x2((keep & 4u) ? v.z * sc : 0.f, (keep & 8u) ? v.w * sc : 0.f);
                *reinterpret_cast<uint2*>(actH + row * ALD + 32 * cw + 4 * ch) = w;
                if (ok) *reinterpret_cast<uint2*>((h16_t*)a.dz1 + m0 * D + ro.o[it]) = w;
            }
        }
    }
    __syncthreads();                                                  // dz1 complete in actH
    // ---- do = dz1 . Wo: leaves through the wave's slice as 64-byte row pieces
    zero<RB>(acc);
    gemm_unit<RB, KS, PD>((const h16_t*)a.WoT, KS, 0, cw, actH, ALD, tb0, acc, ring);
    h16_t* qslice = reinterpret_cast<h16_t*>(slice);
    acc_to_slice_h16<RB>(qslice, acc);
    slice_h16_store_rows<RB>(qslice, (h16_t*)a.dout_attn + (m0 + tb0 * 32) * D + 32 * cw, D, nvalid - tb0 * 32);
}

//   [dqkv] . Wqkv -> LayerNorm-1' (+ dx1) -> [dbase] (+= dsum)
// NP: D-wide parts of the incoming gradient (3: dq | dk | dv against Wqkv^T; 1: the cross-attention layer's dq against Wq^T, vpf_ca_front_bwd)
template <int D, int RB, int TH, int MINW, int NP = 3>
__global__ void __launch_bounds__(2 * D * TH, MINW) sa_rows_bwd_qkv_kernel(VpfSaLayerBwd a)
{
    using C = Cfg<D, RB, TH>;
    constexpr int NWV = C::NWV, TOK = C::TOK, ALD = C::ALD, KS = C::KS, PD = C::PD;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* buf0 = lds;                                               // two buffers of [TOK][ALD]: the q | k | v slices of dqkv
    h16_t* buf1 = lds + C::TILE;
    float* sStat2 = reinterpret_cast<float*>(buf1 + C::TILE);
    float* xded = sStat2 + TOK * NWV * 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % NWV, tb0 = (wave / NWV) * RB, th = wave / NWV;
    const long M = (long)a.M, m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);
    float* slice = (C::XDED ? xded : reinterpret_cast<float*>(buf1)) + wave * 1024;      // buf1 is free once the last part is staged

    WRing<PD> ring;
    ring_fill<PD>((const h16_t*)a.WqkvT, NP * KS, 0, cw, ring);
    tile_load_rows<C>(buf0, (const h16_t*)a.dqkv + m0 * (NP * D), NP * D, nvalid);
    __syncthreads();
    f32x16_t acc[RB];
    zero<RB>(acc);
#pragma unroll
    for (int part = 0; part < NP; ++part) {
        // (the next part is staged behind this part's product: the other buffer is free -- every wave passed the last barrier)
        gemm_unit<RB, KS, PD>((const h16_t*)a.WqkvT, NP * KS, part * KS, cw, (part & 1) ? buf1 : buf0, ALD, tb0, acc, ring);
        if (part + 1 < NP) {
            ring_fill<PD>((const h16_t*)a.WqkvT, NP * KS, (part + 1) * KS, cw, ring);
            tile_load_rows<C>((part & 1) ? buf0 : buf1, (const h16_t*)a.dqkv + m0 * (NP * D) + (part + 1) * D, NP * D, nvalid);
            __syncthreads();
        }
    }
    // ---- dbase = LayerNorm-1'(dn1) + dx1: the LayerNorm input comes in row segments through the slices
    f32x16_t xr[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float4 rx[4];
        rows_load(a.base + m0 * D, row_offsets<D>(tb0 + i, cw, nvalid), rx);
        rows_to_slice(slice, rx);
        slice_to_acc<false>(slice, xr[i]);
    }
    layernorm_bwd<D, RB>(acc, xr, a.mean1 + m0, a.rstd1 + m0, a.ln1_g, sStat2, a.pgrad1 + ((size_t)blockIdx.x * TH + th) * 2 * D, nvalid);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const RowOff ro = row_offsets<D>(tb0 + i, cw, nvalid);
        float4 dv[4], sv[4];
        rows_load(a.dx1 + m0 * D, ro, dv);
        if (a.dsum && !a.dsum_init) rows_load(a.dsum + m0 * D, ro, sv);
        else {
#pragma unroll
            for (int it = 0; it < 4; ++it) sv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        acc_to_slice(slice, acc[i]);
        const int lane = fresh_tid() & 63, tl = lane >> 3, ch = lane & 7;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int tok = 8 * it + tl;
            float4 v = *reinterpret_cast<const float4*>(slice + tok * 32 + ((ch ^ swz(tok)) << 2));
            v.x += dv[it].x; v.y += dv[it].y; v.z += dv[it].z; v.w += dv[it].w;
            if ((ro.ok >> it) & 1u) {
                *reinterpret_cast<float4*>(a.dbase + m0 * D + ro.o[it]) = v;
                if (a.dsum) *reinterpret_cast<float4*>(a.dsum + m0 * D + ro.o[it]) = make_float4(sv[it].x + v.x, sv[it].y + v.y, sv[it].z + v.z, sv[it].w + v.w);
            }
        }
    }
}

template <int D, int RB, int TH>
size_t bwd_lds()
{
    using C = Cfg<D, RB, TH>;
    return (size_t)2 * C::TILE * 2 + (size_t)C::TOK * C::NWV * 8 + (C::XDED ? (size_t)C::NW * 4096 : 0);
}
template <int D, int HID, int RB, int TH, int MINW>
int bwd_mlp_launch(const VpfSaLayerBwd& a, hipStream_t st)
{
    using C = Cfg<D, RB, TH>;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)sa_rows_bwd_mlp_kernel<D, HID, RB, TH, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    const size_t lds = bwd_lds<D, RB, TH>();
    hipLaunchKernelGGL((sa_rows_bwd_mlp_kernel<D, HID, RB, TH, MINW>), dim3(vpf_cdiv((long)a.M, C::TOK)), dim3(C::NT), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
template <int D, int RB, int TH, int MINW>
int bwd_qkv_launch(const VpfSaLayerBwd& a, hipStream_t st)
{
    using C = Cfg<D, RB, TH>;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)sa_rows_bwd_qkv_kernel<D, RB, TH, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    const size_t lds = bwd_lds<D, RB, TH>();
    hipLaunchKernelGGL((sa_rows_bwd_qkv_kernel<D, RB, TH, MINW>), dim3(vpf_cdiv((long)a.M, C::TOK)), dim3(C::NT), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

}   // namespace

// (D = 256, hidden 1024: mlp_widen_factor 4 at the default width -- scripts/pretrain/pt-*-MR4-0.sh of the reference; only these kernels have it)
bool sa_rows_supported(int D, int hidden) { return (D == 256 && (hidden == 512 || hidden == 1024)) || (D == 384 && hidden == 1536); }

int sa_rows_fwd_launch(const VpfSaLayerFwd& a, hipStream_t st)
{
    if (a.D == 256 && a.hidden == 512) {
        switch (vpf_debug().sa_rb) {          // VPF_SA_RB: 12 = <RB 1, TH 2>, 2 = <2, 1>, 1 = <1, 1> whole blocks, 0 = <1, 1> cut to 2 per CU
            case 2: return fwd_launch<256, 512, 2, 1, 4>(a, st);
            case 1: return fwd_launch<256, 512, 1, 1, 4>(a, st);
            case 12: return fwd_launch<256, 512, 1, 2, 4>(a, st);
            case 13:                                                           // round 5: the same 16 waves as two decoupled 8-wave groups
                switch (vpf_debug().sa_store) {                                // VPF_SA_STORE: cache policy of the row stores (st16)
                    case 1: return fwd_launch<256, 512, 1, 2, 4, true, 1>(a, st);
                    case 2: return fwd_launch<256, 512, 1, 2, 4, true, 2>(a, st);
                    case 3: return fwd_launch<256, 512, 1, 2, 4, true, 3>(a, st);
                    default: return fwd_launch<256, 512, 1, 2, 4, true, 0>(a, st);
                }
            default: return fwd_launch<256, 512, 1, 1, 4>(a, st, true);
        }
    }
    if (a.D == 256 && a.hidden == 1024) return fwd_launch<256, 1024, 1, 1, 4>(a, st, true);
    if (a.D == 384 && a.hidden == 1536) return vpf_debug().sa_rb == 2 ? fwd_launch<384, 1536, 2, 1, 3>(a, st) : fwd_launch<384, 1536, 1, 1, 3>(a, st);
    return VPF_ERR_UNSUPPORTED;
}

// tokens per LayerNorm-parameter-gradient partial row of the backward kernels (one row per wave group): the caller sizes pgrad1 /
// pgrad2 as ceil(M / this) rows of 2 D floats
int sa_rows_bwd_pgrad_tokens(int D) { return (D == 384 && vpf_debug().sa_rb == 2) ? 64 : 32; }

int sa_rows_bwd_mlp_launch(const VpfSaLayerBwd& a, hipStream_t st)
{
    if (a.D == 256 && a.hidden == 512) return bwd_mlp_launch<256, 512, 1, 2, 4>(a, st);      // 16 waves x 32 tokens each (the 8-wave x 64 shape spills at 128 registers)
    if (a.D == 256 && a.hidden == 1024) return bwd_mlp_launch<256, 1024, 1, 2, 4>(a, st);
    if (a.D == 384 && a.hidden == 1536) return vpf_debug().sa_rb == 2 ? bwd_mlp_launch<384, 1536, 2, 1, 3>(a, st) : bwd_mlp_launch<384, 1536, 1, 1, 3>(a, st);
    return VPF_ERR_UNSUPPORTED;
}
template <int D, int RB, int TH, int MINW>
static int ca_front_bwd_launch(const VpfSaLayerBwd& a, hipStream_t st)
{
    using C = Cfg<D, RB, TH>;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)sa_rows_bwd_qkv_kernel<D, RB, TH, MINW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    const size_t lds = bwd_lds<D, RB, TH>();
    hipLaunchKernelGGL((sa_rows_bwd_qkv_kernel<D, RB, TH, MINW, 1>), dim3(vpf_cdiv((long)a.M, C::TOK)), dim3(C::NT), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// the cross-attention layer's query-side backward at D = 384 (partial rows per 32 tokens, like the other D = 384 backward kernels)
int sa_rows_ca_front_bwd_launch(const VpfSaLayerBwd& a, hipStream_t st)
{
    if (a.D == 384 && vpf_debug().sa_rb != 2) return ca_front_bwd_launch<384, 1, 1, 3>(a, st);
    return VPF_ERR_UNSUPPORTED;
}
int sa_rows_bwd_qkv_launch(const VpfSaLayerBwd& a, hipStream_t st)
{
    if (a.D == 256) return bwd_qkv_launch<256, 1, 2, 4>(a, st);
    if (a.D == 384) return vpf_debug().sa_rb == 2 ? bwd_qkv_launch<384, 2, 1, 3>(a, st) : bwd_qkv_launch<384, 1, 1, 3>(a, st);
    return VPF_ERR_UNSUPPORTED;
}

// ================================================================================================ K / V producer, backward
// The same chain as adapter_kv_bwd_kernel (sa_layer.hip: dkv [M, 2D] -> . (Wk | Wv) -> kv LayerNorm' -> dxkv h16 -> . W2 -> da1 h16
// [M, 64], identical arithmetic and rounding points) on this file's building blocks: two operand tiles instead of three (the LayerNorm's
// input rows are staged into the dk tile once its product is done), a 4-deep weight ring instead of 16 k-steps of prefetch -- 128 VGPRs
// and 72 KB of LDS, so TWO workgroups share a CU.  The launch has 2 048 workgroups (131 072 points): unlike the encoder's 192-workgroup
// launches the second one is always there, and its products fill the first one's load / LayerNorm / store phases.
template <int D, int RB>
__device__ __forceinline__ void tile_to_acc(const h16_t* sAct, int ald, f32x16_t (&x)[RB])
{
    const int tid_ = fresh_tid();
    int cw, tb0;
    who<D, RB>(tid_, cw, tb0);
    const int lane = tid_ & 63, hl = lane >> 5, t = lane & 31;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 u = *reinterpret_cast<const uint2*>(sAct + ((tb0 + i) * 32 + t) * ald + 32 * cw + 8 * g + 4 * hl);
            x[i][4 * g + 0] = h16_lo(u.x);
            x[i][4 * g + 1] = h16_hi(u.x);
            x[i][4 * g + 2] = h16_lo(u.y);
            x[i][4 * g + 3] = h16_hi(u.y);
        }
}
template <class C>
__device__ __forceinline__ void rows_request(const h16_t* __restrict__ Gu, int ld, int nvalid, uint4 (&r)[C::TOK * C::C8 / C::NT])
{
    const int tid_ = fresh_tid();
#pragma unroll
    for (int it = 0; it < C::TOK * C::C8 / C::NT; ++it) {
        const int e = tid_ + it * C::NT, row = e / C::C8, ch = e - row * C::C8;
        r[it] = row < nvalid ? *reinterpret_cast<const uint4*>(Gu + (unsigned)(row * ld + ch * 8)) : make_uint4(0, 0, 0, 0);
    }
}
template <class C>
__device__ __forceinline__ void rows_commit(h16_t* sAct, const uint4 (&r)[C::TOK * C::C8 / C::NT])
{
    const int tid_ = fresh_tid();
#pragma unroll
    for (int it = 0; it < C::TOK * C::C8 / C::NT; ++it) {
        const int e = tid_ + it * C::NT, row = e / C::C8, ch = e - row * C::C8;
        *reinterpret_cast<uint4*>(sAct + row * C::ALD + ch * 8) = r[it];
    }
}
template <int D, int RB, int TH, int MINW>
__global__ void __launch_bounds__(2 * D * TH, MINW) adapter_kv_bwd_rows_kernel(VpfAdapterKvBwd a)
{
    using C = Cfg<D, RB, TH>;
    constexpr int NWV = C::NWV, TOK = C::TOK, ALD = C::ALD, KS = C::KS, PD = C::PD;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* buf0 = lds;                                               // dk rows, then the LayerNorm input rows, then the da1 slices
    h16_t* buf1 = lds + C::TILE;                                     // dv rows, then dxkv
    float* sStat2 = reinterpret_cast<float*>(buf1 + C::TILE);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % NWV, tb0 = (wave / NWV) * RB, th = wave / NWV;
    const long M = a.M, m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, M - m0);
    const h16_t* dkv = (const h16_t*)a.dkv + m0 * (2 * D);

    WRing<PD> ring;
    ring_fill<PD>((const h16_t*)a.WkvT, 2 * KS, 0, cw, ring);
    uint4 rr[TOK * C::C8 / C::NT];
    rows_request<C>(dkv, 2 * D, nvalid, rr);
    rows_commit<C>(buf0, rr);
    rows_request<C>(dkv + D, 2 * D, nvalid, rr);                    // the dv rows travel while the dk rows settle
    __syncthreads();
    f32x16_t acc[RB];
    zero<RB>(acc);
    gemm_unit<RB, KS, PD>((const h16_t*)a.WkvT, 2 * KS, 0, cw, buf0, ALD, tb0, acc, ring);
    ring_fill<PD>((const h16_t*)a.WkvT, 2 * KS, KS, cw, ring);
    rows_commit<C>(buf1, rr);
    rows_request<C>((const h16_t*)a.xkv + m0 * D, D, nvalid, rr);  // the LayerNorm's input rows, behind the second product
    __syncthreads();                                                  // (every wave is done with the dk tile)
    gemm_unit<RB, KS, PD>((const h16_t*)a.WkvT, 2 * KS, KS, cw, buf1, ALD, tb0, acc, ring);
    rows_commit<C>(buf0, rr);
    // the unfused path stores dnk as h16 before the LayerNorm backward: round the same way
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = h16_to_f32(f32_to_h16(acc[i][r]));
    __syncthreads();
    f32x16_t xr[RB];
    tile_to_acc<D, RB>(buf0, ALD, xr);
    layernorm_bwd<D, RB, true>(acc, xr, a.mean + m0, a.rstd + m0, a.lnkv_g, sStat2, a.pgrad_kv + ((size_t)blockIdx.x * TH + th) * 2 * D, nvalid);
    // (the barrier inside: every wave has finished the dv tile and read its x rows)
    if (cw < 2) ring_fill<PD>((const h16_t*)a.W2T, KS, 0, cw, ring);      // (not earlier: 16 more live registers across the LayerNorm spill)
    acc_to_tile<D, RB>(acc, buf1, ALD);
    __syncthreads();
    tile_store_rows<C>(buf1, (h16_t*)a.dxkv + m0 * D, D, nvalid);
    // ---- da1 = dxkv . W2 (64 hidden channels: the waves of channel blocks 0 and 1), h16 like the unfused dgrad output
    if (cw < 2) {
        zero<RB>(acc);
        gemm_unit<RB, KS, PD>((const h16_t*)a.W2T, KS, 0, cw, buf1, ALD, tb0, acc, ring);
        h16_t* slice = buf0 + wave * (RB * 1024);                   // wave-private; the x rows are in registers since the barriers above
        acc_to_slice_h16<RB>(slice, acc);
        slice_h16_store_rows<RB>(slice, (h16_t*)a.da1 + (m0 + tb0 * 32) * 64 + 32 * cw, 64, nvalid - tb0 * 32);
    }
}
// ================================================================================================ K / V producer, forward (any D)
// adapter_kv_fwd_kernel of sa_layer.hip (D = 256) on this file's building blocks, for the widths it does not cover (D = 384, BASELINE
// config 4): Linear(C, 64) + LayerNorm(64) + ReLU in VALU (8 lanes per point) -> LDS -> . W2 (K = 64) + bias, rounded to h16 as the
// unfused path stores it -> kv LayerNorm -> LDS -> . Wk | . Wv; a1, xkv, nk and k | v leave as whole rows.  Same arithmetic, same
// rounding points.
template <int D, int RB, int TH, int MINW>
__global__ void __launch_bounds__(2 * D * TH, MINW) adapter_kv_fwd_rows_kernel(VpfAdapterKv a)
{
    using C = Cfg<D, RB, TH>;
    constexpr int NWV = C::NWV, TOK = C::TOK, ALD = C::ALD, KS = C::KS, PD = C::PD, A1LD = 72;
    extern __shared__ __attribute__((aligned(16))) h16_t lds[];
    h16_t* sA1 = lds;                                               // [TOK][A1LD] hidden layer
    h16_t* sX = lds + TOK * A1LD;                                   // [TOK][ALD]  per-point embedding, then the k / v halves on their way out
    h16_t* actA = sX + C::TILE;                                     // [TOK][ALD]  normalised embedding
    float2* sPair = reinterpret_cast<float2*>(actA + C::TILE);       // [TOK][NWV]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % NWV, tb0 = (wave / NWV) * RB;
    const long m0 = (long)blockIdx.x * TOK;
    const int nvalid = (int)min((long)TOK, a.M - m0);
    const int Cin = a.C;

    WRing<PD> ring;
    ring_fill<PD>((const h16_t*)a.Wkv, KS, 0, cw, ring);
    // ---- hidden layer: thread = (token, 8 of the 64 channels); LayerNorm over the token's 8 threads (lanes ^1 ^2 ^4)
    {
        const int tid_ = fresh_tid();
        for (int e = tid_; e < TOK * 8; e += C::NT) {
            const int tok = e >> 3, cg = (e & 7) * 8;
            const bool ok = tok < nvalid;
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = (ok && j < Cin) ? a.x[(size_t)(m0 + tok) * Cin + j] : 0.f;
            float h[8], sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = a.b1[cg + k];
                for (int j = 0; j < Cin; ++j) v += a.W1[(cg + k) * Cin + j] * xv[j];
                h[k] = v; sum += v;
            }
            sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
            const float mu = sum * (1.f / 64.f);
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
    }
    __syncthreads();
    // ---- per-point embedding = hidden . W2^T + b2 (K = 64: four k-steps), rounded to h16, then the kv LayerNorm
    f32x16_t acc[RB];
    zero<RB>(acc);
    {
        const unsigned lane = fresh_tid() & 63;
        const uint4* w0 = reinterpret_cast<const uint4*>(a.W2) + (size_t)cw * 4 * 64 + lane;
        const h16_t* xrow = sA1 + (tb0 * 32 + (lane & 31)) * A1LD + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8_t af = __builtin_bit_cast(h16x8_t, w0[ks * 64]);
#pragma unroll
            for (int i = 0; i < RB; ++i)
                acc[i] = vpf_mfma32(af, __builtin_bit_cast(h16x8_t, *reinterpret_cast<const uint4*>(xrow + i * 32 * A1LD + ks * 16)), acc[i]);
        }
    }
    {
        const int lane = fresh_tid() & 63, hl = lane >> 5, t = lane & 31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * cw + 8 * g + 4 * hl;
            const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + c);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                uint2 u;
                u.x = pack_h16x2(acc[i][4 * g + 0] + b2.x, acc[i][4 * g + 1] + b2.y);
                u.y = pack_h16x2(acc[i][4 * g + 2] + b2.z, acc[i][4 * g + 3] + b2.w);
                *reinterpret_cast<uint2*>(sX + ((tb0 + i) * 32 + t) * ALD + c) = u;
                acc[i][4 * g + 0] = h16_lo(u.x); acc[i][4 * g + 1] = h16_hi(u.x);
                acc[i][4 * g + 2] = h16_lo(u.y); acc[i][4 * g + 3] = h16_hi(u.y);
            }
        }
    }
    layernorm<D, RB>(acc, a.lnkv_g, a.lnkv_b, sPair, a.mean + m0, a.rstd + m0, nvalid);
    acc_to_tile<D, RB>(acc, actA, ALD);
    __syncthreads();
    tile_store_rows<C>(sX, (h16_t*)a.xkv + m0 * D, D, nvalid);
    tile_store_rows<C>(actA, (h16_t*)a.nk + m0 * D, D, nvalid);
    // ---- K | V = normalised . Wkv^T: two D-channel halves, each staged in sX and stored as whole half rows
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        zero<RB>(acc);
        gemm_unit<RB, KS, PD>((const h16_t*)a.Wkv, KS, 0, part * NWV + cw, actA, ALD, tb0, acc, ring);
        if (part == 0) ring_fill<PD>((const h16_t*)a.Wkv, KS, 0, NWV + cw, ring);
        __syncthreads();                                             // the row pass that read sX last is done
        acc_to_tile<D, RB>(acc, sX, ALD);
        __syncthreads();
        tile_store_rows<C>(sX, (h16_t*)a.kv + m0 * (2 * D) + part * D, 2 * D, nvalid);
    }
}
template <int D, int RB, int MINW>
static int adapter_kv_fwd_rows_launch(const VpfAdapterKv& a, hipStream_t st)
{
    using C = Cfg<D, RB, 1>;
    const size_t lds = (size_t)C::TOK * 72 * 2 + (size_t)2 * C::TILE * 2 + (size_t)C::TOK * C::NWV * 8;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)adapter_kv_fwd_rows_kernel<D, RB, 1, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((adapter_kv_fwd_rows_kernel<D, RB, 1, MINW>), dim3((int)vpf_cdiv(a.M, (long)C::TOK)), dim3(C::NT), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
int sa_rows_adapter_kv_fwd_launch(const VpfAdapterKv& a, hipStream_t st)
{
    if (a.D == 384) return vpf_debug().sa_rb == 2 ? adapter_kv_fwd_rows_launch<384, 2, 3>(a, st) : adapter_kv_fwd_rows_launch<384, 1, 6>(a, st);
    if (a.D == 256) return vpf_debug().sa_rb == 2 ? adapter_kv_fwd_rows_launch<256, 2, 4>(a, st) : adapter_kv_fwd_rows_launch<256, 1, 6>(a, st);
    return VPF_ERR_UNSUPPORTED;
}
int sa_rows_adapter_kv_tokens(int D) { return (D == 384 && vpf_debug().sa_rb != 2) ? 32 : 64; }

template <int D, int RB, int MINW>
static int adapter_kv_bwd_rows_launch(const VpfAdapterKvBwd& a, hipStream_t st)
{
    using C = Cfg<D, RB, 1>;
    const size_t lds = (size_t)2 * C::TILE * 2 + (size_t)C::TOK * C::NWV * 8;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)adapter_kv_bwd_rows_kernel<D, RB, 1, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((adapter_kv_bwd_rows_kernel<D, RB, 1, MINW>), dim3((int)vpf_cdiv(a.M, (long)C::TOK)), dim3(C::NT), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
int sa_rows_adapter_kv_bwd_launch(const VpfAdapterKvBwd& a, hipStream_t st)
{
    if (a.D == 384) return vpf_debug().sa_rb == 2 ? adapter_kv_bwd_rows_launch<384, 2, 3>(a, st) : adapter_kv_bwd_rows_launch<384, 1, 3>(a, st);
    if (a.D != 256) return VPF_ERR_UNSUPPORTED;
    using C = Cfg<256, 2, 1>;
    const size_t lds = (size_t)2 * C::TILE * 2 + (size_t)C::TOK * C::NWV * 8;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)adapter_kv_bwd_rows_kernel<256, 2, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((adapter_kv_bwd_rows_kernel<256, 2, 1, 4>), dim3((int)vpf_cdiv(a.M, (long)C::TOK)), dim3(C::NT), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

int sa_rows_ca_front_launch(const VpfCaFront& a, hipStream_t st)
{
    if (a.D != 256 || a.hidden != 128 || a.C < 3) return VPF_ERR_UNSUPPORTED;
    using C = Cfg<256, 1, 2>;                                           // 16 waves x 32 tokens each: 64-token blocks
    const size_t lds = (size_t)C::TOK * 136 * 2 + (size_t)C::TILE * 2 + (size_t)C::TOK * C::NWV * 8 + (size_t)C::NW * 4096;
    static VpfPerDevice attr_dev; bool& attr = attr_dev();
    if (!attr) {
        if (hipFuncSetAttribute((const void*)ca_front_fwd_kernel<256, 1, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return VPF_ERR_HIP;
        attr = true;
    }
    hipLaunchKernelGGL((ca_front_fwd_kernel<256, 1, 2, 4>), dim3(vpf_cdiv(a.M, C::TOK)), dim3(C::NT), lds, st, a);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
