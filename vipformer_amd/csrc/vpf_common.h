// vpf_common.h -- shared host/device helpers for libvipformer_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vipformer_hip.h"

#define VPF_WAVE 64

#define VPF_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return VPF_ERR_HIP;           \
    } while (0)

static inline int vpf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE opt-in: a process-wide "done" flag would leave the second GPU a
// process touches without it and the launch would fail.  One flag per device ordinal.
#define VPF_MAX_DEVICES 64
struct VpfPerDevice {
    bool done[VPF_MAX_DEVICES] = {};
    bool& operator()() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= VPF_MAX_DEVICES) d = 0; return done[d]; }
};

// ------------------------------------------------------------------ launch-time experiment knobs, in ONE place
// Every A/B switch the launchers consult (kernel variant, grid caps, ...).  Filled ONCE per process (api.hip: the VPF_* environment
// variables named there, read when the first launcher asks) and changed only through vpf_debug_set (tests / tools).  No launcher reads
// the environment itself.  The defaults are the measured-fastest parity-green variants.
struct VpfDebug {
    int attn_resident;      // VPF_ATTN_RESIDENT     1: whole-head resident attention kernels when the sequence fits (0: tiled ones)
    int g2e_grid;           // VPF_G2E_GRID          persistent Group2Emb workgroups
    int g2e_w4_grid;        // VPF_G2E_W4_GRID       walkers of the sparse dW4 walk (0: 64, each x 4 k-quarter workgroups; < 0: the round-1 kernel, -value workgroups)
    int wgrad_cfg;          // VPF_WGRAD_CFG         tile configuration of the single weight-gradient GEMM (0: by shape)
    int wgrad_wgs;          // VPF_WGRAD_WGS         its target workgroup count (0: default)
    int gemm_cfg;           // VPF_GEMM_CFG          forced tile configuration of vpf_gemm_bf16 (-1: by shape)
    int wgroup_cfg;         // VPF_WGROUP_CFG        tile configuration of the grouped weight gradient
    int wgroup_wgs;         // VPF_WGROUP_WGS        its target workgroup count (0: default)
    int wgroup_uneven;      // VPF_WGROUP_UNEVEN     K slices of alternating length (2), equal (0)
    int wgroup_dbg;         // VPF_WGROUP_DBG        anatomy switches of tools/microbench.py wgroup (0 in production)
    int fps_exclusive_cu;   // VPF_FPS_EXCLUSIVE_CU  1: a CU of its own per cloud (160 KB LDS request); containment of DESIGN.md section 6
    int knn_select;         // VPF_KNN_SELECT        1: bisection select kernel, 0: K successive extractions
    int sa_nj;              // VPF_SA_NJ             channel blocks per wave of the encoder row-block kernels (1: 8 waves, 2: 4 waves)
    int sa_bwd_rows;        // VPF_SA_BWD_ROWS       1: row-coalesced backward row-block kernels
    int smallk_rpb;         // VPF_SMALLK_RPB        rows per block of the K = 3 front kernels (0: default)
    int sa_wg2;             // VPF_SA_WG2            bit 0 / bit 1: the round-3 forward / backward row-block kernels (sa_rows.hip) also at D = 256 (measured slower in the step: 0)
    int attn_ksplit;        // VPF_ATTN_KSPLIT       key splits per query block in the tiled forward when Lq <= 128 and Lkv >= 512 (1, 2 or 4)
    int attn_ca_merged;     // VPF_ATTN_CA_MERGED    N > 0: dQ / dK / dV of a few-queries-many-keys attention in one kernel when B * H >= N (0: two kernels)
    int attn_rng32;         // VPF_ATTN_RNG32        1: 32-bit dropout group indices in the attention kernels when the score tensor allows it (same masks)
    int sa_bwd_fuse;        // VPF_SA_BWD_FUSE       1: qkv backward of a layer + MLP backward of the layer below as one launch (vpf_sa_layer_bwd_qkv_mlp), 0: two
    int wgroup_xlist;       // VPF_WGROUP_XLIST      1: every (problem, K slice) of a grouped weight gradient on ONE XCD when the slices are few (0: plain order)
    int sa_stagger;         // VPF_SA_STAGGER        decoupled wave groups (sa_rows.hip, Grp<.., DEC>): group g starts g x N x 64 cycles late (0: together)
    int sa_store;           // VPF_SA_STORE          cache policy of the row-block kernels' 16-byte row stores: 0 plain, 1 sc1 (write-through), 2 nt, 3 sc0 sc1
    int sa_tpw;             // VPF_SA_TPW            tokens per workgroup of the sa_rows forward kernels (experiment; 0: the geometry's own, < 0: the grid cut to -N workgroups per CU)
    int wgroup_dma;         // VPF_WGROUP_DMA        N > 0: grouped weight gradients of conforming problems with >= N tokens run the LDS-DMA kernel (0: never)
    int wgroup_dma_tn;      // VPF_WGROUP_DMA_TN     tile columns of the LDS-DMA kernel: 128 = always 256 x 128 (default), 0 = 256 x 256 where every problem allows it (measured slower: 151.8 vs 148.2 us, +0.06 ms per step)
    int wgroup_dma_ramp;    // VPF_WGROUP_DMA_RAMP   LDS-DMA kernel, launches of >= 16 K slices per problem: slice lengths ramp over (1 -+ N / 100) x the mean (0: equal)
    int sa_rb;              // VPF_SA_RB             geometry of those kernels at D = 256: 12 = 16 waves x 32 tokens each (default), 2 = 8 waves x 64, 1 = 8 waves x 32
};
VpfDebug& vpf_debug();

// ------------------------------------------------------------------ the 16-bit MFMA operand type ("h16") -- ONE switch
// Every 16-bit tensor of the library (weights' shadow, activations, gradient operands) and every matrix-core product uses ONE type:
//   VPF_OPERAND_FP16 = 1 (default, round 4): IEEE fp16 -- the reference's own autocast dtype (pretrain.py:154,176): 11 significant
//     bits; the gradients of the backward pass carry GradScaler's loss scale (train.Pretrainer / torch.cuda.amp.GradScaler) to stay
//     inside fp16's range.  tests/rounding_budget.py (profiles/r04_rounding_budget_*): gradient cosine 0.9995 against the fp32 oracle
//     where bf16 operands give 0.994, features 4e-4 instead of 3e-3 (c1, 16 pairs, train mode).
//   VPF_OPERAND_FP16 = 0: bf16 (rounds 1-3), kept for A/B runs of the same kernels (python -m vipformer_amd.build --bf16).
// v_mfma_f32_32x32x16_f16 and _bf16 issue at the same rate; conversions cost the same (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 per pair
// in, one v_cvt_f32_f16 [sdwa] / one shift-or-mask per element out).  Storage is uint16_t either way; vpf_operand_dtype() reports it.
#ifndef VPF_OPERAND_FP16
#define VPF_OPERAND_FP16 1
#endif
typedef uint16_t h16_t;
typedef __attribute__((ext_vector_type(2))) float vpf_f32x2_t;
#if VPF_OPERAND_FP16
typedef _Float16 h16_scalar_t;
#define vpf_mfma32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 h16_scalar_t;
#define vpf_mfma32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(8))) h16_scalar_t h16x8_t;          // one MFMA operand fragment (8 consecutive k per lane)
typedef __attribute__((ext_vector_type(2))) h16_scalar_t h16x2_t;

// the two halves of a packed pair as f32 (element 0 = low half)
__device__ __forceinline__ float h16_lo(uint32_t w)
{
#if VPF_OPERAND_FP16
    return (float)__builtin_bit_cast(h16x2_t, w)[0];
#else
    return __uint_as_float(w << 16);
#endif
}
__device__ __forceinline__ float h16_hi(uint32_t w)
{
#if VPF_OPERAND_FP16
    return (float)__builtin_bit_cast(h16x2_t, w)[1];
#else
    return __uint_as_float(w & 0xffff0000u);
#endif
}
__device__ __forceinline__ float h16_to_f32(h16_t v)
{
#if VPF_OPERAND_FP16
    return (float)__builtin_bit_cast(_Float16, v);
#else
    return __uint_as_float(((uint32_t)v) << 16);
#endif
}
// round-to-nearest-even; NaN stays NaN, a value beyond the type's range becomes inf (what GradScaler's overflow check looks for).
// A plain cast lowers to the hardware v_cvt_pk_{f16,bf16}_f32 on gfx950 (one instruction per PAIR).
__device__ __forceinline__ h16_t f32_to_h16(float f)
{
    const h16_scalar_t b = (h16_scalar_t)f;
    return __builtin_bit_cast(h16_t, b);
}
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi)
{
    const vpf_f32x2_t v = {lo, hi};
    const h16x2_t b = __builtin_convertvector(v, h16x2_t);
    return __builtin_bit_cast(uint32_t, b);
}
// acc += a.lo * b.lo + a.hi * b.hi on the packed pairs (v_dot2c_f32_f16 / v_dot2c_f32_bf16)
__device__ __forceinline__ float h16_dot2(uint32_t a, uint32_t b, float acc)
{
#if VPF_OPERAND_FP16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2_t, a), __builtin_bit_cast(h16x2_t, b), acc, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(h16x2_t, a), __builtin_bit_cast(h16x2_t, b), acc, false);
#endif
}
// the packed pair (1, 1)
#if VPF_OPERAND_FP16
#define VPF_H16_ONES2 0x3c003c00u
#else
#define VPF_H16_ONES2 0x3f803f80u
#endif

// ------------------------------------------------------------------ wave-level reductions (64 lanes)
// DPP (data-parallel primitives) cross-lane moves stay in the VALU: quad permutes, row (16-lane) mirrors,
// then row_bcast15 / row_bcast31 fold the four rows; the total lands in lane 63 and is read back as a
// wave-uniform scalar.  ~6 dependent VALU ops instead of 6 LDS round trips (ds_bpermute).
#define VPF_DPP_QUAD_1032 0xB1
#define VPF_DPP_QUAD_2301 0x4E
#define VPF_DPP_ROW_HALF_MIRROR 0x141
#define VPF_DPP_ROW_MIRROR 0x140
#define VPF_DPP_ROW_BCAST15 0x142
#define VPF_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<VPF_DPP_QUAD_1032, 0xF>(0.f, v);
    v += dpp_f<VPF_DPP_QUAD_2301, 0xF>(0.f, v);
    v += dpp_f<VPF_DPP_ROW_HALF_MIRROR, 0xF>(0.f, v);
    v += dpp_f<VPF_DPP_ROW_MIRROR, 0xF>(0.f, v);
    v += dpp_f<VPF_DPP_ROW_BCAST15, 0xA>(0.f, v);
    v += dpp_f<VPF_DPP_ROW_BCAST31, 0xC>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_f<VPF_DPP_QUAD_1032, 0xF>(v, v));
    v = fmaxf(v, dpp_f<VPF_DPP_QUAD_2301, 0xF>(v, v));
    v = fmaxf(v, dpp_f<VPF_DPP_ROW_HALF_MIRROR, 0xF>(v, v));
    v = fmaxf(v, dpp_f<VPF_DPP_ROW_MIRROR, 0xF>(v, v));
    v = fmaxf(v, dpp_f<VPF_DPP_ROW_BCAST15, 0xA>(v, v));
    v = fmaxf(v, dpp_f<VPF_DPP_ROW_BCAST31, 0xC>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, dpp_u<VPF_DPP_QUAD_1032, 0xF>(v, v));
    v = min(v, dpp_u<VPF_DPP_QUAD_2301, 0xF>(v, v));
    v = min(v, dpp_u<VPF_DPP_ROW_HALF_MIRROR, 0xF>(v, v));
    v = min(v, dpp_u<VPF_DPP_ROW_MIRROR, 0xF>(v, v));
    v = min(v, dpp_u<VPF_DPP_ROW_BCAST15, 0xA>(v, v));
    v = min(v, dpp_u<VPF_DPP_ROW_BCAST31, 0xC>(v, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Single-instruction DPP reduction steps (v_max_f32_dpp / v_min_u32_dpp): the compiler's update_dpp lowering spends a
// copy + a DPP move + the op per step, and these reductions sit on the loop-carried chain of the FPS iteration.
// s_nop covers the VALU-write -> DPP-read (2 wait states) and EXEC-write -> DPP (5) hazards, which the assembler
// does not insert inside inline asm.  The result is valid in lane 63 and read from there.
#define VPF_DPP_REDUCE_ASM(op)                                                                  \
    "s_nop 4\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"          \
    "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"          \
    "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"              \
    "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                   \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                 \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                 \
    "s_nop 1"
// max over aligned groups of W lanes (W = 4, 8 or 16), result in every lane of the group
template <int W>
__device__ __forceinline__ float row_max_f32_asm(float v)
{
    asm volatile("s_nop 4\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    if (W >= 8) asm volatile("v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    if (W >= 16) asm volatile("v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    return v;
}
__device__ __forceinline__ float wave_max_f32_asm(float v)
{
    asm volatile(VPF_DPP_REDUCE_ASM("v_max_f32_dpp") : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32_asm(uint32_t v)
{
    asm volatile(VPF_DPP_REDUCE_ASM("v_min_u32_dpp") : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit keys as (hi, lo) pairs
template <int CTRL, int ROW_MASK, bool MAXOP>
__device__ __forceinline__ void dpp_step_u64(uint32_t& hi, uint32_t& lo)
{
    const uint32_t th = dpp_u<CTRL, ROW_MASK>(hi, hi), tl = dpp_u<CTRL, ROW_MASK>(lo, lo);
    const bool take = MAXOP ? (th > hi || (th == hi && tl > lo)) : (th < hi || (th == hi && tl < lo));
    hi = take ? th : hi; lo = take ? tl : lo;
}
template <bool MAXOP>
__device__ __forceinline__ unsigned long long wave_reduce_u64(unsigned long long v)
{
    uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    dpp_step_u64<VPF_DPP_QUAD_1032, 0xF, MAXOP>(hi, lo);
    dpp_step_u64<VPF_DPP_QUAD_2301, 0xF, MAXOP>(hi, lo);
    dpp_step_u64<VPF_DPP_ROW_HALF_MIRROR, 0xF, MAXOP>(hi, lo);
    dpp_step_u64<VPF_DPP_ROW_MIRROR, 0xF, MAXOP>(hi, lo);
    dpp_step_u64<VPF_DPP_ROW_BCAST15, 0xA, MAXOP>(hi, lo);
    dpp_step_u64<VPF_DPP_ROW_BCAST31, 0xC, MAXOP>(hi, lo);
    hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) { return wave_reduce_u64<true>(v); }
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) { return wave_reduce_u64<false>(v); }

// ------------------------------------------------------------------ counter-based dropout RNG
// rng_state (device memory, 4 x uint32): {seed_lo, seed_hi, step, reserved}.  A keep decision is a
// pure function of (state, site, element index) so backward regenerates forward's mask and a
// captured hipGraph gets fresh masks every replay by bumping `step` on the device.
__device__ __forceinline__ uint32_t vpf_hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// Dropout decisions come in GROUPS OF FOUR consecutive elements: one counter-based hash of (seed, site, step, index >> 2)
// yields 64 bits = four 16-bit uniforms; element e is dropped iff its uniform < thresh (thresh = round(p * 65536), so the
// drop probability is p to within 2^-17).  32-bit integer multiplies are quarter-rate on CDNA, and the epilogues that
// apply dropout own 4 consecutive elements per lane anyway: 4 multiplies per group instead of 5 per element.
struct VpfRng { uint32_t k0, k1; uint32_t thresh; float scale; };
__device__ __forceinline__ VpfRng vpf_rng_init(const uint32_t* rng_state, uint32_t site, float p)
{
    VpfRng r;
    uint32_t s0 = rng_state[0], s1 = rng_state[1], st = rng_state[2];
    r.k0 = vpf_hash32(s0 ^ vpf_hash32(site * 0x9E3779B9u + 0x85ebca6bu));
    r.k1 = vpf_hash32(s1 + st * 0x9E3779B9u + 0xc2b2ae35u);
    const float t = p * 65536.0f + 0.5f;
    r.thresh = p <= 0.f ? 0u : (t >= 65536.0f ? 65536u : (uint32_t)t);
    r.scale = p < 1.f ? 1.0f / (1.0f - p) : 0.f;
    return r;
}
// the four 16-bit uniforms of group g (elements 4g .. 4g+3): .x = e0 | e1 << 16, .y = e2 | e3 << 16
__device__ __forceinline__ uint2 vpf_rand4x16(const VpfRng& r, uint64_t g)
{
    const uint32_t lo = (uint32_t)g, hi = (uint32_t)(g >> 32);
    uint32_t a = vpf_hash32(lo ^ r.k0 ^ ((hi << 16) | (hi >> 16))) + r.k1;
    uint32_t w0 = a * 0x9E3779B1u; w0 ^= w0 >> 15;
    uint32_t w1 = (a ^ 0x85ebca6bu) * 0xC2B2AE3Du; w1 ^= w1 >> 16;
    return make_uint2(w0, w1);
}
// the same four uniforms for a group index that fits 32 bits (hi = 0 in vpf_rand4x16: identical values, no 64-bit index arithmetic)
__device__ __forceinline__ uint2 vpf_rand4x16_32(const VpfRng& r, uint32_t g)
{
    uint32_t a = vpf_hash32(g ^ r.k0) + r.k1;
    uint32_t w0 = a * 0x9E3779B1u; w0 ^= w0 >> 15;
    uint32_t w1 = (a ^ 0x85ebca6bu) * 0xC2B2AE3Du; w1 ^= w1 >> 16;
    return make_uint2(w0, w1);
}
__device__ __forceinline__ uint32_t vpf_keep4_32(const VpfRng& r, uint32_t g)
{
    const uint2 w = vpf_rand4x16_32(r, g);
    return ((w.x & 0xffffu) >= r.thresh ? 1u : 0u) | ((w.x >> 16) >= r.thresh ? 2u : 0u) |
           ((w.y & 0xffffu) >= r.thresh ? 4u : 0u) | ((w.y >> 16) >= r.thresh ? 8u : 0u);
}
// bit e of the result = element 4g + e is KEPT
__device__ __forceinline__ uint32_t vpf_keep4(const VpfRng& r, uint64_t g)
{
    const uint2 w = vpf_rand4x16(r, g);
    return ((w.x & 0xffffu) >= r.thresh ? 1u : 0u) | ((w.x >> 16) >= r.thresh ? 2u : 0u) |
           ((w.y & 0xffffu) >= r.thresh ? 4u : 0u) | ((w.y >> 16) >= r.thresh ? 8u : 0u);
}
__device__ __forceinline__ bool vpf_keep(const VpfRng& r, uint64_t idx)
{
    const uint2 w = vpf_rand4x16(r, idx >> 2);
    const uint32_t word = (idx & 2) ? w.y : w.x;
    return ((idx & 1) ? (word >> 16) : (word & 0xffffu)) >= r.thresh;
}
// keep bits of elements idx0 .. idx0+3 for any alignment of idx0 (one hash when idx0 is a multiple of 4)
__device__ __forceinline__ uint32_t vpf_keep4_at(const VpfRng& r, uint64_t idx0)
{
    if ((idx0 & 3) == 0) return vpf_keep4(r, idx0 >> 2);
    return (vpf_keep(r, idx0) ? 1u : 0u) | (vpf_keep(r, idx0 + 1) ? 2u : 0u) | (vpf_keep(r, idx0 + 2) ? 4u : 0u) | (vpf_keep(r, idx0 + 3) ? 8u : 0u);
}

// 2^x as ONE v_exp_f32 (no denormal-range rescue sequence: the arguments here are score - max <= 0, where an
// underflow to 0 is the right answer; exp2f(-inf) = 0 and exp2f(-inf - -inf) is never formed)
__device__ __forceinline__ float vpf_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26) from one v_rcp_f32 and one v_exp_f32: the libm erff costs ~4x
// as many VALU cycles, and GELU runs over every hidden activation of every MLP, forward and backward.
// Returns erf(z) and exp(-z^2) (the Gaussian factor GELU' needs as well).
__device__ __forceinline__ float vpf_erf_fast(float z, float& gauss)
{
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    gauss = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float y = 1.0f - pl * t * gauss;
    return copysignf(y, z);
}
__device__ __forceinline__ float vpf_gelu(float x)
{
    float g;
    return 0.5f * x * (1.0f + vpf_erf_fast(x * 0.70710678118654752f, g));
}
__device__ __forceinline__ float vpf_gelu_grad(float x)
{
    float g;
    const float cdf = 0.5f * (1.0f + vpf_erf_fast(x * 0.70710678118654752f, g));
    return cdf + x * 0.39894228040143268f * g;      // g = exp(-x^2 / 2)
}
// small.hip: fold of the adapter front's per-workgroup partial parameter gradients (host-side helper shared by two entry points)
int vpf_adapter_front_fold(const float* partial, int nblk, int C, float* dW, float* db, float* dgamma, float* dbeta, void* stream);

