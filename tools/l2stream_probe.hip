// How fast can a CU pull L2-resident weights?  Every workgroup (8 waves, one per 32-channel slice as in sa_layer_fwd_kernel) streams the
// same 1 MB "weight" buffer; wave w owns the w-th eighth and reads it in 1 KB wave-instructions (16 B per lane) with DEPTH loads in
// flight, PASSES times.  Reported: bytes per clock per CU (2.4 GHz) for DEPTH = 1 .. 32, for 1 and 2 workgroups per CU, and for the same
// stream issued as LDS-DMA (global_load_lds_dwordx4 into a small LDS ring, nothing consumes it).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/l2stream_probe tools/l2stream_probe.hip ; run: tools/_bin/l2stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int DEPTH>
__global__ void __launch_bounds__(512) stream_regs(const u32x4* __restrict__ w, long slice_vec, int passes, unsigned* sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4* p = w + (long)wave * slice_vec + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < passes; ++it) {
        for (long i = 0; i < slice_vec; i += 64 * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(p + i + 64 * d) ;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int DEPTH>
__global__ void __launch_bounds__(512) stream_regs_plain(const u32x4* __restrict__ w, long slice_vec, int passes, unsigned* sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4* p = w + (long)wave * slice_vec + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < passes; ++it) {
        for (long i = 0; i < slice_vec; i += 64 * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = p[i + 64 * d];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// LDS-DMA: every wave-instruction lands 1 KB in LDS (lane-linear); DEPTH pieces in flight per wave, ring of DEPTH KB per wave
template <int DEPTH>
__global__ void __launch_bounds__(512) stream_ldsdma(const u32x4* __restrict__ w, long slice_vec, int passes, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4* p = w + (long)wave * slice_vec + lane;
    unsigned char* ring = lds + (size_t)wave * DEPTH * 1024;
    for (int it = 0; it < passes; ++it) {
        for (long i = 0; i < slice_vec; i += 64 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(p + i + 64 * d),
                                                 (void __attribute__((address_space(3)))*)(ring + d * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (lds[threadIdx.x] == 0x5a && lds[threadIdx.x + 512] == 0xa5 && passes < 0) sink[0] = 1;
}

// Operand staging of a split-K weight-gradient tile: 256 threads pull `stages` stages of 64 rows x 256 B (one 128-column block of a
// row-major bf16 matrix with `pitch` bytes per row), every byte of the matrix exactly once over the grid.  MODE 0: the same bytes as
// one contiguous 16 KB piece per stage (a column-block-major layout).  Workgroups of one row slice are congruent mod 8 (same XCD).
template <int MODE>
__global__ void __launch_bounds__(256) stage_pattern(const unsigned char* __restrict__ base, long pitch, int ncb, int stages, unsigned* sink)
{
    const int wg = blockIdx.x, t = threadIdx.x;
    const int slice = (wg / (8 * ncb)) * 8 + wg % 8, cb = (wg / 8) % ncb;
    u32x4 acc = {0, 0, 0, 0};
    for (int st = 0; st < stages; ++st) {
        const long r0 = ((long)slice * stages + st) * 64;
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned char* p;
            if (MODE == 1) p = base + (r0 + t / 16 + 16 * j) * pitch + (long)cb * 256 + (t % 16) * 16;
            else p = base + ((long)cb * gridDim.x / ncb * stages * 64 + r0) * 256 + (long)(t + 256 * j) * 16;
            v[j] = *reinterpret_cast<const u32x4*>(p);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// The grouped weight gradient's staging with its real reuse: 4 problems (dW = dY^T X over 12 288 tokens), 32 output tiles of 128 x 128,
// 16 token slices -> 512 workgroups; a workgroup pulls per stage 64 tokens x 256 B of dY (its 128 output rows) and of X (its 128 output
// columns): 201 MB staged, 75 MB unique.  LDS = 0: loads only (8 x 16 B per thread in flight); LDS = 1: + ds_write_b128 into a double
// buffer and one barrier per stage (what gemm_wgrad_group_kernel does around its MFMAs).
struct WgTile { const unsigned char* a; const unsigned char* b; long apitch, bpitch; };
struct WgTable { WgTile t[32]; };
template <int LDS>
__global__ void __launch_bounds__(256) wgrad_staging(WgTable tab, int stages, unsigned* sink)
{
    __shared__ __attribute__((aligned(16))) u32x4 buf[LDS ? 2 * 2048 : 1];
    const int wg = blockIdx.x, t = threadIdx.x;
    const int slice = (wg / 256) * 8 + wg % 8, tile = (wg / 8) % 32;
    const WgTile T = tab.t[tile];
    u32x4 acc = {0, 0, 0, 0};
    for (int st = 0; st < stages; ++st) {
        const long r0 = ((long)slice * stages + st) * 64;
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = *reinterpret_cast<const u32x4*>(T.a + (r0 + t / 16 + 16 * j) * T.apitch + (t % 16) * 16);
            v[4 + j] = *reinterpret_cast<const u32x4*>(T.b + (r0 + t / 16 + 16 * j) * T.bpitch + (t % 16) * 16);
        }
        if (LDS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) buf[(st & 1) * 2048 + j * 256 + t] = v[j];
            __syncthreads();
            acc ^= buf[(st & 1) * 2048 + (t * 7 + st) % 2048];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <typename F>
static double time_us(F launch, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main()
{
    const long bytes = 1 << 20, slice_vec = bytes / 8 / 16;       // 8 slices of 128 KB, in 16-byte vectors
    const int passes = 16;
    u32x4* w; unsigned* sink;
    CHECK(hipMalloc(&w, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(w, 1, bytes)); CHECK(hipMemset(sink, 0, 64));
    const double clk = 2.4e3;                                      // clocks per us
    printf("1 MB of L2-resident weights, 8 waves per workgroup, %d passes; B/clk per CU at 2.4 GHz\n", passes);
    for (int grid : {256, 512, 1024}) {
#define RUN(K, D, LDS) do { \
        double us = time_us([&] { hipLaunchKernelGGL((K<D>), dim3(grid), dim3(512), LDS, 0, w, slice_vec, passes, sink); }, 10); \
        double per_cu = (double)bytes * passes * grid / 256.0; \
        printf("  grid %4d  %-18s depth %2d: %8.1f us  %6.1f B/clk/CU  (%.2f TB/s chip)\n", grid, #K, D, us, per_cu / (us * clk), (double)bytes * passes * grid / us / 1e6); } while (0)
        RUN(stream_regs_plain, 1, 0); RUN(stream_regs_plain, 2, 0); RUN(stream_regs_plain, 4, 0); RUN(stream_regs_plain, 8, 0);
        RUN(stream_regs_plain, 16, 0); RUN(stream_regs_plain, 32, 0);
        RUN(stream_regs, 8, 0);
        RUN(stream_ldsdma, 2, 8 * 2 * 1024); RUN(stream_ldsdma, 4, 8 * 4 * 1024); RUN(stream_ldsdma, 8, 8 * 8 * 1024);
    }
    {
        // 6 column blocks (a [rows][768] bf16 matrix, 1536-byte rows) and 2 (a [rows][256] matrix, 512-byte rows)
        for (int ncb : {6, 2}) {
            const int grid = 8 * ncb * (ncb == 6 ? 10 : 32);                 // 480 / 512 workgroups
            for (int stages : {3, 24, 96}) {
                const long rows = (long)grid / ncb * stages * 64, pitch = 256L * ncb, total = rows * pitch;
                unsigned char* m;
                CHECK(hipMalloc(&m, total)); CHECK(hipMemset(m, 1, total));
                for (int mode = 0; mode < 2; ++mode) {
                    double us = mode ? time_us([&] { hipLaunchKernelGGL((stage_pattern<1>), dim3(grid), dim3(256), 0, 0, m, pitch, ncb, stages, sink); }, 10)
                                     : time_us([&] { hipLaunchKernelGGL((stage_pattern<0>), dim3(grid), dim3(256), 0, 0, m, pitch, ncb, stages, sink); }, 10);
                    printf("  staging %s, %d column blocks, %5.1f MB read once by %d workgroups: %8.1f us  %5.2f TB/s\n",
                           mode ? "256-B pieces of pitched rows" : "contiguous 16 KB pieces     ", ncb, total / 1e6, grid, us, total / us / 1e6);
                }
                CHECK(hipFree(m));
            }
        }
    }
    {
        const long M = 12288;
        const int nout[4] = {256, 512, 256, 768}, kin[4] = {512, 256, 256, 256};
        WgTable tab; int nt = 0; double unique = 0;
        for (int p = 0; p < 4; ++p) {
            unsigned char *dy, *x;
            CHECK(hipMalloc(&dy, M * nout[p] * 2)); CHECK(hipMalloc(&x, M * kin[p] * 2));
            CHECK(hipMemset(dy, 1, M * nout[p] * 2)); CHECK(hipMemset(x, 1, M * kin[p] * 2));
            unique += (double)M * (nout[p] + kin[p]) * 2;
            for (int i = 0; i < nout[p] / 128; ++i)
                for (int j = 0; j < kin[p] / 128; ++j) tab.t[nt++] = WgTile{dy + i * 256, x + j * 256, nout[p] * 2L, kin[p] * 2L};
        }
        const int stages = 12;
        const double staged = 512.0 * stages * 32768;
        double us0 = time_us([&] { hipLaunchKernelGGL((wgrad_staging<0>), dim3(512), dim3(256), 0, 0, tab, stages, sink); }, 20);
        double us1 = time_us([&] { hipLaunchKernelGGL((wgrad_staging<1>), dim3(512), dim3(256), 0, 0, tab, stages, sink); }, 20);
        printf("  grouped-wgrad staging (%d tiles x 16 slices, %.0f MB staged, %.0f MB unique): loads only %.1f us (%.2f TB/s staged); + LDS store + barrier %.1f us (%.2f TB/s)\n",
               nt, staged / 1e6, unique / 1e6, us0, staged / us0 / 1e6, us1, staged / us1 / 1e6);
    }
    CHECK(hipDeviceSynchronize());
    printf("ok\n");
    return 0;
}
