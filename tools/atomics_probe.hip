// Split-K flush probe: 512 workgroups x 64 KB of fp32 atomic adds into 32 tiles (agent vs workgroup scope; a tile's adders on one XCD or
// spread), a racy plain read-modify-write for comparison, and HW_REG_XCC_ID of the first workgroups (round-robin dispatch).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/xcc_probe tools/atomics_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__global__ void who(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int SCOPE>
__global__ void __launch_bounds__(256) flush(float* dst, int tiles, int xcd_local)
{
    // 512 workgroups, each adds a 64 KB tile (16384 floats) into one of `tiles` output tiles.
    // xcd_local = 1: all workgroups adding to a tile have ids congruent mod 8 (the same XCD under round-robin dispatch)
    const int wg = blockIdx.x;
    int tile;
    if (xcd_local) { const int x = wg & 7, u = wg >> 3; tile = (u % (tiles / 8)) * 8 + x; }
    else tile = wg % tiles;
    float* p = dst + (size_t)tile * 16384;
#pragma unroll 4
    for (int i = 0; i < 64; ++i) {
        float* q = p + i * 256 + threadIdx.x;
        if (SCOPE == 0) __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else *q += 1.0f;
    }
}
template <typename F> static double time_us(F launch, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); return ms * 1e3 / reps;
}
int main() {
    unsigned* out; CHECK(hipMalloc(&out, 4096 * 4));
    hipLaunchKernelGGL(who, dim3(64), dim3(64), 0, 0, out);
    unsigned h[64]; CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("XCC_ID of workgroups 0..63:"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\n");
    float* dst; CHECK(hipMalloc(&dst, 32 * 16384 * 4));
    for (int local = 0; local < 2; ++local) {
        CHECK(hipMemset(dst, 0, 32 * 16384 * 4));
        double t0 = time_us([&] { hipLaunchKernelGGL((flush<0>), dim3(512), dim3(256), 0, 0, dst, 32, local); }, 20);
        double t1 = time_us([&] { hipLaunchKernelGGL((flush<1>), dim3(512), dim3(256), 0, 0, dst, 32, local); }, 20);
        double t2 = time_us([&] { hipLaunchKernelGGL((flush<2>), dim3(512), dim3(256), 0, 0, dst, 32, local); }, 20);
        printf("flush 512 x 64 KB into 32 tiles (%s): agent-scope atomics %.1f us, workgroup-scope atomics %.1f us, plain RMW (racy) %.1f us\n",
               local ? "same-XCD slices" : "slices spread over XCDs", t0, t1, t2);
    }
    // correctness of workgroup-scope atomics when all adders of a tile share an XCD: 22 launches x 16 adders each
    CHECK(hipMemset(dst, 0, 32 * 16384 * 4));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((flush<1>), dim3(512), dim3(256), 0, 0, dst, 32, 1);
    CHECK(hipDeviceSynchronize());
    float* hd = (float*)malloc(32 * 16384 * 4); CHECK(hipMemcpy(hd, dst, 32 * 16384 * 4, hipMemcpyDeviceToHost));
    long bad = 0; for (long i = 0; i < 32 * 16384; ++i) bad += hd[i] != 160.0f;
    printf("workgroup-scope, same-XCD: %ld of %d elements differ from 160 (first %.1f)\n", bad, 32 * 16384, hd[0]);
    CHECK(hipMemset(dst, 0, 32 * 16384 * 4));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((flush<1>), dim3(512), dim3(256), 0, 0, dst, 32, 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hd, dst, 32 * 16384 * 4, hipMemcpyDeviceToHost));
    bad = 0; for (long i = 0; i < 32 * 16384; ++i) bad += hd[i] != 160.0f;
    printf("workgroup-scope, slices spread over XCDs: %ld of %d elements differ from 160 (first %.1f)\n", bad, 32 * 16384, hd[0]);
    return 0;
}
