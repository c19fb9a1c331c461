// Minimal reproduction attempt of the fault behind fps_kernel's mis-sampling: packed-fp32 VALU instructions
// (v_pk_add_f32 / v_pk_mul_f32, here with a broadcast second operand as the compiler formed them) in one kernel, LDS-fed MFMAs
// (ds_read_b128 fragments -> v_mfma_f32_32x32x16_bf16) in another, both resident on the same CUs via two streams.  The victim computes
// every squared distance twice -- packed and scalar -- from the same registers and counts disagreements.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/_bin/pkf32_probe tools/pkf32_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ void __launch_bounds__(256) victim(unsigned* bad, int iters)
{
    __shared__ float4 sp[1024];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 256) sp[i] = make_float4(0.001f * ((i * 37 + blockIdx.x) % 977) - 0.4f, 0.001f * ((i * 91) % 811) - 0.3f, 0.001f * ((i * 53) % 911) - 0.2f, 0.f);
    __syncthreads();
    f2 px0 = {sp[4 * t].x, sp[4 * t + 1].x}, py0 = {sp[4 * t].y, sp[4 * t + 1].y}, pz0 = {sp[4 * t].z, sp[4 * t + 1].z};
    f2 px1 = {sp[4 * t + 2].x, sp[4 * t + 3].x}, py1 = {sp[4 * t + 2].y, sp[4 * t + 3].y}, pz1 = {sp[4 * t + 2].z, sp[4 * t + 3].z};
    f2 dist0 = {1e10f, 1e10f}, dist1 = {1e10f, 1e10f};
    float sd[4] = {1e10f, 1e10f, 1e10f, 1e10f};
    unsigned nb = 0;
    int far = (blockIdx.x * 131) & 1023;
    for (int g = 0; g < iters; ++g) {
        const float4 c = sp[far];
        // packed: two points per instruction, the centroid broadcast to both halves
        const f2 cx = {c.x, c.x}, cy = {c.y, c.y}, cz = {c.z, c.z};
        f2 dx = px0 - cx, dy = py0 - cy, dz = pz0 - cz;
        f2 d0 = dx * dx; d0 = d0 + dy * dy; d0 = d0 + dz * dz;
        dx = px1 - cx; dy = py1 - cy; dz = pz1 - cz;
        f2 d1 = dx * dx; d1 = d1 + dy * dy; d1 = d1 + dz * dz;
        dist0.x = d0.x < dist0.x ? d0.x : dist0.x; dist0.y = d0.y < dist0.y ? d0.y : dist0.y;
        dist1.x = d1.x < dist1.x ? d1.x : dist1.x; dist1.y = d1.y < dist1.y ? d1.y : dist1.y;
        // scalar reference from opaque copies (never packed)
        float q[4][3] = {{px0.x, py0.x, pz0.x}, {px0.y, py0.y, pz0.y}, {px1.x, py1.x, pz1.x}, {px1.y, py1.y, pz1.y}};
        const float pk[4] = {d0.x, d0.y, d1.x, d1.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = q[j][0], b = q[j][1], e = q[j][2], ccx = c.x, ccy = c.y, ccz = c.z;
            asm volatile("" : "+v"(a), "+v"(b), "+v"(e), "+v"(ccx), "+v"(ccy), "+v"(ccz));
            const float ux = a - ccx, uy = b - ccy, uz = e - ccz;
            float s = ux * ux; asm volatile("" : "+v"(s)); s = s + uy * uy; asm volatile("" : "+v"(s)); s = s + uz * uz;
            nb += s != pk[j];
            sd[j] = s < sd[j] ? s : sd[j];
        }
        nb += (sd[0] != dist0.x) + (sd[1] != dist0.y) + (sd[2] != dist1.x) + (sd[3] != dist1.y);
        far = (far * 5 + 17 + g) & 1023;
    }
    if (nb) atomicAdd(bad, nb);
}

__global__ void __launch_bounds__(256) aggressor(float* sink, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 128 * 72];
    for (int i = threadIdx.x; i < 2 * 128 * 72; i += 256) lds[i] = 0x3c00 + (i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16_t acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const bf16x8_t*>(lds + ((wave & 1) * 64 + i * 32 + (lane & 31)) * 72 + s * 16 + 8 * (lane >> 5));
                fb[i] = *reinterpret_cast<const bf16x8_t*>(lds + 128 * 72 + ((wave >> 1) * 64 + i * 32 + (lane & 31)) * 72 + s * 16 + 8 * (lane >> 5));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned* bad; float* sink;
    CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&sink, 4));
    hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
    for (int with = 0; with < 2; ++with) {
        CHECK(hipMemset(bad, 0, 4));
        for (int l = 0; l < launches; ++l) {
            if (with) for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(aggressor, dim3(512), dim3(256), 0, s2, sink, 300);
            hipLaunchKernelGGL(victim, dim3(128), dim3(256), 0, s1, bad, 96);
            if ((l & 15) == 15) CHECK(hipDeviceSynchronize());
        }
        CHECK(hipDeviceSynchronize());
        unsigned h; CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("%s: %d launches x 128 workgroups x 96 iterations x 1024 points: %u packed/scalar disagreements\n", with ? "beside LDS-fed MFMAs" : "alone", launches, h);
    }
    return 0;
}
