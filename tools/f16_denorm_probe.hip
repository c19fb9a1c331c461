// does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  A = subnormal 2^-20 everywhere, B = 1024.0: exact result per element = 16 * 2^-10
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
__global__ void k(float* o, uint16_t abits, uint16_t bbits)
{
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, abits); b[i] = __builtin_bit_cast(_Float16, bbits); }
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) o[0] = acc[0];
}
int main()
{
    float* d; hipMalloc(&d, 4); float h;
    // 2^-20 as fp16 subnormal: value = m * 2^-24, m = 16 -> bits 0x0010 ; 1024.0 = 0x6400
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (uint16_t)0x0010, (uint16_t)0x6400);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("subnormal input 2^-20 x 1024 x16 terms = %g (exact %g; 0 means MFMA flushes fp16 denormal inputs)\n", h, 16.0 * 1024.0 / 1048576.0);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (uint16_t)0x0001, (uint16_t)0x7bff);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("smallest subnormal 2^-24 x 65504 x16 = %g (exact %g)\n", h, 16.0 * 65504.0 / 16777216.0);
    return 0;
}
