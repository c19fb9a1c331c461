// Does a hipGraph replay keep the CU mask of the stream a kernel was captured on?  (round 6 question: could the two branches of the step
// be given disjoint CU sets so that co-resident kernels stop slowing each other down 1.3 - 3x?)
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/_bin/cumask_probe && tools/_bin/cumask_probe
// Launches a 4096-workgroup kernel that records (XCC_ID, HW_ID) per workgroup: (a) on a plain stream, (b) eagerly on a stream created with
// hipExtStreamCreateWithCUMask (first half of the mask bits), (c) captured on that stream and replayed as a graph on it, (d) the same graph
// replayed on a plain stream.  Prints the number of distinct (xcc, se, cu) triples seen in each case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void where(unsigned* out)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // some work so that workgroups spread over the machine
    float v = threadIdx.x;
    for (int i = 0; i < 2000; ++i) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
    if (v == 12345.f) out[0] = 0;
}
static int distinct(const std::vector<unsigned>& h)
{
    std::set<unsigned> s;
    for (size_t i = 0; i < h.size(); i += 2) {
        const unsigned hw = h[i + 1], cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID layout
        s.insert((h[i] << 16) | (se << 8) | (sh << 4) | cu);
    }
    return (int)s.size();
}
int main()
{
    const int NB = 4096;
    unsigned* d; CK(hipMalloc(&d, NB * 2 * sizeof(unsigned)));
    std::vector<unsigned> h(NB * 2);
    hipStream_t plain, masked;
    CK(hipStreamCreate(&plain));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> mask(words, 0);
    for (int i = 0; i < ncu / 2; ++i) mask[i / 32] |= 1u << (i % 32);
    CK(hipExtStreamCreateWithCUMask(&masked, words, mask.data()));
    auto run = [&](const char* tag, auto fn) -> int {
        CK(hipMemset(d, 0, NB * 2 * sizeof(unsigned)));
        if (fn()) return 1;
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, NB * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
        printf("%-58s %3d distinct CUs of %d\n", tag, distinct(h), ncu);
        return 0;
    };
    if (run("(a) eager, plain stream", [&]() -> int { hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, plain, d); return 0; })) return 1;
    if (run("(b) eager, stream with half the CUs masked in", [&]() -> int { hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, masked, d); return 0; })) return 1;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(masked, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, masked, d);
    CK(hipStreamEndCapture(masked, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    if (run("(c) captured on the masked stream, replayed on it", [&]() -> int { CK(hipGraphLaunch(ge, masked)); return 0; })) return 1;
    if (run("(d) the same graph replayed on the plain stream", [&]() -> int { CK(hipGraphLaunch(ge, plain)); return 0; })) return 1;
    // (e) two branches: fork from the launch stream to a masked side stream inside the capture
    hipStream_t side; CK(hipExtStreamCreateWithCUMask(&side, words, mask.data()));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(plain, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(e0, plain)); CK(hipStreamWaitEvent(side, e0, 0));
    hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, side, d);
    CK(hipEventRecord(e1, side)); CK(hipStreamWaitEvent(plain, e1, 0));
    CK(hipStreamEndCapture(plain, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    if (run("(e) forked to a masked side stream in the capture, replayed", [&]() -> int { CK(hipGraphLaunch(ge2, plain)); return 0; })) return 1;
    return 0;
}
