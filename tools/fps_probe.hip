// Phase timing of the FPS iteration (clock64 stamps in wave 0 of block 0).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vipformer_amd/csrc -I include tools/fps_probe.hip -o /tmp/fps_probe && /tmp/fps_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long g_ph[8];
#define FPS_STAMP_INIT long long ph_[6] = {0, 0, 0, 0, 0, 0}; long long t0_ = clock64(), t1_;
#define FPS_STAMP(i) do { t1_ = clock64(); ph_[i] += t1_ - t0_; t0_ = t1_; } while (0)
#define FPS_STAMP_FINI if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 6; ++i_) g_ph[i_] = ph_[i_]; }
#include "preproc.hip"
int main()
{
    const int B = 128, N = 1024, G = 96;
    std::vector<float> h((size_t)B * N * 3);
    for (auto& v : h) v = (float)rand() / RAND_MAX;
    float* d; int64_t *st, *out;
    hipMalloc(&d, h.size() * 4); hipMalloc(&st, B * 8); hipMalloc(&out, (size_t)B * G * 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(st, 0, B * 8);
    for (int cfg = 0; cfg < 3; ++cfg) {
        for (int r = 0; r < 3; ++r) {
            if (cfg == 0) launch_fps<1024, 4>(d, B, N, 3, st, G, out, 0);
            if (cfg == 1) launch_fps<256, 4>(d, B, N, 3, st, G, out, 0);
            if (cfg == 2) launch_fps<512, 2>(d, B, N, 3, st, G, out, 0);
        }
        hipDeviceSynchronize();
        long long ph[8];
        hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ph), sizeof(ph));
        printf("cfg %d cycles/iter: loop-top %.0f  centroid+dist %.0f  wave-argmax %.0f  slot+barrier %.0f  merge %.0f\n", cfg,
               ph[0] / (double)(G - 1), ph[1] / (double)(G - 1), ph[2] / (double)(G - 1), ph[3] / (double)(G - 1), ph[4] / (double)(G - 1));
    }
    return 0;
}
