// augment.hip -- the DataLoader-worker augmentations of the reference moved onto the GPU: once the step takes
// 4.6 ms, 64 pairs x (2 x trans_1 on a CPU core + a 38.5 MB fp32 image batch over PCIe) per step is what bounds a real run.
//   * vpf_augment_points: trans_1 / trans_2 of datasets/data.py:16-36 -- PointcloudNormalize, PointcloudScale(0.5, 2),
//     PointcloudRotate (about y), PointcloudTranslate(0.5), PointcloudJitter(0.01, clip 0.05), PointcloudRandomInputDropout(0.875)
//     (datasets/data_utils.py:56-221) -- one workgroup per cloud, the cloud in registers, every reduction (centroid, radius,
//     bounding box) a wave DPP reduction + one LDS exchange.  The random draws come from the library's counter-based hash, NOT from
//     numpy / torch: the per-cloud draws are exported so the deterministic part can be replayed exactly; jitter and dropout are
//     statistically, not bitwise, those of the reference.
//   * vpf_image_u8_normalize: ToTensor + Normalize(mean, std) (+ RandomHorizontalFlip) of utils.py:21-25 on a uint8 HWC batch:
//     the host ships 1 byte per sample instead of 4 (9.6 MB instead of 38.5 MB per 64 images); resize and colour jitter stay
//     with the decoder on the host.
#include "vpf_common.h"

#define AUG_MAX_PPT 16      // points per thread (256 threads): clouds of up to 4096 points

__device__ __forceinline__ float aug_uniform(uint32_t k0, uint32_t k1, uint32_t idx)
{
    uint32_t h = vpf_hash32(idx * 0x9E3779B9u + k0);
    h = vpf_hash32(h ^ k1);
    return (float)(h >> 8) * (1.0f / 16777216.0f);              // [0, 1)
}
__device__ __forceinline__ float block_sum(float v, float* sh)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh)
{
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return r;
}

// params_out [B, 8] = {scale, angle, tx, ty, tz (unit draws in [-0.5, 0.5) BEFORE the bounding-box factor), dropout ratio, radius, 0}
__global__ void __launch_bounds__(256) augment_points_kernel(const float* __restrict__ in, int N, int C, const uint32_t* __restrict__ rng_state,
                                                            uint32_t site, float* __restrict__ out, float* __restrict__ params_out)
{
    __shared__ float sh[4];
    __shared__ float p0[3];
    const int b = blockIdx.x;
    const uint32_t s0 = rng_state[0], s1 = rng_state[1], st = rng_state[2];
    const uint32_t k0 = vpf_hash32(s0 ^ vpf_hash32(site * 0x9E3779B9u + 0x85ebca6bu) ^ vpf_hash32((uint32_t)b + 0x27d4eb2fu));
    const uint32_t k1 = vpf_hash32(s1 + st * 0x9E3779B9u + 0xc2b2ae35u);
    const float* src = in + (size_t)b * N * C;
    float x[AUG_MAX_PPT], y[AUG_MAX_PPT], z[AUG_MAX_PPT];
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int i = 0; i < AUG_MAX_PPT; ++i) {
        const int n = threadIdx.x + 256 * i;
        if (n < N) { x[i] = src[(size_t)n * C]; y[i] = src[(size_t)n * C + 1]; z[i] = src[(size_t)n * C + 2]; sx += x[i]; sy += y[i]; sz += z[i]; }
        else { x[i] = y[i] = z[i] = 0.f; }
    }
    // PointcloudNormalize (data_utils.py:206-221): centre, divide by the largest norm
    const float cx = block_sum(sx, sh) / (float)N, cy = block_sum(sy, sh) / (float)N, cz = block_sum(sz, sh) / (float)N;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < AUG_MAX_PPT; ++i) {
        const int n = threadIdx.x + 256 * i;
        x[i] -= cx; y[i] -= cy; z[i] -= cz;
        if (n < N) m2 = fmaxf(m2, x[i] * x[i] + y[i] * y[i] + z[i] * z[i]);
    }
    const float radius = sqrtf(block_max(m2, sh));
    // the per-cloud draws (indices 0..7 of this cloud's stream; per-point draws start at 16)
    const float scale = 0.5f + 1.5f * aug_uniform(k0, k1, 0);                       // PointcloudScale(lo=0.5, hi=2)   :56-67
    const float angle = aug_uniform(k0, k1, 1) * 6.283185307179586f;                // PointcloudRotate about y        :70-101
    const float t0 = aug_uniform(k0, k1, 2) - 0.5f, t1 = aug_uniform(k0, k1, 3) - 0.5f, t2 = aug_uniform(k0, k1, 4) - 0.5f;   // Translate(0.5) :157-173
    const float ratio = aug_uniform(k0, k1, 5) * 0.875f;                            // RandomInputDropout(0.875)       :181-199
    const float f = scale / radius, cs = cosf(angle), sn = sinf(angle);
    float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < AUG_MAX_PPT; ++i) {
        const int n = threadIdx.x + 256 * i;
        const float px = x[i] * f, py = y[i] * f, pz = z[i] * f;
        x[i] = cs * px + sn * pz; y[i] = py; z[i] = -sn * px + cs * pz;            // points @ R^T, R = angle_axis(angle, y) :6-34
        if (n < N) {
            lo0 = fminf(lo0, x[i]); hi0 = fmaxf(hi0, x[i]); lo1 = fminf(lo1, y[i]); hi1 = fmaxf(hi1, y[i]);
            lo2 = fminf(lo2, z[i]); hi2 = fmaxf(hi2, z[i]);
        }
    }
    const float d0 = block_max(hi0, sh) + block_max(-lo0, sh), d1 = block_max(hi1, sh) + block_max(-lo1, sh), d2 = block_max(hi2, sh) + block_max(-lo2, sh);
    const float tx = t0 * d0, ty = t1 * d1, tz = t2 * d2;
#pragma unroll
    for (int i = 0; i < AUG_MAX_PPT; ++i) {
        const int n = threadIdx.x + 256 * i;
        if (n < N) {
            // PointcloudJitter: N(0, 0.01) clamped to +-0.05 (Box-Muller on two uniforms per coordinate pair)
            const float u1 = fmaxf(aug_uniform(k0, k1, 16 + 4 * n), 1e-7f), u2 = aug_uniform(k0, k1, 17 + 4 * n);
            const float u3 = fmaxf(aug_uniform(k0, k1, 18 + 4 * n), 1e-7f), u4 = aug_uniform(k0, k1, 19 + 4 * n);
            const float r1 = 0.01f * sqrtf(-2.f * __logf(u1)), r2 = 0.01f * sqrtf(-2.f * __logf(u3));
            x[i] += tx + fminf(fmaxf(r1 * __cosf(6.283185307f * u2), -0.05f), 0.05f);
            y[i] += ty + fminf(fmaxf(r1 * __sinf(6.283185307f * u2), -0.05f), 0.05f);
            z[i] += tz + fminf(fmaxf(r2 * __cosf(6.283185307f * u4), -0.05f), 0.05f);
        }
    }
    if (threadIdx.x == 0) { p0[0] = x[0]; p0[1] = y[0]; p0[2] = z[0]; }
    __syncthreads();
    float* dst = out + (size_t)b * N * 3;
#pragma unroll
    for (int i = 0; i < AUG_MAX_PPT; ++i) {
        const int n = threadIdx.x + 256 * i;
        if (n < N) {
            const bool drop = aug_uniform(k0, k1, 0x40000000u + n) <= ratio;          // pc[drop_idx] = pc[0]
            dst[(size_t)n * 3] = drop ? p0[0] : x[i]; dst[(size_t)n * 3 + 1] = drop ? p0[1] : y[i]; dst[(size_t)n * 3 + 2] = drop ? p0[2] : z[i];
        }
    }
    if (params_out && threadIdx.x == 0) {
        float* q = params_out + (size_t)b * 8;
        q[0] = scale; q[1] = angle; q[2] = t0; q[3] = t1; q[4] = t2; q[5] = ratio; q[6] = radius; q[7] = 0.f;
    }
}
extern "C" int vpf_augment_points(const float* pts, int B, int N, int C, const uint32_t* rng_state, uint32_t site, float* out,
                                  float* params_out, void* stream)
{
    (void)hipGetLastError();
    if (!pts || !rng_state || !out) return VPF_ERR_NULL;
    if (B < 0 || N <= 0 || N > 256 * AUG_MAX_PPT || C < 3) return VPF_ERR_BADSHAPE;
    if (B == 0) return VPF_OK;
    hipLaunchKernelGGL(augment_points_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pts, N, C, rng_state, site, out, params_out);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// uint8 [B,H,W,3] (as decoded / resized / colour-jittered on the host) -> float [B,3,H,W]: x / 255, (x - mean) / std, horizontal flip
// of the images whose draw is < p_flip (transforms.RandomHorizontalFlip, utils.py:23); flips_out (nullable) u8 [B] reports them.
__global__ void image_u8_normalize_kernel(const uint8_t* __restrict__ img, int B, int H, int W, float m0, float m1, float m2, float i0,
                                          float i1, float i2, const uint32_t* __restrict__ rng_state, uint32_t site, float p_flip,
                                          float* __restrict__ out, uint8_t* __restrict__ flips_out)
{
    const long total = (long)B * H * W;
    uint32_t k0 = 0, k1 = 0;
    if (rng_state) {
        k0 = vpf_hash32(rng_state[0] ^ vpf_hash32(site * 0x9E3779B9u + 0x85ebca6bu));
        k1 = vpf_hash32(rng_state[1] + rng_state[2] * 0x9E3779B9u + 0xc2b2ae35u);
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W); const long t = i / W; const int h = (int)(t % H); const int b = (int)(t / H);
        const bool flip = rng_state && p_flip > 0.f && aug_uniform(k0, k1, (uint32_t)b) < p_flip;
        const uint8_t* s = img + (((size_t)b * H + h) * W + (flip ? W - 1 - w : w)) * 3;
        const size_t plane = (size_t)H * W, o = (size_t)b * 3 * plane + (size_t)h * W + w;
        out[o] = ((float)s[0] * (1.f / 255.f) - m0) * i0;
        out[o + plane] = ((float)s[1] * (1.f / 255.f) - m1) * i1;
        out[o + 2 * plane] = ((float)s[2] * (1.f / 255.f) - m2) * i2;
        if (flips_out && h == 0 && w == 0) flips_out[b] = flip ? 1 : 0;
    }
}
extern "C" int vpf_image_u8_normalize(const void* img_u8, int B, int H, int W, const float* mean3_host, const float* std3_host,
                                      const uint32_t* rng_state, uint32_t site, float p_flip, float* out, void* flips_out, void* stream)
{
    (void)hipGetLastError();
    if (!img_u8 || !mean3_host || !std3_host || !out) return VPF_ERR_NULL;
    if (B < 0 || H <= 0 || W <= 0) return VPF_ERR_BADSHAPE;
    if (B == 0) return VPF_OK;
    const long total = (long)B * H * W;
    int grid = vpf_cdiv(total, 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(image_u8_normalize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img_u8, B, H, W, mean3_host[0],
                       mean3_host[1], mean3_host[2], 1.f / std3_host[0], 1.f / std3_host[1], 1.f / std3_host[2], rng_state, site, p_flip, out,
                       (uint8_t*)flips_out);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
