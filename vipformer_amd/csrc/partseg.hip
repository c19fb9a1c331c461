// partseg.hip -- the pieces of CrossFormer_partseg (partseg.py:345-470) and PointNetFeaturePropagation (utils.py:192-242) that the
// pre-training kernels do not already cover: LayerNorm of the encoder taps into one concatenated feature, the 3-NN inverse-distance
// interpolation rows (operand of the propagation MLP's first 1x1 convolution) and their backward scatter, zero-padded operand
// copies for the GEMM's 8-element alignment, and the label-smoothed cross entropy of ft_partseg.py:128.
// (The 3-NN search itself lives in preproc.hip next to the exact square_distance recipe it shares.)
#include "vpf_common.h"

// ------------------------------------------------------------------ zero-padded h16 operand copy
// dst h16 [rows_out, Kp] = src [rows, K] (f32 or h16, row stride ld) in the top-left corner, zeros elsewhere.
template <typename T>
__global__ void pad_h16_kernel(const T* __restrict__ src, long rows, int K, long ld, long rows_out, int Kp, h16_t* __restrict__ dst)
{
    const long total = rows_out * (long)Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / Kp; const int c = (int)(i % Kp);
        float v = 0.f;
        if (r < rows && c < K) {
            if constexpr (sizeof(T) == 2) v = h16_to_f32(src[r * ld + c]); else v = src[r * ld + c];
        }
        dst[i] = f32_to_h16(v);
    }
}
extern "C" int vpf_pad_h16(const void* src, int src_is_h16, long rows, int K, long ld, long rows_out, int Kp, void* dst_h16, void* stream)
{
    (void)hipGetLastError();
    if (!src || !dst_h16) return VPF_ERR_NULL;
    if (rows < 0 || K <= 0 || rows_out < rows || Kp < K || ld < K) return VPF_ERR_BADSHAPE;
    if (rows_out == 0) return VPF_OK;
    const long total = rows_out * (long)Kp;
    int grid = vpf_cdiv(total, 256); if (grid > 4096) grid = 4096;
    if (src_is_h16) hipLaunchKernelGGL(pad_h16_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const h16_t*)src, rows, K, ld, rows_out, Kp, (h16_t*)dst_h16);
    else hipLaunchKernelGGL(pad_h16_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)src, rows, K, ld, rows_out, Kp, (h16_t*)dst_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// ------------------------------------------------------------------ LayerNorm of the encoder taps -> concatenated feature
// partseg.py:427-435: feature_list = [self.norm(x) for x in taps]; x = cat(feature_list, channel).  One wave per (row, tap):
// xcat f32 [rows, nl * D] (channel block t = tap t), mean / rstd f32 [nl, rows].  The SAME LayerNorm parameters serve every tap.
struct TapPtrs { const float* x[4]; };
struct TapOut { float* d[4]; };
__global__ void __launch_bounds__(256) ln_taps_fwd_kernel(TapPtrs taps, int nl, long rows, int D, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float* __restrict__ xcat,
                                                         float* __restrict__ mean, float* __restrict__ rstd)
{
    const int lane = threadIdx.x & 63;
    const long wv = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    if (wv >= rows * nl) return;
    const int t = (int)(wv / rows); const long r = wv % rows;
    const float* x = taps.x[t] + r * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += x[c];
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = x[c] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    float* y = xcat + r * (long)nl * D + (long)t * D;
    for (int c = lane; c < D; c += 64) y[c] = (x[c] - mu) * rs * gamma[c] + beta[c];
    if (lane == 0) { mean[(long)t * rows + r] = mu; rstd[(long)t * rows + r] = rs; }
}
extern "C" int vpf_ln_taps_fwd(const float* x0, const float* x1, const float* x2, const float* x3, int nl, long rows, int D,
                               const float* gamma, const float* beta, float eps, float* xcat, float* mean, float* rstd, void* stream)
{
    (void)hipGetLastError();
    if (!x0 || !gamma || !beta || !xcat || !mean || !rstd) return VPF_ERR_NULL;
    if (nl < 1 || nl > 4 || rows < 0 || D <= 0) return VPF_ERR_BADSHAPE;
    TapPtrs tp; tp.x[0] = x0; tp.x[1] = x1; tp.x[2] = x2; tp.x[3] = x3;
    for (int t = 0; t < nl; ++t) if (!tp.x[t]) return VPF_ERR_NULL;
    if (rows == 0) return VPF_OK;
    hipLaunchKernelGGL(ln_taps_fwd_kernel, dim3(vpf_cdiv(rows * nl, 4)), dim3(256), 0, (hipStream_t)stream, tp, nl, rows, D, gamma, beta, eps, xcat, mean, rstd);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// backward: d tap_t [rows, D] = LN'(dxcat[:, t*D:(t+1)*D]);  dgamma / dbeta += over rows AND taps.  Each workgroup (4 waves) walks
// a strided set of (row, tap) items, keeps per-lane partial parameter gradients in registers (D <= 512: 8 per lane), folds its four
// waves through LDS and adds ONE partial per channel with an atomic (grid <= 256 workgroups, so <= 256 adds per address).
__global__ void __launch_bounds__(256) ln_taps_bwd_kernel(const float* __restrict__ dxcat, TapPtrs taps, int nl, long rows, int D,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, TapOut outs, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta)
{
    __shared__ float sg[4][512], sb[4][512];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float pg[8], pb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pg[k] = pb[k] = 0.f;
    const long items = rows * nl;
    for (long it = blockIdx.x * 4L + wid; it < items; it += gridDim.x * 4L) {
        const int t = (int)(it / rows); const long r = it % rows;
        const float* x = taps.x[t] + r * D;
        const float* dy = dxcat + r * (long)nl * D + (long)t * D;
        const float mu = mean[(long)t * rows + r], rs = rstd[(long)t * rows + r];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane + 64 * k;
            if (c < D) {
                const float xh = (x[c] - mu) * rs, g = dy[c] * gamma[c];
                s1 += g; s2 += g * xh;
                pg[k] += dy[c] * xh; pb[k] += dy[c];
            }
        }
        s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
        float* dx = outs.d[t] + r * D;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane + 64 * k;
            if (c < D) { const float xh = (x[c] - mu) * rs; dx[c] = rs * (dy[c] * gamma[c] - s1 - xh * s2); }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { sg[wid][lane + 64 * k] = pg[k]; sb[wid][lane + 64 * k] = pb[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(dgamma + c, (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]));
        atomicAdd(dbeta + c, (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]));
    }
}
extern "C" int vpf_ln_taps_bwd(const float* dxcat, const float* x0, const float* x1, const float* x2, const float* x3, int nl, long rows,
                               int D, const float* mean, const float* rstd, const float* gamma, float* d0, float* d1, float* d2,
                               float* d3, float* dgamma, float* dbeta, void* stream)
{
    (void)hipGetLastError();
    if (!dxcat || !mean || !rstd || !gamma || !dgamma || !dbeta) return VPF_ERR_NULL;
    if (nl < 1 || nl > 4 || rows < 0 || D <= 0 || D > 512) return VPF_ERR_BADSHAPE;
    TapPtrs tp; tp.x[0] = x0; tp.x[1] = x1; tp.x[2] = x2; tp.x[3] = x3;
    TapOut to; to.d[0] = d0; to.d[1] = d1; to.d[2] = d2; to.d[3] = d3;
    for (int t = 0; t < nl; ++t) if (!tp.x[t] || !to.d[t]) return VPF_ERR_NULL;
    if (rows == 0) return VPF_OK;
    int grid = vpf_cdiv(rows * nl, 4); if (grid > 256) grid = 256;
    hipLaunchKernelGGL(ln_taps_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dxcat, tp, nl, rows, D, mean, rstd, gamma, to, dgamma, dbeta);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// ------------------------------------------------------------------ interpolation rows (utils.py:230-236)
// A[b*N + n, :] = [ points1 = xyz[b, n, 0:C]  |  sum_k w[n, k] * feat[b, idx[n, k], 0:F]  |  zero pad to Kp ]   (h16): the operand
// of mlp_convs[0] -- cat([points1, interpolated_points]) -- written once, directly in the GEMM's layout.  One wave per point; the
// three source rows are read as whole contiguous rows (F * 4 bytes each), products and sum in the reference's order.
__global__ void __launch_bounds__(256) interp_rows_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ xyz, int C,
                                                             const int* __restrict__ idx, const float* __restrict__ w, int N, int S,
                                                             int F, int Kp, long rows, h16_t* __restrict__ A)
{
    const int lane = threadIdx.x & 63;
    const long r = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    if (r >= rows) return;
    const long b = r / N;
    const int i0 = idx[r * 3], i1 = idx[r * 3 + 1], i2 = idx[r * 3 + 2];
    const float w0 = w[r * 3], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];
    const float* f0 = feat + (b * S + i0) * (long)F;
    const float* f1 = feat + (b * S + i1) * (long)F;
    const float* f2 = feat + (b * S + i2) * (long)F;
    h16_t* a = A + r * (long)Kp;
    if (lane < C) a[lane] = f32_to_h16(xyz[r * C + lane]);
    for (int c = lane; c < F; c += 64) {
        float v = f0[c] * w0 + f1[c] * w1;
        v = v + f2[c] * w2;
        a[C + c] = f32_to_h16(v);
    }
    for (int c = C + F + lane; c < Kp; c += 64) a[c] = 0;
}
extern "C" int vpf_interp_rows_fwd(const float* feat, const float* xyz, int B, int N, int C, int S, int F, const int* idx,
                                   const float* weight, int Kp, void* A_h16, void* stream)
{
    (void)hipGetLastError();
    if (!feat || !xyz || !idx || !weight || !A_h16) return VPF_ERR_NULL;
    if (B < 0 || N < 0 || S <= 0 || F <= 0 || C < 0 || C > 64 || Kp < C + F) return VPF_ERR_BADSHAPE;
    const long rows = (long)B * N;
    if (rows == 0) return VPF_OK;
    hipLaunchKernelGGL(interp_rows_fwd_kernel, dim3(vpf_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, feat, xyz, C, idx, weight, N, S, F, Kp,
                       rows, (h16_t*)A_h16);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
// backward: dfeat[b, idx[n, k], :] += w[n, k] * dA[b*N + n, C : C + F]   (fp32 atomics into a zeroed buffer: a centre collects from
// every point that has it among its three nearest, ~3N/S points on average)
__global__ void __launch_bounds__(256) interp_rows_bwd_kernel(const h16_t* __restrict__ dA, int C, const int* __restrict__ idx,
                                                             const float* __restrict__ w, int N, int S, int F, int Kp, long rows,
                                                             float* __restrict__ dfeat)
{
    const int lane = threadIdx.x & 63;
    const long r = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    if (r >= rows) return;
    const long b = r / N;
    const h16_t* a = dA + r * (long)Kp + C;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float wk = w[r * 3 + k];
        float* d = dfeat + (b * S + idx[r * 3 + k]) * (long)F;
        for (int c = lane; c < F; c += 64) atomicAdd(d + c, wk * h16_to_f32(a[c]));
    }
}
extern "C" int vpf_interp_rows_bwd(const void* dA_h16, int B, int N, int C, int S, int F, const int* idx, const float* weight, int Kp,
                                   float* dfeat, void* stream)
{
    (void)hipGetLastError();
    if (!dA_h16 || !idx || !weight || !dfeat) return VPF_ERR_NULL;
    if (B < 0 || N < 0 || S <= 0 || F <= 0 || C < 0 || Kp < C + F) return VPF_ERR_BADSHAPE;
    const long rows = (long)B * N;
    if (rows == 0) return VPF_OK;
    hipLaunchKernelGGL(interp_rows_bwd_kernel, dim3(vpf_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)dA_h16, C, idx, weight, N,
                       S, F, Kp, rows, dfeat);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}

// ------------------------------------------------------------------ cross entropy with label smoothing (ft_partseg.py:128,155)
// torch.nn.CrossEntropyLoss(label_smoothing = eps), mean reduction: loss_i = (1 - eps) * (lse - z[y]) + eps / C * sum_c (lse - z[c]);
// dz = (softmax - ((1 - eps) * onehot + eps / C)) / rows.  logits f32 [rows, ld] (the first C columns count), one wave per row;
// per-workgroup partial losses are folded by the last launch in a fixed order.
__global__ void __launch_bounds__(256) ce_smooth_kernel(const float* __restrict__ z, long ld, const long long* __restrict__ y, long rows, int C,
                                                       float eps, float* __restrict__ partial, float* __restrict__ dz, long lddz)
{
    __shared__ float sp[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    for (long r = blockIdx.x * 4L + wid; r < rows; r += gridDim.x * 4L) {
        const float* zr = z + r * ld;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, zr[c]);
        m = wave_max(m);
        float s = 0.f, sum = 0.f;
        for (int c = lane; c < C; c += 64) { s += __expf(zr[c] - m); sum += zr[c]; }
        s = wave_sum(s); sum = wave_sum(sum);
        const float lse = m + __logf(s);
        const long long yl = y[r];
        const bool bad = yl < 0 || yl >= (long long)C;      // torch.nn.CrossEntropyLoss raises on such a target (ft_partseg.py never
        const int yy = bad ? 0 : (int)yl;                   // passes ignore_index); a kernel cannot: the loss and the row's gradient
        const float zy = bad ? __builtin_nanf("") : zr[yy]; // become NaN instead of an out-of-bounds read
        acc += (1.f - eps) * (lse - zy) + eps * (lse - sum / (float)C);
        if (dz) {
            float* d = dz + r * lddz;
            const float inv = 1.f / (float)rows;
            for (int c = lane; c < C; c += 64)
                d[c] = bad ? __builtin_nanf("") : (__expf(zr[c] - lse) - ((c == yy ? 1.f - eps : 0.f) + eps / (float)C)) * inv;
        }
    }
    if (lane == 0) sp[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sp[0] + sp[1]) + (sp[2] + sp[3]);
}
__global__ void ce_fold_kernel(const float* __restrict__ partial, int n, long rows, float* __restrict__ loss)
{
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s / (float)rows;
}
extern "C" int vpf_ce_smooth(const float* logits, long ld, const long long* target, long rows, int C, float eps, float* partial_ws,
                             float* loss, float* dlogits, long lddz, void* stream)
{
    (void)hipGetLastError();
    if (!logits || !target || !partial_ws || !loss) return VPF_ERR_NULL;
    if (rows <= 0 || C <= 0 || ld < C || (dlogits && lddz < C)) return VPF_ERR_BADSHAPE;
    int grid = vpf_cdiv(rows, 4); if (grid > 1024) grid = 1024;         // partial_ws: >= 1024 floats
    hipLaunchKernelGGL(ce_smooth_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, ld, target, rows, C, eps, partial_ws, dlogits, lddz);
    hipLaunchKernelGGL(ce_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)partial_ws, grid, rows, loss);
    VPF_CHECK_LAUNCH();
    return VPF_OK;
}
