/*
 * vpf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the index/integer part of the ViPFormer --mp
 * pre-training hot path (farthest-point sampling, kNN grouping).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (vipformer_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here bit-for-bit against fixtures captured from the imported reference
 * (tests/golden/make_golden.py, run in the build container against
 * /root/reference).
 *
 * Every function cites the reference lines (relative to the reference repo
 * root) whose arithmetic it restates.  fp32 evaluation order is part of the
 * contract: compile with -ffp-contract=off (see oracle/Makefile); the only
 * fused multiply-adds are the explicit fmaf() calls below.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* vipformer/model/pointcloud/utils.py:56-85  farthest_point_sample
 *   distance init 1e10 (:69); start index supplied by the caller (the reference
 *   draws it with torch.randint, :71); per iteration (:74-83)
 *     dist = sum((p - c)**2, -1)  -> ((dx*dx + dy*dy) + dz*dz), each op rounded
 *     distance = min(distance, dist); farthest = argmax(distance) (first max)
 */
int vpf_oracle_fps(const float* pts, int B, int N, int C, const int64_t* start_idx,
                   int G, int64_t* out_idx)
{
    if (!pts || !start_idx || !out_idx || B < 0 || N <= 0 || C < 3 || G < 0) return -1;
    float* dist = (float*)malloc(sizeof(float) * (size_t)N);
    if (!dist) return -2;
    for (int b = 0; b < B; ++b) {
        const float* p = pts + (size_t)b * N * C;
        for (int i = 0; i < N; ++i) dist[i] = 1e10f;
        int64_t far = start_idx[b];
        if (far < 0 || far >= N) { free(dist); return -1; }
        for (int g = 0; g < G; ++g) {
            out_idx[(size_t)b * G + g] = far;
            const float cx = p[far * C + 0], cy = p[far * C + 1], cz = p[far * C + 2];
            float best = -INFINITY; int64_t besti = 0;
            for (int i = 0; i < N; ++i) {
                const float dx = p[(size_t)i * C + 0] - cx;
                const float dy = p[(size_t)i * C + 1] - cy;
                const float dz = p[(size_t)i * C + 2] - cz;
                float d = dx * dx;
                d = d + dy * dy;
                d = d + dz * dz;
                /* torch.min propagates NaN; inputs here are finite by contract */
                if (d < dist[i]) dist[i] = d;
                if (dist[i] > best) { best = dist[i]; besti = i; }
            }
            far = besti;
        }
    }
    free(dist);
    return 0;
}

/* vipformer/model/pointcloud/utils.py:122-141  square_distance(src, dst), C == 3
 *   dist  = -2 * matmul(src, dst^T)   -> m = fma(s2,d2, fma(s1,d1, s0*d0)); -2*m exact
 *   dist += sum(src**2,-1)[:, :, None] -> ((s0*s0 + s1*s1) + s2*s2), no fma
 *   dist += sum(dst**2,-1)[:, None, :]
 * (the matmul's K=3 accumulation order is what torch 2.10 / MKL produces in
 *  the build container; pinned by tests/golden/sqdist_*.npz)
 */
static inline float sq3(const float* a)
{
    float s = a[0] * a[0];
    s = s + a[1] * a[1];
    s = s + a[2] * a[2];
    return s;
}
static inline float sqdist3(const float* s, float sn, const float* d, float dn)
{
    float m = s[0] * d[0];
    m = fmaf(s[1], d[1], m);
    m = fmaf(s[2], d[2], m);
    float r = -2.0f * m;
    r = r + sn;
    r = r + dn;
    return r;
}
int vpf_oracle_square_distance(const float* src, const float* dst, int B, int Ns, int Nd,
                               float* out)
{
    if (!src || !dst || !out) return -1;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < Ns; ++i) {
            const float* s = src + ((size_t)b * Ns + i) * 3;
            const float sn = sq3(s);
            for (int j = 0; j < Nd; ++j) {
                const float* d = dst + ((size_t)b * Nd + j) * 3;
                out[((size_t)b * Ns + i) * Nd + j] = sqdist3(s, sn, d, sq3(d));
            }
        }
    return 0;
}

/* vipformer/model/pointcloud/utils.py:107-119  knn_point(nsample, xyz, new_xyz)
 *   sqrdists = square_distance(new_xyz, xyz); topk(K, largest=False, sorted=False)
 * torch leaves the order of an unsorted topk unspecified, so the oracle (and the
 * HIP kernel) emit the CANONICAL order: ascending distance, ties -> lower index.
 * xyz: [B,N,C>=3] (first three channels used, utils.py:19), centers: [B,G,C].
 */
typedef struct { float d; int64_t i; } vpf_pair;
static int pair_cmp(const void* a, const void* b)
{
    const vpf_pair* x = (const vpf_pair*)a; const vpf_pair* y = (const vpf_pair*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i < y->i) ? -1 : (x->i > y->i);
}
int vpf_oracle_knn(const float* pts, int B, int N, int C, const float* centers, int G,
                   int K, int64_t* out_idx, float* out_dist /* nullable */)
{
    if (!pts || !centers || !out_idx || C < 3 || K > N || K <= 0) return -1;
    vpf_pair* row = (vpf_pair*)malloc(sizeof(vpf_pair) * (size_t)N);
    float* pn = (float*)malloc(sizeof(float) * (size_t)N);
    if (!row || !pn) { free(row); free(pn); return -2; }
    for (int b = 0; b < B; ++b) {
        const float* p = pts + (size_t)b * N * C;
        for (int j = 0; j < N; ++j) pn[j] = sq3(p + (size_t)j * C);
        for (int g = 0; g < G; ++g) {
            const float* c = centers + ((size_t)b * G + g) * C;
            const float cn = sq3(c);
            for (int j = 0; j < N; ++j) {
                row[j].d = sqdist3(c, cn, p + (size_t)j * C, pn[j]);
                row[j].i = j;
            }
            qsort(row, (size_t)N, sizeof(vpf_pair), pair_cmp);
            for (int k = 0; k < K; ++k) {
                out_idx[((size_t)b * G + g) * K + k] = row[k].i;
                if (out_dist) out_dist[((size_t)b * G + g) * K + k] = row[k].d;
            }
        }
    }
    free(row); free(pn);
    return 0;
}

/* vipformer/model/pointcloud/utils.py:6-38  divide_patches(points, G, K)
 *   centers = index_points(points, fps_idx)                      (:16, :88-104)
 *   idx     = knn_point(K, points[:,:,:3], centers[:,:,:3])      (:19)
 *   neighbors = points.reshape(B*N, C)[idx + b*N]                (:22-32)
 *   neighbors[:, :, :3] -= centers.unsqueeze(2)[:, :, :3]        (:36)
 * Line 36 slices the GROUP-MEMBER axis: members 0,1,2 of every group get the
 * centre subtracted on ALL C channels; members 3..K-1 stay absolute.  That is
 * the reference's behaviour and is reproduced when apply_ref_axis_quirk != 0
 * (0 = leave every member absolute; there is no "intended" mode in the
 * reference to restate).
 */
int vpf_oracle_divide_patches(const float* pts, int B, int N, int C, const int64_t* fps_idx,
                              int G, int K, int apply_ref_axis_quirk,
                              int64_t* knn_idx /* nullable */, float* neighbors, float* centers)
{
    if (!pts || !fps_idx || !neighbors || !centers) return -1;
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            const int64_t ci = fps_idx[(size_t)b * G + g];
            if (ci < 0 || ci >= N) return -1;
            memcpy(centers + ((size_t)b * G + g) * C, pts + ((size_t)b * N + ci) * C,
                   sizeof(float) * (size_t)C);
        }
    int64_t* idx = knn_idx ? knn_idx : (int64_t*)malloc(sizeof(int64_t) * (size_t)B * G * K);
    if (!idx) return -2;
    int rc = vpf_oracle_knn(pts, B, N, C, centers, G, K, idx, NULL);
    if (rc == 0) {
        for (int b = 0; b < B; ++b)
            for (int g = 0; g < G; ++g)
                for (int k = 0; k < K; ++k) {
                    const int64_t j = idx[((size_t)b * G + g) * K + k];
                    const float* p = pts + ((size_t)b * N + j) * C;
                    const float* c = centers + ((size_t)b * G + g) * C;
                    float* o = neighbors + (((size_t)b * G + g) * K + k) * C;
                    for (int ch = 0; ch < C; ++ch)
                        o[ch] = (apply_ref_axis_quirk && k < 3) ? (p[ch] - c[ch]) : p[ch];
                }
    }
    if (!knn_idx) free(idx);
    return rc;
}
