/*
 * vipformer_hip.h -- C ABI of libvipformer_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the ViPFormer --mp pre-training hot path.  The
 * reference (auniquesun/ViPFormer) has no FFI of its own: its "operator API" is
 * plain Python callables built from stock torch ops.  Each entry point below
 * replaces one such composition; the comment above it names the reference lines
 * (relative to the reference repo root) it stands in for.  The Python mirror of
 * the reference interface (vipformer_amd/model/pointcloud/{utils,partseg,classifier}.py)
 * is the only caller; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; the caller (torch)
 *     allocates all buffers including workspaces; the library keeps no pointer.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous with
 *     respect to the host and issues no allocation or synchronisation (safe to
 *     capture into a hipGraph).
 *   - return value: 0 = ok, negative = VPF_ERR_*.  No exceptions cross the ABI.
 *   - row-major tensors, innermost index last; fp32 unless the name says bf16
 *     (bf16 buffers are uint16_t bit patterns).
 */
#ifndef VIPFORMER_HIP_H
#define VIPFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPF_OK 0
#define VPF_ERR_BADSHAPE (-1)
#define VPF_ERR_BADALIGN (-2)
#define VPF_ERR_UNSUPPORTED (-3)
#define VPF_ERR_HIP (-4)
#define VPF_ERR_NULL (-5)

int vpf_version(void);
const char* vpf_strerror(int code);

/* ------------------------------------------------------------------ point-cloud preproc */

/* farthest_point_sample(pts, npoint)  vipformer/model/pointcloud/utils.py:56-85.
 * pts [B,N,C>=3] (xyz = first 3 channels), start_idx [B] = the torch.randint draw of
 * utils.py:71 (made by the caller so torch's RNG stream is consumed exactly as in the
 * reference), out_idx int64 [B,G].  fp32, evaluation order ((dx*dx+dy*dy)+dz*dz), no
 * FMA; argmax ties -> lowest index.  Bit-exact against the oracle.  N <= 4096. */
int vpf_fps_f32(const float* pts, int B, int N, int C, const int64_t* start_idx, int G,
                int64_t* out_idx, void* stream);

/* index_points(points, idx)  utils.py:88-104.  points [B,N,C], idx int64 [B,S] (S may be
 * G*K for the rank-3 form), out [B,S,C]. */
int vpf_index_points_f32(const float* points, int B, int N, int C, const int64_t* idx, int S,
                         float* out, void* stream);

/* square_distance(src, dst)  utils.py:122-141, C == 3 recipe
 * ((-2*fma(s2,d2,fma(s1,d1,s0*d0))) + |s|^2) + |d|^2.  src [B,Ns,Cs], dst [B,Nd,Cd]
 * (first 3 channels used), out [B,Ns,Nd]. */
int vpf_square_distance_f32(const float* src, int Cs, const float* dst, int Cd, int B, int Ns,
                            int Nd, float* out, void* stream);

/* knn_point(nsample, xyz, new_xyz)  utils.py:107-119, fused with the gather and the
 * centre subtraction of divide_patches (utils.py:22-36).
 * xyz [B,N,C], centers [B,G,Cc] (first 3 channels are the query position; Cc == C when
 * neighbours are requested).  Outputs (each nullable):
 *   knn_idx  int64 [B,G,K]  CANONICAL order: ascending distance, ties -> lower index
 *                           (torch.topk(sorted=False) leaves the order unspecified)
 *   knn_dist float [B,G,K]  the selected squared distances (bit-exact recipe above)
 *   neighbors float [B,G,K,C]  gathered rows; when apply_ref_axis_quirk != 0 members
 *                           0,1,2 of every group have the centre subtracted on all C
 *                           channels and members 3.. stay absolute (utils.py:36 slices
 *                           the member axis); 0 leaves every member absolute.
 * K <= 64, N <= 4096. */
int vpf_knn_group_f32(const float* xyz, int B, int N, int C, const float* centers, int Cc, int G,
                      int K, int apply_ref_axis_quirk, int64_t* knn_idx, float* knn_dist,
                      float* neighbors, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIPFORMER_HIP_H */
