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
 *   - row-major tensors, innermost index last; fp32 unless the name says h16
 *     (h16 buffers are uint16_t bit patterns).
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
/* The type of every "h16" buffer and matrix-core product of this library: 1 = IEEE fp16 (default: the reference's autocast dtype,
 * pretrain.py:154,176), 0 = bf16 (a build with -DVPF_OPERAND_FP16=0: rounds 1-3, kept for A/B runs). */
int vpf_operand_dtype(void);
/* "VPF_BUILD_ID=<sha256 of the library's sources>": vipformer_amd/build.py rebuilds when it does not match the tree */
const char* vpf_build_id(void);
/* Launch-time experiment knobs (kernel variants, grid caps: csrc/vpf_common.h VpfDebug) by name; the library reads the VPF_*
 * environment variables once, when the first launcher asks, and never again.  Not a user-facing option: tests and tools only. */
int vpf_debug_set(const char* key, int value);
int vpf_debug_get(const char* key, int* value);

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


/* ------------------------------------------------------------------ h16 MFMA GEMM family
 * C[m,n] (+)= sum_k A(m,k) * B(n,k), h16 operands, fp32 accumulate (v_mfma_f32_32x32x16_f16).
 * Replaces every aten linear / conv1d(k=1) / mm of the path and its autograd backward:
 * q/k/v/o projections partseg.py:48-51,67-69,86; MLP partseg.py:191-198; Group2Emb convs
 * utils.py:153-165; adapter / position / patch linears classifier.py:35, partseg.py:500,633;
 * latent_head partseg.py:522,524.
 *   a_kstrided / b_kstrided: 0 = operand stored [rows][K] (K contiguous); 1 = stored [K][rows]
 *     (forward: 0,0 with B = W[N,K];  dgrad dX = dY W: A = dY (0), B = W (1);
 *      wgrad dW = dY^T X: A = dY (1), B = X (1), mode VPF_EPI_ATOMIC, split over K = tokens).
 *   contiguous dimension of each operand and lda/ldb must be multiples of 8, bases 16-byte aligned.
 *   batch > 1: blockIdx.z strides sAb/sBb/sCb (elements).  splitk: 0 = auto (atomic mode only).
 *   mode (epilogue): 0 store (+bias) | 1 bias+GELU, C2 = pre-activation (h16) | 2 C(f32) = res +
 *     dropout(acc+bias) keyed by (rng_state, site, m*N+n) | 3 C = acc * gelu'(aux) | 4 C(f32) += acc
 *     (atomics) | 5 bias+ReLU | 6 acc + gbias[(m/group)*N + n].
 *   dbias (nullable; wgrad only): dbias[m] += sum_k A(m,k), the bias gradient, fused into the same pass. */
#define VPF_EPI_STORE 0
#define VPF_EPI_GELU 1
#define VPF_EPI_DROP_RES 2
#define VPF_EPI_GELU_BWD 3
#define VPF_EPI_ATOMIC 4
#define VPF_EPI_RELU 5
#define VPF_EPI_GROUPBIAS 6
int vpf_gemm_h16(const void* A, int a_kstrided, long lda, const void* B, int b_kstrided, long ldb,
                  int M, int N, int K, int batch, long sAb, long sBb, long sCb,
                  void* C, long ldc, int c_is_f32, int mode, const float* bias,
                  void* C2, long ldc2, const float* res, long ldres, const void* aux, long ldaux,
                  const float* gbias, int group, const uint32_t* rng_state, uint32_t site, float p,
                  int splitk, float* dbias, void* stream);

/* The same GEMM with operand prologues and a max-pool epilogue (Group2Emb's last conv utils.py:163-165,188 and its
 * backward): a_kind / b_kind 1 = relu(scale[c]*x + shift[c]) applied while the operand is staged (BatchNorm + ReLU,
 * the normalised activation never reaches HBM); a_kind 2 = the A operand is the VIRTUAL gradient of a max over
 * a_group consecutive rows, rebuilt from (a_dout f32 [rows/a_group, a_ncols], a_arg uint8) -- A is ignored.
 * mode 0 store | 4 atomic (+dbias) | 7 group max: C f32 [M/group, N] = max over each group of `group` rows of
 * h16(acc + bias), C2 uint8 = first arg-max (group must divide 32). */
#define VPF_EPI_GROUPMAX 7
int vpf_gemm_h16_fused(const void* A, int a_kstrided, long lda, int a_kind, const float* a_scale, const float* a_shift,
                        const float* a_dout, const uint8_t* a_arg, int a_group, long a_ncols,
                        const void* B, int b_kstrided, long ldb, int b_kind, const float* b_scale, const float* b_shift,
                        int M, int N, int K, void* C, long ldc, int c_is_f32, int mode, const float* bias,
                        void* C2, long ldc2, int group, int splitk, float* dbias, void* stream);

/* ------------------------------------------------------------------ fused attention (head dim 64)
 * MultiHeadAttention.forward partseg.py:67-86: softmax(q k^T * scale) -> dropout(p) -> . v, without the
 * [b*h, Lq, Lkv] matrix in HBM.  q/k/v/out are h16 [B, L, H*64] views with row strides ld* (elements),
 * head h at column h*64 (so column slices of a fused [M,3D] projection buffer are valid operands).
 * lse f32 [B,H,Lq] = log-sum-exp of the scaled scores.  Dropout keep decisions are a pure function of
 * (rng_state, site, (b*H+h, q, kv)); backward regenerates them. */
int vpf_attention_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, int B, int H,
                      int Lq, int Lkv, int head_dim, float scale, float dropout_p, const uint32_t* rng_state,
                      uint32_t site, void* out, long ldo, float* lse, void* stream);
/* autograd backward of the above: dq/dk/dv h16, same layouts.  delta_ws: f32 workspace [B*H*Lq]. */
int vpf_attention_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* out,
                      long ldo, const void* dout, long lddo, const float* lse, int B, int H, int Lq, int Lkv,
                      int head_dim, float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                      void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, float* delta_ws, void* stream);
/* The same pair with the key padding mask of partseg.py:54,73-76: pad_mask uint8 [B, Lkv], non-zero = padding key, shared by the
 * H heads of a batch element (`repeat(pad_mask, "b j -> (b h) () j")`).  A padded key's score is replaced by the most negative
 * finite float before the softmax, as masked_fill_(pad_mask, -finfo.max) does: it receives probability 0 beside any real key, a row
 * whose keys are ALL padded attends uniformly to its Lkv keys, and no gradient reaches q / k through a padded score (dv still
 * receives p * dout).  pad_mask must not be NULL (VPF_ERR_NULL); without a mask call the entry points above. */
int vpf_attention_fwd_pad(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, int B, int H,
                          int Lq, int Lkv, int head_dim, float scale, float dropout_p, const uint32_t* rng_state,
                          uint32_t site, void* out, long ldo, float* lse, const uint8_t* pad_mask, void* stream);
int vpf_attention_bwd_pad(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* out,
                          long ldo, const void* dout, long lddo, const float* lse, int B, int H, int Lq, int Lkv,
                          int head_dim, float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                          void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, float* delta_ws,
                          const uint8_t* pad_mask, void* stream);

/* ------------------------------------------------------------------ LayerNorm / dropout / residual
 * nn.LayerNorm of CrossAttention.q_norm/kv_norm, SelfAttention.norm, MLP[0] (partseg.py:100-101,
 * 131,193), eps 1e-5.  y(h16) = LN(x [+ pos[row % pos_rows]]); xsum (nullable) receives x + pos (the
 * residual base of partseg.py:326,335); mean/rstd f32 [rows] are kept for backward.  D <= 512. */
int vpf_layernorm_fwd(const void* x, int x_is_h16, const float* pos, int pos_rows, const float* gamma,
                      const float* beta, void* y_h16, float* xsum, float* mean, float* rstd, long rows, int D,
                      float eps, void* stream);
/* dx = (dres ? dres : 0) + LN'(dy); dgamma/dbeta += (fp32).  ws (nullable, >= 2*1024*D floats): per-block
 * partial sums instead of contended atomics. */
int vpf_layernorm_bwd(const void* dy_h16, const void* x, int x_is_h16, const float* mean, const float* rstd,
                      const float* gamma, const float* dres, void* dx, int dx_is_h16, float* dgamma, float* dbeta,
                      float* ws, long ws_floats, long rows, int D, void* stream);
/* Residual.forward partseg.py:208-212: out = res + dropout(y) (generic path; the attention / MLP blocks
 * fuse this into the producing GEMM, mode 2). */
int vpf_dropout_add_fwd(const void* y_h16, const float* res, float* out, long n, const uint32_t* rng_state,
                        uint32_t site, float p, void* stream);
int vpf_dropout_bwd(const float* dout, void* dy_h16, long n, const uint32_t* rng_state, uint32_t site, float p,
                    void* stream);
/* the keep mask (1/0 bytes) the kernels use for `site` at the current state (test hook) */
int vpf_dropout_mask(uint8_t* out, long n, const uint32_t* rng_state, uint32_t site, float p, void* stream);
/* rng_state = {seed_lo, seed_hi, step, 0}: step += 1 on the device (fresh masks per hipGraph replay) */
int vpf_rng_advance(uint32_t* rng_state, void* stream);
/* Timeline diagnostics (tools/step_timeline.py): slots[slot] = the device's constant-rate clock (vpf_wall_clock_khz ticks per ms) when the
 * stream reaches this point; capturable, so the replayed graph of the step (pretrain.py:173-211) can be timed branch by branch. */
int vpf_stamp(unsigned long long* slots, int slot, void* stream);
int vpf_wall_clock_khz(void);
int vpf_cast_f32_h16(const float* x, void* y_h16, long n, void* stream);
int vpf_cast_h16_f32(const void* x_h16, float* y, long n, void* stream);
/* acc[c] += sum_m x[m,c]; acc2[c] += sum_m x^2 (nullable): bias gradients and BatchNorm statistics */
int vpf_colsum(const void* x, int x_is_h16, long M, int C, float* acc, float* acc2, void* stream);
/* out[c] = sum_r x[r,c] with r ascending (deterministic) */
int vpf_sum_rows_f32(const float* x, int R, int C, float* out, void* stream);
int vpf_axpy_f32(const float* x, float* y, long n, float a, void* stream);
/* acc[r % period, :] += x[r, :]: gradient of CrossFormer_img_mp.position_emb [1,T,D] (partseg.py:637) */
int vpf_rowsum_mod_f32(const float* x, long rows, int D, int period, float* acc, void* stream);

/* ------------------------------------------------------------------ BatchNorm1d (channels-last [M,C])
 * utils.py:156,163 and partseg.py:520,523, momentum 0.1, eps 1e-5.  stat = [mean(C) | rstd(C)]:
 * training -> from batch sums (vpf_colsum), updates running_mean/var (unbiased) and num_batches_tracked;
 * eval -> from the running statistics. */
int vpf_bn_finalize(const float* sums, const float* sumsq, long M, int C, float eps, float momentum, int training,
                    float* running_mean, float* running_var, long long* num_batches, float* stat, void* stream);
/* ab = [rstd*gamma | beta - mean*rstd*gamma]: BatchNorm as a per-channel affine (operand prologue of vpf_gemm_h16_fused) */
int vpf_bn_affine(const float* stat, const float* gamma, const float* beta, int C, float* ab, void* stream);
int vpf_bn_act_fwd(const void* x, int x_is_h16, const float* stat, const float* gamma, const float* beta, void* y,
                   int y_is_h16, long M, int C, int relu, void* stream);
/* dx (nullable) = BN'(relu'(dy)); dgamma/dbeta +=.  tmp2C_zeroed: f32 [2C] scratch, zero on entry. */
int vpf_bn_bwd(const void* dy, int dy_is_h16, const void* x, int x_is_h16, const float* stat, const float* gamma,
               const float* beta, long M, int C, int relu, int training, float* tmp2C_zeroed, void* dx, int dx_is_h16,
               float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------ Group2Emb pieces (utils.py:168-189)
 * first conv (C->64) + BatchNorm(64) + ReLU without materialising the conv output: statistics pass,
 * apply pass (out h16 [M,64]), two-pass backward (weight/bias/BN-affine gradients; the input needs none). */
int vpf_g2e_conv1_stats(const float* x, long M, int C, const float* W, const float* b, float* sums, float* sumsq, void* stream);
/* the same statistics from the C x C second-moment matrix of the inputs (h1 is affine in x); scratch = f32 [72 + 512*72];
 * per-block partials folded in a fixed order (deterministic) */
int vpf_g2e_conv1_stats_moments(const float* x, long M, int C, const float* W, const float* b, float* scratch,
                                float* sums, float* sumsq, void* stream);
int vpf_g2e_conv1_apply(const float* x, long M, int C, const float* W, const float* b, const float* stat, const float* gamma,
                        const float* beta, void* out_h16, void* stream);
/* backward in ONE pass over da (the gradient of the ReLU output): sum g, sum g xhat, sum g x_i per channel, combined with the input
 * moments of the forward pass (mom = the first 72 floats of vpf_g2e_conv1_stats_moments' / vpf_g2e_bn1_prepare's scratch: sum x |
 * sum x x^T; may be null when training == 0) into dW / db / dgamma / dbeta (+=).  ws: per-block partials, (blocks + 1) * 320 floats,
 * up to 1025 * 320. */
int vpf_g2e_conv1_bwd(const float* x, const void* da_h16, long M, int C, const float* W, const float* b, const float* stat,
                      const float* gamma, const float* beta, int training, const float* mom, float* dW, float* db,
                      float* dgamma, float* dbeta, float* ws, long ws_floats, void* stream);
/* ... the same with the second conv's input gradient folded in: da = dh2 . W2 (W2 = Conv1d(64, 128).weight as h16 [128][64]) is formed
 * on the matrix cores and consumed in registers; dh2 h16 [M, 128], 16-byte aligned */
int vpf_g2e_conv1_bwd_fused(const float* x, const void* dh2_h16, long M, int C, const float* W, const float* b, const float* stat,
                            const float* gamma, const float* beta, int training, const float* mom, const void* W2_h16,
                            float* dW, float* db, float* dgamma, float* dbeta, float* ws, long ws_floats, void* stream);
/* Group2Emb forward for group_size == 32 as two persistent weight-stationary kernels (conv weights held in registers as
 * MFMA fragments, activations of a pair of groups in LDS; only the pre-BN2 activation h3 and what backward needs reach HBM):
 *   vpf_g2e_fold_bn1: BatchNorm-1 folded into the first conv (ab1 from vpf_bn_affine)
 *   vpf_g2e_fwd_a:    x -> a1, h2, gmax/arg2 (max over members), h3 = conv3([gmax | h2]) ; partials[wg][512] = column sum | sum^2
 *                     of h3 per workgroup (*nrows_out rows; fold with vpf_sum_rows_f32: deterministic BatchNorm statistics)
 *   vpf_g2e_fwd_b:    h3 -> BN2+ReLU -> conv4 -> max over members: out f32 [NG,Dm], arg4 */
int vpf_g2e_fold_bn1(const float* W1, const float* b1, const float* ab1, int C, float* w1e, float* b1e, void* stream);
int vpf_g2e_fwd_a(const float* x, long NG, int C, const float* w1e, const float* b1e, const void* w2_h16, const float* b2,
                  const void* w3_h16, const float* b3, void* a1, void* h2, void* gmax, uint8_t* arg2, void* h3,
                  float* partials_256x512, int* nrows_out /* host */, void* stream);
int vpf_g2e_fwd_b(const void* h3_h16, long NG, const float* ab2, const void* w4_h16, const float* b4, int Dm, float* out,
                  uint8_t* arg4, void* stream);
/* Group2Emb backward, group_size == 32:
 *   vpf_g2e_wgrad4: dW4 / db4 from the max-pool gradient (ONE non-zero per (group, column): 32x less work than a dense wgrad)
 *   vpf_g2e_bwd (Dm <= 256): conv4 dgrad on MFMA from the rebuilt gradient tile, BatchNorm-2 backward (two passes), dh3, the
 *                per-group sums dgb and dh2 = dh3 . W3[:,128:]  (w4t = W4^T [256,Dm], w3bt = W3[:,128:]^T [128,256], h16)
 *   vpf_transpose_h16: dst[c][r] = src[r][c] */
int vpf_g2e_wgrad4(const void* h3_h16, long NG, const float* ab2, const float* dout, const uint8_t* arg4, int Dm,
                   float* dW4, float* db4, void* stream);
int vpf_g2e_bwd(const float* dout, const uint8_t* arg4, int Dm, long NG, const void* h3_h16, const float* stat2,
                const float* gamma2, const float* beta2, const void* w4t_h16, const void* w3bt_h16, int training,
                float* tmp512_zeroed, void* dh3_h16, float* dgb, void* dh2_h16, float* dgamma2, float* dbeta2,
                long long* dbg_cycles /* nullable diagnostic: [256*2*6] per-phase cycle sums */, void* stream);
int vpf_transpose_h16(const void* src, long ld, int R, int C, void* dst, void* stream);
/* torch.max over the K group members (utils.py:180,188): h h16 [NG,K,C] -> out [NG,C], arg uint8 (first max) */
int vpf_group_max_fwd(const void* h_h16, long NG, int K, int C, void* out, int out_is_h16, uint8_t* arg, void* stream);
int vpf_group_max_bwd(const void* dout, int dout_is_h16, const uint8_t* arg, long NG, int K, int C, void* dh_h16, void* stream);
/* sum over the K members (gradient of the broadcast global feature) and max-pool backward added in place */
int vpf_group_sum(const void* x_h16, long NG, int K, int C, float* out, void* stream);
int vpf_group_max_scatter_add(const void* dg_h16, const uint8_t* arg, long NG, int K, int C, void* dh_h16, void* stream);
/* torch.cat([global.expand, local]) (utils.py:183) and its backward */
int vpf_g2e_concat_fwd(const void* gmax_h16, const void* h_h16, long M, int K, int C, void* feat_h16, void* stream);
int vpf_g2e_concat_bwd(const void* dfeat_h16, const uint8_t* arg, long NG, int K, int C, void* dh_h16, void* stream);

/* ------------------------------------------------------------------ K=3 front-ends, patchify, pooling
 * PointCloudInputAdapter.point_mlp[0:3] (classifier.py:31-34): Linear(C,64) LayerNorm(64) ReLU -> h16 [M,64] */
int vpf_adapter_front_fwd(const float* x, long M, int C, const float* W, const float* b, const float* gamma,
                          const float* beta, void* out_h16, void* stream);
int vpf_adapter_front_bwd(const float* x, const void* da_h16, long M, int C, const float* W, const float* b,
                          const float* gamma, const float* beta, float* dW, float* db, float* dgamma, float* dbeta,
                          float* ws, long ws_floats, void* stream);   /* ws: per-block partials, up to 2048 * 704 floats */
/* y = act(x W^T + b), x f32 [M,C<=8] -> h16 [M,N]; act 1 = GELU(erf): position_emb[0:2] (partseg.py:498-500) */
int vpf_smallk_fwd(const float* x, long M, int C, const float* W, const float* b, int N, int act, void* out_h16, void* stream);
int vpf_smallk_bwd(const float* x, const void* dy_h16, long M, int C, const float* W, const float* b, int N, int act,
                   float* dW, float* db, void* stream);
/* Rearrange('b (h p1) (w p2) c -> b (h w) (p1 p2 c)') (partseg.py:632) on an arbitrary-stride [B,H,W,C] view
 * (pretrain.py:179 hands a permuted NCHW tensor): out h16 [B*T, p*p*C] */
int vpf_patchify(const float* img, long sb, long sh, long sw, long sc, int B, int H, int W, int C, int p, void* out_h16, void* stream);
/* cat[x.max(1)[0], x.mean(1)] (partseg.py:547): x f32 [B,L,D] -> out f32 [B,2D], arg int32 [B,D] */
int vpf_pool_fwd(const float* x, int B, int L, int D, float* out, int* arg, void* stream);
int vpf_pool_bwd(const float* dout, const int* arg, int B, int L, int D, float* dx, void* stream);

/* ------------------------------------------------------------------ loss and optimizer
 * lightly==1.1.21 NTXentLoss(temperature, memory_bank_size=0) (pretrain.py:155,196,202; third party, absent
 * from the reference tree): z0,z1 f32 [b,D]; workspaces zn [2b,D], inv_norm [2b], P [2b,2b], loss_rows [2b];
 * loss = scalar.  bwd: dz0/dz1 from dloss (device scalar). */
int vpf_ntxent_fwd(const float* z0, const float* z1, int b, int D, float temperature, float* zn, float* inv_norm, float* P,
                   float* loss_rows, float* loss, void* stream);
int vpf_ntxent_bwd(const float* zn, const float* inv_norm, const float* P, int b, int D, float temperature, const float* dloss,
                   float* dz0, float* dz1, void* stream);
/* Group2Emb (utils.py:150-189), training mode, the BatchNorm bookkeeping between the kernels as single launches:
 * vpf_g2e_bn1_prepare = vpf_g2e_conv1_stats_moments + vpf_bn_finalize + vpf_bn_affine + vpf_g2e_fold_bn1 (stat = mean | rstd [128],
 * ab = a | b [128], w1e [64*C], b1e [64]; scratch f32 [72 + 512*72], its first 72 floats hold the input moments on return);  vpf_bn_partials_finalize = vpf_sum_rows_f32 + vpf_bn_finalize
 * + vpf_bn_affine on per-workgroup partial rows [nrows][2C] (C % 64 == 0). */
int vpf_g2e_bn1_prepare(const float* x, long M, int C, const float* W, const float* b, float* scratch, const float* gamma, const float* beta,
                        float eps, float momentum, float* running_mean, float* running_var, long long* num_batches, float* stat, float* ab,
                        float* w1e, float* b1e, void* stream);
int vpf_bn_partials_finalize(const float* partials, int nrows, int C, long M, const float* gamma, const float* beta, float eps, float momentum,
                             float* running_mean, float* running_var, long long* num_batches, float* stat, float* ab, void* stream);
/* nn.BatchNorm1d (+ ReLU) over a SMALL batch (M <= 4096 rows, C % 64 == 0: the projection heads, partseg.py:519-525) in training
 * mode as one kernel each way: batch statistics, running-statistics update and normalisation (stat = mean | rstd is kept for the
 * backward); backward: dx (h16 or f32, may be NULL) and dgamma / dbeta += . */
int vpf_bn_small_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, long long* num_batches, float* stat, void* y_h16, int relu, void* stream);
int vpf_bn_small_bwd(const float* dy, const float* x, const float* stat, const float* gamma, const float* beta, int M, int C, int relu,
                     void* dx, int dx_is_h16, float* dgamma, float* dbeta, void* stream);
/* Both pre-training losses in one go (pretrain.py:196-204): f f32 [2b,D] = the two point-cloud views stacked, g f32 [b,D] =
 * the image features; total f32[1] = imid + w*cmid, parts f32[2] = {imid = NTXent(f[:b], f[b:]), cmid = NTXent((f[:b]+f[b:])/2, g)}.
 * Workspaces: zn [2,2b,D], inv_norm [2,2b], P [2,2b,2b], loss_rows [2,2b]; bwd: ws_dz [2,2b,D], df [2b,D], dg [b,D] from
 * dtotal (device scalar). */
int vpf_pretrain_loss_fwd(const float* f, const float* g, int b, int D, float temperature, float cmid_weight, float* zn,
                          float* inv_norm, float* P, float* loss_rows, float* total, float* parts, void* stream);
int vpf_pretrain_loss_bwd(const float* zn, const float* inv_norm, const float* P, int b, int D, float temperature, float cmid_weight,
                          const float* dtotal, float* ws_dz, float* df, float* dg, void* stream);
/* torch.optim.AdamW (pretrain.py:121-124,210) over a flat fp32 buffer, also rewriting the h16 shadow the MFMA
 * kernels read, with torch.cuda.amp.GradScaler's step / update folded in (pretrain.py:154,209-211).  hyper_dev (device, 16 floats) =
 * {lr, beta1, beta2, eps, weight_decay, grad_scale, step, skip, loss_scale (0 = none), growth_tracker, growth_interval, found_inf,
 *  growth_factor, backoff_factor, skipped_steps, already_unscaled}: g carries the factor loss_scale; a step with found_inf set is skipped
 *  and halves the scale, growth_interval good steps in a row double it (the kernel behind the last AdamW launch of a step). */
/* advance_step: bit 0 = advance the bias-correction step + the scaler's update, bit 1 = zero g after use (the next step's optimizer.zero_grad()) */
int vpf_adamw_step(float* p, float* g, float* m, float* v, void* shadow_h16, long n, float* hyper_dev,
                   int advance_step, void* stream);
/* GradScaler's overflow check over the flat gradient (scaler.step's inf check, pretrain.py:210): hyper_dev[11] = 1 if any of the n
 * gradients is inf or NaN (g 16-byte aligned). */
int vpf_grad_check(const float* g, long n, float* hyper_dev, void* stream);

/* Up to 32 weight-gradient GEMMs in one launch (the backward of nn.Linear: the four of a transformer layer, or -- round 3 -- those
 * of a whole encoder stack, so that the split-K flush is paid once per stack; partseg.py:48-51,194-197): dW[N,K] += dy[M,N]^T x[M,K] (h16 operands, fp32 atomics), dbias[N] += column sums of dy
 * (dbias may be NULL).  host_jobs is a HOST array (copied into the kernel arguments: capturable). */
typedef struct VpfWgradJob { const void* dy; const void* x; int M, N, K; float* dW; float* dbias; } VpfWgradJob;
/* ws (nullable): >= 4096 + 65536 * (number of workgroups, <= ~640) bytes of scratch, 16-byte aligned, whose first 4096 bytes were
 * zeroed ONCE by the caller (arrival counters; the kernel leaves them zero), private to the stream: with it the split-K slices
 * exchange their partial tiles through the workspace and the last-arriving slice of a tile writes dW -- no atomics on dW;
 * without it (or if it is too small) fp32 atomics as before. */
/* Round 5: problems with N_out % 128 == 0, K_in % 128 == 0, tokens % 64 == 0 and at least VPF_WGROUP_DMA (default 2048) tokens run the
 * LDS-DMA kernel (csrc/gemm.hip: gemm_wgrad_dma_kernel) when no workspace is given: 256 x 128 tiles when EVERY conforming problem of the
 * group has N_out % 256 == 0, otherwise the whole DMA group runs on 128 x 128 tiles (one N_out that is a multiple of 128 only -- D = 384
 * -- moves all of them); a group that holds conforming and non-conforming problems becomes two launches on `stream`.  Same results up
 * to the order of the fp32 atomics. */
int vpf_wgrad_group(const VpfWgradJob* host_jobs, int njobs, void* ws, long ws_bytes, void* stream);

/* ------------------------------------------------------------------ fused self-attention layer
 * SelfAttentionLayer.forward (partseg.py:170-188; Residual :201-213, MultiHeadAttention :14-86, MLP :191-198) for
 * D = 256, 4 heads of 64, hidden 512 (with attention_done also D = 384, 6 heads, hidden 1536: BASELINE config 4), as ONE kernel per layer: attention -> o_proj + dropout + residual -> LayerNorm
 * -> fc1 + GELU -> fc2 + dropout + residual, and -- when qkv_next is set -- the NEXT layer's (+pos) -> LayerNorm ->
 * q/k/v projection (Encoder.forward re-adds pos before every layer, partseg.py:326-335).  One workgroup owns
 * chunk_rows tokens of one sequence (L <= 96: the whole sequence; L <= 224: chunk_rows <= 128); with attention_done
 * the attention itself is left to vpf_attention_fwd and a workgroup owns any 64 consecutive rows.
 * Weights are read in the MFMA fragment order produced by vpf_pack_wfrag from the natural h16 [N,K] matrices.
 * Everything the backward needs is written as the unfused ops write it. */
typedef struct VpfPackJob { const void* src; void* dst; int N, K; int transposed; int pad_; } VpfPackJob;
#define VPF_PACK_MAX_JOBS 64
/* logical A[N][K] -> fragment order: ((cb*(K/16) + ks)*64 + lane)*8 + j  <-  A[cb*32 + (lane&31)][ks*16 + 8*(lane>>5) + j];
 * A[n][k] = src[n*K + k] (a natural nn.Linear weight, forward) or, with transposed = 1, src[k*N + n] (the same weight
 * read for dgrad: N = in_features, K = out_features).
 * jobs is a HOST array (copied into the kernel arguments: capturable). */
int vpf_pack_wfrag(const VpfPackJob* host_jobs, int njobs, void* stream);

typedef struct VpfSaLayerFwd {
    int B, L, chunk_rows, D, H, hidden;
    const void* qkv;            /* h16 [B*L, 3D]: q | k | v of THIS layer */
    const float* base;          /* f32 [B*L, D]: residual base of this layer (x + pos) */
    const uint32_t* rng;        /* dropout state (vpf_dropout_*) */
    float scale, p_att; uint32_t site_att;
    const void* Wo; const float* bo;               /* packed [D,D], f32 [D] */
    float p_res1; uint32_t site_res1;
    const float* ln2_g; const float* ln2_b;
    const void* W1; const float* b1;               /* packed [hidden,D] */
    const void* W2; const float* b2;               /* packed [D,hidden] */
    float p_res2; uint32_t site_res2;
    /* saved for backward */
    void* o; float* lse;                            /* h16 [M,D], f32 [B,H,L] */
    float* x1; float* mean2; float* rstd2; void* n2;   /* f32 [M,D], [M], [M], h16 [M,D] */
    void* u; void* h;                               /* h16 [M,hidden]: fc1 pre-activation, GELU output */
    float* out;                                     /* f32 [M,D]: x2 (+ pos when pos != NULL) */
    /* next layer's head (all NULL after the last layer) */
    const float* pos; int pos_rows;                 /* f32 [pos_rows, D], row m uses pos[m % pos_rows] */
    const float* ln1n_g; const float* ln1n_b; const void* Wqkv_next;   /* packed [3D,D] */
    float* mean1n; float* rstd1n; void* n1n; void* qkv_next;
    int attention_done;                             /* 1: o / lse were produced by vpf_attention_fwd (o is an INPUT; any L) */
    long long* dbg;                                 /* optional: 8 phase cycle counters of workgroup 0 (profiling aid) */
} VpfSaLayerFwd;
int vpf_sa_layer_fwd(const VpfSaLayerFwd* host_args, void* stream);
/* The dgrad chain of the same layer (backward of partseg.py:170-213) as two row-block kernels around vpf_attention_bwd:
 *   _mlp: d = dL/d(x2) -> dz2 = dropout'(d) -> du = (dz2 W2) * gelu'(u) -> dn = du W1 -> dx1 = LayerNorm2'(dn) + d
 *         -> dz1 = dropout'(dx1) -> dout_attn = dz1 Wo
 *   _qkv: dbase = LayerNorm1'(dqkv Wqkv) + dx1 (also accumulated into dsum when set)
 * Each writes its workgroups' LayerNorm parameter-gradient partials to pgrad2 / pgrad1 (ceil(M/64) rows of 512 floats:
 * dgamma | dbeta); vpf_ln_pgrad_reduce folds any number of those into the gradient buffers in one launch.
 * dz2, du, dz1, dqkv are the h16 operands of the weight-gradient GEMMs (vpf_wgrad_group).  W*T = vpf_pack_wfrag with
 * transposed = 1. */
typedef struct VpfSaLayerBwd {
    int M, D, hidden;
    const uint32_t* rng;
    float p_res1; uint32_t site_res1; float p_res2; uint32_t site_res2;
    /* _mlp */
    const float* d; const void* u; const float* x1; const float* mean2; const float* rstd2; const float* ln2_g;
    const void* W2T; const void* W1T; const void* WoT;
    void* dz2; void* du; float* dx1; void* dz1; void* dout_attn;
    float* pgrad2;
    /* _qkv */
    const void* dqkv; const void* WqkvT; const float* base; const float* mean1; const float* rstd1; const float* ln1_g;
    float* dbase; float* dsum;
    float* pgrad1;
    int dsum_init;          /* 1: dsum = dbase (the first layer of a backward pass initialises the sum: no zero-fill launch); 0: dsum += dbase */
} VpfSaLayerBwd;
int vpf_sa_layer_bwd_mlp(const VpfSaLayerBwd* host_args, void* stream);
int vpf_sa_layer_bwd_qkv(const VpfSaLayerBwd* host_args, void* stream);
/* Round 3: the qkv half of one layer and the mlp half of the layer BELOW it (mlp->d == qkv->dbase: the gradient leaving the upper
 * layer enters the lower one) in ONE launch: a workgroup's gradient rows go from the first body to the second through LDS.  Where the
 * fused kernel does not apply (D = 384, or mlp->d != qkv->dbase) the two launches above run back to back: same results either way. */
int vpf_sa_layer_bwd_qkv_mlp(const VpfSaLayerBwd* qkv_of_layer, const VpfSaLayerBwd* mlp_of_layer_below, void* stream);
/* Backward of a cross-attention layer's query side (CrossAttention.q_norm + q_proj, partseg.py:100-116, and the Residual around it) on
 * the same struct: dqkv = dq h16 [M, D]; WqkvT = vpf_pack_wfrag(transposed = 1) of the h16 [D, D] q weight; base / mean1 / rstd1 / ln1_g =
 * the q LayerNorm's input, statistics and scale; dx1 = the residual's gradient; out: dbase f32 [M, D] (+= into dsum if set), pgrad1.
 * D = 256; VPF_ERR_UNSUPPORTED otherwise. */
int vpf_ca_front_bwd(const VpfSaLayerBwd* host_args, void* stream);
/* ... and of its key / value side when the kv input is an f32 [M, D] tensor (CrossAttention.kv_norm + k_proj | v_proj; the image branch:
 * the point-cloud branch's K / V producer has vpf_adapter_kv_bwd): dqkv = dk | dv h16 [M, 2D]; WqkvT = vpf_pack_wfrag(transposed = 1) of
 * the h16 [2D, D] k | v weights; base / mean1 / rstd1 / ln1_g = the kv LayerNorm's; dx1 may be NULL; out: dbase = dxkv f32 [M, D], pgrad1. */
int vpf_ca_kv_bwd(const VpfSaLayerBwd* host_args, void* stream);
/* The front of the point-cloud branch's cross-attention layer in ONE kernel (D = 256): position_emb (partseg.py:498-501:
 * Linear(3,128) GELU Linear(128,D)) on the group centres, base = tokens + pos (Encoder.forward, partseg.py:326), q_norm and the
 * bias-free q projection (partseg.py:100-116, 48-51).  W1 / Wq = vpf_pack_wfrag of the h16 weights [D,128] / [D,D].
 * Outputs are what the separate kernels write: hpos h16 [M,128] (GELU output), pos f32 [M,D], base f32 [M,D], mean / rstd f32 [M],
 * nq h16 [M,D], q h16 [M,D]. */
typedef struct VpfCaFront {
    long M; int D, hidden, C;
    const float* centers; const float* W0; const float* b0; const void* W1; const float* b1;
    const float* x; const float* lnq_g; const float* lnq_b; const void* Wq;
    void* hpos; float* pos; float* base; float* mean; float* rstd; void* nq; void* q;
} VpfCaFront;
int vpf_ca_front_fwd(const VpfCaFront* host_args, void* stream);
typedef struct VpfPgradJob { const float* partials; int rows; int D; float* dgamma; float* dbeta; } VpfPgradJob;   /* D = 0 means 256 */
#define VPF_PGRAD_MAX_JOBS 32
int vpf_ln_pgrad_reduce(const VpfPgradJob* host_jobs, int njobs, void* stream);
/* number of partial rows (2 D floats each) the two backward kernels write for M tokens: size pgrad1 / pgrad2 and the reduce job with it */
int vpf_sa_layer_pgrad_rows(long M, int D);
/* ... for an MLP hidden width other than the default of D (512 at D = 256, 1536 at D = 384): D = 256 / hidden 1024 (mlp_widen_factor 4) */
int vpf_sa_layer_pgrad_rows_h(long M, int D, int hidden);
/* PointCloudInputAdapter.point_mlp (classifier.py:31-36) + the cross-attention kv LayerNorm and K / V projections
 * (partseg.py:48-51,100-116) in one kernel, 64 (D = 256) or 32 (D = 384) points per workgroup.  x f32 [M,C<=8]; W1 f32 [64,C];
 * W2 = vpf_pack_wfrag of the h16 [D,64] weight; Wkv = vpf_pack_wfrag of the h16 [2D,D] k|v weights.
 * Outputs (all also needed by the backward pass): a1 h16 [M,64] (hidden layer), xkv h16 [M,D] (the per-point
 * embedding), mean / rstd f32 [M] and nk h16 [M,D] (kv LayerNorm), kv h16 [M,2D]. */
typedef struct VpfAdapterKv {
    long M; int C, D;
    const float* x; const float* W1; const float* b1; const float* ln_g; const float* ln_b;
    const void* W2; const float* b2; const float* lnkv_g; const float* lnkv_b; const void* Wkv;
    void* a1; void* xkv; float* mean; float* rstd; void* nk; void* kv;
} VpfAdapterKv;
int vpf_adapter_kv_fwd(const VpfAdapterKv* host_args, void* stream);
/* Backward of the same chain from dkv (h16 [M,2D]) down to the adapter's hidden layer, one kernel: dxkv h16 [M,D] (the
 * operand of the adapter's second-Linear weight gradient dxkv x a1; dkv x nk is the other GEMM) and da1 h16 [M,64] (the
 * input of vpf_adapter_front_bwd); the kv LayerNorm's parameter gradients go to pgrad_kv (ceil(M/64) rows of 2D floats,
 * folded by vpf_ln_pgrad_reduce).  WkvT / W2T = vpf_pack_wfrag(transposed = 1) of the h16 [2D,D] k|v weights
 * (N = D, K = 2D) and of the [D,64] weight (N = 64, K = D). */
typedef struct VpfAdapterKvBwd {
    long M; int C, D;
    const void* dkv; const void* WkvT; const void* xkv; const float* mean; const float* rstd; const float* lnkv_g;
    const void* W2T;
    void* dxkv; void* da1; float* pgrad_kv;
} VpfAdapterKvBwd;
int vpf_adapter_kv_bwd(const VpfAdapterKvBwd* host_args, void* stream);
/* number of partial rows vpf_adapter_kv_bwd writes to pgrad_kv for M points (one per workgroup: 64 points at D = 256, 32 at D = 384) */
int vpf_adapter_kv_pgrad_rows(long M, int D);
/* ------------------------------------------------------------------ part segmentation (BASELINE config 5)
 * CrossFormer_partseg.forward partseg.py:407-470 + PointNetFeaturePropagation.forward utils.py:205-242: everything that the
 * pre-training entry points above do not already cover (the 1x1 convolutions are vpf_gemm_h16, BatchNorm the entries above). */
/* utils.py:219-230: the three nearest of S centres per point by the exact square_distance recipe (the reference sorts all S),
 * weights 1/(d + 1e-8) normalised; idx int32 [B,N,3], weight f32 [B,N,3]; ties -> lower centre index.  S == 1: the broadcast of
 * utils.py:216-217 (weight exactly 1 on the one centre).  S <= 4096 (16 S bytes of LDS), else VPF_ERR_UNSUPPORTED. */
int vpf_three_nn_f32(const float* xyz, int B, int N, int C, const float* centers, int Cc, int S, int* idx, float* weight,
                     void* stream);
/* partseg.py:427-435: LayerNorm (one parameter set) of up to four encoder taps f32 [rows,D] into the concatenated feature
 * xcat f32 [rows, nl*D]; mean / rstd f32 [nl, rows].  Backward: d taps from dxcat, dgamma / dbeta += . */
int vpf_ln_taps_fwd(const float* x0, const float* x1, const float* x2, const float* x3, int nl, long rows, int D,
                    const float* gamma, const float* beta, float eps, float* xcat, float* mean, float* rstd, void* stream);
int vpf_ln_taps_bwd(const float* dxcat, const float* x0, const float* x1, const float* x2, const float* x3, int nl, long rows,
                    int D, const float* mean, const float* rstd, const float* gamma, float* d0, float* d1, float* d2,
                    float* d3, float* dgamma, float* dbeta, void* stream);
/* utils.py:230-236: A h16 [B*N, Kp] = [ xyz (C) | sum_k weight_k * feat[b, idx_k, :] (F) | 0 ]: the operand of mlp_convs[0] on
 * cat([points1, interpolated_points]).  Backward: dfeat f32 [B*S, F] += (zeroed by the caller; fp32 atomics). */
int vpf_interp_rows_fwd(const float* feat, const float* xyz, int B, int N, int C, int S, int F, const int* idx,
                        const float* weight, int Kp, void* A_h16, void* stream);
int vpf_interp_rows_bwd(const void* dA_h16, int B, int N, int C, int S, int F, const int* idx, const float* weight, int Kp,
                        float* dfeat, void* stream);
/* dst h16 [rows_out, Kp] = src [rows, K] (f32 or h16, row stride ld) zero-padded: GEMM operands whose contraction or output
 * dimension is not a multiple of 8 (3 + nl*D input channels of mlp_convs[0]; the 50 part classes of conv3). */
int vpf_pad_h16(const void* src, int src_is_h16, long rows, int K, long ld, long rows_out, int Kp, void* dst_h16, void* stream);
/* ft_partseg.py:128,155: CrossEntropyLoss(label_smoothing = eps), mean over rows.  logits f32 [rows, ld] (first C columns),
 * target int64 [rows]; partial_ws f32 [1024]; loss f32 [1]; dlogits (nullable) f32 [rows, lddz] = d loss / d logits.
 * A target outside [0, C) (torch raises; ignore_index is not used by the reference) makes the loss and that row's gradient NaN. */
int vpf_ce_smooth(const float* logits, long ld, const long long* target, long rows, int C, float eps, float* partial_ws,
                  float* loss, float* dlogits, long lddz, void* stream);
/* ------------------------------------------------------------------ on-device augmentation
 * trans_1 / trans_2 of datasets/data.py:16-36 (datasets/data_utils.py:56-221: Normalize, Scale(0.5,2), Rotate about y, Translate(0.5),
 * Jitter(0.01, clip 0.05), RandomInputDropout(0.875)) on a batch of raw clouds pts f32 [B,N,C>=3] -> out f32 [B,N,3], N <= 4096.
 * Random draws: the library's counter-based stream (rng_state, site), statistically -- not bitwise -- the reference's numpy / torch
 * draws; params_out (nullable) f32 [B,8] = {scale, angle, tx, ty, tz (unit draws), dropout ratio, radius, 0} for replay. */
int vpf_augment_points(const float* pts, int B, int N, int C, const uint32_t* rng_state, uint32_t site, float* out,
                       float* params_out, void* stream);
/* ToTensor + Normalize(mean, std) + RandomHorizontalFlip(p_flip) of utils.py:21-25 on uint8 img [B,H,W,3] -> f32 [B,3,H,W]
 * (mean3 / std3: HOST arrays of 3 floats; rng_state nullable = no flip; flips_out nullable u8 [B]). */
int vpf_image_u8_normalize(const void* img_u8, int B, int H, int W, const float* mean3_host, const float* std3_host,
                           const uint32_t* rng_state, uint32_t site, float p_flip, float* out, void* flips_out, void* stream);
/* sizeof(VpfSaLayerBwd) is vpf_abi_sizeof(3), sizeof(VpfPgradJob) (4), sizeof(VpfAdapterKv) (5), sizeof(VpfAdapterKvBwd) (6) */
/* sizeof(VpfPackJob) (which = 0) / sizeof(VpfSaLayerFwd) (1) / sizeof(VpfWgradJob) (2): lets a binding verify its struct layout */
int vpf_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* VIPFORMER_HIP_H */
