/*
 * sgcdet_amd.h -- C ABI of the MI355X (gfx950) view-transformation library.
 *
 * This is the drop-in boundary for the SGCDet hot path.
 * The reference reaches its native code through the pybind module `dfa3D._ext`
 * (packages/3D-deformable-attention/DFA3D/dfa3D/ops/csrc/pybind.cpp:42-67); the
 * four functions exported there map 1:1 onto the first four entry points below.
 * The remaining entry points are the fused / restructured forms the MI355X host
 * code calls instead of the reference's Python loops (file:line cited per
 * function).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory
 *     unless the name ends in `_host`;
 *   - all tensors are dense row-major ("contiguous"), fp32 unless stated,
 *     spatial-shape / level-start tensors are int64 exactly as the reference
 *     passes them (value_spatial_shapes, value_level_start_index);
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null
 *     stream); no internal synchronisation, no allocation, graph-capture safe;
 *   - caller owns every buffer; outputs documented as "accumulated" must be
 *     zeroed by the caller (same contract as the reference's backward,
 *     TU/multi_scale_3ddeformable_attn_function.py:319-322);
 *   - return 0 on success, a negative SGC_E* code otherwise; sgc_last_error()
 *     returns a static, thread-local description of the last failure.  Launch
 *     failures are returned, never printf-and-continue (contrast
 *     csrc/cuda/wms_deform_attn_cuda.cu:45-48).
 *   - index arithmetic inside kernels is 64-bit where products can pass 2^31.
 *
 * The CPU oracle (oracle/sgc_oracle.c, test infrastructure only) exports the
 * same symbols with the same signatures (stream ignored, pointers = host
 * memory), so one ctypes binding drives both.
 *
 * Shape letters: B batch (= cameras on the hot path), S = sum_l H_l*W_l value
 * pixels, M heads, Cm channels per head, D depth bins, L levels, Q queries,
 * P points per level.
 */
#ifndef SGCDET_AMD_H_
#define SGCDET_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGC_OK 0
#define SGC_EINVAL (-1)   /* bad argument (null pointer, non-positive size, ...) */
#define SGC_ELAUNCH (-2)  /* HIP launch / runtime error                          */
#define SGC_EUNSUP (-3)   /* shape not supported by this build                    */

/* Bumped whenever the argument list of an EXISTING entry point changes (a stale prebuilt .so is then rejected at load
 * instead of being called with shifted arguments).  1 -> 2: sgc_project_points gained `sel`, sgc_nchw_to_nhwc_crop gained
 * `step` (round 2).  New entry points do not bump it: a missing symbol already fails the load. */
#define SGC_ABI_VERSION 4

typedef void *sgc_stream_t; /* hipStream_t */

int sgc_abi_version(void);
/* Development knobs (A/B of kernel variants and launch geometries in one process; keys in csrc/dfa3d_fwd.hip).  Results do not
 * depend on them, with one stated exception: "split_target" / "halo_split_target" (and, round 6, "split_free" / "split_min_steps" /
 * "split_max": the tile kernel splits at any K step, see pick_split_steps in csrc/conv3d.hip) choose over how many workgroups a
 * layer with few voxels splits its reduction -- a different split adds the same partial sums in another order (fp32 rounding,
 * <= 1e-5 of the tensor scale; deterministic for a given value).  sgc_conv3d_workspace_floats follows the current values:
 * query it under the setting the call will run with.                                                                       */
int sgc_set_tuning(const char *key, int value);
/* Arithmetic mode of every bf16 MFMA kernel of the library (sgc_conv3d_cl_bf16x3 and its 2-D / masked forms,
 * sgc_linear_rows_*_bf16x3, sgc_level_tail):
 *   3 (default) = fp32-faithful: operands split a = a_hi + a_lo in bf16, products a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32
 *       accumulate -- agrees with exact fp32 products to ~1e-5 of the tensor scale (the parity mode, the headline);
 *   1 = plain bf16: a_hi*b_hi only, i.e. both operands ROUNDED to bfloat16 (RNE), fp32 accumulate -- the opt-in
 *       reduced-precision mode of BASELINE.json config #2 ("bf16"); 1/3 of the matrix work, ~2^-8 relative per operand;
 *   2 = plain fp16 (ABI version 3): ONE product on v_mfma_f32_32x32x16_f16 -- activations rounded to IEEE half (RNE) and
 *       SATURATED at +-65504 (a NaN stays a NaN), fp32 accumulate; the w_hi planes then hold IEEE-half bit patterns of the
 *       weights (TensorOps.split_f16), not bfloat16.  BASELINE.json config #5 ("fp16"; the reference's fp16 twin of the
 *       operator: TU/multi_scale_3ddeformable_attn_function.py:353-428); the rate of mode 1, ~2^-11 relative per operand.
 * It changes results (that is its purpose) and is therefore NOT a sgc_set_tuning key.  Process-wide; returns SGC_EINVAL for
 * any other value.  The w_lo arguments are ignored in modes 1 and 2.  sgc_conv3d_wgrad_bf16x3 always computes in mode 3;
 * the forward and input-gradient passes of training are the entry points above and FOLLOW the mode.  sgc_pack_conv_weight
 * emits bfloat16 planes, which mode 2 would misread as IEEE half: the fp16 mode is inference-only (the host side refuses it
 * with gradients enabled, sgcdet_amd/functions.py). */
int sgc_set_conv_products(int products);
int sgc_get_conv_products(void);
const char *sgc_last_error(void);
/* "hip-gfx950" for the product library, "cpu-oracle" for oracle/libsgc_oracle.so */
const char *sgc_backend(void);

/* ------------------------------------------------------------------------- *
 * 1. The four `dfa3D._ext` operators (pybind.cpp:42-67)
 * ------------------------------------------------------------------------- */

/* ms_depth_score_sample_forward (csrc/cuda/ms_depth_score_sample_cuda.cu:49-111,
 * kernel common/cuda/ms_depth_score_sample_cuda_kernel.cuh:24-148).
 *   dist   [B,S,M,D]        depth distributions (replicated per head, as the reference passes them)
 *   shapes3[L,3] int64      (H,W,D) per level;  lsi [L] int64 level start index
 *   loc3   [B,Q,M,L,P,3]    normalised (x=w, y=h, z=d) sampling locations
 *   score  [B,Q,M,L,P,4]    OUT (fully written): depth score at the 4 bilinear corners,
 *                           corner order [0]=(h0,w0) [1]=(h0,w1) [2]=(h1,w1) [3]=(h1,w0)
 */
int sgc_depth_score_forward(const float *dist, const int64_t *shapes3, const int64_t *lsi,
                            const float *loc3, float *score,
                            int B, int S, int M, int D, int L, int Q, int P, sgc_stream_t stream);

/* wms_deform_attn_forward (csrc/cuda/wms_deform_attn_cuda.cu:213-288, kernel
 * common/cuda/wms_deform_attn_cuda_kernel.cuh:24-80,240-303).
 *   value  [B,S,M,Cm];  shapes2 [L,2] int64 (H,W);  loc2 [B,Q,M,L,P,2];  attn [B,Q,M,L,P];
 *   score  [B,Q,M,L,P,4];  out [B,Q,M*Cm] OUT (fully written).
 */
int sgc_wms_forward(const float *value, const int64_t *shapes2, const int64_t *lsi,
                    const float *loc2, const float *attn, const float *score, float *out,
                    int B, int S, int M, int Cm, int L, int Q, int P, sgc_stream_t stream);

/* wms_deform_attn_backward (csrc/cuda/wms_deform_attn_cuda.cu:291-370, kernels
 * wms_deform_attn_cuda_kernel.cuh:82-159,305-531).
 *   grad_out   [B,Q,M*Cm]
 *   grad_value [B,S,M,Cm]      ACCUMULATED (+=, float atomics; caller zeroes)
 *   grad_loc2  [B,Q,M,L,P,2]   written: (W * d/dw, H * d/dh)
 *   grad_attn  [B,Q,M,L,P]     written
 *   grad_score [B,Q,M,L,P,4]   written
 */
int sgc_wms_backward(const float *value, const int64_t *shapes2, const int64_t *lsi,
                     const float *loc2, const float *attn, const float *score,
                     const float *grad_out, float *grad_value, float *grad_loc2,
                     float *grad_attn, float *grad_score,
                     int B, int S, int M, int Cm, int L, int Q, int P, sgc_stream_t stream);

/* ms_depth_score_sample_backward (csrc/cuda/ms_depth_score_sample_cuda.cu:137-201,
 * kernel ms_depth_score_sample_cuda_kernel.cuh:150-327).
 *   grad_score [B,Q,M,L,P,4]
 *   grad_dist  [B,S,M,D]       ACCUMULATED (+=; caller zeroes)
 *   grad_loc3  [B,Q,M,L,P,3]   written: (0, 0, D * sum_k g_k (v_d1 - v_d0)); the u,v
 *                              gradient through the score is dropped exactly as the
 *                              reference does (kernel.cuh:238-239).
 */
int sgc_depth_score_backward(const float *dist, const int64_t *shapes3, const int64_t *lsi,
                             const float *loc3, const float *grad_score,
                             float *grad_dist, float *grad_loc3,
                             int B, int S, int M, int D, int L, int Q, int P, sgc_stream_t stream);

/* ------------------------------------------------------------------------- *
 * 2. Fused 3D deformable attention (one launch instead of the reference's
 *    two-stage MultiScale3DDeformableAttnFunction_fp32,
 *    TU/multi_scale_3ddeformable_attn_function.py:275-351)
 * ------------------------------------------------------------------------- */

/* Forward.  dist is [B,S,dist_heads,D] with dist_heads == 1 (un-replicated; what the
 * MI355X host passes) or == M (the reference's `.repeat(1,1,num_heads,1)` layout,
 * TU/deformable_cross_attention.py:82,422).  attn may be NULL (= all ones, the
 * Grid_Sample_3D_Feature case, TU/deformable_cross_attention.py:85).
 * score_or_null: optional [B,Q,M,L,P,4] OUT to materialise the depth scores.      */
int sgc_dfa3d_forward(const float *value, const float *dist, const int64_t *shapes3,
                      const int64_t *lsi, const float *loc3, const float *attn,
                      float *out, float *score_or_null,
                      int B, int S, int M, int Cm, int D, int dist_heads,
                      int L, int Q, int P, sgc_stream_t stream);

/* Backward of the fused op == backward() of MultiScale3DDeformableAttnFunction_fp32
 * (TU/multi_scale_3ddeformable_attn_function.py:303-351):
 *   grad_value [B,S,M,Cm] ACCUMULATED;  grad_dist [B,S,dist_heads,D] ACCUMULATED (for
 *   dist_heads == 1 this is already the sum over heads that autograd of `.repeat`
 *   would produce);  grad_loc3 [B,Q,M,L,P,3] written (uv from the weighted gather,
 *   z from the depth score);  grad_attn_or_null [B,Q,M,L,P] written.               */
int sgc_dfa3d_backward(const float *value, const float *dist, const int64_t *shapes3,
                       const int64_t *lsi, const float *loc3, const float *attn,
                       const float *grad_out, float *grad_value, float *grad_dist,
                       float *grad_loc3, float *grad_attn_or_null,
                       int B, int S, int M, int Cm, int D, int dist_heads,
                       int L, int Q, int P, sgc_stream_t stream);

/* Item-list forms of the fused operator and its backward (training path): item i samples map item_batch[i] (int32,
 * a camera index < B) -- the reference's padded [N, max_len] rebatch (TU/deformable_cross_attention.py:759-773) without
 * its padding rows.  loc3 [n_items,M,L,P,3], attn [n_items,M,L,P] (NULL = 1), out / grad_out [n_items, M*Cm];
 * grad_value [B,S,M,Cm] and grad_dist [B,S,dist_heads,D] are ACCUMULATED (caller zeroes), grad_loc3 / grad_attn are
 * fully written.  Same arithmetic as sgc_dfa3d_forward / sgc_dfa3d_backward.                                       */
int sgc_dfa3d_forward_items(const float *value, const float *dist, const int64_t *shapes3, const int64_t *lsi,
                            const float *loc3, const float *attn_or_null, const int32_t *item_batch, float *out,
                            float *score_or_null, int B, int S, int M, int Cm, int D, int dist_heads, int L,
                            int n_items, int P, sgc_stream_t stream);
int sgc_dfa3d_backward_items(const float *value, const float *dist, const int64_t *shapes3,
                             const int64_t *lsi, const float *loc3, const float *attn_or_null,
                             const int32_t *item_batch, const float *grad_out, float *grad_value, float *grad_dist,
                             float *grad_loc3, float *grad_attn_or_null,
                             int B, int S, int M, int Cm, int D, int dist_heads,
                             int L, int n_items, int P, sgc_stream_t stream);

/* The same backward for ONE level over a BINNED item list (round 6; training path): items = the visible (camera, voxel) pairs in
 * the (camera, bin) order of sgc_bin_pairs, `bin_offset` [N * nbx * nby + 1] as sgc_bin_pairs writes it for bins of
 * bin_w x bin_h feature pixels.  A workgroup owns (camera, bin): the window's slice of grad_value and of grad_dist is accumulated
 * in LDS and flushed once (wms_deform_attn_cuda_kernel.cuh:82-159 and ms_depth_score_sample_cuda_kernel.cuh:150-241 add every corner
 * with a global atomic; TU/multi_scale_3ddeformable_attn_function.py:303-351 merges the two stages as this entry point does).
 * value [N,S,M,Cm], dist [N,S,D] (one depth map per camera: dist_heads == 1), loc3 [n_items, loc_heads, P, 3], attn
 * [n_items, loc_heads, P] or NULL (= 1), grad_out [n_items, M*Cm]; loc_heads = M, or 1 = ONE sample set shared by the M channel
 * groups (the geometry sample's single head over C = M * Cm channels); its gradients are then summed over the groups.
 * grad_value / grad_dist must be zero-filled by the caller (they are accumulated into); grad_loc3 / grad_attn (either may be NULL)
 * are written.  Cm in {16, 32}, P <= 4; the window (bin + halo) must fit LDS (sgc_dfa3d_backward_binned_lds_bytes <= 160 KiB).
 * head_shift_or_null [M][2] int32 (x, y) in pixels: head m's window is shifted by it (its mean sampling offset; speed only).
 * Same function of the inputs as sgc_dfa3d_backward_items; float atomics make the last bits order-dependent in both. */
int sgc_dfa3d_backward_binned(const float *value, const float *dist, const float *loc3, const float *attn_or_null,
                              const int32_t *bin_offset, const int32_t *head_shift_or_null, const float *grad_out, float *grad_value, float *grad_dist,
                              float *grad_loc3_or_null, float *grad_attn_or_null, int N, int S, int H, int W, int M, int Cm,
                              int D, int loc_heads, int P, int bin_w, int bin_h, int halo_x, int halo_y, sgc_stream_t stream);
int64_t sgc_dfa3d_backward_binned_lds_bytes(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y);

/* Geometry-aware sample FUSED with the Linear that consumes it (round 6): y[p] = (sum_k w_k(p) feat[cam(p), corner_k(p), :]) @ W^T +
 * shift -- `Grid_Sample_3D_Feature` (TU/deformable_cross_attention.py:67-116) followed by the fused offsets | logits projection of
 * `MSDeformableAttention3D_DFA3D.forward` (:417-436), without the [pairs, C] tensor between them.  Arguments as
 * sgc_pairs_geometry_sample + the bf16 hi / lo planes [Cout][C] of the weight; workspace >= sgc_pairs_geometry_linear_workspace_bytes(cap)
 * (32 bytes per pair: the sample's corner weights and rows).  Bit-identical to sgc_pairs_geometry_sample + sgc_linear_rows_bf16x3.
 * C in {128, 256} and Cout == 128 (sgc_pairs_geometry_linear_supported). */
int sgc_pairs_geometry_linear_bf16x3(const float *feat, const float *dist, const float *ref_cam, const int32_t *pair_cam,
                                     const int32_t *pair_q, const int32_t *totals, const uint16_t *w_hi, const uint16_t *w_lo,
                                     const float *shift_or_null, float *y, void *workspace, int N, int Nq, int H, int W, int C,
                                     int D, int Cout, int cam_stride_or_0, int n_pairs_or_neg, int cap, sgc_stream_t stream);
int sgc_pairs_geometry_linear_supported(int C, int Cout, int N, int S);
int64_t sgc_pairs_geometry_linear_workspace_bytes(int cap);

/* ------------------------------------------------------------------------- *
 * 3. Voxel -> pixel projection and per-camera compaction
 *    (replaces VoxFormerEncoder_DFA3D.point_sampling, TU/encoder.py:179-223, and
 *     the per-camera nonzero / rebatch loops, TU/deformable_cross_attention.py:759-773)
 * ------------------------------------------------------------------------- */

/* ref3d [Nvox,3] voxel reference points (DenseHead.ref_3d, WITHOUT origin); sel_or_null [Nq] int64: the q-th query
 * is voxel sel[q] (DenseHead.py:66 + transformer.py:145-146 gather; NULL: query q = row q, Nvox = Nq);
 * origin[3], proj [N,3,4] = (K' @ E_i[:3]) -- all device fp32.
 * ref_cam [N,Nq,3] OUT = (u/img_w, v/img_h, zn = (z-d_near)/(d_far-d_near));
 * mask [N,Nq] uint8 OUT = zn>eps & eps<u<1-eps & eps<v<1-eps  (eps = 1e-5; the reference's depth test runs on the
 * slice it has already overwritten with zn, TU/encoder.py:203-213).
 * Arithmetic order is fixed and documented in DESIGN.md (no FMA contraction).      */
int sgc_project_points(const float *ref3d, const int64_t *sel_or_null, const float *origin, const float *proj,
                       float *ref_cam, uint8_t *mask,
                       int N, int Nq, float img_w, float img_h, float d_near, float d_far,
                       sgc_stream_t stream);

/* Compaction of mask[N,Nq] into the (camera, query) pair list, camera-major, query
 * ascending inside a camera == concatenation of the reference's `indexes[i]`.
 *   cam_count [N] int32 OUT, cam_offset [N+1] int32 OUT (exclusive scan),
 *   pair_cam / pair_q [cap] int32 OUT (first n_pairs entries valid),
 *   slot [N,Nq] int32 OUT: pair index of (cam,q) or -1,
 *   vox_count [Nq] int32 OUT: #cameras seeing q, valid_index [Nq] int32 OUT: ascending
 *   q with count>0 (reference `valid_index`, TU/deformable_cross_attention.py:822),
 *   totals [4] int32 OUT: {n_pairs, n_valid, max_len, 0}.
 * workspace: >= (N*Nq + Nq + 2*N + 64) int32 (may be null); on return its first Nq entries hold
 *   row_of [Nq]: the inverse of valid_index (row of q in the compact list of seen voxels, -1 if
 *   no camera sees q) -- the gather index of sgc_level_tail.                                      */
int sgc_compact_pairs(const uint8_t *mask, int N, int Nq,
                      int32_t *cam_count, int32_t *cam_offset,
                      int32_t *pair_cam, int32_t *pair_q, int32_t *slot,
                      int32_t *vox_count, int32_t *valid_index, int32_t *totals,
                      int32_t *workspace, sgc_stream_t stream);

/* ------------------------------------------------------------------------- *
 * 4. Pair-list forms of the gather (no padded `max_len` rebatch, no dense slots)
 * ------------------------------------------------------------------------- */

/* Geometry-aware sample == Grid_Sample_3D_Feature (TU/deformable_cross_attention.py:67-116)
 * evaluated only on visible pairs:  feat [N,S,C], dist [N,S,D], one level (H,W);
 * out[p,:] = depth-weighted bilinear sample of camera pair_cam[p] at ref_cam[pair_cam[p], pair_q[p]].
 *   ref_cam [N,Nq,3];  out [n_pairs,C] fully written.  n_pairs is read from
 *   totals[0] on the device when n_pairs_or_neg < 0 (then `cap` bounds the grid).   */
int sgc_pairs_geometry_sample(const float *feat, const float *dist, const float *ref_cam,
                              const int32_t *pair_cam, const int32_t *pair_q,
                              const int32_t *totals, float *out,
                              int N, int Nq, int H, int W, int C, int D, int cam_stride_or_0,
                              int n_pairs_or_neg, int cap, sgc_stream_t stream);

/* Context-aware deformable gather == the DFA3D call of MSDeformableAttention3D_DFA3D
 * (TU/deformable_cross_attention.py:423-489) with the softmax over the L*P points and
 * the `ref + offset / (W,H,D)` location arithmetic fused in (one level, L = 1):
 *   value [N,S,M,Cm] (value_proj output), dist [N,S,D],
 *   raw [n_pairs, M*P*4]: per pair the three Linear outputs laid out as
 *       [ M*P*2 uv offsets (m,p,xy) | M*P depth offsets (m,p) | M*P attention logits (m,p) ],
 *   out [n_pairs, M*Cm] fully written.
 *   dist_pairs_or_null: optional pair-interleaved copy of dist made by sgc_depth_pairs -- same results,
 *   half the depth load instructions (the depth taps of one image row become 16 contiguous bytes).
 *   value_has_zero_row != 0: the caller appended ONE all-zero row after the N*S rows of `value`
 *   (value then holds N*S+1 rows); corners outside the image are pointed at it instead of being zeroed
 *   by a select per load -- same results, fewer instructions on the load path.                     */
int sgc_pairs_deform_gather(const float *value, const float *dist, const float *dist_pairs_or_null,
                            const float *ref_cam,
                            const float *raw, const int32_t *pair_cam, const int32_t *pair_q,
                            const int32_t *totals, float *out,
                            int N, int Nq, int H, int W, int M, int Cm, int D, int P, int cam_stride_or_0,
                            int value_has_zero_row, int n_pairs_or_neg, int cap, sgc_stream_t stream);

/* ---- LDS-tiled form of the same gather (the hot-path default where the shape allows) ----------------------
 *
 * sgc_bin_pairs: REORDERS the visible pairs of every camera by the feature pixel their reference point projects to
 * (a stable counting sort per camera; the camera-major layout of sgc_compact_pairs is kept, so pair_cam and
 * cam_offset stay valid).
 *   ref_cam [N,Nq,3]; pair_cam / pair_q [cap], cam_offset [N+1] from sgc_compact_pairs (counts stay on the device).
 *   bin(u, v) = (clamp(floor(v*H - 0.5), 0, H-1) / bin_h) * ceil(W / bin_w) + clamp(floor(u*W - 0.5), 0, W-1) / bin_w
 *   (fp32, no FMA contraction).  nb = ceil(W/bin_w) * ceil(H/bin_h) <= 1024 bins per camera.
 *   OUT pair_q_out [cap] int32 (must not alias pair_q): the queries in the new order -- grouped by (camera, bin),
 *       ascending ORIGINAL pair index (= ascending query) inside a group, identical from run to run;
 *   IN/OUT slot [N,Nq] int32: rewritten to the new pair index of every visible (camera, query);
 *   OUT pair_ref [cap][4] fp32, 16-byte aligned: (u, v, zn, bit pattern of the int32 query) of every pair, new order;
 *   OUT bin_offset [N*nb + 1] int32: first pair of every (camera, bin) group; the last entry is n_pairs;
 *   workspace: sgc_bin_pairs_workspace_bytes(...) bytes, 16-byte aligned.                                       */
int sgc_bin_pairs(const float *ref_cam, const int32_t *pair_cam, const int32_t *pair_q, const int32_t *cam_offset,
                  int32_t *pair_q_out, int32_t *slot, float *pair_ref, int32_t *bin_offset, void *workspace,
                  int N, int Nq, int cap, int H, int W, int bin_w, int bin_h, sgc_stream_t stream);
int64_t sgc_bin_pairs_workspace_bytes(int N, int Nq, int cap, int H, int W, int bin_w, int bin_h);

/* sgc_pairs_deform_gather_tiled: the operator of sgc_pairs_deform_gather (MSDeformableAttention3D_DFA3D's DFA3D
 * call, TU/deformable_cross_attention.py:423-489; one level, softmax over the P points and `ref + offset/(W,H,D)`
 * fused in) on a BINNED pair list with HEAD-MAJOR operands.  One workgroup per (camera, bin) walks the heads and
 * stages each head's window of the value map (and the camera's depth window) in LDS:
 *   value_hm [N][M][S][Cm]          value_proj output as written by sgc_linear_rows_headmajor_bf16x3: fp32, or -- with
 *                                   value_bf16 != 0, the opt-in bf16 STORAGE mode -- bfloat16 (half the map bytes and
 *                                   half the LDS per window; taps are widened to fp32, accumulation and output fp32)
 *   dist     [N][S][D]              depth distributions: fp32, or -- ABI version 4 -- bfloat16 TOO when value_bf16 != 0
 *                                   (the storage mode covers both maps: BASELINE.json config #5's reduced-precision twin,
 *                                   TU/multi_scale_3ddeformable_attn_function.py:353-428, casts value and value_dpt_dist)
 *   pair_ref / bin_offset           from sgc_bin_pairs with the same (H, W, bin_w, bin_h)
 *   raw_hm   [n_pairs][M][P][4]     per (pair, head, point): (du, dv, dz, attention logit) -- the three Linear
 *                                   outputs of sgc_pairs_deform_gather's `raw`, columns permuted head-major; rows in
 *                                   the binned pair order
 *   head_shift_or_null [M][2] int32 per-head shift (x, y) in pixels of the staged window (a head's mean sampling
 *                                   offset), |shift| <= max_shift_x / max_shift_y; speed only
 *   out      [n_pairs][M*Cm]        fully written for every pair, binned pair order
 * The staged window is the bin + halo_x / halo_y pixels on each side (clipped to the map); samples outside it are
 * served from global memory: results do not depend on bin_w / bin_h / halo / head_shift.  P == 4, Cm in {16, 32},
 * depth_in_lds != 0: the depth taps are served from an LDS copy of the camera's depth window as well (when it fits;
 * pays when many pairs share a bin).  D >= 2, H*W < 32767; the windows must fit 160 KB of LDS (sgc_tile_window
 * reports what would be staged).                                                                                 */
int sgc_pairs_deform_gather_tiled(const void *value_hm, int value_bf16, const void *dist, const float *pair_ref,
                                  const int32_t *bin_offset, const float *raw_hm, const int32_t *head_shift_or_null,
                                  float *out, int N, int H, int W, int M, int Cm, int D, int P,
                                  int cam_stride_or_0, int bin_w, int bin_h, int halo_x, int halo_y,
                                  int max_shift_x, int max_shift_y, int depth_in_lds, sgc_stream_t stream);
/* (host-side helper, no launch, HOST pointers) value window, LDS bytes, number of value buffers (2 = the next head's
 * window is loaded while the current head is computed) and whether the depth window is staged too */
int sgc_tile_window(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y, int max_shift_x,
                    int max_shift_y, int depth_in_lds, int value_bf16, int *tw_out, int *th_out, int *lds_bytes_out, int *nbuf_out,
                    int *depth_in_lds_out);

/* dp [N,H,W+1,D,2]: dp[n][h][wq][d] = (dist[n][h][wq-1][d] or 0, dist[n][h][wq][d] or 0); dist [N,H*W,D]. */
int sgc_depth_pairs(const float *dist, float *dp, int N, int H, int W, int D, int cam_stride_or_0,
                    sgc_stream_t stream);

/* `cam_stride_or_0` (the three entry points above): number of pixels between consecutive cameras in the
 * channels-last maps (feat / value / dist); 0 = H*W (compact).  A producer that emits channels-last FPN and depth
 * maps hands them over without the NCHW -> NHWC pass: the maps keep the rows that
 * AdaptiveSparseHead.py:53-59 crops away, i.e. cam_stride = Hs*Ws >= H*W with W == Ws, and `value` holds
 * N*cam_stride (+1 zero) rows.                                                                         */

/* ------------------------------------------------------------------------- *
 * 5. Inter-view aggregation (TU/deformable_cross_attention.py:815-837)
 * ------------------------------------------------------------------------- */

/* Row counts that live on the device.  sgc_compact_pairs leaves {n_pairs, n_valid, ...} in totals[]; the entry
 * points below take an optional `*_dev_or_null` pointer to such a count next to the host-side integer: when it
 * is non-NULL the kernels use min(host value, *device value) rows and the host value only bounds the grid
 * (pass the capacity).  A whole scene can then be issued -- or captured into one hipGraph -- without the
 * reference's per-level host read-back (`nonzero`, DenseHead.py:66, TU/deformable_cross_attention.py:759-762).  */

/* Masked mean over the cameras that see a voxel (:819-826):
 *   feat [n_pairs,C], slot [N,Nq], valid_index [n_valid] -> mean [n_valid,C].      */
int sgc_view_mean(const float *feat, const int32_t *slot, const int32_t *valid_index,
                  float *mean, int N, int Nq, int C, const int32_t *n_valid_dev_or_null, int n_valid,
                  sgc_stream_t stream);

/* Softmax over views of nn.MultiheadAttention with query length 1 (:829-833):
 *   q [n_valid,C] (already in-projected, NOT yet scaled), kv [n_pairs,2C] (k | v
 *   in-projected per visible pair), heads -> ctx [n_valid,C] (before out_proj).
 *   Invisible cameras are the reference's key_padding_mask = -inf entries.          */
int sgc_view_attend(const float *q, const float *kv, const int32_t *slot,
                    const int32_t *valid_index, float *ctx,
                    int N, int Nq, int C, int heads, const int32_t *n_valid_dev_or_null, int n_valid,
                    sgc_stream_t stream);
/* Its backward over the same pair list (training; the reference back-propagates through nn.MultiheadAttention on the
 * dense [N, L, C] slots, :829-833): grad_ctx [n_valid,C] -> grad_q [n_valid,C] (w.r.t. the un-scaled q), grad_kv
 * [n_pairs,2C] (k | v gradients per visible pair; every pair row is written exactly once).                          */
int sgc_view_attend_backward(const float *q, const float *kv, const int32_t *slot, const int32_t *valid_index,
                             const float *ctx, const float *grad_ctx, float *grad_q, float *grad_kv,
                             int N, int Nq, int C, int heads, int n_valid, sgc_stream_t stream);

/* The same attention with the K / V in-projections moved off the pair list (round 5; a new entry point: no ABI bump).  Both commute with the
 * softmax over views: score(n, h) = (scale W_k,h^T q_h) . x_n + a constant per (voxel, head) that the softmax drops, and
 * ctx_h = W_v,h (sum_n a(n, h) x_n) + b_v,h.  So instead of a [pairs, C] x [C, 2C] GEMM (2.1 M pairs per scene at 100 views) the
 * caller projects the QUERY once per voxel and head,
 *   qp [n_valid, heads, C] = scale * W_k,h^T q_h        (one Linear C -> heads * C on the voxels),
 * this entry point computes, on the RAW per-pair features x [n_pairs, C],
 *   a(n, h) = softmax_n (qp_h . x_n)    over the cameras n that see the voxel (slot[n, q] >= 0),
 *   s [n_valid, heads, C],  s_h = sum_n a(n, h) x_n,
 * and the caller applies V once per voxel: ctx[h * hd + j] = W_v[h * hd + j, :] . s_h + b_v (a block-diagonal Linear).
 * Same function of (q, x, weights) as sgc_view_attend on the in-projected tensors (TU/deformable_cross_attention.py:826-833);
 * the sums are associated differently (~1e-6 relative).  Supported: heads == 8, C in {128, 256}, N <= 128
 * (sgc_view_attend_pq_supported); the caller keeps sgc_view_attend for everything else.                                   */
int sgc_view_attend_pq(const float *qp, const float *x, const int32_t *slot, const int32_t *valid_index, float *s,
                       int N, int Nq, int C, int heads, const int32_t *n_valid_dev_or_null, int n_valid,
                       sgc_stream_t stream);
int sgc_view_attend_pq_supported(int N, int C, int heads);

/* ------------------------------------------------------------------------- *
 * 6. Volume glue
 * ------------------------------------------------------------------------- */

/* rows [n,C] scattered to vol[idx[i],:] (DenseHead.forward scatter, DenseHead.py:80-81,
 * and `output[:,valid_index,:] = slots_mean`, TU/deformable_cross_attention.py:835-836).
 * idx2_or_null composes two index maps: dst row = idx2[idx[i]].                     */
int sgc_scatter_rows(const float *rows, const int32_t *idx, const int32_t *idx2_or_null,
                     float *vol, const int32_t *n_dev_or_null, int n, int C, sgc_stream_t stream);

/* NCHW -> NHWC crop-and-transpose of the FPN / depth maps
 * (TU/transformer.py:151-170 flatten+permute, AdaptiveSparseHead.py:53-59 crop):
 *   src [N,C,Hs,Ws] -> dst [N,H*W,C], dst[n, h*W + w, c] = src[n, c, h*step, w*step] for h < H, w < W.
 *   step 2 / 4 reads the x1/2, x1/4 nearest-neighbour copies of the depth distribution (SGCDet.py:83-85) straight
 *   from the full-resolution map: they are never materialised.                                     */
int sgc_nchw_to_nhwc_crop(const float *src, float *dst, int N, int C, int Hs, int Ws,
                          int H, int W, int step, sgc_stream_t stream);
/* Its adjoint (training, round 5): dst [N, C, Hd, Wd] (NCHW, the whole plane written) from channels-last rows src [N, H*W, C]:
 * dst[n][c][h][w] = src[n][h*W + w][c] inside the H x W crop, 0 outside (Hd >= H, Wd >= W).  The gradient of the rows the path
 * samples w.r.t. the map the 2D stage holds (TU/transformer.py:151-170 flatten / permute; AdaptiveSparseHead.py:53-59 crop),
 * which autograd otherwise builds from two strided copies per level.                                                        */
int sgc_nhwc_to_nchw_pad(const float *src, float *dst, int N, int C, int H, int W, int Hd, int Wd, sgc_stream_t stream);

/* Coarse-to-fine glue of AdaptiveSparseHead on channels-last volumes (AdaptiveSparseHead.py:64-82):
 *   up [8*ix*iy*iz, C] = trilinear x2 upsample of vol [ix*iy*iz, C] (F.interpolate, align_corners=False);
 *   occ [8*ix*iy*iz]   = sigmoid(up . w + b)  (the occupancy head Sequential(Linear(C,1), Sigmoid)); w/b/occ
 *   may all be NULL to upsample only.                                                        */
int sgc_upsample2x_occ(const float *vol, const float *w_or_null, const float *b_or_null, float *up,
                       float *occ_or_null, int ix, int iy, int iz, int C, sgc_stream_t stream);

/* Backward of that x2 trilinear upsample for the training path (the adjoint of F.interpolate(scale_factor=2,
 * mode='trilinear', align_corners=False), AdaptiveSparseHead.py:64-69) on NCDHW planes, as a gather:
 *   grad_in [C, X, Y, Z] <- grad_out [C, 2X, 2Y, 2Z]; per axis an input index i collects outputs 2i-1, 2i, 2i+1,
 *   2i+2 with weights .25, .75, .75, .25 (border outputs 0 and 2n-1 carry weight 1 on their single source).
 *   torch's upsample_trilinear3d_backward scatters with float atomics (5.2 ms per config-2 step); this reads
 *   every output 8 times from L2 and writes every input once, bit-reproducibly.                              */
int sgc_upsample2x_backward(const float *grad_out, float *grad_in, int C, int X, int Y, int Z, sgc_stream_t stream);

/* vol[idx[i], :] += rows[i, :] -- `upsampled_volume + DenseHead(...)` where the dense head's output is zero
 * outside the selected voxels (AdaptiveSparseHead.py:77-82, DenseHead.py:80-81); idx int64, distinct.   */
int sgc_scatter_add_rows(const float *rows, const int64_t *idx, float *vol, int n, int C, sgc_stream_t stream);

/* ------------------------------------------------------------------------- *
 * 7. Dense 3D convolution of the neck / head on channels-last volumes
 *    (FastIndoorImVoxelNeck, necks/imvoxelnet.py:36-64,146-173; head convs,
 *     dense_heads/imvoxel_head_v2.py:75-78 -- cuDNN / MIOpen in the reference)
 * ------------------------------------------------------------------------- */

/* t = (sum_{tap,ci} x[nbr(v,tap), ci] * wt[tap][co][ci]) * scale[co] + shift[co];
 * relu = 0: y = t + residual;  relu = 1: y = max(t + residual, 0)  (BasicBlock3dV2, imvoxelnet.py:161-173);
 * relu = 2: y = max(t, 0) + residual  (decoder: up_block then skip add, imvoxelnet.py:29-31)
 *   x  [ix*iy*iz, Cin]  channels-last volume, voxel index (x*iy + y)*iz + z  (== torch [1,C,X,Y,Z]
 *      in channels_last_3d memory format);  Cin % 32 == 0;
 *   wt [taps][Cout][Cin]: taps = ksize^3 with tap = (kx*ksize + ky)*ksize + kz (nn.Conv3d weight
 *      [Cout,Cin,kx,ky,kz] permuted), or, transposed = 1, the 8 parities (px*2+py)*2+pz of
 *      nn.ConvTranspose3d(k=2, s=2) weight [Cin,Cout,2,2,2];
 *   scale/shift [Cout] or NULL: folded eval-mode BatchNorm3d (or bias); residual [OV,Cout] or NULL;
 *   ksize in {1,3} (pad = ksize/2), stride in {1,2}, or ksize 2 with stride 2 and no padding (the adjoint geometry of
 *   nn.ConvTranspose3d(2, 2): its input gradient);  y [ox*oy*oz, Cout] fully written.
 * fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).           */
int sgc_conv3d_cl_f32(const float *x, const float *wt, const float *scale, const float *shift,
                      const float *residual_or_null, float *y,
                      int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                      int transposed, int relu, float *workspace_or_null, int64_t workspace_floats,
                      sgc_stream_t stream);

/* Same contract on the bf16 matrix cores with fp32-faithful results: every fp32 operand is split
 * as v = hi + lo (two bf16); the products hi*hi + hi*lo + lo*hi run on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation (the dropped lo*lo term is 2^-16 relative).  w_hi / w_lo [taps][Cout][Cin] are the
 * host-side split of the fp32 weights (raw bf16 bit patterns): w_hi = bf16_rne(w), w_lo = bf16_rne(w - w_hi).
 * Activations are fp32 in memory and split while staged into LDS.  Agreement with sgc_conv3d_cl_f32:
 * ~1e-5 of the tensor scale (tests: 1e-4).  The kernels address x and the weights through 32-bit
 * buffer offsets: x must stay below 4 GiB and the weight tensor below 2 GiB (SGC_EUNSUP otherwise;
 * also sgc_conv2d_nhwc_bf16x3, sgc_linear_rows_*_bf16x3, sgc_conv3d_wgrad_bf16x3 for x and dy).   */
int sgc_conv3d_cl_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                         const float *shift, const float *residual_or_null, float *y,
                         int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                         int transposed, int relu, float *workspace_or_null, int64_t workspace_floats,
                         sgc_stream_t stream);

/* Layers with few output voxels split their reduction over several workgroups (split-K).  With a workspace of
 * sgc_conv3d_workspace_floats(...) floats every split stores its partial tile and a second kernel adds them in a
 * fixed order: results are bit-identical from run to run (and to any other launch order).  Without it
 * (workspace_or_null = NULL or too small) the partial tiles meet in `y` through float atomics -- same values up to
 * the rounding of a different summation order.  The query returns 0 for layers that are not split.            */
int64_t sgc_conv3d_workspace_floats(int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                                    int transposed, int bf16x3);

/* 2-D convolution over a stack of channels-last images -- the producer side of the hand-over: the output
 * convolutions of the image FPN (mmdet `FPN.fpn_convs` / `lateral_convs` as configured in configs/SGCDet_ScanNet.py:84-88
 * and called at detectors/SGCDet.py:67) emitting the [N, H*W, C] rows the view transformation consumes
 * (TU/transformer.py:151-170) with no NCHW round trip.
 *   x [N*H*W, Cin] -> y [N*H*W, Cout]; w_hi / w_lo [ksize^2][Cout][Cin] (nn.Conv2d weight [Cout,Cin,ky,kx] permuted, split as
 *   for sgc_conv3d_cl_bf16x3); ksize in {1,3} with padding ksize/2, stride 1; scale / shift / residual / relu as above;
 *   Cin % 32 == 0, Cout % 4 == 0.                                                                                       */
int sgc_conv2d_nhwc_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                           const float *shift, const float *residual_or_null, float *y, int N, int H, int W,
                           int Cin, int Cout, int ksize, int relu, sgc_stream_t stream);

/* Weight gradient of the same convolutions (the neck / head layers in training; the reference gets it
 * from cuDNN through autograd of nn.Conv3d, necks/imvoxelnet.py:36-64):
 *   dw[tap][co][ci] = sum over output voxels o of dy[o][co] * x[nbr(o, tap)][ci]      (bf16x3 arithmetic, fp32 accumulate)
 *   x [ix*iy*iz, Cin], dy [ox*oy*oz, Cout] channels-last rows; dw [ksize^3][Cout][Cin] = nn.Conv3d.weight.grad permuted
 *   like the forward weights.  ksize / stride as above (ksize 2 = the ConvTranspose3d(2,2) layers with x := the fine-grid
 *   tensor, dy := the coarse one); Cin % 4 == 0, Cout % 4 == 0.  The voxel range is split over workgroups; with a
 *   workspace of sgc_conv3d_wgrad_workspace_floats(...) floats the partial sums are added in a fixed order (run-to-run
 *   bit-identical); without it one workgroup walks the whole range (slower).  The INPUT gradient needs no entry point of
 *   its own: it is sgc_conv3d_cl_bf16x3 on dy with the taps mirrored and Cin/Cout swapped (stride 1), on the
 *   zero-interleaved dy (stride 2), or with ksize 2 / stride 2 (transposed layers) -- sgcdet_amd/functions.py.     */
int sgc_conv3d_wgrad_bf16x3(const float *x, const float *dy, float *dw, int ix, int iy, int iz, int Cin, int Cout,
                            int ksize, int stride, float *workspace_or_null, int64_t workspace_floats,
                            sgc_stream_t stream);
int64_t sgc_conv3d_wgrad_workspace_floats(int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride);

/* Output-masked form of the 3x3x3 stride-1 convolution (north star: "sparse 3D convolution over the occupancy-masked
 * voxels").  The reference's volume is dense, so its convolutions are dense (necks/imvoxelnet.py:47-64); but the
 * head's outputs are only consumed where `valid` (dense_heads/imvoxel_head_v2.py:258,301), hence the finest-level
 * tail needs: the head convolution on valid, out_block_0 on dilate(valid), up_block_1's 3x3x3 on dilate^2(valid).
 *   out_mask [OX*OY*OZ] uint8 {0,1}: rows with 1 are BIT-IDENTICAL to sgc_conv3d_cl_bf16x3; rows with 0 hold the
 *   epilogue of a zero accumulator or the dense value (finite, deterministic, not to be consumed).  64-voxel tiles
 *   without a live row skip their matrix work, 256-voxel bricks without one skip everything.  Layers that do not run
 *   on the brick kernel (few voxels, narrow outputs) are computed dense.                                          */
int sgc_conv3d_cl_bf16x3_masked(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                                const float *shift, const float *residual_or_null, float *y, const uint8_t *out_mask,
                                int ix, int iy, int iz, int Cin, int Cout, int relu,
                                float *workspace_or_null, int64_t workspace_floats, sgc_stream_t stream);
/* The same 3x3x3 stride-1 convolution with an output activation on a column range (round 5): columns [act_c0, act_c1) leave as
 * expf(v * *act_scale_dev), v being the epilogue's result (scale / shift / relu / residual as above) -- ImVoxelHeadV2's
 * `torch.exp(self.scales[i](reg_conv(x)))` (dense_heads/imvoxel_head_v2.py:79,103-110; mmcv Scale = a learnable scalar, read from
 * the device) inside the fused centerness | reg | cls convolution instead of two elementwise launches per scale.
 * out_mask_or_null as in sgc_conv3d_cl_bf16x3_masked.  All other columns are bit-identical to sgc_conv3d_cl_bf16x3.            */
int sgc_conv3d_cl_bf16x3_act(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                             const float *shift, const float *residual_or_null, float *y, const uint8_t *out_mask_or_null,
                             int ix, int iy, int iz, int Cin, int Cout, int relu, int act_c0, int act_c1,
                             const float *act_scale_dev, float *workspace_or_null, int64_t workspace_floats,
                             sgc_stream_t stream);
/* The 3x3x3 stride-1 convolution through a Winograd F(2,3) transform ALONG Z (round 5): 18 instead of 27 tap-GEMMs per output --
 * fewer multiply-adds is the lever left on layers that run at the package power limit.  Same operator as sgc_conv3d_cl_bf16x3
 * (ksize 3, stride 1; nn.Conv3d + folded BatchNorm + residual + ReLU of necks/imvoxelnet.py:36-64,146-173), same epilogue; the sums
 * are associated differently (tests bound the difference at 2e-5 of the tensor scale).
 *   wg_hi / wg_lo: bf16 hi / lo planes of the TRANSFORMED weights [4][9][Cout][Cin]: for the (dx, dy) tap t = dx*3 + dy and the z taps
 *   w0, w1, w2 of the module's weight: G[0][t] = w0, G[1][t] = (w0 + w1 + w2) / 2, G[2][t] = (w0 - w1 + w2) / 2, G[3][t] = w2.
 *   workspace: >= sgc_conv3d_winograd_z_workspace_floats() floats (the transform-domain input and output stacks).
 *   Supported (sgc_conv3d_winograd_z_supported): ix, iy >= 8, iz % 8 == 0, Cin % 32 == 0, Cout % 4 == 0, Cout > 64, 2 * ix * iy * iz >= 2048. */
int sgc_conv3d_winograd_z_bf16x3(const float *x, const uint16_t *wg_hi, const uint16_t *wg_lo, const float *scale,
                                 const float *shift, const float *residual_or_null, float *y, int ix, int iy, int iz,
                                 int Cin, int Cout, int relu, float *workspace, int64_t workspace_floats, sgc_stream_t stream);
int sgc_conv3d_winograd_z_supported(int ix, int iy, int iz, int Cin, int Cout);
int64_t sgc_conv3d_winograd_z_workspace_floats(int ix, int iy, int iz, int Cin, int Cout);
/* mask_out = 3x3x3 dilation of mask_in ([X*Y*Z] uint8 {0,1}, flat index (x*Y + y)*Z + z); must not alias. */
int sgc_mask_dilate3(const uint8_t *mask_in, uint8_t *mask_out, int X, int Y, int Z, sgc_stream_t stream);
/* Head valid mask of scale `factor` (1, 2, 4): nn.Upsample(size, mode='trilinear')(valid.float()).round().bool()
 * (imvoxel_head_v2.py:123,258) as uint8 [(X/f)*(Y/f)*(Z/f)] from valid [X*Y*Z] int64 {0,1}.                     */
int sgc_valid_pyramid(const int64_t *valid, uint8_t *mask_out, int X, int Y, int Z, int factor, sgc_stream_t stream);

/* nn.Linear over a row list whose length lives on the device (the Linears of
 * MSDeformableAttention3D_DFA3D / nn.MultiheadAttention applied to the visible pairs,
 * TU/deformable_cross_attention.py:423-436,829-833):
 *   y[r, :] = x[r, :] @ W^T + shift  for r < min(rows_cap, *rows_dev_or_null);  rows past the count are neither
 *   read nor written.  W as w_hi / w_lo [Cout][Cin] (the split of sgc_conv3d_cl_bf16x3), Cin % 32 == 0,
 *   Cout % 4 == 0; same bf16x3 arithmetic.                                                            */
int sgc_linear_rows_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                           float *y, const int32_t *rows_dev_or_null, int rows_cap, int Cin, int Cout,
                           sgc_stream_t stream);

/* The same Linear with ONE extra all-zero row behind its result (round 5): y holds rows_cap + 1 rows, rows [0, count) are the
 * Linear, row rows_cap is set to zero by the same launch -- the row sgc_pairs_deform_gather's value_has_zero_row points
 * out-of-image corners at (a separate fill was one more launch per level).  rows_cap > 0.                                    */
int sgc_linear_rows_zrow_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                                float *y, const int32_t *rows_dev_or_null, int rows_cap, int Cin, int Cout,
                                sgc_stream_t stream);

/* value_proj (TU/deformable_cross_attention.py:417) with the result stored HEAD-MAJOR for the tiled gather:
 *   x [N*S][Cin] camera-major pixel rows -> y [N][M][S][Cm], y[n][h][s][j] = (x[n*S+s] @ W^T + shift)[h*Cm + j].
 *   Same arithmetic per element as sgc_linear_rows_bf16x3; Cin % 32 == 0, Cm % 4 == 0 and Cm | 128 (Cm | 64 when M * Cm <= 64): a head
 *   stays inside one column tile; other head sizes return SGC_EUNSUP (store row-major and permute instead).
 *   y_bf16 != 0: y is bfloat16 (round-to-nearest-even of the fp32 result) -- the opt-in bf16 storage mode.     */
int sgc_linear_rows_headmajor_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                                     void *y, int y_bf16, int N, int S, int Cin, int M, int Cm, sgc_stream_t stream);

/* Block-diagonal Linear over a row list (round 6): y[r][g * Nh + j] = sum_k x[r][g * K + k] * w[g][j][k] + shift[g * Nh + j]
 * -- the per-voxel V projection of the projected-query attention (head g multiplies the attention-weighted raw feature
 * sgc_view_attend_pq leaves for it with its own rows of nn.MultiheadAttention.in_proj_weight,
 * TU/deformable_cross_attention.py:826-833) without the 7/8 zero blocks of the dense [G K -> G Nh] form.
 *   x [rows_cap][G K] fp32, w_hi / w_lo [G][Nh][K] bf16 split, y [rows_cap][G Nh]; rows as in sgc_linear_rows_bf16x3.
 *   Same products and K order as sgc_linear_rows_bf16x3 on the dense block-diagonal matrix: BIT-IDENTICAL results.
 *   Supported: G == 8, K in {128, 256}, Nh == K / 8 (sgc_linear_rows_blockdiag_supported); SGC_EUNSUP otherwise. */
int sgc_linear_rows_blockdiag_supported(int G, int K, int Nh);
int sgc_linear_rows_blockdiag_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift_or_null,
                                     float *y, const int32_t *rows_dev_or_null, int rows_cap, int G, int K, int Nh,
                                     sgc_stream_t stream);

/* ------------------------------------------------------------------------- *
 * 7b. Row-wise glue of the coarse-to-fine head (no library kernels inside the scene graphs)
 * ------------------------------------------------------------------------- */

/* Hard top-k selection == topk_wo_grad + nonzero + get_valid (AdaptiveSparseHead.py:9-13,74,95-98; DenseHead.py:66):
 *   score [n] fp32 (occupancy of every voxel), 0 < k <= n;
 *   OUT idx_out [k] int64: flat indices of the k largest scores in ASCENDING index order (= nonzero(mask));
 *   OUT valid_or_null [n] int64 {0,1}, mask_or_null [n] fp32 {0,1}: the selection as a dense mask.
 * Ties at the cut are broken by the lowest flat index (torch.topk leaves that order implementation-defined; NaN
 * counts as the largest value, as in torch).  One launch, one workgroup, deterministic.                          */
int sgc_topk_select(const float *score, int n, int k, int64_t *idx_out, int64_t *valid_or_null, float *mask_or_null,
                    sgc_stream_t stream);
/* The same selection over many workgroups (candidate sets of the 80x80x32 / 96x96x32 configurations: one workgroup needs
 * ~0.5 ms for 204 800 scores): four histogram launches, a counting launch and an ordered compaction, identical outputs.
 * workspace: sgc_topk_select_workspace_bytes(n) bytes, contents irrelevant; NULL / too small / a small n: the one-workgroup
 * form above.                                                                                                          */
int sgc_topk_select_ws(const float *score, int n, int k, int64_t *idx_out, int64_t *valid_or_null, float *mask_or_null,
                       void *workspace_or_null, int64_t workspace_bytes, sgc_stream_t stream);
int64_t sgc_topk_select_workspace_bytes(int n);

/* nn.LayerNorm(C) over the first min(rows_cap, *rows_dev_or_null) rows of x [rows_cap, C] (the two norms of
 * VoxFormerLayer, TU/encoder.py:311-338): y = (x - mean) * rsqrt(var + eps) * gamma + beta, biased variance,
 * fp32 two-pass statistics.  y may alias x.                                                                    */
int sgc_layer_norm_rows(const float *x, const float *gamma, const float *beta, float eps, float *y,
                        const int32_t *rows_dev_or_null, int rows_cap, int C, sgc_stream_t stream);

/* Training-mode BatchNorm over channels-last rows x [rows, C]: nn.BatchNorm3d over [1, C, X, Y, Z] is the
 * per-channel statistics of the X*Y*Z rows (necks/imvoxelnet.py:36-64,146-173).
 *   forward : mean_out / invstd_out [C] = batch mean and 1 / sqrt(biased variance + eps); y = (x - mean) * invstd * weight + bias;
 *             running_mean / running_var (optional) are updated as nn.BatchNorm does (momentum, UNBIASED variance).
 *   backward: dbias = sum dy, dweight = sum dy * xhat, dx = weight * invstd * (dy - dbias / rows - xhat * dweight / rows).
 * Reductions run over slabs of rows whose partial results are merged in a fixed order (same bits every run); workspace:
 * sgc_bn_rows_workspace_floats(rows, C) floats for either call.  C % 4 == 0; 16-byte aligned pointers.                   */
int64_t sgc_bn_rows_workspace_floats(int rows, int C);
int sgc_bn_rows_forward(const float *x, const float *weight, const float *bias, float *running_mean_or_null,
                        float *running_var_or_null, float momentum, float eps, float *y, float *mean_out,
                        float *invstd_out, float *workspace, int64_t workspace_floats, int rows, int C,
                        sgc_stream_t stream);
int sgc_bn_rows_backward(const float *x, const float *dy, const float *mean, const float *invstd, const float *weight,
                         float *dx, float *dweight, float *dbias, float *workspace, int64_t workspace_floats, int rows,
                         int C, sgc_stream_t stream);
/* (round 5) The same passes with the elementwise tail of the neck's blocks inside them: y = relu?(bn(x) + residual?) -- `relu(norm(conv))`
 * and the ResBlock tail `relu(norm2(conv2) + identity)` (necks/imvoxelnet.py:36-64) -- instead of one or two more elementwise kernels per
 * layer and pass.  Backward: the incoming gradient counts where y > 0 (y_relu_or_null = the forward's output when relu was set; torch's
 * threshold_backward), then the plain BatchNorm backward; dresidual_or_null receives the gradient of the added identity (= the masked
 * incoming gradient).  Same statistics, same reduction order as the plain entry points. */
int sgc_bn_rows_act_forward(const float *x, const float *weight, const float *bias, float *running_mean_or_null,
                            float *running_var_or_null, float momentum, float eps, const float *residual_or_null, int relu,
                            float *y, float *mean_out, float *invstd_out, float *workspace, int64_t workspace_floats,
                            int rows, int C, sgc_stream_t stream);
int sgc_bn_rows_act_backward(const float *x, const float *dy, const float *y_relu_or_null, const float *mean, const float *invstd,
                             const float *weight, float *dx, float *dweight, float *dbias, float *dresidual_or_null,
                             float *workspace, int64_t workspace_floats, int rows, int C, sgc_stream_t stream);

/* The tail of a VoxFormer level in one launch: for every voxel q of the level
 *     x0 = out_proj(ctx[row_of[q]]) if row_of[q] >= 0 else 0      (nn.MultiheadAttention.out_proj + the slot scatter,
 *                                                                  TU/deformable_cross_attention.py:826-837)
 *     x1 = LayerNorm(x0; ln1)                                      (VoxFormerLayer "norm", TU/encoder.py:311-338)
 *     x2 = W2 relu(W1 x1 + b1) + b2 + x1                           (mmcv FFN, "ffn")
 *     out[q] = LayerNorm(x2; ln2)                                  ("norm")
 * i.e. sgc_linear_rows_bf16x3 + sgc_scatter_rows + sgc_layer_norm_rows + 2 x sgc_conv3d_cl_bf16x3 (1x1x1) +
 * sgc_layer_norm_rows with the intermediates kept on chip; same arithmetic, BIT-IDENTICAL results (bf16x3 products).
 *   ctx [rows, C] fp32 compact rows (view_attend's output), row_of [Nq] int32 (sgc_compact_pairs), weights split as
 *   bf16 hi / lo: wo [C][C], w1 [F][C], w2 [C][F], each FRAGMENT-PACKED: a row-major [N][K] matrix W is passed as
 *   P[N/32][K/16][64][8] with P[b][kk][l][j] = W[32 b + (l & 31)][16 kk + 8 (l >> 5) + j] (the 16 bytes lane l feeds to the
 *   32x32x16 MFMA of k-step kk: one coalesced 1 KiB access per wave instead of 32 row pieces; TensorOps.pack_b_fragments);
 *   biases / LayerNorm parameters fp32; out [Nq, C].
 *   Supported: C in {128, 256} and F == 2 C (sgc_level_tail_supported); SGC_EUNSUP otherwise.                   */
int sgc_level_tail_supported(int C, int F);
int sgc_level_tail(const float *ctx, const int32_t *row_of, const uint16_t *wo_hi, const uint16_t *wo_lo, const float *bo,
                   const float *ln1_gamma, const float *ln1_beta, float eps1, const uint16_t *w1_hi, const uint16_t *w1_lo,
                   const float *b1, const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, const float *ln2_gamma,
                   const float *ln2_beta, float eps2, float *out, int Nq, int C, int F, sgc_stream_t stream);

/* Weight layout passes of the training step (row f-3), one launch each instead of a strided torch copy + three
 * conversion kernels per layer and pass:
 *   sgc_pack_conv_weight: parameter w [A][B][T] fp32 (nn.Conv3d [Cout][Cin][k^3]; nn.ConvTranspose3d [Cin][Cout][8];
 *     nn.Linear T = 1) -> w_hi / w_lo [T][R][C] bf16 (hi = bf16_rne(w), lo = bf16_rne(w - hi)) with
 *     out[t][r][c] = w[a][b][flip ? T-1-t : t], (a, b) = transpose ? (c, r) : (r, c); R >= rows, C >= cols: zero padding.
 *   sgc_unpack_conv_wgrad: the inverse map without the split, for the weight gradient: dw_trc [T][R][C] fp32 -> dw [A][B][T]. */
int sgc_pack_conv_weight(const float *w, uint16_t *w_hi, uint16_t *w_lo, int A, int B, int T, int R, int C, int transpose,
                         int flip, sgc_stream_t stream);
int sgc_unpack_conv_wgrad(const float *dw_trc, float *dw, int A, int B, int T, int R, int C, int transpose, int flip,
                          sgc_stream_t stream);
/* (round 5) sgc_pack_conv_weight for a LIST of parameters in one launch -- a training step packs every parameter of the path in
 * two forms (forward and input-gradient layout), ~100 launches of mostly a few microseconds of work; the host side keeps the
 * planes and the item list across steps and repacks all of them at once when the parameters have changed.
 *   items: n_items descriptors in DEVICE memory (host memory for the oracle); block_start ascending from 0, item i owning
 *   sgc_pack_conv_weight_blocks(A, B, T, transpose) workgroups; total_blocks = their sum; max_T = the largest T.
 *   The padding of the planes (R, C beyond the matrix) is NOT written: allocate them zeroed, once. */
typedef struct sgc_pack_item {
  const float *w;                 /* parameter [A][B][T] */
  uint16_t *hi, *lo;              /* planes [T][R][C] */
  int32_t A, B, T, R, C, transpose, flip, block_start;
  int32_t reserved[2];
} sgc_pack_item;                  /* 64 bytes */
int sgc_pack_conv_weight_blocks(int A, int B, int T, int transpose);
int sgc_pack_conv_weight_batch(const void *items, int n_items, int total_blocks, int max_T, sgc_stream_t stream);

/* ------------------------------------------------------------------------- *
 * 8. Post-processing
 * ------------------------------------------------------------------------- */

/* Greedy NMS of axis-aligned 3D boxes == mmdet3d `aligned_3d_nms`
 * (packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py:131-178; called by
 * ScanNetImVoxelHeadV2._nms, mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py:437-443):
 *   boxes [n,6] (x1,y1,z1,x2,y2,z2) fp32, labels [n] int64, order [n] int64 = argsort(scores) ASCENDING (the
 *   reference's `torch.argsort(scores)`; the best box is order[n-1]);  a lower-scored box is dropped when, against a
 *   kept box, NOT (iou * (same class) <= iou_thr) -- exactly the reference's comparison, NaN IoU included.
 *   keep [n] int64 OUT: indices of the kept boxes in descending score (the reference's return value),
 *   n_keep [1] int32 OUT (device), workspace >= n * ceil(n/64) uint64.  n <= 4096.                       */
int sgc_aligned_nms3d(const float *boxes, const int64_t *order, const int64_t *labels, float iou_thr,
                      int64_t *keep, int32_t *n_keep, uint64_t *workspace, int n, sgc_stream_t stream);

/* Rotated BEV NMS of every class in one call == the loop of mmdet3d `box3d_multiclass_nms`
 * (packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py:52-68) over `nms_bev` (:231-268), i.e.
 * mmcv.ops.nms_rotated on BEV rectangles (mmcv-full 1.5.3, pip dependency docs/install.md:6, not vendored:
 * mmcv/ops/csrc/common/box_iou_rotated_utils.hpp + cuda/nms_rotated_cuda.cuh, T = float); called by
 * SunRgbdImVoxelHeadV2._nms, mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py:565-584 (ARKit configs):
 *   boxes  [K,5] fp32 (x1, y1, x2, y2, ry) = `mlvl_bboxes_for_nms`; converted to (xc, yc, w, h, ry) as :256-262;
 *   order  [C,K] int64: row c = box indices by DESCENDING score of class c (`_scores.sort(0, descending=True)`);
 *   counts [C] int32 (device): only the first counts[c] entries of row c are candidates (score > score_thr);
 *   a candidate is dropped when a kept, better-scored one of the same class has IoU > iou_thr (exact
 *   intersection polygon of the two rotated rectangles, fp32, the reference kernel's operation order);
 *   keep [C,K] int64 OUT: row c = kept box indices in descending score (= `_bboxes[selected]` order),
 *   n_keep [C] int32 OUT (device), workspace >= C * K * ceil(K/64) uint64.  K <= 4096.                     */
int sgc_nms_rotated_bev(const float *boxes, const int64_t *order, const int32_t *counts, float iou_thr,
                        int64_t *keep, int32_t *n_keep, uint64_t *workspace, int K, int C, sgc_stream_t stream);

/* mmcv.ops.box_iou_rotated(a, b, mode='iou', aligned=False) -- the IoU the NMS above thresholds:
 * a [n,5], b [m,5] fp32 (xc, yc, w, h, angle in radians) -> iou [n,m] fp32.                                */
int sgc_box_iou_rotated(const float *a, const float *b, float *iou, int n, int m, sgc_stream_t stream);

/* Target assignment of the FCOS3D-style head == `ImVoxelHeadV2.get_targets`
 * (mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py:361-435 for ScanNetImVoxelHeadV2, :485-561 for
 * SunRgbdImVoxelHeadV2) without its dense [n_points, n_boxes(, 6)] intermediates:
 *   points [n_points,3] fp32 (all scales concatenated, finest first), scales [n_points] int32 (scale id of a point),
 *   boxes [n_boxes,7] fp32 (gravity centre x,y,z, dx,dy,dz, yaw; yaw ignored unless `rotated`), gt_labels [n_boxes] int64.
 *   A point is assigned to the box of minimal volume among those it lies strictly inside of, whose best scale
 *   (smallest scale with >= `limit` inside points, :390-407) is the point's scale, and for which its centerness
 *   exceeds the box's (centerness_topk + 1)-th largest (:413-417).
 *   centerness_t [n_points] fp32, bbox_t [n_points,6] (x0,y0,z0,x1,y1,z1; rotated: [n_points,7] the assigned gt row),
 *   labels [n_points] int64 (-1 = background), geo_occ [n_points] uint8 (inside any box) OUT;
 *   workspace >= n_boxes * (n_scales + 2) int32.  n_boxes >= 1, centerness_topk + 1 <= n_points.             */
int sgc_assign_targets(const float *points, const int32_t *scales, const float *boxes, const int64_t *gt_labels,
                       int rotated, int n_scales, int limit, int centerness_topk, float *centerness_t, float *bbox_t,
                       int64_t *labels, uint8_t *geo_occ, int32_t *workspace, int n_points, int n_boxes,
                       sgc_stream_t stream);

/* ------------------------------------------------------------------------- *
 * 9. Upstream of the path: plane-sweep matching cost of DepthNet_Fusion
 * ------------------------------------------------------------------------- */

/* corr[n,d,y,x] = (1/K) sum_k ( sum_c warp_k(feat[nbr[n,k]])[c,d,y,x] * feat[n,(y,x),c] ) / sqrt(C)
 * == homo_warping + the cost-volume loop of DepthNet_Fusion.forward
 * (mmdet3d_plugin/models/im2voxel/depth_utils/depth_est_fusion.py:87-126, :233-240) without materialising the
 * warped features [N,C,D,H,W]:
 *   feat [N, H*W, C] channels-last matching features (f_mvs), C <= 256;  nbr [N,K] int32 neighbour view ids
 *   (get_closest_frame_ids, :53-64);  rt [N,K,12]: rows of (nei_proj @ inverse(ref_proj))[:3,:4] (:97-99);
 *   depth [D] plane depths (D <= 32);  corr [N,D,H,W] fully written.  Sampling = F.grid_sample(bilinear, zeros
 *   padding, align_corners=False) at the reference's (W-1)/2, (H-1)/2 normalised coordinates.              */
int sgc_plane_sweep_corr(const float *feat, const int32_t *nbr, const float *rt, const float *depth,
                         float *corr, int N, int K, int H, int W, int C, int D, sgc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SGCDET_AMD_H_ */
