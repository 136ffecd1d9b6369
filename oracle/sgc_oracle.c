/*
 * sgc_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the arithmetic of the reference's two CUDA kernel
 * families (the only native code on the SGCDet hot path) and of the torch glue
 * the MI355X library replaces with its own kernels.  It exports the symbols of
 * include/sgcdet_amd.h with host pointers and `stream` ignored, so that the same
 * ctypes binding drives the HIP library and this checker.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (sgcdet_amd/) never does.
 *
 * Pinning status: the reference ships NO test vectors for this path
 * (packages/3D-deformable-attention/unittest_DFA3D.py:71-72 only checks NaN and
 * needs CUDA) and its kernels cannot be built here (CUDA sources, no nvcc).
 * "parity unpinned" by the reference's own tests; this file is pinned instead by
 * (1) the independent closed form  grid_sample(value (x) dist)  evaluated with
 * torch on the CPU (tests/test_oracle_identity.py) and (2) golden vectors made
 * by running the reference's Python glue in the build container with this
 * oracle injected as `dfa3D._ext` (tests/golden/make_golden.py).  The two widened
 * rows are pinned directly: sgc_aligned_nms3d against kept indices of the reference's
 * own aligned_3d_nms (tests/golden/make_golden_nms.py), sgc_plane_sweep_corr against the
 * correlation volume of the reference's own homo_warping loop
 * (tests/golden/make_golden_planesweep.py).  sgc_box_iou_rotated (the IoU inside
 * sgc_nms_rotated_bev) restates box_iou_rotated_utils.hpp, which the reference DOES carry:
 * the DFA3D package vendors mmcv's header (CS/common/box_iou_rotated_utils.hpp).  It is plain
 * C++, so oracle/Makefile's `_ref` target compiles it with g++ from where it lies
 * (oracle/_ref/libref_box_iou.so, its __CUDACC__ branch = what the GPU NMS of the reference
 * executes) and tests/golden/make_golden_iou.py turns it into box_iou_rotated.npz: PINNED to
 * 1e-6 (tests/test_oracle_golden.py).  The greedy suppression loop around it restates
 * mmcv-full 1.5.3's nms_rotated_cuda.cuh (mask = IoU > thr, then a sweep; that file is not in
 * the reference tree); the class-loop glue is pinned to the reference's own
 * box3d_multiclass_nms / nms_bev (tests/golden/make_golden_nms_rotated.py).
 *
 * Reference files restated (paths relative to /root/reference; CS = packages/
 * 3D-deformable-attention/DFA3D/dfa3D/ops/csrc):
 *   depth score fwd  CS/common/cuda/ms_depth_score_sample_cuda_kernel.cuh:24-148
 *   depth score bwd  CS/common/cuda/ms_depth_score_sample_cuda_kernel.cuh:150-327
 *   weighted attn fwd CS/common/cuda/wms_deform_attn_cuda_kernel.cuh:24-80,240-303
 *   weighted attn bwd CS/common/cuda/wms_deform_attn_cuda_kernel.cuh:82-159,305-419
 *   aligned 3D NMS   packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py:131-178
 *   target assignment mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py:334-343,361-435,485-561
 *                    (pinned: tests/golden/make_golden_targets.py runs the reference's own get_targets)
 *   rotated BEV NMS  packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py:52-68,231-268
 *                    + CS/common/box_iou_rotated_utils.hpp:55-341 (the IoU; built as oracle/_ref) and mmcv-full
 *                    1.5.3's nms_rotated_cuda.cuh (the suppression sweep; published algorithm, not in the tree)
 *   plane sweep      mmdet3d_plugin/models/im2voxel/depth_utils/depth_est_fusion.py:87-126,233-240
 *
 * All arithmetic is fp32 in the reference's operation order; compile with
 * -ffp-contract=off so gcc does not fuse multiply-adds the reference's nvcc
 * build may or may not have fused (differences are <= 1 ulp per term either way).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sgcdet_amd.h"

static __thread char g_err[256] = "";

static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

int sgc_abi_version(void) { return SGC_ABI_VERSION; }

/* arithmetic mode of the MFMA entry points (include/sgcdet_amd.h): 3 = fp32 truth of the 3-way split, 1 = both operands
 * rounded to bfloat16 (RNE), 2 = both operands rounded to IEEE half (RNE, saturated at +-65504); products and sums in fp32 */
static int g_conv_products = 3;
int sgc_set_conv_products(int products) {
  if (products != 1 && products != 2 && products != 3) return fail(SGC_EINVAL, "3 (bf16x3), 1 (bf16) or 2 (fp16)");
  g_conv_products = products;
  return SGC_OK;
}
int sgc_get_conv_products(void) { return g_conv_products; }
int sgc_set_tuning(const char *key, int value) { (void)key; (void)value; return SGC_OK; }
const char *sgc_last_error(void) { return g_err; }
const char *sgc_backend(void) { return "cpu-oracle"; }

/* ---- shared helpers ------------------------------------------------------ */

/* depth scores of one sample: ms_depth_score_sample_cuda_kernel.cuh:129-141 (gate)
 * and :24-93 (im2col_trilinear).  dist_px(h,w) -> pointer to the D-vector of the
 * pixel for this (batch, head).  s[] must be zero on entry (output is pre-zeroed
 * in the reference, cuda.cu:83-85, and the helper does `+=`).                     */
static void depth_score_sample(const float *dist_b, int64_t pix_stride, int head_off,
                               int H, int W, int D, float x, float y, float z, float s[4]) {
  const float h_im = y * H - 0.5f;
  const float w_im = x * W - 0.5f;
  const float d_im = z * D - 0.5f;
  s[0] = s[1] = s[2] = s[3] = 0.f;
  if (!(h_im > -1 && w_im > -1 && d_im > -1 && h_im < H && w_im < W && d_im < D)) return;
  const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), d0 = (int)floorf(d_im);
  const int h1 = h0 + 1, w1 = w0 + 1, d1 = d0 + 1;
  const float ld = d_im - d0, hd = 1 - ld;
  const int hs[4] = {h0, h0, h1, h1}; /* reference corner order: (h0,w0) (h0,w1) (h1,w1) (h1,w0) */
  const int ws[4] = {w0, w1, w1, w0};
  for (int k = 0; k < 4; ++k) {
    const int h = hs[k], w = ws[k];
    if (h < 0 || h > H - 1 || w < 0 || w > W - 1) continue;
    const float *p = dist_b + ((int64_t)h * W + w) * pix_stride + head_off;
    const float va = (d0 >= 0) ? p[d0] : 0.f;
    const float vb = (d1 <= D - 1) ? p[d1] : 0.f;
    s[k] = va * hd + vb * ld;
  }
}

/* ---- 1a. depth score forward ------------------------------------------- */
int sgc_depth_score_forward(const float *dist, const int64_t *shapes3, const int64_t *lsi,
                            const float *loc3, float *score,
                            int B, int S, int M, int D, int L, int Q, int P, sgc_stream_t stream) {
  (void)stream;
  if (!dist || !shapes3 || !lsi || !loc3 || !score) return fail(SGC_EINVAL, "null pointer");
#pragma omp parallel for collapse(2) schedule(dynamic, 32)
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q)
      for (int m = 0; m < M; ++m) {
        const int64_t sidx = ((int64_t)b * Q + q) * M + m; /* "sampling_index" :104 */
        for (int l = 0; l < L; ++l) {
          const int H = (int)shapes3[l * 3], W = (int)shapes3[l * 3 + 1], Dl = (int)shapes3[l * 3 + 2];
          const float *dist_b = dist + ((int64_t)b * S + lsi[l]) * M * D;
          for (int p = 0; p < P; ++p) {
            const int64_t pt = (sidx * L + l) * P + p;
            depth_score_sample(dist_b, (int64_t)M * D, m * D, H, W, Dl, loc3[pt * 3], loc3[pt * 3 + 1],
                               loc3[pt * 3 + 2], score + pt * 4);
          }
        }
      }
  return SGC_OK;
}

/* one weighted bilinear term: wms_deform_attn_cuda_kernel.cuh:24-80.
 * vb = value + offset of (batch, level), stride between pixels = M*Cm.             */
static float wms_bilinear(const float *vb, int H, int W, int MC, int mc_off, float h, float w,
                          const float s[4]) {
  const int h0 = (int)floorf(h), w0 = (int)floorf(w);
  const int h1 = h0 + 1, w1 = w0 + 1;
  const float lh = h - h0, lw = w - w0, hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0;
  if (h0 >= 0 && w0 >= 0) { d1 = s[0]; v1 = vb[((int64_t)h0 * W + w0) * MC + mc_off]; }
  if (h0 >= 0 && w1 <= W - 1) { d2 = s[1]; v2 = vb[((int64_t)h0 * W + w1) * MC + mc_off]; }
  if (h1 <= H - 1 && w0 >= 0) { d3 = s[3]; v3 = vb[((int64_t)h1 * W + w0) * MC + mc_off]; }
  if (h1 <= H - 1 && w1 <= W - 1) { d4 = s[2]; v4 = vb[((int64_t)h1 * W + w1) * MC + mc_off]; }
  const float a1 = hh * hw * d1, a2 = hh * lw * d2, a3 = lh * hw * d3, a4 = lh * lw * d4;
  return a1 * v1 + a2 * v2 + a3 * v3 + a4 * v4;
}

/* ---- 1b. weighted deformable attention forward ---------------------------- */
int sgc_wms_forward(const float *value, const int64_t *shapes2, const int64_t *lsi,
                    const float *loc2, const float *attn, const float *score, float *out,
                    int B, int S, int M, int Cm, int L, int Q, int P, sgc_stream_t stream) {
  (void)stream;
  if (!value || !shapes2 || !lsi || !loc2 || !attn || !score || !out)
    return fail(SGC_EINVAL, "null pointer");
  const int MC = M * Cm;
#pragma omp parallel for collapse(2) schedule(dynamic, 32)
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q)
      for (int m = 0; m < M; ++m) {
        const int64_t sidx = ((int64_t)b * Q + q) * M + m;
        for (int c = 0; c < Cm; ++c) {
          float col = 0.f; /* :270 */
          for (int l = 0; l < L; ++l) {
            const int H = (int)shapes2[l * 2], W = (int)shapes2[l * 2 + 1];
            const float *vb = value + ((int64_t)b * S + lsi[l]) * MC;
            for (int p = 0; p < P; ++p) {
              const int64_t pt = (sidx * L + l) * P + p;
              const float h_im = loc2[pt * 2 + 1] * H - 0.5f; /* :286-287 */
              const float w_im = loc2[pt * 2] * W - 0.5f;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                col += wms_bilinear(vb, H, W, MC, m * Cm + c, h_im, w_im, score + pt * 4) * attn[pt];
            }
          }
          out[sidx * Cm + c] = col;
        }
      }
  return SGC_OK;
}

/* ---- 1c. weighted deformable attention backward --------------------------- */
int sgc_wms_backward(const float *value, const int64_t *shapes2, const int64_t *lsi,
                     const float *loc2, const float *attn, const float *score,
                     const float *grad_out, float *grad_value, float *grad_loc2,
                     float *grad_attn, float *grad_score,
                     int B, int S, int M, int Cm, int L, int Q, int P, sgc_stream_t stream) {
  (void)stream;
  if (!value || !shapes2 || !lsi || !loc2 || !attn || !score || !grad_out || !grad_value ||
      !grad_loc2 || !grad_attn || !grad_score)
    return fail(SGC_EINVAL, "null pointer");
  const int MC = M * Cm;
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q)
      for (int m = 0; m < M; ++m) {
        const int64_t sidx = ((int64_t)b * Q + q) * M + m;
        for (int l = 0; l < L; ++l) {
          const int H = (int)shapes2[l * 2], W = (int)shapes2[l * 2 + 1];
          const float *vb = value + ((int64_t)b * S + lsi[l]) * MC;
          float *gvb = grad_value + ((int64_t)b * S + lsi[l]) * MC;
          for (int p = 0; p < P; ++p) {
            const int64_t pt = (sidx * L + l) * P + p;
            const float *s = score + pt * 4;
            const float aw = attn[pt];
            const float h_im = loc2[pt * 2 + 1] * H - 0.5f;
            const float w_im = loc2[pt * 2] * W - 0.5f;
            /* per-(b,q,m,l,p) sums over the Cm channel threads, taken sequentially in
             * c like the tid==0 loop of reduce_v1 (:378-398) */
            float gw = 0, gh = 0, ga = 0, gs[4] = {0, 0, 0, 0};
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
              const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
              const float lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
              const float a1 = hh * hw * s[0], a2 = hh * lw * s[1], a3 = lh * hw * s[3],
                          a4 = lh * lw * s[2]; /* :106 */
              for (int c = 0; c < Cm; ++c) {
                const int off = m * Cm + c;
                const float top = grad_out[sidx * Cm + c];
                const float tgv = top * aw; /* :107 */
                float ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (h0 >= 0 && w0 >= 0) { /* :112-120 */
                  const int64_t o = ((int64_t)h0 * W + w0) * MC + off;
                  v1 = vb[o];
                  ghw -= s[0] * hw * v1; gww -= s[0] * hh * v1;
                  gvb[o] += a1 * tgv;
                  gs[0] += v1 * hh * hw * tgv;
                }
                if (h0 >= 0 && w1 <= W - 1) { /* :123-131 */
                  const int64_t o = ((int64_t)h0 * W + w1) * MC + off;
                  v2 = vb[o];
                  ghw -= s[1] * lw * v2; gww += s[1] * hh * v2;
                  gvb[o] += a2 * tgv;
                  gs[1] += v2 * hh * lw * tgv;
                }
                if (h1 <= H - 1 && w0 >= 0) { /* :134-142 */
                  const int64_t o = ((int64_t)h1 * W + w0) * MC + off;
                  v3 = vb[o];
                  ghw += s[3] * hw * v3; gww -= s[3] * lh * v3;
                  gvb[o] += a3 * tgv;
                  gs[3] += v3 * lh * hw * tgv;
                }
                if (h1 <= H - 1 && w1 <= W - 1) { /* :145-153 */
                  const int64_t o = ((int64_t)h1 * W + w1) * MC + off;
                  v4 = vb[o];
                  ghw += s[2] * lw * v4; gww += s[2] * lh * v4;
                  gvb[o] += a4 * tgv;
                  gs[2] += v4 * lh * lw * tgv;
                }
                const float val = a1 * v1 + a2 * v2 + a3 * v3 + a4 * v4;
                ga += top * val;      /* :156 */
                gw += W * gww * tgv;  /* :157 */
                gh += H * ghw * tgv;  /* :158 */
              }
            }
            grad_loc2[pt * 2] = gw;
            grad_loc2[pt * 2 + 1] = gh;
            grad_attn[pt] = ga;
            memcpy(grad_score + pt * 4, gs, sizeof gs);
          }
        }
      }
  return SGC_OK;
}

/* gradient of one sample's 4 depth scores w.r.t. dist and z:
 * ms_depth_score_sample_cuda_kernel.cuh:150-241 (one case per corner).            */
static float depth_score_sample_bwd(const float *dist_b, float *gdist_b, int64_t pix_stride,
                                    int head_off, int H, int W, int D, float x, float y, float z,
                                    const float g[4]) {
  const float h_im = y * H - 0.5f, w_im = x * W - 0.5f, d_im = z * D - 0.5f;
  if (!(h_im > -1 && w_im > -1 && d_im > -1 && h_im < H && w_im < W && d_im < D)) return 0.f;
  const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), d0 = (int)floorf(d_im);
  const int h1 = h0 + 1, w1 = w0 + 1, d1 = d0 + 1;
  const float ld = d_im - d0, hd = 1 - ld;
  const int hs[4] = {h0, h0, h1, h1};
  const int ws[4] = {w0, w1, w1, w0};
  float gz = 0.f; /* reduced over the 4 corner threads, :304-314 */
  for (int k = 0; k < 4; ++k) {
    const int h = hs[k], w = ws[k];
    float va = 0, vb = 0;
    if (h >= 0 && h <= H - 1 && w >= 0 && w <= W - 1) {
      const int64_t o = ((int64_t)h * W + w) * pix_stride + head_off;
      if (d0 >= 0) { va = dist_b[o + d0]; gdist_b[o + d0] += hd * g[k]; }
      if (d1 <= D - 1) { vb = dist_b[o + d1]; gdist_b[o + d1] += ld * g[k]; }
    }
    gz += D * (g[k] * (vb - va)); /* :193,240 */
  }
  return gz;
}

/* ---- 1d. depth score backward -------------------------------------------- */
int sgc_depth_score_backward(const float *dist, const int64_t *shapes3, const int64_t *lsi,
                             const float *loc3, const float *grad_score,
                             float *grad_dist, float *grad_loc3,
                             int B, int S, int M, int D, int L, int Q, int P, sgc_stream_t stream) {
  (void)stream;
  if (!dist || !shapes3 || !lsi || !loc3 || !grad_score || !grad_dist || !grad_loc3)
    return fail(SGC_EINVAL, "null pointer");
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q)
      for (int m = 0; m < M; ++m) {
        const int64_t sidx = ((int64_t)b * Q + q) * M + m;
        for (int l = 0; l < L; ++l) {
          const int H = (int)shapes3[l * 3], W = (int)shapes3[l * 3 + 1], Dl = (int)shapes3[l * 3 + 2];
          const int64_t boff = ((int64_t)b * S + lsi[l]) * M * D;
          for (int p = 0; p < P; ++p) {
            const int64_t pt = (sidx * L + l) * P + p;
            const float gz = depth_score_sample_bwd(dist + boff, grad_dist + boff, (int64_t)M * D, m * D,
                                                    H, W, Dl, loc3[pt * 3], loc3[pt * 3 + 1],
                                                    loc3[pt * 3 + 2], grad_score + pt * 4);
            grad_loc3[pt * 3] = 0.f; /* :238-239: uv gradient through the score is dropped */
            grad_loc3[pt * 3 + 1] = 0.f;
            grad_loc3[pt * 3 + 2] = gz;
          }
        }
      }
  return SGC_OK;
}

/* ---- 2. fused forms -------------------------------------------------------- */
int sgc_dfa3d_forward(const float *value, const float *dist, const int64_t *shapes3,
                      const int64_t *lsi, const float *loc3, const float *attn,
                      float *out, float *score_or_null,
                      int B, int S, int M, int Cm, int D, int dist_heads,
                      int L, int Q, int P, sgc_stream_t stream) {
  (void)stream;
  if (!value || !dist || !shapes3 || !lsi || !loc3 || !out) return fail(SGC_EINVAL, "null pointer");
  if (dist_heads != 1 && dist_heads != M) return fail(SGC_EINVAL, "dist_heads must be 1 or M");
  const int MC = M * Cm;
#pragma omp parallel for collapse(2) schedule(dynamic, 32)
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q)
      for (int m = 0; m < M; ++m) {
        const int64_t sidx = ((int64_t)b * Q + q) * M + m;
        float *o = out + sidx * Cm;
        for (int c = 0; c < Cm; ++c) o[c] = 0.f;
        for (int l = 0; l < L; ++l) {
          const int H = (int)shapes3[l * 3], W = (int)shapes3[l * 3 + 1], Dl = (int)shapes3[l * 3 + 2];
          const float *vb = value + ((int64_t)b * S + lsi[l]) * MC;
          const float *db = dist + ((int64_t)b * S + lsi[l]) * dist_heads * D;
          for (int p = 0; p < P; ++p) {
            const int64_t pt = (sidx * L + l) * P + p;
            float s[4];
            const float x = loc3[pt * 3], y = loc3[pt * 3 + 1], z = loc3[pt * 3 + 2];
            depth_score_sample(db, (int64_t)dist_heads * D, (dist_heads == 1 ? 0 : m) * D, H, W, Dl, x, y, z, s);
            if (score_or_null) memcpy(score_or_null + pt * 4, s, sizeof s);
            const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
            const float aw = attn ? attn[pt] : 1.0f;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
              for (int c = 0; c < Cm; ++c)
                o[c] += wms_bilinear(vb, H, W, MC, m * Cm + c, h_im, w_im, s) * aw;
          }
        }
      }
  return SGC_OK;
}

int sgc_dfa3d_backward(const float *value, const float *dist, const int64_t *shapes3,
                       const int64_t *lsi, const float *loc3, const float *attn,
                       const float *grad_out, float *grad_value, float *grad_dist,
                       float *grad_loc3, float *grad_attn_or_null,
                       int B, int S, int M, int Cm, int D, int dist_heads,
                       int L, int Q, int P, sgc_stream_t stream) {
  (void)stream;
  if (!value || !dist || !shapes3 || !lsi || !loc3 || !grad_out || !grad_value || !grad_dist || !grad_loc3)
    return fail(SGC_EINVAL, "null pointer");
  if (dist_heads != 1 && dist_heads != M) return fail(SGC_EINVAL, "dist_heads must be 1 or M");
  const int MC = M * Cm;
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q)
      for (int m = 0; m < M; ++m) {
        const int64_t sidx = ((int64_t)b * Q + q) * M + m;
        for (int l = 0; l < L; ++l) {
          const int H = (int)shapes3[l * 3], W = (int)shapes3[l * 3 + 1], Dl = (int)shapes3[l * 3 + 2];
          const int64_t voff = ((int64_t)b * S + lsi[l]) * MC;
          const int64_t doff = ((int64_t)b * S + lsi[l]) * dist_heads * D;
          const int hoff = (dist_heads == 1 ? 0 : m) * D;
          for (int p = 0; p < P; ++p) {
            const int64_t pt = (sidx * L + l) * P + p;
            const float x = loc3[pt * 3], y = loc3[pt * 3 + 1], z = loc3[pt * 3 + 2];
            float s[4];
            depth_score_sample(dist + doff, (int64_t)dist_heads * D, hoff, H, W, Dl, x, y, z, s);
            const float aw = attn ? attn[pt] : 1.0f;
            const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
            float gw = 0, gh = 0, ga = 0, gs[4] = {0, 0, 0, 0};
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
              const float *vb = value + voff;
              float *gvb = grad_value + voff;
              const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
              const float lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
              const float a1 = hh * hw * s[0], a2 = hh * lw * s[1], a3 = lh * hw * s[3], a4 = lh * lw * s[2];
              for (int c = 0; c < Cm; ++c) {
                const int off = m * Cm + c;
                const float top = grad_out[sidx * Cm + c], tgv = top * aw;
                float ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (h0 >= 0 && w0 >= 0) {
                  const int64_t o = ((int64_t)h0 * W + w0) * MC + off; v1 = vb[o];
                  ghw -= s[0] * hw * v1; gww -= s[0] * hh * v1; gvb[o] += a1 * tgv; gs[0] += v1 * hh * hw * tgv;
                }
                if (h0 >= 0 && w1 <= W - 1) {
                  const int64_t o = ((int64_t)h0 * W + w1) * MC + off; v2 = vb[o];
                  ghw -= s[1] * lw * v2; gww += s[1] * hh * v2; gvb[o] += a2 * tgv; gs[1] += v2 * hh * lw * tgv;
                }
                if (h1 <= H - 1 && w0 >= 0) {
                  const int64_t o = ((int64_t)h1 * W + w0) * MC + off; v3 = vb[o];
                  ghw += s[3] * hw * v3; gww -= s[3] * lh * v3; gvb[o] += a3 * tgv; gs[3] += v3 * lh * hw * tgv;
                }
                if (h1 <= H - 1 && w1 <= W - 1) {
                  const int64_t o = ((int64_t)h1 * W + w1) * MC + off; v4 = vb[o];
                  ghw += s[2] * lw * v4; gww += s[2] * lh * v4; gvb[o] += a4 * tgv; gs[2] += v4 * lh * lw * tgv;
                }
                ga += top * (a1 * v1 + a2 * v2 + a3 * v3 + a4 * v4);
                gw += W * gww * tgv;
                gh += H * ghw * tgv;
              }
            }
            const float gz = depth_score_sample_bwd(dist + doff, grad_dist + doff, (int64_t)dist_heads * D,
                                                    hoff, H, W, Dl, x, y, z, gs);
            grad_loc3[pt * 3] = gw;      /* python merge, multi_scale_3ddeformable_attn_function.py:349 */
            grad_loc3[pt * 3 + 1] = gh;
            grad_loc3[pt * 3 + 2] = gz;
            if (grad_attn_or_null) grad_attn_or_null[pt] = ga;
          }
        }
      }
  return SGC_OK;
}

/* item-list forms: item i samples map item_batch[i]; one (B = 1, Q = 1) call of the batch form per item */
int sgc_dfa3d_forward_items(const float *value, const float *dist, const int64_t *shapes3, const int64_t *lsi,
                            const float *loc3, const float *attn_or_null, const int32_t *item_batch, float *out,
                            float *score_or_null, int B, int S, int M, int Cm, int D, int dist_heads, int L,
                            int n_items, int P, sgc_stream_t stream) {
  if (!value || !dist || !shapes3 || !lsi || !loc3 || !item_batch || !out) return fail(SGC_EINVAL, "null pointer");
  const int64_t MC = (int64_t)M * Cm, SPI = (int64_t)M * L * P;
  for (int i = 0; i < n_items; ++i) {
    const int b = item_batch[i];
    if (b < 0 || b >= B) return fail(SGC_EINVAL, "item_batch out of range");
    const int rc = sgc_dfa3d_forward(value + (int64_t)b * S * MC, dist + (int64_t)b * S * dist_heads * D, shapes3, lsi,
                                     loc3 + i * SPI * 3, attn_or_null ? attn_or_null + i * SPI : NULL, out + i * MC,
                                     score_or_null ? score_or_null + i * SPI * 4 : NULL, 1, S, M, Cm, D, dist_heads, L, 1, P, stream);
    if (rc) return rc;
  }
  return SGC_OK;
}

int sgc_dfa3d_backward_items(const float *value, const float *dist, const int64_t *shapes3,
                             const int64_t *lsi, const float *loc3, const float *attn_or_null,
                             const int32_t *item_batch, const float *grad_out, float *grad_value, float *grad_dist,
                             float *grad_loc3, float *grad_attn_or_null,
                             int B, int S, int M, int Cm, int D, int dist_heads,
                             int L, int n_items, int P, sgc_stream_t stream) {
  if (!value || !dist || !shapes3 || !lsi || !loc3 || !item_batch || !grad_out || !grad_value || !grad_dist || !grad_loc3)
    return fail(SGC_EINVAL, "null pointer");
  const int64_t MC = (int64_t)M * Cm, SPI = (int64_t)M * L * P;
  for (int i = 0; i < n_items; ++i) {
    const int b = item_batch[i];
    if (b < 0 || b >= B) return fail(SGC_EINVAL, "item_batch out of range");
    const int rc = sgc_dfa3d_backward(value + (int64_t)b * S * MC, dist + (int64_t)b * S * dist_heads * D, shapes3, lsi,
                                      loc3 + i * SPI * 3, attn_or_null ? attn_or_null + i * SPI : NULL, grad_out + i * MC,
                                      grad_value + (int64_t)b * S * MC, grad_dist + (int64_t)b * S * dist_heads * D,
                                      grad_loc3 + i * SPI * 3, grad_attn_or_null ? grad_attn_or_null + i * SPI : NULL,
                                      1, S, M, Cm, D, dist_heads, L, 1, P, stream);
    if (rc) return rc;
  }
  return SGC_OK;
}

/* Binned form of the item backward (product: csrc/dfa3d_bwd_tile.hip).  The bins only say which workgroup adds what: the
 * function of the inputs is sgc_dfa3d_backward on every item with its camera's maps; a sample set shared by the M channel groups
 * (loc_heads == 1 < M) is the one-head operator over C = M * Cm channels. */
int64_t sgc_dfa3d_backward_binned_lds_bytes(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y) {
  if (H <= 0 || W <= 0 || Cm <= 0 || D <= 0 || bin_w <= 0 || bin_h <= 0 || halo_x < 0 || halo_y < 0) return 0;
  const int64_t tw = bin_w + 2 * halo_x < W ? bin_w + 2 * halo_x : W, th = bin_h + 2 * halo_y < H ? bin_h + 2 * halo_y : H;
  return tw * th * (Cm + 1 + D) * 4;
}
int sgc_dfa3d_backward_binned(const float *value, const float *dist, const float *loc3, const float *attn_or_null,
                              const int32_t *bin_offset, const int32_t *head_shift_or_null, const float *grad_out, float *grad_value, float *grad_dist,
                              float *grad_loc3_or_null, float *grad_attn_or_null, int N, int S, int H, int W, int M, int Cm,
                              int D, int loc_heads, int P, int bin_w, int bin_h, int halo_x, int halo_y, sgc_stream_t stream) {
  (void)halo_x; (void)halo_y; (void)head_shift_or_null;
  if (!value || !dist || !loc3 || !bin_offset || !grad_out || !grad_value || !grad_dist) return fail(SGC_EINVAL, "null pointer");
  if (loc_heads != 1 && loc_heads != M) return fail(SGC_EINVAL, "loc_heads must be 1 or M");
  const int nb = ((W + bin_w - 1) / bin_w) * ((H + bin_h - 1) / bin_h);
  const int64_t MC = (int64_t)M * Cm;
  const int Mo = loc_heads == 1 ? 1 : M, Cmo = loc_heads == 1 ? M * Cm : Cm;      /* the operator's own head split */
  const int64_t SPI = (int64_t)Mo * P;
  const int64_t shapes3[3] = {H, W, D}, lsi[1] = {0};
  float *gl_tmp = (float *)malloc(sizeof(float) * (size_t)SPI * 3), *ga_tmp = (float *)malloc(sizeof(float) * (size_t)SPI);
  int rc = SGC_OK;
  for (int t = 0; t < N * nb && !rc; ++t) {
    const int b = t / nb;
    for (int i = bin_offset[t]; i < bin_offset[t + 1] && !rc; ++i) {
      rc = sgc_dfa3d_backward(value + (int64_t)b * S * MC, dist + (int64_t)b * S * D, shapes3, lsi, loc3 + i * SPI * 3,
                              attn_or_null ? attn_or_null + i * SPI : NULL, grad_out + i * MC, grad_value + (int64_t)b * S * MC,
                              grad_dist + (int64_t)b * S * D, grad_loc3_or_null ? grad_loc3_or_null + i * SPI * 3 : gl_tmp,
                              grad_attn_or_null ? grad_attn_or_null + i * SPI : ga_tmp, 1, S, Mo, Cmo, D, 1, 1, 1, P, stream);
    }
  }
  free(gl_tmp); free(ga_tmp);
  return rc;
}

/* ---- 3. projection + compaction ------------------------------------------ */
/* VoxFormerEncoder_DFA3D.point_sampling, TU/encoder.py:179-223.  Fixed order:
 *   p = ref + origin;  cam_r = ((P_r0*x + P_r1*y) + P_r2*z) + P_r3  (no FMA);
 *   den = max(cam_z, eps);  u = (cam_x/den) * (1/img_w);  v = (cam_y/den) * (1/img_h);
 *   zn = (cam_z - d_near) * (1/(d_far - d_near))
 * (torch's GPU `tensor / python_scalar` multiplies by the fp32 reciprocal).         */
int sgc_project_points(const float *ref3d, const int64_t *sel_or_null, const float *origin, const float *proj,
                       float *ref_cam, uint8_t *mask,
                       int N, int Nq, float img_w, float img_h, float d_near, float d_far,
                       sgc_stream_t stream) {
  (void)stream;
  if (!ref3d || !origin || !proj || !ref_cam || !mask) return fail(SGC_EINVAL, "null pointer");
  const float eps = 1e-5f;
  const float rw = 1.0f / img_w, rh = 1.0f / img_h, rd = 1.0f / (d_far - d_near);
  const float hi = 1.0f - eps;
  for (int n = 0; n < N; ++n) {
    const float *Pm = proj + (int64_t)n * 12;
    for (int q = 0; q < Nq; ++q) {
      const int64_t r = sel_or_null ? sel_or_null[q] : q;
      const float x = ref3d[r * 3] + origin[0], y = ref3d[r * 3 + 1] + origin[1], z = ref3d[r * 3 + 2] + origin[2];
      float cam[3];
      for (int r = 0; r < 3; ++r) cam[r] = ((Pm[r * 4] * x + Pm[r * 4 + 1] * y) + Pm[r * 4 + 2] * z) + Pm[r * 4 + 3];
      const float den = fmaxf(cam[2], eps);
      const float u = (cam[0] / den) * rw, v = (cam[1] / den) * rh;
      const float zn = (cam[2] - d_near) * rd;
      float *o = ref_cam + ((int64_t)n * Nq + q) * 3;
      o[0] = u; o[1] = v; o[2] = zn;
      /* `points_d` aliases reference_points_cam[..., 2:3], overwritten with zn before the test (TU/encoder.py:203-213) */
      mask[(int64_t)n * Nq + q] = (uint8_t)(zn > eps && u > eps && u < hi && v > eps && v < hi);
    }
  }
  return SGC_OK;
}

int sgc_compact_pairs(const uint8_t *mask, int N, int Nq,
                      int32_t *cam_count, int32_t *cam_offset,
                      int32_t *pair_cam, int32_t *pair_q, int32_t *slot,
                      int32_t *vox_count, int32_t *valid_index, int32_t *totals,
                      int32_t *workspace, sgc_stream_t stream) {
  (void)stream;
  if (!mask || !cam_count || !cam_offset || !pair_cam || !pair_q || !slot || !vox_count || !valid_index || !totals)
    return fail(SGC_EINVAL, "null pointer");
  int32_t np = 0, max_len = 0;
  for (int q = 0; q < Nq; ++q) vox_count[q] = 0;
  for (int n = 0; n < N; ++n) { /* indexes[i] = nonzero(mask_i), TU/deformable_cross_attention.py:759-762 */
    cam_offset[n] = np;
    int32_t c = 0;
    for (int q = 0; q < Nq; ++q) {
      if (mask[(int64_t)n * Nq + q]) {
        pair_cam[np] = n; pair_q[np] = q; slot[(int64_t)n * Nq + q] = np;
        ++np; ++c; ++vox_count[q];
      } else {
        slot[(int64_t)n * Nq + q] = -1;
      }
    }
    cam_count[n] = c;
    if (c > max_len) max_len = c;
  }
  cam_offset[N] = np;
  int32_t nv = 0; /* valid_index = count.nonzero(), :822 */
  for (int q = 0; q < Nq; ++q) {
    if (workspace) workspace[q] = vox_count[q] > 0 ? nv : -1;      /* row_of: inverse of valid_index */
    if (vox_count[q] > 0) valid_index[nv++] = q;
  }
  totals[0] = np; totals[1] = nv; totals[2] = max_len; totals[3] = 0;
  return SGC_OK;
}

/* ---- 4. pair-list gathers ---------------------------------------------------- */
int sgc_pairs_geometry_sample(const float *feat, const float *dist, const float *ref_cam,
                              const int32_t *pair_cam, const int32_t *pair_q,
                              const int32_t *totals, float *out,
                              int N, int Nq, int H, int W, int C, int D, int cam_stride_or_0,
                              int n_pairs_or_neg, int cap, sgc_stream_t stream) {
  (void)stream; (void)N;
  if (!feat || !dist || !ref_cam || !pair_cam || !pair_q || !out) return fail(SGC_EINVAL, "null pointer");
  int np = n_pairs_or_neg >= 0 ? n_pairs_or_neg : (totals ? totals[0] : -1);
  if (np < 0 || np > cap) return fail(SGC_EINVAL, "n_pairs out of range");
  if (cam_stride_or_0 > 0 && cam_stride_or_0 < H * W) return fail(SGC_EINVAL, "cam_stride < H*W");
  const int64_t S = cam_stride_or_0 > 0 ? cam_stride_or_0 : (int64_t)H * W;   /* pixels between cameras */
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < np; ++i) {
    const int n = pair_cam[i], q = pair_q[i];
    const float *r = ref_cam + ((int64_t)n * Nq + q) * 3;
    float s[4];
    depth_score_sample(dist + n * S * D, D, 0, H, W, D, r[0], r[1], r[2], s);
    const float h_im = r[1] * H - 0.5f, w_im = r[0] * W - 0.5f;
    float *o = out + (int64_t)i * C;
    const int in = (h_im > -1 && w_im > -1 && h_im < H && w_im < W);
    for (int c = 0; c < C; ++c) {
      float col = 0.f;
      if (in) col += wms_bilinear(feat + n * S * C, H, W, C, c, h_im, w_im, s) * 1.0f;
      o[c] = col;
    }
  }
  return SGC_OK;
}

int sgc_depth_pairs(const float *dist, float *dp, int N, int H, int W, int D, int cam_stride_or_0,
                    sgc_stream_t stream) {
  (void)stream;
  if (!dist || !dp) return fail(SGC_EINVAL, "null pointer");
  if (cam_stride_or_0 > 0 && cam_stride_or_0 < H * W) return fail(SGC_EINVAL, "cam_stride < H*W");
  const int64_t S = cam_stride_or_0 > 0 ? cam_stride_or_0 : (int64_t)H * W;
  for (int64_t nh = 0; nh < (int64_t)N * H; ++nh) {
    const float *row = dist + ((nh / H) * S + (nh % H) * W) * D;
    for (int wq = 0; wq <= W; ++wq)
      for (int d = 0; d < D; ++d) {
        float *o = dp + ((nh * (W + 1) + wq) * D + d) * 2;
        o[0] = wq > 0 ? row[(wq - 1) * D + d] : 0.f;
        o[1] = wq < W ? row[wq * D + d] : 0.f;
      }
  }
  return SGC_OK;
}

int sgc_pairs_deform_gather(const float *value, const float *dist, const float *dist_pairs_or_null,
                            const float *ref_cam,
                            const float *raw, const int32_t *pair_cam, const int32_t *pair_q,
                            const int32_t *totals, float *out,
                            int N, int Nq, int H, int W, int M, int Cm, int D, int P, int cam_stride_or_0,
                            int value_has_zero_row, int n_pairs_or_neg, int cap, sgc_stream_t stream) {
  (void)stream; (void)N; (void)dist_pairs_or_null; (void)value_has_zero_row;   /* the oracle always samples the plain map */
  if (!value || !dist || !ref_cam || !raw || !pair_cam || !pair_q || !out) return fail(SGC_EINVAL, "null pointer");
  if (P > 64) return fail(SGC_EUNSUP, "P > 64");
  int np = n_pairs_or_neg >= 0 ? n_pairs_or_neg : (totals ? totals[0] : -1);
  if (np < 0 || np > cap) return fail(SGC_EINVAL, "n_pairs out of range");
  if (cam_stride_or_0 > 0 && cam_stride_or_0 < H * W) return fail(SGC_EINVAL, "cam_stride < H*W");
  const int64_t S = cam_stride_or_0 > 0 ? cam_stride_or_0 : (int64_t)H * W;   /* pixels between cameras */
  const int MC = M * Cm, MP = M * P;
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < np; ++i) {
    const int n = pair_cam[i], q = pair_q[i];
    const float *r = ref_cam + ((int64_t)n * Nq + q) * 3;
    const float *rw = raw + (int64_t)i * MP * 4;
    for (int m = 0; m < M; ++m) {
      /* softmax over the L*P logits of this head (TU/deformable_cross_attention.py:428-431) */
      const float *lg = rw + MP * 3 + m * P;
      float mx = lg[0];
      for (int p = 1; p < P; ++p) mx = fmaxf(mx, lg[p]);
      float e[64], sum = 0.f;
      for (int p = 0; p < P; ++p) { e[p] = expf(lg[p] - mx); sum += e[p]; }
      float *o = out + (int64_t)i * MC + m * Cm;
      for (int c = 0; c < Cm; ++c) o[c] = 0.f;
      for (int p = 0; p < P; ++p) {
        /* loc = ref + offset / (W, H, D), :445-455 */
        const float x = r[0] + rw[(m * P + p) * 2] / (float)W;
        const float y = r[1] + rw[(m * P + p) * 2 + 1] / (float)H;
        const float z = r[2] + rw[MP * 2 + m * P + p] / (float)D;
        const float aw = e[p] / sum;
        float s[4];
        depth_score_sample(dist + n * S * D, D, 0, H, W, D, x, y, z, s);
        const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
          for (int c = 0; c < Cm; ++c)
            o[c] += wms_bilinear(value + n * S * MC, H, W, MC, m * Cm + c, h_im, w_im, s) * aw;
      }
    }
  }
  return SGC_OK;
}

static float bf16_to_f32(uint16_t b) { union { uint32_t u; float f; } c; c.u = (uint32_t)b << 16; return c.f; }

/* ---- 4b. binned / head-major forms (include/sgcdet_amd.h: sgc_bin_pairs, sgc_pairs_deform_gather_tiled) ---- */
static int oracle_ref_bin(const float *rc, int H, int W, int bw, int bh, int nbx) {
  const float w_im = rc[0] * (float)W - 0.5f, h_im = rc[1] * (float)H - 0.5f;
  int px = (int)floorf(w_im), py = (int)floorf(h_im);
  px = px < 0 ? 0 : (px > W - 1 ? W - 1 : px);
  py = py < 0 ? 0 : (py > H - 1 ? H - 1 : py);
  return (py / bh) * nbx + px / bw;
}

/* stable counting sort of every camera's pairs by bin: ascending ORIGINAL pair index inside a bin */
int64_t sgc_bin_pairs_workspace_bytes(int N, int Nq, int cap, int H, int W, int bin_w, int bin_h) {
  (void)N; (void)Nq; (void)cap; (void)H; (void)W; (void)bin_w; (void)bin_h;
  return 16;
}

int sgc_bin_pairs(const float *ref_cam, const int32_t *pair_cam, const int32_t *pair_q, const int32_t *cam_offset,
                  int32_t *pair_q_out, int32_t *slot, float *pair_ref, int32_t *bin_offset, void *workspace,
                  int N, int Nq, int cap, int H, int W, int bin_w, int bin_h, sgc_stream_t stream) {
  (void)stream; (void)workspace; (void)pair_cam; (void)cap;
  if (!ref_cam || !pair_q || !cam_offset || !pair_q_out || !slot || !pair_ref || !bin_offset) return fail(SGC_EINVAL, "null pointer");
  if (N <= 0 || Nq <= 0 || H <= 0 || W <= 0 || bin_w <= 0 || bin_h <= 0) return fail(SGC_EINVAL, "bad size");
  if (pair_q_out == pair_q) return fail(SGC_EINVAL, "pair_q_out must not alias pair_q");
  const int nbx = (W + bin_w - 1) / bin_w, nby = (H + bin_h - 1) / bin_h, nb = nbx * nby;
  if (nb > 1024) return fail(SGC_EUNSUP, "more than 1024 bins per camera");
  int *next = (int *)malloc(sizeof(int) * (size_t)nb);
  if (!next) return fail(SGC_EINVAL, "out of memory");
  for (int n = 0; n < N; ++n) {
    const int p0 = cam_offset[n], p1 = cam_offset[n + 1];
    const float *rcam = ref_cam + (int64_t)n * Nq * 3;
    for (int b = 0; b < nb; ++b) next[b] = 0;
    for (int p = p0; p < p1; ++p) next[oracle_ref_bin(rcam + (int64_t)pair_q[p] * 3, H, W, bin_w, bin_h, nbx)]++;
    int run = p0;
    for (int b = 0; b < nb; ++b) { const int c = next[b]; bin_offset[(int64_t)n * nb + b] = run; next[b] = run; run += c; }
    for (int p = p0; p < p1; ++p) {
      const int32_t q = pair_q[p];
      const float *rc = rcam + (int64_t)q * 3;
      const int pos = next[oracle_ref_bin(rc, H, W, bin_w, bin_h, nbx)]++;
      float *o = pair_ref + (int64_t)pos * 4;
      o[0] = rc[0]; o[1] = rc[1]; o[2] = rc[2];
      memcpy(o + 3, &q, sizeof(int32_t));
      pair_q_out[pos] = q;
      slot[(int64_t)n * Nq + q] = pos;
    }
    if (n == N - 1) bin_offset[(int64_t)N * nb] = p1;
  }
  free(next);
  return SGC_OK;
}

int sgc_tile_window(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y, int max_shift_x,
                    int max_shift_y, int depth_in_lds, int value_bf16, int *tw_out, int *th_out, int *lds_bytes_out, int *nbuf_out,
                    int *depth_in_lds_out) {
  const int tw = bin_w + 2 * halo_x < W ? bin_w + 2 * halo_x : W;
  const int th = bin_h + 2 * halo_y < H ? bin_h + 2 * halo_y : H;
  (void)max_shift_x; (void)max_shift_y; (void)D; (void)depth_in_lds; (void)value_bf16;
  if (tw_out) *tw_out = tw;
  if (th_out) *th_out = th;
  if (lds_bytes_out) *lds_bytes_out = (tw * th + 1) * Cm * 4;
  if (nbuf_out) *nbuf_out = 1;
  if (depth_in_lds_out) *depth_in_lds_out = 0;
  return SGC_OK;
}

/* Same arithmetic as sgc_pairs_deform_gather (the reference kernels' order), operands head-major, pairs in the
 * binned order; the window parameters (bin / halo / head_shift) only choose what the GPU stages in LDS and cannot
 * change a result. */
int sgc_pairs_deform_gather_tiled(const void *value_hm_any, int value_bf16, const void *dist_any, const float *pair_ref,
                                  const int32_t *bin_offset, const float *raw_hm, const int32_t *head_shift_or_null,
                                  float *out, int N, int H, int W, int M, int Cm, int D, int P,
                                  int cam_stride_or_0, int bin_w, int bin_h, int halo_x, int halo_y,
                                  int max_shift_x, int max_shift_y, int depth_in_lds, sgc_stream_t stream) {
  (void)stream; (void)head_shift_or_null; (void)halo_x; (void)halo_y; (void)max_shift_x; (void)max_shift_y; (void)depth_in_lds;
  if (!value_hm_any || !dist_any || !pair_ref || !bin_offset || !raw_hm || !out) return fail(SGC_EINVAL, "null pointer");
  if (P > 64) return fail(SGC_EUNSUP, "P > 64");
  if (cam_stride_or_0 > 0 && cam_stride_or_0 < H * W) return fail(SGC_EINVAL, "cam_stride < H*W");
  const int64_t S = cam_stride_or_0 > 0 ? cam_stride_or_0 : (int64_t)H * W;
  const float *value_hm = (const float *)value_hm_any;
  const float *dist = (const float *)dist_any;
  float *widened = NULL, *dwidened = NULL;
  if (value_bf16) {          /* bf16 storage mode: value map AND depth maps are bf16; the taps are widened to fp32 (exact) */
    const int64_t n = (int64_t)N * M * S * Cm, nd = (int64_t)N * S * D;
    widened = (float *)malloc(sizeof(float) * (size_t)n);
    dwidened = (float *)malloc(sizeof(float) * (size_t)nd);
    if (!widened || !dwidened) { free(widened); free(dwidened); return fail(SGC_EINVAL, "out of memory"); }
    const uint16_t *h = (const uint16_t *)value_hm_any, *hd = (const uint16_t *)dist_any;
    for (int64_t i = 0; i < n; ++i) widened[i] = bf16_to_f32(h[i]);
    for (int64_t i = 0; i < nd; ++i) dwidened[i] = bf16_to_f32(hd[i]);
    value_hm = widened;
    dist = dwidened;
  }
  const int nb = ((W + bin_w - 1) / bin_w) * ((H + bin_h - 1) / bin_h);
  const int MC = M * Cm;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < N * nb; ++t) {
    const int n = t / nb;
    for (int i = bin_offset[t]; i < bin_offset[t + 1]; ++i) {
      const float *r = pair_ref + (int64_t)i * 4;
      for (int m = 0; m < M; ++m) {
        const float *rw = raw_hm + ((int64_t)i * M + m) * P * 4;         /* [P][4] = (du, dv, dz, logit) */
        float mx = rw[3];
        for (int p = 1; p < P; ++p) mx = fmaxf(mx, rw[p * 4 + 3]);
        float e[64], sum = 0.f;
        for (int p = 0; p < P; ++p) { e[p] = expf(rw[p * 4 + 3] - mx); sum += e[p]; }
        float *o = out + (int64_t)i * MC + m * Cm;
        for (int c = 0; c < Cm; ++c) o[c] = 0.f;
        const float *plane = value_hm + ((int64_t)n * M + m) * S * Cm;
        for (int p = 0; p < P; ++p) {
          const float x = r[0] + rw[p * 4] / (float)W;
          const float y = r[1] + rw[p * 4 + 1] / (float)H;
          const float z = r[2] + rw[p * 4 + 2] / (float)D;
          const float aw = e[p] / sum;
          float s[4];
          depth_score_sample(dist + n * S * D, D, 0, H, W, D, x, y, z, s);
          const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
          if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
            for (int c = 0; c < Cm; ++c) o[c] += wms_bilinear(plane, H, W, Cm, c, h_im, w_im, s) * aw;
        }
      }
    }
  }
  free(widened);
  free(dwidened);
  return SGC_OK;
}

/* ---- 5. inter-view aggregation ---------------------------------------------- */
int sgc_view_mean(const float *feat, const int32_t *slot, const int32_t *valid_index,
                  float *mean, int N, int Nq, int C, const int32_t *n_valid_dev_or_null, int n_valid,
                  sgc_stream_t stream) {
  (void)stream;
  if (!feat || !slot || !valid_index || !mean) return fail(SGC_EINVAL, "null pointer");
  if (n_valid_dev_or_null && *n_valid_dev_or_null < n_valid) n_valid = *n_valid_dev_or_null;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_valid; ++i) {
    const int q = valid_index[i];
    float *o = mean + (int64_t)i * C;
    for (int c = 0; c < C; ++c) o[c] = 0.f;
    int cnt = 0;
    for (int n = 0; n < N; ++n) { /* (valid_slots * valid_mask).sum(dim=0), :826 */
      const int32_t p = slot[(int64_t)n * Nq + q];
      if (p < 0) continue;
      ++cnt;
      for (int c = 0; c < C; ++c) o[c] += feat[(int64_t)p * C + c];
    }
    for (int c = 0; c < C; ++c) o[c] /= (float)cnt;
  }
  return SGC_OK;
}

int sgc_view_attend(const float *q, const float *kv, const int32_t *slot,
                    const int32_t *valid_index, float *ctx,
                    int N, int Nq, int C, int heads, const int32_t *n_valid_dev_or_null, int n_valid,
                    sgc_stream_t stream) {
  (void)stream;
  if (!q || !kv || !slot || !valid_index || !ctx) return fail(SGC_EINVAL, "null pointer");
  if (C % heads) return fail(SGC_EINVAL, "C % heads != 0");
  if (n_valid_dev_or_null && *n_valid_dev_or_null < n_valid) n_valid = *n_valid_dev_or_null;
  const int hd = C / heads;
  const float scale = sqrtf(1.0f / (float)hd); /* torch MHA: q * sqrt(1/head_dim) */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_valid; ++i) {
    const int vq = valid_index[i];
    for (int h = 0; h < heads; ++h) {
      const float *qh = q + (int64_t)i * C + h * hd;
      float mx = -INFINITY;
      for (int n = 0; n < N; ++n) {
        const int32_t p = slot[(int64_t)n * Nq + vq];
        if (p < 0) continue;
        const float *k = kv + (int64_t)p * 2 * C + h * hd;
        float d = 0.f;
        for (int c = 0; c < hd; ++c) d += (qh[c] * scale) * k[c];
        if (d > mx) mx = d;
      }
      float sum = 0.f;
      float *o = ctx + (int64_t)i * C + h * hd;
      for (int c = 0; c < hd; ++c) o[c] = 0.f;
      for (int n = 0; n < N; ++n) {
        const int32_t p = slot[(int64_t)n * Nq + vq];
        if (p < 0) continue;
        const float *k = kv + (int64_t)p * 2 * C + h * hd;
        const float *v = k + C;
        float d = 0.f;
        for (int c = 0; c < hd; ++c) d += (qh[c] * scale) * k[c];
        const float e = expf(d - mx);
        sum += e;
        for (int c = 0; c < hd; ++c) o[c] += e * v[c];
      }
      for (int c = 0; c < hd; ++c) o[c] /= sum;
    }
  }
  return SGC_OK;
}

/* Projected-query form of the view softmax (see the header): qp [n_valid, heads, C], x [n_pairs, C] -> s [n_valid, heads, C].
 * Plain restatement of the formula, double accumulation; the reference function it equals is the MHA above
 * (TU/deformable_cross_attention.py:826-833) with k = W_k x + b_k, v = W_v x + b_v folded by the caller. */
int sgc_view_attend_pq_supported(int N, int C, int heads) { return heads > 0 && C > 0 && N > 0 && N <= 4096; }
int sgc_view_attend_pq(const float *qp, const float *x, const int32_t *slot, const int32_t *valid_index, float *s,
                       int N, int Nq, int C, int heads, const int32_t *n_valid_dev_or_null, int n_valid,
                       sgc_stream_t stream) {
  (void)stream;
  if (!qp || !x || !slot || !valid_index || !s) return fail(SGC_EINVAL, "null pointer");
  if (heads <= 0 || N > 4096) return fail(SGC_EUNSUP, "heads / N");
  if (n_valid_dev_or_null && *n_valid_dev_or_null < n_valid) n_valid = *n_valid_dev_or_null;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_valid; ++i) {
    const int vq = valid_index[i];
    double d[4096];
    for (int h = 0; h < heads; ++h) {
      const float *q = qp + ((int64_t)i * heads + h) * C;
      float *o = s + ((int64_t)i * heads + h) * C;
      double mx = -INFINITY, sum = 0.0;
      for (int n = 0; n < N; ++n) {
        const int32_t p = slot[(int64_t)n * Nq + vq];
        d[n] = 0.0;
        if (p < 0) continue;
        const float *xr = x + (int64_t)p * C;
        double acc = 0.0;
        for (int c = 0; c < C; ++c) acc += (double)q[c] * (double)xr[c];
        d[n] = acc;
        if (acc > mx) mx = acc;
      }
      for (int n = 0; n < N; ++n)
        if (slot[(int64_t)n * Nq + vq] >= 0) sum += exp(d[n] - mx);
      for (int c = 0; c < C; ++c) {
        double acc = 0.0;
        for (int n = 0; n < N; ++n) {
          const int32_t p = slot[(int64_t)n * Nq + vq];
          if (p >= 0) acc += exp(d[n] - mx) / sum * (double)x[(int64_t)p * C + c];
        }
        o[c] = (float)acc;
      }
    }
  }
  return SGC_OK;
}

/* backward of the view softmax above (double accumulation): see the header */
int sgc_view_attend_backward(const float *q, const float *kv, const int32_t *slot, const int32_t *valid_index,
                             const float *ctx, const float *grad_ctx, float *grad_q, float *grad_kv,
                             int N, int Nq, int C, int heads, int n_valid, sgc_stream_t stream) {
  (void)stream; (void)ctx;
  if (!q || !kv || !slot || !valid_index || !grad_ctx || !grad_q || !grad_kv) return fail(SGC_EINVAL, "null pointer");
  if (C % heads) return fail(SGC_EINVAL, "C % heads != 0");
  const int hd = C / heads;
  const double scale = sqrt(1.0 / (double)hd);
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < n_valid; ++i) {
    const int vq = valid_index[i];
    for (int h = 0; h < heads; ++h) {
      const float *qh = q + (int64_t)i * C + h * hd, *dc = grad_ctx + (int64_t)i * C + h * hd;
      double d[256], a[256], mx = -INFINITY, sum = 0.0;
      int ps[256], cnt = 0;
      for (int n = 0; n < N && cnt < 256; ++n) {
        const int32_t p = slot[(int64_t)n * Nq + vq];
        if (p < 0) continue;
        const float *k = kv + (int64_t)p * 2 * C + h * hd;
        double s = 0.0;
        for (int c = 0; c < hd; ++c) s += (double)qh[c] * scale * (double)k[c];
        d[cnt] = s; ps[cnt++] = p;
        if (s > mx) mx = s;
      }
      for (int t = 0; t < cnt; ++t) { a[t] = exp(d[t] - mx); sum += a[t]; }
      double S = 0.0;
      for (int t = 0; t < cnt; ++t) {
        a[t] /= sum;
        const float *v = kv + (int64_t)ps[t] * 2 * C + C + h * hd;
        double da = 0.0;
        for (int c = 0; c < hd; ++c) da += (double)dc[c] * (double)v[c];
        d[t] = da;                                   /* reuse: d[] now holds da_n */
        S += a[t] * da;
      }
      float *gq = grad_q + (int64_t)i * C + h * hd;
      for (int c = 0; c < hd; ++c) gq[c] = 0.f;
      for (int t = 0; t < cnt; ++t) {
        const double ds = a[t] * (d[t] - S);
        const float *k = kv + (int64_t)ps[t] * 2 * C + h * hd;
        float *gk = grad_kv + (int64_t)ps[t] * 2 * C + h * hd;
        for (int c = 0; c < hd; ++c) {
          gq[c] += (float)(ds * (double)k[c] * scale);
          gk[c] = (float)(ds * (double)qh[c] * scale);
          gk[C + c] = (float)(a[t] * (double)dc[c]);
        }
      }
    }
  }
  return SGC_OK;
}

/* ---- 6. volume glue ------------------------------------------------------------ */
int sgc_scatter_rows(const float *rows, const int32_t *idx, const int32_t *idx2_or_null,
                     float *vol, const int32_t *n_dev_or_null, int n, int C, sgc_stream_t stream) {
  (void)stream;
  if (!rows || !idx || !vol) return fail(SGC_EINVAL, "null pointer");
  if (n_dev_or_null && *n_dev_or_null < n) n = *n_dev_or_null;
  for (int i = 0; i < n; ++i) {
    int64_t d = idx[i];
    if (idx2_or_null) d = idx2_or_null[d];
    memcpy(vol + d * C, rows + (int64_t)i * C, sizeof(float) * C);
  }
  return SGC_OK;
}

int sgc_nchw_to_nhwc_crop(const float *src, float *dst, int N, int C, int Hs, int Ws,
                          int H, int W, int step, sgc_stream_t stream) {
  (void)stream;
  if (!src || !dst) return fail(SGC_EINVAL, "null pointer");
  if (step < 1 || (int64_t)(H - 1) * step >= Hs || (int64_t)(W - 1) * step >= Ws) return fail(SGC_EINVAL, "crop larger than source");
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
          dst[(((int64_t)n * H + h) * W + w) * C + c] = src[(((int64_t)n * C + c) * Hs + (int64_t)h * step) * Ws + (int64_t)w * step];
  return SGC_OK;
}

/* adjoint of the crop + transpose: dst [N, C, Hd, Wd] <- src [N, H*W, C], zero outside the H x W crop */
int sgc_nhwc_to_nchw_pad(const float *src, float *dst, int N, int C, int H, int W, int Hd, int Wd, sgc_stream_t stream) {
  (void)stream;
  if (!src || !dst) return fail(SGC_EINVAL, "null pointer");
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Hd < H || Wd < W) return fail(SGC_EINVAL, "bad sizes");
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < Hd; ++h)
        for (int w = 0; w < Wd; ++w)
          dst[(((int64_t)n * C + c) * Hd + h) * Wd + w] = (h < H && w < W) ? src[(((int64_t)n * H + h) * W + w) * C + c] : 0.f;
  return SGC_OK;
}

/* ---- 7. dense 3D convolution, channels-last (naive loops; semantics of nn.Conv3d(k,s,pad=k/2) /
 * nn.ConvTranspose3d(2,2) + folded BatchNorm + residual + ReLU as chained in necks/imvoxelnet.py) ---- */
int sgc_conv3d_cl_f32(const float *x, const float *wt, const float *scale, const float *shift,
                      const float *residual_or_null, float *y,
                      int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                      int transposed, int relu, float *workspace_or_null, int64_t workspace_floats,
                      sgc_stream_t stream) {
  (void)workspace_or_null; (void)workspace_floats;      /* the oracle never splits a reduction */
  (void)stream;
  if (!x || !wt || !y) return fail(SGC_EINVAL, "null pointer");
  const int pad = ksize == 2 ? 0 : ksize / 2;        /* k2 s2 p0: the adjoint geometry of ConvTranspose3d(2, 2) */
  int ox, oy, oz;
  if (transposed) { ox = 2 * ix; oy = 2 * iy; oz = 2 * iz; }
  else { ox = (ix + 2 * pad - ksize) / stride + 1; oy = (iy + 2 * pad - ksize) / stride + 1; oz = (iz + 2 * pad - ksize) / stride + 1; }
#pragma omp parallel for collapse(2) schedule(static)
  for (int a = 0; a < ox; ++a)
    for (int b = 0; b < oy; ++b)
      for (int c = 0; c < oz; ++c) {
        const int64_t orow = ((int64_t)a * oy + b) * oz + c;
        for (int co = 0; co < Cout; ++co) {
          float acc = 0.f;
          if (transposed) {
            const int par = ((a & 1) * 2 + (b & 1)) * 2 + (c & 1);
            const float *xi = x + (((int64_t)(a / 2) * iy + b / 2) * iz + c / 2) * Cin;
            const float *w = wt + ((int64_t)par * Cout + co) * Cin;
            for (int ci = 0; ci < Cin; ++ci) acc += xi[ci] * w[ci];
          } else {
            for (int kx = 0; kx < ksize; ++kx)
              for (int ky = 0; ky < ksize; ++ky)
                for (int kz = 0; kz < ksize; ++kz) {
                  const int xx = a * stride + kx - pad, yy = b * stride + ky - pad, zz = c * stride + kz - pad;
                  if (xx < 0 || xx >= ix || yy < 0 || yy >= iy || zz < 0 || zz >= iz) continue;
                  const float *xi = x + (((int64_t)xx * iy + yy) * iz + zz) * Cin;
                  const float *w = wt + ((int64_t)((kx * ksize + ky) * ksize + kz) * Cout + co) * Cin;
                  for (int ci = 0; ci < Cin; ++ci) acc += xi[ci] * w[ci];
                }
          }
          float v = acc * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
          if (relu == 2 && v < 0.f) v = 0.f;
          if (residual_or_null) v += residual_or_null[orow * Cout + co];
          if (relu == 1 && v < 0.f) v = 0.f;
          y[orow * Cout + co] = v;
        }
      }
  return SGC_OK;
}

/* bf16x3 entry point: the oracle is the fp32 truth -- it rebuilds w = float(hi) + float(lo) and runs
 * the naive fp32 convolution above. */
int64_t sgc_conv3d_workspace_floats(int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                                    int transposed, int bf16x3) {
  (void)ix; (void)iy; (void)iz; (void)Cin; (void)Cout; (void)ksize; (void)stride; (void)transposed; (void)bf16x3;
  return 0;
}

static uint16_t f32_to_bf16_rne(float f);

/* IEEE binary16 <-> binary32 in integer arithmetic (gcc 11 has no _Float16 on x86-64): round to nearest even, values
 * beyond the half range SATURATE at +-65504 (csrc/mma.hpp op_hi), NaN stays NaN, subnormal halves are kept */
static uint16_t f32_to_f16_sat(float f) {
  union { float f; uint32_t u; } c; c.f = f;
  const uint32_t sign = (c.u >> 16) & 0x8000u, a = c.u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                    /* NaN */
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);                   /* >= 65520 rounds past the largest half: saturate */
  if (a < 0x33000001u) return (uint16_t)sign;                                /* < 2^-25 (ties to even at 2^-25): zero */
  const int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;                                  /* 24-bit significand */
  int shift = e >= -14 ? 13 : 13 + (-14 - e);                                /* bits dropped; subnormal halves drop more */
  uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;
  uint32_t h = e >= -14 ? ((uint32_t)(e + 15) << 10) + (q - 0x400u) : q;     /* a carry out of the significand bumps the exponent */
  return (uint16_t)(sign | h);
}
static float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  union { float f; uint32_t u; } c;
  if (e == 0) { c.f = (float)m * 5.9604644775390625e-8f; c.u |= sign; return c.f; }   /* subnormal: m * 2^-24 */
  if (e == 31) { c.u = sign | 0x7f800000u | (m << 13); return c.f; }
  c.u = sign | ((e + 112u) << 23) | (m << 13);
  return c.f;
}
/* an activation / a weight as the mode's MFMA sees it */
static float mode_x(float x) {
  return g_conv_products == 1 ? bf16_to_f32(f32_to_bf16_rne(x)) : g_conv_products == 2 ? f16_to_f32(f32_to_f16_sat(x)) : x;
}
static float mode_w(uint16_t hi, uint16_t lo) {
  return g_conv_products == 3 ? bf16_to_f32(hi) + bf16_to_f32(lo) : g_conv_products == 1 ? bf16_to_f32(hi) : f16_to_f32(hi);
}

int sgc_conv3d_cl_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                         const float *shift, const float *residual_or_null, float *y,
                         int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                         int transposed, int relu, float *workspace_or_null, int64_t workspace_floats,
                         sgc_stream_t stream) {
  (void)workspace_or_null; (void)workspace_floats;
  if (!w_hi || !w_lo) return fail(SGC_EINVAL, "null pointer");
  const int taps = transposed ? 8 : ksize * ksize * ksize;
  const size_t n = (size_t)taps * Cout * Cin;
  float *w = (float *)malloc(n * sizeof(float));
  if (!w) return fail(SGC_EINVAL, "out of memory");
  for (size_t i = 0; i < n; ++i) w[i] = mode_w(w_hi[i], w_lo[i]);
  float *xr = NULL;
  if (g_conv_products != 3) {                     /* one-product modes: the activations are rounded to the operand format as well */
    const size_t nx = (size_t)ix * iy * iz * Cin;
    xr = (float *)malloc(nx * sizeof(float));
    if (!xr) { free(w); return fail(SGC_EINVAL, "out of memory"); }
    for (size_t i = 0; i < nx; ++i) xr[i] = mode_x(x[i]);
  }
  const int rc = sgc_conv3d_cl_f32(xr ? xr : x, w, scale, shift, residual_or_null, y, ix, iy, iz, Cin, Cout, ksize, stride,
                                   transposed, relu, NULL, 0, stream);
  free(w); free(xr);
  return rc;
}

/* nn.Conv2d(k, padding=k/2) over channels-last images + the same epilogue (mmdet FPN lateral / output convolutions) */
int sgc_conv2d_nhwc_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                           const float *shift, const float *residual_or_null, float *y, int N, int H, int W,
                           int Cin, int Cout, int ksize, int relu, sgc_stream_t stream) {
  (void)stream;
  if (!x || !w_hi || !w_lo || !y) return fail(SGC_EINVAL, "null pointer");
  if (ksize != 1 && ksize != 3) return fail(SGC_EUNSUP, "ksize in {1,3}");
  const int pad = ksize / 2;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        const int64_t orow = ((int64_t)n * H + h) * W + w;
        for (int co = 0; co < Cout; ++co) {
          float acc = 0.f;
          for (int ky = 0; ky < ksize; ++ky)
            for (int kx = 0; kx < ksize; ++kx) {
              const int hh = h + ky - pad, ww = w + kx - pad;
              if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
              const float *xi = x + (((int64_t)n * H + hh) * W + ww) * Cin;
              const int64_t wo = ((int64_t)(ky * ksize + kx) * Cout + co) * Cin;
              for (int ci = 0; ci < Cin; ++ci)
                acc += mode_x(xi[ci]) * mode_w(w_hi[wo + ci], w_lo[wo + ci]);
            }
          float v = acc * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
          if (relu == 2 && v < 0.f) v = 0.f;
          if (residual_or_null) v += residual_or_null[orow * Cout + co];
          if (relu == 1 && v < 0.f) v = 0.f;
          y[orow * Cout + co] = v;
        }
      }
  return SGC_OK;
}

/* weight gradient of the convolution above: dW[tap][co][ci] = sum_o dy[o][co] * x[nbr(o, tap)][ci]
 * (what autograd returns for nn.Conv3d.weight, permuted to the kernel's [tap][Cout][Cin] layout); double accumulation */
int64_t sgc_conv3d_wgrad_workspace_floats(int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride) {
  (void)ix; (void)iy; (void)iz; (void)Cin; (void)Cout; (void)ksize; (void)stride;
  return 0;
}

int sgc_conv3d_wgrad_bf16x3(const float *x, const float *dy, float *dw, int ix, int iy, int iz, int Cin, int Cout,
                            int ksize, int stride, float *workspace_or_null, int64_t workspace_floats, sgc_stream_t stream) {
  (void)workspace_or_null; (void)workspace_floats; (void)stream;
  if (!x || !dy || !dw) return fail(SGC_EINVAL, "null pointer");
  const int pad = ksize == 2 ? 0 : ksize / 2;
  const int ox = (ix + 2 * pad - ksize) / stride + 1, oy = (iy + 2 * pad - ksize) / stride + 1, oz = (iz + 2 * pad - ksize) / stride + 1;
  const int taps = ksize * ksize * ksize;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int tap = 0; tap < taps; ++tap)
    for (int co = 0; co < Cout; ++co) {
      const int kx = tap / (ksize * ksize), ky = (tap / ksize) % ksize, kz = tap % ksize;
      double *acc = (double *)calloc((size_t)Cin, sizeof(double));
      for (int a = 0; a < ox; ++a)
        for (int b = 0; b < oy; ++b)
          for (int c = 0; c < oz; ++c) {
            const int xx = a * stride + kx - pad, yy = b * stride + ky - pad, zz = c * stride + kz - pad;
            if (xx < 0 || xx >= ix || yy < 0 || yy >= iy || zz < 0 || zz >= iz) continue;
            const float g = dy[(((int64_t)a * oy + b) * oz + c) * Cout + co];
            const float *xi = x + (((int64_t)xx * iy + yy) * iz + zz) * Cin;
            for (int ci = 0; ci < Cin; ++ci) acc[ci] += (double)g * (double)xi[ci];
          }
      float *o = dw + ((int64_t)tap * Cout + co) * Cin;
      for (int ci = 0; ci < Cin; ++ci) o[ci] = (float)acc[ci];
      free(acc);
    }
  return SGC_OK;
}

/* masked form: the oracle computes every row (the mask only licenses the GPU to skip work) */
int sgc_conv3d_cl_bf16x3_masked(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                                const float *shift, const float *residual_or_null, float *y, const uint8_t *out_mask,
                                int ix, int iy, int iz, int Cin, int Cout, int relu,
                                float *workspace_or_null, int64_t workspace_floats, sgc_stream_t stream) {
  if (!out_mask) return fail(SGC_EINVAL, "null mask");
  return sgc_conv3d_cl_bf16x3(x, w_hi, w_lo, scale, shift, residual_or_null, y, ix, iy, iz, Cin, Cout, 3, 1, 0, relu,
                              workspace_or_null, workspace_floats, stream);
}

/* the head's fused convolution with exp(scale(reg)) on columns [act_c0, act_c1) (dense_heads/imvoxel_head_v2.py:79,103-110):
 * the dense convolution, then the activation as the reference applies it -- x * scale (mmcv Scale), then exp */
int sgc_conv3d_cl_bf16x3_act(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                             const float *shift, const float *residual_or_null, float *y, const uint8_t *out_mask_or_null,
                             int ix, int iy, int iz, int Cin, int Cout, int relu, int act_c0, int act_c1,
                             const float *act_scale_dev, float *workspace_or_null, int64_t workspace_floats,
                             sgc_stream_t stream) {
  (void)out_mask_or_null;
  if (!act_scale_dev || act_c0 < 0 || act_c1 > Cout || act_c1 <= act_c0) return fail(SGC_EINVAL, "bad activation range");
  const int rc = sgc_conv3d_cl_bf16x3(x, w_hi, w_lo, scale, shift, residual_or_null, y, ix, iy, iz, Cin, Cout, 3, 1, 0, relu,
                                      workspace_or_null, workspace_floats, stream);
  if (rc) return rc;
  const int64_t OV = (int64_t)ix * iy * iz;
  const float s = *act_scale_dev;
  for (int64_t v = 0; v < OV; ++v)
    for (int c = act_c0; c < act_c1; ++c) y[v * Cout + c] = expf(y[v * Cout + c] * s);
  return SGC_OK;
}

/* Winograd F(2,3) along z, restated literally (see the header): transform-domain products in double from the TRANSFORMED weights
 * the caller passes (hi + lo), output transform, then the dense entry point's epilogue.  The test that pins it compares with
 * sgc_conv3d_cl_bf16x3 on the ORIGINAL weights. */
int sgc_conv3d_winograd_z_supported(int ix, int iy, int iz, int Cin, int Cout) {
  return ix > 0 && iy > 0 && iz >= 2 && iz % 2 == 0 && Cin > 0 && Cout > 0;
}
int64_t sgc_conv3d_winograd_z_workspace_floats(int ix, int iy, int iz, int Cin, int Cout) {
  return (int64_t)2 * ix * iy * iz * ((int64_t)Cin + Cout);
}
static float bf16_bits_to_f32(uint16_t b) { union { uint32_t u; float f; } c; c.u = (uint32_t)b << 16; return c.f; }
int sgc_conv3d_winograd_z_bf16x3(const float *x, const uint16_t *wg_hi, const uint16_t *wg_lo, const float *scale,
                                 const float *shift, const float *residual_or_null, float *y, int ix, int iy, int iz,
                                 int Cin, int Cout, int relu, float *workspace, int64_t workspace_floats, sgc_stream_t stream) {
  (void)stream; (void)workspace; (void)workspace_floats;
  if (!x || !wg_hi || !wg_lo || !y) return fail(SGC_EINVAL, "null pointer");
  if (iz % 2) return fail(SGC_EUNSUP, "iz must be even");
  const int J = iz / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int xx = 0; xx < ix; ++xx)
    for (int yy = 0; yy < iy; ++yy) {
      double *m = (double *)malloc(sizeof(double) * 4 * (size_t)Cout);
      for (int j = 0; j < J; ++j) {
        for (int i = 0; i < 4 * Cout; ++i) m[i] = 0.0;
        for (int dx = 0; dx < 3; ++dx)
          for (int dy = 0; dy < 3; ++dy) {
            const int sx = xx + dx - 1, sy = yy + dy - 1;
            if (sx < 0 || sx >= ix || sy < 0 || sy >= iy) continue;
            const float *col = x + (((int64_t)sx * iy + sy) * iz) * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
              const float d0 = 2 * j - 1 >= 0 ? col[(int64_t)(2 * j - 1) * Cin + ci] : 0.f, d1 = col[(int64_t)(2 * j) * Cin + ci];
              const float d2 = col[(int64_t)(2 * j + 1) * Cin + ci], d3 = 2 * j + 2 < iz ? col[(int64_t)(2 * j + 2) * Cin + ci] : 0.f;
              const float t[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};          /* one fp32 rounding each, as on the GPU */
              for (int k = 0; k < 4; ++k) {
                if (t[k] == 0.f) continue;
                const int64_t wb = (((int64_t)k * 9 + dx * 3 + dy) * Cout) * Cin + ci;
                for (int co = 0; co < Cout; ++co) {
                  const int64_t wi = wb + (int64_t)co * Cin;
                  m[k * Cout + co] += (double)t[k] * ((double)bf16_bits_to_f32(wg_hi[wi]) + (double)bf16_bits_to_f32(wg_lo[wi]));
                }
              }
            }
          }
        for (int q = 0; q < 2; ++q)
          for (int co = 0; co < Cout; ++co) {
            const double a = q == 0 ? m[co] + m[Cout + co] + m[2 * Cout + co] : m[Cout + co] - m[2 * Cout + co] - m[3 * Cout + co];
            const int64_t o = (((int64_t)xx * iy + yy) * iz + 2 * j + q) * Cout + co;
            float v = (float)a * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
            if (relu == 2) v = v > 0.f ? v : 0.f;
            if (residual_or_null) v += residual_or_null[o];
            if (relu == 1) v = v > 0.f ? v : 0.f;
            y[o] = v;
          }
      }
      free(m);
    }
  return SGC_OK;
}

int sgc_mask_dilate3(const uint8_t *mask_in, uint8_t *mask_out, int X, int Y, int Z, sgc_stream_t stream) {
  (void)stream;
  if (!mask_in || !mask_out || mask_in == mask_out) return fail(SGC_EINVAL, "null or aliased pointers");
  for (int x = 0; x < X; ++x)
    for (int y = 0; y < Y; ++y)
      for (int z = 0; z < Z; ++z) {
        uint8_t any = 0;
        for (int dx = -1; dx <= 1; ++dx)
          for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
              const int a = x + dx, b = y + dy, c = z + dz;
              if (a >= 0 && a < X && b >= 0 && b < Y && c >= 0 && c < Z) any |= mask_in[((int64_t)a * Y + b) * Z + c];
            }
        mask_out[((int64_t)x * Y + y) * Z + z] = any ? 1 : 0;
      }
  return SGC_OK;
}

/* nn.Upsample(size, mode='trilinear')(valid.float()).round().bool() (imvoxel_head_v2.py:123,258), integer factors:
 * align_corners=False samples half-way between fine voxels f*d + f/2 - 1 and f*d + f/2; round() is half-to-even */
int sgc_valid_pyramid(const int64_t *valid, uint8_t *mask_out, int X, int Y, int Z, int factor, sgc_stream_t stream) {
  (void)stream;
  if (!valid || !mask_out) return fail(SGC_EINVAL, "null pointer");
  if (factor < 1 || (factor & (factor - 1)) || X % factor || Y % factor || Z % factor) return fail(SGC_EUNSUP, "bad factor");
  const int f = factor, cx = X / f, cy = Y / f, cz = Z / f, o = f / 2 - 1;
  for (int x = 0; x < cx; ++x)
    for (int y = 0; y < cy; ++y)
      for (int z = 0; z < cz; ++z) {
        int cnt = 0;
        if (f == 1) cnt = valid[((int64_t)x * Y + y) * Z + z] != 0 ? 8 : 0;
        else
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
              for (int c = 0; c < 2; ++c)
                cnt += valid[((int64_t)(x * f + o + a) * Y + (y * f + o + b)) * Z + (z * f + o + c)] != 0;
        mask_out[((int64_t)x * cy + y) * cz + z] = cnt >= 5;
      }
  return SGC_OK;
}

/* Linear over a row list whose length lives on the "device" (here: host memory): fp32 truth of
 * sgc_linear_rows_bf16x3; rows past the count are left untouched. */
int sgc_linear_rows_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                           float *y, const int32_t *rows_dev_or_null, int rows_cap, int Cin, int Cout,
                           sgc_stream_t stream) {
  if (!x || !w_hi || !w_lo || !y) return fail(SGC_EINVAL, "null pointer");
  int rows = rows_cap;
  if (rows_dev_or_null && *rows_dev_or_null < rows) rows = *rows_dev_or_null;
  if (rows <= 0) return SGC_OK;
  return sgc_conv3d_cl_bf16x3(x, w_hi, w_lo, NULL, shift, NULL, y, rows, 1, 1, Cin, Cout, 1, 1, 0, 0, NULL, 0, stream);
}

/* the geometry-aware sample followed by the Linear that consumes it (product: one fused pass, csrc/rows_gemm.hip GATHER form): here
 * simply the two steps through a temporary */
int sgc_pairs_geometry_linear_supported(int C, int Cout, int N, int S) { (void)N; (void)S; return C % 32 == 0 && Cout % 4 == 0; }
int64_t sgc_pairs_geometry_linear_workspace_bytes(int cap) { return cap > 0 ? (int64_t)cap * 32 : 0; }
int sgc_pairs_geometry_linear_bf16x3(const float *feat, const float *dist, const float *ref_cam, const int32_t *pair_cam,
                                     const int32_t *pair_q, const int32_t *totals, const uint16_t *w_hi, const uint16_t *w_lo,
                                     const float *shift_or_null, float *y, void *workspace, int N, int Nq, int H, int W, int C,
                                     int D, int Cout, int cam_stride_or_0, int n_pairs_or_neg, int cap, sgc_stream_t stream) {
  (void)workspace;
  if (!feat || !dist || !ref_cam || !pair_cam || !pair_q || !w_hi || !w_lo || !y) return fail(SGC_EINVAL, "null pointer");
  int np = n_pairs_or_neg >= 0 ? n_pairs_or_neg : (totals ? totals[0] : -1);
  if (np < 0 || np > cap) return fail(SGC_EINVAL, "n_pairs out of range");
  if (np == 0) return SGC_OK;
  float *geo = (float *)malloc(sizeof(float) * (size_t)np * C);
  int rc = sgc_pairs_geometry_sample(feat, dist, ref_cam, pair_cam, pair_q, totals, geo, N, Nq, H, W, C, D, cam_stride_or_0, np, np, stream);
  if (!rc) rc = sgc_linear_rows_bf16x3(geo, w_hi, w_lo, shift_or_null, y, NULL, np, C, Cout, stream);
  free(geo);
  return rc;
}

/* block-diagonal Linear: group g's K inputs -> its Nh outputs (product: csrc/rows_blockdiag.hip; reference: the V rows of
 * nn.MultiheadAttention.in_proj_weight applied per head, TU/deformable_cross_attention.py:826-833).  Here: one sgc_linear_rows_bf16x3
 * per group on a copy of its K columns. */
int sgc_linear_rows_blockdiag_supported(int G, int K, int Nh) { return G > 0 && K % 32 == 0 && Nh % 4 == 0; }
int sgc_linear_rows_blockdiag_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift_or_null,
                                     float *y, const int32_t *rows_dev_or_null, int rows_cap, int G, int K, int Nh,
                                     sgc_stream_t stream) {
  if (!x || !w_hi || !w_lo || !y) return fail(SGC_EINVAL, "null pointer");
  int rows = rows_cap;
  if (rows_dev_or_null && *rows_dev_or_null < rows) rows = *rows_dev_or_null;
  if (rows <= 0) return SGC_OK;
  float *xg = (float *)malloc(sizeof(float) * (size_t)rows * K), *yg = (float *)malloc(sizeof(float) * (size_t)rows * Nh);
  int rc = SGC_OK;
  for (int g = 0; g < G && !rc; ++g) {
    for (int r = 0; r < rows; ++r) memcpy(xg + (size_t)r * K, x + (size_t)r * G * K + (size_t)g * K, sizeof(float) * K);
    rc = sgc_linear_rows_bf16x3(xg, w_hi + (size_t)g * Nh * K, w_lo + (size_t)g * Nh * K, shift_or_null ? shift_or_null + g * Nh : NULL,
                                yg, NULL, rows, K, Nh, stream);
    for (int r = 0; r < rows && !rc; ++r) memcpy(y + (size_t)r * G * Nh + (size_t)g * Nh, yg + (size_t)r * Nh, sizeof(float) * Nh);
  }
  free(xg); free(yg);
  return rc;
}

/* the same with one all-zero row behind the result (y holds rows_cap + 1 rows) */
int sgc_linear_rows_zrow_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                                float *y, const int32_t *rows_dev_or_null, int rows_cap, int Cin, int Cout,
                                sgc_stream_t stream) {
  if (rows_cap <= 0 || !y) return fail(SGC_EINVAL, "needs rows_cap > 0 and an output");
  for (int c = 0; c < Cout; ++c) y[(int64_t)rows_cap * Cout + c] = 0.f;
  return sgc_linear_rows_bf16x3(x, w_hi, w_lo, shift, y, rows_dev_or_null, rows_cap, Cin, Cout, stream);
}

/* value_proj with a head-major result: y[n][h][s][j] = (x[n*S+s] @ W^T + shift)[h*Cm + j] */
static uint16_t f32_to_bf16_rne(float f) {
  union { float f; uint32_t u; } c; c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((c.u >> 16) | 0x40);      /* NaN stays NaN */
  return (uint16_t)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}

int sgc_linear_rows_headmajor_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                                     void *y_any, int y_bf16, int N, int S, int Cin, int M, int Cm, sgc_stream_t stream) {
  float *y = (float *)y_any;
  if (!x || !w_hi || !w_lo || !y) return fail(SGC_EINVAL, "null pointer");
  if (N <= 0 || S <= 0 || M <= 0 || Cm <= 0) return fail(SGC_EINVAL, "bad size");
  const int64_t rows = (int64_t)N * S;
  const int C = M * Cm;
  float *tmp = (float *)malloc(sizeof(float) * (size_t)rows * C);
  if (!tmp) return fail(SGC_EINVAL, "out of memory");
  const int rc = sgc_conv3d_cl_bf16x3(x, w_hi, w_lo, NULL, shift, NULL, tmp, (int)rows, 1, 1, Cin, C, 1, 1, 0, 0, NULL, 0, stream);
  if (rc == SGC_OK)
    for (int64_t r = 0; r < rows; ++r)
      for (int c = 0; c < C; ++c)
        {
          const int64_t o = (((r / S) * M + c / Cm) * S + r % S) * Cm + c % Cm;
          if (y_bf16) ((uint16_t *)y_any)[o] = f32_to_bf16_rne(tmp[r * C + c]);
          else y[o] = tmp[r * C + c];
        }
  free(tmp);
  return rc;
}

/* ---- 7b. top-k selection with a DEFINED tie rule, LayerNorm ------------------------------------------------ */
typedef struct { float v; int32_t i; } topk_item_t;
static int cmp_topk_item(const void *a, const void *b) {      /* value descending (NaN first), index ascending */
  const topk_item_t *x = (const topk_item_t *)a, *y = (const topk_item_t *)b;
  const int xn = x->v != x->v, yn = y->v != y->v;
  if (xn != yn) return yn - xn;
  if (!xn) {
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
  }
  return (x->i > y->i) - (x->i < y->i);
}

/* topk_wo_grad + nonzero (AdaptiveSparseHead.py:9-13,74; DenseHead.py:66) with ties at the cut broken by the lowest
 * flat index -- the rule the product's kernel implements (torch.topk's own tie order is implementation-defined) */
int sgc_topk_select(const float *score, int n, int k, int64_t *idx_out, int64_t *valid_or_null, float *mask_or_null,
                    sgc_stream_t stream) {
  (void)stream;
  if (!score || !idx_out) return fail(SGC_EINVAL, "null pointer");
  if (n <= 0 || k <= 0 || k > n) return fail(SGC_EINVAL, "need 0 < k <= n");
  topk_item_t *it = (topk_item_t *)malloc(sizeof(topk_item_t) * (size_t)n);
  uint8_t *sel = (uint8_t *)calloc((size_t)n, 1);
  if (!it || !sel) { free(it); free(sel); return fail(SGC_EINVAL, "out of memory"); }
  for (int i = 0; i < n; ++i) { it[i].v = score[i]; it[i].i = i; }
  qsort(it, (size_t)n, sizeof(topk_item_t), cmp_topk_item);
  for (int j = 0; j < k; ++j) sel[it[j].i] = 1;
  int c = 0;
  for (int i = 0; i < n; ++i) {
    if (sel[i]) idx_out[c++] = i;
    if (valid_or_null) valid_or_null[i] = sel[i];
    if (mask_or_null) mask_or_null[i] = sel[i] ? 1.f : 0.f;
  }
  free(it); free(sel);
  return SGC_OK;
}

/* many-workgroup entry point of the product: the oracle has one implementation */
int64_t sgc_topk_select_workspace_bytes(int n) { (void)n; return 0; }
int sgc_topk_select_ws(const float *score, int n, int k, int64_t *idx_out, int64_t *valid_or_null, float *mask_or_null,
                       void *workspace_or_null, int64_t workspace_bytes, sgc_stream_t stream) {
  (void)workspace_or_null; (void)workspace_bytes;
  return sgc_topk_select(score, n, k, idx_out, valid_or_null, mask_or_null, stream);
}

/* Training-mode BatchNorm over rows (test infrastructure; product: csrc/batch_norm.hip).  Statistics in double precision:
 * this is the checker of the product's fp32 Welford / Chan reductions, not a restatement of their order. */
int64_t sgc_bn_rows_workspace_floats(int rows, int C) { return rows > 0 && C > 0 ? 4 : 0; }

int sgc_bn_rows_forward(const float *x, const float *weight, const float *bias, float *running_mean_or_null,
                        float *running_var_or_null, float momentum, float eps, float *y, float *mean_out,
                        float *invstd_out, float *workspace, int64_t workspace_floats, int rows, int C,
                        sgc_stream_t stream) {
  (void)workspace; (void)workspace_floats; (void)stream;
  if (!x || !weight || !bias || !y || !mean_out || !invstd_out) return fail(SGC_EINVAL, "sgc_bn_rows_forward: null pointer");
  if (rows <= 0 || C <= 0) return fail(SGC_EINVAL, "sgc_bn_rows_forward: bad size");
  if (C % 4) return fail(SGC_EUNSUP, "sgc_bn_rows_forward: needs C % 4 == 0");
  for (int c = 0; c < C; ++c) {
    double m = 0.0, v = 0.0;
    for (int r = 0; r < rows; ++r) m += x[(int64_t)r * C + c];
    m /= rows;
    for (int r = 0; r < rows; ++r) { const double d = x[(int64_t)r * C + c] - m; v += d * d; }
    const double var = v / rows, is = 1.0 / sqrt(var + (double)eps);
    mean_out[c] = (float)m;
    invstd_out[c] = (float)is;
    if (running_mean_or_null) running_mean_or_null[c] = (float)((1.0 - momentum) * running_mean_or_null[c] + momentum * m);
    if (running_var_or_null)
      running_var_or_null[c] = (float)((1.0 - momentum) * running_var_or_null[c] + momentum * (rows > 1 ? v / (rows - 1) : var));
    for (int r = 0; r < rows; ++r)
      y[(int64_t)r * C + c] = (float)((x[(int64_t)r * C + c] - m) * is * weight[c] + bias[c]);
  }
  return SGC_OK;
}

int sgc_bn_rows_backward(const float *x, const float *dy, const float *mean, const float *invstd, const float *weight,
                         float *dx, float *dweight, float *dbias, float *workspace, int64_t workspace_floats, int rows,
                         int C, sgc_stream_t stream) {
  (void)workspace; (void)workspace_floats; (void)stream;
  if (!x || !dy || !mean || !invstd || !weight || !dx || !dweight || !dbias) return fail(SGC_EINVAL, "sgc_bn_rows_backward: null pointer");
  if (rows <= 0 || C <= 0) return fail(SGC_EINVAL, "sgc_bn_rows_backward: bad size");
  if (C % 4) return fail(SGC_EUNSUP, "sgc_bn_rows_backward: needs C % 4 == 0");
  for (int c = 0; c < C; ++c) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < rows; ++r) {
      const double g = dy[(int64_t)r * C + c], xh = ((double)x[(int64_t)r * C + c] - mean[c]) * invstd[c];
      s1 += g; s2 += g * xh;
    }
    dbias[c] = (float)s1;
    dweight[c] = (float)s2;
    for (int r = 0; r < rows; ++r) {
      const double g = dy[(int64_t)r * C + c], xh = ((double)x[(int64_t)r * C + c] - mean[c]) * invstd[c];
      dx[(int64_t)r * C + c] = (float)((double)weight[c] * invstd[c] * (g - s1 / rows - xh * s2 / rows));
    }
  }
  return SGC_OK;
}

/* y = relu?(bn(x) + residual?) and its backward, restated: the plain passes above around the elementwise steps (float, as torch does them) */
int sgc_bn_rows_act_forward(const float *x, const float *weight, const float *bias, float *running_mean_or_null,
                            float *running_var_or_null, float momentum, float eps, const float *residual_or_null, int relu,
                            float *y, float *mean_out, float *invstd_out, float *workspace, int64_t workspace_floats,
                            int rows, int C, sgc_stream_t stream) {
  const int rc = sgc_bn_rows_forward(x, weight, bias, running_mean_or_null, running_var_or_null, momentum, eps, y, mean_out, invstd_out,
                                     workspace, workspace_floats, rows, C, stream);
  if (rc) return rc;
  for (int64_t i = 0; i < (int64_t)rows * C; ++i) {
    float v = y[i];
    if (residual_or_null) v += residual_or_null[i];
    if (relu) v = v > 0.f ? v : 0.f;
    y[i] = v;
  }
  return SGC_OK;
}
int sgc_bn_rows_act_backward(const float *x, const float *dy, const float *y_relu_or_null, const float *mean, const float *invstd,
                             const float *weight, float *dx, float *dweight, float *dbias, float *dresidual_or_null,
                             float *workspace, int64_t workspace_floats, int rows, int C, sgc_stream_t stream) {
  if (!dy) return fail(SGC_EINVAL, "sgc_bn_rows_act_backward: null pointer");
  const int64_t n = (int64_t)rows * C;
  float *g = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; ++i) g[i] = (!y_relu_or_null || y_relu_or_null[i] > 0.f) ? dy[i] : 0.f;
  if (dresidual_or_null) memcpy(dresidual_or_null, g, sizeof(float) * (size_t)n);
  const int rc = sgc_bn_rows_backward(x, g, mean, invstd, weight, dx, dweight, dbias, workspace, workspace_floats, rows, C, stream);
  free(g);
  return rc;
}

int sgc_layer_norm_rows(const float *x, const float *gamma, const float *beta, float eps, float *y,
                        const int32_t *rows_dev_or_null, int rows_cap, int C, sgc_stream_t stream) {
  (void)stream;
  if (!x || !gamma || !beta || !y) return fail(SGC_EINVAL, "null pointer");
  int rows = rows_cap;
  if (rows_dev_or_null && *rows_dev_or_null < rows) rows = *rows_dev_or_null;
  for (int r = 0; r < rows; ++r) {
    const float *xr = x + (int64_t)r * C;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < C; ++c) s += xr[c];
    const double mean = s / C;
    for (int c = 0; c < C; ++c) q += (xr[c] - mean) * (xr[c] - mean);
    const double rstd = 1.0 / sqrt(q / C + (double)eps);
    for (int c = 0; c < C; ++c) y[(int64_t)r * C + c] = (float)((xr[c] - mean) * rstd * gamma[c] + beta[c]);
  }
  return SGC_OK;
}

/* The tail of a VoxFormer level, restated as the composition it replaces (fp32 truth of sgc_level_tail):
 * out_proj on the seen voxels + zero rows elsewhere (TU/deformable_cross_attention.py:826-837), LayerNorm, mmcv FFN with
 * its identity, LayerNorm (TU/encoder.py:311-338). */
int sgc_level_tail_supported(int C, int F) { return C > 0 && F > 0; }

int sgc_level_tail(const float *ctx, const int32_t *row_of, const uint16_t *wo_hi, const uint16_t *wo_lo, const float *bo,
                   const float *ln1_gamma, const float *ln1_beta, float eps1, const uint16_t *w1_hi, const uint16_t *w1_lo,
                   const float *b1, const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, const float *ln2_gamma,
                   const float *ln2_beta, float eps2, float *out, int Nq, int C, int F, sgc_stream_t stream) {
  if (!ctx || !row_of || !wo_hi || !wo_lo || !bo || !ln1_gamma || !ln1_beta || !w1_hi || !w1_lo || !b1 || !w2_hi || !w2_lo ||
      !b2 || !ln2_gamma || !ln2_beta || !out)
    return fail(SGC_EINVAL, "null pointer");
  if (Nq <= 0) return SGC_OK;
  if (C % 32 || F % 32) return fail(SGC_EUNSUP, "fragment-packed weights need C, F multiples of 32");
  /* weights arrive fragment-packed P[N/32][K/16][64][8] (include/sgcdet_amd.h): back to row-major [N][K] */
  uint16_t *wm[6];
  const uint16_t *src[6] = {wo_hi, wo_lo, w1_hi, w1_lo, w2_hi, w2_lo};
  const int nn[6] = {C, C, F, F, C, C}, kk_[6] = {C, C, C, C, F, F};
  int rc = SGC_OK;
  for (int m = 0; m < 6; ++m) {
    wm[m] = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)nn[m] * kk_[m]);
    if (!wm[m]) { rc = fail(SGC_EINVAL, "out of memory"); continue; }
    const int N_ = nn[m], K_ = kk_[m];
    for (int b = 0; b < N_ / 32; ++b)
      for (int ks = 0; ks < K_ / 16; ++ks)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j)
            wm[m][(size_t)(32 * b + (l & 31)) * K_ + 16 * ks + 8 * (l >> 5) + j] = src[m][(((size_t)b * (K_ / 16) + ks) * 64 + l) * 8 + j];
  }
  float *x0 = (float *)calloc((size_t)Nq * C, sizeof(float));
  float *x1 = (float *)malloc(sizeof(float) * (size_t)Nq * C);
  float *h = (float *)malloc(sizeof(float) * (size_t)Nq * F);
  float *x2 = (float *)malloc(sizeof(float) * (size_t)Nq * C);
  float *row = (float *)malloc(sizeof(float) * (size_t)C);
  if (rc == SGC_OK && !(x0 && x1 && h && x2 && row)) rc = fail(SGC_EINVAL, "out of memory");
  for (int q = 0; rc == SGC_OK && q < Nq; ++q) {
    if (row_of[q] < 0) continue;                                   /* the slot scatter leaves unseen voxels at zero */
    rc = sgc_conv3d_cl_bf16x3(ctx + (int64_t)row_of[q] * C, wm[0], wm[1], NULL, bo, NULL, row, 1, 1, 1, C, C, 1, 1, 0, 0, NULL, 0, stream);
    memcpy(x0 + (int64_t)q * C, row, sizeof(float) * (size_t)C);
  }
  if (rc == SGC_OK) rc = sgc_layer_norm_rows(x0, ln1_gamma, ln1_beta, eps1, x1, NULL, Nq, C, stream);
  if (rc == SGC_OK) rc = sgc_conv3d_cl_bf16x3(x1, wm[2], wm[3], NULL, b1, NULL, h, Nq, 1, 1, C, F, 1, 1, 0, 2, NULL, 0, stream);
  if (rc == SGC_OK) rc = sgc_conv3d_cl_bf16x3(h, wm[4], wm[5], NULL, b2, x1, x2, Nq, 1, 1, F, C, 1, 1, 0, 0, NULL, 0, stream);
  if (rc == SGC_OK) rc = sgc_layer_norm_rows(x2, ln2_gamma, ln2_beta, eps2, out, NULL, Nq, C, stream);
  for (int m = 0; m < 6; ++m) free(wm[m]);
  free(x0); free(x1); free(h); free(x2); free(row);
  return rc;
}

/* weight layout passes of the training step (include/sgcdet_amd.h) */
int sgc_pack_conv_weight(const float *w, uint16_t *w_hi, uint16_t *w_lo, int A, int B, int T, int R, int C, int transpose,
                         int flip, sgc_stream_t stream) {
  (void)stream;
  if (!w || !w_hi || !w_lo) return fail(SGC_EINVAL, "null pointer");
  if (A <= 0 || B <= 0 || T <= 0 || R < (transpose ? B : A) || C < (transpose ? A : B)) return fail(SGC_EINVAL, "bad size");
  for (int t = 0; t < T; ++t)
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) {
        const int a = transpose ? c : r, b = transpose ? r : c;
        const float v = (a < A && b < B) ? w[((int64_t)a * B + b) * T + (flip ? T - 1 - t : t)] : 0.f;
        const uint16_t hb = f32_to_bf16_rne(v);
        const int64_t o = ((int64_t)t * R + r) * C + c;
        w_hi[o] = hb;
        w_lo[o] = f32_to_bf16_rne(v - bf16_to_f32(hb));
      }
  return SGC_OK;
}

/* the batch form: the same map per item (include/sgcdet_amd.h: sgc_pack_item); padding is left as the caller zeroed it */
int sgc_pack_conv_weight_blocks(int A, int B, int T, int transpose) {
  if (A <= 0 || B <= 0 || T <= 0) return 0;
  return transpose ? ((B + 7) / 8) * ((A + 31) / 32) : ((B + 31) / 32) * ((A + 7) / 8);
}
int sgc_pack_conv_weight_batch(const void *items, int n_items, int total_blocks, int max_T, sgc_stream_t stream) {
  (void)stream; (void)max_T;
  if (!items || n_items <= 0 || total_blocks <= 0) return fail(SGC_EINVAL, "bad arguments");
  const sgc_pack_item *it = (const sgc_pack_item *)items;
  for (int i = 0; i < n_items; ++i) {
    const int A = it[i].A, B = it[i].B, T = it[i].T, R = it[i].R, C = it[i].C, tr = it[i].transpose, fl = it[i].flip;
    if (!it[i].w || !it[i].hi || !it[i].lo || A <= 0 || B <= 0 || T <= 0 || R < (tr ? B : A) || C < (tr ? A : B)) return fail(SGC_EINVAL, "bad item");
    for (int t = 0; t < T; ++t)
      for (int a = 0; a < A; ++a)
        for (int b = 0; b < B; ++b) {
          const int r = tr ? b : a, c = tr ? a : b;
          const float v = it[i].w[((int64_t)a * B + b) * T + (fl ? T - 1 - t : t)];
          const uint16_t hb = f32_to_bf16_rne(v);
          const int64_t o = ((int64_t)t * R + r) * C + c;
          it[i].hi[o] = hb;
          it[i].lo[o] = f32_to_bf16_rne(v - bf16_to_f32(hb));
        }
  }
  return SGC_OK;
}

int sgc_unpack_conv_wgrad(const float *dw_trc, float *dw, int A, int B, int T, int R, int C, int transpose, int flip,
                          sgc_stream_t stream) {
  (void)stream;
  if (!dw_trc || !dw) return fail(SGC_EINVAL, "null pointer");
  if (A <= 0 || B <= 0 || T <= 0 || R < (transpose ? B : A) || C < (transpose ? A : B)) return fail(SGC_EINVAL, "bad size");
  for (int a = 0; a < A; ++a)
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < T; ++t) {
        const int r = transpose ? b : a, c = transpose ? a : b;
        dw[((int64_t)a * B + b) * T + (flip ? T - 1 - t : t)] = dw_trc[((int64_t)t * R + r) * C + c];
      }
  return SGC_OK;
}

/* ---- coarse-to-fine glue (AdaptiveSparseHead.py:64-82), torch upsample_trilinear3d index rule ---- */
int sgc_upsample2x_occ(const float *vol, const float *w_or_null, const float *b_or_null, float *up,
                       float *occ_or_null, int ix, int iy, int iz, int C, sgc_stream_t stream) {
  (void)stream;
  if (!vol || !up) return fail(SGC_EINVAL, "null pointer");
  const int ox = 2 * ix, oy = 2 * iy, oz = 2 * iz;
  const int n[3] = {ix, iy, iz};
#pragma omp parallel for collapse(2) schedule(static)
  for (int x = 0; x < ox; ++x)
    for (int y = 0; y < oy; ++y)
      for (int z = 0; z < oz; ++z) {
        const int dst[3] = {x, y, z};
        int i0[3], i1[3];
        float l0[3], l1[3];
        for (int a = 0; a < 3; ++a) {
          float t = 0.5f * ((float)dst[a] + 0.5f) - 0.5f;
          if (t < 0.f) t = 0.f;
          i0[a] = (int)t;
          i1[a] = i0[a] + (i0[a] < n[a] - 1 ? 1 : 0);
          l1[a] = t - (float)i0[a];
          l0[a] = 1.f - l1[a];
        }
        const int64_t v = ((int64_t)x * oy + y) * oz + z;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
#define V(a, b, d) vol[((((int64_t)(a)) * iy + (b)) * iz + (d)) * C + c]
          const float r = l0[0] * (l0[1] * (l0[2] * V(i0[0], i0[1], i0[2]) + l1[2] * V(i0[0], i0[1], i1[2])) +
                                   l1[1] * (l0[2] * V(i0[0], i1[1], i0[2]) + l1[2] * V(i0[0], i1[1], i1[2]))) +
                          l1[0] * (l0[1] * (l0[2] * V(i1[0], i0[1], i0[2]) + l1[2] * V(i1[0], i0[1], i1[2])) +
                                   l1[1] * (l0[2] * V(i1[0], i1[1], i0[2]) + l1[2] * V(i1[0], i1[1], i1[2])));
#undef V
          up[v * C + c] = r;
          if (w_or_null) dot += r * w_or_null[c];
        }
        if (w_or_null) occ_or_null[v] = 1.f / (1.f + expf(-(dot + b_or_null[0])));
      }
  return SGC_OK;
}

int sgc_scatter_add_rows(const float *rows, const int64_t *idx, float *vol, int n, int C, sgc_stream_t stream) {
  (void)stream;
  if (!rows || !idx || !vol) return fail(SGC_EINVAL, "null pointer");
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < C; ++c) vol[idx[i] * C + c] += rows[(int64_t)i * C + c];
  return SGC_OK;
}


/* ---- 7b. adjoint of the x2 trilinear upsample (AdaptiveSparseHead.py:64-69; torch's upsample_trilinear3d index
 * rule, align_corners = False: src = max(0.5 (o + 0.5) - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, n - 1)), written
 * as the scatter torch's backward performs; pinned by torch autograd in tests/test_oracle_conv.py ------------- */
static void up2_src(int o, int n, int *i0, int *i1, float *l0, float *l1) {
  float src = 0.5f * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  *i0 = (int)src;
  *i1 = *i0 + (*i0 < n - 1 ? 1 : 0);
  *l1 = src - (float)*i0;
  *l0 = 1.f - *l1;
}
int sgc_upsample2x_backward(const float *grad_out, float *grad_in, int C, int X, int Y, int Z, sgc_stream_t stream) {
  (void)stream;
  if (C <= 0 || X <= 0 || Y <= 0 || Z <= 0) return SGC_OK;
  if (!grad_out || !grad_in) return fail(SGC_EINVAL, "null pointer");
  const int64_t vin = (int64_t)X * Y * Z, vout = vin * 8;
  memset(grad_in, 0, sizeof(float) * (size_t)(C * vin));
  for (int c = 0; c < C; ++c)
    for (int ox = 0; ox < 2 * X; ++ox) {
      int x0, x1; float a0, a1;
      up2_src(ox, X, &x0, &x1, &a0, &a1);
      for (int oy = 0; oy < 2 * Y; ++oy) {
        int y0, y1; float b0, b1;
        up2_src(oy, Y, &y0, &y1, &b0, &b1);
        for (int oz = 0; oz < 2 * Z; ++oz) {
          int z0, z1; float c0, c1;
          up2_src(oz, Z, &z0, &z1, &c0, &c1);
          const float g = grad_out[c * vout + ((int64_t)ox * 2 * Y + oy) * 2 * Z + oz];
          float *gi = grad_in + c * vin;
          gi[((int64_t)x0 * Y + y0) * Z + z0] += a0 * b0 * c0 * g;
          gi[((int64_t)x0 * Y + y0) * Z + z1] += a0 * b0 * c1 * g;
          gi[((int64_t)x0 * Y + y1) * Z + z0] += a0 * b1 * c0 * g;
          gi[((int64_t)x0 * Y + y1) * Z + z1] += a0 * b1 * c1 * g;
          gi[((int64_t)x1 * Y + y0) * Z + z0] += a1 * b0 * c0 * g;
          gi[((int64_t)x1 * Y + y0) * Z + z1] += a1 * b0 * c1 * g;
          gi[((int64_t)x1 * Y + y1) * Z + z0] += a1 * b1 * c0 * g;
          gi[((int64_t)x1 * Y + y1) * Z + z1] += a1 * b1 * c1 * g;
        }
      }
    }
  return SGC_OK;
}


/* ---- 8. post-processing: mmdet3d aligned_3d_nms (box3d_nms.py:131-178), restated loop for loop -------------- */
int sgc_aligned_nms3d(const float *boxes, const int64_t *order, const int64_t *labels, float iou_thr,
                      int64_t *keep, int32_t *n_keep, uint64_t *workspace, int n, sgc_stream_t stream) {
  (void)stream; (void)workspace;
  if (!n_keep) return fail(SGC_EINVAL, "null pointer");
  *n_keep = 0;
  if (n <= 0) return SGC_OK;
  if (!boxes || !order || !labels || !keep) return fail(SGC_EINVAL, "null pointer");
  int64_t *sorted = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);      /* `score_sorted`, ascending */
  if (!sorted) return fail(SGC_EINVAL, "out of memory");
  memcpy(sorted, order, sizeof(int64_t) * (size_t)n);
  int len = n, cnt = 0;
  while (len != 0) {
    const int64_t i = sorted[len - 1];
    keep[cnt++] = i;                                                     /* pick.append(i) */
    const float *a = boxes + i * 6;
    const volatile float area_i = (a[3] - a[0]) * (a[4] - a[1]) * (a[5] - a[2]);
    int out = 0;
    for (int t = 0; t < len - 1; ++t) {                                  /* score_sorted[:last - 1] */
      const int64_t j = sorted[t];
      const float *b = boxes + j * 6;
      const volatile float area_j = (b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]);
      const float xx1 = a[0] > b[0] ? a[0] : b[0], yy1 = a[1] > b[1] ? a[1] : b[1], zz1 = a[2] > b[2] ? a[2] : b[2];
      const float xx2 = a[3] < b[3] ? a[3] : b[3], yy2 = a[4] < b[4] ? a[4] : b[4], zz2 = a[5] < b[5] ? a[5] : b[5];
      float il = xx2 - xx1, iw = yy2 - yy1, ih = zz2 - zz1;
      il = il > 0.f ? il : 0.f; iw = iw > 0.f ? iw : 0.f; ih = ih > 0.f ? ih : 0.f;
      const volatile float inter = il * iw * ih;
      const volatile float denom = area_i + area_j - inter;
      volatile float iou = inter / denom;
      iou = iou * (labels[i] == labels[j] ? 1.f : 0.f);
      if (iou <= iou_thr) sorted[out++] = j;                             /* nonzero(iou <= thresh) */
    }
    len = out;
  }
  free(sorted);
  *n_keep = cnt;
  return SGC_OK;
}


/* ---- 8b. rotated BEV NMS: mmdet3d nms_bev (box3d_nms.py:231-268) -> mmcv.ops.nms_rotated -------------------
 * The IoU follows the reference's own copy of mmcv's header, CS/common/box_iou_rotated_utils.hpp (get_rotated_vertices
 * :55-72, get_intersection_points :74-156, convex_hull_graham :158-261 [its __CUDACC__ branch: the O(n^2) exchange
 * sort the GPU kernels run], polygon_area :263-275, single_box_iou_rotated :307-341) in T = float as the CUDA kernel
 * instantiates it; PINNED to that header built with g++ (oracle/_ref, tests/golden/box_iou_rotated.npz, <= 1e-6).
 * The suppression itself restates mmcv-full 1.5.3's nms_rotated_cuda.cuh (mask = IoU(row, col) > thr for col after
 * row in descending score; host sweep) -- that file is a pip dependency (docs/install.md:6), not in the tree.   */
typedef struct { float x, y; } rpt_t;
static inline float rcross(rpt_t a, rpt_t b) { return a.x * b.y - b.x * a.y; }
static inline float rdot(rpt_t a, rpt_t b) { return a.x * b.x + a.y * b.y; }
static inline rpt_t rsub(rpt_t a, rpt_t b) { rpt_t r = {a.x - b.x, a.y - b.y}; return r; }

static void rot_vertices(float xc, float yc, float w, float h, float a, rpt_t *pts) {
  const double theta = (double)a;
  const float cos2 = (float)cos(theta) * 0.5f, sin2 = (float)sin(theta) * 0.5f;
  pts[0].x = xc - sin2 * h - cos2 * w;
  pts[0].y = yc + cos2 * h - sin2 * w;
  pts[1].x = xc + sin2 * h - cos2 * w;
  pts[1].y = yc - cos2 * h - sin2 * w;
  pts[2].x = 2 * xc - pts[0].x;
  pts[2].y = 2 * yc - pts[0].y;
  pts[3].x = 2 * xc - pts[1].x;
  pts[3].y = 2 * yc - pts[1].y;
}

static int rot_intersections(const rpt_t *p1, const rpt_t *p2, rpt_t *out) {
  rpt_t v1[4], v2[4];
  for (int i = 0; i < 4; ++i) { v1[i] = rsub(p1[(i + 1) % 4], p1[i]); v2[i] = rsub(p2[(i + 1) % 4], p2[i]); }
  int num = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const float det = rcross(v2[j], v1[i]);
      if (fabs((double)det) <= 1e-14) continue;                  /* parallel edges */
      const rpt_t v12 = rsub(p2[j], p1[i]);
      const float t1 = rcross(v2[j], v12) / det, t2 = rcross(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        out[num].x = p1[i].x + v1[i].x * t1;
        out[num].y = p1[i].y + v1[i].y * t1;
        ++num;
      }
    }
  {                                                              /* vertices of rect 1 inside rect 2 */
    const rpt_t AB = v2[0], DA = v2[3];
    const float ABdotAB = rdot(AB, AB), ADdotAD = rdot(DA, DA);
    for (int i = 0; i < 4; ++i) {
      const rpt_t AP = rsub(p1[i], p2[0]);
      const float APdotAB = rdot(AP, AB), APdotAD = -rdot(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) out[num++] = p1[i];
    }
  }
  {                                                              /* and the reverse */
    const rpt_t AB = v1[0], DA = v1[3];
    const float ABdotAB = rdot(AB, AB), ADdotAD = rdot(DA, DA);
    for (int i = 0; i < 4; ++i) {
      const rpt_t AP = rsub(p2[i], p1[0]);
      const float APdotAB = rdot(AP, AB), APdotAD = -rdot(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) out[num++] = p2[i];
    }
  }
  return num;
}

static int rot_hull(const rpt_t *p, int n, rpt_t *q) {            /* convex_hull_graham(..., shift_to_zero = true) */
  int t = 0;
  for (int i = 1; i < n; ++i)
    if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
  const rpt_t start = p[t];
  for (int i = 0; i < n; ++i) q[i] = rsub(p[i], start);
  { const rpt_t tmp = q[0]; q[0] = q[t]; q[t] = tmp; }
  float dist[24];
  for (int i = 0; i < n; ++i) dist[i] = rdot(q[i], q[i]);
  for (int i = 1; i < n - 1; ++i)                                /* device branch: exchange sort by polar angle */
    for (int j = i + 1; j < n; ++j) {
      const float cp = rcross(q[i], q[j]);
      if (((double)cp < -1e-6) || (fabs((double)cp) < 1e-6 && dist[i] > dist[j])) {
        const rpt_t qt = q[i]; q[i] = q[j]; q[j] = qt;
        const float dt = dist[i]; dist[i] = dist[j]; dist[j] = dt;
      }
    }
  int k;
  for (k = 1; k < n; ++k)
    if ((double)dist[k] > 1e-8) break;
  if (k == n) return 1;                                          /* all points coincide */
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < n; ++i) {
    while (m > 1 && rcross(rsub(q[i], q[m - 2]), rsub(q[m - 1], q[m - 2])) >= 0) --m;
    q[m++] = q[i];
  }
  return m;
}

static float rot_iou(const float *b1, const float *b2) {          /* single_box_iou_rotated(box1, box2, mode 0); xywhr */
  const float sx = (b1[0] + b2[0]) * 0.5f, sy = (b1[1] + b2[1]) * 0.5f;   /* "/ 2.0" in double == exact halving */
  const float x1 = b1[0] - sx, y1 = b1[1] - sy, x2 = b2[0] - sx, y2 = b2[1] - sy;
  const float area1 = b1[2] * b1[3], area2 = b2[2] * b2[3];
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;
  rpt_t p1[4], p2[4], ip[24], hull[24];
  rot_vertices(x1, y1, b1[2], b1[3], b1[4], p1);
  rot_vertices(x2, y2, b2[2], b2[3], b2[4], p2);
  const int num = rot_intersections(p1, p2, ip);
  float inter = 0.f;
  if (num > 2) {
    const int m = rot_hull(ip, num, hull);
    if (m > 2) {
      float area = 0.f;
      for (int i = 1; i < m - 1; ++i) area += fabsf(rcross(rsub(hull[i], hull[0]), rsub(hull[i + 1], hull[0])));
      inter = area * 0.5f;
    }
  }
  return inter / (area1 + area2 - inter);
}

/* mmcv.ops.box_iou_rotated(a, b, mode='iou', aligned=False): boxes (xc, yc, w, h, angle in radians) */
int sgc_box_iou_rotated(const float *a, const float *b, float *iou, int n, int m, sgc_stream_t stream) {
  (void)stream;
  if (n <= 0 || m <= 0) return SGC_OK;
  if (!a || !b || !iou) return fail(SGC_EINVAL, "null pointer");
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) iou[(int64_t)i * m + j] = rot_iou(a + (int64_t)i * 5, b + (int64_t)j * 5);
  return SGC_OK;
}

/* per class c: nms_bev(boxes[order[c][:counts[c]]], thr) -- xyxyr -> xywhr (box3d_nms.py:256-262), then the
 * nms_rotated mask + sweep over the already sorted boxes */
int sgc_nms_rotated_bev(const float *boxes, const int64_t *order, const int32_t *counts, float iou_thr,
                        int64_t *keep, int32_t *n_keep, uint64_t *workspace, int K, int C, sgc_stream_t stream) {
  (void)stream; (void)workspace;
  if (C <= 0) return SGC_OK;
  if (!counts || !n_keep) return fail(SGC_EINVAL, "null pointer");
  if (K > 4096) return fail(SGC_EUNSUP, "at most 4096 candidates");
  for (int c = 0; c < C; ++c) {
    const int n = counts[c] < K ? counts[c] : K;
    n_keep[c] = 0;
    if (n <= 0) continue;
    if (!boxes || !order || !keep) return fail(SGC_EINVAL, "null pointer");
    float *xywhr = (float *)malloc(sizeof(float) * 5 * (size_t)n);
    unsigned char *removed = (unsigned char *)calloc((size_t)n, 1);
    if (!xywhr || !removed) { free(xywhr); free(removed); return fail(SGC_EINVAL, "out of memory"); }
    const int64_t *ord = order + (int64_t)c * K;
    for (int p = 0; p < n; ++p) {
      const float *b = boxes + ord[p] * 5;
      xywhr[p * 5 + 0] = (b[0] + b[2]) / 2;
      xywhr[p * 5 + 1] = (b[1] + b[3]) / 2;
      xywhr[p * 5 + 2] = b[2] - b[0];
      xywhr[p * 5 + 3] = b[3] - b[1];
      xywhr[p * 5 + 4] = b[4];
    }
    int cnt = 0;
    for (int p = 0; p < n; ++p) {
      if (removed[p]) continue;
      keep[(int64_t)c * K + cnt++] = ord[p];
      for (int q = p + 1; q < n; ++q)
        if (!removed[q] && rot_iou(xywhr + p * 5, xywhr + q * 5) > iou_thr) removed[q] = 1;
    }
    n_keep[c] = cnt;
    free(xywhr); free(removed);
  }
  return SGC_OK;
}


/* ---- 8c. head target assignment: ImVoxelHeadV2.get_targets (imvoxel_head_v2.py:361-435 axis-aligned,
 * :485-561 rotated) restated with loops over the reference's dense [n_points, n_boxes] tensors -------------- */
static void target_faces(const float *pt, const float *b, int rotated, float *t /*[6]*/) {
  if (!rotated) {                                             /* :379-384 */
    t[0] = pt[0] - b[0] + b[3] / 2; t[1] = b[0] + b[3] / 2 - pt[0];
    t[2] = pt[1] - b[1] + b[4] / 2; t[3] = b[1] + b[4] / 2 - pt[1];
    t[4] = pt[2] - b[2] + b[5] / 2; t[5] = b[2] + b[5] / 2 - pt[2];
    return;
  }
  /* :503-515: shift rotated by -yaw about z (rotation_3d_in_axis: x' = x cos + y (-sin), y' = x sin + y cos) */
  const float sx = pt[0] - b[0], sy = pt[1] - b[1], sz = pt[2] - b[2];
  const float a = -b[6], sn = sinf(a), cs = cosf(a);
  const float rx = sx * cs + sy * (-sn), ry = sx * sn + sy * cs;
  const float cx = b[0] + rx, cy = b[1] + ry, cz = b[2] + sz;
  t[0] = cx - b[0] + b[3] / 2; t[1] = b[0] + b[3] / 2 - cx;
  t[2] = cy - b[1] + b[4] / 2; t[3] = b[1] + b[4] / 2 - cy;
  t[4] = cz - b[2] + b[5] / 2; t[5] = b[2] + b[5] / 2 - cz;
}
static float target_centerness(const float *t) {               /* compute_centerness, :334-343 */
  const float xm = t[0] < t[1] ? t[0] : t[1], xM = t[0] > t[1] ? t[0] : t[1];
  const float ym = t[2] < t[3] ? t[2] : t[3], yM = t[2] > t[3] ? t[2] : t[3];
  const float zm = t[4] < t[5] ? t[4] : t[5], zM = t[4] > t[5] ? t[4] : t[5];
  return sqrtf(xm / xM * ym / yM * zm / zM);
}
static int cmp_desc_f32(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return x < y ? 1 : (x > y ? -1 : 0);
}
int sgc_assign_targets(const float *points, const int32_t *scales, const float *boxes, const int64_t *gt_labels,
                       int rotated, int n_scales, int limit, int centerness_topk, float *centerness_t, float *bbox_t,
                       int64_t *labels, uint8_t *geo_occ, int32_t *workspace, int n_points, int n_boxes,
                       sgc_stream_t stream) {
  (void)stream; (void)workspace;
  if (n_points <= 0) return SGC_OK;
  if (!points || !scales || !centerness_t || !bbox_t || !labels || !geo_occ) return fail(SGC_EINVAL, "null pointer");
  if (n_boxes <= 0 || !boxes || !gt_labels) return fail(SGC_EINVAL, "get_targets needs at least one box");
  if (centerness_topk + 1 > n_points) return fail(SGC_EINVAL, "centerness_topk + 1 > n_points");
  const float float_max = 1e8f;
  const int bw = rotated ? 7 : 6;
  int *best = (int *)malloc(sizeof(int) * (size_t)n_boxes);
  float *top_c = (float *)malloc(sizeof(float) * (size_t)n_boxes);
  float *col = (float *)malloc(sizeof(float) * (size_t)n_points);
  if (!best || !top_c || !col) { free(best); free(top_c); free(col); return fail(SGC_EINVAL, "out of memory"); }
  for (int j = 0; j < n_boxes; ++j) {
    const float *b = boxes + (int64_t)j * 7;
    /* condition 2 (:390-407): positive points per scale, best scale of the box */
    int lower_index = -1, all_upper = 1;
    int best_val = 0;
    for (int s = 0; s < n_scales; ++s) {
      int cnt = 0;
      for (int i = 0; i < n_points; ++i) {
        if (scales[i] != s) continue;
        float t[6];
        target_faces(points + (int64_t)i * 3, b, rotated, t);
        float mn = t[0];
        for (int k = 1; k < 6; ++k) mn = t[k] < mn ? t[k] : mn;
        cnt += mn > 0;
      }
      const int lower = cnt < limit;
      if (lower) all_upper = 0;
      const int v = lower * (n_scales - s);                   /* argmax(lower_limit_mask * extra): first maximum */
      if (s == 0 || v > best_val) { best_val = v; lower_index = s; }
    }
    lower_index -= 1;
    if (lower_index < 0) lower_index = 0;
    best[j] = all_upper ? n_scales - 1 : lower_index;
    /* condition 3 (:413-417): (centerness_topk + 1)-th largest masked centerness of the box */
    for (int i = 0; i < n_points; ++i) {
      float t[6];
      target_faces(points + (int64_t)i * 3, b, rotated, t);
      float mn = t[0];
      for (int k = 1; k < 6; ++k) mn = t[k] < mn ? t[k] : mn;
      col[i] = (mn > 0 && scales[i] == best[j]) ? target_centerness(t) : -1.f;
    }
    qsort(col, (size_t)n_points, sizeof(float), cmp_desc_f32);
    top_c[j] = col[centerness_topk];
  }
  for (int i = 0; i < n_points; ++i) {
    const float *pt = points + (int64_t)i * 3;
    float min_area = 0.f;
    int arg = 0, any_inside = 0;
    for (int j = 0; j < n_boxes; ++j) {
      const float *b = boxes + (int64_t)j * 7;
      float t[6];
      target_faces(pt, b, rotated, t);
      float mn = t[0];
      for (int k = 1; k < 6; ++k) mn = t[k] < mn ? t[k] : mn;
      const int inside = mn > 0;
      any_inside |= inside;
      float vol = b[3] * b[4] * b[5];
      if (!inside) vol = float_max;
      if (scales[i] != best[j]) vol = float_max;
      const float c = (inside && scales[i] == best[j]) ? target_centerness(t) : -1.f;
      if (!(c > top_c[j])) vol = float_max;
      if (j == 0 || vol < min_area) { min_area = vol; arg = j; }       /* volumes.min(dim=1): first minimum */
    }
    const float *b = boxes + (int64_t)arg * 7;
    float t[6];
    target_faces(pt, b, rotated, t);
    labels[i] = min_area == float_max ? -1 : gt_labels[arg];
    centerness_t[i] = target_centerness(t);
    geo_occ[i] = (uint8_t)any_inside;
    float *o = bbox_t + (int64_t)i * bw;
    if (rotated) {
      for (int k = 0; k < 7; ++k) o[k] = b[k];                /* gt_bboxes[range(n_points), min_area_inds], :561 */
    } else {                                                  /* _bbox_pred_to_bbox(points, bbox_targets), :456-464 */
      o[0] = pt[0] - t[0]; o[1] = pt[1] - t[2]; o[2] = pt[2] - t[4];
      o[3] = pt[0] + t[1]; o[4] = pt[1] + t[3]; o[5] = pt[2] + t[5];
    }
  }
  free(best); free(top_c); free(col);
  return SGC_OK;
}


/* ---- 9. plane-sweep matching cost (depth_est_fusion.py:87-126 homo_warping, :233-240 cost volume) ----------- */
int sgc_plane_sweep_corr(const float *feat, const int32_t *nbr, const float *rt, const float *depth,
                         float *corr, int N, int K, int H, int W, int C, int D, sgc_stream_t stream) {
  (void)stream;
  if (!feat || !nbr || !rt || !depth || !corr) return fail(SGC_EINVAL, "null pointer");
  const int64_t HW = (int64_t)H * W;
  const float inv_sqrt_c = 1.0f / sqrtf((float)C);
  const float half_w = (float)(W - 1) / 2.0f, half_h = (float)(H - 1) / 2.0f;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int64_t pix = 0; pix < HW; ++pix) {
      const float fx = (float)(pix % W), fy = (float)(pix / W);
      const float *own = feat + (n * HW + pix) * C;
      for (int d = 0; d < D; ++d) {
        float total = 0.f;
        for (int k = 0; k < K; ++k) {
          const float *m = rt + ((int64_t)n * K + k) * 12;
          const float rx = m[0] * fx + m[1] * fy + m[2], ry = m[4] * fx + m[5] * fy + m[6], rz = m[8] * fx + m[9] * fy + m[10];
          const float px = rx * depth[d] + m[3], py = ry * depth[d] + m[7], pz = rz * depth[d] + m[11];
          const float gx = (px / pz) / half_w - 1.0f, gy = (py / pz) / half_h - 1.0f;
          const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
          float dot = 0.f;
          if (ix > -1.0f && iy > -1.0f && ix < (float)W && iy < (float)H) {
            const float x0f = floorf(ix), y0f = floorf(iy);
            const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
            const float lx = ix - x0f, ly = iy - y0f, hx = 1.0f - lx, hy = 1.0f - ly;
            const float *src = feat + (int64_t)nbr[n * K + k] * HW * C;
            for (int c = 0; c < C; ++c) {
              float s = 0.f;
              if (y0 >= 0 && x0 >= 0) s += src[((int64_t)y0 * W + x0) * C + c] * (hx * hy);
              if (y0 >= 0 && x1 <= W - 1) s += src[((int64_t)y0 * W + x1) * C + c] * (lx * hy);
              if (y1 <= H - 1 && x0 >= 0) s += src[((int64_t)y1 * W + x0) * C + c] * (hx * ly);
              if (y1 <= H - 1 && x1 <= W - 1) s += src[((int64_t)y1 * W + x1) * C + c] * (lx * ly);
              dot += s * own[c];
            }
          }
          total += dot * inv_sqrt_c;
        }
        corr[((int64_t)n * D + d) * HW + pix] = total / (float)K;
      }
    }
  return SGC_OK;
}
