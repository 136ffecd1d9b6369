// Target assignment of the FCOS3D-style head == ImVoxelHeadV2.get_targets of the reference
// (mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py:361-435 ScanNetImVoxelHeadV2, :485-561
// SunRgbdImVoxelHeadV2; compute_centerness :334-343).  The reference materialises [n_points, n_boxes(, 6)] tensors
// (n_points = 29 200 at config 2) through ~40 torch launches; here nothing of that size exists:
//   1. tgt_count_kernel     (point, box) -> inside test, per-scale counts per box            (:386-393)
//   2. tgt_best_scale_kernel  one thread per box: smallest scale with >= limit inside points  (:394-407)
//   3. tgt_topc_kernel      one workgroup per box: exact (centerness_topk + 1)-th largest masked centerness by a
//                           4-pass radix select over the float bits                          (:413-417)
//   4. tgt_assign_kernel    one thread per point: minimal-volume box among those passing the three conditions,
//                           label / box target / centerness target / occupancy               (:419-435)
// Face distances and centerness are re-evaluated where needed (12 flops) instead of being stored.  All arithmetic
// in the reference's operation order, no FMA contraction: the axis-aligned head is bit-exact against the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgcdet_amd.h"
#include "common.hpp"

#pragma clang fp contract(off)

namespace sgc {

struct TBox { float b[7]; float sn, cs; };     // gravity centre, dims, yaw; sin / cos of -yaw (rotated heads)

__device__ __forceinline__ TBox load_tbox(const float *boxes, int j, int rotated) {
  TBox t;
#pragma unroll
  for (int k = 0; k < 7; ++k) t.b[k] = boxes[(int64_t)j * 7 + k];
  t.sn = 0.f; t.cs = 1.f;
  if (rotated) { const float a = -t.b[6]; t.sn = sinf(a); t.cs = cosf(a); }
  return t;
}

__device__ __forceinline__ void faces(const float *pt, const TBox &B, int rotated, float *t) {
  const float *b = B.b;
  float cx = pt[0], cy = pt[1], cz = pt[2];
  if (rotated) {       // shift rotated by -yaw about z, then re-centred on the box (:503-510)
    const float sx = pt[0] - b[0], sy = pt[1] - b[1], sz = pt[2] - b[2];
    const float rx = sx * B.cs + sy * (-B.sn), ry = sx * B.sn + sy * B.cs;
    cx = b[0] + rx; cy = b[1] + ry; cz = b[2] + sz;
  }
  t[0] = cx - b[0] + b[3] / 2; t[1] = b[0] + b[3] / 2 - cx;
  t[2] = cy - b[1] + b[4] / 2; t[3] = b[1] + b[4] / 2 - cy;
  t[4] = cz - b[2] + b[5] / 2; t[5] = b[2] + b[5] / 2 - cz;
}

__device__ __forceinline__ bool inside_box(const float *t) {
  return fminf(fminf(fminf(t[0], t[1]), fminf(t[2], t[3])), fminf(t[4], t[5])) > 0.f;
}

__device__ __forceinline__ float centerness_of(const float *t) {
  const float xm = fminf(t[0], t[1]), xM = fmaxf(t[0], t[1]);
  const float ym = fminf(t[2], t[3]), yM = fmaxf(t[2], t[3]);
  const float zm = fminf(t[4], t[5]), zM = fmaxf(t[4], t[5]);
  return sqrtf(xm / xM * ym / yM * zm / zM);
}

__global__ void tgt_zero_kernel(int32_t *ws, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ws[i] = 0;
}

__global__ __launch_bounds__(256) void tgt_count_kernel(const float *__restrict__ points, const int32_t *__restrict__ scales,
                                                        const float *__restrict__ boxes, int rotated, int n_scales,
                                                        int32_t *__restrict__ counts, int n_points, int n_boxes) {
  __shared__ int cnt[16];
  const int j = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
  __syncthreads();
  const TBox B = load_tbox(boxes, j, rotated);
  if (i < n_points) {
    float t[6];
    faces(points + (int64_t)i * 3, B, rotated, t);
    const int s = scales[i];
    if (inside_box(t) && s >= 0 && s < n_scales) atomicAdd(&cnt[s], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < n_scales && cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x * n_boxes + j], cnt[threadIdx.x]);
}

__global__ void tgt_best_scale_kernel(const int32_t *__restrict__ counts, int32_t *__restrict__ best, int n_scales,
                                      int limit, int n_boxes) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_boxes) return;
  int lower_index = 0, best_val = 0, all_upper = 1;
  for (int s = 0; s < n_scales; ++s) {
    const int lower = counts[s * n_boxes + j] < limit;
    if (lower) all_upper = 0;
    const int v = lower * (n_scales - s);          // argmax(lower_limit_mask * extra): first maximum
    if (s == 0 || v > best_val) { best_val = v; lower_index = s; }
  }
  lower_index = max(lower_index - 1, 0);
  best[j] = all_upper ? n_scales - 1 : lower_index;
}

// key of a masked centerness: 0 for the reference's -1 fill, bits + 1 for a centerness >= 0 (monotonic)
__device__ __forceinline__ unsigned key_of(const float *pt, int scale, const TBox &B, int rotated, int best) {
  float t[6];
  faces(pt, B, rotated, t);
  if (!(inside_box(t) && scale == best)) return 0u;
  return __float_as_uint(centerness_of(t)) + 1u;
}

__global__ __launch_bounds__(256) void tgt_topc_kernel(const float *__restrict__ points, const int32_t *__restrict__ scales,
                                                       const float *__restrict__ boxes, int rotated,
                                                       const int32_t *__restrict__ best, float *__restrict__ top_c, int kth,
                                                       int n_points) {
  __shared__ int hist[256];
  __shared__ unsigned sel_prefix;
  __shared__ int sel_k;
  const int j = blockIdx.x, tid = threadIdx.x;
  const TBox B = load_tbox(boxes, j, rotated);
  const int bs = best[j];
  if (tid == 0) { sel_prefix = 0u; sel_k = kth; }           // kth: 1-based rank from the top
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = sel_prefix;
    const unsigned pmask = pass == 0 ? 0u : 0xffffffffu << (shift + 8);
    for (int i = tid; i < n_points; i += 256) {
      const unsigned key = key_of(points + (int64_t)i * 3, scales[i], B, rotated, bs);
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int k = sel_k, d = 255;
      for (; d > 0; --d) {
        if (hist[d] >= k) break;
        k -= hist[d];
      }
      sel_prefix = prefix | ((unsigned)d << shift);
      sel_k = k;
    }
    __syncthreads();
  }
  if (tid == 0) top_c[j] = sel_prefix == 0u ? -1.f : __uint_as_float(sel_prefix - 1u);
}

__global__ __launch_bounds__(256) void tgt_assign_kernel(const float *__restrict__ points, const int32_t *__restrict__ scales,
                                                         const float *__restrict__ boxes, const int64_t *__restrict__ gt_labels,
                                                         int rotated, const int32_t *__restrict__ best,
                                                         const float *__restrict__ top_c, float *__restrict__ centerness_t,
                                                         float *__restrict__ bbox_t, int64_t *__restrict__ labels,
                                                         uint8_t *__restrict__ geo_occ, int n_points, int n_boxes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_points) return;
  const float pt[3] = {points[(int64_t)i * 3], points[(int64_t)i * 3 + 1], points[(int64_t)i * 3 + 2]};
  const int sc = scales[i];
  const float float_max = 1e8f;
  float min_area = 0.f;
  int arg = 0;
  bool any_inside = false;
  for (int j = 0; j < n_boxes; ++j) {
    const TBox B = load_tbox(boxes, j, rotated);
    float t[6];
    faces(pt, B, rotated, t);
    const bool inside = inside_box(t);
    any_inside |= inside;
    const bool scale_ok = sc == best[j];
    float vol = B.b[3] * B.b[4] * B.b[5];
    const float c = (inside && scale_ok) ? centerness_of(t) : -1.f;
    if (!inside || !scale_ok || !(c > top_c[j])) vol = float_max;
    if (j == 0 || vol < min_area) { min_area = vol; arg = j; }           // volumes.min(dim=1): first minimum
  }
  const TBox B = load_tbox(boxes, arg, rotated);
  float t[6];
  faces(pt, B, rotated, t);
  labels[i] = min_area == float_max ? -1 : gt_labels[arg];
  centerness_t[i] = centerness_of(t);
  geo_occ[i] = any_inside ? 1 : 0;
  if (rotated) {
    float *o = bbox_t + (int64_t)i * 7;
#pragma unroll
    for (int k = 0; k < 7; ++k) o[k] = B.b[k];
  } else {
    float *o = bbox_t + (int64_t)i * 6;
    o[0] = pt[0] - t[0]; o[1] = pt[1] - t[2]; o[2] = pt[2] - t[4];
    o[3] = pt[0] + t[1]; o[4] = pt[1] + t[3]; o[5] = pt[2] + t[5];
  }
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_assign_targets(const float *points, const int32_t *scales, const float *boxes, const int64_t *gt_labels,
                                  int rotated, int n_scales, int limit, int centerness_topk, float *centerness_t,
                                  float *bbox_t, int64_t *labels, uint8_t *geo_occ, int32_t *workspace, int n_points,
                                  int n_boxes, sgc_stream_t stream) {
  if (n_points <= 0) return SGC_OK;
  if (!points || !scales || !centerness_t || !bbox_t || !labels || !geo_occ || !workspace)
    return set_error(SGC_EINVAL, "sgc_assign_targets: null pointer");
  if (n_boxes <= 0 || !boxes || !gt_labels) return set_error(SGC_EINVAL, "sgc_assign_targets: at least one box is required");
  if (n_scales < 1 || n_scales > 16) return set_error(SGC_EUNSUP, "sgc_assign_targets: 1..16 scales (got %d)", n_scales);
  if (centerness_topk + 1 > n_points || centerness_topk < 0)
    return set_error(SGC_EINVAL, "sgc_assign_targets: centerness_topk + 1 must be in [1, n_points]");
  if (n_boxes > 65535) return set_error(SGC_EUNSUP, "sgc_assign_targets: at most 65535 boxes");
  hipStream_t st = (hipStream_t)stream;
  int32_t *counts = workspace, *best = workspace + (int64_t)n_scales * n_boxes;
  float *top_c = reinterpret_cast<float *>(best + n_boxes);
  const int nz = n_scales * n_boxes;
  hipLaunchKernelGGL(tgt_zero_kernel, dim3(ceil_div(nz, 256)), dim3(256), 0, st, counts, nz);
  int rc = check_launch("tgt_zero_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(tgt_count_kernel, dim3(ceil_div(n_points, 256), n_boxes), dim3(256), 0, st, points, scales, boxes, rotated,
                     n_scales, counts, n_points, n_boxes);
  if ((rc = check_launch("tgt_count_kernel"))) return rc;
  hipLaunchKernelGGL(tgt_best_scale_kernel, dim3(ceil_div(n_boxes, 64)), dim3(64), 0, st, counts, best, n_scales, limit, n_boxes);
  if ((rc = check_launch("tgt_best_scale_kernel"))) return rc;
  hipLaunchKernelGGL(tgt_topc_kernel, dim3(n_boxes), dim3(256), 0, st, points, scales, boxes, rotated, best, top_c,
                     centerness_topk + 1, n_points);
  if ((rc = check_launch("tgt_topc_kernel"))) return rc;
  hipLaunchKernelGGL(tgt_assign_kernel, dim3(ceil_div(n_points, 256)), dim3(256), 0, st, points, scales, boxes, gt_labels, rotated,
                     best, top_c, centerness_t, bbox_t, labels, geo_occ, n_points, n_boxes);
  return check_launch("tgt_assign_kernel");
}
