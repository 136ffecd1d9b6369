// Greedy NMS of axis-aligned 3D boxes == mmdet3d `aligned_3d_nms`
// (packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py:131-178 of the reference; called from
// ScanNetImVoxelHeadV2._nms, mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py:437-443).  The reference is a
// Python while-loop with ~10 torch launches and a `nonzero` host sync per kept box; here:
//   1. nms_mask_kernel: bit (p, q) of an n x n bit matrix = "box at processing position q (lower score) is dropped
//      when the box at position p is kept", with the reference's arithmetic and comparison (`iou * same_class <=
//      thresh` keeps -- so a NaN IoU drops the box whatever its class);
//   2. nms_sweep_kernel: one workgroup walks the positions in order; the `removed` bit set lives in the registers
//      of one wave (lane w = word w), the matrix streams through LDS 64 rows at a time.
// Positions: p = 0 is the best score = order[n-1] of the reference's ascending `argsort(scores)`.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgcdet_amd.h"
#include "common.hpp"

#pragma clang fp contract(off)   // the keep/drop decision compares fp32 values: no fused multiply-add

namespace sgc {

struct Box7 { float x1, y1, z1, x2, y2, z2, area; };

__device__ __forceinline__ Box7 load_box(const float *boxes, int64_t i) {
  const float *b = boxes + i * 6;
  Box7 r = {b[0], b[1], b[2], b[3], b[4], b[5], 0.f};
  r.area = (r.x2 - r.x1) * (r.y2 - r.y1) * (r.z2 - r.z1);
  return r;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, const int64_t *__restrict__ order,
                                                      const int64_t *__restrict__ labels, float thr,
                                                      unsigned long long *__restrict__ mask, int n, int words) {
  __shared__ Box7 cb[64];
  __shared__ int64_t cl[64];
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * 64, p = blockIdx.y * 64 + tid;
  if (blockIdx.x < blockIdx.y) {        // every q of this column block precedes every p of this row block
    if (p < n) mask[(int64_t)p * words + blockIdx.x] = 0ull;
    return;
  }
  if (q0 + tid < n) {
    const int64_t j = order[n - 1 - (q0 + tid)];
    cb[tid] = load_box(boxes, j);
    cl[tid] = labels[j];
  }
  __syncthreads();
  if (p >= n) return;
  const int64_t i = order[n - 1 - p];
  const Box7 a = load_box(boxes, i);
  const int64_t la = labels[i];
  unsigned long long bits = 0ull;
  for (int t = 0; t < 64; ++t) {
    const int q = q0 + t;
    if (q >= n || q <= p) continue;
    const Box7 b = cb[t];
    const float xx1 = fmaxf(a.x1, b.x1), yy1 = fmaxf(a.y1, b.y1), zz1 = fmaxf(a.z1, b.z1);
    const float xx2 = fminf(a.x2, b.x2), yy2 = fminf(a.y2, b.y2), zz2 = fminf(a.z2, b.z2);
    const float il = fmaxf(0.f, xx2 - xx1), iw = fmaxf(0.f, yy2 - yy1), ih = fmaxf(0.f, zz2 - zz1);
    const float inter = il * iw * ih;
    float iou = inter / (a.area + b.area - inter);
    iou = iou * (la == cl[t] ? 1.f : 0.f);
    if (!(iou <= thr)) bits |= 1ull << t;
  }
  mask[(int64_t)p * words + blockIdx.x] = bits;
}

__global__ __launch_bounds__(256) void nms_sweep_kernel(const unsigned long long *__restrict__ mask,
                                                        const int64_t *__restrict__ order, int64_t *__restrict__ keep,
                                                        int32_t *__restrict__ n_keep, int n, int words) {
  __shared__ unsigned long long tile[64 * 64];     // 64 rows x up to 64 words
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long removed = 0ull;               // wave 0: lane w owns word w of the removed set
  int cnt = 0;
  for (int b = 0; b * 64 < n; ++b) {
    __syncthreads();
    for (int e = tid; e < 64 * words; e += 256) {
      const int r = e / words, w = e - r * words;
      const int p = b * 64 + r;
      tile[r * 64 + w] = p < n ? mask[(int64_t)p * words + w] : 0ull;
    }
    __syncthreads();
    if (tid < 64) {
      const int rows = min(64, n - b * 64);
      for (int r = 0; r < rows; ++r) {
        const unsigned long long wb = __shfl(removed, b);          // word b is owned by lane b
        if (!((wb >> r) & 1ull)) {
          if (lane == 0) keep[cnt] = order[n - 1 - (b * 64 + r)];
          ++cnt;
          if (lane < words) removed |= tile[r * 64 + lane];
        }
      }
    }
  }
  if (tid == 0) *n_keep = cnt;
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_aligned_nms3d(const float *boxes, const int64_t *order, const int64_t *labels, float iou_thr,
                                 int64_t *keep, int32_t *n_keep, uint64_t *workspace, int n, sgc_stream_t stream) {
  if (!n_keep) return set_error(SGC_EINVAL, "sgc_aligned_nms3d: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) {
    const hipError_t e = hipMemsetAsync(n_keep, 0, sizeof(int32_t), st);
    return e == hipSuccess ? SGC_OK : set_error(SGC_ELAUNCH, "sgc_aligned_nms3d: %s", hipGetErrorString(e));
  }
  if (!boxes || !order || !labels || !keep || !workspace) return set_error(SGC_EINVAL, "sgc_aligned_nms3d: null pointer");
  if (n > 4096) return set_error(SGC_EUNSUP, "sgc_aligned_nms3d: at most 4096 candidates (got %d)", n);
  const int words = (n + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, st, boxes, order, labels, iou_thr,
                     reinterpret_cast<unsigned long long *>(workspace), n, words);
  int rc = check_launch("nms_mask_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(workspace),
                     order, keep, n_keep, n, words);
  return check_launch("nms_sweep_kernel");
}
