// Rotated BEV NMS for all classes of a scene in two launches == the class loop of mmdet3d `box3d_multiclass_nms`
// (packages/mmdetection3d/mmdet3d/core/post_processing/box3d_nms.py:52-68) over `nms_bev` (:231-268) ->
// mmcv.ops.nms_rotated (the IoU: box_iou_rotated_utils.hpp, which the reference's DFA3D package vendors under
// packages/3D-deformable-attention/DFA3D/dfa3D/ops/csrc/common/ -- tests pin this file to a g++ build of it,
// tests/golden/box_iou_rotated.npz; the sweep: mmcv-full 1.5.3 nms_rotated_cuda.cuh, a pip dependency); called from SunRgbdImVoxelHeadV2._nms (imvoxel_head_v2.py:565-584, ARKit configs:
// score_thr 0, 17 classes x up to 3000 candidates each -- the reference launches 17 mask kernels, copies 17 bit
// matrices to the host and sweeps them on the CPU).
//   1. rnms_mask_kernel, grid (column block, row block, class): bit (p, q) = IoU(box at sorted position p, box at
//      position q > p) > thr, with the reference kernel's fp32 operation order (vertex construction after the
//      pair-midpoint shift, 16 edge/edge tests, 2 x 4 containment tests, exchange sort by polar angle, Graham scan,
//      fan area).  Pairs that are provably disjoint with a 0.1 % margin (circumcircles, then a separating-axis test)
//      cannot produce an intersection point in any arithmetic and skip all of that -- in a room almost all pairs do;
//      the survivors of a 64 x 64 tile are packed so that every lane clips a real pair.
//   2. rnms_sweep_kernel, one workgroup per class: the greedy pass over the bit matrix (as nms3d.hip).
// Candidate counts per class stay on the device (`counts`): no host read-back between sort, mask and sweep.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgcdet_amd.h"
#include "common.hpp"

#pragma clang fp contract(off)   // threshold decisions on fp32 values: same roundings as the unfused reference build

namespace sgc {

struct RPt { float x, y; };
__device__ __forceinline__ float rcross(RPt a, RPt b) { return a.x * b.y - b.x * a.y; }
__device__ __forceinline__ float rdot(RPt a, RPt b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ RPt rsub(RPt a, RPt b) { return RPt{a.x - b.x, a.y - b.y}; }

// (xc, yc, w, h) + the four products of get_rotated_vertices that do not depend on the pair
struct RBox { float x, y, w, h, sh, cw, ch, sw, rad; };      // rad: circumradius

__device__ __forceinline__ RBox make_rbox(float x, float y, float w, float h, float a) {
  const double theta = (double)a;                                  // the reference evaluates cos / sin in double
  const float cos2 = (float)cos(theta) * 0.5f, sin2 = (float)sin(theta) * 0.5f;
  return RBox{x, y, w, h, sin2 * h, cos2 * w, cos2 * h, sin2 * w, 0.5f * sqrtf(w * w + h * h)};
}

__device__ __forceinline__ void rot_vertices(const RBox &b, float xc, float yc, RPt *pts) {
  pts[0].x = xc - b.sh - b.cw;
  pts[0].y = yc + b.ch - b.sw;
  pts[1].x = xc + b.sh - b.cw;
  pts[1].y = yc - b.ch - b.sw;
  pts[2].x = 2 * xc - pts[0].x;
  pts[2].y = 2 * yc - pts[0].y;
  pts[3].x = 2 * xc - pts[1].x;
  pts[3].y = 2 * yc - pts[1].y;
}

// Cheap pre-test of rot_iou: false only when the two rectangles are provably disjoint with a 0.1 % margin -- separated
// circumcircles, or a separating axis among the four edge directions -- in which case the reference's 16 edge/edge
// and 8 containment tests (whose rounding errors are ~1e-7 relative) find no point and its IoU is exactly 0.
// hw = (cw, sw) and hh = (-sh, ch) are the half-edge vectors of a box; on the axis hw1 box 1 reaches |hw1|^2 and
// box 2 reaches |hw2.hw1| + |hh2.hw1| (all in units of |hw1|, which cancels).
__device__ __forceinline__ bool sat_separated(float dx, float dy, float ax, float ay, float bx, float by, float cx, float cy) {
  // axis a (a half-edge of one box, whose other half-edge is perpendicular to it); b, c: the half-edges of the other
  const float reach = (ax * ax + ay * ay) + fabsf(bx * ax + by * ay) + fabsf(cx * ax + cy * ay);
  return fabsf(dx * ax + dy * ay) > 1.001f * reach;
}
__device__ __forceinline__ bool may_overlap(const RBox &b1, const RBox &b2) {
  const float area1 = b1.w * b1.h, area2 = b2.w * b2.h;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return false;
  const float dx = b1.x - b2.x, dy = b1.y - b2.y;
  const float r = 1.001f * (b1.rad + b2.rad);
  if (dx * dx + dy * dy > r * r) return false;
  if (sat_separated(dx, dy, b1.cw, b1.sw, b2.cw, b2.sw, -b2.sh, b2.ch)) return false;
  if (sat_separated(dx, dy, -b1.sh, b1.ch, b2.cw, b2.sw, -b2.sh, b2.ch)) return false;
  if (sat_separated(dx, dy, b2.cw, b2.sw, b1.cw, b1.sw, -b1.sh, b1.ch)) return false;
  if (sat_separated(dx, dy, -b2.sh, b2.ch, b1.cw, b1.sw, -b1.sh, b1.ch)) return false;
  return true;
}

__device__ float rot_iou(const RBox &b1, const RBox &b2) {
  const float area1 = b1.w * b1.h, area2 = b2.w * b2.h;
  if (!may_overlap(b1, b2)) return 0.f;
  const float sx = (b1.x + b2.x) * 0.5f, sy = (b1.y + b2.y) * 0.5f;
  RPt p1[4], p2[4], ip[24], q[24];
  rot_vertices(b1, b1.x - sx, b1.y - sy, p1);
  rot_vertices(b2, b2.x - sx, b2.y - sy, p2);
  RPt v1[4], v2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v1[i] = rsub(p1[(i + 1) & 3], p1[i]); v2[i] = rsub(p2[(i + 1) & 3], p2[i]); }
  int num = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float det = rcross(v2[j], v1[i]);
      if (fabs((double)det) <= 1e-14) continue;
      const RPt v12 = rsub(p2[j], p1[i]);
      const float t1 = rcross(v2[j], v12) / det, t2 = rcross(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        ip[num].x = p1[i].x + v1[i].x * t1;
        ip[num].y = p1[i].y + v1[i].y * t1;
        ++num;
      }
    }
  {
    const RPt AB = v2[0], DA = v2[3];
    const float ABdotAB = rdot(AB, AB), ADdotAD = rdot(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const RPt AP = rsub(p1[i], p2[0]);
      const float APdotAB = rdot(AP, AB), APdotAD = -rdot(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) ip[num++] = p1[i];
    }
  }
  {
    const RPt AB = v1[0], DA = v1[3];
    const float ABdotAB = rdot(AB, AB), ADdotAD = rdot(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const RPt AP = rsub(p2[i], p1[0]);
      const float APdotAB = rdot(AP, AB), APdotAD = -rdot(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) ip[num++] = p2[i];
    }
  }
  float inter = 0.f;
  if (num > 2) {
    int t = 0;
    for (int i = 1; i < num; ++i)
      if (ip[i].y < ip[t].y || (ip[i].y == ip[t].y && ip[i].x < ip[t].x)) t = i;
    const RPt start = ip[t];
    for (int i = 0; i < num; ++i) q[i] = rsub(ip[i], start);
    { const RPt tmp = q[0]; q[0] = q[t]; q[t] = tmp; }
    float dist[24];
    for (int i = 0; i < num; ++i) dist[i] = rdot(q[i], q[i]);
    for (int i = 1; i < num - 1; ++i)
      for (int j = i + 1; j < num; ++j) {
        const float cp = rcross(q[i], q[j]);
        if (((double)cp < -1e-6) || (fabs((double)cp) < 1e-6 && dist[i] > dist[j])) {
          const RPt qt = q[i]; q[i] = q[j]; q[j] = qt;
          const float dt = dist[i]; dist[i] = dist[j]; dist[j] = dt;
        }
      }
    int k;
    for (k = 1; k < num; ++k)
      if ((double)dist[k] > 1e-8) break;
    if (k < num) {
      q[1] = q[k];
      int m = 2;
      for (int i = k + 1; i < num; ++i) {
        while (m > 1 && rcross(rsub(q[i], q[m - 2]), rsub(q[m - 1], q[m - 2])) >= 0) --m;
        q[m++] = q[i];
      }
      if (m > 2) {
        float area = 0.f;
        for (int i = 1; i < m - 1; ++i) area += fabsf(rcross(rsub(q[i], q[0]), rsub(q[i + 1], q[0])));
        inter = area * 0.5f;
      }
    }
  }
  return inter / (area1 + area2 - inter);
}

__device__ __forceinline__ RBox load_bev(const float *boxes, int64_t i) {       // xyxyr -> xywhr (box3d_nms.py:256-262)
  const float *b = boxes + i * 5;
  return make_rbox((b[0] + b[2]) / 2, (b[1] + b[3]) / 2, b[2] - b[0], b[3] - b[1], b[4]);
}

__global__ __launch_bounds__(256) void rnms_mask_kernel(const float *__restrict__ boxes, const int64_t *__restrict__ order,
                                                        const int32_t *__restrict__ counts, float thr,
                                                        unsigned long long *__restrict__ mask, int K, int words) {
  __shared__ RBox cb[64], rb[64];
  __shared__ unsigned long long bits[64];
  __shared__ unsigned short todo[64 * 64];       // (row << 6 | column) of the pairs that need the full IoU
  __shared__ int wave_total[4];
  const int c = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = min(counts[c], K);
  const int q0 = blockIdx.x * 64, p0 = blockIdx.y * 64;
  if (q0 >= n || p0 >= n) return;
  if (blockIdx.x < blockIdx.y) {                 // every q of this column block precedes every p of this row block
    if (tid < 64 && p0 + tid < n) mask[((int64_t)c * K + p0 + tid) * words + blockIdx.x] = 0ull;
    return;
  }
  const int64_t *ord = order + (int64_t)c * K;
  if (wv == 0 && q0 + lane < n) cb[lane] = load_bev(boxes, ord[q0 + lane]);
  if (wv == 1 && p0 + lane < n) rb[lane] = load_bev(boxes, ord[p0 + lane]);
  if (wv == 2) bits[lane] = 0ull;
  __syncthreads();
  // 1. lane = row box, wave = a quarter of the 64 column boxes: which pairs can overlap at all (in a room: few)
  unsigned need = 0u;
  if (p0 + lane < n) {
    const RBox a = rb[lane];
    const int t0 = max(wv * 16, blockIdx.x == blockIdx.y ? lane + 1 : 0), t1 = min(wv * 16 + 16, n - q0);
    for (int t = t0; t < t1; ++t)
      if (may_overlap(a, cb[t])) need |= 1u << (t - wv * 16);
  }
  // 2. the surviving pairs of the tile, packed: every lane then clips a real pair instead of idling through the
  //    other lanes' polygons
  const int mine = __popc(need);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_total[wv] = incl;
  __syncthreads();
  int at = incl - mine, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wv) at += wave_total[w];
    total += wave_total[w];
  }
  while (need) {
    const int t = __ffs((int)need) - 1 + wv * 16;
    need &= need - 1;
    todo[at++] = (unsigned short)(lane << 6 | t);
  }
  __syncthreads();
  for (int e = tid; e < total; e += 256) {
    const int r = todo[e] >> 6, t = todo[e] & 63;
    if (rot_iou(rb[r], cb[t]) > thr) atomicOr(&bits[r], 1ull << t);
  }
  __syncthreads();
  if (tid < 64 && p0 + tid < n) mask[((int64_t)c * K + p0 + tid) * words + blockIdx.x] = bits[tid];
}

__global__ __launch_bounds__(256) void rnms_sweep_kernel(const unsigned long long *__restrict__ mask,
                                                         const int64_t *__restrict__ order, const int32_t *__restrict__ counts,
                                                         int64_t *__restrict__ keep, int32_t *__restrict__ n_keep, int K,
                                                         int words) {
  __shared__ unsigned long long tile[64 * 64];     // 64 rows x up to 64 words
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int n = min(counts[c], K);
  const int nw = (n + 63) / 64;                    // words the mask kernel wrote for this class
  const unsigned long long *cm = mask + (int64_t)c * K * words;
  const int64_t *ord = order + (int64_t)c * K;
  int64_t *kp = keep + (int64_t)c * K;
  unsigned long long removed = 0ull;               // wave 0: lane w owns word w of the removed set
  int cnt = 0;
  for (int b = 0; b * 64 < n; ++b) {
    __syncthreads();
    for (int e = tid; e < 64 * nw; e += 256) {
      const int r = e / nw, w = e - r * nw;
      const int p = b * 64 + r;
      tile[r * 64 + w] = p < n ? cm[(int64_t)p * words + w] : 0ull;
    }
    __syncthreads();
    if (tid < 64) {
      const int rows = min(64, n - b * 64);
      for (int r = 0; r < rows; ++r) {
        const unsigned long long wb = __shfl(removed, b);          // word b is owned by lane b
        if (!((wb >> r) & 1ull)) {
          if (lane == 0) kp[cnt] = ord[b * 64 + r];
          ++cnt;
          if (lane < nw) removed |= tile[r * 64 + lane];
        }
      }
    }
  }
  if (tid == 0) n_keep[c] = cnt;
}

__global__ void box_iou_rotated_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ iou,
                                       int n, int m) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * m) return;
  const int i = (int)(e / m), j = (int)(e - (int64_t)i * m);
  const float *pa = a + (int64_t)i * 5, *pb = b + (int64_t)j * 5;
  iou[e] = rot_iou(make_rbox(pa[0], pa[1], pa[2], pa[3], pa[4]), make_rbox(pb[0], pb[1], pb[2], pb[3], pb[4]));
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_nms_rotated_bev(const float *boxes, const int64_t *order, const int32_t *counts, float iou_thr,
                                   int64_t *keep, int32_t *n_keep, uint64_t *workspace, int K, int C,
                                   sgc_stream_t stream) {
  if (C <= 0) return SGC_OK;
  if (!counts || !n_keep) return set_error(SGC_EINVAL, "sgc_nms_rotated_bev: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (K <= 0) {
    const hipError_t e = hipMemsetAsync(n_keep, 0, sizeof(int32_t) * C, st);
    return e == hipSuccess ? SGC_OK : set_error(SGC_ELAUNCH, "sgc_nms_rotated_bev: %s", hipGetErrorString(e));
  }
  if (!boxes || !order || !keep || !workspace) return set_error(SGC_EINVAL, "sgc_nms_rotated_bev: null pointer");
  if (K > 4096) return set_error(SGC_EUNSUP, "sgc_nms_rotated_bev: at most 4096 candidates per class (got %d)", K);
  if (C > 65535) return set_error(SGC_EUNSUP, "sgc_nms_rotated_bev: at most 65535 classes (got %d)", C);
  const int words = (K + 63) / 64;
  hipLaunchKernelGGL(rnms_mask_kernel, dim3(words, words, C), dim3(256), 0, st, boxes, order, counts, iou_thr,
                     reinterpret_cast<unsigned long long *>(workspace), K, words);
  int rc = check_launch("rnms_mask_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(rnms_sweep_kernel, dim3(C), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(workspace),
                     order, counts, keep, n_keep, K, words);
  return check_launch("rnms_sweep_kernel");
}

extern "C" int sgc_box_iou_rotated(const float *a, const float *b, float *iou, int n, int m, sgc_stream_t stream) {
  if (n <= 0 || m <= 0) return SGC_OK;
  if (!a || !b || !iou) return set_error(SGC_EINVAL, "sgc_box_iou_rotated: null pointer");
  const int64_t total = (int64_t)n * m;
  hipLaunchKernelGGL(box_iou_rotated_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream, a, b,
                     iou, n, m);
  return check_launch("box_iou_rotated_kernel");
}
