// Plane-sweep matching cost of DepthNet_Fusion, fused:
//   corr[n, d, y, x] = (1/K) * sum_k  sum_c  bilinear(f[nbr(n,k)], warp_{n,k}(x, y, depth_d))[c] * f[n, (y,x), c] / sqrt(C)
// Reference: mmdet3d_plugin/models/im2voxel/depth_utils/depth_est_fusion.py -- homo_warping (:87-126: pixel ->
// rot * (x, y, 1) * depth + trans -> perspective divide -> normalised with (W-1)/2, (H-1)/2 -> F.grid_sample,
// bilinear, zeros padding, align_corners = False) and the cost-volume loop of DepthNet_Fusion.forward (:233-240).
// The reference materialises the warped neighbour features [N, C, D, H, W] per neighbour (1.18 GB at 40 views x
// 128 ch x 12 planes x 60x80) and reduces over C afterwards; here a wave owns a pixel, keeps its own C-vector in
// registers, gathers the 4 bilinear corner rows of the neighbour (channels-last, one coalesced C-row each) per
// (neighbour, plane) and reduces the dot product in registers -- the warped volume never exists.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgcdet_amd.h"
#include "common.hpp"

#pragma clang fp contract(off)   // the sample position decides which pixels are read: keep the reference's op order

namespace sgc {

constexpr int PS_PIX = 64;    // pixels per workgroup (4 waves x 16), results staged in LDS for coalesced stores
constexpr int PS_MAXD = 32;

template <int VPL>            // channels per lane: C = 64 * VPL
__global__ __launch_bounds__(256) void plane_sweep_corr_kernel(const float *__restrict__ feat, const int32_t *__restrict__ nbr,
                                                               const float *__restrict__ rt, const float *__restrict__ depth,
                                                               float *__restrict__ corr, int N, int K, int H, int W, int C,
                                                               int D) {
  __shared__ float out_s[PS_MAXD][PS_PIX];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int HW = H * W;
  const int tiles = (HW + PS_PIX - 1) / PS_PIX;
  const int n = blockIdx.x / tiles, pix0 = (blockIdx.x % tiles) * PS_PIX;
  const float inv_sqrt_c = 1.0f / sqrtf((float)C);
  const float half_w = (float)(W - 1) / 2.0f, half_h = (float)(H - 1) / 2.0f;
  for (int pp = 0; pp < PS_PIX / 4; ++pp) {
    const int pl = wid * (PS_PIX / 4) + pp;
    const int pix = pix0 + pl;
    if (pix >= HW) break;
    const float fx = (float)(pix % W), fy = (float)(pix / W);
    float own[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) own[v] = v * 64 + lane < C ? feat[((int64_t)n * HW + pix) * C + v * 64 + lane] : 0.f;
    // Planes in groups of DG: all 4 * DG * VPL corner loads of a group are issued together (clamped index + zero
    // weight instead of a branch per corner); the dot products stay per lane until every neighbour is in, then one
    // butterfly per plane.
    constexpr int DG = 4;
    for (int d0 = 0; d0 < D; d0 += DG) {
      float part[DG];
#pragma unroll
      for (int j = 0; j < DG; ++j) part[j] = 0.f;
      for (int k = 0; k < K; ++k) {
        const float *m = rt + ((int64_t)n * K + k) * 12;  // rows of (src_proj @ inv(ref_proj))[:3, :4]
        // rot_xyz = rot @ (x, y, 1)
        const float rx = m[0] * fx + m[1] * fy + m[2], ry = m[4] * fx + m[5] * fy + m[6], rz = m[8] * fx + m[9] * fy + m[10];
        const float *src = feat + (int64_t)nbr[n * K + k] * HW * C;
        int off[DG][4];
        float wgt[DG][4];
#pragma unroll
        for (int j = 0; j < DG; ++j) {
          const float dep = depth[min(d0 + j, D - 1)];
          const float px = rx * dep + m[3], py = ry * dep + m[7], pz = rz * dep + m[11];   // * depth + trans
          const float u = px / pz, v_ = py / pz;
          const float gx = u / half_w - 1.0f, gy = v_ / half_h - 1.0f;          // the reference's normalisation
          const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;              // grid_sample, align_corners = False
          const float iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
          const bool in = ix > -1.0f && iy > -1.0f && ix < (float)W && iy < (float)H;   // false for NaN / inf too
          const float x0f = in ? floorf(ix) : 0.f, y0f = in ? floorf(iy) : 0.f;
          const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
          const float lx = ix - x0f, ly = iy - y0f, hx = 1.0f - lx, hy = 1.0f - ly;
          const bool okx0 = in && x0 >= 0, okx1 = in && x1 <= W - 1, oky0 = in && y0 >= 0, oky1 = in && y1 <= H - 1;
          const int cx0 = max(x0, 0), cx1 = min(x1, W - 1), cy0 = max(y0, 0), cy1 = min(y1, H - 1);
          off[j][0] = (cy0 * W + cx0) * C; wgt[j][0] = (oky0 && okx0) ? hx * hy : 0.f;   // nw, ne, sw, se
          off[j][1] = (cy0 * W + cx1) * C; wgt[j][1] = (oky0 && okx1) ? lx * hy : 0.f;
          off[j][2] = (cy1 * W + cx0) * C; wgt[j][2] = (oky1 && okx0) ? hx * ly : 0.f;
          off[j][3] = (cy1 * W + cx1) * C; wgt[j][3] = (oky1 && okx1) ? lx * ly : 0.f;
        }
        float val[DG][4][VPL];
#pragma unroll
        for (int j = 0; j < DG; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
              const int c = v * 64 + lane;
              val[j][q][v] = c < C ? src[off[j][q] + c] : 0.f;
            }
#pragma unroll
        for (int j = 0; j < DG; ++j) {
          float dot = 0.f;
#pragma unroll
          for (int v = 0; v < VPL; ++v) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) s += val[j][q][v] * wgt[j][q];
            dot += s * own[v];
          }
          part[j] += dot * inv_sqrt_c;      // per lane: the sum over lanes commutes with the sum over neighbours
        }
      }
#pragma unroll
      for (int j = 0; j < DG; ++j) {
        float t = part[j];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0 && d0 + j < D) out_s[d0 + j][pl] = t / (float)K;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < D * PS_PIX; e += 256) {
    const int d = e / PS_PIX, pl = e % PS_PIX;
    if (pix0 + pl < HW) corr[((int64_t)n * D + d) * HW + pix0 + pl] = out_s[d][pl];
  }
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_plane_sweep_corr(const float *feat, const int32_t *nbr, const float *rt, const float *depth,
                                    float *corr, int N, int K, int H, int W, int C, int D, sgc_stream_t stream) {
  if (!feat || !nbr || !rt || !depth || !corr) return set_error(SGC_EINVAL, "sgc_plane_sweep_corr: null pointer");
  if (N <= 0 || K <= 0 || H <= 0 || W <= 0 || C <= 0 || D <= 0) return set_error(SGC_EINVAL, "sgc_plane_sweep_corr: bad size");
  if (D > PS_MAXD) return set_error(SGC_EUNSUP, "sgc_plane_sweep_corr: at most %d depth planes", PS_MAXD);
  if (C > 256) return set_error(SGC_EUNSUP, "sgc_plane_sweep_corr: at most 256 channels");
  const int tiles = (H * W + PS_PIX - 1) / PS_PIX;
  const dim3 grid((unsigned)(N * tiles)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch ((C + 63) / 64) {
    case 1: hipLaunchKernelGGL(plane_sweep_corr_kernel<1>, grid, block, 0, st, feat, nbr, rt, depth, corr, N, K, H, W, C, D); break;
    case 2: hipLaunchKernelGGL(plane_sweep_corr_kernel<2>, grid, block, 0, st, feat, nbr, rt, depth, corr, N, K, H, W, C, D); break;
    case 3: hipLaunchKernelGGL(plane_sweep_corr_kernel<3>, grid, block, 0, st, feat, nbr, rt, depth, corr, N, K, H, W, C, D); break;
    default: hipLaunchKernelGGL(plane_sweep_corr_kernel<4>, grid, block, 0, st, feat, nbr, rt, depth, corr, N, K, H, W, C, D); break;
  }
  return check_launch("plane_sweep_corr_kernel");
}
