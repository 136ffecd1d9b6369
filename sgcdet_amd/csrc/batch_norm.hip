// Training-mode BatchNorm over channels-last rows [rows, C]: nn.BatchNorm3d over [1, C, X, Y, Z] IS the
// per-channel statistics of the X*Y*Z rows (necks/imvoxelnet.py:36-64,146-173 -- every convolution of the neck is followed by one).
//
//   sgc_bn_rows_forward    batch mean / biased variance per channel (Welford per thread, Chan's merge across threads and
//                          workgroups in a FIXED order: the same bits every run), the running-statistics update of
//                          nn.BatchNorm (unbiased variance, momentum), y = (x - mean) * invstd * weight + bias.
//   sgc_bn_rows_backward   dweight = sum dy * xhat, dbias = sum dy, dx = weight * invstd * (dy - dbias / rows - xhat * dweight / rows).
//
// Round 2 ran torch's channels-last batch-norm kernels here: 43 us (statistics) + 52 us (backward reduction) per layer for
// 26 MB tensors, 1.5 ms of a 25 ms training step.  These are plain column reductions: a workgroup walks a slab of rows with
// 16-byte loads (a thread owns 4 channels), partial results meet in a small workspace.
#include "common.hpp"

namespace sgc {

constexpr int BN_T = 256;          // threads per workgroup
constexpr int BN_MAXG = 256;       // slabs (workgroups of the partial kernels)

// merge (nb, mb, M2b) into (na, ma, M2a): Chan et al.
__device__ __forceinline__ void chan_merge(float &na, float &ma, float &M2a, float nb, float mb, float M2b) {
  if (nb == 0.f) return;
  const float n = na + nb, d = mb - ma, f = nb / n;
  ma += d * f;
  M2a += M2b + d * d * na * f;
  na = n;
}

// partial statistics of slab g: ws[(g * 2 + 0) * C + c] = mean, ws[(g * 2 + 1) * C + c] = M2 (sum of squared deviations)
__global__ __launch_bounds__(BN_T) void bn_stats_partial_kernel(const float *__restrict__ x, float *__restrict__ ws, int rows, int C,
                                                                int rows_per_slab) {
  extern __shared__ float bn_lds[];                    // [RL][CG * 4][2]
  const int C4 = C >> 2;
  const int CG = C4 < BN_T ? C4 : BN_T;                // column groups (float4) handled at once
  const int RL = BN_T / CG;                            // row lanes
  const int tid = threadIdx.x, cg = tid % CG, rl = tid / CG;
  const int r_lo = blockIdx.x * rows_per_slab, r_hi = min(rows, r_lo + rows_per_slab);
  for (int c0 = 0; c0 < C4; c0 += CG) {
    const int c4 = c0 + cg;
    float n = 0.f, mean[4] = {0.f, 0.f, 0.f, 0.f}, M2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4 && rl < RL) {
      for (int r = r_lo + rl; r < r_hi; r += RL) {
        const float4 v4 = *reinterpret_cast<const float4 *>(x + (int64_t)r * C + c4 * 4);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        n += 1.f;
        const float inv = __frcp_rn(n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - mean[e];
          mean[e] += d * inv;
          M2[e] += d * (v[e] - mean[e]);
        }
      }
    }
    // row lanes -> one partial per channel, merged in lane order by lane 0 of the column group
    if (rl < RL && c4 < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bn_lds[((rl * CG + cg) * 4 + e) * 3 + 0] = n;
        bn_lds[((rl * CG + cg) * 4 + e) * 3 + 1] = mean[e];
        bn_lds[((rl * CG + cg) * 4 + e) * 3 + 2] = M2[e];
      }
    }
    __syncthreads();
    if (rl == 0 && c4 < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float na = n, ma = mean[e], Ma = M2[e];
        for (int l = 1; l < RL; ++l) {
          const float *q = bn_lds + ((l * CG + cg) * 4 + e) * 3;
          chan_merge(na, ma, Ma, q[0], q[1], q[2]);
        }
        ws[((int64_t)blockIdx.x * 2 + 0) * C + c4 * 4 + e] = ma;
        ws[((int64_t)blockIdx.x * 2 + 1) * C + c4 * 4 + e] = Ma;
      }
    }
    __syncthreads();
  }
}

// merge the slabs; batch statistics, running statistics (nn.BatchNorm: unbiased variance), invstd.  BN_FJ lanes per channel each
// merge a contiguous run of slabs in order, lane 0 merges the BN_FJ results in lane order: a fixed tree, the same bits every run.
constexpr int BN_FJ = 8, BN_FC = BN_T / BN_FJ;       // lanes per channel, channels per workgroup
__global__ __launch_bounds__(BN_T) void bn_stats_final_kernel(const float *__restrict__ ws, int G, int rows, int rows_per_slab, int C, float eps,
                                                              float momentum, float *__restrict__ mean_out, float *__restrict__ invstd_out,
                                                              float *__restrict__ running_mean, float *__restrict__ running_var) {
  __shared__ float part[BN_FJ][BN_FC][3];
  const int cl = threadIdx.x % BN_FC, j = threadIdx.x / BN_FC;
  const int c = blockIdx.x * BN_FC + cl;
  const int per = (G + BN_FJ - 1) / BN_FJ;
  float na = 0.f, ma = 0.f, Ma = 0.f;
  if (c < C) {
    for (int g = j * per; g < min(G, (j + 1) * per); ++g) {
      const int cnt = min(rows, (g + 1) * rows_per_slab) - g * rows_per_slab;
      if (cnt <= 0) break;
      chan_merge(na, ma, Ma, (float)cnt, ws[((int64_t)g * 2 + 0) * C + c], ws[((int64_t)g * 2 + 1) * C + c]);
    }
  }
  part[j][cl][0] = na; part[j][cl][1] = ma; part[j][cl][2] = Ma;
  __syncthreads();
  if (j != 0 || c >= C) return;
  for (int l = 1; l < BN_FJ; ++l) chan_merge(na, ma, Ma, part[l][cl][0], part[l][cl][1], part[l][cl][2]);
  const float var = Ma / (float)rows;
  mean_out[c] = ma;
  invstd_out[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * ma;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (rows > 1 ? Ma / (float)(rows - 1) : var);
}

// residual / relu (round 5, sgc_bn_rows_act_forward): y = relu?(bn(x) + residual?) -- the ResBlock tail `relu(norm2(conv2) + identity)` and the
// `relu(norm(conv))` pairs of the neck (necks/imvoxelnet.py:36-64) in the normalisation pass instead of two more elementwise kernels
__global__ __launch_bounds__(BN_T) void bn_apply_kernel(const float4 *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                        const float *__restrict__ w, const float *__restrict__ b, float4 *__restrict__ y,
                                                        int64_t total4, int C4, const float4 *__restrict__ residual, int relu) {
  for (int64_t i = (int64_t)blockIdx.x * BN_T + threadIdx.x; i < total4; i += (int64_t)gridDim.x * BN_T) {
    const int c = (int)(i % C4) * 4;
    const float4 v = x[i];
    const float4 m = *reinterpret_cast<const float4 *>(mean + c), s = *reinterpret_cast<const float4 *>(invstd + c);
    const float4 ww = *reinterpret_cast<const float4 *>(w + c), bb = *reinterpret_cast<const float4 *>(b + c);
    float4 o;
    o.x = (v.x - m.x) * s.x * ww.x + bb.x; o.y = (v.y - m.y) * s.y * ww.y + bb.y;
    o.z = (v.z - m.z) * s.z * ww.z + bb.z; o.w = (v.w - m.w) * s.w * ww.w + bb.w;
    if (residual) { const float4 r = residual[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    y[i] = o;
  }
}

// partial sums of slab g: ws[(g * 2 + 0) * C + c] = sum dy, ws[(g * 2 + 1) * C + c] = sum dy * xhat
// y_or_null: the forward's output when it ended in a ReLU -- the incoming gradient counts only where y > 0 (torch's threshold_backward)
__global__ __launch_bounds__(BN_T) void bn_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, float *__restrict__ ws, int rows, int C,
                                                              int rows_per_slab, const float *__restrict__ y_or_null) {
  extern __shared__ float bn_lds[];                    // [RL][CG * 4][2]
  const int C4 = C >> 2;
  const int CG = C4 < BN_T ? C4 : BN_T, RL = BN_T / CG;
  const int tid = threadIdx.x, cg = tid % CG, rl = tid / CG;
  const int r_lo = blockIdx.x * rows_per_slab, r_hi = min(rows, r_lo + rows_per_slab);
  for (int c0 = 0; c0 < C4; c0 += CG) {
    const int c4 = c0 + cg;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4 && rl < RL) {
      const float4 m4 = *reinterpret_cast<const float4 *>(mean + c4 * 4), i4 = *reinterpret_cast<const float4 *>(invstd + c4 * 4);
      const float m[4] = {m4.x, m4.y, m4.z, m4.w}, is[4] = {i4.x, i4.y, i4.z, i4.w};
      for (int r = r_lo + rl; r < r_hi; r += RL) {
        const float4 v4 = *reinterpret_cast<const float4 *>(x + (int64_t)r * C + c4 * 4);
        float4 g4 = *reinterpret_cast<const float4 *>(dy + (int64_t)r * C + c4 * 4);
        if (y_or_null) {
          const float4 y4 = *reinterpret_cast<const float4 *>(y_or_null + (int64_t)r * C + c4 * 4);
          g4.x = y4.x > 0.f ? g4.x : 0.f; g4.y = y4.y > 0.f ? g4.y : 0.f; g4.z = y4.z > 0.f ? g4.z : 0.f; g4.w = y4.w > 0.f ? g4.w : 0.f;
        }
        const float v[4] = {v4.x, v4.y, v4.z, v4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += g[e];
          s2[e] += g[e] * ((v[e] - m[e]) * is[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bn_lds[((rl * CG + cg) * 4 + e) * 2 + 0] = s1[e];
        bn_lds[((rl * CG + cg) * 4 + e) * 2 + 1] = s2[e];
      }
    }
    __syncthreads();
    if (rl == 0 && c4 < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = s1[e], bsum = s2[e];
        for (int l = 1; l < RL; ++l) {
          a += bn_lds[((l * CG + cg) * 4 + e) * 2 + 0];
          bsum += bn_lds[((l * CG + cg) * 4 + e) * 2 + 1];
        }
        ws[((int64_t)blockIdx.x * 2 + 0) * C + c4 * 4 + e] = a;
        ws[((int64_t)blockIdx.x * 2 + 1) * C + c4 * 4 + e] = bsum;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(BN_T) void bn_bwd_final_kernel(const float *__restrict__ ws, int G, int C, float *__restrict__ dw, float *__restrict__ db) {
  __shared__ float part[BN_FJ][BN_FC][2];
  const int cl = threadIdx.x % BN_FC, j = threadIdx.x / BN_FC;
  const int c = blockIdx.x * BN_FC + cl;
  const int per = (G + BN_FJ - 1) / BN_FJ;
  float a = 0.f, b = 0.f;
  if (c < C) {
    for (int g = j * per; g < min(G, (j + 1) * per); ++g) {     // runs of slabs in order, then the runs in order: the same bits every run
      a += ws[((int64_t)g * 2 + 0) * C + c];
      b += ws[((int64_t)g * 2 + 1) * C + c];
    }
  }
  part[j][cl][0] = a; part[j][cl][1] = b;
  __syncthreads();
  if (j != 0 || c >= C) return;
  for (int l = 1; l < BN_FJ; ++l) { a += part[l][cl][0]; b += part[l][cl][1]; }
  db[c] = a;
  dw[c] = b;
}

__global__ __launch_bounds__(BN_T) void bn_bwd_apply_kernel(const float4 *__restrict__ x, const float4 *__restrict__ dy, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, const float *__restrict__ w,
                                                            const float *__restrict__ dw, const float *__restrict__ db, float4 *__restrict__ dx,
                                                            int64_t total4, int C4, float inv_rows, const float4 *__restrict__ y_or_null,
                                                            float4 *__restrict__ dres_or_null) {
  for (int64_t i = (int64_t)blockIdx.x * BN_T + threadIdx.x; i < total4; i += (int64_t)gridDim.x * BN_T) {
    const int c = (int)(i % C4) * 4;
    const float4 v4 = x[i];
    float4 g4 = dy[i];
    if (y_or_null) {
      const float4 y4 = y_or_null[i];
      g4.x = y4.x > 0.f ? g4.x : 0.f; g4.y = y4.y > 0.f ? g4.y : 0.f; g4.z = y4.z > 0.f ? g4.z : 0.f; g4.w = y4.w > 0.f ? g4.w : 0.f;
    }
    if (dres_or_null) dres_or_null[i] = g4;        // the gradient of the added identity: the masked incoming gradient
    const float v[4] = {v4.x, v4.y, v4.z, v4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float is = invstd[c + e], xhat = (v[e] - mean[c + e]) * is;
      o[e] = w[c + e] * is * (g[e] - db[c + e] * inv_rows - xhat * dw[c + e] * inv_rows);
    }
    dx[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

static void bn_geometry(int rows, int &G, int &rows_per_slab) {
  G = ceil_div(rows, 64);
  if (G > BN_MAXG) G = BN_MAXG;
  rows_per_slab = ceil_div(rows, G);
  G = ceil_div(rows, rows_per_slab);
}

}  // namespace sgc

using namespace sgc;

extern "C" int64_t sgc_bn_rows_workspace_floats(int rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  int G, rps;
  bn_geometry(rows, G, rps);
  return (int64_t)G * 2 * C;
}

static int bn_forward_impl(const float *x, const float *weight, const float *bias, float *running_mean_or_null,
                           float *running_var_or_null, float momentum, float eps, const float *residual_or_null, int relu, float *y,
                           float *mean_out, float *invstd_out, float *workspace, int64_t workspace_floats, int rows, int C,
                           sgc_stream_t stream) {
  if (!x || !weight || !bias || !y || !mean_out || !invstd_out || !workspace)
    return set_error(SGC_EINVAL, "sgc_bn_rows_forward: null pointer");
  if ((uintptr_t)residual_or_null & 15) return set_error(SGC_EINVAL, "sgc_bn_rows_forward: pointers must be 16-byte aligned");
  if (rows <= 0 || C <= 0) return set_error(SGC_EINVAL, "sgc_bn_rows_forward: bad size");
  if (C % 4) return set_error(SGC_EUNSUP, "sgc_bn_rows_forward: needs C %% 4 == 0");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)mean_out | (uintptr_t)invstd_out) & 15)
    return set_error(SGC_EINVAL, "sgc_bn_rows_forward: pointers must be 16-byte aligned");
  int G, rps;
  bn_geometry(rows, G, rps);
  if (workspace_floats < (int64_t)G * 2 * C) return set_error(SGC_EINVAL, "sgc_bn_rows_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int C4 = C / 4, CG = C4 < BN_T ? C4 : BN_T, RL = BN_T / CG;
  const size_t smem = (size_t)RL * CG * 4 * 3 * sizeof(float);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(G), dim3(BN_T), smem, st, x, workspace, rows, C, rps);
  int rc = check_launch("bn_stats_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(C, BN_FC)), dim3(BN_T), 0, st, (const float *)workspace, G, rows, rps, C, eps, momentum,
                     mean_out, invstd_out, running_mean_or_null, running_var_or_null);
  rc = check_launch("bn_stats_final_kernel");
  if (rc) return rc;
  const int64_t total4 = (int64_t)rows * C4;
  const int g = (int)((total4 + BN_T - 1) / BN_T < 4096 ? (total4 + BN_T - 1) / BN_T : 4096);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(g), dim3(BN_T), 0, st, reinterpret_cast<const float4 *>(x), (const float *)mean_out,
                     (const float *)invstd_out, weight, bias, reinterpret_cast<float4 *>(y), total4, C4,
                     reinterpret_cast<const float4 *>(residual_or_null), relu);
  return check_launch("bn_apply_kernel");
}

extern "C" int sgc_bn_rows_forward(const float *x, const float *weight, const float *bias, float *running_mean_or_null,
                                   float *running_var_or_null, float momentum, float eps, float *y, float *mean_out,
                                   float *invstd_out, float *workspace, int64_t workspace_floats, int rows, int C,
                                   sgc_stream_t stream) {
  return bn_forward_impl(x, weight, bias, running_mean_or_null, running_var_or_null, momentum, eps, nullptr, 0, y, mean_out, invstd_out,
                         workspace, workspace_floats, rows, C, stream);
}
extern "C" int sgc_bn_rows_act_forward(const float *x, const float *weight, const float *bias, float *running_mean_or_null,
                                       float *running_var_or_null, float momentum, float eps, const float *residual_or_null, int relu,
                                       float *y, float *mean_out, float *invstd_out, float *workspace, int64_t workspace_floats,
                                       int rows, int C, sgc_stream_t stream) {
  return bn_forward_impl(x, weight, bias, running_mean_or_null, running_var_or_null, momentum, eps, residual_or_null, relu ? 1 : 0, y, mean_out,
                         invstd_out, workspace, workspace_floats, rows, C, stream);
}

static int bn_backward_impl(const float *x, const float *dy, const float *y_or_null, const float *mean, const float *invstd, const float *weight,
                            float *dx, float *dweight, float *dbias, float *dres_or_null, float *workspace, int64_t workspace_floats, int rows,
                            int C, sgc_stream_t stream) {
  if (!x || !dy || !mean || !invstd || !weight || !dx || !dweight || !dbias || !workspace)
    return set_error(SGC_EINVAL, "sgc_bn_rows_backward: null pointer");
  if (((uintptr_t)y_or_null | (uintptr_t)dres_or_null) & 15) return set_error(SGC_EINVAL, "sgc_bn_rows_backward: pointers must be 16-byte aligned");
  if (rows <= 0 || C <= 0) return set_error(SGC_EINVAL, "sgc_bn_rows_backward: bad size");
  if (C % 4) return set_error(SGC_EUNSUP, "sgc_bn_rows_backward: needs C %% 4 == 0");
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)mean | (uintptr_t)invstd) & 15)
    return set_error(SGC_EINVAL, "sgc_bn_rows_backward: pointers must be 16-byte aligned");
  int G, rps;
  bn_geometry(rows, G, rps);
  if (workspace_floats < (int64_t)G * 2 * C) return set_error(SGC_EINVAL, "sgc_bn_rows_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int C4 = C / 4, CG = C4 < BN_T ? C4 : BN_T, RL = BN_T / CG;
  const size_t smem = (size_t)RL * CG * 4 * 2 * sizeof(float);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(G), dim3(BN_T), smem, st, x, dy, mean, invstd, workspace, rows, C, rps, y_or_null);
  int rc = check_launch("bn_bwd_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(ceil_div(C, BN_FC)), dim3(BN_T), 0, st, (const float *)workspace, G, C, dweight, dbias);
  rc = check_launch("bn_bwd_final_kernel");
  if (rc) return rc;
  const int64_t total4 = (int64_t)rows * C4;
  const int g = (int)((total4 + BN_T - 1) / BN_T < 4096 ? (total4 + BN_T - 1) / BN_T : 4096);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(g), dim3(BN_T), 0, st, reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(dy), mean,
                     invstd, weight, (const float *)dweight, (const float *)dbias, reinterpret_cast<float4 *>(dx), total4, C4, 1.0f / (float)rows,
                     reinterpret_cast<const float4 *>(y_or_null), reinterpret_cast<float4 *>(dres_or_null));
  return check_launch("bn_bwd_apply_kernel");
}

extern "C" int sgc_bn_rows_backward(const float *x, const float *dy, const float *mean, const float *invstd, const float *weight,
                                    float *dx, float *dweight, float *dbias, float *workspace, int64_t workspace_floats, int rows,
                                    int C, sgc_stream_t stream) {
  return bn_backward_impl(x, dy, nullptr, mean, invstd, weight, dx, dweight, dbias, nullptr, workspace, workspace_floats, rows, C, stream);
}
// backward of sgc_bn_rows_act_forward: y_relu_or_null = its output when relu was set (the gradient passes where y > 0), dresidual_or_null
// = where to put the gradient of the added identity (the masked incoming gradient; null when there was none or nobody needs it)
extern "C" int sgc_bn_rows_act_backward(const float *x, const float *dy, const float *y_relu_or_null, const float *mean, const float *invstd,
                                        const float *weight, float *dx, float *dweight, float *dbias, float *dresidual_or_null,
                                        float *workspace, int64_t workspace_floats, int rows, int C, sgc_stream_t stream) {
  return bn_backward_impl(x, dy, y_relu_or_null, mean, invstd, weight, dx, dweight, dbias, dresidual_or_null, workspace, workspace_floats, rows, C,
                          stream);
}
