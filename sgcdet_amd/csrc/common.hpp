// Shared helpers of the gfx950 library (include/sgcdet_amd.h).  CDNA4 only: wave = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/sgcdet_amd.h"

namespace sgc {

constexpr int kWave = 64;
constexpr int kXcd = 8;  // MI355X: 8 XCDs, blocks are dealt round-robin over them

int set_error(int code, const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SGC_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return SGC_OK;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: `done` is a bitmask over device ordinals
// (one static per call site), so a process that drives several GPUs sets it once on each of them; thread-safe
// (setting it twice is harmless, the mask only saves the driver call on the hot path).
inline void ensure_dynamic_lds(const void *fn, int bytes, std::atomic<uint64_t> &done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.fetch_or(bit, std::memory_order_release);
}

// XCD-aware, bijective block -> tile map: the blocks that share an XCD (equal
// blockIdx % 8 under round-robin dispatch) walk one contiguous chunk of the tile
// range, so tiles that touch the same camera's maps meet in one 4 MiB L2.  Placement
// is a speed assumption only; any dispatch order gives the same results.
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
  const int x = bid % kXcd, j = bid / kXcd;
  const int base = ntiles / kXcd, rem = ntiles % kXcd;
  return x * base + (x < rem ? x : rem) + j;
}

typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));  // 4-byte aligned pair load

constexpr int kOutside = (int)0x80000000;
__device__ __forceinline__ int off_index(int off) { return off & 0x7fffffff; }

// Butterfly exchange with lane ^ o.  o = 1, 2 stay on the VALU as DPP quad permutes (no trip through the
// LDS crossbar that ds_bpermute takes); larger strides fall back to __shfl_xor.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_xor(float v, int o) {
  if (o == 1) return dpp_move<0xB1>(v);    // quad_perm:[1,0,3,2]
  if (o == 2) return dpp_move<0x4E>(v);    // quad_perm:[2,3,0,1]
  return __shfl_xor(v, o);
}

// Sample coordinate `loc * size - 0.5` with the REFERENCE's two roundings: `loc_h * spatial_h` is a float product
// (int promoted to float) and `- 0.5` a double subtraction rounded back to float on assignment
// (ms_depth_score_sample_cuda_kernel.cuh:133-135, wms_deform_attn_cuda_kernel.cuh:286-287) -- nvcc cannot contract
// that into an fma, so neither may hipcc: floor() of the result picks the pixel, a one-rounding fma flips it for
// ~1 sample in 1e7.  (float product exact in double, one rounding of the exact difference == float subtraction.)
__device__ __forceinline__ float sample_coord(float loc, float size) {
#pragma clang fp contract(off)
  const float prod = loc * size;
  return prod - 0.5f;
}

// a / size, correctly rounded (== the IEEE division of the reference's `sampling_offsets / offset_normalizer`,
// TU/deformable_cross_attention.py:428-455), for a wave-uniform size and rcp = RN(1 / size): one Newton correction of
// the reciprocal product (Markstein): q0 = a * rcp is within 1.5 ulp, r = a - q0 * size is exact in an fma, RN(q0 + r * rcp)
// is the correctly rounded quotient (checked against `/` on 3e8 random operands for every map size in use; a * rcp alone
// differs from the quotient in the last bit for 20 % of the operands at size 80).  3 instructions instead of ~10.
__device__ __forceinline__ float div_by_size(float a, float size, float rcp) {
  const float q0 = a * rcp;
  const float r = __builtin_fmaf(-q0, size, a);
  return __builtin_fmaf(r, rcp, q0);
}

// One trilinear sample of the DFA3D operator, reduced to what the gather needs:
// 4 corner weights (bilinear * depth score * attention weight) and 4 pixel indices
// (-1 = corner outside the map).  Semantics: ms_depth_score_sample_cuda_kernel.cuh:24-148
// and wms_deform_attn_cuda_kernel.cuh:24-80,286-294 of the reference.
struct Sample {
  float w[4];   // order: (h0,w0) (h0,w1) (h1,w0) (h1,w1)
  int off[4];   // pixel index inside the level (h*W + w); corners outside the map carry the sign bit
                // (kOutside) on top of a CLAMPED in-range index, so consumers can load unconditionally
                // (no exec-mask branch per load) and zero the value with a select
  float s[4];   // depth scores in the REFERENCE order (h0,w0) (h0,w1) (h1,w1) (h1,w0)
  // pieces the backward needs
  float lh, lw, ld;
  int d0;
  bool in2, in3;
};

// `taps` (optional, 8 floats): the two depth taps (d0, d0 + 1; 0 where gated off) of every corner in the GATHER order
// (h0,w0) (h0,w1) (h1,w0) (h1,w1) -- what the depth-score backward needs again (dfa3d_bwd_tile.hip keeps them instead of re-loading)
__device__ __forceinline__ void make_sample(Sample &sm, const float *__restrict__ dist_px0,
                                            int64_t pix_stride, int H, int W, int D,
                                            float x, float y, float z, float aw, float *taps = nullptr) {
  const float h_im = sample_coord(y, (float)H);
  const float w_im = sample_coord(x, (float)W);
  const float d_im = sample_coord(z, (float)D);
  sm.in2 = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  sm.in3 = sm.in2 && d_im > -1.f && d_im < (float)D;
  const float hf = floorf(h_im), wf = floorf(w_im), df = floorf(d_im);
  const int h0 = (int)hf, w0 = (int)wf, d0 = (int)df;
  const int h1 = h0 + 1, w1 = w0 + 1, d1 = d0 + 1;
  sm.lh = h_im - hf;
  sm.lw = w_im - wf;
  sm.ld = d_im - df;
  sm.d0 = d0;
  const float hh = 1.f - sm.lh, hw = 1.f - sm.lw, hd = 1.f - sm.ld;
  const bool okh0 = h0 >= 0, okh1 = h1 <= H - 1, okw0 = w0 >= 0, okw1 = w1 <= W - 1;
  const bool ok[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
  const int px[4] = {h0 * W + w0, h0 * W + w1, h1 * W + w0, h1 * W + w1};
  const int ch0 = min(max(h0, 0), H - 1), ch1 = min(max(h1, 0), H - 1);
  const int cw0 = min(max(w0, 0), W - 1), cw1 = min(max(w1, 0), W - 1);
  const int cpx[4] = {ch0 * W + cw0, ch0 * W + cw1, ch1 * W + cw0, ch1 * W + cw1};
  float sc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool use = sm.in2 && ok[k];
    sm.off[k] = use ? px[k] : (cpx[k] | kOutside);
    float v = 0.f;
    if (sm.in3 && ok[k]) {
      // one 8-byte load covers (d0, d1): every depth load touches 64 different cache lines per
      // wave instruction, so the instruction count is what the L1/TA path pays for
      const float *p = dist_px0 + (int64_t)px[k] * pix_stride;
      float va, vb;
      if (D >= 2) {
        const int base = d0 < 0 ? 0 : (d0 > D - 2 ? D - 2 : d0);
        const float2_u pr = *reinterpret_cast<const float2_u *>(p + base);
        va = d0 < 0 ? 0.f : (d0 == base ? pr.x : pr.y);
        vb = d1 > D - 1 ? 0.f : (d0 == base ? pr.y : pr.x);
      } else {
        va = d0 >= 0 ? p[d0] : 0.f;
        vb = d1 <= D - 1 ? p[d1] : 0.f;
      }
      v = va * hd + vb * sm.ld;
      if (taps) { taps[2 * k] = va; taps[2 * k + 1] = vb; }
    } else if (taps) {
      taps[2 * k] = 0.f; taps[2 * k + 1] = 0.f;
    }
    sc[k] = v;
  }
  sm.s[0] = sc[0]; sm.s[1] = sc[1]; sm.s[2] = sc[3]; sm.s[3] = sc[2];
  sm.w[0] = sm.in2 ? hh * hw * sc[0] * aw : 0.f;
  sm.w[1] = sm.in2 ? hh * sm.lw * sc[1] * aw : 0.f;
  sm.w[2] = sm.in2 ? sm.lh * hw * sc[2] * aw : 0.f;
  sm.w[3] = sm.in2 ? sm.lh * sm.lw * sc[3] * aw : 0.f;
}

// Same sample from the PAIR-INTERLEAVED depth map dp[h][wq][d][2] (wq = w + 1 in [0, W]):
//   dp[h][wq][d] = (dist[h][wq-1][d] or 0, dist[h][wq][d] or 0)
// so the (w0, w1) x (d0, d1) taps of one image row are 16 contiguous bytes: 2 loads per sample
// instead of 4 (each depth load of a wave touches 64 different cache lines).
typedef float float4_u __attribute__((ext_vector_type(4), aligned(8)));

__device__ __forceinline__ void make_sample_dp(Sample &sm, const float *__restrict__ dp_cam, int H, int W, int D,
                                               float x, float y, float z, float aw) {
  const float h_im = sample_coord(y, (float)H);
  const float w_im = sample_coord(x, (float)W);
  const float d_im = sample_coord(z, (float)D);
  sm.in2 = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  sm.in3 = sm.in2 && d_im > -1.f && d_im < (float)D;
  const float hf = floorf(h_im), wf = floorf(w_im), df = floorf(d_im);
  const int h0 = (int)hf, w0 = (int)wf, d0 = (int)df;
  const int h1 = h0 + 1, w1 = w0 + 1, d1 = d0 + 1;
  sm.lh = h_im - hf; sm.lw = w_im - wf; sm.ld = d_im - df; sm.d0 = d0;
  const float hh = 1.f - sm.lh, hw = 1.f - sm.lw, hd = 1.f - sm.ld;
  const bool okh0 = h0 >= 0, okh1 = h1 <= H - 1, okw0 = w0 >= 0, okw1 = w1 <= W - 1;
  const bool ok[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
  const int px[4] = {h0 * W + w0, h0 * W + w1, h1 * W + w0, h1 * W + w1};
  const int ch0 = min(max(h0, 0), H - 1), ch1 = min(max(h1, 0), H - 1);
  const int cw0 = min(max(w0, 0), W - 1), cw1 = min(max(w1, 0), W - 1);
  const int cpx[4] = {ch0 * W + cw0, ch0 * W + cw1, ch1 * W + cw0, ch1 * W + cw1};
  float sc[4] = {0.f, 0.f, 0.f, 0.f};
  if (sm.in3) {
    const int base = d0 < 0 ? 0 : (d0 > D - 2 ? D - 2 : d0);
    const int wq = w0 + 1;                       // in [0, W] whenever in2 holds
    const bool lo = d0 == base;                  // false only at the two depth borders
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int h = rr ? h1 : h0;
      if (h < 0 || h > H - 1) continue;
      const float4_u t = *reinterpret_cast<const float4_u *>(dp_cam + (((int64_t)h * (W + 1) + wq) * D + base) * 2);
      // t = (w0@base, w1@base, w0@base+1, w1@base+1)
      const float a0 = d0 < 0 ? 0.f : (lo ? t.x : t.z), a1 = d1 > D - 1 ? 0.f : (lo ? t.z : t.x);
      const float b0 = d0 < 0 ? 0.f : (lo ? t.y : t.w), b1 = d1 > D - 1 ? 0.f : (lo ? t.w : t.y);
      sc[rr * 2] = a0 * hd + a1 * sm.ld;
      sc[rr * 2 + 1] = b0 * hd + b1 * sm.ld;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool use = sm.in2 && ok[k];
    sm.off[k] = use ? px[k] : (cpx[k] | kOutside);
    if (!ok[k]) sc[k] = 0.f;
  }
  sm.s[0] = sc[0]; sm.s[1] = sc[1]; sm.s[2] = sc[3]; sm.s[3] = sc[2];
  sm.w[0] = sm.in2 ? hh * hw * sc[0] * aw : 0.f;
  sm.w[1] = sm.in2 ? hh * sm.lw * sc[1] * aw : 0.f;
  sm.w[2] = sm.in2 ? sm.lh * hw * sc[2] * aw : 0.f;
  sm.w[3] = sm.in2 ? sm.lh * sm.lw * sc[3] * aw : 0.f;
}

// ---- LayerNorm over one row of C = 64 * VPL channels held by ONE wave (rows.hip: layer_norm_rows_kernel; level_tail.hip uses the
// same functions on rows that live in LDS, so the fused level tail reproduces the stand-alone kernel bit for bit).  Lane l holds
// channels 4 l .. 4 l + 3 (VPL % 4 == 0: one 16-byte access) or l + 64 j. ----
template <int VPL>
__device__ __forceinline__ int ln_channel(int lane, int j) {
  return VPL % 4 == 0 ? ((j / 4) * 64 + lane) * 4 + (j & 3) : j * 64 + lane;
}
template <int VPL>
__device__ __forceinline__ void ln_row_load(const float *xr, int lane, float (&v)[VPL]) {
  if constexpr (VPL % 4 == 0) {
#pragma unroll
    for (int j = 0; j < VPL / 4; ++j) {
      const float4 t = *reinterpret_cast<const float4 *>(xr + (j * 64 + lane) * 4);
      v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j) v[j] = xr[j * 64 + lane];
  }
}
template <int VPL>
__device__ __forceinline__ void ln_row_store(float *yr, int lane, const float (&y)[VPL]) {
  if constexpr (VPL % 4 == 0) {
#pragma unroll
    for (int j = 0; j < VPL / 4; ++j)
      *reinterpret_cast<float4 *>(yr + (j * 64 + lane) * 4) = make_float4(y[4 * j], y[4 * j + 1], y[4 * j + 2], y[4 * j + 3]);
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j) yr[j * 64 + lane] = y[j];
  }
}
template <int VPL>
__device__ __forceinline__ void ln_row_stats(const float (&v)[VPL], float eps, float &mean, float &rstd) {
  constexpr int C = 64 * VPL;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) s += v[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  mean = s * (1.0f / (float)C);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) { const float d = v[j] - mean; q += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  rstd = rsqrtf(q * (1.0f / (float)C) + eps);
}
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float g, float b) { return (v - mean) * rstd * g + b; }

}  // namespace sgc
