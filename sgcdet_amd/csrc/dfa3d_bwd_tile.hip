// LDS-tiled backward of the 3D deformable attention over a BINNED pair list (training path; gfx950, wave64).
//
// Same gradients as dfa3d_bwd_kernel (dfa3d_bwd.hip; reference: wms_deform_attn_cuda_kernel.cuh:82-159,305-419 and
// ms_depth_score_sample_cuda_kernel.cuh:150-241, merged as TU/multi_scale_3ddeformable_attn_function.py:303-351 does) for
// the shape every SGCDet config trains: one level, items = visible (camera, voxel) pairs.
//
// What is different.  The item kernel scatters every corner contribution with a memory-side float atomic: 16 KB of
// added bytes per pair for grad_value (1.3 GB per launch at the finest config-2 level, at the chip's ~1.3 TB/s atomic
// rate) and 8 single-dword atomics per sample for grad_dist (64 lanes in 64 different rows: ~17x slower per byte,
// MI355X_MICROARCH.md "Global float atomics") -- 2.45 ms of a 17.9-ms training step.  Here the forward's binning is
// reused (sgc_bin_pairs: a camera's pairs grouped by the feature pixel their reference point projects to), so the
// corners of a workgroup's pairs fall into one window of the map:
//   * a workgroup owns (camera, bin) and walks the heads; per head the window's slice of grad_value ([th][tw][Cm] fp32)
//     lives in LDS and takes the corner contributions as LDS atomics (ds_add_f32; rows pitched Cm + 1 floats so that the
//     rows of different pixels start on different banks); the window is flushed ONCE per head with global float atomics
//     (whole 64- / 128-byte head segments per pixel row: the full-rate shape) -- windows of neighbouring bins overlap by
//     their halos, so the flush has to add, but it adds each touched element once per bin instead of once per sample;
//   * grad_dist ([th][tw][D], shared by the heads: dist_heads == 1 on this path) is accumulated in LDS over ALL heads
//     and flushed once per workgroup;
//   * corners outside the window (large learned offsets) fall back to the global atomics of the item kernel, lane by lane;
//   * lanes as in the tiled forward: phase 1 one lane per sample (unit = (pair, head): 4 lanes = its 4 points), phase 2 in
//     the unit's quad -- lane c owns channels 4c .. 4c+3 (+16 for Cm = 32), sample descriptors by DPP quad broadcasts, the
//     per-sample scalars (d/dx, d/dy, d/dattn, d/dscore[4]) reduced over the quad by two DPP steps.
// Float atomics (LDS and global) make grad_value / grad_dist order-dependent in the last bits, like the item kernel and
// like the reference's atomicAdd; grad_loc / grad_attn are written by their owning lane (deterministic).
#include "common.hpp"

namespace sgc {

struct BwdTileParams {
  const float *value;          // [N][S][M][CM]
  const float *dist;           // [N][S][D]
  const float *loc;            // [items][LM][P][3]
  const float *attn;           // [items][LM][P] or null (= 1)
  const int32_t *bin_offset;   // [N * nb + 1]
  const float *grad_out;       // [items][M * CM]
  float *grad_value, *grad_dist;
  float *grad_loc;             // [items][LM][P][3] or null
  float *grad_attn;            // [items][LM][P] or null
  int N, S, H, W, D, M, LM, P;
  int bw, bh, nbx, nby, tw, th, hx, hy;
};

// LDS pointers carry their address space in the type: through the lambda below the compiler no longer proves that a plain float *
// points into LDS and emits flat atomics (the slow path of both memories)
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ void lds_add(lds_float *q, float v) { __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int S> __device__ __forceinline__ float qb(float v) { return dpp_move<S | (S << 2) | (S << 4) | (S << 6)>(v); }
template <int S> __device__ __forceinline__ int qbi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, true);
}

template <int CM, int NW>
__global__ __launch_bounds__(NW * 64) void dfa3d_bwd_tile_kernel(const BwdTileParams p) {
  constexpr int NCH = CM / 16;           // 16-byte chunks of a head row per lane (a unit's 4 lanes cover the row)
  constexpr int CMP = CM + 1;            // LDS row pitch in floats: odd, so the rows of different pixels start on different banks
  constexpr int UPW = 16, NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char bt_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nb = p.nbx * p.nby;
  const int t = blockIdx.x;                                         // (camera, bin)
  const int i0 = p.bin_offset[t], i1 = p.bin_offset[t + 1];
  if (i0 >= i1) return;
  const int cnt = i1 - i0;
  const int n = t / nb, b = t - n * nb;
  const int by = b / p.nbx, bx = b - by * p.nbx;
  const int npx = p.tw * p.th;
  const int x0 = max(0, min(bx * p.bw - p.hx, p.W - p.tw)), y0 = max(0, min(by * p.bh - p.hy, p.H - p.th));
  lds_float *gval = (lds_float *)bt_smem;                           // [npx][CMP]
  lds_float *gdist = gval + npx * CMP;                              // [npx][D]
  for (int i = tid; i < npx * CMP + npx * p.D; i += NT) gval[i] = 0.f;
  const int MC = p.M * CM;
  const float *vcam = p.value + (int64_t)n * p.S * MC;
  float *gvcam = p.grad_value + (int64_t)n * p.S * MC;
  const float *dcam = p.dist + (int64_t)n * p.S * p.D;
  float *gdcam = p.grad_dist + (int64_t)n * p.S * p.D;
  const int ul = lane >> 2, pt = lane & 3, c16 = lane & 3;
  const float fW = (float)p.W, fH = (float)p.H, fD = (float)p.D;
  __syncthreads();

  for (int m = 0; m < p.M; ++m) {
    const int lm = p.LM == 1 ? 0 : m;
    for (int g0 = wid * UPW; g0 < cnt; g0 += NW * UPW) {
      const bool unit_live = g0 + ul < cnt;
      const int item = i0 + min(g0 + ul, cnt - 1);
      // ---------------- phase 1: lane = (unit, point) ----------------
      const bool samp_live = unit_live && pt < p.P;
      const int64_t g = ((int64_t)item * p.LM + lm) * p.P + min(pt, p.P - 1);
      const float x = p.loc[g * 3], y = p.loc[g * 3 + 1], z = p.loc[g * 3 + 2];
      const float aw = p.attn ? p.attn[g] : 1.f;
      Sample sm;
      make_sample(sm, dcam, p.D, p.H, p.W, p.D, x, y, z, 1.f);
      const int h0 = (int)fminf(fmaxf(floorf(sample_coord(y, fH)), -2.f), fH);
      const int w0 = (int)fminf(fmaxf(floorf(sample_coord(x, fW)), -2.f), fW);
      // bit k: corner k (gather order (h0,w0) (h0,w1) (h1,w0) (h1,w1)) lies in the map and the sample passes the 2-D gate
      int okm = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) okm |= (samp_live && sm.off[k] >= 0) ? (1 << k) : 0;
      const float sgx = sm.s[0], sgy = sm.s[1], sgz = sm.s[3], sgw = sm.s[2];      // gather order

      // ---------------- phase 2: the unit's quad; lane c owns channels 4c .. 4c+3 (+16 j) ----------------
      float4 top[NCH];
#pragma unroll
      for (int j = 0; j < NCH; ++j)
        top[j] = unit_live ? *reinterpret_cast<const float4 *>(p.grad_out + (int64_t)item * MC + m * CM + (c16 + 4 * j) * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
      float res[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // of THIS lane's sample: gw, gh, ga, gs[4] (gather order)
      auto sample = [&](const int s, const float lh, const float lw, const float aws, const int sh0, const int sw0, const int som,
                        const float s0, const float s1, const float s2, const float s3) {
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float sg[4] = {s0, s1, s2, s3};
        const float bil[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
        const float dh_c[4] = {-hw, -lw, hw, lw};                   // d(bilinear weight)/dh, /dw: wms_deform_attn_cuda_kernel.cuh:116-150
        const float dw_c[4] = {-hh, hh, -lh, lh};
        float part[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float4 vv[4][NCH];
        int pix[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int hk = min(max(sh0 + (k >> 1), 0), p.H - 1), wk = min(max(sw0 + (k & 1), 0), p.W - 1);
          pix[k] = hk * p.W + wk;
#pragma unroll
          for (int j = 0; j < NCH; ++j)
            vv[k][j] = *reinterpret_cast<const float4 *>(vcam + (int64_t)pix[k] * MC + m * CM + (c16 + 4 * j) * 4);
        }
        float val[NCH][4], ghw[NCH][4], gww[NCH][4], tgv[NCH][4];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const float tv[4] = {top[j].x, top[j].y, top[j].z, top[j].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { val[j][e] = 0.f; ghw[j][e] = 0.f; gww[j][e] = 0.f; tgv[j][e] = tv[e] * aws; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool okk = (som >> k) & 1;
          const float ak = bil[k] * sg[k];
          const int hk = sh0 + (k >> 1), wk = sw0 + (k & 1);
          const int tx = wk - x0, ty = hk - y0;
          const bool inside = ((unsigned)tx < (unsigned)p.tw) & ((unsigned)ty < (unsigned)p.th);
          float gsk = 0.f;
          float add[NCH][4];
#pragma unroll
          for (int j = 0; j < NCH; ++j) {
            const float v4[4] = {vv[k][j].x, vv[k][j].y, vv[k][j].z, vv[k][j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = okk ? v4[e] : 0.f;
              ghw[j][e] += sg[k] * dh_c[k] * v;
              gww[j][e] += sg[k] * dw_c[k] * v;
              gsk += v * bil[k] * tgv[j][e];
              val[j][e] += ak * v;
              add[j][e] = ak * tgv[j][e];
            }
          }
          // the corner's contribution to grad_value: the lanes of a quad share the corner, so the branches diverge between units only
          if (okk) {
            if (inside) {
              lds_float *row = gval + (ty * p.tw + tx) * CMP + c16 * 4;
#pragma unroll
              for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) lds_add(row + 16 * j + e, add[j][e]);
            } else {
              float *row = gvcam + (int64_t)pix[k] * MC + m * CM + c16 * 4;
#pragma unroll
              for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(row + 16 * j + e, add[j][e]);
            }
          }
          part[3 + k] = gsk;
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const float tv[4] = {top[j].x, top[j].y, top[j].z, top[j].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            part[2] += tv[e] * val[j][e];
            part[0] += fW * gww[j][e] * tgv[j][e];
            part[1] += fH * ghw[j][e] * tgv[j][e];
          }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          float r = part[k];
          r += lane_xor(r, 1);
          r += lane_xor(r, 2);
          res[k] = pt == s ? r : res[k];
        }
      };
      sample(0, qb<0>(sm.lh), qb<0>(sm.lw), qb<0>(aw), qbi<0>(h0), qbi<0>(w0), qbi<0>(okm), qb<0>(sgx), qb<0>(sgy), qb<0>(sgz), qb<0>(sgw));
      if (p.P > 1) {
        sample(1, qb<1>(sm.lh), qb<1>(sm.lw), qb<1>(aw), qbi<1>(h0), qbi<1>(w0), qbi<1>(okm), qb<1>(sgx), qb<1>(sgy), qb<1>(sgz), qb<1>(sgw));
        if (p.P > 2) sample(2, qb<2>(sm.lh), qb<2>(sm.lw), qb<2>(aw), qbi<2>(h0), qbi<2>(w0), qbi<2>(okm), qb<2>(sgx), qb<2>(sgy), qb<2>(sgz), qb<2>(sgw));
        if (p.P > 3) sample(3, qb<3>(sm.lh), qb<3>(sm.lw), qb<3>(aw), qbi<3>(h0), qbi<3>(w0), qbi<3>(okm), qb<3>(sgx), qb<3>(sgy), qb<3>(sgz), qb<3>(sgw));
      }

      // ---------------- phase 3: lane = its own sample again: depth-score backward, grad_loc / grad_attn ----------------
      if (samp_live) {
        // scores' gradients back in the reference order (h0,w0) (h0,w1) (h1,w1) (h1,w0)
        const float gs_ref[4] = {res[3], res[4], res[6], res[5]};
        float gz = 0.f;
        if (sm.in3) {
          const int d0 = sm.d0, d1 = d0 + 1;
          const float ld = sm.ld, hd = 1.f - ld;
          const int hs[4] = {h0, h0, h0 + 1, h0 + 1}, ws[4] = {w0, w0 + 1, w0 + 1, w0};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float va = 0.f, vb = 0.f;
            if (hs[k] >= 0 && hs[k] <= p.H - 1 && ws[k] >= 0 && ws[k] <= p.W - 1) {
              const int64_t o = ((int64_t)hs[k] * p.W + ws[k]) * p.D;
              const int tx = ws[k] - x0, ty = hs[k] - y0;
              const bool inside = ((unsigned)tx < (unsigned)p.tw) & ((unsigned)ty < (unsigned)p.th);
              if (d0 >= 0) va = dcam[o + d0];
              if (d1 <= p.D - 1) vb = dcam[o + d1];
              if (gs_ref[k] != 0.f) {
                if (inside) {                                  // explicit branches: one pointer that is LDS or global would make the atomics flat
                  lds_float *gd = gdist + (ty * p.tw + tx) * p.D;
                  if (d0 >= 0) lds_add(gd + d0, hd * gs_ref[k]);
                  if (d1 <= p.D - 1) lds_add(gd + d1, ld * gs_ref[k]);
                } else {
                  float *gd = gdcam + o;
                  if (d0 >= 0) atomicAdd(gd + d0, hd * gs_ref[k]);
                  if (d1 <= p.D - 1) atomicAdd(gd + d1, ld * gs_ref[k]);
                }
              }
            }
            gz += fD * (gs_ref[k] * (vb - va));
          }
        }
        // a sample shared by the channel groups (LM == 1 < M: the geometry sample's one "head" over C channels run as M groups of
        // CM): this lane owns the entry in every iteration of the head loop (same bin, same unit deal) and accumulates it
        const bool first = p.LM != 1 || m == 0;
        if (p.grad_loc) {
          float *gl = p.grad_loc + g * 3;
          gl[0] = first ? res[0] : gl[0] + res[0];
          gl[1] = first ? res[1] : gl[1] + res[1];
          gl[2] = first ? gz : gl[2] + gz;
        }
        if (p.grad_attn) p.grad_attn[g] = first ? res[2] : p.grad_attn[g] + res[2];
      }
    }
    __syncthreads();
    // ---- flush this head's window (and clear it for the next head): whole head segments of a pixel row per wave instruction ----
    {
      constexpr int RPP = NT / CM;                                // window rows per pass
      const int ch = tid % CM;
      int row = tid / CM;
      int ty = row / p.tw, tx = row - ty * p.tw;
      for (; row < npx; row += RPP) {
        const float v = gval[row * CMP + ch];
        if (v != 0.f) {
          gval[row * CMP + ch] = 0.f;
          atomicAdd(gvcam + ((int64_t)(y0 + ty) * p.W + x0 + tx) * MC + m * CM + ch, v);
        }
        tx += RPP;
        while (tx >= p.tw) { tx -= p.tw; ++ty; }
      }
    }
    __syncthreads();
  }
  // ---- flush the depth-gradient window ----
  for (int i = tid; i < npx * p.D; i += NT) {
    const float v = gdist[i];
    if (v != 0.f) {
      const int row = i / p.D, d = i - row * p.D;
      const int ty = row / p.tw, tx = row - ty * p.tw;
      atomicAdd(gdcam + ((int64_t)(y0 + ty) * p.W + x0 + tx) * p.D + d, v);
    }
  }
}

int g_tune_bwd_tile_nw = 8;     // waves per workgroup of the tiled backward (4 | 8 | 16)

}  // namespace sgc

using namespace sgc;

template <int CM, int NW>
static int launch_bwd_tile(const BwdTileParams &p, size_t smem, hipStream_t st) {
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void *)dfa3d_bwd_tile_kernel<CM, NW>, 160 * 1024, attr_done);
  hipLaunchKernelGGL((dfa3d_bwd_tile_kernel<CM, NW>), dim3((unsigned)(p.N * p.nbx * p.nby)), dim3(NW * 64), smem, st, p);
  return check_launch("dfa3d_bwd_tile_kernel");
}

extern "C" int64_t sgc_dfa3d_backward_binned_lds_bytes(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y) {
  if (H <= 0 || W <= 0 || Cm <= 0 || D <= 0 || bin_w <= 0 || bin_h <= 0 || halo_x < 0 || halo_y < 0) return 0;
  const int64_t tw = bin_w + 2 * halo_x < W ? bin_w + 2 * halo_x : W, th = bin_h + 2 * halo_y < H ? bin_h + 2 * halo_y : H;
  return tw * th * (Cm + 1 + D) * 4;
}

extern "C" int sgc_dfa3d_backward_binned(const float *value, const float *dist, const float *loc3, const float *attn_or_null,
                                         const int32_t *bin_offset, const float *grad_out, float *grad_value, float *grad_dist,
                                         float *grad_loc3_or_null, float *grad_attn_or_null, int N, int S, int H, int W, int M, int Cm,
                                         int D, int loc_heads, int P, int bin_w, int bin_h, int halo_x, int halo_y,
                                         sgc_stream_t stream) {
  if (!value || !dist || !loc3 || !bin_offset || !grad_out || !grad_value || !grad_dist)
    return set_error(SGC_EINVAL, "sgc_dfa3d_backward_binned: null pointer");
  if (N <= 0 || S < H * W || H <= 0 || W <= 0 || M <= 0 || D < 1 || bin_w <= 0 || bin_h <= 0 || halo_x < 0 || halo_y < 0)
    return set_error(SGC_EINVAL, "sgc_dfa3d_backward_binned: bad size");
  if ((Cm != 16 && Cm != 32) || P < 1 || P > 4 || (loc_heads != 1 && loc_heads != M))
    return set_error(SGC_EUNSUP, "sgc_dfa3d_backward_binned: Cm in {16, 32}, 1 <= P <= 4, loc_heads in {1, M} (got Cm %d, P %d, loc_heads %d)", Cm, P, loc_heads);
  if (((uintptr_t)value | (uintptr_t)grad_out) & 15) return set_error(SGC_EINVAL, "sgc_dfa3d_backward_binned: value / grad_out must be 16-byte aligned");
  const int64_t lds = sgc_dfa3d_backward_binned_lds_bytes(H, W, Cm, D, bin_w, bin_h, halo_x, halo_y);
  if (lds > 160 * 1024) return set_error(SGC_EUNSUP, "sgc_dfa3d_backward_binned: the window needs %lld bytes of LDS", (long long)lds);
  BwdTileParams p = {};
  p.value = value; p.dist = dist; p.loc = loc3; p.attn = attn_or_null; p.bin_offset = bin_offset; p.grad_out = grad_out;
  p.grad_value = grad_value; p.grad_dist = grad_dist; p.grad_loc = grad_loc3_or_null; p.grad_attn = grad_attn_or_null;
  p.N = N; p.S = S; p.H = H; p.W = W; p.D = D; p.M = M; p.LM = loc_heads; p.P = P;
  p.bw = bin_w; p.bh = bin_h; p.nbx = ceil_div(W, bin_w); p.nby = ceil_div(H, bin_h);
  p.hx = halo_x; p.hy = halo_y;
  p.tw = bin_w + 2 * halo_x < W ? bin_w + 2 * halo_x : W;
  p.th = bin_h + 2 * halo_y < H ? bin_h + 2 * halo_y : H;
  hipStream_t st = (hipStream_t)stream;
  const int nw = g_tune_bwd_tile_nw == 4 ? 4 : g_tune_bwd_tile_nw == 16 ? 16 : 8;
#define SGC_BT_CASE(CMV, NWV) if (Cm == CMV && nw == NWV) return launch_bwd_tile<CMV, NWV>(p, (size_t)lds, st)
  SGC_BT_CASE(32, 8); SGC_BT_CASE(16, 8); SGC_BT_CASE(32, 4); SGC_BT_CASE(16, 4); SGC_BT_CASE(32, 16); SGC_BT_CASE(16, 16);
#undef SGC_BT_CASE
  return set_error(SGC_EUNSUP, "sgc_dfa3d_backward_binned: no kernel for this shape");
}
