// Backward of the 3D deformable attention for gfx950 (wave64).
//
// One kernel does what the reference splits over wms_deform_attn_backward (blocks of Cm =
// 16/32 threads: a quarter/half-filled wave64) and ms_depth_score_sample_backward (4-thread
// blocks), TU/multi_scale_3ddeformable_attn_function.py:303-351:
//
//   phase 1  one lane per sample: rebuild the sample (location, scores, corner indices)
//            and park it in LDS;
//   phase 2  one lane per 4 channels of a head: gather the 4 corner rows, scatter
//            grad_value with float atomics (memory-side on gfx950), and reduce the
//            per-sample scalars (d/dx, d/dy, d/dattn, d/dscore[4]) over the head's lanes
//            with wave shuffles instead of the reference's LDS + thread-0 serial loop
//            (wms_deform_attn_cuda_kernel.cuh:377-407);
//   phase 3  one lane per sample again: depth-score backward (grad_dist atomics, d/dz),
//            write grad_loc / grad_attn (or grad_score for the split `_ext` operator).
//
// Float atomics make grad_value / grad_dist order-dependent in the last bits, exactly like
// the reference (THC atomicAdd).
#include "common.hpp"

namespace sgc {

struct BwdParams {
  const float *value, *dist;
  const int64_t *shapes;  // [L,3] (fused) or [L,2] (split)
  int shape_stride;       // 3 or 2
  const int64_t *lsi;
  const float *loc;       // [items,M,L,P,loc_stride]
  int loc_stride;         // 3 or 2
  const float *attn;      // may be null (= 1)
  const float *score_in;  // split mode: precomputed depth scores (reference order)
  const float *grad_out;
  float *grad_value, *grad_dist, *grad_loc, *grad_attn, *grad_score;
  int grad_loc_stride;    // 3 (fused) or 2 (split)
  const int32_t *item_batch;  // optional per-item batch index (pair lists); null -> item / Q
  int S, M, Cm, D, dist_heads, L, Q, P;
  int n_items, TP;
  int fused;              // 1: depth score evaluated / back-propagated in-kernel
};

struct SampleRec {   // 96 B per sample in LDS
  int4 off;          // corner pixel indices (level-relative + lsi), -1 = outside
  float4 sg;         // depth scores in GATHER order (h0,w0)(h0,w1)(h1,w0)(h1,w1)
  float4 misc;       // lh, lw, aw, in2
  float2 wh;         // (float)W, (float)H
  float res[7];      // gw, gh, ga, gs[4] (gather order) -- reduced over channels
  float pad_[3];
};

__device__ __forceinline__ void decode_sample(const BwdParams &p, int item, int r, int &b, int &m, int &l,
                                              int &H, int &W, int &D, int &lvl0, float &x, float &y, float &z,
                                              float &aw, int64_t &g) {
  const int LP = p.L * p.P;
  m = r / LP;
  const int lp = r - m * LP;
  l = lp / p.P;
  b = p.item_batch ? p.item_batch[item] : item / p.Q;
  H = (int)p.shapes[l * p.shape_stride];
  W = (int)p.shapes[l * p.shape_stride + 1];
  D = p.shape_stride == 3 ? (int)p.shapes[l * 3 + 2] : p.D;
  lvl0 = (int)p.lsi[l];
  g = (int64_t)item * (p.M * LP) + r;
  x = p.loc[g * p.loc_stride];
  y = p.loc[g * p.loc_stride + 1];
  z = p.loc_stride == 3 ? p.loc[g * 3 + 2] : 0.f;
  aw = p.attn ? p.attn[g] : 1.f;
}

template <int VEC>
__global__ __launch_bounds__(256) void dfa3d_bwd_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPI = p.M * p.L * p.P;
  SampleRec *rec = reinterpret_cast<SampleRec *>(smem_raw);
  int *lds_b = reinterpret_cast<int *>(smem_raw + (size_t)p.TP * SPI * sizeof(SampleRec));

  const int ntiles = (p.n_items + p.TP - 1) / p.TP;
  if ((int)blockIdx.x >= ntiles) return;
  const int tile = xcd_tile(blockIdx.x, ntiles);
  const int item0 = tile * p.TP;
  const int tid = threadIdx.x;
  const int MC = p.M * p.Cm;
  const int nsamp = p.TP * SPI;

  // ---- phase 1 ----
  for (int t = tid; t < nsamp; t += blockDim.x) {
    const int il = t / SPI, r = t - il * SPI;
    const int item = item0 + il;
    SampleRec &R = rec[t];
#pragma unroll
    for (int k = 0; k < 7; ++k) R.res[k] = 0.f;
    if (item >= p.n_items) {
      R.misc = make_float4(0.f, 0.f, 0.f, 0.f); R.off = make_int4(-1, -1, -1, -1);
      R.sg = make_float4(0.f, 0.f, 0.f, 0.f); R.wh = make_float2(0.f, 0.f);
      if (r == 0) lds_b[il] = 0;
      continue;
    }
    int b, m, l, H, W, D, lvl0; float x, y, z, aw; int64_t g;
    decode_sample(p, item, r, b, m, l, H, W, D, lvl0, x, y, z, aw, g);
    Sample sm;
    if (p.fused) {
      const int dh = p.dist_heads == 1 ? 0 : m;
      const float *dpx = p.dist + (((int64_t)b * p.S + lvl0) * p.dist_heads + dh) * p.D;
      make_sample(sm, dpx, (int64_t)p.dist_heads * p.D, H, W, D, x, y, z, 1.f);
    } else {
      // geometry only (no depth), scores come from the caller
      const float h_im = sample_coord(y, (float)H), w_im = sample_coord(x, (float)W);
      sm.in2 = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
      sm.lh = h_im - hf; sm.lw = w_im - wf;
      const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w1 <= W - 1, h1 <= H - 1 && w0 >= 0, h1 <= H - 1 && w1 <= W - 1};
      const int px[4] = {h0 * W + w0, h0 * W + w1, h1 * W + w0, h1 * W + w1};
#pragma unroll
      for (int k = 0; k < 4; ++k) sm.off[k] = (sm.in2 && ok[k]) ? px[k] : -1;
      const float4 sc = reinterpret_cast<const float4 *>(p.score_in)[g];
      sm.s[0] = sc.x; sm.s[1] = sc.y; sm.s[2] = sc.z; sm.s[3] = sc.w;
    }
    R.off = make_int4(sm.off[0] < 0 ? -1 : sm.off[0] + lvl0, sm.off[1] < 0 ? -1 : sm.off[1] + lvl0,
                      sm.off[2] < 0 ? -1 : sm.off[2] + lvl0, sm.off[3] < 0 ? -1 : sm.off[3] + lvl0);  // < 0 = outside
    R.sg = make_float4(sm.s[0], sm.s[1], sm.s[3], sm.s[2]);
    R.misc = make_float4(sm.lh, sm.lw, aw, sm.in2 ? 1.f : 0.f);
    R.wh = make_float2((float)W, (float)H);
    if (r == 0) lds_b[il] = b;
  }
  __syncthreads();

  // ---- phase 2 ----
  const int CV = p.Cm / VEC;
  const int LPI = p.M * CV;
  const int LP = p.L * p.P;
  // shuffle-reduce width: the lanes of one (item, head) group that sit in one wave
  int RW = CV < kWave ? CV : kWave;
  const bool pow2 = (RW & (RW - 1)) == 0 && (CV % RW) == 0;
  if (!pow2) RW = 1;
  const int span = ((p.TP * LPI + kWave - 1) / kWave) * kWave;
  for (int idx = tid; idx < span; idx += blockDim.x) {
    const bool in_tile = idx < p.TP * LPI;
    const int id = in_tile ? idx : p.TP * LPI - 1;
    const int il = id / LPI;
    const int item = item0 + il;
    const bool live = in_tile && item < p.n_items;
    const int r = id - il * LPI;
    const int m = r / CV;
    const int c0 = (r - m * CV) * VEC;
    const int64_t boff = (int64_t)lds_b[il] * p.S * MC + m * p.Cm + c0;
    const float *vbase = p.value + boff;
    float *gbase = p.grad_value + boff;
    float top[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) top[v] = live ? p.grad_out[(int64_t)item * MC + m * p.Cm + c0 + v] : 0.f;
    for (int s = 0; s < LP; ++s) {
      SampleRec &R = rec[il * SPI + m * LP + s];
      const float lh = R.misc.x, lw = R.misc.y, aw = R.misc.z;
      const bool in2 = live && R.misc.w != 0.f;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const int ok[4] = {R.off.x, R.off.y, R.off.z, R.off.w};
      const float sg[4] = {R.sg.x, R.sg.y, R.sg.z, R.sg.w};
      const float bil[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
      // d(bilinear weight)/dh and /dw per corner, signs as wms_deform_attn_cuda_kernel.cuh:116-150
      const float dh_c[4] = {-hw, -lw, hw, lw};
      const float dw_c[4] = {-hh, hh, -lh, lh};
      float part[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (in2) {
        float val[VEC], ghw[VEC], gww[VEC], tgv[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { val[v] = 0.f; ghw[v] = 0.f; gww[v] = 0.f; tgv[v] = top[v] * aw; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (ok[k] < 0) continue;
          const int64_t o = (int64_t)ok[k] * MC;
          float vv[VEC];
          if (VEC == 4) {
            const float4 x4 = *reinterpret_cast<const float4 *>(vbase + o);
            vv[0] = x4.x; vv[1 % VEC] = x4.y; vv[2 % VEC] = x4.z; vv[3 % VEC] = x4.w;
          } else {
            vv[0] = vbase[o];
          }
          const float ak = bil[k] * sg[k];
          float gsk = 0.f;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            ghw[v] += sg[k] * dh_c[k] * vv[v];
            gww[v] += sg[k] * dw_c[k] * vv[v];
            if (VEC == 1) atomicAdd(gbase + o + v, ak * tgv[v]);        // VEC == 4: phase 2b scatters whole rows
            gsk += vv[v] * bil[k] * tgv[v];
            val[v] += ak * vv[v];
          }
          part[3 + k] = gsk;
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          part[2] += top[v] * val[v];
          part[0] += R.wh.x * gww[v] * tgv[v];
          part[1] += R.wh.y * ghw[v] * tgv[v];
        }
      }
      // reduce over the RW lanes of this (item, head) group, then one LDS atomic per wave-group
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        float x = part[k];
        for (int o = 1; o < RW; o <<= 1) x += __shfl_xor(x, o);
        part[k] = x;
      }
      if (in_tile && ((r - m * CV) % RW) == 0) {
        if (RW == CV) {                 // the whole (item, head) group sits in this wave: the only writer
#pragma unroll
          for (int k = 0; k < 7; ++k) R.res[k] = part[k];
        } else {
#pragma unroll
          for (int k = 0; k < 7; ++k)
            if (part[k] != 0.f) atomicAdd(&R.res[k], part[k]);
        }
      }
    }
  }
  // ---- phase 2b: grad_value scatter, one wave per item, lanes along the channels of the pixel row ----
  // Device-scope float atomics are served memory-side in 64-byte requests (the L2s of the 8 XCDs are not coherent
  // with each other); with the lane = (head, 4 channels) deal of phase 2 every wave instruction touched 4 dwords
  // of each of its 16 requests.  Here instruction j of a lane adds channel 64 j + lane: 16 useful dwords per
  // request, a quarter of the requests (measured on the config-2 finest level: 4.9 -> see DESIGN.md 4.3).
  if (VEC == 4) {
    const int lane = tid & (kWave - 1), wv = tid / kWave, nwv = blockDim.x / kWave;
    for (int il = wv; il < p.TP; il += nwv) {
      const int item = item0 + il;
      if (item >= p.n_items) break;
      float *grow = p.grad_value + (int64_t)lds_b[il] * p.S * MC;
      for (int c = lane; c < MC; c += kWave) {
        const float t = p.grad_out[(int64_t)item * MC + c];
        if (!__builtin_amdgcn_ballot_w64(t != 0.f)) continue;      // padded rows of the rebatch: nothing to add
        const int m = c / p.Cm;
        for (int s2 = 0; s2 < LP; ++s2) {
          const SampleRec &R = rec[il * SPI + m * LP + s2];
          if (R.misc.w == 0.f) continue;
          const float lh = R.misc.x, lw = R.misc.y, tg = t * R.misc.z;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const int ok[4] = {R.off.x, R.off.y, R.off.z, R.off.w};
          const float ak[4] = {hh * hw * R.sg.x, hh * lw * R.sg.y, lh * hw * R.sg.z, lh * lw * R.sg.w};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ok[k] >= 0) atomicAdd(grow + (int64_t)ok[k] * MC + c, ak[k] * tg);
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 3 ----
  for (int t = tid; t < nsamp; t += blockDim.x) {
    const int il = t / SPI, r = t - il * SPI;
    const int item = item0 + il;
    if (item >= p.n_items) continue;
    const SampleRec &R = rec[t];
    int b, m, l, H, W, D, lvl0; float x, y, z, aw; int64_t g;
    decode_sample(p, item, r, b, m, l, H, W, D, lvl0, x, y, z, aw, g);
    // scores' gradients back in the reference order (h0,w0) (h0,w1) (h1,w1) (h1,w0)
    const float gs_ref[4] = {R.res[3], R.res[4], R.res[6], R.res[5]};
    if (p.grad_attn) p.grad_attn[g] = R.res[2];
    float gz = 0.f;
    if (p.fused) {
      const float h_im = sample_coord(y, (float)H), w_im = sample_coord(x, (float)W), d_im = sample_coord(z, (float)D);
      const bool in3 = h_im > -1.f && w_im > -1.f && d_im > -1.f && h_im < (float)H && w_im < (float)W && d_im < (float)D;
      if (in3) {
        const float hf = floorf(h_im), wf = floorf(w_im), df = floorf(d_im);
        const int h0 = (int)hf, w0 = (int)wf, d0 = (int)df, h1 = h0 + 1, w1 = w0 + 1, d1 = d0 + 1;
        const float ld = d_im - df, hd = 1.f - ld;
        const int hs[4] = {h0, h0, h1, h1}, ws[4] = {w0, w1, w1, w0};
        const int dh = p.dist_heads == 1 ? 0 : m;
        const int64_t doff = (((int64_t)b * p.S + lvl0) * p.dist_heads + dh) * p.D;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float va = 0.f, vb = 0.f;
          if (hs[k] >= 0 && hs[k] <= H - 1 && ws[k] >= 0 && ws[k] <= W - 1) {
            const int64_t o = doff + ((int64_t)hs[k] * W + ws[k]) * p.dist_heads * p.D;
            if (d0 >= 0) { va = p.dist[o + d0]; if (gs_ref[k] != 0.f) atomicAdd(p.grad_dist + o + d0, hd * gs_ref[k]); }
            if (d1 <= D - 1) { vb = p.dist[o + d1]; if (gs_ref[k] != 0.f) atomicAdd(p.grad_dist + o + d1, ld * gs_ref[k]); }
          }
          gz += (float)D * (gs_ref[k] * (vb - va));
        }
      }
    } else if (p.grad_score) {
      reinterpret_cast<float4 *>(p.grad_score)[g] = make_float4(gs_ref[0], gs_ref[1], gs_ref[2], gs_ref[3]);
    }
    float *gl = p.grad_loc + g * p.grad_loc_stride;
    gl[0] = R.res[0];
    gl[1] = R.res[1];
    if (p.grad_loc_stride == 3) gl[2] = gz;
  }
}

static int launch_bwd(BwdParams p, hipStream_t stream) {
  const int SPI = p.M * p.L * p.P;
  if ((int64_t)SPI * sizeof(SampleRec) > 60000) return set_error(SGC_EUNSUP, "backward: M*L*P = %d samples per query exceed the LDS tile", SPI);
  int tp = 256 / SPI;
  if (tp < 1) tp = 1;
  if (SPI <= 2 && tp > 32) tp = 32;
  while (tp > 1 && (int64_t)tp * SPI * sizeof(SampleRec) > 49152) tp >>= 1;
  p.TP = tp;
  const bool vec4 = (p.Cm % 4 == 0) && !((uintptr_t)p.value & 15) && !((uintptr_t)p.grad_out & 15);
  const size_t smem = (size_t)tp * SPI * sizeof(SampleRec) + tp * sizeof(int);
  const int grid = ceil_div(p.n_items, tp);
  if (grid <= 0) return SGC_OK;
  if (vec4) hipLaunchKernelGGL(dfa3d_bwd_kernel<4>, dim3(grid), dim3(256), smem, stream, p);
  else hipLaunchKernelGGL(dfa3d_bwd_kernel<1>, dim3(grid), dim3(256), smem, stream, p);
  return check_launch("dfa3d_bwd_kernel");
}

// split operator: depth score backward, one lane per sample
__global__ void depth_score_bwd_kernel(const float *__restrict__ dist, const int64_t *__restrict__ shapes3,
                                       const int64_t *__restrict__ lsi, const float *__restrict__ loc3,
                                       const float *__restrict__ grad_score, float *__restrict__ grad_dist,
                                       float *__restrict__ grad_loc3, int64_t total, int S, int M, int D, int L,
                                       int Q, int P) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)((g / P) % L);
    const int m = (int)((g / ((int64_t)P * L)) % M);
    const int b = (int)(g / ((int64_t)P * L * M) / Q);
    const int H = (int)shapes3[l * 3], W = (int)shapes3[l * 3 + 1], Dl = (int)shapes3[l * 3 + 2];
    const float x = loc3[g * 3], y = loc3[g * 3 + 1], z = loc3[g * 3 + 2];
    const float h_im = sample_coord(y, (float)H), w_im = sample_coord(x, (float)W), d_im = sample_coord(z, (float)Dl);
    float gz = 0.f;
    if (h_im > -1.f && w_im > -1.f && d_im > -1.f && h_im < (float)H && w_im < (float)W && d_im < (float)Dl) {
      const float hf = floorf(h_im), wf = floorf(w_im), df = floorf(d_im);
      const int h0 = (int)hf, w0 = (int)wf, d0 = (int)df, h1 = h0 + 1, w1 = w0 + 1, d1 = d0 + 1;
      const float ld = d_im - df, hd = 1.f - ld;
      const int hs[4] = {h0, h0, h1, h1}, ws[4] = {w0, w1, w1, w0};
      const float4 gs4 = reinterpret_cast<const float4 *>(grad_score)[g];
      const float gs[4] = {gs4.x, gs4.y, gs4.z, gs4.w};
      const int64_t doff = (((int64_t)b * S + lsi[l]) * M + m) * D;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float va = 0.f, vb = 0.f;
        if (hs[k] >= 0 && hs[k] <= H - 1 && ws[k] >= 0 && ws[k] <= W - 1) {
          const int64_t o = doff + ((int64_t)hs[k] * W + ws[k]) * M * D;
          if (d0 >= 0) { va = dist[o + d0]; atomicAdd(grad_dist + o + d0, hd * gs[k]); }
          if (d1 <= Dl - 1) { vb = dist[o + d1]; atomicAdd(grad_dist + o + d1, ld * gs[k]); }
        }
        gz += (float)Dl * (gs[k] * (vb - va));
      }
    }
    grad_loc3[g * 3] = 0.f;      // uv gradient through the score is dropped (kernel.cuh:238-239)
    grad_loc3[g * 3 + 1] = 0.f;
    grad_loc3[g * 3 + 2] = gz;
  }
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_dfa3d_backward(const float *value, const float *dist, const int64_t *shapes3,
                                  const int64_t *lsi, const float *loc3, const float *attn,
                                  const float *grad_out, float *grad_value, float *grad_dist,
                                  float *grad_loc3, float *grad_attn_or_null,
                                  int B, int S, int M, int Cm, int D, int dist_heads,
                                  int L, int Q, int P, sgc_stream_t stream) {
  if (!value || !dist || !shapes3 || !lsi || !loc3 || !grad_out || !grad_value || !grad_dist || !grad_loc3)
    return set_error(SGC_EINVAL, "sgc_dfa3d_backward: null pointer");
  if (dist_heads != 1 && dist_heads != M) return set_error(SGC_EINVAL, "sgc_dfa3d_backward: dist_heads must be 1 or M");
  if ((int64_t)B * Q >= (1ll << 31)) return set_error(SGC_EUNSUP, "sgc_dfa3d_backward: B*Q >= 2^31");
  BwdParams p = {};
  p.value = value; p.dist = dist; p.shapes = shapes3; p.shape_stride = 3; p.lsi = lsi; p.loc = loc3; p.loc_stride = 3;
  p.attn = attn; p.grad_out = grad_out; p.grad_value = grad_value; p.grad_dist = grad_dist; p.grad_loc = grad_loc3;
  p.grad_attn = grad_attn_or_null; p.grad_loc_stride = 3;
  p.S = S; p.M = M; p.Cm = Cm; p.D = D; p.dist_heads = dist_heads; p.L = L; p.Q = Q; p.P = P;
  p.n_items = B * Q; p.fused = 1;
  return launch_bwd(p, (hipStream_t)stream);
}

extern "C" int sgc_dfa3d_backward_items(const float *value, const float *dist, const int64_t *shapes3,
                                        const int64_t *lsi, const float *loc3, const float *attn_or_null,
                                        const int32_t *item_batch, const float *grad_out, float *grad_value, float *grad_dist,
                                        float *grad_loc3, float *grad_attn_or_null,
                                        int B, int S, int M, int Cm, int D, int dist_heads,
                                        int L, int n_items, int P, sgc_stream_t stream) {
  if (!value || !dist || !shapes3 || !lsi || !loc3 || !item_batch || !grad_out || !grad_value || !grad_dist || !grad_loc3)
    return set_error(SGC_EINVAL, "sgc_dfa3d_backward_items: null pointer");
  if (dist_heads != 1 && dist_heads != M) return set_error(SGC_EINVAL, "sgc_dfa3d_backward_items: dist_heads must be 1 or M");
  if (B <= 0 || n_items < 0) return set_error(SGC_EINVAL, "sgc_dfa3d_backward_items: bad size");
  if (n_items == 0) return SGC_OK;
  BwdParams p = {};
  p.value = value; p.dist = dist; p.shapes = shapes3; p.shape_stride = 3; p.lsi = lsi; p.loc = loc3; p.loc_stride = 3;
  p.attn = attn_or_null; p.grad_out = grad_out; p.grad_value = grad_value; p.grad_dist = grad_dist; p.grad_loc = grad_loc3;
  p.grad_attn = grad_attn_or_null; p.grad_loc_stride = 3; p.item_batch = item_batch;
  p.S = S; p.M = M; p.Cm = Cm; p.D = D; p.dist_heads = dist_heads; p.L = L; p.Q = 1; p.P = P;
  p.n_items = n_items; p.fused = 1;
  return launch_bwd(p, (hipStream_t)stream);
}

extern "C" int sgc_wms_backward(const float *value, const int64_t *shapes2, const int64_t *lsi,
                                const float *loc2, const float *attn, const float *score,
                                const float *grad_out, float *grad_value, float *grad_loc2,
                                float *grad_attn, float *grad_score,
                                int B, int S, int M, int Cm, int L, int Q, int P, sgc_stream_t stream) {
  if (!value || !shapes2 || !lsi || !loc2 || !attn || !score || !grad_out || !grad_value || !grad_loc2 ||
      !grad_attn || !grad_score)
    return set_error(SGC_EINVAL, "sgc_wms_backward: null pointer");
  BwdParams p = {};
  p.value = value; p.shapes = shapes2; p.shape_stride = 2; p.lsi = lsi; p.loc = loc2; p.loc_stride = 2;
  p.attn = attn; p.score_in = score; p.grad_out = grad_out; p.grad_value = grad_value; p.grad_loc = grad_loc2;
  p.grad_attn = grad_attn; p.grad_score = grad_score; p.grad_loc_stride = 2;
  p.S = S; p.M = M; p.Cm = Cm; p.D = 1; p.dist_heads = 1; p.L = L; p.Q = Q; p.P = P;
  p.n_items = B * Q; p.fused = 0;
  return launch_bwd(p, (hipStream_t)stream);
}

extern "C" int sgc_depth_score_backward(const float *dist, const int64_t *shapes3, const int64_t *lsi,
                                        const float *loc3, const float *grad_score,
                                        float *grad_dist, float *grad_loc3,
                                        int B, int S, int M, int D, int L, int Q, int P, sgc_stream_t stream) {
  if (!dist || !shapes3 || !lsi || !loc3 || !grad_score || !grad_dist || !grad_loc3)
    return set_error(SGC_EINVAL, "sgc_depth_score_backward: null pointer");
  const int64_t total = (int64_t)B * Q * M * L * P;
  if (total == 0) return SGC_OK;
  const int grid = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
  hipLaunchKernelGGL(depth_score_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dist, shapes3, lsi, loc3,
                     grad_score, grad_dist, grad_loc3, total, S, M, D, L, Q, P);
  return check_launch("depth_score_bwd_kernel");
}
