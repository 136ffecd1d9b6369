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
// corners of a workgroup's pairs fall into one window of the map, and a workgroup = (camera, bin, head) keeps that
// head's window of grad_value ([th][tw][Cm] fp32) in LDS.
//
// LDS float atomics are NOT the way to fill it: measured (tools/bwd_tile_bench.py, round 6) a ds_add_f32 wave instruction
// takes ~240 cycles -- the first form of this kernel, one ds_add per corner element, ran 2.96 ms against the item
// kernel's 1.88 and 0.58 ms with the adds removed.  So nobody adds atomically; the window has OWNERS:
//   phase A  (lanes as in the tiled forward: a unit = (pair, head) is a quad, lane = (unit, point) for the sample
//            arithmetic, lane c = channels 4c .. 4c+3 (+16) for the row arithmetic)  every sample's descriptor -- LDS
//            offsets of its two window rows and the four corner weights bil * score * attn -- goes to a queue in LDS,
//            the unit's grad_out row beside it; the per-sample scalars (d/dx, d/dy, d/dattn, d/dscore[4]) are reduced
//            over the quad by two DPP steps and finished by the sample's own lane (depth-score backward, grad_loc /
//            grad_attn); corners that are in the map but outside the window take the item kernel's global atomic;
//   phase B  window row y belongs to wave y % NW: phase A appends each sample's two row entries to the OWNER's queue (one sub-queue
//            per (owner, producing wave), slots by ballot + popcount: no atomics), and every wave walks only its own entries
//            (wave-uniform control flow): lane = (pixel of the row pair, channel), a plain ds_read / fma / ds_write of 2 x Cm
//            consecutive floats -- no conflicts inside the instruction (two adjacent pixels), none between waves (disjoint
//            rows); a wave's own accesses to one address stay in program order (two entries are in flight together only when
//            their pixel pairs do not overlap).
// The window is flushed ONCE with global float atomics (whole 64- / 128-byte head segments per pixel row: the full-rate
// shape): windows of neighbouring bins overlap by their halos, so the flush has to add, but it adds each touched element
// once per (bin, head) instead of once per sample.  grad_dist ([th][tw][D] per workgroup; dist_heads == 1) takes the few
// depth atomics (8 per sample against 512 row elements) in LDS and is flushed the same way.
// Float atomics make grad_value / grad_dist order-dependent in the last bits, like the item kernel and like the
// reference's atomicAdd; grad_loc / grad_attn are written by their owning lane (deterministic) -- except for a sample set
// shared by the channel groups (loc_heads == 1 < M), whose gradients are summed over the groups' workgroups atomically.
#include <stdlib.h>

#include "common.hpp"

namespace sgc {

struct BwdTileParams {
  const float *value;          // [N][S][M][CM]
  const float *dist;           // [N][S][D]
  const float *loc;            // [items][LM][P][3]
  const float *attn;           // [items][LM][P] or null (= 1)
  const int32_t *bin_offset;   // [N * nb + 1]
  const int32_t *head_shift;   // [M][2] window shift per head in pixels (x, y), or null: the window follows the head's mean sampling offset
  const float *grad_out;       // [items][M * CM]
  float *grad_value, *grad_dist;
  float *grad_loc;             // [items][LM][P][3] or null
  float *grad_attn;            // [items][LM][P] or null
  int N, S, H, W, D, M, LM, P;
  int bw, bh, nbx, nby, tw, th, hx, hy;
  int diag;                    // TIMING ONLY (sgc_set_tuning "bwd_tile_diag", honoured with SGC_DIAG=1): 1 no phase B, 2 no flush,
                               // 4 no value-row loads, 8 no depth atomics, 16 no row arithmetic in phase A
};

// LDS pointers carry their address space in the type: through a lambda the compiler no longer proves that a plain float *
// points into LDS and emits flat atomics (the slow path of both memories)
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void lds_add(lds_float *q, float v) { __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int S> __device__ __forceinline__ float qb(float v) { return dpp_move<S | (S << 2) | (S << 4) | (S << 6)>(v); }
template <int S> __device__ __forceinline__ int qbi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, true);
}

constexpr int kBtQueue = 64;   // entries per (owner, producer) sub-queue: a lane's two rows have different owners, so a wave step adds <= 64
typedef int BtEntry __attribute__((ext_vector_type(4)));     // x: window float index of (row, first pixel, channel 0); y, z: bits of the
                                                             // weights of the two adjacent pixels; w: unit slot of the batch

// (Measured and not kept, tools/bwd_tile_bench.py on the finest config-2 level: eight waves per workgroup -- 2.7 ms against 1.57 with
//  four, in every form of phase B; producer and owner roles on different waves with double-buffered queues -- 2.7 ms: a bin holds
//  ~70 pairs, i.e. one full batch and a remainder, so there is nothing to overlap.)
template <int CM, int NW>
__global__ __launch_bounds__(NW * 64) void dfa3d_bwd_tile_kernel(const BwdTileParams p) {
  constexpr int NCH = CM / 16;           // 16-byte chunks of a head row per lane (a unit's 4 lanes cover the row)
  constexpr int UPW = 16, NT = NW * 64, UB = NW * UPW;       // units per wave step / per batch of the workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char bt_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nb = p.nbx * p.nby;
  const int m = blockIdx.x % p.M, t = blockIdx.x / p.M;             // (camera, bin), head innermost: a bin's heads run side by side
  const int i0 = p.bin_offset[t], i1 = p.bin_offset[t + 1];
  if (i0 >= i1) return;
  const int cnt = i1 - i0;
  const int n = t / nb, b = t - n * nb;
  const int by = b / p.nbx, bx = b - by * p.nbx;
  const int npx = p.tw * p.th;
  // the window of head m is shifted by that head's mean sampling offset (as in the tiled forward): the samples of a head sit around
  // reference pixel + offset, and without the shift a third of the corners fell outside a 2-pixel halo (the global-atomic fall-back)
  const int sx = p.head_shift ? p.head_shift[m * 2] : 0, sy = p.head_shift ? p.head_shift[m * 2 + 1] : 0;
  const int x0 = max(0, min(bx * p.bw - p.hx + sx, p.W - p.tw)), y0 = max(0, min(by * p.bh - p.hy + sy, p.H - p.th));
  lds_float *win = (lds_float *)bt_smem;                            // [npx][CM]
  lds_float *gdist = win + npx * CM;                                // [npx][D]
  lds_float *tops = gdist + npx * p.D;                              // [UB][CM]: grad_out rows of the batch's units
  typedef __attribute__((address_space(3))) BtEntry lds_entry;
  lds_entry *queue = (lds_entry *)(win + ((npx * (CM + p.D) + UB * CM + 3) & ~3));     // [owner][producer][kBtQueue], 16-byte aligned
  lds_int *qcount = (lds_int *)(queue + NW * NW * kBtQueue);        // [owner][producer]
  for (int i = tid; i < npx * CM + npx * p.D; i += NT) win[i] = 0.f;
  const int MC = p.M * CM;
  const float *vcam = p.value + (int64_t)n * p.S * MC;
  float *gvcam = p.grad_value + (int64_t)n * p.S * MC;
  const float *dcam = p.dist + (int64_t)n * p.S * p.D;
  float *gdcam = p.grad_dist + (int64_t)n * p.S * p.D;
  const int ul = lane >> 2, pt = lane & 3, c16 = lane & 3;
  const float fW = (float)p.W, fH = (float)p.H, fD = (float)p.D;
  const int lm = p.LM == 1 ? 0 : m;
  const bool shared_samples = p.LM == 1 && p.M > 1;                 // gradients of the sample set are summed over the groups' workgroups

  for (int base = 0; base < cnt; base += UB) {
    __syncthreads();                                                // window zeroed / the previous batch's queues are consumed
    // ======================= phase A =======================
    {
      const int us = wid * UPW + ul;                                // unit slot in the batch
      const bool unit_live = base + us < cnt;
      const int item = i0 + min(base + us, cnt - 1);
      const bool samp_live = unit_live && pt < p.P;
      const int64_t g = ((int64_t)item * p.LM + lm) * p.P + min(pt, p.P - 1);
      const float x = p.loc[g * 3], y = p.loc[g * 3 + 1], z = p.loc[g * 3 + 2];
      const float aw = p.attn ? p.attn[g] : 1.f;
      Sample sm;
      float taps[8];                                                // the depth taps of the four corners: phase 3 needs them again
      make_sample(sm, dcam, p.D, p.H, p.W, p.D, x, y, z, 1.f, taps);
      const int h0 = (int)__builtin_amdgcn_fmed3f(floorf(sample_coord(y, fH)), -2.f, fH);
      const int w0 = (int)__builtin_amdgcn_fmed3f(floorf(sample_coord(x, fW)), -2.f, fW);
      // bit k: corner k (gather order (h0,w0) (h0,w1) (h1,w0) (h1,w1)) lies in the map and the sample passes the 2-D gate
      int okm = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) okm |= (samp_live && sm.off[k] >= 0) ? (1 << k) : 0;
      const float sgx = sm.s[0], sgy = sm.s[1], sgz = sm.s[3], sgw = sm.s[2];      // gather order
      // ---- this lane's sample -> the queue.  Window rows ty0 = h0 - y0, ty0 + 1; pixels tx0 = w0 - x0, tx0 + 1.  A row pair is applied
      //      as TWO ADJACENT window pixels starting at bx0 = clamp(tx0, 0, tw - 2): the weights move with the clamp, a corner whose
      //      pixel is outside the window keeps weight 0 here (the row arithmetic below adds it with a global atomic) ----
      {
        const float hh = 1.f - sm.lh, hw = 1.f - sm.lw;
        const float ak[4] = {hh * hw * sgx * aw, hh * sm.lw * sgy * aw, sm.lh * hw * sgz * aw, sm.lh * sm.lw * sgw * aw};
        const int tx0 = w0 - x0, ty0 = h0 - y0;
        const int bx0 = min(max(tx0, 0), p.tw - 2);
        int offs[2], own[2];
        float wq[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int ty = ty0 + r;
          const bool row_in = (unsigned)ty < (unsigned)p.th;
          float w_p0 = 0.f, w_p1 = 0.f;                              // weights of window pixels bx0, bx0 + 1 of this row
          bool any = false;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int k = r * 2 + c, tx = tx0 + c;
            const bool use = row_in && ((okm >> k) & 1) && (unsigned)tx < (unsigned)p.tw;
            if (use && tx == bx0) w_p0 = ak[k];
            if (use && tx == bx0 + 1) w_p1 = ak[k];
            any |= use;
          }
          offs[r] = (any && samp_live) ? (ty * p.tw + bx0) * CM : -1;
          own[r] = ty & (NW - 1);
          wq[r * 2] = w_p0; wq[r * 2 + 1] = w_p1;
        }
        // append to the owners' queues: slot = rank among the wave's lanes that add to the same owner (row 0 entries first)
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int o = 0; o < NW; ++o) {
          const bool a0 = offs[0] >= 0 && own[0] == o, a1 = offs[1] >= 0 && own[1] == o;
          const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
          const int c0 = __popcll(m0);
          lds_entry *q = queue + (o * NW + wid) * kBtQueue;
          if (a0) { const BtEntry e = {offs[0], __float_as_int(wq[0]), __float_as_int(wq[1]), us}; q[__popcll(m0 & lt)] = e; }
          if (a1) { const BtEntry e = {offs[1], __float_as_int(wq[2]), __float_as_int(wq[3]), us}; q[c0 + __popcll(m1 & lt)] = e; }
          if (lane == 0) qcount[o * NW + wid] = c0 + __popcll(m1);
        }
      }

      // ---- row arithmetic in the unit's quad; lane c owns channels 4c .. 4c+3 (+16 j) ----
      float4 top[NCH];
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        top[j] = unit_live ? *reinterpret_cast<const float4 *>(p.grad_out + (int64_t)item * MC + m * CM + (c16 + 4 * j) * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        lds_float *tq = tops + us * CM + (c16 + 4 * j) * 4;
        tq[0] = top[j].x; tq[1] = top[j].y; tq[2] = top[j].z; tq[3] = top[j].w;
      }
      float res[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // of THIS lane's sample: gw, gh, ga, gs[4] (gather order)
      // The corner rows of ALL the unit's samples are requested before the first one is used: a bin holds one or two batches, so a
      // wave passes here once or twice per launch and every dependent round trip to L2 is exposed (measured: the phase took 0.8 ms of a
      // 1.25-ms launch with one sample's rows in flight at a time).  4 samples x 4 corners x NCH 16-byte chunks = 128 registers at Cm = 32.
      float4 vv[4][4][NCH];
      int pixs[4][4];
      auto request = [&](const int s, const int sh0, const int sw0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int hk = min(max(sh0 + (k >> 1), 0), p.H - 1), wk = min(max(sw0 + (k & 1), 0), p.W - 1);
          pixs[s][k] = hk * p.W + wk;
#pragma unroll
          for (int j = 0; j < NCH; ++j)
            vv[s][k][j] = (p.diag & 4) ? make_float4(1.f, 2.f, 3.f, 4.f)
                                       : *reinterpret_cast<const float4 *>(vcam + (int64_t)pixs[s][k] * MC + m * CM + (c16 + 4 * j) * 4);
        }
      };
      auto sample = [&](const int s, const float lh, const float lw, const float aws, const int sh0, const int sw0, const int som,
                        const float s0, const float s1, const float s2, const float s3) {
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float sg[4] = {s0, s1, s2, s3};
        const float bil[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
        const float dh_c[4] = {-hw, -lw, hw, lw};                   // d(bilinear weight)/dh, /dw: wms_deform_attn_cuda_kernel.cuh:116-150
        const float dw_c[4] = {-hh, hh, -lh, lh};
        float part[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
          const int tx = sw0 + (k & 1) - x0, ty = sh0 + (k >> 1) - y0;
          const bool inside = ((unsigned)tx < (unsigned)p.tw) & ((unsigned)ty < (unsigned)p.th);
          float gsk = 0.f;
#pragma unroll
          for (int j = 0; j < NCH; ++j) {
            const float v4[4] = {vv[s][k][j].x, vv[s][k][j].y, vv[s][k][j].z, vv[s][k][j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = okk ? v4[e] : 0.f;
              ghw[j][e] += sg[k] * dh_c[k] * v;
              gww[j][e] += sg[k] * dw_c[k] * v;
              gsk += v * bil[k] * tgv[j][e];
              val[j][e] += ak * v;
            }
          }
          // rare: the corner is in the map but outside the staged window -> the item kernel's global atomic, lane by lane
          if (okk && !inside) {
            float *row = gvcam + (int64_t)pixs[s][k] * MC + m * CM + c16 * 4;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) atomicAdd(row + 16 * j + e, ak * tgv[j][e]);
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
      if (!(p.diag & 16)) {
        request(0, qbi<0>(h0), qbi<0>(w0));
        if (p.P > 1) request(1, qbi<1>(h0), qbi<1>(w0));
        if (p.P > 2) request(2, qbi<2>(h0), qbi<2>(w0));
        if (p.P > 3) request(3, qbi<3>(h0), qbi<3>(w0));
        sample(0, qb<0>(sm.lh), qb<0>(sm.lw), qb<0>(aw), qbi<0>(h0), qbi<0>(w0), qbi<0>(okm), qb<0>(sgx), qb<0>(sgy), qb<0>(sgz), qb<0>(sgw));
        if (p.P > 1) {
          sample(1, qb<1>(sm.lh), qb<1>(sm.lw), qb<1>(aw), qbi<1>(h0), qbi<1>(w0), qbi<1>(okm), qb<1>(sgx), qb<1>(sgy), qb<1>(sgz), qb<1>(sgw));
          if (p.P > 2) sample(2, qb<2>(sm.lh), qb<2>(sm.lw), qb<2>(aw), qbi<2>(h0), qbi<2>(w0), qbi<2>(okm), qb<2>(sgx), qb<2>(sgy), qb<2>(sgz), qb<2>(sgw));
          if (p.P > 3) sample(3, qb<3>(sm.lh), qb<3>(sm.lw), qb<3>(aw), qbi<3>(h0), qbi<3>(w0), qbi<3>(okm), qb<3>(sgx), qb<3>(sgy), qb<3>(sgz), qb<3>(sgw));
        }
      }

      // ---- lane = its own sample again: depth-score backward, grad_loc / grad_attn ----
      if (samp_live) {
        float gz = 0.f;
        if (sm.in3) {
          const int d0 = sm.d0, d1 = d0 + 1;
          const float ld = sm.ld, hd = 1.f - ld;
#pragma unroll
          for (int k = 0; k < 4; ++k) {                             // gather order: res[3 + k] is d/dscore of corner k, taps[2k], [2k+1] its depth taps
            const int hk = h0 + (k >> 1), wk = w0 + (k & 1);
            const float gs = res[3 + k];
            if (hk >= 0 && hk <= p.H - 1 && wk >= 0 && wk <= p.W - 1 && gs != 0.f && !(p.diag & 8)) {
              const int tx = wk - x0, ty = hk - y0;
              const bool inside = ((unsigned)tx < (unsigned)p.tw) & ((unsigned)ty < (unsigned)p.th);
              if (inside) {                                  // explicit branches: one pointer that is LDS or global would make the atomics flat
                lds_float *gd = gdist + (ty * p.tw + tx) * p.D;
                if (d0 >= 0) lds_add(gd + d0, hd * gs);
                if (d1 <= p.D - 1) lds_add(gd + d1, ld * gs);
              } else {
                float *gd = gdcam + ((int64_t)hk * p.W + wk) * p.D;
                if (d0 >= 0) atomicAdd(gd + d0, hd * gs);
                if (d1 <= p.D - 1) atomicAdd(gd + d1, ld * gs);
              }
            }
            gz += fD * (gs * (taps[2 * k + 1] - taps[2 * k]));
          }
        }
        if (p.grad_loc) {
          float *gl = p.grad_loc + g * 3;
          if (shared_samples) { atomicAdd(gl, res[0]); atomicAdd(gl + 1, res[1]); atomicAdd(gl + 2, gz); }      // the caller zero-filled it
          else { gl[0] = res[0]; gl[1] = res[1]; gl[2] = gz; }
        }
        if (p.grad_attn) {
          if (shared_samples) atomicAdd(p.grad_attn + g, res[2]);
          else p.grad_attn[g] = res[2];
        }
      }
    }
    __syncthreads();
    // ======================= phase B: every wave applies the window rows it owns =======================
    if (!(p.diag & 1)) {
      const int ch = lane % CM;                                    // lane = (pixel of the adjacent pair, channel); CM = 16: lanes 32 .. 63 idle
      const bool half = lane >= CM, lane_on = lane < 2 * CM;
      // G entries in flight: their queue records, grad_out values and window values are read together (one LDS round trip each instead
      // of one per entry); entries whose pixel pairs overlap must stay in program order, so a group with an overlap runs one by one
      constexpr int G = 4;
      for (int prod = 0; prod < NW; ++prod) {
        const lds_entry *q = queue + (wid * NW + prod) * kBtQueue;
        const int c = __builtin_amdgcn_readfirstlane(qcount[wid * NW + prod]);
        for (int e = 0; e < c; e += G) {
          BtEntry en[G];
          int o[G];
          float tv[G], wv[G];
#pragma unroll
          for (int i = 0; i < G; ++i) en[i] = q[e + i < c ? e + i : c - 1];
#pragma unroll
          for (int i = 0; i < G; ++i) {
            o[i] = __builtin_amdgcn_readfirstlane(en[i].x);
            tv[i] = lane_on ? tops[en[i].w * CM + ch] : 0.f;
            wv[i] = __int_as_float(half ? en[i].z : en[i].y);
          }
          const int ng = min(G, c - e);
          bool clash = false;
#pragma unroll
          for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = i + 1; j < G; ++j) clash |= j < ng && o[i] - o[j] < 2 * CM && o[j] - o[i] < 2 * CM;
          if (lane_on) {
            if (!clash) {
              float v[G];
#pragma unroll
              for (int i = 0; i < G; ++i) v[i] = i < ng ? win[o[i] + lane] : 0.f;
#pragma unroll
              for (int i = 0; i < G; ++i)
                if (i < ng) win[o[i] + lane] = v[i] + wv[i] * tv[i];
            } else {
#pragma unroll
              for (int i = 0; i < G; ++i)
                if (i < ng) { lds_float *qq = win + o[i] + lane; *qq = *qq + wv[i] * tv[i]; }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- flush: whole head segments of a pixel row per wave instruction ----
  if (!(p.diag & 2)) {
    constexpr int RPP = NT / CM;                                // window rows per pass
    const int ch = tid % CM;
    int row = tid / CM;
    int ty = row / p.tw, tx = row - ty * p.tw;
    for (; row < npx; row += RPP) {
      const float v = win[row * CM + ch];
      if (v != 0.f) atomicAdd(gvcam + ((int64_t)(y0 + ty) * p.W + x0 + tx) * MC + m * CM + ch, v);
      tx += RPP;
      while (tx >= p.tw) { tx -= p.tw; ++ty; }
    }
    for (int i = tid; i < npx * p.D; i += NT) {
      const float v = gdist[i];
      if (v != 0.f) {
        const int r = i / p.D, d = i - r * p.D;
        const int yy = r / p.tw, xx = r - yy * p.tw;
        atomicAdd(gdcam + ((int64_t)(y0 + yy) * p.W + x0 + xx) * p.D + d, v);
      }
    }
  }
}

int g_tune_bwd_tile_diag = 0;   // timing experiments only (BwdTileParams.diag); inert unless SGC_DIAG=1 is in the environment

}  // namespace sgc

using namespace sgc;

constexpr int kBtWaves = 4;     // waves per workgroup = window-row owners

template <int CM>
static int launch_bwd_tile(const BwdTileParams &p, size_t smem, hipStream_t st) {
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void *)dfa3d_bwd_tile_kernel<CM, kBtWaves>, 160 * 1024, attr_done);
  hipLaunchKernelGGL((dfa3d_bwd_tile_kernel<CM, kBtWaves>), dim3((unsigned)(p.N * p.nbx * p.nby * p.M)), dim3(kBtWaves * 64), smem, st, p);
  return check_launch("dfa3d_bwd_tile_kernel");
}

// window (grad_value slice of one head + grad_dist) + the batch's grad_out rows + its sample queues
static int64_t bwd_tile_lds(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y) {
  const int64_t tw = bin_w + 2 * halo_x < W ? bin_w + 2 * halo_x : W, th = bin_h + 2 * halo_y < H ? bin_h + 2 * halo_y : H;
  const int64_t nw = kBtWaves, ub = nw * 16;
  return ((tw * th * (Cm + D) + ub * Cm + 3) & ~(int64_t)3) * 4 + nw * nw * kBtQueue * (int64_t)sizeof(BtEntry) + nw * nw * 4;
}

extern "C" int64_t sgc_dfa3d_backward_binned_lds_bytes(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y) {
  if (H <= 0 || W <= 0 || Cm <= 0 || D <= 0 || bin_w <= 0 || bin_h <= 0 || halo_x < 0 || halo_y < 0) return 0;
  return bwd_tile_lds(H, W, Cm, D, bin_w, bin_h, halo_x, halo_y);
}

extern "C" int sgc_dfa3d_backward_binned(const float *value, const float *dist, const float *loc3, const float *attn_or_null,
                                         const int32_t *bin_offset, const int32_t *head_shift_or_null, const float *grad_out, float *grad_value, float *grad_dist,
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
  const int64_t lds = bwd_tile_lds(H, W, Cm, D, bin_w, bin_h, halo_x, halo_y);
  if (lds > 160 * 1024) return set_error(SGC_EUNSUP, "sgc_dfa3d_backward_binned: the window needs %lld bytes of LDS", (long long)lds);
  BwdTileParams p = {};
  p.value = value; p.dist = dist; p.loc = loc3; p.attn = attn_or_null; p.bin_offset = bin_offset; p.grad_out = grad_out;
  p.head_shift = head_shift_or_null;
  p.grad_value = grad_value; p.grad_dist = grad_dist; p.grad_loc = grad_loc3_or_null; p.grad_attn = grad_attn_or_null;
  p.N = N; p.S = S; p.H = H; p.W = W; p.D = D; p.M = M; p.LM = loc_heads; p.P = P;
  p.bw = bin_w; p.bh = bin_h; p.nbx = ceil_div(W, bin_w); p.nby = ceil_div(H, bin_h);
  p.hx = halo_x; p.hy = halo_y;
  {
    static const bool diag_ok = getenv("SGC_DIAG") && atoi(getenv("SGC_DIAG")) == 1;
    p.diag = diag_ok ? g_tune_bwd_tile_diag : 0;
  }
  p.tw = bin_w + 2 * halo_x < W ? bin_w + 2 * halo_x : W;
  p.th = bin_h + 2 * halo_y < H ? bin_h + 2 * halo_y : H;
  if (p.tw < 2) return set_error(SGC_EUNSUP, "sgc_dfa3d_backward_binned: the window must be at least two pixels wide");
  hipStream_t st = (hipStream_t)stream;
  return Cm == 32 ? launch_bwd_tile<32>(p, (size_t)lds, st) : launch_bwd_tile<16>(p, (size_t)lds, st);
}
