// LDS-tiled form of the context-aware deformable gather (the north-star kernel) for gfx950.
//
// Same operator as dfa3d_fwd_wave_kernel<kPairsDeform> (dfa3d_fwd.hip): the DFA3D call of
// MSDeformableAttention3D_DFA3D.forward (TU/deformable_cross_attention.py:423-489; kernels
// CS/common/cuda/ms_depth_score_sample_cuda_kernel.cuh:24-148, wms_deform_attn_cuda_kernel.cuh:24-80,240-303)
// evaluated on the visible (camera, voxel) pairs, softmax over the points and `ref + offset / (W,H,D)` fused in.
//
// What is different: the wave kernel fetches every corner row of every sample from L2 (one 128-byte head segment
// per corner: ~1.3 GB of L2 -> L1 traffic per launch at config 2 against 0.33 GB of compulsory HBM bytes, and it
// runs at the practical L2 -> CU rate, profiles/r01_gather_pmc_sq_v9.json).  Here
//   * the value map is HEAD-MAJOR, [N][M][S][Cm] (written that way by value_proj's GEMM epilogue,
//     sgc_linear_rows_headmajor_bf16x3), so one head's pixels are contiguous 4*Cm-byte rows;
//   * the pairs of a camera are REORDERED by the feature pixel their reference point projects to (sgc_bin_pairs:
//     bins of bin_w x bin_h pixels; everything downstream of the pair list -- geometry sample, the raw projection,
//     this kernel, the K/V projection -- simply runs in that order, `slot` still maps (camera, voxel) -> pair);
//   * a workgroup owns (camera, bin) and walks a group of heads: per head it stages the bin's window (+ halo, shifted
//     by the head's mean sampling offset) of that head's map in LDS with full-width LDS-DMA loads
//     (global_load_lds_dwordx4: 1 KiB per wave instruction, no registers; the next head's window lands in a second
//     buffer while the current head is computed), the camera's depth window once, then serves the 16 corner reads of
//     each (pair, head) unit from LDS (ds_read_b128).  Corners outside the staged window (large learned offsets) are
//     fetched from global memory in a rare wave-uniform fix-up branch; results never depend on the window.
//
// Lanes: phase 1 one lane per SAMPLE (16 units x 4 points per wave step): raw Linear outputs (head-major
// [pairs][M][P][4] = (du, dv, dz, logit): one 16-byte load per lane, prefetched one step ahead), softmax over the 4
// points by DPP, trilinear gate, depth taps from the LDS depth window -- all branch-free (clamped addresses +
// selects: the first version spent more scalar than vector instructions on exec-mask bookkeeping) -> 4 corner weights
// + 4 packed 16-bit LDS row offsets.  Phase 2 stays in the unit's own quad: lane c reads the 16-byte chunks c (and
// c + 4 for Cm = 32) of every corner row, so the descriptors of the unit's 4 samples arrive by DPP quad broadcasts on
// the VALU (no LDS-crossbar shuffles), one sample's 4 rows in flight at a time (ds_read_b128), FMA, 16-byte stores.
#include <stdlib.h>

#include "common.hpp"

namespace sgc {

// ---------------------------------------------------------------------------------------------
// sgc_bin_pairs: a stable counting sort of every camera's pairs by bin, in four small launches that keep every CU
// busy (the first version used one workgroup per camera and was bound by its chain of dependent loads: 20-110 us):
//   keys    one thread per pair: bin of its reference pixel + a (u, v, zn, q) record;
//   hist    grid (segments, cameras): per-segment bin histogram;
//   scan    one workgroup per camera: bases of every (segment, bin), bin_offset;
//   place   grid (segments, cameras): per-wave private counts (ballots over equal bins, no atomics: the order
//           inside a bin is the ascending original pair index, identical from run to run), then the move.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ref_bin(float u, float v, int H, int W, int bw, int bh, int nbx) {
#pragma clang fp contract(off)
  const float w_im = u * (float)W - 0.5f, h_im = v * (float)H - 0.5f;
  int px = (int)floorf(w_im), py = (int)floorf(h_im);
  px = min(max(px, 0), W - 1);
  py = min(max(py, 0), H - 1);
  return (py / bh) * nbx + px / bw;
}

__global__ void bin_keys_kernel(const float *__restrict__ ref_cam, const int32_t *__restrict__ pair_cam,
                                const int32_t *__restrict__ pair_q, const int32_t *__restrict__ cam_offset, int N, int Nq,
                                int H, int W, int bw, int bh, int nbx, int32_t *__restrict__ key, float4 *__restrict__ rec) {
  const int n_pairs = cam_offset[N];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += gridDim.x * blockDim.x) {
    const int n = pair_cam[p], q = pair_q[p];
    const float *rc = ref_cam + ((int64_t)n * Nq + q) * 3;
    const float u = rc[0], v = rc[1], z = rc[2];
    key[p] = ref_bin(u, v, H, W, bw, bh, nbx);
    rec[p] = make_float4(u, v, z, __int_as_float(q));
  }
}

// segment g of camera n: pairs [s0, s1)
__device__ __forceinline__ void bin_segment(const int32_t *cam_offset, int n, int g, int G, int *s0, int *s1) {
  const int p0 = cam_offset[n], p1 = cam_offset[n + 1];
  const int seg = ((p1 - p0 + G - 1) / G + 63) / 64 * 64;
  *s0 = min(p0 + g * seg, p1);
  *s1 = min(p0 + (g + 1) * seg, p1);
}

__global__ __launch_bounds__(256) void bin_hist_kernel(const int32_t *__restrict__ key, const int32_t *__restrict__ cam_offset,
                                                       int G, int nb, int32_t *__restrict__ hist) {
  extern __shared__ int bh_smem[];
  const int g = blockIdx.x, n = blockIdx.y;
  int s0, s1;
  bin_segment(cam_offset, n, g, G, &s0, &s1);
  for (int i = threadIdx.x; i < nb; i += blockDim.x) bh_smem[i] = 0;
  __syncthreads();
  for (int p = s0 + threadIdx.x; p < s1; p += blockDim.x) atomicAdd(&bh_smem[key[p]], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += blockDim.x) hist[((int64_t)n * G + g) * nb + i] = bh_smem[i];
}

__global__ __launch_bounds__(1024) void bin_scan_kernel(const int32_t *__restrict__ cam_offset, int N, int G, int nb,
                                                        int32_t *__restrict__ hist, int32_t *__restrict__ bin_offset) {
  __shared__ int wave_tot[16];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int tot_b = 0;
  if (tid < nb)
    for (int g = 0; g < G; ++g) tot_b += hist[((int64_t)n * G + g) * nb + tid];
  int incl = tot_b;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wid; ++w) wbase += wave_tot[w];
  const int p0 = cam_offset[n];
  if (tid < nb) {
    int run = p0 + wbase + incl - tot_b;            // first pair of bin `tid` of this camera
    bin_offset[(int64_t)n * nb + tid] = run;
    for (int g = 0; g < G; ++g) {                   // hist <- base of (segment g, bin tid)
      const int64_t i = ((int64_t)n * G + g) * nb + tid;
      const int c = hist[i];
      hist[i] = run;
      run += c;
    }
  }
  if (n == N - 1 && tid == 0) bin_offset[(int64_t)N * nb] = cam_offset[N];
}

__global__ __launch_bounds__(256) void bin_place_kernel(const int32_t *__restrict__ key, const float4 *__restrict__ rec,
                                                        const int32_t *__restrict__ cam_offset, const int32_t *__restrict__ base,
                                                        int Nq, int G, int nb, int32_t *__restrict__ pair_q_out,
                                                        int32_t *__restrict__ slot, float4 *__restrict__ pair_ref) {
  extern __shared__ int bp_smem[];                  // cnt[4][nb]
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int s0, s1;
  bin_segment(cam_offset, n, g, G, &s0, &s1);
  if (s0 >= s1) return;
  const int wseg = ((s1 - s0 + 3) / 4 + 63) / 64 * 64;       // the segment is dealt to the 4 waves in order
  const int w0 = min(s0 + wid * wseg, s1), w1 = min(s0 + (wid + 1) * wseg, s1);
  for (int i = tid; i < 4 * nb; i += 256) bp_smem[i] = 0;
  __syncthreads();
  int *mine = bp_smem + wid * nb;
  for (int i = w0; i < w1; i += 64) {               // private counts
    const int p = i + lane;
    const bool act = p < w1;
    const int b = act ? key[p] : -1;
    unsigned long long rem = __ballot(act);
    while (rem) {
      const int l = __ffsll((long long)rem) - 1;
      const int b0 = __builtin_amdgcn_readlane(b, l);      // l is wave-uniform: a VALU read, no LDS round trip
      const unsigned long long same = __ballot(act && b == b0);
      if (lane == l) mine[b0] += __popcll(same);
      rem &= ~same;
    }
  }
  __syncthreads();
  for (int b = tid; b < nb; b += 256) {             // counts -> bases (bin-major, then wave order)
    int run = base[((int64_t)n * G + g) * nb + b];
    for (int w = 0; w < 4; ++w) {
      const int c = bp_smem[w * nb + b];
      bp_smem[w * nb + b] = run;
      run += c;
    }
  }
  __syncthreads();
  for (int i = w0; i < w1; i += 64) {               // place
    const int p = i + lane;
    const bool act = p < w1;
    const int b = act ? key[p] : -1;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) r = rec[p];
    unsigned long long rem = __ballot(act);
    int pos = 0;
    while (rem) {
      const int l = __ffsll((long long)rem) - 1;
      const int b0 = __builtin_amdgcn_readlane(b, l);      // l is wave-uniform: a VALU read, no LDS round trip
      const unsigned long long same = __ballot(act && b == b0);
      const int bs = mine[b0];                      // every lane of the wave reads before the leader writes
      __builtin_amdgcn_wave_barrier();
      if (act && b == b0) pos = bs + __popcll(same & ((1ull << lane) - 1ull));
      if (lane == l) mine[b0] = bs + __popcll(same);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      rem &= ~same;
    }
    if (act) {
      const int q = __float_as_int(r.w);
      pair_q_out[pos] = q;
      slot[(int64_t)n * Nq + q] = pos;
      pair_ref[pos] = r;
    }
  }
}

// ---------------------------------------------------------------------------------------------
struct TileParams {
  const void *value;          // [N][M][S][CM] head-major, fp32 or bf16 (the kernel's VB template argument)
  const void *dist;           // [N][S][D] fp32, or bf16 in the storage mode (VB = 2: the value map AND the depth maps are bfloat16)
  const float4 *pair_ref;     // [pairs] (u, v, zn, q bits) in (camera, bin) order
  const int32_t *bin_offset;  // [N*nb + 1]
  const float4 *raw;          // [pairs][M][P] x (du, dv, dz, logit)
  float *out;                 // [pairs][M*CM]
  const int32_t *head_shift;  // [M][2] window shift per head in pixels (x, y), or null
  int N, S, H, W, D, M;
  int bw, bh, nbx, nby;
  int tw, th;                 // staged value window (already clipped to the map: tw <= W, th <= H)
  int dw, dh;                 // staged depth window (covers every head's shifted value window)
  int hx, hy;                 // halo on each side of the bin
  int smx, smy;               // largest |shift| over the heads
  int HG;                     // heads per workgroup (divides M)
  int xcd_map;                // 1: all work of camera n on XCD n % 8 (see the kernel), 0: plain (camera, bin, head) order
  int diag;                   // timing experiments only (sgc_set_tuning "tile_diag"): 1 = no compute, 2 = no fill
};

constexpr unsigned kFallbackBit = 0x8000u;  // rare-path descriptor: low 15 bits = pixel index in the map

// rows of `row_bytes` contiguous bytes (global: `g_row0 + r * g_pitch`) -> dense LDS image [rows][row_bytes], by
// LDS-DMA.  Rows are dealt to the NW waves; a row goes out in 1-KiB pieces (the last one partial): every address is
// wave-uniform scalar arithmetic plus lane * 16 (an image-order deal needs a per-lane division and carries).
// row_bytes is a multiple of 16.
template <int NW>
__device__ __forceinline__ void lds_dma_rows(const char *__restrict__ g_row0, int64_t g_pitch, int rows, int row_bytes,
                                             unsigned char *lds_img, int wid, int lane) {
  const int lo = lane * 16;
  for (int r = wid; r < rows; r += NW) {
    const char *g = g_row0 + (int64_t)r * g_pitch + lo;
    unsigned char *l = lds_img + r * row_bytes;
    for (int c = 0; c < row_bytes; c += 1024)
      if (c + lo < row_bytes)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + c),
                                         (__attribute__((address_space(3))) void *)(l + c), 16, 0, 0);
  }
}

template <int S> __device__ __forceinline__ float quad_bcast(float v) {
  return dpp_move<S | (S << 2) | (S << 4) | (S << 6)>(v);
}
template <int S> __device__ __forceinline__ unsigned quad_bcast_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, true);
}

// one 4-channel chunk of a value row: 16 bytes of fp32, or 8 bytes of bf16 widened to fp32 (bf16 -> fp32 is exact)
template <int VB> __device__ __forceinline__ float4 load_chunk(const unsigned char *p);
template <> __device__ __forceinline__ float4 load_chunk<4>(const unsigned char *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 load_chunk<2>(const unsigned char *p) {
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}

// DS (with DL, one head per workgroup): the staged depth window IS the head's value window (same origin, same extent), so a corner's
// depth row is its value row and one window test serves both -- 14 vector instructions per step less in phase 1, which is what bounds
// the kernel at Cm = 16 (round-6 counters: the vector ALU is busy 83 % of the kernel's time, profiles/r06_pmc_gather_cm16.json)
template <int CM, int NW, bool DL, int NBUF, int VB, bool DS = false>
__global__ __launch_bounds__(NW * 64) void dfa3d_fwd_tile_kernel(const TileParams p) {
  static_assert(!DS || (DL && NBUF == 1), "DS: depth window in LDS, one head per workgroup");
  constexpr int P = 4;
  constexpr int NCH = CM / 16;          // 16-byte chunks of a row per lane in phase 2 (a unit's 4 lanes cover the row)
  constexpr int CV = CM / 4;            // 16-byte chunks per row
  constexpr int UPW = 64 / P;           // units per wave step (16)
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ngrp = p.M / p.HG;
  const int nb = p.nbx * p.nby;
  int hg, t;
  if (p.xcd_map) {
    // (option, off by default -- see g_tune_tile_xcd)  Workgroups go to the 8 XCDs round-robin by blockIdx, and every XCD
    // has its own L2.  All work of camera n runs on XCD n % 8, head-major with the bins innermost: a bin's halo rows are
    // re-read by its neighbours a few workgroups later, and the camera's depth map (shared by the 8 heads; re-fetched by
    // 8 different L2s when head h simply goes to XCD h) is re-read by the next head about 1 MB of traffic later.
    const int x = blockIdx.x & 7, r = blockIdx.x >> 3, per_cam = nb * ngrp;
    const int n = (r / per_cam) * 8 + x;
    if (n >= p.N) return;
    const int j = r - (r / per_cam) * per_cam;
    hg = j / nb;
    t = n * nb + (j - hg * nb);
  } else {
    hg = blockIdx.x % ngrp;
    t = blockIdx.x / ngrp;                      // (camera, bin)
  }
  const int i0 = p.bin_offset[t], i1 = p.bin_offset[t + 1];
  // the first head's window shift travels with the bin bounds (two dependent scalar round trips otherwise)
  int sx_first = 0, sy_first = 0;
  if (p.head_shift) { sx_first = p.head_shift[hg * p.HG * 2]; sy_first = p.head_shift[hg * p.HG * 2 + 1]; }
  if (i0 >= i1) return;
  const int cnt = i1 - i0;
  const int n = t / nb, b = t - n * nb;
  const int by = b / p.nbx, bx = b - by * p.nbx;
  const int npx = p.tw * p.th;
  constexpr int CHB = 4 * VB;                                       // bytes of a 4-channel chunk
  const int buf_bytes = (npx + 1) * CM * VB;                        // window + one all-zero row
  unsigned char *val0 = tile_smem;
  // depth maps: fp32, or bfloat16 in the storage mode (widened by a shift: exact).  DB = bytes per depth value
  constexpr int DB = VB == 2 ? 2 : 4;
  const unsigned char *dep = tile_smem + ((NBUF * buf_bytes + 15) & ~15);                 // [dh][dw][D]
  const unsigned char *dcam = reinterpret_cast<const unsigned char *>(p.dist) + (int64_t)n * p.S * p.D * DB;
  auto depth_pair = [&](const unsigned char *q, float &a, float &b) {                      // two consecutive depth bins
    if constexpr (DB == 4) {
      const float2_u pr = *reinterpret_cast<const float2_u *>(q);
      a = pr.x; b = pr.y;
    } else {                                           // 2-byte aligned only: two 16-bit reads
      a = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(q) << 16);
      b = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(q + 2) << 16);
    }
  };
  int xd0 = max(0, min(bx * p.bw - p.hx - p.smx, p.W - p.dw));
  int yd0 = max(0, min(by * p.bh - p.hy - p.smy, p.H - p.dh));
  if (p.HG == 1 && p.dw == p.tw && p.dh == p.th) {   // one head per workgroup: the depth window is the head's own value window
    xd0 = max(0, min(bx * p.bw - p.hx + sx_first, p.W - p.tw));
    yd0 = max(0, min(by * p.bh - p.hy + sy_first, p.H - p.th));
  }

  auto window_origin = [&](int m, int &x0, int &y0) {
    int sx = sx_first, sy = sy_first;
    if (p.head_shift && m != hg * p.HG) { sx = p.head_shift[m * 2]; sy = p.head_shift[m * 2 + 1]; }
    x0 = max(0, min(bx * p.bw - p.hx + sx, p.W - p.tw));
    y0 = max(0, min(by * p.bh - p.hy + sy, p.H - p.th));
  };
  auto fill_value = [&](int m, int buf) {
    int x0, y0;
    window_origin(m, x0, y0);
    const char *plane = reinterpret_cast<const char *>(p.value) + ((int64_t)n * p.M + m) * p.S * CM * VB;
    lds_dma_rows<NW>(plane + ((int64_t)y0 * p.W + x0) * CM * VB, (int64_t)p.W * CM * VB, p.th,
                     p.tw * CM * VB, val0 + buf * buf_bytes, wid, lane);
  };

  const int m_first = hg * p.HG;
  if (p.diag != 2) {
    if (DL)
      lds_dma_rows<NW>(reinterpret_cast<const char *>(dcam + ((int64_t)yd0 * p.W + xd0) * p.D * DB), (int64_t)p.W * p.D * DB, p.dh,
                       p.dw * p.D * DB, const_cast<unsigned char *>(dep), wid, lane);
    fill_value(m_first, 0);
  }
  if (tid < NBUF * CV) {                                            // the zero rows (one 4-channel chunk per thread)
    const int bsel = tid / CV;
    unsigned char *z = val0 + bsel * buf_bytes + npx * CM * VB + (tid - bsel * CV) * CHB;
    if (VB == 4) *reinterpret_cast<float4 *>(z) = make_float4(0.f, 0.f, 0.f, 0.f);
    else *reinterpret_cast<uint2 *>(z) = make_uint2(0u, 0u);
  }

  const float rW = 1.0f / (float)p.W, rH = 1.0f / (float)p.H, rD = 1.0f / (float)p.D;
  const float fW = (float)p.W, fH = (float)p.H, fD = (float)p.D;
  const int ul = lane / P, pt = lane % P;
  const int Dm2 = p.D - 2;

  for (int hi = 0; hi < p.HG; ++hi) {
    const int m = m_first + hi;
    const int buf = NBUF == 2 ? (hi & 1) : 0;
    const unsigned char *val = val0 + buf * buf_bytes;
    int x0, y0;
    window_origin(m, x0, y0);
    const unsigned char *plane = reinterpret_cast<const unsigned char *>(p.value) + ((int64_t)n * p.M + m) * p.S * CM * VB;
    // first step's operands: issued before the wait below, they travel with the window
    int g0 = wid * UPW;
    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f), r4 = rec;
    if (g0 < cnt) {
      const int iu = min(i0 + g0 + ul, i1 - 1);
      rec = p.pair_ref[iu];
      r4 = p.raw[((int64_t)iu * p.M + m) * P + pt];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                      // window of head m complete; every wave is done with the other buffer
    if (NBUF == 2 && hi + 1 < p.HG && p.diag != 2) fill_value(m + 1, buf ^ 1);
    if (p.diag != 1) {
    for (; g0 < cnt; g0 += NW * UPW) {
      // ---------------- phase 1: lane = (unit, point), branch-free ----------------
      // the reference's forms (TU/deformable_cross_attention.py:428-455): offset / (W, H, D) with IEEE rounding (div_by_size),
      // then the addition; softmax over the 4 points with expf and a true division.  (Round 3 multiplied by the reciprocal and
      // used v_exp / v_rcp, <= 2 ulp off: same kernel time -- 109.4 vs 109.2 us config 2, 374 vs 365 us config 4, alternated
      // rounds, profiles/r04_gather_ieee.txt -- so there is no reason to differ from the reference in the last bit.)
      const float x = rec.x + div_by_size(r4.x, fW, rW), y = rec.y + div_by_size(r4.y, fH, rH), z = rec.z + div_by_size(r4.z, fD, rD);
      float mx = fmaxf(r4.w, lane_xor(r4.w, 1));
      mx = fmaxf(mx, lane_xor(mx, 2));
      const float e = expf(r4.w - mx);
      float sum = e + lane_xor(e, 1);
      sum += lane_xor(sum, 2);
      const float aw = e / sum;
      // next step's operands (one step ahead)
      const int gn = g0 + NW * UPW;
      float4 rec_n = rec, r4_n = r4;
      if (gn < cnt) {
        const int iun = min(i0 + gn + ul, i1 - 1);
        rec_n = p.pair_ref[iun];
        r4_n = p.raw[((int64_t)iun * p.M + m) * P + pt];
      }

      const float h_im = sample_coord(y, fH), w_im = sample_coord(x, fW), d_im = sample_coord(z, fD);
      // bitwise & on purpose: with && the compiler evaluates the later comparisons under an exec mask (an s_and_saveexec /
      // s_or pair per chain -- 13 of them per step); every operand here is cheap and side-effect free
      const bool in2 = (h_im > -1.f) & (w_im > -1.f) & (h_im < fH) & (w_im < fW);
      const bool in3 = in2 & (d_im > -1.f) & (d_im < fD);
      const float hf = floorf(h_im), wf = floorf(w_im), df = floorf(d_im);
      // (int) of a huge float is undefined: clamp the floats first (in2 / in3 already hold the decision)
      const int h0 = (int)__builtin_amdgcn_fmed3f(hf, -2.f, fH), w0 = (int)__builtin_amdgcn_fmed3f(wf, -2.f, fW);
      const int d0 = (int)__builtin_amdgcn_fmed3f(df, -2.f, fD);
      const float lh = h_im - hf, lw = w_im - wf, ld = d_im - df;
      const float hh = 1.f - lh, hw = 1.f - lw, hd = 1.f - ld;
      const int dbase = min(max(d0, 0), Dm2);
      const bool dlo = d0 == dbase;               // false only at the two depth borders
      const bool d0ok = d0 >= 0, d1ok = d0 + 1 <= p.D - 1;
      const unsigned char *depb = dep + dbase * DB;
      float ta[4], tb[4];
      bool ok[4], inside[4];
      int pix[4], trow[4];
      bool need_g = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {               // corner order (h0,w0) (h0,w1) (h1,w0) (h1,w1)
        const int hk = h0 + (k >> 1), wk = w0 + (k & 1);
        ok[k] = in2 & (hk >= 0) & (hk <= p.H - 1) & (wk >= 0) & (wk <= p.W - 1);
        const int ch = min(max(hk, 0), p.H - 1), cw = min(max(wk, 0), p.W - 1);
        pix[k] = __mul24(ch, p.W) + cw;
        const int tx = cw - x0, ty = ch - y0;
        inside[k] = ((unsigned)tx < (unsigned)p.tw) & ((unsigned)ty < (unsigned)p.th);
        const int trow_in = __mul24(ty, p.tw) + tx;
        trow[k] = inside[k] ? trow_in : npx;
        if (DL) {
          bool din;
          int drow;
          if constexpr (DS) {
            din = inside[k]; drow = trow_in;
          } else {
            const int dx = cw - xd0, dy = ch - yd0;
            din = ((unsigned)dx < (unsigned)p.dw) & ((unsigned)dy < (unsigned)p.dh);
            drow = __mul24(dy, p.dw) + dx;
          }
          const unsigned char *dp = depb + __umul24((unsigned)(din ? drow : 0), (unsigned)(p.D * DB));
          if constexpr (DB == 4) { ta[k] = reinterpret_cast<const float *>(dp)[0]; tb[k] = reinterpret_cast<const float *>(dp)[1]; }
          else depth_pair(dp, ta[k], tb[k]);
          need_g |= in3 & ok[k] & !din;
        } else {
          depth_pair(dcam + (unsigned)(__mul24(pix[k], p.D) + dbase) * DB, ta[k], tb[k]);
        }
      }
      if (DL && __ballot(need_g)) {               // rare, wave-uniform: depth taps outside the staged depth window
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int cw = pix[k] % p.W, ch = pix[k] / p.W;
          const int dx = cw - xd0, dy = ch - yd0;
          const bool din = (unsigned)dx < (unsigned)p.dw && (unsigned)dy < (unsigned)p.dh;
          if (!din) {
            depth_pair(dcam + ((int64_t)pix[k] * p.D + dbase) * DB, ta[k], tb[k]);
          }
        }
      }
      float wgt[4];
      unsigned rowb[4], fbs[4];                      // byte offset of the corner's row in the staged window
      bool any_fb = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float va = d0ok ? (dlo ? ta[k] : tb[k]) : 0.f;
        const float vb = d1ok ? (dlo ? tb[k] : ta[k]) : 0.f;
        const float sc = va * hd + vb * ld;
        const float bil = (k == 0 ? hh * hw : k == 1 ? hh * lw : k == 2 ? lh * hw : lh * lw);
        // ok[k] holds the 2-D gate and in3 implies it: ONE select instead of the score's and the weight's (a gated-off score is 0, and
        // bil * 0 * aw is 0 for the finite operands the maps hold)
        wgt[k] = (in3 & ok[k]) ? bil * sc * aw : 0.f;
        const bool fb = ok[k] & !inside[k];
        rowb[k] = (unsigned)(ok[k] ? trow[k] : npx) * (unsigned)(CM * VB); // outside the map / outside the window -> zero row (a shift)
        fbs[k] = fb ? (kFallbackBit | (unsigned)pix[k]) : 0u;
        any_fb |= fb;
      }
      const unsigned long long fbm = __ballot(any_fb);

      // ---------------- phase 2: the unit's own quad; lane c owns channels 4c .. 4c+3 (+16 j for Cm = 32) ----------------
      // Descriptors of the unit's four samples come from the lanes of the same quad: DPP quad broadcasts on the VALU, no
      // LDS-crossbar shuffles.  One sample (4 corner rows x NCH 16-byte chunks) is in flight at a time per lane.
      {
        const int c16 = lane % P;                         // 16-byte chunk of the row this lane reads (and chunk c16 + 4 j)
        const unsigned char *vrow = val + c16 * CHB;
        float4 acc[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        // every broadcast has ONE consumer (the address add, or -- per corner weight -- the four multiply-adds of its chunk), so
        // the row offsets travel as four plain byte offsets rather than two packed pairs: no unpacking, and the compiler can fold
        // the quad broadcast into the add (v_add_u32_dpp)
        // The row offsets travel as four plain byte offsets, each with ONE consumer (the address add), so the compiler folds the quad
        // broadcast into it (v_add_u32_dpp) and nothing is unpacked.  (The weights keep their v_mov_b32_dpp broadcasts: reading them
        // through the DPP operand of every multiply-add instead -- inline-asm v_fmac_f32_dpp, 16 instructions per step less -- changed
        // nothing measurable: 0.379 vs 0.374 / 0.383 vs 0.391 / 0.446 vs 0.457 of 8 TB/s at configs 4 / 5 / 2.)
        auto sample = [&](const float w0, const float w1, const float w2, const float w3, const unsigned b0, const unsigned b1,
                          const unsigned b2, const unsigned b3) {
          const float w[4] = {w0, w1, w2, w3};
          const unsigned r[4] = {b0, b1, b2, b3};
          float4 v[4][NCH];
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < NCH; ++j) v[k][j] = load_chunk<VB>(vrow + r[k] + 4 * j * CHB);
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
              acc[j].x += w[k] * v[k][j].x; acc[j].y += w[k] * v[k][j].y;
              acc[j].z += w[k] * v[k][j].z; acc[j].w += w[k] * v[k][j].w;
            }
        };
        sample(quad_bcast<0>(wgt[0]), quad_bcast<0>(wgt[1]), quad_bcast<0>(wgt[2]), quad_bcast<0>(wgt[3]),
               quad_bcast_u<0>(rowb[0]), quad_bcast_u<0>(rowb[1]), quad_bcast_u<0>(rowb[2]), quad_bcast_u<0>(rowb[3]));
        sample(quad_bcast<1>(wgt[0]), quad_bcast<1>(wgt[1]), quad_bcast<1>(wgt[2]), quad_bcast<1>(wgt[3]),
               quad_bcast_u<1>(rowb[0]), quad_bcast_u<1>(rowb[1]), quad_bcast_u<1>(rowb[2]), quad_bcast_u<1>(rowb[3]));
        sample(quad_bcast<2>(wgt[0]), quad_bcast<2>(wgt[1]), quad_bcast<2>(wgt[2]), quad_bcast<2>(wgt[3]),
               quad_bcast_u<2>(rowb[0]), quad_bcast_u<2>(rowb[1]), quad_bcast_u<2>(rowb[2]), quad_bcast_u<2>(rowb[3]));
        sample(quad_bcast<3>(wgt[0]), quad_bcast<3>(wgt[1]), quad_bcast<3>(wgt[2]), quad_bcast<3>(wgt[3]),
               quad_bcast_u<3>(rowb[0]), quad_bcast_u<3>(rowb[1]), quad_bcast_u<3>(rowb[2]), quad_bcast_u<3>(rowb[3]));
        if (fbm) {            // rare, wave-uniform: some corner of some unit lies outside the staged window
          const int src = lane & ~(P - 1);
#pragma unroll
          for (int s = 0; s < P; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const unsigned sl = (unsigned)__shfl((int)fbs[k], src + s);
              const float wk = __shfl(wgt[k], src + s);
              if (sl & kFallbackBit) {
                const unsigned char *grow = plane + (int64_t)(sl & 0x7fffu) * CM * VB + c16 * CHB;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                  const float4 gv = load_chunk<VB>(grow + 4 * j * CHB);
                  acc[j].x += wk * gv.x; acc[j].y += wk * gv.y; acc[j].z += wk * gv.z; acc[j].w += wk * gv.w;
                }
              }
            }
        }
        if (g0 + ul < cnt) {
          float4 *orow = reinterpret_cast<float4 *>(p.out + ((int64_t)(i0 + g0 + ul) * p.M + m) * CM) + c16;
#pragma unroll
          for (int j = 0; j < NCH; ++j) orow[4 * j] = acc[j];
        }
      }
      rec = rec_n; r4 = r4_n;
    }
    }
    if (NBUF == 1 && hi + 1 < p.HG) {
      __syncthreads();                    // every wave is done reading the single buffer
      if (p.diag != 2) fill_value(m + 1, 0);
    }
  }
  if (p.diag == 1 && tid == 0) p.out[(int64_t)i0 * p.M * CM] = (float)val0[0] + (DL ? dep[0] : 0.f);
}

// A/B knobs (sgc_set_tuning); 0 = the library's own choice.  Results never depend on them.
int g_tune_tile_nw = 0;         // waves per workgroup (8 or 16).  auto: 8 when two workgroups fit a CU's LDS (<= 80 KB
                                // each: one computes while the other stages its window), else 16
int g_tune_tile_depth_lds = -1; // >= 0 overrides the caller's depth_in_lds
int g_tune_tile_diag = 0;       // TIMING EXPERIMENTS ONLY (1 = no compute, 2 = no fill: the outputs are garbage); inert unless
                                // SGC_DIAG=1 is in the environment, so a stray SGC_TUNE cannot corrupt a production run
static bool diag_allowed() {
  static const bool ok = getenv("SGC_DIAG") && atoi(getenv("SGC_DIAG")) == 1;
  return ok;
}
int g_tune_tile_nbuf = 0;       // value-window buffers; 2 = the next head's window lands while the current head is computed
                                // (needs heads-per-workgroup > 1).  auto: 1
int g_tune_tile_xcd = -1;       // 1: camera n's workgroups on XCD n % 8 (head-major, bins innermost); 0: (camera, bin, head) order,
                                // i.e. head h on XCD h; -1 (default): 1 at Cm = 32, 0 at Cm = 16.  Same time at config 2 (round 3,
                                // alternated bench runs: 95.8 / 97.5 vs 96.6 / 96.4 us) but a third less traffic on the memory side
                                // of the L2s: rocprofv3 --pmc FETCH_SIZE (doubled) + WRITE_SIZE = 435 MB -> 335 MB per launch = 1.30 ->
                                // 1.00 x the algorithmic bytes (profiles/r03_gather_tile_pmc_hbm.json) -- with head h on XCD h every
                                // camera's depth map was fetched through all eight L2s.  Config 4 (Cm = 16, depth window in LDS) was
                                // 8 % slower with 1 (round 2: 385 vs 420 us) and keeps 0.
int g_tune_tile_hg = 0;         // heads per workgroup.  auto: 1 (most workgroups: (camera, bin, head))
int g_tune_tile_ds = 1;         // 1: one window test for value and depth where the two windows coincide (template flag DS); 0: A/B

}  // namespace sgc

using namespace sgc;

static int bin_segments(int Nq) { return Nq <= 4096 ? 1 : (Nq + 4095) / 4096 > 32 ? 32 : (Nq + 4095) / 4096; }

extern "C" int64_t sgc_bin_pairs_workspace_bytes(int N, int Nq, int cap, int H, int W, int bin_w, int bin_h) {
  if (N <= 0 || Nq <= 0 || cap <= 0 || H <= 0 || W <= 0 || bin_w <= 0 || bin_h <= 0) return 0;
  const int64_t nb = (int64_t)ceil_div(W, bin_w) * ceil_div(H, bin_h);
  // rec [cap] float4 | key [cap] int32 | hist [N * G * nb] int32
  return (int64_t)cap * 16 + (((int64_t)cap * 4 + 15) / 16) * 16 + (int64_t)N * bin_segments(Nq) * nb * 4;
}

extern "C" int sgc_bin_pairs(const float *ref_cam, const int32_t *pair_cam, const int32_t *pair_q, const int32_t *cam_offset,
                             int32_t *pair_q_out, int32_t *slot, float *pair_ref, int32_t *bin_offset, void *workspace,
                             int N, int Nq, int cap, int H, int W, int bin_w, int bin_h, sgc_stream_t stream) {
  if (!ref_cam || !pair_cam || !pair_q || !cam_offset || !pair_q_out || !slot || !pair_ref || !bin_offset || !workspace)
    return set_error(SGC_EINVAL, "sgc_bin_pairs: null pointer");
  if (N <= 0 || Nq <= 0 || cap <= 0 || H <= 0 || W <= 0 || bin_w <= 0 || bin_h <= 0 || N > 65535)
    return set_error(SGC_EINVAL, "sgc_bin_pairs: bad size");
  if (pair_q_out == pair_q) return set_error(SGC_EINVAL, "sgc_bin_pairs: pair_q_out must not alias pair_q");
  const int nbx = ceil_div(W, bin_w), nby = ceil_div(H, bin_h), nb = nbx * nby;
  if (nb > 1024) return set_error(SGC_EUNSUP, "sgc_bin_pairs: more than 1024 bins per camera (%d)", nb);
  if ((reinterpret_cast<uintptr_t>(pair_ref) | reinterpret_cast<uintptr_t>(workspace)) & 15)
    return set_error(SGC_EINVAL, "sgc_bin_pairs: pair_ref / workspace must be 16-byte aligned");
  const int G = bin_segments(Nq);
  float4 *rec = reinterpret_cast<float4 *>(workspace);
  int32_t *key = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(workspace) + (int64_t)cap * 16);
  int32_t *hist = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(key) + (((int64_t)cap * 4 + 15) / 16) * 16);
  hipStream_t st = (hipStream_t)stream;
  const int kgrid = ceil_div(cap, 256) < 2048 ? ceil_div(cap, 256) : 2048;
  hipLaunchKernelGGL(bin_keys_kernel, dim3(kgrid), dim3(256), 0, st, ref_cam, pair_cam, pair_q, cam_offset, N, Nq, H, W,
                     bin_w, bin_h, nbx, key, rec);
  hipLaunchKernelGGL(bin_hist_kernel, dim3(G, N), dim3(256), (size_t)nb * 4, st, key, cam_offset, G, nb, hist);
  hipLaunchKernelGGL(bin_scan_kernel, dim3(N), dim3(1024), 0, st, cam_offset, N, G, nb, hist, bin_offset);
  hipLaunchKernelGGL(bin_place_kernel, dim3(G, N), dim3(256), (size_t)4 * nb * 4, st, key, rec, cam_offset, hist, Nq, G, nb,
                     pair_q_out, slot, reinterpret_cast<float4 *>(pair_ref));
  return check_launch("sgc_bin_pairs");
}

namespace {
struct TileGeom { int tw, th, dw, dh, smx, smy, nbuf; bool dl; size_t lds; };
}

// window sizes and LDS bytes for a (bin, halo, max shift) choice; nbuf falls back to 1 when two value buffers do not fit
static TileGeom tile_geometry(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y, int smx, int smy,
                              int depth_in_lds, int vb = 4) {
  TileGeom g;
  g.tw = bin_w + 2 * halo_x < W ? bin_w + 2 * halo_x : W;
  g.th = bin_h + 2 * halo_y < H ? bin_h + 2 * halo_y : H;
  g.smx = smx; g.smy = smy;
  // depth window: with one head per workgroup it is that head's own (shifted) value window; a workgroup that walks several
  // heads stages the union of their windows once
  const bool one_head = !(g_tune_tile_hg > 1);
  g.dw = one_head ? g.tw : (bin_w + 2 * (halo_x + smx) < W ? bin_w + 2 * (halo_x + smx) : W);
  g.dh = one_head ? g.th : (bin_h + 2 * (halo_y + smy) < H ? bin_h + 2 * (halo_y + smy) : H);
  g.dl = (g_tune_tile_depth_lds >= 0 ? g_tune_tile_depth_lds : depth_in_lds) != 0 && D % 4 == 0 && D >= 2;
  const size_t db = vb == 2 ? 2 : 4;               // storage mode: bf16 depth maps beside the bf16 value map
  const size_t vbuf = ((size_t)g.tw * g.th + 1) * Cm * vb, dbuf = ((size_t)g.dw * g.dh * D * db + 15) & ~(size_t)15;
  if ((g.dw * D * db) % 16) g.dl = false;          // the LDS-DMA deals 16-byte pieces of a row
  // (the window's SOURCE address is pixel-aligned only: 48 bytes per fp32 pixel at D = 12, but 24 per bf16 pixel -- an odd
  //  origin column then starts 8 bytes into a 16-byte unit.  global_load_lds_dwordx4 needs dword alignment of the global
  //  address, not 16 bytes; test_bf16_storage_mode_of_the_tiled_gather runs odd origins against the oracle.)
  g.nbuf = g_tune_tile_nbuf == 2 ? 2 : 1;
  if (g.nbuf == 2 && 2 * vbuf + (g.dl ? dbuf : 0) > 160 * 1024) g.nbuf = 1;
  if (g.dl && g.nbuf * vbuf + dbuf > 160 * 1024) g.dl = false;       // depth taps from global memory instead
  g.lds = ((g.nbuf * vbuf + 15) & ~(size_t)15) + (g.dl ? dbuf : 0);
  return g;
}

extern "C" int sgc_tile_window(int H, int W, int Cm, int D, int bin_w, int bin_h, int halo_x, int halo_y, int max_shift_x,
                               int max_shift_y, int depth_in_lds, int value_bf16, int *tw_out, int *th_out, int *lds_bytes_out, int *nbuf_out,
                               int *depth_in_lds_out) {
  const TileGeom g = tile_geometry(H, W, Cm, D, bin_w, bin_h, halo_x, halo_y, max_shift_x, max_shift_y, depth_in_lds, value_bf16 ? 2 : 4);
  if (tw_out) *tw_out = g.tw;
  if (th_out) *th_out = g.th;
  if (lds_bytes_out) *lds_bytes_out = (int)g.lds;
  if (nbuf_out) *nbuf_out = g.nbuf;
  if (depth_in_lds_out) *depth_in_lds_out = g.dl ? 1 : 0;
  return SGC_OK;
}

template <int CM, int NW, bool DL, int NBUF, int VB, bool DS>
static int launch_tile_ds(const TileParams &p, size_t smem, hipStream_t st) {
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void *)dfa3d_fwd_tile_kernel<CM, NW, DL, NBUF, VB, DS>, 160 * 1024, attr_done);
  const int64_t grid = (int64_t)(p.xcd_map ? (p.N + 7) / 8 * 8 : p.N) * p.nbx * p.nby * (p.M / p.HG);
  hipLaunchKernelGGL((dfa3d_fwd_tile_kernel<CM, NW, DL, NBUF, VB, DS>), dim3((unsigned)grid), dim3(NW * 64), smem, st, p);
  return check_launch("dfa3d_fwd_tile_kernel");
}

template <int CM, int NW, bool DL, int NBUF, int VB>
static int launch_tile(const TileParams &p, size_t smem, hipStream_t st) {
  if constexpr (DL && NBUF == 1) {
    if (g_tune_tile_ds && p.HG == 1 && p.dw == p.tw && p.dh == p.th) return launch_tile_ds<CM, NW, DL, NBUF, VB, true>(p, smem, st);
  }
  return launch_tile_ds<CM, NW, DL, NBUF, VB, false>(p, smem, st);
}

extern "C" int sgc_pairs_deform_gather_tiled(const void *value_hm, int value_bf16, const void *dist, const float *pair_ref,
                                             const int32_t *bin_offset, const float *raw_hm,
                                             const int32_t *head_shift_or_null, float *out, int N, int H, int W, int M,
                                             int Cm, int D, int P, int cam_stride_or_0, int bin_w, int bin_h, int halo_x,
                                             int halo_y, int max_shift_x, int max_shift_y, int depth_in_lds, sgc_stream_t stream) {
  if (!value_hm || !dist || !pair_ref || !bin_offset || !raw_hm || !out)
    return set_error(SGC_EINVAL, "sgc_pairs_deform_gather_tiled: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || M <= 0 || D <= 0 || bin_w <= 0 || bin_h <= 0 || halo_x < 0 || halo_y < 0)
    return set_error(SGC_EINVAL, "sgc_pairs_deform_gather_tiled: bad size");
  if (P != 4 || (Cm != 16 && Cm != 32))
    return set_error(SGC_EUNSUP, "sgc_pairs_deform_gather_tiled: P = 4 and Cm in {16, 32} only (got P = %d, Cm = %d)", P, Cm);
  if ((int64_t)H * W >= 0x7fff) return set_error(SGC_EUNSUP, "sgc_pairs_deform_gather_tiled: H*W must stay below 32767");
  if (D < 2) return set_error(SGC_EUNSUP, "sgc_pairs_deform_gather_tiled: D >= 2 required");
  if (cam_stride_or_0 > 0 && cam_stride_or_0 < H * W) return set_error(SGC_EINVAL, "sgc_pairs_deform_gather_tiled: cam_stride < H*W");
  if (max_shift_x < 0 || max_shift_y < 0) return set_error(SGC_EINVAL, "sgc_pairs_deform_gather_tiled: negative max_shift");
  if ((reinterpret_cast<uintptr_t>(value_hm) | reinterpret_cast<uintptr_t>(dist) | reinterpret_cast<uintptr_t>(pair_ref) |
       reinterpret_cast<uintptr_t>(raw_hm) | reinterpret_cast<uintptr_t>(out)) & 15)
    return set_error(SGC_EINVAL, "sgc_pairs_deform_gather_tiled: pointers must be 16-byte aligned");
  const int smx = head_shift_or_null ? max_shift_x : 0, smy = head_shift_or_null ? max_shift_y : 0;
  const TileGeom g = tile_geometry(H, W, Cm, D, bin_w, bin_h, halo_x, halo_y, smx, smy, depth_in_lds, value_bf16 ? 2 : 4);
  TileParams p = {};
  p.value = value_hm; p.dist = dist; p.pair_ref = reinterpret_cast<const float4 *>(pair_ref); p.bin_offset = bin_offset;
  p.raw = reinterpret_cast<const float4 *>(raw_hm); p.out = out; p.head_shift = head_shift_or_null;
  p.N = N; p.S = cam_stride_or_0 > 0 ? cam_stride_or_0 : H * W; p.H = H; p.W = W; p.D = D; p.M = M;
  p.bw = bin_w; p.bh = bin_h; p.nbx = ceil_div(W, bin_w); p.nby = ceil_div(H, bin_h);
  p.hx = halo_x; p.hy = halo_y; p.smx = smx; p.smy = smy; p.diag = diag_allowed() ? g_tune_tile_diag : 0;
  p.tw = g.tw; p.th = g.th; p.dw = g.dw; p.dh = g.dh;
  p.HG = (g_tune_tile_hg > 0 && M % g_tune_tile_hg == 0) ? g_tune_tile_hg : 1;
  p.xcd_map = g_tune_tile_xcd >= 0 ? g_tune_tile_xcd : (Cm == 32 ? 1 : 0);
  if (g.lds > 160 * 1024)
    return set_error(SGC_EUNSUP, "sgc_pairs_deform_gather_tiled: window %dx%d needs %zu bytes of LDS", g.tw, g.th, g.lds);
  if (((int64_t)g.tw * g.th + 1) * (Cm / 4) >= 0x10000)
    return set_error(SGC_EUNSUP, "sgc_pairs_deform_gather_tiled: window too large for 16-bit row offsets");
  hipStream_t st = (hipStream_t)stream;
  const int nw = g_tune_tile_nw == 8 ? 8 : g_tune_tile_nw == 16 ? 16 : (g.lds <= 80 * 1024 ? 8 : 16);
#define SGC_TILE_CASE(CMV, NWV, DLV, NB)                                                             \
  if (Cm == CMV && nw == NWV && g.dl == DLV && g.nbuf == NB)                                         \
    return value_bf16 ? launch_tile<CMV, NWV, DLV, NB, 2>(p, g.lds, st) : launch_tile<CMV, NWV, DLV, NB, 4>(p, g.lds, st)
  SGC_TILE_CASE(32, 16, true, 2); SGC_TILE_CASE(32, 16, false, 2); SGC_TILE_CASE(32, 8, true, 2); SGC_TILE_CASE(32, 8, false, 2);
  SGC_TILE_CASE(16, 16, true, 2); SGC_TILE_CASE(16, 16, false, 2); SGC_TILE_CASE(16, 8, true, 2); SGC_TILE_CASE(16, 8, false, 2);
  SGC_TILE_CASE(32, 16, true, 1); SGC_TILE_CASE(32, 16, false, 1); SGC_TILE_CASE(32, 8, true, 1); SGC_TILE_CASE(32, 8, false, 1);
  SGC_TILE_CASE(16, 16, true, 1); SGC_TILE_CASE(16, 16, false, 1); SGC_TILE_CASE(16, 8, true, 1); SGC_TILE_CASE(16, 8, false, 1);
#undef SGC_TILE_CASE
  return set_error(SGC_EUNSUP, "sgc_pairs_deform_gather_tiled: no kernel for this shape");
}
