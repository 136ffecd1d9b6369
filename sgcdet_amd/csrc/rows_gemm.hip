// Persistent, weight-stationary row GEMM on the bf16 matrix cores (gfx950):  y[M, N] = x[M, K] @ W^T (+ epilogue)
// with the 3-way bf16 split of conv3d.hip (a = a_hi + a_lo, products a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulate).
//
// What it is for: every nn.Linear of a level of the view transformation (value_proj over N*S pixel rows,
// the fused offset / logit projection and the K|V in-projection over the visible pairs, the q / out projections and the
// FFN over the selected voxels -- TU/deformable_cross_attention.py:417-436,826-833 of the reference, mmcv's FFN) and the
// 1x1x1 layers of the neck (necks/imvoxelnet.py:36-64).  All of them have K <= 512 and 10^3 .. 10^5 rows: on the
// tile-per-workgroup implicit-GEMM kernel a K = 256 tile is a serial chain (weight + activation prologue, 8 short
// K-steps with one global-load latency each, store-heavy epilogue) and ran at 0.3 of its HBM floor on the large calls
// and at a 17 us latency floor on the small ones.
//
// Structure (CDNA4-first, nothing of it exists in the reference, which calls cuBLAS through torch):
//   * a WAVE owns 32 output columns for the whole kernel and keeps their weights -- hi and lo planes, all of K -- as
//     MFMA B fragments in REGISTERS (K = 256: 128 VGPRs).  No weight traffic after the prologue, no B fragment reads
//     from LDS, no weight staging barrier.
//   * a workgroup (NW waves = NW * 32 columns) walks 32-row tiles of x persistently (tile = stripe + i * stripes);
//     the tile is loaded fp32 (16 B per lane, one row = one contiguous wave access), split once into bf16 hi / lo and
//     written to a double-buffered LDS image; loads run DEPTH tiles ahead in registers so that the HBM latency of
//     tile i + DEPTH is hidden behind the MFMA phases of tiles i .. i + DEPTH - 1.
//   * A fragments: one ds_read_b128 per plane per 16-deep k-step (row pitch K + 8 bf16: conflict-free), 3 MFMAs per
//     k-step on ONE accumulator chain (v_mfma_f32_32x32x16_bf16 issues back to back on a single chain).
//   * epilogue straight from the accumulator registers: in the 32x32 C layout a store instruction covers two rows x
//     32 columns = two whole 128-byte lines, so neither LDS staging nor a second barrier is needed; the stores of
//     tile i drain while tile i + 1 computes (they are issued after the loads of the tiles ahead, so the counted
//     vmcnt of the next conversion does not wait for them).
//   * ONE barrier per tile.
// A row's result does not depend on the number of rows, the grid or the tile a row falls in (fixed k order, no split-K):
// the host-counted and the device-counted (m_dev) launches are bit-identical, and so are eager and graph replays.
// The k order and the product order are those of conv3d_igemm_bf16x3_kernel: both kernels give identical bits.
#include <stdlib.h>

#include "common.hpp"
#include "mma.hpp"
#include "diag.hpp"

namespace sgc {

int g_tune_rows_gemm = 1;        // 0: every row GEMM on the tile-per-workgroup implicit-GEMM kernel (round-2 path)
int g_tune_rows_diag = 0;        // TIMING EXPERIMENTS ONLY, honoured only with SGC_DIAG=1 in the environment (results are then
                                 // invalid): bit 0 = stores dropped by the range check, bit 1 = loads dropped (zeros), bit 2 = no MFMA
int g_tune_rows_cu_pct = 100;   // persistent row GEMM: share of the CUs it occupies (it is memory-bound: with scenes in flight the rest serve MFMA kernels)
int g_tune_rows_depth = 1;       // 8-wave form: 1 / 2 = lockstep with that many tiles in flight ahead of the one being multiplied,
                                 // 0 = staggered halves (waves 4-7 half a period behind waves 0-3); the 4-wave form (two workgroups
                                 // per CU) is lockstep, 1 ahead.  Interleaved A/B on the 204,800 x 256 -> 256 Linear (3 rounds x 40
                                 // launches): lockstep-1 90-93 us row-major / 94-98 head-major, lockstep-2 92-93 / 98-101, staggered
                                 // 96-100 / 97-104 (in-kernel stamps: a staging phase issues ~350 instructions per wave and tile and
                                 // slows the partner wave's MFMA chain from 1536 to 2000-3000 cycles, so separating the phases in
                                 // time does not pay); the memory-only form of the kernel (no MFMA) takes 80 us = 5.2 TB/s


struct RowsGemmParams {
  const float *x;              // [M, K] rows, row stride ldx floats
  int64_t ldx;
  const __bf16 *w_hi, *w_lo;   // [N][K]
  const float *scale, *shift;  // [N] or null
  const float *residual;       // [M, N] or null
  void *y;                     // [M, N] fp32; head-major [cam][head][s][cm] fp32 / bf16 when hm_cm > 0
  const int32_t *m_dev;        // live row count on the device or null
  int M, N;
  int relu;                    // 0 none, 1 relu(y + residual), 2 relu(y) + residual (conv3d.hip's modes)
  int hm_S, hm_cm, hm_bf16;
  int ncg;                     // column groups of NW * 32 columns
  int diag;                    // see g_tune_rows_diag
  unsigned long long *stamps;  // diagnostic builds only
  int64_t y_bytes;             // head-major output: bytes of the whole buffer (the range the stores are checked against)
  float *zero_row;             // optional: N floats the launch sets to zero (the all-zero row behind the value map the wave gather
                               // points out-of-image corners at, sgc_linear_rows_zrow_bf16x3) -- by workgroup 0, before its tiles
  // GATHER form (sgc_pairs_geometry_linear_bf16x3): row r of the A operand is sum_k gw[r][k] * x[go[r][k]][:] -- the geometry-aware
  // sample of a visible pair, built while the tile is staged instead of written to HBM by one kernel and read back by this one
  const float *gw;             // [M][4] corner weights (bilinear * depth score)
  const int32_t *go;           // [M][4] row of x of each corner (a valid row also where the weight is 0)
  int64_t x_rows;              // rows of x (the whole feature map), GATHER only
};

constexpr int RG_ROWS = 32;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned RG_OOB = 0xfffffff0u;    // a byte offset no buffer of < 4 GiB reaches: the load returns 0, the store is dropped

// EPI: 0 = y = acc * scale + shift (optional relu), row-major; 1 = head-major fp32 store (value_proj); 2 = row-major with
// residual; 3 = head-major bf16 store (opt-in storage mode)
template <int K, int NW, int DEPTH, int EPI, int NP = 3, bool GATHER = false>   // NP: bf16 products per multiply-add (conv3d.hip: g_conv_products)
__global__ __launch_bounds__(NW * 64, 2) void rows_gemm_bf16x3_kernel(const RowsGemmParams p) {
  static_assert(!GATHER || (DEPTH == 1 && EPI == 0), "the gather form runs the lockstep schedule with the plain epilogue");
  constexpr int NT = NW * 64;
  constexpr int KS = K / 16;                       // 16-deep k-steps
  constexpr int K4 = K / 4;                        // float4 chunks per row
  constexpr int CH = RG_ROWS * K4 / NT;            // float4 chunks per thread per tile
  constexpr int PITCH = K + 8;                     // bf16 per LDS row: 16-byte pad -> conflict-free ds_read_b128 fragments
  constexpr int PLANE = RG_ROWS * PITCH;
  static_assert(RG_ROWS * K4 % NT == 0, "tile must deal evenly");
  extern __shared__ __attribute__((aligned(16))) unsigned char rg_smem[];
  __bf16 *lds = reinterpret_cast<__bf16 *>(rg_smem);   // [2 buffers][hi | lo][32][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  if (p.zero_row && blockIdx.x == 0)
    for (int c = tid; c < p.N; c += NT) p.zero_row[c] = 0.f;
  const int Mrows = p.m_dev ? min(p.M, *p.m_dev) : p.M;
  const int ntiles = (Mrows + RG_ROWS - 1) / RG_ROWS;
  // blocks that share an XCD (equal blockIdx % 8) take the column groups of the same stripes: the second group's
  // read of a tile hits the L2 the first one filled.  Speed only.
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int cg = loc % p.ncg, stripe = (loc / p.ncg) * 8 + xcd, nstripes = gridDim.x / p.ncg;
  if (stripe >= ntiles) return;
  const int col = (cg * NW + wid) * 32 + fr;       // this lane's output column

  // Buffer descriptors over the LIVE rows: a load past them returns zeros and a store past them is dropped by the
  // range check, so the tile loop has no branch (every wave issues the same memory instructions every iteration:
  // the compiler's counted vmcnt stays exact and the loads of the tiles ahead stay in flight across the waits).
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.x), 0, (p.diag & 2) ? 0 : (int)(unsigned)((int64_t)(GATHER ? p.x_rows : Mrows) * p.ldx * 4), 0x00020000);
  // gather descriptors over the LIVE rows: past them both loads return zeros (weights 0, row 0: a valid row)
  const __amdgpu_buffer_rsrc_t gwr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(GATHER ? p.gw : p.x), 0,
                                                                        GATHER ? (int)(unsigned)((int64_t)Mrows * 16) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t gor = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(GATHER ? p.go : (const int32_t *)p.x), 0,
                                                                        GATHER ? (int)(unsigned)((int64_t)Mrows * 16) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
      p.y, 0, (p.diag & 1) ? 0 : (int)(unsigned)((EPI == 1 || EPI == 3) ? p.y_bytes : (int64_t)Mrows * p.N * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(EPI == 2 ? p.residual : p.x), 0, (int)(unsigned)(EPI == 2 ? (int64_t)Mrows * p.N * 4 : 0), 0x00020000);

  // ---- weights: all of K for this wave's 32 columns, as B fragments in registers ----
  bf16x8 bh[KS], bl[KS];
  {
    const __bf16 *wh = p.w_hi + (int64_t)col * K + fh * 8, *wl = p.w_lo + (int64_t)col * K + fh * 8;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      bh[kk] = *reinterpret_cast<const bf16x8 *>(wh + kk * 16);
      if constexpr (NP == 3) bl[kk] = *reinterpret_cast<const bf16x8 *>(wl + kk * 16);
    }
  }
  const float sc = p.scale ? p.scale[col] : 1.f, sh = p.shift ? p.shift[col] : 0.f;
  const bool relu = p.relu != 0, relu1 = p.relu == 1, relu2 = p.relu == 2;

  // Staging deal.  NTL threads cooperate on RL rows of a tile: all NT threads on all 32 rows, or -- staggered form
  // (DEPTH == 0, 8 waves) -- each half of the workgroup (waves 0-3 / 4-7) on its own 16 rows.  Chunk i of a thread:
  // f = tl + i * NTL -> (row, 16-byte chunk) = (tl / K4 + i * (NTL / K4), tl % K4): one per-lane offset plus a SCALAR
  // multiple of the row pitch per chunk (made opaque per call: not hoisted into CH long-lived VGPRs)
  constexpr bool STAG = DEPTH == 0;
  static_assert(!STAG || NW == 8, "the staggered form pairs waves w and w + 4 of an 8-wave workgroup");
  constexpr int NTL = STAG ? NT / 2 : NT, RL = STAG ? RG_ROWS / 2 : RG_ROWS;
  static_assert(NTL % K4 == 0 && RL * K4 / NTL == CH, "chunks of a thread must share their column");
  const int late = STAG ? __builtin_amdgcn_readfirstlane(wid >> 2) : 0;      // 1: waves 4-7, half a period behind
  const int tl = STAG ? (tid & (NTL - 1)) : tid;
  const int ld_row = tl / K4 + late * RL, ld_c4 = tl % K4;
  // A tile in flight: CH 16-byte chunks per thread -- or, GATHER, the four corner rows' chunks and their weights
  constexpr int TV = GATHER ? 4 * CH + CH : CH;
  auto load_tile = [&](int t, float4 (&ra)[TV]) {
    int ldx4 = (int)p.ldx * 4;
    asm volatile("" : "+s"(ldx4));
    if constexpr (GATHER) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const unsigned doff = t < ntiles ? (unsigned)(t * RG_ROWS + ld_row + i * (NTL / K4)) * 16u : RG_OOB;
        const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(gor, doff, 0, 0);
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(gwr, doff, 0, 0);
        ra[4 * CH + i] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, o[k] * (unsigned)ldx4 + ld_c4 * 16, 0, 0);
          ra[4 * i + k] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
      }
      return;
    }
    const unsigned base = t < ntiles ? (unsigned)(t * RG_ROWS + ld_row) * (unsigned)ldx4 + ld_c4 * 16 : RG_OOB;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, t < ntiles ? base + i * (NTL / K4) * ldx4 : RG_OOB, 0, 0);
      ra[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
  };
  auto split_tile = [&](const float4 (&ra)[TV], int buf) {
    __bf16 *a_hi = lds + buf * 2 * PLANE, *a_lo = a_hi + PLANE;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int row = ld_row + i * (NTL / K4), c4 = ld_c4;
      float v[4];
      if constexpr (GATHER) {
        // the geometry sample's arithmetic (dfa3d_fwd_kernel<kPairsGeom>: acc += w[k] * v[k] over the corners in order, contracted
        // to fmas): the staged row is the value sgc_pairs_geometry_sample would have written, bit for bit
        const float wk[4] = {ra[4 * CH + i].x, ra[4 * CH + i].y, ra[4 * CH + i].z, ra[4 * CH + i].w};
        v[0] = v[1] = v[2] = v[3] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[0] = __builtin_fmaf(wk[k], ra[4 * i + k].x, v[0]); v[1] = __builtin_fmaf(wk[k], ra[4 * i + k].y, v[1]);
          v[2] = __builtin_fmaf(wk[k], ra[4 * i + k].z, v[2]); v[3] = __builtin_fmaf(wk[k], ra[4 * i + k].w, v[3]);
        }
      } else {
        v[0] = ra[i].x; v[1] = ra[i].y; v[2] = ra[i].z; v[3] = ra[i].w;
      }
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = op_hi<NP>(v[e]);
        h[e] = hb;
        l[e] = op_lo<NP>(v[e], hb);
      }
      *reinterpret_cast<bf16x4 *>(a_hi + row * PITCH + c4 * 4) = h;
      if constexpr (NP == 3) *reinterpret_cast<bf16x4 *>(a_lo + row * PITCH + c4 * 4) = l;
    }
  };
  // A fragments are read PD k-steps ahead of the MFMAs that use them (ring of PD + 1 register slots, static indices
  // after unrolling); the scheduling barrier per step keeps that distance in the emitted code (left alone, the
  // compiler issues each read one step ahead: ~96 MFMA cycles of cover for an LDS round trip, which a single wave per
  // SIMD -- the staggered form -- cannot hide)
  constexpr int PD = 3;
  auto multiply = [&](int buf, f32x16 &acc) {
    const __bf16 *a_hi = lds + buf * 2 * PLANE + fr * PITCH + fh * 8, *a_lo = a_hi + PLANE;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    if (p.diag & 4) return;
    bf16x8 ah[PD + 1], al[PD + 1];
#pragma unroll
    for (int kk = 0; kk < PD; ++kk) {
      ah[kk] = *reinterpret_cast<const bf16x8 *>(a_hi + kk * 16);
      if constexpr (NP == 3) al[kk] = *reinterpret_cast<const bf16x8 *>(a_lo + kk * 16);
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if (kk + PD < KS) {
        ah[(kk + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8 *>(a_hi + (kk + PD) * 16);
        if constexpr (NP == 3) al[(kk + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8 *>(a_lo + (kk + PD) * 16);
      }
      if constexpr (NP == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk % (PD + 1)], bh[kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk % (PD + 1)], bl[kk], acc, 0, 0, 0);
      }
      acc = mma_hh<NP>(ah[kk % (PD + 1)], bh[kk], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // head-major constants: row m = cam * S + s, column = head * cm + j  ->  [cam][head][s][cm]
  constexpr bool HM = EPI == 1 || EPI == 3;
  const int hm_heads = HM ? p.N / p.hm_cm : 1, hm_head = HM ? col / p.hm_cm : 0;
  const int hm_j = HM ? col - hm_head * p.hm_cm : 0;
  auto store_tile = [&](int t, const f32x16 &acc) {
    const int m0 = t * RG_ROWS;
    if constexpr (HM) {
      // rows of a tile are consecutive pixels of a camera: byte offset = base(tile, lane) + row * cm * es, plus one
      // constant jump for the rows past a camera border (hm_S >= 32: at most one border per tile)
      constexpr int es = EPI == 3 ? 2 : 4;
      const int cam0 = m0 / p.hm_S, s0 = m0 - cam0 * p.hm_S;       // scalar: one division per tile
      int cme = p.hm_cm * es;
      asm volatile("" : "+s"(cme));
      const unsigned base = (unsigned)(((cam0 * hm_heads + hm_head) * p.hm_S + s0 + 4 * fh) * p.hm_cm + hm_j) * es;
      const unsigned jump = (unsigned)((hm_heads - 1) * p.hm_S) * (unsigned)cme;
      const int rows_left = p.hm_S - s0 - 4 * fh;                  // rows of camera cam0 from this lane's first row on
      const int live = Mrows - m0 - 4 * fh;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int r = (k & 3) + 8 * (k >> 2);                      // row - 4 * fh
        unsigned off = base + r * cme;
        off += r >= rows_left ? jump : 0u;
        off = r < live ? off : RG_OOB;
        const float v = acc[k] + sh;
        if constexpr (EPI == 3) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), yr, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, off, 0, 0);
      }
    } else {
      // row offsets as SCALAR multiples of the row pitch added to one per-lane base: the pitch is made opaque per call
      // so that the 16 offsets are not hoisted out of the tile loop into 16 long-lived VGPRs (the kernel sits at the
      // 256-register limit of two waves per SIMD)
      int n4 = p.N * 4;
      asm volatile("" : "+s"(n4));
      float res[16];
      const unsigned base = (unsigned)(m0 + 4 * fh) * (unsigned)n4 + col * 4;
      if constexpr (EPI == 2) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          res[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, base + ((k & 3) + 8 * (k >> 2)) * n4, 0, 0));
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float v = acc[k] * sc;                              // two roundings (product, then sum) like the staged epilogue of
        asm volatile("" : "+v"(v));                          // conv3d.hip and the oracle's plain C: the empty asm keeps the
        v += sh;                                             // compiler from contracting them into one fma
        if constexpr (EPI == 2) {
          v = relu2 ? fmaxf(v, 0.f) : v;
          v += res[k];
          v = relu1 ? fmaxf(v, 0.f) : v;
        } else {
          v = relu ? fmaxf(v, 0.f) : v;
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, base + ((k & 3) + 8 * (k >> 2)) * n4, 0, 0);
      }
    }
  };

  // Loop shape.  The compiler's counted s_waitcnt vmcnt(N) in front of a split must leave the loads of the tiles ahead
  // (and the stores of the tiles behind) in flight; its count is exact only where every path into a block has issued
  // the same vector-memory operations.  Hence: the first iteration (pair) is peeled, the loop has ONE exit at its
  // bottom (a break in the middle leaves a structurizer edge from the break to the header on which the registers
  // being waited for were "just reloaded": the waits then degrade to "all but the last 16"), and an odd last tile
  // runs in a tail copy behind the loop.
  f32x16 acc;
  const int n = (ntiles - stripe + nstripes - 1) / nstripes;     // tiles of this workgroup, >= 1
  int t = stripe;                                                // tile multiplied in the current iteration
  if constexpr (STAG) {
    // Staggered halves.  Waves w and w + 4 share a SIMD; a period has two half-periods separated by a barrier: in the
    // first, waves 0-3 multiply tile i while waves 4-7 stage (store tile i - 1, split their 16 rows of tile i + 1, load
    // tile i + 2); in the second they swap.  Each SIMD then always holds one wave in its MFMA chain and one in the
    // VALU / LDS-write / memory-issue part, which a lockstep schedule (both waves multiply, then both stage) serialises
    // (measured on the 204,800-row Linear: staging 30 us + MFMA 33 us + memory, all additive).  Two LDS buffers are
    // enough: the half of tile i + 1 staged by waves 4-7 during the first half-period lands in the buffer of tile
    // i - 1, whose last reader (their own multiply) finished a barrier earlier.
    float4 r0[TV];
    RG_STAMP_PTR(p, wid, late);
    load_tile(t, r0);
    split_tile(r0, 0);
    load_tile(t + nstripes, r0);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      RG_STAMP(0);
      if (!late) {
        multiply(i & 1, acc);
        asm volatile("" :: "v"(acc[0]));
        RG_STAMP(1);
      } else {
        if (i > 0) store_tile(t - nstripes, acc);
        RG_STAMP(1);
        split_tile(r0, (i + 1) & 1);
        RG_STAMP(2);
        load_tile(t + 2 * nstripes, r0);
        RG_STAMP(3);
      }
      __syncthreads();
      RG_STAMP(4);
      if (!late) {
        split_tile(r0, (i + 1) & 1);
        RG_STAMP(5);
        load_tile(t + 2 * nstripes, r0);
        store_tile(t, acc);
        RG_STAMP(6);
      } else {
        multiply(i & 1, acc);
        asm volatile("" :: "v"(acc[0]));
        RG_STAMP(5);
      }
      t += nstripes;
      __syncthreads();
      RG_STAMP(7);
    }
    if (late) store_tile(t - nstripes, acc);
  } else if constexpr (DEPTH == 2) {
    float4 r0[TV], r1[TV];                 // tile j of this workgroup lives in set j & 1 until it is split
    load_tile(t, r0);
    load_tile(t + nstripes, r1);
    split_tile(r0, 0);
    load_tile(t + 2 * nstripes, r0);
    __syncthreads();
    // even iteration: tile t in buffer 0, t + ns in r1, t + 2 ns in r0 (in flight); odd: the mirror image
    auto even = [&]() {
      multiply(0, acc);
      split_tile(r1, 1);
      load_tile(t + 3 * nstripes, r1);
      store_tile(t, acc);
      t += nstripes;
    };
    auto odd = [&]() {
      multiply(1, acc);
      split_tile(r0, 0);
      load_tile(t + 3 * nstripes, r0);
      store_tile(t, acc);
      t += nstripes;
    };
    even();
    if (n == 1) return;
    __syncthreads();
    odd();
    if (n == 2) return;
    __syncthreads();
    const int pairs = (n - 2) >> 1;
    for (int i = 0; i < pairs; ++i) {
      even();
      __syncthreads();
      odd();
      __syncthreads();
    }
    if (n & 1) even();
  } else {
    float4 r0[TV];
    int buf = 0;
    load_tile(t, r0);
    split_tile(r0, 0);
    load_tile(t + nstripes, r0);
    __syncthreads();
    auto step = [&]() {
      multiply(buf, acc);
      split_tile(r0, buf ^ 1);
      load_tile(t + 2 * nstripes, r0);
      store_tile(t, acc);
      t += nstripes;
      buf ^= 1;
    };
    step();
    for (int i = 1; i < n; ++i) {
      __syncthreads();
      step();
    }
  }
}

extern int g_conv_products;      // conv3d.hip

// ---------------------------------------------------------------------------------------------
// Gather form at K = 256 (round 6): PRODUCER and CONSUMER waves.  In the kernel above a wave stages its share of the next tile AND
// multiplies the current one: at K = 256 with the gather that is the resident weights of its 32 columns (128 registers) plus a tile in
// flight of 8 chunks x 4 corner rows (128 registers + weights of the corners) -- 388 bytes of scratch per lane.  Here the two jobs
// sit on different waves of an 8-wave workgroup, so neither register set meets the other:
//   waves 0-3 (consumers)  hold the weights of 32 columns each (N = 128) and multiply tile i from LDS buffer i & 1, then store it;
//   waves 4-7 (producers)  request the corner rows of tile i + 2, build tile i + 1's rows (the sample's fmas), split them to bf16 hi | lo
//                          and write LDS buffer (i + 1) & 1;
// ONE barrier per tile: behind barrier i buffer i & 1 holds tile i and every consumer is done with buffer (i + 1) & 1 (tile i - 1).
// Same A values (the fmas of the geometry sample), same k order and product order as the plain kernel: bit-identical results.
// ---------------------------------------------------------------------------------------------
template <int K, int NP>
__global__ __launch_bounds__(512) void rows_gemm_gather_pc_kernel(const RowsGemmParams p) {
  constexpr int KS = K / 16, K4 = K / 4, PITCH = K + 8, PLANE = RG_ROWS * PITCH;
  constexpr int NTP = 256;                          // producer threads
  constexpr int CH = RG_ROWS * K4 / NTP;            // chunks per producer thread and tile (8 at K = 256)
  static_assert(RG_ROWS * K4 % NTP == 0 && NTP % K4 == 0, "tile must deal evenly");
  extern __shared__ __attribute__((aligned(16))) unsigned char rg_smem[];
  __bf16 *lds = reinterpret_cast<__bf16 *>(rg_smem);   // [2 buffers][hi | lo][32][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int Mrows = p.m_dev ? min(p.M, *p.m_dev) : p.M;
  const int ntiles = (Mrows + RG_ROWS - 1) / RG_ROWS;
  const int stripe = blockIdx.x, nstripes = gridDim.x;           // N == 128: one column group
  if (stripe >= ntiles) return;
  const int n = (ntiles - stripe + nstripes - 1) / nstripes;    // tiles of this workgroup, >= 1
  const bool consumer = wid < 4;
  if (consumer) {
    const int col = wid * 32 + fr;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(unsigned)((int64_t)Mrows * p.N * 4), 0x00020000);
    bf16x8 bh[KS], bl[KS];
    {
      const __bf16 *wh = p.w_hi + (int64_t)col * K + fh * 8, *wl = p.w_lo + (int64_t)col * K + fh * 8;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        bh[kk] = *reinterpret_cast<const bf16x8 *>(wh + kk * 16);
        if constexpr (NP == 3) bl[kk] = *reinterpret_cast<const bf16x8 *>(wl + kk * 16);
      }
    }
    const float sc = p.scale ? p.scale[col] : 1.f, sh = p.shift ? p.shift[col] : 0.f;
    constexpr int PD = 3;
    int t = stripe;
    for (int i = 0; i < n; ++i, t += nstripes) {
      __syncthreads();                                           // tile i is in buffer i & 1
      const __bf16 *a_hi = lds + (i & 1) * 2 * PLANE + fr * PITCH + fh * 8, *a_lo = a_hi + PLANE;
      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
      bf16x8 ah[PD + 1], al[PD + 1];
#pragma unroll
      for (int kk = 0; kk < PD; ++kk) {
        ah[kk] = *reinterpret_cast<const bf16x8 *>(a_hi + kk * 16);
        if constexpr (NP == 3) al[kk] = *reinterpret_cast<const bf16x8 *>(a_lo + kk * 16);
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        if (kk + PD < KS) {
          ah[(kk + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8 *>(a_hi + (kk + PD) * 16);
          if constexpr (NP == 3) al[(kk + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8 *>(a_lo + (kk + PD) * 16);
        }
        if constexpr (NP == 3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk % (PD + 1)], bh[kk], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk % (PD + 1)], bl[kk], acc, 0, 0, 0);
        }
        acc = mma_hh<NP>(ah[kk % (PD + 1)], bh[kk], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      int n4 = p.N * 4;
      asm volatile("" : "+s"(n4));
      const unsigned base = (unsigned)(t * RG_ROWS + 4 * fh) * (unsigned)n4 + col * 4;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float v = acc[k] * sc;                              // two roundings, as the plain kernel's epilogue
        asm volatile("" : "+v"(v));
        v += sh;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, base + ((k & 3) + 8 * (k >> 2)) * n4, 0, 0);
      }
    }
    __syncthreads();                                             // the producers' last barrier
    return;
  }
  // ---------------- producers ----------------
  const int pt = tid - 256;
  const int ld_row = pt / K4, ld_c4 = pt % K4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)(unsigned)(p.x_rows * p.ldx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t gwr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gw), 0, (int)(unsigned)((int64_t)Mrows * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t gor = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.go), 0, (int)(unsigned)((int64_t)Mrows * 16), 0x00020000);
  float4 rv[4 * CH], rw[CH];
  auto request = [&](int t) {
    int ldx4 = (int)p.ldx * 4;
    asm volatile("" : "+s"(ldx4));
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const unsigned doff = t < ntiles ? (unsigned)(t * RG_ROWS + ld_row + i * (NTP / K4)) * 16u : RG_OOB;
      const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(gor, doff, 0, 0);
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(gwr, doff, 0, 0);
      rw[i] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, o[k] * (unsigned)ldx4 + ld_c4 * 16, 0, 0);
        rv[4 * i + k] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
      }
    }
  };
  auto build = [&](int buf) {
    __bf16 *a_hi = lds + buf * 2 * PLANE, *a_lo = a_hi + PLANE;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int row = ld_row + i * (NTP / K4);
      const float wk[4] = {rw[i].x, rw[i].y, rw[i].z, rw[i].w};
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[0] = __builtin_fmaf(wk[k], rv[4 * i + k].x, v[0]); v[1] = __builtin_fmaf(wk[k], rv[4 * i + k].y, v[1]);
        v[2] = __builtin_fmaf(wk[k], rv[4 * i + k].z, v[2]); v[3] = __builtin_fmaf(wk[k], rv[4 * i + k].w, v[3]);
      }
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = op_hi<NP>(v[e]);
        h[e] = hb;
        l[e] = op_lo<NP>(v[e], hb);
      }
      *reinterpret_cast<bf16x4 *>(a_hi + row * PITCH + ld_c4 * 4) = h;
      if constexpr (NP == 3) *reinterpret_cast<bf16x4 *>(a_lo + row * PITCH + ld_c4 * 4) = l;
    }
  };
  int t = stripe;
  request(t);
  build(0);
  request(t + nstripes);
  for (int i = 0; i < n; ++i) {
    __syncthreads();                                             // tile i published; buffer (i + 1) & 1 is free
    if (i + 1 < n) {
      build((i + 1) & 1);
      request(t + (i + 2) * nstripes);
    }
  }
  __syncthreads();
}

template <int K, int NW, int DEPTH, int EPI, int NP>
static int launch_rows_gemm_np(const RowsGemmParams &p, int grid, hipStream_t st) {
  constexpr int smem = 2 * 2 * RG_ROWS * (K + 8) * (int)sizeof(uint16_t);
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void *)rows_gemm_bf16x3_kernel<K, NW, DEPTH, EPI, NP>, smem, attr_done);
  hipLaunchKernelGGL((rows_gemm_bf16x3_kernel<K, NW, DEPTH, EPI, NP>), dim3(grid), dim3(NW * 64), smem, st, p);
  return check_launch("rows_gemm_bf16x3_kernel");
}

template <int K, int NW, int DEPTH, int EPI>
static int launch_rows_gemm_e(const RowsGemmParams &p, int grid, hipStream_t st) {
  if (g_conv_products == 1) return launch_rows_gemm_np<K, NW, 1, EPI, 1>(p, grid, st);      // single-product modes: the lockstep-1 form
  if (g_conv_products == 2) return launch_rows_gemm_np<K, NW, 1, EPI, 2>(p, grid, st);
  return launch_rows_gemm_np<K, NW, DEPTH, EPI, 3>(p, grid, st);
}

template <int K, int NW, int DEPTH>
static int launch_rows_gemm(const RowsGemmParams &p, int grid, hipStream_t st) {
  if (p.hm_cm > 0 && p.hm_bf16) return launch_rows_gemm_e<K, NW, DEPTH, 3>(p, grid, st);
  if (p.hm_cm > 0) return launch_rows_gemm_e<K, NW, DEPTH, 1>(p, grid, st);
  if (p.residual) return launch_rows_gemm_e<K, NW, DEPTH, 2>(p, grid, st);
  return launch_rows_gemm_e<K, NW, DEPTH, 0>(p, grid, st);
}

// Can the persistent kernel take this GEMM?  K in {128, 256}; whole 128-column groups; 32-bit byte offsets into x and y;
// head-major: a camera holds at least one tile (the store handles one camera border per tile) and a wave's 32 columns
// are whole heads or part of one.
bool rows_gemm_supported(int K, int N, int hm_cm, int hm_S, int64_t rows, int64_t ldx) {
  if (!g_tune_rows_gemm) return false;
  if (K != 128 && K != 256) return false;
  if (N <= 0 || N % 128) return false;
  if (rows <= 0 || (rows + 64) * ldx * 4 >= (int64_t)RG_OOB || (rows + 64) * N * 4 >= (int64_t)RG_OOB) return false;
  if (hm_cm > 0 && (128 % hm_cm || hm_S < RG_ROWS)) return false;      // per-lane head / offset arithmetic: any hm_cm | 128
  return true;
}

int device_cus() {      // also conv3d.hip (the wave-quantisation model of the halo kernel)
  static std::atomic<int> cached{0};
  int c = cached.load(std::memory_order_relaxed);
  if (c > 0) return c;
  int dev = 0;
  hipDeviceProp_t prop;
  c = 256;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    c = prop.multiProcessorCount;
  cached.store(c, std::memory_order_relaxed);
  return c;
}

// Gather form (see RowsGemmParams): N == 128 columns; K == 128 on the kernel above (four waves, the lockstep schedule), K == 256 on
// the producer / consumer kernel (rows_gemm_gather_pc_kernel).
bool rows_gemm_gather_supported(int K, int N, int64_t x_rows, int64_t rows) {
  return rows_gemm_supported(K, N, 0, 0, rows, K) && N == 128 && x_rows > 0 && (x_rows + 1) * K * 4 < (int64_t)RG_OOB;
}

template <int NP>
static void launch_gather_pc(const RowsGemmParams &p, int grid, hipStream_t st) {
  constexpr int smem = 2 * 2 * RG_ROWS * (256 + 8) * (int)sizeof(uint16_t);
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void *)rows_gemm_gather_pc_kernel<256, NP>, smem, attr_done);
  hipLaunchKernelGGL((rows_gemm_gather_pc_kernel<256, NP>), dim3(grid), dim3(512), smem, st, p);
}

template <int K>
static int launch_rows_gemm_gather(const RowsGemmParams &p, int grid, hipStream_t st) {
  constexpr int smem = 2 * 2 * RG_ROWS * (K + 8) * (int)sizeof(uint16_t);
  static std::atomic<uint64_t> attr_done{0};
  if (g_conv_products == 1) {
    ensure_dynamic_lds((const void *)rows_gemm_bf16x3_kernel<K, 4, 1, 0, 1, true>, smem, attr_done);
    hipLaunchKernelGGL((rows_gemm_bf16x3_kernel<K, 4, 1, 0, 1, true>), dim3(grid), dim3(256), smem, st, p);
  } else if (g_conv_products == 2) {
    static std::atomic<uint64_t> attr2{0};
    ensure_dynamic_lds((const void *)rows_gemm_bf16x3_kernel<K, 4, 1, 0, 2, true>, smem, attr2);
    hipLaunchKernelGGL((rows_gemm_bf16x3_kernel<K, 4, 1, 0, 2, true>), dim3(grid), dim3(256), smem, st, p);
  } else {
    static std::atomic<uint64_t> attr3{0};
    ensure_dynamic_lds((const void *)rows_gemm_bf16x3_kernel<K, 4, 1, 0, 3, true>, smem, attr3);
    hipLaunchKernelGGL((rows_gemm_bf16x3_kernel<K, 4, 1, 0, 3, true>), dim3(grid), dim3(256), smem, st, p);
  }
  return check_launch("rows_gemm_bf16x3_kernel (gather)");
}

int rows_gemm_gather_launch(const float *x, int64_t x_rows, const float *gw, const int32_t *go, const uint16_t *w_hi, const uint16_t *w_lo,
                            const float *shift, float *y, const int32_t *m_dev, int M, int K, int N, hipStream_t st) {
  RowsGemmParams p = {};
  p.x = x; p.ldx = K; p.x_rows = x_rows; p.gw = gw; p.go = go;
  p.w_hi = reinterpret_cast<const __bf16 *>(w_hi); p.w_lo = reinterpret_cast<const __bf16 *>(w_lo);
  p.shift = shift; p.y = y; p.m_dev = m_dev; p.M = M; p.N = N;
  p.y_bytes = (int64_t)M * N * 4;
  p.ncg = N / 128;
  const int cap_tiles = ceil_div(M, RG_ROWS);
  int stripes = device_cus() * 2 / p.ncg;                       // two 4-wave workgroups per CU
  if (g_tune_rows_cu_pct > 0 && g_tune_rows_cu_pct < 100) stripes = stripes * g_tune_rows_cu_pct / 100;
  if (stripes > cap_tiles) stripes = cap_tiles;
  stripes = (stripes + 7) / 8 * 8;
  const int grid = stripes * p.ncg;
  if (K == 256) {
    int s8 = device_cus();                                      // one 8-wave workgroup per CU
    if (g_tune_rows_cu_pct > 0 && g_tune_rows_cu_pct < 100) s8 = s8 * g_tune_rows_cu_pct / 100;
    if (s8 > cap_tiles) s8 = cap_tiles;
    if (g_conv_products == 1) launch_gather_pc<1>(p, s8, st);
    else if (g_conv_products == 2) launch_gather_pc<2>(p, s8, st);
    else launch_gather_pc<3>(p, s8, st);
    return check_launch("rows_gemm_gather_pc_kernel");
  }
  return launch_rows_gemm_gather<128>(p, grid, st);
}

int rows_gemm_launch(const float *x, int64_t ldx, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                     const float *shift, const float *residual, void *y, const int32_t *m_dev, int M, int K, int N, int relu,
                     int hm_S, int hm_cm, int hm_bf16, hipStream_t st, float *zero_row) {
  RowsGemmParams p = {};
  p.zero_row = zero_row;
  p.x = x; p.ldx = ldx;
  p.w_hi = reinterpret_cast<const __bf16 *>(w_hi); p.w_lo = reinterpret_cast<const __bf16 *>(w_lo);
  p.scale = scale; p.shift = shift; p.residual = residual; p.y = y; p.m_dev = m_dev;
  p.M = M; p.N = N; p.relu = relu;
  p.hm_S = hm_S; p.hm_cm = hm_cm; p.hm_bf16 = hm_bf16;
  p.y_bytes = (int64_t)M * N * (hm_bf16 ? 2 : 4);
  {
    static const bool diag_ok = getenv("SGC_DIAG") && atoi(getenv("SGC_DIAG")) == 1;
    p.diag = diag_ok ? g_tune_rows_diag : 0;
  }
  RG_STAMP_BIND(p);
  const int nw = (N % 256 == 0) ? 8 : 4;
  p.ncg = N / (nw * 32);
  const int cap_tiles = ceil_div(M, RG_ROWS);
  // one 8-wave workgroup or two 4-wave workgroups per CU; stripes are a multiple of 8 (the XCD-aware deal of the kernel)
  const int per_cu = nw == 8 ? 1 : 2;
  int stripes = device_cus() * per_cu / p.ncg;
  if (g_tune_rows_cu_pct > 0 && g_tune_rows_cu_pct < 100) stripes = stripes * g_tune_rows_cu_pct / 100;   // leave CUs to the other streams
  if (stripes > cap_tiles) stripes = cap_tiles;
  stripes = (stripes + 7) / 8 * 8;
  const int grid = stripes * p.ncg;
  const int depth = nw == 8 ? g_tune_rows_depth : 1;
  if (K == 256) {
    if (nw == 8) return depth == 0 ? launch_rows_gemm<256, 8, 0>(p, grid, st) : depth == 2 ? launch_rows_gemm<256, 8, 2>(p, grid, st) : launch_rows_gemm<256, 8, 1>(p, grid, st);
    return launch_rows_gemm<256, 4, 1>(p, grid, st);
  }
  if (nw == 8) return depth == 0 ? launch_rows_gemm<128, 8, 0>(p, grid, st) : depth == 2 ? launch_rows_gemm<128, 8, 2>(p, grid, st) : launch_rows_gemm<128, 8, 1>(p, grid, st);
  return launch_rows_gemm<128, 4, 1>(p, grid, st);
}

}  // namespace sgc

#if defined(SGC_RG_STAMPS)
extern "C" void sgc_diag_rows_stamp_buffer(unsigned long long *buf) { sgc::g_rows_stamp_buf = buf; }
#endif
