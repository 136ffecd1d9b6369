// Implicit-GEMM 3D convolution on MFMA for gfx950, channels-last volumes.
//
// Replaces the dense nn.Conv3d / ConvTranspose3d + BatchNorm3d(eval) + ReLU (+ residual) chains
// of FastIndoorImVoxelNeck (necks/imvoxelnet.py:36-64,146-173) and the three head convolutions
// (dense_heads/imvoxel_head_v2.py:75-78).  The reference runs them through cuDNN; on ROCm torch
// lowers them to MIOpen's Im3d2Col + GEMM (a 27x blown-up column buffer through HBM, measured
// 14.7 of 17.8 ms per scene at config 2).  Here:
//
//   GEMM view   out[v, co] = sum_{tap} sum_{ci} in[nbr(v, tap), ci] * W[tap][co][ci]
//               M = voxels, N = Cout, K = taps * Cin; no column buffer: the A tile of a K-step is
//               gathered straight from the channels-last volume (one tap, BK consecutive
//               channels, zero rows outside the volume) into LDS.
//   tile        128 voxels x BN channels x 32 (K) per 256-thread workgroup, 4 waves, each wave
//               (BM/WM x BN/WN) made of 32x32 MFMA tiles; fp32 operands on
//               v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
//   LDS         A[128][32+4] and B[BN][32+4] floats, K contiguous, double buffered; the +4 pad
//               makes both the ds_write_b128 rows and the ds_read_b128 fragment reads
//               conflict-free (row stride 36 dwords -> 16 distinct 4-bank slots per lane group).
//   fragments   lane (r = l&31, h = l>>5) reads 16 B at [row r][8*kk + 4*h]: the 4 floats feed 4
//               consecutive MFMAs.  A and B use the same k permutation, so the sum is unchanged.
//   pipeline    global loads of K-step s+1 are issued into registers before the MFMAs of step s
//               and written to the other LDS buffer after them (one barrier per K-step).
//   split-K     layers with few voxels (400 / 3200) split the taps over blockIdx.z and add
//               partial tiles with float atomics into a zeroed output; a small epilogue kernel
//               applies BN/ReLU/residual.  Single-pass layers fuse the epilogue.
//   epilogue    y = acc * scale[co] + shift[co], then relu mode 1: relu(y + residual) (residual
//               block), mode 2: relu(y) + residual (decoder skip add); scale/shift carry the
//               folded eval-mode BatchNorm (or the conv bias).
#include "common.hpp"
#include "mma.hpp"
#include "diag.hpp"

namespace sgc {
int g_conv_products = 3;     // NOT a tuning knob (it changes results; sgc_set_conv_products): the NP of csrc/mma.hpp -- 3 = fp32-faithful
                             // 3-way bf16 split (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi), 1 = plain bf16 (a_hi*b_hi only: operands rounded
                             // to bf16, fp32 accumulate), 2 = plain fp16 (operands rounded to IEEE half): the opt-in reduced-precision
                             // modes of BASELINE.json configs #2 / #5
int g_tune_igemm_xcd = 0;    // tile implicit GEMM, XCD deal of the split / transposed layers (ConvParams.xcd_deal)
int g_tune_conv_waves = 8;   // implicit-GEMM kernel: 4 or 8 waves per 128x128 tile
int g_tune_conv_halo = 1;    // 3x3x3 stride-1 layers: 0 per-tap kernel, 1 halo-resident kernel
int g_tune_halo_split_target = 192;   // halo kernel: channel slices are split over workgroups until a launch has this many
int g_tune_halo_2d = 1;      // 3x3 layers of sgc_conv2d_nhwc_bf16x3: 1 the 2-D form of the halo kernel (16 x 16 pixel bricks), 2 bricks of 4 images x 8 x 8
                             // where they tile the stack, 0 the tile kernel
int g_tune_halo_brick = 0;        // 0: brick shape by grid (below), 1: prefer 4x8x8, 2: force 8x8x4, 3: the round-2 rule (4x4x16 at depth >= 16).
                                  // Round 3, interleaved A/B of the three shapes on the 40x40x16 and 80x80x32 layers (bit-identical
                                  // results): 8x8x4 is 1.5 - 2.5 % faster than 4x4x16 (236 vs 241 us, 129.5 vs 133, 534 vs 546;
                                  // 600 halo rows instead of 648) -> it is the choice wherever it tiles the grid exactly
int g_tune_wgrad_waves = 8;       // weight-gradient kernel: 4 or 8 waves per 128 x 128 tile
int g_tune_split_free = 1, g_tune_split_min_steps = 8, g_tune_split_max = 32;   // tile kernel, round 6: see pick_split_steps
int g_tune_split_target = 512;    // implicit GEMM: tap groups are split until the launch has this many workgroups (interleaved A/B,
                                  // tools/split_ab.py: 128 / 256 are 20-30 % slower on the stride-2 and 400-voxel layers, 1024+ no better)
int g_tune_halo_narrow = 1;       // halo kernel: 1 = 64-column tiles for layers with <= 64 output channels and 32-column tiles (8 x 1
                                  // waves) for <= 32; 64 = never below 64 columns (the round-2..4 form, A/B); 0 = always 128
int g_tune_halo_min_m = 2048;     // fewest output voxels for the halo kernel
int g_tune_halo_min_cout = 16;   // fewest output channels for which the halo kernel (128-column tiles) is used: the head's
                                 // 28-channel convolutions run 105 -> 67 us on it although 3/4 of the tile columns are padding



// rows_gemm.hip: persistent weight-stationary form of the K <= 256 row GEMMs (every Linear of a level, the 1x1x1 layers)
bool rows_gemm_supported(int K, int N, int hm_cm, int hm_S, int64_t rows, int64_t ldx);
int device_cus();                // rows_gemm.hip: multiProcessorCount of the current device, cached
int rows_gemm_launch(const float *x, int64_t ldx, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                     const float *shift, const float *residual, void *y, const int32_t *m_dev, int M, int K, int N, int relu,
                     int hm_S, int hm_cm, int hm_bf16, hipStream_t st, float *zero_row = nullptr);

struct ConvParams {
  const float *x;         // [IV, Cin] channels-last input volume
  const float *w;         // [taps][Cout][Cin]
  const float *scale;     // [Cout] or null (= 1)
  const float *shift;     // [Cout] or null (= 0)
  const float *residual;  // [OV, Cout] or null
  float *y;               // [OV, Cout]
  int Cin, Cout;
  int ix, iy, iz;         // input grid
  int gx, gy, gz;         // GEMM-row grid (conv: output grid; transposed: input grid)
  int ksize, stride, pad; // conv geometry (transposed: ksize = 1 per parity)
  int transposed;         // 1: ConvTranspose3d k=2 s=2, parity = blockIdx.z % 8
  int relu;
  int taps;               // ksize^3
  int splitk;             // number of tap groups (divides taps); >1 -> atomic accumulate, no epilogue
  int steps_per;          // bf16x3 tile kernel, splitk > 1: K steps (32 channels of one tap) per split; the last split may hold fewer
  int M;                  // gx*gy*gz
  float *ws;              // optional split-K workspace [splitk][OV][Cout]: every split stores its partial tile there and
  int64_t ws_stride;      // the epilogue kernel sums them in split order (deterministic); null: float atomics into y
  int64_t ws_floats;      // capacity of ws
  const int32_t *m_dev;   // optional: the live row count lives on the device (sgc_linear_rows_*); rows >= *m_dev
                          // are neither read nor written and workgroups past it exit at once
  const uint8_t *out_mask; // optional [OV] {0,1}: OUTPUT mask of a 3x3x3 stride-1 layer on the halo kernel (sgc_conv3d_cl_bf16x3_masked):
                          // rows with mask 0 are not needed by the caller.  Tiles of 64 voxels (one wave) without a live row skip
                          // their MFMAs, bricks without one skip everything; what they store is the epilogue of a zero
                          // accumulator (finite, deterministic).  Live rows are bit-identical to the dense launch.
  int two_d;              // 2-D convolution over a stack of images: grid (x, y, z) = (image, row, column), the taps only span (y, z)
                          // (sgc_conv2d_nhwc_bf16x3: the FPN output convolutions)
  unsigned long long *stamps;  // diagnostic builds only (SGC_HALO_STAMPS)
  int xcd_deal;           // tile kernel: how workgroups are dealt to the 8 XCDs (hardware: linear id % 8).  0 = as launched;
                          // 1 = consecutive ROW tiles of one (column tile, split) on one XCD (they share a weight slab);
                          // 2 = consecutive COLUMN tiles of one (row tile, split) on one XCD (they share the gathered rows)
  int wz_Z;               // WZ kernels: z extent of the raw volume behind the virtual image stack (J = wz_Z / 2 pairs per position)
  int w_group_images;     // 2-D form only, > 0: the image stack is made of groups of this many images, group g convolves with the
                          // weight set w + g * taps * Cout * Cin (the four transform-domain positions of sgc_conv3d_winograd_z_bf16x3)
  float *zero_row;        // optional: Cout floats this launch sets to zero (workgroup (0, 0, 0); sgc_linear_rows_zrow_bf16x3)
  const float *act_scale; // optional: columns [act_c0, act_c1) leave as expf(v * *act_scale) -- the head's `exp(scale(reg))` (dense_heads/
  int act_c0, act_c1;     // imvoxel_head_v2.py:79,110: mmcv Scale then torch.exp) applied last in the epilogue (sgc_conv3d_cl_bf16x3_act)
  int hm_bf16;            // head-major output stored as bfloat16 (RNE of the fp32 result)
  int hm_S, hm_cm;        // hm_cm > 0: HEAD-MAJOR output of a row-list GEMM -- row r = n * hm_S + s, column c = h * hm_cm + j
                          // is stored at y[((n * (Cout / hm_cm) + h) * hm_S + s) * hm_cm + j] (sgc_linear_rows_headmajor_bf16x3)
};

// the optional output activation of a column range (ConvParams.act_*): applied after scale / shift / relu / residual
__device__ __forceinline__ float act_col(float v, int col, int c0, int c1, float s) { return (col >= c0 && col < c1) ? expf(v * s) : v; }
__device__ __forceinline__ float4 act_col4(float4 v, int col, int c0, int c1, float s) {
  if (c1 <= c0) return v;
  return make_float4(act_col(v.x, col, c0, c1, s), act_col(v.y, col + 1, c0, c1, s), act_col(v.z, col + 2, c0, c1, s), act_col(v.w, col + 3, c0, c1, s));
}

constexpr int BM = 128, BK = 32, LDK = BK + 4;

template <int BN, int WM, int WN>  // WM x WN waves; wave tile (BM/WM) x (BN/WN)
__global__ __launch_bounds__(256) void conv3d_igemm_f32_kernel(const ConvParams p) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // MFMA tiles per wave
  constexpr int BROWS = BN / 32;                       // B rows per thread (passes of 32 rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem;                      // [2][BM][LDK]
  float *Bs = smem + 2 * BM * LDK;       // [2][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  int zid = blockIdx.z;
  int parity = 0;
  if (p.transposed) { parity = zid % 8; zid /= 8; }
  const int taps_per = p.taps / p.splitk;
  const int tap_lo = zid * taps_per;
  const int ksteps_c = p.Cin / BK;
  const int nsteps = taps_per * ksteps_c;

  // --- this thread's load slots: 4 A rows and BROWS B rows, one float4 (c4) each ---
  const int c4 = tid & 7, r0 = tid >> 3;
  int ax[4], ay[4], az[4];
  bool arow_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    arow_ok[i] = m < p.M;
    const int mm = arow_ok[i] ? m : 0;
    az[i] = mm % p.gz;
    ay[i] = (mm / p.gz) % p.gy;
    ax[i] = mm / (p.gz * p.gy);
  }

  float4 ra[4], rb[BROWS];
  auto load_step = [&](int s) {
    const int tap = p.transposed ? parity : tap_lo + s / ksteps_c;
    const int ci0 = (s % ksteps_c) * BK + c4 * 4;
    int dx = 0, dy = 0, dz = 0;
    if (!p.transposed && p.ksize > 1) {
      dx = tap / (p.ksize * p.ksize); dy = (tap / p.ksize) % p.ksize; dz = tap % p.ksize;
      if (p.two_d) { dx = p.pad; dy = tap / p.ksize; dz = tap % p.ksize; }      // k x k taps in the (y, z) plane of every x slice
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int xx = ax[i] * p.stride + dx - p.pad, yy = ay[i] * p.stride + dy - p.pad,
                zz = az[i] * p.stride + dz - p.pad;
      const bool ok = arow_ok[i] && xx >= 0 && xx < p.ix && yy >= 0 && yy < p.iy && zz >= 0 && zz < p.iz;
      ra[i] = ok ? *reinterpret_cast<const float4 *>(p.x + ((int64_t)(xx * p.iy + yy) * p.iz + zz) * p.Cin + ci0)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const int n = n0 + r0 + 32 * i;
      rb[i] = n < p.Cout ? *reinterpret_cast<const float4 *>(p.w + ((int64_t)tap * p.Cout + n) * p.Cin + ci0)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_step = [&](int buf) {
    float *a = As + buf * BM * LDK, *b = Bs + buf * BN * LDK;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(a + (r0 + 32 * i) * LDK + c4 * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *reinterpret_cast<float4 *>(b + (r0 + 32 * i) * LDK + c4 * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  load_step(0);
  store_step(0);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) load_step(s + 1);
    const float *a = As + buf * BM * LDK + (wm * (BM / WM) + fr) * LDK + fh * 4;
    const float *b = Bs + buf * BN * LDK + (wn * (BN / WN) + fr) * LDK + fh * 4;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(a + i * 32 * LDK + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(b + j * 32 * LDK + kk * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (s + 1 < nsteps) {
      store_step(buf ^ 1);   // the other buffer was last read in step s-1, before the barrier below
    }
    __syncthreads();
  }

  // --- epilogue: C/D layout col = lane&31, row = (j&3) + 8*(j>>2) + 4*(lane>>5) ---
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
      if (col >= p.Cout) continue;
      const float sc = p.scale ? p.scale[col] : 1.f, sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int m = m0 + wm * (BM / WM) + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        int64_t orow = m;
        if (p.transposed) {
          const int z = m % p.gz, y = (m / p.gz) % p.gy, x = m / (p.gz * p.gy);
          const int px = parity >> 2, py = (parity >> 1) & 1, pz = parity & 1;
          orow = ((int64_t)(2 * x + px) * (2 * p.gy) + (2 * y + py)) * (2 * p.gz) + (2 * z + pz);
        }
        float *dst = p.y + orow * p.Cout + col;
        if (p.splitk > 1) {
          if (p.ws) p.ws[(int64_t)zid * p.ws_stride + orow * p.Cout + col] = acc[i][j][k];
          else atomicAdd(dst, acc[i][j][k]);
        } else {
          float v = acc[i][j][k] * sc + sh;
          if (p.relu == 2) v = fmaxf(v, 0.f);
          if (p.residual) v += p.residual[orow * p.Cout + col];
          if (p.relu == 1) v = fmaxf(v, 0.f);
          if (p.act_scale) v = act_col(v, col, p.act_c0, p.act_c1, *p.act_scale);
          *dst = v;
        }
      }
    }
}


// ---------------------------------------------------------------------------------------------
// bf16x3 variant: fp32 operands split as a = a_hi + a_lo (two bf16 each), products
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  bf16 x bf16
// products are exact in fp32, the dropped a_lo*b_lo term is 2^-16 relative: the result agrees with
// the fp32 kernel to ~1e-5 (tests bound it at 1e-4 of the tensor scale, north-star bar 1e-3) at
// 3/16 of the fp32-MFMA cycles.  Activations stay fp32 in HBM and are split while they are staged
// into LDS; weights are split once on the host.
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDKH = BK + 8;   // bf16 elements per LDS row (80 B): conflict-free ds_read_b128

struct ConvParamsB : ConvParams {
  const __bf16 *w_hi, *w_lo;   // [taps][Cout][Cin]
};

// BMT: rows of the workgroup tile (128, or 256 = the tall tile of the split / transposed layers: wave tile 64 x 64, two thirds of the
// LDS fragment reads per MFMA of the 32 x 64 wave tile).  NP: bf16 products per multiply-add (3 = fp32-faithful split, 1 = hi * hi only)
template <int BN, int WM, int WN, int NP = 3, int BMT = 128>
__global__ __launch_bounds__(WM * WN * 64) void conv3d_igemm_bf16x3_kernel(const ConvParamsB p) {
  constexpr int NT = WM * WN * 64;                     // threads per workgroup (256 or 512)
  constexpr int TM = BMT / WM / 32, TN = BN / WN / 32;
  constexpr int ACH = BMT * 8 / NT;                     // float4 A chunks per thread (rows r0 + (NT/8) i)
  constexpr int AROWS = NT / 8;
  constexpr int BCH = BN * 4 / NT;                     // 16-byte weight chunks per thread per plane
  constexpr int BROWS_ = NT / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  // per buffer: A_hi[BMT][LDKH], A_lo[BMT][LDKH], B_hi[BN][LDKH], B_lo[BN][LDKH]
  constexpr int A_PLANE = BMT * LDKH, B_PLANE = BN * LDKH, BUF = 2 * A_PLANE + 2 * B_PLANE;
  __bf16 *base = reinterpret_cast<__bf16 *>(smem_b);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_deal) {
    // the hardware deals workgroup L (x fastest) to XCD L % 8: XCD c holds L = c, c + 8, ...  Renumber so that the tiles an
    // XCD works on are CONSECUTIVE in the chosen order -- neighbours then find their shared operand in that XCD's L2
    const int total = gridDim.x * gridDim.y * gridDim.z;
    const int L = bx + gridDim.x * (by + gridDim.y * bz);
    const int q = total >> 3, r = total & 7, c = L & 7;
    int t = c * q + min(c, r) + (L >> 3);
    if (p.xcd_deal == 1) { bx = t % gridDim.x; t /= gridDim.x; by = t % gridDim.y; bz = t / gridDim.y; }
    else                 { by = t % gridDim.y; t /= gridDim.y; bx = t % gridDim.x; bz = t / gridDim.x; }
  }
  const int m0 = bx * BMT, n0 = by * BN;
  if (p.zero_row && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int c = threadIdx.x; c < p.Cout; c += NT) p.zero_row[c] = 0.f;
  const int Mrows = p.m_dev ? min(p.M, *p.m_dev) : p.M;
  if (m0 >= Mrows) return;
  int zid = bz;
  int parity = 0;
  if (p.transposed) { parity = zid % 8; zid /= 8; }
  // Split z of a split launch owns the K steps [zid * steps_per, ...) of the (tap, channel chunk) sequence: a split boundary may fall
  // inside a tap, so any number of splits balances a launch (round 6; groups of whole taps only allowed 3 / 9 / 27)
  const int ksteps_c = p.Cin / BK;
  const int total_steps = (p.transposed ? 1 : p.taps) * ksteps_c;
  const int step_lo = p.splitk > 1 ? zid * p.steps_per : 0;
  const int nsteps = p.splitk > 1 ? min(p.steps_per, total_steps - step_lo) : total_steps;
  if (nsteps <= 0) return;                              // (the host never launches an empty split)

  // Staging rows are dealt so that the lanes one LDS write pass covers (32 lanes x 8 B for A, 16 lanes x 16 B for B)
  // sit in rows {r, r+4, r+8, r+12}: with the 20-dword row pitch those start 16 banks apart and tile all 64 banks;
  // consecutive rows (the plain tid >> 3 deal) overlap by 12 banks and every pass took two turns.
  const int c4 = tid & 7, rs8 = (tid >> 3) & 7;
  const int r0 = 16 * (wid >> 1) + 2 * (wid & 1) + (rs8 >> 2) + 4 * (rs8 & 3);   // A: row r0 + AROWS i, 4 floats at c4*4
  const int bc = tid & 3, rs16 = (tid >> 2) & 15;
  const int br0 = 16 * wid + (rs16 >> 2) + 4 * (rs16 & 3);                        // B: row br0 + BROWS_ i, 8 bf16 at bc*8
  // Addressing is split by how often it changes.  Per TAP: the input row of each of this thread's A chunks (neighbour lookup,
  // padding test) -> a 32-bit byte offset, 0xfffffff0 for "no such row".  Per STEP: one uniform offset (the channel chunk, and
  // for the weights the tap's slab).  Loads go through buffer descriptors: an offset past the tensor returns zeros, so the loop
  // has no branch and no per-step index arithmetic in the vector unit (it used to spend 280 instructions per step, 100 of them
  // scalar divisions of the tap decode, on 12 MFMAs per wave).
  constexpr unsigned OOB = 0xfffffff0u;
  int ax[ACH], ay[ACH], az[ACH];
  bool arow_ok[ACH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int m = m0 + r0 + AROWS * i;
    arow_ok[i] = m < Mrows;
    const int mm = arow_ok[i] ? m : 0;
    az[i] = mm % p.gz;
    ay[i] = (mm / p.gz) % p.gy;
    ax[i] = mm / (p.gz * p.gy);
  }
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.x), 0, (int)(unsigned)((int64_t)p.ix * p.iy * p.iz * p.Cin * 4), 0x00020000);
  const int w_bytes = (int)(unsigned)((int64_t)(p.transposed ? 8 : p.taps) * p.Cout * p.Cin * 2);
  const __amdgpu_buffer_rsrc_t whr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.w_hi), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wlr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.w_lo), 0, w_bytes, 0x00020000);
  unsigned boff[BCH];                                  // this thread's weight rows: fixed for the whole kernel
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int n = n0 + br0 + BROWS_ * i;
    boff[i] = n < p.Cout ? (unsigned)(n * p.Cin + bc * 8) * 2u : OOB;
  }
  unsigned aoff[ACH];                                  // this thread's input rows under the tap being loaded
  auto set_tap = [&](int tap) {
    int dx = 0, dy = 0, dz = 0;
    if (!p.transposed && p.ksize > 1) {
      dx = tap / (p.ksize * p.ksize); dy = (tap / p.ksize) % p.ksize; dz = tap % p.ksize;
      if (p.two_d) { dx = p.pad; dy = tap / p.ksize; dz = tap % p.ksize; }      // k x k taps in the (y, z) plane of every x slice
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int xx = ax[i] * p.stride + dx - p.pad, yy = ay[i] * p.stride + dy - p.pad,
                zz = az[i] * p.stride + dz - p.pad;
      const bool ok = arow_ok[i] && xx >= 0 && xx < p.ix && yy >= 0 && yy < p.iy && zz >= 0 && zz < p.iz;
      aoff[i] = ok ? ((unsigned)((xx * p.iy + yy) * p.iz + zz) * (unsigned)p.Cin + c4 * 4) * 4u : OOB;
    }
  };
  // (a second register stage -- loads of step s + 2 issued before the MFMAs of step s -- was tried: 156 VGPRs and
  //  one workgroup per CU, or 128 with spills; 404 -> 507 us on the per-tap 90 GF layer, 143 -> 180-200 us on the
  //  split-K layers.  Two resident workgroups at 88 VGPRs hide more latency than the deeper prefetch.)
  float4 ra[ACH];
  uint4 rbh[BCH], rbl[BCH];
  if constexpr ((SGC_TILE_SKIP & 6) != 0) {               // timing builds: the registers the skipped loads would have filled
#pragma unroll
    for (int i = 0; i < ACH; ++i) ra[i] = make_float4(1.f + tid, 2.f, 3.f, 4.f);
#pragma unroll
    for (int i = 0; i < BCH; ++i) { rbh[i] = make_uint4(tid, 1, 2, 3); rbl[i] = make_uint4(3, 2, 1, tid); }
  }
  int ld_tap = p.transposed ? parity : step_lo / ksteps_c;       // (tap, channel chunk) of the NEXT load_step
  int ld_kc = p.transposed ? step_lo : step_lo % ksteps_c;
  set_tap(ld_tap);
  auto load_step = [&]() {
    const int soff_a = __builtin_amdgcn_readfirstlane(ld_kc * (BK * 4));
    const int soff_b = __builtin_amdgcn_readfirstlane((ld_tap * p.Cout * p.Cin + ld_kc * BK) * 2);
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      if constexpr ((SGC_TILE_SKIP & 2) != 0) break;      // timing builds (diag.hpp): no input loads
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, aoff[i], soff_a, 0);
      ra[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      if constexpr ((SGC_TILE_SKIP & 4) != 0) break;      // timing builds: no weight loads
      const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(whr, boff[i], soff_b, 0);
      rbh[i] = make_uint4(h[0], h[1], h[2], h[3]);
      if constexpr (NP == 3) {
        const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(wlr, boff[i], soff_b, 0);
        rbl[i] = make_uint4(l[0], l[1], l[2], l[3]);
      } else {
        rbl[i] = make_uint4(0, 0, 0, 0);
      }
    }
    if (++ld_kc == ksteps_c) {                          // next tap: uniform branch, once per Cin / 32 steps
      ld_kc = 0;
      ++ld_tap;
      if (!p.transposed && ld_tap < p.taps) set_tap(ld_tap);
    }
  };
  auto store_step = [&](int buf) {
    if constexpr ((SGC_TILE_SKIP & 8) != 0) { if (p.relu != 77) return; }    // timing builds: no split, no LDS stores
    __bf16 *a_hi = base + buf * BUF, *a_lo = a_hi + A_PLANE, *b_hi = a_lo + A_PLANE, *b_lo = b_hi + B_PLANE;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = op_hi<NP>(v[e]);
        h[e] = hb;
        l[e] = op_lo<NP>(v[e], hb);
      }
      const int o = (r0 + AROWS * i) * LDKH + c4 * 4;
      *reinterpret_cast<bf16x4 *>(a_hi + o) = h;
      if constexpr (NP == 3) *reinterpret_cast<bf16x4 *>(a_lo + o) = l;
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int o = (br0 + BROWS_ * i) * LDKH + bc * 8;
      *reinterpret_cast<uint4 *>(b_hi + o) = rbh[i];
      if constexpr (NP == 3) *reinterpret_cast<uint4 *>(b_lo + o) = rbl[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  load_step();
  store_step(0);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) load_step();
    const __bf16 *a_hi = base + buf * BUF + (wm * (BMT / WM) + fr) * LDKH + fh * 8;
    const __bf16 *a_lo = a_hi + A_PLANE;
    const __bf16 *b_hi = base + buf * BUF + 2 * A_PLANE + (wn * (BN / WN) + fr) * LDKH + fh * 8;
    const __bf16 *b_lo = b_hi + B_PLANE;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
      if constexpr ((SGC_TILE_SKIP & 16) != 0) {           // timing builds: no fragment reads
#pragma unroll
        for (int i = 0; i < TM; ++i) { ah[i] = (bf16x8)(__bf16)(float)(lane + kk); al[i] = (bf16x8)(__bf16)(float)(lane + 2 * kk); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bh[j] = (bf16x8)(__bf16)(float)(wid + kk); bl[j] = (bf16x8)(__bf16)(float)(wid + 3 * kk); }
      } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8 *>(a_hi + i * 32 * LDKH + kk * 16);
        if constexpr (NP == 3) al[i] = *reinterpret_cast<const bf16x8 *>(a_lo + i * 32 * LDKH + kk * 16);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8 *>(b_hi + j * 32 * LDKH + kk * 16);
        if constexpr (NP == 3) bl[j] = *reinterpret_cast<const bf16x8 *>(b_lo + j * 32 * LDKH + kk * 16);
      }
      }
      if constexpr ((SGC_TILE_SKIP & 1) != 0) {            // timing builds: everything but the MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (NP == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = mma_hh<NP>(ah[i], bh[j], acc[i][j]);
        }
    }
    if (s + 1 < nsteps) store_step(buf ^ 1);
    if constexpr ((SGC_TILE_SKIP & 32) == 0) __syncthreads();     // timing builds: no barrier per step
  }
  if constexpr ((SGC_TILE_SKIP & 64) != 0) { if (p.relu != 77) return; }      // timing builds: no epilogue

  // Epilogue through LDS: in the MFMA layout a lane owns ONE column and 16 rows of a tile, i.e. 4-byte stores, 32 per
  // lane -- store-issue bound (PMC on the K = 256 Linears: waves parked 54 % of their cycles, matrix pipe busy 20 %).
  // The staging buffers are free now: the tile goes to LDS once and leaves as 16-byte row-contiguous stores (and the
  // partial tiles of a split reduction, the residual and the scale / shift vectors move 16 bytes at a time too).
  if ((p.Cout & 3) == 0 && (p.splitk == 1 || p.ws)) {
    constexpr int LDC = BN + 8;                              // floats per staged row: rows r and r + 4 (lane halves) 32 banks apart
    float *cs = reinterpret_cast<float *>(smem_b);           // [BMT][LDC] <= the 2 x (A + B) staging buffers
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k)
          cs[(wm * (BMT / WM) + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)) * LDC + wn * (BN / WN) + j * 32 + (lane & 31)] =
              acc[i][j][k];
    __syncthreads();
    constexpr int C4 = BN / 4;
    if (p.hm_cm > 0) {
      // head-major store: the lanes of a wave instruction walk ROWS of one head (a head's rows are hm_cm * 4 bytes apart in
      // its plane), so a wave writes one contiguous 1 KiB run instead of 8 head segments 1 plane apart
      const int cvh = p.hm_cm / 4, per_head = BMT * cvh, heads = p.Cout / p.hm_cm;
      const int ncam0 = m0 / p.hm_S, s0 = m0 - ncam0 * p.hm_S;
      for (int e = tid; e < BMT * C4; e += NT) {
        const int hl = e / per_head, rr = e - hl * per_head;
        const int rl = rr / cvh, c4 = hl * cvh + (rr - rl * cvh);
        const int m = m0 + rl, col = n0 + c4 * 4;
        if (m >= Mrows || col >= p.Cout) continue;
        float4 v = *reinterpret_cast<const float4 *>(cs + rl * LDC + c4 * 4);
        if (p.shift) {
          const float4 sh4 = *reinterpret_cast<const float4 *>(p.shift + col);
          v.x += sh4.x; v.y += sh4.y; v.z += sh4.z; v.w += sh4.w;
        }
        int ncam = ncam0, spx = s0 + rl;             // rows of one tile straddle at most a few cameras: no division per element
        while (spx >= p.hm_S) { spx -= p.hm_S; ++ncam; }
        const int head = (n0 / p.hm_cm) + hl;
        const int64_t o = (((int64_t)ncam * heads + head) * p.hm_S + spx) * p.hm_cm + (col - head * p.hm_cm);
        if (p.hm_bf16) {
          bf16x4 h;
          h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
          *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(p.y) + o) = h;
        } else {
          *reinterpret_cast<float4 *>(p.y + o) = v;
        }
      }
      return;
    }
    for (int e = tid; e < BMT * C4; e += NT) {
      const int rl = e / C4, c4 = e - rl * C4;
      const int m = m0 + rl, col = n0 + c4 * 4;
      if (m >= Mrows || col >= p.Cout) continue;
      int64_t orow = m;
      if (p.transposed) {
        const int z = m % p.gz, y = (m / p.gz) % p.gy, x = m / (p.gz * p.gy);
        const int px = parity >> 2, py = (parity >> 1) & 1, pz = parity & 1;
        orow = ((int64_t)(2 * x + px) * (2 * p.gy) + (2 * y + py)) * (2 * p.gz) + (2 * z + pz);
      }
      float4 v = *reinterpret_cast<const float4 *>(cs + rl * LDC + c4 * 4);
      if (p.splitk > 1) {
        *reinterpret_cast<float4 *>(p.ws + (int64_t)zid * p.ws_stride + orow * p.Cout + col) = v;
        continue;
      }
      if (p.scale) {
        const float4 sc4 = *reinterpret_cast<const float4 *>(p.scale + col);
        v.x *= sc4.x; v.y *= sc4.y; v.z *= sc4.z; v.w *= sc4.w;
      }
      if (p.shift) {
        const float4 sh4 = *reinterpret_cast<const float4 *>(p.shift + col);
        v.x += sh4.x; v.y += sh4.y; v.z += sh4.z; v.w += sh4.w;
      }
      if (p.relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (p.residual) {
        const float4 r4 = *reinterpret_cast<const float4 *>(p.residual + orow * p.Cout + col);
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
      }
      if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (p.act_scale) v = act_col4(v, col, p.act_c0, p.act_c1, *p.act_scale);
      *reinterpret_cast<float4 *>(p.y + orow * p.Cout + col) = v;
    }
    return;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
      if (col >= p.Cout) continue;
      const float sc = p.scale ? p.scale[col] : 1.f, sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int m = m0 + wm * (BMT / WM) + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
        if (m >= Mrows) continue;
        int64_t orow = m;
        if (p.transposed) {
          const int z = m % p.gz, y = (m / p.gz) % p.gy, x = m / (p.gz * p.gy);
          const int px = parity >> 2, py = (parity >> 1) & 1, pz = parity & 1;
          orow = ((int64_t)(2 * x + px) * (2 * p.gy) + (2 * y + py)) * (2 * p.gz) + (2 * z + pz);
        }
        float *dst = p.y + orow * p.Cout + col;
        if (p.splitk > 1) {
          if (p.ws) p.ws[(int64_t)zid * p.ws_stride + orow * p.Cout + col] = acc[i][j][k];
          else atomicAdd(dst, acc[i][j][k]);
        } else {
          float v = acc[i][j][k] * sc + sh;
          if (p.relu == 2) v = fmaxf(v, 0.f);
          if (p.residual) v += p.residual[orow * p.Cout + col];
          if (p.relu == 1) v = fmaxf(v, 0.f);
          if (p.act_scale) v = act_col(v, col, p.act_c0, p.act_c1, *p.act_scale);
          *dst = v;
        }
      }
    }
}


// ---------------------------------------------------------------------------------------------
// v2 for the 3x3x3 stride-1 layers (90 % of the neck's FLOPs): halo-resident A.
// A workgroup owns a brick of BX*BY*BZ = 256 output voxels.  For one 32-channel slice it stages the
// brick's input HALO ((BX+2)(BY+2)(BZ+2) rows, split to bf16 hi/lo once) in LDS and then walks the 27
// taps with the SAME staged rows: the A fragment of output row r at tap t is the halo row
// hr(r) + toff(t), a wave-uniform offset on a per-lane base.  Only the weights stream (16 KB hi+lo per
// tap, double buffered).  L2->LDS traffic per MAC drops 3.4x against the per-tap gather above, and the
// fp32->bf16 split runs once per halo element instead of 27 times.
// 512 threads = 8 waves as 4 (M) x 2 (N), wave tile 64 x 64, BN = 128 output channels.
// LDS: A 2 planes x HROWS x 80 B (<= 104 KB) + B 2 buffers x 2 planes x 128 x 80 B (41 KB).
// ---------------------------------------------------------------------------------------------
// z-pitch (in rows) of the halo image in LDS: the smallest pitch >= BZ + 2 for which every 32-row MFMA tile
// of the brick holds each halo-row residue mod 16 exactly twice (checked offline for the three brick shapes:
// 18 for BZ = 16, 12 for BZ = 8, 6 for BZ = 4) -- the precondition of the conflict-free lane assignment.
__host__ __device__ constexpr int halo_pitch(int BZ) { return BZ == 8 ? 12 : BZ + 2; }
__host__ __device__ constexpr size_t halo_tab_offset(int lrows, int mrows = 256, int bnv = 128) {
  const size_t planes = (size_t)(2 * lrows + 2 * 2 * bnv) * LDKH * sizeof(uint16_t);   // A hi|lo + 2 x B hi|lo
  const size_t stage = (size_t)mrows * (bnv + 8) * sizeof(float);                      // epilogue tile [MROWS][BNV + 8]
  return planes > stage ? planes : stage;
}

// BNV: output columns per workgroup, 128 (wave tile 64 x 64) or 64 (wave tile 64 x 32: the head's 28 / 32-column layers, which
// otherwise spend three quarters of their matrix work on padding columns).
// TD: 2-D form (sgc_conv2d_nhwc_bf16x3: the FPN's 3 x 3 output convolutions) -- the grid is (image, row, column),
// a brick is BX images x BY x BZ pixels, there is no halo and no tap along x: 9 taps, (BY + 2)(BZ + 2) halo rows per image.
// STG: software-pipelined schedule with the barrier in the MIDDLE of a tap (round 4; the body explains the hazards).  The
// lockstep form (STG = false) put the barrier at the end of a tap: behind it every wave first had to fetch the freshly published
// weight fragments from LDS (MFMA pipe idle for an LDS round trip with 96 reads queued), and in front of it every wave waited
// for its two weight ds_write_b128 to drain.  Timing builds (tools/halo_skip.py, 90-GF layer, warm): 232 us as shipped, 195
// without the weight ds_writes, 215 without the barrier, 183 without the weight loads and writes, 179 with nothing but the MFMAs
// and the loop -- the weight path cost a fifth of the kernel although it moves 16 KB per tap.  With the barrier at mid-tap the
// operands of BOTH k-halves are in registers before the MFMAs that use them are reached, the weight tile is written a
// half-tap before the barrier that publishes it, and nothing but wave skew is left at the barrier.  Every accumulator still
// sees (tap, k-half, product) in the same order: bit-identical to the lockstep form.
// WZ (2-D form only): the image stack is VIRTUAL -- image k * J + j, pixel (xx, yy) is the Winograd F(2,3)-along-z input transform
// t_k of the raw volume p.x [rows][cols][Z = 2 J][Cin] at the output pair j (sgc_conv3d_winograd_z_bf16x3): every staged chunk is
// loaded from TWO voxel rows and combined (a - b, or a + b for k = 1) in front of the hi / lo split; no transformed copy exists.
template <int BX, int BY, int BZ, int BNV = 128, int NP = 3, bool TD = false, bool STG = true, bool WZ = false>
__global__ __launch_bounds__(512) void conv3d_halo_bf16x3_kernel(const ConvParamsB p) {
  static_assert(!WZ || TD, "the virtual Winograd stack is a 2-D form");
  constexpr int NTAP = TD ? 9 : 27, XO = TD ? 0 : 1;    // taps; halo width along x
  // MFMA rows of the brick: 256 for the standard bricks; a brick with another voxel count (a whole small grid: 10 x 10 x 4, the
  // coarsest config-2 scale) is padded to a multiple of 128 rows (4 wave rows x 32) -- pad rows work on voxel 0 and are dropped
  // wave layout: WMV (rows) x WNV (columns) = 8 waves.  4 x 2 for 128- and 64-column tiles; 8 x 1 for the 32-column tile of the
  // head's fused 28-column convolution (round 5: on 64 columns more than half of its matrix work was padding)
  constexpr int WNV = BNV >= 64 ? 2 : 1, WMV = 8 / WNV;
  constexpr int NVOX = BX * BY * BZ, MROWS = (NVOX + 32 * WMV - 1) / (32 * WMV) * (32 * WMV);
  constexpr int RT = MROWS / (32 * WMV);                // 32-row tiles per wave
  constexpr unsigned short kPadRow = 0x8000;            // vox_tab flag of a pad row
  static_assert(MROWS <= 512, "one table entry per thread");
  constexpr int TN = BNV / (32 * WNV), WCOL = BNV / WNV; // 32-column tiles per wave, columns per wave
  constexpr int HX = BX + 2 * XO, HY = BY + 2, HZ = BZ + 2, HROWS = HX * HY * HZ;
  constexpr int HZP = halo_pitch(BZ), LROWS = HX * HY * HZP;   // z-pitch of the LDS image (see halo_pitch)
  constexpr int NT = 512;
  constexpr int NA = (HROWS * 8 + NT - 1) / NT;     // float4 halo chunks per thread
  constexpr int A_PLANE = LROWS * LDKH, B_PLANE = BNV * LDKH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  __bf16 *A_hi = reinterpret_cast<__bf16 *>(smem_h), *A_lo = A_hi + A_PLANE;
  __bf16 *Bbase = A_lo + A_PLANE;                   // [2][hi|lo][BNV][LDKH]
  // [8 tiles][32 lanes], behind both the staging planes and the epilogue's output tile that later overlays them
  unsigned short *vox_tab = reinterpret_cast<unsigned short *>(smem_h + halo_tab_offset(LROWS, MROWS, MROWS > 256 ? BNV : 128));

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WNV, wn = wid % WNV;
  const int nby = (p.gy + BY - 1) / BY, nbz = (p.gz + BZ - 1) / BZ;
  int bid = blockIdx.x;
  const int bk = bid % nbz; bid /= nbz;
  const int bj = bid % nby; const int bi = bid / nby;
  const int X0 = bi * BX, Y0 = bj * BY, Z0 = bk * BZ;
  const int n0 = blockIdx.y * BNV;
  const int nchunks = p.Cin / BK;
  const int per = (nchunks + p.splitk - 1) / p.splitk;
  const int c_lo = blockIdx.z * per, c_hi = min(nchunks, c_lo + per);
  if (c_lo >= c_hi) return;

  // Which output voxel of the brick each MFMA row (= lane & 31 of a 32-row tile) works on.  ds_read_b128
  // serves a wave in four fixed 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) over 64 banks,
  // i.e. with the 80-byte row stride a group is conflict-free iff its halo rows are distinct mod 16.  The
  // natural order (lane = z-run position) is not: a tile spans several z-runs whose halo rows are HZP apart
  // (measured: 38 % of the LDS cycles of this kernel were bank-conflict cycles).  Every tile holds each
  // residue exactly twice (halo_pitch guarantees it), so lane l takes the first (l < 16) or second voxel of
  // the tile whose halo row is == l mod 16 -- any assignment works as long as the epilogue uses the same one.
  // (Other brick shapes -- a whole small grid -- do not keep that precondition for every tile: the greedy pass below IS the
  //  assignment above wherever it exists and degrades to a few two-way conflicts elsewhere, never to a wrong permutation.)
  // One wave per tile, lane j = row j of the tile (both wave halves compute the same; the upper half does not store): a row whose
  // residue it is the first / second to carry takes slot residue / 16 + residue; further rows of a crowded residue fill the
  // slots of the residues that came short, in order.
  for (int t = wid; t < MROWS / 32; t += NT / 64) {
    const int j = lane & 31;
    const int r = t * 32 + j, rv = r < NVOX ? r : 0;
    const int x = rv / (BY * BZ), y = (rv / BZ) % BY, z = rv % BZ;
    const int res = (((x + XO) * HY + (y + 1)) * HZP + (z + 1)) & 15;
    unsigned same = 0, mine = 0;                      // rows with this row's residue / with this SLOT's residue (slot j: j & 15)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const unsigned m = (unsigned)__ballot(res == q);
      if (res == q) same = m;
      if ((j & 15) == q) mine = m;
    }
    const unsigned below = (1u << j) - 1u;
    const int rank = __popc(same & below);
    const bool slot_empty = __popc(mine) < (j >> 4) + 1;
    const unsigned left = (unsigned)__ballot(rank >= 2), empty = (unsigned)__ballot(slot_empty);
    const unsigned short val = r < NVOX ? (unsigned short)r : kPadRow;
    unsigned short *tab = vox_tab + t * 32, *tmp = vox_tab + MROWS + (wid & 7) * 32;
    if (lane < 32) {
      if (rank < 2) tab[rank * 16 + res] = val;
      else tmp[__popc(left & below)] = val;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // one wave, LDS operations complete in order: its own stores are visible
    if (lane < 32 && slot_empty) tab[j] = tmp[__popc(empty & below)];
  }
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  int arow[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int r = vox_tab[(wm * RT + i) * 32 + fr] & 0x7fff;          // pad rows read voxel 0's halo rows (valid LDS, result dropped)
    const int x = r / (BY * BZ), y = (r / BZ) % BY, z = r % BZ;
    arow[i] = ((x + XO) * HY + (y + 1)) * HZP + (z + 1);
  }
  // output mask: does this wave's 64-voxel tile / this brick hold a row the caller needs?
  bool wave_live = true;
  if (p.out_mask) {
    bool mine = false;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int r = vox_tab[(wm * RT + i) * 32 + fr];
      const int x = X0 + r / (BY * BZ), y = Y0 + (r / BZ) % BY, z = Z0 + r % BZ;
      if (!(r & kPadRow) && x < p.gx && y < p.gy && z < p.gz) mine |= p.out_mask[((int64_t)x * p.gy + y) * p.gz + z] != 0;
    }
    wave_live = __ballot(mine) != 0ull;
    if (!__syncthreads_or(wave_live ? 1 : 0)) {
      // dead brick: store the epilogue of a zero accumulator and leave (no staging, no taps)
      if (p.splitk > 1 && !p.ws) return;             // atomics path: y was zero-filled, the epilogue kernel finishes it
      for (int e = tid; e < NVOX * (BNV / 4); e += NT) {
        const int rl = e / (BNV / 4), c4 = e - rl * (BNV / 4);
        const int col = n0 + c4 * 4;
        if (col >= p.Cout) continue;
        const int x = X0 + rl / (BY * BZ), y = Y0 + (rl / BZ) % BY, z = Z0 + rl % BZ;
        if (x >= p.gx || y >= p.gy || z >= p.gz) continue;
        const int64_t orow = ((int64_t)x * p.gy + y) * p.gz + z;
        for (int q = 0; q < 4 && col + q < p.Cout; ++q) {
          float v = 0.f;
          if (p.splitk > 1) { p.ws[(int64_t)blockIdx.z * p.ws_stride + orow * p.Cout + col + q] = 0.f; continue; }
          v = v * (p.scale ? p.scale[col + q] : 1.f) + (p.shift ? p.shift[col + q] : 0.f);
          if (p.relu == 2) v = fmaxf(v, 0.f);
          if (p.residual) v += p.residual[orow * p.Cout + col + q];
          if (p.relu == 1) v = fmaxf(v, 0.f);
          if (p.act_scale) v = act_col(v, col + q, p.act_c0, p.act_c1, *p.act_scale);
          p.y[orow * p.Cout + col + q] = v;
        }
      }
      return;
    }
  }
  // B staging slot of this thread: 8 bf16 at (tid&3)*8 of row bn; the 16 lanes of one ds_write_b128 pass take rows
  // {r, r+4, r+8, r+12} (16 banks apart at the 20-dword pitch) instead of 4 consecutive rows that overlap by 12 banks
  const int bc = tid & 3, rs16 = (tid >> 2) & 15;
  const int bn = 16 * wid + (rs16 >> 2) + 4 * (rs16 & 3);
  const bool bn_ok = tid < BNV * 4 && n0 + bn < p.Cout;     // BNV * 4 sixteen-byte chunks per plane and tap

  f32x16 acc[RT][TN];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  SGC_HALO_STAMP(0);

  float4 ra[NA], rw[WZ ? NA : 1];      // rw: the second voxel row of a Winograd transform chunk
  uint4 rbh, rbl;
  // Addressing is fixed per thread for the whole kernel (the halo rows a thread stages and its weight row do not depend on the
  // channel slice or the tap): one 32-bit byte offset per chunk, 0xfffffff0 = "outside the volume / padding slot", computed once;
  // a slice / a tap then only moves a uniform offset.  Buffer loads return zeros past the tensor, so the loads carry no branch.
  constexpr unsigned OOB = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.x), 0, (int)(unsigned)((int64_t)(WZ ? p.wz_Z : p.ix) * p.iy * p.iz * p.Cin * 4), 0x00020000);
  const int w_bytes = (int)(unsigned)((int64_t)NTAP * p.Cout * p.Cin * 2);
  // weight set of this brick: one for the whole launch, or -- 2-D form with image groups -- that of the group its images belong to
  const int64_t w_set = (TD && p.w_group_images > 0) ? (int64_t)(X0 / p.w_group_images) * NTAP * p.Cout * p.Cin : 0;
  const __amdgpu_buffer_rsrc_t whr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.w_hi + w_set), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wlr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.w_lo + w_set), 0, w_bytes, 0x00020000);
  unsigned aoff[NA], aoffb[WZ ? NA : 1];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = i * NT + tid;
    const int row = idx >> 3, c4 = idx & 7;
    const int hz = row % HZ, hy = (row / HZ) % HY, hx = row / (HZ * HY);
    const int gx = X0 + hx - XO, gy = Y0 + hy - 1, gz = Z0 + hz - 1;
    const bool in = row < HROWS && gx >= 0 && gx < p.ix && gy >= 0 && gy < p.iy && gz >= 0 && gz < p.iz;
    if constexpr (WZ) {
      // image gx = kpos * J + j: the two voxel rows of the raw volume [iy][iz][Z] whose combination is this transform row
      const int J = p.wz_Z >> 1, kpos = X0 / J, j = gx - kpos * J;       // a brick's images belong to one position (J % BX == 0)
      const int za = kpos == 0 ? 2 * j - 1 : kpos == 2 ? 2 * j + 1 : 2 * j;
      const int zb = kpos <= 1 ? 2 * j + 1 : kpos == 2 ? 2 * j : 2 * j + 2;
      const unsigned col = (unsigned)((gy * p.iz + gz) * p.wz_Z);
      aoff[i] = in && za >= 0 ? ((col + (unsigned)za) * (unsigned)p.Cin + c4 * 4) * 4u : OOB;
      aoffb[i] = in && zb < p.wz_Z ? ((col + (unsigned)zb) * (unsigned)p.Cin + c4 * 4) * 4u : OOB;
    } else {
      aoff[i] = in ? ((unsigned)((gx * p.iy + gy) * p.iz + gz) * (unsigned)p.Cin + c4 * 4) * 4u : OOB;
    }
  }
  const float wz_sign = WZ && (X0 / max(p.wz_Z >> 1, 1)) == 1 ? 1.f : -1.f;      // t1 = d1 + d2; t0, t2, t3 are differences
  const unsigned boff = bn_ok ? (unsigned)((n0 + bn) * p.Cin + bc * 8) * 2u : OOB;
  auto load_A = [&](int cc) {
    const int soff = __builtin_amdgcn_readfirstlane(cc * (BK * 4));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, aoff[i], soff, 0);
      ra[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
      if constexpr (WZ) {
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(xr, aoffb[i], soff, 0);
        rw[i] = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
      }
    }
  };
  // the split of a loaded halo chunk, in place: ra[i] = (hi.xy, hi.zw, lo.xy, lo.zw) as packed bf16 pairs.  Called under the
  // last tap of a slice (the loads went out three taps earlier), so that between the slice's last barrier and the next
  // slice's first tap only the ds_writes remain -- the vector work of the split overlaps the other wave's MFMAs instead of
  // sitting between two barriers.
  auto split_A = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      if constexpr (WZ) {                       // the input transform: one fp32 rounding per element (sign * b is exact)
        v[0] += wz_sign * rw[i].x; v[1] += wz_sign * rw[i].y; v[2] += wz_sign * rw[i].z; v[3] += wz_sign * rw[i].w;
      }
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = op_hi<NP>(v[e]);
        h[e] = hb;
        l[e] = op_lo<NP>(v[e], hb);
      }
      const uint2 hu = __builtin_bit_cast(uint2, h), lu = __builtin_bit_cast(uint2, l);
      ra[i] = make_float4(__uint_as_float(hu.x), __uint_as_float(hu.y), __uint_as_float(lu.x), __uint_as_float(lu.y));
    }
  };
  auto store_A = [&]() {               // ra[] holds split chunks (split_A)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      // LDS slot of the chunk, recomputed once per slice (two constant divisions) rather than held in NA registers
      const int idx = i * NT + tid;
      const int row = idx >> 3, c4 = idx & 7;
      if (row < HROWS) {
        const int o = ((row / HZ) * HZP + row % HZ) * LDKH + c4 * 4;
        *reinterpret_cast<uint2 *>(A_hi + o) = make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y));
        if constexpr (NP == 3) *reinterpret_cast<uint2 *>(A_lo + o) = make_uint2(__float_as_uint(ra[i].z), __float_as_uint(ra[i].w));
      }
    }
  };
  auto load_B = [&](int tap, int cc) {
    const int soff = __builtin_amdgcn_readfirstlane((tap * p.Cout * p.Cin + cc * BK) * 2);
    const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(whr, boff, soff, 0);
    rbh = make_uint4(h[0], h[1], h[2], h[3]);
    if constexpr (NP == 3) {
      const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(wlr, boff, soff, 0);
      rbl = make_uint4(l[0], l[1], l[2], l[3]);
    } else {
      rbl = make_uint4(0, 0, 0, 0);
    }
  };
  auto store_B = [&](int buf) {
    if (tid >= BNV * 4) return;
    __bf16 *b = Bbase + buf * 2 * B_PLANE + bn * LDKH + bc * 8;
    *reinterpret_cast<uint4 *>(b) = rbh;
    if constexpr (NP == 3) *reinterpret_cast<uint4 *>(b + B_PLANE) = rbl;
  };
  auto tap_off = [&](int tap) {
    const int dx = TD ? XO : tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    return ((dx - XO) * HY + (dy - 1)) * HZP + (dz - 1);
  };
  // one k-half (16 channels) of a tap: A fragments of the wave's two row tiles, B fragments of its TN column tiles
  struct Frag { bf16x8 ah[RT], al[RT], bh[TN], bl[TN]; };
  auto read_A = [&](Frag &f, int toff, int kk) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int o = (arow[i] + toff) * LDKH + fh * 8 + kk * 16;
      f.ah[i] = *reinterpret_cast<const bf16x8 *>(A_hi + o);
      if constexpr (NP == 3) f.al[i] = *reinterpret_cast<const bf16x8 *>(A_lo + o);
    }
  };
  auto read_B = [&](Frag &f, int buf, int kk) {
    const __bf16 *b = Bbase + buf * 2 * B_PLANE + (wn * WCOL + fr) * LDKH + fh * 8 + kk * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f.bh[j] = *reinterpret_cast<const bf16x8 *>(b + j * 32 * LDKH);
      if constexpr (NP == 3) f.bl[j] = *reinterpret_cast<const bf16x8 *>(b + B_PLANE + j * 32 * LDKH);
    }
  };
  auto mfma_half = [&](const Frag &f) {
    if constexpr ((SGC_HALO_SKIP & 32) != 0) {          // timing / power builds: everything but the MFMAs
#pragma unroll
      for (int i = 0; i < RT; ++i) asm volatile("" ::"v"(f.ah[i]), "v"(f.al[i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(f.bh[j]), "v"(f.bl[j]));
      return;
    }
    // (s_setprio 2 around this cluster -- the wave that feeds the matrix pipe first at the issue arbiter -- or around everything else:
    //  181.9 / 181.7 against 182.7 us on the 90-GF Winograd layer, 149.7 / 150.1 against 151.3 on 512 -> 512, bit-identical: noise;
    //  profiles/r06_wz_prio.txt.  Not kept.)
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (NP == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = mma_hh<NP>(f.ah[i], f.bh[j], acc[i][j]);
      }
  };
  const int steps_total = (c_hi - c_lo) * NTAP;
  auto step_tap = [&](int st) { return st % NTAP; };
  auto step_cc = [&](int st) { return c_lo + st / NTAP; };

  if constexpr (STG) {
    // P: operands of a tap's first k-half, Q: of its second k-half (32 registers each).  Tap g, every wave:
    //   first half   issue the reads of Q(g); write the weight tile of tap g + 1 (registers loaded during tap g - 1) and
    //                start the load of tap g + 2; multiply P(g) -- in registers since the second half of tap g - 1
    //   BARRIER      publishes tile g + 1; its own waits (Q(g) reads, the tile's ds_writes) ended long before
    //   second half  issue the reads of P(g + 1); multiply Q(g)
    // Hazards: tile g + 1 overwrites tile g - 1, whose last reads (Q(g - 1)) completed before barrier g - 1; P(g + 1) is read
    // behind barrier g, which follows every wave's ds_writes of tile g + 1.  At a slice's last tap the second half reads
    // nothing from the halo image (P of the next slice needs the new image), so barrier g also ends the slice's halo reads:
    // the next slice's image is written under the MFMAs of that second half, one more barrier publishes it.
    // (Weight tiles by LDS-DMA instead of registers + ds_write_b128 -- unpadded swizzled rows, issued a whole tap ahead of the
    //  barrier that publishes them -- were built into this schedule and measured: 0.90 of the lockstep form's time against 0.83
    //  for the register-staged tiles, same box, bit-identical.  Two DMA pieces per wave cost more issue time than two ds_writes.)
    load_A(c_lo);
    load_B(0, c_lo);
    split_A();
    store_A();
    store_B(0);
    if (steps_total > 1) load_B(step_tap(1), step_cc(1));      // stays in registers until tap 0 publishes it
    __syncthreads();
    Frag P = {}, Q = {};
    constexpr int SKIP = SGC_HALO_SKIP;      // timing builds only (diag.hpp); 0 in the product
    if (wave_live) { read_A(P, tap_off(0), 0); read_B(P, 0, 0); }
    int g = 0;
    for (int cc = c_lo; cc < c_hi; ++cc) {
      // the taps are unrolled: a tap's halo offset, its place in the slice and (with the slice's parity) its weight buffer are
      // compile-time constants, so the fragment addresses are one register + an immediate and the tap decode -- ~50 scalar and
      // ~12 vector instructions per tap in the rolled loop (SQ_INSTS_SALU > SQ_INSTS_VALU in round 3's counters) -- is gone
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap, ++g) {
        const bool last_tap = tap == NTAP - 1;
        const bool more = g + 1 < steps_total;
        if (wave_live) {
          if (!(SKIP & 16) || g == 0) read_A(Q, tap_off(tap), 1);
          if (!(SKIP & 8) || g == 0) read_B(Q, g & 1, 1);
        }
        if (!(SKIP & 2) && more) store_B((g + 1) & 1);
        if (!(SKIP & 4) && g + 2 < steps_total) load_B(step_tap(g + 2), step_cc(g + 2));
        if (!(SKIP & 64) && tap == NTAP - 3 && cc + 1 < c_hi) load_A(cc + 1);    // next slice's halo rides under the last taps
        if (wave_live) mfma_half(P);
        if (!(SKIP & 64) && last_tap && cc + 1 < c_hi) split_A();
        if (!(SKIP & 1)) __syncthreads();
        // the fence keeps the refill of P behind the MFMAs that consumed it (hoisted above them it needs a second set of
        // registers) and behind the barrier that publishes the tile it reads
        __builtin_amdgcn_sched_barrier(0);
        if (wave_live && !last_tap) {
          if (!(SKIP & 16)) read_A(P, tap_off(tap + 1), 0);
          if (!(SKIP & 8)) read_B(P, (g + 1) & 1, 0);
        }
        if (!(SKIP & 64) && last_tap && cc + 1 < c_hi) store_A();                // every wave's halo reads of this slice are complete
        if (wave_live) mfma_half(Q);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (cc + 1 < c_hi) {
        __syncthreads();                                          // the new halo image is published
        if (wave_live) { read_A(P, tap_off(0), 0); read_B(P, g & 1, 0); }
      }
    }
  } else {
    // lockstep form.  SGC_HALO_SKIP (diag.hpp; 0 in the product) removes parts of the tap loop in timing builds
    constexpr int SKIP = SGC_HALO_SKIP;
    int g = 0;                       // global step counter -> B buffer
    load_A(c_lo);
    load_B(0, c_lo);
    split_A();
    store_A();
    store_B(0);
    __syncthreads();
    // A fragments of the NEXT tap's first k-half are read before the barrier (the halo is static within a
    // channel slice), so after the barrier only the freshly written B tile has to come out of LDS
    Frag P = {}, Q = {};
    read_A(P, tap_off(0), 0);
    for (int cc = c_lo; cc < c_hi; ++cc) {
      for (int tap = 0; tap < NTAP; ++tap, ++g) {
        const bool last_tap = tap == NTAP - 1;
        if (!(SKIP & 4) && g + 1 < steps_total) load_B(step_tap(g + 1), step_cc(g + 1));
        if (tap == NTAP - 3 && cc + 1 < c_hi) load_A(cc + 1);      // next slice's halo rides under the last taps
        const int toff = tap_off(tap);
        if (wave_live) {
          if (!(SKIP & 8) || g == 0) read_B(P, g & 1, 0);
          if (!(SKIP & 16) || g == 0) read_A(Q, toff, 1);
          if (!(SKIP & 8) || g == 0) read_B(Q, g & 1, 1);
          mfma_half(P);
          mfma_half(Q);
          if (!last_tap && (!(SKIP & 16) || g == 0)) read_A(P, tap_off(tap + 1), 0);
        }
        if (!(SKIP & 2) && g + 1 < steps_total) store_B((g + 1) & 1);
        if (last_tap && cc + 1 < c_hi) split_A();
        if (!(SKIP & 1)) __syncthreads();
      }
      if (cc + 1 < c_hi) {            // every wave is past the last tap: the halo can be replaced
        store_A();
        __syncthreads();
        if (wave_live && (!(SKIP & 16))) read_A(P, tap_off(0), 0);
      }
    }
  }
  SGC_HALO_STAMP(2);
  if constexpr ((SGC_HALO_SKIP & 128) != 0) { if (p.relu != 77) return; }     // timing builds: no epilogue (the condition keeps the MFMAs alive)

  // Epilogue through LDS (as in the implicit-GEMM kernel): the halo / weight buffers are free, the 256 x 128 tile
  // leaves as 16-byte row-contiguous stores instead of 64 four-byte stores per lane.
  if ((p.Cout & 3) == 0 && (p.splitk == 1 || p.ws)) {
    constexpr int LDC = BNV + 8;
    float *cs = reinterpret_cast<float *>(smem_h);           // [MROWS][LDC] floats (139 KB for 256 x 128; launch_halo sizes LDS for it)
    if constexpr (STG) __syncthreads();                      // the second half of the last tap ran after the loop's last barrier
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k)
          cs[(wm * (RT * 32) + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)) * LDC + wn * WCOL + j * 32 + (lane & 31)] = acc[i][j][k];
    __syncthreads();
    constexpr int C4 = BNV / 4;
    for (int e = tid; e < MROWS * C4; e += NT) {
      const int rl = e / C4, c4 = e - rl * C4;
      const int col = n0 + c4 * 4;
      if (col >= p.Cout) continue;
      const int r = vox_tab[rl];
      if (r & kPadRow) continue;
      const int x = X0 + r / (BY * BZ), y = Y0 + (r / BZ) % BY, z = Z0 + r % BZ;
      if (x >= p.gx || y >= p.gy || z >= p.gz) continue;
      const int64_t orow = ((int64_t)x * p.gy + y) * p.gz + z;
      float4 v = *reinterpret_cast<const float4 *>(cs + rl * LDC + c4 * 4);
      if (p.splitk > 1) {
        *reinterpret_cast<float4 *>(p.ws + (int64_t)blockIdx.z * p.ws_stride + orow * p.Cout + col) = v;
        continue;
      }
      if (p.scale) {
        const float4 sc4 = *reinterpret_cast<const float4 *>(p.scale + col);
        v.x *= sc4.x; v.y *= sc4.y; v.z *= sc4.z; v.w *= sc4.w;
      }
      if (p.shift) {
        const float4 sh4 = *reinterpret_cast<const float4 *>(p.shift + col);
        v.x += sh4.x; v.y += sh4.y; v.z += sh4.z; v.w += sh4.w;
      }
      if (p.relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (p.residual) {
        const float4 r4 = *reinterpret_cast<const float4 *>(p.residual + orow * p.Cout + col);
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
      }
      if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (p.act_scale) v = act_col4(v, col, p.act_c0, p.act_c1, *p.act_scale);
      *reinterpret_cast<float4 *>(p.y + orow * p.Cout + col) = v;
    }
    return;
  }

#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WCOL + j * 32 + (lane & 31);
      if (col >= p.Cout) continue;
      const float sc = p.scale ? p.scale[col] : 1.f, sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int r = vox_tab[(wm * RT + i) * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)];
        const int x = X0 + r / (BY * BZ), y = Y0 + (r / BZ) % BY, z = Z0 + r % BZ;
        if ((r & kPadRow) || x >= p.gx || y >= p.gy || z >= p.gz) continue;
        const int64_t orow = ((int64_t)x * p.gy + y) * p.gz + z;
        float *dst = p.y + orow * p.Cout + col;
        if (p.splitk > 1) {
          if (p.ws) p.ws[(int64_t)blockIdx.z * p.ws_stride + orow * p.Cout + col] = acc[i][j][k];
          else atomicAdd(dst, acc[i][j][k]);
        } else {
          float v = acc[i][j][k] * sc + sh;
          if (p.relu == 2) v = fmaxf(v, 0.f);
          if (p.residual) v += p.residual[orow * p.Cout + col];
          if (p.relu == 1) v = fmaxf(v, 0.f);
          if (p.act_scale) v = act_col(v, col, p.act_c0, p.act_c1, *p.act_scale);
          *dst = v;
        }
      }
    }
}

// Forms of this kernel that were built, bit-identical, and measured slower or equal (the code is in the
// history up to round 3): weights by LDS-DMA into a four-stage ring with swizzled unpadded rows (237 vs 230 us warm on the 90-GF
// layer); weights straight from L2 into registers, no barrier per tap (236 vs 229-240, 2-7 % slower elsewhere); three weight
// buffers staged two taps ahead (236 vs 232); v_mfma_f32_16x16x32_bf16 (234 vs 230); a one-wave-per-SIMD form with 512
// registers (309 vs 251).  Round 6, Winograd (2-D, 9 taps per slice) form: the next slice's halo loads issued at the slice's second tap
// and split two chunks per tap under the last four taps instead of one split under the last tap -- 191.5 vs 191.2 us on the 90-GF layer,
// 155.8 vs 151.8 on 512 -> 512 @ 20x20x8 (profiles/r06_wz_skip.txt: the 13 us the restaging costs are not vector-issue time).

// Zero-fill of a split-K accumulation target as a KERNEL, not hipMemsetAsync: a memset captured into a large
// hipGraph (the whole-scene graph) is not ordered with the kernel nodes around it on ROCm 7.2 -- from the second
// replay on the accumulators started from whatever earlier nodes had left in the recycled pool memory
// (tools/scene_graph_check2.py); a kernel node is.
__global__ __launch_bounds__(256) void zero_fill_kernel(float4 *__restrict__ p, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
static int zero_fill(float *y, int64_t n, hipStream_t st) {      // n floats, n % 4 == 0, y 16-byte aligned
  const int64_t n4 = n / 4;
  const int g = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(g > 0 ? g : 1), dim3(256), 0, st, reinterpret_cast<float4 *>(y), n4);
  return check_launch("zero_fill_kernel");
}

// channel-chunk splits of the halo kernel: enough workgroups for 3/4 of the CUs, and no empty split
// (An XCD-aware tile order for the implicit-GEMM kernel -- 1-D launch decoded so that the n-blocks of an m tile, or the
//  m-blocks sharing a weight tile, run back to back on ONE XCD and its L2 serves the repeats -- was measured on the
//  Linears and on the 400-voxel 1024-channel layers: no change (162 vs 165 us, 140 vs 141 us).  The repeats are
//  served by the memory-side cache either way; the limiter is the load -> LDS -> MFMA latency chain, section 4.5.
//  A BK = 64 single-LDS-buffer form of the same kernel -- twice the bytes per thread in flight at the same LDS
//  footprint and occupancy, two barriers per step, half the steps -- was built and is correct but slower where it
//  matters: 203 vs 164 us on the 188,800-row Linear, 168 vs 140 us on the 400-voxel 1024-channel layer; it only wins
//  on the 6,400-row Linears (16 vs 20 us).  The exposed regs -> LDS phase between its two barriers costs more than
//  the deeper loads hide.)
// brick of the halo kernel for a grid: 0 = 4x4x16, 1 = 4x8x8, 2 = 8x8x4
static int halo_brick_shape(int gx, int gy, int gz) {
  if (g_tune_halo_brick == 2) return 2;
  if (g_tune_halo_brick == 0 && gz >= 16 && gx % 8 == 0 && gy % 8 == 0 && gz % 4 == 0) return 2;
  if (gz >= 16 && (g_tune_halo_brick == 0 || g_tune_halo_brick == 3)) return 0;
  if (gz >= 8) return 1;
  return 2;
}

// Whole-grid bricks for the coarsest scale of configs 2 / 3 (10 x 10 x 4 = 400 and 12 x 12 x 4 = 2 x 288 voxels; too few for 256-voxel
// bricks: 61 % of their rows would be padding, and on the tile kernel these weight-streaming layers -- 113 MB of weights for
// 400 voxels -- run at a quarter of the MFMA rate): 1 = one 10 x 10 x 4 brick, 2 = two 6 x 12 x 4 bricks, 0 = none.
// 64-column tiles (the 512 / 384 MFMA rows of such a brick leave the accumulators room for no more).
int g_tune_halo_small = 1;
static int halo_small_grid(int gx, int gy, int gz, int Cout) {
  // measured (tools/small_grid_ab.py, alternated): 1024 -> 1024 109 -> 88 us at 10x10x4, 137 -> 119 at 12x12x4; with 128 output
  // channels (two column tiles x 32 splits) the tile kernel stays ahead (39 vs 47 / 40 us)
  if (!g_tune_halo_small || Cout < 512) return 0;
  if (gx == 10 && gy == 10 && gz == 4) return 1;
  if (gx == 12 && gy == 12 && gz == 4) return 2;
  return 0;
}

int g_tune_halo_wave_fix = 1;  // 1: latency geometry only -- one more split when the workgroups of a launch overflow the CUs by a small
                               // remainder (cfg3: 288 workgroups on 256 CUs took two full rounds); 2: in the throughput geometry too (A/B)

static int halo_splitk(int bricks, int nb, int nchunks) {
  int splitk = 1;
  while (splitk < nchunks && (int64_t)bricks * nb * splitk < g_tune_halo_split_target) splitk *= 2;
  // Wave quantisation (round 5).  One workgroup per CU: W workgroups take ceil(W / CUs) rounds of (fixed + slices * per_slice).  When a
  // launch overflows the CUs by a small remainder -- 144 bricks x 2 column tiles = 288 workgroups on 256 CUs, BASELINE config 3's
  // 256 -> 256 layers at 48 x 48 x 16: two full rounds, 0.36 of the MFMA peak against 0.47 for config 2's 200 workgroups -- halving the
  // slices per workgroup (576 workgroups, three rounds of half the length) is faster alone although it adds a reduce pass.  Only in
  // the LATENCY geometry (halo_split_target >= 192): with scenes in flight the other streams fill the idle CUs of the second round and
  // what counts is CU-time, which a split only raises.  Cost model: 13.6 us fixed + 24.2 us per slice
  // (profiles/r04_halo_fixed_cost.txt), + 20 us for the workspace round trip of a split.
  if (g_tune_halo_wave_fix && (g_tune_halo_split_target >= 192 || g_tune_halo_wave_fix == 2) && splitk == 1 && nchunks >= 2) {
    const int cus = device_cus();                 // queried (a partitioned device or another SKU has another count)
    const int64_t w1 = (int64_t)bricks * nb, w2 = 2 * w1;
    const double t1 = (double)((w1 + cus - 1) / cus) * (13.6 + 24.2 * nchunks);
    const double t2 = (double)((w2 + cus - 1) / cus) * (13.6 + 24.2 * ((nchunks + 1) / 2)) + 20.0;
    if (t2 < 0.92 * t1) splitk = 2;
  }
  const int per = (nchunks + splitk - 1) / splitk;
  return (nchunks + per - 1) / per;
}

int g_tune_halo_stagger = 1;   // halo kernel: 1 software-pipelined schedule with the barrier at mid-tap (see the kernel), 0 lockstep form

template <int BX, int BY, int BZ, int BNV, int NP, bool TD, bool STG, bool WZ = false>
static int launch_halo_k(ConvParamsB &p, int64_t OV, hipStream_t st) {
#if defined(SGC_HALO_STAMPS)
  p.stamps = g_halo_stamp_buf;
#endif
  constexpr int LROWS = (TD ? BX : BX + 2) * (BY + 2) * halo_pitch(BZ);
  constexpr int RGRAN = BNV >= 64 ? 128 : 256;          // rows per (wave rows x 32): see the kernel's wave layout
  constexpr int MROWS = (BX * BY * BZ + RGRAN - 1) / RGRAN * RGRAN;
  const size_t smem = halo_tab_offset(LROWS, MROWS, MROWS > 256 ? BNV : 128) + (MROWS + 256) * sizeof(uint16_t);   // table + 8 x 32 scratch
  static_assert(halo_tab_offset(LROWS, MROWS, MROWS > 256 ? BNV : 128) + (MROWS + 256) * sizeof(uint16_t) <= 160 * 1024, "brick does not fit the LDS");
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void *)conv3d_halo_bf16x3_kernel<BX, BY, BZ, BNV, NP, TD, STG, WZ>, (int)smem, attr_done);
  const int bricks = ceil_div(p.gx, BX) * ceil_div(p.gy, BY) * ceil_div(p.gz, BZ);
  const int nb = ceil_div(p.Cout, BNV);
  const int nchunks = p.Cin / BK;
  int splitk = halo_splitk(bricks, nb, nchunks);
  // the 2-D entry point carries no workspace: one split rather than float atomics (the result must not depend on the run)
  if (TD && !(p.ws && p.ws_floats >= (int64_t)splitk * OV * p.Cout)) splitk = 1;
  p.splitk = splitk;
  if (splitk > 1) {
    if (p.Cout % 4) return set_error(SGC_EUNSUP, "conv3d: split-K path needs Cout %% 4 == 0");
    if (p.ws && p.ws_floats >= (int64_t)splitk * OV * p.Cout) {
      p.ws_stride = OV * p.Cout;
    } else {
      p.ws = nullptr;
      const int rcz = zero_fill(p.y, OV * p.Cout, st);
      if (rcz) return rcz;
    }
  }
  hipLaunchKernelGGL((conv3d_halo_bf16x3_kernel<BX, BY, BZ, BNV, NP, TD, STG, WZ>), dim3(bricks, nb, splitk), dim3(512), smem, st, p);
  return check_launch("conv3d_halo_bf16x3_kernel");
}

template <int BX, int BY, int BZ, int BNV = 128, bool TD = false, bool WZ = false>
static int launch_halo(ConvParamsB &p, int64_t OV, hipStream_t st) {
  if (g_conv_products == 1) return launch_halo_k<BX, BY, BZ, BNV, 1, TD, true, WZ>(p, OV, st);
  if (g_conv_products == 2) return launch_halo_k<BX, BY, BZ, BNV, 2, TD, true, WZ>(p, OV, st);
  if constexpr (WZ) return launch_halo_k<BX, BY, BZ, BNV, 3, TD, true, true>(p, OV, st);
  // the lockstep form is kept for the fp32-faithful mode only: it is the reference of the schedule's bit-identity test, and the
  // form of the whole-grid bricks (four row tiles per wave: the unrolled pipelined loop spills 600 registers there)
  if (!g_tune_halo_stagger || BX * BY * BZ > 256) return launch_halo_k<BX, BY, BZ, BNV, 3, TD, false>(p, OV, st);
  return launch_halo_k<BX, BY, BZ, BNV, 3, TD, true>(p, OV, st);
}

__global__ void conv_epilogue_kernel(float *__restrict__ y, const float *__restrict__ scale,
                                     const float *__restrict__ shift, const float *__restrict__ residual,
                                     int64_t total4, int C4, int relu, const float *__restrict__ ws, int splits,
                                     const float *__restrict__ act_scale, int act_c0, int act_c1) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    float4 v;
    if (ws) {                      // partial tiles of the splits, summed in split order: same bits every run
      v = reinterpret_cast<const float4 *>(ws)[i];
      for (int sidx = 1; sidx < splits; ++sidx) {
        const float4 t = reinterpret_cast<const float4 *>(ws)[(int64_t)sidx * total4 + i];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    } else {
      v = reinterpret_cast<float4 *>(y)[i];
    }
    const float4 sc = scale ? reinterpret_cast<const float4 *>(scale)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? reinterpret_cast<const float4 *>(shift)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (residual) {
      const float4 r = reinterpret_cast<const float4 *>(residual)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (act_scale) v = act_col4(v, c * 4, act_c0, act_c1, *act_scale);
    reinterpret_cast<float4 *>(y)[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Winograd F(2,3) along z for the 3x3x3 stride-1 layers (round 5).  With four scenes in flight the package sits at its power
// limit and the bf16 x 3 products of these layers are about half of a scene's joules: the only lever left on them
// is fewer multiply-adds.  Per (dx, dy) the z direction is a 1-D 3-tap convolution; for an output pair (z = 2j, 2j + 1) with the
// inputs d_k = x[.., 2j - 1 + k], k = 0..3 (zero outside the grid):
//     t0 = d0 - d2,  t1 = d1 + d2,  t2 = d2 - d1,  t3 = d1 - d3                    (input transform)
//     G0 = w0,  G1 = (w0 + w1 + w2) / 2,  G2 = (w0 - w1 + w2) / 2,  G3 = w2        (weight transform, once per module)
//     m_k = sum over (dx, dy, ci) of t_k G_k                                        (four 3 x 3 convolutions over (x, y))
//     y[2j] = m0 + m1 + m2,   y[2j + 1] = m1 - m2 - m3                              (output transform)
// 18 instead of 27 tap-GEMMs per output.  Two launches: the halo kernel's 2-D form convolves a VIRTUAL stack of 4 * Z/2 "images" of
// X x Y pixels (position-major; bricks of 4 images x 8 x 8 pixels, no halo along the image axis) with one weight set per position --
// the input transform is applied while the halo rows are staged (template flag WZ: two voxel rows per chunk, one add, then the
// hi / lo split; a transformed copy of the input never exists) -- and the output transform combines the four results and applies
// the epilogue (scale / shift / relu / residual -- the direct kernel's expressions in the direct kernel's order).
// The fused form does not fit: four accumulators per output pair and four weight tiles per tap need 164 - 189 KB of LDS.
// (The output transform in the TAIL of the convolution launch -- tiles to the scratch with sc1 stores, an arrival counter per brick, the
//  last of a brick's four position workgroups reads the four tiles back behind an agent-scope acquire and stores the output -- was
//  built, bit-identical, and measured: the convolution launch grows from 200-209 to 230-237 us, more than the 12.5 us launch it
//  removes; 557-560 against 563-572 scenes/s with four scenes in flight, 384-388 against 395-400 with one
//  (profiles/r05_wzfuse_ab.txt, also with the four positions adjacent on one XCD).  One CU moves the 1.25 MB of a brick's tiles,
//  residual and output at 50-100 GB/s; 256 CUs in a separate launch do it at the chip's rate.)
// Entries 0, +-1, +-1/2: every transform is exact up to one fp32 rounding per element; measured error against the direct form
// in tests/test_gpu_conv3d.py.  Deterministic (no atomics, fixed order).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void winograd_z_out_kernel(const float4 *__restrict__ m, float4 *__restrict__ y,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             const float4 *__restrict__ residual, int X, int Y, int Z, int C4, int relu,
                                                             const float *__restrict__ act_scale, int act_c0, int act_c1) {
  const int J = Z >> 1;
  const int64_t total = (int64_t)X * Y * J * C4, plane = (int64_t)X * Y * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int j = (int)(r % J); r /= J;
    const int yy = (int)(r % Y), xx = (int)(r / Y);
    const float4 *mi = m + ((int64_t)j * X + xx) * Y * C4 + (int64_t)yy * C4 + c;
    const float4 m0 = mi[0], m1 = mi[(int64_t)J * plane], m2 = mi[(int64_t)2 * J * plane], m3 = mi[(int64_t)3 * J * plane];
    float4 v[2] = {make_float4((m0.x + m1.x) + m2.x, (m0.y + m1.y) + m2.y, (m0.z + m1.z) + m2.z, (m0.w + m1.w) + m2.w),
                   make_float4((m1.x - m2.x) - m3.x, (m1.y - m2.y) - m3.y, (m1.z - m2.z) - m3.z, (m1.w - m2.w) - m3.w)};
    const float4 sc = scale ? reinterpret_cast<const float4 *>(scale)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? reinterpret_cast<const float4 *>(shift)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t o0 = (((int64_t)xx * Y + yy) * Z + 2 * j) * C4 + c;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float4 w = v[q];
      w.x = w.x * sc.x + sh.x; w.y = w.y * sc.y + sh.y; w.z = w.z * sc.z + sh.z; w.w = w.w * sc.w + sh.w;
      if (relu == 2) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
      if (residual) {
        const float4 rr = residual[o0 + (int64_t)q * C4];
        w.x += rr.x; w.y += rr.y; w.z += rr.z; w.w += rr.w;
      }
      if (relu == 1) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
      if (act_scale) w = act_col4(w, c * 4, act_c0, act_c1, *act_scale);
      y[o0 + (int64_t)q * C4] = w;
    }
  }
}

}  // namespace sgc

using namespace sgc;



static int conv_setup(ConvParams &p, const char *who, const float *x, const void *w1, const void *w2, float *y,
                      int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride, int transposed, int relu,
                      int &ox, int &oy, int &oz) {
  if (!x || !w1 || !w2 || !y) return set_error(SGC_EINVAL, "%s: null pointer", who);
  if (Cin <= 0 || Cout <= 0 || ix <= 0 || iy <= 0 || iz <= 0) return set_error(SGC_EINVAL, "%s: bad size", who);
  if (Cin % BK) return set_error(SGC_EUNSUP, "%s: Cin must be a multiple of %d", who, BK);
  if (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)y) & 15)
    return set_error(SGC_EINVAL, "%s: pointers must be 16-byte aligned", who);
  p.x = x; p.y = y; p.Cin = Cin; p.Cout = Cout; p.ix = ix; p.iy = iy; p.iz = iz; p.relu = relu; p.transposed = transposed;
  if (transposed) {
    if (ksize != 2 || stride != 2) return set_error(SGC_EUNSUP, "%s: transposed supports k=2, s=2", who);
    p.gx = ix; p.gy = iy; p.gz = iz; p.ksize = 1; p.stride = 1; p.pad = 0; p.taps = 1;
    ox = 2 * ix; oy = 2 * iy; oz = 2 * iz;
  } else {
    if (!((ksize == 3 || ksize == 1) && (stride == 1 || stride == 2)) && !(ksize == 2 && stride == 2))
      return set_error(SGC_EUNSUP, "%s: ksize in {1,3} with stride in {1,2}, or ksize 2 with stride 2", who);
    p.pad = ksize == 2 ? 0 : ksize / 2; p.ksize = ksize; p.stride = stride; p.taps = ksize * ksize * ksize;   // k2s2: no padding (the
                                                                                                       // dgrad of ConvTranspose3d k2s2)
    ox = (ix + 2 * p.pad - ksize) / stride + 1; oy = (iy + 2 * p.pad - ksize) / stride + 1;
    oz = (iz + 2 * p.pad - ksize) / stride + 1;
    p.gx = ox; p.gy = oy; p.gz = oz;
  }
  if (p.two_d) {                               // images are independent: no taps, no padding, no stride along x
    if (transposed || stride != 1) return set_error(SGC_EUNSUP, "%s: the 2-D form is stride 1, not transposed", who);
    p.taps = ksize * ksize;
    ox = ix; p.gx = ox;
  }
  p.M = p.gx * p.gy * p.gz;
  return SGC_OK;
}

static int pick_splitk(const ConvParams &p, int mb, int nb, int target_blocks) {
  int splitk = 1;
  if (!p.transposed && p.taps == 27) {
    const int64_t blocks = (int64_t)mb * nb;
    while (splitk < 27 && blocks * splitk < target_blocks) splitk *= 3;
  }
  return splitk;
}

// bf16x3 tile kernel (round 6): splits of whole K STEPS.  Groups of whole taps (above) only allow 3 / 9 / 27 splits and overshoot the
// target (100 tiles -> 300 or 900 workgroups on 256 CUs: some CUs carry twice the work of others); steps let a launch land on
// floor(target / tiles) splits of equal length.  split_free = 0 restores the tap groups (A/B), split_min_steps bounds how short a
// split may get (its prologue + the partial tile it stores are fixed costs), split_max the partial tiles the epilogue kernel re-reads.
static int pick_split_steps(const ConvParams &p, int mb, int nb, int target_blocks, int *steps_per) {
  const int total = (p.transposed ? 1 : p.taps) * (p.Cin / BK);
  *steps_per = total;
  if (p.two_d) return 1;                       // the 2-D entry point carries no workspace
  if (!p.transposed && p.taps == 1) return 1;  // 1x1x1 layers are plain GEMMs: one K order everywhere (sgc_level_tail and the row GEMM are
                                               // tested bit for bit against this kernel); splitting them bought 1 us of 18 alone
  if (!g_tune_split_free) {
    const int k = pick_splitk(p, mb, nb, target_blocks);
    *steps_per = total / k;
    return k;
  }
  const int64_t tiles = (int64_t)mb * nb * (p.transposed ? 8 : 1);
  int64_t want = target_blocks / (tiles > 0 ? tiles : 1);
  if (want > total / g_tune_split_min_steps) want = total / g_tune_split_min_steps;
  if (want > g_tune_split_max) want = g_tune_split_max;
  if (want <= 1) return 1;
  const int per = (total + (int)want - 1) / (int)want;
  *steps_per = per;
  return (total + per - 1) / per;
}

static int conv_finish(const ConvParams &p, int64_t OV, hipStream_t st) {
  if (p.splitk <= 1) return SGC_OK;
  if (p.Cout % 4) return set_error(SGC_EUNSUP, "conv3d: split-K path needs Cout %% 4 == 0");
  const int64_t total4 = OV * p.Cout / 4;
  const int g = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(conv_epilogue_kernel, dim3(g), dim3(256), 0, st, p.y, p.scale, p.shift, p.residual, total4,
                     p.Cout / 4, p.relu, (const float *)p.ws, p.splitk, p.act_scale, p.act_c0, p.act_c1);
  return check_launch("conv_epilogue_kernel");
}

// x [ix*iy*iz, Cin] channels-last; wt [taps][Cout][Cin]; y [ox*oy*oz, Cout].
//   ksize 3 (pad 1) or 1 (pad 0), stride 1 or 2;  transposed = 1: ConvTranspose3d(k=2, s=2), wt [8][Cout][Cin]
//   with parity index (px*2+py)*2+pz.  Cin must be a multiple of 32 (zero-pad the channel dim otherwise).
extern "C" int sgc_conv3d_cl_f32(const float *x, const float *wt, const float *scale, const float *shift,
                                 const float *residual_or_null, float *y,
                                 int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                                 int transposed, int relu, float *workspace_or_null, int64_t workspace_floats,
                                 sgc_stream_t stream) {
  ConvParams p = {};
  int ox, oy, oz;
  int rc = conv_setup(p, "sgc_conv3d_cl_f32", x, wt, wt, y, ix, iy, iz, Cin, Cout, ksize, stride, transposed, relu, ox, oy, oz);
  if (rc) return rc;
  p.w = wt; p.scale = scale; p.shift = shift; p.residual = residual_or_null;
  p.ws = workspace_or_null; p.ws_floats = workspace_or_null ? workspace_floats : 0;
  const int64_t OV = (int64_t)ox * oy * oz;
  const bool narrow = Cout <= 32;
  const int bn = narrow ? 32 : 128;
  const int mb = ceil_div(p.M, BM), nb = ceil_div(Cout, bn);
  p.splitk = pick_splitk(p, mb, nb, g_tune_split_target);   // >= 2 workgroups per CU
  hipStream_t st = (hipStream_t)stream;
  if (p.splitk > 1) {
    if (Cout % 4) return set_error(SGC_EUNSUP, "conv3d: split-K path needs Cout %% 4 == 0");
    if (p.ws && p.ws_floats >= (int64_t)p.splitk * OV * Cout) {
      p.ws_stride = OV * Cout;                 // partial tiles -> workspace, summed in order by the epilogue kernel
    } else {
      p.ws = nullptr;                          // no (or too small a) workspace: float atomics into a zeroed y
      const int rcz = zero_fill(y, OV * Cout, st);
      if (rcz) return rcz;
    }
  }
  const dim3 grid(mb, nb, (transposed ? 8 : 1) * p.splitk);
  const size_t smem = (size_t)2 * (BM + bn) * LDK * sizeof(float);
  if (narrow) {
    hipLaunchKernelGGL((conv3d_igemm_f32_kernel<32, 4, 1>), grid, dim3(256), smem, st, p);
  } else {
    static std::atomic<uint64_t> attr_done{0};
    ensure_dynamic_lds((const void *)conv3d_igemm_f32_kernel<128, 2, 2>, (int)smem, attr_done);
    hipLaunchKernelGGL((conv3d_igemm_f32_kernel<128, 2, 2>), grid, dim3(256), smem, st, p);
  }
  rc = check_launch("conv3d_igemm_f32_kernel");
  if (rc) return rc;
  return conv_finish(p, OV, st);
}

// the tile kernel addresses its input and its weights with 32-bit byte offsets into buffer descriptors
static bool igemm_fits_32bit(const ConvParamsB &p) {
  const int64_t lim = 0xfffffff0ll - 65536;       // input: unsigned byte offsets; weights: the per-step offset is a signed scalar
  return (int64_t)p.ix * p.iy * p.iz * p.Cin * 4 < lim && (int64_t)(p.transposed ? 8 : p.taps) * p.Cout * p.Cin * 2 < 0x7fffffffll;
}

// one launch of the tile-per-workgroup implicit-GEMM kernel in the arithmetic mode of g_conv_products
template <int NP>
static void launch_igemm_np(const ConvParamsB &p, bool narrow, dim3 grid, size_t smem, hipStream_t st) {
  static std::atomic<uint64_t> done[2];
  const int big = (int)((size_t)2 * (2 * BM + 2 * 128) * LDKH * sizeof(uint16_t));
  ensure_dynamic_lds((const void *)conv3d_igemm_bf16x3_kernel<128, 2, 2, NP>, big, done[0]);
  ensure_dynamic_lds((const void *)conv3d_igemm_bf16x3_kernel<128, 4, 2, NP>, big, done[1]);
  if (narrow) hipLaunchKernelGGL((conv3d_igemm_bf16x3_kernel<64, 4, 1, NP>), grid, dim3(256), smem, st, p);
  else if (g_tune_conv_waves == 8) hipLaunchKernelGGL((conv3d_igemm_bf16x3_kernel<128, 4, 2, NP>), grid, dim3(512), smem, st, p);
  else hipLaunchKernelGGL((conv3d_igemm_bf16x3_kernel<128, 2, 2, NP>), grid, dim3(256), smem, st, p);
}
static void launch_igemm(const ConvParamsB &p, bool narrow, dim3 grid, size_t smem, hipStream_t st) {
  if (g_conv_products == 1) launch_igemm_np<1>(p, narrow, grid, smem, st);
  else if (g_conv_products == 2) launch_igemm_np<2>(p, narrow, grid, smem, st);
  else launch_igemm_np<3>(p, narrow, grid, smem, st);
}

// Same contract with the weights pre-split on the host: w_hi = bf16(w), w_lo = bf16(w - float(w_hi)),
// both [taps][Cout][Cin] bf16 (raw 16-bit patterns).
static int conv3d_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                         const float *shift, const float *residual_or_null, float *y,
                         int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                         int transposed, int relu, float *workspace_or_null, int64_t workspace_floats,
                         const uint8_t *out_mask_or_null, sgc_stream_t stream, int two_d = 0,
                         const float *act_scale = nullptr, int act_c0 = 0, int act_c1 = 0, int w_group_images = 0, int wz_Z = 0) {
  ConvParamsB p = {};
  p.out_mask = out_mask_or_null;
  p.two_d = two_d;
  p.w_group_images = w_group_images;
  p.wz_Z = wz_Z;
  p.act_scale = act_c1 > act_c0 ? act_scale : nullptr; p.act_c0 = act_c0; p.act_c1 = act_c1;
  int ox, oy, oz;
  int rc = conv_setup(p, "sgc_conv3d_cl_bf16x3", x, w_hi, w_lo, y, ix, iy, iz, Cin, Cout, ksize, stride, transposed, relu, ox, oy, oz);
  if (rc) return rc;
  p.w_hi = reinterpret_cast<const __bf16 *>(w_hi); p.w_lo = reinterpret_cast<const __bf16 *>(w_lo);
  p.scale = scale; p.shift = shift; p.residual = residual_or_null;
  p.ws = workspace_or_null; p.ws_floats = workspace_or_null ? workspace_floats : 0;
  if (!igemm_fits_32bit(p)) return set_error(SGC_EUNSUP, "sgc_conv3d_cl_bf16x3: the input must stay below 4 GiB and the weights below 2 GiB");
  const int64_t OV = (int64_t)ox * oy * oz;
  hipStream_t st = (hipStream_t)stream;
  // 1x1x1 stride-1 layers are row GEMMs (the FFN of a level, TU/encoder.py:311-338): persistent weight-stationary kernel
  if (!transposed && ksize == 1 && stride == 1 && !p.two_d && rows_gemm_supported(Cin, Cout, 0, 0, OV, Cin))
    return rows_gemm_launch(x, Cin, w_hi, w_lo, scale, shift, residual_or_null, y, nullptr, (int)OV, Cin, Cout, relu, 0, 0, 0, st);
  // 3x3x3 stride-1 layers with enough voxels: halo-resident kernel (bricks of 256 voxels)
  if (g_tune_conv_halo && !p.two_d && !transposed && ksize == 3 && stride == 1 && Cout >= g_tune_halo_min_cout && p.M >= g_tune_halo_min_m) {
    const bool narrow_n = g_tune_halo_narrow && Cout <= 64;          // 64-column tiles for layers with <= 64 output channels,
    const bool narrow_32 = g_tune_halo_narrow >= 1 && g_tune_halo_narrow != 64 && Cout <= 32;   // 32-column tiles (8 x 1 waves) for the head's 28 columns
    const int brick = halo_brick_shape(p.gx, p.gy, p.gz);
    if (brick == 0)
      rc = narrow_32 ? launch_halo<4, 4, 16, 32>(p, OV, st) : narrow_n ? launch_halo<4, 4, 16, 64>(p, OV, st) : launch_halo<4, 4, 16>(p, OV, st);
    else if (brick == 1)
      rc = narrow_32 ? launch_halo<4, 8, 8, 32>(p, OV, st) : narrow_n ? launch_halo<4, 8, 8, 64>(p, OV, st) : launch_halo<4, 8, 8>(p, OV, st);
    else
      rc = narrow_32 ? launch_halo<8, 8, 4, 32>(p, OV, st) : narrow_n ? launch_halo<8, 8, 4, 64>(p, OV, st) : launch_halo<8, 8, 4>(p, OV, st);
    if (rc) return rc;
    return conv_finish(p, OV, st);
  }
  if (g_tune_conv_halo && !p.two_d && !transposed && ksize == 3 && stride == 1 && Cout >= g_tune_halo_min_cout && !p.out_mask &&
      halo_small_grid(p.gx, p.gy, p.gz, Cout)) {
    rc = halo_small_grid(p.gx, p.gy, p.gz, Cout) == 1 ? launch_halo<10, 10, 4, 64>(p, OV, st) : launch_halo<6, 12, 4, 64>(p, OV, st);
    if (rc) return rc;
    return conv_finish(p, OV, st);
  }
  // 3 x 3 layers over a stack of images (the FPN's output convolutions): the 2-D form of the halo kernel, bricks of 16 x 16 pixels
  if (g_tune_conv_halo && g_tune_halo_2d && p.two_d && ksize == 3 && stride == 1 && Cout >= g_tune_halo_min_cout &&
      p.M >= g_tune_halo_min_m && p.gy >= 8 && p.gz >= 8) {
    const bool narrow_n = g_tune_halo_narrow && Cout <= 64;
    // bricks of 4 images x 8 x 8 pixels -- no halo along the image axis, 400 staged rows per 256 outputs: the geometry of the
    // transform-domain convolutions of sgc_conv3d_winograd_z_bf16x3 (image groups = positions); halo_2d = 2 selects it for any stack
    if (p.wz_Z > 0 && !narrow_n && p.w_group_images == p.wz_Z / 2 && p.w_group_images % 4 == 0 && p.gx == 2 * p.wz_Z)
      rc = launch_halo<4, 8, 8, 128, true, true>(p, OV, st);
    else if (p.wz_Z > 0) return set_error(SGC_EUNSUP, "conv: the virtual Winograd stack needs Z / 2 a multiple of 4 and > 64 output channels");
    else if ((g_tune_halo_2d == 2 || p.w_group_images > 0) && !narrow_n && p.gx % 4 == 0 && (p.w_group_images == 0 || p.w_group_images % 4 == 0))
      rc = launch_halo<4, 8, 8, 128, true>(p, OV, st);
    else if (p.w_group_images > 0) return set_error(SGC_EUNSUP, "conv: grouped 2-D form needs groups of a multiple of 4 images and > 64 output channels");
    else
    rc = narrow_n ? launch_halo<1, 16, 16, 64, true>(p, OV, st) : launch_halo<1, 16, 16, 128, true>(p, OV, st);
    if (rc) return rc;
    return conv_finish(p, OV, st);
  }
  if (p.w_group_images > 0 || p.wz_Z > 0) return set_error(SGC_EUNSUP, "conv: the grouped 2-D form runs on the halo kernel only (stack too small?)");
  const bool narrow = Cout <= 64;
  const int bn = narrow ? 64 : 128;
  const int mb = ceil_div(p.M, BM), nb = ceil_div(Cout, bn);
  p.splitk = pick_split_steps(p, mb, nb, g_tune_split_target, &p.steps_per);
  if (p.splitk > 1) {
    if (Cout % 4) return set_error(SGC_EUNSUP, "conv3d: split-K path needs Cout %% 4 == 0");
    if (p.ws && p.ws_floats >= (int64_t)p.splitk * OV * Cout) {
      p.ws_stride = OV * Cout;                 // partial tiles -> workspace, summed in order by the epilogue kernel
    } else {
      p.ws = nullptr;                          // no (or too small a) workspace: float atomics into a zeroed y
      const int rcz = zero_fill(y, OV * Cout, st);
      if (rcz) return rcz;
    }
  }
  const dim3 grid(mb, nb, (transposed ? 8 : 1) * p.splitk);
  const size_t smem = (size_t)2 * (2 * BM + 2 * bn) * LDKH * sizeof(uint16_t);
  p.xcd_deal = (p.taps > 1 || transposed) && !p.two_d ? g_tune_igemm_xcd : 0;
  launch_igemm(p, narrow, grid, smem, st);
  rc = check_launch("conv3d_igemm_bf16x3_kernel");
  if (rc) return rc;
  return conv_finish(p, OV, st);
}


// ---------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions (training): dW[tap][co][ci] = sum_o dy[o][co] * x[nbr(o, tap)][ci],
// a GEMM per tap with M = Cout, N = Cin and the OUTPUT VOXELS as the reduction dimension.  Both operands are stored
// voxel-major (rows = k), so the staging pass transposes: a thread loads a 4-voxel x 4-channel block (four 16-byte loads
// from four rows), splits it hi/lo and writes four 8-byte runs of 4 consecutive k into the [channel][k] LDS image the
// forward kernel's fragment reads expect.  128 x 128 tile, K-step 32 voxels, 4 waves (2 x 2, 64 x 64 each), register
// prefetch of step s + 1 under the MFMAs of step s, double-buffered LDS.  The voxel range is split over blockIdx.z
// (taps x splits); partial tiles go to a workspace and are summed in split order (deterministic), or straight to dW
// when there is one split.  ksize 1 | 3 (pad k/2, stride 1 | 2) or 2 (stride 2, no pad: the ConvTranspose3d k2s2 layers
// with x := the fine-grid tensor and dy := the coarse one).
// ---------------------------------------------------------------------------------------------
struct WgradParams {
  const float *x, *dy;
  float *out;               // dW [taps][Cout][Cin] (one split) or the workspace [splits][taps][Cout][Cin]
  int Cin, Cout;
  int ix, iy, iz, ox, oy, oz;
  int ksize, stride, pad, taps;
  int OV, ksteps, splits, steps_per_split;
  int ax, by, cz;           // 32 = ax * (oy * oz) + by * oz + cz: the per-step advance of a voxel's (x, y, z) (see load_step)
};

// WM = 2: 4 waves (2 x 2, 64 x 64 each), every thread stages one block of BOTH operands; WM = 4: 8 waves (4 x 2, 32 x 64 each),
// threads 0-255 stage the dy tile and 256-511 the x tile (half the loads, conversions and registers per thread, twice the
// waves to hide them).
template <int WM>
__global__ __launch_bounds__(WM * 128) void conv3d_wgrad_bf16x3_kernel(const WgradParams p) {
  constexpr int TMW = 4 / WM * 1;                           // 32-row tiles per wave along M: 2 (WM = 2) or 1 (WM = 4)
  constexpr int BMW = 128, BNW = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  constexpr int PLANE = 128 * LDKH, BUF = 4 * PLANE;        // per buffer: A_hi, A_lo, B_hi, B_lo of [128][LDKH]
  __bf16 *base = reinterpret_cast<__bf16 *>(smem_w);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int co0 = blockIdx.x * BMW, ci0 = blockIdx.y * BNW;
  const int tap = blockIdx.z % p.taps, split = blockIdx.z / p.taps;
  const int s_lo = split * p.steps_per_split, s_hi = min(p.ksteps, s_lo + p.steps_per_split);
  int dx = 0, dy_ = 0, dz = 0;
  if (p.ksize > 1) { dx = tap / (p.ksize * p.ksize); dy_ = (tap / p.ksize) % p.ksize; dz = tap % p.ksize; }

  const int role = WM == 4 ? __builtin_amdgcn_readfirstlane(tid >> 8) : 2;   // 0: stages dy, 1: stages x, 2: both (wave-uniform)
  const int kb = tid & 7, cb = (tid & 255) >> 3;            // this thread's block: voxels 4 kb .. + 3 of the step, channels 4 cb .. + 3
  const bool a_ok = role != 1 && co0 + 4 * cb < p.Cout, b_ok = role != 0 && ci0 + 4 * cb < p.Cin;
  float4 ra[4], rb[4];
  // Addressing without a division in the loop: a thread's four voxels advance by 32 per step, so their (x, y, z) are carried
  // (32 = ax * oy * oz + by * oz + cz, uniform digits from the host: z += cz, y += by + carry, x += ax + carry) instead of decoded from the flat index with two runtime divisions per voxel and step; loads go
  // through buffer descriptors, an out-of-range offset (voxel past OV, neighbour outside the volume, channel block past the
  // tensor) returns zeros, so there is no branch either.
  constexpr unsigned OOB = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.dy), 0, (int)(unsigned)((int64_t)p.OV * p.Cout * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.x), 0, (int)(unsigned)((int64_t)p.ix * p.iy * p.iz * p.Cin * 4), 0x00020000);
  int vx[4], vy[4], vz[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = s_lo * 32 + 4 * kb + j;                   // may be >= OV: the coordinates then run past ox and the loads are OOB
    vz[j] = o % p.oz; vy[j] = (o / p.oz) % p.oy; vx[j] = o / (p.oz * p.oy);
  }
  int o_base = s_lo * 32 + 4 * kb;
  auto load_step = [&]() {                                  // loads the step the carried coordinates stand at, then advances them
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = o_base + j;
      const bool live = o < p.OV;
      if (role != 1) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(dyr, live && a_ok ? (unsigned)(o * p.Cout + co0 + 4 * cb) * 4u : OOB, 0, 0);
        ra[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
      }
      if (role != 0) {
        const int xx = vx[j] * p.stride + dx - p.pad, yy = vy[j] * p.stride + dy_ - p.pad, zz = vz[j] * p.stride + dz - p.pad;
        const bool in = live && b_ok && xx >= 0 && xx < p.ix && yy >= 0 && yy < p.iy && zz >= 0 && zz < p.iz;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
            xr, in ? ((unsigned)((xx * p.iy + yy) * p.iz + zz) * (unsigned)p.Cin + ci0 + 4 * cb) * 4u : OOB, 0, 0);
        rb[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
      }
    }
    if (role != 0) {                                        // carry the coordinates to the next step: selects only, no branch
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vz[j] += p.cz;                                      // 32 = ax * (oy * oz) + by * oz + cz: one carry per digit, always exact
        const int c1 = vz[j] >= p.oz ? 1 : 0;
        vz[j] -= c1 ? p.oz : 0;
        vy[j] += p.by + c1;
        const int c2 = vy[j] >= p.oy ? 1 : 0;
        vy[j] -= c2 ? p.oy : 0;
        vx[j] += p.ax + c2;
      }
    }
    o_base += 32;
  };
  auto store_block = [&](const float4 (&r)[4], __bf16 *hi, __bf16 *lo) {
    const float v[4][4] = {{r[0].x, r[0].y, r[0].z, r[0].w}, {r[1].x, r[1].y, r[1].z, r[1].w},
                           {r[2].x, r[2].y, r[2].z, r[2].w}, {r[3].x, r[3].y, r[3].z, r[3].w}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {                           // channel 4 cb + c: its 4 consecutive k
      bf16x4 h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 hb = (__bf16)v[j][c];
        h[j] = hb;
        l[j] = (__bf16)(v[j][c] - (float)hb);
      }
      const int o = (4 * cb + c) * LDKH + 4 * kb;
      *reinterpret_cast<bf16x4 *>(hi + o) = h;
      *reinterpret_cast<bf16x4 *>(lo + o) = l;
    }
  };
  auto store_step = [&](int buf) {
    __bf16 *a_hi = base + buf * BUF;
    if (role != 1) store_block(ra, a_hi, a_hi + PLANE);
    if (role != 0) store_block(rb, a_hi + 2 * PLANE, a_hi + 3 * PLANE);
  };

  f32x16 acc[TMW][2];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  if (s_lo < s_hi) {
    load_step();
    store_step(0);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int s = s_lo; s < s_hi; ++s) {
      const int buf = (s - s_lo) & 1;
      if (s + 1 < s_hi) load_step();
      const __bf16 *a_hi = base + buf * BUF + (wm * (32 * TMW) + fr) * LDKH + fh * 8;
      const __bf16 *a_lo = a_hi + PLANE;
      const __bf16 *b_hi = base + buf * BUF + 2 * PLANE + (wn * 64 + fr) * LDKH + fh * 8;
      const __bf16 *b_lo = b_hi + PLANE;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 ah[TMW], al[TMW], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8 *>(a_hi + i * 32 * LDKH + kk * 16);
          al[i] = *reinterpret_cast<const bf16x8 *>(a_lo + i * 32 * LDKH + kk * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bh[i] = *reinterpret_cast<const bf16x8 *>(b_hi + i * 32 * LDKH + kk * 16);
          bl[i] = *reinterpret_cast<const bf16x8 *>(b_lo + i * 32 * LDKH + kk * 16);
        }
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      if (s + 1 < s_hi) store_step(buf ^ 1);
      __syncthreads();
    }
  }
  // a lane owns column ci = lane & 31 of a 32 x 32 tile: for a fixed register the 32 lanes of a half-wave store 128
  // contiguous bytes of one dW row
  float *out = p.out + ((int64_t)split * p.taps + tap) * p.Cout * p.Cin;
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
      if (ci >= p.Cin) continue;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int co = co0 + wm * (32 * TMW) + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
        if (co < p.Cout) out[(int64_t)co * p.Cin + ci] = acc[i][j][k];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Halo form of the weight gradient for the 3x3x3 stride-1 layers (round 4).  The tile kernel above is one GEMM per tap: every
// (co, ci) tile streams dy AND the tap-shifted x for its voxel range, so a layer moves 27 x both operands (2.8 GB for
// 256 -> 256 at 40x40x16) and splits each element to bf16 27 times: it runs at 0.29 PF/s, bound by its staging.  Here a
// workgroup owns 32 output channels x 32 input channels and ALL 27 taps and walks bricks of 8 x 8 x 4 voxels: per brick it
// stages dy [256 voxels][32 co] and the brick's x halo [10 x 10 x 6 rows][32 ci] ONCE (split hi / lo once) and multiplies
// them 27 times.  Both operands are stored voxel-major -- rows = the reduction index -- which is what the MFMA wants
// transposed: the fragments come out of LDS through ds_read_b64_tr_b16 (a 4-row x 16-column block per 16 lanes, delivered
// column-major; lane map checked in tools/probe/tr16_probe.hip), so no transposing pass exists anywhere.  Wave w owns taps
// w, w + 8, w + 16 (, w + 24): its accumulators are 3 - 4 tiles of 32 x 32; the dy fragments of a k-step (16 voxels) are read
// once per wave and reused for its taps, the x fragments of a tap are the same LDS rows shifted by the tap's halo offset --
// with the k-steps unrolled every read is one register + an immediate.  The brick range is split over workgroups; partial
// sums go through the workspace and wgrad_reduce_kernel (fixed order: deterministic).
// ---------------------------------------------------------------------------------------------
struct WgradHaloParams {
  const float *x, *dy;
  float *out;               // dW [27][Cout][Cin] (one split) or the workspace [splits][27][Cout][Cin]
  int Cin, Cout;
  int gx, gy, gz;           // grid (input = output grid: stride 1, padding 1)
  int nbricks, bricks_per_split;
};

typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

// MT = 32-channel dy tiles per workgroup (2: 64 output channels x 32 input channels; every x fragment feeds two MFMA triples,
// which halves the LDS reads per MFMA -- with MT = 1 the k-loop is LDS-read-bound).  Rows are stored with NO padding (pitch
// 32 bf16 = 16 banks): a transposed read touches 4 rows x 16 banks per half-wave and the staging writes are contiguous, both
// conflict-free; a padded pitch of 40 makes row q = 3 alias row 0.
// NW = waves per workgroup (8: two per SIMD, 256 registers each).  Fragments are read right before their MFMAs: with two waves
// per SIMD the partner's MFMAs cover the LDS round trip (a one-wave-per-SIMD variant of THIS form with read-ahead was
// built and spilled 168 registers; the double-buffered kernel below is the form that uses one wave per SIMD).
template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void conv3d_wgrad_halo_kernel(const WgradHaloParams p) {
  constexpr int BX = 8, BY = 8, BZ = 4, HY = BY + 2, HZ = BZ + 2, HROWS = (BX + 2) * HY * HZ;      // 600 halo rows
  constexpr int PW = 32;                                                                            // row pitch (bf16)
  constexpr int NT = 64 * NW;
  constexpr int X_PLANE = HROWS * PW, D_IMG = 256 * PW, D_PLANE = MT * D_IMG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
  __bf16 *X_hi = reinterpret_cast<__bf16 *>(smem_g), *X_lo = X_hi + X_PLANE;
  __bf16 *D_hi = X_lo + X_PLANE, *D_lo = D_hi + D_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups go to the 8 XCDs round-robin by linear id: all (co, ci) tiles of one brick range are put on ONE XCD, so the
  // 2.3 x (Cout / 64) re-reads of x and the (Cin / 32) re-reads of dy are L2 hits (without it a 256 -> 256 layer at
  // 40 x 40 x 16 pulls 455 MB through the fabric and the loads, not the MFMAs, set the time).  gridDim.z is the split count
  // rounded up to a multiple of 8; the surplus workgroups leave at once.
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), tiles = gridDim.x * gridDim.y;
  int split = blockIdx.z, tile = blockIdx.x + gridDim.x * blockIdx.y;
  if ((gridDim.z & 7) == 0) { split = (lin & 7) + 8 * ((lin >> 3) / tiles); tile = (lin >> 3) % tiles; }
  const int co0 = (tile % gridDim.x) * (32 * MT), ci0 = (tile / gridDim.x) * 32;
  const int b_lo = split * p.bricks_per_split, b_hi = min(p.nbricks, b_lo + p.bricks_per_split);
  if (b_lo >= b_hi) return;
  const int nby = (p.gy + BY - 1) / BY, nbz = (p.gz + BZ - 1) / BZ;
  constexpr unsigned OOB = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.x), 0, (int)(unsigned)((int64_t)p.gx * p.gy * p.gz * p.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.dy), 0, (int)(unsigned)((int64_t)p.gx * p.gy * p.gz * p.Cout * 4), 0x00020000);
  // staging assignment: the x halo goes slab by slab (hx = 0 .. 9; a slab is 60 rows (hy, hz) x 8 float4, APASS passes of
  // RPP rows), dy voxel-row by voxel-row: the x index of every global load is wave-uniform, (y, z) are per-thread constants
  constexpr int APASS = 512 / NT, RPP = HY * HZ / APASS, NA = (BX + 2) * APASS;
  constexpr int DCH = 8 * MT, DROWS = NT / DCH, ND = 256 / DROWS;      // float4 per dy row, dy rows per pass, passes
  const int a_r = tid >> 3, a_c4 = tid & 7;
  const int d_r = tid / DCH, d_c4 = tid % DCH;
  float4 ra[NA], rd[ND];
  auto load_brick = [&](int b) {
    const int bk = b % nbz, bj = (b / nbz) % nby, bi = b / (nbz * nby);
    const int X0 = bi * BX, Y0 = bj * BY, Z0 = bk * BZ;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int row = j * RPP + a_r, y = Y0 + row / HZ - 1, z = Z0 + row % HZ - 1;
      const bool in = a_r < RPP && y >= 0 && y < p.gy && z >= 0 && z < p.gz;
      const unsigned voff = in ? ((unsigned)(y * p.gz + z) * (unsigned)p.Cin + ci0 + a_c4 * 4) * 4u : OOB;
#pragma unroll
      for (int i = 0; i < BX + 2; ++i) {
        const int x = X0 + i - 1;                                                       // uniform
        const bool xin = x >= 0 && x < p.gx;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, xin ? voff : OOB, xin ? (int)((unsigned)x * (unsigned)(p.gy * p.gz) * (unsigned)p.Cin * 4u) : 0, 0);
        ra[i * APASS + j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int vx = i * DROWS + d_r, r = vx % (BY * BZ);                               // voxel of the brick, its (by, bz) row
      const int x = X0 + vx / (BY * BZ), y = Y0 + r / BZ, z = Z0 + r % BZ;
      const bool xin = x < p.gx, in = y < p.gy && z < p.gz;
      const unsigned voff = in && xin ? ((unsigned)(y * p.gz + z) * (unsigned)p.Cout + co0 + d_c4 * 4) * 4u : OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(dr, voff, xin ? (int)((unsigned)x * (unsigned)(p.gy * p.gz) * (unsigned)p.Cout * 4u) : 0, 0);
      rd[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
  };
  auto split_store = [&](const float4 &r, __bf16 *hi, __bf16 *lo, int o) {
    const float v[4] = {r.x, r.y, r.z, r.w};
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 hb = (__bf16)v[e];
      h[e] = hb;
      l[e] = (__bf16)(v[e] - (float)hb);
    }
    *reinterpret_cast<bf16x4 *>(hi + o) = h;
    *reinterpret_cast<bf16x4 *>(lo + o) = l;
  };
  auto store_brick = [&]() {
    if (a_r < RPP) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
        split_store(ra[i], X_hi, X_lo, ((i / APASS) * (HY * HZ) + (i % APASS) * RPP + a_r) * PW + a_c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < ND; ++i)
      split_store(rd[i], D_hi, D_lo, (d_c4 >> 3) * D_IMG + (i * DROWS + d_r) * PW + (d_c4 & 7) * 4);
  };
  // transposed fragment reads: group g = lane >> 4 reads the block of rows (8 (g >> 1) + q [+ 4]) x columns 16 (g & 1) .. + 15;
  // lane 4 q + pp of the group supplies the address of row q, columns 4 pp .. 4 pp + 3
  const int g4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, hh = g4 >> 1;
  const int d_lane = (8 * hh + q) * PW + 16 * (g4 & 1) + 4 * pp;           // + (16 s + 4 rd) * PW
  const int x_lane = (12 * hh + q) * PW + 16 * (g4 & 1) + 4 * pp;          // + (row(s, rd) + toff(tap)) * PW, see below
  auto tr4 = [&](const __bf16 *ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v *)ptr);
  };
  auto frag = [&](const __bf16 *p0, const __bf16 *p1) {
    const bf16x4v a = tr4(p0), b = tr4(p1);
    bf16x8 f;
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    return f;
  };
  constexpr int TPW = (27 + NW - 1) / NW;              // tap slots of a wave: taps wid + NW t; the last slot may be empty
  const bool has_last = wid + NW * (TPW - 1) < 27;     // wave-uniform
  int toff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tt = (t < TPW - 1 || has_last) ? wid + NW * t : 13;
    toff[t] = (((tt / 9 - 1) * HY + ((tt / 3) % 3 - 1)) * HZ + (tt % 3 - 1)) * PW;
  }
  f32x16 acc[TPW][MT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[t][m][k] = 0.f;

  bf16x8 ah[1][MT], al[1][MT], bh[1], bl[1];
  auto read_A = [&](int s, int buf) {                  // dy fragments of k-step s: voxels 16 s .. 16 s + 15 of the brick
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const __bf16 *dp = D_hi + m * D_IMG + d_lane + 16 * s * PW;
      ah[buf][m] = frag(dp, dp + 4 * PW);
      al[buf][m] = frag(dp + D_PLANE, dp + D_PLANE + 4 * PW);
    }
  };
  auto read_B = [&](int s, int t, int buf) {           // x fragments of k-step s shifted by tap slot t
    // halo row of voxel 16 s + 8 hh + 4 rd + q:  ((s >> 1) + 1) * 60 + (4 (s & 1) + 2 hh + rd + 1) * 6 + q + 1
    const int r0 = ((s >> 1) + 1) * (HY * HZ) + (4 * (s & 1) + 1) * HZ + 1;
    const __bf16 *xp = X_hi + x_lane + r0 * PW + toff[t];
    bh[buf] = frag(xp, xp + HZ * PW);
    bl[buf] = frag(xp + X_PLANE, xp + X_PLANE + HZ * PW);
  };

  constexpr int WSKIP = SGC_WGRAD_SKIP;
  if (WSKIP & 6) {
#pragma unroll
    for (int m = 0; m < MT; ++m) ah[0][m] = al[0][m] = bf16x8{};
    bh[0] = bl[0] = bf16x8{};
  }
  if (WSKIP & 16) {
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int i = 0; i < ND; ++i) rd[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  }
  if (b_lo < b_hi && !(WSKIP & 16)) load_brick(b_lo);
  for (int b = b_lo; b < b_hi; ++b) {
    __syncthreads();                                   // every wave is done with the previous brick's images
    if (!(WSKIP & 8) || b == b_lo) store_brick();
    __syncthreads();
    if (b + 1 < b_hi && !(WSKIP & 16)) load_brick(b + 1);               // lands under this brick's MFMAs
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (!(WSKIP & 4)) read_A(s, 0);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (!(WSKIP & 2) && (t < TPW - 1 || has_last)) read_B(s, t, 0);
        if (WSKIP & 1) {                               // keep the reads alive
          asm volatile("" ::"v"(bh[0]), "v"(bl[0]));
          if (t == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(ah[0][m]), "v"(al[0][m]));
          }
        } else if (t < TPW - 1 || has_last) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0][m], bh[0], acc[t][m], 0, 0, 0);
            acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0][m], bl[0], acc[t][m], 0, 0, 0);
            acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0][m], bh[0], acc[t][m], 0, 0, 0);
          }
        }
      }
    }
  }
  // a lane owns column ci = lane & 31 of its 32 x 32 tiles: the 32 lanes of a half-wave store 128 contiguous bytes of one dW row
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (t == TPW - 1 && !has_last) break;
    float *out = p.out + ((int64_t)split * 27 + (wid + NW * t)) * p.Cout * p.Cin;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int co = co0 + 32 * m + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
        out[(int64_t)co * p.Cin + ci0 + (lane & 31)] = acc[t][m][k];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Double-buffered form of the halo weight gradient (round 4, late).  Timing builds of the form above (SGC_WGRAD_SKIP) show its
// phases ADD: MFMAs alone 145 us, + fragment reads 186, + the split / LDS stores of a brick and its global loads 238 -- between
// the two barriers of a brick nothing multiplies.  Here a brick is 4 x 8 x 4 voxels (halo 6 x 10 x 6 = 360 rows), BOTH LDS images
// exist twice (154 KB), four waves (one per SIMD, 512 registers) own 7 taps x MT tiles each, and the staging of brick b + 1 is
// cut into four pieces that ride behind the MFMAs of k-steps 0 - 3 of brick b (register -> split -> LDS), the global loads of
// brick b + 2 behind k-steps 4 - 7: ONE barrier per brick, the matrix pipe never waits for staging.  Fragments of the next tap
// are read ahead of the MFMAs of the current one.  Same sums in the same order per workgroup as the form above is NOT
// guaranteed (bricks differ): the two forms agree to fp32 summation order.
// ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void conv3d_wgrad_halo2_kernel(const WgradHaloParams p) {
  constexpr int BX = 4, BY = 8, BZ = 4, NVB = BX * BY * BZ, HY = BY + 2, HZ = BZ + 2, HROWS = (BX + 2) * HY * HZ;   // 360 halo rows
  constexpr int PW = 32, NT = 256, NW = 4, KS = NVB / 16;
  constexpr int X_PLANE = HROWS * PW, D_IMG = NVB * PW, D_PLANE = MT * D_IMG;
  constexpr int X_BUF = 2 * X_PLANE, D_BUF = 2 * D_PLANE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_g2[];
  __bf16 *Xb = reinterpret_cast<__bf16 *>(smem_g2);             // [2 buffers][hi | lo][HROWS][32]
  __bf16 *Db = Xb + 2 * X_BUF;                                  // [2 buffers][hi | lo][MT][128][32]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), tiles = gridDim.x * gridDim.y;
  int split = blockIdx.z, tile = blockIdx.x + gridDim.x * blockIdx.y;
  if ((gridDim.z & 7) == 0) { split = (lin & 7) + 8 * ((lin >> 3) / tiles); tile = (lin >> 3) % tiles; }   // a brick range on one XCD
  const int co0 = (tile % gridDim.x) * (32 * MT), ci0 = (tile / gridDim.x) * 32;
  const int b_lo = split * p.bricks_per_split, b_hi = min(p.nbricks, b_lo + p.bricks_per_split);
  if (b_lo >= b_hi) return;
  const int nby = (p.gy + BY - 1) / BY, nbz = (p.gz + BZ - 1) / BZ;
  constexpr unsigned OOB = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.x), 0, (int)(unsigned)((int64_t)p.gx * p.gy * p.gz * p.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.dy), 0, (int)(unsigned)((int64_t)p.gx * p.gy * p.gz * p.Cout * 4), 0x00020000);
  // staging: x halo slab by slab (hx = 0 .. 5; 60 rows x 8 float4 in two passes of 30 rows), dy in passes of DROWS voxels
  constexpr int RPP = HY * HZ / 2, NA = (BX + 2) * 2;
  constexpr int DCH = 8 * MT, DROWS = NT / DCH, ND = NVB / DROWS;
  constexpr int NCH = NA + ND;                                    // float4 chunks of a brick per thread: x halo, then dy
  static_assert(NCH <= 4 * 6, "one chunk per tap slot 0 .. 5 of four k-steps");
  // threads 240 .. 255 have no halo row of their own in a pass: they repeat row RPP - 1 (same loads, same values, same LDS
  // address -- a benign duplicate) so that the staging code has no predicate and can be interleaved with the MFMAs
  const int a_r = min(tid >> 3, RPP - 1), a_c4 = tid & 7;
  const int d_r = tid / DCH, d_c4 = tid % DCH;
  float4 rs[NCH];
  int X0 = 0, Y0 = 0, Z0 = 0;                                     // origin of the brick being loaded
  unsigned a_voff[2];
  auto load_begin = [&](int b) {
    const int bk = b % nbz, bj = (b / nbz) % nby, bi = b / (nbz * nby);
    X0 = bi * BX; Y0 = bj * BY; Z0 = bk * BZ;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * RPP + a_r, y = Y0 + row / HZ - 1, z = Z0 + row % HZ - 1;
      const bool in = y >= 0 && y < p.gy && z >= 0 && z < p.gz;
      a_voff[j] = in ? ((unsigned)(y * p.gz + z) * (unsigned)p.Cin + ci0 + a_c4 * 4) * 4u : OOB;
    }
  };
  auto load_chunk = [&](int c) {
    u32x4 v;
    if (c < NA) {
      const int slab = c >> 1, j = c & 1;
      const int x = X0 + slab - 1;                                                      // uniform
      const bool xin = x >= 0 && x < p.gx;
      v = __builtin_amdgcn_raw_buffer_load_b128(xr, xin ? a_voff[j] : OOB, xin ? (int)((unsigned)x * (unsigned)(p.gy * p.gz) * (unsigned)p.Cin * 4u) : 0, 0);
    } else {
      const int vx = (c - NA) * DROWS + d_r, r = vx % (BY * BZ);
      const int x = X0 + vx / (BY * BZ), y = Y0 + r / BZ, z = Z0 + r % BZ;
      const bool in = x < p.gx && y < p.gy && z < p.gz;
      const unsigned voff = in ? ((unsigned)((x * p.gy + y) * p.gz + z) * (unsigned)p.Cout + co0 + d_c4 * 4) * 4u : OOB;
      v = __builtin_amdgcn_raw_buffer_load_b128(dr, voff, 0, 0);
    }
    rs[c] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  };
  auto store_chunk = [&](int c, int buf) {
    const float v[4] = {rs[c].x, rs[c].y, rs[c].z, rs[c].w};
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 hb = (__bf16)v[e];
      h[e] = hb;
      l[e] = (__bf16)(v[e] - (float)hb);
    }
    __bf16 *hi;
    int plane;
    if (c < NA) {
      hi = Xb + buf * X_BUF + ((c >> 1) * (HY * HZ) + (c & 1) * RPP + a_r) * PW + a_c4 * 4;
      plane = X_PLANE;
    } else {
      hi = Db + buf * D_BUF + (d_c4 >> 3) * D_IMG + ((c - NA) * DROWS + d_r) * PW + (d_c4 & 7) * 4;
      plane = D_PLANE;
    }
    *reinterpret_cast<bf16x4 *>(hi) = h;
    *reinterpret_cast<bf16x4 *>(hi + plane) = l;
  };
  const int g4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, hh = g4 >> 1;
  const int d_lane = (8 * hh + q) * PW + 16 * (g4 & 1) + 4 * pp;
  const int x_lane = (12 * hh + q) * PW + 16 * (g4 & 1) + 4 * pp;
  auto tr4 = [&](const __bf16 *ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v *)ptr);
  };
  auto frag = [&](const __bf16 *p0, const __bf16 *p1) {
    const bf16x4v a = tr4(p0), b = tr4(p1);
    bf16x8 f;
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    return f;
  };
  constexpr int TPW = 7;                               // tap slots of a wave: taps wid + 4 t; wave 3's last slot is empty
  const bool has_last = wid + NW * (TPW - 1) < 27;     // wave-uniform
  int toff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tt = (t < TPW - 1 || has_last) ? wid + NW * t : 13;
    toff[t] = (((tt / 9 - 1) * HY + ((tt / 3) % 3 - 1)) * HZ + (tt % 3 - 1)) * PW;
  }
  f32x16 acc[TPW][MT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[t][m][k] = 0.f;
  bf16x8 ah[2][MT], al[2][MT], bh[3], bl[3];      // x fragments two tap slots ahead of their MFMAs
  auto read_A = [&](int s, int fb, int buf) {
    const __bf16 *D_hi = Db + buf * D_BUF;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const __bf16 *dp = D_hi + m * D_IMG + d_lane + 16 * s * PW;
      ah[fb][m] = frag(dp, dp + 4 * PW);
      al[fb][m] = frag(dp + D_PLANE, dp + D_PLANE + 4 * PW);
    }
  };
  auto read_B = [&](int s, int t, int fb, int buf) {
    const int r0 = ((s >> 1) + 1) * (HY * HZ) + (4 * (s & 1) + 1) * HZ + 1;
    const __bf16 *xp = Xb + buf * X_BUF + x_lane + r0 * PW + toff[t];
    bh[fb] = frag(xp, xp + HZ * PW);
    bl[fb] = frag(xp + X_PLANE, xp + X_PLANE + HZ * PW);
  };

  constexpr int WSKIP = SGC_WGRAD_SKIP;                // timing builds only (diag.hpp); 0 in the product
  if (WSKIP & 6) {
#pragma unroll
    for (int i = 0; i < 3; ++i) bh[i] = bl[i] = bf16x8{};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int m = 0; m < MT; ++m) ah[i][m] = al[i][m] = bf16x8{};
  }
  load_begin(b_lo);
#pragma unroll
  for (int c = 0; c < NCH; ++c) load_chunk(c);
#pragma unroll
  for (int c = 0; c < NCH; ++c) store_chunk(c, 0);
  load_begin(min(b_lo + 1, b_hi - 1));
#pragma unroll
  for (int c = 0; c < NCH; ++c) load_chunk(c);
  __syncthreads();
  for (int b = b_lo; b < b_hi; ++b) {
    const int buf = (b - b_lo) & 1;
    // the last brick(s) of the range restage themselves once more (into the buffer nobody reads): no condition, so the
    // staging chunks below sit in the same straight-line regions as the MFMAs
    const int b_load = min(b + 2, b_hi - 1);
    // slot n = s * TPW + t of the brick; its x fragments live in ring entry n % 3 and are read two slots ahead
    if (!(WSKIP & 4)) read_A(0, 0, buf);
    if (!(WSKIP & 2)) { read_B(0, 0, 0, buf); read_B(0, 1, 1, buf); }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int n = s * TPW + t, n2 = n + 2, s2 = n2 / TPW, t2 = n2 % TPW;
        if (s2 < KS) {
          if (t2 == 0 && !(WSKIP & 4)) read_A(s2, s2 & 1, buf);        // dy fragments of the next k-step, two slots ahead as well
          if (!(WSKIP & 2)) read_B(s2, t2, n2 % 3, buf);
        }
        // one staging chunk per tap slot 0 .. 5: k-steps 0 - 3 split + store brick b + 1 (loaded during the previous brick),
        // k-steps 4 - 7 load brick b + 2.  The chunk's ~30 vector instructions go BETWEEN this slot's MFMAs (group fences):
        // a lone wave per SIMD issues in order, so a lump of staging code after the MFMAs would leave the pipe idle
        const int ch = (s & 3) * 6 + t;
        const bool stage = t < 6 && ch < NCH;
        if (WSKIP & 1) {
          asm volatile("" ::"v"(bh[n % 3]), "v"(bl[n % 3]));
          if (t == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(ah[s & 1][m]), "v"(al[s & 1][m]));
          }
        } else if (t < TPW - 1 || has_last) {
          // product-major: consecutive MFMAs go to different accumulators (the group fences below keep this order)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s & 1][m], bh[n % 3], acc[t][m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s & 1][m], bl[n % 3], acc[t][m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s & 1][m], bh[n % 3], acc[t][m], 0, 0, 0);
        }
        if (stage) {
          if (s < 4) { if (!(WSKIP & 8)) store_chunk(ch, buf ^ 1); }
          else if (!(WSKIP & 16)) { if (ch == 0) load_begin(b_load); load_chunk(ch); }
        }
        // issue order of the slot: after every MFMA a share of the slot's LDS reads (they feed the slot after next) and of
        // the staging chunk's vector work.  A wave issues in order and an LDS instruction holds the issue port for several
        // cycles: four reads in a row, or a lump of staging code, let the matrix pipe run dry behind the one MFMA in flight
        {
          const bool two = s2 < KS && t2 == 0, st = stage && s < 4;       // 4 + 4 MT reads in the slot / a store chunk in it
#pragma unroll
          for (int i = 0; i < 3 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (two) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (st) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          }
          if (st) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                                   // publishes brick b + 1; every wave is done reading brick b
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (t == TPW - 1 && !has_last) break;
    float *out = p.out + ((int64_t)split * 27 + (wid + NW * t)) * p.Cout * p.Cin;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int co = co0 + 32 * m + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
        out[(int64_t)co * p.Cin + ci0 + (lane & 31)] = acc[t][m][k];
      }
  }
}

namespace sgc { int g_tune_wgrad_halo = 1; }       // 3x3x3 stride-1 layers with Cin, Cout multiples of 32: 1 double-buffered halo form, 2 single-buffered, 0 tile kernel
static bool wgrad_halo_geometry(WgradHaloParams &h, int &mt, int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride) {
  if (!g_tune_wgrad_halo || ksize != 3 || stride != 1 || (Cin & 31) || (Cout & 31)) return false;
  if ((int64_t)ix * iy * iz * (Cin > Cout ? Cin : Cout) * 4 >= 0xfffffff0ll - 65536) return false;
  h.Cin = Cin; h.Cout = Cout; h.gx = ix; h.gy = iy; h.gz = iz;
  const bool db = g_tune_wgrad_halo == 1;                                       // double-buffered form (default): bricks of 4 x 8 x 4
  h.nbricks = ceil_div(ix, db ? 4 : 8) * ceil_div(iy, 8) * ceil_div(iz, 4);
  if ((int64_t)h.nbricks * (db ? 128 : 256) > (int64_t)2 * ix * iy * iz) return false;       // bricks mostly padding: the tile kernel wins
  mt = (Cout & 63) ? 1 : 2;
  const int tiles = (Cout / (32 * mt)) * (Cin / 32);
  const int splits = std::max(1, std::min(h.nbricks / 4, ceil_div(256, tiles)));      // fill the chip; >= 4 bricks per workgroup
  h.bricks_per_split = ceil_div(h.nbricks, splits);
  return true;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float4 *__restrict__ ws, float4 *__restrict__ dw, int64_t n4, int splits) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = ws[i];
    for (int s = 1; s < splits; ++s) {                      // fixed order: deterministic
      const float4 b = ws[(int64_t)s * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    dw[i] = a;
  }
}

static int wgrad_geometry(WgradParams &p, int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride) {
  if (Cin <= 0 || Cout <= 0 || ix <= 0 || iy <= 0 || iz <= 0) return set_error(SGC_EINVAL, "sgc_conv3d_wgrad_bf16x3: bad size");
  if ((Cin & 3) || (Cout & 3)) return set_error(SGC_EUNSUP, "sgc_conv3d_wgrad_bf16x3: Cin and Cout must be multiples of 4");
  if (ksize == 2) {
    if (stride != 2 || (ix & 1) || (iy & 1) || (iz & 1)) return set_error(SGC_EUNSUP, "sgc_conv3d_wgrad_bf16x3: ksize 2 needs stride 2 and an even grid");
    p.pad = 0;
  } else if ((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2)) {
    p.pad = ksize / 2;
  } else {
    return set_error(SGC_EUNSUP, "sgc_conv3d_wgrad_bf16x3: ksize in {1,2,3}, stride in {1,2}");
  }
  p.Cin = Cin; p.Cout = Cout; p.ix = ix; p.iy = iy; p.iz = iz; p.ksize = ksize; p.stride = stride;
  p.taps = ksize * ksize * ksize;
  p.ox = (ix + 2 * p.pad - ksize) / stride + 1; p.oy = (iy + 2 * p.pad - ksize) / stride + 1; p.oz = (iz + 2 * p.pad - ksize) / stride + 1;
  p.OV = p.ox * p.oy * p.oz;
  p.ksteps = ceil_div(p.OV, 32);
  const int tiles = ceil_div(Cout, 128) * ceil_div(Cin, 128) * p.taps;
  int splits = 1;
  while (tiles * splits < 512 && p.ksteps / (splits * 2) >= 16) splits *= 2;     // fill the chip twice over; >= 16 K-steps per split
  // a layer that still leaves most CUs idle (the nn.Linear layers of a level: 800 rows, 4 tiles) is bound by the load latency
  // of its serial K-steps (~2.5 us each), not by flops: spread the steps over idle CUs, down to 3 per workgroup
  while (tiles * splits < 256 && p.ksteps / (splits * 2) >= 3) splits *= 2;
  p.steps_per_split = ceil_div(p.ksteps, splits);
  p.splits = ceil_div(p.ksteps, p.steps_per_split);
  p.ax = 32 / (p.oy * p.oz); p.by = (32 % (p.oy * p.oz)) / p.oz; p.cz = 32 % p.oz;
  if ((int64_t)p.OV * Cout * 4 >= 0xfffffff0ll - 65536 || (int64_t)ix * iy * iz * Cin * 4 >= 0xfffffff0ll - 65536)
    return set_error(SGC_EUNSUP, "sgc_conv3d_wgrad_bf16x3: x and dy must stay below 4 GiB each");
  return SGC_OK;
}

extern "C" int64_t sgc_conv3d_wgrad_workspace_floats(int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride) {
  WgradHaloParams h = {};
  int mt = 1;
  if (wgrad_halo_geometry(h, mt, ix, iy, iz, Cin, Cout, ksize, stride)) {
    const int splits = ceil_div(h.nbricks, h.bricks_per_split);
    return splits > 1 ? (int64_t)splits * 27 * Cout * Cin : 0;
  }
  WgradParams p = {};
  if (wgrad_geometry(p, ix, iy, iz, Cin, Cout, ksize, stride)) return -1;
  return p.splits > 1 ? (int64_t)p.splits * p.taps * Cout * Cin : 0;
}

extern "C" int sgc_conv3d_wgrad_bf16x3(const float *x, const float *dy, float *dw, int ix, int iy, int iz, int Cin, int Cout,
                                       int ksize, int stride, float *workspace_or_null, int64_t workspace_floats,
                                       sgc_stream_t stream) {
  if (!x || !dy || !dw) return set_error(SGC_EINVAL, "sgc_conv3d_wgrad_bf16x3: null pointer");
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace_or_null) & 15)
    return set_error(SGC_EINVAL, "sgc_conv3d_wgrad_bf16x3: pointers must be 16-byte aligned");
  hipStream_t st0 = (hipStream_t)stream;
  WgradHaloParams h = {};
  int mt = 1;
  if (wgrad_halo_geometry(h, mt, ix, iy, iz, Cin, Cout, ksize, stride)) {
    int splits = ceil_div(h.nbricks, h.bricks_per_split);
    const int64_t n27 = (int64_t)27 * Cout * Cin;
    if (splits > 1 && !(workspace_or_null && workspace_floats >= splits * n27)) { splits = 1; h.bricks_per_split = h.nbricks; }
    h.x = x; h.dy = dy; h.out = splits > 1 ? workspace_or_null : dw;
    const size_t smem_h = (size_t)2 * (600 + 256 * mt) * 32 * sizeof(uint16_t);
    static std::atomic<uint64_t> attr_h[4] = {};
    const dim3 grid_h(Cout / (32 * mt), Cin / 32, splits >= 8 ? (splits + 7) / 8 * 8 : splits);
    if (g_tune_wgrad_halo == 1) {                      // double-buffered bricks of 4 x 8 x 4 (default)
      const size_t smem2 = (size_t)2 * 2 * (360 + 128 * mt) * 32 * sizeof(uint16_t);
      if (mt == 2) {
        ensure_dynamic_lds((const void *)conv3d_wgrad_halo2_kernel<2>, (int)smem2, attr_h[0]);
        hipLaunchKernelGGL(conv3d_wgrad_halo2_kernel<2>, grid_h, dim3(256), smem2, st0, h);
      } else {
        ensure_dynamic_lds((const void *)conv3d_wgrad_halo2_kernel<1>, (int)smem2, attr_h[1]);
        hipLaunchKernelGGL(conv3d_wgrad_halo2_kernel<1>, grid_h, dim3(256), smem2, st0, h);
      }
    } else if (mt == 2) {                              // single-buffered bricks of 8 x 8 x 4, eight waves
      ensure_dynamic_lds((const void *)conv3d_wgrad_halo_kernel<2, 8>, (int)smem_h, attr_h[2]);
      hipLaunchKernelGGL((conv3d_wgrad_halo_kernel<2, 8>), grid_h, dim3(512), smem_h, st0, h);
    } else {
      ensure_dynamic_lds((const void *)conv3d_wgrad_halo_kernel<1, 8>, (int)smem_h, attr_h[3]);
      hipLaunchKernelGGL((conv3d_wgrad_halo_kernel<1, 8>), grid_h, dim3(512), smem_h, st0, h);
    }
    int rch = check_launch("conv3d_wgrad_halo_kernel");
    if (rch) return rch;
    if (splits > 1) {
      const int64_t n4 = n27 / 4;
      const int g = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(g), dim3(256), 0, st0, reinterpret_cast<const float4 *>(workspace_or_null),
                         reinterpret_cast<float4 *>(dw), n4, splits);
      rch = check_launch("wgrad_reduce_kernel");
    }
    return rch;
  }
  WgradParams p = {};
  int rc = wgrad_geometry(p, ix, iy, iz, Cin, Cout, ksize, stride);
  if (rc) return rc;
  p.x = x; p.dy = dy;
  const int64_t n = (int64_t)p.taps * Cout * Cin;
  if (p.splits > 1 && !(workspace_or_null && workspace_floats >= p.splits * n)) {     // no workspace: one split (slower, same result class)
    p.splits = 1;
    p.steps_per_split = p.ksteps;
  }
  p.out = p.splits > 1 ? workspace_or_null : dw;
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)2 * 4 * 128 * LDKH * sizeof(uint16_t);
  static std::atomic<uint64_t> attr_done{0};
  static std::atomic<uint64_t> attr_done8{0};
  ensure_dynamic_lds((const void *)conv3d_wgrad_bf16x3_kernel<2>, (int)smem, attr_done);
  ensure_dynamic_lds((const void *)conv3d_wgrad_bf16x3_kernel<4>, (int)smem, attr_done8);
  const dim3 wgrid(ceil_div(Cout, 128), ceil_div(Cin, 128), p.taps * p.splits);
  if (g_tune_wgrad_waves == 8) hipLaunchKernelGGL(conv3d_wgrad_bf16x3_kernel<4>, wgrid, dim3(512), smem, st, p);
  else hipLaunchKernelGGL(conv3d_wgrad_bf16x3_kernel<2>, wgrid, dim3(256), smem, st, p);
  rc = check_launch("conv3d_wgrad_bf16x3_kernel");
  if (rc) return rc;
  if (p.splits > 1) {
    const int64_t n4 = n / 4;
    const int g = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(g), dim3(256), 0, st, reinterpret_cast<const float4 *>(workspace_or_null),
                       reinterpret_cast<float4 *>(dw), n4, p.splits);
    rc = check_launch("wgrad_reduce_kernel");
  }
  return rc;
}

// Split-K workspace (floats) the convolution above would use for a deterministic reduction; 0 = the layer is not
// split.  Mirrors the dispatch of sgc_conv3d_cl_f32 (bf16x3 = 0) / sgc_conv3d_cl_bf16x3 (bf16x3 = 1).
extern "C" int sgc_conv3d_cl_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                                    const float *shift, const float *residual_or_null, float *y,
                                    int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                                    int transposed, int relu, float *workspace_or_null, int64_t workspace_floats,
                                    sgc_stream_t stream) {
  return conv3d_bf16x3(x, w_hi, w_lo, scale, shift, residual_or_null, y, ix, iy, iz, Cin, Cout, ksize, stride, transposed, relu,
                       workspace_or_null, workspace_floats, nullptr, stream);
}

// 2-D convolution over a stack of channels-last images (the producer side of the hand-over -- the FPN's
// output convolutions, mmdet FPN.fpn_convs as configured by configs/SGCDet_ScanNet.py:84-88 and called at detectors/
// SGCDet.py:67, emitting the [N, H*W, C] rows the view transformation consumes, TU/transformer.py:151-170, without an
// NCHW round trip).  x [N*H*W, Cin] -> y [N*H*W, Cout], ksize 1 | 3 (pad k/2), stride 1; same bf16x3 arithmetic and
// epilogue (scale / shift / residual / relu) as the 3-D entry point, on the implicit-GEMM kernel with the taps confined to
// the (row, column) plane.
extern "C" int sgc_conv2d_nhwc_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                                      const float *shift, const float *residual_or_null, float *y, int N, int H, int W,
                                      int Cin, int Cout, int ksize, int relu, sgc_stream_t stream) {
  if (ksize != 1 && ksize != 3) return set_error(SGC_EUNSUP, "sgc_conv2d_nhwc_bf16x3: ksize in {1,3}");
  return conv3d_bf16x3(x, w_hi, w_lo, scale, shift, residual_or_null, y, N, H, W, Cin, Cout, ksize, 1, 0, relu, nullptr, 0,
                       nullptr, stream, 1);
}

// Output-masked 3x3x3 stride-1 convolution (the decoder tail / head of the neck, where only voxels in -- or next to --
// the refined set are consumed, necks/imvoxelnet.py:47-64, dense_heads/imvoxel_head_v2.py:258,301): live rows are
// bit-identical to sgc_conv3d_cl_bf16x3.  Layers the halo kernel does not take run dense (the mask is a licence, not a
// duty).
extern "C" int sgc_conv3d_cl_bf16x3_masked(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                                           const float *shift, const float *residual_or_null, float *y,
                                           const uint8_t *out_mask, int ix, int iy, int iz, int Cin, int Cout, int relu,
                                           float *workspace_or_null, int64_t workspace_floats, sgc_stream_t stream) {
  if (!out_mask) return set_error(SGC_EINVAL, "sgc_conv3d_cl_bf16x3_masked: null mask");
  return conv3d_bf16x3(x, w_hi, w_lo, scale, shift, residual_or_null, y, ix, iy, iz, Cin, Cout, 3, 1, 0, relu,
                       workspace_or_null, workspace_floats, out_mask, stream);
}

// 3x3x3 stride-1 convolution whose columns [act_c0, act_c1) leave as expf(v * *act_scale_dev) -- the head's fused
// centerness | reg | cls convolution with `torch.exp(scale(reg))` (dense_heads/imvoxel_head_v2.py:79,103-110: mmcv Scale, a
// learnable scalar, then exp) in the epilogue instead of two elementwise launches per scale.  out_mask_or_null as in
// sgc_conv3d_cl_bf16x3_masked.  The other columns are bit-identical to sgc_conv3d_cl_bf16x3.
extern "C" int sgc_conv3d_cl_bf16x3_act(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *scale,
                                        const float *shift, const float *residual_or_null, float *y,
                                        const uint8_t *out_mask_or_null, int ix, int iy, int iz, int Cin, int Cout, int relu,
                                        int act_c0, int act_c1, const float *act_scale_dev,
                                        float *workspace_or_null, int64_t workspace_floats, sgc_stream_t stream) {
  if (!act_scale_dev || act_c0 < 0 || act_c1 > Cout || act_c1 <= act_c0)
    return set_error(SGC_EINVAL, "sgc_conv3d_cl_bf16x3_act: needs 0 <= act_c0 < act_c1 <= Cout and a scale pointer");
  return conv3d_bf16x3(x, w_hi, w_lo, scale, shift, residual_or_null, y, ix, iy, iz, Cin, Cout, 3, 1, 0, relu,
                       workspace_or_null, workspace_floats, out_mask_or_null, stream, 0, act_scale_dev, act_c0, act_c1);
}

extern "C" int sgc_conv3d_winograd_z_supported(int ix, int iy, int iz, int Cin, int Cout) {
  // Z/2 images per position, in bricks of 4; the 2-D halo form wants slices of at least 8 x 8 pixels and g_tune_halo_min_m rows in the stack
  return ix >= 8 && iy >= 8 && iz >= 8 && iz % 8 == 0 && Cin % 32 == 0 && Cout % 4 == 0 && Cout > 64 && Cout >= g_tune_halo_min_cout &&
                 (int64_t)2 * iz * ix * iy >= g_tune_halo_min_m && g_tune_conv_halo && g_tune_halo_2d
             ? 1 : 0;
}
extern "C" int64_t sgc_conv3d_winograd_z_workspace_floats(int ix, int iy, int iz, int Cin, int Cout) {
  (void)Cin;
  return sgc_conv3d_winograd_z_supported(ix, iy, iz, Cin, Cout) ? (int64_t)2 * ix * iy * iz * Cout : 0;    // the four transform-domain outputs
}

// 3x3x3 stride-1 convolution through the Winograd F(2,3) transform along z (see the kernels above): wg_hi / wg_lo are the bf16 hi / lo
// planes of the TRANSFORMED weights [4][9][Cout][Cin] (position, (dx, dy) tap); workspace >= 2 V Cout floats (the four transform-domain
// outputs; the input transform is fused into the halo staging, so no transformed copy of the input exists): two launches.
extern "C" int sgc_conv3d_winograd_z_bf16x3(const float *x, const uint16_t *wg_hi, const uint16_t *wg_lo, const float *scale,
                                            const float *shift, const float *residual_or_null, float *y, int ix, int iy, int iz,
                                            int Cin, int Cout, int relu, float *workspace, int64_t workspace_floats,
                                            sgc_stream_t stream) {
  if (!x || !wg_hi || !wg_lo || !y || !workspace) return set_error(SGC_EINVAL, "sgc_conv3d_winograd_z_bf16x3: null pointer");
  if (!sgc_conv3d_winograd_z_supported(ix, iy, iz, Cin, Cout))
    return set_error(SGC_EUNSUP, "sgc_conv3d_winograd_z_bf16x3: needs ix, iy >= 8, iz %% 8 == 0, Cin %% 32 == 0, Cout %% 4 == 0, Cout > 64, >= 2048 stack rows");
  if (workspace_floats < sgc_conv3d_winograd_z_workspace_floats(ix, iy, iz, Cin, Cout))
    return set_error(SGC_EINVAL, "sgc_conv3d_winograd_z_bf16x3: workspace too small");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)workspace | (uintptr_t)residual_or_null | (uintptr_t)scale | (uintptr_t)shift) & 15)
    return set_error(SGC_EINVAL, "sgc_conv3d_winograd_z_bf16x3: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t V = (int64_t)ix * iy * iz;
  const int J = iz / 2;
  float *m = workspace;
  const int64_t n_out = V / 2 * (Cout / 4);
  // four 3 x 3 convolutions over (x, y) as ONE launch of the halo kernel's 2-D form on the VIRTUAL stack of 4 J images (the input
  // transform happens while the halo rows are staged: template flag WZ), weight set = image / J
  int rc = conv3d_bf16x3(x, wg_hi, wg_lo, nullptr, nullptr, nullptr, m, 4 * J, ix, iy, Cin, Cout, 3, 1, 0, 0, nullptr, 0, nullptr, stream, 1,
                         nullptr, 0, 0, J, iz);
  if (rc) return rc;
  hipLaunchKernelGGL(winograd_z_out_kernel, dim3((unsigned)std::min<int64_t>((n_out + 255) / 256, 65536)), dim3(256), 0, st,
                     reinterpret_cast<const float4 *>(m), reinterpret_cast<float4 *>(y), scale, shift,
                     reinterpret_cast<const float4 *>(residual_or_null), ix, iy, iz, Cout / 4, relu, nullptr, 0, 0);
  return check_launch("winograd_z_out_kernel");
}

// 3x3x3 dilation of a {0,1} voxel mask (what a 3x3x3 convolution must produce so that its consumer is exact on `in`)
__global__ void mask_dilate3_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int X, int Y, int Z) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= (int64_t)X * Y * Z) return;
  const int z = (int)(v % Z), y = (int)((v / Z) % Y), x = (int)(v / ((int64_t)Z * Y));
  uint8_t any = 0;
  for (int dx = -1; dx <= 1; ++dx)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dz = -1; dz <= 1; ++dz) {
        const int a = x + dx, b = y + dy, c = z + dz;
        if (a >= 0 && a < X && b >= 0 && b < Y && c >= 0 && c < Z) any |= in[((int64_t)a * Y + b) * Z + c];
      }
  out[v] = any ? 1 : 0;
}

extern "C" int sgc_mask_dilate3(const uint8_t *mask_in, uint8_t *mask_out, int X, int Y, int Z, sgc_stream_t stream) {
  if (!mask_in || !mask_out || mask_in == mask_out) return set_error(SGC_EINVAL, "sgc_mask_dilate3: null or aliased pointers");
  if (X <= 0 || Y <= 0 || Z <= 0) return set_error(SGC_EINVAL, "sgc_mask_dilate3: bad size");
  const int64_t n = (int64_t)X * Y * Z;
  hipLaunchKernelGGL(mask_dilate3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask_in, mask_out, X, Y, Z);
  return check_launch("mask_dilate3_kernel");
}

// valid masks of the head's scales: nn.Upsample(size, mode='trilinear')(valid.float()).round().bool()
// (dense_heads/imvoxel_head_v2.py:123,258) for integer factors f = 2^s: align_corners=False puts every coarse voxel half-way
// between fine voxels f*d + f/2 - 1 and f*d + f/2 on each axis, i.e. the mean of 8 fine voxels; round() is half-to-even,
// so a coarse voxel is valid iff at least 5 of the 8 are.
__global__ void valid_pyramid_kernel(const int64_t *__restrict__ valid, uint8_t *__restrict__ out, int X, int Y, int Z, int f) {
  const int cx = X / f, cy = Y / f, cz = Z / f;
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= (int64_t)cx * cy * cz) return;
  const int z = (int)(v % cz), y = (int)((v / cz) % cy), x = (int)(v / ((int64_t)cz * cy));
  if (f == 1) { out[v] = valid[v] != 0; return; }
  const int o = f / 2 - 1;
  int cnt = 0;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int c = 0; c < 2; ++c)
        cnt += valid[((int64_t)(x * f + o + a) * Y + (y * f + o + b)) * Z + (z * f + o + c)] != 0;
  out[v] = cnt >= 5;
}

extern "C" int sgc_valid_pyramid(const int64_t *valid, uint8_t *mask_out, int X, int Y, int Z, int factor, sgc_stream_t stream) {
  if (!valid || !mask_out) return set_error(SGC_EINVAL, "sgc_valid_pyramid: null pointer");
  if (X <= 0 || Y <= 0 || Z <= 0 || factor < 1 || (factor & (factor - 1)) || X % factor || Y % factor || Z % factor)
    return set_error(SGC_EUNSUP, "sgc_valid_pyramid: factor must be a power of two dividing the grid");
  const int64_t n = (int64_t)(X / factor) * (Y / factor) * (Z / factor);
  hipLaunchKernelGGL(valid_pyramid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, valid, mask_out, X, Y, Z, factor);
  return check_launch("valid_pyramid_kernel");
}

extern "C" int64_t sgc_conv3d_workspace_floats(int ix, int iy, int iz, int Cin, int Cout, int ksize, int stride,
                                               int transposed, int bf16x3) {
  if (ix <= 0 || iy <= 0 || iz <= 0 || Cin <= 0 || Cout <= 0) return 0;
  int ox, oy, oz, gx, gy, gz;
  if (transposed) { ox = 2 * ix; oy = 2 * iy; oz = 2 * iz; gx = ix; gy = iy; gz = iz; }
  else {
    const int pad = ksize == 2 ? 0 : ksize / 2;
    ox = (ix + 2 * pad - ksize) / stride + 1; oy = (iy + 2 * pad - ksize) / stride + 1; oz = (iz + 2 * pad - ksize) / stride + 1;
    gx = ox; gy = oy; gz = oz;
  }
  const int64_t OV = (int64_t)ox * oy * oz, M = (int64_t)gx * gy * gz;
  int splitk = 1;
  if (bf16x3 && g_tune_conv_halo && !transposed && ksize == 3 && stride == 1 && Cout >= g_tune_halo_min_cout && M >= g_tune_halo_min_m) {
    const int shape = halo_brick_shape(gx, gy, gz);
    const int bx = shape == 2 ? 8 : 4, by = shape == 0 ? 4 : 8, bz = shape == 0 ? 16 : (shape == 1 ? 8 : 4);
    splitk = halo_splitk(ceil_div(gx, bx) * ceil_div(gy, by) * ceil_div(gz, bz), ceil_div(Cout, (g_tune_halo_narrow && Cout <= 64) ? 64 : 128), Cin / BK);
  } else if (bf16x3 && g_tune_conv_halo && !transposed && ksize == 3 && stride == 1 && Cout >= g_tune_halo_min_cout && halo_small_grid(gx, gy, gz, Cout)) {
    splitk = halo_splitk(halo_small_grid(gx, gy, gz, Cout), ceil_div(Cout, 64), Cin / BK);
    // a MASKED call on these grids takes the tile kernel (the whole-grid brick carries no output mask): size for whichever
    // form splits further, so that neither ever falls back to float atomics for want of workspace
    ConvParams p = {};
    p.taps = 27; p.Cin = Cin; p.M = (int)M; p.gx = gx; p.gy = gy; p.gz = gz;
    int per;
    const int tile_split = pick_split_steps(p, ceil_div((int)M, BM), ceil_div(Cout, Cout <= 64 ? 64 : 128), g_tune_split_target, &per);
    if (tile_split > splitk) splitk = tile_split;
  } else {
    ConvParams p = {};
    p.transposed = transposed; p.taps = transposed ? 1 : ksize * ksize * ksize;
    p.Cin = Cin; p.M = (int)M; p.gx = gx; p.gy = gy; p.gz = gz;
    p.taps = transposed ? 8 : ksize * ksize * ksize;
    const int bn = bf16x3 ? (Cout <= 64 ? 64 : 128) : (Cout <= 32 ? 32 : 128);
    int per;
    splitk = bf16x3 ? pick_split_steps(p, ceil_div((int)M, BM), ceil_div(Cout, bn), g_tune_split_target, &per)
                    : pick_splitk(p, ceil_div((int)M, BM), ceil_div(Cout, bn), g_tune_split_target);
  }
  return splitk > 1 ? (int64_t)splitk * OV * Cout : 0;
}

// y[rows, Cout] = x[rows, Cin] @ W^T + shift with the row count on the DEVICE: the pair-list stages size their
// GEMMs by the number of visible (camera, voxel) pairs, which sgc_compact_pairs leaves in totals[] -- reading it
// back costs a host round trip per level.  The grid covers rows_cap; workgroups past *rows_dev exit at once.
static int linear_rows(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                       float *y, const int32_t *rows_dev_or_null, int rows_cap, int Cin, int Cout, int hm_S, int hm_cm,
                       int hm_bf16, sgc_stream_t stream, float *zero_row = nullptr) {
  if (!x || !w_hi || !w_lo || !y) return set_error(SGC_EINVAL, "sgc_linear_rows_bf16x3: null pointer");
  if (rows_cap <= 0) return SGC_OK;
  if (Cin % 32 || Cout % 4) return set_error(SGC_EUNSUP, "sgc_linear_rows_bf16x3: needs Cin %% 32 == 0 and Cout %% 4 == 0");
  // (a variant with the A operand resident in registers -- one wave owning 32 rows for the whole K = 256
  //  reduction, weights streamed through LDS in 32-column tiles -- was built, bit-identical, and measured:
  //  159 vs 175 us on 188,800 rows but 57 vs 41 us on 77,000 x 128 and 41 vs 18 us on 6,400 rows: its 10 us
  //  load-and-split prologue per workgroup is not amortised.  Not adopted.)
  if (((uintptr_t)x | (uintptr_t)w_hi | (uintptr_t)w_lo | (uintptr_t)y) & 15)
    return set_error(SGC_EINVAL, "sgc_linear_rows_bf16x3: pointers must be 16-byte aligned");
  if (rows_gemm_supported(Cin, Cout, hm_cm, hm_S, rows_cap, Cin))
    return rows_gemm_launch(x, Cin, w_hi, w_lo, nullptr, shift, nullptr, y, rows_dev_or_null, rows_cap, Cin, Cout, 0, hm_S, hm_cm,
                            hm_bf16, (hipStream_t)stream, zero_row);
  ConvParamsB p = {};
  p.zero_row = zero_row;
  p.x = x; p.w_hi = reinterpret_cast<const __bf16 *>(w_hi); p.w_lo = reinterpret_cast<const __bf16 *>(w_lo);
  p.y = y; p.shift = shift;
  p.Cin = Cin; p.Cout = Cout;
  p.ix = rows_cap; p.iy = 1; p.iz = 1; p.gx = rows_cap; p.gy = 1; p.gz = 1;
  p.ksize = 1; p.stride = 1; p.pad = 0; p.taps = 1; p.splitk = 1; p.M = rows_cap; p.m_dev = rows_dev_or_null;
  p.hm_S = hm_S; p.hm_cm = hm_cm; p.hm_bf16 = hm_bf16;
  const bool narrow = Cout <= 64;
  const int bn = narrow ? 64 : 128;
  const dim3 grid(ceil_div(rows_cap, BM), ceil_div(Cout, bn), 1);
  const size_t smem = (size_t)2 * (2 * BM + 2 * bn) * LDKH * sizeof(uint16_t);
  hipStream_t st = (hipStream_t)stream;
  if (!igemm_fits_32bit(p)) return set_error(SGC_EUNSUP, "sgc_linear_rows_bf16x3: the input must stay below 4 GiB and the weights below 2 GiB");
  launch_igemm(p, narrow, grid, smem, st);
  return check_launch("conv3d_igemm_bf16x3_kernel (linear rows)");
}

extern "C" int sgc_linear_rows_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                                      float *y, const int32_t *rows_dev_or_null, int rows_cap, int Cin, int Cout,
                                      sgc_stream_t stream) {
  return linear_rows(x, w_hi, w_lo, shift, y, rows_dev_or_null, rows_cap, Cin, Cout, 0, 0, 0, stream);
}

// ... with one extra all-zero row behind the result: y holds rows_cap + 1 rows and row rows_cap is set to zero by the SAME launch
// (the wave gather points out-of-image corners at that row, sgc_pairs_deform_gather's value_has_zero_row -- a separate fill was one
// more launch per level).  rows_cap > 0.
extern "C" int sgc_linear_rows_zrow_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift,
                                           float *y, const int32_t *rows_dev_or_null, int rows_cap, int Cin, int Cout,
                                           sgc_stream_t stream) {
  if (rows_cap <= 0 || !y) return set_error(SGC_EINVAL, "sgc_linear_rows_zrow_bf16x3: needs rows_cap > 0 and an output");
  return linear_rows(x, w_hi, w_lo, shift, y, rows_dev_or_null, rows_cap, Cin, Cout, 0, 0, 0, stream, y + (int64_t)rows_cap * Cout);
}

// The same GEMM with a HEAD-MAJOR result: x holds N * S rows (camera-major pixels), the Cout columns are M heads of
// Cm channels, and y is [N][M][S][Cm] -- the layout the LDS-tiled deformable gather stages one head's window from
// (dfa3d_tile.hip).  Same arithmetic per element as sgc_linear_rows_bf16x3: only the store address differs.
extern "C" int sgc_linear_rows_headmajor_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo,
                                                const float *shift, void *y, int y_bf16, int N, int S, int Cin, int M, int Cm,
                                                sgc_stream_t stream) {
  if (N <= 0 || S <= 0 || M <= 0 || Cm <= 0 || Cm % 4 || (int64_t)N * S >= (1ll << 31))
    return set_error(SGC_EINVAL, "sgc_linear_rows_headmajor_bf16x3: bad size (Cm %% 4 == 0 required)");
  // the head-major epilogues keep a head inside one column tile (128 columns on the tile kernel -- 64 when M * Cm <= 64 --,
  // 32 per wave on the persistent kernel, which also takes whole-wave multiples up to 128)
  const int tile_cols = M * Cm <= 64 ? 64 : 128;
  if (tile_cols % Cm)
    return set_error(SGC_EUNSUP, "sgc_linear_rows_headmajor_bf16x3: Cm must divide the %d-column tile (got %d)", tile_cols, Cm);
  return linear_rows(x, w_hi, w_lo, shift, reinterpret_cast<float *>(y), nullptr, N * S, Cin, M * Cm, S, Cm, y_bf16 ? 1 : 0, stream);
}

#if defined(SGC_HALO_STAMPS)
extern "C" void sgc_diag_halo_stamp_buffer(unsigned long long *buf) { sgc::g_halo_stamp_buf = buf; }
#endif

// Arithmetic mode of every bf16 MFMA kernel of the library (convolutions, Linears, the fused level tail): 3 = the
// fp32-faithful 3-way split (default), 1 = plain bf16 products.  Changes results (that is its purpose): not a tuning knob.
extern "C" int sgc_set_conv_products(int products) {
  if (products != 1 && products != 2 && products != 3)
    return set_error(SGC_EINVAL, "sgc_set_conv_products: 3 (bf16x3, fp32-faithful), 1 (one bf16 product) or 2 (one fp16 product)");
  sgc::g_conv_products = products;
  return SGC_OK;
}
extern "C" int sgc_get_conv_products(void) { return sgc::g_conv_products; }
