// The tail of a VoxFormer level in ONE launch (gfx950): for a tile of 32 voxels
//
//     x0 = out_proj(ctx)  on voxels some camera sees, 0 elsewhere     (nn.MultiheadAttention.out_proj + the reference's
//                                                                      slot scatter, TU/deformable_cross_attention.py:826-837)
//     x1 = LayerNorm_1(x0)                                             (VoxFormerLayer "norm", TU/encoder.py:311-338)
//     x2 = W2 relu(W1 x1 + b1) + b2 + x1                               (mmcv FFN, "ffn")
//     y  = LayerNorm_2(x2)                                             ("norm")
//
// i.e. what ran as six launches (row GEMM, scatter, LayerNorm, two 1x1x1 GEMMs, LayerNorm: ~80 us per level at
// config 2, each of them at its launch-latency floor) with the [Nq, C] / [Nq, 2C] intermediates going through HBM four
// times.  Here they stay in LDS:
//   * a workgroup (C / 32 waves) owns 32 voxels; every GEMM stage uses the persistent row GEMM's scheme (rows_gemm.hip):
//     a wave keeps the weights of its 32 output columns for a K chunk of C as MFMA B fragments in registers (loaded
//     from L2 once per stage), A fragments come from a bf16 hi / lo image in LDS, 3 bf16 products per multiply-add;
//   * stage outputs are written back to LDS in the form the next stage reads (fp32 rows for the LayerNorms and the
//     residual, hi / lo split images for the next GEMM's A operand);
//   * LayerNorm is the arithmetic of layer_norm_rows_kernel (common.hpp: ln_row_stats / ln_apply), one wave per row.
// Every product, sum order and epilogue expression is that of the unfused kernels, so the fused level is
// BIT-IDENTICAL to the six launches (tests/test_gpu_kernels.py checks it) -- it is a scheduling change, not a numerical one.
#include "common.hpp"
#include "mma.hpp"

namespace sgc {

int g_tune_level_tail = 1;       // 0: the six separate launches (round-2 path)


struct LevelTailParams {
  const float *ctx;              // [rows, C] compact rows of the voxels some camera sees (view_attend's output)
  const int32_t *row_of;         // [Nq]: compact row of voxel q, -1 = seen by no camera
  const __bf16 *wo_hi, *wo_lo;   // out_proj [C][C], fragment-packed (see load_B)
  const float *bo;
  const float *ln1_g, *ln1_b;
  const __bf16 *w1_hi, *w1_lo;   // FFN layer 1 [2C][C]
  const float *b1;
  const __bf16 *w2_hi, *w2_lo;   // FFN layer 2 [C][2C]
  const float *b2;
  const float *ln2_g, *ln2_b;
  float *out;                    // [Nq, C]
  float eps1, eps2;
  int Nq;
};

constexpr int LT_ROWS = 32;

template <int C, int NP = 3>   // NP: bf16 products per multiply-add (conv3d.hip: g_conv_products)
__global__ __launch_bounds__(C * 2, 2) void level_tail_kernel(const LevelTailParams p) {
  constexpr int NW = C / 32, NT = NW * 64, F = 2 * C;
  constexpr int KS = C / 16;                       // k-steps of a K chunk of C
  constexpr int K4 = C / 4, CH = LT_ROWS * K4 / NT;
  constexpr int PA = C + 8, PH = F + 8;            // bf16 row pitches of the two split images (16-byte pad: conflict-free fragments)
  constexpr int VPL = C / 64, RPW = LT_ROWS / NW;  // LayerNorm: floats per lane, rows per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char lt_smem[];
  __bf16 *A_hi = reinterpret_cast<__bf16 *>(lt_smem), *A_lo = A_hi + LT_ROWS * PA;           // x0-stage input, then x1
  float *X = reinterpret_cast<float *>(A_lo + LT_ROWS * PA);                                  // [32][C] fp32: x0, x1, x2
  __bf16 *H_hi = reinterpret_cast<__bf16 *>(X + LT_ROWS * C), *H_lo = H_hi + LT_ROWS * PH;    // relu(W1 x1 + b1)
  int *rowv = reinterpret_cast<int *>(H_lo + LT_ROWS * PH);                                   // [32] compact row or -1

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int col = wid * 32 + fr;                   // this lane's column in an N = C stage
  const int ntiles = (p.Nq + LT_ROWS - 1) / LT_ROWS;
  const int ld_row = tid / K4, ld_c4 = tid % K4;

  // Weights arrive FRAGMENT-PACKED (include/sgcdet_amd.h, sgc_level_tail): [N / 32][K / 16][64 lanes][8 bf16], i.e. the
  // 16 bytes lane l of the wave that owns columns 32 b .. 32 b + 31 feeds to the MFMA of k-step kk sit at
  // ((b * K/16 + kk) * 64 + l) * 8 -- one contiguous, fully coalesced 1 KiB access per wave and k-step.  (From a row-major
  // [N][K] matrix the same fragment is 32 rows x 32 bytes: every 128-byte line is touched by four different
  // instructions, and with all workgroups streaming the same 128 KB at the same time the level took 66 us instead of ~20.)
  // Buffer loads: ONE per-lane offset (lane * 16 bytes) for every fragment of the kernel, the (block, k-step) part of the
  // address in the scalar offset.  (With flat addresses only k-steps 0..3 fit the 12-bit immediate; the other 2 x 12
  // 64-bit addresses per stage were hoisted out of the tile loop: 292 spilled VGPRs, every load followed by vmcnt(0).)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  bf16x8 bh[KS], bl[KS];
  const int lane16 = lane * 16;
  auto load_B = [&](const __bf16 *w_hi, const __bf16 *w_lo, int n_total, int blk, int ksteps_total, int kk0) {
    const int bytes = n_total * ksteps_total * 32;                       // [n_total / 32][ksteps_total][64][8] bf16
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(w_hi), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(w_lo), 0, bytes, 0x00020000);
    const int soff = __builtin_amdgcn_readfirstlane((blk * ksteps_total + kk0) * 1024);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const u32x4 vh = __builtin_amdgcn_raw_buffer_load_b128(rh, lane16, soff + kk * 1024, 0);
      bh[kk] = __builtin_bit_cast(bf16x8, vh);
      if constexpr (NP == 3) {
        const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(rl, lane16, soff + kk * 1024, 0);
        bl[kk] = __builtin_bit_cast(bf16x8, vl);
      }
    }
  };
  constexpr int PD = 3;                            // A fragments read PD k-steps ahead (see rows_gemm.hip)
  auto multiply = [&](const __bf16 *hi, const __bf16 *lo, int pitch, int koff, f32x16 &acc) {
    const __bf16 *a_hi = hi + fr * pitch + koff + fh * 8, *a_lo = lo + fr * pitch + koff + fh * 8;
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
  auto zero = [&](f32x16 &acc) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  };
  // LayerNorm of this wave's RPW rows of X; result back into X and, split, into the A image; `to_global`: the level's output
  auto layer_norm = [&](const float *g, const float *b, float eps, bool to_global, int q0) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wid * RPW + rr;
      float v[VPL];
      ln_row_load<VPL>(X + row * C, lane, v);
      float mean, rstd;
      ln_row_stats<VPL>(v, eps, mean, rstd);
      float y[VPL];
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        const int c = ln_channel<VPL>(lane, j);
        y[j] = ln_apply(v[j], mean, rstd, g[c], b[c]);
      }
      if (to_global) {
        if (q0 + row < p.Nq) ln_row_store<VPL>(p.out + (int64_t)(q0 + row) * C, lane, y);
      } else {
        ln_row_store<VPL>(X + row * C, lane, y);
        bf16x4 h4, l4;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          const __bf16 hb = op_hi<NP>(y[j]);
          const __bf16 lb = op_lo<NP>(y[j], hb);
          if constexpr (VPL == 4) {
            h4[j] = hb; l4[j] = lb;
          } else {
            const int c = ln_channel<VPL>(lane, j);
            A_hi[row * PA + c] = hb;
            A_lo[row * PA + c] = lb;
          }
        }
        if constexpr (VPL == 4) {
          *reinterpret_cast<bf16x4 *>(A_hi + row * PA + lane * 4) = h4;
          *reinterpret_cast<bf16x4 *>(A_lo + row * PA + lane * 4) = l4;
        }
      }
    }
  };

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int q0 = t * LT_ROWS;
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage 0: out_proj weights on their way; gather + split the ctx rows of the tile ----
    load_B(p.wo_hi, p.wo_lo, C, wid, KS, 0);
    if (tid < LT_ROWS) rowv[tid] = q0 + tid < p.Nq ? p.row_of[q0 + tid] : -1;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int row = ld_row + i * (NT / K4);
      const int q = q0 + row;
      const int r = q < p.Nq ? p.row_of[q] : -1;
      const float4 v4 = r >= 0 ? *reinterpret_cast<const float4 *>(p.ctx + (int64_t)r * C + ld_c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = op_hi<NP>(v[e]);
        h[e] = hb;
        l[e] = op_lo<NP>(v[e], hb);
      }
      *reinterpret_cast<bf16x4 *>(A_hi + row * PA + ld_c4 * 4) = h;
      *reinterpret_cast<bf16x4 *>(A_lo + row * PA + ld_c4 * 4) = l;
    }
    __syncthreads();
    f32x16 acc;
    zero(acc);
    multiply(A_hi, A_lo, PA, 0, acc);
    {
      const float sh = p.bo[col];
      float *xb = X + 4 * fh * C + col;            // one per-lane base; the row part of every address is an immediate offset
      const int *rv = rowv + 4 * fh;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float v = acc[k] + sh;
        v = rv[(k & 3) + 8 * (k >> 2)] >= 0 ? v : 0.f;     // voxels no camera sees keep the zero row of the reference's slot scatter
        xb[((k & 3) + 8 * (k >> 2)) * C] = v;
      }
    }
    __syncthreads();
    // ---- x1 = LayerNorm_1(x0): fp32 into X (the FFN's residual), split into the A image ----
    __builtin_amdgcn_sched_barrier(0);
    load_B(p.w1_hi, p.w1_lo, F, wid, KS, 0);          // FFN layer 1, first half of its 2C columns (in flight under the LayerNorm)
    layer_norm(p.ln1_g, p.ln1_b, p.eps1, false, q0);
    __syncthreads();
    // ---- h = relu(W1 x1 + b1), two passes of C columns, split into the H image ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {
        __builtin_amdgcn_sched_barrier(0);           // not above the multiply that still reads the first half's fragments
        load_B(p.w1_hi, p.w1_lo, F, NW + wid, KS, 0);
      }
      zero(acc);
      multiply(A_hi, A_lo, PA, 0, acc);
      const int hc = pass * C + col;
      const float sh = p.b1[hc];
      __bf16 *hh = H_hi + 4 * fh * PH + hc, *hl = H_lo + 4 * fh * PH + hc;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float v = fmaxf(acc[k] + sh, 0.f);
        const __bf16 hb = op_hi<NP>(v);
        hh[((k & 3) + 8 * (k >> 2)) * PH] = hb;
        hl[((k & 3) + 8 * (k >> 2)) * PH] = op_lo<NP>(v, hb);
      }
    }
    __syncthreads();
    // ---- x2 = W2 h + b2 + x1: one accumulator chain over the two K chunks (k ascending, as the unfused K = 2C GEMM) ----
    zero(acc);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      __builtin_amdgcn_sched_barrier(0);             // one set of weight fragments lives at a time (128 VGPRs at C = 256)
      load_B(p.w2_hi, p.w2_lo, C, wid, 2 * KS, kc * KS);
      multiply(H_hi, H_lo, PH, kc * C, acc);
    }
    {
      const float sh = p.b2[col];
      float *xb = X + 4 * fh * C + col;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float v = acc[k] + sh;
        v += xb[((k & 3) + 8 * (k >> 2)) * C];
        xb[((k & 3) + 8 * (k >> 2)) * C] = v;
      }
    }
    __syncthreads();
    // ---- y = LayerNorm_2(x2) straight to the level's output rows ----
    layer_norm(p.ln2_g, p.ln2_b, p.eps2, true, q0);
    __syncthreads();                               // X / rowv are rewritten by the next tile
  }
}

extern int g_conv_products;      // conv3d.hip

template <int C, int NP>
static int launch_level_tail(const LevelTailParams &p, hipStream_t st) {
  constexpr int smem = 2 * LT_ROWS * (C + 8) * 2 + LT_ROWS * C * 4 + 2 * LT_ROWS * (2 * C + 8) * 2 + LT_ROWS * 4;
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void *)level_tail_kernel<C, NP>, smem, attr_done);
  const int ntiles = ceil_div(p.Nq, LT_ROWS);
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    static std::atomic<int> cached{0};
    if (cached.load() == 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cached.store(prop.multiProcessorCount);
    if (cached.load() > 0) cus = cached.load();
  }
  const int per_cu = C == 256 ? 1 : 2;
  const int grid = ntiles < cus * per_cu ? ntiles : cus * per_cu;
  hipLaunchKernelGGL((level_tail_kernel<C, NP>), dim3(grid), dim3(C * 2), smem, st, p);
  return check_launch("level_tail_kernel");
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_level_tail_supported(int C, int F) { return g_tune_level_tail && (C == 128 || C == 256) && F == 2 * C; }

extern "C" int sgc_level_tail(const float *ctx, const int32_t *row_of, const uint16_t *wo_hi, const uint16_t *wo_lo, const float *bo,
                              const float *ln1_gamma, const float *ln1_beta, float eps1, const uint16_t *w1_hi,
                              const uint16_t *w1_lo, const float *b1, const uint16_t *w2_hi, const uint16_t *w2_lo,
                              const float *b2, const float *ln2_gamma, const float *ln2_beta, float eps2, float *out, int Nq,
                              int C, int F, sgc_stream_t stream) {
  if (!ctx || !row_of || !wo_hi || !wo_lo || !bo || !ln1_gamma || !ln1_beta || !w1_hi || !w1_lo || !b1 || !w2_hi || !w2_lo ||
      !b2 || !ln2_gamma || !ln2_beta || !out)
    return set_error(SGC_EINVAL, "sgc_level_tail: null pointer");
  if (Nq <= 0) return SGC_OK;
  if (!((C == 128 || C == 256) && F == 2 * C))
    return set_error(SGC_EUNSUP, "sgc_level_tail: C in {128, 256} and F == 2 C (got C = %d, F = %d)", C, F);
  if (((uintptr_t)ctx | (uintptr_t)wo_hi | (uintptr_t)wo_lo | (uintptr_t)w1_hi | (uintptr_t)w1_lo | (uintptr_t)w2_hi | (uintptr_t)w2_lo |
       (uintptr_t)out) & 15)
    return set_error(SGC_EINVAL, "sgc_level_tail: pointers must be 16-byte aligned");
  LevelTailParams p = {};
  p.ctx = ctx; p.row_of = row_of;
  p.wo_hi = reinterpret_cast<const __bf16 *>(wo_hi); p.wo_lo = reinterpret_cast<const __bf16 *>(wo_lo); p.bo = bo;
  p.ln1_g = ln1_gamma; p.ln1_b = ln1_beta; p.eps1 = eps1;
  p.w1_hi = reinterpret_cast<const __bf16 *>(w1_hi); p.w1_lo = reinterpret_cast<const __bf16 *>(w1_lo); p.b1 = b1;
  p.w2_hi = reinterpret_cast<const __bf16 *>(w2_hi); p.w2_lo = reinterpret_cast<const __bf16 *>(w2_lo); p.b2 = b2;
  p.ln2_g = ln2_gamma; p.ln2_b = ln2_beta; p.eps2 = eps2;
  p.out = out; p.Nq = Nq;
  if (g_conv_products == 1)
    return C == 256 ? launch_level_tail<256, 1>(p, (hipStream_t)stream) : launch_level_tail<128, 1>(p, (hipStream_t)stream);
  if (g_conv_products == 2)
    return C == 256 ? launch_level_tail<256, 2>(p, (hipStream_t)stream) : launch_level_tail<128, 2>(p, (hipStream_t)stream);
  return C == 256 ? launch_level_tail<256, 3>(p, (hipStream_t)stream) : launch_level_tail<128, 3>(p, (hipStream_t)stream);
}
