// Block-diagonal Linear over a row list (round 6):
//     y[r][g * NH + j] = sum_k x[r][g * K + k] * w[g][j][k] + shift[g * NH + j],     g < 8,  NH = K / 8,  K in {128, 256}
// -- the per-voxel V projection of the projected-query attention (sgc_view_attend_pq leaves the attention-weighted raw feature of
// every head, [voxels][8][C]; head h multiplies it with ITS rows of nn.MultiheadAttention's in_proj_weight:
// TU/deformable_cross_attention.py:826-833).  Rounds 5 ran it as a dense [8 C -> C] GEMM with 7/8 zero blocks on the tile kernel,
// whose time is its operand traffic (128 x 128 x 32 steps over K = 8 C): 135 us for 73 600 x 1024 -> 128 at config 5, 2.2 TB/s of x.
// Here x is read ONCE at streaming rate and that is all the kernel does to HBM: 8 waves = 8 groups; a wave holds the bf16 hi / lo
// fragments of its group's [NH x K] weights in registers for the whole kernel (the weight-stationary idea of rows_gemm.hip) and
// streams its own K-slice of 32 rows at a time: coalesced 512-byte row pieces -> registers -> hi | lo split -> a PRIVATE LDS image
// (no workgroup barrier anywhere: a wave's LDS operations execute in order) -> MFMA fragments; the next chunk's loads are in flight
// while this one multiplies.  Same products (lo*hi + hi*lo + hi*hi) and the same K order as the dense GEMM over the group's K range:
// the zero blocks of the dense form add exact zeros, so the two agree bit for bit.
#include "common.hpp"
#include "mma.hpp"

namespace sgc {

extern int g_conv_products;
int device_cus();

struct BdParams {
  const float *x;              // [rows_cap][8 K]
  const __bf16 *w_hi, *w_lo;   // [8][NH][K]
  const float *shift;          // [8 NH] or null
  float *y;                    // [rows_cap][8 NH]
  const int32_t *rows_dev;     // live rows on the device, or null (= rows_cap)
  int rows_cap;
};

constexpr int BD_KC = 128;            // channels of a staged chunk: 32 rows x 512 B per wave
constexpr int BD_PITCH = BD_KC + 8;   // bf16 elements per LDS row (272 B: rows 4 banks apart, ds_read_b128 of 16 rows conflict-free)

template <int K, int NP>
__global__ __launch_bounds__(512) void rows_blockdiag_kernel(const BdParams p) {
  constexpr int NH = K / 8, NKK = K / 16, NPH = K / BD_KC, XS = 8 * K, YS = 8 * NH;
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem[];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  __bf16 *a_hi = reinterpret_cast<__bf16 *>(bd_smem) + g * (2 * 32 * BD_PITCH), *a_lo = a_hi + 32 * BD_PITCH;
  const int rows = p.rows_dev ? min(p.rows_cap, *p.rows_dev) : p.rows_cap;
  const int ntiles = (rows + 31) >> 5;
  if ((int)blockIdx.x >= ntiles) return;
  const int col = lane & 31, half = lane >> 5;

  // this group's weights as B fragments: column `col` (zero past NH), k = 16 kk + 8 half .. + 8
  bf16x8 bh[NKK], bl[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    const size_t o = ((size_t)(g * NH + (col < NH ? col : 0))) * K + kk * 16 + half * 8;
    const bf16x8 z = (bf16x8)__builtin_bit_cast(__bf16, (unsigned short)0);
    bh[kk] = col < NH ? *reinterpret_cast<const bf16x8 *>(p.w_hi + o) : z;
    if constexpr (NP == 3) bl[kk] = col < NH ? *reinterpret_cast<const bf16x8 *>(p.w_lo + o) : z;
  }
  const float bias = (p.shift && col < NH) ? p.shift[g * NH + col] : 0.f;

  // chunk c = (tile, phase): rows 32 tile .. + 32, channels g K + 128 phase .. + 128.  Lane l stages row 2 i + (l >> 5), floats 4 (l & 31)
  float4 ra[16];
  auto load_chunk = [&](int tile, int ph) {
    const float *base = p.x + (size_t)(tile * 32 + half) * XS + g * K + ph * BD_KC + col * 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = tile * 32 + 2 * i + half;
      ra[i] = row < rows ? *reinterpret_cast<const float4 *>(base + (size_t)(2 * i) * XS) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = op_hi<NP>(v[e]);
        h[e] = hb;
        l[e] = op_lo<NP>(v[e], hb);
      }
      const int o = (2 * i + half) * BD_PITCH + col * 4;
      *reinterpret_cast<bf16x4 *>(a_hi + o) = h;
      if constexpr (NP == 3) *reinterpret_cast<bf16x4 *>(a_lo + o) = l;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  int tile = blockIdx.x;
  load_chunk(tile, 0);
  while (tile < ntiles) {
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      store_chunk();
      // the next chunk's loads fly while this one multiplies
      const int ntile = ph + 1 < NPH ? tile : tile + (int)gridDim.x;
      if (ntile < ntiles) load_chunk(ntile, ph + 1 < NPH ? ph + 1 : 0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const __bf16 *fa = a_hi + col * BD_PITCH + half * 8;
#pragma unroll
      for (int kk = 0; kk < BD_KC / 16; ++kk) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(fa + kk * 16);
        if constexpr (NP == 3) {
          const bf16x8 al = *reinterpret_cast<const bf16x8 *>(fa + 32 * BD_PITCH + kk * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ph * (BD_KC / 16) + kk], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ph * (BD_KC / 16) + kk], acc, 0, 0, 0);
        }
        acc = mma_hh<NP>(ah, bh[ph * (BD_KC / 16) + kk], acc);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // accumulator layout of the 32 x 32 tile: lane = column, element k = row (k & 3) + 8 (k >> 2) + 4 (lane >> 5)
    if (col < NH) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = tile * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
        if (row < rows) p.y[(size_t)row * YS + g * NH + col] = acc[k] + bias;
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    tile += gridDim.x;
  }
}

template <int K>
static int launch_blockdiag(const BdParams &p, hipStream_t st) {
  const size_t smem = (size_t)8 * 2 * 32 * BD_PITCH * sizeof(uint16_t);      // 136 KB: one workgroup per CU
  const int tiles = ceil_div(p.rows_cap, 32);
  const int grid = tiles < device_cus() ? tiles : device_cus();
  static std::atomic<uint64_t> done[3] = {};
  if (g_conv_products == 1) {
    ensure_dynamic_lds((const void *)rows_blockdiag_kernel<K, 1>, (int)smem, done[0]);
    hipLaunchKernelGGL((rows_blockdiag_kernel<K, 1>), dim3(grid), dim3(512), smem, st, p);
  } else if (g_conv_products == 2) {
    ensure_dynamic_lds((const void *)rows_blockdiag_kernel<K, 2>, (int)smem, done[1]);
    hipLaunchKernelGGL((rows_blockdiag_kernel<K, 2>), dim3(grid), dim3(512), smem, st, p);
  } else {
    ensure_dynamic_lds((const void *)rows_blockdiag_kernel<K, 3>, (int)smem, done[2]);
    hipLaunchKernelGGL((rows_blockdiag_kernel<K, 3>), dim3(grid), dim3(512), smem, st, p);
  }
  return check_launch("rows_blockdiag_kernel");
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_linear_rows_blockdiag_supported(int G, int K, int Nh) {
  return G == 8 && (K == 128 || K == 256) && Nh * 8 == K;
}

extern "C" int sgc_linear_rows_blockdiag_bf16x3(const float *x, const uint16_t *w_hi, const uint16_t *w_lo, const float *shift_or_null,
                                                float *y, const int32_t *rows_dev_or_null, int rows_cap, int G, int K, int Nh,
                                                sgc_stream_t stream) {
  if (!x || !w_hi || !w_lo || !y) return set_error(SGC_EINVAL, "sgc_linear_rows_blockdiag_bf16x3: null pointer");
  if (rows_cap <= 0) return SGC_OK;
  if (!sgc_linear_rows_blockdiag_supported(G, K, Nh))
    return set_error(SGC_EUNSUP, "sgc_linear_rows_blockdiag_bf16x3: 8 groups of K in {128, 256} inputs and K / 8 outputs (got %d x %d -> %d)", G, K, Nh);
  if (((uintptr_t)x | (uintptr_t)w_hi | (uintptr_t)w_lo | (uintptr_t)y) & 15)
    return set_error(SGC_EINVAL, "sgc_linear_rows_blockdiag_bf16x3: pointers must be 16-byte aligned");
  if ((int64_t)rows_cap * 8 * K * 4 >= ((int64_t)1 << 40)) return set_error(SGC_EUNSUP, "sgc_linear_rows_blockdiag_bf16x3: too many rows");
  BdParams p = {};
  p.x = x; p.w_hi = reinterpret_cast<const __bf16 *>(w_hi); p.w_lo = reinterpret_cast<const __bf16 *>(w_lo);
  p.shift = shift_or_null; p.y = y; p.rows_dev = rows_dev_or_null; p.rows_cap = rows_cap;
  return K == 128 ? launch_blockdiag<128>(p, (hipStream_t)stream) : launch_blockdiag<256>(p, (hipStream_t)stream);
}
