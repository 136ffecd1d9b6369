// Weight layout passes of the TRAINING step, one launch each:
//
//   sgc_pack_conv_weight    module parameter [A][B][T] fp32 (nn.Conv3d: [Cout][Cin][k^3], nn.ConvTranspose3d: [Cin][Cout][8],
//                           nn.Linear: T = 1)  ->  the kernels' [T][R][C] layout, split into bf16 hi / lo planes, optionally
//                           transposed (R = B: the input-gradient pass multiplies by W^T), tap-mirrored (flip: the input
//                           gradient of a 3x3x3 convolution walks the taps backwards) and zero-padded (R, C up to multiples the
//                           kernels need).
//   sgc_unpack_conv_wgrad   the weight-gradient kernel's [T][R][C] fp32 result -> the parameter's [A][B][T] layout.
//   sgc_pack_conv_weight_batch   (round 5) the pack of MANY parameters in one launch: a training step packs ~100 parameters (two
//                           forms each: forward and input-gradient layout), most of them a few microseconds of work -- 96 launches
//                           were 1.2 ms of device time and 2.2 ms of wall per config-2 step (the host could not keep the GPU fed:
//                           tools/trace_gaps.py); the batch moves the same bytes in one launch at the rate of the large layers.
// (Summing the weight gradient's split slabs inside the unpack pass -- one launch and one HBM round trip of the gradient less per
//  layer -- was built, bit-identical, and is slower: the unpack kernel holds a 110-KB block per CU and reads each slab at a fraction of
//  the rate of the streaming reduce kernel it replaces: 0.72 + 0.39 ms against 0.31 + 0.51 ms per config-2 step.)
//
// In round 2 these were torch ops per layer and per pass (permute + contiguous: an uncoalesced strided copy of up to 113 MB;
// then to(bfloat16), subtract, to(bfloat16) for the split): 3.5 ms of strided copies + 1.4 ms of conversion kernels per
// config-2 step (profiles/r02_train_step_kernels.txt).  Here a workgroup moves a 32 x 32 x T block through LDS: the reads
// are contiguous runs of 32 T floats, the writes 64-byte row segments, the split happens on the way.
// Reference: the passes exist only because cuDNN takes the parameter layout directly (necks/imvoxelnet.py:36-64).
#include "common.hpp"

namespace sgc {

struct PackParams {
  const float *w;        // [A][B][T]
  __bf16 *hi, *lo;       // [T][R][C] (pack)
  float *out;            // [A][B][T] (unpack: w is then the [T][R][C] source)
  int A, B, T;
  int R, C;              // padded output extents
  int transpose, flip;
};

constexpr int PK = 32;

// Block of the pack kernel: PA x PB x T parameter elements, PB * T of them contiguous in the source.  The OUTPUT is contiguous
// along c, and a block has to carry 32 of them (64-byte segments per plane): c = b in the plain layout -> 8 a x 32 b; c = a in
// the transposed one -> 32 a x 8 b.  27.6 KB of LDS at 27 taps: five workgroups per CU (the first version moved 32 x 32 x T
// blocks, 110 KB and one 8-wave workgroup per CU, through one barrier: 0.5 TB/s, 2.4 ms per training step).
template <int PA, int PB>
__device__ __forceinline__ void pack_block(const PackParams &p, int bx, int by) {
  extern __shared__ float pk_lds[];                 // [PA a][PB b][T] (+1 pad per a-row)
  constexpr bool TR = PA == 32;                     // the transposed layout (c = a)
  const int T = p.T, run = PB * T, pitch = run + 1;
  const int a0 = by * PA, b0 = bx * PB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // load: for every a of the block the run [b0, b0 + PB) x T is contiguous in w; a wave takes whole runs
  const int nvalid_b = (p.B - b0 < PB ? p.B - b0 : PB) * T;
  if ((((int64_t)p.B * T) & 3) == 0 && ((uintptr_t)p.w & 15) == 0) {
    // 16-byte loads, four in flight per thread (round 5: with 4-byte loads one wave kept 256 bytes in flight per instruction and the
    // batch kernel moved 2 TB/s); run = PB * T is a multiple of 8, a row starts on a 16-byte boundary
    const int r4 = run >> 2, total4 = PA * r4;
    for (int base = tid; base < total4; base += 256 * 4) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = base + 256 * k;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < total4) {
          const int a = e / r4, i = (e - a * r4) * 4;
          const int nvalid = a0 + a < p.A ? nvalid_b : 0;
          const float *src = p.w + ((int64_t)(a0 + a) * p.B + b0) * T + i;
          if (i + 3 < nvalid) v[k] = *reinterpret_cast<const float4 *>(src);
          else if (i < nvalid) { v[k].x = src[0]; if (i + 1 < nvalid) v[k].y = src[1]; if (i + 2 < nvalid) v[k].z = src[2]; }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = base + 256 * k;
        if (e < total4) {
          const int a = e / r4, i = (e - a * r4) * 4;
          float *d = pk_lds + a * pitch + i;
          d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
        }
      }
    }
  } else {
    for (int a = wid; a < PA; a += 4) {
      const int nvalid = a0 + a < p.A ? nvalid_b : 0;
      const float *wrow = p.w + ((int64_t)(a0 + a) * p.B + b0) * T;
      for (int i = lane; i < run; i += 64) pk_lds[a * pitch + i] = i < nvalid ? wrow[i] : 0.f;
    }
  }
  __syncthreads();
  // store: units of 8 consecutive c of one (t, r): 16-byte hi and lo stores, the 4 units of a 32-column row segment on 4 lanes
  const int units = T * 8 * 4;
  const bool vec = (p.C & 7) == 0;
  for (int u = tid; u < units; u += 256) {
    const int c8 = u & 3, r = (u >> 2) & 7, t = u >> 5;
    const int gr = (TR ? b0 : a0) + r, gc = (TR ? a0 : b0) + c8 * 8;
    if (gr >= p.R || gc >= p.C) continue;
    const int ts = p.flip ? T - 1 - t : t;
    const float *src = TR ? pk_lds + (c8 * 8) * pitch + r * T + ts : pk_lds + r * pitch + (c8 * 8) * T + ts;
    const int step = TR ? pitch : T;
    __bf16 h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = src[j * step];                     // zero outside [A) x [B): the padding rows / columns
      h[j] = (__bf16)v;
      l[j] = (__bf16)(v - (float)h[j]);
    }
    const int64_t o = ((int64_t)t * p.R + gr) * p.C + gc;
    if (vec) {
      *reinterpret_cast<uint4 *>(p.hi + o) = *reinterpret_cast<const uint4 *>(h);
      *reinterpret_cast<uint4 *>(p.lo + o) = *reinterpret_cast<const uint4 *>(l);
    } else {
      for (int j = 0; j < 8 && gc + j < p.C; ++j) { p.hi[o + j] = h[j]; p.lo[o + j] = l[j]; }
    }
  }
}

template <int PA, int PB>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const PackParams p) { pack_block<PA, PB>(p, blockIdx.x, blockIdx.y); }

// One launch for a list of parameters (sgc_pack_conv_weight_batch): block -> item by binary search over the items' first blocks,
// then the single-parameter block above.  The item list lives in device memory (the host side builds it once per model).
struct PackItem {                 // = sgc_pack_item of include/sgcdet_amd.h (64 bytes)
  const float *w;
  uint16_t *hi, *lo;
  int A, B, T, R, C, transpose, flip, block_start;
  int pad_[2];
};
static_assert(sizeof(PackItem) == 64, "sgc_pack_item is 64 bytes");
__device__ __host__ inline int pack_blocks_x(int B, int transpose) { return transpose ? (B + 7) / 8 : (B + 31) / 32; }
__device__ __host__ inline int pack_blocks_y(int A, int transpose) { return transpose ? (A + 31) / 32 : (A + 7) / 8; }

__global__ __launch_bounds__(256) void pack_conv_weight_batch_kernel(const PackItem *__restrict__ items, int n) {
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;                           // last item with block_start <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_start <= b) lo = mid; else hi = mid - 1;
  }
  const PackItem it = items[lo];
  PackParams p = {};
  p.w = it.w; p.hi = reinterpret_cast<__bf16 *>(it.hi); p.lo = reinterpret_cast<__bf16 *>(it.lo);
  p.A = it.A; p.B = it.B; p.T = it.T; p.R = it.R; p.C = it.C; p.transpose = it.transpose; p.flip = it.flip;
  const int local = b - it.block_start, gx = pack_blocks_x(it.B, it.transpose);
  if (local >= gx * pack_blocks_y(it.A, it.transpose)) return;
  if (it.transpose) pack_block<32, 8>(p, local % gx, local / gx);
  else pack_block<8, 32>(p, local % gx, local / gx);
}

__global__ __launch_bounds__(512) void unpack_conv_wgrad_kernel(const PackParams p) {
  extern __shared__ float pk_lds[];                 // [PK a][PK b][T]
  const int T = p.T, pitch = PK * T + 1;
  const int a0 = blockIdx.y * PK, b0 = blockIdx.x * PK;
  const int tid = threadIdx.x;
  const int units = T * PK * 8;                     // 4 consecutive c (16 bytes) of one (t, r)
  const bool vec = (p.C & 3) == 0;
  for (int u = tid; u < units; u += 512) {
    const int c4 = u & 7, r = (u >> 3) & (PK - 1), t = u >> 8;
    const int gr = (p.transpose ? b0 : a0) + r, gc = (p.transpose ? a0 : b0) + c4 * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gr < p.R && gc < p.C) {
      const float *s = p.w + ((int64_t)t * p.R + gr) * p.C + gc;
      if (vec) {
        const float4 q = *reinterpret_cast<const float4 *>(s);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
        for (int j = 0; j < 4 && gc + j < p.C; ++j) v[j] = s[j];
      }
    }
    const int ts = p.flip ? T - 1 - t : t;
    float *dst = p.transpose ? pk_lds + (c4 * 4) * pitch + r * T + ts : pk_lds + r * pitch + (c4 * 4) * T + ts;
    const int step = p.transpose ? pitch : T;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j * step] = v[j];
  }
  __syncthreads();
  const int run = PK * T;
  const int nvalid_b = (p.B - b0 < PK ? p.B - b0 : PK) * T;
  for (int a = 0; a < PK && a0 + a < p.A; ++a) {
    float *orow = p.out + ((int64_t)(a0 + a) * p.B + b0) * T;
    for (int i = tid; i < run; i += 512)
      if (i < nvalid_b) orow[i] = pk_lds[a * pitch + i];
  }
}

static int pack_launch(bool unpack, const PackParams &p, hipStream_t st) {
  if (!unpack) {
    const size_t smem = (size_t)8 * 32 * p.T * sizeof(float) + 32 * sizeof(float);       // [PA][PB * T + 1], PA * PB = 256
    if (smem > 64 * 1024) return set_error(SGC_EUNSUP, "weight pack: %d taps do not fit the block in LDS", p.T);
    if (p.transpose) {
      hipLaunchKernelGGL((pack_conv_weight_kernel<32, 8>), dim3(ceil_div(p.B, 8), ceil_div(p.A, 32)), dim3(256), smem, st, p);
    } else {
      hipLaunchKernelGGL((pack_conv_weight_kernel<8, 32>), dim3(ceil_div(p.B, 32), ceil_div(p.A, 8)), dim3(256), smem, st, p);
    }
    return check_launch("pack_conv_weight_kernel");
  }
  const size_t smem = (size_t)PK * (PK * p.T + 1) * sizeof(float);
  if (smem > 160 * 1024) return set_error(SGC_EUNSUP, "weight pack: %d taps do not fit the 32 x 32 block in LDS", p.T);
  static std::atomic<uint64_t> done_b{0};
  ensure_dynamic_lds((const void *)unpack_conv_wgrad_kernel, 160 * 1024, done_b);
  const dim3 grid(ceil_div(p.B, PK), ceil_div(p.A, PK));
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, grid, dim3(512), smem, st, p);
  return check_launch("unpack_conv_wgrad_kernel");
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_pack_conv_weight(const float *w, uint16_t *w_hi, uint16_t *w_lo, int A, int B, int T, int R, int C,
                                    int transpose, int flip, sgc_stream_t stream) {
  if (!w || !w_hi || !w_lo) return set_error(SGC_EINVAL, "sgc_pack_conv_weight: null pointer");
  if (A <= 0 || B <= 0 || T <= 0 || R < (transpose ? B : A) || C < (transpose ? A : B))
    return set_error(SGC_EINVAL, "sgc_pack_conv_weight: bad size (R x C must cover the %s matrix)", transpose ? "transposed" : "");
  PackParams p = {};
  p.w = w; p.hi = reinterpret_cast<__bf16 *>(w_hi); p.lo = reinterpret_cast<__bf16 *>(w_lo);
  p.A = A; p.B = B; p.T = T; p.R = R; p.C = C; p.transpose = transpose ? 1 : 0; p.flip = flip ? 1 : 0;
  // the grid covers [A) x [B); padding rows / columns beyond it are zero-filled here
  const int rows = transpose ? B : A, cols = transpose ? A : B;
  hipStream_t st = (hipStream_t)stream;
  if (R > rows || C > cols) {
    const size_t bytes = (size_t)T * R * C * sizeof(uint16_t);
    if (hipMemsetAsync(w_hi, 0, bytes, st) != hipSuccess || hipMemsetAsync(w_lo, 0, bytes, st) != hipSuccess)
      return set_error(SGC_ELAUNCH, "sgc_pack_conv_weight: memset failed");
  }
  return pack_launch(false, p, st);
}

extern "C" int sgc_pack_conv_weight_blocks(int A, int B, int T, int transpose) {
  if (A <= 0 || B <= 0 || T <= 0) return 0;
  return pack_blocks_x(B, transpose) * pack_blocks_y(A, transpose);
}

// items_dev: n_items descriptors in DEVICE memory, block_start ascending from 0 with item i owning sgc_pack_conv_weight_blocks()
// blocks; total_blocks = their sum; max_T = the largest tap count among them (sizes the LDS block).  The padding rows / columns of
// the planes (R, C beyond the matrix) are NOT written: the caller allocates the planes zeroed, once.
extern "C" int sgc_pack_conv_weight_batch(const void *items_dev, int n_items, int total_blocks, int max_T, sgc_stream_t stream) {
  if (!items_dev) return set_error(SGC_EINVAL, "sgc_pack_conv_weight_batch: null pointer");
  if (n_items <= 0 || total_blocks <= 0 || max_T <= 0) return set_error(SGC_EINVAL, "sgc_pack_conv_weight_batch: bad size");
  const size_t smem = (size_t)8 * 32 * max_T * sizeof(float) + 32 * sizeof(float);
  if (smem > 64 * 1024) return set_error(SGC_EUNSUP, "sgc_pack_conv_weight_batch: %d taps do not fit the block in LDS", max_T);
  hipLaunchKernelGGL(pack_conv_weight_batch_kernel, dim3(total_blocks), dim3(256), smem, (hipStream_t)stream,
                     reinterpret_cast<const PackItem *>(items_dev), n_items);
  return check_launch("pack_conv_weight_batch_kernel");
}

extern "C" int sgc_unpack_conv_wgrad(const float *dw_trc, float *dw, int A, int B, int T, int R, int C, int transpose, int flip,
                                     sgc_stream_t stream) {
  if (!dw_trc || !dw) return set_error(SGC_EINVAL, "sgc_unpack_conv_wgrad: null pointer");
  if (A <= 0 || B <= 0 || T <= 0 || R < (transpose ? B : A) || C < (transpose ? A : B))
    return set_error(SGC_EINVAL, "sgc_unpack_conv_wgrad: bad size");
  PackParams p = {};
  p.w = dw_trc; p.out = dw;
  p.A = A; p.B = B; p.T = T; p.R = R; p.C = C; p.transpose = transpose ? 1 : 0; p.flip = flip ? 1 : 0;
  return pack_launch(true, p, (hipStream_t)stream);
}
