// Weight layout passes of the TRAINING step, one launch each:
//
//   sgc_pack_conv_weight    module parameter [A][B][T] fp32 (nn.Conv3d: [Cout][Cin][k^3], nn.ConvTranspose3d: [Cin][Cout][8],
//                           nn.Linear: T = 1)  ->  the kernels' [T][R][C] layout, split into bf16 hi / lo planes, optionally
//                           transposed (R = B: the input-gradient pass multiplies by W^T), tap-mirrored (flip: the input
//                           gradient of a 3x3x3 convolution walks the taps backwards) and zero-padded (R, C up to multiples the
//                           kernels need).
//   sgc_unpack_conv_wgrad   the weight-gradient kernel's [T][R][C] fp32 result -> the parameter's [A][B][T] layout.
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

__global__ __launch_bounds__(512) void pack_conv_weight_kernel(const PackParams p) {
  extern __shared__ float pk_lds[];                 // [PK a][PK b][T] (+1 pad per a-row)
  const int T = p.T, pitch = PK * T + 1;
  const int a0 = blockIdx.y * PK, b0 = blockIdx.x * PK;
  const int tid = threadIdx.x;
  // load: for every a of the tile the run [b0, b0 + PK) x T is contiguous in w (4 bytes per lane, 256-byte wave accesses)
  const int run = PK * T;
  const int nvalid_b = (p.B - b0 < PK ? p.B - b0 : PK) * T;
  for (int a = 0; a < PK; ++a) {
    const int nvalid = a0 + a < p.A ? nvalid_b : 0;
    const float *wrow = p.w + ((int64_t)(a0 + a) * p.B + b0) * T;
    for (int i = tid; i < run; i += 512) pk_lds[a * pitch + i] = i < nvalid ? wrow[i] : 0.f;
  }
  __syncthreads();
  // store: units of 8 consecutive c of one (t, r): 16-byte hi and lo stores, a 32-column row segment = 4 lanes
  const int units = T * PK * 4;
  const bool vec = (p.C & 7) == 0;
  for (int u = tid; u < units; u += 512) {
    const int c8 = u & 3, r = (u >> 2) & (PK - 1), t = u >> 7;
    const int gr = (p.transpose ? b0 : a0) + r, gc = (p.transpose ? a0 : b0) + c8 * 8;
    if (gr >= p.R || gc >= p.C) continue;
    const int ts = p.flip ? T - 1 - t : t;
    const float *src = p.transpose ? pk_lds + (c8 * 8) * pitch + r * T + ts : pk_lds + r * pitch + (c8 * 8) * T + ts;
    const int step = p.transpose ? pitch : T;
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
  const size_t smem = (size_t)PK * (PK * p.T + 1) * sizeof(float);
  if (smem > 160 * 1024) return set_error(SGC_EUNSUP, "weight pack: %d taps do not fit the 32 x 32 block in LDS", p.T);
  static std::atomic<uint64_t> done_a{0}, done_b{0};
  ensure_dynamic_lds((const void *)pack_conv_weight_kernel, 160 * 1024, done_a);
  ensure_dynamic_lds((const void *)unpack_conv_wgrad_kernel, 160 * 1024, done_b);
  const dim3 grid(ceil_div(p.B, PK), ceil_div(p.A, PK));
  if (unpack) hipLaunchKernelGGL(unpack_conv_wgrad_kernel, grid, dim3(512), smem, st, p);
  else hipLaunchKernelGGL(pack_conv_weight_kernel, grid, dim3(512), smem, st, p);
  return check_launch(unpack ? "unpack_conv_wgrad_kernel" : "pack_conv_weight_kernel");
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
