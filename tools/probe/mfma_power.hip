// What the matrix pipe sustains under the package power limit: every SIMD issues v_mfma_f32_32x32x16_bf16 back to back on
// register-resident operands (random bf16 values, or zeros: argv[1] = 0) for a few seconds; prints TFLOP/s per interval.
// Run it next to `rocm-smi --showclocks --showpower` (tools/jobs/r04_mfma_power.sh).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ORDER>   // 0: A and B alternate between consecutive MFMAs; 1: the same A and B for every MFMA; 2: A stationary, B alternates
__global__ __launch_bounds__(512) void mfma_loop(const bf16x8 *in, float *out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a0 = in[4 * (t & 4095)], a1 = in[4 * (t & 4095) + 1], b0 = in[4 * (t & 4095) + 2], b1 = in[4 * (t & 4095) + 3];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ORDER == 0 ? a1 : a0, b0, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, ORDER == 1 ? b0 : b1, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ORDER == 0 ? a1 : a0, ORDER == 1 ? b0 : b1, c3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k] + (float)a1[0] + (float)b1[0];
  out[t] = s;
}

int main(int argc, char **argv) {
  const int random_data = argc > 1 ? atoi(argv[1]) : 1;
  const float seconds = argc > 2 ? atof(argv[2]) : 4.f;
  const int waves_per_simd = argc > 3 ? atoi(argv[3]) : 2;
  const int order = argc > 4 ? atoi(argv[4]) : 0;
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 4096;
  std::vector<unsigned short> h(4096 * 4 * 8);
  srand(1);
  for (auto &v : h) {
    float f = random_data ? ((rand() % 2001) - 1000) * 1e-3f : 0.f;
    unsigned u; std::memcpy(&u, &f, 4);
    v = (unsigned short)(u >> 16);
  }
  bf16x8 *din; float *dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, (size_t)blocks * threads * 4);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const double flop = (double)blocks * (threads / 64) * iters * 32.0 * 2.0 * 32 * 32 * 16;
  auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    auto a = std::chrono::steady_clock::now();
    for (int r = 0; r < 8; ++r) {
      if (order == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
      else if (order == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
      else hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
    }
    hipDeviceSynchronize();
    auto b = std::chrono::steady_clock::now();
    const double dt = std::chrono::duration<double>(b - a).count();
    printf("%.2f s: %.1f TFLOP/s\n", std::chrono::duration<double>(b - t0).count(), 8 * flop / dt / 1e12);
    fflush(stdout);
    if (std::chrono::duration<double>(b - t0).count() > seconds) break;
  }
  return 0;
}
