// Minimal reproducer: packed-FP32 VALU results of one kernel while another kernel's waves run
// MFMA instructions on the same SIMDs (two streams).  Build: hipcc --offload-arch=gfx950 -O2 -o pk_mfma_repro pk_mfma_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <dlfcn.h>
typedef void *sgc_stream_t;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// victim: out = xy * (H, W) - 0.5 with the operand swizzle the gather kernel's compiler output uses
template <int MODE>
__global__ __launch_bounds__(256) void victim(const float *xs, unsigned *bad_per_lane, int iters, int H, int W) {
  const int lane = threadIdx.x & 63;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float x = xs[gid & 1023], y = xs[(gid + 17) & 1023];
  float hf, wf;
  asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(hf) : "s"(H));
  asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(wf) : "s"(W));
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 xy = {x, y}, hw = {hf, wf}, r;
    float ex, ey;
    if (MODE == 0) {        // packed fma with op_sel (lo result uses hw.y = W, hi result uses hw.x = H)
      asm volatile("v_pk_fma_f32 %0, %1, %2, -0.5 op_sel:[0,1,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(xy), "v"(hw));
    } else if (MODE == 1) { // packed fma, plain operand order (lo: x*H, hi: y*W)
      asm volatile("v_pk_fma_f32 %0, %1, %2, -0.5 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(xy), "v"(hw));
    } else {                // packed mul
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(xy), "v"(hw));
    }
    if (MODE == 0) {
      asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(ex) : "v"(x), "v"(wf));
      asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(ey) : "v"(y), "v"(hf));
    } else if (MODE == 1) {
      asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(ex) : "v"(x), "v"(hf));
      asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(ey) : "v"(y), "v"(wf));
    } else {
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ex) : "v"(x), "v"(hf));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ey) : "v"(y), "v"(wf));
    }
    if (__float_as_uint(r.x) != __float_as_uint(ex) || __float_as_uint(r.y) != __float_as_uint(ey)) ++bad;
    x = x * 0.999f + 0.0007f; y = y * 0.998f + 0.0011f;
  }
  if (bad) atomicAdd(&bad_per_lane[lane], bad);
}

// aggressors
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float *sink, int iters) {
  const int lane = threadIdx.x & 63;
  if (KIND == 0) {          // bf16 MFMA 32x32x16 (gfx950)
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
    f32x16 acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; ++it) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 123.456f) sink[0] = s;
  } else if (KIND == 1) {   // fp32 MFMA 32x32x2
    f32x16 acc0 = {0}, acc1 = {0};
    float a = 0.01f * lane, b = 0.02f * lane;
    for (int it = 0; it < iters; ++it) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 123.456f) sink[0] = s;
  } else {                  // plain VALU
    float s = lane;
    for (int it = 0; it < iters * 16; ++it) s = s * 1.0001f + 0.5f;
    if (s == 123.456f) sink[0] = s;
  }
}

static float *g_x, *g_y; static void *g_wh, *g_wl;
typedef int (*conv_fn)(const float *, const uint16_t *, const uint16_t *, const float *, const float *, const float *, float *,
                       int, int, int, int, int, int, int, int, int, float *, int64_t, sgc_stream_t);
static conv_fn sgc_conv3d_cl_bf16x3 = nullptr;
static const char *(*sgc_last_error)() = nullptr;
static void launch_igemm(hipStream_t st) {
  // 25600 x 256 -> 256, 1x1x1: the bf16x3 implicit-GEMM kernel (8 waves, 80 KiB LDS, v_mfma_f32_32x32x16_bf16)
  int rc = sgc_conv3d_cl_bf16x3(g_x, (const uint16_t *)g_wh, (const uint16_t *)g_wl, nullptr, nullptr, nullptr, g_y, 25600, 1, 1, 256, 256, 1, 1, 0, 0,
                                nullptr, 0, (sgc_stream_t)st);
  if (rc) { printf("conv rc %d %s\n", rc, sgc_last_error()); exit(1); }
}

// more aggressor pieces of the implicit-GEMM kernel, one at a time
template <int KIND, int NT>
__global__ __launch_bounds__(NT) void aggressor2(float *sink, const float *src, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  if (KIND == 5) {          // fp32 -> bf16 hi/lo split (v_cvt_pk_bf16_f32, v_sub_f32, v_pk_add_f32 ...)
    float v[4] = {src[tid & 255], src[(tid + 1) & 255], src[(tid + 2) & 255], src[(tid + 3) & 255]};
    float s = 0.f;
    for (int it = 0; it < iters * 8; ++it) {
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = (__bf16)v[e];
        const __bf16 lb = (__bf16)(v[e] - (float)hb);
        s += (float)lb; v[e] = v[e] * 1.0001f + (float)hb * 1e-6f;
      }
    }
    if (s == 123.456f) sink[0] = s;
  } else if (KIND == 6) {   // LDS traffic only: b64 writes + b128 reads over 80 KiB
    uint2 *w = reinterpret_cast<uint2 *>(smem);
    const uint4 *r = reinterpret_cast<const uint4 *>(smem);
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      w[(tid + it * NT) % 10240] = make_uint2(it, tid);
      __syncthreads();
      const uint4 t = r[(tid * 5 + it) % 5120];
      acc.x ^= t.x; acc.y += t.y; acc.z ^= t.z; acc.w += t.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) sink[0] = 1.f;
  } else if (KIND == 7) {   // MFMA fed from LDS: ds_read_b128 fragments -> 6 independent bf16 MFMAs per step
    __bf16 *base = reinterpret_cast<__bf16 *>(smem);
    for (int i = tid; i < 40960; i += NT) base[i] = (__bf16)(0.001f * (i % 113));
    __syncthreads();
    f32x16 acc[6];
    for (int j = 0; j < 6; ++j) for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
    for (int it = 0; it < iters / 4; ++it) {
      const __bf16 *a = base + ((lane & 31) * 40 + (lane >> 5) * 8 + (it & 7) * 1280);
      bf16x8 ah = *reinterpret_cast<const bf16x8 *>(a), al = *reinterpret_cast<const bf16x8 *>(a + 10240);
      bf16x8 bh = *reinterpret_cast<const bf16x8 *>(a + 20480), bl = *reinterpret_cast<const bf16x8 *>(a + 30720);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[5], 0, 0, 0);
    }
    float s = 0; for (int j = 0; j < 6; ++j) for (int k = 0; k < 16; ++k) s += acc[j][k];
    if (s == 123.456f) sink[0] = s;
  }
}

template <int MODE>
static void run_case(const char *vname, int kind, const char *aname, hipStream_t sv, hipStream_t sa, const float *xs, unsigned *bad, float *sink) {
  CHECK(hipMemset(bad, 0, 64 * sizeof(unsigned)));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 20; ++rep) {
    if (kind == 0) hipLaunchKernelGGL(aggressor<0>, dim3(2048), dim3(256), 0, sa, sink, 4000);
    else if (kind == 1) hipLaunchKernelGGL(aggressor<1>, dim3(2048), dim3(256), 0, sa, sink, 4000);
    else if (kind == 2) hipLaunchKernelGGL(aggressor<2>, dim3(2048), dim3(256), 0, sa, sink, 4000);
    else if (kind == 4) { for (int k = 0; k < 8; ++k) launch_igemm(sa); }
    else if (kind == 5) hipLaunchKernelGGL((aggressor2<5, 512>), dim3(1024), dim3(512), 0, sa, sink, xs, 4000);
    else if (kind == 6) hipLaunchKernelGGL((aggressor2<6, 512>), dim3(1024), dim3(512), 81920, sa, sink, xs, 4000);
    else if (kind == 7) hipLaunchKernelGGL((aggressor2<7, 512>), dim3(1024), dim3(512), 81920, sa, sink, xs, 4000);
    else if (kind == 8) hipLaunchKernelGGL((aggressor2<7, 256>), dim3(1024), dim3(256), 81920, sa, sink, xs, 4000);
    hipLaunchKernelGGL(victim<MODE>, dim3(4096), dim3(256), 0, sv, xs, bad, 2000, 15, 20);
  }
  CHECK(hipDeviceSynchronize());
  unsigned h[64];
  CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long long tot = 0; unsigned q[4] = {0, 0, 0, 0};
  for (int i = 0; i < 64; ++i) { tot += h[i]; q[i / 16] += h[i]; }
  printf("victim %-28s aggressor %-22s mismatches %llu  per lane quarter [%u %u %u %u]\n", vname, aname, tot, q[0], q[1], q[2], q[3]);
}

int main(int argc, char **argv) {
  const char *libp = argc > 1 ? argv[1] : "sgcdet_amd/csrc/libsgcdet_amd.so";
  void *h = dlopen(libp, RTLD_NOW);
  if (!h) { printf("dlopen %s: %s\n", libp, dlerror()); return 1; }
  sgc_conv3d_cl_bf16x3 = (conv_fn)dlsym(h, "sgc_conv3d_cl_bf16x3");
  sgc_last_error = (const char *(*)())dlsym(h, "sgc_last_error");
  const bool only_igemm = argc > 2;
  printf("aggressor library %s\n", libp);
  hipStream_t sv, sa;
  CHECK(hipStreamCreate(&sv)); CHECK(hipStreamCreate(&sa));
  float *xs, *sink; unsigned *bad;
  CHECK(hipMalloc(&xs, 1024 * sizeof(float))); CHECK(hipMalloc(&sink, 16 * sizeof(float))); CHECK(hipMalloc(&bad, 64 * sizeof(unsigned)));
  std::vector<float> hx(1024);
  for (int i = 0; i < 1024; ++i) hx[i] = (float)(i % 97) / 97.0f;
  CHECK(hipMemcpy(xs, hx.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
  CHECK(hipMalloc(&g_x, 25600 * 256 * 4)); CHECK(hipMalloc(&g_y, 25600 * 256 * 4));
  CHECK(hipMalloc(&g_wh, 256 * 256 * 2)); CHECK(hipMalloc(&g_wl, 256 * 256 * 2));
  CHECK(hipMemset(g_x, 0, 25600 * 256 * 4)); CHECK(hipMemset(g_wh, 0, 256 * 256 * 2)); CHECK(hipMemset(g_wl, 0, 256 * 256 * 2));
  CHECK(hipFuncSetAttribute((const void *)aggressor2<6, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  CHECK(hipFuncSetAttribute((const void *)aggressor2<7, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  CHECK(hipFuncSetAttribute((const void *)aggressor2<7, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  const char *an[9] = {"mfma_32x32x16_bf16", "mfma_32x32x2_f32", "plain VALU", "none", "sgc igemm bf16x3",
                       "bf16 hi/lo split VALU", "LDS traffic 80KiB/512thr", "LDS-fed bf16 MFMA 512thr", "LDS-fed bf16 MFMA 256thr"};
  for (int kind = 8; kind >= 0; --kind) {
    if (only_igemm && kind != 4) continue;
    run_case<0>("v_pk_fma_f32 op_sel", kind, an[kind], sv, sa, xs, bad, sink);
    if (kind == 4) {
      run_case<1>("v_pk_fma_f32", kind, an[kind], sv, sa, xs, bad, sink);
      run_case<2>("v_pk_mul_f32", kind, an[kind], sv, sa, xs, bad, sink);
    }
  }
  return 0;
}
