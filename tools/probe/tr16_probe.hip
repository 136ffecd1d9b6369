// Lane map of ds_read_b64_tr_b16 (gfx950): every 16-lane group reads a block of 4 rows x 16 columns of 16-bit elements; lane 4q+p
// of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i of the 4 rows (row q in element q).
// Build + run: hipcc --offload-arch=gfx950 -O2 -o tr16_probe tools/probe/tr16_probe.hip && ./tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;          // element (row r, col c) = r * 64 + c
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const short* addr = lds + (g * 4 + q) * 64 + 4 * pp;                      // group g: rows 4g..4g+3, columns 0..15
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 512); short h[256];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
    const int g = lane >> 4, i = lane & 15, want = (g * 4 + e) * 64 + i;   // row 4g+e, column i
    if (h[lane * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got (r %d, c %d) want (r %d, c %d)\n", lane, e, h[lane*4+e] / 64, h[lane*4+e] % 64, want / 64, want % 64); ++bad; }
  }
  printf("%s (%d mismatches)\n", bad ? "MAP DIFFERS" : "map as documented", bad);
  return bad != 0;
}
