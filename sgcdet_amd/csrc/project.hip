// Voxel -> pixel projection, visibility mask and (camera, query) pair compaction.
//
// Replaces VoxFormerEncoder_DFA3D.point_sampling (TU/encoder.py:179-223: ~15 small torch
// ops + 3 host->device copies per call) and the per-camera `nonzero` / rebatch Python loops
// of DeformCrossAttention_DFA3D.forward (TU/deformable_cross_attention.py:759-773, N host
// syncs per level) by four tiny launches and ONE optional host read (the 4 ints of `totals`).
#include "common.hpp"

namespace sgc {

// Fixed arithmetic order, no FMA contraction (HIP's __f*_rn are plain operators that hipcc
// would contract, so contraction is switched off for this kernel by pragma):
//   p = ref + origin;  cam_r = ((P_r0*x + P_r1*y) + P_r2*z) + P_r3
//   den = max(cam_z, eps); u = (cam_x/den)*(1/img_w); v = (cam_y/den)*(1/img_h);
//   zn = (cam_z - d_near) * (1/(d_far - d_near));  mask = zn > eps & eps < u < 1-eps & eps < v < 1-eps
// torch's GPU `tensor / python_scalar` (TU/encoder.py:209-211) multiplies by the fp32
// reciprocal, which is what rw/rh/rd reproduce.  The oracle uses the same order.
__global__ void project_points_kernel(const float *__restrict__ ref3d, const int64_t *__restrict__ sel,
                                      const float *__restrict__ origin,
                                      const float *__restrict__ proj, float *__restrict__ ref_cam,
                                      uint8_t *__restrict__ mask, int N, int Nq, float rw, float rh,
                                      float d_near, float rd) {
#pragma clang fp contract(off)
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (q >= Nq) return;
  const float eps = 1e-5f;
  const float hi = 1.0f - eps;
  const int64_t r = sel ? sel[q] : q;             // the q-th query is voxel sel[q] (DenseHead.py:66, transformer.py:145-146)
  const float x = ref3d[r * 3] + origin[0];
  const float y = ref3d[r * 3 + 1] + origin[1];
  const float z = ref3d[r * 3 + 2] + origin[2];
  const float *P = proj + (int64_t)n * 12;
  float cam[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    cam[r] = ((P[r * 4] * x + P[r * 4 + 1] * y) + P[r * 4 + 2] * z) + P[r * 4 + 3];
  const float den = fmaxf(cam[2], eps);
  const float u = (cam[0] / den) * rw;
  const float v = (cam[1] / den) * rh;
  const float zn = (cam[2] - d_near) * rd;
  float *o = ref_cam + ((int64_t)n * Nq + q) * 3;
  o[0] = u; o[1] = v; o[2] = zn;
  // the reference's `points_d` is a VIEW of reference_points_cam[..., 2:3] and that slice is overwritten in place
  // with the normalised depth before `volume_mask = points_d > eps` runs (TU/encoder.py:203-213): the depth test is
  // on zn, i.e. points closer than d_near (+ eps * range) are dropped, not only points behind the camera
  mask[(int64_t)n * Nq + q] = (uint8_t)(zn > eps && u > eps && u < hi && v > eps && v < hi);
}

// block-wide exclusive scan of one flag per thread (1024 threads = 16 waves)
__device__ __forceinline__ int block_scan_flags(bool flag, int *total, int *wave_sums) {
  const unsigned long long bal = __ballot(flag);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_sums[wid] = __popcll(bal);
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    const int s = wave_sums[w];
    if (w < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + in_wave;
}

// one block per camera: local rank of every visible query (ascending q) and the count
__global__ __launch_bounds__(1024) void cam_rank_kernel(const uint8_t *__restrict__ mask, int Nq,
                                                        int32_t *__restrict__ slot, int32_t *__restrict__ cam_count) {
  __shared__ int wave_sums[16];
  const int n = blockIdx.x;
  int running = 0;
  for (int q0 = 0; q0 < Nq; q0 += blockDim.x) {
    const int q = q0 + threadIdx.x;
    const bool f = q < Nq && mask[(int64_t)n * Nq + q] != 0;
    int tot;
    const int r = block_scan_flags(f, &tot, wave_sums);
    if (q < Nq) slot[(int64_t)n * Nq + q] = f ? running + r : -1;
    running += tot;
  }
  if (threadIdx.x == 0) cam_count[n] = running;
}

// single block: exclusive scan of the camera counts, n_pairs, max_len
__global__ void cam_offset_kernel(const int32_t *__restrict__ cam_count, int N, int32_t *__restrict__ cam_offset,
                                  int32_t *__restrict__ totals) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0, mx = 0;
    for (int n = 0; n < N; ++n) {
      cam_offset[n] = acc;
      const int c = cam_count[n];
      acc += c;
      mx = c > mx ? c : mx;
    }
    cam_offset[N] = acc;
    totals[0] = acc;
    totals[2] = mx;
    totals[3] = 0;
  }
}

__global__ void fill_pairs_kernel(const int32_t *__restrict__ cam_offset, int N, int Nq, int32_t *__restrict__ slot,
                                  int32_t *__restrict__ pair_cam, int32_t *__restrict__ pair_q) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (q >= Nq) return;
  const int r = slot[(int64_t)n * Nq + q];
  if (r < 0) return;
  const int p = cam_offset[n] + r;
  slot[(int64_t)n * Nq + q] = p;
  pair_cam[p] = n;
  pair_q[p] = q;
}

__global__ void vox_count_kernel(const uint8_t *__restrict__ mask, int N, int Nq, int32_t *__restrict__ vox_count) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Nq) return;
  int c = 0;
  for (int n = 0; n < N; ++n) c += mask[(int64_t)n * Nq + q] != 0;
  vox_count[q] = c;
}

// single block: ascending list of queries seen by at least one camera
__global__ __launch_bounds__(1024) void valid_index_kernel(const int32_t *__restrict__ vox_count, int Nq,
                                                           int32_t *__restrict__ valid_index,
                                                           int32_t *__restrict__ totals, int32_t *__restrict__ row_of) {
  __shared__ int wave_sums[16];
  int running = 0;
  for (int q0 = 0; q0 < Nq; q0 += blockDim.x) {
    const int q = q0 + threadIdx.x;
    const bool f = q < Nq && vox_count[q] > 0;
    int tot;
    const int r = block_scan_flags(f, &tot, wave_sums);
    if (f) valid_index[running + r] = q;
    if (q < Nq && row_of) row_of[q] = f ? running + r : -1;      // inverse of valid_index (sgc_level_tail gathers by it)
    running += tot;
  }
  if (threadIdx.x == 0) totals[1] = running;
}

// ---------------------------------------------------------------------------------------------
// Round 5: the same compaction in TWO launches of independent 1024-query segments (the five kernels above are one block per
// camera walking all Nq queries, a one-thread scan of the camera counts, and ONE block walking all Nq voxels: 5 launches and
// 30 us per level at config 2, 100 us per level at config 5 where Nq = 73 728).
//   scan1  grid (S, N + 1), S = ceil(Nq / 1024).  Row y < N: camera y, segment x -- local rank of every visible query inside the
//          segment -> slot, the segment's count -> seg_cam[y * S + x].  Row y == N: voxel segment x -- #cameras seeing q ->
//          vox_count, local rank of the seen voxels -> row_of, the segment's count -> seg_vox[x].
//   scan2  same grid.  Every block sums the segment counts in front of it (<= N * S ints, a block reduction) and adds that
//          base: slot / pair_cam / pair_q (camera rows), valid_index / row_of (voxel row); the blocks of segment 0 also write
//          cam_count / cam_offset, block (0, N) the totals.
// Pure integer work: the outputs are identical to the five-kernel form (tested), which stays for workspace == null.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_sum_1024(int v, int *wave_sums) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();                                  // wave_sums may still be read by a previous use
  if (lane == 0) wave_sums[wid] = v;
  __syncthreads();
  int tot = 0;
  for (int w = 0; w < 16; ++w) tot += wave_sums[w];
  return tot;
}

__global__ __launch_bounds__(1024) void pairs_scan1_kernel(const uint8_t *__restrict__ mask, int N, int Nq, int S,
                                                           int32_t *__restrict__ slot, int32_t *__restrict__ vox_count,
                                                           int32_t *__restrict__ row_of, int32_t *__restrict__ seg_cam,
                                                           int32_t *__restrict__ seg_vox) {
  __shared__ int wave_sums[16];
  const int seg = blockIdx.x, y = blockIdx.y, q = seg * 1024 + threadIdx.x;
  int tot;
  if (y < N) {
    const bool f = q < Nq && mask[(int64_t)y * Nq + q] != 0;
    const int r = block_scan_flags(f, &tot, wave_sums);
    if (q < Nq) slot[(int64_t)y * Nq + q] = f ? r : -1;
    if (threadIdx.x == 0) seg_cam[y * S + seg] = tot;
  } else {
    int c = 0;
    if (q < Nq)
      for (int n = 0; n < N; ++n) c += mask[(int64_t)n * Nq + q] != 0;
    const bool f = c > 0;
    const int r = block_scan_flags(f, &tot, wave_sums);
    if (q < Nq) { vox_count[q] = c; row_of[q] = f ? r : -1; }
    if (threadIdx.x == 0) seg_vox[seg] = tot;
  }
}

__global__ __launch_bounds__(1024) void pairs_scan2_kernel(int N, int Nq, int S, int32_t *__restrict__ slot,
                                                           int32_t *__restrict__ pair_cam, int32_t *__restrict__ pair_q,
                                                           int32_t *__restrict__ cam_count, int32_t *__restrict__ cam_offset,
                                                           int32_t *__restrict__ valid_index, int32_t *__restrict__ row_of,
                                                           int32_t *__restrict__ totals, const int32_t *__restrict__ seg_cam,
                                                           const int32_t *__restrict__ seg_vox) {
  __shared__ int wave_sums[16];
  const int seg = blockIdx.x, y = blockIdx.y, tid = threadIdx.x, q = seg * 1024 + tid;
  if (y < N) {
    const int before = y * S + seg;                  // segment counts in front of this one, camera-major
    int part = 0;
    for (int i = tid; i < before; i += 1024) part += seg_cam[i];
    const int base = block_sum_1024(part, wave_sums);
    if (seg == 0) {                                  // base == pairs of the cameras before y
      int mine = 0;
      for (int i = tid; i < S; i += 1024) mine += seg_cam[y * S + i];
      const int cnt = block_sum_1024(mine, wave_sums);
      if (tid == 0) { cam_offset[y] = base; cam_count[y] = cnt; }
    }
    if (q < Nq) {
      const int r = slot[(int64_t)y * Nq + q];
      if (r >= 0) {
        const int p = base + r;
        slot[(int64_t)y * Nq + q] = p;
        pair_cam[p] = y;
        pair_q[p] = q;
      }
    }
  } else {
    int part = 0;
    for (int i = tid; i < seg; i += 1024) part += seg_vox[i];
    const int base = block_sum_1024(part, wave_sums);
    if (q < Nq) {
      const int r = row_of[q];
      if (r >= 0) { valid_index[base + r] = q; row_of[q] = base + r; }
    }
    if (seg == 0) {                                  // totals: {n_pairs, n_valid, max pairs of a camera, 0}
      int vt = 0;
      for (int i = tid; i < S; i += 1024) vt += seg_vox[i];
      const int n_valid = block_sum_1024(vt, wave_sums);
      int psum = 0, pmax = 0;
      for (int n = tid; n < N; n += 1024) {
        int c = 0;
        for (int i = 0; i < S; ++i) c += seg_cam[n * S + i];
        psum += c;
        pmax = c > pmax ? c : pmax;
      }
      const int n_pairs = block_sum_1024(psum, wave_sums);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) pmax = max(pmax, __shfl_xor(pmax, o));
      __syncthreads();
      if ((tid & 63) == 0) wave_sums[tid >> 6] = pmax;
      __syncthreads();
      if (tid == 0) {
        int mx = 0;
        for (int w = 0; w < 16; ++w) mx = wave_sums[w] > mx ? wave_sums[w] : mx;
        totals[0] = n_pairs; totals[1] = n_valid; totals[2] = mx; totals[3] = 0;
        cam_offset[N] = n_pairs;
      }
    }
  }
}

int g_tune_compact2 = 1;       // 1: the two-launch segment form of sgc_compact_pairs (needs the workspace), 0: the five kernels

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_project_points(const float *ref3d, const int64_t *sel_or_null, const float *origin, const float *proj,
                                  float *ref_cam, uint8_t *mask,
                                  int N, int Nq, float img_w, float img_h, float d_near, float d_far,
                                  sgc_stream_t stream) {
  if (!ref3d || !origin || !proj || !ref_cam || !mask) return set_error(SGC_EINVAL, "sgc_project_points: null pointer");
  if (N <= 0 || Nq < 0 || N > 65535) return set_error(SGC_EINVAL, "sgc_project_points: bad N/Nq");
  if (Nq == 0) return SGC_OK;
  const float rw = 1.0f / img_w, rh = 1.0f / img_h, rd = 1.0f / (d_far - d_near);
  hipLaunchKernelGGL(project_points_kernel, dim3(ceil_div(Nq, 256), N), dim3(256), 0, (hipStream_t)stream, ref3d,
                     sel_or_null, origin, proj, ref_cam, mask, N, Nq, rw, rh, d_near, rd);
  return check_launch("project_points_kernel");
}

extern "C" int sgc_compact_pairs(const uint8_t *mask, int N, int Nq,
                                 int32_t *cam_count, int32_t *cam_offset,
                                 int32_t *pair_cam, int32_t *pair_q, int32_t *slot,
                                 int32_t *vox_count, int32_t *valid_index, int32_t *totals,
                                 int32_t *workspace, sgc_stream_t stream) {
  if (!mask || !cam_count || !cam_offset || !pair_cam || !pair_q || !slot || !vox_count || !valid_index || !totals)
    return set_error(SGC_EINVAL, "sgc_compact_pairs: null pointer");
  if (N <= 0 || Nq <= 0 || N > 65535) return set_error(SGC_EINVAL, "sgc_compact_pairs: bad N/Nq");
  hipStream_t st = (hipStream_t)stream;
  if (workspace && g_tune_compact2 && N <= 4096) {
    // workspace: row_of [Nq] | seg_cam [N * S] | seg_vox [S]   (N * S + S <= N * Nq + 2 N + 64: the documented size covers it)
    const int S = ceil_div(Nq, 1024);
    int32_t *row_of = workspace, *seg_cam = workspace + Nq, *seg_vox = seg_cam + (int64_t)N * S;
    hipLaunchKernelGGL(pairs_scan1_kernel, dim3(S, N + 1), dim3(1024), 0, st, mask, N, Nq, S, slot, vox_count, row_of, seg_cam, seg_vox);
    hipLaunchKernelGGL(pairs_scan2_kernel, dim3(S, N + 1), dim3(1024), 0, st, N, Nq, S, slot, pair_cam, pair_q, cam_count, cam_offset,
                       valid_index, row_of, totals, seg_cam, seg_vox);
    return check_launch("sgc_compact_pairs (segment form)");
  }
  hipLaunchKernelGGL(cam_rank_kernel, dim3(N), dim3(1024), 0, st, mask, Nq, slot, cam_count);
  hipLaunchKernelGGL(cam_offset_kernel, dim3(1), dim3(64), 0, st, cam_count, N, cam_offset, totals);
  hipLaunchKernelGGL(fill_pairs_kernel, dim3(ceil_div(Nq, 256), N), dim3(256), 0, st, cam_offset, N, Nq, slot,
                     pair_cam, pair_q);
  hipLaunchKernelGGL(vox_count_kernel, dim3(ceil_div(Nq, 256)), dim3(256), 0, st, mask, N, Nq, vox_count);
  hipLaunchKernelGGL(valid_index_kernel, dim3(1), dim3(1024), 0, st, vox_count, Nq, valid_index, totals, workspace);
  return check_launch("sgc_compact_pairs");
}
