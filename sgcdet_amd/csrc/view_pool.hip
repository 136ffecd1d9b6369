// Inter-view aggregation and volume glue for gfx950.
//
// The reference scatters per-camera results into dense slots [N,1,Nq,C] (262 MB at
// config 2), projects K/V for every slot including the invisible ones and masks them
// with -inf (TU/deformable_cross_attention.py:815-833).  Here the visible (camera, query)
// pairs stay a compact list; `slot[n,q]` (pair index or -1) is the only dense object.
#include "common.hpp"

namespace sgc {

int g_tune_pq_depth = 4;      // knob view_depth: pair rows in flight per lane in view_attend_pq_kernel (1 | 2 | 4 | 8) and in the
                               // group kernels of view_mean / view_attend (>= 4: four, else one) -- A/B; results identical
int g_tune_view_group = 1;     // 0: the per-camera loops of rounds 1-2 in view_mean / view_attend (A/B; results identical)

// mean over the cameras that see voxel valid_index[i]; C/4 lanes per voxel (float4 rows)
// LG = C/4 in {32, 64}: the lanes of a voxel form a GROUP inside one wave.  Round 3: the group reads the voxel's slot column
// 32 / 64 cameras at a time with ONE load per lane, and walks only the visible cameras (ballot + shuffle), with the next
// camera's row requested before the current one is added -- instead of one dependent slot load per camera (28 of config 2's 40
// cameras do not see a voxel) followed by a dependent row load.  Same cameras in the same order, same sums: bit-identical.
template <int LG, int PD = 4>
__global__ __launch_bounds__(256) void view_mean_group_kernel(const float *__restrict__ feat, const int32_t *__restrict__ slot,
                                                              const int32_t *__restrict__ valid_index, float *__restrict__ mean,
                                                              int N, int Nq, int n_valid, const int32_t *__restrict__ n_dev) {
  if (n_dev) n_valid = min(n_valid, *n_dev);
  constexpr int C = LG * 4;
  const int lane = threadIdx.x & 63, gl = lane & (LG - 1), gbase = lane & ~(LG - 1);
  const int64_t total = (int64_t)n_valid * LG, span = ((total + 63) / 64) * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < span; idx += (int64_t)gridDim.x * blockDim.x) {
    const bool live = idx < total;
    const int i = (int)((live ? idx : total - 1) / LG);
    const int q = valid_index[i];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int nb = 0; nb < N; nb += LG) {
      const int pl = nb + gl < N ? slot[(int64_t)(nb + gl) * Nq + q] : -1;
      unsigned long long m = __ballot(pl >= 0);
      if (LG == 32) m = (m >> gbase) & 0xffffffffull;
      if (!m) continue;
      // PD rows in flight per lane (round 5; one row ahead left the kernel a chain of dependent loads): r[k] holds the row of the
      // (j0 + k)-th visible camera, refilled with the row PD cameras later as soon as it has been added.  Same order, same sums.
      const int nvis = __popcll(m);
      float4 r[PD];
#pragma unroll
      for (int k = 0; k < PD; ++k) {
        r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m) {
          const int pn = __shfl(pl, gbase + __builtin_ctzll(m));
          m &= m - 1;
          r[k] = reinterpret_cast<const float4 *>(feat + (int64_t)pn * C)[gl];
        }
      }
      for (int j0 = 0; j0 < nvis; j0 += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
          if (j0 + k >= nvis) break;
          const float4 v = r[k];
          if (m) {
            const int pn = __shfl(pl, gbase + __builtin_ctzll(m));
            m &= m - 1;
            r[k] = reinterpret_cast<const float4 *>(feat + (int64_t)pn * C)[gl];
          }
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          ++cnt;
        }
      }
    }
    if (live) {
      const float fc = (float)cnt;
      reinterpret_cast<float4 *>(mean + (int64_t)i * C)[gl] = make_float4(acc.x / fc, acc.y / fc, acc.z / fc, acc.w / fc);
    }
  }
}

__global__ __launch_bounds__(256) void view_mean_kernel(const float *__restrict__ feat,
                                                        const int32_t *__restrict__ slot,
                                                        const int32_t *__restrict__ valid_index,
                                                        float *__restrict__ mean, int N, int Nq, int C, int n_valid,
                                                        const int32_t *__restrict__ n_dev) {
  if (n_dev) n_valid = min(n_valid, *n_dev);     // row count produced on the device (no host read-back)
  const int C4 = C >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)n_valid * C4;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / C4), c4 = (int)(idx - (int64_t)i * C4);
    const int q = valid_index[i];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int n = 0; n < N; ++n) {
      const int p = slot[(int64_t)n * Nq + q];
      if (p < 0) continue;
      const float4 v = reinterpret_cast<const float4 *>(feat + (int64_t)p * C)[c4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      ++cnt;
    }
    const float fc = (float)cnt;
    reinterpret_cast<float4 *>(mean + (int64_t)i * C)[c4] =
        make_float4(acc.x / fc, acc.y / fc, acc.z / fc, acc.w / fc);
  }
}

__global__ void view_mean_scalar_kernel(const float *__restrict__ feat, const int32_t *__restrict__ slot,
                                        const int32_t *__restrict__ valid_index, float *__restrict__ mean,
                                        int N, int Nq, int C, int n_valid, const int32_t *__restrict__ n_dev) {
  if (n_dev) n_valid = min(n_valid, *n_dev);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)n_valid * C;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / C), c = (int)(idx - (int64_t)i * C);
    const int q = valid_index[i];
    float acc = 0.f;
    int cnt = 0;
    for (int n = 0; n < N; ++n) {
      const int p = slot[(int64_t)n * Nq + q];
      if (p < 0) continue;
      acc += feat[(int64_t)p * C + c];
      ++cnt;
    }
    mean[idx] = acc / (float)cnt;
  }
}

// Group form of view_attend_kernel<4> for heads * G in {32, 64} lanes per voxel (C = 128 / 256 with 8 heads): the voxel's slot
// column comes in with one load per lane, only the visible cameras are walked, and the next camera's k | v row is requested
// before the current one enters the online softmax (see view_mean_group_kernel).  Same cameras, same order, same arithmetic as
// the generic kernel below: bit-identical.
template <int LG, int PD = 4>
__global__ __launch_bounds__(256) void view_attend_group_kernel(const float *__restrict__ q, const float *__restrict__ kv,
                                                                const int32_t *__restrict__ slot,
                                                                const int32_t *__restrict__ valid_index, float *__restrict__ ctx,
                                                                int N, int Nq, int heads, int n_valid, float scale,
                                                                const int32_t *__restrict__ n_dev) {
  if (n_dev) n_valid = min(n_valid, *n_dev);
  if (n_valid <= 0) return;
  constexpr int C = LG * 4;
  const int G = LG / heads;                          // lanes per head
  const int lane = threadIdx.x & 63, gl = lane & (LG - 1), gbase = lane & ~(LG - 1);
  const int64_t total = (int64_t)n_valid * LG, span = ((total + 63) / 64) * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < span; idx += (int64_t)gridDim.x * blockDim.x) {
    const bool live = idx < total;
    const int i = (int)((live ? idx : total - 1) / LG);
    const int vq = valid_index[i];
    const int c0 = gl * 4;                           // == h * hd + g * 4 of the generic kernel
    const float4 q4 = *reinterpret_cast<const float4 *>(q + (int64_t)i * C + c0);
    const float qv[4] = {q4.x * scale, q4.y * scale, q4.z * scale, q4.w * scale};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float mx = -INFINITY, sum = 0.f;
    for (int nb = 0; nb < N; nb += LG) {
      const int pl = nb + gl < N ? slot[(int64_t)(nb + gl) * Nq + vq] : -1;
      unsigned long long m = __ballot(pl >= 0);
      if (LG == 32) m = (m >> gbase) & 0xffffffffull;
      if (!m) continue;
      // PD k | v rows in flight per lane (see view_mean_group_kernel)
      const int nvis = __popcll(m);
      float4 rk[PD], rv[PD];
#pragma unroll
      for (int k = 0; k < PD; ++k) {
        rk[k] = rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m) {
          const int pn = __shfl(pl, gbase + __builtin_ctzll(m));
          m &= m - 1;
          rk[k] = *reinterpret_cast<const float4 *>(kv + (int64_t)pn * 2 * C + c0);
          rv[k] = *reinterpret_cast<const float4 *>(kv + (int64_t)pn * 2 * C + C + c0);
        }
      }
      for (int j0 = 0; j0 < nvis; j0 += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
          if (j0 + k >= nvis) break;
          const float4 k4 = rk[k], v4 = rv[k];
          if (m) {
            const int pn = __shfl(pl, gbase + __builtin_ctzll(m));
            m &= m - 1;
            rk[k] = *reinterpret_cast<const float4 *>(kv + (int64_t)pn * 2 * C + c0);
            rv[k] = *reinterpret_cast<const float4 *>(kv + (int64_t)pn * 2 * C + C + c0);
          }
          const float kx[4] = {k4.x, k4.y, k4.z, k4.w}, vx[4] = {v4.x, v4.y, v4.z, v4.w};
          float d = 0.f;
#pragma unroll
          for (int v = 0; v < 4; ++v) d += qv[v] * kx[v];
          for (int o = 1; o < G; o <<= 1) d += __shfl_xor(d, o);
          const float nm = fmaxf(mx, d);
          const float corr = expf(mx - nm);
          const float e = expf(d - nm);
          sum = sum * corr + e;
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] = acc[v] * corr + e * vx[v];
          mx = nm;
        }
      }
    }
    if (live) *reinterpret_cast<float4 *>(ctx + (int64_t)i * C + c0) = make_float4(acc[0] / sum, acc[1] / sum, acc[2] / sum, acc[3] / sum);
  }
}

// ---------------------------------------------------------------------------------------------
// Projected-query form of the same attention (round 5).  nn.MultiheadAttention over the views (query length 1,
// TU/deformable_cross_attention.py:826-833) in-projects K and V for EVERY visible (camera, voxel) pair: a [pairs, C] x
// [C, 2C] GEMM whose result (2 KB per pair at C = 256) is written, read back once by the softmax, and thrown away --
// at 100 views 2.1 M pairs per scene, the largest Linear of the path.  Both projections commute with the softmax:
//   score(n, h) = (scale q_h) . (W_k,h x_n + b_k,h) = (scale W_k,h^T q_h) . x_n + const(h)        (the constant
//                 is the same for every camera n and drops out of the softmax over n),
//   ctx_h       = sum_n a(n, h) (W_v,h x_n + b_v,h) = W_v,h (sum_n a(n, h) x_n) + b_v,h          (sum_n a = 1),
// so the per-pair work is the raw pair feature x_n against a PROJECTED QUERY qp_h = scale W_k,h^T q_h (C floats per voxel
// and head, one [n_valid, C] x [C, heads C] GEMM on the voxels), and V is applied once per voxel to the attention-weighted
// feature s_h = sum_n a(n, h) x_n ([n_valid, heads C] x block-diagonal [heads C, C]).  Same function of the inputs as
// sgc_view_attend on the in-projected tensors; sums are associated differently (~1e-6 relative).
//
// One group of LG = C / 4 lanes per voxel (one wave at C = 256, half a wave at C = 128), 8 heads.  Per visible camera
// the lane holds 4 channels of x_n and the same 4 channels of the 8 projected queries: 8 partial dot products, reduced
// over the group with a packed butterfly (the first three exchange steps halve the number of values a lane carries:
// 4 + 2 + 1 shuffles, then one value over the remaining lane bits), scores parked in LDS.  An exact two-pass softmax
// (max, expf, sum, true division -- torch's softmax) over the voxel's cameras, then a second walk over the same pair
// rows (L2 hits) accumulates the 8 weighted features.  No atomics, fixed order: the same bits every run.
// ---------------------------------------------------------------------------------------------
constexpr int kPqHeads = 8, kPqMaxViews = 128;

template <int LG, int PD = 4>
__global__ __launch_bounds__(256) void view_attend_pq_kernel(const float *__restrict__ qp, const float *__restrict__ x,
                                                             const int32_t *__restrict__ slot,
                                                             const int32_t *__restrict__ valid_index, float *__restrict__ s,
                                                             int N, int Nq, int n_valid, const int32_t *__restrict__ n_dev) {
  if (n_dev) n_valid = min(n_valid, *n_dev);
  if (n_valid <= 0) return;
  constexpr int C = LG * 4, H = kPqHeads, GPW = 64 / LG, GPB = 4 * GPW;      // groups per wave / per block
  __shared__ __attribute__((aligned(16))) float sc[GPB][kPqMaxViews][H];     // scores, then softmax weights: [camera][head]
  __shared__ int plist[GPB][kPqMaxViews];                                     // pair index of the voxel's j-th visible camera
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, gl = lane & (LG - 1), gbase = lane & ~(LG - 1);
  const int grp = wid * GPW + (lane / LG);
  float (*my_sc)[H] = sc[grp];
  int *my_pl = plist[grp];
  const bool b0 = gl & 1, b1 = gl & 2, b2 = gl & 4;
  const int hm = (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0);                 // the head whose sum this lane holds after the butterfly
  const int64_t ngroups = n_valid, gstride = (int64_t)gridDim.x * GPB;
  // every lane of a wave runs the same number of iterations (shuffles need the whole wave): dead groups redo the last voxel
  const int64_t first = (int64_t)blockIdx.x * GPB + grp;
  const int64_t iters = (ngroups - (int64_t)blockIdx.x * GPB - wid * GPW + gstride - 1) / gstride;     // of this wave's first group
  int64_t i_raw = first;
  for (int64_t it = 0; it < iters; ++it, i_raw += gstride) {
    const bool live = i_raw < ngroups;
    const int i = (int)(live ? i_raw : ngroups - 1);
    const int vq = valid_index[i];
    // ---- the voxel's visible cameras, ascending camera index ----
    int cnt = 0;
    for (int nb = 0; nb < N; nb += LG) {
      const int pl = nb + gl < N ? slot[(int64_t)(nb + gl) * Nq + vq] : -1;
      unsigned long long m = __ballot(pl >= 0);
      if (LG == 32) m = (m >> gbase) & 0xffffffffull;
      if (pl >= 0) my_pl[cnt + __popcll(m & ((1ull << gl) - 1ull))] = pl;
      cnt += __popcll(m);
    }
    float4 q[H];
#pragma unroll
    for (int h = 0; h < H; ++h) q[h] = *reinterpret_cast<const float4 *>(qp + ((int64_t)i * H + h) * C + gl * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave: its LDS operations complete in order
    __builtin_amdgcn_wave_barrier();
    // ---- pass 1: scores ----
    // PD pair rows in flight per lane (round 5: with one row ahead the kernel ran at a third of the HBM rate -- a chain of ~30 dependent
    // loads per voxel -- although it issues ~170 instructions per pair): the rows of cameras j .. j + PD - 1 sit in xr[]
    float4 xr[PD];
#pragma unroll
    for (int k = 0; k < PD; ++k)
      xr[k] = k < cnt ? *reinterpret_cast<const float4 *>(x + (int64_t)my_pl[k] * C + gl * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = 0; j0 < cnt; j0 += PD) {
#pragma unroll
     for (int k = 0; k < PD; ++k) {
      const int j = j0 + k;
      if (j >= cnt) break;
      const float4 xv = xr[k];
      if (j + PD < cnt) xr[k] = *reinterpret_cast<const float4 *>(x + (int64_t)my_pl[j + PD] * C + gl * 4);
      float v[H];
#pragma unroll
      for (int h = 0; h < H; ++h) v[h] = ((q[h].x * xv.x + q[h].y * xv.y) + q[h].z * xv.z) + q[h].w * xv.w;
      float t[4], u[2];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = (b0 ? v[4 + k] : v[k]) + __shfl_xor(b0 ? v[k] : v[4 + k], 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) u[k] = (b1 ? t[2 + k] : t[k]) + __shfl_xor(b1 ? t[k] : t[2 + k], 2);
      float d = (b2 ? u[1] : u[0]) + __shfl_xor(b2 ? u[0] : u[1], 4);
#pragma unroll
      for (int o = 8; o < LG; o <<= 1) d += __shfl_xor(d, o);
      if (gl < 8) my_sc[j][hm] = d;
     }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- softmax over the cameras, per head: lane = (head gl & 7, cameras (gl >> 3) + k LG / 8) ----
    {
      const int h = gl & 7, j0 = gl >> 3;
      constexpr int JS = LG / 8;
      float mx = -INFINITY;
      for (int j = j0; j < cnt; j += JS) mx = fmaxf(mx, my_sc[j][h]);
#pragma unroll
      for (int o = 8; o < LG; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.f;
      for (int j = j0; j < cnt; j += JS) {
        const float e = expf(my_sc[j][h] - mx);
        my_sc[j][h] = e;
        sum += e;
      }
#pragma unroll
      for (int o = 8; o < LG; o <<= 1) sum += __shfl_xor(sum, o);
      for (int j = j0; j < cnt; j += JS) my_sc[j][h] = my_sc[j][h] / sum;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- pass 2: the 8 attention-weighted features ----
    float4 acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < PD; ++k)
      if (k < cnt) xr[k] = *reinterpret_cast<const float4 *>(x + (int64_t)my_pl[k] * C + gl * 4);
    for (int j0 = 0; j0 < cnt; j0 += PD) {
#pragma unroll
     for (int k = 0; k < PD; ++k) {
      const int j = j0 + k;
      if (j >= cnt) break;
      const float4 xv = xr[k];
      if (j + PD < cnt) xr[k] = *reinterpret_cast<const float4 *>(x + (int64_t)my_pl[j + PD] * C + gl * 4);
      const float4 a0 = *reinterpret_cast<const float4 *>(&my_sc[j][0]), a1 = *reinterpret_cast<const float4 *>(&my_sc[j][4]);
      const float a[H] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int h = 0; h < H; ++h) {
        acc[h].x += a[h] * xv.x; acc[h].y += a[h] * xv.y; acc[h].z += a[h] * xv.z; acc[h].w += a[h] * xv.w;
      }
     }
    }
    if (live) {
#pragma unroll
      for (int h = 0; h < H; ++h) *reinterpret_cast<float4 *>(s + ((int64_t)i * H + h) * C + gl * 4) = acc[h];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the next voxel overwrites this group's LDS
    __builtin_amdgcn_wave_barrier();
  }
}

// Softmax over views for query length 1 (nn.MultiheadAttention, :829-833): one group of
// G = head_dim/VEC lanes per (voxel, head); the dot product is reduced over the group with
// wave shuffles, the softmax over views runs online (running max / sum) in registers.
template <int VEC>
__global__ __launch_bounds__(256) void view_attend_kernel(const float *__restrict__ q,
                                                          const float *__restrict__ kv,
                                                          const int32_t *__restrict__ slot,
                                                          const int32_t *__restrict__ valid_index,
                                                          float *__restrict__ ctx, int N, int Nq, int C,
                                                          int heads, int n_valid, int G, float scale,
                                                          const int32_t *__restrict__ n_dev) {
  if (n_dev) n_valid = min(n_valid, *n_dev);
  if (n_valid <= 0) return;
  const int hd = C / heads;
  const int64_t total = (int64_t)n_valid * heads * G;
  const int64_t span = (((int64_t)total + 63) / 64) * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < span;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const bool live = idx < total;
    const int64_t id = live ? idx : total - 1;
    const int g = (int)(id % G);
    const int h = (int)((id / G) % heads);
    const int i = (int)(id / ((int64_t)G * heads));
    const int vq = valid_index[i];
    const int c0 = h * hd + g * VEC;
    float qv[VEC], acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      qv[v] = q[(int64_t)i * C + c0 + v] * scale;  // torch: q_scaled = q * sqrt(1/head_dim)
      acc[v] = 0.f;
    }
    float mx = -INFINITY, sum = 0.f;
    for (int n = 0; n < N; ++n) {
      const int p = slot[(int64_t)n * Nq + vq];
      if (p < 0) continue;  // uniform over the G lanes of a group
      const float *kp = kv + (int64_t)p * 2 * C + c0;
      float kx[VEC], vx[VEC];
      if (VEC == 4) {
        const float4 k4 = *reinterpret_cast<const float4 *>(kp);
        const float4 v4 = *reinterpret_cast<const float4 *>(kp + C);
        kx[0] = k4.x; kx[1 % VEC] = k4.y; kx[2 % VEC] = k4.z; kx[3 % VEC] = k4.w;
        vx[0] = v4.x; vx[1 % VEC] = v4.y; vx[2 % VEC] = v4.z; vx[3 % VEC] = v4.w;
      } else {
        kx[0] = kp[0]; vx[0] = kp[C];
      }
      float d = 0.f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) d += qv[v] * kx[v];
      for (int o = 1; o < G; o <<= 1) d += __shfl_xor(d, o);
      const float nm = fmaxf(mx, d);
      const float corr = expf(mx - nm);  // exp(-inf) = 0 on the first visible view
      const float e = expf(d - nm);
      sum = sum * corr + e;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = acc[v] * corr + e * vx[v];
      mx = nm;
    }
    if (live) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) ctx[(int64_t)i * C + c0 + v] = acc[v] / sum;
    }
  }
}

// Backward of view_attend_kernel (training over the pair list: no dense [N, L, C] slots, no full-size K/V projections --
// the reference back-propagates through nn.MultiheadAttention on the padded slots, TU/deformable_cross_attention.py:829-833).
// Same lane grouping as the forward: G lanes per (voxel, head).  With a_n the softmax weights, ctx = sum_n a_n v_n:
//   S = <d_ctx, ctx>;  da_n = <d_ctx, v_n>;  ds_n = a_n (da_n - S);
//   d_q = scale * sum_n ds_n k_n;  d_k_n = ds_n * scale * q;  d_v_n = a_n * d_ctx.
// Every pair belongs to exactly one voxel, so grad_kv rows are written once, without atomics.
template <int VEC>
__global__ __launch_bounds__(256) void view_attend_backward_kernel(const float *__restrict__ q, const float *__restrict__ kv,
                                                                   const int32_t *__restrict__ slot,
                                                                   const int32_t *__restrict__ valid_index,
                                                                   const float *__restrict__ ctx, const float *__restrict__ gctx,
                                                                   float *__restrict__ gq, float *__restrict__ gkv, int N, int Nq,
                                                                   int C, int heads, int n_valid, int G, float scale) {
  const int hd = C / heads;
  const int64_t total = (int64_t)n_valid * heads * G;
  const int64_t span = (((int64_t)total + 63) / 64) * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < span; idx += (int64_t)gridDim.x * blockDim.x) {
    const bool live = idx < total;
    const int64_t id = live ? idx : total - 1;
    const int g = (int)(id % G);
    const int h = (int)((id / G) % heads);
    const int i = (int)(id / ((int64_t)G * heads));
    const int vq = valid_index[i];
    const int c0 = h * hd + g * VEC;
    float qv[VEC], dc[VEC], dq[VEC];
    float S = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      qv[v] = q[(int64_t)i * C + c0 + v] * scale;
      dc[v] = gctx[(int64_t)i * C + c0 + v];
      S += dc[v] * ctx[(int64_t)i * C + c0 + v];
      dq[v] = 0.f;
    }
    for (int o = 1; o < G; o <<= 1) S += __shfl_xor(S, o);
    float mx = -INFINITY, sum = 0.f;
    for (int n = 0; n < N; ++n) {                      // pass 1: the softmax normalisation (as the forward computes it)
      const int p = slot[(int64_t)n * Nq + vq];
      if (p < 0) continue;
      const float *kp = kv + (int64_t)p * 2 * C + c0;
      float d = 0.f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) d += qv[v] * kp[v];
      for (int o = 1; o < G; o <<= 1) d += __shfl_xor(d, o);
      const float nm = fmaxf(mx, d);
      sum = sum * expf(mx - nm) + expf(d - nm);
      mx = nm;
    }
    for (int n = 0; n < N; ++n) {                      // pass 2: gradients
      const int p = slot[(int64_t)n * Nq + vq];
      if (p < 0) continue;
      const float *kp = kv + (int64_t)p * 2 * C + c0;
      float kx[VEC], vx[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) { kx[v] = kp[v]; vx[v] = kp[C + v]; }
      float d = 0.f, da = 0.f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) { d += qv[v] * kx[v]; da += dc[v] * vx[v]; }
      for (int o = 1; o < G; o <<= 1) { d += __shfl_xor(d, o); da += __shfl_xor(da, o); }
      const float a = expf(d - mx) / sum;
      const float ds = a * (da - S);
      if (live) {
        float *gp = gkv + (int64_t)p * 2 * C + c0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          dq[v] += ds * kx[v];
          gp[v] = ds * qv[v];
          gp[C + v] = a * dc[v];
        }
      }
    }
    if (live) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) gq[(int64_t)i * C + c0 + v] = dq[v] * scale;
    }
  }
}

__global__ void scatter_rows_kernel(const float *__restrict__ rows, const int32_t *__restrict__ idx,
                                    const int32_t *__restrict__ idx2, float *__restrict__ vol, int n, int C, int VEC,
                                    const int32_t *__restrict__ n_dev) {
  if (n_dev) n = min(n, *n_dev);
  const int CV = C / VEC;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (int64_t)n * CV;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / CV), c = (int)(t - (int64_t)i * CV);
    int64_t d = idx[i];
    if (idx2) d = idx2[d];
    if (VEC == 4)
      reinterpret_cast<float4 *>(vol + d * C)[c] = reinterpret_cast<const float4 *>(rows + (int64_t)i * C)[c];
    else
      vol[d * C + c] = rows[(int64_t)i * C + c];
  }
}

// [N,C,Hs,Ws] -> [N,H*W,C] crop + transpose through a padded 32x33 LDS tile:
// reads coalesced along w, writes coalesced along c.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                           int C, int Hs, int Ws, int H, int W, int step) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32;  // pixel tile over the cropped H*W
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int HW = H * W;
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, px = p0 + tx;
    float v = 0.f;
    if (c < C && px < HW) {
      const int h = px / W, w = px - h * W;
      v = src[(((int64_t)n * C + c) * Hs + (int64_t)h * step) * Ws + (int64_t)w * step];
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int px = p0 + j, c = c0 + tx;
    if (c < C && px < HW) dst[((int64_t)n * HW + px) * C + c] = tile[tx][j];
  }
}


// Adjoint of the crop + transpose above (training: the gradient of the channels-last rows w.r.t. the NCHW map the producer holds):
// dst[n][c][h][w] = src[n][h * W + w][c] for h < H, w < W, written for the WHOLE [Hd, Wd] plane of dst (zero outside the crop when
// Hd > H or Wd > W).  32 x 32 tiles through LDS: 128-byte reads along c, 128-byte writes along the destination pixels.
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                           int C, int H, int W, int Hd, int Wd) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32;                          // destination pixel tile over Hd * Wd
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int HWd = Hd * Wd;
  for (int j = ty; j < 32; j += 8) {                       // pixel p0 + j, channel c0 + tx
    const int pd = p0 + j, c = c0 + tx;
    float v = 0.f;
    if (pd < HWd && c < C) {
      const int h = pd / Wd, w = pd - h * Wd;
      if (h < H && w < W) v = src[((int64_t)n * H * W + (int64_t)h * W + w) * C + c];
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {                       // channel c0 + j, pixel p0 + tx
    const int c = c0 + j, pd = p0 + tx;
    if (c < C && pd < HWd) dst[((int64_t)n * C + c) * HWd + pd] = tile[tx][j];
  }
}

// Trilinear x2 upsample of a channels-last volume fused with the occupancy head:
//   up[v', :] = F.interpolate(vol, scale_factor=2, mode='trilinear', align_corners=False)   (AdaptiveSparseHead.py:64-69)
//   occ[v']   = sigmoid(dot(up[v', :], w) + b)                                               (:71, Sequential(Linear(C,1), Sigmoid))
// One group of C/4 lanes per OUTPUT voxel (a full wave at C = 256): up to 8 input rows are read as
// float4 lanes (L2-resident, the input is 1/8 of the output), the row is written once, and the dot
// product is reduced over the group with wave shuffles.  Source index / weights as torch's
// upsample_trilinear3d: t = max(0.5*(dst+0.5)-0.5, 0), i0 = floor(t), i1 = i0 + (i0 < n-1), l1 = t - i0.
__global__ __launch_bounds__(256) void upsample2x_occ_kernel(const float *__restrict__ vol, const float *__restrict__ w,
                                                             const float *__restrict__ b, float *__restrict__ up,
                                                             float *__restrict__ occ, int ix, int iy, int iz, int C, int G) {
  const int C4 = C >> 2;
  const int ox = 2 * ix, oy = 2 * iy, oz = 2 * iz;
  const int64_t nvox = (int64_t)ox * oy * oz;
  const int gpb = blockDim.x / G;                       // voxel groups per block
  const int g = threadIdx.x / G, l = threadIdx.x % G;
  for (int64_t v = (int64_t)blockIdx.x * gpb + g; v < nvox; v += (int64_t)gridDim.x * gpb) {
    const int z = (int)(v % oz), y = (int)((v / oz) % oy), x = (int)(v / ((int64_t)oz * oy));
    int i0[3], i1[3];
    float l0[3], l1[3];
    const int dst[3] = {x, y, z}, n[3] = {ix, iy, iz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float t = 0.5f * ((float)dst[a] + 0.5f) - 0.5f;
      t = t < 0.f ? 0.f : t;
      i0[a] = (int)t;
      i1[a] = i0[a] + (i0[a] < n[a] - 1 ? 1 : 0);
      l1[a] = t - (float)i0[a];
      l0[a] = 1.f - l1[a];
    }
    float dot = 0.f;
    for (int c4 = l; c4 < C4; c4 += G) {
      auto row = [&](int a, int bq, int c) {
        return reinterpret_cast<const float4 *>(vol + (((int64_t)a * iy + bq) * iz + c) * C)[c4];
      };
      const float4 v000 = row(i0[0], i0[1], i0[2]), v001 = row(i0[0], i0[1], i1[2]);
      const float4 v010 = row(i0[0], i1[1], i0[2]), v011 = row(i0[0], i1[1], i1[2]);
      const float4 v100 = row(i1[0], i0[1], i0[2]), v101 = row(i1[0], i0[1], i1[2]);
      const float4 v110 = row(i1[0], i1[1], i0[2]), v111 = row(i1[0], i1[1], i1[2]);
      float4 r;
#define SGC_TRI(f)                                                                                     \
  r.f = l0[0] * (l0[1] * (l0[2] * v000.f + l1[2] * v001.f) + l1[1] * (l0[2] * v010.f + l1[2] * v011.f)) + \
        l1[0] * (l0[1] * (l0[2] * v100.f + l1[2] * v101.f) + l1[1] * (l0[2] * v110.f + l1[2] * v111.f));
      SGC_TRI(x) SGC_TRI(y) SGC_TRI(z) SGC_TRI(w)
#undef SGC_TRI
      reinterpret_cast<float4 *>(up + v * C)[c4] = r;
      if (w) {
        const float4 ww = reinterpret_cast<const float4 *>(w)[c4];
        dot += r.x * ww.x + r.y * ww.y + r.z * ww.z + r.w * ww.w;
      }
    }
    if (w) {
      for (int o = 1; o < G; o <<= 1) dot += __shfl_xor(dot, o);
      if (l == 0) occ[v] = 1.f / (1.f + expf(-(dot + b[0])));
    }
  }
}

// vol[idx[i], :] += rows[i, :]  (volume = upsampled + DenseHead(selected voxels), AdaptiveSparseHead.py:77-82:
// the dense head's output is zero outside the selected voxels, so the full-volume add is a row scatter-add)
__global__ void scatter_add_rows_kernel(const float *__restrict__ rows, const int64_t *__restrict__ idx,
                                        float *__restrict__ vol, int n, int C4) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (int64_t)n * C4;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / C4), c = (int)(t - (int64_t)i * C4);
    float4 *dst = reinterpret_cast<float4 *>(vol + idx[i] * (int64_t)C4 * 4) + c;
    const float4 a = reinterpret_cast<const float4 *>(rows + (int64_t)i * C4 * 4)[c];
    float4 d = *dst;
    d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
    *dst = d;
  }
}

}  // namespace sgc

using namespace sgc;

static int grid_for(int64_t work, int block) {
  int64_t g = (work + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int sgc_view_attend_pq_supported(int N, int C, int heads) {
  return heads == sgc::kPqHeads && (C == 128 || C == 256) && N > 0 && N <= sgc::kPqMaxViews ? 1 : 0;
}

extern "C" int sgc_view_mean(const float *feat, const int32_t *slot, const int32_t *valid_index,
                             float *mean, int N, int Nq, int C, const int32_t *n_valid_dev_or_null, int n_valid,
                             sgc_stream_t stream) {
  const int32_t *n_dev = n_valid_dev_or_null;
  if (!feat || !slot || !valid_index || !mean) return set_error(SGC_EINVAL, "sgc_view_mean: null pointer");
  if (n_valid <= 0) return SGC_OK;
  const bool al16 = !((uintptr_t)feat & 15) && !((uintptr_t)mean & 15);
  const bool deep = g_tune_pq_depth >= 4;            // rows in flight per lane: 4 (default) | 1 (the round-3 form, A/B)
  if (C == 256 && al16 && g_tune_view_group && deep)
    hipLaunchKernelGGL((view_mean_group_kernel<64, 4>), dim3(grid_for((int64_t)n_valid * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, slot, valid_index, mean, N, Nq, n_valid, n_dev);
  else if (C == 256 && al16 && g_tune_view_group)
    hipLaunchKernelGGL((view_mean_group_kernel<64, 1>), dim3(grid_for((int64_t)n_valid * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, slot, valid_index, mean, N, Nq, n_valid, n_dev);
  else if (C == 128 && al16 && g_tune_view_group && deep)
    hipLaunchKernelGGL((view_mean_group_kernel<32, 4>), dim3(grid_for((int64_t)n_valid * 32, 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, slot, valid_index, mean, N, Nq, n_valid, n_dev);
  else if (C == 128 && al16 && g_tune_view_group)
    hipLaunchKernelGGL((view_mean_group_kernel<32, 1>), dim3(grid_for((int64_t)n_valid * 32, 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, slot, valid_index, mean, N, Nq, n_valid, n_dev);
  else if (C % 4 == 0 && al16)
    hipLaunchKernelGGL(view_mean_kernel, dim3(grid_for((int64_t)n_valid * (C / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, slot, valid_index, mean, N, Nq, C, n_valid, n_dev);
  else
    hipLaunchKernelGGL(view_mean_scalar_kernel, dim3(grid_for((int64_t)n_valid * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, feat, slot, valid_index, mean, N, Nq, C, n_valid, n_dev);
  return check_launch("view_mean_kernel");
}

extern "C" int sgc_view_attend(const float *q, const float *kv, const int32_t *slot,
                               const int32_t *valid_index, float *ctx,
                               int N, int Nq, int C, int heads, const int32_t *n_valid_dev_or_null, int n_valid,
                               sgc_stream_t stream) {
  const int32_t *n_dev = n_valid_dev_or_null;
  if (!q || !kv || !slot || !valid_index || !ctx) return set_error(SGC_EINVAL, "sgc_view_attend: null pointer");
  if (heads <= 0 || C % heads) return set_error(SGC_EINVAL, "sgc_view_attend: C %% heads != 0");
  if (n_valid <= 0) return SGC_OK;
  const int hd = C / heads;
  const float scale = sqrtf(1.0f / (float)hd);
  // lanes per (voxel, head): head_dim/4 when that is a power of two <= 64, else one lane per channel
  int vec = 4, G = hd / 4;
  if (hd % 4 || (G & (G - 1)) || G > 64 || ((uintptr_t)q & 15) || ((uintptr_t)kv & 15)) { vec = 1; G = hd; }
  if ((G & (G - 1)) || G > 64) return set_error(SGC_EUNSUP, "sgc_view_attend: head_dim %d not supported", hd);
  const int64_t work = (int64_t)n_valid * heads * G;
  const bool deep = g_tune_pq_depth >= 4;
  if (vec == 4 && heads * G == 64 && g_tune_view_group && deep)
    hipLaunchKernelGGL((view_attend_group_kernel<64, 4>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, q, kv,
                       slot, valid_index, ctx, N, Nq, heads, n_valid, scale, n_dev);
  else if (vec == 4 && heads * G == 64 && g_tune_view_group)
    hipLaunchKernelGGL((view_attend_group_kernel<64, 1>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, q, kv,
                       slot, valid_index, ctx, N, Nq, heads, n_valid, scale, n_dev);
  else if (vec == 4 && heads * G == 32 && g_tune_view_group && deep)
    hipLaunchKernelGGL((view_attend_group_kernel<32, 4>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, q, kv,
                       slot, valid_index, ctx, N, Nq, heads, n_valid, scale, n_dev);
  else if (vec == 4 && heads * G == 32 && g_tune_view_group)
    hipLaunchKernelGGL((view_attend_group_kernel<32, 1>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, q, kv,
                       slot, valid_index, ctx, N, Nq, heads, n_valid, scale, n_dev);
  else if (vec == 4)
    hipLaunchKernelGGL(view_attend_kernel<4>, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, q, kv,
                       slot, valid_index, ctx, N, Nq, C, heads, n_valid, G, scale, n_dev);
  else
    hipLaunchKernelGGL(view_attend_kernel<1>, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, q, kv,
                       slot, valid_index, ctx, N, Nq, C, heads, n_valid, G, scale, n_dev);
  return check_launch("view_attend_kernel");
}

extern "C" int sgc_view_attend_pq(const float *qp, const float *x, const int32_t *slot, const int32_t *valid_index, float *s,
                                  int N, int Nq, int C, int heads, const int32_t *n_valid_dev_or_null, int n_valid,
                                  sgc_stream_t stream) {
  if (!qp || !x || !slot || !valid_index || !s) return set_error(SGC_EINVAL, "sgc_view_attend_pq: null pointer");
  if (!sgc_view_attend_pq_supported(N, C, heads))
    return set_error(SGC_EUNSUP, "sgc_view_attend_pq: needs heads == %d, C in {128, 256}, N <= %d (got heads %d, C %d, N %d)",
                     kPqHeads, kPqMaxViews, heads, C, N);
  if (((uintptr_t)qp | (uintptr_t)x | (uintptr_t)s) & 15) return set_error(SGC_EINVAL, "sgc_view_attend_pq: pointers must be 16-byte aligned");
  if (n_valid <= 0) return SGC_OK;
  hipStream_t st = (hipStream_t)stream;
  const int gpb = C == 256 ? 4 : 8;                                    // voxel groups per block
  const dim3 grid((unsigned)std::min<int64_t>(((int64_t)n_valid + gpb - 1) / gpb, 256 * 16));
#define SGC_PQ_LAUNCH(LG, PD) hipLaunchKernelGGL((view_attend_pq_kernel<LG, PD>), grid, dim3(256), 0, st, qp, x, slot, valid_index, s, N, Nq, n_valid, n_valid_dev_or_null)
  const int pd = g_tune_pq_depth;
  if (C == 256) {
    if (pd >= 8) SGC_PQ_LAUNCH(64, 8); else if (pd >= 4) SGC_PQ_LAUNCH(64, 4); else if (pd >= 2) SGC_PQ_LAUNCH(64, 2); else SGC_PQ_LAUNCH(64, 1);
  } else {
    if (pd >= 8) SGC_PQ_LAUNCH(32, 8); else if (pd >= 4) SGC_PQ_LAUNCH(32, 4); else if (pd >= 2) SGC_PQ_LAUNCH(32, 2); else SGC_PQ_LAUNCH(32, 1);
  }
#undef SGC_PQ_LAUNCH
  return check_launch("view_attend_pq_kernel");
}

extern "C" int sgc_view_attend_backward(const float *q, const float *kv, const int32_t *slot, const int32_t *valid_index,
                                        const float *ctx, const float *grad_ctx, float *grad_q, float *grad_kv,
                                        int N, int Nq, int C, int heads, int n_valid, sgc_stream_t stream) {
  if (!q || !kv || !slot || !valid_index || !ctx || !grad_ctx || !grad_q || !grad_kv)
    return set_error(SGC_EINVAL, "sgc_view_attend_backward: null pointer");
  if (heads <= 0 || C % heads) return set_error(SGC_EINVAL, "sgc_view_attend_backward: C %% heads != 0");
  if (n_valid <= 0) return SGC_OK;
  const int hd = C / heads;
  const float scale = sqrtf(1.0f / (float)hd);
  int vec = 4, G = hd / 4;
  if (hd % 4 || (G & (G - 1)) || G > 64) { vec = 1; G = hd; }
  if ((G & (G - 1)) || G > 64) return set_error(SGC_EUNSUP, "sgc_view_attend_backward: head_dim %d not supported", hd);
  const int64_t work = (int64_t)n_valid * heads * G;
  if (vec == 4)
    hipLaunchKernelGGL(view_attend_backward_kernel<4>, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, q, kv, slot,
                       valid_index, ctx, grad_ctx, grad_q, grad_kv, N, Nq, C, heads, n_valid, G, scale);
  else
    hipLaunchKernelGGL(view_attend_backward_kernel<1>, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, q, kv, slot,
                       valid_index, ctx, grad_ctx, grad_q, grad_kv, N, Nq, C, heads, n_valid, G, scale);
  return check_launch("view_attend_backward_kernel");
}

extern "C" int sgc_scatter_rows(const float *rows, const int32_t *idx, const int32_t *idx2_or_null,
                                float *vol, const int32_t *n_dev_or_null, int n, int C, sgc_stream_t stream) {
  if (!rows || !idx || !vol) return set_error(SGC_EINVAL, "sgc_scatter_rows: null pointer");
  if (n <= 0) return SGC_OK;
  const int vec = (C % 4 == 0 && !((uintptr_t)rows & 15) && !((uintptr_t)vol & 15)) ? 4 : 1;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((int64_t)n * (C / vec), 256)), dim3(256), 0,
                     (hipStream_t)stream, rows, idx, idx2_or_null, vol, n, C, vec, n_dev_or_null);
  return check_launch("scatter_rows_kernel");
}

namespace sgc {
// The same through a 64 x 64 tile with 16-byte accesses on both sides (C % 64 == 0, W, Ws % 4 == 0, 16-byte aligned
// pointers: the feature maps): a float4 covers 4 pixels of one row on the way in and 4 channels of one pixel on the
// way out; tile[c][px] with a 65-float pitch keeps both LDS phases conflict-free.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel64(const float *__restrict__ src, float *__restrict__ dst,
                                                             int C, int Hs, int Ws, int H, int W) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int HW = H * W;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int c = e >> 4, q = e & 15;
    const int px = p0 + q * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (px < HW) {                                   // HW % 4 == 0: a group is in or out as a whole
      const int h = px / W, w = px - h * W;
      v = *reinterpret_cast<const float4 *>(src + (((int64_t)n * C + c0 + c) * Hs + h) * Ws + w);
    }
    tile[c][q * 4] = v.x; tile[c][q * 4 + 1] = v.y; tile[c][q * 4 + 2] = v.z; tile[c][q * 4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int px = e >> 4, g = e & 15;
    if (p0 + px < HW)
      *reinterpret_cast<float4 *>(dst + ((int64_t)n * HW + p0 + px) * C + c0 + g * 4) =
          make_float4(tile[g * 4][px], tile[g * 4 + 1][px], tile[g * 4 + 2][px], tile[g * 4 + 3][px]);
  }
}
}  // namespace sgc

extern "C" int sgc_nchw_to_nhwc_crop(const float *src, float *dst, int N, int C, int Hs, int Ws,
                                     int H, int W, int step, sgc_stream_t stream) {
  if (!src || !dst) return set_error(SGC_EINVAL, "sgc_nchw_to_nhwc_crop: null pointer");
  if (step < 1 || (int64_t)(H - 1) * step >= Hs || (int64_t)(W - 1) * step >= Ws || N <= 0 || C <= 0 || H <= 0 || W <= 0)
    return set_error(SGC_EINVAL, "sgc_nchw_to_nhwc_crop: bad sizes");
  if (N > 65535) return set_error(SGC_EUNSUP, "sgc_nchw_to_nhwc_crop: N > 65535");
  if (step == 1 && C % 64 == 0 && W % 4 == 0 && Ws % 4 == 0 && !(((uintptr_t)src | (uintptr_t)dst) & 15)) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel64, dim3(ceil_div(H * W, 64), C / 64, N), dim3(256), 0, (hipStream_t)stream, src,
                       dst, C, Hs, Ws, H, W);
    return check_launch("nchw_to_nhwc_kernel64");
  }
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ceil_div(H * W, 32), ceil_div(C, 32), N), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, Hs, Ws, H, W, step);
  return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int sgc_nhwc_to_nchw_pad(const float *src, float *dst, int N, int C, int H, int W, int Hd, int Wd, sgc_stream_t stream) {
  if (!src || !dst) return set_error(SGC_EINVAL, "sgc_nhwc_to_nchw_pad: null pointer");
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Hd < H || Wd < W) return set_error(SGC_EINVAL, "sgc_nhwc_to_nchw_pad: bad sizes");
  if (N > 65535) return set_error(SGC_EUNSUP, "sgc_nhwc_to_nchw_pad: N > 65535");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ceil_div(Hd * Wd, 32), ceil_div(C, 32), N), dim3(256), 0, (hipStream_t)stream, src, dst,
                     C, H, W, Hd, Wd);
  return check_launch("nhwc_to_nchw_kernel");
}

extern "C" int sgc_upsample2x_occ(const float *vol, const float *w_or_null, const float *b_or_null, float *up,
                                  float *occ_or_null, int ix, int iy, int iz, int C, sgc_stream_t stream) {
  if (!vol || !up) return set_error(SGC_EINVAL, "sgc_upsample2x_occ: null pointer");
  if ((w_or_null != nullptr) != (occ_or_null != nullptr) || (w_or_null && !b_or_null))
    return set_error(SGC_EINVAL, "sgc_upsample2x_occ: w, b and occ go together");
  if (C % 4 || ix <= 0 || iy <= 0 || iz <= 0 || (((uintptr_t)vol | (uintptr_t)up | (uintptr_t)w_or_null) & 15))
    return set_error(SGC_EUNSUP, "sgc_upsample2x_occ: C %% 4 == 0 and 16-byte aligned pointers required");
  int G = 1;
  while (G < 64 && G < C / 4) G <<= 1;      // lanes per output voxel (power of two <= 64)
  const int64_t nvox = (int64_t)8 * ix * iy * iz;
  const int gpb = 256 / G;
  hipLaunchKernelGGL(upsample2x_occ_kernel, dim3(grid_for(nvox * G, 256)), dim3(256), 0, (hipStream_t)stream, vol,
                     w_or_null, b_or_null, up, occ_or_null, ix, iy, iz, C, G);
  (void)gpb;
  return check_launch("upsample2x_occ_kernel");
}

// Adjoint of the x2 trilinear upsample on NCDHW planes as a gather: one thread per input voxel collects its <= 4 x 4 x 4
// outputs (per axis: 2i-1, 2i, 2i+1, 2i+2) with the weights the forward's index rule gives them.  torch's backward
// scatters 8 float atomics per output element (5.2 ms per config-2 training step for the two upsamples of the path).
__device__ __forceinline__ float up2_weight(int o, int n, int i) {     // weight of input i in output o along one axis
  float t = 0.5f * ((float)o + 0.5f) - 0.5f;
  t = t < 0.f ? 0.f : t;
  const int i0 = (int)t;
  const int i1 = i0 + (i0 < n - 1 ? 1 : 0);
  const float l1 = t - (float)i0;
  return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void upsample2x_backward_kernel(const float *__restrict__ go, float *__restrict__ gi, int C,
                                                                  int X, int Y, int Z) {
  const int64_t vin = (int64_t)X * Y * Z, total = vin * C;
  const int OY = 2 * Y, OZ = 2 * Z;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e / vin);
    const int64_t v = e - (int64_t)c * vin;
    const int z = (int)(v % Z), y = (int)((v / Z) % Y), x = (int)(v / ((int64_t)Z * Y));
    const float *plane = go + (int64_t)c * vin * 8;
    float acc = 0.f;
    for (int ox = max(2 * x - 1, 0); ox <= min(2 * x + 2, 2 * X - 1); ++ox) {
      const float wx = up2_weight(ox, X, x);
      for (int oy = max(2 * y - 1, 0); oy <= min(2 * y + 2, OY - 1); ++oy) {
        const float wxy = wx * up2_weight(oy, Y, y);
        const float *line = plane + ((int64_t)ox * OY + oy) * OZ;
        for (int oz = max(2 * z - 1, 0); oz <= min(2 * z + 2, OZ - 1); ++oz) acc += wxy * up2_weight(oz, Z, z) * line[oz];
      }
    }
    gi[e] = acc;
  }
}

extern "C" int sgc_upsample2x_backward(const float *grad_out, float *grad_in, int C, int X, int Y, int Z, sgc_stream_t stream) {
  if (C <= 0 || X <= 0 || Y <= 0 || Z <= 0) return SGC_OK;
  if (!grad_out || !grad_in) return set_error(SGC_EINVAL, "sgc_upsample2x_backward: null pointer");
  const int64_t total = (int64_t)C * X * Y * Z;
  hipLaunchKernelGGL(upsample2x_backward_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, grad_out,
                     grad_in, C, X, Y, Z);
  return check_launch("upsample2x_backward_kernel");
}

extern "C" int sgc_scatter_add_rows(const float *rows, const int64_t *idx, float *vol, int n, int C, sgc_stream_t stream) {
  if (!rows || !idx || !vol) return set_error(SGC_EINVAL, "sgc_scatter_add_rows: null pointer");
  if (C % 4 || (((uintptr_t)rows | (uintptr_t)vol) & 15)) return set_error(SGC_EUNSUP, "sgc_scatter_add_rows: C %% 4 == 0 required");
  if (n <= 0) return SGC_OK;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for((int64_t)n * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, rows, idx, vol, n, C / 4);
  return check_launch("scatter_add_rows_kernel");
}
