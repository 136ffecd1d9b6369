// Row-wise glue of the coarse-to-fine head that used to run as library kernels inside the scene graphs:
//
//   sgc_topk_select   `torch.topk(occ, k)` + `scatter_` mask + `nonzero` of AdaptiveSparseHead (topk_wo_grad,
//                     mmdet3d_plugin/models/im2voxel/AdaptiveSparseHead.py:9-13,74; DenseHead.py:66; get_valid :95-98):
//                     the reference needs the SET of the k highest occupancy scores, then its ascending index list.
//                     One workgroup: exact k-th largest value by a 4-pass radix select over the float bits, then one
//                     ordered compaction.  Ties at the cut are broken by the LOWEST flat voxel index (torch.topk leaves
//                     that order implementation-defined; voxels no camera sees carry bit-identical scores, so ties are
//                     real) -- the same rule as the oracle, so both sides select the same voxels whenever their scores
//                     agree bit for bit.  Replaces gatherTopK + 2 radix sorts + merges + scatter + 3 elementwise
//                     kernels (~150 us per scene at config 2) by one launch; candidate sets of more than 32 768 scores go
//                     through the many-workgroup form further down (same outputs).
//   sgc_layer_norm_rows  nn.LayerNorm(C) over rows (the two norms of VoxFormerLayer, TU/encoder.py:311-338 via
//                     build_norm_layer(dict(type='LN')), TU/custom_base_transformer_layer.py:153-156): one wave per
//                     row, two-pass mean / biased variance in registers, rsqrt(var + eps), affine.
#include "common.hpp"

namespace sgc {

// order-preserving map float -> uint32 (larger float <=> larger key); -0.0 < +0.0, NaNs sort above +inf like torch.topk
__device__ __forceinline__ uint32_t float_key(float f) {
  if (f != f) return 0xffffffffu;           // NaN: treated as the largest value (torch.topk does the same)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// KPT > 0: the workgroup keeps all keys in registers (n <= 1024 * KPT): ONE pass over global memory with every load of a
// thread in flight together; the first version re-read the scores in each of its 5 loops, one dependent ~1 us load per
// iteration (76 us per launch at n = 25 600).  KPT == 0: any n, keys re-read from global memory in batches of 8.
template <int KPT>
__global__ __launch_bounds__(1024) void topk_select_kernel(const float *__restrict__ score, int n, int k,
                                                           int64_t *__restrict__ idx_out, int64_t *__restrict__ valid_out,
                                                           float *__restrict__ mask_out) {
  __shared__ int hist[256];
  __shared__ int wave_sums[16];
  __shared__ uint32_t s_prefix;
  __shared__ int s_remaining;
  const int tid = threadIdx.x;
  constexpr int B = KPT > 0 ? KPT : 8;          // keys per thread per batch
  uint32_t keys[B];
  const int n_batches = KPT > 0 ? 1 : (n + 1024 * B - 1) / (1024 * B);
  auto load_batch = [&](int bi) {               // element j of the batch: i = (bi * B + j) * 1024 + tid (coalesced)
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int i = (bi * B + j) * 1024 + tid;
      keys[j] = i < n ? float_key(score[i]) : 0u;
    }
  };
  if (KPT > 0) load_batch(0);
  if (tid == 0) { s_prefix = 0; s_remaining = k; }
  // ---- radix select, most significant byte first: after pass p the top (p+1) bytes of the k-th largest key are known
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = tid; i < 256; i += 1024) hist[i] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    const uint32_t hi_mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int bi = 0; bi < n_batches; ++bi) {
      if (KPT == 0) load_batch(bi);
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const int i = (bi * B + j) * 1024 + tid;
        if ((bi * B + j) * 1024 >= n) break;                  // wave-uniform
        const uint32_t key = keys[j];
        bool act = i < n && (key & hi_mask) == prefix;
        const unsigned digit = (key >> shift) & 255u;
        // occupancy scores share their leading bytes, so most lanes of a wave hit ONE bin (a same-address LDS atomic runs
        // lane by lane): the wave's three most common digits are counted by ballot and added once, the rest individually
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const unsigned long long rem = __ballot(act);
          if (!rem) break;
          const int l = __ffsll((long long)rem) - 1;
          const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)digit, l);   // l is wave-uniform: a VALU read, no LDS round trip
          const unsigned long long same = __ballot(act && digit == d0);
          if ((int)(threadIdx.x & 63) == l) atomicAdd(&hist[d0], __popcll(same));
          if (digit == d0) act = false;
        }
        if (act) atomicAdd(&hist[digit], 1);
      }
    }
    __syncthreads();
    // which digit holds the k-th element: suffix sums of the histogram from the top, in parallel over 256 threads (a serial
    // walk by one thread was ~450 dependent LDS round trips per launch -- most of the first version's 60-70 us)
    {
      const int rem = s_remaining;
      int v = 0, incl = 0;
      if (tid < 256) {
        v = hist[255 - tid];
        incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o);
          if ((tid & 63) >= o) incl += t;
        }
        if ((tid & 63) == 63) wave_sums[tid >> 6] = incl;
      }
      __syncthreads();
      if (tid < 256) {
        for (int w = 0; w < (tid >> 6); ++w) incl += wave_sums[w];
        const int excl = incl - v;
        if (excl < rem && rem <= incl) {          // exactly one thread: the digit whose bin contains the k-th element
          s_prefix = prefix | ((uint32_t)(255 - tid) << shift);
          s_remaining = rem - excl;               // how many elements with this prefix are still to be taken
        }
      }
    }
    __syncthreads();
  }
  const uint32_t thr = s_prefix;                // key of the k-th largest score
  const int need_eq = s_remaining;              // elements equal to it that belong to the selection (lowest indices first)
  // ---- ordered compaction: ascending flat index, ties by lowest index.  Per chunk of 1024 consecutive elements ONE
  //      packed scan over the 16 waves: (elements above the cut) << 16 | (elements equal to it); inside a wave the ranks are
  //      popcounts of ballots.  Equal elements are taken in index order until need_eq is used up, so the output position is
  //      out_base + greater_before + min(eq_rank, need_eq) - min(eq_base, need_eq).
  __shared__ int wave_cnt[2][16];
  const int lane = tid & 63, wid = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int out_base = 0, eq_base = 0, parity = 0;
  for (int bi = 0; bi < n_batches; ++bi) {
    if (KPT == 0) load_batch(bi);
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int i = (bi * B + j) * 1024 + tid;
      if ((bi * B + j) * 1024 >= n) break;                    // wave- and block-uniform
      const uint32_t key = keys[j];
      const bool is_gt = i < n && key > thr, is_eq = i < n && key == thr;
      const unsigned long long bg = __ballot(is_gt), be = __ballot(is_eq);
      if (lane == 0) wave_cnt[parity][wid] = (__popcll(bg) << 16) | __popcll(be);
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const int c = wave_cnt[parity][w];
        if (w < wid) before += c;
        total += c;
      }
      parity ^= 1;                                            // the other buffer is free: every wave has passed the barrier since
      const int gt_before = (before >> 16) + __popcll(bg & lt);
      const int eq_rank = eq_base + (before & 0xffff) + __popcll(be & lt);
      const bool sel = is_gt || (is_eq && eq_rank < need_eq);
      if (sel) idx_out[out_base + gt_before + min(eq_rank, need_eq) - min(eq_base, need_eq)] = i;
      if (i < n) {
        if (valid_out) valid_out[i] = sel;
        if (mask_out) mask_out[i] = sel ? 1.f : 0.f;
      }
      const int eq_tot = total & 0xffff;
      out_base += (total >> 16) + min(eq_base + eq_tot, need_eq) - min(eq_base, need_eq);
      eq_base += eq_tot;
    }
  }
}

// ---- many-workgroup form for large candidate sets (config 4 / 5: 204 800 voxels; one workgroup took ~500 us there) ----
// Same selection, same tie rule, same outputs.  Four histogram launches (one per key byte, most significant first), one
// counting launch and one ordered-compaction launch; workgroup w owns the 4096 consecutive candidates [4096 w, 4096 w + 4096).
// The digit picked after pass q is a pure function of hist[0..q], which is complete when the next launch starts, so every
// workgroup re-derives (prefix, remaining) itself: no single-workgroup "pick" launches and no device-side state hand-over.
// workspace: int hist[4][256], then int cnt[G][2] = (elements above the cut, elements equal to it) per workgroup.
constexpr int TK_T = 256, TK_ITEMS = 16, TK_CHUNK = TK_T * TK_ITEMS;

__global__ void topk_init_kernel(int *__restrict__ ws, int n_ints) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ints; i += gridDim.x * blockDim.x) ws[i] = 0;
}

// (prefix, remaining) after the first `passes` bytes; all TK_T threads call it, hist_g is complete for those passes
__device__ __forceinline__ void topk_resolve(const int *__restrict__ hist_g, int passes, int k, uint32_t &prefix_out, int &rem_out,
                                             int *s_wave, uint32_t *s_pref, int *s_rem) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0;
  int rem = k;
  for (int q = 0; q < passes; ++q) {
    const int shift = 24 - 8 * q;
    const int v = hist_g[q * 256 + 255 - tid];
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if ((tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); ++w) incl += s_wave[w];
    const int excl = incl - v;
    if (excl < rem && rem <= incl) {            // exactly one thread
      *s_pref = prefix | ((uint32_t)(255 - tid) << shift);
      *s_rem = rem - excl;
    }
    __syncthreads();
    prefix = *s_pref;
    rem = *s_rem;
    __syncthreads();
  }
  prefix_out = prefix;
  rem_out = rem;
}

__global__ __launch_bounds__(TK_T) void topk_hist_kernel(const float *__restrict__ score, int n, int k, int pass, int *__restrict__ ws) {
  __shared__ int hist[256];
  __shared__ int s_wave[4];
  __shared__ uint32_t s_pref;
  __shared__ int s_rem;
  const int tid = threadIdx.x;
  hist[tid] = 0;
  uint32_t prefix;
  int rem;
  topk_resolve(ws, pass, k, prefix, rem, s_wave, &s_pref, &s_rem);
  (void)rem;
  __syncthreads();                                 // the zeroed histogram (pass 0 resolves nothing and has no barrier)
  const int shift = 24 - 8 * pass;
  const uint32_t hi_mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
  const int base = blockIdx.x * TK_CHUNK;
  uint32_t keys[TK_ITEMS];
#pragma unroll
  for (int j = 0; j < TK_ITEMS; ++j) {
    const int i = base + j * TK_T + tid;
    keys[j] = i < n ? float_key(score[i]) : 0u;
  }
#pragma unroll
  for (int j = 0; j < TK_ITEMS; ++j) {
    const int i = base + j * TK_T + tid;
    const uint32_t key = keys[j];
    bool act = i < n && (key & hi_mask) == prefix;
    const unsigned digit = (key >> shift) & 255u;
#pragma unroll
    for (int r = 0; r < 3; ++r) {               // a wave's most common digits are counted by ballot (see the one-workgroup form)
      const unsigned long long remb = __ballot(act);
      if (!remb) break;
      const int l = __ffsll((long long)remb) - 1;
      const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)digit, l);
      const unsigned long long same = __ballot(act && digit == d0);
      if ((int)(tid & 63) == l) atomicAdd(&hist[d0], __popcll(same));
      if (digit == d0) act = false;
    }
    if (act) atomicAdd(&hist[digit], 1);
  }
  __syncthreads();
  const int c = hist[tid];
  if (c) atomicAdd(&ws[pass * 256 + tid], c);
}

__global__ __launch_bounds__(TK_T) void topk_count_kernel(const float *__restrict__ score, int n, int k, int *__restrict__ ws) {
  __shared__ int s_wave[4];
  __shared__ uint32_t s_pref;
  __shared__ int s_rem;
  __shared__ int s_cnt[2];
  const int tid = threadIdx.x;
  if (tid < 2) s_cnt[tid] = 0;
  uint32_t thr;
  int need_eq;
  topk_resolve(ws, 4, k, thr, need_eq, s_wave, &s_pref, &s_rem);
  (void)need_eq;
  const int base = blockIdx.x * TK_CHUNK;
  int gt = 0, eq = 0;
#pragma unroll
  for (int j = 0; j < TK_ITEMS; ++j) {
    const int i = base + j * TK_T + tid;
    if (i < n) {
      const uint32_t key = float_key(score[i]);
      gt += key > thr;
      eq += key == thr;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { gt += __shfl_xor(gt, o); eq += __shfl_xor(eq, o); }
  if ((tid & 63) == 0) { atomicAdd(&s_cnt[0], gt); atomicAdd(&s_cnt[1], eq); }
  __syncthreads();
  if (tid < 2) ws[4 * 256 + blockIdx.x * 2 + tid] = s_cnt[tid];
}

__global__ __launch_bounds__(TK_T) void topk_compact_kernel(const float *__restrict__ score, int n, int k, const int *__restrict__ ws,
                                                            int64_t *__restrict__ idx_out, int64_t *__restrict__ valid_out,
                                                            float *__restrict__ mask_out) {
  __shared__ int s_wave[4];
  __shared__ uint32_t s_pref;
  __shared__ int s_rem;
  __shared__ int s_before[2];
  __shared__ int wave_cnt[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid < 2) s_before[tid] = 0;
  uint32_t thr;
  int need_eq;
  topk_resolve(ws, 4, k, thr, need_eq, s_wave, &s_pref, &s_rem);
  // candidates above / equal to the cut in the workgroups before this one
  int gtb = 0, eqb = 0;
  for (int w = tid; w < (int)blockIdx.x; w += TK_T) { gtb += ws[4 * 256 + 2 * w]; eqb += ws[4 * 256 + 2 * w + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { gtb += __shfl_xor(gtb, o); eqb += __shfl_xor(eqb, o); }
  if (lane == 0) { atomicAdd(&s_before[0], gtb); atomicAdd(&s_before[1], eqb); }
  __syncthreads();
  int eq_base = s_before[1];
  int out_base = s_before[0] + min(eq_base, need_eq);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int base = blockIdx.x * TK_CHUNK;
  int parity = 0;
#pragma unroll 1
  for (int j = 0; j < TK_ITEMS; ++j) {
    if (base + j * TK_T >= n) break;                          // block-uniform
    const int i = base + j * TK_T + tid;
    const uint32_t key = i < n ? float_key(score[i]) : 0u;
    const bool is_gt = i < n && key > thr, is_eq = i < n && key == thr;
    const unsigned long long bg = __ballot(is_gt), be = __ballot(is_eq);
    if (lane == 0) wave_cnt[parity][wid] = (__popcll(bg) << 16) | __popcll(be);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = wave_cnt[parity][w];
      if (w < wid) before += c;
      total += c;
    }
    parity ^= 1;
    const int gt_before = (before >> 16) + __popcll(bg & lt);
    const int eq_rank = eq_base + (before & 0xffff) + __popcll(be & lt);
    const bool sel = is_gt || (is_eq && eq_rank < need_eq);
    if (sel) idx_out[out_base + gt_before + min(eq_rank, need_eq) - min(eq_base, need_eq)] = i;
    if (i < n) {
      if (valid_out) valid_out[i] = sel;
      if (mask_out) mask_out[i] = sel ? 1.f : 0.f;
    }
    const int eq_tot = total & 0xffff;
    out_base += (total >> 16) + min(eq_base + eq_tot, need_eq) - min(eq_base, need_eq);
    eq_base += eq_tot;
  }
}

template <int VPL>     // C = 64 * VPL channels, VPL floats per lane
__global__ __launch_bounds__(256) void layer_norm_rows_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float eps, float *__restrict__ y,
                                                              const int32_t *__restrict__ rows_dev, int rows_cap) {
  constexpr int C = 64 * VPL;
  const int rows = rows_dev ? min(rows_cap, *rows_dev) : rows_cap;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[VPL];
  ln_row_load<VPL>(x + (int64_t)row * C, lane, v);
  float mean, rstd;
  ln_row_stats<VPL>(v, eps, mean, rstd);
  float o[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = ln_channel<VPL>(lane, j);
    o[j] = ln_apply(v[j], mean, rstd, gamma[c], beta[c]);
  }
  ln_row_store<VPL>(y + (int64_t)row * C, lane, o);
}

// generic width (any C): one wave per row, strided loops
__global__ __launch_bounds__(256) void layer_norm_rows_generic_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                                      const float *__restrict__ beta, float eps,
                                                                      float *__restrict__ y, const int32_t *__restrict__ rows_dev,
                                                                      int rows_cap, int C) {
  const int rows = rows_dev ? min(rows_cap, *rows_dev) : rows_cap;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *xr = x + (int64_t)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
  for (int c = lane; c < C; c += 64) y[(int64_t)row * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

}  // namespace sgc

using namespace sgc;

// candidate sets of at least this size use the many-workgroup form (7 launches) when a workspace is given.  Round 5: 32 769 -- up to 32 768
// candidates ONE workgroup with the keys in registers is as fast with one or four scenes in flight (573 / 566 / 579 against 574 / 567 / 564
// scenes/s, 388 against 389 with one stream: profiles/r05_topk_ab.txt) and is one launch; it was 16 384 since round 3
namespace sgc { int g_tune_topk_multi_min = 32769; }

extern "C" int64_t sgc_topk_select_workspace_bytes(int n) {
  return n > 0 ? (int64_t)(4 * 256 + 2 * ceil_div(n, TK_CHUNK)) * (int64_t)sizeof(int) : 0;
}

extern "C" int sgc_topk_select_ws(const float *score, int n, int k, int64_t *idx_out, int64_t *valid_or_null, float *mask_or_null,
                                  void *workspace_or_null, int64_t workspace_bytes, sgc_stream_t stream) {
  if (!score || !idx_out) return set_error(SGC_EINVAL, "sgc_topk_select: null pointer");
  if (n <= 0 || k <= 0 || k > n) return set_error(SGC_EINVAL, "sgc_topk_select: need 0 < k <= n (k = %d, n = %d)", k, n);
  if (!workspace_or_null || workspace_bytes < sgc_topk_select_workspace_bytes(n) || n < g_tune_topk_multi_min)
    return sgc_topk_select(score, n, k, idx_out, valid_or_null, mask_or_null, stream);
  hipStream_t st = (hipStream_t)stream;
  int *ws = reinterpret_cast<int *>(workspace_or_null);
  const int G = ceil_div(n, TK_CHUNK);
  hipLaunchKernelGGL(topk_init_kernel, dim3(4), dim3(256), 0, st, ws, 4 * 256 + 2 * G);
  for (int pass = 0; pass < 4; ++pass) hipLaunchKernelGGL(topk_hist_kernel, dim3(G), dim3(TK_T), 0, st, score, n, k, pass, ws);
  hipLaunchKernelGGL(topk_count_kernel, dim3(G), dim3(TK_T), 0, st, score, n, k, ws);
  hipLaunchKernelGGL(topk_compact_kernel, dim3(G), dim3(TK_T), 0, st, score, n, k, ws, idx_out, valid_or_null, mask_or_null);
  return check_launch("topk_compact_kernel");
}

extern "C" int sgc_topk_select(const float *score, int n, int k, int64_t *idx_out, int64_t *valid_or_null,
                               float *mask_or_null, sgc_stream_t stream) {
  if (!score || !idx_out) return set_error(SGC_EINVAL, "sgc_topk_select: null pointer");
  if (n <= 0 || k <= 0 || k > n) return set_error(SGC_EINVAL, "sgc_topk_select: need 0 < k <= n (k = %d, n = %d)", k, n);
  if (n <= 1024 * 8)
    hipLaunchKernelGGL(topk_select_kernel<8>, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, n, k, idx_out, valid_or_null, mask_or_null);
  else if (n <= 1024 * 32)
    hipLaunchKernelGGL(topk_select_kernel<32>, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, n, k, idx_out, valid_or_null, mask_or_null);
  else
    hipLaunchKernelGGL(topk_select_kernel<0>, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, n, k, idx_out, valid_or_null, mask_or_null);
  return check_launch("topk_select_kernel");
}

extern "C" int sgc_layer_norm_rows(const float *x, const float *gamma, const float *beta, float eps, float *y,
                                   const int32_t *rows_dev_or_null, int rows_cap, int C, sgc_stream_t stream) {
  if (!x || !gamma || !beta || !y) return set_error(SGC_EINVAL, "sgc_layer_norm_rows: null pointer");
  if (rows_cap <= 0) return SGC_OK;
  if (C <= 0) return set_error(SGC_EINVAL, "sgc_layer_norm_rows: bad C");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ceil_div(rows_cap, 4)), block(256);
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) |
                    reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
  if (C == 256 && al)
    hipLaunchKernelGGL(layer_norm_rows_kernel<4>, grid, block, 0, st, x, gamma, beta, eps, y, rows_dev_or_null, rows_cap);
  else if (C == 128)
    hipLaunchKernelGGL(layer_norm_rows_kernel<2>, grid, block, 0, st, x, gamma, beta, eps, y, rows_dev_or_null, rows_cap);
  else if (C == 64)
    hipLaunchKernelGGL(layer_norm_rows_kernel<1>, grid, block, 0, st, x, gamma, beta, eps, y, rows_dev_or_null, rows_cap);
  else
    hipLaunchKernelGGL(layer_norm_rows_generic_kernel, grid, block, 0, st, x, gamma, beta, eps, y, rows_dev_or_null, rows_cap, C);
  return check_launch("layer_norm_rows_kernel");
}
