// Fused 3D deformable attention forward for gfx950 (wave64).
//
// Replaces the reference's two launches per call (ms_depth_score_sample_forward +
// wms_deform_attn_forward, TU/multi_scale_3ddeformable_attn_function.py:285-299) and the
// [B,Q,M,L,P,4] depth-score round trip through HBM by ONE kernel built in two phases
// per tile of TP items (an item = one (camera, query)):
//
//   phase 1  one lane per SAMPLE (item, head, level, point): reads its location /
//            attention weight (or the raw Linear outputs, with the softmax over the
//            points done by wave shuffles), evaluates the depth score from the
//            un-replicated depth map, and leaves a 32-byte descriptor in LDS:
//            4 corner weights (bilinear x depth score x attention) + 4 pixel indices.
//            No lane repeats another lane's scalar work (the reference re-reads the
//            same loc/attn/4 scores in every one of its Cm channel threads).
//   phase 2  one lane per 4 CHANNELS of a head: reads descriptors (LDS broadcast),
//            gathers the corner rows as float4 (a wave covers M*Cm*4 B of contiguous
//            channels-last row per corner) and accumulates in registers.
//
// Tiles are dealt to XCDs in contiguous chunks (common.hpp: xcd_tile) because items are
// camera-major: neighbouring tiles hit the same camera's value map in one L2.
#include <string.h>

#include "common.hpp"

namespace sgc {

enum Mode { kBatch = 0, kPairsDeform = 1, kPairsGeom = 2 };

struct FwdParams {
  // maps
  const float *value;       // [B,S,M,Cm]
  const float *dist;        // [B,S,dist_heads,D]
  const float *dist_pairs;  // optional pair-interleaved depth [N,H,W+1,D,2] (pairs modes), see make_sample_dp
  const int64_t *shapes3;   // [L,3] (batch mode)  -- device
  const int64_t *lsi;       // [L]
  // batch-mode sample source
  const float *loc3;        // [items,M,L,P,3]
  const float *attn;        // [items,M,L,P] or null (= 1)
  // pairs-mode sample source
  const float *ref_cam;     // [N,Nq,3]
  const float *raw;         // [pairs, M*P*4]
  const int32_t *pair_cam, *pair_q, *totals;
  const int32_t *item_batch; // batch mode, optional: per-item index of the map it samples (item lists: sgc_dfa3d_forward_items)
  // outputs
  float *out;               // [items, M*Cm]
  float *score;             // optional [items,M,L,P,4]
  int S, M, Cm, D, dist_heads, L, Q, P, Nq;
  int H, W;                 // pairs mode (single level)
  int64_t value_bytes;      // size of the value map in bytes (pairs modes)
  int zero_row;             // >= 0: row index (in units of M*Cm floats from `value`) of an all-zero row the caller
                            // appended to the map; corners outside the image are pointed at it (no select needed)
  int n_items;              // < 0: read totals[0]
  int TP;                   // items per tile
};

template <int MODE, int VEC>
__global__ __launch_bounds__(256) void dfa3d_fwd_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SPI = p.M * p.L * p.P;  // samples per item
  float4 *lds_w = reinterpret_cast<float4 *>(smem_raw);
  int4 *lds_o = reinterpret_cast<int4 *>(smem_raw + (size_t)p.TP * SPI * sizeof(float4));
  int *lds_b = reinterpret_cast<int *>(smem_raw + (size_t)p.TP * SPI * (sizeof(float4) + sizeof(int4)));

  const int n_items = p.n_items >= 0 ? p.n_items : p.totals[0];
  const int ntiles = (n_items + p.TP - 1) / p.TP;
  if ((int)blockIdx.x >= ntiles) return;
  const int tile = xcd_tile(blockIdx.x, ntiles);
  const int item0 = tile * p.TP;
  const int tid = threadIdx.x;
  const int MC = p.M * p.Cm;

  // ---------------- phase 1: sample descriptors ----------------
  const int nsamp = p.TP * SPI;
  for (int t = tid; t < ((nsamp + kWave - 1) / kWave) * kWave; t += blockDim.x) {
    const bool in_tile = t < nsamp;
    const int tt = in_tile ? t : nsamp - 1;
    const int il = tt / SPI;          // item inside the tile
    const int r = tt - il * SPI;      // (m, l, p), p fastest
    int item = item0 + il;
    const bool live = in_tile && item < n_items;
    if (item >= n_items) item = n_items - 1;
    const int m = r / (p.L * p.P);
    const int lp = r - m * (p.L * p.P);
    const int l = lp / p.P;
    const int pt = lp - l * p.P;

    int b, H, W, D, lvl0;
    float x, y, z, aw;
    if (MODE == kBatch) {
      b = p.item_batch ? p.item_batch[item] : item / p.Q;
      H = (int)p.shapes3[l * 3]; W = (int)p.shapes3[l * 3 + 1]; D = (int)p.shapes3[l * 3 + 2];
      lvl0 = (int)p.lsi[l];
      const int64_t g = (int64_t)item * SPI + r;
      x = p.loc3[g * 3]; y = p.loc3[g * 3 + 1]; z = p.loc3[g * 3 + 2];
      aw = p.attn ? p.attn[g] : 1.f;
    } else {
      b = p.pair_cam[item];
      const int q = p.pair_q[item];
      H = p.H; W = p.W; D = p.D; lvl0 = 0;
      const float *rc = p.ref_cam + ((int64_t)b * p.Nq + q) * 3;
      x = rc[0]; y = rc[1]; z = rc[2];
      aw = 1.f;
      if (MODE == kPairsDeform) {
        const int MP = p.M * p.P;
        const float *rw = p.raw + (int64_t)item * MP * 4;
        const int mp = m * p.P + pt;
        // loc = ref + offset / (W,H,D): TU/deformable_cross_attention.py:445-455 (IEEE division)
        x = x + rw[mp * 2] / (float)W;
        y = y + rw[mp * 2 + 1] / (float)H;
        z = z + rw[MP * 2 + mp] / (float)D;
        // softmax over the P points of this head (:428-431): the P lanes are adjacent
        const float lg = rw[MP * 3 + mp];
        float mx = lg;
        for (int o = 1; o < p.P; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = expf(lg - mx);
        float sum = e;
        for (int o = 1; o < p.P; o <<= 1) sum += __shfl_xor(sum, o);
        aw = e / sum;
      }
    }
    const int dh = p.dist_heads == 1 ? 0 : m;
    const float *dpx = p.dist + (((int64_t)b * p.S + lvl0) * p.dist_heads + dh) * p.D;
    Sample sm;
    make_sample(sm, dpx, (int64_t)p.dist_heads * p.D, H, W, D, x, y, z, aw);
    if (in_tile) {
      lds_w[tt] = make_float4(sm.w[0], sm.w[1], sm.w[2], sm.w[3]);
      lds_o[tt] = make_int4(sm.off[0] + lvl0, sm.off[1] + lvl0, sm.off[2] + lvl0, sm.off[3] + lvl0);  // sign bit survives
      if (r == 0) lds_b[il] = b;
      if (live && p.score)
        reinterpret_cast<float4 *>(p.score)[(int64_t)item * SPI + r] =
            make_float4(sm.s[0], sm.s[1], sm.s[2], sm.s[3]);
    }
  }
  __syncthreads();

  // ---------------- phase 2: row gather ----------------
  const int CV = p.Cm / VEC;   // lanes per head
  const int LPI = p.M * CV;    // lanes per item
  const int LP = p.L * p.P;
  for (int idx = tid; idx < p.TP * LPI; idx += blockDim.x) {
    const int il = idx / LPI;
    const int item = item0 + il;
    if (item >= n_items) break;
    const int r = idx - il * LPI;
    const int m = r / CV;
    const int c0 = (r - m * CV) * VEC;
    const float *vbase = p.value + (int64_t)lds_b[il] * p.S * MC + m * p.Cm + c0;
    const int d0 = il * SPI + m * LP;
    if (VEC == 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < LP; ++s) {
        const float4 w = lds_w[d0 + s];
        const int4 o = lds_o[d0 + s];
        const float wk[4] = {w.x, w.y, w.z, w.w};
        const int ok[4] = {o.x, o.y, o.z, o.w};
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // unconditional load from the clamped index, value zeroed by a select
          v[k] = *reinterpret_cast<const float4 *>(vbase + (int64_t)off_index(ok[k]) * MC);
          if (ok[k] < 0) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc.x += wk[k] * v[k].x; acc.y += wk[k] * v[k].y;
          acc.z += wk[k] * v[k].z; acc.w += wk[k] * v[k].w;
        }
      }
      *reinterpret_cast<float4 *>(p.out + (int64_t)item * MC + m * p.Cm + c0) = acc;
    } else {
      float acc = 0.f;
      for (int s = 0; s < LP; ++s) {
        const float4 w = lds_w[d0 + s];
        const int4 o = lds_o[d0 + s];
        const float wk[4] = {w.x, w.y, w.z, w.w};
        const int ok[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float x = vbase[(int64_t)off_index(ok[k]) * MC];
          acc += wk[k] * (ok[k] < 0 ? 0.f : x);
        }
      }
      p.out[(int64_t)item * MC + m * p.Cm + c0] = acc;
    }
  }
}


// Wave-private variant for the pair-list modes (L = 1, SPI = M*P divides 64): every wave owns
// IPW = 64/SPI items end to end -- descriptors are produced and consumed by the same wave, so the
// only synchronisation is a wavefront-scope fence (LDS operations of one wave complete in order).
// Waves of a workgroup drift apart and the dependent-load chain of phase 1 (pair -> ref -> depth)
// of one wave overlaps the row gather of the others.  PT > 0 fixes the point count at compile time
// so the gather of one (item, head) is fully unrolled: all 4*PT row loads are in flight together.
// (forcing more waves per SIMD with __launch_bounds__(256, 6|8) spills the 16 in-flight rows to scratch:
//  2-3x slower, measured -- the kernel wants its 99 VGPRs and 4 waves/SIMD.)
template <int MODE, int PT, int MT = 0, int CMT = 0, int SPL = 1, bool ZR = false>
__global__ __launch_bounds__(256) void dfa3d_fwd_wave_kernel(const FwdParams p) {
  // SPL = samples per lane in phase 1: a wave owns SPL * 64 / SPI items.  Phase 1 is a chain of dependent
  // loads (pair -> reference point -> depth); with SPL > 1 the chains of SPL samples are issued together,
  // which is where its time goes (measured: phase 1 alone 45 us at SPL = 1, latency- not bandwidth-bound).
  __shared__ float4 lds_w[256 * SPL];
  __shared__ int4 lds_o[256 * SPL];
  __shared__ int lds_b[256 * SPL];
  // MT / CMT / PT > 0 pin heads / channels-per-head / points at compile time (the hot shapes M = 8,
  // P = 4, Cm = 32 | 16): every divide and modulo of the index arithmetic below folds to shifts.
  const int M = MT > 0 ? MT : p.M;
  const int P = PT > 0 ? PT : p.P;
  const int Cm = CMT > 0 ? CMT : p.Cm;
  const int SPI = M * P;
  const int IPW = SPL * kWave / SPI;         // items per wave
  const int n_items = p.n_items >= 0 ? p.n_items : p.totals[0];
  const int per_block = 4 * IPW;
  const int ntiles = (n_items + per_block - 1) / per_block;
  if ((int)blockIdx.x >= ntiles) return;
  const int tile = xcd_tile(blockIdx.x, ntiles);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int item0 = tile * per_block + wid * IPW;
  if (item0 >= n_items) return;              // whole wave idle (tail)
  const int MC = M * Cm;
  const float rW = 1.0f / (float)p.W, rH = 1.0f / (float)p.H, rD = 1.0f / (float)p.D;
  float4 *w_ = lds_w + wid * kWave * SPL;
  int4 *o_ = lds_o + wid * kWave * SPL;
  int *b_ = lds_b + wid * kWave * SPL;

  {  // ---- phase 1: lane = SPL samples (slot = j*64 + lane) ----
    int bb[SPL];
    float x[SPL], y[SPL], z[SPL], aw[SPL];
    float2 uv[SPL];
    float dzv[SPL], lgv[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {          // stage A: everything that only needs the item index
      const int slot = j * kWave + lane;
      const int il = slot / SPI, r = slot - il * SPI;
      int item = item0 + il;
      if (item >= n_items) item = n_items - 1;
      bb[j] = p.pair_cam[item];
      const int q = p.pair_q[item];
      const float *rc = p.ref_cam + ((int64_t)bb[j] * p.Nq + q) * 3;
      x[j] = rc[0]; y[j] = rc[1]; z[j] = rc[2];
      aw[j] = 1.f;
      if (MODE == kPairsDeform) {
        const int MP = M * P;
        const float *rw = p.raw + (int64_t)item * (MP * 4);
        uv[j] = *reinterpret_cast<const float2 *>(rw + r * 2);
        dzv[j] = rw[MP * 2 + r];
        lgv[j] = rw[MP * 3 + r];
      }
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j) {          // stage B: locations, softmax over the P adjacent lanes
      if (MODE == kPairsDeform) {
        // the reference's forms: offset / (W,H,D) with IEEE rounding (div_by_size), expf, a true division (see dfa3d_tile.hip)
        x[j] = x[j] + div_by_size(uv[j].x, (float)p.W, rW);
        y[j] = y[j] + div_by_size(uv[j].y, (float)p.H, rH);
        z[j] = z[j] + div_by_size(dzv[j], (float)p.D, rD);
        float mx = lgv[j];
#pragma unroll
        for (int o = 1; o < P; o <<= 1) mx = fmaxf(mx, lane_xor(mx, o));
        const float e = expf(lgv[j] - mx);
        float sum = e;
#pragma unroll
        for (int o = 1; o < P; o <<= 1) sum += lane_xor(sum, o);
        aw[j] = e / sum;
      }
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j) {          // stage C: depth loads + descriptors
      const int slot = j * kWave + lane;
      const int il = slot / SPI, r = slot - il * SPI;
      Sample sm;
      if (p.dist_pairs)
        make_sample_dp(sm, p.dist_pairs + (int64_t)bb[j] * p.H * (p.W + 1) * p.D * 2, p.H, p.W, p.D, x[j], y[j], z[j], aw[j]);
      else
        make_sample(sm, p.dist + (int64_t)bb[j] * p.S * p.D, (int64_t)p.D, p.H, p.W, p.D, x[j], y[j], z[j], aw[j]);
      w_[slot] = make_float4(sm.w[0], sm.w[1], sm.w[2], sm.w[3]);
      if (ZR) {   // global row index, outside corners -> the appended zero row
        const int cb = bb[j] * p.S;
        o_[slot] = make_int4(sm.off[0] < 0 ? p.zero_row : cb + sm.off[0], sm.off[1] < 0 ? p.zero_row : cb + sm.off[1],
                             sm.off[2] < 0 ? p.zero_row : cb + sm.off[2], sm.off[3] < 0 ? p.zero_row : cb + sm.off[3]);
      } else {
        o_[slot] = make_int4(sm.off[0], sm.off[1], sm.off[2], sm.off[3]);
      }
      if (r == 0) b_[il] = bb[j];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- phase 2: lane = 4 channels of a head ----
  const int CV = Cm / 4, LPI = M * CV;
  const int LP = P;
  for (int idx = lane; idx < IPW * LPI; idx += kWave) {
    const int il = idx / LPI;
    const int item = item0 + il;
    if (item >= n_items) break;
    const int r = idx - il * LPI;
    const int m = r / CV;
    const int c0 = (r - m * CV) * 4;
    // 32-bit byte offsets on the uniform base pointer (host checks N*S*M*Cm*4 < 2^32): the loads take the
    // saddr + voffset form and the per-corner address is one 32-bit mad instead of 64-bit vector arithmetic
    const char *vbytes = reinterpret_cast<const char *>(p.value);
    const unsigned rowb = ((ZR ? 0u : (unsigned)b_[il] * (unsigned)p.S * (unsigned)MC) + (unsigned)(m * Cm + c0)) * 4u;
    const unsigned rstride = (unsigned)MC * 4u;
    const float *vbase = p.value + (int64_t)b_[il] * p.S * MC + (m * Cm + c0);
    (void)vbase;
    const int d0 = il * SPI + m * LP;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PT > 0) {
      float4 w[PT > 0 ? PT : 1];
      int4 o[PT > 0 ? PT : 1];
      float4 v[PT > 0 ? PT : 1][4];
#pragma unroll
      for (int s = 0; s < PT; ++s) { w[s] = w_[d0 + s]; o[s] = o_[d0 + s]; }
#pragma unroll
      for (int s = 0; s < PT; ++s) {
        const int ok[4] = {o[s].x, o[s].y, o[s].z, o[s].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // no branch per load: clamped index + select (see Sample::off);
                                        // 32-bit element offset (host checks S*M*Cm < 2^31)
          v[s][k] = *reinterpret_cast<const float4 *>(vbytes + (rowb + (unsigned)(ZR ? ok[k] : off_index(ok[k])) * rstride));
          if (!ZR && ok[k] < 0) v[s][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int s = 0; s < PT; ++s) {
        const float wk[4] = {w[s].x, w[s].y, w[s].z, w[s].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc.x += wk[k] * v[s][k].x; acc.y += wk[k] * v[s][k].y;
          acc.z += wk[k] * v[s][k].z; acc.w += wk[k] * v[s][k].w;
        }
      }
    } else {
      for (int s = 0; s < LP; ++s) {
        const float4 w = w_[d0 + s];
        const int4 o = o_[d0 + s];
        const float wk[4] = {w.x, w.y, w.z, w.w};
        const int ok[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float4 v = *reinterpret_cast<const float4 *>(vbytes + (rowb + (unsigned)off_index(ok[k]) * rstride));
          if (ok[k] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
          acc.x += wk[k] * v.x; acc.y += wk[k] * v.y; acc.z += wk[k] * v.z; acc.w += wk[k] * v.w;
        }
      }
    }
    *reinterpret_cast<float4 *>(p.out + (int64_t)item * MC + (m * Cm + c0)) = acc;
  }
}

// (A software-pipelined persistent form -- rows of tile t, depth loads of t+1 and inputs of t+2 in flight
//  together -- was built and measured: 168-238 us against 138 us for the plain wave kernel on the same
//  inputs.  The gather is bound by instruction issue on the load path, not by exposed latency; what did
//  help is removing the exec-mask branch around every corner load, see Sample::off.)

int g_tune_fwd_variant = 1;
int g_tune_fwd_spl = 1;         // samples per lane in phase 1 of the wave kernel (1, 2, 4)
extern int g_tune_conv_waves;   // conv3d.hip: 4 or 8 waves per 128x128 tile
extern int g_tune_halo_min_cout;
extern int g_tune_halo_min_m;
extern int g_tune_halo_brick;
extern int g_tune_halo_stagger;
extern int g_tune_halo_small;
extern int g_tune_wgrad_halo;
extern int g_tune_view_group;
extern int g_tune_pq_depth;         // view_pool.hip
extern int g_tune_halo_wave_fix;     // conv3d.hip
extern int g_tune_compact2;          // project.hip: two-launch segment form of sgc_compact_pairs
extern int g_tune_halo_narrow;
extern int g_tune_split_target, g_tune_split_free, g_tune_split_min_steps, g_tune_split_max;
extern int g_tune_wgrad_waves;
extern int g_tune_rows_gemm, g_tune_rows_depth, g_tune_rows_diag, g_tune_igemm_xcd, g_tune_halo_2d, g_tune_rows_cu_pct, g_tune_halo_split_target;
extern int g_tune_topk_multi_min;
extern int g_tune_tile_nw;          // dfa3d_tile.hip
extern int g_tune_tile_depth_lds;
extern int g_tune_tile_diag;
extern int g_tune_tile_nbuf;
extern int g_tune_tile_hg;
extern int g_tune_tile_ds;
extern int g_tune_bwd_tile_diag;    // dfa3d_bwd_tile.hip
extern int g_tune_tile_xcd;
extern int g_tune_conv_halo;    // conv3d.hip: halo-resident kernel for the 3x3x3 stride-1 layers   // 0: block-barrier kernel, 1: wave-private kernel (when the shape allows)

static int pick_tp(int SPI, int LPI) {
  // enough samples to occupy the block in phase 1, bounded LDS (<= 32 KiB of descriptors)
  int tp = 256 / (SPI > 0 ? SPI : 1);
  if (tp < 1) tp = 1;
  if (SPI <= 2 && tp > 32) tp = 32;
  while (tp > 1 && (int64_t)tp * SPI * 32 > 32768) tp >>= 1;
  (void)LPI;
  return tp;
}

template <int MODE>
static int launch_fwd(FwdParams p, int grid_items, hipStream_t stream) {
  const int SPI = p.M * p.L * p.P;
  const bool vec4 = (p.Cm % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.value) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
  // (with one sample per item -- the geometry sample -- the block kernel's wider phase 2 wins: 64 vs 87 us)
  if (MODE != kBatch && g_tune_fwd_variant == 1 && vec4 && p.L == 1 && p.dist_heads == 1 && SPI >= 8 && SPI <= kWave &&
      p.value_bytes > 0 && p.value_bytes < (1ll << 32) &&   // 32-bit byte offsets in the wave kernel
      kWave % SPI == 0 && (p.P & (p.P - 1)) == 0 && ((reinterpret_cast<uintptr_t>(p.raw) & 7) == 0)) {
    const int per_block = 4 * (kWave / SPI);
    const int grid = ceil_div(grid_items, per_block);
    if (grid <= 0) return SGC_OK;
    if (MODE == kPairsDeform && p.P == 4 && p.M == 8 && p.Cm == 32 && g_tune_fwd_spl == 4)
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<kPairsDeform, 4, 8, 32, 4>), dim3(ceil_div(grid_items, 32)), dim3(256), 0, stream, p);
    else if (MODE == kPairsDeform && p.P == 4 && p.M == 8 && p.Cm == 32 && g_tune_fwd_spl == 2)
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<kPairsDeform, 4, 8, 32, 2>), dim3(ceil_div(grid_items, 16)), dim3(256), 0, stream, p);
    else if (MODE == kPairsDeform && p.P == 4 && p.M == 8 && p.Cm == 32 && p.zero_row >= 0)
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<kPairsDeform, 4, 8, 32, 1, true>), dim3(grid), dim3(256), 0, stream, p);
    else if (MODE == kPairsDeform && p.P == 4 && p.M == 8 && p.Cm == 32)
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<kPairsDeform, 4, 8, 32>), dim3(grid), dim3(256), 0, stream, p);
    else if (MODE == kPairsDeform && p.P == 4 && p.M == 8 && p.Cm == 16 && p.zero_row >= 0)
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<kPairsDeform, 4, 8, 16, 1, true>), dim3(grid), dim3(256), 0, stream, p);
    else if (MODE == kPairsDeform && p.P == 4 && p.M == 8 && p.Cm == 16)
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<kPairsDeform, 4, 8, 16>), dim3(grid), dim3(256), 0, stream, p);
    else if (MODE == kPairsDeform && p.P == 4)
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<kPairsDeform, 4>), dim3(grid), dim3(256), 0, stream, p);
    else if (MODE == kPairsGeom)
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<kPairsGeom, 1>), dim3(grid), dim3(256), 0, stream, p);
    else
      hipLaunchKernelGGL((dfa3d_fwd_wave_kernel<MODE == kBatch ? kPairsDeform : MODE, 0>), dim3(grid), dim3(256), 0, stream, p);
    return check_launch("dfa3d_fwd_wave_kernel");
  }
  p.TP = pick_tp(SPI, p.M * p.Cm / (vec4 ? 4 : 1));
  if ((int64_t)SPI * 32 > 60000) return set_error(SGC_EUNSUP, "M*L*P = %d samples per query exceed the LDS tile", SPI);
  const size_t smem = (size_t)p.TP * SPI * 32 + (size_t)p.TP * sizeof(int);
  const int grid = ceil_div(grid_items, p.TP);
  if (grid <= 0) return SGC_OK;
  if (vec4)
    hipLaunchKernelGGL((dfa3d_fwd_kernel<MODE, 4>), dim3(grid), dim3(256), smem, stream, p);
  else
    hipLaunchKernelGGL((dfa3d_fwd_kernel<MODE, 1>), dim3(grid), dim3(256), smem, stream, p);
  return check_launch("dfa3d_fwd_kernel");
}

// ---- split operators (dfa3D._ext compatibility): one thread per sample / per output ----
__global__ void depth_score_fwd_kernel(const float *__restrict__ dist, const int64_t *__restrict__ shapes3,
                                       const int64_t *__restrict__ lsi, const float *__restrict__ loc3,
                                       float *__restrict__ score, int64_t total, int S, int M, int D,
                                       int L, int Q, int P) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int pt = (int)(g % P);
    const int l = (int)((g / P) % L);
    const int m = (int)((g / ((int64_t)P * L)) % M);
    const int64_t bq = g / ((int64_t)P * L * M);
    const int b = (int)(bq / Q);
    (void)pt;
    const int H = (int)shapes3[l * 3], W = (int)shapes3[l * 3 + 1], Dl = (int)shapes3[l * 3 + 2];
    const float *dpx = dist + (((int64_t)b * S + lsi[l]) * M + m) * D;
    Sample sm;
    make_sample(sm, dpx, (int64_t)M * D, H, W, Dl, loc3[g * 3], loc3[g * 3 + 1], loc3[g * 3 + 2], 1.f);
    reinterpret_cast<float4 *>(score)[g] = make_float4(sm.s[0], sm.s[1], sm.s[2], sm.s[3]);
  }
}

// wms forward with precomputed scores: descriptors are rebuilt from (loc2, attn, score).
template <int VEC>
__global__ __launch_bounds__(256) void wms_fwd_kernel(const float *__restrict__ value,
                                                      const int64_t *__restrict__ shapes2,
                                                      const int64_t *__restrict__ lsi,
                                                      const float *__restrict__ loc2,
                                                      const float *__restrict__ attn,
                                                      const float *__restrict__ score, float *__restrict__ out,
                                                      int64_t total, int S, int M, int Cm, int L, int Q, int P) {
  const int CV = Cm / VEC;
  const int MC = M * Cm;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(g % CV) * VEC;
    const int64_t sidx = g / CV;  // (b,q,m)
    const int m = (int)(sidx % M);
    const int b = (int)(sidx / M / Q);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes2[l * 2], W = (int)shapes2[l * 2 + 1];
      const float *vb = value + ((int64_t)b * S + lsi[l]) * MC + m * Cm + c0;
      for (int pt = 0; pt < P; ++pt) {
        const int64_t s = (sidx * L + l) * P + pt;
        const float h_im = sample_coord(loc2[s * 2 + 1], (float)H), w_im = sample_coord(loc2[s * 2], (float)W);
        if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) continue;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const float4 sc = reinterpret_cast<const float4 *>(score)[s];
        const float aw = attn[s];
        const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w1 <= W - 1, h1 <= H - 1 && w0 >= 0,
                            h1 <= H - 1 && w1 <= W - 1};
        const int px[4] = {h0 * W + w0, h0 * W + w1, h1 * W + w0, h1 * W + w1};
        const float wk[4] = {hh * hw * sc.x, hh * lw * sc.y, lh * hw * sc.w, lh * lw * sc.z};
        float val[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) val[v] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!ok[k]) continue;
          const float *src = vb + (int64_t)px[k] * MC;
          if (VEC == 4) {
            const float4 x4 = *reinterpret_cast<const float4 *>(src);
            val[0] += wk[k] * x4.x; val[1 % VEC] += wk[k] * x4.y; val[2 % VEC] += wk[k] * x4.z; val[3 % VEC] += wk[k] * x4.w;
          } else {
            val[0] += wk[k] * src[0];
          }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += val[v] * aw;
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) out[sidx * Cm + c0 + v] = acc[v];
  }
}

// dp[n][h][wq][d] = (dist[n][h][wq-1][d] or 0, dist[n][h][wq][d] or 0), wq in [0, W]
__global__ void depth_pairs_kernel(const float *__restrict__ dist, float2 *__restrict__ dp, int64_t total, int H, int W, int D,
                                   int64_t cam_stride) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int wq = (int)((i / D) % (W + 1));
    const int64_t nh = i / ((int64_t)D * (W + 1));
    const float *row = dist + ((nh / H) * cam_stride + (nh % H) * W) * D;
    dp[i] = make_float2(wq > 0 ? row[(wq - 1) * D + d] : 0.f, wq < W ? row[wq * D + d] : 0.f);
  }
}

}  // namespace sgc

using namespace sgc;

extern "C" int sgc_depth_pairs(const float *dist, float *dp, int N, int H, int W, int D, int cam_stride_or_0,
                               sgc_stream_t stream) {
  if (!dist || !dp) return set_error(SGC_EINVAL, "sgc_depth_pairs: null pointer");
  const int64_t cam_stride = cam_stride_or_0 > 0 ? cam_stride_or_0 : (int64_t)H * W;
  if (cam_stride < (int64_t)H * W) return set_error(SGC_EINVAL, "sgc_depth_pairs: cam_stride < H*W");
  const int64_t total = (int64_t)N * H * (W + 1) * D;
  if (total <= 0) return set_error(SGC_EINVAL, "sgc_depth_pairs: bad size");
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(depth_pairs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dist,
                     reinterpret_cast<float2 *>(dp), total, H, W, D, cam_stride);
  return check_launch("depth_pairs_kernel");
}

extern "C" int sgc_set_tuning(const char *key, int value) {
  if (!key) return set_error(SGC_EINVAL, "sgc_set_tuning: null key");
  if (!strcmp(key, "fwd_variant")) { g_tune_fwd_variant = value; return SGC_OK; }
  if (!strcmp(key, "conv_waves")) { g_tune_conv_waves = value; return SGC_OK; }
  if (!strcmp(key, "fwd_spl")) { g_tune_fwd_spl = value; return SGC_OK; }
  if (!strcmp(key, "conv_halo")) { g_tune_conv_halo = value; return SGC_OK; }
  if (!strcmp(key, "halo_min_cout")) { g_tune_halo_min_cout = value; return SGC_OK; }
  if (!strcmp(key, "halo_min_m")) { g_tune_halo_min_m = value; return SGC_OK; }
  if (!strcmp(key, "halo_brick")) { g_tune_halo_brick = value; return SGC_OK; }
  if (!strcmp(key, "halo_stagger")) { g_tune_halo_stagger = value; return SGC_OK; }
  if (!strcmp(key, "halo_small")) { g_tune_halo_small = value; return SGC_OK; }
  if (!strcmp(key, "wgrad_halo")) { g_tune_wgrad_halo = value; return SGC_OK; }
  if (!strcmp(key, "view_group")) { g_tune_view_group = value; return SGC_OK; }
  if (!strcmp(key, "view_depth")) { g_tune_pq_depth = value; return SGC_OK; }
  if (!strcmp(key, "compact2")) { g_tune_compact2 = value; return SGC_OK; }
  if (!strcmp(key, "halo_wave_fix")) { g_tune_halo_wave_fix = value; return SGC_OK; }
  if (!strcmp(key, "halo_narrow")) { g_tune_halo_narrow = value; return SGC_OK; }
  if (!strcmp(key, "split_target")) { g_tune_split_target = value; return SGC_OK; }
  if (!strcmp(key, "split_free")) { g_tune_split_free = value; return SGC_OK; }
  if (!strcmp(key, "split_min_steps")) { g_tune_split_min_steps = value > 0 ? value : 1; return SGC_OK; }
  if (!strcmp(key, "split_max")) { g_tune_split_max = value > 0 ? value : 1; return SGC_OK; }
  if (!strcmp(key, "wgrad_waves")) { g_tune_wgrad_waves = value; return SGC_OK; }
  if (!strcmp(key, "rows_gemm")) { g_tune_rows_gemm = value; return SGC_OK; }
  if (!strcmp(key, "igemm_xcd")) { g_tune_igemm_xcd = value; return SGC_OK; }
  if (!strcmp(key, "halo_2d")) { g_tune_halo_2d = value; return SGC_OK; }
  if (!strcmp(key, "rows_cu_pct")) { g_tune_rows_cu_pct = value; return SGC_OK; }
  if (!strcmp(key, "halo_split_target")) { g_tune_halo_split_target = value; return SGC_OK; }
  if (!strcmp(key, "rows_depth")) { g_tune_rows_depth = value; return SGC_OK; }
  if (!strcmp(key, "rows_diag")) { g_tune_rows_diag = value; return SGC_OK; }      // inert without SGC_DIAG=1 (rows_gemm.hip)
  if (!strcmp(key, "topk_multi_min")) { g_tune_topk_multi_min = value; return SGC_OK; }
  if (!strcmp(key, "tile_nw")) { g_tune_tile_nw = value; return SGC_OK; }
  if (!strcmp(key, "tile_depth_lds")) { g_tune_tile_depth_lds = value; return SGC_OK; }
  if (!strcmp(key, "tile_diag")) { g_tune_tile_diag = value; return SGC_OK; }
  if (!strcmp(key, "tile_nbuf")) { g_tune_tile_nbuf = value; return SGC_OK; }
  if (!strcmp(key, "tile_hg")) { g_tune_tile_hg = value; return SGC_OK; }
  if (!strcmp(key, "tile_ds")) { g_tune_tile_ds = value; return SGC_OK; }
  if (!strcmp(key, "bwd_tile_diag")) { g_tune_bwd_tile_diag = value; return SGC_OK; }
  if (!strcmp(key, "tile_xcd")) { g_tune_tile_xcd = value; return SGC_OK; }
  return set_error(SGC_EINVAL, "sgc_set_tuning: unknown key %s", key);
}

extern "C" int sgc_dfa3d_forward(const float *value, const float *dist, const int64_t *shapes3,
                                 const int64_t *lsi, const float *loc3, const float *attn,
                                 float *out, float *score_or_null,
                                 int B, int S, int M, int Cm, int D, int dist_heads,
                                 int L, int Q, int P, sgc_stream_t stream) {
  if (!value || !dist || !shapes3 || !lsi || !loc3 || !out) return set_error(SGC_EINVAL, "sgc_dfa3d_forward: null pointer");
  if (B < 0 || S <= 0 || M <= 0 || Cm <= 0 || D <= 0 || L <= 0 || Q < 0 || P <= 0)
    return set_error(SGC_EINVAL, "sgc_dfa3d_forward: non-positive size");
  if (dist_heads != 1 && dist_heads != M) return set_error(SGC_EINVAL, "sgc_dfa3d_forward: dist_heads must be 1 or M");
  if ((int64_t)B * Q >= (1ll << 31)) return set_error(SGC_EUNSUP, "sgc_dfa3d_forward: B*Q >= 2^31");
  FwdParams p = {};
  p.value = value; p.dist = dist; p.shapes3 = shapes3; p.lsi = lsi; p.loc3 = loc3; p.attn = attn;
  p.out = out; p.score = score_or_null;
  p.S = S; p.M = M; p.Cm = Cm; p.D = D; p.dist_heads = dist_heads; p.L = L; p.Q = Q; p.P = P;
  p.n_items = B * Q;
  return launch_fwd<kBatch>(p, p.n_items, (hipStream_t)stream);
}

// Item-list form of the fused operator: item i samples map item_batch[i] (a camera) -- the padded [N, max_len] rebatch of
// the reference (TU/deformable_cross_attention.py:759-773) without its padding rows; used by the training path.
extern "C" int sgc_dfa3d_forward_items(const float *value, const float *dist, const int64_t *shapes3, const int64_t *lsi,
                                       const float *loc3, const float *attn_or_null, const int32_t *item_batch, float *out,
                                       float *score_or_null, int B, int S, int M, int Cm, int D, int dist_heads, int L,
                                       int n_items, int P, sgc_stream_t stream) {
  if (!value || !dist || !shapes3 || !lsi || !loc3 || !item_batch || !out) return set_error(SGC_EINVAL, "sgc_dfa3d_forward_items: null pointer");
  if (B <= 0 || S <= 0 || M <= 0 || Cm <= 0 || D <= 0 || L <= 0 || n_items < 0 || P <= 0)
    return set_error(SGC_EINVAL, "sgc_dfa3d_forward_items: non-positive size");
  if (dist_heads != 1 && dist_heads != M) return set_error(SGC_EINVAL, "sgc_dfa3d_forward_items: dist_heads must be 1 or M");
  if (n_items == 0) return SGC_OK;
  FwdParams p = {};
  p.value = value; p.dist = dist; p.shapes3 = shapes3; p.lsi = lsi; p.loc3 = loc3; p.attn = attn_or_null;
  p.item_batch = item_batch; p.out = out; p.score = score_or_null;
  p.S = S; p.M = M; p.Cm = Cm; p.D = D; p.dist_heads = dist_heads; p.L = L; p.Q = 1; p.P = P;
  p.n_items = n_items;
  return launch_fwd<kBatch>(p, p.n_items, (hipStream_t)stream);
}

extern "C" int sgc_pairs_geometry_sample(const float *feat, const float *dist, const float *ref_cam,
                                         const int32_t *pair_cam, const int32_t *pair_q,
                                         const int32_t *totals, float *out,
                                         int N, int Nq, int H, int W, int C, int D, int cam_stride_or_0,
                                         int n_pairs_or_neg, int cap, sgc_stream_t stream) {
  if (!feat || !dist || !ref_cam || !pair_cam || !pair_q || !out)
    return set_error(SGC_EINVAL, "sgc_pairs_geometry_sample: null pointer");
  if (n_pairs_or_neg < 0 && !totals) return set_error(SGC_EINVAL, "sgc_pairs_geometry_sample: totals required");
  if (n_pairs_or_neg > cap) return set_error(SGC_EINVAL, "sgc_pairs_geometry_sample: n_pairs > cap");
  if (N <= 0 || Nq <= 0 || H <= 0 || W <= 0 || C <= 0 || D <= 0) return set_error(SGC_EINVAL, "sgc_pairs_geometry_sample: bad size");
  FwdParams p = {};
  p.value = feat; p.dist = dist; p.ref_cam = ref_cam; p.pair_cam = pair_cam; p.pair_q = pair_q;
  p.totals = totals; p.out = out;
  if (cam_stride_or_0 > 0 && cam_stride_or_0 < H * W) return set_error(SGC_EINVAL, "sgc_pairs_geometry_sample: cam_stride < H*W");
  p.S = cam_stride_or_0 > 0 ? cam_stride_or_0 : H * W;       // pixels between consecutive cameras in feat / dist
  p.M = 1; p.Cm = C; p.D = D; p.dist_heads = 1; p.L = 1; p.Q = 1; p.P = 1; p.Nq = Nq;
  p.H = H; p.W = W; p.n_items = n_pairs_or_neg;
  return launch_fwd<kPairsGeom>(p, n_pairs_or_neg >= 0 ? n_pairs_or_neg : cap, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Geometry-aware sample FUSED with the Linear that consumes it (round 6).  `Grid_Sample_3D_Feature` (TU/deformable_cross_attention.py:
// 67-116) produces one C-channel row per visible pair, and the only reader of that row is the fused offsets / logits projection
// (`sampling_offsets` | `sampling_offsets_depth` | `attention_weights`, :417-436): [pairs, C] written and read back -- 2.2 GB per scene
// at BASELINE config 5.  Here a first launch leaves 32 bytes per pair (four corner weights = bilinear * depth score, four rows of the
// feature map) and the row GEMM builds its A operand from them while it stages a tile (rows_gemm.hip, GATHER form): the sampled row
// exists in registers only.  Same arithmetic as the two launches it replaces (the staged value is the fmas of the sample kernel, the
// GEMM is the same kernel): bit-identical results.
// ---------------------------------------------------------------------------------------------
namespace sgc {
bool rows_gemm_gather_supported(int K, int N, int64_t x_rows, int64_t rows);
int rows_gemm_gather_launch(const float *x, int64_t x_rows, const float *gw, const int32_t *go, const uint16_t *w_hi, const uint16_t *w_lo,
                            const float *shift, float *y, const int32_t *m_dev, int M, int K, int N, hipStream_t st);

__global__ __launch_bounds__(256) void pairs_geometry_desc_kernel(const float *__restrict__ dist, const float *__restrict__ ref_cam,
                                                                  const int32_t *__restrict__ pair_cam, const int32_t *__restrict__ pair_q,
                                                                  const int32_t *__restrict__ totals, int n_items, int Nq, int S, int H, int W,
                                                                  int D, float4 *__restrict__ gw, int4 *__restrict__ go) {
  const int n = n_items >= 0 ? n_items : totals[0];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int b = pair_cam[i], q = pair_q[i];
    const float *rc = ref_cam + ((int64_t)b * Nq + q) * 3;
    Sample sm;
    make_sample(sm, dist + (int64_t)b * S * D, D, H, W, D, rc[0], rc[1], rc[2], 1.f);
    gw[i] = make_float4(sm.w[0], sm.w[1], sm.w[2], sm.w[3]);       // 0 for a corner outside the map or gated off
    const int row0 = b * S;
    go[i] = make_int4(row0 + off_index(sm.off[0]), row0 + off_index(sm.off[1]), row0 + off_index(sm.off[2]), row0 + off_index(sm.off[3]));
  }
}
}  // namespace sgc

extern "C" int sgc_pairs_geometry_linear_supported(int C, int Cout, int N, int S) {
  return rows_gemm_gather_supported(C, Cout, (int64_t)N * S, 1) ? 1 : 0;
}
extern "C" int64_t sgc_pairs_geometry_linear_workspace_bytes(int cap) { return cap > 0 ? (int64_t)cap * 32 : 0; }

extern "C" int sgc_pairs_geometry_linear_bf16x3(const float *feat, const float *dist, const float *ref_cam, const int32_t *pair_cam,
                                                const int32_t *pair_q, const int32_t *totals, const uint16_t *w_hi, const uint16_t *w_lo,
                                                const float *shift_or_null, float *y, void *workspace, int N, int Nq, int H, int W, int C,
                                                int D, int Cout, int cam_stride_or_0, int n_pairs_or_neg, int cap, sgc_stream_t stream) {
  if (!feat || !dist || !ref_cam || !pair_cam || !pair_q || !w_hi || !w_lo || !y || !workspace)
    return set_error(SGC_EINVAL, "sgc_pairs_geometry_linear_bf16x3: null pointer");
  if (n_pairs_or_neg < 0 && !totals) return set_error(SGC_EINVAL, "sgc_pairs_geometry_linear_bf16x3: totals required");
  if (n_pairs_or_neg > cap) return set_error(SGC_EINVAL, "sgc_pairs_geometry_linear_bf16x3: n_pairs > cap");
  if (N <= 0 || Nq <= 0 || H <= 0 || W <= 0 || C <= 0 || D <= 0 || cap <= 0) return set_error(SGC_EINVAL, "sgc_pairs_geometry_linear_bf16x3: bad size");
  if (cam_stride_or_0 > 0 && cam_stride_or_0 < H * W) return set_error(SGC_EINVAL, "sgc_pairs_geometry_linear_bf16x3: cam_stride < H*W");
  const int S = cam_stride_or_0 > 0 ? cam_stride_or_0 : H * W;
  if (!rows_gemm_gather_supported(C, Cout, (int64_t)N * S, cap))
    return set_error(SGC_EUNSUP, "sgc_pairs_geometry_linear_bf16x3: needs C in {128, 256}, Cout == 128 and a map below 4 GiB (got C %d, Cout %d)", C, Cout);
  if (((uintptr_t)feat | (uintptr_t)y | (uintptr_t)workspace | (uintptr_t)w_hi | (uintptr_t)w_lo) & 15)
    return set_error(SGC_EINVAL, "sgc_pairs_geometry_linear_bf16x3: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  float4 *gw = reinterpret_cast<float4 *>(workspace);
  int4 *go = reinterpret_cast<int4 *>(gw + cap);
  const int items = n_pairs_or_neg >= 0 ? n_pairs_or_neg : cap;
  if (items == 0) return SGC_OK;
  const int grid = ceil_div(items, 256) < 4096 ? ceil_div(items, 256) : 4096;
  hipLaunchKernelGGL(pairs_geometry_desc_kernel, dim3(grid), dim3(256), 0, st, dist, ref_cam, pair_cam, pair_q, totals, n_pairs_or_neg, Nq, S, H, W,
                     D, gw, go);
  int rc = check_launch("pairs_geometry_desc_kernel");
  if (rc) return rc;
  return rows_gemm_gather_launch(feat, (int64_t)N * S, reinterpret_cast<const float *>(gw), reinterpret_cast<const int32_t *>(go), w_hi, w_lo,
                                 shift_or_null, y, n_pairs_or_neg >= 0 ? nullptr : totals, items, C, Cout, st);
}

extern "C" int sgc_pairs_deform_gather(const float *value, const float *dist, const float *dist_pairs_or_null,
                                       const float *ref_cam,
                                       const float *raw, const int32_t *pair_cam, const int32_t *pair_q,
                                       const int32_t *totals, float *out,
                                       int N, int Nq, int H, int W, int M, int Cm, int D, int P, int cam_stride_or_0,
                                       int value_has_zero_row, int n_pairs_or_neg, int cap, sgc_stream_t stream) {
  if (!value || !dist || !ref_cam || !raw || !pair_cam || !pair_q || !out)
    return set_error(SGC_EINVAL, "sgc_pairs_deform_gather: null pointer");
  if (n_pairs_or_neg < 0 && !totals) return set_error(SGC_EINVAL, "sgc_pairs_deform_gather: totals required");
  if (n_pairs_or_neg > cap) return set_error(SGC_EINVAL, "sgc_pairs_deform_gather: n_pairs > cap");
  if (N <= 0 || Nq <= 0 || H <= 0 || W <= 0 || M <= 0 || Cm <= 0 || D <= 0 || P <= 0)
    return set_error(SGC_EINVAL, "sgc_pairs_deform_gather: bad size");
  if (P > 64 || (P & (P - 1)))
    return set_error(SGC_EUNSUP, "sgc_pairs_deform_gather: P must be a power of two <= 64 (wave-shuffle softmax)");
  FwdParams p = {};
  p.value = value; p.dist = dist; p.dist_pairs = dist_pairs_or_null; p.ref_cam = ref_cam; p.raw = raw;
  p.pair_cam = pair_cam; p.pair_q = pair_q;
  p.totals = totals; p.out = out;
  if (cam_stride_or_0 > 0 && cam_stride_or_0 < H * W) return set_error(SGC_EINVAL, "sgc_pairs_deform_gather: cam_stride < H*W");
  p.S = cam_stride_or_0 > 0 ? cam_stride_or_0 : H * W;       // pixels between consecutive cameras in value / dist
  p.M = M; p.Cm = Cm; p.D = D; p.dist_heads = 1; p.L = 1; p.Q = 1; p.P = P; p.Nq = Nq;
  p.value_bytes = ((int64_t)N * p.S + 1) * M * Cm * 4;
  p.zero_row = value_has_zero_row ? N * p.S : -1;
  p.H = H; p.W = W; p.n_items = n_pairs_or_neg;
  return launch_fwd<kPairsDeform>(p, n_pairs_or_neg >= 0 ? n_pairs_or_neg : cap, (hipStream_t)stream);
}

extern "C" int sgc_depth_score_forward(const float *dist, const int64_t *shapes3, const int64_t *lsi,
                                       const float *loc3, float *score,
                                       int B, int S, int M, int D, int L, int Q, int P, sgc_stream_t stream) {
  if (!dist || !shapes3 || !lsi || !loc3 || !score) return set_error(SGC_EINVAL, "sgc_depth_score_forward: null pointer");
  const int64_t total = (int64_t)B * Q * M * L * P;
  if (total == 0) return SGC_OK;
  const int grid = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
  hipLaunchKernelGGL(depth_score_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dist, shapes3, lsi,
                     loc3, score, total, S, M, D, L, Q, P);
  return check_launch("depth_score_fwd_kernel");
}

extern "C" int sgc_wms_forward(const float *value, const int64_t *shapes2, const int64_t *lsi,
                               const float *loc2, const float *attn, const float *score, float *out,
                               int B, int S, int M, int Cm, int L, int Q, int P, sgc_stream_t stream) {
  if (!value || !shapes2 || !lsi || !loc2 || !attn || !score || !out)
    return set_error(SGC_EINVAL, "sgc_wms_forward: null pointer");
  const bool vec4 = (Cm % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) & 15) == 0);
  const int64_t total = (int64_t)B * Q * M * (Cm / (vec4 ? 4 : 1));
  if (total == 0) return SGC_OK;
  const int grid = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
  if (vec4)
    hipLaunchKernelGGL(wms_fwd_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, value, shapes2, lsi, loc2,
                       attn, score, out, total, S, M, Cm, L, Q, P);
  else
    hipLaunchKernelGGL(wms_fwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, value, shapes2, lsi, loc2,
                       attn, score, out, total, S, M, Cm, L, Q, P);
  return check_launch("wms_fwd_kernel");
}
