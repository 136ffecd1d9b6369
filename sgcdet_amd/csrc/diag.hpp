// Diagnostic switches of the library, in ONE place so that the product loops read clean.  Nothing here is compiled into the
// shipped library: the macros expand to nothing unless a diagnostic build defines them (tools/diag_build.sh <name> <file>
// -D<MACRO> builds a side library under tools/diag/ that the A/B tools load through SGC_DIAG_LIB).
//
//   SGC_HALO_STAMPS   shader-clock (s_memtime) and real-time (s_memrealtime) stamps around the tap loop of every workgroup of
//                     the halo convolution -> the clock the chip holds while the kernel runs (MI355X_MICROARCH.md, DVFS
//                     give-back item 6; tools/halo_clock.py).  The stamps go to a buffer of their own
//                     (sgc_diag_halo_stamp_buffer); no output value depends on them.
//   SGC_HALO_SKIP     bit mask, lockstep form of the halo convolution (halo_stagger 0), TIMING ONLY (results are garbage): 1 no
//                     barrier per tap, 2 no weight ds_write, 4 no weight global load, 8 weight fragments read from LDS once
//                     instead of every tap, 16 halo fragments read once, 32 no MFMAs, 64 (software-pipelined form) the halo image of the first channel slice is never
//                     replaced: no loads / split / LDS stores per slice, 128 no epilogue -- what each part costs (tools/halo_skip.py,
//                     tools/wz_skip.py, tools/kernel_power.py)
//   SGC_RG_STAMPS     s_memtime stamps inside the staggered form of the persistent row GEMM (rows_depth 0; tools/rows_gemm_stamps.py):
//                     buffer [workgroup < 8][wave parity 2][iteration < 32][8] x uint64 (sgc_diag_rows_stamp_buffer)
#pragma once

#if !defined(SGC_WGRAD_SKIP)
#define SGC_WGRAD_SKIP 0   // timing builds of the halo weight gradient: 1 no MFMAs, 2 no x-fragment reads, 4 no dy-fragment reads,
#endif                     // 8 no split + LDS stores of a brick, 16 no global loads (tools/wgrad_skip.py; results are garbage)
#if !defined(SGC_HALO_SKIP)
#define SGC_HALO_SKIP 0
#endif
#if !defined(SGC_TILE_SKIP)
#define SGC_TILE_SKIP 0    // timing builds of the tile implicit GEMM (conv3d_igemm_bf16x3_kernel): 1 no MFMAs, 2 no input loads, 4 no weight
#endif                     // loads, 8 no split + LDS stores, 16 no fragment reads, 32 no barrier per step, 64 no epilogue (tools/tile_skip.py)

#if defined(SGC_HALO_STAMPS)
namespace sgc { inline unsigned long long *g_halo_stamp_buf = nullptr; }
#define SGC_HALO_STAMP(slot)                                                                                     \
  do {                                                                                                           \
    if (p.stamps && threadIdx.x == 0) {                                                                          \
      unsigned long long *s_ = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4);                   \
      s_[slot] = __builtin_readcyclecounter();                                                                   \
      s_[(slot) + 1] = __builtin_amdgcn_s_memrealtime();                                                         \
    }                                                                                                            \
  } while (0)
#else
#define SGC_HALO_STAMP(slot) do {} while (0)
#endif

#if defined(SGC_RG_STAMPS)
namespace sgc { inline unsigned long long *g_rows_stamp_buf = nullptr; }
#define RG_STAMP_PTR(p, wid, late) \
  unsigned long long *stamp_ptr = ((p).stamps && blockIdx.x < 8 && ((wid) & 3) == 0) ? (p).stamps + ((blockIdx.x * 2 + (late)) * 32) * 8 : nullptr
#define RG_STAMP(slot) do { if (stamp_ptr && i < 32 && lane == 0) stamp_ptr[i * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define RG_STAMP_BIND(p) (p).stamps = sgc::g_rows_stamp_buf
#else
#define RG_STAMP_PTR(p, wid, late) do {} while (0)
#define RG_STAMP(slot) do {} while (0)
#define RG_STAMP_BIND(p) do {} while (0)
#endif
