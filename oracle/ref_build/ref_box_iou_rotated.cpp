// oracle/_ref/libref_box_iou.so -- the REFERENCE's own rotated-box IoU, compiled from the source where it lies:
//   /root/reference/packages/3D-deformable-attention/DFA3D/dfa3D/ops/csrc/common/box_iou_rotated_utils.hpp
// (the DFA3D package vendors mmcv's header: <cassert>/<cmath>/<algorithm> only, a host build needs nothing else).
// This wrapper is ours; the header is included from the reference tree at build time (oracle/Makefile, target
// `_ref`, build container only) and never copied.  TEST INFRASTRUCTURE: it pins oracle/sgc_oracle.c's
// sgc_box_iou_rotated and, through tests/golden/box_iou_rotated.npz, the HIP kernels of csrc/nms_rotated.hip.
//
// The header has two variants of its convex-hull sort: `#ifdef __CUDACC__` an in-place exchange sort (what mmcv's
// nms_rotated / box_iou_rotated CUDA kernels -- the ones the reference runs -- execute), otherwise std::sort with a
// comparator.  This file is compiled twice (oracle/Makefile): plainly (suffix _cpu) and with
// `-D__CUDACC__ -D__host__= -D__device__= -D__forceinline__=inline -DREF_SUFFIX_CUDA` (suffix _cuda), i.e. the CUDA
// branch of the same header built for the host; only CUDA's three function qualifiers are defined away.
#include "box_iou_rotated_utils.hpp"

#ifdef REF_SUFFIX_CUDA
#define REF_NAME(x) x##_cuda
#else
#define REF_NAME(x) x##_cpu
#endif

extern "C" {

// iou[i*m + j] = single_box_iou_rotated(a[i], b[j], mode 0 = IoU); boxes (xc, yc, w, h, angle in radians), fp32
void REF_NAME(ref_box_iou_rotated)(const float *a, const float *b, float *iou, int n, int m) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) iou[(long)i * m + j] = single_box_iou_rotated<float>(a + 5 * i, b + 5 * j, 0);
}

// the same in double precision (a tighter yardstick for the tolerance quoted in the tests)
void REF_NAME(ref_box_iou_rotated_f64)(const double *a, const double *b, double *iou, int n, int m) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) iou[(long)i * m + j] = single_box_iou_rotated<double>(a + 5 * i, b + 5 * j, 0);
}
}
