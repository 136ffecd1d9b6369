"""sgcdet_amd -- MI355X (gfx950) native view-transformation hot path of SGCDet.

Layout (only what the path needs):
  csrc/         hand-written HIP kernels + the C ABI of include/sgcdet_amd.h
  ext.py        ``dfa3D._ext``-compatible operator module over that ABI
  functions.py  autograd Functions with the reference's names
  mmcv_lite.py  the registry / config slice of mmcv the reference's configs need
  plugin/       ``mmdet3d_plugin`` counterparts (AdaptiveSparseHead ... ImVoxelHeadV2)
  scene.py      synthetic ScanNet / ARKit shaped scenes
"""
__version__ = "0.1.0"

