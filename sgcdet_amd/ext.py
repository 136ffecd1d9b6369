"""``dfa3D._ext``-compatible operator module backed by the gfx950 library.

The reference binds four functions through pybind
(packages/3D-deformable-attention/DFA3D/dfa3D/ops/csrc/pybind.cpp:42-67) and calls them
with keyword arguments from its autograd Functions
(mmdet3d_plugin/models/im2voxel/transformer_utils/multi_scale_3ddeformable_attn_function.py).
This module exposes the same four names, the same keyword names and the same
allocate-and-return / write-in-place behaviour, so it can be handed to the reference as
``ext_module``.  ``im2col_step`` is accepted and ignored: the HIP
kernels do not chunk the batch (csrc/cuda/wms_deform_attn_cuda.cu:250-253 only used it to
bound per-launch sizes).

Inputs must be contiguous CUDA(HIP) tensors on one device -- anything else raises
``RuntimeError`` (reference: AT_ASSERTM, wms_deform_attn_cuda.cu:220-238).  There is no
CPU path.
"""
import os

import torch  # noqa: F401  (torch first: one HIP runtime per process)

from ._lib import library
from .tensor_api import TensorOps

_OPS = None


def ops():
    """The CUDA-only tensor front end of the library (also exposes the fused entry points)."""
    global _OPS
    if _OPS is None:
        _OPS = TensorOps(library(), "cuda")
        # development knobs of the kernels (variant selection only -- results never depend on them): SGC_TUNE="key=value,..."
        for kv in filter(None, os.environ.get("SGC_TUNE", "").split(",")):
            key, val = kv.split("=")
            _OPS.lib.call("sgc_set_tuning", key.strip().encode(), int(val))
    return _OPS


def ms_depth_score_sample_forward(value, value_spatial_shapes, value_level_start_index,
                                  sampling_locations, im2col_step=64):
    return ops().depth_score_forward(value, value_spatial_shapes, value_level_start_index,
                                     sampling_locations)


def ms_depth_score_sample_backward(value, value_spatial_shapes, value_level_start_index,
                                   sampling_locations, grad_output, grad_value, grad_sampling_loc,
                                   im2col_step=64):
    ops().depth_score_backward(value, value_spatial_shapes, value_level_start_index,
                               sampling_locations, grad_output, grad_value, grad_sampling_loc)


def wms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                            sampling_locations, attention_weights, depth_scores, im2col_step=64):
    return ops().wms_forward(value, value_spatial_shapes, value_level_start_index,
                             sampling_locations, attention_weights, depth_scores)


def wms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index,
                             sampling_locations, attention_weights, depth_scores, grad_output,
                             grad_value, grad_sampling_loc, grad_attn_weight, grad_depth_score,
                             im2col_step=64):
    ops().wms_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                       attention_weights, depth_scores, grad_output, grad_value, grad_sampling_loc,
                       grad_attn_weight, grad_depth_score)


__all__ = ["ms_depth_score_sample_forward", "ms_depth_score_sample_backward",
           "wms_deform_attn_forward", "wms_deform_attn_backward", "ops"]
