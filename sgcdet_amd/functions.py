"""Autograd Functions with the reference's names and argument order.

Reference classes mirrored (mmdet3d_plugin/models/im2voxel/transformer_utils/
multi_scale_3ddeformable_attn_function.py and the DFA3D package twin
dfa3D/ops/multi_scale_3D_deform_attn.py):

* ``MultiScale3DDeformableAttnFunction_fp32`` (:275-351)  -- returns ``(output, depth_score)``;
* ``MultiScaleDepthScoreSampleFunction_fp32`` (:228-273)
* ``WeightedMultiScaleDeformableAttnFunction_fp32`` (:102-178)

Differences, all behind the same call signature:

* the one-stage Function runs ONE fused HIP kernel forward and ONE backward (no
  ``[B,Q,M,L,P,4]`` round trip, no ``[..., :2].contiguous()`` slice copies);
* ``value_dpt_dist`` may be passed un-replicated (``[B,S,1,D]``) -- the reference's
  ``.repeat(1,1,num_heads,1)`` (TU/deformable_cross_attention.py:422) is not needed, and the
  returned gradient then already holds the sum over heads;
* no host sync in backward: the reference's ``grad_depth_score_.sum() != 0.0`` check
  (:314) is replaced by ``ctx.mark_non_differentiable`` on the returned depth score;
* the ``_fp16`` twins of the reference cannot run (its kernels have no half dispatch,
  SURVEY.md section 0 fact 2); they are aliases of the fp32 Functions here, which is what
  ``custom_fwd(cast_inputs=torch.float32)`` amounts to.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ext


class MultiScale3DDeformableAttnFunction_fp32(Function):
    @staticmethod
    def forward(ctx, value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step=64):
        value = value.float().contiguous()
        value_dpt_dist = value_dpt_dist.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        output, depth_score = ext.ops().dfa3d_forward(
            value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
            sampling_locations, attention_weights, want_score=True)
        ctx.save_for_backward(value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        ctx.mark_non_differentiable(depth_score)
        return output, depth_score

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, grad_depth_score_=None):
        value, dist, shapes3, lsi, loc, attn = ctx.saved_tensors
        gv, gd, gl, ga = ext.ops().dfa3d_backward(value, dist, shapes3, lsi, loc, attn,
                                                  grad_output.float().contiguous())
        return gv, gd, None, None, gl, ga, None


class MultiScaleDepthScoreSampleFunction_fp32(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                im2col_step=64):
        value = value.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        out = ext.ms_depth_score_sample_forward(value, value_spatial_shapes, value_level_start_index,
                                                sampling_locations, im2col_step=im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes3, lsi, loc = ctx.saved_tensors
        grad_value = torch.zeros_like(value)
        grad_loc = torch.zeros_like(loc)
        ext.ms_depth_score_sample_backward(value, shapes3, lsi, loc, grad_output.float().contiguous(),
                                           grad_value, grad_loc)
        return grad_value, None, None, grad_loc, None


class WeightedMultiScaleDeformableAttnFunction_fp32(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, depth_score, im2col_step=64):
        value = value.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        depth_score = depth_score.float().contiguous()
        out = ext.wms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                          sampling_locations, attention_weights, depth_score,
                                          im2col_step=im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights, depth_score)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes2, lsi, loc2, attn, score = ctx.saved_tensors
        grad_value = torch.zeros_like(value)
        grad_loc = torch.zeros_like(loc2)
        grad_attn = torch.zeros_like(attn)
        grad_score = torch.zeros_like(score)
        ext.wms_deform_attn_backward(value, shapes2, lsi, loc2, attn, score,
                                     grad_output.float().contiguous(), grad_value, grad_loc,
                                     grad_attn, grad_score)
        return grad_value, None, None, grad_loc, grad_attn, grad_score, None


class PairListDeformAttnFunction(Function):
    """The fused DFA3D operator over an ITEM LIST: item i = (camera ``item_batch[i]``, its sampling locations / weights).
    Training-path counterpart of ``MultiScale3DDeformableAttnFunction_fp32`` without the reference's padded
    [N, max_len] rebatch (TU/deformable_cross_attention.py:759-773): only visible (camera, voxel) pairs are sampled and
    back-propagated.  value [B,S,M,Cm], dist [B,S,1|M,D], loc3 [n,M,L,P,3], attn [n,M,L,P] -> [n, M*Cm]."""

    @staticmethod
    def forward(ctx, value, value_dpt_dist, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, item_batch):
        value = value.float().contiguous()
        value_dpt_dist = value_dpt_dist.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        item_batch = item_batch.to(torch.int32).contiguous()
        out = ext.ops().dfa3d_forward_items(value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
                                            sampling_locations, attention_weights, item_batch)
        ctx.save_for_backward(value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights, item_batch)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, dist, shapes3, lsi, loc, attn, item_batch = ctx.saved_tensors
        gv, gd, gl, ga = ext.ops().dfa3d_backward_items(value, dist, shapes3, lsi, loc, attn, item_batch,
                                                        grad_output.float().contiguous())
        return gv, gd, None, None, gl, ga, None


# the DFA3D package spells them without the suffix (dfa3D/ops/multi_scale_3D_deform_attn.py:22,67,146)
MultiScale3DDeformableAttnFunction = MultiScale3DDeformableAttnFunction_fp32
MultiScaleDepthScoreSampleFunction = MultiScaleDepthScoreSampleFunction_fp32
WeightedMultiScaleDeformableAttnFunction = WeightedMultiScaleDeformableAttnFunction_fp32
MultiScale3DDeformableAttnFunction_fp16 = MultiScale3DDeformableAttnFunction_fp32
MultiScaleDepthScoreSampleFunction_fp16 = MultiScaleDepthScoreSampleFunction_fp32
WeightedMultiScaleDeformableAttnFunction_fp16 = WeightedMultiScaleDeformableAttnFunction_fp32
