"""MI355X-native counterparts of the ``mmdet3d_plugin`` modules on the SGCDet hot path.

Importing this package registers every ``type=`` name the reference's configs use for the
path in the registries of ``sgcdet_amd.mmcv_lite``.
"""
from .voxformer import (MSDeformableAttention3D_DFA3D, DeformCrossAttention_DFA3D, MyCustomBaseTransformerLayer,
                        VoxFormerLayer, VoxFormerEncoder_DFA3D, PerceptionTransformer_DFA3D, compute_projection)
from .voxel_heads import AdaptiveSparseHead, DenseHead, topk_wo_grad
from .neck3d import FastIndoorImVoxelNeck, BasicBlock3dV2
from .bbox_head import ImVoxelHeadV2, ScanNetImVoxelHeadV2, SunRgbdImVoxelHeadV2, get_points
from .detector import SGCDet
from .depth_net import DepthNet_Fusion, ResNetFPN, SimpleUnet2D, ConvBnReLU2D
from .fpn import FPN

__all__ = [
    "MSDeformableAttention3D_DFA3D", "DeformCrossAttention_DFA3D", "MyCustomBaseTransformerLayer",
    "VoxFormerLayer", "VoxFormerEncoder_DFA3D", "PerceptionTransformer_DFA3D", "compute_projection",
    "AdaptiveSparseHead", "DenseHead", "topk_wo_grad", "FastIndoorImVoxelNeck", "BasicBlock3dV2",
    "ImVoxelHeadV2", "ScanNetImVoxelHeadV2", "SunRgbdImVoxelHeadV2", "get_points", "SGCDet",
    "DepthNet_Fusion", "ResNetFPN", "SimpleUnet2D", "ConvBnReLU2D", "FPN",
]
