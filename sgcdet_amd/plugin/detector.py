"""``SGCDet`` detector shell: the three calls of the reference that form the hot path.

Reference: mmdet3d_plugin/models/detectors/SGCDet.py:61-129.  The 2D stage (ResNet + FPN,
``DepthNet_Fusion``) is upstream of the path and is NOT
built here: ``backbone`` / ``neck`` / ``depth_head`` configs are accepted and kept, and the
path starts from what they produce -- FPN maps ``x[l] = [1,N,C,H_l,W_l]`` and the depth
distribution ``[1,N,D,H_0,W_0]``:

    volume, valid, occ = voxel_head(x, img_metas[0], mlvl_dpt_dists)      # SGCDet.py:87
    feats = neck_3d(volume)                                               # :96
    outs  = bbox_head(feats); bbox_head.get_bboxes(*outs, valid.float(), img_metas)  # :123-124
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..mmcv_lite import DETECTORS, build_head, build_neck


@DETECTORS.register_module()
class SGCDet(nn.Module):
    def __init__(self, backbone=None, neck=None, depth_head=None, neck_3d=None, bbox_head=None, n_voxels=None,
                 voxel_size=None, voxel_head=None, head_2d=None, train_cfg=None, test_cfg=None,
                 use_gt_dpt=False, depth_loss=False, occ_loss=False, lighting_augmentation=False):
        super().__init__()
        self.upstream_cfg = dict(backbone=backbone, neck=neck, depth_head=depth_head, head_2d=head_2d)
        self.neck_3d = build_neck(neck_3d)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.bbox_head.voxel_size = voxel_size
        self.voxel_head = build_head(voxel_head) if voxel_head is not None else None
        self.n_voxels = n_voxels
        self.voxel_size = voxel_size
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.occ_loss = occ_loss

    @staticmethod
    def depth_pyramid(dpt_dist):
        """nearest x1/2, x1/4 copies of the depth distribution (SGCDet.py:83-85)."""
        return [dpt_dist,
                F.interpolate(dpt_dist, scale_factor=(1, 0.5, 0.5), mode="nearest"),
                F.interpolate(dpt_dist, scale_factor=(1, 0.25, 0.25), mode="nearest")]

    def build_volume_from_features(self, x, img_metas, dpt_dist):
        volume, valid, occ = self.voxel_head(x, img_metas[0], self.depth_pyramid(dpt_dist))
        if valid is None:
            _, _, vh, vw, vz = volume.shape
            valid = torch.ones([volume.shape[0], 1, vh, vw, vz], device=volume.device)
        return volume, valid, occ

    def extract_feat(self, volumes):
        return self.neck_3d(volumes)

    def forward_features(self, x, img_metas, dpt_dist):
        """FPN maps + depth distribution -> head tensors (the timed hot path)."""
        volume, valid, occ = self.build_volume_from_features(x, img_metas, dpt_dist)
        outs = self.bbox_head(self.extract_feat(volume))
        return dict(volume=volume, valid=valid, occ=occ, centerness=outs[0], bbox_pred=outs[1], cls_score=outs[2])

    def simple_test_from_features(self, x, img_metas, dpt_dist):
        r = self.forward_features(x, img_metas, dpt_dist)
        return self.bbox_head.get_bboxes(r["centerness"], r["bbox_pred"], r["cls_score"], r["valid"].float(),
                                         img_metas)
