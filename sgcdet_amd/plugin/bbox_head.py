"""FCOS3D-style voxel heads: forward convolutions and box decoding.

Reference: mmdet3d_plugin/models/dense_heads/imvoxel_head_v2.py -- ``ImVoxelHeadV2``
(:42-88, :237-317), ``ScanNetImVoxelHeadV2`` (:346-359, :437-464), ``SunRgbdImVoxelHeadV2``
(:467-483, :563-613); ``get_points`` from mmdet3d_plugin/models/detectors/utils.py:5-14.
Parameter names (``centerness_conv``, ``reg_conv``, ``cls_conv``, ``scales.i.scale``) are
the reference's.  Training side (row f-3 of SURVEY.md section 8): target assignment (:361-435, :485-561) is one
fused pass on the GPU (``sgc_assign_targets``), ``loss`` / ``forward_train`` (:90-235) follow the reference with the
losses of ``plugin/losses.py``.

NMS (row f-4): ScanNet's ``aligned_3d_nms`` (mmdet3d, a Python while-loop in the reference) runs on the GPU
(``sgc_aligned_nms3d``: one mask kernel + one sweep, same keep/drop arithmetic); ARKit's
``box3d_multiclass_nms`` + mmcv ``nms_rotated`` runs all classes in two launches (``sgc_nms_rotated_bev``: exact
rotated-rectangle intersection in the reference kernel's fp32 operation order, pinned to the copy of mmcv's
``box_iou_rotated_utils.hpp`` the reference carries in its DFA3D package -- tests/golden/box_iou_rotated.npz).
"""
import torch
from torch import nn

from .. import ext
from ..mmcv_lite import HEADS, LOSSES, Scale, bias_init_with_prob, multi_apply, normal_init
from .conv_plan import train_conv_on_hip, ConvSpec, module_fingerprint, rows_to_ncdhw, to_channels_last_rows
from . import losses  # noqa: F401  (registers the LOSSES entries the head builds)


@torch.no_grad()
def get_points(n_voxels, voxel_size, origin):
    """Voxel CORNER coordinates [3,nx,ny,nz] = idx*size + origin - n/2*size (utils.py:5-14)."""
    grid = torch.stack(torch.meshgrid([torch.arange(n_voxels[0]), torch.arange(n_voxels[1]),
                                       torch.arange(n_voxels[2])], indexing="ij"))
    new_origin = origin - n_voxels / 2.0 * voxel_size
    return grid * voxel_size.view(3, 1, 1, 1) + new_origin.view(3, 1, 1, 1)


def rotation_3d_in_axis_z(points, angles):
    """points [N,K,3] rotated about z by angles [N] (mmdet3d rotation_3d_in_axis, axis=2,
    counter-clockwise for positive angles in the depth/lidar convention)."""
    cos, sin = torch.cos(angles), torch.sin(angles)
    zeros, ones = torch.zeros_like(cos), torch.ones_like(cos)
    rot_t = torch.stack([torch.stack([cos, sin, zeros]), torch.stack([-sin, cos, zeros]),
                         torch.stack([zeros, zeros, ones])])            # [3,3,N]
    return torch.einsum("aij,jka->aik", points, rot_t)


class ImVoxelHeadV2(nn.Module):
    default_loss_bbox = None                      # set per head class (the configs always name one, :111 / :114)

    def __init__(self, n_classes, n_channels, n_reg_outs, n_scales, limit, centerness_topk=-1,
                 loss_centerness=None, loss_bbox=None, loss_cls=None, train_cfg=None, test_cfg=None,
                 nms_fn=None):
        super().__init__()
        self.n_classes = n_classes
        self.n_scales = n_scales
        self.limit = limit
        self.centerness_topk = centerness_topk
        # the three losses are built from the LOSSES registry like the reference's build_loss (:67-69); defaults = :46-57
        self.loss_centerness = LOSSES.build(loss_centerness or dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0))
        self.loss_bbox = LOSSES.build(loss_bbox or dict(type=self.default_loss_bbox, loss_weight=1.0))
        self.loss_cls = LOSSES.build(loss_cls or dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25,
                                                      loss_weight=1.0))
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.nms_fn = nms_fn
        self.voxel_size = None                    # set by the detector (SGCDet.py:36)
        self.centerness_conv = nn.Conv3d(n_channels, 1, 3, padding=1, bias=False)
        self.reg_conv = nn.Conv3d(n_channels, n_reg_outs, 3, padding=1, bias=False)
        self.cls_conv = nn.Conv3d(n_channels, n_classes, 3, padding=1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in range(n_scales)])

    def init_weights(self):
        normal_init(self.centerness_conv, std=0.01)
        normal_init(self.reg_conv, std=0.01)
        normal_init(self.cls_conv, std=0.01, bias=bias_init_with_prob(0.01))

    def _plan(self):
        """The three 3x3x3 convolutions share their input: one fused conv with
        Cout = 1 + n_reg + n_classes (centerness | reg | cls) on the MFMA kernel."""
        fp = module_fingerprint(self)
        if getattr(self, "_hip_plan", None) is not None and self._hip_plan[0] == fp:
            return self._hip_plan[1]
        w = torch.cat([self.centerness_conv.weight, self.reg_conv.weight, self.cls_conv.weight], 0)
        n_reg = self.reg_conv.weight.shape[0]
        bias = torch.cat([self.cls_conv.bias.new_zeros(1 + n_reg), self.cls_conv.bias])
        spec = ConvSpec(w, None, bias=bias, ksize=3, pad_out=False)
        self._hip_plan = (fp, (spec, n_reg))
        return self._hip_plan[1]

    def _forward_hip(self, feats, valid_masks=None):
        """``valid_masks``: per scale a uint8 [X*Y*Z] mask (the head's own valid pyramid, :123,258): the tensors are only
        consumed there (scores are multiplied by it, :301), so the convolution may skip tiles without a valid voxel."""
        from .conv_plan import CONV_MODE
        spec, n_reg = self._plan()
        ctr, reg, cls = [], [], []
        # exp(scale(reg)) of the box distances (:79,110; the first 6 regression outputs of either head class) runs in the
        # convolution's epilogue on the MFMA path -- no elementwise launches; the strict-fp32 convolution keeps the torch ops
        fused = CONV_MODE == "bf16x3" and self.reg_exp_cols > 0
        for i, (x, scale) in enumerate(zip(feats, self.scales)):
            rows, grid = to_channels_last_rows(x)
            mask = None if valid_masks is None else valid_masks[i]
            if fused:
                y, g = spec(rows, grid, out_mask=mask, act=(1, 1 + self.reg_exp_cols, scale.scale))
            else:
                y, g = spec(rows, grid, out_mask=mask)
            full = rows_to_ncdhw(y, g, spec.cout)
            ctr.append(full[:, :1])
            reg.append(full[:, 1:1 + n_reg] if fused else self._reg_activation(full[:, 1:1 + n_reg], scale))
            cls.append(full[:, 1 + n_reg:])
        return ctr, reg, cls

    reg_exp_cols = 6          # both head classes: exp(scale(.)) on the six face distances; SunRgbd's seventh output (yaw) stays raw

    def _forward_autograd_hip(self, feats):
        """Training / autograd path on the HIP kernels: the three 3x3x3 convolutions of a scale as ONE convolution with the
        concatenated weights (autograd splits the weight gradient back), imvoxel_head_v2.py:75-78,103-110."""
        from ..functions import ChannelsLastConv3dFunction, train_weight_planes
        w = torch.cat([self.centerness_conv.weight, self.reg_conv.weight, self.cls_conv.weight], 0)
        train_weight_planes().mark_ephemeral(w)          # a fresh temporary every step: packed per use (once for the three scales), never registered
        n_reg = self.reg_conv.weight.shape[0]
        ctr, reg, cls = [], [], []
        for x, scale in zip(feats, self.scales):
            rows, grid = to_channels_last_rows(x)
            y = ChannelsLastConv3dFunction.apply(rows, w, tuple(grid), 3, 1)
            full = rows_to_ncdhw(y, grid, y.shape[1])
            ctr.append(full[:, :1])
            reg.append(self._reg_activation(full[:, 1:1 + n_reg], scale))
            cls.append(full[:, 1 + n_reg:] + self.cls_conv.bias.view(1, -1, 1, 1, 1))
        return ctr, reg, cls

    def forward(self, x, valid_masks=None):
        if not self.training and not torch.is_grad_enabled() and x[0].is_cuda and x[0].shape[0] == 1:
            return self._forward_hip(x, valid_masks)
        if train_conv_on_hip(x[0], [self.cls_conv.in_channels]):
            return self._forward_autograd_hip(x)
        return multi_apply(self.forward_single, [t.contiguous() for t in x], self.scales)   # packed NCDHW for MIOpen

    def _reg_activation(self, reg, scale):
        raise NotImplementedError

    @torch.no_grad()
    def get_points(self, featmap_sizes, origin, device):
        out = []
        for i, size in enumerate(featmap_sizes):
            pts = get_points(n_voxels=torch.tensor(size), voxel_size=torch.tensor(self.voxel_size) * (2 ** i),
                             origin=torch.tensor(origin))
            out.append(pts.reshape(3, -1).transpose(0, 1).to(device))
        return out

    def get_bboxes(self, centernesses, bbox_preds, cls_scores, valid, img_metas):
        assert len(centernesses[0]) == len(bbox_preds[0]) == len(cls_scores[0]) == len(img_metas)
        valids = [nn.Upsample(size=x.shape[-3:], mode="trilinear")(valid).round().bool() for x in centernesses]
        results = []
        for b in range(len(img_metas)):
            results.append(self._get_bboxes_single(
                [x[b].detach() for x in centernesses], [x[b].detach() for x in bbox_preds],
                [x[b].detach() for x in cls_scores], [x[b].detach() for x in valids], img_metas[b]))
        return results

    def decode_candidates(self, centernesses, bbox_preds, cls_scores, valids, img_meta):
        """Per-level sigmoid scores x centerness x valid, top ``nms_pre`` per level, box decode
        (imvoxel_head_v2.py:286-315).  Returns (boxes [K, 6|7], scores [K, n_classes])."""
        sizes = [f.size()[-3:] for f in centernesses]
        mlvl_points = self.get_points(sizes, img_meta["lidar2img"]["origin"], centernesses[0].device)
        n_reg = bbox_preds[0].shape[0]
        nms_pre = self.test_cfg["nms_pre"] if self.test_cfg is not None else -1
        boxes, scores_out = [], []
        for ctr, reg, cls, valid, pts in zip(centernesses, bbox_preds, cls_scores, valids, mlvl_points):
            ctr = ctr.permute(1, 2, 3, 0).reshape(-1).sigmoid()
            reg = reg.permute(1, 2, 3, 0).reshape(-1, n_reg)
            scores = cls.permute(1, 2, 3, 0).reshape(-1, self.n_classes).sigmoid()
            valid = valid.permute(1, 2, 3, 0).reshape(-1)
            scores = scores * ctr[:, None] * valid[:, None]
            max_scores, _ = scores.max(dim=1)
            if len(scores) > nms_pre > 0:
                _, ids = max_scores.topk(nms_pre)
                reg, scores, pts = reg[ids], scores[ids], pts[ids]
            boxes.append(self._bbox_pred_to_bbox(pts, reg))
            scores_out.append(scores)
        return torch.cat(boxes), torch.cat(scores_out)

    def _get_bboxes_single(self, centernesses, bbox_preds, cls_scores, valids, img_meta):
        boxes, scores = self.decode_candidates(centernesses, bbox_preds, cls_scores, valids, img_meta)
        return self._nms(boxes, scores, img_meta)

    # ---- training side (row f-3): target assignment + losses, imvoxel_head_v2.py:90-235 --------------------
    rotated_targets = False          # SunRgbdImVoxelHeadV2: boxes with yaw, targets = the assigned gt row

    def forward_train(self, x, valid, img_metas, gt_bboxes, gt_labels):
        return self.loss(*(self(x) + (valid, img_metas, gt_bboxes, gt_labels)))

    def loss(self, centernesses, bbox_preds, cls_scores, valid, img_metas, gt_bboxes, gt_labels):
        """-> (dict(loss_centerness, loss_bbox, loss_cls), sem_occ [B, n_points], geo_occ [B, n_points]) (:95-145)."""
        assert len(centernesses[0]) == len(bbox_preds[0]) == len(cls_scores[0]) == len(valid) == len(img_metas) \
            == len(gt_bboxes) == len(gt_labels)
        valids = [nn.Upsample(size=x.shape[-3:], mode="trilinear")(valid).round().bool() for x in centernesses]
        per_img = [self._loss_single([x[i] for x in centernesses], [x[i] for x in bbox_preds], [x[i] for x in cls_scores],
                                     [x[i] for x in valids], img_metas[i], gt_bboxes[i], gt_labels[i])
                   for i in range(len(img_metas))]
        lc, lb, ls, sem, geo = zip(*per_img)
        return dict(loss_centerness=torch.mean(torch.stack(lc)), loss_bbox=torch.mean(torch.stack(lb)),
                    loss_cls=torch.mean(torch.stack(ls))), torch.stack(sem), torch.stack(geo)

    @staticmethod
    def _gt_rows(gt_bboxes, device):
        """[n_boxes, 7] (gravity centre, dims, yaw) from an mmdet3d box structure or a plain tensor in that layout."""
        if torch.is_tensor(gt_bboxes):
            rows = gt_bboxes
        else:
            rows = torch.cat((gt_bboxes.gravity_center, gt_bboxes.tensor[:, 3:]), dim=1)
        if rows.shape[1] == 6:
            rows = torch.cat((rows, rows.new_zeros(rows.shape[0], 1)), dim=1)
        return rows.to(device=device, dtype=torch.float32).contiguous()

    @torch.no_grad()
    def get_targets(self, points, gt_bboxes, gt_labels):
        """(:361-435 / :485-561) -> centerness_targets [n], bbox_targets [n, 6|7], labels [n] (-1 background),
        geo_occ_box [n] bool; one fused pass on the GPU (sgc_assign_targets), no [n_points, n_boxes] tensors."""
        dev = gt_labels.device
        scales = torch.cat([torch.full((len(p),), i, dtype=torch.int32, device=dev) for i, p in enumerate(points)])
        pts = torch.cat(points, dim=0).to(device=dev, dtype=torch.float32).contiguous()
        return ext.ops().assign_targets(pts, scales, self._gt_rows(gt_bboxes, dev), gt_labels.to(torch.int64).contiguous(),
                                        self.rotated_targets, self.n_scales, self.limit, self.centerness_topk)

    def _loss_single(self, centernesses, bbox_preds, cls_scores, valids, img_meta, gt_bboxes, gt_labels):
        """(:147-235)."""
        dev = centernesses[0].device
        sizes = [f.size()[-3:] for f in centernesses]
        mlvl_points = self.get_points(sizes, img_meta["lidar2img"]["origin"], dev)
        ctr_t, box_t, labels, geo_occ = self.get_targets(mlvl_points, gt_bboxes, gt_labels.to(dev))
        n_reg = bbox_preds[0].shape[0]
        ctr = torch.cat([c.permute(1, 2, 3, 0).reshape(-1) for c in centernesses])
        reg = torch.cat([r.permute(1, 2, 3, 0).reshape(-1, n_reg) for r in bbox_preds])
        cls = torch.cat([c.permute(1, 2, 3, 0).reshape(-1, self.n_classes) for c in cls_scores])
        val = torch.cat([v.permute(1, 2, 3, 0).reshape(-1) for v in valids])
        points = torch.cat(mlvl_points)
        pos_inds = torch.nonzero(torch.logical_and(labels >= 0, val)).reshape(-1)
        n_pos = torch.tensor(len(pos_inds), dtype=torch.float, device=dev)
        if torch.distributed.is_available() and torch.distributed.is_initialized():      # mmdet reduce_mean
            n_pos = n_pos.clone()
            torch.distributed.all_reduce(n_pos.div_(torch.distributed.get_world_size()))
        n_pos = max(float(n_pos), 1.0)
        if torch.any(val):
            loss_cls = self.loss_cls(cls[val], labels[val], avg_factor=n_pos)
        else:
            loss_cls = cls[val].sum()
        pos_ctr, pos_reg = ctr[pos_inds], reg[pos_inds]
        if len(pos_inds) > 0:
            pos_ctr_t = ctr_t[pos_inds]
            loss_centerness = self.loss_centerness(pos_ctr, pos_ctr_t, avg_factor=n_pos)
            loss_bbox = self.loss_bbox(self._bbox_pred_to_bbox(points[pos_inds], pos_reg), box_t[pos_inds],
                                       weight=pos_ctr_t, avg_factor=pos_ctr_t.sum())
        else:
            loss_centerness, loss_bbox = pos_ctr.sum(), pos_reg.sum()
        return loss_centerness, loss_bbox, loss_cls, labels, geo_occ

    def forward_single(self, x, scale):
        raise NotImplementedError

    def _bbox_pred_to_bbox(self, points, bbox_pred):
        raise NotImplementedError

    def _nms(self, bboxes, scores, img_meta):
        raise NotImplementedError


@HEADS.register_module()
class ScanNetImVoxelHeadV2(ImVoxelHeadV2):
    default_loss_bbox = "AxisAlignedIoULoss"      # configs/SGCDet_ScanNet.py:111

    def forward_single(self, x, scale):
        return self.centerness_conv(x), torch.exp(scale(self.reg_conv(x))), self.cls_conv(x)

    def _reg_activation(self, reg, scale):
        return torch.exp(scale(reg))

    def _bbox_pred_to_bbox(self, points, bbox_pred):
        """point -/+ distances -> (x0,y0,z0,x1,y1,z1), :456-464."""
        lo = points - bbox_pred[:, [0, 2, 4]]
        hi = points + bbox_pred[:, [1, 3, 5]]
        return torch.cat([lo, hi], -1)

    def _nms(self, bboxes, scores, img_meta):
        scores, labels = scores.max(dim=1)
        keep = scores > self.test_cfg["score_thr"]
        bboxes, scores, labels = bboxes[keep], scores[keep], labels[keep]
        if self.nms_fn is not None:
            ids = self.nms_fn(bboxes, scores, labels, self.test_cfg["iou_thr"])
            bboxes, scores, labels = bboxes[ids], scores[ids], labels[ids]
        elif bboxes.is_cuda:      # mmdet3d aligned_3d_nms on the GPU (sgc_aligned_nms3d): mask kernel + one sweep
            ids = ext.ops().aligned_nms3d(bboxes.float().contiguous(), scores.float().contiguous(), labels,
                                          self.test_cfg["iou_thr"])
            bboxes, scores, labels = bboxes[ids], scores[ids], labels[ids]
        centers = (bboxes[:, :3] + bboxes[:, 3:6]) / 2.0
        bboxes = torch.cat([centers, bboxes[:, 3:6] - bboxes[:, :3]], dim=1)
        box_type = img_meta.get("box_type_3d") if isinstance(img_meta, dict) else None
        if box_type is not None:
            bboxes = box_type(bboxes, origin=(0.5, 0.5, 0.5), box_dim=6, with_yaw=False)
        return bboxes, scores, labels


@HEADS.register_module()
class SunRgbdImVoxelHeadV2(ImVoxelHeadV2):
    rotated_targets = True
    default_loss_bbox = "RotatedIoU3DLoss"        # configs/SGCDet_ARKit.py:114

    def forward_single(self, x, scale):
        reg = self.reg_conv(x)
        return self.centerness_conv(x), torch.cat((torch.exp(scale(reg[:, :6])), reg[:, 6:]), dim=1), self.cls_conv(x)

    def _reg_activation(self, reg, scale):
        return torch.cat((torch.exp(scale(reg[:, :6])), reg[:, 6:]), dim=1)

    @staticmethod
    def _bbox_pred_to_bbox(points, bbox_pred):
        """(dx-,dx+,dy-,dy+,dz-,dz+,alpha) -> (cx,cy,cz,w,l,h,alpha), :595-613."""
        if bbox_pred.shape[0] == 0:
            return bbox_pred
        shift = torch.stack(((bbox_pred[:, 1] - bbox_pred[:, 0]) / 2, (bbox_pred[:, 3] - bbox_pred[:, 2]) / 2,
                             (bbox_pred[:, 5] - bbox_pred[:, 4]) / 2), dim=-1).view(-1, 1, 3)
        shift = rotation_3d_in_axis_z(shift, bbox_pred[:, 6])[:, 0, :]
        size = torch.stack((bbox_pred[:, 0] + bbox_pred[:, 1], bbox_pred[:, 2] + bbox_pred[:, 3],
                            bbox_pred[:, 4] + bbox_pred[:, 5]), dim=-1)
        return torch.cat((points + shift, size, bbox_pred[:, 6:7]), dim=-1)

    def _nms(self, bboxes, scores, img_meta):
        """box3d_multiclass_nms on BEV rectangles with rotated IoU (:565-584).  The reference appends a dummy
        background column only because mmdet3d's loop runs over ``shape[1] - 1`` classes; here every column of
        ``scores`` is a class."""
        if self.nms_fn is not None:
            return self.nms_fn(bboxes, scores, self.test_cfg, img_meta)
        cfg = self.test_cfg
        if not cfg.get("use_rotate_nms", True):
            raise NotImplementedError("SunRgbdImVoxelHeadV2: nms_normal_bev (use_rotate_nms=False) is not built; "
                                      "pass nms_fn= (the reference's ARKit configs use rotated NMS)")
        bboxes, scores, labels = box3d_multiclass_nms_rotated(
            ext.ops(), bboxes.float().contiguous(), scores.float().contiguous(), cfg["score_thr"], cfg["nms_pre"],
            cfg["nms_thr"])
        box_type = img_meta.get("box_type_3d") if isinstance(img_meta, dict) else None
        if box_type is not None:
            bboxes = box_type(bboxes, origin=(0.5, 0.5, 0.5))
        return bboxes, scores, labels


def box3d_multiclass_nms_rotated(ops, bboxes, scores, score_thr, max_num, nms_thr):
    """mmdet3d ``box3d_multiclass_nms`` (box3d_nms.py:8-128) with ``cfg.use_rotate_nms`` as the ARKit head calls it:
    bboxes [K,7] (cx,cy,cz,w,l,h,yaw), scores [K,C] -> (boxes [n,7], scores [n], labels [n]) concatenated class by
    class, each class in descending score; more than ``max_num`` survivors: the best ``max_num`` by score.
    ``ops``: a TensorOps (the HIP library in the product, the CPU oracle in tests).  One host read-back (the
    per-class survivor counts) -- the outputs are variable-length."""
    K, C = scores.shape
    if K == 0:
        return bboxes.new_zeros((0, bboxes.shape[-1])), scores.new_zeros((0,)), scores.new_zeros((0,), dtype=torch.long)
    bev = torch.stack((bboxes[:, 0] - bboxes[:, 3] / 2, bboxes[:, 1] - bboxes[:, 4] / 2,
                       bboxes[:, 0] + bboxes[:, 3] / 2, bboxes[:, 1] + bboxes[:, 4] / 2, bboxes[:, 6]), dim=1)
    keep, n_keep = ops.nms_rotated_bev(bev.contiguous(), scores, score_thr, nms_thr)
    live = torch.arange(K, device=keep.device)[None, :] < n_keep[:, None]        # [C,K], row-major = class order
    sel = keep[live]
    labels = torch.arange(C, device=keep.device)[:, None].expand(C, K)[live]
    out_boxes, out_scores = bboxes[sel], scores[sel, labels]
    if out_boxes.shape[0] > max_num:
        _, inds = out_scores.sort(descending=True)
        inds = inds[:max_num]
        out_boxes, out_scores, labels = out_boxes[inds], out_scores[inds], labels[inds]
    return out_boxes, out_scores, labels
