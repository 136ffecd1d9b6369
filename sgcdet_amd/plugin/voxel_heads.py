"""Coarse-to-fine sparse volume construction (``AdaptiveSparseHead`` / ``DenseHead``).

Reference: mmdet3d_plugin/models/im2voxel/AdaptiveSparseHead.py:9-102 and
mmdet3d_plugin/models/im2voxel/DenseHead.py:10-84.  Registry names, kwargs, buffers
(``vox_coords``, ``ref_3d``) and parameter names (``base_heads.i...``,
``occ_pred_heads.i.0``) are the reference's; call contract:

    AdaptiveSparseHead(mlvl_feats, img_meta, mlvl_dpt_dists)
        -> (volume [1,C,nx,ny,nz], valid [1,1,nx,ny,nz] int64, occ_preds [1, sum Nvox])

MI355X differences: the top-k indices are sorted and handed to ``DenseHead`` directly, so
the ``nonzero`` host sync of DenseHead.py:66 disappears (the selected set and its ascending
order are identical); the per-level scatter into the dense volume is one HIP launch.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..mmcv_lite import HEADS, build_head, build_transformer
from .. import ext


class _TrilinearUp2x(torch.autograd.Function):
    """``F.interpolate(x, scale_factor=2, mode='trilinear', align_corners=False)`` (AdaptiveSparseHead.py:64-69) whose
    backward is a gather (``sgc_upsample2x_backward``) instead of torch's atomic scatter."""

    @staticmethod
    def forward(ctx, x):
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[0] == 1:
            # the inference path's kernel (sgc_upsample2x_occ without the occupancy head) on channels-last rows; the result is
            # returned as the channels-last [1, C, 2X, 2Y, 2Z] view, so that the occupancy Linear that follows reads it in place
            # (torch's trilinear kernel took 250 us for the 26 MB output of the finest level, and its NCDHW result cost the
            # Linear a permuting copy)
            _, C, X, Y, Z = x.shape
            rows = x[0].permute(1, 2, 3, 0).reshape(X * Y * Z, C).contiguous()
            up, _, (ox, oy, oz) = ext.ops().upsample2x_occ(rows, (X, Y, Z))
            return up.view(ox, oy, oz, C).permute(3, 0, 1, 2).unsqueeze(0)
        return F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)

    @staticmethod
    def backward(ctx, grad_out):
        return ext.ops().upsample2x_backward(grad_out.contiguous().float())


def trilinear_up2x(x):
    if x.is_cuda and x.requires_grad and torch.is_grad_enabled() and x.dtype == torch.float32:
        return _TrilinearUp2x.apply(x)
    return F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)


def topk_wo_grad(occ_preds_flatten, topk=10):
    """Hard top-k mask, no gradient (AdaptiveSparseHead.py:9-13).  On the GPU the selection runs in one HIP launch with
    ties at the cut broken by the lowest flat index; elsewhere (CPU tensors) ties follow torch.topk."""
    if occ_preds_flatten.is_cuda and occ_preds_flatten.dtype == torch.float32 and occ_preds_flatten.shape[0] == 1:
        _, _, mask = ext.ops().topk_select(occ_preds_flatten.detach().contiguous(), topk, want_mask=True)
        return mask.view_as(occ_preds_flatten)
    _, idx = torch.topk(occ_preds_flatten, k=topk, dim=1)
    return torch.zeros_like(occ_preds_flatten).scatter_(1, idx, 1.0)


@HEADS.register_module()
class DenseHead(nn.Module):
    def __init__(self, *args, voxel_size=None, n_voxels=None, embed_dims, cross_transformer, **kwargs):
        super().__init__()
        self.voxel_size = torch.tensor(voxel_size)
        self.n_voxels = torch.tensor(n_voxels)
        self.embed_dims = embed_dims
        self.cross_transformer = build_transformer(cross_transformer)
        vox_coords, ref_3d = self.get_voxel_indices()
        self.register_buffer("vox_coords", vox_coords)
        self.register_buffer("ref_3d", ref_3d)
        self._tag_flat_coords()

    def _tag_flat_coords(self):
        """Declare ``vox_coords[:, 3] == arange`` (true by construction in get_voxel_indices) on the tensor OBJECT, keyed by
        its version counter: the transformer then never needs the host read-back of ``_coords_are_flat`` -- also not when the
        first use of a moved buffer falls into a graph capture.  An in-place overwrite (``load_state_dict``) bumps the version
        and voids the tag; the content is then checked once on the host."""
        vc = self.vox_coords
        vc._sgc_flat = (vc._version, True)

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .float() build a new tensor object: carry the tag over -- but only if the tensor that is being
        moved still held a VALID one.  In the usual order build -> load_state_dict -> .cuda() the in-place load has bumped the
        version and voided the tag; the moved copy then stays untagged and ``_coords_are_flat`` checks the loaded content once
        on the host (a checkpoint's persistent buffer, DenseHead.py:29, is data, not something to assert about)."""
        vc = self.vox_coords
        tag = getattr(vc, "_sgc_flat", None)
        was_flat = tag is not None and tag[0] == vc._version and tag[1]
        out = super()._apply(fn, *args, **kwargs)
        if was_flat:
            self._tag_flat_coords()
        return out

    def get_voxel_indices(self):
        """vox_coords [Nvox,4] = (x,y,z,flat) with flat = (x*ny + y)*nz + z; ref_3d [Nvox,3] =
        idx * voxel_size - n_voxels/2 * voxel_size (voxel corner, no +0.5; DenseHead.py:32-48)."""
        nx, ny, nz = (int(v) for v in self.n_voxels)
        grid = torch.stack(torch.meshgrid(torch.arange(nx), torch.arange(ny), torch.arange(nz), indexing="ij"))
        flat = torch.arange(nx * ny * nz)
        vox_coords = torch.cat([grid.reshape(3, -1).t(), flat[:, None]], dim=-1)
        new_origin = -self.n_voxels / 2.0 * self.voxel_size
        points = grid * self.voxel_size.view(3, 1, 1, 1) + new_origin.view(3, 1, 1, 1)
        # contiguous [Nvox, 3]: the projection kernel reads it as rows (a permuted view cost one strided copy per level and scene)
        return vox_coords, points.view(3, -1).permute(1, 0).contiguous()

    def seed_rows(self, mlvl_feats, img_meta, idx=None, **kwargs):
        """Inference building block: features of the selected voxels as rows [Nq, C] (``idx`` ascending
        int64 flat voxel indices, None = every voxel) -- what ``forward`` scatters into the dense volume."""
        device = mlvl_feats[0].device
        if idx is None:
            idx = self.__dict__.get("_all_idx")
            if idx is None or idx.device != device:
                idx = torch.arange(int(self.n_voxels.prod()), device=device)
                self.__dict__["_all_idx"] = idx
        return self.cross_transformer.get_vox_features(
            mlvl_feats, None, ref_3d=self.ref_3d, vox_coords=self.vox_coords, unmasked_idx=idx, bev_pos=None,
            prev_bev=None, img_meta=img_meta, **kwargs).squeeze(0)

    def forward(self, mlvl_feats, img_meta=None, proposal=None, proposal_idx=None, **kwargs):
        """mlvl_feats: list of [1,N,C,H,W]; proposal: [Nvox] {0,1} mask or None (= all);
        proposal_idx: optional ascending int64 indices equal to ``nonzero(proposal > 0)``.
        Returns [1,C,nx,ny,nz]."""
        bs = mlvl_feats[0].shape[0]
        device = mlvl_feats[0].device
        assert bs == 1
        n_vox = int(self.n_voxels.prod())
        C = self.embed_dims
        if proposal_idx is not None:
            unmasked_idx = proposal_idx
        elif proposal is None:
            unmasked_idx = torch.arange(n_vox, device=device)
        else:
            unmasked_idx = torch.nonzero(proposal > 0).view(-1)          # host sync, reference behaviour
        # content-free queries (:63): only materialised where autograd needs the reference's data flow
        volume_queries = torch.zeros((n_vox, C), device=device) if torch.is_grad_enabled() else None
        seed_feats = self.cross_transformer.get_vox_features(
            mlvl_feats, volume_queries, ref_3d=self.ref_3d, vox_coords=self.vox_coords,
            unmasked_idx=unmasked_idx, bev_pos=None, prev_bev=None, img_meta=img_meta, **kwargs).squeeze(0)
        if torch.is_grad_enabled() and seed_feats.requires_grad:
            volume_out = torch.zeros((n_vox, C), device=device).index_put((unmasked_idx,), seed_feats)
        else:
            volume_out = torch.zeros((n_vox, C), device=device)
            ext.ops().scatter_rows(seed_feats.contiguous(), unmasked_idx.to(torch.int32), volume_out)
        nx, ny, nz = (int(v) for v in self.n_voxels)
        return volume_out.view(nx, ny, nz, C).permute(3, 0, 1, 2).unsqueeze(0)


@HEADS.register_module()
class AdaptiveSparseHead(nn.Module):
    def __init__(self, embed_dims=256, topk_list=None, voxel_size_list=None, n_voxels_list=None,
                 base_head_configs=None, **kwargs):
        super().__init__()
        self.embed_dims = embed_dims
        self.topk_list = list(topk_list) if topk_list is not None else []
        self.voxel_size_list = list(voxel_size_list) if voxel_size_list is not None else []
        self.n_voxels_list = list(n_voxels_list) if n_voxels_list is not None else []
        self.base_heads = nn.ModuleList([build_head(cfg) for cfg in (base_head_configs or [])])
        self.occ_pred_heads = nn.ModuleList(
            [nn.Sequential(nn.Linear(embed_dims, 1), nn.Sigmoid()) for _ in range(len(self.base_heads) - 1)])
        self.loss = nn.BCELoss()
        # Controlled selection for synthetic scenes ("optionally override the mask"): a list with one score
        # tensor [V_i] (float32, on the module's device) or None per refined level; the level's top-k then ranks THESE scores
        # instead of the predicted occupancy (which is still computed and returned).  None = the reference's behaviour.
        # sgcdet_amd.scene.clustered_occupancy builds a surface-clustered one.
        self.occupancy_override = None

    def _selection_scores(self, i, occ):
        """scores the top-k of refined level i (1-based) ranks: the predicted occupancy, or the controlled override"""
        ov = self.occupancy_override
        if ov is None or ov[i - 1] is None:
            return occ
        s = ov[i - 1].view_as(occ)
        if s.device != occ.device or s.dtype != occ.dtype:
            raise RuntimeError("occupancy_override: scores must live on the module's device as float32")
        return s

    def _level_inputs(self, i, mlvl_feats, img_meta, mlvl_dpt_dists):
        """level i samples FPN map n_lvl-1-i cropped to the un-padded image (AdaptiveSparseHead.py:52-59)"""
        n_lvl = len(self.base_heads)
        ds = 4 * (2 ** (n_lvl - 1 - i))
        h, w = img_meta["img_shape"][0] // ds, img_meta["img_shape"][1] // ds
        k = n_lvl - 1 - i
        return mlvl_feats[k][:, :, :, :h, :w], mlvl_dpt_dists[k][:, :, :, :h, :w]

    def _forward_rows(self, mlvl_feats, img_meta, mlvl_dpt_dists):
        """Inference path on channels-last rows [V, C]: trilinear upsample + occupancy head in one HIP
        launch, `upsampled + DenseHead(selected)` as a row scatter-add (the dense head's output is zero
        outside the selection), no NCDHW <-> NDHWC copies, no dense zero volumes."""
        ops = ext.ops()
        C = self.embed_dims
        # "_sgc_static": no host read-backs (device-side counts, worst-case buffers): whole-scene hipGraph capture
        extra = dict(static_counts=True) if img_meta.get("_sgc_static") else {}
        feat, dpt = self._level_inputs(0, mlvl_feats, img_meta, mlvl_dpt_dists)
        rows = self.base_heads[0].seed_rows([feat], img_meta, None, mlvl_dpt_dists=[dpt], **extra).contiguous()
        grid = tuple(int(v) for v in self.base_heads[0].n_voxels)
        occ_list, valid = [], None
        # the reference returns cat(occ_preds_list[::-1], dim=1) (finest level first, AdaptiveSparseHead.py:99): every level's
        # scores are written straight into their slice of that vector (no cat launch)
        sizes = [int(h.n_voxels.prod()) for h in self.base_heads[1:]]
        occ_all = torch.empty((1, sum(sizes)), dtype=torch.float32, device=rows.device) if sizes else None
        ends = [sum(sizes[j:]) for j in range(len(sizes))]                    # level i occupies [ends[i-1] - sizes[i-1], ends[i-1])
        for i in range(1, len(self.base_heads)):
            lin = self.occ_pred_heads[i - 1][0]
            up, occ, grid = ops.upsample2x_occ(rows, grid, lin.weight.reshape(-1), lin.bias,
                                               occ_out=occ_all[0, ends[i - 1] - sizes[i - 1]:ends[i - 1]])
            occ_list.append(occ.view(1, -1))
            feat, dpt = self._level_inputs(i, mlvl_feats, img_meta, mlvl_dpt_dists)
            if (i - 1) < len(self.topk_list):
                # the k highest occupancy scores as an ascending index list (== nonzero(mask)) in ONE launch; ties at the
                # cut go to the lowest flat index (csrc/rows.hip) -- no topk / sort / scatter library kernels, no host sync
                last = i == len(self.base_heads) - 1
                idx, valid, _ = ops.topk_select(self._selection_scores(i, occ), self.topk_list[i - 1], want_valid=last)
                seed = self.base_heads[i].seed_rows([feat], img_meta, idx, mlvl_dpt_dists=[dpt], **extra)
                ops.scatter_add_rows(seed.contiguous(), idx, up)
            else:
                valid = None
                up.add_(self.base_heads[i].seed_rows([feat], img_meta, None, mlvl_dpt_dists=[dpt], **extra))
            rows = up
        nx, ny, nz = grid
        volume = rows.view(nx, ny, nz, C).permute(3, 0, 1, 2).unsqueeze(0)
        if not occ_list:
            return volume, torch.ones([1, 1, nx, ny, nz], device=volume.device), None
        occ_preds = occ_all
        if valid is None:                       # no top-k at the last level (not an SGCDet config): everything is valid
            valid = torch.ones(nx * ny * nz, dtype=torch.int64, device=volume.device)
        return volume, valid.view(1, 1, nx, ny, nz), occ_preds

    def forward(self, mlvl_feats, img_meta, mlvl_dpt_dists):
        assert mlvl_feats[0].shape[0] == 1
        if not torch.is_grad_enabled() and mlvl_feats[0].is_cuda and self.embed_dims % 4 == 0:
            return self._forward_rows(mlvl_feats, img_meta, mlvl_dpt_dists)
        n_lvl = len(self.base_heads)
        volume = None
        occ_preds_list = []
        mask = None
        for i in range(n_lvl):
            # level i samples FPN map n_lvl-1-i cropped to the un-padded image (:52-59)
            ds = 4 * (2 ** (n_lvl - 1 - i))
            h, w = img_meta["img_shape"][0] // ds, img_meta["img_shape"][1] // ds
            k = n_lvl - 1 - i
            feat = mlvl_feats[k][:, :, :, :h, :w]
            dpt = mlvl_dpt_dists[k][:, :, :, :h, :w]
            if i == 0:
                volume = self.base_heads[0]([feat], img_meta, mlvl_dpt_dists=[dpt])
                continue
            up = trilinear_up2x(volume)
            occ = self.occ_pred_heads[i - 1](up.permute(0, 2, 3, 4, 1)).reshape(1, -1)
            occ_preds_list.append(occ)
            mask, idx = None, None
            if (i - 1) < len(self.topk_list):
                sel = self._selection_scores(i, occ)
                if occ.is_cuda and occ.dtype == torch.float32:
                    idx, _, mask = ext.ops().topk_select(sel.detach().contiguous(), self.topk_list[i - 1], want_mask=True)
                else:
                    _, top = torch.topk(sel, k=self.topk_list[i - 1], dim=1)
                    mask = torch.zeros_like(occ).scatter_(1, top, 1.0).squeeze(0)
                    idx = top.squeeze(0).sort().values                   # == nonzero(mask), no host sync
            volume = up + self.base_heads[i]([feat], img_meta, proposal=mask, proposal_idx=idx,
                                             mlvl_dpt_dists=[dpt])
        if not occ_preds_list:
            _, _, vh, vw, vz = volume.shape
            return volume, torch.ones([1, 1, vh, vw, vz], device=volume.device), None
        occ_preds = torch.cat(occ_preds_list[::-1], dim=1)
        valid = self.get_valid(mask).unsqueeze(0).unsqueeze(0).detach()
        return volume, valid, occ_preds

    def get_valid(self, indices_0):
        nx, ny, nz = self.n_voxels_list[-1]
        return indices_0.view(nx, ny, nz).bool().long()

    def occ_loss(self, occ_pred, sem_occ_gt, geo_occ_gt):
        n = occ_pred.shape[1]
        return {"loss_occ": self.loss(occ_pred, geo_occ_gt[:, 0:n].float()).mean() * 0.5}
