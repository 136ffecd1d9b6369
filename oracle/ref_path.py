"""CPU ORACLE, module level -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A functional torch-CPU restatement of the reference's module path in the reference's own
operation order and data layout (padded per-camera rebatch, dense slots, full-size
``nn.MultiheadAttention`` with ``key_padding_mask``), driven by a plain ``state_dict`` with
the reference's parameter names.  The two native operators run on the C restatement
(``oracle.ops()``).  It shares no code with ``sgcdet_amd/plugin``: that is the point.

Restated (paths under /root/reference/mmdet3d_plugin/models/):
  im2voxel/AdaptiveSparseHead.py:43-98      adaptive_sparse_head()
  im2voxel/DenseHead.py:32-84               dense_head()
  im2voxel/transformer_utils/transformer.py:118-185   (flatten / permute of the maps)
  im2voxel/transformer_utils/encoder.py:168-223,262-340   project(), layer order
  im2voxel/transformer_utils/deformable_cross_attention.py:67-116,364-501,705-837
  necks/imvoxelnet.py:22-34,146-173         neck()
  dense_heads/imvoxel_head_v2.py:237-317,346-359,456-477,595-613   head(), decode()

Pinned by tests/golden/*.npz produced by running the reference's own Python for these
modules in the build container (tests/golden/make_golden.py).
"""

import torch
import torch.nn.functional as F

from . import ops as _oracle_ops


class RefPath:
    def __init__(self, state_dict, cfg, omp=False):
        """cfg: dict(embed_dims, n_voxels_list, voxel_size_list, topk_list, dbound, num_heads,
        num_points, n_classes, n_reg_outs, head ('scannet'|'sunrgbd'), nms_pre)."""
        self.sd = {k: v.detach().cpu().float() if v.is_floating_point() else v.detach().cpu()
                   for k, v in state_dict.items()}
        self.cfg = cfg
        self.ops = _oracle_ops(omp=omp)
        self.timing = None          # dict stage -> seconds when enabled (bench.py's cpu_baseline per-stage split)

    def _t(self, stage):
        """with self._t("stage"): ...  accumulates wall time into self.timing[stage] when timing is enabled."""
        import contextlib
        import time
        if self.timing is None:
            return contextlib.nullcontext()
        rp = self

        class _Timer:
            def __enter__(self_):
                self_.t0 = time.perf_counter()

            def __exit__(self_, *exc):
                rp.timing[stage] = rp.timing.get(stage, 0.0) + time.perf_counter() - self_.t0
                return False
        return _Timer()

    # ---------------------------------------------------------------- helpers
    def lin(self, x, prefix):
        return F.linear(x, self.sd[prefix + ".weight"], self.sd.get(prefix + ".bias"))

    @staticmethod
    def compute_projection(img_meta, stride=1):
        intrinsic = torch.tensor(img_meta["lidar2img"]["intrinsic"][:3, :3])
        ratio = img_meta["ori_shape"][0] / (img_meta["img_shape"][0] / stride)
        intrinsic[:2] /= ratio
        return torch.stack([intrinsic @ torch.tensor(e)[:3] for e in img_meta["lidar2img"]["extrinsic"]])

    def project(self, ref3d, img_meta):
        """encoder.py:179-223 on the C oracle's fixed arithmetic order."""
        proj = self.compute_projection(img_meta).float().contiguous()
        origin = torch.tensor(img_meta["lidar2img"]["origin"]).float()
        db = self.cfg["dbound"]
        return self.ops.project_points(ref3d.contiguous(), origin, proj, img_meta["img_shape"][1],
                                       img_meta["img_shape"][0], db[0], db[1])

    # ---------------------------------------------------------------- one level
    def cross_attention(self, pre, feat_flat, dist_flat, H, W, ref_cam, mask):
        """DeformCrossAttention_DFA3D.forward (deformable_cross_attention.py:705-837).
        feat_flat [N,S,C], dist_flat [N,S,D], ref_cam [N,Nq,3], mask [N,Nq] bool -> [1,Nq,C]."""
        cfg = self.cfg
        C, M, P = cfg["embed_dims"], cfg["num_heads"], cfg["num_points"]
        N, Nq = mask.shape
        S = H * W
        D = dist_flat.shape[-1]
        with self._t("project_compact"):
            indexes = [mask[i].nonzero().squeeze(-1) for i in range(N)]            # :759-762
            max_len = max(len(ix) for ix in indexes)
            ref_rebatch = torch.zeros(N, max_len, 1, 3)                           # :766-773
            for i, ix in enumerate(indexes):
                ref_rebatch[i, :len(ix), 0] = ref_cam[i, ix]
        shapes3 = torch.tensor([[H, W, D]], dtype=torch.int64)
        lsi = torch.zeros(1, dtype=torch.int64)
        # Grid_Sample_3D_Feature (:67-116): one head, one point, weight one, replicated depth
        loc = ref_rebatch.view(N, max_len, 1, 1, 1, 3).contiguous()
        ones = torch.ones(N, max_len, 1, 1, 1)
        with self._t("geometry_sample"):
            q_img, _ = self.ops.dfa3d_forward(feat_flat.view(N, S, 1, C).contiguous(), dist_flat.view(N, S, 1, D).contiguous(),
                                              shapes3, lsi, loc, ones)
        da = pre + ".deformable_attention"
        # MSDeformableAttention3D_DFA3D.forward (:364-501)
        with self._t("gemms"):
            value = self.lin(feat_flat, da + ".value_proj").view(N, S, M, C // M).contiguous()
            dist_rep = dist_flat.view(N, S, 1, D).repeat(1, 1, M, 1).contiguous()  # :422
            off_uv = self.lin(q_img, da + ".sampling_offsets").view(N, max_len, M, 1, P, 2)
            off_d = self.lin(q_img, da + ".sampling_offsets_depth").view(N, max_len, M, 1, P, 1)
            offs = torch.cat([off_uv, off_d], -1)
            attn = self.lin(q_img, da + ".attention_weights").view(N, max_len, M, P).softmax(-1).view(N, max_len, M, 1, P)
            normalizer = torch.tensor([[W, H, D]], dtype=torch.int64)
            offs = offs / normalizer[None, None, None, :, None, :]
            loc = (ref_rebatch[:, :, None, None, None, :, :] + offs.view(N, max_len, M, 1, P, 1, 3)).view(N, max_len, M, 1, P, 3)
        with self._t("deform_gather"):
            queries, _ = self.ops.dfa3d_forward(value, dist_rep, shapes3, lsi, loc.contiguous(), attn.contiguous())
        # dense slots, masked mean, attention over views (:815-837)
        with self._t("view_pooling"):
            slots = torch.zeros(N, 1, Nq, C)
            for i, ix in enumerate(indexes):
                slots[i, 0, ix] = queries[i, :len(ix)]
            count = mask.sum(0)
            valid_index = count.nonzero()[:, 0]
            valid_slots = slots[:, :, valid_index, :]
            valid_mask = mask[:, None, valid_index, None]
            mean = (valid_slots * valid_mask).sum(dim=0) / count[None, valid_index, None]
            mean = self.lin(mean, pre + ".output_proj")
            mha = pre + ".attention_pooling"
            pooled, _ = F.multi_head_attention_forward(
                mean, valid_slots.squeeze(1), valid_slots.squeeze(1), C, 8,
                self.sd[mha + ".in_proj_weight"], self.sd[mha + ".in_proj_bias"], None, None, False, 0.0,
                self.sd[mha + ".out_proj.weight"], self.sd[mha + ".out_proj.bias"], training=False,
                key_padding_mask=~mask[:, valid_index].t(), need_weights=True)
            out = torch.zeros(1, Nq, C)
            out[:, valid_index, :] = pooled
        return out                                                            # dropout(0) + zero queries

    def layer(self, pre, feat_flat, dist_flat, H, W, ref_cam, mask):
        """VoxFormerLayer, order cross_attn -> norm -> ffn -> norm (encoder.py:310-338)."""
        C = self.cfg["embed_dims"]
        x = self.cross_attention(pre + ".attentions.0", feat_flat, dist_flat, H, W, ref_cam, mask)
        with self._t("ln_ffn"):
            x = F.layer_norm(x, (C,), self.sd[pre + ".norms.0.weight"], self.sd[pre + ".norms.0.bias"])
            h = F.relu(self.lin(x, pre + ".ffns.0.layers.0.0"))
            x = x + self.lin(h, pre + ".ffns.0.layers.1")
            return F.layer_norm(x, (C,), self.sd[pre + ".norms.1.weight"], self.sd[pre + ".norms.1.bias"])

    def dense_head(self, i, feat, dpt, img_meta, proposal=None):
        """DenseHead.forward (DenseHead.py:50-84). feat [1,N,C,h,w], dpt [1,N,D,h,w]."""
        pre = f"base_heads.{i}"
        nx, ny, nz = self.cfg["n_voxels_list"][i]
        C = self.cfg["embed_dims"]
        n_vox = nx * ny * nz
        if proposal is None:
            proposal = torch.ones(n_vox)
        idx = torch.nonzero(proposal > 0).view(-1)
        ref3d = self.sd[pre + ".ref_3d"][self.sd[pre + ".vox_coords"][idx, 3]]
        _, N, _, h, w = feat.shape
        with self._t("flatten_permute"):
            feat_flat = feat[0].flatten(2).permute(0, 2, 1).contiguous()          # transformer.py:154-169
            dist_flat = dpt[0].flatten(2).permute(0, 2, 1).contiguous()
        with self._t("project_compact"):
            ref_cam, mask = self.project(ref3d.float(), img_meta)
        x = self.layer(pre + ".cross_transformer.encoder.layers.0", feat_flat, dist_flat, h, w, ref_cam, mask.bool())
        vol = torch.zeros(n_vox, C)
        vol[idx] = x[0]
        return vol.view(nx, ny, nz, C).permute(3, 0, 1, 2).unsqueeze(0), dict(ref_cam=ref_cam, mask=mask, idx=idx)

    # ---------------------------------------------------------------- whole head
    def adaptive_sparse_head(self, mlvl_feats, img_meta, mlvl_dpt_dists, return_aux=False):
        """AdaptiveSparseHead.forward (AdaptiveSparseHead.py:43-93)."""
        n_lvl = len(self.cfg["n_voxels_list"])
        volumes, occ_list, aux = [], [], []
        mask = None
        for i in range(n_lvl):
            ds = 4 * (2 ** (n_lvl - 1 - i))
            h, w = img_meta["img_shape"][0] // ds, img_meta["img_shape"][1] // ds
            k = n_lvl - 1 - i
            feat = mlvl_feats[k][:, :, :, :h, :w]
            dpt = mlvl_dpt_dists[k][:, :, :, :h, :w]
            if i == 0:
                v, a = self.dense_head(0, feat, dpt, img_meta)
            else:
                with self._t("upsample_occ_topk"):
                    up = F.interpolate(volumes[-1], scale_factor=2, mode="trilinear", align_corners=False)
                    occ = torch.sigmoid(self.lin(up.permute(0, 2, 3, 4, 1), f"occ_pred_heads.{i - 1}.0")).reshape(1, -1)
                    occ_list.append(occ)
                    # topk_wo_grad (AdaptiveSparseHead.py:9-13): torch.topk's order among EQUAL scores is implementation-
                    # defined and voxels no camera sees carry bit-identical scores, so the tie rule is pinned here
                    # (lowest flat index first) -- the same set as torch.topk whenever the cut is not an exact tie
                    _, _, mask = self.ops.topk_select(occ.contiguous(), self.cfg["topk_list"][i - 1], want_mask=True)
                v, a = self.dense_head(i, feat, dpt, img_meta, proposal=mask)
                a["occ"] = occ
                v = up + v
            volumes.append(v)
            aux.append(a)
        nx, ny, nz = self.cfg["n_voxels_list"][-1]
        valid = mask.view(nx, ny, nz).bool().long()[None, None]
        occ_preds = torch.cat(occ_list[::-1], dim=1)
        if return_aux:
            return volumes[-1], valid, occ_preds, aux, volumes
        return volumes[-1], valid, occ_preds

    def coarse_topk_gap(self, mlvl_feats, img_meta, mlvl_dpt_dists):
        """Gap between the k-th and (k+1)-th occupancy score at the FIRST top-k of AdaptiveSparseHead (levels 0 and 1
        only: cheap).  A gap at rounding-noise level means two correct fp32 implementations may refine different voxels."""
        n_lvl = len(self.cfg["n_voxels_list"])
        ds = 4 * (2 ** (n_lvl - 1))
        h, w = img_meta["img_shape"][0] // ds, img_meta["img_shape"][1] // ds
        v0, _ = self.dense_head(0, mlvl_feats[n_lvl - 1][:, :, :, :h, :w], mlvl_dpt_dists[n_lvl - 1][:, :, :, :h, :w], img_meta)
        up = F.interpolate(v0, scale_factor=2, mode="trilinear", align_corners=False)
        occ = torch.sigmoid(self.lin(up.permute(0, 2, 3, 4, 1), "occ_pred_heads.0.0")).flatten()
        srt = occ.sort(descending=True).values
        k = self.cfg["topk_list"][0]
        return float(srt[k - 1] - srt[k])

    # ---------------------------------------------------------------- neck + head
    def _bn(self, x, pre):
        return F.batch_norm(x, self.sd[pre + ".running_mean"], self.sd[pre + ".running_var"],
                            self.sd[pre + ".weight"], self.sd[pre + ".bias"], False, 0.0, 1e-5)

    def _block(self, x, pre, stride):
        out = F.relu(self._bn(F.conv3d(x, self.sd[pre + ".conv1.weight"], None, stride, 1), pre + ".norm1"))
        out = self._bn(F.conv3d(out, self.sd[pre + ".conv2.weight"], None, 1, 1), pre + ".norm2")
        if stride != 1:
            x = self._bn(F.conv3d(x, self.sd[pre + ".downsample.0.weight"], None, stride), pre + ".downsample.1")
        return F.relu(out + x)

    def neck(self, x, prefix="", n_scales=3):
        """FastIndoorImVoxelNeck.forward, eval-mode BatchNorm (imvoxelnet.py:22-34)."""
        p = prefix
        skips = []
        for i in range(n_scales):
            x = self._block(x, f"{p}down_layer_{i}.0", 1 if i == 0 else 2)
            skips.append(x)
        outs = []
        for i in range(n_scales - 1, -1, -1):
            if i < n_scales - 1:
                u = f"{p}up_block_{i + 1}"
                x = F.relu(self._bn(F.conv_transpose3d(x, self.sd[u + ".0.weight"], None, 2), u + ".1"))
                x = F.relu(self._bn(F.conv3d(x, self.sd[u + ".3.weight"], None, 1, 1), u + ".4"))
                x = skips[i] + x
            o = f"{p}out_block_{i}"
            outs.append(F.relu(self._bn(F.conv3d(x, self.sd[o + ".0.weight"], None, 1, 1), o + ".1")))
        return outs[::-1]

    def head(self, feats, prefix=""):
        """forward_single per scale (imvoxel_head_v2.py:348-353 / :469-477)."""
        p = prefix
        ctr, reg, cls = [], [], []
        for i, x in enumerate(feats):
            ctr.append(F.conv3d(x, self.sd[p + "centerness_conv.weight"], None, 1, 1))
            r = F.conv3d(x, self.sd[p + "reg_conv.weight"], None, 1, 1)
            s = self.sd[f"{p}scales.{i}.scale"]
            if self.cfg["head"] == "scannet":
                r = torch.exp(r * s)
            else:
                r = torch.cat((torch.exp(r[:, :6] * s), r[:, 6:]), dim=1)
            reg.append(r)
            cls.append(F.conv3d(x, self.sd[p + "cls_conv.weight"], self.sd[p + "cls_conv.bias"], 1, 1))
        return ctr, reg, cls

    def decode(self, ctr, reg, cls, valid, img_meta, voxel_size):
        """get_bboxes / _get_bboxes_single up to (not including) NMS (imvoxel_head_v2.py:248-315)."""
        n_classes = self.cfg["n_classes"]
        nms_pre = self.cfg["nms_pre"]
        origin = torch.tensor(img_meta["lidar2img"]["origin"])
        boxes, scores_all = [], []
        for i, (c, r, s) in enumerate(zip(ctr, reg, cls)):
            v = F.interpolate(valid, size=c.shape[-3:], mode="trilinear").round().bool()[0]
            size = torch.tensor(c.shape[-3:])
            vs = torch.tensor(voxel_size) * (2 ** i)
            grid = torch.stack(torch.meshgrid([torch.arange(size[0]), torch.arange(size[1]), torch.arange(size[2])],
                                              indexing="ij"))
            pts = (grid * vs.view(3, 1, 1, 1) + (origin - size / 2.0 * vs).view(3, 1, 1, 1)).reshape(3, -1).t()
            c = c[0].permute(1, 2, 3, 0).reshape(-1).sigmoid()
            r = r[0].permute(1, 2, 3, 0).reshape(-1, r.shape[1])
            sc = s[0].permute(1, 2, 3, 0).reshape(-1, n_classes).sigmoid()
            sc = sc * c[:, None] * v.permute(1, 2, 3, 0).reshape(-1)[:, None]
            mx, _ = sc.max(dim=1)
            if len(sc) > nms_pre > 0:
                _, ids = mx.topk(nms_pre)
                r, sc, pts = r[ids], sc[ids], pts[ids]
            if self.cfg["head"] == "scannet":
                b = torch.stack([pts[:, 0] - r[:, 0], pts[:, 1] - r[:, 2], pts[:, 2] - r[:, 4],
                                 pts[:, 0] + r[:, 1], pts[:, 1] + r[:, 3], pts[:, 2] + r[:, 5]], -1)
            else:
                shift = torch.stack(((r[:, 1] - r[:, 0]) / 2, (r[:, 3] - r[:, 2]) / 2, (r[:, 5] - r[:, 4]) / 2), -1)
                cs, sn = torch.cos(r[:, 6]), torch.sin(r[:, 6])
                rot = torch.stack([shift[:, 0] * cs - shift[:, 1] * sn, shift[:, 0] * sn + shift[:, 1] * cs, shift[:, 2]], -1)
                b = torch.cat((pts + rot, torch.stack((r[:, 0] + r[:, 1], r[:, 2] + r[:, 3], r[:, 4] + r[:, 5]), -1),
                               r[:, 6:7]), -1)
            boxes.append(b)
            scores_all.append(sc)
        return torch.cat(boxes), torch.cat(scores_all)
