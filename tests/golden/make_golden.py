#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own Python for the hot-path
modules.  Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite read the committed .npz files, never this script's imports.

How the reference is made importable here (nothing of it is copied into the repo):
  * its third-party stack (mmcv / mmdet / mmdet3d / cv2 / torchvision) is absent, so stub
    modules are injected into ``sys.modules``; the registry / BaseModule / FFN / Scale pieces
    come from a PRIVATE second copy of ``sgcdet_amd/mmcv_lite.py`` (fresh registries, so the
    reference's classes do not collide with the product's);
  * ``dfa3D._ext`` (CUDA-only in the reference) is served by the CPU oracle
    (``oracle/libsgc_oracle.so``) -- the module-level goldens therefore pin the Python glue
    (projection, rebatching, slicing, MHA, LayerNorm/FFN, top-k, neck, head, decode), while
    the kernel arithmetic itself is pinned by tests/test_oracle_identity.py;
  * the reference only assigns its op output under ``if torch.cuda.is_available() and
    value.is_cuda`` (deformable_cross_attention.py:108,482); the generator loads that one
    file with the guard replaced by ``True`` in memory.

Usage:  python tests/golden/make_golden.py      (writes next to this file)
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import oracle  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    spec = importlib.util.spec_from_file_location("_refstub_mmcv_lite", os.path.join(ROOT, "sgcdet_amd", "mmcv_lite.py"))
    ml = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ml)
    oops = oracle.ops()

    class _OracleExt:  # dfa3D._ext served by the CPU oracle
        @staticmethod
        def ms_depth_score_sample_forward(value, shapes, lsi, loc, im2col_step=64):
            return oops.depth_score_forward(value.contiguous(), shapes.contiguous(), lsi.contiguous(), loc.contiguous())

        @staticmethod
        def ms_depth_score_sample_backward(value, shapes, lsi, loc, grad_output, grad_value, grad_loc, im2col_step=64):
            oops.depth_score_backward(value, shapes, lsi, loc, grad_output, grad_value, grad_loc)

        @staticmethod
        def wms_deform_attn_forward(value, shapes, lsi, loc, attn, score, im2col_step=64):
            return oops.wms_forward(value.contiguous(), shapes, lsi, loc, attn.contiguous(), score)

        @staticmethod
        def wms_deform_attn_backward(value, shapes, lsi, loc, attn, score, grad_output, grad_value, grad_loc,
                                     grad_attn, grad_score, im2col_step=64):
            oops.wms_backward(value, shapes, lsi, loc, attn, score, grad_output, grad_value, grad_loc, grad_attn,
                              grad_score)

    class _Missing:
        def __getattr__(self, k):
            raise RuntimeError(f"mmcv _ext.{k} is not available in the golden generator")

    ext_loader = types.SimpleNamespace(load_ext=lambda name, funcs: _Missing())
    dfa_loader = types.SimpleNamespace(load_ext=lambda name, funcs: _OracleExt())

    def deprecated_api_warning(*a, **k):
        return lambda fn: fn

    _mod("cv2")
    _mod("torchvision"); _mod("torchvision.transforms")
    _mod("torchvision.transforms.functional", rotate=None)
    _mod("mmcv", ConfigDict=ml.ConfigDict, deprecated_api_warning=deprecated_api_warning)
    _mod("mmcv.cnn", xavier_init=ml.xavier_init, constant_init=ml.constant_init, Linear=torch.nn.Linear,
         build_activation_layer=ml.build_activation_layer, build_norm_layer=ml.build_norm_layer, Scale=ml.Scale,
         bias_init_with_prob=ml.bias_init_with_prob, normal_init=ml.normal_init)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.registry", ATTENTION=ml.ATTENTION, FEEDFORWARD_NETWORK=ml.FEEDFORWARD_NETWORK,
         POSITIONAL_ENCODING=ml.Registry("pe"), TRANSFORMER_LAYER=ml.TRANSFORMER_LAYER,
         TRANSFORMER_LAYER_SEQUENCE=ml.TRANSFORMER_LAYER_SEQUENCE)
    _mod("mmcv.cnn.bricks.transformer", build_attention=ml.build_attention,
         build_feedforward_network=ml.build_feedforward_network, TransformerLayerSequence=ml.TransformerLayerSequence,
         build_transformer_layer=ml.build_transformer_layer,
         build_transformer_layer_sequence=ml.build_transformer_layer_sequence)
    _mod("mmcv.runner", force_fp32=ml.force_fp32, auto_fp16=ml.auto_fp16)
    _mod("mmcv.runner.base_module", BaseModule=ml.BaseModule, ModuleList=ml.ModuleList, Sequential=ml.Sequential)
    _mod("mmcv.utils", ext_loader=ext_loader, TORCH_VERSION=torch.__version__, digit_version=lambda v: (1, 10))
    _mod("mmcv.ops")
    _mod("mmcv.ops.multi_scale_deform_attn", multi_scale_deformable_attn_pytorch=None,
         MultiScaleDeformableAttention=object)
    _mod("mmdet")
    _mod("mmdet.models", HEADS=ml.HEADS, NECKS=ml.NECKS, DETECTORS=ml.DETECTORS, build_head=ml.build_head,
         build_neck=ml.build_neck, build_backbone=lambda cfg: None)
    _mod("mmdet.models.utils", build_transformer=ml.build_transformer)
    _mod("mmdet.models.utils.builder", TRANSFORMER=ml.TRANSFORMER)
    _mod("mmdet.models.builder", HEADS=ml.HEADS, build_loss=lambda cfg: None)
    _mod("mmdet.core", multi_apply=ml.multi_apply, reduce_mean=lambda x: x)
    _mod("dfa3D", ext_loader=dfa_loader)
    # mmdet3d: the vendored rotation helper is loaded from the reference tree itself
    _mod("mmdet3d"); _mod("mmdet3d.core")
    _mod("mmdet3d.core.utils", array_converter=lambda **kw: (lambda fn: fn))
    us = importlib.util.spec_from_file_location(
        "mmdet3d.core.bbox.structures.utils",
        os.path.join(REF, "packages/mmdetection3d/mmdet3d/core/bbox/structures/utils.py"))
    um = importlib.util.module_from_spec(us)
    us.loader.exec_module(um)
    _mod("mmdet3d.core.bbox"); _mod("mmdet3d.core.bbox.structures", rotation_3d_in_axis=um.rotation_3d_in_axis)
    _mod("mmdet3d.core.post_processing", aligned_3d_nms=None, box3d_multiclass_nms=None)
    # the plugin packages, WITHOUT running their __init__ (which would import datasets, losses ...)
    for pkg, sub in [("mmdet3d_plugin", "mmdet3d_plugin"), ("mmdet3d_plugin.models", "mmdet3d_plugin/models"),
                     ("mmdet3d_plugin.models.im2voxel", "mmdet3d_plugin/models/im2voxel"),
                     ("mmdet3d_plugin.models.im2voxel.transformer_utils", "mmdet3d_plugin/models/im2voxel/transformer_utils"),
                     ("mmdet3d_plugin.models.detectors", "mmdet3d_plugin/models/detectors"),
                     ("mmdet3d_plugin.models.necks", "mmdet3d_plugin/models/necks"),
                     ("mmdet3d_plugin.models.dense_heads", "mmdet3d_plugin/models/dense_heads")]:
        m = _mod(pkg)
        m.__path__ = [os.path.join(REF, sub)]
    # deformable_cross_attention.py with the CUDA-only guard neutralised (in memory only)
    name = "mmdet3d_plugin.models.im2voxel.transformer_utils.deformable_cross_attention"
    path = os.path.join(REF, "mmdet3d_plugin/models/im2voxel/transformer_utils/deformable_cross_attention.py")
    src = open(path).read().replace("torch.cuda.is_available() and value.is_cuda", "True")
    m = types.ModuleType(name)
    m.__file__ = path
    m.__package__ = "mmdet3d_plugin.models.im2voxel.transformer_utils"
    sys.modules[name] = m
    exec(compile(src, path, "exec"), m.__dict__)
    return ml


def small_img_meta(n_views, seed):
    from sgcdet_amd.scene import camera_ring
    rng = np.random.RandomState(seed)
    ext, _ = camera_ring(n_views, rng)
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = np.array([[290.0, 0, 160.0], [0, 290.0, 120.0], [0, 0, 1]], dtype=np.float32)
    return dict(img_shape=(59, 80, 3), ori_shape=(240, 320, 3),
                lidar2img=dict(extrinsic=ext, intrinsic=K, origin=np.array([0.0, 0.0, 0.5], dtype=np.float32)))


def meta_arrays(meta):
    return dict(meta_extrinsic=np.stack(meta["lidar2img"]["extrinsic"]), meta_intrinsic=meta["lidar2img"]["intrinsic"],
                meta_origin=meta["lidar2img"]["origin"], meta_img_shape=np.array(meta["img_shape"]),
                meta_ori_shape=np.array(meta["ori_shape"]))


def randomize_(module, gen, scale=0.15):
    """Non-trivial deterministic weights: default init leaves offsets/attention data-independent."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("sampling_offsets.bias") or n.endswith("sampling_offsets_depth.bias"):
                p.add_(torch.randn(p.shape, generator=gen) * 0.3)
            elif n.endswith("norms.0.weight") or n.endswith("norms.1.weight") or ".norm" in n and n.endswith("weight") \
                    or n.endswith(".1.weight") and p.dim() == 1 or n.endswith(".4.weight") and p.dim() == 1:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif n.endswith("scale"):
                p.copy_(torch.tensor(1.0) + 0.1 * torch.randn((), generator=gen))
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * scale)
        for n, b in module.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
            elif n.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=gen))


def sd_arrays(module, prefix="sd::"):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def voxel_head_cfg(C, grids, sizes, topk):
    xf = dict(type="PerceptionTransformer_DFA3D", embed_dims=C, encoder=dict(
        type="VoxFormerEncoder_DFA3D", num_layers=1, return_intermediate=False, dbound=[0.2, 5, 0.4],
        transformerlayers=dict(
            type="VoxFormerLayer",
            attn_cfgs=[dict(type="DeformCrossAttention_DFA3D", embed_dims=C, inter_view_aggregation="attn", dropout=0,
                            deformable_attention=dict(type="MSDeformableAttention3D_DFA3D", embed_dims=C, num_heads=8,
                                                      num_points=4, num_levels=1, im2col_step=128))],
            ffn_cfgs=dict(type="FFN", embed_dims=C, feedforward_channels=C * 2, num_fcs=2, ffn_drop=0.1,
                          act_cfg=dict(type="ReLU", inplace=True)),
            operation_order=("cross_attn", "norm", "ffn", "norm"))))
    heads = [dict(type="DenseHead", voxel_size=s, n_voxels=g, embed_dims=C, cross_transformer=xf)
             for s, g in zip(sizes, grids)]
    return dict(type="AdaptiveSparseHead", embed_dims=C, topk_list=topk, voxel_size_list=sizes, n_voxels_list=grids,
                base_head_configs=heads)


def main():
    ml = install_stubs()
    tu = "mmdet3d_plugin.models.im2voxel.transformer_utils."
    fn3d = importlib.import_module(tu + "multi_scale_3ddeformable_attn_function")
    importlib.import_module(tu + "encoder")
    importlib.import_module(tu + "transformer")
    importlib.import_module("mmdet3d_plugin.models.im2voxel.DenseHead")
    importlib.import_module("mmdet3d_plugin.models.im2voxel.AdaptiveSparseHead")
    neck_mod = importlib.import_module("mmdet3d_plugin.models.necks.imvoxelnet")
    head_mod = importlib.import_module("mmdet3d_plugin.models.dense_heads.imvoxel_head_v2")
    gen = torch.Generator().manual_seed(1234)

    # ---- (A) the reference's autograd composition of the two operators ------------------
    B, M, Cm, D, Q, P = 2, 8, 4, 12, 37, 4
    levels = [(7, 9), (4, 5)]
    S = sum(h * w for h, w in levels)
    shapes3 = torch.tensor([[h, w, D] for h, w in levels])
    lsi = torch.tensor([0, levels[0][0] * levels[0][1]])
    value = torch.randn(B, S, M, Cm, generator=gen, requires_grad=True)
    dist = torch.randn(B, S, M, D, generator=gen).mul(2).softmax(-1).detach().requires_grad_()
    loc = (torch.rand(B, Q, M, 2, P, 3, generator=gen) * 1.3 - 0.15).requires_grad_()
    attn = torch.rand(B, Q, M, 2, P, generator=gen, requires_grad=True)
    out, score = fn3d.MultiScale3DDeformableAttnFunction_fp32.apply(value, dist, shapes3, lsi, loc, attn, 64)
    go = torch.randn(out.shape, generator=gen)
    gv, gd, gl, ga = torch.autograd.grad(out, [value, dist, loc, attn], go)
    save("op_autograd", value=value, dist=dist, shapes3=shapes3, lsi=lsi, loc=loc, attn=attn, out=out, score=score,
         grad_out=go, grad_value=gv, grad_dist=gd, grad_loc=gl, grad_attn=ga)

    # ---- (B..D) voxel head on a small scene --------------------------------------------
    C, N = 32, 3
    grids = [(4, 4, 2), (8, 8, 4), (16, 16, 8)]
    sizes = [(.64, .64, .8), (.32, .32, .4), (.16, .16, .2)]
    topk = [64, 512]
    meta = small_img_meta(N, 1234)
    head = ml.build_head(voxel_head_cfg(C, grids, sizes, topk)).eval()
    randomize_(head, gen)
    feats = [torch.randn(1, N, C, h, w, generator=gen) for h, w in [(15, 20), (8, 10), (4, 5), (2, 3)]]
    dpt = torch.randn(1, N, D, 15, 20, generator=gen).mul(2).softmax(2)
    import torch.nn.functional as F
    dpts = [dpt, F.interpolate(dpt, scale_factor=(1, 0.5, 0.5), mode="nearest"),
            F.interpolate(dpt, scale_factor=(1, 0.25, 0.25), mode="nearest")]
    with torch.no_grad():
        # (B) point_sampling of the finest DenseHead's encoder on all its voxels
        dh = head.base_heads[2]
        enc = dh.cross_transformer.encoder
        ref_cam, mask = enc.point_sampling(dh.ref_3d[None, None], img_meta=meta)
        save("point_sampling", ref_3d=dh.ref_3d, ref_cam=ref_cam, mask=mask.to(torch.uint8), dbound=np.array([0.2, 5.0]),
             **meta_arrays(meta))
        # (C) one dense level (coarsest head, all voxels)
        f0 = feats[2][:, :, :, :59 // 16, :80 // 16]
        d0 = dpts[2][:, :, :, :59 // 16, :80 // 16]
        vol0 = head.base_heads[0]([f0], meta, mlvl_dpt_dists=[d0])
        # (D) the whole coarse-to-fine head
        volume, valid, occ = head(feats, meta, dpts)
    save("voxel_head", feat0=feats[0], feat1=feats[1], feat2=feats[2], feat3=feats[3], dpt=dpt, level0_volume=vol0,
         volume=volume, valid=valid, occ=occ, grids=np.array(grids), sizes=np.array(sizes), topk=np.array(topk),
         **meta_arrays(meta), **sd_arrays(head))

    # ---- (E) neck ------------------------------------------------------------------------
    neck = neck_mod.FastIndoorImVoxelNeck(in_channels=16, n_blocks=[1, 1, 1], out_channels=8).eval()
    randomize_(neck, gen, scale=0.08)
    x = torch.randn(1, 16, 8, 8, 4, generator=gen)
    with torch.no_grad():
        outs = neck(x)
    save("neck", x=x, out0=outs[0], out1=outs[1], out2=outs[2], **sd_arrays(neck))

    # ---- (F) heads: forward_single + decode up to NMS --------------------------------------
    test_cfg = ml.ConfigDict(nms_pre=50, iou_thr=.25, score_thr=.01)
    for tag, cls_name, n_cls, n_reg in (("scannet", "ScanNetImVoxelHeadV2", 18, 6), ("sunrgbd", "SunRgbdImVoxelHeadV2", 17, 7)):
        bh = getattr(head_mod, cls_name)(n_classes=n_cls, n_channels=8, n_reg_outs=n_reg, n_scales=3, limit=27,
                                         centerness_topk=18, test_cfg=test_cfg).eval()
        bh.voxel_size = (.16, .16, .2)
        randomize_(bh, gen, scale=0.08)
        bh._nms = lambda bboxes, scores, img_meta: (bboxes, scores, None)
        v = (torch.rand(1, 1, 8, 8, 4, generator=gen) > 0.5).float()
        fs = [torch.randn(1, 8, 8, 8, 4, generator=gen), torch.randn(1, 8, 4, 4, 2, generator=gen),
              torch.randn(1, 8, 2, 2, 1, generator=gen)]
        with torch.no_grad():
            ctr, reg, cls = bh(fs)
            boxes, scores, _ = bh.get_bboxes(ctr, reg, cls, v, [meta])[0]
        save("head_" + tag, f0=fs[0], f1=fs[1], f2=fs[2], valid=v, ctr0=ctr[0], ctr1=ctr[1], ctr2=ctr[2], reg0=reg[0],
             reg1=reg[1], reg2=reg[2], cls0=cls[0], cls1=cls[1], cls2=cls[2], boxes=boxes, scores=scores,
             voxel_size=np.array(bh.voxel_size), nms_pre=np.array(50), **meta_arrays(meta), **sd_arrays(bh))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    torch.set_num_threads(1)
    main()
