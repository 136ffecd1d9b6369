"""GPU parity of the product modules (sgcdet_amd.plugin, HIP kernels through the C ABI):
  * against the golden vectors produced by the reference's own Python, with the golden
    state_dict loaded by the reference's parameter names (checkpoint compatibility);
  * against the CPU oracle on larger seeded scenes (indices bit-exact, features 1e-3)."""
import pytest
import torch

from golden_util import load, img_meta, depth_pyramid, max_err

pytestmark = pytest.mark.gpu


def _build_voxel_head(d, C=32):
    import sgcdet_amd.plugin  # noqa: F401  (registers the type= names)
    from sgcdet_amd.mmcv_lite import build_head
    from sgcdet_amd.scene import model_config
    w = dict(embed_dims=C, n_voxels_list=[tuple(int(v) for v in g) for g in d["grids"]],
             voxel_size_list=[tuple(float(v) for v in s) for s in d["sizes"]], topk_list=[int(v) for v in d["topk"]],
             head="ScanNetImVoxelHeadV2", n_classes=18, n_reg_outs=6)
    return build_head(model_config(w)["voxel_head"])


def test_voxel_head_against_reference_golden():
    d, sd = load("voxel_head")
    head = _build_voxel_head(d)
    missing, unexpected = head.load_state_dict(sd, strict=True), None
    head = head.cuda().eval()
    meta = img_meta(d)
    feats = [d[f"feat{i}"].cuda() for i in range(4)]
    dpts = [t.cuda() for t in depth_pyramid(d["dpt"])]
    with torch.no_grad():
        lvl0 = head.base_heads[0]([feats[2][:, :, :, :59 // 16, :80 // 16]], meta,
                                  mlvl_dpt_dists=[dpts[2][:, :, :, :59 // 16, :80 // 16]])
        volume, valid, occ = head(feats, meta, dpts)
    assert max_err(lvl0, d["level0_volume"]) < 1e-4
    assert torch.equal(valid.cpu(), d["valid"])            # selected voxel set: bit-exact
    assert valid.dtype == torch.int64 and tuple(valid.shape) == (1, 1, 16, 16, 8)
    assert max_err(occ, d["occ"]) < 1e-4
    assert max_err(volume, d["volume"]) < 1e-3            # north-star tolerance (observed ~1e-5)
    assert max_err(volume, d["volume"]) < 2e-4


def test_point_sampling_against_reference_golden():
    d, sd = load("voxel_head")
    ps, _ = load("point_sampling")
    head = _build_voxel_head(d).cuda().eval()
    enc = head.base_heads[2].cross_transformer.encoder
    ref_cam, mask = enc.point_sampling(ps["ref_3d"].cuda()[None, None], img_meta=img_meta(ps))
    assert tuple(ref_cam.shape) == tuple(ps["ref_cam"].shape) and mask.dtype == torch.bool
    assert torch.equal(mask.cpu().to(torch.uint8), ps["mask"])
    assert max_err(ref_cam, ps["ref_cam"]) < 2e-6


def test_point_sampling_near_plane_rule_against_reference_golden():
    """tests/golden/point_sampling_near.npz: in-image voxels closer than d_near are invisible in the reference (its depth
    test reads the slice already overwritten with the normalised depth, encoder.py:203-213); bit-exact mask."""
    d, sd = load("voxel_head")
    ps, _ = load("point_sampling_near")
    head = _build_voxel_head(d).cuda().eval()
    enc = head.base_heads[2].cross_transformer.encoder
    ref_cam, mask = enc.point_sampling(ps["ref_3d"].cuda()[None, None], img_meta=img_meta(ps))
    assert int(ps["n_band"]) >= 8
    assert torch.equal(mask.cpu().to(torch.uint8), ps["mask"])
    g = ps["ref_cam"]
    inside = (g[..., 0] > 1e-5) & (g[..., 0] < 1 - 1e-5) & (g[..., 1] > 1e-5) & (g[..., 1] < 1 - 1e-5)
    assert max_err(ref_cam.cpu()[inside], g[inside]) < 2e-6


def test_neck_and_heads_against_reference_golden():
    import sgcdet_amd.plugin as P
    d, sd = load("neck")
    neck = P.FastIndoorImVoxelNeck(in_channels=16, n_blocks=[1, 1, 1], out_channels=8)
    neck.load_state_dict(sd, strict=True)
    neck = neck.cuda().eval()
    with torch.no_grad():
        outs = neck(d["x"].cuda())
    for i, o in enumerate(outs):
        assert max_err(o, d[f"out{i}"]) < 1e-3 * max(1.0, d[f"out{i}"].abs().max().item())
    for tag, cls, n_cls, n_reg in (("scannet", P.ScanNetImVoxelHeadV2, 18, 6), ("sunrgbd", P.SunRgbdImVoxelHeadV2, 17, 7)):
        d, sd = load("head_" + tag)
        bh = cls(n_classes=n_cls, n_channels=8, n_reg_outs=n_reg, n_scales=3, limit=27, centerness_topk=18,
                 test_cfg=dict(nms_pre=int(d["nms_pre"]), iou_thr=.25, score_thr=.01))
        bh.load_state_dict(sd, strict=True)
        bh.voxel_size = tuple(float(v) for v in d["voxel_size"])
        bh = bh.cuda().eval()
        with torch.no_grad():
            ctr, reg, cl = bh([d["f0"].cuda(), d["f1"].cuda(), d["f2"].cuda()])
            valids = [torch.nn.Upsample(size=x.shape[-3:], mode="trilinear")(d["valid"].cuda()).round().bool()[0] for x in ctr]
            boxes, scores = bh.decode_candidates([x[0] for x in ctr], [x[0] for x in reg], [x[0] for x in cl], valids,
                                                 img_meta(d))
        for i in range(3):
            assert max_err(ctr[i], d[f"ctr{i}"]) < 1e-4 and max_err(cl[i], d[f"cls{i}"]) < 1e-4
            assert max_err(reg[i], d[f"reg{i}"]) < 1e-4 * max(1.0, d[f"reg{i}"].abs().max().item())
        assert boxes.shape == d["boxes"].shape
        # top-k candidate ORDER may differ on equal scores; compare as sorted sets
        key_g = torch.argsort(scores.max(1).values.cpu(), stable=True)
        key_r = torch.argsort(d["scores"].max(1).values, stable=True)
        assert max_err(scores.cpu()[key_g], d["scores"][key_r]) < 1e-5
        assert max_err(boxes.cpu()[key_g], d["boxes"][key_r]) < 1e-3


@pytest.mark.parametrize("name,n_views,pq", [("cfg1_plumbing", 2, "auto"), ("cfg2_scannet", 6, "auto"), ("cfg3_arkit", 4, "auto"),
                                             ("cfg4_scannet200_large", 3, "auto"), ("cfg5_arkit_large", 3, "auto"),
                                             ("cfg2_scannet", 6, True), ("cfg5_arkit_large", 3, True)])
def test_hot_path_against_oracle(name, n_views, pq):
    """Seeded synthetic scene of the BASELINE shapes (views reduced for cfg2 so the CPU oracle
    finishes in seconds) -- volume / neck / head tensors within 1e-3, masks and top-k sets bit-exact.
    ``pq`` True: the projected-query form of the inter-view attention (sgc_view_attend_pq) forced on at these small view
    counts ("auto" takes it from 48 views on: the full-view-count tests of configs 3 / 4 / 5 below run it) -- against the same oracle, which
    restates nn.MultiheadAttention on the reference's dense slots."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    from oracle.ref_path import RefPath
    w = workload(name)
    torch.manual_seed(7)
    det = build_detector(model_config(w)).eval()
    n_pq = 0
    for m in det.modules():
        if hasattr(m, "projected_query"):
            m.projected_query = pq
            n_pq += 1
    assert n_pq == 3
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():   # default init leaves the attention data-independent: perturb deterministically
        for n, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.05)
    feats, dpt, meta = make_scene(n_views, w["embed_dims"], kind=w["kind"], seed=5)
    rp = RefPath(det.voxel_head.state_dict(), dict(embed_dims=w["embed_dims"], n_voxels_list=w["n_voxels_list"],
                                                   voxel_size_list=w["voxel_size_list"], topk_list=w["topk_list"],
                                                   dbound=(0.2, 5.0), num_heads=8, num_points=4))
    vol_c, valid_c, occ_c = rp.adaptive_sparse_head(feats, meta, depth_pyramid(dpt))
    det = det.cuda()
    with torch.no_grad():
        r = det.forward_features([f.cuda() for f in feats], [meta], dpt.cuda())
    from oracle.compare import check_sparse_head
    n_fin = w["n_voxels_list"][-1][0] * w["n_voxels_list"][-1][1] * w["n_voxels_list"][-1][2]
    res = check_sparse_head(r["volume"], r["valid"], r["occ"], vol_c, valid_c, occ_c, n_fin, w["topk_list"])
    assert res["tie_flips"] <= 4       # selected sets bit-exact up to near ties at the cut
    if pq is True:                     # the form under test really ran (call log of the library binding)
        from sgcdet_amd import ext
        ops = ext.ops()
        ops.event_log, ops.event_names = [], {"sgc_view_attend_pq", "sgc_view_attend"}
        with torch.no_grad():
            det.forward_features([f.cuda() for f in feats], [meta], dpt.cuda())
        torch.cuda.synchronize()
        names = [e[0] for e in ops.event_log]
        ops.event_log, ops.event_names = None, None
        assert names.count("sgc_view_attend_pq") == 3 and "sgc_view_attend" not in names
    # neck + head on the oracle's volume (torch-CPU conv3d) vs the GPU path
    rp2 = RefPath({**{"neck." + k: v for k, v in det.neck_3d.state_dict().items()},
                   **{"head." + k: v for k, v in det.bbox_head.state_dict().items()}},
                  dict(head="scannet" if w["head"].startswith("ScanNet") else "sunrgbd", n_classes=w["n_classes"],
                       nms_pre=1000))
    # neck + head: the HIP modules on the ORACLE's volume (a near-tie flip in the top-k changes a voxel's whole 3x3x3
    # neighbourhood, so the product's own volume is only comparable when tie_flips == 0; the oracle's volume always is)
    _check_neck_head_on(det, vol_c, rp2)
    if res["tie_flips"] == 0:
        feats3d = rp2.neck(vol_c, prefix="neck.")
        ctr, reg, cls = rp2.head(feats3d, prefix="head.")
        for a, b in zip(r["centerness"] + r["bbox_pred"] + r["cls_score"], ctr + reg + cls):
            assert max_err(a, b) < 1e-3 * max(1.0, b.abs().max().item())


def test_throughput_launch_geometry_keeps_parity():
    """conv_plan.set_throughput_mode (what bench.py selects for scenes in flight): the row GEMM on half the CUs is bit-identical;
    the coarser reduction splits of seven neck layers move the head tensors by fp32 summation order only (<= 1e-5 of their
    scale) -- and the path under that geometry passes the same oracle comparison as the default one."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.plugin.conv_plan import set_throughput_mode
    from sgcdet_amd.scene import make_scene, model_config, workload
    from oracle.ref_path import RefPath
    from oracle.compare import check_sparse_head
    w = workload("cfg2_scannet")
    torch.manual_seed(7)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.05)
    feats, dpt, meta = make_scene(6, w["embed_dims"], kind=w["kind"], seed=5)
    rp = RefPath(det.voxel_head.state_dict(), dict(embed_dims=w["embed_dims"], n_voxels_list=w["n_voxels_list"],
                                                   voxel_size_list=w["voxel_size_list"], topk_list=w["topk_list"],
                                                   dbound=(0.2, 5.0), num_heads=8, num_points=4))
    vol_c, valid_c, occ_c = rp.adaptive_sparse_head(feats, meta, depth_pyramid(dpt))
    det = det.cuda()
    det.use_graph = det.scene_graph = False            # a captured graph keeps the geometry it was captured with
    gf, gd = [f.cuda() for f in feats], dpt.cuda()
    with torch.no_grad():
        lat = det.forward_features(gf, [meta], gd)
        lat = {k: [t.clone() for t in v] if isinstance(v, (list, tuple)) else v.clone() for k, v in lat.items() if k != "feats"}
        try:
            set_throughput_mode(True)
            thr = det.forward_features(gf, [meta], gd)
            thr2 = det.forward_features(gf, [meta], gd)
            n_fin = w["n_voxels_list"][-1][0] * w["n_voxels_list"][-1][1] * w["n_voxels_list"][-1][2]
            res = check_sparse_head(thr["volume"], thr["valid"], thr["occ"], vol_c, valid_c, occ_c, n_fin, w["topk_list"])
            assert res["tie_flips"] <= 4
            rp2 = RefPath({**{"neck." + k: v for k, v in det.neck_3d.state_dict().items()},
                           **{"head." + k: v for k, v in det.bbox_head.state_dict().items()}},
                          dict(head="scannet", n_classes=w["n_classes"], nms_pre=1000))
            _check_neck_head_on(det, vol_c, rp2)                   # the neck / head under the throughput geometry vs the oracle
        finally:
            set_throughput_mode(False)
    assert torch.equal(thr["volume"], lat["volume"]) and torch.equal(thr["occ"], lat["occ"])      # view transform: same bits
    moved = 0.0
    for k in ("centerness", "bbox_pred", "cls_score"):
        for a, b, c in zip(thr[k], lat[k], thr2[k]):
            assert torch.equal(a, c)                                   # deterministic within the mode
            moved = max(moved, float((a - b).abs().max() / b.abs().max().clamp(min=1.0)))
    assert 0.0 < moved < 1e-5, moved                                   # a different summation order, nothing more


@pytest.mark.parametrize("name,n_views", [("cfg1_plumbing", 2), ("cfg2_scannet", 6)])
def test_hot_path_in_strict_fp32_mode(name, n_views):
    """The same scenes with --conv-mode f32 (exact fp32 products on v_mfma_f32_32x32x2_f32 for every Linear and
    convolution; the tiled gather behind a permuting copy): the volume must agree with the fp32 oracle at least as well
    as in the default bf16x3 mode, and the near-tie flips of the finest cut are counted in both modes -- exact products
    may not produce MORE flips than the 3-way split (AdaptiveSparseHead.py:9-13: top-k of fp32 scores)."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.plugin.conv_plan import set_conv_mode
    from sgcdet_amd.scene import make_scene, model_config, workload
    from oracle.ref_path import RefPath
    from oracle.compare import check_sparse_head
    w = workload(name)
    torch.manual_seed(7)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.05)
    feats, dpt, meta = make_scene(n_views, w["embed_dims"], kind=w["kind"], seed=5)
    rp = RefPath(det.voxel_head.state_dict(), dict(embed_dims=w["embed_dims"], n_voxels_list=w["n_voxels_list"],
                                                   voxel_size_list=w["voxel_size_list"], topk_list=w["topk_list"],
                                                   dbound=(0.2, 5.0), num_heads=8, num_points=4))
    vol_c, valid_c, occ_c = rp.adaptive_sparse_head(feats, meta, depth_pyramid(dpt))
    det = det.cuda()
    det.use_graph = det.scene_graph = False
    n_fin = w["n_voxels_list"][-1][0] * w["n_voxels_list"][-1][1] * w["n_voxels_list"][-1][2]
    res = {}
    try:
        for mode in ("bf16x3", "f32"):
            set_conv_mode(mode)
            with torch.no_grad():
                r = det.forward_features([f.cuda() for f in feats], [meta], dpt.cuda())
            res[mode] = check_sparse_head(r["volume"], r["valid"], r["occ"], vol_c, valid_c, occ_c, n_fin, w["topk_list"])
            if mode == "f32":                       # neck + head in strict mode on the product's own volume when no voxel flipped
                rp2 = _neck_head_oracle(det, w)
                _check_neck_head_on(det, vol_c, rp2)
    finally:
        set_conv_mode("bf16x3")
    assert res["f32"]["tie_flips"] <= res["bf16x3"]["tie_flips"] <= 4, res
    assert res["f32"]["max_err"] <= max(res["bf16x3"]["max_err"] * 1.5, 2e-5 * res["f32"]["scale"]), res


def _check_neck_head_on(det, vol_c, rp2):
    """FastIndoorImVoxelNeck + ImVoxelHeadV2 of the product (HIP convolutions) against the oracle's torch-CPU
    restatement on the SAME input volume: every head tensor within 1e-3 of its scale (north-star bar)."""
    feats3d = rp2.neck(vol_c, prefix="neck.")
    ctr, reg, cls = rp2.head(feats3d, prefix="head.")
    with torch.no_grad():
        outs = det._neck_head_eager(vol_c.cuda())
    got = list(outs[0]) + list(outs[1]) + list(outs[2])
    assert len(got) == len(ctr + reg + cls) == 9
    for a, b in zip(got, ctr + reg + cls):
        assert a.shape == b.shape
        assert max_err(a, b) < 1e-3 * max(1.0, b.abs().max().item())


def test_reduced_precision_modes_through_neck_and_head():
    """The opt-in one-product modes (`set_conv_mode("bf16")` / `("fp16")`, BASELINE.json configs #2 / #5) through the whole neck +
    head (twelve chained convolutions per scale) on the oracle's fp32 volume: every head tensor stays within 2^-5 (bf16) /
    2^-8 (fp16) of its scale against the fp32 oracle, and fp16 -- 11 significant bits against 8 at the same MFMA rate -- is at
    least 3x closer than bf16.  The view transform of these modes is covered by the kernel tests; its top-k selections are not
    comparable across arithmetic modes."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.plugin.conv_plan import set_conv_mode
    from sgcdet_amd.scene import model_config, workload
    w = workload("cfg1_plumbing")
    torch.manual_seed(21)
    det = build_detector(model_config(w)).eval()
    rp2 = _neck_head_oracle(det, w)
    g = torch.Generator().manual_seed(4)
    nx, ny, nz = w["n_voxels_list"][-1]
    vol = torch.randn(1, w["embed_dims"], nx, ny, nz, generator=g)
    feats3d = rp2.neck(vol, prefix="neck.")
    ctr, reg, cls = rp2.head(feats3d, prefix="head.")
    det = det.cuda()
    det.use_graph = det.scene_graph = False
    worst = {}
    try:
        for mode in ("bf16x3", "bf16", "fp16"):
            set_conv_mode(mode)
            with torch.no_grad():
                outs = det._neck_head_eager(vol.cuda())
            got = list(outs[0]) + list(outs[1]) + list(outs[2])
            worst[mode] = max(max_err(a, b) / max(1.0, b.abs().max().item()) for a, b in zip(got, ctr + reg + cls))
    finally:
        set_conv_mode("bf16x3")
    assert worst["bf16x3"] < 1e-4 and worst["bf16"] <= 2.0 ** -5 and worst["fp16"] <= 2.0 ** -8, worst
    assert worst["fp16"] * 3 <= worst["bf16"], worst


def test_fp16_mode_with_gradients_enabled_never_feeds_bf16_planes_to_the_half_mfma():
    """the training Functions pack bfloat16 weight planes (sgc_pack_conv_weight), which the fp16 arithmetic mode
    would read as IEEE half.  With gradients enabled in that mode (a) the Functions themselves refuse loudly, (b) the neck's
    training path takes torch's library convolutions instead, so its output and input gradient agree with the fp32-faithful
    mode to the library path's own tolerance rather than being garbage."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.functions import ChannelsLastConv3dFunction, LinearRowsFunction
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.plugin.conv_plan import set_conv_mode, train_conv_on_hip
    from sgcdet_amd.scene import model_config, workload
    w = workload("cfg1_plumbing")
    torch.manual_seed(5)
    det = build_detector(model_config(w)).cuda().train()
    nx, ny, nz = w["n_voxels_list"][-1]
    vol = torch.randn(1, w["embed_dims"], nx, ny, nz, device="cuda")
    outs = {}
    try:
        for mode in ("bf16x3", "fp16"):
            set_conv_mode(mode)
            torch.manual_seed(6)                               # same BatchNorm statistics path
            x = vol.clone().requires_grad_(True)
            assert train_conv_on_hip(x, [w["embed_dims"]]) == (mode == "bf16x3")
            y = det.neck_3d(x)
            loss = sum((t.float() ** 2).mean() for t in y)
            loss.backward()
            outs[mode] = ([t.detach().clone() for t in y], x.grad.detach().clone())
        rows = torch.randn(512, 64, device="cuda", requires_grad=True)
        wt = torch.randn(32, 64, 3, 3, 3, device="cuda", requires_grad=True)
        with pytest.raises(RuntimeError, match="inference-only"):
            ChannelsLastConv3dFunction.apply(rows, wt, (8, 8, 8), 3, 1)
        with pytest.raises(RuntimeError, match="inference-only"):
            LinearRowsFunction.apply(rows, torch.randn(32, 64, device="cuda", requires_grad=True), None)
    finally:
        set_conv_mode("bf16x3")
    for a, b in zip(outs["fp16"][0], outs["bf16x3"][0]):
        assert torch.isfinite(a).all()
        assert max_err(a, b) <= 2e-3 * max(1.0, b.abs().max().item())
    ga, gb = outs["fp16"][1], outs["bf16x3"][1]
    assert torch.isfinite(ga).all()
    cos = torch.nn.functional.cosine_similarity(ga.flatten(), gb.flatten(), dim=0).item()
    assert cos > 0.999, cos


def _neck_head_oracle(det, w):
    from oracle.ref_path import RefPath
    return RefPath({**{"neck." + k: v for k, v in det.neck_3d.state_dict().items()},
                    **{"head." + k: v for k, v in det.bbox_head.state_dict().items()}},
                   dict(head="scannet" if w["head"].startswith("ScanNet") else "sunrgbd", n_classes=w["n_classes"],
                        nms_pre=1000))


@pytest.mark.parametrize("name,img_hw", [("cfg3_arkit", None), ("cfg4_scannet200_large", None), ("cfg5_arkit_large", None),
                                         ("cfg2_scannet_100v", None)])
def test_full_view_count_scenes_against_the_oracle(name, img_hw):
    """BASELINE.json configs[2..4] at their FULL view counts (60 / 50 / 100 views; 48x48x16, 80x80x32 with 189
    classes, 96x96x32), and config 2 at the reference's own TEST-time view count (100 views, `n_images=100` of
    configs/SGCDet_ScanNet.py:151-164: the shape its README numbers are produced on, and the only C = 256 shape that takes the
    projected-query inter-view attention by default) -- GPU vs the OpenMP build of the oracle: selected voxel sets identical up to near ties at the
    cut, voxel features and occupancy within 1e-3, and neck + head (HIP) on the oracle's volume within 1e-3.
    These are the C = 128 (Cm = 16) shapes of the LDS-tiled gather at full pair counts (0.8 M / 2.4 M pairs)."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    from oracle.ref_path import RefPath
    from oracle.compare import check_sparse_head, topk_cut
    w = workload(name)
    torch.manual_seed(23)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(29)
    with torch.no_grad():
        for n, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.02)
    rp = RefPath(det.voxel_head.state_dict(), dict(embed_dims=w["embed_dims"], n_voxels_list=w["n_voxels_list"],
                                                   voxel_size_list=w["voxel_size_list"], topk_list=w["topk_list"],
                                                   dbound=(0.2, 5.0), num_heads=8, num_points=4), omp=True)
    # top-k is discontinuous in the fp32 scores.  Whether the coarse cut of a scene is resolvable (gap above rounding
    # noise) is a property of the oracle's scene, decided here on the CPU from levels 0-1 only; the first resolvable
    # seed of a fixed list is used, else the last one (then only the tie-independent checks below apply)
    gaps = []
    for seed in (31, 32, 33, 34):
        feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=seed, img_hw=img_hw)
        gaps.append((seed, rp.coarse_topk_gap(feats, meta, depth_pyramid(dpt))))
        if gaps[-1][1] > 2e-6:
            break
    vol_c, valid_c, occ_c, aux, volumes_c = rp.adaptive_sparse_head(feats, meta, depth_pyramid(dpt), return_aux=True)
    det = det.cuda()
    # (1) level by level, every level fed with the ORACLE's previous volume and the ORACLE's selection: every kernel of
    #     the view transform at full size, independent of how a top-k tie is broken
    _check_levels_against_oracle(det.voxel_head, [f.cuda() for f in feats], meta, [t.cuda() for t in depth_pyramid(dpt)],
                                 aux, volumes_c)
    # (2) end to end.  top-k is discontinuous in the fp32 scores: with 9 216 of 36 864 coarse candidates (config 5) the
    #     gap at the coarse cut is routinely ~1e-7, i.e. rounding noise decides which voxel is refined and the two
    #     volumes then legitimately differ around it.  Whether the cut is resolvable is a property of the oracle's
    #     scene, decided here on the CPU; the voxel-by-voxel comparison runs when it is, the structural checks always.
    with torch.no_grad():
        r = det.forward_features([f.cuda() for f in feats], [meta], dpt.cuda())
    n_fin = w["n_voxels_list"][-1][0] * w["n_voxels_list"][-1][1] * w["n_voxels_list"][-1][2]
    assert int(r["valid"].sum()) == w["topk_list"][1]
    assert (r["occ"].cpu()[0, n_fin:] - occ_c[0, n_fin:]).abs().max() < 1e-4          # coarse occupancy: no top-k upstream
    gap = topk_cut(aux[1]["occ"], w["topk_list"][0])[1]
    if gap > 2e-6:
        res = check_sparse_head(r["volume"], r["valid"], r["occ"], vol_c, valid_c, occ_c, n_fin, w["topk_list"])
    else:
        # every seed's coarse cut is at rounding-noise level (config 5: 9 216 of 36 864 candidates): which voxel is refined
        # there is a coin toss on EITHER side.  The end-to-end comparison still runs -- with the oracle's coarse selection
        # forced into the product's first top-k, everything downstream (level 1 on that selection, the second upsample +
        # occupancy, the finest top-k, level 2, the scatter-adds) is the product's own and is compared voxel by voxel.
        r = _forward_with_injected_coarse_selection(det, feats, meta, dpt, aux[1]["idx"])
        assert int(r["valid"].sum()) == w["topk_list"][1]
        res = check_sparse_head(r["volume"], r["valid"], r["occ"], vol_c, valid_c, occ_c, n_fin, w["topk_list"], coarse_injected=True)
    assert res["tie_flips"] <= max(8, n_fin // 20000), res
    # (3) neck + head (HIP convolutions) on the oracle's volume
    _check_neck_head_on(det, vol_c, _neck_head_oracle(det, w))


def _forward_with_injected_coarse_selection(det, feats, meta, dpt, coarse_idx):
    """det.forward_features with the FIRST top-k of AdaptiveSparseHead replaced by ``coarse_idx`` (the oracle's ascending
    selection); every other launch is the product's."""
    from sgcdet_amd import ext
    ops = ext.ops()
    orig, calls = ops.topk_select, []

    def patched(occ, k, **kw):
        if not calls:
            calls.append(k)
            assert coarse_idx.numel() == k
            return coarse_idx.to(device=occ.device, dtype=torch.int64).contiguous(), None, None
        return orig(occ, k, **kw)

    ops.topk_select = patched
    graph = det.use_graph, det.scene_graph
    det.use_graph = det.scene_graph = False
    try:
        with torch.no_grad():
            r = det.forward_features([f.cuda() for f in feats], [meta], dpt.cuda())
    finally:
        del ops.topk_select                       # back to the class's method
        det.use_graph, det.scene_graph = graph
    assert len(calls) == 1
    return r


def _check_levels_against_oracle(head, feats, meta, dpts, aux, volumes_c):
    """AdaptiveSparseHead level by level on the GPU with the oracle's inputs per level: dense level 0; then for every
    finer level the fused trilinear x2 + occupancy kernel on the ORACLE's previous volume (occupancy within 1e-5) and
    the level's DenseHead on the ORACLE's selected voxels, `upsampled + refined` within 1e-3 of the oracle's volume."""
    from sgcdet_amd import ext
    ops = ext.ops()
    C = head.embed_dims
    with torch.no_grad():
        feat, dpt = head._level_inputs(0, feats, meta, dpts)
        rows = head.base_heads[0].seed_rows([feat], meta, None, mlvl_dpt_dists=[dpt])
        vc = volumes_c[0][0].permute(1, 2, 3, 0).reshape(-1, C)
        assert max_err(rows, vc) < 1e-3 * max(1.0, vc.abs().max().item())
        for i in range(1, len(head.base_heads)):
            prev = volumes_c[i - 1]
            grid = tuple(prev.shape[2:])
            prev_rows = prev[0].permute(1, 2, 3, 0).reshape(-1, C).contiguous().cuda()
            lin = head.occ_pred_heads[i - 1][0]
            up, occ, grid = ops.upsample2x_occ(prev_rows, grid, lin.weight.reshape(-1), lin.bias)
            assert max_err(occ, aux[i]["occ"].reshape(-1)) < 1e-5
            idx = aux[i]["idx"].cuda()
            feat, dpt = head._level_inputs(i, feats, meta, dpts)
            seed = head.base_heads[i].seed_rows([feat], meta, idx, mlvl_dpt_dists=[dpt])
            ops.scatter_add_rows(seed.contiguous(), idx, up)
            vc = volumes_c[i][0].permute(1, 2, 3, 0).reshape(-1, C)
            assert max_err(up, vc) < 1e-3 * max(1.0, vc.abs().max().item()), i


def test_full_size_config2_scene_against_the_oracle():
    """BASELINE.json configs[1] at its full size -- 40 views x 256 ch, 64x80 / 32x40 / 16x20 maps, 40x40x16 voxels,
    top-k [800, 6400] -- GPU vs the CPU oracle (its OpenMP build: same arithmetic, outputs parallelised): selected
    voxel sets bit-exact up to near ties at the cut, voxel features and occupancy within 1e-3."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    from oracle.ref_path import RefPath
    from oracle.compare import check_sparse_head
    w = workload("cfg2_scannet")
    torch.manual_seed(17)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(13)
    with torch.no_grad():
        for n, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.02)
    feats, dpt, meta = make_scene(40, w["embed_dims"], kind="scannet", seed=25, img_hw=(256, 320))
    rp = RefPath(det.voxel_head.state_dict(), dict(embed_dims=w["embed_dims"], n_voxels_list=w["n_voxels_list"],
                                                   voxel_size_list=w["voxel_size_list"], topk_list=w["topk_list"],
                                                   dbound=(0.2, 5.0), num_heads=8, num_points=4), omp=True)
    vol_c, valid_c, occ_c = rp.adaptive_sparse_head(feats, meta, depth_pyramid(dpt))
    det = det.cuda()
    with torch.no_grad():
        r = det.forward_features([f.cuda() for f in feats], [meta], dpt.cuda())
    res = check_sparse_head(r["volume"], r["valid"], r["occ"], vol_c, valid_c, occ_c, 40 * 40 * 16, w["topk_list"])
    assert res["tie_flips"] <= 8, res
    assert int(r["valid"].sum()) == 6400
    # neck + head at full size (490 + 5 GFLOP): HIP convolutions on the oracle's volume vs torch-CPU conv3d
    rp2 = _neck_head_oracle(det, w)
    _check_neck_head_on(det, vol_c, rp2)
    # ... and in the configuration bench.py TIMES (round-5 review, parity item 3: the oracle comparison of the as-benched combination
    # ran at 6 views only): the throughput launch geometry, where every 3x3x3 layer with >= 256 channels takes the Winograd-z form
    # (also the 20 x 20 x 8 layers the latency geometry leaves on the direct kernel) and the few-voxel layers use the coarser splits
    from sgcdet_amd.plugin import conv_plan
    from sgcdet_amd import ext
    ops = ext.ops()
    try:
        conv_plan.set_throughput_mode(True)
        log = []
        orig = ops.conv3d_winograd_z
        ops.conv3d_winograd_z = lambda *a, **k: (log.append(a[3]), orig(*a, **k))[1]
        _check_neck_head_on(det, vol_c, rp2)
        assert (20, 20, 8) in log and (40, 40, 16) in log and len(log) == 7, log       # seven layers of config 2 on the form
    finally:
        ops.conv3d_winograd_z = orig
        conv_plan.set_throughput_mode(False)


def test_training_path_gradients_match_oracle_backward():
    """autograd through the HIP forward/backward Function == oracle backward (a11b)."""
    from sgcdet_amd.functions import MultiScale3DDeformableAttnFunction_fp32 as Fn
    import oracle
    d, _ = load("op_autograd")
    leaves = [d[k].cuda().requires_grad_() for k in ("value", "dist", "loc", "attn")]
    out, score = Fn.apply(leaves[0], leaves[1], d["shapes3"].cuda(), d["lsi"].cuda(), leaves[2], leaves[3], 64)
    assert max_err(out, d["out"]) < 1e-5 and max_err(score, d["score"]) < 1e-6
    grads = torch.autograd.grad(out, leaves, d["grad_out"].cuda())
    for g, k in zip(grads, ("grad_value", "grad_dist", "grad_loc", "grad_attn")):
        assert max_err(g, d[k]) < 2e-4 * max(1.0, d[k].abs().max().item()), k


def test_module_training_path_runs_and_matches_inference_path():
    d, sd = load("voxel_head")
    head = _build_voxel_head(d)
    head.load_state_dict(sd)
    head = head.cuda().train()
    for m in head.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    meta = img_meta(d)
    feats = [d[f"feat{i}"].cuda().requires_grad_() for i in range(4)]
    dpts = [t.cuda() for t in depth_pyramid(d["dpt"])]
    volume, valid, occ = head(feats, meta, dpts)
    assert max_err(volume, d["volume"]) < 1e-3
    assert torch.equal(valid.cpu(), d["valid"])
    (volume.square().mean() + occ.mean()).backward()
    assert feats[0].grad is not None and torch.isfinite(feats[0].grad).all() and feats[0].grad.abs().sum() > 0
    n_with_grad = sum(p.grad is not None and p.grad.abs().sum() > 0 for p in head.parameters())
    assert n_with_grad > 20


def test_projection_composition_on_this_host():
    """single-mm == reference loop on the GPU box's host CPU as well (BLAS code paths are host specific)"""
    from test_host_logic import test_single_mm_projection_equals_reference_loop
    test_single_mm_projection_equals_reference_loop()


def test_two_scenes_in_flight_give_the_same_results():
    """ScenePipeline (alternating HIP streams, per-stream hipGraph buffers) == one scene at a time."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.pipeline import ScenePipeline
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload("cfg1_plumbing")
    torch.manual_seed(11)
    det = build_detector(model_config(w)).eval().cuda()
    gen = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for _, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen).to(p.device) * 0.05)
    scenes = []
    for s in range(5):
        feats, dpt, meta = make_scene(4, w["embed_dims"], kind=w["kind"], seed=40 + s, device="cuda")
        scenes.append((feats, [meta], dpt))
    serial = ScenePipeline(det, n_streams=1).run(scenes)
    for _ in range(3):                                  # repeat: races are timing dependent
        piped = ScenePipeline(det, n_streams=2).run(scenes)
        torch.cuda.synchronize()
        for a, b in zip(serial, piped):
            assert torch.equal(a["valid"], b["valid"]) and torch.equal(a["occ"], b["occ"])
            assert torch.equal(a["volume"], b["volume"])
            for k in ("centerness", "bbox_pred", "cls_score"):
                for x, y in zip(a[k], b[k]):
                    assert torch.equal(x, y)            # split-K layers reduce in a fixed order (workspace)


@pytest.mark.parametrize("use_graph", [True, False])
def test_overlapping_scenes_are_bit_exact_at_full_size(use_graph):
    """Config 2 (40 views, 40x40x16) stepped the way bench.py does -- consecutive scenes on alternating
    streams, nothing cloned, no sync in between, so the neck convolutions of one scene run beside the
    gathers of the next -- must reproduce the serial, eagerly launched voxel features bit for bit.
    (Before the packed-FP32 forms were switched off in the build this failed on ~9 of 10 scenes: the
    gfx950 hazard of DESIGN.md 4.7.)"""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload("cfg2_scannet")
    torch.manual_seed(0)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for _, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.02)
    det = det.cuda()
    scenes = []
    for s in range(3):
        feats, dpt, meta = make_scene(w["n_views"], w["embed_dims"], kind=w["kind"], seed=s, device="cuda")
        scenes.append((feats, dpt, [meta]))
    det.use_graph = False
    serial = []
    with torch.no_grad():
        for feats, dpt, metas in scenes:
            r = det.forward_features(feats, metas, dpt)
            serial.append((r["volume"].clone(), r["occ"].clone(), r["valid"].clone()))
    torch.cuda.synchronize()
    det.use_graph = use_graph
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    runs = []
    with torch.no_grad():
        for i in range(24):
            feats, dpt, metas = scenes[i % 3]
            with torch.cuda.stream(streams[i % 2]):
                r = det.forward_features(feats, metas, dpt)
                runs.append((i, r["volume"], r["occ"], r["valid"]))
    torch.cuda.synchronize()
    wrong = [i for i, v, o, m in runs
             if not (torch.equal(v, serial[i % 3][0]) and torch.equal(o, serial[i % 3][1]) and torch.equal(m, serial[i % 3][2]))]
    assert not wrong, f"scene runs {wrong} differ from the serial result"


@pytest.mark.parametrize("name,n_views", [("cfg1_plumbing", 4), ("cfg2_scannet", 40), ("cfg3_arkit", 6)])
def test_scene_graph_replay_is_bit_identical_to_eager_launches(name, n_views):
    """``SGCDet.scene_graph``: the whole scene as one hipGraph replay (pair / voxel counts left on the device,
    worst-case buffers) == the eager path with its per-level host read-back, bit for bit on the voxel features,
    masks and occupancy; replaying on new scene content through the same buffers follows the content."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload(name)
    torch.manual_seed(3)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for _, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.03)
    det = det.cuda()
    scenes = [make_scene(n_views, w["embed_dims"], kind=w["kind"], seed=70 + s, device="cuda") for s in range(2)]
    feats = [f.clone() for f in scenes[0][0]]
    dpt = scenes[0][1].clone()
    for s, (f_s, d_s, meta) in enumerate(scenes):
        for dst, src in zip(feats, f_s):                 # same buffers, new content (and new cameras)
            dst.copy_(src)
        dpt.copy_(d_s)
        with torch.no_grad():
            det.scene_graph, det.use_graph = False, False
            r0 = det.forward_features(feats, [meta], dpt)
            want = {k: r0[k].clone() for k in ("volume", "valid", "occ")}
            heads = [t.clone() for t in r0["centerness"] + r0["bbox_pred"] + r0["cls_score"]]
            det.scene_graph = True
            r1 = det.forward_features(feats, [meta], dpt)
            torch.cuda.synchronize()
        assert len(det._scene_graph_cache) == 1          # the second scene replays the graph captured for the first
        for k in want:
            assert torch.equal(r1[k], want[k]), (s, k)
        for a, b in zip(r1["centerness"] + r1["bbox_pred"] + r1["cls_score"], heads):
            assert torch.equal(a, b)                    # split-K layers reduce in a fixed order (workspace)


def test_scene_graph_with_fresh_input_buffers_falls_back_to_copies():
    """A producer that hands over new buffers every scene must not trigger a capture per scene: past
    ``scene_graph_capacity`` aliased graphs one input-copying graph serves every further address set."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload("cfg1_plumbing")
    torch.manual_seed(3)
    det = build_detector(model_config(w)).eval().cuda()
    det.scene_graph_capacity = 2
    keep = []                # the earlier scenes stay alive: a NEW address set every scene whatever the caching allocator would recycle
    for s in range(6):
        feats, dpt, meta = make_scene(4, w["embed_dims"], kind=w["kind"], seed=90 + s, device="cuda")   # new tensors
        keep.append((feats, dpt))
        with torch.no_grad():
            det.scene_graph, det.use_graph = False, False
            want = det.forward_features(feats, [meta], dpt)
            want = {k: want[k].clone() for k in ("volume", "valid", "occ")}
            det.scene_graph = True
            got = det.forward_features(feats, [meta], dpt)
            torch.cuda.synchronize()
        for k in want:
            assert torch.equal(got[k], want[k]), (s, k)
    kinds = [e["inputs"] is None for e in det._scene_graph_cache.values()]
    assert kinds.count(True) == 2 and kinds.count(False) == 1      # 2 aliased graphs + ONE copying graph for the rest


@pytest.mark.parametrize("scene_graph", [False, True])
def test_channels_last_inputs_skip_the_transpose_and_change_nothing(scene_graph):
    """FPN / depth maps handed over channels-last in memory (same logical [1,N,C,H,W] shape) are
    consumed in place -- no NCHW->NHWC pass, the cropped-away last row stays in the buffers -- with bit-identical
    results (config-2 shapes at the reference's 239x320, i.e. maps 60x80 cropped to 59x80)."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd import ext
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload("cfg2_scannet")
    torch.manual_seed(5)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for _, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.03)
    det = det.cuda()
    feats, dpt, meta = make_scene(8, w["embed_dims"], kind=w["kind"], seed=33, device="cuda")
    cl = lambda t: t[0].contiguous(memory_format=torch.channels_last).unsqueeze(0)     # [1,N,C,H,W], NHWC memory
    feats_cl, dpt_cl = [cl(f) for f in feats], cl(dpt)
    assert feats_cl[0].shape == feats[0].shape and feats_cl[0].stride() != feats[0].stride()
    det.scene_graph, det.use_graph = scene_graph, scene_graph
    ops = ext.ops()
    calls = []
    orig = ops.nchw_to_nhwc_crop
    ops.nchw_to_nhwc_crop = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            want = det.forward_features(feats, [meta], dpt)
            want = {k: want[k].clone() for k in ("volume", "valid", "occ")}
            n_transposes = len(calls)
            got = det.forward_features(feats_cl, [meta], dpt_cl)
            torch.cuda.synchronize()
    finally:
        ops.nchw_to_nhwc_crop = orig
    assert n_transposes > 0 and len(calls) == n_transposes        # the channels-last call launched no transpose
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_get_bboxes_end_to_end_with_gpu_nms_matches_oracle_decode_and_nms(oracle_ops):
    """SGCDet.simple_test_from_features (SGCDet.py:119-129): head tensors -> decode -> aligned_3d_nms on the GPU ==
    the oracle's decode followed by the oracle's NMS, fed the SAME head tensors (boxes 1e-4, kept set identical)."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    from oracle.ref_path import RefPath
    w = workload("cfg1_plumbing")
    torch.manual_seed(9)
    cfg = model_config(w)
    det = build_detector(cfg).eval()
    gen = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for _, p in list(det.voxel_head.named_parameters()) + list(det.bbox_head.named_parameters()):
            p.add_(torch.randn(p.shape, generator=gen) * 0.05)
    det = det.cuda()
    det.bbox_head.test_cfg = dict(nms_pre=200, iou_thr=0.25, score_thr=0.05)
    feats, dpt, meta = make_scene(4, w["embed_dims"], kind=w["kind"], seed=12, device="cuda")
    with torch.no_grad():
        r = det.forward_features(feats, [meta], dpt)
        (boxes, scores, labels), = det.bbox_head.get_bboxes(r["centerness"], r["bbox_pred"], r["cls_score"],
                                                            r["valid"].float(), [meta])
    rp = RefPath({}, dict(head="scannet", n_classes=w["n_classes"], nms_pre=200))
    cpu = lambda ts: [t.cpu() for t in ts]
    b_c, s_c = rp.decode(cpu(r["centerness"]), cpu(r["bbox_pred"]), cpu(r["cls_score"]), r["valid"].float().cpu(), meta,
                         det.bbox_head.voxel_size)
    sc, lb = s_c.max(dim=1)
    keep = sc > 0.05
    b_c, sc, lb = b_c[keep], sc[keep], lb[keep]
    ids = oracle_ops.aligned_nms3d(b_c.contiguous(), sc.contiguous(), lb, 0.25)
    want = torch.cat([(b_c[ids, :3] + b_c[ids, 3:6]) / 2, b_c[ids, 3:6] - b_c[ids, :3]], 1)
    assert boxes.shape[0] == want.shape[0] and 0 < want.shape[0] < int(keep.sum())     # NMS removed something
    assert max_err(boxes, want) < 1e-4 and max_err(scores, sc[ids]) < 1e-5
    assert torch.equal(labels.cpu(), lb[ids])


def test_arkit_head_get_bboxes_runs_rotated_multiclass_nms_on_the_gpu(oracle_ops):
    """SunRgbdImVoxelHeadV2.get_bboxes (imvoxel_head_v2.py:248-317, :565-584) end to end on the GPU with the ARKit
    test_cfg (score_thr 0, rotated BEV NMS at 0.15, max_num = nms_pre): the survivors equal the reference's
    multiclass glue run through the CPU oracle on the same decoded candidates."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.plugin.bbox_head import box3d_multiclass_nms_rotated
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload("cfg1_plumbing")
    w.update(kind="arkit", head="SunRgbdImVoxelHeadV2", n_classes=17, n_reg_outs=7)
    torch.manual_seed(19)
    det = build_detector(model_config(w, nms_pre=150)).eval()
    gen = torch.Generator().manual_seed(18)
    with torch.no_grad():
        for _, p in list(det.voxel_head.named_parameters()) + list(det.bbox_head.named_parameters()):
            p.add_(torch.randn(p.shape, generator=gen) * 0.05)
    det = det.cuda()
    assert det.bbox_head.test_cfg["use_rotate_nms"] and det.bbox_head.test_cfg["score_thr"] == 0.0
    feats, dpt, meta = make_scene(4, w["embed_dims"], kind="arkit", seed=13, device="cuda")
    with torch.no_grad():
        r = det.forward_features(feats, [meta], dpt)
        head = det.bbox_head
        (boxes, scores, labels), = head.get_bboxes(r["centerness"], r["bbox_pred"], r["cls_score"], r["valid"].float(), [meta])
        valids = [torch.nn.Upsample(size=x.shape[-3:], mode="trilinear")(r["valid"].float()).round().bool() for x in r["centerness"]]
        cand_b, cand_s = head.decode_candidates([x[0] for x in r["centerness"]], [x[0] for x in r["bbox_pred"]],
                                                [x[0] for x in r["cls_score"]], [v[0] for v in valids], meta)
    assert cand_b.shape == (350, 7) and cand_s.shape == (350, 17)              # 150 + 150 + all 50 coarse voxels
    wb, ws, wl = box3d_multiclass_nms_rotated(oracle_ops, cand_b.cpu().contiguous(), cand_s.cpu().contiguous(), 0.0, 150, 0.15)
    with torch.no_grad():
        res, = det.simple_test_from_features(feats, [meta], dpt, as_results=True)             # bbox3d2result format
    assert set(res) == {"boxes_3d", "scores_3d", "labels_3d"} and not res["scores_3d"].is_cuda
    assert torch.equal(res["scores_3d"], scores.cpu()) and torch.equal(res["labels_3d"], labels.cpu())
    assert boxes.shape == (150, 7)                                   # more than max_num survive over 17 classes: cut
    assert torch.equal(labels.cpu(), wl) and torch.equal(scores.cpu(), ws) and torch.equal(boxes.cpu(), wb)
    assert len(set(labels.tolist())) > 3


def test_forward_train_losses_match_cpu_recomputation_with_oracle_targets(oracle_ops):
    """SGCDet.forward_train from the FPN maps on (SGCDet.py:98-114) for the ScanNet head: the three detection
    losses and the occupancy loss equal a CPU recomputation from the same head tensors with the ORACLE's target
    assignment, and every trainable parameter of the path receives a finite gradient."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.plugin import losses as L
    from sgcdet_amd.scene import make_scene, model_config, workload
    from targets_contract import random_boxes
    w = workload("cfg1_plumbing")
    torch.manual_seed(23)
    cfg = model_config(w)
    cfg["occ_loss"] = True
    det = build_detector(cfg).cuda().train()
    for m in det.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    feats, dpt, meta = make_scene(3, w["embed_dims"], kind="scannet", seed=14, device="cuda")
    boxes, gl = random_boxes(9, 6, False)
    boxes[:, :3] *= 0.55                                        # inside the 6.4 x 6.4 x 3.2 m plumbing volume
    losses = det.forward_train_from_features(feats, [meta], dpt, [boxes.cuda()], [gl.cuda()])
    assert set(losses) == {"loss_centerness", "loss_bbox", "loss_cls", "loss_occ"}
    total = sum(losses.values())
    total.backward()
    for name, p in det.named_parameters():
        if p.requires_grad and ("voxel_head" in name or "neck_3d" in name or "bbox_head" in name):
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
    # CPU recomputation from the head tensors of the same (deterministic) forward
    with torch.no_grad():
        volume, valid, occ = det.build_volume_from_features(feats, [meta], dpt)
        ctr, reg, cls = det.bbox_head(det.neck_3d(volume))
    head = det.bbox_head
    pts = head.get_points([c.shape[-3:] for c in ctr], meta["lidar2img"]["origin"], "cpu")
    scales = torch.cat([torch.full((len(p),), i, dtype=torch.int32) for i, p in enumerate(pts)])
    P = torch.cat(pts).float().contiguous()
    ct_t, bx_t, lb, geo = oracle_ops.assign_targets(P, scales, boxes, gl, False, head.n_scales, head.limit, head.centerness_topk)
    vals = [torch.nn.Upsample(size=c.shape[-3:], mode="trilinear")(valid.float()).round().bool() for c in ctr]
    flat = lambda ts, k: torch.cat([t[0].permute(1, 2, 3, 0).reshape(-1, k) for t in ts]).cpu()
    c_, r_, s_, v_ = flat(ctr, 1)[:, 0], flat(reg, 6), flat(cls, head.n_classes), flat(vals, 1)[:, 0]
    pos = torch.nonzero((lb >= 0) & v_).reshape(-1)
    assert len(pos) > 10
    n_pos = float(len(pos))
    want_cls = L.sigmoid_focal_loss(s_[v_], lb[v_], avg_factor=n_pos)
    want_ctr = L.sigmoid_bce_loss(c_[pos], ct_t[pos], avg_factor=n_pos)
    want_box = L.axis_aligned_iou_loss(head._bbox_pred_to_bbox(P[pos], r_[pos]), bx_t[pos], weight=ct_t[pos], avg_factor=ct_t[pos].sum())
    want_occ = torch.nn.functional.binary_cross_entropy(occ.cpu(), geo[None, :occ.shape[1]].float()) * 0.5
    for k, want in (("loss_cls", want_cls), ("loss_centerness", want_ctr), ("loss_bbox", want_box), ("loss_occ", want_occ)):
        assert abs(float(losses[k].detach()) - float(want)) < 2e-4 * max(1.0, abs(float(want))), (k, float(losses[k].detach()), float(want))


def test_three_sgd_steps_with_the_batched_weight_planes():
    """functions.TrainWeightPlanes end to end: three optimiser steps through the whole training path (torch.optim.SGD updates the
    parameters in place between forwards, so every plane has to be repacked every step).  After every forward and after every
    backward EVERY registered plane pair equals a fresh pack of the parameter it was made from, bit for bit (the per-step views of
    the fused in_proj weight included) -- nothing is served stale, nothing is packed from the wrong parameter; steps 2 and 3 each
    cost ONE pack launch.  The losses agree with the pack-per-use form within the run-to-run noise of the path (the DFA3D backward
    and torch's index_add accumulate with float atomics, and the discrete top-k amplifies their last bits: two runs of ONE
    configuration differ by 2e-4 at the second loss, 2e-3 at the third)."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd import ext, functions
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload("cfg1_plumbing")
    feats, dpt, meta = make_scene(3, w["embed_dims"], kind="scannet", seed=14, device="cuda")
    planes, ops = functions.train_weight_planes(), ext.ops()

    def fresh_everywhere():
        for (ptr, shape, transpose, flip, pr, pc), e in planes.entries.items():
            hi, lo = ops.pack_conv_weight(e["w"], transpose=transpose, flip=flip, pad_rows=pr, pad_cols=pc)
            assert torch.equal(hi.view(torch.int16), e["hi"].view(torch.int16)) and torch.equal(lo.view(torch.int16), e["lo"].view(torch.int16)), (tuple(shape), transpose, flip)

    def run(batch, check):
        torch.manual_seed(23)
        det = build_detector(model_config(w)).cuda().train()
        for m in det.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        opt = torch.optim.SGD([p for p in det.parameters() if p.requires_grad], lr=1e-3)
        planes.clear()
        planes.enabled = batch
        n0, losses = planes.launches, []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            r = det.forward_features(feats, [meta], dpt)
            if check:
                fresh_everywhere()
            loss = sum((t ** 2).mean() for k in ("centerness", "bbox_pred", "cls_score") for t in r[k]) + r["occ"].mean()
            loss.backward()
            if check:
                fresh_everywhere()
            opt.step()
            losses.append(float(loss.detach()))
        return losses, planes.launches - n0, len(planes.entries)

    try:
        a = run(True, True)
        b = run(False, False)
    finally:
        planes.enabled = True
        planes.clear()
    assert a[1] == 2 and a[2] > 20 and b[1] == 0 and b[2] == 0       # steps 2 and 3: every registered plane repacked in ONE launch each
    assert abs(a[0][2] - a[0][0]) > 1e-2 * abs(a[0][0]), a[0]        # the updates matter
    for la, lb, tol in zip(a[0], b[0], (1e-6, 2e-3, 2e-2)):
        assert abs(la - lb) <= tol * abs(la), (a[0], b[0])


def test_indoor_eval_with_rotated_boxes_uses_the_gpu_iou():
    from eval_contract import check_case
    check_case(1)


def test_arkit_forward_train_uses_rotated_targets_and_rotated_iou_loss(oracle_ops):
    """The ARKit head's training step (rotated target assignment + RotatedIoU3DLoss): finite losses and gradients,
    loss_bbox equal to a CPU recomputation from the same head tensors with the oracle's targets."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.plugin import losses as L
    from sgcdet_amd.scene import make_scene, model_config, workload
    from targets_contract import random_boxes
    w = workload("cfg1_plumbing")
    w.update(kind="arkit", head="SunRgbdImVoxelHeadV2", n_classes=17, n_reg_outs=7)
    torch.manual_seed(29)
    det = build_detector(model_config(w)).cuda().train()
    for m in det.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    feats, dpt, meta = make_scene(3, w["embed_dims"], kind="arkit", seed=15, device="cuda")
    boxes, gl = random_boxes(8, 9, True)
    boxes[:, :3] *= 0.5
    gl = gl % 17
    losses = det.forward_train_from_features(feats, [meta], dpt, [boxes.cuda()], [gl.cuda()])
    assert set(losses) == {"loss_centerness", "loss_bbox", "loss_cls"}
    sum(losses.values()).backward()
    assert all(torch.isfinite(v) for v in losses.values())
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in det.parameters())
    assert det.bbox_head.reg_conv.weight.grad.abs().sum() > 0
    with torch.no_grad():
        volume, valid, occ = det.build_volume_from_features(feats, [meta], dpt)
        ctr, reg, cls = det.bbox_head(det.neck_3d(volume))
    head = det.bbox_head
    pts = head.get_points([c.shape[-3:] for c in ctr], meta["lidar2img"]["origin"], "cpu")
    scales = torch.cat([torch.full((len(p),), i, dtype=torch.int32) for i, p in enumerate(pts)])
    P = torch.cat(pts).float().contiguous()
    ct_t, bx_t, lb, geo = oracle_ops.assign_targets(P, scales, boxes, gl, True, head.n_scales, head.limit, head.centerness_topk)
    vals = [torch.nn.Upsample(size=c.shape[-3:], mode="trilinear")(valid.float()).round().bool() for c in ctr]
    flat = lambda ts, k: torch.cat([t[0].permute(1, 2, 3, 0).reshape(-1, k) for t in ts]).cpu()
    r_, v_ = flat(reg, 7), flat(vals, 1)[:, 0]
    pos = torch.nonzero((lb >= 0) & v_).reshape(-1)
    assert len(pos) > 10
    want = L.rotated_iou_3d_loss(head._bbox_pred_to_bbox(P[pos], r_[pos]), bx_t[pos], weight=ct_t[pos], avg_factor=ct_t[pos].sum())
    assert abs(float(losses["loss_bbox"].detach()) - float(want)) < 5e-4, (float(losses["loss_bbox"].detach()), float(want))


def test_scene_graph_follows_weight_updates():
    """A parameter changed in place (optimizer step, checkpoint load) must not be served by a stale graph or a stale
    prepared plan: the graph cache is keyed by (address, version) of every parameter / buffer of the path."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload("cfg1_plumbing")
    torch.manual_seed(3)
    det = build_detector(model_config(w)).eval().cuda()
    feats, dpt, meta = make_scene(4, w["embed_dims"], kind=w["kind"], seed=5, device="cuda")
    det.scene_graph = True
    with torch.no_grad():
        a = {k: v.clone() for k, v in det.forward_features(feats, [meta], dpt).items() if torch.is_tensor(v)}
        for p in list(det.voxel_head.parameters())[:6] + list(det.neck_3d.parameters())[:4]:
            p.mul_(1.05).add_(0.01)
        b = det.forward_features(feats, [meta], dpt)
        det.scene_graph, det.use_graph = False, False
        want = det.forward_features(feats, [meta], dpt)
        torch.cuda.synchronize()
    assert not torch.equal(a["volume"], b["volume"])
    assert torch.equal(b["volume"], want["volume"]) and torch.equal(b["occ"], want["occ"])
    for x, y in zip(b["centerness"] + b["cls_score"], want["centerness"] + want["cls_score"]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("occupancy", ["predicted", "clustered"])
def test_masked_decoder_tail_gives_the_dense_detections(occupancy):
    """Row N1 (north star: "sparse 3D convolution over the occupancy-masked voxels"): with ``masked_tail`` the finest
    decoder convolutions and the head convolutions only have to be right where the head's valid pyramid (or its 3x3x3
    dilations) is 1.  On a config-2 scene: head tensors bit-identical to the dense path wherever valid, everything
    finite, and the decoded + NMS'ed detections identical.  ``clustered``: the surface-clustered occupancy override
    (scene.clustered_occupancy) -- the workload on which whole bricks are dead, so the skip paths of the masked kernel run."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import clustered_occupancy, make_scene, model_config, workload
    w = workload("cfg2_scannet")
    torch.manual_seed(3)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in list(det.voxel_head.named_parameters()) + list(det.bbox_head.named_parameters()):
            p.add_(torch.randn(p.shape, generator=gen) * 0.05)
    det = det.cuda()
    det.use_graph = False
    if occupancy == "clustered":
        det.voxel_head.occupancy_override = clustered_occupancy(w["n_voxels_list"], seed=0, device="cuda")
    # the masked kernel is the DIRECT 3x3x3 kernel with an output mask: its bit-identity claim is against the dense direct kernel, so the
    # dense run of this test does not take the Winograd-z form (round 5) for the two wide layers of the tail
    from sgcdet_amd.plugin import conv_plan
    request = conv_plan.WINOGRAD_Z
    conv_plan.set_winograd_z(False)
    feats, dpt, meta = make_scene(12, w["embed_dims"], kind="scannet", seed=2, device="cuda")
    try:
        with torch.no_grad():
            det.masked_tail = False
            dense = det.forward_features(feats, [meta], dpt)
            dense = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in dense.items()}
            dets_d = det.bbox_head.get_bboxes(dense["centerness"], dense["bbox_pred"], dense["cls_score"], dense["valid"].float(), [meta])
            det.masked_tail = True
            sparse = det.forward_features(feats, [meta], dpt)
            dets_s = det.bbox_head.get_bboxes(sparse["centerness"], sparse["bbox_pred"], sparse["cls_score"], sparse["valid"].float(), [meta])
            det.masked_tail = False
    finally:
        conv_plan.set_winograd_z(request)
    assert torch.equal(dense["valid"], sparse["valid"]) and torch.equal(dense["volume"], sparse["volume"])
    valid = dense["valid"].float()
    assert int(valid.sum()) == w["topk_list"][-1]
    if occupancy == "clustered":                         # the override really clusters: a third of the 8x8x4 bricks hold no valid voxel
        live = torch.nn.functional.max_pool3d(valid, (8, 8, 4), (8, 8, 4)).mean().item()
        assert live < 0.75, live
        # ... and it is the selection of the override's scores, ties and all (lowest flat index first, as sgc_topk_select)
        sc = det.voxel_head.occupancy_override[-1]
        order = torch.sort(sc, descending=True, stable=True).indices[:w["topk_list"][-1]]
        want = torch.zeros_like(sc, dtype=torch.int64)
        want[order] = 1
        assert torch.equal(want.view_as(dense["valid"]), dense["valid"])
    n_checked = 0
    for key in ("centerness", "bbox_pred", "cls_score"):
        for a, b in zip(dense[key], sparse[key]):
            assert torch.isfinite(b).all()
            v = torch.nn.Upsample(size=a.shape[-3:], mode="trilinear")(valid).round().bool().expand_as(a)
            assert torch.equal(a[v], b[v]), key
            n_checked += int(v.sum())
    assert n_checked > 10000
    (bd, sd, ld), (bs, ss, ls) = dets_d[0], dets_s[0]
    assert bd.tensor.shape[0] > 0 if hasattr(bd, "tensor") else bd.shape[0] > 0
    tb = (lambda t: t.tensor if hasattr(t, "tensor") else t)
    assert torch.equal(tb(bd), tb(bs)) and torch.equal(sd, ss) and torch.equal(ld, ls)


def test_pair_list_training_path_equals_the_padded_reference_layout():
    """f-3: the differentiable path on the visible (camera, voxel) pairs (item-list operator + backward, no padded
    [N, max_len] rebatch) gives the same volume and the same gradients -- parameters, feature maps, depth maps -- as the
    reference's padded layout (`_forward_reference_layout`, pinned to the reference golden by
    test_module_training_path_runs_and_matches_inference_path)."""
    from sgcdet_amd.plugin.voxformer import DeformCrossAttention_DFA3D
    d, sd = load("voxel_head")
    results = {}
    for mode in (True, False):
        head = _build_voxel_head(d)
        head.load_state_dict(sd)
        head = head.cuda().train()
        for m in head.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, DeformCrossAttention_DFA3D):
                m.train_pair_list = mode
        feats = [d[f"feat{i}"].cuda().requires_grad_() for i in range(4)]
        dpt = d["dpt"].cuda().requires_grad_()
        dpts = depth_pyramid(dpt)
        volume, valid, occ = head(feats, img_meta(d), dpts)
        g = torch.Generator().manual_seed(1)
        w = torch.randn(volume.shape, generator=g).cuda()
        ((volume * w).sum() + occ.square().sum()).backward()
        results[mode] = (volume.detach(), valid, [f.grad for f in feats[:3]], dpt.grad,
                         {n: p.grad for n, p in head.named_parameters() if p.grad is not None})
    v1, m1, fg1, dg1, pg1 = results[True]
    v0, m0, fg0, dg0, pg0 = results[False]
    assert torch.equal(m1, m0) and max_err(v1, v0) < 1e-5 * max(1.0, v0.abs().max().item())
    assert max_err(v1, d["volume"]) < 1e-3
    for a, b in zip(fg1, fg0):
        assert max_err(a, b) < 2e-4 * max(1.0, b.abs().max().item())
    assert max_err(dg1, dg0) < 2e-4 * max(1.0, dg0.abs().max().item())
    assert set(pg1) == set(pg0) and len(pg1) > 20
    for k in pg0:
        assert max_err(pg1[k], pg0[k]) < 2e-4 * max(1.0, pg0[k].abs().max().item()), k


def test_depth_net_inference_uses_the_fused_plane_sweep_and_matches_the_reference():
    """DepthNet_Fusion (row f-2) on the GPU in inference: the cost volume comes from the fused HIP plane-sweep kernel (no
    warped [N,C,D,H,W] tensor); the depth distribution equals the reference class's output (tests/golden/depth_net.npz)
    and, through the detector shell, feeds the view transform."""
    import numpy as np
    import os
    import sgcdet_amd.plugin as P
    from golden_util import fill_by_name
    d, _ = load("depth_net")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth_net.npz"))
    stride, dbound = int(z["stride"]), [float(v) for v in z["dbound"]]
    net = P.DepthNet_Fusion(neighbor_img_num=2, downsample_factor=stride, dbound=dbound, mono_channels=d["xs"].shape[2],
                            loss_weight=0.5, max_tol=0, init_weight="none").eval()
    fill_by_name(net, base_seed=7, scale=0.15)
    net = net.cuda()
    meta = img_meta(d)
    from sgcdet_amd import ext
    ops = ext.ops()
    ops.event_log, ops.event_names = [], {"sgc_plane_sweep_corr"}
    try:
        with torch.no_grad():
            pred = net(d["xs"].cuda(), d["imgs"].cuda(), [meta], stride)
        assert len(ops.event_log) == 1                            # the fused kernel ran (once for the scene)
    finally:
        ops.event_log = ops.event_names = None
    assert max_err(pred, d["pred"]) < 5e-5
    assert max_err(pred.sum(2), torch.ones_like(pred.sum(2))) < 1e-5
    loss = net.loss(d["depth_maps"].cuda(), pred)["loss_dpt"]
    assert abs(float(loss) - float(d["loss"])) < 1e-4


def test_neck_and_head_autograd_on_hip_kernels_matches_library_convolutions():
    """Training mode (row f-3): FastIndoorImVoxelNeck + ScanNetImVoxelHeadV2 with every convolution pass (forward, input
    gradient, weight gradient) on the HIP kernels against the same modules on torch's library convolutions.

    * outputs, loss and running statistics: 1e-4 of the tensor scale;
    * every convolution's input / weight gradient IN ISOLATION (the tensors captured from the library run fed to the HIP
      Functions) against float64 autograd of that single op: 5e-5 -- the 3-way bf16 split is fp32-faithful per op;
    * end-to-end parameter gradients: direction (cosine > 0.9999) and 5e-2 of the gradient's scale -- BatchNorm's
      backward subtracts the mean of a nearly mean-free field, which amplifies the per-op rounding of BOTH paths (the
      library path moves by 1e-6 .. 1e-5 against float64 on the same sums, the split by 1e-4 .. 1.5e-2)."""
    from sgcdet_amd.plugin import conv_plan
    from sgcdet_amd.plugin.neck3d import FastIndoorImVoxelNeck
    from sgcdet_amd.plugin.bbox_head import ScanNetImVoxelHeadV2
    from sgcdet_amd.functions import ChannelsLastConv3dFunction, ChannelsLastConvTranspose3dFunction
    import copy
    import torch.nn.functional as F
    torch.manual_seed(3)
    neck = FastIndoorImVoxelNeck(in_channels=64, n_blocks=[1, 1, 1], out_channels=32).cuda().train()
    head = ScanNetImVoxelHeadV2(n_classes=5, n_channels=32, n_reg_outs=6, n_scales=3, limit=27, centerness_topk=18).cuda().train()
    neck_b, head_b = copy.deepcopy(neck), copy.deepcopy(head)
    x = torch.randn(1, 64, 16, 12, 8, device="cuda")
    xa = x.to(memory_format=torch.channels_last_3d).clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    captured = {}
    for n, m in neck_b.named_modules():
        if isinstance(m, (torch.nn.Conv3d, torch.nn.ConvTranspose3d)):
            m.register_forward_hook(lambda mod, inp, out, n=n: captured.__setitem__(n + ".x", inp[0].detach().clone()))
            m.register_full_backward_hook(lambda mod, gin, gout, n=n: captured.__setitem__(n + ".dy", gout[0].detach().clone()))

    def run(nk, hd, inp, mode):
        conv_plan.set_train_conv(mode)
        feats = nk(inp)
        ctr, reg, cls = hd(feats)
        loss = sum((t.float() ** 2).mean() for ts in (ctr, reg, cls) for t in ts) + sum((f ** 2).mean() for f in feats)
        loss.backward()
        return feats, (ctr, reg, cls), loss.detach()

    try:
        fa, ha, la = run(neck, head, xa, "hip")
        fb, hb, lb = run(neck_b, head_b, xb, "library")
    finally:
        conv_plan.set_train_conv("hip")
    for a, b in list(zip(fa, fb)) + [(a, b) for ta, tb in zip(ha, hb) for a, b in zip(ta, tb)]:
        assert a.shape == b.shape and max_err(a, b) < 1e-4 * max(1.0, float(b.detach().abs().max()))
    assert abs(float(la) - float(lb)) < 1e-5 * abs(float(lb))
    for (n, ba), (_, bb) in zip(neck.named_buffers(), neck_b.named_buffers()):
        assert max_err(ba.float(), bb.float()) < 1e-4 * max(1.0, float(bb.float().abs().max())), n

    def rows(t):
        return t[0].permute(1, 2, 3, 0).reshape(-1, t.shape[1]).contiguous()

    for n, m in neck_b.named_modules():                      # every convolution in isolation, on the real training tensors
        if not isinstance(m, (torch.nn.Conv3d, torch.nn.ConvTranspose3d)):
            continue
        xin, dy = captured[n + ".x"], captured[n + ".dy"]
        grid = tuple(xin.shape[2:])
        xr, w = rows(xin).requires_grad_(True), m.weight.detach().clone().requires_grad_(True)
        xd, wd = xin.double().requires_grad_(True), m.weight.detach().double().requires_grad_(True)
        if isinstance(m, torch.nn.ConvTranspose3d):
            y = ChannelsLastConvTranspose3dFunction.apply(xr, w, grid)
            yd = F.conv_transpose3d(xd, wd, None, 2)
        else:
            y = ChannelsLastConv3dFunction.apply(xr, w, grid, m.kernel_size[0], m.stride[0])
            yd = F.conv3d(xd, wd, None, m.stride, m.padding)
        y.backward(rows(dy))
        yd.backward(dy.double())
        assert max_err(w.grad, wd.grad) < 5e-5 * float(wd.grad.abs().max()), n
        assert max_err(xr.grad, rows(xd.grad)) < 5e-5 * float(xd.grad.abs().max()), n

    assert max_err(xa.grad, xb.grad) < 1e-2 * float(xb.grad.abs().max())
    for (n, pa), (_, pb) in zip(list(neck.named_parameters()) + list(head.named_parameters()),
                                list(neck_b.named_parameters()) + list(head_b.named_parameters())):
        if pb.grad is None:
            assert pa.grad is None, n
            continue
        ga, gb = pa.grad.double().flatten(), pb.grad.double().flatten()
        assert float((ga - gb).abs().max()) < 5e-2 * max(1e-12, float(gb.abs().max())), n
        if gb.numel() > 1:
            assert float(torch.dot(ga, gb) / (ga.norm() * gb.norm())) > 0.9999, n


def test_fpn_on_hip_emits_channels_last_maps_the_path_reads_in_place():
    """Row f-1, producer side: plugin.fpn.FPN in eval mode on the GPU == its torch formulation (1e-4 of the scale), its
    outputs are channels-last in memory, and feeding them to the view transformation launches NO NCHW -> NHWC pass of the
    feature maps while giving the results of the NCHW hand-over (same values, so bit-identical volume / valid / occ)."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd import ext
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import make_scene, model_config, workload
    w = workload("cfg2_scannet")
    torch.manual_seed(5)
    cfg = model_config(w)
    cfg["neck"] = dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=w["embed_dims"], num_outs=4)
    det = build_detector(cfg).eval().cuda()
    det.neck.init_weights()
    N = 6
    gen = torch.Generator().manual_seed(9)
    backbone = [torch.randn(N, c, h, wd, generator=gen).cuda() * 0.5
                for c, (h, wd) in zip([256, 512, 1024, 2048], [(64, 80), (32, 40), (16, 20), (8, 10)])]   # 239x320 padded to 256x320
    backbone[1] = backbone[1].contiguous(memory_format=torch.channels_last)      # one level already channels-last: read in place
    with torch.no_grad():
        maps = det.image_features(backbone)
        want = [f.unsqueeze(0) for f in det.neck._forward_torch(backbone)]
    for a, b in zip(maps, want):
        assert a.shape == b.shape and max_err(a, b) < 1e-4 * float(b.abs().max())
        assert a[0].is_contiguous(memory_format=torch.channels_last)
    _, dpt, meta = make_scene(N, w["embed_dims"], kind=w["kind"], seed=33, device="cuda", img_hw=(239, 320))
    ops = ext.ops()
    calls = []
    orig = ops.nchw_to_nhwc_crop
    ops.nchw_to_nhwc_crop = lambda src, *a, **k: (calls.append(src.shape[1]), orig(src, *a, **k))[1]
    try:
        with torch.no_grad():
            got = det.forward_features(maps[:3], [meta], dpt)
            got = {k: got[k].clone() for k in ("volume", "valid", "occ")}
            n_cl = sum(1 for c in calls if c == w["embed_dims"])
            ref = det.forward_features([m.contiguous() for m in maps[:3]], [meta], dpt)
            torch.cuda.synchronize()
    finally:
        ops.nchw_to_nhwc_crop = orig
    assert n_cl == 0 and sum(1 for c in calls if c == w["embed_dims"]) > 0      # only the NCHW hand-over transposed feature maps
    for k in got:
        assert torch.equal(got[k], ref[k]), k


def _dense_2d_cross_attention(att, feat, ref_uv, mask, H, W):
    """The published formulation of the 2-D classes on EVERY (camera, voxel), masked afterwards: bilinear sample
    (``F.grid_sample``, zeros padding, align_corners=False), 2-D deformable attention around it plus the sample
    (deformable_cross_attention.py:215-341, :632-647), mean / attention over the visible views (:652-676).  Torch ops only."""
    import torch.nn.functional as F
    da = att.deformable_attention
    N, S, C = feat.shape
    Nq = ref_uv.shape[1]
    M, P = da.num_heads, da.num_points
    fmap = feat.view(N, H, W, C).permute(0, 3, 1, 2)
    geo = F.grid_sample(fmap, (ref_uv * 2 - 1).view(N, Nq, 1, 2), mode="bilinear", padding_mode="zeros",
                        align_corners=False)[..., 0].permute(0, 2, 1)                       # [N,Nq,C]
    v = da.value_proj(feat).view(N, H, W, M, C // M).permute(0, 3, 4, 1, 2).reshape(N * M, C // M, H, W)
    off = da.sampling_offsets(geo).view(N, Nq, M, P, 2) / torch.tensor([W, H], dtype=feat.dtype, device=feat.device)
    a = da.attention_weights(geo).view(N, Nq, M, P).softmax(-1)
    loc = ref_uv[:, :, None, None, :] + off
    samp = F.grid_sample(v, (loc * 2 - 1).permute(0, 2, 1, 3, 4).reshape(N * M, Nq, P, 2), mode="bilinear",
                         padding_mode="zeros", align_corners=False).view(N, M, C // M, Nq, P)
    per = (samp * a.permute(0, 2, 1, 3)[:, :, None]).sum(-1).permute(0, 3, 1, 2).reshape(N, Nq, C) + geo
    count = mask.sum(0)
    valid = count > 0
    mean = (per * mask[..., None]).sum(0)[valid] / count[valid][:, None]
    pooled = att.output_proj(mean)
    pooled, _ = att.attention_pooling(pooled[None], per[:, valid], per[:, valid], ~mask[:, valid].t())
    out = torch.zeros(Nq, C, dtype=feat.dtype, device=feat.device)
    return out.index_put((valid.nonzero()[:, 0],), pooled[0])[None]


def test_2d_registry_classes_run_the_hip_path_and_equal_the_published_formulation():
    """SURVEY 8b registry surface: ``PerceptionTransformer`` / ``VoxFormerEncoder`` / ``DeformCrossAttention`` /
    ``MSDeformableAttention3D`` (the 2-D names; no SGCDet config selects them) build from a config and run on the same HIP
    kernels as the DFA3D classes (one unit depth bin).  Checked against torch's grid_sample formulation: the inference
    pair-list path, both differentiable paths, and their gradients."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd import ext
    from sgcdet_amd.mmcv_lite import TRANSFORMER
    from sgcdet_amd.plugin import voxformer as vf
    from sgcdet_amd.scene import make_img_meta
    C, N, H, W = 64, 5, 15, 20
    cfg = dict(type="PerceptionTransformer", embed_dims=C, encoder=dict(
        type="VoxFormerEncoder", num_layers=1, return_intermediate=False, dbound=[0.2, 5.0],
        transformerlayers=dict(
            type="VoxFormerLayer",
            attn_cfgs=[dict(type="DeformCrossAttention", embed_dims=C, inter_view_aggregation="attn", dropout=0,
                            deformable_attention=dict(type="MSDeformableAttention3D", embed_dims=C, num_heads=4,
                                                      num_levels=1, num_points=4))],
            ffn_cfgs=dict(type="FFN", embed_dims=C, feedforward_channels=C * 2, num_fcs=2, ffn_drop=0.0,
                          act_cfg=dict(type="ReLU", inplace=True)),
            operation_order=("cross_attn", "norm", "ffn", "norm"))))
    torch.manual_seed(0)
    xf = TRANSFORMER.build(cfg).cuda()
    layer = xf.encoder.layers[0]
    att = layer.attentions[0]
    assert type(xf) is vf.PerceptionTransformer and type(xf.encoder) is vf.VoxFormerEncoder
    assert type(att) is vf.DeformCrossAttention and type(att.deformable_attention) is vf.MSDeformableAttention3D
    assert not any("depth" in k for k in xf.state_dict())            # the 2-D classes own no depth-offset Linear (:165-170)
    with torch.no_grad():                                            # offsets / logits / FFN that depend on the data
        for p in xf.parameters():
            if p.dim() > 1:
                p.add_(0.05 * torch.randn_like(p))
    meta = make_img_meta(N, "scannet", seed=2)
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(1, N, C, H, W, generator=g).cuda()
    gx, gy, gz = 12, 12, 6
    ax = [torch.arange(n, dtype=torch.float32) for n in (gx, gy, gz)]
    pts = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    ref_3d = ((pts + 0.5) / torch.tensor([gx, gy, gz]) - 0.5) * torch.tensor([6.4, 6.4, 2.56])
    n_vox = ref_3d.shape[0]
    coords = torch.cat([pts.long(), torch.arange(n_vox)[:, None]], 1).cuda()
    idx = torch.arange(0, n_vox, 2).cuda()                           # every other voxel is a query
    ref_3d = ref_3d.cuda()
    # what the encoder hands the attention: (u, v) only, and the shared visibility mask (encoder.py:45-99)
    ref_uv, mask = xf.encoder.point_sampling(ref_3d[idx][None, None], meta)
    assert ref_uv.shape == (N, 1, idx.numel(), 1, 2) and 0 < int(mask.sum()) < mask.numel()
    ref_uv, mask = ref_uv.reshape(N, -1, 2), mask.reshape(N, -1)
    rows = feat[0].flatten(2).permute(0, 2, 1).contiguous()

    def tail(x):
        return layer.norms[1](layer.ffns[0](layer.norms[0](x)))
    with torch.no_grad():
        want = tail(_dense_2d_cross_attention(att, rows, ref_uv, mask, H, W))
        n0 = ext.ops().n_calls
        got = xf.get_vox_features([feat], None, ref_3d, coords, idx, img_meta=meta)
        assert ext.ops().n_calls > n0                                # the kernels of the library ran, not a torch fallback
    scale = max(1.0, want.abs().max().item())
    assert got.shape == want.shape and max_err(got, want) < 2e-4 * scale, max_err(got, want)
    # the differentiable paths (pair list / the reference's padded layout): the whole level forward; gradients through the
    # attention alone -- behind it sits a ReLU, and one pre-activation within rounding noise of zero (|h| = 2e-6 in this very
    # scene) flips its mask between two correct Linear kernels and moves single gradient entries by 0.5 %
    q0 = torch.zeros(1, idx.numel(), C, device="cuda")
    ss, lsi = torch.tensor([[H, W]], device="cuda"), torch.zeros(1, dtype=torch.long, device="cuda")
    names = ["feat"] + [n for n, _ in att.named_parameters()]
    params = [p for p in att.parameters()]
    featg = feat.clone().requires_grad_()
    want_att = _dense_2d_cross_attention(att, featg[0].flatten(2).permute(0, 2, 1), ref_uv, mask, H, W)
    w = torch.randn(want_att.shape, generator=g).cuda()
    (want_att * w).sum().backward()
    want_grads = [featg.grad.clone()] + [None if p.grad is None else p.grad.clone() for p in params]
    assert sum(b is not None for b in want_grads) >= 13
    for mode in (True, False):
        att.train_pair_list = mode
        out = xf.get_vox_features([feat.clone().requires_grad_()], None, ref_3d, coords, idx, img_meta=meta)
        assert max_err(out, want) < 2e-4 * scale, (mode, max_err(out, want))
        xf.zero_grad(set_to_none=True)
        featg = feat.clone().requires_grad_()
        got_att = att(q0, None, featg[0].flatten(2).permute(0, 2, 1).unsqueeze(2), reference_points_cam=ref_uv.view(N, 1, -1, 1, 2),
                      bev_mask=mask.view(N, 1, -1, 1), spatial_shapes=ss, level_start_index=lsi)
        assert max_err(got_att, want_att) < 1e-4 * max(1.0, want_att.abs().max().item())
        (got_att * w).sum().backward()
        got_grads = [featg.grad] + [p.grad for p in params]
        bad = [(n, tuple(b.shape), round(max_err(a, b), 6), round(b.abs().max().item(), 4)) for n, a, b in zip(names, got_grads, want_grads)
               if (a is None) != (b is None) or (b is not None and max_err(a, b) >= 2e-4 * max(1.0, b.abs().max().item()))]
        assert not bad, (mode, bad)
