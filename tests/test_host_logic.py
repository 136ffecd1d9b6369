"""Host-side logic that must agree with the reference's own torch calls bit for bit."""
import numpy as np
import pytest
import torch

from sgcdet_amd.plugin.voxformer import compute_projection, compute_projection_loop
from sgcdet_amd.scene import make_img_meta, workload, model_config


def test_single_mm_projection_equals_reference_loop():
    rng = np.random.RandomState(0)
    for trial in range(50):
        n = int(rng.randint(1, 101))
        meta = make_img_meta(n, "scannet" if trial % 2 else "arkit", seed=trial)
        meta["lidar2img"]["extrinsic"] = [(e + rng.randn(4, 4).astype(np.float32) * 0.3) for e in meta["lidar2img"]["extrinsic"]]
        assert torch.equal(compute_projection(meta), compute_projection_loop(meta))


def test_workloads_match_baseline_configs():
    w = workload("cfg2_scannet")
    assert w["n_voxels_list"] == [(10, 10, 4), (20, 20, 8), (40, 40, 16)] and w["topk_list"] == [800, 6400]
    assert workload("cfg1_plumbing")["topk_list"] == [100, 800]
    assert workload("cfg5_arkit_large")["topk_list"] == [9216, 73728]
    cfg = model_config(w)
    assert cfg["voxel_head"]["base_head_configs"][2]["n_voxels"] == (40, 40, 16)


def test_head_losses_follow_their_definitions():
    """plugin/losses.py: the vendored axis-aligned IoU (golden of the reference's own function), the focal loss
    against mmdet's pure-torch formulation (one-hot targets, background = all-zero row), BCE against its definition"""
    import os
    import numpy as np
    import torch.nn.functional as F
    from sgcdet_amd.plugin import losses
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "head_targets.npz"))
    a, b = torch.from_numpy(d["iou_a"]), torch.from_numpy(d["iou_b"])
    iou = losses.axis_aligned_iou(a, b)
    assert torch.allclose(iou, torch.from_numpy(d["iou_aligned"]), rtol=0, atol=1e-7) and (iou > 0).sum() > 20 and (iou == 0).sum() > 0
    w = torch.rand(64)
    got = losses.axis_aligned_iou_loss(a, b, weight=w, avg_factor=w.sum())
    assert torch.allclose(got, ((1 - iou) * w).sum() / w.sum())
    assert losses.axis_aligned_iou_loss(a, b, weight=torch.zeros(64), avg_factor=1.0) == 0
    g = torch.Generator().manual_seed(0)
    pred = torch.randn(200, 18, generator=g) * 3
    tgt = torch.randint(-1, 18, (200,), generator=g)
    onehot = F.one_hot(tgt.clamp(min=0), 18).float() * (tgt >= 0)[:, None]
    p = pred.sigmoid()
    pt = (1 - p) * onehot + p * (1 - onehot)
    ref = F.binary_cross_entropy_with_logits(pred, onehot, reduction="none") * (0.25 * onehot + 0.75 * (1 - onehot)) * pt.pow(2.0)
    got = losses.sigmoid_focal_loss(pred, tgt, avg_factor=37.0)
    assert torch.allclose(got, ref.sum() / 37.0, rtol=1e-5)
    x, t = torch.randn(50, generator=g), torch.rand(50, generator=g)
    want = -(t * torch.log(x.sigmoid()) + (1 - t) * torch.log(1 - x.sigmoid())).sum() / 12.0
    assert torch.allclose(losses.sigmoid_bce_loss(x, t, avg_factor=12.0), want, rtol=1e-5)


def test_losses_registry_builds_what_the_configs_name():
    """SURVEY 8b registry surface: the configs' ``loss_bbox=dict(type=...)`` (SGCDet_ScanNet.py:111, SGCDet_ARKit.py:114)
    and the head's default centerness / classification losses (imvoxel_head_v2.py:46-57) resolve in LOSSES, follow
    mmdet's calling convention, and are what the head classes hold."""
    from sgcdet_amd.mmcv_lite import LOSSES
    from sgcdet_amd.plugin import losses
    from sgcdet_amd.plugin.bbox_head import ScanNetImVoxelHeadV2, SunRgbdImVoxelHeadV2
    g = torch.Generator().manual_seed(3)
    lo = torch.rand(40, 3, generator=g)
    a = torch.cat([lo, lo + 0.2 + torch.rand(40, 3, generator=g)], 1)
    b = a + 0.1 * torch.randn(40, 6, generator=g)
    w = torch.rand(40, generator=g)
    aa = LOSSES.build(dict(type="AxisAlignedIoULoss", loss_weight=2.0))
    assert torch.equal(aa(a, b, weight=w, avg_factor=w.sum()), losses.axis_aligned_iou_loss(a, b, w, w.sum(), loss_weight=2.0))
    assert torch.equal(aa(a, b, reduction_override="none"), 2.0 * (1 - losses.axis_aligned_iou(a, b)))
    assert torch.equal(aa(a, b, reduction_override="sum"), (2.0 * (1 - losses.axis_aligned_iou(a, b))).sum())
    with pytest.raises(ValueError):
        aa(a, b, avg_factor=3.0, reduction_override="sum")
    with pytest.raises(ValueError):
        LOSSES.build(dict(type="AxisAlignedIoULoss", reduction="max"))
    p7 = torch.cat([torch.rand(16, 3, generator=g), 0.5 + torch.rand(16, 3, generator=g), torch.rand(16, 1, generator=g)], 1)
    t7 = p7 + 0.05 * torch.randn(16, 7, generator=g)
    rot = LOSSES.build(dict(type="RotatedIoU3DLoss", loss_weight=1.0))
    w7 = torch.rand(16, 7, generator=g)                              # per-coordinate weights are averaged (rotated_iou_loss.py:75)
    assert torch.allclose(rot(p7, t7, weight=w7, avg_factor=4.0), losses.rotated_iou_3d_loss(p7, t7, w7.mean(-1), 4.0))
    assert float(rot(p7, t7, weight=torch.zeros(16), avg_factor=1.0)) == 0.0
    none0 = rot(p7, t7, weight=torch.zeros(16), reduction_override="none")       # all-zero weights keep the [n] shape
    assert none0.shape == (16,) and float(none0.abs().max()) == 0.0
    foc = LOSSES.build(dict(type="FocalLoss", use_sigmoid=True, gamma=1.5, alpha=0.3, loss_weight=0.5))
    x, y = torch.randn(30, 5, generator=g), torch.randint(-1, 5, (30,), generator=g)
    assert torch.equal(foc(x, y, avg_factor=7.0), losses.sigmoid_focal_loss(x, y, 1.5, 0.3, None, 7.0, 0.5))
    ce = LOSSES.build(dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0))
    z, zt = torch.randn(30, generator=g), torch.rand(30, generator=g)
    assert torch.equal(ce(z, zt, avg_factor=5.0), losses.sigmoid_bce_loss(z, zt, avg_factor=5.0))
    for bad in (dict(type="FocalLoss", use_sigmoid=False), dict(type="CrossEntropyLoss"), dict(type="CrossEntropyLoss", use_mask=True, use_sigmoid=True)):
        with pytest.raises(NotImplementedError):
            LOSSES.build(bad)
    with pytest.raises(KeyError):
        LOSSES.build(dict(type="IoU3DLoss"))                         # the reference's (never used) base default :51 does not exist in mmdet3d either
    kw = dict(n_classes=3, n_channels=8, n_scales=3, limit=27)
    head = ScanNetImVoxelHeadV2(n_reg_outs=6, loss_bbox=dict(type="AxisAlignedIoULoss", loss_weight=3.0), **kw)
    assert isinstance(head.loss_bbox, losses.AxisAlignedIoULoss) and head.loss_bbox.loss_weight == 3.0
    assert isinstance(head.loss_cls, losses.FocalLoss) and isinstance(head.loss_centerness, losses.CrossEntropyLoss)
    assert isinstance(SunRgbdImVoxelHeadV2(n_reg_outs=7, **kw).loss_bbox, losses.RotatedIoU3DLoss)
    swapped = SunRgbdImVoxelHeadV2(n_reg_outs=7, loss_bbox=dict(type="AxisAlignedIoULoss"), **kw)   # a config may swap it
    assert isinstance(swapped.loss_bbox, losses.AxisAlignedIoULoss)


def test_indoor_eval_reproduces_the_reference_metrics_on_upright_boxes():
    """row f-4: AP / recall / result keys of the reference's own indoor_eval (fixture), ScanNet-style boxes (no yaw:
    the closed-form IoU on the host); the rotated case runs on the GPU (tests/test_gpu_modules.py)."""
    from eval_contract import check_case
    check_case(0)
    check_case(2)


def test_rotated_iou_3d_loss_value_and_gradients():
    """plugin/losses.py::rotated_iou_3d (the ARKit config's RotatedIoU3DLoss): forward against the exact float64
    polygon clip of the NMS fixture (pairs taken from its 40 x 40 IoU table), gradients by gradcheck in float64,
    and the degenerate pairs (identical boxes, disjoint boxes, one inside the other)."""
    import os
    import numpy as np
    from sgcdet_amd.plugin import losses
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "nms_rotated_multiclass.npz"))
    a, b, want = torch.from_numpy(d["iou_a"]).double(), torch.from_numpy(d["iou_b"]).double(), torch.from_numpy(d["iou_f64"])
    ia, ib = torch.meshgrid(torch.arange(40), torch.arange(40), indexing="ij")
    pa, pb = a[ia.reshape(-1)], b[ib.reshape(-1)]
    inter = losses.rotated_bev_intersection(pa, pb)
    iou2d = inter / (pa[:, 2] * pa[:, 3] + pb[:, 2] * pb[:, 3] - inter)
    assert (iou2d - want.reshape(-1)).abs().max() < 1e-9
    # 3-D: unit height overlap scaling
    g = torch.Generator().manual_seed(3)
    n = 24
    p = torch.cat([(torch.rand(n, 3, generator=g) - 0.5), 0.5 + torch.rand(n, 3, generator=g), (torch.rand(n, 1, generator=g) - 0.5) * 3], 1).double()
    t = p + torch.randn(n, 7, generator=g).double() * 0.08
    t[:, 3:6] = t[:, 3:6].abs() + 0.1
    iou = losses.rotated_iou_3d(p, t)
    assert ((iou > 0.2) & (iou < 1)).all()
    assert torch.allclose(losses.rotated_iou_3d(p, p), torch.ones(n, dtype=torch.float64), atol=1e-9)
    far = p.clone(); far[:, 0] += 10
    assert (losses.rotated_iou_3d(p, far) == 0).all()
    small = p.clone(); small[:, 3:6] *= 0.5
    assert torch.allclose(losses.rotated_iou_3d(small, p), torch.full((n,), 0.125, dtype=torch.float64), atol=1e-9)
    pp = p.clone().requires_grad_()
    assert torch.autograd.gradcheck(lambda x: losses.rotated_iou_3d(x, t), (pp,), eps=1e-6, atol=1e-5)
    w = torch.rand(n, generator=g).double()
    loss = losses.rotated_iou_3d_loss(pp, t, weight=w, avg_factor=w.sum())
    assert torch.allclose(loss, ((1 - iou) * w).sum() / w.sum())
    loss.backward()
    assert torch.isfinite(pp.grad).all() and pp.grad.abs().sum() > 0


def test_depth_net_matches_the_reference_class_on_cpu_paths():
    """DepthNet_Fusion (row f-2) against tests/golden/depth_net.npz, made by the reference's own class
    (tests/golden/make_golden_depthnet.py): identical state-dict key set, the label down-sampling / one-hot / error
    tolerance (pure tensor logic, exact) and -- through the differentiable torch formulation of the plane sweep that the
    module runs under autograd -- the depth distribution and the depth loss."""
    import os
    import sgcdet_amd.plugin as P
    from golden_util import fill_by_name, load, img_meta
    d, _ = load("depth_net")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth_net.npz"))
    stride, dbound = int(z["stride"]), [float(v) for v in z["dbound"]]
    net = P.DepthNet_Fusion(neighbor_img_num=2, downsample_factor=stride, dbound=dbound, mono_channels=d["xs"].shape[2],
                            loss_weight=0.5, max_tol=0, init_weight="none").eval()
    assert sorted(net.state_dict().keys()) == [str(k) for k in z["keys"]]
    fill_by_name(net, base_seed=7, scale=0.15)
    labels = net.get_downsampled_gt_depth(d["depth_maps"])
    assert torch.equal(labels, d["labels"])
    net_tol = P.DepthNet_Fusion(neighbor_img_num=2, downsample_factor=stride, dbound=dbound, mono_channels=d["xs"].shape[2],
                                max_tol=1, init_weight="none")
    assert torch.equal(net_tol.get_downsampled_gt_depth(d["depth_maps"]), d["labels_tol"])
    meta = img_meta(d)
    xs = d["xs"].clone().requires_grad_()                  # autograd on: the grid_sample formulation, runs on the CPU
    pred = net(xs, d["imgs"], [meta], stride)
    assert (pred.detach() - d["pred"]).abs().max() < 2e-5
    loss = net.loss(d["depth_maps"], pred)["loss_dpt"]
    assert abs(float(loss) - float(d["loss"])) < 1e-4
    loss.backward()
    assert xs.grad is not None and torch.isfinite(xs.grad).all() and float(xs.grad.abs().sum()) > 0


def test_bn_rows_is_batchnorm3d_on_the_channels_last_rows():
    """conv_plan.bn_rows (the BatchNorm of the HIP training path, applied to [V, C] rows) == nn.BatchNorm3d on the
    [1, C, X, Y, Z] volume: output, running statistics, num_batches_tracked, gradients; momentum=None (cumulative average)
    and eval mode; a non-BatchNorm3d module goes through the 5-D view."""
    import copy
    import torch
    from sgcdet_amd.plugin.conv_plan import bn_rows, rows_to_ncdhw
    torch.manual_seed(0)
    grid, C = (4, 3, 5), 8
    for momentum, training in ((0.1, True), (None, True), (0.1, False)):
        bn = torch.nn.BatchNorm3d(C, momentum=momentum)
        bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        bn.train(training)
        ref = copy.deepcopy(bn)
        for _ in range(2):                                   # two steps: the cumulative-average momentum changes per step
            rows = torch.randn(grid[0] * grid[1] * grid[2], C, requires_grad=True)
            vol = rows_to_ncdhw(rows.detach(), grid, C).contiguous().requires_grad_(True)
            y = bn_rows(bn, rows, grid)
            y_ref = ref(vol)
            assert torch.allclose(rows_to_ncdhw(y, grid, C), y_ref, atol=1e-6)
            y.square().sum().backward(); y_ref.square().sum().backward()
            assert torch.allclose(rows_to_ncdhw(rows.grad, grid, C), vol.grad, atol=1e-5)
        assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-6) and torch.allclose(bn.running_var, ref.running_var, atol=1e-6)
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)
        assert torch.allclose(bn.weight.grad, ref.weight.grad, atol=1e-4) and torch.allclose(bn.bias.grad, ref.bias.grad, atol=1e-4)
    gn = torch.nn.GroupNorm(2, C)
    rows = torch.randn(60, C)
    assert torch.allclose(rows_to_ncdhw(bn_rows(gn, rows, grid), grid, C), gn(rows_to_ncdhw(rows, grid, C)), atol=1e-6)
    # the block tail (`+ identity`, ReLU) through bn_rows: the torch composition on every path that does not fuse it
    bn = torch.nn.BatchNorm3d(C).train()
    ref = copy.deepcopy(bn)
    rows, skip = torch.randn(60, C, requires_grad=True), torch.randn(60, C, requires_grad=True)
    y = bn_rows(bn, rows, grid, residual=skip, relu=True)
    vol, vskip = rows_to_ncdhw(rows.detach(), grid, C).contiguous().requires_grad_(True), rows_to_ncdhw(skip.detach(), grid, C).contiguous().requires_grad_(True)
    y_ref = torch.relu(ref(vol) + vskip)
    assert torch.allclose(rows_to_ncdhw(y, grid, C), y_ref, atol=1e-6)
    y.square().sum().backward(); y_ref.square().sum().backward()
    assert torch.allclose(rows_to_ncdhw(rows.grad, grid, C), vol.grad, atol=1e-5) and torch.allclose(rows_to_ncdhw(skip.grad, grid, C), vskip.grad, atol=1e-5)


def test_fpn_restates_the_published_module():
    """plugin.fpn.FPN (torch formulation) against the module's definition written out with plain functional calls: lateral
    1x1 + bias, top-down nearest to the finer size (odd sizes: 8 -> 15), 3x3 + bias; state-dict keys as mmdet's."""
    import torch
    import torch.nn.functional as F
    from sgcdet_amd.plugin.fpn import FPN, _nearest_index
    torch.manual_seed(0)
    fpn = FPN([32, 64, 96, 128], 32, 4).eval()
    fpn.init_weights()
    for p in fpn.parameters():
        p.data.add_(torch.randn_like(p) * 0.05)
    feats = [torch.randn(2, c, h, w) for c, (h, w) in zip([32, 64, 96, 128], [(60, 80), (30, 40), (15, 20), (8, 10)])]
    sd = fpn.state_dict()
    assert set(sd) == {f"{g}.{i}.conv.{t}" for g in ("lateral_convs", "fpn_convs") for i in range(4) for t in ("weight", "bias")}
    lat = [F.conv2d(f, sd[f"lateral_convs.{i}.conv.weight"], sd[f"lateral_convs.{i}.conv.bias"]) for i, f in enumerate(feats)]
    for i in (3, 2, 1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
    want = [F.conv2d(l, sd[f"fpn_convs.{i}.conv.weight"], sd[f"fpn_convs.{i}.conv.bias"], padding=1) for i, l in enumerate(lat)]
    got = fpn(feats)
    assert len(got) == 4 and all(torch.allclose(g, w, atol=1e-5) for g, w in zip(got, want))
    # the index rule the kernel path uses for the top-down step == F.interpolate(mode="nearest")
    x = torch.arange(8 * 10, dtype=torch.float32).view(1, 1, 8, 10)
    up = F.interpolate(x, size=(15, 20), mode="nearest")
    assert torch.equal(up[0, 0], x[0, 0][_nearest_index(15, 8, "cpu")][:, _nearest_index(20, 10, "cpu")])


def test_bench_refuses_to_mislabel_the_gpu_count():
    """bench.py --gpus N: without a launcher it starts N ranks itself and needs N visible GPUs (here: none -> non-zero
    exit, before anything touches a GPU); under a launcher WORLD_SIZE must equal N."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 2 requested but only" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_committed_bench_lines_follow_from_their_own_fields():
    """Every bench line of this round under profiles/ is self-consistent (tools/roofline_check.py): `roofline.frac` =
    achieved / peak with achieved = algorithmic bytes (or issued FLOPs) / the measured average launch time, `path_roofline.frac`
    = floor / measured time per scene, `value` = scenes per second of the timed region -- no hand-edited or stale field."""
    import glob
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from roofline_check import check_line
    files = sorted(glob.glob(os.path.join(root, "profiles", "r04_bench_*.json")) + glob.glob(os.path.join(root, "profiles", "r05_bench_*.json"))
                   + glob.glob(os.path.join(root, "profiles", "r06_bench_*.json")))
    assert len(files) >= 8
    for f in files:
        line = open(f).readline()
        d = json.loads(line)
        assert check_line(d, os.path.basename(f)) == []
        assert d["self_check"]["mismatching"] == 0


def test_dense_head_flat_tag_follows_the_content_not_the_move():
    """DenseHead tags the vox_coords it BUILDS as flat (column 3 == arange); a module move (.to / .cuda builds a
    new tensor object) carries the tag over only while it is still valid.  build -> load_state_dict -> move: the in-place load
    voids the tag, so the moved buffer is untagged and the transformer checks the loaded content itself."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_head
    w = workload("cfg1_plumbing")
    head = build_head(model_config(w)["voxel_head"]).base_heads[0]
    enc = head.cross_transformer

    def valid_tag(t):
        tag = getattr(t, "_sgc_flat", None)
        return tag is not None and tag[0] == t._version

    assert valid_tag(head.vox_coords) and enc._coords_are_flat(head.vox_coords)
    head._apply(lambda t: t.clone())                       # what .to() / .cuda() do: a new tensor object per buffer
    assert valid_tag(head.vox_coords) and head.vox_coords._sgc_flat[1]
    sd = head.state_dict()
    sd["vox_coords"] = sd["vox_coords"].clone()
    sd["vox_coords"][:, 3] = sd["vox_coords"][:, 3].flip(0)          # a checkpoint whose column 3 is NOT arange
    head.load_state_dict(sd)
    assert not valid_tag(head.vox_coords)                  # the in-place copy bumped the version
    head._apply(lambda t: t.clone())
    assert not valid_tag(head.vox_coords)                  # ... and the move must not re-assert "flat"
    assert enc._coords_are_flat(head.vox_coords) is False  # one host check of the loaded content
    assert valid_tag(head.vox_coords) and head.vox_coords._sgc_flat[1] is False


def test_oracle_batched_weight_pack_is_its_single_pack(oracle_ops):
    """The CPU twin of sgc_pack_conv_weight_batch (host item list, same 64-byte sgc_pack_item) == sgc_pack_conv_weight per item;
    the GPU comparison is tests/test_gpu_conv3d.py::test_batched_weight_pack_is_the_single_pack."""
    import torch
    g = torch.Generator().manual_seed(5)
    entries, want = [], []
    for shp, (transpose, flip, pr, pc) in (((12, 7, 27), (False, False, 4, 1)), ((12, 7, 27), (True, True, 1, 32)), ((9, 40, 1), (True, False, 1, 32)),
                                           ((16, 8, 8), (False, False, 1, 1))):
        w = torch.randn(*shp, generator=g)
        T, R, C = oracle_ops.packed_shape(shp, transpose, pr, pc)
        hi = torch.zeros(T, R, C, dtype=torch.bfloat16)
        entries.append((w, hi, torch.zeros_like(hi), transpose, flip))
        want.append(oracle_ops.pack_conv_weight(w, transpose=transpose, flip=flip, pad_rows=pr, pad_cols=pc))
    oracle_ops.run_pack_plan(oracle_ops.pack_conv_weight_plan(entries))
    for (w, hi, lo, *_), (whi, wlo) in zip(entries, want):
        assert torch.equal(hi.view(torch.int16), whi.view(torch.int16)) and torch.equal(lo.view(torch.int16), wlo.view(torch.int16))
