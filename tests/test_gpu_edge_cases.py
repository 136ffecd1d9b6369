"""Edge cases of the pair-list path: no visible pairs at all, cameras that see nothing (ragged pair
list), voxels seen by exactly one camera, empty query sets, samples entirely outside the maps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(C=32):
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_head
    from sgcdet_amd.scene import model_config
    w = dict(embed_dims=C, n_voxels_list=[(4, 4, 2), (8, 8, 4), (16, 16, 8)],
             voxel_size_list=[(.64, .64, .8), (.32, .32, .4), (.16, .16, .2)], topk_list=[64, 512],
             head="ScanNetImVoxelHeadV2", n_classes=18, n_reg_outs=6)
    torch.manual_seed(3)
    head = build_head(model_config(w)["voxel_head"]).eval()
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for _, p in head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.05)
    return head, w


def _scene(n_views, C, seed, look_away=()):
    from sgcdet_amd.scene import make_scene
    feats, dpt, meta = make_scene(n_views, C, seed=seed, pad_shape=(60, 80))
    meta["img_shape"], meta["ori_shape"] = (59, 80, 3), (240, 320, 3)
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = np.array([[290.0, 0, 160.0], [0, 290.0, 120.0], [0, 0, 1]], dtype=np.float32)
    meta["lidar2img"]["intrinsic"] = K
    for i in look_away:                       # flip the camera: everything is behind it
        E = meta["lidar2img"]["extrinsic"][i].copy()
        E[2, :] *= -1
        E[0, :] *= -1
        meta["lidar2img"]["extrinsic"][i] = E
    return feats, dpt, meta


def _run_both(head, w, feats, dpt, meta):
    from oracle.ref_path import RefPath
    import torch.nn.functional as F
    dpts = [dpt, F.interpolate(dpt, scale_factor=(1, 0.5, 0.5), mode="nearest"),
            F.interpolate(dpt, scale_factor=(1, 0.25, 0.25), mode="nearest")]
    rp = RefPath(head.state_dict(), dict(embed_dims=w["embed_dims"], n_voxels_list=w["n_voxels_list"],
                                         voxel_size_list=w["voxel_size_list"], topk_list=w["topk_list"],
                                         dbound=(0.2, 5.0), num_heads=8, num_points=4))
    vol_c, valid_c, occ_c = rp.adaptive_sparse_head(feats, meta, dpts)
    head = head.cuda()
    with torch.no_grad():
        vol_g, valid_g, occ_g = head([f.cuda() for f in feats], meta, [d.cuda() for d in dpts])
    return (vol_g.cpu(), valid_g.cpu(), occ_g.cpu()), (vol_c, valid_c, occ_c)


def test_ragged_pair_list_some_cameras_see_nothing():
    head, w = _build()
    feats, dpt, meta = _scene(4, 32, seed=21, look_away=(1, 3))
    (vol_g, valid_g, occ_g), (vol_c, valid_c, occ_c) = _run_both(head, w, feats, dpt, meta)
    from oracle.compare import check_sparse_head
    res = check_sparse_head(vol_g, valid_g, occ_g, vol_c, valid_c, occ_c, 16 * 16 * 8, w["topk_list"], feat_tol=1e-4)
    assert res["tie_flips"] <= 4


def test_no_camera_sees_any_voxel():
    """Every (camera, voxel) pair invisible: the cross attention contributes zeros and the path must not
    divide by a zero count, launch empty grids with garbage, or read past the empty pair list."""
    head, w = _build()
    feats, dpt, meta = _scene(3, 32, seed=22, look_away=(0, 1, 2))
    head = head.cuda()
    import torch.nn.functional as F
    dpts = [dpt, F.interpolate(dpt, scale_factor=(1, 0.5, 0.5), mode="nearest"),
            F.interpolate(dpt, scale_factor=(1, 0.25, 0.25), mode="nearest")]
    with torch.no_grad():
        vol, valid, occ = head([f.cuda() for f in feats], meta, [d.cuda() for d in dpts])
    assert torch.isfinite(vol).all() and torch.isfinite(occ).all()
    assert int(valid.sum()) == w["topk_list"][-1]
    # with no image evidence every voxel gets the same feature (LayerNorm/FFN of zeros), up to trilinear mixing
    with torch.no_grad():
        lvl0 = head.base_heads[0]([feats[2].cuda()[:, :, :, :3, :5]], meta, mlvl_dpt_dists=[dpts[2].cuda()[:, :, :, :3, :5]])
    flat = lvl0[0].reshape(32, -1)
    assert (flat - flat[:, :1]).abs().max() < 1e-6
    # the autograd (reference-layout) path must survive an empty pair list too
    for m in head.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    lvl0_t = head.base_heads[0]([feats[2].cuda()[:, :, :, :3, :5]], meta, mlvl_dpt_dists=[dpts[2].cuda()[:, :, :, :3, :5]])
    assert lvl0_t.requires_grad and (lvl0_t - lvl0).abs().max() < 1e-4     # inference FFN: bf16x3 MFMA (~5e-6), training: torch fp32


def test_kernel_level_empty_and_outside(gpu_ops, oracle_ops):
    N, H, W, C, D, M, P, Nq = 2, 5, 6, 32, 12, 8, 4, 9
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(N, H * W, C, generator=g).cuda()
    dist = torch.randn(N, H * W, D, generator=g).softmax(-1).contiguous().cuda()
    # all-zero mask -> empty pair list, totals = 0
    mask = torch.zeros(N, Nq, dtype=torch.uint8).cuda()
    pc = gpu_ops.compact_pairs(mask)
    assert pc["totals"].tolist()[:3] == [0, 0, 0]
    assert (pc["slot"] == -1).all() and pc["cam_offset"].tolist() == [0, 0, 0]
    ref_cam = torch.rand(N, Nq, 3, generator=g).cuda()
    out = gpu_ops.pairs_geometry_sample(feat, dist, ref_cam, pc["pair_cam"], pc["pair_q"], 0, H, W)
    assert out.shape == (0, C)
    # device-side pair count of zero: the launch covers `cap` rows but must write nothing
    out = gpu_ops.pairs_geometry_sample(feat, dist, ref_cam, pc["pair_cam"], pc["pair_q"], -1, H, W, totals=pc["totals"])
    torch.cuda.synchronize()
    # samples entirely outside the map / depth range -> exact zeros (reference gates, kernel.cuh:137,289)
    shapes3 = torch.tensor([[H, W, D]]).cuda()
    lsi = torch.zeros(1, dtype=torch.int64).cuda()
    loc = torch.full((N, 3, M, 1, P, 3), 0.5).cuda()
    loc[:, 0, ..., 0] = 1.5       # x beyond the right border
    loc[:, 1, ..., 1] = -0.3      # y above the top border
    loc[:, 2, ..., 2] = 1.2       # depth beyond the last bin
    attn = torch.rand(N, 3, M, 1, P, generator=g).cuda()
    o, sc = gpu_ops.dfa3d_forward(feat.view(N, H * W, M, C // M), dist.view(N, H * W, 1, D), shapes3, lsi, loc, attn, want_score=True)
    assert (o == 0).all() and (sc == 0).all()
    oc, scc = oracle_ops.dfa3d_forward(feat.cpu().view(N, H * W, M, C // M), dist.cpu().view(N, H * W, 1, D), shapes3.cpu(),
                                       lsi.cpu(), loc.cpu(), attn.cpu(), want_score=True)
    assert (oc == 0).all() and (scc == 0).all()
    # zero queries
    empty = gpu_ops.dfa3d_forward(feat.view(N, H * W, M, C // M), dist.view(N, H * W, 1, D), shapes3, lsi,
                                  torch.zeros(N, 0, M, 1, P, 3).cuda(), torch.zeros(N, 0, M, 1, P).cuda())[0]
    assert empty.shape == (N, 0, C)


def test_round2_entry_points_on_degenerate_shapes(oracle_ops, gpu_ops):
    """wgrad with fewer output voxels than one K-step, minimal channel counts and an odd stride-2 grid; the 2-D convolution on
    1 x 1 images; the many-workgroup top-k at exact chunk multiples; the view-softmax backward with a single camera."""
    g = torch.Generator().manual_seed(0)
    for cin, cout, grid, k, s in [(4, 4, (2, 3, 2), 3, 1), (32, 8, (7, 5, 3), 3, 2), (8, 12, (3, 1, 1), 1, 1), (16, 4, (2, 2, 2), 2, 2)]:
        pad = 0 if k == 2 else k // 2
        og = tuple((d + 2 * pad - k) // s + 1 for d in grid)
        x = torch.randn(grid[0] * grid[1] * grid[2], cin, generator=g)
        dy = torch.randn(og[0] * og[1] * og[2], cout, generator=g)
        ref = oracle_ops.conv3d_wgrad_bf16x3(x, dy, grid, k, s)
        got = gpu_ops.conv3d_wgrad_bf16x3(x.cuda(), dy.cuda(), grid, k, s).cpu()
        assert float((got - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), (cin, cout, grid, k, s)
    x = torch.randn(5, 32, generator=g)                               # five 1 x 1 images
    w = torch.randn(9, 4, 32, generator=g) * 0.1
    hi, lo = gpu_ops.split_bf16(w)
    ref = oracle_ops.conv2d_nhwc_bf16x3(x, hi, lo, (5, 1, 1), 3)
    got = gpu_ops.conv2d_nhwc_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), (5, 1, 1), 3).cpu()
    assert float((got - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    try:
        gpu_ops.lib.call("sgc_set_tuning", b"topk_multi_min", 1)
        for n, k in [(4096, 1), (8192, 8192), (12288, 4097), (1, 1)]:
            s = torch.rand(n, generator=g)
            s[: n // 2] = 0.5
            a = gpu_ops.topk_select(s.cuda(), k, want_valid=True, want_mask=True)
            b = oracle_ops.topk_select(s, k, want_valid=True, want_mask=True)
            assert all(torch.equal(u.cpu(), v) for u, v in zip(a, b)), (n, k)
    finally:
        gpu_ops.lib.call("sgc_set_tuning", b"topk_multi_min", 32769)
    N, Nq, C, heads = 1, 9, 32, 8                                     # one camera: softmax over one view, d_score = 0
    slot = torch.arange(Nq, dtype=torch.int32).view(1, Nq)
    vi = torch.arange(Nq, dtype=torch.int32)
    q, kv, go = torch.randn(Nq, C, generator=g), torch.randn(Nq, 2 * C, generator=g), torch.randn(Nq, C, generator=g)
    ctx = gpu_ops.view_attend(q.cuda(), kv.cuda(), slot.cuda(), vi.cuda(), heads)
    gq, gkv = gpu_ops.view_attend_backward(q.cuda(), kv.cuda(), slot.cuda(), vi.cuda(), heads, ctx, go.cuda())
    assert float(gq.abs().max()) < 1e-6 and float(gkv[:, :C].abs().max()) < 1e-6        # softmax of one score is constant
    assert torch.allclose(gkv[:, C:].cpu(), go, atol=1e-6) and torch.allclose(ctx.cpu(), kv[:, C:], atol=1e-6)
