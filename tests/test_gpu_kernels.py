"""GPU parity: every entry point of the HIP library (through the C ABI) against the CPU
oracle on the same seeded inputs.  Tolerances: indices / masks bit-exact; fp32 features
1e-5 relative to the tensor scale (north star bar: 1e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, M, Cm, D, Q, P, levels[(H,W)], dist_heads_is_M)
SHAPES = [
    (3, 8, 32, 12, 257, 4, [(15, 20)], False),     # hot shape A (context-aware gather), C=256
    (3, 1, 256, 12, 300, 1, [(15, 20)], False),    # hot shape B (geometry sample)
    (2, 8, 16, 12, 130, 4, [(14, 20)], False),     # large configs: C=128
    (2, 8, 4, 12, 77, 4, [(7, 9)], True),          # tiny C=32, replicated depth
    (2, 4, 8, 6, 50, 3, [(5, 7), (3, 4)], True),   # multi-level, P not a power of two
    (2, 2, 5, 7, 33, 2, [(6, 5)], False),          # Cm % 4 != 0 -> scalar path
    (1, 8, 32, 16, 40, 8, [(12, 20), (6, 10), (3, 5), (2, 3)], True),  # unittest_DFA3D-like: L=4, P=8
]


def make_inputs(shape, seed, dev="cpu"):
    B, M, Cm, D, Q, P, levels, rep = shape
    g = torch.Generator().manual_seed(seed)
    L = len(levels)
    S = sum(h * w for h, w in levels)
    shapes3 = torch.tensor([[h, w, D] for h, w in levels], dtype=torch.int64)
    lsi = torch.tensor([0] + [h * w for h, w in levels], dtype=torch.int64).cumsum(0)[:-1].contiguous()
    value = torch.randn(B, S, M, Cm, generator=g)
    dh = M if rep else 1
    dist = torch.randn(B, S, dh, D, generator=g).mul(2).softmax(-1).contiguous()
    # locations spill over the borders on every axis to exercise the gates
    loc = (torch.rand(B, Q, M, L, P, 3, generator=g) * 1.3 - 0.15).contiguous()
    attn = torch.rand(B, Q, M, L, P, generator=g)
    t = dict(value=value, dist=dist, shapes3=shapes3, lsi=lsi, loc=loc, attn=attn)
    return {k: v.to(dev) for k, v in t.items()}


def max_abs(t):
    return float(t.detach().abs().max())


def close(a, b, tol=1e-5):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"max abs err {err} (scale {scale})"


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_forward_matches_oracle(shape, oracle_ops, gpu_ops):
    c = make_inputs(shape, 0)
    g = {k: v.cuda() for k, v in c.items()}
    out_c, sc_c = oracle_ops.dfa3d_forward(c["value"], c["dist"], c["shapes3"], c["lsi"], c["loc"], c["attn"], want_score=True)
    out_g, sc_g = gpu_ops.dfa3d_forward(g["value"], g["dist"], g["shapes3"], g["lsi"], g["loc"], g["attn"], want_score=True)
    close(out_g, out_c)
    close(sc_g, sc_c)
    # attention weights omitted == all ones (Grid_Sample_3D_Feature)
    out_c1, _ = oracle_ops.dfa3d_forward(c["value"], c["dist"], c["shapes3"], c["lsi"], c["loc"], None)
    out_g1, _ = gpu_ops.dfa3d_forward(g["value"], g["dist"], g["shapes3"], g["lsi"], g["loc"], None)
    close(out_g1, out_c1)


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[7]])
def test_split_ext_operators_match_oracle_and_fused(shape, oracle_ops, gpu_ops):
    """The implied KAT of unittest_DFA3D.py:9-29: two-stage == one-stage."""
    from sgcdet_amd import ext
    c = make_inputs(shape, 1)
    g = {k: v.cuda() for k, v in c.items()}
    sc_c = oracle_ops.depth_score_forward(c["dist"], c["shapes3"], c["lsi"], c["loc"])
    sc_g = ext.ms_depth_score_sample_forward(g["dist"], g["shapes3"], g["lsi"], g["loc"], im2col_step=32)
    close(sc_g, sc_c)
    s2c, l2c = c["shapes3"][:, :2].contiguous(), c["loc"][..., :2].contiguous()
    s2g, l2g = g["shapes3"][:, :2].contiguous(), g["loc"][..., :2].contiguous()
    out_c = oracle_ops.wms_forward(c["value"], s2c, c["lsi"], l2c, c["attn"], sc_c)
    out_g = ext.wms_deform_attn_forward(g["value"], s2g, g["lsi"], l2g, g["attn"], sc_g, im2col_step=32)
    close(out_g, out_c)
    fused, _ = gpu_ops.dfa3d_forward(g["value"], g["dist"], g["shapes3"], g["lsi"], g["loc"], g["attn"])
    close(fused, out_g, tol=2e-6)


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_backward_matches_oracle(shape, oracle_ops, gpu_ops):
    c = make_inputs(shape, 2)
    g = {k: v.cuda() for k, v in c.items()}
    B, M, Cm = shape[0], shape[1], shape[2]
    go = torch.randn(B, shape[4], M * Cm, generator=torch.Generator().manual_seed(9))
    rc = oracle_ops.dfa3d_backward(c["value"], c["dist"], c["shapes3"], c["lsi"], c["loc"], c["attn"], go)
    rg = gpu_ops.dfa3d_backward(g["value"], g["dist"], g["shapes3"], g["lsi"], g["loc"], g["attn"], go.cuda())
    for name, a, b in zip(("grad_value", "grad_dist", "grad_loc", "grad_attn"), rg, rc):
        close(a, b, tol=2e-5)


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[7]][:2])
def test_split_backward_matches_oracle(shape, oracle_ops, gpu_ops):
    from sgcdet_amd import ext
    c = make_inputs(shape, 3)
    g = {k: v.cuda() for k, v in c.items()}
    B, M, Cm, D, Q, P, levels, _ = shape
    L = len(levels)
    go = torch.randn(B, Q, M * Cm, generator=torch.Generator().manual_seed(5))
    sc = oracle_ops.depth_score_forward(c["dist"], c["shapes3"], c["lsi"], c["loc"])
    s2, l2 = c["shapes3"][:, :2].contiguous(), c["loc"][..., :2].contiguous()

    def run(o_wms, o_dsb, t, go_, sc_, s2_, l2_):
        gv = torch.zeros_like(t["value"]); gl2 = torch.zeros_like(l2_)
        ga = torch.zeros_like(t["attn"]); gs = torch.zeros_like(sc_)
        o_wms(t["value"], s2_, t["lsi"], l2_, t["attn"], sc_, go_, gv, gl2, ga, gs)
        gd = torch.zeros_like(t["dist"]); gl3 = torch.zeros_like(t["loc"])
        o_dsb(t["dist"], t["shapes3"], t["lsi"], t["loc"], gs, gd, gl3)
        return gv, gl2, ga, gs, gd, gl3

    rc = run(oracle_ops.wms_backward, oracle_ops.depth_score_backward, c, go, sc, s2, l2)
    rg = run(lambda *a: ext.wms_deform_attn_backward(*a, im2col_step=64),
             lambda *a: ext.ms_depth_score_sample_backward(*a, im2col_step=64),
             g, go.cuda(), sc.cuda(), s2.cuda(), l2.cuda())
    for a, b in zip(rg, rc):
        close(a, b, tol=2e-5)


def _scene(N, Nq, seed):
    g = torch.Generator().manual_seed(seed)
    ref3d = (torch.rand(Nq, 3, generator=g) - 0.5) * torch.tensor([6.4, 6.4, 3.2])
    origin = torch.tensor([0.0, 0.0, 0.5])
    proj = torch.zeros(N, 3, 4)
    for i in range(N):
        a = 2 * 3.14159265 * i / N
        pos = torch.tensor([2.2 * torch.cos(torch.tensor(a)), 2.2 * torch.sin(torch.tensor(a)), 1.4])
        fwd = torch.tensor([0., 0., 0.6]) - pos
        fwd = fwd / fwd.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0., 0., 1.])); right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        R = torch.stack([right, down, fwd])
        E = torch.cat([R, (-R @ pos)[:, None]], 1)
        K = torch.tensor([[288.8, 0, 159.6], [0, 288.2, 121.0], [0, 0, 1.0]])
        proj[i] = K @ E
    return ref3d.contiguous(), origin, proj.contiguous()


@pytest.mark.parametrize("N,Nq,form", [(5, 400, 1), (13, 3001, 1), (13, 3001, 0), (100, 7000, 1), (3, 1024, 1), (40, 1025, 1), (2, 9, 1)])
def test_projection_and_compaction_bit_exact(N, Nq, form, oracle_ops, gpu_ops):
    """``form`` 1: the two-launch segment form of sgc_compact_pairs (round 5; one and many 1024-query segments, ragged last
    segment, a single query), 0: the five kernels it replaces -- both against the oracle, every output, and the inverse map."""
    gpu_ops.lib.call("sgc_set_tuning", b"compact2", form)
    try:
        _compaction_case(N, Nq, oracle_ops, gpu_ops)
    finally:
        gpu_ops.lib.call("sgc_set_tuning", b"compact2", 1)


def _compaction_case(N, Nq, oracle_ops, gpu_ops):
    ref3d, origin, proj = _scene(N, Nq, 3)
    rc_c, mk_c = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    rc_g, mk_g = gpu_ops.project_points(ref3d.cuda(), origin.cuda(), proj.cuda(), 320., 239., 0.2, 5.0)
    assert torch.equal(mk_g.cpu(), mk_c)                      # visibility mask: bit-exact
    assert torch.equal(rc_g.cpu(), rc_c)                      # fixed arithmetic order: bit-exact
    assert 0 < mk_c.sum() < mk_c.numel()
    pc = oracle_ops.compact_pairs(mk_c)
    pg = gpu_ops.compact_pairs(mk_g)
    tot = pg["totals"].cpu()
    assert torch.equal(tot[:3], pc["totals"][:3])
    n_pairs, n_valid = int(tot[0]), int(tot[1])
    for k in ("cam_count", "cam_offset", "slot", "vox_count"):
        assert torch.equal(pg[k].cpu(), pc[k]), k
    assert torch.equal(pg["pair_cam"][:n_pairs].cpu(), pc["pair_cam"][:n_pairs])
    assert torch.equal(pg["pair_q"][:n_pairs].cpu(), pc["pair_q"][:n_pairs])
    assert torch.equal(pg["valid_index"][:n_valid].cpu(), pc["valid_index"][:n_valid])
    assert int(tot[3]) == 0 and torch.equal(pg["row_of"].cpu(), pc["row_of"])          # inverse of valid_index (sgc_level_tail's gather index)
    inv = torch.full((Nq,), -1, dtype=torch.int32)
    inv[pc["valid_index"][:n_valid].long()] = torch.arange(n_valid, dtype=torch.int32)
    assert torch.equal(pg["row_of"].cpu(), inv)


@pytest.mark.parametrize("C,M,P,HW", [(256, 8, 4, (15, 20)), (128, 8, 4, (14, 20)), (32, 8, 4, (7, 10))])
def test_pair_list_gathers_and_view_pool(C, M, P, HW, oracle_ops, gpu_ops):
    N, Nq, D = 6, 700, 12
    H, W = HW
    ref3d, origin, proj = _scene(N, Nq, 4)
    rc, mk = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    pc = oracle_ops.compact_pairs(mk)
    n_pairs, n_valid = int(pc["totals"][0]), int(pc["totals"][1])
    g = torch.Generator().manual_seed(11)
    feat = torch.randn(N, H * W, C, generator=g)
    dist = torch.randn(N, H * W, D, generator=g).mul(2).softmax(-1).contiguous()
    raw = torch.randn(n_pairs, M * P * 4, generator=g)
    raw[:, :M * P * 3] *= 3.0
    cu = lambda t: t.cuda()
    geo_c = oracle_ops.pairs_geometry_sample(feat, dist, rc, pc["pair_cam"], pc["pair_q"], n_pairs, H, W)
    geo_g = gpu_ops.pairs_geometry_sample(cu(feat), cu(dist), cu(rc), cu(pc["pair_cam"]), cu(pc["pair_q"]), n_pairs, H, W)
    close(geo_g, geo_c)
    # same call with the pair count left on the device
    geo_g2 = gpu_ops.pairs_geometry_sample(cu(feat), cu(dist), cu(rc), cu(pc["pair_cam"]), cu(pc["pair_q"]), -1, H, W,
                                           totals=cu(pc["totals"]))
    close(geo_g2[:n_pairs], geo_c)
    value = feat.view(N, H * W, M, C // M)
    dg_c = oracle_ops.pairs_deform_gather(value, dist, rc, raw, pc["pair_cam"], pc["pair_q"], n_pairs, H, W, M, P)
    dg_g = gpu_ops.pairs_deform_gather(cu(value), cu(dist), cu(rc), cu(raw), cu(pc["pair_cam"]), cu(pc["pair_q"]),
                                       n_pairs, H, W, M, P)
    close(dg_g, dg_c)
    dp = gpu_ops.depth_pairs(cu(dist), H, W)
    assert torch.equal(dp.cpu(), oracle_ops.depth_pairs(dist, H, W))
    dg_g2 = gpu_ops.pairs_deform_gather(cu(value), cu(dist), cu(rc), cu(raw), cu(pc["pair_cam"]), cu(pc["pair_q"]),
                                        n_pairs, H, W, M, P, dist_pairs=dp)
    close(dg_g2, dg_c)                                       # pair-interleaved depth taps: same results
    vbuf = torch.cat([cu(value).reshape(N * H * W, C), torch.zeros(1, C, device="cuda")])
    dg_g3 = gpu_ops.pairs_deform_gather(vbuf[:N * H * W].view(N, H * W, M, C // M), cu(dist), cu(rc), cu(raw),
                                        cu(pc["pair_cam"]), cu(pc["pair_q"]), n_pairs, H, W, M, P, dist_pairs=dp,
                                        zero_row=True)
    close(dg_g3, dg_c)                                       # outside corners -> appended zero row: same results
    mean_c = oracle_ops.view_mean(dg_c, pc["slot"], pc["valid_index"], n_valid)
    mean_g = gpu_ops.view_mean(dg_g, cu(pc["slot"]), cu(pc["valid_index"]), n_valid)
    close(mean_g, mean_c)
    q = torch.randn(n_valid, C, generator=g)
    kv = torch.randn(n_pairs, 2 * C, generator=g)
    ctx_c = oracle_ops.view_attend(q, kv, pc["slot"], pc["valid_index"], 8)
    ctx_g = gpu_ops.view_attend(cu(q), cu(kv), cu(pc["slot"]), cu(pc["valid_index"]), 8)
    close(ctx_g, ctx_c)
    # scatter + transpose glue
    vol_c = torch.zeros(Nq, C); vol_g = torch.zeros(Nq, C).cuda()
    oracle_ops.scatter_rows(ctx_c, pc["valid_index"][:n_valid].contiguous(), vol_c)
    gpu_ops.scatter_rows(ctx_g, cu(pc["valid_index"][:n_valid].contiguous()), vol_g)
    close(vol_g, vol_c)
    src = torch.randn(N, C, H + 1, W + 3, generator=g)
    assert torch.equal(gpu_ops.nchw_to_nhwc_crop(cu(src), H, W).cpu(), oracle_ops.nchw_to_nhwc_crop(src, H, W))
    # the 64 x 64 / 16-byte form (C % 64 == 0, W and the source pitch % 4 == 0): crops, partial pixel tiles, two channel tiles
    for n_, c_, hs, ws, h_, w_ in ((2, 64, 9, 12, 7, 8), (3, 128, 15, 20, 14, 20), (1, 256, 64, 80, 64, 80), (2, 64, 5, 8, 5, 4)):
        src = torch.randn(n_, c_, hs, ws, generator=g)
        assert torch.equal(gpu_ops.nchw_to_nhwc_crop(cu(src), h_, w_).cpu(), oracle_ops.nchw_to_nhwc_crop(src, h_, w_)), (c_, h_, w_)


def test_errors_are_raised_not_printed(gpu_ops):
    v = torch.zeros(1, 4, 1, 4, device="cuda")
    d = torch.zeros(1, 4, 1, 3, device="cuda")
    sh = torch.tensor([[2, 2, 3]], device="cuda")
    lsi = torch.zeros(1, dtype=torch.int64, device="cuda")
    loc = torch.zeros(1, 1, 1, 1, 1, 3, device="cuda")
    with pytest.raises(RuntimeError):
        gpu_ops.dfa3d_forward(v.cpu(), d, sh, lsi, loc)           # device mismatch
    with pytest.raises(RuntimeError):
        gpu_ops.dfa3d_forward(v.transpose(1, 3), d, sh, lsi, loc)  # non-contiguous
    out, _ = gpu_ops.dfa3d_forward(v, d, sh, lsi, loc)
    assert out.shape == (1, 1, 4)


@pytest.mark.parametrize("C,grid", [(256, (5, 6, 4)), (128, (10, 10, 4)), (32, (3, 4, 2))])
def test_upsample_occ_and_scatter_add(C, grid, oracle_ops, gpu_ops):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(C)
    V = grid[0] * grid[1] * grid[2]
    vol = torch.randn(V, C, generator=g)
    w, b = torch.randn(C, generator=g) * 0.1, torch.randn(1, generator=g)
    up_c, occ_c, og = oracle_ops.upsample2x_occ(vol, grid, w, b)
    up_g, occ_g, og_g = gpu_ops.upsample2x_occ(vol.cuda(), grid, w.cuda(), b.cuda())
    assert og == og_g
    close(up_g, up_c, tol=1e-6)
    close(occ_g, occ_c, tol=2e-6)
    # and against torch's own trilinear kernel (the reference's call, AdaptiveSparseHead.py:64-69)
    x = vol.view(*grid, C).permute(3, 0, 1, 2)[None]
    ref = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)[0].permute(1, 2, 3, 0).reshape(-1, C)
    close(up_g, ref, tol=1e-6)
    up_only, none, _ = gpu_ops.upsample2x_occ(vol.cuda(), grid)
    assert none is None and torch.equal(up_only, up_g)
    idx = torch.randperm(8 * V, generator=g)[: 2 * V].sort().values
    rows = torch.randn(2 * V, C, generator=g)
    want = up_c.clone()
    oracle_ops.scatter_add_rows(rows, idx, want)
    got = gpu_ops.scatter_add_rows(rows.cuda(), idx.cuda(), up_g.clone())
    close(got, want, tol=1e-6)


def test_row_counts_left_on_the_device(oracle_ops, gpu_ops):
    from count_contract import check_row_counts
    check_row_counts(gpu_ops, oracle_ops, "cuda")


def test_nms_matches_reference_golden_and_oracle(oracle_ops, gpu_ops):
    import numpy as np
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "nms_aligned.npz"))
    for k in range(int(d["n_cases"])):
        boxes, scores, labels = (torch.from_numpy(d[f"{n}{k}"]) for n in ("boxes", "scores", "labels"))
        keep = gpu_ops.aligned_nms3d(boxes.cuda(), scores.cuda(), labels.cuda(), float(d[f"thr{k}"]))
        assert torch.equal(keep.cpu(), torch.from_numpy(d[f"keep{k}"])), k
    g = torch.Generator().manual_seed(21)
    for n in (3000, 4096):                                  # the head's 3 x nms_pre candidates; the size limit
        c = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([6.4, 6.4, 2.5])
        c = c[torch.randint(0, n // 20, (n,), generator=g)] + torch.randn(n, 3, generator=g) * 0.05
        s = 0.4 + torch.rand(n, 3, generator=g)
        boxes = torch.cat([c - s / 2, c + s / 2], 1)
        scores = torch.rand(n, generator=g)
        labels = torch.randint(0, 18, (n,), generator=g)
        want = oracle_ops.aligned_nms3d(boxes, scores, labels, 0.25)
        got = gpu_ops.aligned_nms3d(boxes.cuda(), scores.cuda(), labels.cuda(), 0.25)
        assert torch.equal(got.cpu(), want) and 0 < want.numel() < n
    assert gpu_ops.aligned_nms3d(torch.zeros(0, 6).cuda(), torch.zeros(0).cuda(), torch.zeros(0, dtype=torch.int64).cuda(), 0.25).numel() == 0
    with pytest.raises(Exception):
        gpu_ops.aligned_nms3d(torch.zeros(5000, 6).cuda(), torch.zeros(5000).cuda(), torch.zeros(5000, dtype=torch.int64).cuda(), 0.25)


def test_rotated_nms_matches_reference_golden_and_oracle(oracle_ops, gpu_ops):
    """ARKit post-processing (row f-4): golden of the reference's multiclass glue, the float64 clip table, the
    textbook loop, and -- at the head's real size, 3 x nms_pre candidates x 17 classes, score_thr 0 -- the CPU
    oracle: same survivors in the same order for every class."""
    from nms_rotated_contract import (arkit_like, bev_of, check_iou_against_float64_clip, check_mask_sweep_equals_textbook_loop,
                                      check_multiclass_golden)
    check_multiclass_golden(gpu_ops, "cuda")
    check_iou_against_float64_clip(gpu_ops, "cuda")
    check_mask_sweep_equals_textbook_loop(gpu_ops, "cuda")
    # pairwise IoU against the REFERENCE's own header (its __CUDACC__ branch built with g++, tests/golden/make_golden_iou.py)
    from golden_util import load
    gd, _ = load("box_iou_rotated")
    iou_ref = gpu_ops.box_iou_rotated(gd["a"].cuda(), gd["b"].cuda()).cpu()
    assert (iou_ref - gd["iou"]).abs().max() <= 1e-6 and (iou_ref == gd["iou"]).float().mean() > 0.95
    # and against the oracle at NMS size: the same fp32 operation order on both sides (cos / sin in double as the reference)
    boxes, scores = arkit_like(700, 3, seed=8)
    bev = bev_of(boxes)
    xywhr = torch.stack(((bev[:, 0] + bev[:, 2]) / 2, (bev[:, 1] + bev[:, 3]) / 2, bev[:, 2] - bev[:, 0],
                         bev[:, 3] - bev[:, 1], bev[:, 4]), 1).contiguous()
    iou_c = oracle_ops.box_iou_rotated(xywhr, xywhr)
    iou_g = gpu_ops.box_iou_rotated(xywhr.cuda(), xywhr.cuda()).cpu()
    assert (iou_c > 0.15).sum() > 5000
    assert (iou_g - iou_c).abs().max() < 1e-6
    assert (iou_g != iou_c).float().mean() < 1e-3, (iou_g != iou_c).float().mean()
    for n, n_cls, seed in ((3000, 17, 9), (4096, 2, 10), (65, 3, 11)):
        boxes, scores = arkit_like(n, n_cls, seed)
        bev = bev_of(boxes)
        keep_c, nk_c = oracle_ops.nms_rotated_bev(bev, scores, 0.0, 0.15)
        keep_g, nk_g = gpu_ops.nms_rotated_bev(bev.cuda(), scores.cuda(), 0.0, 0.15)
        assert torch.equal(nk_g.cpu(), nk_c), (n, nk_g.cpu(), nk_c)
        for c in range(n_cls):
            k = int(nk_c[c])
            assert 0 < k < n and torch.equal(keep_g[c, :k].cpu(), keep_c[c, :k]), (n, c)
    # empty inputs, classes without candidates, the size limit
    keep, nk = gpu_ops.nms_rotated_bev(torch.zeros(0, 5).cuda(), torch.zeros(0, 4).cuda(), 0.0, 0.15)
    assert keep.shape == (4, 0) and nk.tolist() == [0, 0, 0, 0]
    boxes, scores = arkit_like(100, 3, seed=12)
    scores[:, 1] = 0.0
    keep, nk = gpu_ops.nms_rotated_bev(bev_of(boxes).cuda(), scores.cuda(), 0.0, 0.15)
    assert nk[1].item() == 0 and nk[0].item() > 0
    with pytest.raises(Exception):
        gpu_ops.nms_rotated_bev(torch.zeros(5000, 5).cuda(), torch.zeros(5000, 2).cuda(), 0.0, 0.15)


def test_target_assignment_matches_reference_golden_and_oracle(oracle_ops, gpu_ops):
    """row f-3: ImVoxelHeadV2.get_targets as one fused pass (sgc_assign_targets)"""
    from targets_contract import check_against_oracle, check_targets_golden
    check_targets_golden(gpu_ops, "cuda")
    check_against_oracle(gpu_ops, oracle_ops, "cuda")
    with pytest.raises(Exception):        # the reference cannot assign without boxes either
        gpu_ops.assign_targets(torch.zeros(30, 3).cuda(), torch.zeros(30, dtype=torch.int32).cuda(), torch.zeros(0, 7).cuda(),
                               torch.zeros(0, dtype=torch.int64).cuda(), False, 3, 27, 18)


def test_upsample_backward_gather_matches_oracle_and_autograd(oracle_ops, gpu_ops):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    for shape in ((3, 4, 5, 3), (2, 1, 6, 2), (1, 7, 1, 1), (256, 20, 20, 8)):
        go = torch.randn(1, shape[0], 2 * shape[1], 2 * shape[2], 2 * shape[3], generator=g)
        got = gpu_ops.upsample2x_backward(go.cuda()).cpu()
        if shape[0] <= 4:
            assert (got - oracle_ops.upsample2x_backward(go)).abs().max() < 1e-5, shape
        x = torch.zeros(1, *shape, device="cuda", requires_grad=True)
        want, = torch.autograd.grad(F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False), x, go.cuda())
        assert (got - want.cpu()).abs().max() < 2e-5, shape
    # the training path routes through it and agrees with torch's own backward
    from sgcdet_amd.plugin.voxel_heads import trilinear_up2x
    x = torch.randn(1, 8, 5, 6, 4, generator=g).cuda().requires_grad_()
    go = torch.randn(1, 8, 10, 12, 8, generator=g).cuda()
    a, = torch.autograd.grad(trilinear_up2x(x), x, go)
    b, = torch.autograd.grad(F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False), x, go)
    assert (a - b).abs().max() < 1e-5


def test_camera_stride_of_channels_last_maps(oracle_ops, gpu_ops):
    from count_contract import check_camera_stride
    check_camera_stride(gpu_ops, oracle_ops, "cuda")


def test_plane_sweep_cost_volume_matches_reference_golden_and_oracle(oracle_ops, gpu_ops):
    import numpy as np
    import os
    from test_oracle_golden import _plane_sweep_case
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "plane_sweep.npz"))
    for k in range(int(d["n_cases"])):
        f_mvs, rows, nbr, rt, depth, (H, W) = _plane_sweep_case(d, k)
        corr = gpu_ops.plane_sweep_corr(rows.cuda(), nbr.cuda(), rt.cuda(), depth.cuda(), H, W)
        want = torch.from_numpy(d[f"corr{k}"])
        assert float((corr.cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max())), k
    # DepthNet shapes (128-channel matching features at 1/4 resolution), a few views, against the oracle
    from sgcdet_amd.scene import make_img_meta
    from sgcdet_amd.plugin.plane_sweep import plane_sweep_correlation, closest_frame_ids, relative_projections
    N, C, H, W = 6, 128, 60, 80
    meta = make_img_meta(N, "scannet", 4)
    g = torch.Generator().manual_seed(4)
    f_mvs = torch.randn(N, C, H, W, generator=g)
    depth = np.arange(0.2, 5.0, 0.4, dtype=np.float32) + 0.2
    got = plane_sweep_correlation(f_mvs.cuda(), meta, 4, depth, neighbor_img_num=2)
    w2c = torch.tensor(np.array(meta["lidar2img"]["extrinsic"]))
    intr = torch.tensor(np.array(meta["lidar2img"]["intrinsic"])).clone()
    intr[:2] /= meta["ori_shape"][0] / (meta["img_shape"][0] / 4)
    nbr = closest_frame_ids(N, 2)
    rt = relative_projections(w2c, intr, nbr).reshape(N, 2, 12).contiguous()
    want = oracle_ops.plane_sweep_corr(f_mvs.permute(0, 2, 3, 1).reshape(N, H * W, C).contiguous(), nbr.to(torch.int32).contiguous(),
                                       rt, torch.from_numpy(depth), H, W)
    assert float((got.cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    assert float((want != 0).float().mean()) > 0.3


def test_full_size_properties_of_the_path_kernels(gpu_ops):
    """Size-independent properties at BASELINE.json's config-2 sizes (the oracle pins the arithmetic at sizes it
    finishes in seconds; these hold at any size): the deformable gather is linear in the value map, the halo
    convolution is linear in its input and commutes with the epilogue scale, both NMS forms are idempotent, the
    target assignment does not depend on the order of the boxes."""
    from sgcdet_amd.scene import make_img_meta
    from sgcdet_amd.plugin.voxformer import compute_projection
    from nms_rotated_contract import arkit_like, bev_of
    from targets_contract import random_boxes
    dev = "cuda"
    g = torch.Generator().manual_seed(31)
    # --- gather: 40 views, 64x80 maps, 6400 voxels -> ~77 k visible pairs ---
    N, C, H, W, D, M, P = 40, 256, 64, 80, 12, 8, 4
    meta = make_img_meta(N, "scannet", 0, img_hw=(256, 320))
    proj = compute_projection(meta).float().to(dev).contiguous()
    origin = torch.tensor(meta["lidar2img"]["origin"]).to(dev)
    idx = torch.randperm(40 * 40 * 16, generator=g)[:6400].sort().values
    xs = torch.stack([idx // (40 * 16), (idx // 16) % 40, idx % 16], 1).float()
    ref3d = (xs * torch.tensor([.16, .16, .2]) - torch.tensor([40, 40, 16]) / 2 * torch.tensor([.16, .16, .2])).to(dev).contiguous()
    ref_cam, mask = gpu_ops.project_points(ref3d, origin, proj, 320, 256, 0.2, 5.0)
    pc = gpu_ops.compact_pairs(mask)
    n_pairs = int(pc["totals"][0])
    assert n_pairs > 50000
    dist = torch.randn(N, H * W, D, generator=g).mul(2).softmax(-1).contiguous().to(dev)
    raw = torch.randn(n_pairs, M * P * 4, generator=g).to(dev)
    v1 = torch.randn(N, H * W, M, C // M, generator=g).to(dev)
    v2 = torch.randn(N, H * W, M, C // M, generator=g).to(dev)
    run = lambda v: gpu_ops.pairs_deform_gather(v, dist, ref_cam, raw, pc["pair_cam"], pc["pair_q"], n_pairs, H, W, M, P)
    o1, o2, o12 = run(v1), run(v2), run((0.5 * v1 - 2.0 * v2).contiguous())
    assert o1.shape == (n_pairs, C) and o1.abs().max() > 0.1
    assert (o12 - (0.5 * o1 - 2.0 * o2)).abs().max() < 2e-5 * max(1.0, o12.abs().max().item())
    # --- halo convolution 256 -> 256 at 40x40x16 (the 90-GF layer) ---
    x1, x2 = torch.randn(25600, 256, generator=g).to(dev), torch.randn(25600, 256, generator=g).to(dev)
    wt = (torch.randn(27, 256, 256, generator=g) * (1.0 / (27 * 256) ** 0.5)).to(dev)
    w_hi, w_lo = gpu_ops.split_bf16(wt)
    sc = (torch.rand(256, generator=g) + 0.5).to(dev)
    conv = lambda x, s=None: gpu_ops.conv3d_cl_bf16x3(x, w_hi, w_lo, (40, 40, 16), 3, 1, False, s, None, None, 0)[0]
    y1, y2, y12 = conv(x1), conv(x2), conv((x1 + 3.0 * x2).contiguous())
    tol = 1e-4 * y12.abs().max().item()
    assert (y12 - (y1 + 3.0 * y2)).abs().max() < tol
    assert (conv(x1, sc) - y1 * sc[None]).abs().max() < tol
    # --- NMS idempotence at the heads' candidate counts ---
    n = 3000
    c = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([6.4, 6.4, 2.5])
    c = c[torch.randint(0, 150, (n,), generator=g)] + torch.randn(n, 3, generator=g) * 0.05
    s = 0.4 + torch.rand(n, 3, generator=g)
    boxes, scores, labels = torch.cat([c - s / 2, c + s / 2], 1).to(dev), torch.rand(n, generator=g).to(dev), torch.randint(0, 18, (n,), generator=g).to(dev)
    keep = gpu_ops.aligned_nms3d(boxes, scores, labels, 0.25)
    again = gpu_ops.aligned_nms3d(boxes[keep].contiguous(), scores[keep].contiguous(), labels[keep].contiguous(), 0.25)
    assert 0 < keep.numel() < n and torch.equal(again, torch.arange(keep.numel(), device=dev))
    rb, rs = arkit_like(3000, 17, seed=33)
    bev = bev_of(rb).to(dev)
    kp, nk = gpu_ops.nms_rotated_bev(bev, rs.to(dev), 0.0, 0.15)
    for cls in (0, 8, 16):
        sel = kp[cls, :int(nk[cls])]
        kp2, nk2 = gpu_ops.nms_rotated_bev(bev[sel].contiguous(), rs.to(dev)[sel][:, cls:cls + 1].contiguous(), 0.0, 0.15)
        assert int(nk2[0]) == sel.numel() and torch.equal(kp2[0, :sel.numel()], torch.arange(sel.numel(), device=dev))
    # --- target assignment: permuting the boxes permutes nothing but the indices ---
    import numpy as np, os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "head_targets.npz"))
    pts, scl = torch.from_numpy(d["points"]).to(dev).contiguous(), torch.from_numpy(d["scales"]).to(dev).contiguous()
    tb, tl = random_boxes(40, 7, False)
    tb[:, 3:6] += torch.arange(40)[:, None] * 1e-3          # distinct volumes: the minimal-volume rule has no ties
    perm = torch.randperm(40, generator=g)
    a = gpu_ops.assign_targets(pts, scl, tb.to(dev), tl.to(dev), False, 3, 27, 18)
    b = gpu_ops.assign_targets(pts, scl, tb[perm].contiguous().to(dev), tl[perm].contiguous().to(dev), False, 3, 27, 18)
    pos = a[2] >= 0
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[1][pos], b[1][pos]) and torch.equal(a[0][pos], b[0][pos])


@pytest.mark.parametrize("C,HW,bins,halo,scale", [
    (256, (15, 20), (7, 5), (2, 2), 3.0),        # small window, offsets of several pixels: the global fix-up pass runs
    (256, (30, 40), (13, 10), (4, 3), 1.0),      # hot shape family (Cm = 32)
    (128, (14, 20), (20, 14), (0, 0), 2.0),      # Cm = 16, one bin per camera (window = whole map)
    (128, (29, 40), (8, 29), (1, 0), 6.0),       # Cm = 16, column bins, large offsets
])
def test_tiled_gather_against_oracle(C, HW, bins, halo, scale, oracle_ops, gpu_ops):
    """sgc_bin_pairs (bit-exact) and sgc_pairs_deform_gather_tiled (1e-5) against the oracle; results must not depend
    on the bin size, the halo, the per-head window shift, the workgroup size, the number of value buffers, the head
    grouping or where the depth taps are read from."""
    from tests.tile_contract import check_bins, raw_to_headmajor, value_to_headmajor
    N, Nq, D, M, P = 6, 900, 12, 8, 4
    H, W = HW
    ref3d, origin, proj = _scene(N, Nq, 4)
    rc, mk = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    pc = oracle_ops.compact_pairs(mk)
    n_pairs, cap = int(pc["totals"][0]), pc["pair_q"].numel()
    g = torch.Generator().manual_seed(21)
    value = torch.randn(N, H * W, M, C // M, generator=g)
    dist = torch.randn(N, H * W, D, generator=g).mul(2).softmax(-1).contiguous()
    raw = torch.randn(cap, M * P * 4, generator=g)
    raw[:, :M * P * 3] *= scale
    want = oracle_ops.pairs_deform_gather(value, dist, rc, raw, pc["pair_cam"], pc["pair_q"], n_pairs, H, W, M, P)
    cu = lambda t: t.cuda()
    gpc = {k: cu(v) for k, v in pc.items()}
    bw, bh = bins
    before = dict(pc, slot=pc["slot"].clone())
    b_c = oracle_ops.bin_pairs(rc, dict(pc, slot=pc["slot"].clone()), H, W, bw, bh)
    b_g = gpu_ops.bin_pairs(cu(rc), dict(gpc, slot=gpc["slot"].clone()), H, W, bw, bh)
    for k in ("bin_offset", "slot"):
        assert torch.equal(b_g[k].cpu(), b_c[k]), k
    assert torch.equal(b_g["pair_q"][:n_pairs].cpu(), b_c["pair_q"][:n_pairs])
    assert torch.equal(b_g["pair_ref"][:n_pairs].cpu().view(torch.int32), b_c["pair_ref"][:n_pairs].view(torch.int32))
    old = check_bins(b_g, before, rc, n_pairs, H, W, bw, bh)
    raw_new = torch.zeros_like(raw)
    raw_new[:n_pairs] = raw[old]
    vhm, rhm = cu(value_to_headmajor(value)), cu(raw_to_headmajor(raw_new, M, P))
    shift = torch.randint(-3, 4, (M, 2), generator=g, dtype=torch.int32)
    knobs = ("tile_nw", "tile_depth_lds", "tile_nbuf", "tile_hg")
    try:
        for nw, dl, nbuf, hg, hs in [(16, 1, 2, 0, None), (8, 1, 1, 2, shift), (16, 0, 2, 4, shift), (8, 0, 1, 1, None),
                                     (16, 1, 2, 8, shift)]:
            for key, val in zip(knobs, (nw, dl, nbuf, hg)):
                gpu_ops.lib.call("sgc_set_tuning", key.encode(), val)
            got = gpu_ops.pairs_deform_gather_tiled(vhm, cu(dist), b_g["pair_ref"], b_g["bin_offset"], rhm, H, W, P, bw, bh,
                                                    halo[0], halo[1], head_shift=None if hs is None else cu(hs),
                                                    max_shift=(3, 3))
            close(got[:n_pairs], want[old])
    finally:
        for key, val in zip(knobs, (0, -1, 0, 0)):
            gpu_ops.lib.call("sgc_set_tuning", key.encode(), val)
    # a different binning of the same pairs: same operator
    bw2, bh2 = max(1, bw // 2), bh + 3
    b2 = gpu_ops.bin_pairs(cu(rc), dict(gpc, slot=gpc["slot"].clone()), H, W, bw2, bh2)
    old2 = check_bins(b2, before, rc, n_pairs, H, W, bw2, bh2)
    raw2 = torch.zeros_like(raw)
    raw2[:n_pairs] = raw[old2]
    got2 = gpu_ops.pairs_deform_gather_tiled(vhm, cu(dist), b2["pair_ref"], b2["bin_offset"], cu(raw_to_headmajor(raw2, M, P)),
                                             H, W, P, bw2, bh2, 1, 1)
    close(got2[:n_pairs], want[old2])


def test_headmajor_value_projection_is_the_row_gemm_permuted(oracle_ops, gpu_ops):
    """sgc_linear_rows_headmajor_bf16x3 == sgc_linear_rows_bf16x3 with the store address permuted: bit-identical
    elements; and within the bf16x3 bound (1e-4 of the scale) of the fp32 oracle."""
    N, S, Cin, M = 3, 333, 64, 8
    g = torch.Generator().manual_seed(3)
    for Cm in (32, 16):
        x = torch.randn(N * S, Cin, generator=g)
        w = torch.randn(M * Cm, Cin, generator=g) * 0.2
        b = torch.randn(M * Cm, generator=g)
        hi, lo = gpu_ops.split_bf16(w.view(1, M * Cm, Cin))
        y_rows = gpu_ops.linear_rows_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), b.cuda())
        y_hm = gpu_ops.linear_rows_headmajor_bf16x3(x.cuda(), hi.cuda(), lo.cuda(), b.cuda(), N, S, M)
        assert torch.equal(y_hm, y_rows.view(N, S, M, Cm).permute(0, 2, 1, 3).contiguous())
        y_c = oracle_ops.linear_rows_headmajor_bf16x3(x, hi, lo, b, N, S, M)
        close(y_hm, y_c, tol=1e-4)


@pytest.mark.parametrize("n,k", [(3200, 800), (25600, 6400), (294912, 73728), (1000, 1000), (777, 1)])
def test_topk_select_matches_oracle_with_the_defined_tie_rule(n, k, oracle_ops, gpu_ops):
    """sgc_topk_select: same voxel set as torch.topk on tie-free scores; with exact ties at the cut (blocks of voxels no
    camera sees carry bit-identical occupancy) the lowest flat indices win on both sides: indices, valid and mask
    bit-exact against the oracle, sizes up to config 5's finest level."""
    g = torch.Generator().manual_seed(n + k)
    s = torch.sigmoid(torch.randn(n, generator=g))
    idx_g, valid_g, mask_g = gpu_ops.topk_select(s.cuda(), k, want_valid=True, want_mask=True)
    idx_c, valid_c, mask_c = oracle_ops.topk_select(s, k, want_valid=True, want_mask=True)
    assert torch.equal(idx_g.cpu(), idx_c) and torch.equal(valid_g.cpu(), valid_c) and torch.equal(mask_g.cpu(), mask_c)
    if k < n:
        thr = s.sort(descending=True).values
        if thr[k - 1] > thr[k]:                                    # no tie at the cut: torch.topk picks the same set
            assert torch.equal(idx_c, torch.topk(s, k).indices.sort().values)
    # heavy exact ties: 40 % of the scores share the value that ends up at the cut
    t = s.clone()
    tie_val = float(s.sort(descending=True).values[min(k, n - 1)])
    t[torch.randperm(n, generator=g)[: int(0.4 * n)]] = tie_val
    idx_g, valid_g, _ = gpu_ops.topk_select(t.cuda(), k, want_valid=True)
    idx_c, valid_c, _ = oracle_ops.topk_select(t, k, want_valid=True)
    assert torch.equal(idx_g.cpu(), idx_c) and torch.equal(valid_g.cpu(), valid_c)
    assert int(valid_c.sum()) == k and bool((idx_c[1:] > idx_c[:-1]).all())
    sel, rest = t[valid_c.bool()], t[~valid_c.bool()]
    assert rest.numel() == 0 or float(sel.min()) >= float(rest.max())
    eq = (t == sel.min()).nonzero().flatten()                        # among the tied scores the lowest indices are taken
    taken = valid_c[eq].bool()
    assert bool(taken[: int(taken.sum())].all())


@pytest.mark.parametrize("rows,C", [(6400, 256), (51200, 128), (333, 64), (100, 96)])
def test_layer_norm_rows_matches_torch_and_oracle(rows, C, oracle_ops, gpu_ops):
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 3 + 0.7
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    want = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5)
    got = gpu_ops.layer_norm_rows(x.cuda(), w.cuda(), b.cuda(), 1e-5)
    close(got, want, tol=2e-6)
    close(oracle_ops.layer_norm_rows(x, w, b, 1e-5), want, tol=2e-6)
    cnt = torch.tensor([rows // 2], dtype=torch.int32)
    part = gpu_ops.layer_norm_rows(x.cuda(), w.cuda(), b.cuda(), 1e-5, count=cnt.cuda(), out=torch.full((rows, C), 7.0).cuda())
    close(part[: rows // 2], want[: rows // 2], tol=2e-6)
    assert bool((part[rows // 2:] == 7.0).all())                     # rows past the device-side count are untouched


def test_projection_with_a_query_selection_equals_gather_then_project(oracle_ops, gpu_ops):
    N, Nvox = 7, 5000
    ref3d, origin, proj = _scene(N, Nvox, 9)
    sel = torch.randperm(Nvox, generator=torch.Generator().manual_seed(1))[:1234].sort().values
    rc_a, mk_a = gpu_ops.project_points(ref3d[sel].contiguous().cuda(), origin.cuda(), proj.cuda(), 320., 239., 0.2, 5.0)
    rc_b, mk_b = gpu_ops.project_points(ref3d.cuda(), origin.cuda(), proj.cuda(), 320., 239., 0.2, 5.0, sel=sel.cuda())
    assert torch.equal(rc_a, rc_b) and torch.equal(mk_a, mk_b)
    rc_c, mk_c = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0, sel=sel)
    assert torch.equal(rc_b.cpu(), rc_c) and torch.equal(mk_b.cpu(), mk_c)


@pytest.mark.parametrize("C,HW,bins", [(256, (30, 40), (13, 10)), (128, (29, 40), (20, 15))])
def test_bf16_storage_mode_of_the_tiled_gather(C, HW, bins, oracle_ops, gpu_ops):
    """Opt-in bf16 STORAGE mode (row N2; BASELINE.json configs #2 / #5): the head-major value map AND the depth distributions
    are bfloat16, taps are widened to fp32, accumulation and outputs stay fp32.  (1) Against the oracle fed the SAME bf16 map: 1e-5 (it is the
    same arithmetic).  (2) Against the fp32 oracle: every value carries one bf16 rounding (relative 2^-9), the output is a
    convex-ish combination of <= 16 taps with weights summing to <= 1, so |err| <= 2^-9 * max|value|; measured ~1e-3 of
    the value scale -- asserted at 4e-3 (2^-8)."""
    from tests.tile_contract import check_bins, raw_to_headmajor, value_to_headmajor
    N, Nq, D, M, P = 5, 800, 12, 8, 4
    H, W = HW
    ref3d, origin, proj = _scene(N, Nq, 6)
    rc, mk = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    pc = oracle_ops.compact_pairs(mk)
    n_pairs, cap = int(pc["totals"][0]), pc["pair_q"].numel()
    g = torch.Generator().manual_seed(33)
    value = torch.randn(N, H * W, M, C // M, generator=g)
    dist = torch.randn(N, H * W, D, generator=g).mul(2).softmax(-1).contiguous()
    raw = torch.randn(cap, M * P * 4, generator=g)
    raw[:, :M * P * 3] *= 1.5
    bw, bh = bins
    # a bf16 depth pixel is D * 2 = 24 bytes, so the depth window of a bin whose origin column xd0 = bx * bw - 3
    # is ODD starts 8 bytes (not 16) into a line: the LDS-DMA source of the `depth_in_lds` path is then only 8-byte aligned.
    # Both shapes have bin columns with an odd origin -- this test is the coverage of that case.
    assert (D * 2) % 16 == 8 and any((bx * bw - 3) % 2 == 1 for bx in range(1, -(-W // bw)))
    before = dict(pc, slot=pc["slot"].clone())
    b_c = oracle_ops.bin_pairs(rc, dict(pc, slot=pc["slot"].clone()), H, W, bw, bh)
    old = check_bins(b_c, before, rc, n_pairs, H, W, bw, bh)
    raw_new = torch.zeros_like(raw)
    raw_new[:n_pairs] = raw[old]
    rhm = raw_to_headmajor(raw_new, M, P)
    vhm = value_to_headmajor(value)
    vhm16 = vhm.to(torch.bfloat16)
    want_f32 = oracle_ops.pairs_deform_gather_tiled(vhm, dist, b_c["pair_ref"], b_c["bin_offset"], rhm, H, W, P, bw, bh, 3, 3)
    dist16 = dist.to(torch.bfloat16)                    # ABI 4: the storage mode covers the depth distributions too
    want_b16 = oracle_ops.pairs_deform_gather_tiled(vhm16, dist16, b_c["pair_ref"], b_c["bin_offset"], rhm, H, W, P, bw, bh, 3, 3)
    cu = lambda t: t.cuda()
    for dl in (True, False):                            # depth taps from the LDS window / from global memory
        got = gpu_ops.pairs_deform_gather_tiled(cu(vhm16), cu(dist16), cu(b_c["pair_ref"]), cu(b_c["bin_offset"]), cu(rhm), H, W, P,
                                                bw, bh, 3, 3, depth_in_lds=dl)
        close(got[:n_pairs], want_b16[:n_pairs])
    scale = float(value.abs().max())
    err = float((got[:n_pairs].cpu() - want_f32[:n_pairs]).abs().max())
    # value taps carry one bf16 rounding (2^-9), the depth scores (probabilities <= 1, a convex combination of 8 taps) another
    assert err <= 2.0 ** -7 * scale, (err, scale)
    with pytest.raises(RuntimeError):                   # mixed storage is refused
        gpu_ops.pairs_deform_gather_tiled(cu(vhm16), cu(dist), cu(b_c["pair_ref"]), cu(b_c["bin_offset"]), cu(rhm), H, W, P, bw, bh, 3, 3)
    # the producer: value_proj's epilogue writing bf16 == RNE of the fp32 head-major result
    x = torch.randn(N * H * W, 64, generator=g)
    w = torch.randn(C, 64, generator=g) * 0.2
    hi, lo = gpu_ops.split_bf16(w.view(1, C, 64))
    y32 = gpu_ops.linear_rows_headmajor_bf16x3(cu(x), cu(hi), cu(lo), None, N, H * W, M)
    y16 = gpu_ops.linear_rows_headmajor_bf16x3(cu(x), cu(hi), cu(lo), None, N, H * W, M, out_dtype=torch.bfloat16)
    assert torch.equal(y16, y32.to(torch.bfloat16))


@pytest.mark.parametrize("C", [128, 256])
@pytest.mark.parametrize("HW,stride_extra", [((29, 40), 0), ((15, 20), 0), ((14, 20), 20)])
def test_geometry_sample_fused_with_its_linear(HW, stride_extra, C, oracle_ops, gpu_ops):
    """sgc_pairs_geometry_linear_bf16x3 (round 6): the geometry-aware sample of every visible pair built while the row GEMM stages
    its tile == sgc_pairs_geometry_sample + sgc_linear_rows_bf16x3, BIT FOR BIT (same fmas, same GEMM kernel), host-counted and
    device-counted, with the cropped-row camera stride; and within the bf16x3 bound of the oracle's two steps.  C = 128: the gather
    form of the persistent row GEMM; C = 256: its producer / consumer form (rows_gemm_gather_pc_kernel)."""
    N, Nq, D, Cout = 5, 800, 12, 128
    H, W = HW
    S = H * W + stride_extra
    ref3d, origin, proj = _scene(N, Nq, 13)
    rc, mk = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    pc = oracle_ops.compact_pairs(mk)
    n = int(pc["totals"][0])
    g = torch.Generator().manual_seed(31)
    feat = torch.randn(N, S, C, generator=g)
    dist = torch.randn(N, S, D, generator=g).mul(2).softmax(-1).contiguous()
    w = torch.randn(1, Cout, C, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    hi, lo = gpu_ops.split_bf16(w)
    cu = lambda t: t.cuda()
    gpc = {k: cu(v) for k, v in pc.items()}
    assert gpu_ops.pairs_geometry_linear_supported(C, Cout, N, S)
    geo = gpu_ops.pairs_geometry_sample(cu(feat), cu(dist), cu(rc), gpc["pair_cam"], gpc["pair_q"], n, H, W)
    two = gpu_ops.linear_rows_bf16x3(geo, cu(hi), cu(lo), cu(b))
    one = gpu_ops.pairs_geometry_linear(cu(feat), cu(dist), cu(rc), gpc["pair_cam"], gpc["pair_q"], n, H, W, cu(hi), cu(lo), cu(b))
    assert one.shape == two.shape and torch.equal(one, two)
    dev = gpu_ops.pairs_geometry_linear(cu(feat), cu(dist), cu(rc), gpc["pair_cam"], gpc["pair_q"], -1, H, W, cu(hi), cu(lo), cu(b),
                                        totals=gpc["totals"])
    assert torch.equal(dev[:n], two)
    want = oracle_ops.pairs_geometry_linear(feat, dist, rc, pc["pair_cam"], pc["pair_q"], n, H, W, hi, lo, b)
    close(one, want, tol=1e-4)
    assert not gpu_ops.pairs_geometry_linear_supported(C, 256, N, S)        # only the 128-column projection (M * P * 4)


def test_item_list_operator_and_backward_against_oracle(oracle_ops, gpu_ops):
    """sgc_dfa3d_forward_items / sgc_dfa3d_backward_items (training path on pair lists) vs the oracle."""
    B, S_hw, M, Cm, D, L, P, n = 4, (9, 11), 8, 8, 6, 1, 4, 333
    g = torch.Generator().manual_seed(77)
    S = S_hw[0] * S_hw[1]
    value = torch.randn(B, S, M, Cm, generator=g)
    dist = torch.randn(B, S, 1, D, generator=g).mul(2).softmax(-1).contiguous()
    shapes3 = torch.tensor([[S_hw[0], S_hw[1], D]], dtype=torch.int64)
    lsi = torch.zeros(1, dtype=torch.int64)
    loc = (torch.rand(n, M, L, P, 3, generator=g) * 1.3 - 0.15).contiguous()
    attn = torch.rand(n, M, L, P, generator=g)
    item = torch.randint(0, B, (n,), generator=g, dtype=torch.int32).sort().values.contiguous()
    go = torch.randn(n, M * Cm, generator=g)
    out_c = oracle_ops.dfa3d_forward_items(value, dist, shapes3, lsi, loc, attn, item)
    cu = lambda t: t.cuda()
    out_g = gpu_ops.dfa3d_forward_items(cu(value), cu(dist), cu(shapes3), cu(lsi), cu(loc), cu(attn), cu(item))
    close(out_g, out_c)
    gc = oracle_ops.dfa3d_backward_items(value, dist, shapes3, lsi, loc, attn, item, go)
    gg = gpu_ops.dfa3d_backward_items(cu(value), cu(dist), cu(shapes3), cu(lsi), cu(loc), cu(attn), cu(item), cu(go))
    for a, b in zip(gg, gc):
        close(a, b, tol=2e-5)


@pytest.mark.parametrize("Cm,HW,bins,halo,spread", [
    (32, (29, 40), (16, 22), (3, 3), 0.08),     # the training default: bins of the forward, window 22 x 28
    (32, (15, 20), (7, 5), (1, 1), 0.25),       # small windows, far offsets: many corners take the global fall-back
    (16, (30, 40), (13, 10), (2, 2), 0.10),     # Cm = 16 (the C = 128 configs)
    (16, (14, 20), (20, 14), (0, 0), 0.30),     # one bin per camera: the window is the whole map
])
def test_binned_backward_against_oracle(Cm, HW, bins, halo, spread, oracle_ops, gpu_ops):
    """sgc_dfa3d_backward_binned (LDS-tiled backward of the training path, round 6) against the oracle's item backward on the same
    items: grad_value, grad_dist, every grad_loc entry and grad_attn within 2e-5 of the scale (the bound of the item kernel's test);
    the result must not depend on the bins or the halo; the shared-sample form (one head over C channels run
    as channel groups: the geometry sample) against the oracle's one-head operator; and the HIP item kernel as a second opinion."""
    N, Nq, D, M, P = 5, 700, 12, 8, 4
    H, W = HW
    ref3d, origin, proj = _scene(N, Nq, 9)
    rc, mk = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    pc = oracle_ops.compact_pairs(mk)
    n = int(pc["totals"][0])
    assert n > 300
    g = torch.Generator().manual_seed(5 + Cm)
    S = H * W
    value = torch.randn(N, S, M, Cm, generator=g)
    dist = torch.randn(N, S, 1, D, generator=g).mul(2).softmax(-1).contiguous()
    shapes3 = torch.tensor([[H, W, D]], dtype=torch.int64)
    lsi = torch.zeros(1, dtype=torch.int64)
    cu = lambda t: t.cuda()
    gpc = {k: cu(v) for k, v in pc.items()}
    for bw, bh, hx, hy in [(bins[0], bins[1], halo[0], halo[1]), (max(1, bins[0] // 2), bins[1] + 3, 1, 2)]:
        b = gpu_ops.bin_pairs(cu(rc), dict(gpc, slot=gpc["slot"].clone()), H, W, bw, bh)
        cam, q = b["pair_cam"][:n].cpu().long(), b["pair_q"][:n].cpu().long()
        ref = rc[cam, q]                                                       # the items' reference points, binned order
        loc = (ref.view(n, 1, 1, 1, 3) + (torch.rand(n, M, 1, P, 3, generator=g) - 0.5) * spread).contiguous()
        attn = torch.rand(n, M, 1, P, generator=g)
        go = torch.randn(n, M * Cm, generator=g)
        want = oracle_ops.dfa3d_backward_items(value, dist, shapes3, lsi, loc, attn, cam.to(torch.int32), go)
        got = gpu_ops.dfa3d_backward_binned(cu(value), cu(dist), cu(loc), cu(attn), b["bin_offset"], cu(go), H, W, bw, bh, (hx, hy))
        for a, w_ in zip(got, want):
            close(a, w_, tol=2e-5)
        # a per-head window shift moves the windows, never the result
        shift = torch.randint(-4, 5, (M, 2), generator=g, dtype=torch.int32)
        got_s = gpu_ops.dfa3d_backward_binned(cu(value), cu(dist), cu(loc), cu(attn), b["bin_offset"], cu(go), H, W, bw, bh, (hx, hy),
                                              head_shift=cu(shift))
        for a, w_ in zip(got_s, want):
            close(a, w_, tol=2e-5)
        item_k = gpu_ops.dfa3d_backward_items(cu(value), cu(dist), cu(shapes3), cu(lsi), cu(loc), cu(attn), cu(cam.to(torch.int32)), cu(go))
        for a, w_ in zip(got, item_k):
            close(a, w_, tol=2e-5)
        # the oracle's own binned twin walks the bins: same numbers as its item form
        twin = oracle_ops.dfa3d_backward_binned(value, dist, loc, attn, b["bin_offset"].cpu(), go, H, W, bw, bh, (hx, hy))
        for a, w_ in zip(twin, want):
            close(a, w_, tol=1e-6)
    # one sample set shared by the M channel groups, no attention weights, P = 1: the geometry sample's single head over C = M * Cm
    loc1 = (ref.view(n, 1, 1, 1, 3) + (torch.rand(n, 1, 1, 1, 3, generator=g) - 0.5) * spread).contiguous()
    go = torch.randn(n, M * Cm, generator=g)
    want1 = oracle_ops.dfa3d_backward_items(value.view(N, S, 1, M * Cm), dist, shapes3, lsi, loc1, torch.ones(n, 1, 1, 1), cam.to(torch.int32), go)
    got1 = gpu_ops.dfa3d_backward_binned(cu(value), cu(dist), cu(loc1), None, b["bin_offset"], cu(go), H, W, bw, bh, (hx, hy))
    close(got1[0].view(N, S, 1, M * Cm), want1[0], tol=2e-5)
    for a, w_ in zip(got1[1:], want1[1:]):
        close(a, w_, tol=2e-5)
    # gradients nobody asked for are not written
    gv, gd, gl, ga = gpu_ops.dfa3d_backward_binned(cu(value), cu(dist), cu(loc1), None, b["bin_offset"], cu(go), H, W, bw, bh, (hx, hy),
                                                   want_grad_loc=False, want_grad_attn=False)
    assert gl is None and ga is None
    close(gv.view(N, S, 1, M * Cm), want1[0], tol=2e-5)
    close(gd, want1[1], tol=2e-5)


def test_training_level_takes_the_binned_backward_and_keeps_its_gradients(gpu_ops):
    """DeformCrossAttention_DFA3D in training mode: with the LDS-tiled backward (pairs binned, default) and with the item kernel
    (SGC_TRAIN_BWD=0 / TRAIN_BWD_TILED disabled, pairs in ascending-voxel order) the loss and every parameter / input gradient agree to
    float-atomic noise -- the pair ORDER and the backward kernel are scheduling choices, not part of the function."""
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.plugin import voxformer
    from sgcdet_amd.mmcv_lite import build_attention
    torch.manual_seed(3)
    C, N, Nq, D = 256, 6, 500, 12
    H, W = 29, 40
    att = build_attention(dict(type="DeformCrossAttention_DFA3D", embed_dims=C, inter_view_aggregation="attn", dropout=0.0,
                               deformable_attention=dict(type="MSDeformableAttention3D_DFA3D", embed_dims=C, num_heads=8, num_points=4,
                                                         num_levels=1))).cuda().train()
    with torch.no_grad():
        for prm in att.parameters():
            prm.add_(torch.randn_like(prm) * 0.02)
    ref3d, origin, proj = _scene(N, Nq, 11)
    rc, mk = gpu_ops.project_points(ref3d.cuda(), origin.cuda(), proj.cuda(), 320., 239., 0.2, 5.0)
    g = torch.Generator().manual_seed(8)
    feat0 = torch.randn(N, H * W, 1, C, generator=g).cuda()
    dist0 = torch.randn(N, H * W, 1, D, generator=g).mul(2).softmax(-1).cuda()
    query0 = torch.randn(1, Nq, C, generator=g).cuda()
    shapes = torch.tensor([[H, W]], dtype=torch.int64).cuda()
    lsi = torch.zeros(1, dtype=torch.int64).cuda()
    results = []
    for enabled in (True, False):
        old = dict(voxformer.TRAIN_BWD_TILED)
        voxformer.TRAIN_BWD_TILED["enabled"] = enabled
        try:
            feat, dist, query = feat0.clone().requires_grad_(), dist0.clone().requires_grad_(), query0.clone().requires_grad_()
            att.zero_grad()
            out = att(query, feat, feat, reference_points_cam=rc.view(N, 1, Nq, 1, 3), bev_mask=mk.view(N, 1, Nq, 1), spatial_shapes=shapes,
                      level_start_index=lsi, value_dpt_dist=dist, spatial_hw=(H, W))
            loss = (out * torch.linspace(-1, 1, C, device="cuda")).sum()
            loss.backward()
            results.append((loss.detach(), feat.grad, dist.grad, query.grad, {k: p.grad.clone() for k, p in att.named_parameters()}))
        finally:
            voxformer.TRAIN_BWD_TILED.update(old)
    (l1, f1, d1, q1, p1), (l0, f0, d0, q0, p0) = results
    close(l1, l0, tol=1e-5)
    close(f1, f0, tol=5e-5); close(d1, d0, tol=5e-5); close(q1, q0, tol=5e-5)
    for k in p0:
        close(p1[k], p0[k], tol=1e-4)


@pytest.mark.parametrize("n,k", [(204800, 51200), (294912, 73728), (25600, 6400), (16385, 1), (20000, 20000), (40961, 40960)])
def test_topk_many_workgroup_form_equals_the_one_workgroup_form(n, k, gpu_ops):
    """sgc_topk_select_ws (histogram / count / compaction launches over 4096-candidate chunks) against the one-workgroup
    kernel: indices, valid and mask bit-exact -- random scores, heavy exact ties at the cut, all-equal scores, NaNs and
    infinities, chunk-boundary sizes; repeated to catch an ordering problem between the launches."""
    g = torch.Generator().manual_seed(n + k)
    s = torch.sigmoid(torch.randn(n, generator=g))
    tied = s.clone()
    tied[torch.randperm(n, generator=g)[: int(0.4 * n)]] = float(s.sort(descending=True).values[min(k, n - 1)])
    special = torch.randn(n, generator=g)
    special[::97] = float("nan"); special[5::131] = float("inf"); special[7::113] = float("-inf"); special[3::89] = 0.0; special[4::89] = -0.0
    for scores in (s, tied, torch.full((n,), 0.25), special):
        x = scores.cuda()
        try:
            gpu_ops.lib.call("sgc_set_tuning", b"topk_multi_min", 1 << 30)         # one workgroup
            ref = gpu_ops.topk_select(x, k, want_valid=True, want_mask=True)
            gpu_ops.lib.call("sgc_set_tuning", b"topk_multi_min", 1)               # many workgroups
            for _ in range(3):
                got = gpu_ops.topk_select(x, k, want_valid=True, want_mask=True)
                for a, b in zip(got, ref):
                    assert torch.equal(a, b)
        finally:
            gpu_ops.lib.call("sgc_set_tuning", b"topk_multi_min", 32769)


def test_view_attend_backward_matches_oracle_and_multihead_attention(oracle_ops, gpu_ops):
    """sgc_view_attend_backward / ViewAttendFunction: against the double-accumulating oracle, and against autograd through
    nn.MultiheadAttention on the dense [N, L, C] slots with the key-padding mask (the reference's formulation,
    TU/deformable_cross_attention.py:829-833) including the in-/out-projection weights."""
    from sgcdet_amd.functions import ViewAttendFunction
    g = torch.Generator().manual_seed(3)
    N, Nq, C, heads = 6, 50, 64, 8
    mask = torch.rand(N, Nq, generator=g) < 0.45
    mask[:, 7] = False                                                     # a voxel no camera sees
    mask[2, 9] = True
    cam, qi = mask.nonzero(as_tuple=True)
    n_pairs = cam.shape[0]
    slot = torch.full((N, Nq), -1, dtype=torch.int32)
    slot[cam, qi] = torch.arange(n_pairs, dtype=torch.int32)
    valid_index = mask.sum(0).nonzero()[:, 0]
    n_valid = valid_index.shape[0]
    q = torch.randn(n_valid, C, generator=g)
    kv = torch.randn(n_pairs, 2 * C, generator=g)
    gout = torch.randn(n_valid, C, generator=g)
    ctx_c = oracle_ops.view_attend(q, kv, slot, valid_index.int(), heads)
    gq_c, gkv_c = oracle_ops.view_attend_backward(q, kv, slot, valid_index.int(), heads, ctx_c, gout)
    qg, kvg = q.cuda().requires_grad_(True), kv.cuda().requires_grad_(True)
    ctx_g = ViewAttendFunction.apply(qg, kvg, slot.cuda(), valid_index.int().cuda(), heads)
    ctx_g.backward(gout.cuda())
    assert max_abs(ctx_g.detach().cpu() - ctx_c) < 1e-5
    assert max_abs(qg.grad.cpu() - gq_c) < 2e-5 * max(1.0, float(gq_c.abs().max()))
    assert max_abs(kvg.grad.cpu() - gkv_c) < 2e-5 * max(1.0, float(gkv_c.abs().max()))
    # the reference's formulation: MHA over dense slots, identity projections folded out by feeding projected tensors
    mha = torch.nn.MultiheadAttention(C, heads).double()
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.eye(C).repeat(3, 1)); mha.in_proj_bias.zero_()
        mha.out_proj.weight.copy_(torch.eye(C)); mha.out_proj.bias.zero_()
    qd, kvd = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    k_slots = torch.zeros(N, Nq, C, dtype=torch.float64).index_put((cam, qi), kvd[:, :C])[:, valid_index]
    v_slots = torch.zeros(N, Nq, C, dtype=torch.float64).index_put((cam, qi), kvd[:, C:])[:, valid_index]
    out, _ = mha(qd[None], k_slots, v_slots, key_padding_mask=~mask[:, valid_index].t())
    out[0].backward(gout.double())
    assert max_abs(ctx_g.detach().cpu().double() - out[0].detach()) < 1e-5
    assert max_abs(qg.grad.cpu().double() - qd.grad) < 2e-5 * max(1.0, float(qd.grad.abs().max()))
    assert max_abs(kvg.grad.cpu().double() - kvd.grad) < 2e-5 * max(1.0, float(kvd.grad.abs().max()))


@pytest.mark.parametrize("rows,cin,cout", [(1, 256, 256), (33, 256, 128), (3200, 256, 512), (6401, 256, 256), (77, 128, 128),
                                           (5000, 128, 256), (20001, 256, 128)])
def test_persistent_row_gemm_against_oracle_and_the_tile_kernel(rows, cin, cout, oracle_ops, gpu_ops):
    """csrc/rows_gemm.hip (persistent, weights in registers) takes every K in {128, 256}, N % 128 == 0 row GEMM:
    fp32 oracle within the bf16x3 bound (1e-4 of the scale), bit-identical to the tile-per-workgroup kernel it replaces
    (same k order, same product order), a row's bits independent of the row count (host count == device count ==
    a prefix of a longer call), rows past the device count untouched.  Reference work: every nn.Linear of a level,
    TU/deformable_cross_attention.py:417-436,826-833."""
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    hi, lo = gpu_ops.split_bf16(w.view(1, cout, cin))
    xc, hc, lc, bc = x.cuda(), hi.cuda(), lo.cuda(), b.cuda()
    y = gpu_ops.linear_rows_bf16x3(xc, hc, lc, bc)
    close(y, oracle_ops.linear_rows_bf16x3(x, hi, lo, b), tol=1e-4)
    try:
        gpu_ops.lib.call("sgc_set_tuning", b"rows_gemm", 0)
        y_tile = gpu_ops.linear_rows_bf16x3(xc, hc, lc, bc)
    finally:
        gpu_ops.lib.call("sgc_set_tuning", b"rows_gemm", 1)
    assert torch.equal(y, y_tile)
    for form in (0, 2, 1):                    # staggered / two tiles ahead / the default: variants of one arithmetic
        gpu_ops.lib.call("sgc_set_tuning", b"rows_depth", form)
        assert torch.equal(gpu_ops.linear_rows_bf16x3(xc, hc, lc, bc), y), form
    cnt = max(1, (rows * 2) // 3)
    out = torch.full((rows, cout), 7.0, device="cuda")
    gpu_ops.linear_rows_bf16x3(xc, hc, lc, bc, count=torch.tensor([cnt], dtype=torch.int32, device="cuda"), out=out)
    assert torch.equal(out[:cnt], y[:cnt]) and bool((out[cnt:] == 7.0).all())
    assert torch.equal(gpu_ops.linear_rows_bf16x3(xc[:cnt].contiguous(), hc, lc, bc), y[:cnt])


@pytest.mark.parametrize("N,S,cin,M", [(3, 47, 256, 8), (5, 333, 128, 8), (2, 32, 256, 8), (4, 1280, 256, 8)])
def test_persistent_row_gemm_head_major_and_epilogues(N, S, cin, M, oracle_ops, gpu_ops):
    """Head-major store of the persistent kernel (tiles that straddle camera borders, fp32 and bf16 storage) and its
    scale / shift / relu / residual epilogues (the FFN's two launches, mmcv FFN as used by TU/encoder.py:311-338)
    against the oracle and, bit for bit, against the tile kernel."""
    g = torch.Generator().manual_seed(S)
    cout = cin
    x = torch.randn(N * S, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    hi, lo = gpu_ops.split_bf16(w.view(1, cout, cin))
    xc, hc, lc, bc = x.cuda(), hi.cuda(), lo.cuda(), b.cuda()
    rows = gpu_ops.linear_rows_bf16x3(xc, hc, lc, bc)
    for dt in (torch.float32, torch.bfloat16):
        y = gpu_ops.linear_rows_headmajor_bf16x3(xc, hc, lc, bc, N, S, M, out_dtype=dt)
        assert torch.equal(y, rows.view(N, S, M, cout // M).permute(0, 2, 1, 3).contiguous().to(dt))
    close(gpu_ops.linear_rows_headmajor_bf16x3(xc, hc, lc, bc, N, S, M), oracle_ops.linear_rows_headmajor_bf16x3(x, hi, lo, b, N, S, M), tol=1e-4)
    sc = torch.rand(cout, generator=g) + 0.5
    res = torch.randn(N * S, cout, generator=g)
    for relu, r in ((2, None), (0, res), (1, res), (2, res)):
        rc = None if r is None else r.cuda()
        y, _ = gpu_ops.conv3d_cl_bf16x3(xc, hc, lc, (N * S, 1, 1), 1, 1, False, sc.cuda(), bc, rc, relu)
        y_o, _ = oracle_ops.conv3d_cl_bf16x3(x, hi, lo, (N * S, 1, 1), 1, 1, False, sc, b, r, relu)
        close(y, y_o, tol=1e-4)
        try:
            gpu_ops.lib.call("sgc_set_tuning", b"rows_gemm", 0)
            y_tile, _ = gpu_ops.conv3d_cl_bf16x3(xc, hc, lc, (N * S, 1, 1), 1, 1, False, sc.cuda(), bc, rc, relu)
        finally:
            gpu_ops.lib.call("sgc_set_tuning", b"rows_gemm", 1)
        assert torch.equal(y, y_tile), (relu, r is not None)


@pytest.mark.parametrize("Nq,C,seen", [(400, 256, 0.7), (6401, 256, 0.95), (33, 256, 0.0), (1000, 128, 0.6), (31, 128, 1.0)])
def test_level_tail_equals_the_six_launches_it_replaces(Nq, C, seen, oracle_ops, gpu_ops):
    """sgc_level_tail (out_proj on the seen voxels + zero rows elsewhere -> LayerNorm -> FFN with identity -> LayerNorm in one
    launch; reference: TU/deformable_cross_attention.py:826-837, TU/encoder.py:311-338, mmcv FFN) against (a) the CPU
    oracle's composition, 1e-4 of the scale, and (b) the separate HIP launches it replaces -- bit for bit."""
    F = 2 * C
    g = torch.Generator().manual_seed(Nq + C)
    vis = torch.rand(Nq, generator=g) < seen
    row_of = torch.full((Nq,), -1, dtype=torch.int32)
    row_of[vis] = torch.arange(int(vis.sum()), dtype=torch.int32)
    n_valid = max(1, int(vis.sum()))
    ctx = torch.randn(n_valid, C, generator=g)
    mk = lambda o, i, s: torch.randn(o, i, generator=g) * s            # noqa: E731
    wo, w1, w2 = mk(C, C, 0.08), mk(F, C, 0.08), mk(C, F, 0.06)
    bo, b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(F, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    ln1 = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, 1e-5)
    ln2 = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, 1e-5)
    sp = lambda w: gpu_ops.split_bf16(w.view(1, *w.shape))             # noqa: E731
    so, s1, s2 = sp(wo), sp(w1), sp(w2)
    cu = lambda t: t.cuda() if isinstance(t, torch.Tensor) else t      # noqa: E731
    cup = lambda pr: tuple(cu(t) for t in pr)                          # noqa: E731
    y = gpu_ops.level_tail(cu(ctx), cu(row_of), cup(so), cu(bo), cup(ln1), cup(s1), cu(b1), cup(s2), cu(b2), cup(ln2))
    y_o = oracle_ops.level_tail(ctx, row_of, so, bo, ln1, s1, b1, s2, b2, ln2)
    close(y, y_o, tol=1e-4)
    # the six launches
    valid_index = torch.nonzero(vis).view(-1).to(torch.int32).cuda()
    x0 = torch.zeros(Nq, C, device="cuda")
    if int(vis.sum()):
        pooled = gpu_ops.linear_rows_bf16x3(cu(ctx), so[0].cuda(), so[1].cuda(), cu(bo))
        gpu_ops.scatter_rows(pooled, valid_index, x0)
    x1 = gpu_ops.layer_norm_rows(x0, cu(ln1[0]), cu(ln1[1]), ln1[2])
    h, _ = gpu_ops.conv3d_cl_bf16x3(x1, s1[0].cuda(), s1[1].cuda(), (Nq, 1, 1), 1, 1, False, None, cu(b1), None, 2)
    x2, _ = gpu_ops.conv3d_cl_bf16x3(h, s2[0].cuda(), s2[1].cuda(), (Nq, 1, 1), 1, 1, False, None, cu(b2), x1, 0)
    want = gpu_ops.layer_norm_rows(x2, cu(ln2[0]), cu(ln2[1]), ln2[2])
    assert torch.equal(y, want)


@pytest.mark.timeout(1200)
def test_reference_stress_shape_at_full_size():
    """The reference's own stress shape (packages/3D-deformable-attention/unittest_DFA3D.py:43-56: value [6, 30825, 8, 32],
    depth distributions of 112 bins, 4 levels 116x200 .. 15x25, 9 502 queries, 8 points -- the reference only checks it for
    NaN) at FULL size against the OpenMP build of the oracle: fused forward, one-stage == two-stage (the reference's
    MultiScale3DDeformableAttnFunction_fp32 runs the two _ext operators back to back), and the fused backward."""
    import oracle
    from sgcdet_amd import ext
    oo, go_ = oracle.ops(omp=True), ext.ops()
    levels = [(116, 200), (58, 100), (29, 50), (15, 25)]
    B, M, Cm, D, Q, P, L = 6, 8, 32, 112, 9502, 8, 4
    S = sum(h * w for h, w in levels)
    assert S == 30825
    g = torch.Generator().manual_seed(2024)
    shapes3 = torch.tensor([[h, w, D] for h, w in levels], dtype=torch.int64)
    lsi = torch.tensor([0, 23200, 29000, 30450], dtype=torch.int64)
    value = torch.randn(B, S, M, Cm, generator=g)
    dist = torch.randn(B, S, M, D, generator=g).softmax(-1).contiguous()
    loc = torch.rand(B, Q, M, L, P, 3, generator=g)
    attn = torch.rand(B, Q, M, L, P, generator=g)
    cu = lambda t: t.cuda()                                             # noqa: E731
    out_c, sc_c = oo.dfa3d_forward(value, dist, shapes3, lsi, loc, attn, want_score=True)
    out_g, sc_g = go_.dfa3d_forward(cu(value), cu(dist), cu(shapes3), cu(lsi), cu(loc), cu(attn), want_score=True)
    close(out_g, out_c)
    # per-corner depth scores: the split of a sample over its four corners is discontinuous at a pixel / bin border, so this
    # comparison of ALL 58 M scores holds only because the kernels compute `loc * size - 0.5` with the reference's two
    # roundings (float product, then the subtraction: csrc/common.hpp sample_coord) -- no border mask
    close(sc_g.cpu(), sc_c)
    # two-stage form through the _ext-compatible operators == the fused kernel
    sc2 = ext.ms_depth_score_sample_forward(cu(dist), cu(shapes3), cu(lsi), cu(loc), im2col_step=32)
    out2 = ext.wms_deform_attn_forward(cu(value), cu(shapes3)[:, :2].contiguous(), cu(lsi), cu(loc)[..., :2].contiguous(), cu(attn), sc2,
                                       im2col_step=32)
    assert torch.equal(sc2, sc_g)
    close(out2, out_c)
    go = torch.randn(B, Q, M * Cm, generator=g)
    rc = oo.dfa3d_backward(value, dist, shapes3, lsi, loc, attn, go)
    rg = go_.dfa3d_backward(cu(value), cu(dist), cu(shapes3), cu(lsi), cu(loc), cu(attn), cu(go))
    for name, a, b in zip(("grad_value", "grad_dist", "grad_loc", "grad_attn"), rg, rc):
        a = a.cpu()                                # grad_loc is discontinuous across a pixel border too: every entry is compared
        try:
            close(a, b, tol=5e-5)
        except AssertionError as e:
            raise AssertionError(f"{name}: {e}")


@pytest.mark.parametrize("rows,C", [(25600, 256), (3200, 512), (400, 1024), (1237, 128), (77, 1040), (6, 4)])
def test_batch_norm_rows_kernels_against_the_oracle_and_torch(rows, C, oracle_ops, gpu_ops):
    """sgc_bn_rows_forward / _backward (csrc/batch_norm.hip: the BatchNorm of the neck's training path on channels-last rows):
    output, batch statistics and the running-statistics update against the double-precision oracle, gradients against the
    oracle and F.batch_norm autograd in float64; ragged row / column counts (C > 1024 takes the column loop), a mean far from
    zero (Welford / Chan, not sum and sum of squares); two launches give the same bits (ordered merges, no atomics)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 1.5 + 40.0                      # |mean| >> std: a naive E[x^2] - E[x]^2 would lose 3 digits
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_o, rv_o, rm_g, rv_g = rm.clone(), rv.clone(), rm.cuda(), rv.cuda()
    y_o, mean_o, is_o = oracle_ops.bn_rows_forward(x, w, b, rm_o, rv_o, momentum=0.1, eps=1e-5)
    y_g, mean_g, is_g = gpu_ops.bn_rows_forward(x.cuda(), w.cuda(), b.cuda(), rm_g, rv_g, momentum=0.1, eps=1e-5)
    assert float((mean_g.cpu() - mean_o).abs().max()) < 1e-5 * 40
    assert float(((is_g.cpu() - is_o) / is_o).abs().max()) < 2e-5
    assert float((y_g.cpu() - y_o).abs().max()) < 2e-4                      # (x - mean) of 40-sized values: 4e-6 absolute per element, times invstd * w
    assert torch.allclose(rm_g.cpu(), rm_o, atol=1e-5) and torch.allclose(rv_g.cpu(), rv_o, rtol=3e-5)
    y_2, mean_2, is_2 = gpu_ops.bn_rows_forward(x.cuda(), w.cuda(), b.cuda(), None, None, momentum=0.1, eps=1e-5)
    assert torch.equal(y_2, y_g) and torch.equal(mean_2, mean_g) and torch.equal(is_2, is_g)
    dy = torch.randn(rows, C, generator=g)
    dx_o, dw_o, db_o = oracle_ops.bn_rows_backward(x, dy, mean_o, is_o, w)
    dx_g, dw_g, db_g = gpu_ops.bn_rows_backward(x.cuda(), dy.cuda(), mean_g, is_g, w.cuda())
    s = max(1.0, float(dw_o.abs().max()))
    assert float((dw_g.cpu() - dw_o).abs().max()) < 2e-4 * s and float((db_g.cpu() - db_o).abs().max()) < 2e-4 * max(1.0, float(db_o.abs().max()))
    assert float((dx_g.cpu() - dx_o).abs().max()) < 2e-4 * max(1.0, float(dx_o.abs().max()))
    dx_2, dw_2, db_2 = gpu_ops.bn_rows_backward(x.cuda(), dy.cuda(), mean_g, is_g, w.cuda())
    assert torch.equal(dx_2, dx_g) and torch.equal(dw_2, dw_g) and torch.equal(db_2, db_g)
    # the autograd Function the training path uses, against F.batch_norm in float64
    from sgcdet_amd.functions import BatchNormRowsFunction
    xg = x.cuda().requires_grad_(True)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = BatchNormRowsFunction.apply(xg, wg, bg, None, None, 0.1, 1e-5)
    y.backward(dy.cuda())
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    F.batch_norm(xd, None, None, wd, bd, True, 0.1, 1e-5).backward(dy.double())
    assert float((xg.grad.cpu().double() - xd.grad).abs().max()) < 2e-4 * max(1.0, float(xd.grad.abs().max()))
    assert float((wg.grad.cpu().double() - wd.grad).abs().max()) < 2e-4 * max(1.0, float(wd.grad.abs().max()))
    assert float((bg.grad.cpu().double() - bd.grad).abs().max()) < 2e-4 * max(1.0, float(bd.grad.abs().max()))


@pytest.mark.parametrize("rows,C,res,relu", [(25600, 256, True, True), (3200, 512, False, True), (1237, 128, True, False), (77, 1040, True, True)])
def test_batch_norm_rows_with_the_block_tail_inside(rows, C, res, relu, oracle_ops, gpu_ops):
    """sgc_bn_rows_act_forward / _backward: y = relu(bn(x) + residual) in the normalisation pass and its backward (gradient masked
    where y > 0, identity gradient = the masked gradient) -- bit-identical to the plain kernels followed by torch's add / relu and
    threshold_backward (the elementwise steps are the same float operations on the same values); the oracle twin; and
    BatchNormRowsFunction(residual, relu) against F.batch_norm + add + relu autograd in float64."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(rows + C + 1)
    x = (torch.randn(rows, C, generator=g) * 1.5 + 3.0).cuda()
    w, b = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    r = torch.randn(rows, C, generator=g).cuda() if res else None
    dy = torch.randn(rows, C, generator=g).cuda()
    y0, mean, invstd = gpu_ops.bn_rows_forward(x, w, b, None, None, momentum=0.1, eps=1e-5)
    want = y0 + r if res else y0
    want = torch.relu(want) if relu else want
    y1, mean1, invstd1 = gpu_ops.bn_rows_forward(x, w, b, None, None, momentum=0.1, eps=1e-5, residual=r, relu=relu)
    assert torch.equal(mean1, mean) and torch.equal(invstd1, invstd)
    assert torch.equal(y1, want)                       # fmaf-free elementwise tail: the same bits as the separate kernels
    gm = dy * (y1 > 0) if relu else dy
    dx0, dw0, db0 = gpu_ops.bn_rows_backward(x, gm.contiguous(), mean, invstd, w)
    out = gpu_ops.bn_rows_backward(x, dy, mean, invstd, w, y_relu=y1 if relu else None, want_dresidual=res)
    assert torch.equal(out[0], dx0) and torch.equal(out[1], dw0) and torch.equal(out[2], db0)
    if res:
        assert torch.equal(out[3], gm)
    tw = oracle_ops.bn_rows_forward(x.cpu(), w.cpu(), b.cpu(), None, None, momentum=0.1, eps=1e-5, residual=None if r is None else r.cpu(), relu=relu)
    assert float((y1.cpu() - tw[0]).abs().max()) < 2e-4
    # the autograd Function against torch in float64
    from sgcdet_amd.functions import BatchNormRowsFunction
    xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rg = r.clone().requires_grad_(True) if res else None
    BatchNormRowsFunction.apply(xg, wg, bg, None, None, 0.1, 1e-5, rg, relu).backward(dy)
    xd, wd, bd = x.cpu().double().requires_grad_(True), w.cpu().double().requires_grad_(True), b.cpu().double().requires_grad_(True)
    rd = r.cpu().double().requires_grad_(True) if res else None
    t = F.batch_norm(xd, None, None, wd, bd, True, 0.1, 1e-5)
    t = t + rd if res else t
    # the float64 reference masks on ITS OWN sign; entries whose fp32 pre-activation is within rounding of zero may differ: use the fp32 mask
    if relu:
        t = t * (y1.cpu() > 0).double()
    t.backward(dy.cpu().double())
    for a, d in ((xg.grad, xd.grad), (wg.grad, wd.grad), (bg.grad, bd.grad)) + (((rg.grad, rd.grad),) if res else ()):
        assert float((a.cpu().double() - d).abs().max()) < 2e-4 * max(1.0, float(d.abs().max()))


@pytest.mark.parametrize("N,Nq,C", [(40, 700, 256), (100, 900, 128), (3, 100, 256), (128, 300, 128)])
def test_projected_query_attention_against_oracle(N, Nq, C, oracle_ops, gpu_ops):
    """sgc_view_attend_pq vs its oracle twin (double accumulation) on random visible-pair lists: one and many cameras per voxel,
    voxels seen by exactly one camera (softmax weight 1), the device-side count, and -- composed with the V projection -- against
    sgc_view_attend on the in-projected k | v (the function it replaces).  Repeated launches are bit-identical."""
    from tests.test_oracle_identity import _pq_case
    heads = 8
    pooled, x, slot, valid_index, mha = _pq_case(N, Nq, C, heads, seed=5 + N, vis=0.3 if N > 3 else 0.6)
    hd = C // heads
    w, b = mha.in_proj_weight.detach().double(), mha.in_proj_bias.detach().double()
    q = pooled.double() @ w[:C].t() + b[:C]
    scale = (1.0 / hd) ** 0.5
    qp = torch.cat([scale * q[:, h * hd:(h + 1) * hd] @ w[C:2 * C][h * hd:(h + 1) * hd] for h in range(heads)], 1).float().contiguous()
    want = oracle_ops.view_attend_pq(qp, x, slot, valid_index, heads)
    cu = lambda t: t.cuda()
    got = gpu_ops.view_attend_pq(cu(qp), cu(x), cu(slot), cu(valid_index), heads)
    scale_s = max(1.0, float(want.abs().max()))
    assert (got.cpu() - want).abs().max() < 1e-5 * scale_s
    again = gpu_ops.view_attend_pq(cu(qp), cu(x), cu(slot), cu(valid_index), heads)
    assert torch.equal(got, again)
    # device-side row count: rows past it are not written
    n_valid = valid_index.numel()
    cnt = torch.tensor([n_valid - 3], dtype=torch.int32, device="cuda")
    part = gpu_ops.view_attend_pq(cu(qp), cu(x), cu(slot), cu(valid_index), heads, count=cnt)
    assert torch.equal(part[:n_valid - 3], got[:n_valid - 3])
    # composed with V: the attention it replaces
    ctx_pq = torch.cat([got.cpu()[:, h * C:(h + 1) * C].double() @ w[2 * C:][h * hd:(h + 1) * hd].t() for h in range(heads)], 1) + b[2 * C:]
    kv = (x.double() @ w[C:].t() + b[C:]).float().contiguous()
    ctx_ref = gpu_ops.view_attend(cu(q.float().contiguous()), cu(kv), cu(slot), cu(valid_index), heads).cpu()
    assert (ctx_pq.float() - ctx_ref).abs().max() < 2e-5 * max(1.0, float(ctx_ref.abs().max()))
    assert gpu_ops.view_attend_pq_supported(N, C, heads) and not gpu_ops.view_attend_pq_supported(129, C, heads)
    with pytest.raises(RuntimeError):
        gpu_ops.view_attend_pq(cu(qp[:, :4 * C].contiguous()), cu(x), cu(slot), cu(valid_index), 4)


@pytest.mark.parametrize("rows,K", [(6400, 256), (77, 256), (5000, 128), (33, 128), (40001, 128)])
def test_block_diagonal_linear_against_oracle_and_its_dense_form(rows, K, oracle_ops, gpu_ops):
    """sgc_linear_rows_blockdiag_bf16x3 (the V projection of the projected-query attention, head by head): against the oracle's
    per-group Linear, BIT-IDENTICAL to sgc_linear_rows_bf16x3 on the dense block-diagonal matrix it replaces (same products, same
    K order; the zero blocks add exact zeros), ragged row counts, the device-side count (rows past it untouched), repeatable."""
    G, Nh = 8, K // 8
    g = torch.Generator().manual_seed(rows + K)
    n_small = min(rows, 600)                                # the scalar oracle on a slice, the dense GPU form on everything
    x = torch.randn(rows, G * K, generator=g)
    w = torch.randn(G, Nh, K, generator=g) * (1.0 / K ** 0.5)
    b = torch.randn(G * Nh, generator=g) * 0.1
    hi, lo = gpu_ops.split_bf16(w)
    assert gpu_ops.linear_rows_blockdiag_supported(G, K, Nh) and not gpu_ops.linear_rows_blockdiag_supported(4, K, Nh)
    y = gpu_ops.linear_rows_blockdiag(x.cuda(), hi.cuda(), lo.cuda(), b.cuda())
    want = oracle_ops.linear_rows_blockdiag(x[:n_small].contiguous(), hi, lo, b)
    assert float((y[:n_small].cpu() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    dense = torch.zeros(G * Nh, G * K)
    for h in range(G):
        dense[h * Nh:(h + 1) * Nh, h * K:(h + 1) * K] = w[h]
    dhi, dlo = gpu_ops.split_bf16(dense.view(1, G * Nh, G * K))
    y_dense = gpu_ops.linear_rows_bf16x3(x.cuda(), dhi.cuda(), dlo.cuda(), b.cuda())
    assert torch.equal(y, y_dense)
    assert torch.equal(y, gpu_ops.linear_rows_blockdiag(x.cuda(), hi.cuda(), lo.cuda(), b.cuda()))
    # device-side count: rows past it are neither read nor written
    cnt = torch.tensor([max(1, rows - 7)], dtype=torch.int32, device="cuda")
    part = gpu_ops.linear_rows_blockdiag(x.cuda(), hi.cuda(), lo.cuda(), b.cuda(), count=cnt)
    n = int(cnt.item())
    assert torch.equal(part[:n], y[:n])
    with pytest.raises(RuntimeError):
        gpu_ops.linear_rows_blockdiag(x[:, :G * K - 32].contiguous().cuda(), hi.cuda(), lo.cuda(), b.cuda())


def test_block_diagonal_linear_in_the_one_product_modes(gpu_ops):
    """The opt-in arithmetic modes (plain bf16 / plain fp16 products, sgc_set_conv_products 1 / 2) of the block-diagonal Linear: the
    dense form in the same mode, bit for bit (the entry point follows the process-wide mode like every MFMA kernel of the library)."""
    G, K, Nh, rows = 8, 256, 32, 1000
    g = torch.Generator().manual_seed(99)
    x = torch.randn(rows, G * K, generator=g).cuda()
    w = (torch.randn(G, Nh, K, generator=g) * (1.0 / K ** 0.5)).cuda()
    b = (torch.randn(G * Nh, generator=g) * 0.1).cuda()
    dense = torch.zeros(1, G * Nh, G * K, device="cuda")
    for h in range(G):
        dense[0, h * Nh:(h + 1) * Nh, h * K:(h + 1) * K] = w[h]
    try:
        for mode in (1, 2):
            gpu_ops.lib.call("sgc_set_conv_products", mode)
            hi, lo = gpu_ops.split_operand(w)
            dhi, dlo = gpu_ops.split_operand(dense)
            y = gpu_ops.linear_rows_blockdiag(x, hi, lo, b)
            assert torch.equal(y, gpu_ops.linear_rows_bf16x3(x, dhi, dlo, b)), mode
            ref = torch.cat([x[:, h * K:(h + 1) * K].double() @ w[h].double().t() for h in range(G)], 1) + b.double()
            err = float((y.double() - ref).abs().max()) / float(ref.abs().max())
            assert 1e-6 < err < (2.0 ** -6 if mode == 1 else 2.0 ** -9), (mode, err)
    finally:
        gpu_ops.lib.call("sgc_set_conv_products", 3)


@pytest.mark.parametrize("rows,cin,cout", [(3000, 256, 256), (777, 64, 36), (12800, 128, 128)])
def test_linear_rows_with_the_zero_row_behind_the_result(rows, cin, cout, oracle_ops, gpu_ops):
    """sgc_linear_rows_zrow_bf16x3: the Linear's rows are bit-identical to sgc_linear_rows_bf16x3 (persistent row GEMM and tile
    kernel), the row behind them is zero after the SAME launch even when the buffer held garbage, with and without a
    device-side count."""
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, cin, generator=g).cuda()
    w = torch.randn(1, cout, cin, generator=g) * 0.1
    hi, lo = gpu_ops.split_bf16(w)
    hi, lo = hi.cuda(), lo.cuda()
    b = torch.randn(cout, generator=g).cuda()
    plain = gpu_ops.linear_rows_bf16x3(x, hi, lo, b)
    for trial in range(3):
        junk = torch.full((rows + 1, cout), float("nan"), device="cuda")       # whatever the allocator hands out next
        del junk
        y = gpu_ops.linear_rows_bf16x3(x, hi, lo, b, zero_tail=True)
        assert torch.equal(y, plain)
        tail = torch.as_strided(y, (1, cout), (cout, 1), storage_offset=y.storage_offset() + rows * cout)
        assert torch.equal(tail, torch.zeros_like(tail))
    cnt = torch.tensor([rows - 5], dtype=torch.int32, device="cuda")
    y = gpu_ops.linear_rows_bf16x3(x, hi, lo, b, count=cnt, zero_tail=True)
    assert torch.equal(y[:rows - 5], plain[:rows - 5])
    tail = torch.as_strided(y, (1, cout), (cout, 1), storage_offset=y.storage_offset() + rows * cout)
    assert torch.equal(tail, torch.zeros_like(tail))
    want = oracle_ops.linear_rows_bf16x3(x.cpu(), hi.cpu(), lo.cpu(), b.cpu(), zero_tail=True)
    assert (y[:rows - 5].cpu() - want[:rows - 5]).abs().max() < 1e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("N,C,Hs,Ws,H,W", [(5, 256, 64, 80, 64, 80), (3, 12, 60, 80, 59, 80), (2, 40, 17, 23, 15, 20), (1, 7, 9, 9, 9, 9)])
def test_rows_transpose_and_its_adjoint(N, C, Hs, Ws, H, W, oracle_ops, gpu_ops):
    """sgc_nhwc_to_nchw_pad is the adjoint of sgc_nchw_to_nhwc_crop (pure data movement: bit-exact against the oracle and
    against torch), and functions.NchwToRowsFunction gives the crop view of a map exactly the gradient autograd's
    flatten / permute / contiguous chain gives it (AdaptiveSparseHead.py:58-59, TU/transformer.py:151-170)."""
    from sgcdet_amd.functions import NchwToRowsFunction
    g = torch.Generator().manual_seed(N * C + H)
    rows = torch.randn(N, H * W, C, generator=g)
    want = oracle_ops.nhwc_to_nchw_pad(rows, H, W, Hs, Ws)
    got = gpu_ops.nhwc_to_nchw_pad(rows.cuda(), H, W, Hs, Ws)
    assert torch.equal(got.cpu(), want)
    ref = torch.zeros(N, C, Hs, Ws)
    ref[:, :, :H, :W] = rows.view(N, H, W, C).permute(0, 3, 1, 2)
    assert torch.equal(want, ref)
    x = torch.randn(1, N, C, Hs, Ws, generator=g).cuda()
    gy = torch.randn(N, H * W, C, generator=g).cuda()
    xa = x.clone().requires_grad_(True)
    ya = NchwToRowsFunction.apply(xa[:, :, :, :H, :W][0])
    ya.backward(gy)
    xb = x.clone().requires_grad_(True)
    yb = xb[:, :, :, :H, :W][0].flatten(2).permute(0, 2, 1).contiguous()
    yb.backward(gy)
    assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad)
