"""The device-side row-count contract of include/sgcdet_amd.h (section 5, `*_dev_or_null`), checked on whichever
library implements the ABI: the HIP build (tests/test_gpu_kernels.py) and the CPU oracle (tests/test_abi_cpu.py)."""
import math

import torch


def _scene(N, Nq, seed):
    g = torch.Generator().manual_seed(seed)
    ref3d = (torch.rand(Nq, 3, generator=g) - 0.5) * torch.tensor([6.4, 6.4, 2.56])
    origin = torch.tensor([0.0, 0.0, 0.5])
    K = torch.tensor([[288.9, 0, 159.6], [0, 288.2, 120.9], [0, 0, 1.0]])
    proj = []
    for i in range(N):
        a = 2 * math.pi * i / N
        c = torch.tensor([2.2 * math.cos(a), 2.2 * math.sin(a), 1.4])
        f = torch.tensor([0.0, 0.0, 0.6]) - c
        f = f / f.norm()
        r = torch.linalg.cross(f, torch.tensor([0.0, 0.0, 1.0])); r = r / r.norm()
        d = torch.linalg.cross(f, r)
        R = torch.stack([r, d, f])
        E = torch.cat([R, (-R @ c)[:, None]], 1)
        proj.append(K @ E)
    return ref3d.contiguous(), origin, torch.stack(proj).contiguous()


def check_row_counts(ops, oracle_ops, dev):
    """view_mean / view_attend / scatter_rows / linear_rows with the row count in device memory and the buffers
    at capacity == the host-count calls on the live rows; rows past the count stay untouched.  Same contract
    on the oracle library (it implements the same ABI)."""
    N, Nq, C, H, W = 5, 300, 64, 7, 10
    ref3d, origin, proj = _scene(N, Nq, 9)
    rc, mk = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    pc = {k: v.to(dev) for k, v in oracle_ops.compact_pairs(mk).items()}
    n_pairs, n_valid = int(pc["totals"][0]), int(pc["totals"][1])
    assert 0 < n_valid < Nq and 0 < n_pairs < N * Nq
    pairs_cnt, valid_cnt = pc["totals"][0:1], pc["totals"][1:2]
    g = torch.Generator().manual_seed(5)
    cap = N * Nq
    feat = torch.randn(cap, C, generator=g).to(dev)
    sentinel = 12345.0
    # view_mean
    want = ops.view_mean(feat[:n_pairs].contiguous(), pc["slot"], pc["valid_index"], n_valid)
    got = ops.view_mean(feat, pc["slot"], pc["valid_index"], Nq, count=valid_cnt)
    assert got.shape[0] == Nq and torch.equal(got[:n_valid], want)
    # view_attend
    q = torch.randn(Nq, C, generator=g).to(dev)
    kv = torch.randn(cap, 2 * C, generator=g).to(dev)
    want = ops.view_attend(q[:n_valid].contiguous(), kv[:n_pairs].contiguous(), pc["slot"], pc["valid_index"], 8)
    got = ops.view_attend(q, kv, pc["slot"], pc["valid_index"], 8, count=valid_cnt)
    assert torch.equal(got[:n_valid], want)
    # scatter_rows: only the first count rows are scattered
    rows = torch.randn(Nq, C, generator=g).to(dev)
    vol_a = torch.full((Nq, C), sentinel).to(dev)
    vol_b = vol_a.clone()
    ops.scatter_rows(rows[:n_valid].contiguous(), pc["valid_index"][:n_valid].contiguous(), vol_a)
    ops.scatter_rows(rows, pc["valid_index"], vol_b, count=valid_cnt)
    assert torch.equal(vol_a, vol_b) and int((vol_b == sentinel).all(1).sum()) == Nq - n_valid
    # linear over the pair list: bit for bit the 1x1x1 convolution entry point on the live rows
    for cin, cout in ((C, 32), (128, 128), (256, 512), (256, 132)):
        xin = torch.randn(cap, cin, generator=g).to(dev)
        wt = (torch.randn(1, cout, cin, generator=g) * 0.1).to(dev)
        shift = torch.randn(cout, generator=g).to(dev)
        w_hi, w_lo = ops.split_bf16(wt)
        want, _ = ops.conv3d_cl_bf16x3(xin[:n_pairs].contiguous(), w_hi, w_lo, (n_pairs, 1, 1), 1, 1, False, None, shift)
        out = torch.full((cap, cout), sentinel).to(dev)
        got = ops.linear_rows_bf16x3(xin, w_hi, w_lo, shift, count=pairs_cnt, out=out)
        assert torch.equal(got[:n_pairs], want) and bool((got[n_pairs:] == sentinel).all()), (cin, cout)
        full = ops.linear_rows_bf16x3(xin[:n_pairs].contiguous(), w_hi, w_lo, shift)        # host-side count
        assert torch.equal(full, want), (cin, cout)
        zero = torch.zeros(1, dtype=torch.int32, device=dev)       # a count of zero: nothing happens
        out2 = torch.full((cap, cout), sentinel).to(dev)
        ops.linear_rows_bf16x3(xin, w_hi, w_lo, shift, count=zero, out=out2)
        assert bool((out2 == sentinel).all())
    ops.view_attend(q, kv, pc["slot"], pc["valid_index"], 8, count=torch.zeros(1, dtype=torch.int32, device=dev))


def check_camera_stride(ops, oracle_ops, dev):
    """`cam_stride_or_0` of the pair-list gathers and sgc_depth_pairs: maps that keep rows past the H x W crop
    (channels-last producer contract) give exactly the results of the compact maps."""
    N, Nq, C, H, W, Hs, D, M, P = 4, 200, 64, 7, 10, 9, 12, 8, 4
    ref3d, origin, proj = _scene(N, Nq, 13)
    rc, mk = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    pc = {k: v.to(dev) for k, v in oracle_ops.compact_pairs(mk).items()}
    rc = rc.to(dev)
    n_pairs = int(pc["totals"][0])
    g = torch.Generator().manual_seed(6)
    feat_full = torch.randn(N, Hs * W, C, generator=g).to(dev)           # Hs rows kept, the crop uses the first H
    dist_full = torch.randn(N, Hs * W, D, generator=g).mul(2).softmax(-1).contiguous().to(dev)
    feat, dist = feat_full[:, :H * W].contiguous(), dist_full[:, :H * W].contiguous()
    raw = (torch.randn(n_pairs, M * P * 4, generator=g) * 2).to(dev)
    a = ops.pairs_geometry_sample(feat, dist, rc, pc["pair_cam"], pc["pair_q"], n_pairs, H, W)
    b = ops.pairs_geometry_sample(feat_full, dist_full, rc, pc["pair_cam"], pc["pair_q"], n_pairs, H, W)
    assert torch.equal(a, b)
    assert torch.equal(ops.depth_pairs(dist, H, W), ops.depth_pairs(dist_full, H, W))
    for use_dp in (False, True):
        dp_c = ops.depth_pairs(dist, H, W) if use_dp else None
        dp_f = ops.depth_pairs(dist_full, H, W) if use_dp else None
        a = ops.pairs_deform_gather(feat.view(N, H * W, M, C // M), dist, rc, raw, pc["pair_cam"], pc["pair_q"], n_pairs,
                                    H, W, M, P, dist_pairs=dp_c)
        b = ops.pairs_deform_gather(feat_full.view(N, Hs * W, M, C // M), dist_full, rc, raw, pc["pair_cam"], pc["pair_q"],
                                    n_pairs, H, W, M, P, dist_pairs=dp_f)
        assert torch.equal(a, b), use_dp
