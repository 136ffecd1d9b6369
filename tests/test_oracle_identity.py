"""Pins the C oracle's kernel arithmetic to an INDEPENDENT formulation: the DFA3D operator
equals 5-D ``F.grid_sample`` (bilinear, zeros padding, align_corners=False) over the outer
product value (x) depth_dist followed by the attention-weighted sum;
its backward equals torch autograd of that formulation."""
import pytest
import torch
import torch.nn.functional as F

CASES = [  # B, M, Cm, D, Q, P, levels
    (2, 4, 8, 6, 40, 3, [(5, 7), (3, 4)]),
    (1, 8, 4, 12, 64, 4, [(15, 20)]),
    (2, 1, 16, 12, 50, 1, [(7, 10)]),
    (1, 2, 5, 7, 30, 2, [(6, 5), (3, 3), (2, 2)]),
]


def closed_form(value, dist, loc, attn, levels):
    B, S, M, Cm = value.shape
    D = dist.shape[-1]
    Q, P = loc.shape[1], loc.shape[4]
    out, start = 0, 0
    for l, (H, W) in enumerate(levels):
        v = value[:, start:start + H * W].view(B, H, W, M, Cm)
        d = dist[:, start:start + H * W].view(B, H, W, M, D)
        vol = torch.einsum("bhwmc,bhwmd->bmcdhw", v, d).reshape(B * M, Cm, D, H, W)
        g = loc[:, :, :, l].permute(0, 2, 1, 3, 4).reshape(B * M, Q, P, 1, 3) * 2 - 1
        samp = F.grid_sample(vol, g, mode="bilinear", padding_mode="zeros", align_corners=False).view(B, M, Cm, Q, P)
        out = out + torch.einsum("bmcqp,bmqp->bqmc", samp, attn[:, :, :, l].permute(0, 2, 1, 3))
        start += H * W
    return out.reshape(B, Q, M * Cm)


def inputs(case, seed):
    B, M, Cm, D, Q, P, levels = case
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in levels)
    L = len(levels)
    shapes3 = torch.tensor([[h, w, D] for h, w in levels])
    lsi = torch.tensor([0] + [h * w for h, w in levels]).cumsum(0)[:-1].contiguous()
    value = torch.randn(B, S, M, Cm, generator=g)
    dist = torch.randn(B, S, M, D, generator=g).softmax(-1).contiguous()
    loc = (torch.rand(B, Q, M, L, P, 3, generator=g) * 1.4 - 0.2).contiguous()
    attn = torch.rand(B, Q, M, L, P, generator=g)
    return value, dist, shapes3, lsi, loc, attn


@pytest.mark.parametrize("case", CASES)
def test_forward_equals_grid_sample(case, oracle_ops):
    value, dist, shapes3, lsi, loc, attn = inputs(case, 0)
    out, score = oracle_ops.dfa3d_forward(value, dist, shapes3, lsi, loc, attn, want_score=True)
    ref = closed_form(value, dist, loc, attn, case[6])
    assert (out - ref).abs().max() < 2e-6
    # two-stage == one-stage (the implied KAT of unittest_DFA3D.py:9-29), bit for bit
    sc = oracle_ops.depth_score_forward(dist, shapes3, lsi, loc)
    out2 = oracle_ops.wms_forward(value, shapes3[:, :2].contiguous(), lsi, loc[..., :2].contiguous(), attn, sc)
    assert torch.equal(sc, score) and torch.equal(out2, out)


@pytest.mark.parametrize("case", CASES)
def test_backward_equals_autograd_of_grid_sample(case, oracle_ops):
    value, dist, shapes3, lsi, loc, attn = inputs(case, 1)
    leaves = [t.clone().requires_grad_() for t in (value, dist, loc, attn)]
    ref = closed_form(*leaves, case[6])
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
    want = torch.autograd.grad(ref, leaves, go)
    got = oracle_ops.dfa3d_backward(value, dist, shapes3, lsi, loc, attn, go.contiguous())
    for w, g in zip(want, got):
        assert (w - g).abs().max() <= 2e-5 * max(1.0, w.abs().max().item())


def test_one_unit_depth_bin_is_the_2d_deformable_attention(oracle_ops):
    """The 2-D classes (MSDeformableAttention3D / DeformCrossAttention, deformable_cross_attention.py:119,504) run the
    DFA3D operator on a depth map of ones with one bin and every sample at that bin's centre (z = 0.5): the result must be
    plain multi-scale deformable attention -- 4-D ``F.grid_sample`` (bilinear, zeros, align_corners=False) per level, the
    published formulation mmcv's ``multi_scale_deformable_attn_pytorch`` uses -- and every in-map depth score exactly 1."""
    for case in ((2, 4, 8, 1, 40, 3, [(5, 7), (3, 4)]), (1, 8, 4, 1, 64, 4, [(15, 20)])):
        B, M, Cm, _, Q, P, levels = case
        value, _, shapes3, lsi, loc, attn = inputs(case, 5)
        loc[..., 2] = 0.5
        ones = torch.ones(B, value.shape[1], 1, 1)
        out, score = oracle_ops.dfa3d_forward(value, ones, shapes3, lsi, loc, attn, want_score=True)
        assert bool(((score == 1) | (score == 0)).all()) and float(score.mean()) > 0.4   # per bilinear corner: 1 inside the map, 0 off it
        want, start = 0, 0
        for l, (H, W) in enumerate(levels):
            v = value[:, start:start + H * W].permute(0, 2, 3, 1).reshape(B * M, Cm, H, W)
            g = loc[:, :, :, l, :, :2].permute(0, 2, 1, 3, 4).reshape(B * M, Q, P, 2) * 2 - 1
            samp = F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False).view(B, M, Cm, Q, P)
            want = want + torch.einsum("bmcqp,bmqp->bqmc", samp, attn[:, :, :, l].permute(0, 2, 1, 3))
            start += H * W
        assert (out - want.reshape(B, Q, M * Cm)).abs().max() < 1e-5          # fp32 rounding of sums of up to L*P*4 products of O(1)


def test_unreplicated_depth_equals_replicated(oracle_ops):
    """dist_heads == 1 is the reference's `.repeat(1,1,num_heads,1)` without the copy; its
    gradient is the head-sum autograd would produce through the repeat."""
    value, dist, shapes3, lsi, loc, attn = inputs(CASES[1], 3)
    d1 = dist[:, :, :1].contiguous()
    drep = d1.repeat(1, 1, value.shape[2], 1).contiguous()
    a, _ = oracle_ops.dfa3d_forward(value, d1, shapes3, lsi, loc, attn)
    b, _ = oracle_ops.dfa3d_forward(value, drep, shapes3, lsi, loc, attn)
    assert torch.equal(a, b)
    go = torch.randn(a.shape, generator=torch.Generator().manual_seed(4))
    g1 = oracle_ops.dfa3d_backward(value, d1, shapes3, lsi, loc, attn, go)
    gr = oracle_ops.dfa3d_backward(value, drep, shapes3, lsi, loc, attn, go)
    assert (g1[1] - gr[1].sum(2, keepdim=True)).abs().max() < 1e-5
    assert torch.equal(g1[0], gr[0]) and torch.equal(g1[2], gr[2])


def test_tiled_gather_restatement_equals_the_plain_pair_list_form(oracle_ops):
    """The binned / head-major operator (sgc_bin_pairs + sgc_pairs_deform_gather_tiled) is the pair-list gather on a
    reordered pair list with permuted operands: identical rows (the oracle uses the same arithmetic order for both),
    for several bin sizes."""
    from tests.tile_contract import check_bins, raw_to_headmajor, value_to_headmajor
    from tests.test_gpu_kernels import _scene
    N, Nq, D, M, P, C, H, W = 5, 600, 12, 8, 4, 64, 15, 20
    ref3d, origin, proj = _scene(N, Nq, 4)
    rc, mk = oracle_ops.project_points(ref3d, origin, proj, 320., 239., 0.2, 5.0)
    pc = oracle_ops.compact_pairs(mk)
    n_pairs = int(pc["totals"][0])
    g = torch.Generator().manual_seed(5)
    value = torch.randn(N, H * W, M, C // M, generator=g)
    dist = torch.randn(N, H * W, D, generator=g).mul(2).softmax(-1).contiguous()
    raw = torch.randn(pc["pair_q"].numel(), M * P * 4, generator=g)
    raw[:, :M * P * 3] *= 3.0
    want = oracle_ops.pairs_deform_gather(value, dist, rc, raw, pc["pair_cam"], pc["pair_q"], n_pairs, H, W, M, P)
    for bw, bh in [(20, 15), (7, 4), (3, 16), (1, 1)]:
        before = dict(pc, slot=pc["slot"].clone())
        binned = oracle_ops.bin_pairs(rc, dict(pc, slot=pc["slot"].clone()), H, W, bw, bh)
        old = check_bins(binned, before, rc, n_pairs, H, W, bw, bh)
        raw_new = torch.zeros_like(raw)
        raw_new[:n_pairs] = raw[old]
        got = oracle_ops.pairs_deform_gather_tiled(value_to_headmajor(value), dist, binned["pair_ref"], binned["bin_offset"],
                                                   raw_to_headmajor(raw_new, M, P), H, W, P, bw, bh, 2, 2)
        assert torch.equal(got[:n_pairs], want[old])


def test_oracle_view_attend_backward_is_multihead_attention_autograd():
    """oracle sgc_view_attend / sgc_view_attend_backward over the pair list == nn.MultiheadAttention (query length 1, dense
    [N, L, C] key / value slots, key_padding_mask for the cameras that do not see the voxel) and its autograd in float64."""
    import torch
    import oracle
    oracle.build()
    ops = oracle.ops()
    g = torch.Generator().manual_seed(3)
    N, Nq, C, heads = 6, 50, 64, 8
    mask = torch.rand(N, Nq, generator=g) < 0.45
    mask[:, 7] = False
    cam, qi = mask.nonzero(as_tuple=True)
    n_pairs = cam.shape[0]
    slot = torch.full((N, Nq), -1, dtype=torch.int32)
    slot[cam, qi] = torch.arange(n_pairs, dtype=torch.int32)
    valid_index = mask.sum(0).nonzero()[:, 0]
    q = torch.randn(valid_index.shape[0], C, generator=g)
    kv = torch.randn(n_pairs, 2 * C, generator=g)
    gout = torch.randn(valid_index.shape[0], C, generator=g)
    ctx = ops.view_attend(q, kv, slot, valid_index.int(), heads)
    gq, gkv = ops.view_attend_backward(q, kv, slot, valid_index.int(), heads, ctx, gout)
    mha = torch.nn.MultiheadAttention(C, heads).double()
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.eye(C).repeat(3, 1)); mha.in_proj_bias.zero_()
        mha.out_proj.weight.copy_(torch.eye(C)); mha.out_proj.bias.zero_()
    qd, kvd = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    ks = torch.zeros(N, Nq, C, dtype=torch.float64).index_put((cam, qi), kvd[:, :C])[:, valid_index]
    vs = torch.zeros(N, Nq, C, dtype=torch.float64).index_put((cam, qi), kvd[:, C:])[:, valid_index]
    out, _ = mha(qd[None], ks, vs, key_padding_mask=~mask[:, valid_index].t())
    out[0].backward(gout.double())
    assert (ctx.double() - out[0].detach()).abs().max() < 2e-6
    assert (gq.double() - qd.grad).abs().max() < 2e-6 and (gkv.double() - kvd.grad).abs().max() < 2e-6


def _pq_case(N, Nq, C, heads, seed, vis=0.35):
    """Random visible-pair list + MHA weights: (q_in [n_valid, C] pooled features, x [pairs, C], slot, valid_index, mha)."""
    g = torch.Generator().manual_seed(seed)
    mask = torch.rand(N, Nq, generator=g) < vis
    mask[0, ::7] = True                                   # some voxels seen by exactly the first camera
    slot = torch.full((N, Nq), -1, dtype=torch.int32)
    slot[mask] = torch.arange(int(mask.sum()), dtype=torch.int32)
    valid_index = torch.nonzero(mask.any(0)).reshape(-1).to(torch.int32)
    x = torch.randn(int(mask.sum()), C, generator=g)
    pooled = torch.randn(valid_index.numel(), C, generator=g)
    torch.manual_seed(seed)
    mha = torch.nn.MultiheadAttention(C, heads, batch_first=False)
    with torch.no_grad():
        mha.in_proj_bias.copy_(torch.randn(3 * C, generator=g) * 0.3)
        mha.in_proj_weight.mul_(3.0)                     # sharper softmax than the xavier default
    return pooled, x, slot, valid_index, mha


@pytest.mark.parametrize("N,Nq,C,heads", [(5, 60, 32, 8), (40, 150, 128, 8), (13, 90, 256, 8)])
def test_projected_query_attention_equals_multihead_attention(N, Nq, C, heads, oracle_ops):
    """sgc_view_attend_pq (K and V off the pair list) is the same function as nn.MultiheadAttention over the views
    (TU/deformable_cross_attention.py:826-833): against torch's module itself on the reference's dense [N, L, C] slots with
    its key_padding_mask, and against the oracle's sgc_view_attend on the in-projected k | v."""
    pooled, x, slot, valid_index, mha = _pq_case(N, Nq, C, heads, seed=11)
    hd = C // heads
    w, b = mha.in_proj_weight.detach().double(), mha.in_proj_bias.detach().double()
    q = (pooled.double() @ w[:C].t() + b[:C])
    scale = (1.0 / hd) ** 0.5
    qp = torch.cat([scale * q[:, h * hd:(h + 1) * hd] @ w[C:2 * C][h * hd:(h + 1) * hd] for h in range(heads)], 1).float().contiguous()
    s = oracle_ops.view_attend_pq(qp, x, slot, valid_index, heads)                       # [n_valid, heads * C]
    ctx_pq = torch.cat([s[:, h * C:(h + 1) * C].double() @ w[2 * C:][h * hd:(h + 1) * hd].t() for h in range(heads)], 1) + b[2 * C:]
    # (1) the oracle's MHA on in-projected pairs
    kv = (x.double() @ w[C:].t() + b[C:]).float().contiguous()
    ctx_ref = oracle_ops.view_attend(q.float().contiguous(), kv, slot, valid_index, heads)
    assert (ctx_pq.float() - ctx_ref).abs().max() < 2e-5 * max(1.0, float(ctx_ref.abs().max()))
    # (2) torch's nn.MultiheadAttention on dense slots with the padding mask (the reference's formulation)
    n_valid = valid_index.numel()
    dense = torch.zeros(N, n_valid, C)
    sl = slot[:, valid_index.long()]
    dense[sl >= 0] = x[sl[sl >= 0].long()]
    with torch.no_grad():
        out, _ = mha(pooled[None], dense, dense, key_padding_mask=(sl < 0).t())
    want = out[0]
    got = (ctx_pq @ mha.out_proj.weight.detach().double().t() + mha.out_proj.bias.detach().double()).float()
    assert (got - want).abs().max() < 2e-5 * max(1.0, float(want.abs().max()))
