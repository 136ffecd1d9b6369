"""Comparison helpers (test infrastructure): top-k voxel selection is a discontinuous function
of fp32 scores, so "bit-exact indices" between two correct implementations can only be asked
for where the k-th / (k+1)-th score gap exceeds rounding noise.  These helpers make that
explicit instead of hiding it in lucky seeds."""
import torch


def topk_cut(occ, k):
    """(k-th largest value, gap to the (k+1)-th) of a flat score vector."""
    srt = occ.flatten().sort(descending=True).values
    return float(srt[k - 1]), float(srt[k - 1] - srt[k])


def check_sparse_head(volume_g, valid_g, occ_g, volume_c, valid_c, occ_c, n_vox_finest, topk_list,
                      tie_tol=2e-6, feat_tol=1e-3, coarse_injected=False):
    """Product (g) vs oracle (c) outputs of AdaptiveSparseHead.

    Returns a dict with the number of near-tie flips; raises AssertionError on a real mismatch:
      * occupancy scores agree within tie_tol * 50;
      * the finest selected sets are identical except for voxels whose oracle score sits within
        tie_tol of the cut value (near ties);
      * the coarser level's selected set (recomputed from each side's own scores) is identical
        (otherwise the caller should pick another seed: flips there move neighbours too);
      * voxel features agree within feat_tol on every voxel whose selection agrees.
    ``coarse_injected``: the product ran with the ORACLE's coarse selection forced in (scenes whose coarse cut sits at
    rounding-noise level: either side's own coarse top-k is then a coin toss, the rest of the path is not), so the
    coarse sets are equal by construction and are not re-derived from the scores."""
    volume_g, valid_g, occ_g = volume_g.detach().cpu(), valid_g.detach().cpu(), occ_g.detach().cpu()
    assert (occ_g - occ_c).abs().max() < 50 * tie_tol, "occupancy scores differ"
    occ2_c, occ1_c = occ_c[0, :n_vox_finest], occ_c[0, n_vox_finest:]
    occ1_g = occ_g[0, n_vox_finest:]
    set1_c = set(torch.topk(occ1_c, topk_list[0]).indices.tolist())
    set1_g = set(torch.topk(occ1_g, topk_list[0]).indices.tolist())
    assert coarse_injected or set1_c == set1_g, "coarse-level top-k sets differ (near tie at the coarse cut: use another seed)"
    cut2, _ = topk_cut(occ2_c, topk_list[1])
    diff = (valid_g != valid_c).flatten()
    if diff.any():
        worst = (occ2_c[diff] - cut2).abs().max().item()
        assert worst <= tie_tol, f"selected voxel sets differ beyond near ties (|score - cut| = {worst:.2e})"
    agree = ~diff
    C = volume_c.shape[1]
    vg = volume_g.reshape(C, -1)[:, agree]
    vc = volume_c.reshape(C, -1)[:, agree]
    err = (vg - vc).abs().max().item()
    scale = max(1.0, vc.abs().max().item())
    assert err <= feat_tol * scale, f"voxel features differ by {err:.3e} (scale {scale:.2f})"
    return dict(tie_flips=int(diff.sum()), max_err=err, scale=scale)
