"""Shared inputs / layout conversions for the binned, head-major ("tiled") form of the deformable gather
(include/sgcdet_amd.h: sgc_bin_pairs, sgc_pairs_deform_gather_tiled, sgc_linear_rows_headmajor_bf16x3)."""
import torch


def raw_to_headmajor(raw, M, P):
    """[pairs, M*P*4] = [uv (m,p,xy) | dz (m,p) | logit (m,p)]  ->  [pairs, M*P*4] = (m, p, (du, dv, dz, logit))."""
    n = raw.shape[0]
    uv = raw[:, :M * P * 2].view(n, M, P, 2)
    dz = raw[:, M * P * 2:M * P * 3].view(n, M, P, 1)
    lg = raw[:, M * P * 3:].view(n, M, P, 1)
    return torch.cat([uv, dz, lg], -1).reshape(n, M * P * 4).contiguous()


def raw_row_permutation(M, P):
    """Row order of the fused [uv | dz | logit] projection weight that makes the GEMM emit the head-major layout."""
    idx = []
    for m in range(M):
        for p in range(P):
            mp = m * P + p
            idx += [mp * 2, mp * 2 + 1, M * P * 2 + mp, M * P * 3 + mp]
    return torch.tensor(idx, dtype=torch.long)


def value_to_headmajor(value):
    """[N, S, M, Cm] -> [N, M, S, Cm]."""
    return value.permute(0, 2, 1, 3).contiguous()


def check_bins(binned, before, ref_cam, n_pairs, H, W, bw, bh):
    """Structural check of ``bin_pairs``' result (dict) against the definition in include/sgcdet_amd.h.
    ``before``: the ``compact_pairs`` dict the call started from, with a CLONE of its slot table.  Returns the
    permutation ``old_index[new_pair]``."""
    N, Nq = ref_cam.shape[:2]
    nbx, nby = -(-W // bw), -(-H // bh)
    nb = nbx * nby
    off = binned["bin_offset"].cpu()
    ref = binned["pair_ref"].cpu()[:n_pairs]
    cam = before["pair_cam"].cpu().long()[:n_pairs]          # camera-major layout is kept
    q_new = binned["pair_q"].cpu().long()[:n_pairs]
    q_old = before["pair_q"].cpu().long()[:n_pairs]
    assert off.numel() == N * nb + 1 and int(off[0]) == 0 and int(off[-1]) == n_pairs
    assert bool((off[1:] >= off[:-1]).all())
    old_slot = before["slot"].cpu().long()
    old_index = old_slot[cam, q_new]                          # where the pair sat before
    assert bool((old_index >= 0).all()) and torch.equal(old_index.sort().values, torch.arange(n_pairs))
    assert torch.equal(q_old[old_index], q_new)
    assert torch.equal(cam[old_index], cam)                   # reordered inside each camera only
    new_slot = binned["slot"].cpu().long()
    assert torch.equal(new_slot[cam, q_new], torch.arange(n_pairs))          # slot rewritten to the new indices
    assert torch.equal(new_slot < 0, old_slot < 0)
    assert torch.equal(ref[:, :3], ref_cam.cpu()[cam, q_new])                # records carry (u, v, zn) verbatim
    assert torch.equal(ref[:, 3].contiguous().view(torch.int32).long(), q_new)
    px = torch.floor(ref[:, 0] * W - 0.5).clamp(0, W - 1).long()
    py = torch.floor(ref[:, 1] * H - 0.5).clamp(0, H - 1).long()
    grp = cam * nb + (py // bh) * nbx + px // bw
    want = torch.repeat_interleave(torch.arange(N * nb), (off[1:] - off[:-1]).long())
    assert torch.equal(grp, want)                                            # grouped by (camera, bin)
    same = grp[1:] == grp[:-1]
    assert bool((old_index[1:][same] > old_index[:-1][same]).all())          # stable: ascending original index
    return old_index
