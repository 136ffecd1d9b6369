"""Tensor-level front end of the C ABI (``include/sgcdet_amd.h``).

``TensorOps`` checks what the reference's C++ entry points check (same device,
contiguous, dtypes -- csrc/cuda/wms_deform_attn_cuda.cu:220-238,
csrc/common/pytorch_device_registry.hpp:111-126 -> ``RuntimeError``), allocates
the outputs the reference allocates (``at::zeros``), and forwards raw pointers
plus the current HIP stream to the shared library.  torch is only plumbing here:
device memory and the stream handle.

The same class fronts the CPU oracle in the tests (device_type ``"cpu"``); the
product instantiates it exactly once, for ``"cuda"``, in ``sgcdet_amd.ext``.
"""
import torch


def _stream_ptr(device_type, index=None):
    if device_type == "cuda":
        if index is not None and hasattr(torch._C, "_cuda_getCurrentRawStream"):
            return torch._C._cuda_getCurrentRawStream(index)          # no Stream object round trip
        return torch.cuda.current_stream().cuda_stream
    return None


class TensorOps:
    def __init__(self, library, device_type):
        self.lib = library
        self.device_type = device_type
        # when set to a list, every CUDA call is bracketed by HIP events on the launch stream
        # and (name, meta, start, end) is appended -- bench.py's per-kernel timing
        self.event_log = None
        self.event_names = None      # optional set of entry points to time (None = all)

    # ---- argument checks ------------------------------------------------
    def _check(self, **tensors):
        dev = None
        for name, t in tensors.items():
            if t is None:
                continue
            if not isinstance(t, torch.Tensor):
                raise RuntimeError(f"{name} must be a tensor")
            if t.device.type != self.device_type:
                raise RuntimeError(
                    f"{name} must be a {self.device_type} tensor for the "
                    f"'{self.lib.backend}' backend (got {t.device})")
            if not t.is_contiguous():
                raise RuntimeError(f"{name} tensor has to be contiguous")
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise RuntimeError(f"{name} is on {t.device}, expected {dev}")
        return dev

    @staticmethod
    def _f32(**tensors):
        for name, t in tensors.items():
            if t is not None and t.dtype != torch.float32:
                raise RuntimeError(f"{name} must be float32 (got {t.dtype})")

    @staticmethod
    def _i64(**tensors):
        for name, t in tensors.items():
            if t.dtype != torch.int64:
                raise RuntimeError(f"{name} must be int64 (got {t.dtype})")

    @staticmethod
    def _i32(**tensors):
        for name, t in tensors.items():
            if t is not None and t.dtype != torch.int32:
                raise RuntimeError(f"{name} must be int32 (got {t.dtype})")

    n_calls = 0          # library entry points called through this front end (bench.py reports calls per scene)

    def _call(self, name, *args, _meta=None):
        self.n_calls += 1
        ptrs = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
        if self.device_type == "cuda":
            dev = next(a.device for a in args if isinstance(a, torch.Tensor))
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            if idx != torch.cuda.current_device():
                with torch.cuda.device(dev):
                    return self._call(name, *args, _meta=_meta)
            if self.event_log is None or (self.event_names is not None and name not in self.event_names):
                return self.lib.call(name, *ptrs, _stream_ptr("cuda", idx))
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = self.lib.call(name, *ptrs, _stream_ptr("cuda", idx))
            e1.record()
            self.event_log.append((name, _meta or {}, e0, e1))
            return rc
        return self.lib.call(name, *ptrs, None)

    # ---- 1. the four dfa3D._ext operators ----------------------------------
    def depth_score_forward(self, dist, shapes3, lsi, loc3):
        self._check(value=dist, value_spatial_shapes=shapes3, value_level_start_index=lsi,
                    sampling_locations=loc3)
        self._f32(value=dist, sampling_locations=loc3)
        self._i64(value_spatial_shapes=shapes3, value_level_start_index=lsi)
        B, S, M, D = dist.shape
        _, Q, M2, L, P, three = loc3.shape
        if M2 != M or three != 3 or shapes3.shape != (L, 3) or loc3.shape[0] != B:
            raise RuntimeError("ms_depth_score_sample_forward: inconsistent shapes")
        score = torch.empty((B, Q, M, L, P, 4), dtype=dist.dtype, device=dist.device)
        self._call("sgc_depth_score_forward", dist, shapes3, lsi, loc3, score, B, S, M, D, L, Q, P)
        return score

    def wms_forward(self, value, shapes2, lsi, loc2, attn, score):
        self._check(value=value, value_spatial_shapes=shapes2, value_level_start_index=lsi,
                    sampling_locations=loc2, attention_weights=attn, depth_scores=score)
        self._f32(value=value, sampling_locations=loc2, attention_weights=attn, depth_scores=score)
        self._i64(value_spatial_shapes=shapes2, value_level_start_index=lsi)
        B, S, M, Cm = value.shape
        _, Q, _, L, P, two = loc2.shape
        if two != 2 or shapes2.shape != (L, 2) or attn.shape != (B, Q, M, L, P) \
                or score.shape != (B, Q, M, L, P, 4):
            raise RuntimeError("wms_deform_attn_forward: inconsistent shapes")
        out = torch.empty((B, Q, M * Cm), dtype=value.dtype, device=value.device)
        self._call("sgc_wms_forward", value, shapes2, lsi, loc2, attn, score, out, B, S, M, Cm, L, Q, P)
        return out

    def wms_backward(self, value, shapes2, lsi, loc2, attn, score, grad_out,
                     grad_value, grad_loc2, grad_attn, grad_score):
        self._check(value=value, value_spatial_shapes=shapes2, value_level_start_index=lsi,
                    sampling_locations=loc2, attention_weights=attn, depth_scores=score,
                    grad_output=grad_out, grad_value=grad_value, grad_sampling_loc=grad_loc2,
                    grad_attn_weight=grad_attn, grad_depth_score=grad_score)
        self._f32(value=value, sampling_locations=loc2, attention_weights=attn, depth_scores=score,
                  grad_output=grad_out, grad_value=grad_value, grad_sampling_loc=grad_loc2,
                  grad_attn_weight=grad_attn, grad_depth_score=grad_score)
        self._i64(value_spatial_shapes=shapes2, value_level_start_index=lsi)
        B, S, M, Cm = value.shape
        _, Q, _, L, P, _ = loc2.shape
        self._call("sgc_wms_backward", value, shapes2, lsi, loc2, attn, score, grad_out, grad_value,
                   grad_loc2, grad_attn, grad_score, B, S, M, Cm, L, Q, P)

    def depth_score_backward(self, dist, shapes3, lsi, loc3, grad_score, grad_dist, grad_loc3):
        self._check(value=dist, value_spatial_shapes=shapes3, value_level_start_index=lsi,
                    sampling_locations=loc3, grad_output=grad_score, grad_value=grad_dist,
                    grad_sampling_loc=grad_loc3)
        self._f32(value=dist, sampling_locations=loc3, grad_output=grad_score, grad_value=grad_dist,
                  grad_sampling_loc=grad_loc3)
        self._i64(value_spatial_shapes=shapes3, value_level_start_index=lsi)
        B, S, M, D = dist.shape
        _, Q, _, L, P, _ = loc3.shape
        self._call("sgc_depth_score_backward", dist, shapes3, lsi, loc3, grad_score, grad_dist,
                   grad_loc3, B, S, M, D, L, Q, P)

    # ---- 2. fused forms ---------------------------------------------------
    def dfa3d_forward(self, value, dist, shapes3, lsi, loc3, attn=None, want_score=False):
        self._check(value=value, value_dpt_dist=dist, value_spatial_shapes=shapes3,
                    value_level_start_index=lsi, sampling_locations=loc3, attention_weights=attn)
        self._f32(value=value, value_dpt_dist=dist, sampling_locations=loc3, attention_weights=attn)
        self._i64(value_spatial_shapes=shapes3, value_level_start_index=lsi)
        B, S, M, Cm = value.shape
        dist_heads, D = dist.shape[2], dist.shape[3]
        _, Q, _, L, P, _ = loc3.shape
        if dist.shape[:2] != (B, S) or dist_heads not in (1, M) or loc3.shape[2] != M:
            raise RuntimeError("dfa3d_forward: inconsistent shapes")
        out = torch.empty((B, Q, M * Cm), dtype=value.dtype, device=value.device)
        score = torch.empty((B, Q, M, L, P, 4), dtype=value.dtype, device=value.device) if want_score else None
        if B * Q == 0:                        # nothing to sample (e.g. no camera sees any voxel)
            return out, score
        self._call("sgc_dfa3d_forward", value, dist, shapes3, lsi, loc3, attn, out, score,
                   B, S, M, Cm, D, dist_heads, L, Q, P)
        return out, score

    def dfa3d_backward(self, value, dist, shapes3, lsi, loc3, attn, grad_out, want_grad_attn=True):
        self._check(value=value, value_dpt_dist=dist, value_spatial_shapes=shapes3,
                    value_level_start_index=lsi, sampling_locations=loc3, attention_weights=attn,
                    grad_output=grad_out)
        self._f32(value=value, value_dpt_dist=dist, sampling_locations=loc3, attention_weights=attn,
                  grad_output=grad_out)
        B, S, M, Cm = value.shape
        dist_heads, D = dist.shape[2], dist.shape[3]
        _, Q, _, L, P, _ = loc3.shape
        grad_value = torch.zeros_like(value)
        grad_dist = torch.zeros_like(dist)
        grad_loc3 = torch.empty_like(loc3)
        grad_attn = torch.empty((B, Q, M, L, P), dtype=value.dtype, device=value.device) \
            if want_grad_attn else None
        if B * Q == 0:
            return grad_value, grad_dist, grad_loc3, grad_attn
        self._call("sgc_dfa3d_backward", value, dist, shapes3, lsi, loc3, attn, grad_out, grad_value,
                   grad_dist, grad_loc3, grad_attn, B, S, M, Cm, D, dist_heads, L, Q, P)
        return grad_value, grad_dist, grad_loc3, grad_attn

    def dfa3d_forward_items(self, value, dist, shapes3, lsi, loc3, attn, item_batch):
        """Item-list form: value [B,S,M,Cm], dist [B,S,dh,D], loc3 [n,M,L,P,3], attn [n,M,L,P] | None, item_batch [n] int32
        -> out [n, M*Cm]."""
        self._check(value=value, value_dpt_dist=dist, value_spatial_shapes=shapes3, value_level_start_index=lsi,
                    sampling_locations=loc3, attention_weights=attn, item_batch=item_batch)
        self._f32(value=value, value_dpt_dist=dist, sampling_locations=loc3, attention_weights=attn)
        self._i64(value_spatial_shapes=shapes3, value_level_start_index=lsi)
        self._i32(item_batch=item_batch)
        B, S, M, Cm = value.shape
        dist_heads, D = dist.shape[2], dist.shape[3]
        n, M2, L, P, _ = loc3.shape
        if M2 != M or item_batch.numel() != n or dist.shape[:2] != (B, S):
            raise RuntimeError("dfa3d_forward_items: inconsistent shapes")
        out = torch.empty((n, M * Cm), dtype=value.dtype, device=value.device)
        if n:
            self._call("sgc_dfa3d_forward_items", value, dist, shapes3, lsi, loc3, attn, item_batch, out, None,
                       B, S, M, Cm, D, dist_heads, L, n, P)
        return out

    def dfa3d_backward_items(self, value, dist, shapes3, lsi, loc3, attn, item_batch, grad_out, want_grad_attn=True):
        self._check(value=value, value_dpt_dist=dist, sampling_locations=loc3, attention_weights=attn,
                    item_batch=item_batch, grad_output=grad_out)
        self._f32(value=value, value_dpt_dist=dist, sampling_locations=loc3, attention_weights=attn, grad_output=grad_out)
        B, S, M, Cm = value.shape
        dist_heads, D = dist.shape[2], dist.shape[3]
        n, _, L, P, _ = loc3.shape
        grad_value = torch.zeros_like(value)
        grad_dist = torch.zeros_like(dist)
        grad_loc3 = torch.empty_like(loc3)
        grad_attn = torch.empty((n, M, L, P), dtype=value.dtype, device=value.device) if want_grad_attn else None
        if n:
            self._call("sgc_dfa3d_backward_items", value, dist, shapes3, lsi, loc3, attn, item_batch, grad_out, grad_value,
                       grad_dist, grad_loc3, grad_attn, B, S, M, Cm, D, dist_heads, L, n, P)
        return grad_value, grad_dist, grad_loc3, grad_attn

    def dfa3d_backward_binned_fits(self, H, W, Cm, D, bin_w, bin_h, halo):
        """Does the (bin + halo) window of the binned backward fit the LDS of a CU?"""
        n = int(self.lib._dll.sgc_dfa3d_backward_binned_lds_bytes(int(H), int(W), int(Cm), int(D), int(bin_w), int(bin_h),
                                                                   int(halo[0]), int(halo[1])))
        return 0 < n <= 160 * 1024 and Cm in (16, 32)

    def dfa3d_backward_binned(self, value, dist, loc3, attn, bin_offset, grad_out, H, W, bin_w, bin_h, halo=(2, 2),
                              want_grad_loc=True, want_grad_attn=True, head_shift=None):
        """Backward of the one-level DFA3D operator over a BINNED item list (``sgc_dfa3d_backward_binned``): items in the
        (camera, bin) order of ``bin_pairs``.  value [N,S,M,Cm]; dist [N,S,D] or [N,S,1,D]; loc3 [n,LM,(1,)P,3]; attn
        [n,LM,(1,)P] or None (= 1); grad_out [n, M*Cm].  LM = M, or 1 = one sample set shared by the M channel groups."""
        self._check(value=value, value_dpt_dist=dist, sampling_locations=loc3, attention_weights=attn, bin_offset=bin_offset,
                    grad_output=grad_out)
        self._f32(value=value, value_dpt_dist=dist, sampling_locations=loc3, attention_weights=attn, grad_output=grad_out)
        if bin_offset.dtype != torch.int32:
            raise RuntimeError("bin_offset must be int32")
        N, S, M, Cm = value.shape
        if head_shift is not None:
            self._check(head_shift=head_shift)
            if head_shift.dtype != torch.int32 or head_shift.numel() != 2 * M or not head_shift.is_contiguous():
                raise RuntimeError("head_shift must be a contiguous int32 [M, 2]")
        D = dist.shape[-1]
        if dist.numel() != N * S * D:
            raise RuntimeError("dfa3d_backward_binned: one depth map per camera expected (dist_heads == 1)")
        n, LM, P = loc3.shape[0], loc3.shape[1], loc3.shape[-2]
        if loc3.numel() != n * LM * P * 3 or grad_out.shape != (n, M * Cm) or (attn is not None and attn.numel() != n * LM * P):
            raise RuntimeError("dfa3d_backward_binned: inconsistent shapes")
        nb = -(-W // bin_w) * -(-H // bin_h)
        if bin_offset.numel() < N * nb + 1:
            raise RuntimeError("dfa3d_backward_binned: bin_offset too short for these bins")
        grad_value = torch.zeros_like(value)
        grad_dist = torch.zeros_like(dist)
        # a sample set shared by the M channel groups: its gradients are summed over the groups' workgroups (atomics into zeros)
        alloc = torch.zeros if (LM == 1 and M > 1) else torch.empty
        grad_loc3 = alloc(loc3.shape, dtype=value.dtype, device=value.device) if want_grad_loc else None
        grad_attn = alloc(loc3.shape[:-1], dtype=value.dtype, device=value.device) if want_grad_attn else None
        if n:
            self._call("sgc_dfa3d_backward_binned", value, dist, loc3, attn, bin_offset, head_shift, grad_out, grad_value, grad_dist, grad_loc3,
                       grad_attn, N, S, int(H), int(W), M, Cm, D, LM, P, int(bin_w), int(bin_h), int(halo[0]), int(halo[1]))
        return grad_value, grad_dist, grad_loc3, grad_attn

    # ---- 3. projection + compaction ---------------------------------------
    def project_points(self, ref3d, origin, proj, img_w, img_h, d_near, d_far, sel=None):
        """``sel``: optional int64 [Nq] -- query q is voxel ``sel[q]`` of ``ref3d`` (the reference gathers
        ``ref_3d[vox_coords[unmasked_idx, 3]]`` first, transformer.py:145-146)."""
        self._check(ref3d=ref3d, origin=origin, proj=proj, sel=sel)
        self._f32(ref3d=ref3d, origin=origin, proj=proj)
        if sel is not None:
            self._i64(sel=sel)
        N, Nq = proj.shape[0], (ref3d.shape[0] if sel is None else sel.numel())
        if proj.shape[1:] != (3, 4) or ref3d.shape[1] != 3 or origin.numel() != 3:
            raise RuntimeError("project_points: inconsistent shapes")
        ref_cam = torch.empty((N, Nq, 3), dtype=torch.float32, device=ref3d.device)
        mask = torch.empty((N, Nq), dtype=torch.uint8, device=ref3d.device)
        if Nq:
            self._call("sgc_project_points", ref3d, sel, origin, proj, ref_cam, mask, N, Nq,
                       float(img_w), float(img_h), float(d_near), float(d_far))
        return ref_cam, mask

    def compact_pairs(self, mask, cap=None):
        """Returns a dict of int32 tensors; ``totals`` stays on the device."""
        self._check(mask=mask)
        if mask.dtype != torch.uint8:
            raise RuntimeError("mask must be uint8")
        N, Nq = mask.shape
        cap = N * Nq if cap is None else cap
        dev = mask.device
        i32 = dict(dtype=torch.int32, device=dev)
        out = dict(
            cam_count=torch.empty(N, **i32), cam_offset=torch.empty(N + 1, **i32),
            pair_cam=torch.empty(cap, **i32), pair_q=torch.empty(cap, **i32),
            slot=torch.empty((N, Nq), **i32), vox_count=torch.empty(Nq, **i32),
            valid_index=torch.empty(Nq, **i32), totals=torch.empty(4, **i32))
        ws = torch.empty(N * Nq + Nq + 2 * N + 64, **i32)
        self._call("sgc_compact_pairs", mask, N, Nq, out["cam_count"], out["cam_offset"],
                   out["pair_cam"], out["pair_q"], out["slot"], out["vox_count"],
                   out["valid_index"], out["totals"], ws)
        out["row_of"] = ws[:Nq]            # inverse of valid_index (-1: seen by no camera), left in the workspace
        return out

    def pairs_geometry_linear_supported(self, C, Cout, N, S):
        return bool(self.lib._dll.sgc_pairs_geometry_linear_supported(int(C), int(Cout), int(N), int(S)))

    def pairs_geometry_linear(self, feat, dist, ref_cam, pair_cam, pair_q, n_pairs, H, W, w_hi, w_lo, shift=None, totals=None):
        """The geometry-aware sample of every visible pair FUSED with the Linear that consumes it
        (``sgc_pairs_geometry_linear_bf16x3``): y [cap, Cout] = sample(feat, dist, ref)[cap, C] @ W^T + shift without the sampled rows
        in memory.  Arguments as ``pairs_geometry_sample`` (``n_pairs`` < 0: the count is ``totals[0]`` on the device) + the bf16
        hi / lo planes [1, Cout, C] of the weight.  Bit-identical to ``pairs_geometry_sample`` + ``linear_rows_bf16x3``."""
        self._check(feat=feat, dist=dist, ref_cam=ref_cam, pair_cam=pair_cam, pair_q=pair_q, totals=totals, w_hi=w_hi, w_lo=w_lo, shift=shift)
        self._f32(feat=feat, dist=dist, ref_cam=ref_cam, shift=shift)
        self._i32(pair_cam=pair_cam, pair_q=pair_q, totals=totals)
        N, S, C = feat.shape
        D = dist.shape[-1]
        Nq = ref_cam.shape[1]
        cap = pair_q.numel()
        Cout = w_hi.shape[-2]
        if w_hi.dtype != torch.bfloat16 or w_hi.shape != w_lo.shape or w_hi.shape[-1] != C or w_hi.numel() != Cout * C:
            raise RuntimeError("pairs_geometry_linear: w_hi / w_lo must be bfloat16 [1, Cout, C]")
        if S < H * W or dist.numel() != N * S * D:
            raise RuntimeError("pairs_geometry_linear: inconsistent map shapes")
        y = torch.empty((cap, Cout), dtype=torch.float32, device=feat.device)
        if cap == 0 or n_pairs == 0:
            return y
        ws = torch.empty(max(int(self.lib._dll.sgc_pairs_geometry_linear_workspace_bytes(cap)), 16), dtype=torch.uint8, device=feat.device)
        self._call("sgc_pairs_geometry_linear_bf16x3", feat, dist, ref_cam, pair_cam, pair_q, totals, w_hi, w_lo, shift, y, ws, N, ref_cam.shape[1],
                   int(H), int(W), C, D, Cout, S if S != H * W else 0, int(n_pairs), cap,
                   _meta=dict(V=cap if n_pairs < 0 else int(n_pairs), Cin=C, Cout=Cout, taps=1, OV=cap if n_pairs < 0 else int(n_pairs),
                              N=N, H=int(H), W=int(W), C=C, D=D, n_pairs=cap if n_pairs < 0 else int(n_pairs)))
        return y if n_pairs < 0 else y[:int(n_pairs)]

    # ---- 4. pair-list gathers ------------------------------------------------
    def pairs_geometry_sample(self, feat, dist, ref_cam, pair_cam, pair_q, n_pairs, H, W, totals=None):
        self._check(feat=feat, dist=dist, ref_cam=ref_cam, pair_cam=pair_cam, pair_q=pair_q, totals=totals)
        self._f32(feat=feat, dist=dist, ref_cam=ref_cam)
        self._i32(pair_cam=pair_cam, pair_q=pair_q, totals=totals)
        N, S, Cc = feat.shape          # S >= H*W: pixels between consecutive cameras (channels-last maps that kept
        D = dist.shape[-1]             # the rows the reference crops away, include/sgcdet_amd.h `cam_stride_or_0`)
        Nq = ref_cam.shape[1]
        if S < H * W or dist.shape[:2] != (N, S) or ref_cam.shape != (N, Nq, 3):
            raise RuntimeError("pairs_geometry_sample: inconsistent shapes")
        cap = pair_cam.numel()
        rows = n_pairs if n_pairs >= 0 else cap
        out = torch.empty((rows, Cc), dtype=torch.float32, device=feat.device)
        if rows == 0:
            return out
        self._call("sgc_pairs_geometry_sample", feat, dist, ref_cam, pair_cam, pair_q, totals, out,
                   N, Nq, H, W, Cc, D, S, n_pairs, cap,
                   _meta=dict(N=N, H=H, W=W, C=Cc, D=D, n_pairs=rows))
        return out

    def depth_pairs(self, dist, H, W):
        """dist [N, S >= H*W, D] -> pair-interleaved copy [N, H, W+1, D, 2] for ``pairs_deform_gather``."""
        self._check(dist=dist)
        self._f32(dist=dist)
        N, S, D = dist.shape
        if S < H * W:
            raise RuntimeError("depth_pairs: inconsistent shapes")
        dp = torch.empty((N, H, W + 1, D, 2), dtype=torch.float32, device=dist.device)
        self._call("sgc_depth_pairs", dist, dp, N, H, W, D, S)
        return dp

    def pairs_deform_gather(self, value, dist, ref_cam, raw, pair_cam, pair_q, n_pairs, H, W, M, P,
                            totals=None, dist_pairs=None, zero_row=False):
        """``zero_row=True`` promises that ``value`` is a view of a buffer holding one extra all-zero row
        right behind its N*S rows (see ``LinearSpec.__call__(extra_zero_row=True)``)."""
        self._check(value=value, dist=dist, ref_cam=ref_cam, raw=raw, pair_cam=pair_cam,
                    pair_q=pair_q, totals=totals, dist_pairs=dist_pairs)
        self._f32(value=value, dist=dist, ref_cam=ref_cam, raw=raw, dist_pairs=dist_pairs)
        self._i32(pair_cam=pair_cam, pair_q=pair_q, totals=totals)
        N, S, Cc = value.shape[0], value.shape[1], value.shape[-1] * (value.shape[2] if value.dim() == 4 else 1)
        Cm = Cc // M
        D = dist.shape[-1]
        Nq = ref_cam.shape[1]
        cap = pair_cam.numel()
        rows = n_pairs if n_pairs >= 0 else cap
        if S < H * W or dist.shape[:2] != (N, S) or raw.shape[-1] != M * P * 4 or raw.shape[0] < rows:
            raise RuntimeError("pairs_deform_gather: inconsistent shapes")
        out = torch.empty((rows, Cc), dtype=torch.float32, device=value.device)
        if rows == 0:
            return out
        if dist_pairs is not None and dist_pairs.shape != (N, H, W + 1, D, 2):
            raise RuntimeError("pairs_deform_gather: dist_pairs must be [N, H, W+1, D, 2]")
        self._call("sgc_pairs_deform_gather", value, dist, dist_pairs, ref_cam, raw, pair_cam, pair_q, totals, out,
                   N, Nq, H, W, M, Cm, D, P, S, 1 if zero_row else 0, n_pairs, cap,
                   _meta=dict(N=N, H=H, W=W, C=Cc, D=D, M=M, P=P, n_pairs=rows))
        return out

    # ---- 4b. LDS-tiled gather: binning, head-major operands ---------------------------------
    def tile_window(self, H, W, Cm, D, bin_w, bin_h, halo_x, halo_y, max_shift=(0, 0), depth_in_lds=True, value_bf16=False):
        """dict(tw, th, lds_bytes, nbuf, depth_in_lds) of what ``pairs_deform_gather_tiled`` would stage."""
        import ctypes
        v = [ctypes.c_int() for _ in range(5)]
        self.lib._dll.sgc_tile_window(H, W, Cm, D, bin_w, bin_h, halo_x, halo_y, int(max_shift[0]), int(max_shift[1]),
                                      int(bool(depth_in_lds)), int(bool(value_bf16)), *[ctypes.byref(x) for x in v])
        return dict(zip(("tw", "th", "lds_bytes", "nbuf", "depth_in_lds"), (x.value for x in v)))

    def bin_pairs(self, ref_cam, pc, H, W, bin_w, bin_h):
        """Reorders every camera's visible pairs by the feature pixel of their reference point (``sgc_bin_pairs``).
        ``pc``: the dict of ``compact_pairs``; returns a new dict with ``pair_q`` replaced, ``slot`` REWRITTEN IN PLACE
        to the new pair indices and two more entries, ``pair_ref`` [cap,4] fp32 and ``bin_offset`` [N*nb+1] int32.
        The pair counts stay on the device."""
        pair_cam, pair_q, cam_offset, slot = pc["pair_cam"], pc["pair_q"], pc["cam_offset"], pc["slot"]
        self._check(ref_cam=ref_cam, pair_cam=pair_cam, pair_q=pair_q, cam_offset=cam_offset, slot=slot)
        self._f32(ref_cam=ref_cam)
        self._i32(pair_cam=pair_cam, pair_q=pair_q, cam_offset=cam_offset, slot=slot)
        N, Nq, three = ref_cam.shape
        cap = pair_q.numel()
        if three != 3 or cam_offset.numel() != N + 1 or slot.shape != (N, Nq) or pair_cam.numel() != cap:
            raise RuntimeError("bin_pairs: inconsistent shapes")
        nb = -(-W // bin_w) * -(-H // bin_h)
        dev = ref_cam.device
        out = dict(pc)
        out["pair_q"] = torch.empty_like(pair_q)
        out["pair_ref"] = torch.empty((cap, 4), dtype=torch.float32, device=dev)
        out["bin_offset"] = torch.empty(N * nb + 1, dtype=torch.int32, device=dev)
        out["bin"] = (bin_w, bin_h)
        if cap == 0:
            out["bin_offset"].zero_()
            return out
        nbytes = int(self.lib._dll.sgc_bin_pairs_workspace_bytes(N, Nq, cap, H, W, bin_w, bin_h))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        self._call("sgc_bin_pairs", ref_cam, pair_cam, pair_q, cam_offset, out["pair_q"], slot, out["pair_ref"],
                   out["bin_offset"], ws, N, Nq, cap, H, W, bin_w, bin_h)
        return out

    def pairs_deform_gather_tiled(self, value_hm, dist, pair_ref, bin_offset, raw_hm, H, W, P, bin_w, bin_h, halo_x,
                                  halo_y, head_shift=None, max_shift=(0, 0), depth_in_lds=True, out=None):
        """value_hm [N,M,S,Cm]; dist [N,S,D]; raw_hm [cap, M*P*4] head-major (du, dv, dz, logit) per point, rows in
        the binned pair order; returns out [cap, M*Cm] (rows past the pair count are not written)."""
        self._check(value_hm=value_hm, dist=dist, pair_ref=pair_ref, bin_offset=bin_offset, raw_hm=raw_hm,
                    head_shift=head_shift, out=out)
        if value_hm.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError("pairs_deform_gather_tiled: value_hm must be float32 or bfloat16 (bf16 storage mode)")
        if dist.dtype != value_hm.dtype:     # the storage mode covers BOTH maps (sgcdet_amd.h, ABI version 4)
            raise RuntimeError("pairs_deform_gather_tiled: dist must have value_hm's dtype (float32, or bfloat16 in the storage mode)")
        self._f32(pair_ref=pair_ref, raw_hm=raw_hm, out=out)
        self._i32(bin_offset=bin_offset, head_shift=head_shift)
        N, M, S, Cm = value_hm.shape
        D = dist.shape[-1]
        nb = -(-W // bin_w) * -(-H // bin_h)
        if S < H * W or dist.shape[:2] != (N, S) or raw_hm.shape[-1] != M * P * 4 or bin_offset.numel() != N * nb + 1 \
                or (head_shift is not None and head_shift.numel() != 2 * M) or pair_ref.shape[0] < raw_hm.shape[0]:
            raise RuntimeError("pairs_deform_gather_tiled: inconsistent shapes")
        rows = raw_hm.shape[0]
        if out is None:
            out = torch.empty((rows, M * Cm), dtype=torch.float32, device=value_hm.device)
        if rows == 0:
            return out
        self._call("sgc_pairs_deform_gather_tiled", value_hm, int(value_hm.dtype == torch.bfloat16), dist, pair_ref,
                   bin_offset, raw_hm, head_shift, out,
                   N, H, W, M, Cm, D, P, S, bin_w, bin_h, halo_x, halo_y, int(max_shift[0]), int(max_shift[1]),
                   int(bool(depth_in_lds)),
                   _meta=dict(N=N, H=H, W=W, C=M * Cm, D=D, M=M, P=P, n_pairs=rows, bin=(bin_w, bin_h), halo=(halo_x, halo_y),
                              value_bytes=value_hm.element_size(), depth_bytes=dist.element_size()))
        return out

    def linear_rows_headmajor_bf16x3(self, x, w_hi, w_lo, shift, N, S, M, out_dtype=torch.float32):
        """x [N*S, Cin] -> y [N, M, S, Cm] head-major (``sgc_linear_rows_headmajor_bf16x3``); ``out_dtype`` bfloat16 = the
        opt-in bf16 storage mode (RNE of the fp32 result)."""
        self._check(x=x, w_hi=w_hi, w_lo=w_lo, shift=shift)
        self._f32(x=x, shift=shift)
        rows, Cin = x.shape
        Cout = w_hi.shape[-2]
        if rows != N * S or Cout % M or w_hi.shape[-1] != Cin or w_hi.dtype != torch.bfloat16 or w_lo.dtype != torch.bfloat16:
            raise RuntimeError("linear_rows_headmajor_bf16x3: inconsistent shapes")
        Cm = Cout // M
        if out_dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError("linear_rows_headmajor_bf16x3: out_dtype must be float32 or bfloat16")
        y = torch.empty((N, M, S, Cm), dtype=out_dtype, device=x.device)
        self._call("sgc_linear_rows_headmajor_bf16x3", x, w_hi, w_lo, shift, y, int(out_dtype == torch.bfloat16), N, S, Cin, M, Cm,
                   _meta=dict(V=rows, Cin=Cin, Cout=Cout, taps=1, OV=rows))
        return y

    # ---- 5. inter-view aggregation ------------------------------------------
    def view_mean(self, feat, slot, valid_index, n_valid, count=None):
        """``count``: optional int32 device tensor holding the live row count (an element of compact_pairs'
        totals); ``n_valid`` is then the capacity -- the result has n_valid rows, the first count of them written."""
        self._check(feat=feat, slot=slot, valid_index=valid_index, count=count)
        self._f32(feat=feat)
        self._i32(slot=slot, valid_index=valid_index, count=count)
        N, Nq = slot.shape
        Cc = feat.shape[1]
        mean = torch.empty((n_valid, Cc), dtype=torch.float32, device=feat.device)
        if n_valid == 0:
            return mean
        self._call("sgc_view_mean", feat, slot, valid_index, mean, N, Nq, Cc, count, n_valid)
        return mean

    def view_attend_backward(self, q, kv, slot, valid_index, heads, ctx, grad_ctx):
        """(grad_q [n_valid, C], grad_kv [n_pairs, 2C]) of ``view_attend`` (``sgc_view_attend_backward``)."""
        self._check(q=q, kv=kv, slot=slot, valid_index=valid_index, ctx=ctx, grad_ctx=grad_ctx)
        self._f32(q=q, kv=kv, ctx=ctx, grad_ctx=grad_ctx)
        self._i32(slot=slot, valid_index=valid_index)
        N, Nq = slot.shape
        n_valid, Cc = q.shape
        gq = torch.zeros_like(q)
        gkv = torch.zeros_like(kv)
        if n_valid:
            self._call("sgc_view_attend_backward", q, kv, slot, valid_index, ctx, grad_ctx, gq, gkv, N, Nq, Cc, heads, n_valid)
        return gq, gkv

    def view_attend(self, q, kv, slot, valid_index, heads, count=None):
        self._check(q=q, kv=kv, slot=slot, valid_index=valid_index, count=count)
        self._f32(q=q, kv=kv)
        self._i32(slot=slot, valid_index=valid_index, count=count)
        N, Nq = slot.shape
        n_valid, Cc = q.shape
        if kv.shape[1] != 2 * Cc:
            raise RuntimeError("view_attend: kv must be [n_pairs, 2C]")
        ctx = torch.empty_like(q)
        if n_valid == 0:
            return ctx
        self._call("sgc_view_attend", q, kv, slot, valid_index, ctx, N, Nq, Cc, heads, count, n_valid)
        return ctx

    def view_attend_pq_supported(self, N, C, heads):
        return bool(self.lib._dll.sgc_view_attend_pq_supported(int(N), int(C), int(heads)))

    def view_attend_pq(self, qp, x, slot, valid_index, heads, count=None):
        """Projected-query form of the softmax over views (``sgc_view_attend_pq``): qp [n_valid, heads * C] (= scale W_k,h^T q_h
        per head), x [n_pairs, C] raw per-pair features, slot [N, Nq], valid_index [n_valid] -> s [n_valid, heads * C], the
        attention-weighted pair feature of every head (the caller applies V per voxel)."""
        self._check(qp=qp, x=x, slot=slot, valid_index=valid_index, count=count)
        self._f32(qp=qp, x=x)
        self._i32(slot=slot, valid_index=valid_index, count=count)
        N, Nq = slot.shape
        n_valid, HC = qp.shape
        Cc = x.shape[1]
        if HC != heads * Cc:
            raise RuntimeError("view_attend_pq: qp must be [n_valid, heads * C]")
        s = torch.empty_like(qp)
        if n_valid == 0:
            return s
        self._call("sgc_view_attend_pq", qp, x, slot, valid_index, s, N, Nq, Cc, heads, count, n_valid)
        return s

    # ---- 6. volume glue --------------------------------------------------------
    def scatter_rows(self, rows, idx, vol, idx2=None, count=None):
        self._check(rows=rows, idx=idx, vol=vol, idx2=idx2, count=count)
        self._f32(rows=rows, vol=vol)
        self._i32(idx=idx, idx2=idx2, count=count)
        n, Cc = rows.shape
        if vol.shape[-1] != Cc:
            raise RuntimeError("scatter_rows: channel mismatch")
        if n == 0:
            return vol
        self._call("sgc_scatter_rows", rows, idx, idx2, vol, count, n, Cc)
        return vol

    def nchw_to_nhwc_crop(self, src, H, W):
        """``src`` [N,C,Hs,Ws], a top-left crop *view* of it (the reference crops with ``x[..., :height, :width]``,
        AdaptiveSparseHead.py:58-59) or an every-``step``-th-pixel view of it (``x[..., ::2, ::2]``: the nearest x1/2,
        x1/4 copies of the depth distribution, SGCDet.py:83-85 -- read in place, never materialised); returns [N, H*W, C]."""
        if src.dim() != 4:
            raise RuntimeError("nchw_to_nhwc_crop expects [N,C,H,W]")
        N, Cc, h_in, w_in = src.shape
        if H > h_in or W > w_in:
            raise RuntimeError("nchw_to_nhwc_crop: crop larger than the map")
        st = src.stride()
        step = st[3] if w_in > 1 else 1
        viewable = False
        if step >= 1 and h_in > 1 and st[2] % step == 0:
            Ws = st[2] // step
            Hs = st[1] // Ws if Ws > 0 and st[1] % Ws == 0 else 0
            viewable = (Ws >= (w_in - 1) * step + 1 and Hs >= (h_in - 1) * step + 1 and st[1] == Hs * Ws
                        and (N == 1 or st[0] == Cc * Hs * Ws))
        if not viewable:
            src = src.contiguous()
            Hs, Ws, step = h_in, w_in, 1
        if src.device.type != self.device_type:
            raise RuntimeError(f"src must be a {self.device_type} tensor")
        self._f32(src=src)
        dst = torch.empty((N, H * W, Cc), dtype=torch.float32, device=src.device)
        self._call("sgc_nchw_to_nhwc_crop", src, dst, N, Cc, Hs, Ws, H, W, step)
        return dst

    def nhwc_to_nchw_pad(self, rows, H, W, Hd=None, Wd=None):
        """rows [N, H*W, C] -> [N, C, Hd, Wd] (contiguous NCHW; zero outside the H x W crop): the adjoint of
        ``nchw_to_nhwc_crop`` (``sgc_nhwc_to_nchw_pad``)."""
        self._check(rows=rows)
        self._f32(rows=rows)
        N, S, Cc = rows.shape
        Hd, Wd = H if Hd is None else Hd, W if Wd is None else Wd
        if S != H * W or Hd < H or Wd < W:
            raise RuntimeError("nhwc_to_nchw_pad: inconsistent shapes")
        dst = torch.empty((N, Cc, Hd, Wd), dtype=torch.float32, device=rows.device)
        self._call("sgc_nhwc_to_nchw_pad", rows, dst, N, Cc, H, W, Hd, Wd)
        return dst

    # ---- 7. channels-last 3D convolution -----------------------------------------------
    @staticmethod
    def split_bf16(w):
        """fp32 -> (hi, lo) bf16 with hi = bf16_rne(w), lo = bf16_rne(w - hi)."""
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        return hi.contiguous(), lo.contiguous()

    @staticmethod
    def split_f16(w):
        """fp32 -> the operand planes of the fp16 mode (``sgc_set_conv_products(2)``): hi = the IEEE-half bits of w (rounded to
        nearest even, saturated at +-65504) in a bfloat16-TYPED container (the planes are 16-bit patterns to the library), lo =
        zeros (ignored in the one-product modes)."""
        hi = w.clamp(-65504.0, 65504.0).to(torch.float16).view(torch.bfloat16)
        return hi.contiguous(), torch.zeros_like(hi)

    def split_operand(self, w):
        """Weight planes for the library's CURRENT arithmetic mode (``sgc_get_conv_products``): the bf16 hi / lo split for 3 and 1,
        half bits for 2.  Prepared plans must be rebuilt after a mode switch (plugin/conv_plan.set_conv_mode does not do it)."""
        return self.split_f16(w) if self.lib._dll.sgc_get_conv_products() == 2 else self.split_bf16(w)

    def conv3d_cl_bf16x3(self, x, w_hi, w_lo, grid, ksize, stride=1, transposed=False, scale=None, shift=None,
                         residual=None, relu=False, out=None, out_mask=None, act=None):
        """As ``conv3d_cl`` with pre-split bf16 weights (see ``split_bf16``).  ``out_mask`` (uint8 [OV], 3x3x3 stride-1
        layers only): rows with 0 are not needed by the caller (``sgc_conv3d_cl_bf16x3_masked``).  ``act`` = (c0, c1, scale
        tensor on the device): columns [c0, c1) leave as exp(v * scale) (``sgc_conv3d_cl_bf16x3_act``, 3x3x3 stride 1)."""
        self._check(x=x, w_hi=w_hi, w_lo=w_lo, scale=scale, shift=shift, residual=residual, out_mask=out_mask)
        self._f32(x=x, scale=scale, shift=shift, residual=residual)
        if w_hi.dtype != torch.bfloat16 or w_lo.dtype != torch.bfloat16 or w_hi.shape != w_lo.shape:
            raise RuntimeError("conv3d_cl_bf16x3: w_hi / w_lo must be bfloat16 tensors of one shape")
        ix, iy, iz = grid
        V, Cin = x.shape
        taps, Cout, Cin2 = w_hi.shape
        if V != ix * iy * iz or Cin2 != Cin or taps != (8 if transposed else ksize ** 3):
            raise RuntimeError("conv3d_cl_bf16x3: inconsistent shapes")
        pad = 0 if ksize == 2 else ksize // 2              # ksize 2 (stride 2): the adjoint geometry of ConvTranspose3d(2, 2)
        og = (2 * ix, 2 * iy, 2 * iz) if transposed else tuple((d + 2 * pad - ksize) // stride + 1 for d in grid)
        if out is not None:
            self._check(out=out)
            if out.shape != (og[0] * og[1] * og[2], Cout) or out.dtype != torch.float32:
                raise RuntimeError("conv3d_cl_bf16x3: bad `out` tensor")
        y = out if out is not None else torch.empty((og[0] * og[1] * og[2], Cout), dtype=torch.float32, device=x.device)
        if residual is not None and residual.shape != y.shape:
            raise RuntimeError("conv3d_cl_bf16x3: residual shape mismatch")
        ws, ws_n = self._conv_workspace(x.device, ix, iy, iz, Cin, Cout, ksize, stride, transposed, 1)
        if act is not None:
            c0, c1, act_scale = act
            self._check(act_scale=act_scale)
            self._f32(act_scale=act_scale)
            if transposed or ksize != 3 or stride != 1 or act_scale.numel() != 1:
                raise RuntimeError("conv3d_cl_bf16x3: `act` needs a 3x3x3 stride-1 layer and a one-element scale tensor")
            if out_mask is not None and (out_mask.dtype != torch.uint8 or out_mask.numel() != y.shape[0]):
                raise RuntimeError("conv3d_cl_bf16x3: out_mask must be a uint8 mask of [OV]")
            self._call("sgc_conv3d_cl_bf16x3_act", x, w_hi, w_lo, scale, shift, residual, y, out_mask, ix, iy, iz, Cin, Cout,
                       int(relu), int(c0), int(c1), act_scale, ws, ws_n,
                       _meta=dict(V=V, Cin=Cin, Cout=Cout, taps=taps, OV=y.shape[0], masked=out_mask is not None))
            return y, og
        if out_mask is not None:
            if transposed or ksize != 3 or stride != 1 or out_mask.dtype != torch.uint8 or out_mask.numel() != y.shape[0]:
                raise RuntimeError("conv3d_cl_bf16x3: out_mask needs a 3x3x3 stride-1 layer and a uint8 mask of [OV]")
            self._call("sgc_conv3d_cl_bf16x3_masked", x, w_hi, w_lo, scale, shift, residual, y, out_mask, ix, iy, iz, Cin,
                       Cout, int(relu), ws, ws_n, _meta=dict(V=V, Cin=Cin, Cout=Cout, taps=taps, OV=y.shape[0], masked=True))
            return y, og
        self._call("sgc_conv3d_cl_bf16x3", x, w_hi, w_lo, scale, shift, residual, y, ix, iy, iz, Cin, Cout, ksize,
                   stride, 1 if transposed else 0, int(relu), ws, ws_n,
                   _meta=dict(V=V, Cin=Cin, Cout=Cout, taps=taps, OV=y.shape[0], flop_taps=1 if transposed else taps))
        return y, og

    @staticmethod
    def winograd_z_weights(wt):
        """[27, Cout, Cin] fp32 (tap = (dx*3 + dy)*3 + dz) -> [4, 9, Cout, Cin] fp32: the F(2,3)-along-z transform of the z taps
        (G0 = w0, G1 = (w0 + w1 + w2) / 2, G2 = (w0 - w1 + w2) / 2, G3 = w2), composed in float64."""
        t, cout, cin = wt.shape
        if t != 27:
            raise RuntimeError("winograd_z_weights: a 3x3x3 kernel expected")
        w = wt.double().view(9, 3, cout, cin)
        g = torch.stack([w[:, 0], (w[:, 0] + w[:, 1] + w[:, 2]) / 2, (w[:, 0] - w[:, 1] + w[:, 2]) / 2, w[:, 2]])
        return g.float().contiguous()

    def conv3d_winograd_z_supported(self, grid, Cin, Cout):
        return bool(self.lib._dll.sgc_conv3d_winograd_z_supported(int(grid[0]), int(grid[1]), int(grid[2]), int(Cin), int(Cout)))

    def conv3d_winograd_z(self, x, g_hi, g_lo, grid, scale=None, shift=None, residual=None, relu=False, out=None):
        """3x3x3 stride-1 convolution through the Winograd F(2,3) transform along z (``sgc_conv3d_winograd_z_bf16x3``):
        g_hi / g_lo = the split of ``winograd_z_weights`` ([4, 9, Cout, Cin] bf16).  Same operator and epilogue as
        ``conv3d_cl_bf16x3(ksize=3)``; 2/3 of its multiply-adds."""
        self._check(x=x, g_hi=g_hi, g_lo=g_lo, scale=scale, shift=shift, residual=residual, out=out)
        self._f32(x=x, scale=scale, shift=shift, residual=residual, out=out)
        if g_hi.dtype != torch.bfloat16 or g_lo.dtype != torch.bfloat16 or g_hi.shape != g_lo.shape or g_hi.dim() != 4:
            raise RuntimeError("conv3d_winograd_z: g_hi / g_lo must be bfloat16 [4, 9, Cout, Cin]")
        ix, iy, iz = grid
        V, Cin = x.shape
        _, _, Cout, Cin2 = g_hi.shape
        if V != ix * iy * iz or Cin2 != Cin or g_hi.shape[:2] != (4, 9):
            raise RuntimeError("conv3d_winograd_z: inconsistent shapes")
        y = out if out is not None else torch.empty((V, Cout), dtype=torch.float32, device=x.device)
        if y.shape != (V, Cout) or (residual is not None and residual.shape != y.shape):
            raise RuntimeError("conv3d_winograd_z: bad `out` / residual shape")
        n = int(self.lib._dll.sgc_conv3d_winograd_z_workspace_floats(ix, iy, iz, Cin, Cout))
        ws = torch.empty(max(n, 4), dtype=torch.float32, device=x.device)
        self._call("sgc_conv3d_winograd_z_bf16x3", x, g_hi, g_lo, scale, shift, residual, y, ix, iy, iz, Cin, Cout, int(relu), ws, n,
                   _meta=dict(V=V, Cin=Cin, Cout=Cout, taps=27, OV=V, mac_frac=2.0 / 3.0))
        return y, tuple(grid)

    @staticmethod
    def pack_b_fragments(w):
        """[..., N, K] (N % 32 == 0, K % 16 == 0) -> the fragment-packed layout [N/32, K/16, 64, 8] of ``sgc_level_tail``:
        packed[b, kk, l, j] = w[32 b + (l & 31), 16 kk + 8 (l >> 5) + j]."""
        N, K = w.shape[-2], w.shape[-1]
        if N % 32 or K % 16:
            raise RuntimeError("pack_b_fragments: needs N % 32 == 0 and K % 16 == 0")
        return w.reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(N // 32, K // 16, 64, 8)

    def level_tail_supported(self, C, F):
        return bool(self.lib._dll.sgc_level_tail_supported(int(C), int(F)))

    def level_tail(self, ctx, row_of, wo, bo, ln1, w1, b1, w2, b2, ln2, out=None):
        """The tail of a VoxFormer level in one launch (``sgc_level_tail``): out_proj on the seen voxels (zero rows
        elsewhere) -> LayerNorm -> FFN with identity -> LayerNorm.  ``wo`` / ``w1`` / ``w2``: (hi, lo) bf16 pairs of
        ``split_bf16`` ([1, N, K] or [N, K]); they are fragment-packed here (``pack_b_fragments``) unless the caller passes
        already packed 4-D tensors (a module packs once and caches).  ``ln1`` / ``ln2``: (gamma, beta, eps).  Returns [Nq, C]."""
        self._check(ctx=ctx, row_of=row_of, bo=bo, b1=b1, b2=b2, out=out)
        self._f32(ctx=ctx, bo=bo, b1=b1, b2=b2, out=out)
        self._i32(row_of=row_of)
        Nq, C = row_of.shape[0], ctx.shape[1]
        def packed(name, pair, n, k):
            hi, lo = pair
            if hi.dtype != torch.bfloat16 or lo.dtype != torch.bfloat16 or hi.shape != lo.shape:
                raise RuntimeError(f"level_tail: {name} must be a bfloat16 (hi, lo) pair")
            if hi.dim() == 4 and tuple(hi.shape) == (n // 32, k // 16, 64, 8):
                return hi, lo
            if tuple(hi.shape[-2:]) != (n, k) or hi.numel() != n * k:
                raise RuntimeError(f"level_tail: {name} must be [{n}, {k}] (or fragment-packed)")
            return self.pack_b_fragments(hi.reshape(n, k)), self.pack_b_fragments(lo.reshape(n, k))
        F = b1.shape[0]
        wo, w1, w2 = packed("wo", wo, C, C), packed("w1", w1, F, C), packed("w2", w2, C, F)
        y = out if out is not None else torch.empty((Nq, C), dtype=torch.float32, device=ctx.device)
        if y.shape != (Nq, C):
            raise RuntimeError("level_tail: bad `out` tensor")
        self._call("sgc_level_tail", ctx, row_of, wo[0], wo[1], bo, ln1[0], ln1[1], float(ln1[2]), w1[0], w1[1], b1, w2[0], w2[1],
                   b2, ln2[0], ln2[1], float(ln2[2]), y, Nq, C, F, _meta=dict(V=Nq, Cin=C, Cout=5 * C, taps=1, OV=Nq))
        return y

    def linear_rows_bf16x3(self, x, w_hi, w_lo, shift=None, count=None, out=None, useful=None, zero_tail=False):
        """y[r] = x[r] @ W^T + shift for the first ``count`` rows (int32 device tensor; None = all rows) of
        x [rows_cap, Cin]; W as the bf16 split [1, Cout, Cin] of ``split_bf16``.  Rows past the count are left
        untouched (uninitialised in a fresh result).  ``zero_tail``: the result is a view of a [rows + 1, Cout] buffer whose last
        row the same launch sets to zero (``sgc_linear_rows_zrow_bf16x3``)."""
        self._check(x=x, w_hi=w_hi, w_lo=w_lo, shift=shift, count=count, out=out)
        self._f32(x=x, shift=shift, out=out)
        self._i32(count=count)
        if w_hi.dtype != torch.bfloat16 or w_lo.dtype != torch.bfloat16 or w_hi.shape != w_lo.shape:
            raise RuntimeError("linear_rows_bf16x3: w_hi / w_lo must be bfloat16 tensors of one shape")
        rows, Cin = x.shape
        Cout = w_hi.shape[-2]
        if w_hi.shape[-1] != Cin or w_hi.numel() != Cout * Cin:
            raise RuntimeError("linear_rows_bf16x3: inconsistent shapes")
        if out is not None and (out.shape != (rows, Cout)):
            raise RuntimeError("linear_rows_bf16x3: bad `out` tensor")
        if zero_tail:
            if out is not None or rows == 0:
                raise RuntimeError("linear_rows_bf16x3: zero_tail allocates its own [rows + 1, Cout] buffer (rows > 0)")
            buf = torch.empty((rows + 1, Cout), dtype=torch.float32, device=x.device)
            self._call("sgc_linear_rows_zrow_bf16x3", x, w_hi, w_lo, shift, buf, count, rows, Cin, Cout,
                       _meta=dict(V=rows, Cin=Cin, Cout=Cout, taps=1, OV=rows))
            return buf[:rows]
        y = out if out is not None else torch.empty((rows, Cout), dtype=torch.float32, device=x.device)
        if rows:
            meta = dict(V=rows, Cin=Cin, Cout=Cout, taps=1, OV=rows)
            if useful is not None:
                meta["useful"] = useful           # block-diagonal weights run as a dense GEMM: the non-zero fraction (flop accounting)
            self._call("sgc_linear_rows_bf16x3", x, w_hi, w_lo, shift, y, count, rows, Cin, Cout, _meta=meta)
        return y

    def linear_rows_blockdiag_supported(self, G, K, Nh):
        return bool(self.lib._dll.sgc_linear_rows_blockdiag_supported(int(G), int(K), int(Nh)))

    def linear_rows_blockdiag(self, x, w_hi, w_lo, shift=None, count=None):
        """y[r, g * Nh + j] = x[r, g * K : (g + 1) * K] @ w[g, j] + shift[g * Nh + j] for the first ``count`` rows: x [rows_cap, G * K],
        w_hi / w_lo [G, Nh, K] (the operand split of the current arithmetic mode) -> [rows_cap, G * Nh]
        (``sgc_linear_rows_blockdiag_bf16x3``: the V projection of the projected-query attention, head by head)."""
        self._check(x=x, w_hi=w_hi, w_lo=w_lo, shift=shift, count=count)
        self._f32(x=x, shift=shift)
        self._i32(count=count)
        if w_hi.dtype != torch.bfloat16 or w_lo.dtype != torch.bfloat16 or w_hi.shape != w_lo.shape or w_hi.dim() != 3:
            raise RuntimeError("linear_rows_blockdiag: w_hi / w_lo must be bfloat16 tensors [G, Nh, K] of one shape")
        G, Nh, K = w_hi.shape
        rows = x.shape[0]
        if x.dim() != 2 or x.shape[1] != G * K:
            raise RuntimeError("linear_rows_blockdiag: x must be [rows, G * K]")
        y = torch.empty((rows, G * Nh), dtype=torch.float32, device=x.device)
        if rows:
            self._call("sgc_linear_rows_blockdiag_bf16x3", x, w_hi, w_lo, shift, y, count, rows, G, K, Nh,
                       _meta=dict(V=rows, Cin=K, Cout=G * Nh, taps=1, OV=rows))     # 2 K Nh multiply-adds per row and group: the algorithmic count
        return y

    def conv2d_nhwc_bf16x3(self, x, w_hi, w_lo, nhw, ksize, scale=None, shift=None, residual=None, relu=False, out=None):
        """2-D convolution (k in {1,3}, padding k//2, stride 1) over channels-last image rows: x [N*H*W, Cin] ->
        [N*H*W, Cout]; weights [k*k, Cout, Cin] split as ``split_bf16`` (``sgc_conv2d_nhwc_bf16x3``)."""
        self._check(x=x, w_hi=w_hi, w_lo=w_lo, scale=scale, shift=shift, residual=residual, out=out)
        self._f32(x=x, scale=scale, shift=shift, residual=residual, out=out)
        if w_hi.dtype != torch.bfloat16 or w_lo.dtype != torch.bfloat16 or w_hi.shape != w_lo.shape:
            raise RuntimeError("conv2d_nhwc_bf16x3: w_hi / w_lo must be bfloat16 tensors of one shape")
        N, H, W = nhw
        rows, Cin = x.shape
        taps, Cout, Cin2 = w_hi.shape
        if rows != N * H * W or Cin2 != Cin or taps != ksize * ksize:
            raise RuntimeError("conv2d_nhwc_bf16x3: inconsistent shapes")
        y = out if out is not None else torch.empty((rows, Cout), dtype=torch.float32, device=x.device)
        if y.shape != (rows, Cout) or (residual is not None and residual.shape != y.shape):
            raise RuntimeError("conv2d_nhwc_bf16x3: bad `out` / residual shape")
        self._call("sgc_conv2d_nhwc_bf16x3", x, w_hi, w_lo, scale, shift, residual, y, N, H, W, Cin, Cout, ksize, int(relu),
                   _meta=dict(V=rows, Cin=Cin, Cout=Cout, taps=taps, OV=rows))
        return y

    def pack_conv_weight(self, w, transpose=False, flip=False, pad_rows=1, pad_cols=1, out=None):
        """Module parameter [A, B, *taps] (Conv3d [Cout, Cin, k, k, k], ConvTranspose3d [Cin, Cout, 2, 2, 2], Linear [Cout, Cin])
        -> (hi, lo) bf16 [T, R, C] in the kernels' layout: rows = A (or B with ``transpose``), taps mirrored with ``flip``,
        R / C zero-padded to multiples of ``pad_rows`` / ``pad_cols`` (``sgc_pack_conv_weight``).  ``out`` = (hi, lo) to fill."""
        self._check(w=w)
        self._f32(w=w)
        w = w.contiguous()
        T, R, Cc = self.packed_shape(w.shape, transpose, pad_rows, pad_cols)
        A, B = w.shape[0], w.shape[1]
        if out is None:
            hi = torch.empty((T, R, Cc), dtype=torch.bfloat16, device=w.device)
            lo = torch.empty_like(hi)
        else:
            hi, lo = out
            if any(t.shape != (T, R, Cc) or t.dtype != torch.bfloat16 or not t.is_contiguous() or t.device != w.device for t in out):
                raise RuntimeError("pack_conv_weight: `out` must be two contiguous bfloat16 [T, R, C] tensors on the parameter's device")
        self._call("sgc_pack_conv_weight", w, hi, lo, A, B, T, R, Cc, int(bool(transpose)), int(bool(flip)))
        return hi, lo

    @staticmethod
    def packed_shape(wshape, transpose=False, pad_rows=1, pad_cols=1):
        """(T, R, C) of the planes ``pack_conv_weight`` makes of a parameter of shape ``wshape``."""
        A, B = wshape[0], wshape[1]
        T = 1
        for d in wshape[2:]:
            T *= d
        rows, cols = (B, A) if transpose else (A, B)
        return T, -(-rows // pad_rows) * pad_rows, -(-cols // pad_cols) * pad_cols

    def pack_conv_weight_plan(self, entries):
        """The launch plan of ``sgc_pack_conv_weight_batch`` for ``entries`` = [(w, hi, lo, transpose, flip), ...] (the planes keep
        their zero padding: allocate them with ``torch.zeros``): the device item list + counts, valid for as long as those tensors
        stay where they are.  ``run_pack_plan(plan)`` repacks all of them in ONE launch."""
        import struct
        if not entries:
            return None
        total, max_t, blob = 0, 1, b""
        for w, hi, lo, transpose, flip in entries:
            self._check(w=w, hi=hi, lo=lo)
            self._f32(w=w)
            A, B = w.shape[0], w.shape[1]
            T, R, Cc = hi.shape
            if (not w.is_contiguous() or w.numel() != A * B * T or hi.shape != lo.shape or hi.dtype != torch.bfloat16 or lo.dtype != torch.bfloat16
                    or R < (B if transpose else A) or Cc < (A if transpose else B) or not hi.is_contiguous() or not lo.is_contiguous()):
                raise RuntimeError("pack_conv_weight_plan: inconsistent entry")
            nb = int(self.lib._dll.sgc_pack_conv_weight_blocks(A, B, T, int(bool(transpose))))
            blob += struct.pack("<QQQ10i", w.data_ptr(), hi.data_ptr(), lo.data_ptr(), A, B, T, R, Cc, int(bool(transpose)), int(bool(flip)), total, 0, 0)
            total += nb
            max_t = max(max_t, T)
        items = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(entries[0][0].device)
        return items, len(entries), total, max_t

    def run_pack_plan(self, plan):
        if plan is not None:
            self._call("sgc_pack_conv_weight_batch", *plan)

    def unpack_conv_wgrad(self, dw_trc, shape, transpose=False, flip=False):
        """[T, R, C] fp32 (``conv3d_wgrad_bf16x3``) -> the parameter's layout ``shape`` = [A, B, *taps] (``sgc_unpack_conv_wgrad``)."""
        self._check(dw_trc=dw_trc)
        self._f32(dw_trc=dw_trc)
        T, R, Cc = dw_trc.shape
        A, B = shape[0], shape[1]
        out = torch.empty(tuple(shape), dtype=torch.float32, device=dw_trc.device)
        if out.numel() != A * B * T:
            raise RuntimeError("unpack_conv_wgrad: tap count mismatch")
        self._call("sgc_unpack_conv_wgrad", dw_trc.contiguous(), out, A, B, T, R, Cc, int(bool(transpose)), int(bool(flip)))
        return out

    def conv3d_wgrad_bf16x3(self, x, dy, grid, ksize, stride=1):
        """dW [ksize^3, Cout, Cin] of the channels-last convolution: x [IV, Cin] on ``grid``, dy [OV, Cout] on the output
        grid (``sgc_conv3d_wgrad_bf16x3``; ksize 2 = stride 2, no padding)."""
        self._check(x=x, dy=dy)
        self._f32(x=x, dy=dy)
        ix, iy, iz = grid
        pad = 0 if ksize == 2 else ksize // 2
        og = tuple((d + 2 * pad - ksize) // stride + 1 for d in grid)
        V, Cin = x.shape
        OV, Cout = dy.shape
        if V != ix * iy * iz or OV != og[0] * og[1] * og[2]:
            raise RuntimeError("conv3d_wgrad_bf16x3: inconsistent shapes")
        dw = torch.empty((ksize ** 3, Cout, Cin), dtype=torch.float32, device=x.device)
        n = int(self.lib._dll.sgc_conv3d_wgrad_workspace_floats(ix, iy, iz, Cin, Cout, ksize, stride))
        if n < 0:
            raise RuntimeError("conv3d_wgrad_bf16x3: " + self.lib.last_error())
        ws = torch.empty(n, dtype=torch.float32, device=x.device) if n > 0 else None
        self._call("sgc_conv3d_wgrad_bf16x3", x, dy, dw, ix, iy, iz, Cin, Cout, ksize, stride, ws, n,
                   _meta=dict(V=OV, Cin=Cin, Cout=Cout, taps=ksize ** 3, OV=OV))
        return dw

    def _conv_workspace(self, device, ix, iy, iz, Cin, Cout, ksize, stride, transposed, bf16x3):
        """Split-K layers get a workspace so that their partial sums are added in a fixed order (bit-identical
        results from run to run); (None, 0) for layers that are not split."""
        n = int(self.lib._dll.sgc_conv3d_workspace_floats(ix, iy, iz, Cin, Cout, ksize, stride, 1 if transposed else 0, bf16x3))
        if n <= 0:
            return None, 0
        return torch.empty(n, dtype=torch.float32, device=device), n

    # ---- 7b. row-wise glue ---------------------------------------------------------------------
    def topk_select(self, score, k, want_valid=False, want_mask=False):
        """score [n] (or [1,n]) fp32 -> (idx [k] int64 ascending, valid [n] int64 | None, mask [n] fp32 | None): the k
        largest scores, ties at the cut broken by the lowest index (``sgc_topk_select``)."""
        self._check(score=score)
        self._f32(score=score)
        flat = score.reshape(-1)
        n = flat.numel()
        if not 0 < k <= n:
            raise RuntimeError(f"topk_select: need 0 < k <= n (k = {k}, n = {n})")
        dev = score.device
        idx = torch.empty(k, dtype=torch.int64, device=dev)
        valid = torch.empty(n, dtype=torch.int64, device=dev) if want_valid else None
        mask = torch.empty(n, dtype=torch.float32, device=dev) if want_mask else None
        wsb = int(self.lib._dll.sgc_topk_select_workspace_bytes(n))
        ws = torch.empty(max(wsb, 4) // 4, dtype=torch.int32, device=dev) if wsb > 0 else None   # many-workgroup form for large n
        self._call("sgc_topk_select_ws", flat, n, int(k), idx, valid, mask, ws, wsb)
        return idx, valid, mask

    def bn_rows_forward(self, x, weight, bias, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, residual=None, relu=False):
        """Training-mode BatchNorm over rows x [rows, C] -> (y, mean [C], invstd [C]); the running statistics are updated in
        place as nn.BatchNorm does.  ``residual`` / ``relu``: y = relu(bn(x) + residual) in the same pass (``sgc_bn_rows_act_forward``)."""
        self._check(x=x, weight=weight, bias=bias, running_mean=running_mean, running_var=running_var, residual=residual)
        self._f32(x=x, weight=weight, bias=bias, running_mean=running_mean, running_var=running_var, residual=residual)
        rows, Cc = x.shape
        if weight.numel() != Cc or bias.numel() != Cc or (residual is not None and residual.shape != x.shape):
            raise RuntimeError("bn_rows_forward: inconsistent shapes")
        y = torch.empty_like(x)
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        n = int(self.lib._dll.sgc_bn_rows_workspace_floats(rows, Cc))
        ws = torch.empty(max(n, 4), dtype=torch.float32, device=x.device)
        if residual is None and not relu:
            self._call("sgc_bn_rows_forward", x, weight, bias, running_mean, running_var, float(momentum), float(eps), y, mean, invstd, ws,
                       ws.numel(), rows, Cc)
        else:
            self._call("sgc_bn_rows_act_forward", x, weight, bias, running_mean, running_var, float(momentum), float(eps), residual, int(bool(relu)),
                       y, mean, invstd, ws, ws.numel(), rows, Cc)
        return y, mean, invstd

    def bn_rows_backward(self, x, dy, mean, invstd, weight, y_relu=None, want_dresidual=False):
        """-> (dx [rows, C], dweight [C], dbias [C]) of ``bn_rows_forward``; with ``y_relu`` (the forward's output when it ended in a
        ReLU) and / or ``want_dresidual`` -> (dx, dweight, dbias, dresidual) through ``sgc_bn_rows_act_backward``."""
        self._check(x=x, dy=dy, mean=mean, invstd=invstd, weight=weight, y_relu=y_relu)
        self._f32(x=x, dy=dy, mean=mean, invstd=invstd, weight=weight, y_relu=y_relu)
        rows, Cc = x.shape
        if dy.shape != x.shape or mean.numel() != Cc or invstd.numel() != Cc or weight.numel() != Cc or (y_relu is not None and y_relu.shape != x.shape):
            raise RuntimeError("bn_rows_backward: inconsistent shapes")
        dx = torch.empty_like(x)
        dw = torch.empty(Cc, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dw)
        n = int(self.lib._dll.sgc_bn_rows_workspace_floats(rows, Cc))
        ws = torch.empty(max(n, 4), dtype=torch.float32, device=x.device)
        if y_relu is None and not want_dresidual:
            self._call("sgc_bn_rows_backward", x, dy, mean, invstd, weight, dx, dw, db, ws, ws.numel(), rows, Cc)
            return dx, dw, db
        dres = torch.empty_like(x) if want_dresidual else None
        self._call("sgc_bn_rows_act_backward", x, dy, y_relu, mean, invstd, weight, dx, dw, db, dres, ws, ws.numel(), rows, Cc)
        return dx, dw, db, dres

    def layer_norm_rows(self, x, gamma, beta, eps=1e-5, count=None, out=None):
        """nn.LayerNorm over the last dim of x [rows, C]; ``count``: int32 device tensor with the live row count."""
        self._check(x=x, gamma=gamma, beta=beta, count=count, out=out)
        self._f32(x=x, gamma=gamma, beta=beta, out=out)
        self._i32(count=count)
        rows, Cc = x.shape
        if gamma.numel() != Cc or beta.numel() != Cc:
            raise RuntimeError("layer_norm_rows: inconsistent shapes")
        y = out if out is not None else torch.empty_like(x)
        if rows:
            self._call("sgc_layer_norm_rows", x, gamma, beta, float(eps), y, count, rows, Cc)
        return y

    def mask_dilate3(self, mask, grid):
        """3x3x3 dilation of a uint8 {0,1} voxel mask [X*Y*Z]."""
        self._check(mask=mask)
        if mask.dtype != torch.uint8 or mask.numel() != grid[0] * grid[1] * grid[2]:
            raise RuntimeError("mask_dilate3: uint8 mask of [X*Y*Z] expected")
        out = torch.empty_like(mask)
        self._call("sgc_mask_dilate3", mask, out, *grid)
        return out

    def valid_pyramid(self, valid, grid, factor):
        """Head valid mask of scale ``factor``: nn.Upsample(trilinear)(valid.float()).round().bool() as uint8."""
        self._check(valid=valid)
        self._i64(valid=valid)
        X, Y, Z = grid
        if valid.numel() != X * Y * Z:
            raise RuntimeError("valid_pyramid: valid must hold X*Y*Z elements")
        out = torch.empty((X // factor) * (Y // factor) * (Z // factor), dtype=torch.uint8, device=valid.device)
        self._call("sgc_valid_pyramid", valid.reshape(-1), out, X, Y, Z, factor)
        return out

    # ---- 8. post-processing ----------------------------------------------------------------
    def aligned_nms3d(self, boxes, scores, labels, iou_thr):
        """mmdet3d ``aligned_3d_nms(boxes [n,6], scores [n], classes [n], thresh)`` -> kept indices (int64,
        descending score).  One host read-back (the number of kept boxes) -- this is post-processing."""
        self._check(boxes=boxes, scores=scores, labels=labels)
        self._f32(boxes=boxes, scores=scores)
        n = boxes.shape[0]
        if boxes.shape != (n, 6) or scores.shape != (n,) or labels.shape != (n,):
            raise RuntimeError("aligned_nms3d: boxes [n,6], scores [n], labels [n] expected")
        labels = labels.to(torch.int64)
        keep = torch.empty(n, dtype=torch.int64, device=boxes.device)
        n_keep = torch.zeros(1, dtype=torch.int32, device=boxes.device)
        if n == 0:
            return keep
        order = torch.argsort(scores)                              # the reference's call: ascending
        ws = torch.empty(n * ((n + 63) // 64), dtype=torch.int64, device=boxes.device)
        self._call("sgc_aligned_nms3d", boxes, order, labels, float(iou_thr), keep, n_keep, ws, n)
        return keep[: int(n_keep.item())]

    def nms_rotated_bev(self, boxes, scores, score_thr, iou_thr):
        """The class loop of mmdet3d ``box3d_multiclass_nms`` (box3d_nms.py:52-68) with ``use_rotate_nms``: for
        every class c, ``nms_bev(boxes[scores[:, c] > score_thr], ...)``.  boxes [K,5] (x1,y1,x2,y2,ry), scores
        [K,C].  Returns (keep [C,K] int64: kept box indices per class in descending score, n_keep [C] int32);
        nothing is read back to the host here."""
        self._check(boxes=boxes, scores=scores)
        self._f32(boxes=boxes, scores=scores)
        K, C = scores.shape
        if boxes.shape != (K, 5):
            raise RuntimeError("nms_rotated_bev: boxes [K,5] (x1,y1,x2,y2,ry) and scores [K,C] expected")
        keep = torch.zeros((C, K), dtype=torch.int64, device=boxes.device)
        n_keep = torch.zeros(C, dtype=torch.int32, device=boxes.device)
        if K == 0 or C == 0:
            return keep, n_keep
        st = scores.t()
        cand = st > score_thr
        counts = cand.sum(dim=1).to(torch.int32)
        # candidates first, by descending score (stable: ties keep ascending box index); the rest of a row is unused
        order = torch.where(cand, st, torch.full_like(st, float("-inf"))).sort(dim=1, descending=True, stable=True)[1]
        ws = torch.empty(C * K * ((K + 63) // 64), dtype=torch.int64, device=boxes.device)
        self._call("sgc_nms_rotated_bev", boxes, order.contiguous(), counts, float(iou_thr), keep, n_keep, ws, K, C)
        return keep, n_keep

    def box_iou_rotated(self, a, b):
        """mmcv ``box_iou_rotated(a [n,5], b [m,5])`` (xc, yc, w, h, radians) -> [n,m] IoU."""
        self._check(a=a, b=b)
        self._f32(a=a, b=b)
        n, m = a.shape[0], b.shape[0]
        if a.shape != (n, 5) or b.shape != (m, 5):
            raise RuntimeError("box_iou_rotated: [n,5] and [m,5] boxes expected")
        iou = torch.empty((n, m), dtype=torch.float32, device=a.device)
        self._call("sgc_box_iou_rotated", a, b, iou, n, m)
        return iou

    def assign_targets(self, points, scales, boxes, gt_labels, rotated, n_scales, limit, centerness_topk):
        """``ImVoxelHeadV2.get_targets`` (imvoxel_head_v2.py:361-435 / :485-561): points [n,3], scales [n] int32,
        boxes [n_boxes,7] (gravity centre, dims, yaw), gt_labels [n_boxes] int64 ->
        (centerness_targets [n], bbox_targets [n,6|7], labels [n] int64, geo_occ_box [n] bool)."""
        self._check(points=points, scales=scales, boxes=boxes, gt_labels=gt_labels)
        self._f32(points=points, boxes=boxes)
        self._i32(scales=scales)
        self._i64(gt_labels=gt_labels)
        n, nb = points.shape[0], boxes.shape[0]
        if points.shape != (n, 3) or scales.shape != (n,) or boxes.shape != (nb, 7) or gt_labels.shape != (nb,):
            raise RuntimeError("assign_targets: points [n,3], scales [n], boxes [n_boxes,7], gt_labels [n_boxes] expected")
        if nb == 0:
            raise RuntimeError("assign_targets: at least one ground-truth box is required (as the reference)")
        dev = points.device
        ct = torch.empty(n, dtype=torch.float32, device=dev)
        bt = torch.empty((n, 7 if rotated else 6), dtype=torch.float32, device=dev)
        lb = torch.empty(n, dtype=torch.int64, device=dev)
        occ = torch.empty(n, dtype=torch.uint8, device=dev)
        ws = torch.empty(nb * (n_scales + 2), dtype=torch.int32, device=dev)
        self._call("sgc_assign_targets", points, scales, boxes, gt_labels, int(bool(rotated)), int(n_scales), int(limit),
                   int(centerness_topk), ct, bt, lb, occ, ws, n, nb)
        return ct, bt, lb, occ.bool()

    # ---- 9. upstream: plane-sweep matching cost ------------------------------------------------
    def plane_sweep_corr(self, feat, nbr, rt, depth, H, W):
        """feat [N, H*W, C] channels-last; nbr [N,K] int32; rt [N,K,12]; depth [D] -> corr [N,D,H,W]."""
        self._check(feat=feat, nbr=nbr, rt=rt, depth=depth)
        self._f32(feat=feat, rt=rt, depth=depth)
        self._i32(nbr=nbr)
        N, S, Cc = feat.shape
        K = nbr.shape[1]
        if S != H * W or nbr.shape != (N, K) or rt.shape != (N, K, 12):
            raise RuntimeError("plane_sweep_corr: inconsistent shapes")
        D = depth.numel()
        corr = torch.empty((N, D, H, W), dtype=torch.float32, device=feat.device)
        self._call("sgc_plane_sweep_corr", feat, nbr, rt, depth, corr, N, K, H, W, Cc, D)
        return corr

    def conv3d_cl(self, x, wt, grid, ksize, stride=1, transposed=False, scale=None, shift=None,
                  residual=None, relu=False):
        """x [X*Y*Z, Cin] channels-last; wt [taps, Cout, Cin]; grid = (X, Y, Z) of the input;
        relu: 0/False none, 1/True relu(t + residual), 2 relu(t) + residual.
        Returns (y [OX*OY*OZ, Cout], (OX, OY, OZ))."""
        self._check(x=x, wt=wt, scale=scale, shift=shift, residual=residual)
        self._f32(x=x, wt=wt, scale=scale, shift=shift, residual=residual)
        ix, iy, iz = grid
        V, Cin = x.shape
        taps, Cout, Cin2 = wt.shape
        if V != ix * iy * iz or Cin2 != Cin or taps != (8 if transposed else ksize ** 3):
            raise RuntimeError("conv3d_cl: inconsistent shapes")
        if transposed:
            og = (2 * ix, 2 * iy, 2 * iz)
        else:
            pad = 0 if ksize == 2 else ksize // 2
            og = tuple((d + 2 * pad - ksize) // stride + 1 for d in grid)
        y = torch.empty((og[0] * og[1] * og[2], Cout), dtype=torch.float32, device=x.device)
        if residual is not None and residual.shape != y.shape:
            raise RuntimeError("conv3d_cl: residual shape mismatch")
        ws, ws_n = self._conv_workspace(x.device, ix, iy, iz, Cin, Cout, ksize, stride, transposed, 0)
        self._call("sgc_conv3d_cl_f32", x, wt, scale, shift, residual, y, ix, iy, iz, Cin, Cout, ksize, stride,
                   1 if transposed else 0, int(relu), ws, ws_n,
                   _meta=dict(V=V, Cin=Cin, Cout=Cout, taps=taps, OV=y.shape[0], flop_taps=1 if transposed else taps))
        return y, og

    # ---- coarse-to-fine glue -------------------------------------------------------------
    def upsample2x_occ(self, vol, grid, w=None, b=None, occ_out=None):
        """vol [X*Y*Z, C] rows -> (up [8*X*Y*Z, C], occ [8*X*Y*Z] or None, (2X, 2Y, 2Z)).  ``occ_out``: a contiguous fp32
        [8*X*Y*Z] tensor (e.g. a slice of the level-concatenated occupancy vector) to write the scores into."""
        self._check(vol=vol, w=w, b=b, occ_out=occ_out)
        self._f32(vol=vol, w=w, b=b, occ_out=occ_out)
        ix, iy, iz = grid
        V, Cc = vol.shape
        if V != ix * iy * iz or (w is not None and w.numel() != Cc):
            raise RuntimeError("upsample2x_occ: inconsistent shapes")
        if occ_out is not None and (w is None or occ_out.shape != (8 * V,)):
            raise RuntimeError("upsample2x_occ: occ_out must be [8 * X * Y * Z] (and needs the occupancy Linear)")
        up = torch.empty((8 * V, Cc), dtype=torch.float32, device=vol.device)
        occ = occ_out if occ_out is not None else (torch.empty(8 * V, dtype=torch.float32, device=vol.device) if w is not None else None)
        self._call("sgc_upsample2x_occ", vol, w, b, up, occ, ix, iy, iz, Cc)
        return up, occ, (2 * ix, 2 * iy, 2 * iz)

    def upsample2x_backward(self, grad_out):
        """grad_out [1|none, C, 2X, 2Y, 2Z] (NCDHW, contiguous) -> grad_in [same leading dims, C, X, Y, Z]: the adjoint
        of ``F.interpolate(scale_factor=2, mode='trilinear', align_corners=False)``."""
        self._check(grad_out=grad_out)
        self._f32(grad_out=grad_out)
        *lead, ox, oy, oz = grad_out.shape
        if ox % 2 or oy % 2 or oz % 2:
            raise RuntimeError("upsample2x_backward: even output sizes expected")
        planes = 1
        for d in lead:
            planes *= d
        gi = torch.empty((*lead, ox // 2, oy // 2, oz // 2), dtype=torch.float32, device=grad_out.device)
        self._call("sgc_upsample2x_backward", grad_out, gi, planes, ox // 2, oy // 2, oz // 2)
        return gi

    def scatter_add_rows(self, rows, idx, vol):
        self._check(rows=rows, idx=idx, vol=vol)
        self._f32(rows=rows, vol=vol)
        self._i64(idx=idx)
        if rows.shape[1] != vol.shape[1] or idx.numel() != rows.shape[0]:
            raise RuntimeError("scatter_add_rows: inconsistent shapes")
        if rows.shape[0] == 0:
            return vol
        self._call("sgc_scatter_add_rows", rows, idx, vol, rows.shape[0], rows.shape[1])
        return vol
