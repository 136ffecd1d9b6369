"""Autograd Functions with the reference's names and argument order.

Reference classes mirrored (mmdet3d_plugin/models/im2voxel/transformer_utils/
multi_scale_3ddeformable_attn_function.py and the DFA3D package twin
dfa3D/ops/multi_scale_3D_deform_attn.py):

* ``MultiScale3DDeformableAttnFunction_fp32`` (:275-351)  -- returns ``(output, depth_score)``;
* ``MultiScaleDepthScoreSampleFunction_fp32`` (:228-273)
* ``WeightedMultiScaleDeformableAttnFunction_fp32`` (:102-178)

Differences, all behind the same call signature:

* the one-stage Function runs ONE fused HIP kernel forward and ONE backward (no
  ``[B,Q,M,L,P,4]`` round trip, no ``[..., :2].contiguous()`` slice copies);
* ``value_dpt_dist`` may be passed un-replicated (``[B,S,1,D]``) -- the reference's
  ``.repeat(1,1,num_heads,1)`` (TU/deformable_cross_attention.py:422) is not needed, and the
  returned gradient then already holds the sum over heads;
* no host sync in backward: the reference's ``grad_depth_score_.sum() != 0.0`` check
  (:314) is replaced by ``ctx.mark_non_differentiable`` on the returned depth score;
* the ``_fp16`` twins of the reference cannot run (its kernels have no half dispatch,
  SURVEY.md section 0 fact 2); they are aliases of the fp32 Functions here, which is what
  ``custom_fwd(cast_inputs=torch.float32)`` amounts to.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ext


class MultiScale3DDeformableAttnFunction_fp32(Function):
    @staticmethod
    def forward(ctx, value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step=64):
        value = value.float().contiguous()
        value_dpt_dist = value_dpt_dist.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        output, depth_score = ext.ops().dfa3d_forward(
            value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
            sampling_locations, attention_weights, want_score=True)
        ctx.save_for_backward(value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        ctx.mark_non_differentiable(depth_score)
        return output, depth_score

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, grad_depth_score_=None):
        value, dist, shapes3, lsi, loc, attn = ctx.saved_tensors
        gv, gd, gl, ga = ext.ops().dfa3d_backward(value, dist, shapes3, lsi, loc, attn,
                                                  grad_output.float().contiguous())
        return gv, gd, None, None, gl, ga, None


class MultiScaleDepthScoreSampleFunction_fp32(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                im2col_step=64):
        value = value.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        out = ext.ms_depth_score_sample_forward(value, value_spatial_shapes, value_level_start_index,
                                                sampling_locations, im2col_step=im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes3, lsi, loc = ctx.saved_tensors
        grad_value = torch.zeros_like(value)
        grad_loc = torch.zeros_like(loc)
        ext.ms_depth_score_sample_backward(value, shapes3, lsi, loc, grad_output.float().contiguous(),
                                           grad_value, grad_loc)
        return grad_value, None, None, grad_loc, None


class WeightedMultiScaleDeformableAttnFunction_fp32(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, depth_score, im2col_step=64):
        value = value.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        depth_score = depth_score.float().contiguous()
        out = ext.wms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                          sampling_locations, attention_weights, depth_score,
                                          im2col_step=im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights, depth_score)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes2, lsi, loc2, attn, score = ctx.saved_tensors
        grad_value = torch.zeros_like(value)
        grad_loc = torch.zeros_like(loc2)
        grad_attn = torch.zeros_like(attn)
        grad_score = torch.zeros_like(score)
        ext.wms_deform_attn_backward(value, shapes2, lsi, loc2, attn, score,
                                     grad_output.float().contiguous(), grad_value, grad_loc,
                                     grad_attn, grad_score)
        return grad_value, None, None, grad_loc, grad_attn, grad_score, None


TRAIN_BWD_TILED_SHARED = os.environ.get("SGC_TRAIN_BWD_SHARED", "0") == "1"     # the one-head geometry sample on the tiled backward too (A/B)


class PairListDeformAttnFunction(Function):
    """The fused DFA3D operator over an ITEM LIST: item i = (camera ``item_batch[i]``, its sampling locations / weights).
    Training-path counterpart of ``MultiScale3DDeformableAttnFunction_fp32`` without the reference's padded
    [N, max_len] rebatch (TU/deformable_cross_attention.py:759-773): only visible (camera, voxel) pairs are sampled and
    back-propagated.  value [B,S,M,Cm], dist [B,S,1|M,D], loc3 [n,M,L,P,3], attn [n,M,L,P] -> [n, M*Cm]."""

    @staticmethod
    def forward(ctx, value, value_dpt_dist, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, item_batch, bins=None):
        """``bins`` (one level only): (bin_offset, H, W, bin_w, bin_h, halo) when the items are in the (camera, bin) order of
        ``sgc_bin_pairs`` -- the backward then takes the LDS-tiled kernel (``sgc_dfa3d_backward_binned``) instead of the item kernel's
        global atomics.  The forward does not care about the order."""
        value = value.float().contiguous()
        value_dpt_dist = value_dpt_dist.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        item_batch = item_batch.to(torch.int32).contiguous()
        out = ext.ops().dfa3d_forward_items(value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
                                            sampling_locations, attention_weights, item_batch)
        ctx.save_for_backward(value, value_dpt_dist, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights, item_batch)
        ctx.bins = bins
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, dist, shapes3, lsi, loc, attn, item_batch = ctx.saved_tensors
        ops = ext.ops()
        grad_output = grad_output.float().contiguous()
        bins = ctx.bins
        N, S, M, Cm = value.shape
        n, LM, L, P = loc.shape[:4]
        if bins is not None and L == 1 and dist.shape[2] == 1 and n > 0:
            bin_offset, H, W, bw, bh, halo = bins[:6]
            head_shift = bins[6] if len(bins) > 6 and M > 1 else None
            # the kernel's head split: channel groups of 32 (16) channels; one head over C channels (the geometry sample) runs as C / 32
            # groups that share its sample set
            cmb = Cm if Cm in (16, 32) else (32 if Cm % 32 == 0 else 16 if Cm % 16 == 0 else 0)
            # measured (tools/bwd_tile_bench.py, profiles/r06_bwd_tile_bench_cfg2.txt): the deformable call (8 heads x 4 points) 1.25 ms
            # tiled against 1.89 ms on the item kernel; the geometry sample (ONE sample per pair: 4 KB of atomics, not 16) 0.57 against
            # 0.42 -- it keeps the item kernel unless TRAIN_BWD_TILED_SHARED says otherwise
            if M == 1 and Cm != cmb and not TRAIN_BWD_TILED_SHARED:
                cmb = 0
            if cmb and (cmb == Cm or M == 1) and P <= 4 and ops.dfa3d_backward_binned_fits(H, W, cmb, dist.shape[-1], bw, bh, halo):
                mb = M * Cm // cmb
                gv, gd, gl, ga = ops.dfa3d_backward_binned(value.view(N, S, mb, cmb), dist, loc, attn, bin_offset, grad_output, H, W, bw, bh,
                                                           halo, want_grad_loc=ctx.needs_input_grad[4], want_grad_attn=ctx.needs_input_grad[5],
                                                           head_shift=head_shift)
                return gv.view_as(value), gd, None, None, gl, ga, None, None
        gv, gd, gl, ga = ops.dfa3d_backward_items(value, dist, shapes3, lsi, loc, attn, item_batch, grad_output)
        return gv, gd, None, None, gl, ga, None, None


def _require_bf16_planes(who):
    """``sgc_pack_conv_weight`` emits bfloat16 bit patterns; in the fp16 arithmetic mode (``sgc_set_conv_products(2)``) the MFMA
    kernels would read them as IEEE half.  The training Functions therefore refuse that mode instead of computing garbage."""
    from .plugin import conv_plan
    if conv_plan.CONV_PRODUCTS == 2:
        raise RuntimeError(f"{who}: the fp16 arithmetic mode (set_conv_mode('fp16')) is inference-only -- the training Functions "
                           "pack bfloat16 weight planes; switch to 'bf16x3' / 'bf16' or run under torch.no_grad()")


class TrainWeightPlanes:
    """The bf16 hi | lo planes the training Functions multiply with, kept across steps and repacked TOGETHER.

    Every pass of a training step needs each parameter in the kernels' layout: [taps, Cout, Cin] for the forward, [taps, Cin, Cout]
    with mirrored taps for the input gradient (cuDNN does the same internally, necks/imvoxelnet.py:36-64).  Packing them where they
    are used is ~100 launches per config-2 step, most of them a few microseconds of work.  Here a plane pair is allocated the first
    time a (parameter storage, shape, form) is asked for and is registered; ``begin_step()`` -- the detector calls it at the start
    of every training forward -- repacks ALL registered planes in one launch (``sgc_pack_conv_weight_batch``): the same bytes in one
    launch instead of a hundred.  A plane is never served stale: ``get`` compares the version counter the parameter shares with
    the alias kept here (an in-place update between ``begin_step`` and the use repacks); edits through ``.data`` do not move
    that counter, which is why ``begin_step`` repacks unconditionally.  Entries are keyed by storage address + shape, so the
    per-step views of a fused parameter (``in_proj_weight[:C]``) map to one entry; an entry no ``get`` has asked for during two
    steps is dropped (its parameter is gone or has moved).  ``SGC_TRAIN_PACK_BATCH=0`` restores the pack-per-use form."""

    MAX_ENTRIES = 1024             # callers that never reach begin_step (stand-alone Functions, tests) cannot grow the table without bound

    def __init__(self):
        self.entries = {}          # (data_ptr, shape, transpose, flip, pad_rows, pad_cols) -> dict
        self.ephemeral = {}        # data_ptr of a per-step temporary (mark_ephemeral) -> dict(ref=tensor, planes={form key: (hi, lo)})
        self.plans = None          # per device: (entries, launch plan)
        self.enabled = os.environ.get("SGC_TRAIN_PACK_BATCH", "1") != "0"
        self.launches = 0          # batched launches so far (tests)
        self.step = 0

    def mark_ephemeral(self, weight):
        """``weight`` is a temporary of this step (e.g. the head's ``torch.cat`` of three parameters): its planes are packed where they
        are first used and shared by the step's later uses, but never REGISTERED -- a fresh temporary every step would add entries,
        reset the batched plan (a host-side rebuild + a synchronous copy of the item list) and pin the dead storages for two steps."""
        # the entry holds the tensor: its storage cannot be freed and re-used under the same address while the entry lives (entries go at
        # begin_step, or oldest-first beyond eight for callers that never reach it)
        if weight.data_ptr() not in self.ephemeral:
            while len(self.ephemeral) >= 8:
                del self.ephemeral[next(iter(self.ephemeral))]
            self.ephemeral[weight.data_ptr()] = dict(ref=weight, planes={})

    def get(self, weight, transpose=False, flip=False, pad_rows=1, pad_cols=1):
        ops = ext.ops()
        if not self.enabled or not weight.is_cuda or weight.dtype != torch.float32 or not weight.is_contiguous():
            return ops.pack_conv_weight(weight.detach().float(), transpose=transpose, flip=flip, pad_rows=pad_rows, pad_cols=pad_cols)
        eph = self.ephemeral.get(weight.data_ptr())
        if eph is not None:
            fk = (weight.shape, weight._version, transpose, flip, pad_rows, pad_cols)
            if fk not in eph["planes"]:
                eph["planes"][fk] = ops.pack_conv_weight(weight.detach(), transpose=transpose, flip=flip, pad_rows=pad_rows, pad_cols=pad_cols)
            return eph["planes"][fk]
        key = (weight.data_ptr(), weight.shape, transpose, flip, pad_rows, pad_cols)
        e = self.entries.get(key)
        if e is None:
            if len(self.entries) >= self.MAX_ENTRIES:          # least recently used half goes
                for k in sorted(self.entries, key=lambda k: self.entries[k]["used"])[:self.MAX_ENTRIES // 2]:
                    del self.entries[k]
            shape = ops.packed_shape(weight.shape, transpose, pad_rows, pad_cols)
            hi = torch.zeros(shape, dtype=torch.bfloat16, device=weight.device)
            lo = torch.zeros_like(hi)
            w = weight.detach()                        # an alias: keeps the storage where it is, shares the version counter
            ops.pack_conv_weight(w, transpose=transpose, flip=flip, pad_rows=pad_rows, pad_cols=pad_cols, out=(hi, lo))
            self.entries[key] = dict(w=w, hi=hi, lo=lo, version=w._version, transpose=bool(transpose), flip=bool(flip), used=self.step)
            self.plans = None
            return hi, lo
        e["used"] = self.step
        if e["version"] != weight._version:
            self.repack()
        return e["hi"], e["lo"]

    def repack(self):
        """One launch per device over every registered plane pair."""
        if self.plans is None:
            by_dev = {}
            for e in self.entries.values():
                by_dev.setdefault(e["hi"].device, []).append(e)
            ops = ext.ops()
            self.plans = [(es, ops.pack_conv_weight_plan([(e["w"], e["hi"], e["lo"], e["transpose"], e["flip"]) for e in es]))
                          for es in by_dev.values()]
        for es, plan in self.plans:
            ext.ops().run_pack_plan(plan)
            self.launches += 1
            for e in es:
                e["version"] = e["w"]._version

    def begin_step(self):
        self.ephemeral = {}
        if not self.enabled or not self.entries:
            return
        self.step += 1
        stale = [k for k, e in self.entries.items() if e["used"] < self.step - 2]
        for k in stale:
            del self.entries[k]
        if stale:
            self.plans = None
        if self.entries:
            self.repack()

    def clear(self):
        self.entries, self.plans, self.ephemeral = {}, None, {}


_TRAIN_PLANES = TrainWeightPlanes()


def train_weight_planes():
    return _TRAIN_PLANES


class NchwToRowsFunction(Function):
    """[N, C, h, w] map (any strides: the crop view of AdaptiveSparseHead.py:58-59) -> channels-last rows [N, h*w, C]
    (TU/transformer.py:151-170 ``flatten(2).permute``) with the transpose on the HIP kernels in both directions: forward
    ``sgc_nchw_to_nhwc_crop``, backward ``sgc_nhwc_to_nchw_pad`` (a contiguous NCHW gradient).  Autograd's own chain for
    ``flatten / permute / contiguous`` is two strided copies per level (0.26 ms each at config 2's finest level)."""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[2], x.shape[3])
        return ext.ops().nchw_to_nhwc_crop(x.detach().float(), x.shape[2], x.shape[3])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_rows):
        h, w = ctx.hw
        return ext.ops().nhwc_to_nchw_pad(grad_rows.float().contiguous(), h, w)


def _pad_cols(t, mult):
    """[rows, C] -> [rows, ceil(C / mult) * mult] with zero columns (a view when nothing is added)."""
    c = t.shape[-1]
    cp = (c + mult - 1) // mult * mult
    return t if cp == c else torch.nn.functional.pad(t, (0, cp - c))


class ChannelsLastConv3dFunction(Function):
    """``nn.Conv3d(k, stride, padding=k//2, bias=False)`` on channels-last rows, forward AND backward on the MFMA kernels
    (the neck / head convolutions in training; the reference gets all three passes from cuDNN,
    necks/imvoxelnet.py:36-64, dense_heads/imvoxel_head_v2.py:75-78).

    x [X*Y*Z, Cin] rows of the volume, weight [Cout, Cin, k, k, k] (the module's parameter, untouched layout), grid =
    (X, Y, Z), k in {1, 3}, stride in {1, 2} -> y [OX*OY*OZ, Cout].  Cin % 32 == 0.

    * forward: ``sgc_conv3d_cl_bf16x3`` (no epilogue: BatchNorm in training mode needs the raw output);
    * input gradient: the same kernel on dy with the taps mirrored and Cin / Cout swapped; for stride 2 on the
      zero-interleaved dy (dy at the even fine-grid positions: 8x the necessary matrix work on two small layers, no new
      kernel -- the two stride-2 layers are 6 % of the neck's FLOPs);
    * weight gradient: ``sgc_conv3d_wgrad_bf16x3``.
    """

    @staticmethod
    def forward(ctx, x, weight, grid, ksize, stride):
        _require_bf16_planes("ChannelsLastConv3dFunction")
        ops = ext.ops()
        cout, cin = weight.shape[:2]
        # parameter [Cout, Cin, k, k, k] -> the kernel's [taps, Cout (multiple of 4), Cin] bf16 hi / lo planes in ONE launch
        hi, lo = _TRAIN_PLANES.get(weight, pad_rows=4)
        x = x.float().contiguous()
        y, og = ops.conv3d_cl_bf16x3(x, hi, lo, grid, ksize, stride)
        ctx.save_for_backward(x, weight)
        ctx.geom = (tuple(grid), tuple(og), ksize, stride)
        return y[:, :cout] if y.shape[1] != cout else y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        ops = ext.ops()
        x, weight = ctx.saved_tensors
        grid, og, ksize, stride = ctx.geom
        cout, cin = weight.shape[:2]
        dy = dy.float().contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dx[i] = sum_d dy[(i - d + pad) / stride] W[d]^T: a stride-1 convolution of (zero-interleaved) dy with the
            # mirrored taps; its "input channels" are Cout, padded to the kernel's multiple of 32
            hi, lo = _TRAIN_PLANES.get(weight, transpose=True, flip=True, pad_cols=32)
            g = _pad_cols(dy, 32)
            if stride == 2:
                up = torch.zeros((grid[0], grid[1], grid[2], g.shape[1]), dtype=torch.float32, device=dy.device)
                up[:2 * og[0]:2, :2 * og[1]:2, :2 * og[2]:2] = g.view(og[0], og[1], og[2], -1)
                g = up.view(-1, g.shape[1])
            dx, _ = ops.conv3d_cl_bf16x3(g.contiguous(), hi, lo, grid, ksize, 1)
        if ctx.needs_input_grad[1]:
            # the halo form of the weight gradient wants 32-channel dy tiles: the head's 3 x 3 x 3 convolutions (Cout = 28 ...)
            # get four zero columns rather than the per-tap tile kernel
            mult = 32 if ksize == 3 and stride == 1 and cin % 32 == 0 else 4
            dwk = ops.conv3d_wgrad_bf16x3(x, _pad_cols(dy, mult).contiguous(), grid, ksize, stride)     # [taps, cout_p, cin]
            dw = ops.unpack_conv_wgrad(dwk, weight.shape).to(weight.dtype)
        return dx, dw, None, None, None


class ChannelsLastConvTranspose3dFunction(Function):
    """``nn.ConvTranspose3d(2, 2, bias=False)`` on channels-last rows (up_block_*, necks/imvoxelnet.py:56-58): forward on the
    parity form of the MFMA kernel, input gradient = the k2 s2 convolution of dy, weight gradient =
    ``sgc_conv3d_wgrad_bf16x3`` with the roles of the two tensors exchanged.  x [X*Y*Z, Cin], weight [Cin, Cout, 2, 2, 2]
    -> y [8*X*Y*Z, Cout]; Cin % 32 == 0, Cout % 32 == 0."""

    @staticmethod
    def forward(ctx, x, weight, grid):
        _require_bf16_planes("ChannelsLastConvTranspose3dFunction")
        ops = ext.ops()
        cin, cout = weight.shape[:2]
        hi, lo = _TRAIN_PLANES.get(weight, transpose=True)            # [Cin, Cout, 8] -> [8, Cout, Cin]
        x = x.float().contiguous()
        y, og = ops.conv3d_cl_bf16x3(x, hi, lo, grid, 2, 2, True)
        ctx.save_for_backward(x, weight)
        ctx.geom = (tuple(grid), tuple(og))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        ops = ext.ops()
        x, weight = ctx.saved_tensors
        grid, og = ctx.geom
        cin, cout = weight.shape[:2]
        dy = dy.float().contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            hi, lo = _TRAIN_PLANES.get(weight)                        # [Cin, Cout, 8] -> [8, Cin, Cout]
            dx, _ = ops.conv3d_cl_bf16x3(dy, hi, lo, og, 2, 2)                  # dx[x] = sum_p dy[2x + p] W[:, :, p]^T
        if ctx.needs_input_grad[1]:
            dwk = ops.conv3d_wgrad_bf16x3(dy, x, og, 2, 2)                      # [8, cin, cout]
            dw = ops.unpack_conv_wgrad(dwk, weight.shape).to(weight.dtype)
        return dx, dw, None


class BatchNormRowsFunction(Function):
    """Training-mode ``nn.BatchNorm3d`` on channels-last rows [rows, C] (necks/imvoxelnet.py:36-64: every convolution of the neck
    is followed by one): statistics, normalisation and the backward reductions on ``sgc_bn_rows_forward / _backward`` (ordered
    reductions: the same bits every run) instead of torch's channels-last batch-norm kernels.  The running statistics are
    updated in place by the forward kernel, as ``F.batch_norm(training=True)`` does.

    ``residual`` / ``relu`` (round 5): the elementwise tail of the neck's blocks inside the same passes -- ``relu(norm(conv))`` and the
    ResBlock tail ``relu(norm2(conv2) + identity)`` -- i.e. ``y = relu(bn(rows) + residual)`` forward, and backward the incoming
    gradient masked where ``y > 0`` (torch's ``threshold_backward``) before the BatchNorm backward; the identity's gradient is that
    masked gradient (``sgc_bn_rows_act_forward / _backward``)."""

    @staticmethod
    def forward(ctx, rows, weight, bias, running_mean, running_var, momentum, eps, residual=None, relu=False):
        ops = ext.ops()
        x = rows.contiguous()
        res = None if residual is None else residual.detach().contiguous()
        y, mean, invstd = ops.bn_rows_forward(x, weight.detach().contiguous(), bias.detach().contiguous(), running_mean, running_var,
                                              momentum, eps, residual=res, relu=relu)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        if relu:
            ctx.save_for_backward(x, mean, invstd, weight.detach(), y)
        else:
            ctx.save_for_backward(x, mean, invstd, weight.detach())
        return y

    @staticmethod
    def backward(ctx, grad_y):
        saved = ctx.saved_tensors
        x, mean, invstd, weight = saved[:4]
        g = grad_y.contiguous()
        if not ctx.relu and not ctx.has_res:
            dx, dw, db = ext.ops().bn_rows_backward(x, g, mean, invstd, weight.contiguous())
            return dx, dw, db, None, None, None, None, None, None
        want_res = ctx.has_res and ctx.needs_input_grad[7]
        if not ctx.relu:                               # no mask: the identity's gradient IS the incoming gradient
            dx, dw, db = ext.ops().bn_rows_backward(x, g, mean, invstd, weight.contiguous())
            return dx, dw, db, None, None, None, None, (g if want_res else None), None
        dx, dw, db, dres = ext.ops().bn_rows_backward(x, g, mean, invstd, weight.contiguous(), y_relu=saved[4], want_dresidual=want_res)
        return dx, dw, db, None, None, None, None, dres, None


class LinearRowsFunction(Function):
    """``nn.Linear`` over a row list with all three passes on the MFMA kernels (training path of the view transform:
    value_proj / sampling_offsets / attention_weights / output_proj, TU/deformable_cross_attention.py:417-436,826): forward
    and input gradient on ``sgc_linear_rows_bf16x3`` (the latter with W^T), weight gradient on ``sgc_conv3d_wgrad_bf16x3``
    with ksize 1 (the rows are its "voxels").  x [rows, Cin] (Cin % 32 == 0), weight [Cout, Cin], bias [Cout] | None."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_bf16_planes("LinearRowsFunction")
        ops = ext.ops()
        cout, cin = weight.shape
        x = x.float().contiguous()
        hi, lo = _TRAIN_PLANES.get(weight, pad_rows=4)            # [1, Cout (multiple of 4), Cin]
        shift = None if bias is None else _pad_cols(bias.detach().float().view(1, -1), 4).view(-1).contiguous()
        y = ops.linear_rows_bf16x3(x, hi, lo, shift)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y if y.shape[1] == cout else y[:, :cout]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        ops = ext.ops()
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape
        dy = dy.float().contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            hi, lo = _TRAIN_PLANES.get(weight, transpose=True, pad_cols=32)   # [1, Cin, Cout -> mult of 32]: dx = dy @ W
            dx = ops.linear_rows_bf16x3(_pad_cols(dy, 32).contiguous(), hi, lo, None)
        if ctx.needs_input_grad[1]:
            dwk = ops.conv3d_wgrad_bf16x3(x, _pad_cols(dy, 4).contiguous(), (x.shape[0], 1, 1), 1, 1)     # [1, cout_p, cin]
            dw = dwk[0, :cout].to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def linear_rows(module, x):
    """``module(x)`` for an ``nn.Linear`` with the passes on the HIP kernels when the shapes allow (CUDA fp32, in_features %
    32 == 0, at least a tile of rows); any leading shape."""
    from .plugin.conv_plan import TRAIN_CONV, train_products_ok
    if (TRAIN_CONV != "hip" or not train_products_ok() or not x.is_cuda or x.dtype != torch.float32 or module.in_features % 32
            or x.numel() // max(1, module.in_features) < 128):
        return module(x)
    y = LinearRowsFunction.apply(x.reshape(-1, module.in_features), module.weight, module.bias)
    return y.reshape(*x.shape[:-1], module.out_features)


class ViewAttendFunction(Function):
    """Softmax over the views that see a voxel (``nn.MultiheadAttention`` with query length 1 and a key-padding mask,
    TU/deformable_cross_attention.py:829-833) on the PAIR LIST, forward and backward: q [n_valid, C] (in-projected query of
    every visible voxel), kv [n_pairs, 2C] (k | v in-projected per visible pair), slot [N, Nq] int32 (pair index or -1),
    valid_index [n_valid] int32 -> ctx [n_valid, C] (before out_proj)."""

    @staticmethod
    def forward(ctx_, q, kv, slot, valid_index, heads):
        ops = ext.ops()
        q, kv = q.float().contiguous(), kv.float().contiguous()
        out = ops.view_attend(q, kv, slot, valid_index, heads)
        ctx_.save_for_backward(q, kv, slot, valid_index, out)
        ctx_.heads = heads
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx_, grad_out):
        q, kv, slot, valid_index, out = ctx_.saved_tensors
        gq, gkv = ext.ops().view_attend_backward(q, kv, slot, valid_index, ctx_.heads, out, grad_out.float().contiguous())
        return gq, gkv, None, None, None


# the DFA3D package spells them without the suffix (dfa3D/ops/multi_scale_3D_deform_attn.py:22,67,146)
MultiScale3DDeformableAttnFunction = MultiScale3DDeformableAttnFunction_fp32
MultiScaleDepthScoreSampleFunction = MultiScaleDepthScoreSampleFunction_fp32
WeightedMultiScaleDeformableAttnFunction = WeightedMultiScaleDeformableAttnFunction_fp32
MultiScale3DDeformableAttnFunction_fp16 = MultiScale3DDeformableAttnFunction_fp32
MultiScaleDepthScoreSampleFunction_fp16 = MultiScaleDepthScoreSampleFunction_fp32
WeightedMultiScaleDeformableAttnFunction_fp16 = WeightedMultiScaleDeformableAttnFunction_fp32
