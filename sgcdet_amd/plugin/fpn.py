"""Image FPN whose output convolutions EMIT the layout the view transformation consumes.

Reference call site: ``x = list(self.neck(x))`` (detectors/SGCDet.py:67) with ``neck=dict(type='FPN',
in_channels=[256, 512, 1024, 2048], out_channels=embed_dims, num_outs=4)`` (configs/SGCDet_ScanNet.py:84-88); the maps are
then reshaped to [B, N, C, H, W] (:68-69) and every level is flattened / permuted to [N, H*W, C] per use
(TU/transformer.py:151-170).  The class itself is mmdet's (v2.x ``mmdet/models/necks/fpn.py``), which is NOT vendored in
the reference tree: its semantics are restated here from the published module (lateral 1x1 convolutions with bias, top-down
nearest upsampling to the finer level's size + addition, 3x3 output convolutions with bias; extra levels by stride-2
max-pool subsampling of the last output) and are *unpinned*; state-dict keys are mmdet's
(``lateral_convs.{i}.conv.{weight,bias}``, ``fpn_convs.{i}.conv.{weight,bias}``).

Eval mode on the GPU (``forward`` with CUDA inputs, no grad) runs every convolution on ``sgc_conv2d_nhwc_bf16x3`` over
channels-last rows and returns [N, C, H, W] tensors that ARE channels-last in memory (``y.permute(0, 3, 1, 2)`` of the
[N, H, W, C] result): `SGCDet.forward_features` reads them in place, the NCHW -> NHWC pass of the path disappears, and
nothing else changes for the caller.  Backbone maps that arrive channels-last in memory are read in place too; NCHW ones
are transposed once by ``sgc_nchw_to_nhwc_crop``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ext
from ..mmcv_lite import NECKS
from .conv_plan import module_fingerprint


class _ConvModule(nn.Module):
    """mmcv ``ConvModule`` without norm / activation: a Conv2d with bias under the key ``conv``."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2)

    def forward(self, x):
        return self.conv(x)


def _nearest_index(dst, src, device):
    """source index of F.interpolate(mode='nearest') for every destination index: floor(d * src / dst)."""
    return torch.div(torch.arange(dst, device=device) * src, dst, rounding_mode="floor").clamp_(max=src - 1)


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__()
        if add_extra_convs or norm_cfg is not None or act_cfg is not None or conv_cfg is not None:
            raise NotImplementedError("FPN: only the plain configuration of the SGCDet configs is restated")
        if upsample_cfg.get("mode", "nearest") != "nearest" or "scale_factor" in upsample_cfg:
            raise NotImplementedError("FPN: nearest upsampling to the finer level's size only")
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.start_level = start_level
        self.backbone_end_level = len(in_channels) if end_level in (-1, len(in_channels) - 1) else end_level + 1
        self.lateral_convs = nn.ModuleList(_ConvModule(in_channels[i], out_channels, 1)
                                           for i in range(start_level, self.backbone_end_level))
        self.fpn_convs = nn.ModuleList(_ConvModule(out_channels, out_channels, 3)
                                       for _ in range(start_level, self.backbone_end_level))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    # ---- reference formulation (any device, autograd) ----------------------------------------------------------
    def _forward_torch(self, inputs):
        lat = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        outs = [conv(lat[i]) for i, conv in enumerate(self.fpn_convs)]
        while len(outs) < self.num_outs:
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return tuple(outs)

    # ---- MFMA kernels on channels-last rows ----------------------------------------------------------------------
    def _plan(self):
        fp = module_fingerprint(self)
        if getattr(self, "_hip_plan", None) is not None and self._hip_plan[0] == fp:
            return self._hip_plan[1]
        ops = ext.ops()

        def spec(conv):
            w = conv.weight.detach().float()
            k = w.shape[2]
            hi, lo = ops.split_operand(w.permute(2, 3, 0, 1).reshape(k * k, w.shape[0], w.shape[1]).contiguous())
            return hi, lo, conv.bias.detach().float().contiguous(), k
        plan = ([spec(m.conv) for m in self.lateral_convs], [spec(m.conv) for m in self.fpn_convs])
        self._hip_plan = (fp, plan)
        return plan

    @staticmethod
    def _rows(x):
        """[N, C, H, W] (any strides) -> ([N*H*W, C] fp32 rows, (N, H, W)); zero-copy for channels-last memory."""
        ops = ext.ops()
        N, C, H, W = x.shape
        if x.is_contiguous(memory_format=torch.channels_last) and x.dtype == torch.float32:
            return x.permute(0, 2, 3, 1).reshape(N * H * W, C), (N, H, W)
        return ops.nchw_to_nhwc_crop(x.float(), H, W).view(N * H * W, C), (N, H, W)

    def _forward_hip(self, inputs):
        ops = ext.ops()
        lat_specs, out_specs = self._plan()
        lat, dims = [], []
        for i, (hi, lo, b, k) in enumerate(lat_specs):
            rows, nhw = self._rows(inputs[i + self.start_level])
            lat.append(ops.conv2d_nhwc_bf16x3(rows, hi, lo, nhw, k, shift=b))
            dims.append(nhw)
        C = self.out_channels
        for i in range(len(lat) - 1, 0, -1):                 # top-down: nearest upsample to the finer size, add in place
            (N, Hs, Ws), (_, Hd, Wd) = dims[i], dims[i - 1]
            src = lat[i].view(N, Hs, Ws, C)
            ih, iw = _nearest_index(Hd, Hs, src.device), _nearest_index(Wd, Ws, src.device)
            lat[i - 1].view(N, Hd, Wd, C).add_(src[:, ih][:, :, iw])
        outs = []
        for i, (hi, lo, b, k) in enumerate(out_specs):
            N, H, W = dims[i]
            y = ops.conv2d_nhwc_bf16x3(lat[i], hi, lo, dims[i], k, shift=b)
            outs.append(y.view(N, H, W, C).permute(0, 3, 1, 2))          # logical NCHW, channels-last memory
        while len(outs) < self.num_outs:
            outs.append(outs[-1][:, :, ::2, ::2])
        return tuple(outs)

    def forward(self, inputs):
        x0 = inputs[0]
        if (not self.training and not torch.is_grad_enabled() and x0.is_cuda
                and all(c % 32 == 0 for c in self.in_channels) and self.out_channels % 32 == 0):
            return self._forward_hip(inputs)
        return self._forward_torch(inputs)
