"""``FastIndoorImVoxelNeck``: dense 3D residual U-Net over the voxel volume.

Reference: mmdet3d_plugin/models/necks/imvoxelnet.py:8-67 (neck), :146-173 (BasicBlock3dV2).
State-dict keys are the reference's: ``down_layer_{i}.{b}.{conv1,norm1,conv2,norm2,
downsample.0,downsample.1}``, ``up_block_{i}.{0,1,3,4}``, ``out_block_{i}.{0,1}``.

"Sparse" in SGCDet means sparse QUERIES; the volume entering the neck is dense (every voxel
carries the upsampled coarse feature, AdaptiveSparseHead.py:77-82), so the convolutions
stay dense -- a submanifold-sparse convolution would change the results (SURVEY.md
section 0, fact 3).
"""
import torch
from torch import nn

from ..mmcv_lite import NECKS
from .conv_plan import (ConvSpec, bn_rows, conv_rows, conv_transpose_rows, module_fingerprint, rows_to_ncdhw,
                        to_channels_last_rows, train_conv_on_hip)


class BasicBlock3dV2(nn.Module):
    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv3d(in_channels, out_channels, 3, stride, 1, bias=False)
        self.norm1 = nn.BatchNorm3d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv3d(out_channels, out_channels, 3, 1, 1, bias=False)
        self.norm2 = nn.BatchNorm3d(out_channels)
        if stride != 1:
            self.downsample = nn.Sequential(nn.Conv3d(in_channels, out_channels, 1, stride, bias=False),
                                            nn.BatchNorm3d(out_channels))

    def forward(self, x):
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        skip = self.downsample(x) if self.stride != 1 else x
        return self.relu(out + skip)


def _conv_bn_relu(cin, cout):
    return nn.Sequential(nn.Conv3d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


@NECKS.register_module()
class FastIndoorImVoxelNeck(nn.Module):
    def __init__(self, in_channels, n_blocks, out_channels):
        super().__init__()
        self.n_scales = len(n_blocks)
        width = in_channels
        for i, nb in enumerate(n_blocks):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(nb):
                if b == 0 and stride != 1:
                    blocks.append(BasicBlock3dV2(width, width * 2, stride))
                    width *= 2
                else:
                    blocks.append(BasicBlock3dV2(width, width))
            setattr(self, f"down_layer_{i}", nn.Sequential(*blocks))
            if i > 0:
                setattr(self, f"up_block_{i}", nn.Sequential(
                    nn.ConvTranspose3d(width, width // 2, 2, 2, bias=False), nn.BatchNorm3d(width // 2),
                    nn.ReLU(inplace=True),
                    nn.Conv3d(width // 2, width // 2, 3, 1, 1, bias=False), nn.BatchNorm3d(width // 2),
                    nn.ReLU(inplace=True)))
            setattr(self, f"out_block_{i}", _conv_bn_relu(width, out_channels))

    # ---- eval-mode lowering onto the MFMA implicit-GEMM kernel ---------------------------
    def _plan(self):
        fp = module_fingerprint(self)
        if getattr(self, "_hip_plan", None) is not None and self._hip_plan[0] == fp:
            return self._hip_plan[1]
        plan = {}
        for i in range(self.n_scales):
            blocks = []
            for blk in getattr(self, f"down_layer_{i}"):
                blocks.append(dict(
                    c1=ConvSpec(blk.conv1.weight, blk.norm1, ksize=3, stride=blk.stride),
                    c2=ConvSpec(blk.conv2.weight, blk.norm2, ksize=3),
                    ds=ConvSpec(blk.downsample[0].weight, blk.downsample[1], ksize=1, stride=blk.stride)
                    if blk.stride != 1 else None))
            plan[f"down_{i}"] = blocks
            if i > 0:
                up = getattr(self, f"up_block_{i}")
                plan[f"up_{i}"] = (ConvSpec(up[0].weight, up[1], ksize=2, stride=2, transposed=True),
                                   ConvSpec(up[3].weight, up[4], ksize=3))
            ob = getattr(self, f"out_block_{i}")
            plan[f"out_{i}"] = ConvSpec(ob[0].weight, ob[1], ksize=3)
        self._hip_plan = (fp, plan)
        return plan

    def _forward_hip(self, x, tail_masks=None):
        """``tail_masks`` = (mask for up_block_1's 3x3x3, mask for out_block_0), uint8 [X*Y*Z] of the finest grid: the only
        layers whose outputs feed nothing but the finest head scale (x1 also feeds the next ConvTranspose, so every coarser
        layer stays dense).  Live rows are bit-identical to the dense launch."""
        plan = self._plan()
        rows, grid = to_channels_last_rows(x)
        skips = []
        for i in range(self.n_scales):
            for b in plan[f"down_{i}"]:
                h, g1 = b["c1"](rows, grid, relu=1)
                skip = rows if b["ds"] is None else b["ds"](rows, grid)[0]
                rows, grid = b["c2"](h, g1, residual=skip, relu=1)
            skips.append((rows, grid))
        outs = []
        for i in reversed(range(self.n_scales)):
            if i < self.n_scales - 1:
                up_t, up_c = plan[f"up_{i + 1}"]
                h, g = up_t(rows, grid, relu=1)
                m_up = tail_masks[0] if (tail_masks is not None and i == 0) else None
                rows, grid = up_c(h, g, residual=skips[i][0], relu=2, out_mask=m_up)      # relu(bn(conv)) + skip
            spec = plan[f"out_{i}"]
            m_out = tail_masks[1] if (tail_masks is not None and i == 0) else None
            o, g = spec(rows, grid, relu=1, out_mask=m_out)
            outs.append(rows_to_ncdhw(o, g, spec.cout))
        return outs[::-1]

    def _forward_autograd_hip(self, x):
        """Training / autograd path on the HIP kernels: every convolution -- forward, input gradient,
        weight gradient -- through ``ChannelsLastConv3dFunction`` / ``ChannelsLastConvTranspose3dFunction`` on the
        channels-last rows the voxel head hands over; BatchNorm with batch statistics on ``sgc_bn_rows_*``, the ReLU behind it and
        the ResBlock's identity addition inside the same passes (``conv_plan.bn_rows(residual=, relu=)``).  Same chain as
        imvoxelnet.py:22-34,146-173."""
        rows, grid = to_channels_last_rows(x)
        skips = []
        for i in range(self.n_scales):
            for blk in getattr(self, f"down_layer_{i}"):
                h, g1 = conv_rows(blk.conv1, rows, grid)
                h = bn_rows(blk.norm1, h, g1, relu=True)
                o, _ = conv_rows(blk.conv2, h, g1)
                if blk.stride != 1:
                    skip, _ = conv_rows(blk.downsample[0], rows, grid)
                    skip = bn_rows(blk.downsample[1], skip, g1)
                else:
                    skip = rows
                rows, grid = bn_rows(blk.norm2, o, g1, residual=skip, relu=True), g1      # relu(norm2(conv2) + identity)
            skips.append((rows, grid))
        outs = []
        for i in reversed(range(self.n_scales)):
            if i < self.n_scales - 1:
                up = getattr(self, f"up_block_{i + 1}")
                h, g = conv_transpose_rows(up[0], rows, grid)
                h = bn_rows(up[1], h, g, relu=True)
                h, _ = conv_rows(up[3], h, g)
                h = bn_rows(up[4], h, g, relu=True)
                rows, grid = skips[i][0] + h, g
            ob = getattr(self, f"out_block_{i}")
            o, _ = conv_rows(ob[0], rows, grid)
            o = bn_rows(ob[1], o, grid, relu=True)
            outs.append(rows_to_ncdhw(o, grid, o.shape[1]))
        return outs[::-1]

    def forward(self, x, tail_masks=None):
        """[1,C,nx,ny,nz] -> [out@1x, out@1/2, out@1/4], finest first (imvoxelnet.py:22-34)."""
        if not self.training and not torch.is_grad_enabled() and x.is_cuda and x.shape[0] == 1:
            return self._forward_hip(x, tail_masks)
        if train_conv_on_hip(x, [m.in_channels for m in self.modules() if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d))]):
            return self._forward_autograd_hip(x)
        # library convolutions: the voxel head hands over a channels-last strided view, for which MIOpen has only its
        # naive_conv_*_nonpacked kernels (2.6 s per config-2 step instead of ~0.1 s)
        x = x.contiguous()
        skips = []
        for i in range(self.n_scales):
            x = getattr(self, f"down_layer_{i}")(x)
            skips.append(x)
        outs = []
        for i in reversed(range(self.n_scales)):
            if i < self.n_scales - 1:
                x = skips[i] + getattr(self, f"up_block_{i + 1}")(x)
            outs.append(getattr(self, f"out_block_{i}")(x))
        return outs[::-1]

    def init_weights(self):
        pass
