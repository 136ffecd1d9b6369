"""``DepthNet_Fusion``: the producer of the depth distribution the view transform consumes.

Reference: mmdet3d_plugin/models/im2voxel/depth_utils/depth_est_fusion.py:166-329 (``DepthNet_Fusion``), :129-164
(``ConvBnReLU2D``, ``SimpleUnet2D``), depth_utils/extractor_matching.py:7-89 (``ResNetFPN``: the 1/4-resolution
matching-feature extractor) and depth_utils/layer_matching.py:107-134 (``BasicBlock``, ``conv1x1`` / ``conv3x3``).
Same ``type=`` name, constructor arguments, call signature ``forward(xs, imgs, img_metas, stride) -> [B, N, D, H, W]``
and state-dict keys, so released checkpoints load (``fnet_mvs.layer2.0.bn3`` and ``...downsample.1`` are the same
BatchNorm registered twice, exactly as in the reference).

What differs on MI355X: in inference the plane-sweep cost volume (homography warp of the K neighbour views at D depth
planes + correlation, :229-240) is ONE fused HIP kernel (``sgc_plane_sweep_corr``, csrc/plane_sweep.hip) -- the warped
features [N, C, D, H, W] (1.18 GB per neighbour at config 2) never exist.  With autograd enabled the reference's
``F.grid_sample`` formulation runs (differentiable, same numbers).  The 2-D convolutions are library convolutions
(MIOpen through torch): dense 2-D CNNs are not part of the hot path.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..mmcv_lite import HEADS
from .plane_sweep import closest_frame_ids, plane_sweep_correlation


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=0)


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1)


class BasicBlock(nn.Module):
    """layer_matching.py:107-134: two 3x3 conv + BN + ReLU, ReLU(x + y) with a 1x1 / BN shortcut when the shape changes."""

    def __init__(self, in_planes, planes, stride=1, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = conv3x3(in_planes, planes, stride)
        self.conv2 = conv3x3(planes, planes)
        self.bn1 = norm_layer(planes)
        self.bn2 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        if stride == 1 and in_planes == planes:
            self.downsample = None
        else:
            self.bn3 = norm_layer(planes)
            self.downsample = nn.Sequential(conv1x1(in_planes, planes, stride=stride), self.bn3)

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class ResNetFPN(nn.Module):
    """extractor_matching.py:7-89: ResNet-18 stem + layer1 (1/2) + layer2 (1/4) + 1x1 to ``output_dim``."""

    def __init__(self, input_dim=3, output_dim=256, ratio=1.0, norm_layer=nn.BatchNorm2d, init_weight="ImageNet"):
        super().__init__()
        block_dims = [int(d * ratio) for d in (64, 128, 256)]
        self.init_weight = init_weight
        self.input_dim = input_dim
        self.in_planes = 64
        self.conv1 = nn.Conv2d(input_dim, 64, kernel_size=7, stride=2, padding=3)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block_dims[0], 1, norm_layer, 2)
        self.layer2 = self._make_layer(block_dims[1], 2, norm_layer, 2)
        self.final_conv_3ddet = conv1x1(block_dims[1], output_dim)
        self._init_weights()

    def _make_layer(self, dim, stride, norm_layer, num):
        layers = [BasicBlock(self.in_planes, dim, stride=stride, norm_layer=norm_layer)]
        layers += [BasicBlock(dim, dim, stride=1, norm_layer=norm_layer) for _ in range(num - 1)]
        self.in_planes = dim
        return nn.Sequential(*layers)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        if self.init_weight == "ImageNet":
            # the reference copies torchvision's pretrained resnet18 tensors whose names match (:53-64); torchvision and its
            # weight download are not available in this image -- checkpoints are expected to carry these tensors
            try:
                from torchvision.models import resnet18
                pretrained = resnet18(pretrained=True).state_dict()
                own = self.state_dict()
                own.update({k: v for k, v in pretrained.items() if k in own and v.shape == own[k].shape})
                self.load_state_dict(own, strict=False)
            except Exception:       # noqa: BLE001  (no torchvision / no network: keep the kaiming init)
                pass

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.layer2(self.layer1(x))
        return self.final_conv_3ddet(x)


class ConvBnReLU2D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class SimpleUnet2D(nn.Module):
    """depth_est_fusion.py:139-163: two stride-2 stages down, two transposed 3x3 stages up, additive skips."""

    def __init__(self, in_channel):
        super().__init__()
        c = in_channel
        self.conv1 = ConvBnReLU2D(c, 2 * c, stride=2)
        self.conv2 = ConvBnReLU2D(2 * c, 2 * c)
        self.conv3 = ConvBnReLU2D(2 * c, 4 * c, stride=2)
        self.conv4 = ConvBnReLU2D(4 * c, 4 * c)
        self.conv9 = nn.Sequential(
            nn.ConvTranspose2d(4 * c, 2 * c, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
            nn.BatchNorm2d(2 * c), nn.ReLU(inplace=True))
        self.conv11 = nn.Sequential(
            nn.ConvTranspose2d(2 * c, c, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
            nn.BatchNorm2d(c), nn.ReLU(inplace=True))

    def forward(self, x):
        conv0 = x
        conv2 = self.conv2(self.conv1(conv0))
        x = self.conv4(self.conv3(conv2))
        x = conv2 + self.conv9(x)
        return conv0 + self.conv11(x)


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    """Differentiable warp of the neighbour features onto the depth planes of the reference view (:87-126):
    src_fea [B,C,H,W], projections [B,4,4], depth_values [B,D] -> [B,C,D,H,W].  Used only under autograd; inference
    runs the fused kernel."""
    batch, channels, height, width = src_fea.shape
    num_depth = depth_values.shape[1]
    with torch.no_grad():
        proj = torch.matmul(src_proj, torch.inverse(ref_proj))
        rot, trans = proj[:, :3, :3], proj[:, :3, 3:4]
        y, x = torch.meshgrid([torch.arange(0, height, dtype=torch.float32, device=src_fea.device),
                               torch.arange(0, width, dtype=torch.float32, device=src_fea.device)], indexing="ij")
        xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(height * width, device=src_fea.device)))
        rot_xyz = torch.matmul(rot, xyz.unsqueeze(0).repeat(batch, 1, 1))
        rot_depth_xyz = rot_xyz.unsqueeze(2).repeat(1, 1, num_depth, 1) * depth_values.view(batch, 1, num_depth, 1)
        proj_xyz = rot_depth_xyz + trans.view(batch, 3, 1, 1)
        proj_xy = proj_xyz[:, :2] / proj_xyz[:, 2:3]
        grid = torch.stack((proj_xy[:, 0] / ((width - 1) / 2) - 1, proj_xy[:, 1] / ((height - 1) / 2) - 1), dim=3)
    warped = F.grid_sample(src_fea, grid.view(batch, num_depth * height, width, 2), mode="bilinear", padding_mode="zeros",
                           align_corners=False)
    return warped.view(batch, channels, num_depth, height, width)


@HEADS.register_module()
class DepthNet_Fusion(nn.Module):
    def __init__(self, neighbor_img_num, downsample_factor, dbound, mono_channels=256, loss_weight=0.5, max_tol=0,
                 init_weight="ImageNet"):
        super().__init__()
        self.fp16_enabled = False
        self.max_tol = max_tol
        self.downsample_factor = downsample_factor
        self.loss_weight = loss_weight
        self.neighbor_img_num = neighbor_img_num
        self.dbound = dbound
        self.depth_channels = round((dbound[1] - dbound[0]) / dbound[2])
        self.depth_values = np.arange(dbound[0], dbound[1], dbound[2], dtype=np.float32) + dbound[2] / 2   # bin centres
        self.fnet_mvs = ResNetFPN(input_dim=3, output_dim=128, ratio=1.0, norm_layer=nn.BatchNorm2d, init_weight=init_weight)
        self.correlation_regulation = SimpleUnet2D(in_channel=self.depth_channels)
        self.fnet_mono = ConvBnReLU2D(in_channels=mono_channels, out_channels=128)
        self.mono_regulation = SimpleUnet2D(in_channel=128)
        self.fusion_regulation = SimpleUnet2D(in_channel=self.depth_channels + 128)
        self.depth_reg = nn.Conv2d(self.depth_channels + 128, self.depth_channels, kernel_size=3, stride=1, padding=1)

    def correlation(self, f_mvs, img_meta, stride):
        """Plane-sweep matching cost [N, D, H, W] of the N views against their time-adjacent neighbours (:219-240)."""
        num_src, channel_num, H, W = f_mvs.shape
        k = min(self.neighbor_img_num, num_src - 1)
        if f_mvs.is_cuda and not (torch.is_grad_enabled() and f_mvs.requires_grad):
            return plane_sweep_correlation(f_mvs, img_meta, stride, self.depth_values, self.neighbor_img_num)
        dev = f_mvs.device
        src_w2c = torch.tensor(np.array(img_meta["lidar2img"]["extrinsic"]), device=dev)
        intr = torch.tensor(np.array(img_meta["lidar2img"]["intrinsic"]), device=dev).clone()
        ratio = img_meta["ori_shape"][0] / (img_meta["img_shape"][0] / stride)
        if intr.dim() == 2:
            intr[:2] /= ratio
            intr = intr.unsqueeze(0).repeat(num_src, 1, 1)
        else:
            intr[:, :2] /= ratio
        neighbor_ids = closest_frame_ids(num_src, k)
        proj = torch.matmul(intr, src_w2c)
        depth_values = torch.tensor(self.depth_values, device=dev).unsqueeze(0).repeat(num_src, 1)
        corr = torch.zeros((num_src, self.depth_channels, H, W), device=dev)
        for j in range(k):
            warped = homo_warping(f_mvs[neighbor_ids[:, j]], proj[neighbor_ids[:, j]], proj, depth_values)
            corr = corr + (warped * f_mvs.unsqueeze(2)).sum(dim=1) / torch.sqrt(torch.tensor(channel_num).float())
        return corr / k

    def forward(self, xs, imgs, img_metas, stride):
        """xs [B,N,C,H,W] (finest FPN map), imgs [B,N,3,4H,4W], img_metas list of B dicts -> depth probability
        [B,N,D,H,W] (softmax over the D bins)."""
        B, num_src, C, H, W = xs.shape
        depth_preds = torch.empty((B, num_src, self.depth_channels, H, W), device=xs.device)
        for b, (x, img, img_meta) in enumerate(zip(xs, imgs, img_metas)):
            f_mvs = self.fnet_mvs(img)
            cost_reg = self.correlation_regulation(self.correlation(f_mvs, img_meta, stride))
            mono_reg = self.mono_regulation(self.fnet_mono(x))
            prob = self.depth_reg(self.fusion_regulation(torch.cat([cost_reg, mono_reg], dim=1)))
            depth_preds[b] = F.softmax(prob, dim=1)
        return depth_preds

    def get_downsampled_gt_depth(self, gt_depths):
        """[B,N,H,W] metric depth maps -> [B*N*h*w, D] one-hot bins at feature resolution (:254-296): min over every
        ds x ds patch ignoring zeros, bin index (d - (near - step)) / step, out-of-range -> no bin."""
        ds = self.downsample_factor
        if ds % 1 == 0:
            ds = int(ds)
            B, N, H, W = gt_depths.shape
            g = gt_depths.view(B * N, H // ds, ds, W // ds, ds, 1).permute(0, 1, 3, 5, 2, 4).contiguous().view(-1, ds * ds)
            g = torch.where(g == 0.0, 1e5 * torch.ones_like(g), g)
            g = torch.min(g, dim=-1).values.view(B * N, H // ds, W // ds)
        else:
            g = F.interpolate(gt_depths, scale_factor=1 / ds, mode="nearest")
            B, N, H, W = g.shape
            g = g.view(B * N, H, W)
        g = (g - (self.dbound[0] - self.dbound[2])) / self.dbound[2]
        g = torch.where((g < self.depth_channels + 1) & (g >= 0.0), g, torch.zeros_like(g))
        onehot = F.one_hot(g.long(), num_classes=self.depth_channels + 1).view(-1, self.depth_channels + 1)[:, 1:]
        return self.error_tol(onehot).float()

    def error_tol(self, onehot_):
        if self.max_tol < 1:
            return onehot_
        padding = onehot_.new_zeros(onehot_.shape[0], 1)
        onehot = onehot_.clone()
        for error in range(-self.max_tol, self.max_tol + 1):
            if error < 0:
                onehot = onehot + torch.cat([onehot[..., 1:], padding], dim=-1)
            elif error > 0:
                onehot = onehot + torch.cat([padding, onehot[..., :-1]], dim=-1)
        return onehot / (onehot + 1e-5)

    def loss(self, depth_labels, depth_preds):
        """BCE between the predicted distribution and the one-hot depth bin on pixels that have a label (:314-329)."""
        if depth_labels.dim() == 3:
            depth_labels = depth_labels.unsqueeze(0)
        labels = self.get_downsampled_gt_depth(depth_labels)
        preds = depth_preds.float().permute(0, 1, 3, 4, 2).contiguous().view(-1, self.depth_channels)
        fg = torch.max(labels, dim=1).values > 0.0
        preds = torch.clamp(preds, min=1e-7, max=1 - 1e-7)
        depth_loss = F.binary_cross_entropy(preds[fg], labels[fg], reduction="none").sum() / max(1.0, fg.sum())
        return {"loss_dpt": self.loss_weight * depth_loss}
