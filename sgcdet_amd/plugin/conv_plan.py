"""Eval-mode lowering of the neck / head convolution chains onto ``sgc_conv3d_cl_f32``.

BatchNorm3d (running statistics) is folded into a per-channel scale/shift applied in the
kernel's epilogue, weights are permuted once to the kernel's ``[tap][Cout][Cin]`` layout, and
channel counts are zero-padded to multiples of 32 (the K tile) -- a no-op for every SGCDet
config (128 ... 1024 channels).  Plans are cached per module and rebuilt when a parameter or
buffer changes (``Tensor._version``) or moves device.
"""
import os

import torch

from .. import ext

_PAD = 32

# "bf16x3": fp32 operands split hi/lo onto the bf16 matrix cores (fp32-faithful to ~1e-5, default);
# "f32": exact fp32 products on the fp32 MFMA (5x slower matrix pipe).  Both are tested against the oracle.
# "bf16" / "fp16" (opt-in, BASELINE.json configs #2 "bf16" / #5 "fp16"): the bf16x3 code path with ONE product per multiply-add
# -- both operands rounded to bfloat16 (sgc_set_conv_products(1)) or to IEEE half (2; v_mfma_f32_32x32x16_f16: the same rate,
# 11 instead of 8 significant bits, operands saturated at +-65504), fp32 accumulate; 1/3 of the matrix work, NOT parity-exact:
# their own bench lines, never the headline.  CONV_MODE stays "bf16x3" (same kernels, same launch sequence); CONV_PRODUCTS says
# which.  Call set_conv_mode BEFORE the modules prepare their weight plans (the planes hold bf16 or half bits accordingly).
CONV_MODE = "bf16x3"
CONV_PRODUCTS = 3


def set_conv_mode(mode):
    global CONV_MODE, CONV_PRODUCTS
    if mode not in ("bf16x3", "f32", "bf16", "fp16"):
        raise ValueError(mode)
    CONV_MODE = "bf16x3" if mode in ("bf16", "fp16") else mode
    CONV_PRODUCTS = {"bf16": 1, "fp16": 2}.get(mode, 3)
    import torch
    if torch.cuda.is_available():
        ext.ops().lib.call("sgc_set_conv_products", CONV_PRODUCTS)


# Winograd F(2,3) along z for the wide 3x3x3 stride-1 layers (sgc_conv3d_winograd_z_bf16x3): 2/3 of the multiply-adds of
# the layers that hold the chip at its power limit, at the price of a transform pass on either side.  "auto" = wherever the entry
# point supports the shape (z extent a multiple of 8, x / y multiples of 8 so that the 8 x 8 pixel bricks tile the slices) and the
# layer has at least WINOGRAD_Z_MIN_CH input channels; False = never.  Results differ from the direct kernel by summation order
# (<= 2e-5 of the tensor scale, tested); each choice is deterministic.  Set before the first forward: plans are cached.
WINOGRAD_Z = {"0": False, "1": True}.get(os.environ.get("SGC_WINOGRAD_Z", ""), "auto")
WINOGRAD_Z_MIN_CH = int(os.environ.get("SGC_WINOGRAD_Z_MIN_CH", "256"))
WINOGRAD_Z_RAGGED = os.environ.get("SGC_WINOGRAD_Z_RAGGED", "1") != "0"     # also slices the 8 x 8 pixel bricks do not tile exactly (20 x 20)
# layers that keep the direct kernel although "auto" would give them the form: (Cin, Cout, gx, gy, gz) tuples, from per-layer A/B runs
# with scenes in flight (profiles/r06_winograd_layers_ab.txt).  Env (A/B runs): SGC_WINOGRAD_Z_DENY="512:128:20:20:8,512:512:20:20:8"
WINOGRAD_Z_DENY = {tuple(int(v) for v in item.split(":")) for item in os.environ.get("SGC_WINOGRAD_Z_DENY", "").split(",") if item}


def set_winograd_z(mode, min_channels=None):
    global WINOGRAD_Z, WINOGRAD_Z_MIN_CH
    if mode not in ("auto", True, False):
        raise ValueError(mode)
    WINOGRAD_Z = mode
    if min_channels is not None:
        WINOGRAD_Z_MIN_CH = int(min_channels)


def _pad_to(n, m=_PAD):
    return (n + m - 1) // m * m


class ConvSpec:
    """One prepared convolution: wt [taps, Cout_p, Cin_p], scale/shift [Cout_p]."""

    def __init__(self, weight, bn=None, bias=None, ksize=3, stride=1, transposed=False, pad_out=True):
        w = weight.detach().float()
        if transposed:                       # nn.ConvTranspose3d weight [Cin, Cout, 2, 2, 2]
            cin, cout = w.shape[0], w.shape[1]
            wt = w.permute(2, 3, 4, 1, 0).reshape(8, cout, cin)
        else:                                # nn.Conv3d weight [Cout, Cin, k, k, k]
            cout, cin = w.shape[0], w.shape[1]
            wt = w.permute(2, 3, 4, 0, 1).reshape(ksize ** 3, cout, cin)
        cin_p = _pad_to(cin)
        cout_p = _pad_to(cout) if pad_out else _pad_to(cout, 4)
        wp = torch.zeros((wt.shape[0], cout_p, cin_p), dtype=torch.float32, device=w.device)
        wp[:, :cout, :cin] = wt
        scale = torch.ones(cout_p, dtype=torch.float32, device=w.device)
        shift = torch.zeros(cout_p, dtype=torch.float32, device=w.device)
        if bn is not None:                   # y = (x - mean) / sqrt(var + eps) * gamma + beta
            s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            scale[:cout] = s
            shift[:cout] = bn.bias.detach().float() - bn.running_mean.detach().float() * s
        if bias is not None:
            shift[:cout] += bias.detach().float() * scale[:cout]
        self.wt, self.scale, self.shift = wp.contiguous(), scale, shift
        self.w_hi, self.w_lo = ext.ops().split_operand(self.wt) if w.is_cuda else (None, None)
        self.cin, self.cout, self.cin_p, self.cout_p = cin, cout, cin_p, cout_p
        self.ksize, self.stride, self.transposed = ksize, stride, transposed
        self._wino = None                    # (g_hi, g_lo): transformed weight planes, built on first use

    def _winograd_planes(self, grid):
        """The transformed weight planes when this call should take the Winograd-z form, else None."""
        if (WINOGRAD_Z is False or self.ksize != 3 or self.stride != 1 or self.transposed or CONV_MODE != "bf16x3"
                or (WINOGRAD_Z == "auto" and self.cin_p < WINOGRAD_Z_MIN_CH)
                or (not WINOGRAD_Z_RAGGED and (grid[0] % 8 or grid[1] % 8))):
            return None
        if (self.cin_p, self.cout_p) + tuple(grid) in WINOGRAD_Z_DENY:
            return None
        ops = ext.ops()
        if not ops.conv3d_winograd_z_supported(grid, self.cin_p, self.cout_p):
            return None
        # The transform-domain launch has Z/8 x ceil(X/8) x ceil(Y/8) bricks per position x 4 positions x ceil(Cout/128) column tiles
        # and no reduction split.  A layer ALONE needs enough of them to fill the chip (512 -> 128 @ 20x20x8: 36 workgroups, 154 us
        # against 55 us direct); with scenes in flight the other streams fill it and the saved multiply-adds count (+2 % at config 2).
        if WINOGRAD_Z == "auto" and not WINOGRAD_IN_FLIGHT:
            wgs = (grid[2] // 8) * -(-grid[0] // 8) * -(-grid[1] // 8) * 4 * -(-self.cout_p // 128)
            if wgs < 192:
                return None
        if self._wino is None:
            self._wino = ops.split_operand(ops.winograd_z_weights(self.wt))
        return self._wino

    def __call__(self, x, grid, residual=None, relu=0, out_mask=None, act=None):
        """``out_mask`` (uint8 [OV], bf16x3 mode, 3x3x3 stride-1 layers): only rows with 1 are needed downstream.
        ``act`` = (c0, c1, scale tensor): columns [c0, c1) leave as exp(v * scale) (bf16x3 mode; the caller applies it itself in
        the strict-fp32 mode -- ``supports_act``)."""
        if CONV_MODE == "bf16x3":
            wino = self._winograd_planes(grid) if out_mask is None and act is None else None
            if wino is not None:
                return ext.ops().conv3d_winograd_z(x, wino[0], wino[1], grid, self.scale, self.shift, residual, relu)
            return ext.ops().conv3d_cl_bf16x3(x, self.w_hi, self.w_lo, grid, self.ksize, self.stride,
                                              self.transposed, self.scale, self.shift, residual, relu, out_mask=out_mask, act=act)
        if act is not None:
            raise NotImplementedError("the output activation lives in the bf16x3 convolution's epilogue")
        return ext.ops().conv3d_cl(x, self.wt, grid, self.ksize, self.stride, self.transposed, self.scale,
                                   self.shift, residual, relu)


# training / autograd path of the neck and head convolutions: "hip" = forward, input and weight gradients on the MFMA
# kernels (functions.ChannelsLastConv3dFunction), "library" = torch's convolutions (MIOpen) as in round 1
TRAIN_CONV = os.environ.get("SGC_TRAIN_CONV", "hip")
BN_ON_HIP = os.environ.get("SGC_BN_HIP", "1") != "0"      # training-mode BatchNorm of the neck on sgc_bn_rows_* (0: torch's kernels)
BN_FUSE_TAIL = os.environ.get("SGC_BN_FUSE_TAIL", "1") != "0"   # `+ identity` / ReLU behind a BatchNorm inside its passes (0: torch ops, A/B)


THROUGHPUT_GEOMETRY = False      # set_throughput_mode(True): scenes in flight, kernels sized for CU-time
WINOGRAD_IN_FLIGHT = False       # the Winograd-z gate's view of it: follows set_throughput_mode unless that call says keep_winograd=True


def set_throughput_mode(on, keep_winograd=False):
    """Launch geometry for several scenes in flight (one hipGraph per scene on its own stream, bench.py).  With four scenes
    overlapping the chip is CU-time bound -- the sum of workgroup residency of all kernels, not any kernel's latency, sets the
    throughput -- so kernels are sized for work per CU-second instead of for their own latency:
      * the persistent row GEMM runs on HALF the CUs with two tiles in flight per workgroup (alone: 92 -> 136 us on the
        204 800-row Linear, but 27 % less CU-time; same bits -- a row's result does not depend on the tiling of the call);
      * the layers with few voxels split their reductions over FEWER workgroups (tile kernel: until 256 instead of 512 workgroups,
        halo kernel: 96 instead of 192): less prologue / epilogue and workspace traffic per unit of work, 20 - 30 % more latency
        per layer.  A different split adds the same partial sums in a different order: seven convolution layers of the neck
        differ from the latency geometry by fp32 summation order (<= 1e-5 of the tensor scale, tested); each mode is
        deterministic and bit-identical between graph replays and eager launches.
    Together +3.5 % scenes/s (alternated runs).  Off = the latency-optimal geometry (one scene at a time, the default of the
    library).  Call it before the first scene: captured graphs keep the geometry they were captured with.
    ``keep_winograd``: the reduction splits change, the choice of the Winograd-z form per layer does NOT (bench.py's
    ``--eager-geometry latency`` pass must time the kernels the graphs replay, not another form of the small layers)."""
    global THROUGHPUT_GEOMETRY, WINOGRAD_IN_FLIGHT
    THROUGHPUT_GEOMETRY = bool(on)
    if not keep_winograd:
        WINOGRAD_IN_FLIGHT = bool(on)
    from .. import ext
    lib = ext.ops().lib
    explicit = os.environ.get("SGC_TUNE", "")
    for key, hot, cold in (("rows_cu_pct", 50, 100), ("rows_depth", 2, 1), ("split_target", 256, 512), ("halo_split_target", 96, 192)):
        if key + "=" not in explicit:
            lib.call("sgc_set_tuning", key.encode(), hot if on else cold)


def set_train_conv(mode):
    global TRAIN_CONV
    if mode not in ("hip", "library"):
        raise ValueError(mode)
    TRAIN_CONV = mode


def train_conv_on_hip(x, channels):
    """Can the autograd path of a conv chain run on the HIP kernels?  (one scene, CUDA, channel counts the K tile accepts)"""
    return (TRAIN_CONV == "hip" and CONV_MODE == "bf16x3" and train_products_ok() and x.is_cuda and x.shape[0] == 1
            and all(c % _PAD == 0 for c in channels))


def train_products_ok():
    """The autograd Functions pack their weight planes with ``sgc_pack_conv_weight``, which emits bfloat16 hi / lo bits; the
    forward and input-gradient kernels follow the process-wide arithmetic mode.  Modes 3 and 1 read those bits as bfloat16;
    mode 2 (fp16) would read them as IEEE half (bf16 1.0 = 0x3F80 is 1.875 as a half) -- so the fp16 mode is inference-only:
    with gradients enabled the convolutions / Linears take torch's library path (fp32) instead."""
    return CONV_PRODUCTS != 2


def conv_rows(conv, rows, grid):
    """``nn.Conv3d`` module (k in {1,3}, padding k//2) applied to channels-last rows with autograd on the HIP kernels."""
    from ..functions import ChannelsLastConv3dFunction
    k, s = conv.kernel_size[0], conv.stride[0]
    y = ChannelsLastConv3dFunction.apply(rows, conv.weight, tuple(grid), k, s)
    if conv.bias is not None:
        y = y + conv.bias
    return y, tuple((d + 2 * (k // 2) - k) // s + 1 for d in grid)


def conv_transpose_rows(conv, rows, grid):
    """``nn.ConvTranspose3d(2, 2, bias=False)`` on channels-last rows."""
    from ..functions import ChannelsLastConvTranspose3dFunction
    return ChannelsLastConvTranspose3dFunction.apply(rows, conv.weight, tuple(grid)), tuple(2 * d for d in grid)


def bn_rows(bn, rows, grid, residual=None, relu=False):
    """A BatchNorm3d-like module on channels-last rows [V, C], optionally followed by ``+ residual`` and ``relu`` (the tails of the
    neck's blocks; on the HIP path they run inside the normalisation passes).  ``nn.BatchNorm3d`` over [1, C, X, Y, Z] is the
    statistics of the V rows per channel, i.e. ``F.batch_norm`` on the [V, C] matrix (same running-statistics update); any other
    module (SyncBatchNorm3d, GroupNorm ...) gets the 5-D channels-last view."""
    def tail(y):
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    if type(bn) is torch.nn.BatchNorm3d:
        use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
        momentum = 0.0 if bn.momentum is None else bn.momentum
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
            if bn.momentum is None:
                momentum = 1.0 / float(bn.num_batches_tracked)
        if (use_batch and TRAIN_CONV == "hip" and BN_ON_HIP and rows.is_cuda and rows.dtype == torch.float32 and rows.shape[1] % 4 == 0
                and bn.weight is not None and bn.bias is not None and rows.shape[0] > 1):
            from ..functions import BatchNormRowsFunction
            track = bn.training and bn.track_running_stats
            fuse = BN_FUSE_TAIL and (residual is None or (residual.dtype == torch.float32 and residual.shape == rows.shape))
            y = BatchNormRowsFunction.apply(rows, bn.weight, bn.bias, bn.running_mean if track else None,
                                            bn.running_var if track else None, float(momentum), float(bn.eps),
                                            residual if fuse else None, bool(relu and fuse))
            return y if fuse else tail(y)
        return tail(torch.nn.functional.batch_norm(rows, bn.running_mean if not bn.training or bn.track_running_stats else None,
                                                   bn.running_var if not bn.training or bn.track_running_stats else None,
                                                   bn.weight, bn.bias, use_batch, momentum, bn.eps))
    y = bn(rows_to_ncdhw(rows, grid, rows.shape[1]))
    return tail(y[0].permute(1, 2, 3, 0).reshape(rows.shape[0], rows.shape[1]))


def module_fingerprint(module):
    """Cheap change detector for the cached plans: (storage address, in-place version) of every
    parameter / buffer.  The tensor list itself is cached on the module (rebuilt on .train()/.to())."""
    tensors = module.__dict__.get("_fp_tensors")
    if tensors is None or module.__dict__.get("_fp_training") != module.training:
        tensors = list(module.parameters()) + list(module.buffers())
        module.__dict__["_fp_tensors"] = tensors
        module.__dict__["_fp_training"] = module.training
    return (CONV_PRODUCTS,) + tuple((t.data_ptr(), t._version) for t in tensors)       # a mode switch rebuilds the weight planes


def to_channels_last_rows(x):
    """[1,C,X,Y,Z] (any strides) -> ([X*Y*Z, C_padded] contiguous rows, (X,Y,Z)); no copy when the
    tensor already is channels-last in memory and C % 32 == 0."""
    _, C, X, Y, Z = x.shape
    rows = x[0].permute(1, 2, 3, 0).reshape(X * Y * Z, C)          # view if channels-last, copy otherwise
    cp = _pad_to(C)
    if cp != C:
        rows = torch.nn.functional.pad(rows, (0, cp - C))
    return rows.contiguous().float(), (X, Y, Z)


def rows_to_ncdhw(rows, grid, channels):
    """[V, Cp] rows -> [1, channels, X, Y, Z] view (channels-last memory, no copy)."""
    X, Y, Z = grid
    return rows.view(X, Y, Z, rows.shape[1])[..., :channels].permute(3, 0, 1, 2).unsqueeze(0)


class BlockDiagSpec:
    """A block-diagonal Linear [G * K -> G * Nh] kept as its G blocks (``sgc_linear_rows_blockdiag_bf16x3``): group g's K inputs times
    its own [Nh, K] weight.  ``weight`` [G, Nh, K], ``bias`` [G * Nh]."""

    def __init__(self, weight, bias):
        w = weight.detach().float().contiguous()
        self.G, self.Nh, self.K = w.shape
        self.w_hi, self.w_lo = ext.ops().split_operand(w)
        self.shift = bias.detach().float().contiguous() if bias is not None else None

    @staticmethod
    def supported(G, K, Nh):
        return CONV_MODE == "bf16x3" and ext.ops().linear_rows_blockdiag_supported(G, K, Nh)

    def __call__(self, x, count=None):
        return ext.ops().linear_rows_blockdiag(x, self.w_hi, self.w_lo, self.shift, count=count)


class LinearSpec:
    """nn.Linear (or a row-block of stacked weights) prepared for ``sgc_conv3d_cl_bf16x3`` as a 1x1x1
    convolution over M rows: y[M, out] = x[M, in] @ W^T + b on the bf16 matrix cores with the 3-way split."""

    def __init__(self, weight, bias, useful=None):
        """``useful``: fraction of the weight matrix that is structurally non-zero (a block-diagonal matrix run as a dense
        GEMM); only bench.py's flop accounting reads it (the zero blocks are not algorithmic work)."""
        w = weight.detach().float()
        cout, cin = w.shape
        self.useful = useful
        if cin % _PAD:
            raise ValueError("LinearSpec needs in_features % 32 == 0")
        cout_p = _pad_to(cout, 4)
        wp = torch.zeros((1, cout_p, cin), dtype=torch.float32, device=w.device)
        wp[0, :cout] = w
        self.wt = wp
        self.w_hi, self.w_lo = ext.ops().split_operand(wp)
        self.shift = torch.zeros(cout_p, dtype=torch.float32, device=w.device)
        if bias is not None:
            self.shift[:cout] = bias.detach().float()
        self.cout, self.cout_p = cout, cout_p

    def __call__(self, x, extra_zero_row=False, count=None):
        """``extra_zero_row``: the result is a view of an [M+1, out] buffer whose last row is zero (the
        gather kernel points out-of-image corners at it).  ``count``: int32 device tensor with the number of
        live rows of ``x`` (its remaining rows are capacity): rows past it are neither read nor written."""
        M = x.shape[0]
        if CONV_MODE == "bf16x3" and self.cout_p == self.cout:
            return ext.ops().linear_rows_bf16x3(x, self.w_hi, self.w_lo, self.shift, count=count, useful=self.useful,
                                                zero_tail=bool(extra_zero_row) and M > 0)
        if count is not None:
            raise NotImplementedError("device-side row counts need the bf16x3 path and out_features % 4 == 0")
        if CONV_MODE == "bf16x3":
            y, _ = ext.ops().conv3d_cl_bf16x3(x, self.w_hi, self.w_lo, (M, 1, 1), 1, 1, False, None, self.shift)
        else:
            y, _ = ext.ops().conv3d_cl(x, self.wt, (M, 1, 1), 1, 1, False, None, self.shift)
        return y if self.cout_p == self.cout else y[:, :self.cout]

    def headmajor(self, x, N, S, M, out_dtype=torch.float32):
        """x [N*S, in] -> [N, M, S, out/M]: the same GEMM with the result stored head-major (the layout the LDS-tiled
        gather stages a head's window from); bf16x3 path only.  ``out_dtype=torch.bfloat16``: bf16 storage mode."""
        if CONV_MODE != "bf16x3" or self.cout_p != self.cout:
            raise NotImplementedError("head-major output needs the bf16x3 path and out_features % 4 == 0")
        return ext.ops().linear_rows_headmajor_bf16x3(x, self.w_hi, self.w_lo, self.shift, N, S, M, out_dtype=out_dtype)
