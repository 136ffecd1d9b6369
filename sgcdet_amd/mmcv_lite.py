"""The slice of mmcv / mmdet the SGCDet hot path leans on, restated without mmcv.

The reference builds every hot-path module through mmcv registries from python-dict
configs (SURVEY.md section 8b "registry surface").  mmcv-full 1.5.3 / mmdet 2.25.1 are
third-party and not vendored under /root/reference; what is restated here follows their
published behaviour and is exercised only through the reference's
call sites:

* ``Registry`` / ``build_from_cfg``            -- ``type=`` dispatch with kwargs
* ``Config.fromfile``                          -- python-file configs (``configs/*.py``)
* ``BaseModule``, ``ModuleList``, ``Sequential``
* ``FFN``                                       -- TU/custom_base_transformer_layer.py:137-151
* ``build_norm_layer(dict(type='LN'), C)``      -- :153-156
* ``TransformerLayerSequence``                  -- TU/encoder.py:19
* ``Scale``                                     -- dense_heads/imvoxel_head_v2.py:79
* ``auto_fp16`` / ``force_fp32``                -- no-ops (fp16_enabled is False everywhere)
* ``xavier_init`` / ``constant_init`` / ``normal_init`` / ``bias_init_with_prob``
"""
import copy
import math
import os
import types

import torch
import torch.nn as nn


# --------------------------------------------------------------------------- registry
class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)

    def __repr__(self):
        return f"Registry({self.name}: {sorted(self._modules)})"


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, got {type(cfg)}")
    if "type" not in cfg:
        raise KeyError(f'`cfg` must contain the key "type", got {cfg}')
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    else:
        cls = obj_type
    return cls(**args)


# registries named as the reference names them (mmdet.models / mmcv.cnn.bricks.registry)
DETECTORS = Registry("detector")
HEADS = Registry("head")
NECKS = Registry("neck")
LOSSES = Registry("loss")
ATTENTION = Registry("attention")
FEEDFORWARD_NETWORK = Registry("feed-forward network")
TRANSFORMER_LAYER = Registry("transformerLayer")
TRANSFORMER_LAYER_SEQUENCE = Registry("transformer-layers sequence")
TRANSFORMER = Registry("Transformer")


def build_head(cfg):
    return HEADS.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_detector(cfg):
    return DETECTORS.build(cfg)


def build_attention(cfg, default_args=None):
    return build_from_cfg(cfg, ATTENTION, default_args)


def build_feedforward_network(cfg, default_args=None):
    return build_from_cfg(cfg, FEEDFORWARD_NETWORK, default_args)


def build_transformer_layer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)


def build_transformer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER, default_args)


# --------------------------------------------------------------------------- config
class ConfigDict(dict):
    """dict with attribute access (``cfg.model.voxel_head``), nested."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


class Config(ConfigDict):
    """``Config.fromfile('configs/SGCDet_ScanNet.py')`` for python-file configs.

    The file is executed and its public, picklable module-level names become the config
    (what ``mmcv.Config.fromfile`` does for ``.py`` files; the reference's four configs use
    no ``_base_`` inheritance).  ``main.py:33-37`` then merges the CLI namespace with
    ``cfg.update(vars(args))`` -- plain ``dict.update`` works here as well.
    """

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        scope = {"__file__": filename, "__name__": "_sgc_config_"}
        with open(filename, "r") as f:
            code = compile(f.read(), filename, "exec")
        exec(code, scope)
        cfg = {}
        for k, v in scope.items():
            if k.startswith("__") or isinstance(v, (types.ModuleType, types.FunctionType, type)):
                continue
            cfg[k] = v
        if "_base_" in cfg:
            base = cfg.pop("_base_")
            merged = {}
            for b in ([base] if isinstance(base, str) else base):
                merged.update(Config.fromfile(os.path.join(os.path.dirname(filename), b)))
            merged.update(cfg)
            cfg = merged
        out = Config(_wrap(cfg))
        dict.__setattr__(out, "filename", filename)
        return out


# --------------------------------------------------------------------------- modules
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        # mmcv recurses into children that define init_weights
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights()
        self._is_init = True


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def auto_fp16(apply_to=None, out_fp32=False):
    def deco(fn):
        return fn
    return deco


def force_fp32(apply_to=None, out_fp16=False):
    def deco(fn):
        return fn
    return deco


def constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    assert distribution in ("uniform", "normal")
    if hasattr(module, "weight") and module.weight is not None:
        if distribution == "uniform":
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.normal_(module.weight, mean, std)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))


_ACT = {"ReLU": nn.ReLU, "GELU": nn.GELU, "LeakyReLU": nn.LeakyReLU, "Sigmoid": nn.Sigmoid, "Tanh": nn.Tanh}


def build_activation_layer(cfg):
    args = dict(cfg)
    return _ACT[args.pop("type")](**args)


def build_norm_layer(cfg, num_features, postfix=""):
    args = dict(cfg)
    t = args.pop("type")
    args.pop("requires_grad", None)
    if t == "LN":
        args.setdefault("eps", 1e-5)
        return "ln" + str(postfix), nn.LayerNorm(num_features, **args)
    if t in ("BN", "BN2d"):
        return "bn" + str(postfix), nn.BatchNorm2d(num_features, **args)
    if t == "BN3d":
        return "bn" + str(postfix), nn.BatchNorm3d(num_features, **args)
    raise KeyError(f"norm layer {t} is not supported by mmcv_lite")


@FEEDFORWARD_NETWORK.register_module()
class FFN(BaseModule):
    """mmcv-full 1.5.3 ``FFN``; state-dict keys ``layers.0.0.{weight,bias}``, ``layers.1.{weight,bias}``."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type="ReLU", inplace=True), ffn_drop=0.0, dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(nn.Linear(in_channels, feedforward_channels),
                                     build_activation_layer(act_cfg), nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = ModuleList()
        for i in range(num_layers):
            self.layers.append(build_transformer_layer(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def multi_apply(func, *args, **kwargs):
    """mmdet.core.multi_apply: map then transpose the result tuples."""
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    results = map(pfunc, *args)
    return tuple(map(list, zip(*results)))
