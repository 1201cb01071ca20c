"""Backbones the GDRNPP configs name, re-declared without timm (not installable here).

``timm/convnext_base`` (configs/gdrn/ycbv/convnext_a6_…ycbv.py:74-84) and ``timm/resnet34``
(configs/_base_/gdrn_base.py:28-34) are created in the reference by
``timm.create_model(model_name=…, features_only=True, out_indices=…)``
(core/utils/timm_utils.py:34, models/net_factory.py:73-74).  timm==0.6.7 is a third-party
dependency outside /root/reference; the module/parameter names below follow its published
``FeatureListNet(flatten_sequential=True)`` naming (``stem_0``, ``stem_1``, ``stages_0…3`` /
``conv1``, ``bn1``, ``layer1…4``) so that GDRNPP checkpoints (``backbone.*`` keys,
GDRN_double_mask.py:39-43) load by name.  The mapping is UNVERIFIED against a real checkpoint
(none is available offline) — see DESIGN.md.

Layout: activations are kept channels-last in memory end to end, so the ConvNeXt MLPs are
plain row-major GEMMs on [N*H*W, C] (hipBLASLt) with no permute copies and the depthwise 7x7
runs in MIOpen's NHWC path.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_layers


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel dim of an NCHW tensor (timm.models.layers.LayerNorm2d)."""

    def forward(self, x):
        return hip_layers.layernorm2d(self, x)  # NHWC HIP kernel on the GPU, F.layer_norm on a permuted view otherwise


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class ConvNeXtBlock(nn.Module):
    """dwconv7x7 -> LN -> Linear(4x) -> GELU -> Linear -> layer-scale -> residual."""

    def __init__(self, dim, ls_init_value=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, 4 * dim)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))
        self._cache = {}  # tap-major depthwise weights for the HIP kernel (inference only)

    def forward(self, x):
        shortcut = x
        rows = hip_layers.mlp_takes_rows(self.mlp, self.conv_dw, x, self._cache)   # fc1's operand handed over already split (HIP path)
        x = hip_layers.dwconv_ln(self.conv_dw, self.norm, x, self._cache, y_rows=rows)  # NHWC view, LayerNorm applied
        # Mlp + layer scale + residual: shortcut + gamma * fc2(gelu(fc1(x)))  (timm: x.mul(gamma); drop_path(x) + shortcut)
        x = hip_layers.convnext_mlp(self.mlp, self.gamma, x, shortcut.permute(0, 2, 3, 1), self._cache, a_rows=rows)
        return x.permute(0, 3, 1, 2)


class ConvNeXtStage(nn.Module):
    def __init__(self, in_chs, out_chs, stride, depth):
        super().__init__()
        if in_chs != out_chs or stride > 1:
            self.downsample = nn.Sequential(LayerNorm2d(in_chs, eps=1e-6),
                                            nn.Conv2d(in_chs, out_chs, kernel_size=stride, stride=stride))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[ConvNeXtBlock(out_chs) for _ in range(depth)])

    def forward(self, x):
        if isinstance(self.downsample, nn.Sequential):   # LayerNorm2d -> 2x2 / stride-2 conv (implicit GEMM on the GPU)
            x = hip_layers.conv2d(self.downsample[1], self.downsample[0](x))
        return self.blocks(x)


class ConvNeXtFeatures(nn.Module):
    """timm ConvNeXt wrapped by FeatureListNet: returns a list with the maps of ``out_indices``."""

    def __init__(self, depths=(3, 3, 27, 3), dims=(128, 256, 512, 1024), in_chans=3, out_indices=(3,)):
        super().__init__()
        self.out_indices = tuple(out_indices)
        self.stem_0 = nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4)
        self.stem_1 = LayerNorm2d(dims[0], eps=1e-6)
        prev = dims[0]
        for i in range(max(self.out_indices) + 1):  # FeatureListNet prunes modules after the last used one
            setattr(self, f"stages_{i}", ConvNeXtStage(prev, dims[i], 2 if i > 0 else 1, depths[i]))
            prev = dims[i]
        self.num_features = prev

    def forward(self, x):
        x = hip_layers.stem(self.stem_0, self.stem_1, x)   # fused conv + bias + LayerNorm2d on the GPU
        feats = []
        for i in range(max(self.out_indices) + 1):
            x = getattr(self, f"stages_{i}")(x)
            if i in self.out_indices:
                feats.append(x)
        return feats


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes))

    def forward(self, x):
        if hip_layers.enabled_for(x) and not self.training:
            # inference on the GPU: BatchNorms folded into the convolutions, one elementwise kernel behind each of the two
            # 3x3 convolutions (bias + ReLU; bias + shortcut + ReLU); the downsample branch's folded bias joins conv2's
            sc, sc_bias = x, None
            if self.downsample is not None:
                _, sc_bias = hip_layers.folded_conv_bn(self.downsample[0], self.downsample[1])
                sc = hip_layers.folded_conv(self.downsample[0], self.downsample[1], x)
            y = hip_layers.conv_bn_act(self.conv1, self.bn1, x, relu=True)
            return hip_layers.conv_bn_act(self.conv2, self.bn2, y, relu=True, resid=sc, extra_bias=sc_bias)
        sc = x if self.downsample is None else self.downsample(x)
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.bn2(self.conv2(x))
        return self.act2(x + sc)


class ResNetFeatures(nn.Module):
    """timm resnet34, features_only: conv1/bn1/act1/maxpool/layer1..4 (out_indices=(4,) -> layer4, 512 ch)."""

    def __init__(self, layers=(3, 4, 6, 3), in_chans=3, out_indices=(4,)):
        super().__init__()
        self.out_indices = tuple(out_indices)
        self.conv1 = nn.Conv2d(in_chans, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.act1 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for i, (n, planes) in enumerate(zip(layers, (64, 128, 256, 512))):
            blocks = []
            for j in range(n):
                blocks.append(BasicBlock(inpl, planes, stride=2 if (j == 0 and i > 0) else 1))
                inpl = planes
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.num_features = 512

    def forward(self, x):
        x = x.contiguous(memory_format=torch.channels_last)
        feats = []
        if hip_layers.enabled_for(x) and not self.training:
            x = hip_layers.conv_bn_act(self.conv1, self.bn1, x, relu=True)
        else:
            x = self.act1(self.bn1(self.conv1(x)))
        if 0 in self.out_indices:
            feats.append(x)
        hip_layers.foreign("ResNetFeatures: max-pooling as a PyTorch operator", x)
        x = self.maxpool(x)
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
            if i in self.out_indices:
                feats.append(x)
        return feats


def create_backbone(type: str, in_chans=3, features_only=True, out_indices=None, pretrained=False, **_unused):
    """BACKBONES[type](**init_cfg) of models/net_factory.py:73-74.  ``pretrained`` cannot be honoured
    offline (no network / no weights): parameters keep their default initialisation."""
    name = type.split("/")[-1]
    if name == "convnext_base":
        return ConvNeXtFeatures((3, 3, 27, 3), (128, 256, 512, 1024), in_chans, out_indices or (3,))
    if name == "convnext_tiny":
        return ConvNeXtFeatures((3, 3, 9, 3), (96, 192, 384, 768), in_chans, out_indices or (3,))
    if name == "convnext_small":
        return ConvNeXtFeatures((3, 3, 27, 3), (96, 192, 384, 768), in_chans, out_indices or (3,))
    if name == "convnext_large":
        return ConvNeXtFeatures((3, 3, 27, 3), (192, 384, 768, 1536), in_chans, out_indices or (3,))
    if name in ("resnet34", "tv_resnet34"):
        return ResNetFeatures((3, 4, 6, 3), in_chans, out_indices or (4,))
    if name == "resnet18":
        return ResNetFeatures((2, 2, 2, 2), in_chans, out_indices or (4,))
    raise KeyError(f"backbone {type!r} is not re-declared in this build")
