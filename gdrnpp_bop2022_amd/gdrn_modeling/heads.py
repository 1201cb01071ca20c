"""Geometry head and Patch-PnP head of GDRNPP, mirroring the reference modules by name:

* ``TopDownDoubleMaskXyzRegionHead`` / ``TopDownMaskXyzRegionHead``
  (models/heads/top_down_doublemask_xyz_region_head.py:9-211, top_down_mask_xyz_region_head.py)
* ``ConvPnPNet`` (models/heads/conv_pnp_net.py:10-183)
* ``ConvModule`` (lib/torch_utils/layers/conv_module.py:103-236): sub-modules ``conv``, ``gn``, ``activate``.

Parameter names equal the reference's so that ``geo_head_net.*`` / ``pnp_net.*`` checkpoint keys load
by name.  Inference only: initialisers follow the reference, losses/dropblock are not carried.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip_layers
from .. import hip_lib


def get_nn_act_func(act: str):
    """lib/torch_utils/layers/layer_utils.py:63-100 (subset used by the GDRNPP configs)."""
    a = act.lower()
    if a == "relu":
        return nn.ReLU(inplace=True)
    if a in ("lrelu", "leaky_relu", "leakyrelu"):
        return nn.LeakyReLU(negative_slope=0.1, inplace=True)
    if a == "gelu":
        return nn.GELU()
    if a in ("silu", "swish"):
        return nn.SiLU(inplace=True)
    if a == "mish":
        return nn.Mish(inplace=True)
    raise ValueError(f"Unknown activation: {act}")


def get_norm(norm: str, ch: int, num_gn_groups: int = 32):
    """layer_utils.py:32-60."""
    if norm is None or norm == "" or norm.lower() == "none":
        return nn.Identity()
    if norm == "GN":
        return nn.GroupNorm(num_gn_groups, ch)
    if norm == "BN":
        return nn.BatchNorm2d(ch)
    raise ValueError(f"Unknown norm: {norm}")


class ConvModule(nn.Module):
    """conv (bias only without norm) -> norm (registered as ``gn``/``bn``) -> act (``activate``)."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, padding=0, norm=None, num_gn_groups=32, act=None):
        super().__init__()
        with_norm = norm is not None and norm != "" and norm.lower() != "none"
        self.conv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, padding, bias=not with_norm)
        self.norm_name = None
        if with_norm:
            self.norm_name = {"GN": "gn", "BN": "bn"}[norm]
            self.add_module(self.norm_name, get_norm(norm, out_ch, num_gn_groups))
        self.activate = get_nn_act_func(act) if act else None
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        if self.norm_name == "gn":   # conv3x3 -> GN (-> GELU): GroupNorm statistics from the convolution's epilogue
            y = hip_layers.conv3x3_groupnorm_act(self.conv, getattr(self, "gn"), self.activate, x)
            if y is not None:
                return y
        x = hip_layers.conv2d(self.conv, x)
        if self.norm_name == "gn":
            return hip_layers.groupnorm_act(getattr(self, "gn"), self.activate, x)  # fused GN(+GELU) on the GPU
        if self.norm_name is not None or self.activate is not None:
            hip_layers.foreign("ConvModule: norm / activation as PyTorch operators", x)
        if self.norm_name is not None:
            x = getattr(self, self.norm_name)(x)
        if self.activate is not None:
            x = self.activate(x)
        return x

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # the reference registers the norm twice (`self.norm = …` and add_module("gn", …),
        # conv_module.py:175-182), so its checkpoints carry duplicate `norm.*` keys: drop them.
        for k in [k for k in state_dict if k.startswith(prefix + "norm.")]:
            state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)



def run_features(features, x):
    """Run a ``features`` ModuleList, fusing [GroupNorm, act] pairs and routing bilinear x2 upsampling to the
    NHWC HIP kernels when on the GPU (module structure and parameter names are untouched)."""
    i, n = 0, len(features)
    while i < n:
        layer = features[i]
        if isinstance(layer, nn.GroupNorm):
            nxt = features[i + 1] if i + 1 < n else None
            if isinstance(nxt, (nn.GELU, nn.ReLU, nn.LeakyReLU, nn.SiLU, nn.Mish)):
                if isinstance(nxt, nn.GELU):
                    x = hip_layers.groupnorm_act(layer, nxt, x)
                else:
                    hip_layers.foreign("run_features: " + type(nxt).__name__ + " behind GroupNorm as a PyTorch operator", x)
                    x = nxt(hip_layers.groupnorm_act(layer, None, x))
                i += 2
                continue
            x = hip_layers.groupnorm_act(layer, None, x)
        elif isinstance(layer, nn.UpsamplingBilinear2d) and layer.scale_factor in (2, 2.0):
            x = hip_layers.upsample2x(layer, x)
        elif isinstance(layer, nn.Conv2d):
            gn = features[i + 1] if i + 1 < n else None
            if isinstance(gn, nn.GroupNorm):   # conv -> GN (-> exact GELU): statistics in the convolution's epilogue
                nxt = features[i + 2] if i + 2 < n else None
                gelu = nxt if isinstance(nxt, nn.GELU) and getattr(nxt, "approximate", "none") == "none" else None
                y = hip_layers.conv3x3_groupnorm_act(layer, gn, gelu, x)
                if y is not None:
                    x = y
                    i += 3 if gelu is not None else 2
                    continue
            x = hip_layers.conv2d(layer, x)
        elif isinstance(layer, nn.ConvTranspose2d):
            gn = features[i + 1] if i + 1 < n else None
            if isinstance(gn, nn.GroupNorm):   # deconv -> GN (-> exact GELU): statistics from the col2im gather
                nxt = features[i + 2] if i + 2 < n else None
                gelu = nxt if isinstance(nxt, nn.GELU) and getattr(nxt, "approximate", "none") == "none" else None
                y = hip_layers.conv_transpose2d_groupnorm_act(layer, gn, gelu, x)
                if y is not None:
                    x = y
                    i += 3 if gelu is not None else 2
                    continue
            x = hip_layers.conv_transpose2d(layer, x)
        else:
            if type(layer).__module__.startswith("torch.nn") and not isinstance(layer, nn.Identity):   # (ConvModule dispatches by itself)
                hip_layers.foreign("run_features: " + type(layer).__name__ + " as a PyTorch operator", x)
            x = layer(x)
        i += 1
    return x


def _normal_init(m, std):
    nn.init.normal_(m.weight, 0.0, std)
    if getattr(m, "bias", None) is not None:
        nn.init.constant_(m.bias, 0.0)


class TopDownMaskXyzRegionHead(nn.Module):
    """Shared trunk 8x8 -> 64x64 and a shared 1x1 output layer.  ``double_mask`` selects the
    [vis C | full C | xyz 3C | region 65C] split of the DoubleMask variant
    (top_down_doublemask_xyz_region_head.py:184-197), else [mask | xyz | region]."""

    double_mask = False

    def __init__(self, in_dim, up_types=("deconv", "bilinear", "bilinear"), deconv_kernel_size=3,
                 num_conv_per_block=2, feat_dim=256, feat_kernel_size=3, norm="GN", num_gn_groups=32, act="GELU",
                 out_kernel_size=1, out_layer_shared=True, mask_num_classes=1, xyz_num_classes=1,
                 region_num_classes=1, mask_out_dim=1, xyz_out_dim=3, region_out_dim=65, **_unused):
        super().__init__()
        assert out_layer_shared and out_kernel_size == 1, "GDRNPP configs use the shared 1x1 output layer"
        self.features = nn.ModuleList()
        for i, up_type in enumerate(up_types):
            _in = in_dim if i == 0 else feat_dim
            if up_type == "deconv":
                pad, out_pad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[deconv_kernel_size]
                self.features.append(nn.ConvTranspose2d(_in, feat_dim, deconv_kernel_size, stride=2, padding=pad,
                                                        output_padding=out_pad, bias=False))
                self.features.append(get_norm(norm, feat_dim, num_gn_groups))
                self.features.append(get_nn_act_func(act))
            elif up_type == "bilinear":
                self.features.append(nn.UpsamplingBilinear2d(scale_factor=2))
            elif up_type == "nearest":
                self.features.append(nn.UpsamplingNearest2d(scale_factor=2))
            else:
                raise ValueError(f"Unknown up_type: {up_type}")
            for i_conv in range(num_conv_per_block):
                cin = in_dim if (i == 0 and i_conv == 0 and up_type in ("bilinear", "nearest")) else feat_dim
                self.features.append(ConvModule(cin, feat_dim, feat_kernel_size, padding=(feat_kernel_size - 1) // 2,
                                                norm=norm, num_gn_groups=num_gn_groups, act=act))
        self.mask_num_classes, self.xyz_num_classes = mask_num_classes, xyz_num_classes
        self.region_num_classes = region_num_classes
        self.mask_out_dim, self.xyz_out_dim, self.region_out_dim = mask_out_dim, xyz_out_dim, region_out_dim
        out_dim = mask_out_dim * mask_num_classes + xyz_out_dim * xyz_num_classes + region_out_dim * region_num_classes
        self.out_layer = nn.Conv2d(feat_dim, out_dim, kernel_size=1, bias=True)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                _normal_init(m, 0.001)
            elif isinstance(m, nn.GroupNorm):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)
        _normal_init(self.out_layer, 0.01)

    def trunk(self, x):
        if isinstance(x, (tuple, list)) and len(x) == 1:
            x = x[0]
        return run_features(self.features, x)

    def split(self, out):
        """Channel split of the full output (reference forward tail)."""
        mask_dim = self.mask_out_dim * self.mask_num_classes
        xyz_dim = self.xyz_out_dim * self.xyz_num_classes
        xyz = out[:, mask_dim:mask_dim + xyz_dim]
        region = out[:, mask_dim + xyz_dim:]
        bs, c, h, w = xyz.shape
        xyz = xyz.view(bs, 3, xyz_dim // 3, h, w)
        coor = (xyz[:, 0], xyz[:, 1], xyz[:, 2])
        if self.double_mask:
            return (out[:, :mask_dim // 2], out[:, mask_dim // 2:mask_dim]) + coor + (region,)
        return (out[:, :mask_dim],) + coor + (region,)

    def forward(self, x):
        return self.split(self.out_layer(self.trunk(x)))

    # ---- class-sliced output layer ------------------------------------
    def class_channel_index(self, num_classes: int) -> torch.Tensor:
        """Rows of ``out_layer.weight`` that the class-aware gather of GDRN_double_mask.py:107-126 keeps for class c, in the
        order [vis(m), full(m), x, y, z, region(65)] -> i64[C, 2m + 3 + 65] (m = channels per mask: 1, or 2 for the CE
        flavour; single-mask heads have no ``full`` block).  The reference views each block as (bs, C, k, h, w), i.e. class
        c owns the k consecutive channels c*k .. c*k + k - 1 of its block; xyz is viewed (bs, 3, C, h, w) first."""
        C = num_classes
        assert self.xyz_out_dim == 3 and self.mask_num_classes == self.xyz_num_classes == self.region_num_classes == C
        n_masks = 2 if self.double_mask else 1
        m = self.mask_out_dim // n_masks
        mask_dim = self.mask_out_dim * C
        rows = []
        for c in range(C):
            r = [blk * m * C + c * m + j for blk in range(n_masks) for j in range(m)]     # vis (, full)
            r += [mask_dim + k * C + c for k in range(3)]                                   # x, y, z
            r += [mask_dim + 3 * C + c * self.region_out_dim + j for j in range(self.region_out_dim)]
            rows.append(r)
        return torch.tensor(rows, dtype=torch.long)


class TopDownDoubleMaskXyzRegionHead(TopDownMaskXyzRegionHead):
    double_mask = True

    def __init__(self, in_dim, mask_out_dim=2, **kw):
        super().__init__(in_dim, mask_out_dim=mask_out_dim, **kw)


class ConvPnPNet(nn.Module):
    """Patch-PnP: 3 x [conv3x3 s2 -> GN -> act] -> flatten -> fc1 -> fc2 -> (fc_r, fc_t)."""

    def __init__(self, nIn, num_regions=8, mask_attention_type="none", featdim=128, rot_dim=6, num_stride2_layers=3,
                 num_extra_layers=0, norm="GN", num_gn_groups=32, act="relu", drop_prob=0.0, flat_op="flatten",
                 final_spatial_size=(8, 8), denormalize_by_extent=True, **_unused):
        super().__init__()
        assert flat_op == "flatten" and drop_prob == 0.0
        self.mask_attention_type = mask_attention_type
        self.denormalize_by_extent = denormalize_by_extent
        # legacy quirk kept: cfg act "relu" means ReLU in the convs but LeakyReLU(0.1) in the fcs (conv_pnp_net.py:43-48)
        self.act = get_nn_act_func("lrelu") if act == "relu" else get_nn_act_func(act)
        self.features = nn.ModuleList()
        for i in range(num_stride2_layers):
            self.features.append(nn.Conv2d(nIn if i == 0 else featdim, featdim, 3, stride=2, padding=1, bias=False))
            self.features.append(get_norm(norm, featdim, num_gn_groups))
            self.features.append(get_nn_act_func(act))
        for _ in range(num_extra_layers):
            self.features.append(nn.Conv2d(featdim, featdim, 3, stride=1, padding=1, bias=False))
            self.features.append(get_norm(norm, featdim, num_gn_groups))
            self.features.append(get_nn_act_func(act))
        fh, fw = final_spatial_size
        self.fc1 = nn.Linear(featdim * fh * fw, 1024)
        self.fc2 = nn.Linear(1024, 256)
        self.fc_r = nn.Linear(256, rot_dim)
        self.fc_t = nn.Linear(256, 3)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                _normal_init(m, 0.001)
            elif isinstance(m, nn.GroupNorm):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)
        _normal_init(self.fc_r, 0.01)
        _normal_init(self.fc_t, 0.01)

    def forward(self, coor_feat, region=None, extents=None, mask_attention=None, pose=None):
        bs, in_c, fh, fw = coor_feat.shape
        hip_layers.foreign("ConvPnPNet.forward: input de-normalisation / concatenation as PyTorch operators (module path)", coor_feat)
        if in_c in (3, 5) and self.denormalize_by_extent and extents is not None:
            coor_feat[:, :3] = (coor_feat[:, :3] - 0.5) * extents.view(bs, 3, 1, 1)  # in place, like :130-131
        x = torch.cat([coor_feat, region], dim=1) if region is not None else coor_feat
        if self.mask_attention_type == "mul":
            x = x * mask_attention
        elif self.mask_attention_type == "concat":
            x = torch.cat([x, mask_attention], dim=1)
        x = run_features(self.features, x)
        return self._fc_tail(x, pose)

    def _fc_tail(self, x, pose=None):
        x = x.flatten(2).flatten(1)  # NCHW order, like the reference (weights of fc1 depend on it)
        if isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none":
            x = hip_layers.linear(self.fc2, hip_layers.linear(self.fc1, x, gelu=True), gelu=True)   # GELU in the GEMM epilogues
        else:
            hip_layers.foreign("ConvPnPNet: " + type(self.act).__name__ + " behind fc1 / fc2 as a PyTorch operator", x)
            x = self.act(hip_layers.linear(self.fc1, x))
            x = self.act(hip_layers.linear(self.fc2, x))
        return hip_layers.pnp_fc_heads(self.fc_r, self.fc_t, x, pose)

    # ---- NHWC entry used by the fused head tail (hip_lib.head_tail_nhwc) ------------------------------------------------
    def accepts_prepared_input(self) -> bool:
        """The prepared [xyz * extent | coord2d | region softmax | zero pad] NHWC input replaces ``forward``'s own de-normalisation
        and concatenation: only for the plain configuration (69 input channels, no mask attention)."""
        c0 = self.features[0] if len(self.features) else None
        return (isinstance(c0, nn.Conv2d) and c0.in_channels == 69 and c0.out_channels % 128 == 0 and c0.kernel_size == (3, 3)
                and c0.stride == (2, 2) and c0.padding == (1, 1) and c0.bias is None and self.mask_attention_type == "none"
                and self.denormalize_by_extent)

    def forward_prepared(self, x96_cl, pose=None):
        """``x96_cl``: [B, 96, H, W] channels_last, channels 69..95 zero.  First convolution with its weight zero-padded to 96
        input channels on the implicit-GEMM split kernel, the rest as ``forward``."""
        c0 = self.features[0]
        cache = c0.__dict__.setdefault("_gdrnpp_cache", {})
        def w96(w):
            out = torch.zeros((w.shape[0], 96, 3, 3), dtype=w.dtype, device=w.device)
            out[:, :69] = w
            return out

        # always the six-product kernel: the input (metres, [0, 1) coordinates, softmax weights) is not a normalised tensor and sits
        # below the range of the three-product form at most pixels (its range word would say so on the first step)
        w_pk = hip_layers._packed_weight(cache, "w96_pk", c0.weight, lambda w: hip_lib.pack_conv_weight_bf16x3(w96(w)))
        x = hip_lib.conv2d_f32_split(x96_cl, w_pk, None, 3, 3, 2, 1)
        x = run_features(self.features[1:], x)
        return self._fc_tail(x, pose)


HEADS = {
    "TopDownMaskXyzRegionHead": TopDownMaskXyzRegionHead,
    "TopDownDoubleMaskXyzRegionHead": TopDownDoubleMaskXyzRegionHead,
    "ConvPnPNet": ConvPnPNet,
}
