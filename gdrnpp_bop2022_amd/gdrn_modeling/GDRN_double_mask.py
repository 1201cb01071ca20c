"""``GDRN_Net`` for inference: the forward/API surface of the reference kept verbatim
(core/gdrn_modeling/models/GDRN_double_mask.py:35-214,539-615 and models/GDRN.py:66-205),
the schedule re-designed for MI355X:

* backbone / geometry head / Patch-PnP run as PyTorch-ROCm modules (fp32, channels-last);
* CLASS-SLICED OUTPUT LAYER: the reference computes all ``(2+3+65)*C`` output maps
  (1470 channels = 24 MB/ROI for YCB-V) and then keeps the 70 of the ROI's class
  (GDRN_double_mask.py:107-126).  Here the 70 rows of ``out_layer.weight`` belonging to each ROI's
  class are gathered and applied with one batched GEMM ``[B,70,256] x [B,256,4096]`` — same
  arithmetic per kept channel, 21x less MFMA work and none of the 24 MB/ROI of HBM traffic
  (SURVEY.md §7 item 9.ii).  ``exact_reference_order=True`` runs the reference's own graph instead;
* rot6d -> R, centroid/z -> t and allocentric -> egocentric run in one HIP kernel on the device
  (``gdrnpp_pose_from_pred_centroid_z``) instead of the reference's D2H sync + per-ROI NumPy loop
  (pose_from_pred_centroid_z.py:118-154), so ``forward`` never synchronises with the host.
  Consequence: ``out["rot"]`` is a DEVICE tensor (the reference returns a CPU tensor).
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_lib
from . import hip_layers
from .backbones import create_backbone
from .heads import HEADS


def get_xyz_mask_region_out_dim(cfg, double_mask=False):
    """models/model_utils.py:40-65 (and the doublemask variant): (xyz, mask, region) output dims."""
    net_cfg = cfg.MODEL.POSE_NET
    loss_cfg = net_cfg.LOSS_CFG
    if loss_cfg.XYZ_LOSS_TYPE in ("MSE", "L1", "L2", "SmoothL1"):
        xyz_out_dim = 3
    elif loss_cfg.XYZ_LOSS_TYPE in ("CE_coor", "CE"):
        xyz_out_dim = 3 * (net_cfg.GEO_HEAD.XYZ_BIN + 1)
    else:
        raise NotImplementedError(loss_cfg.XYZ_LOSS_TYPE)
    if loss_cfg.MASK_LOSS_TYPE in ("L1", "BCE", "RW_BCE", "dice"):
        mask_out_dim = 2 if double_mask else 1
    elif loss_cfg.MASK_LOSS_TYPE == "CE":
        mask_out_dim = 4 if double_mask else 2
    else:
        raise NotImplementedError(loss_cfg.MASK_LOSS_TYPE)
    region_out_dim = net_cfg.GEO_HEAD.NUM_REGIONS + 1
    assert region_out_dim > 2
    return xyz_out_dim, mask_out_dim, region_out_dim


def get_mask_prob(pred_mask, mask_loss_type):
    """models/model_utils.py:362-380."""
    bs = pred_mask.shape[0]
    if mask_loss_type == "L1":
        mx = pred_mask.view(bs, -1).max(-1)[0].view(bs, 1, 1, 1)
        mn = pred_mask.view(bs, -1).min(-1)[0].view(bs, 1, 1, 1)
        return (pred_mask - mn) / (mx - mn)
    if mask_loss_type in ("BCE", "RW_BCE", "dice"):
        return torch.sigmoid(pred_mask)
    raise NotImplementedError(mask_loss_type)


class GDRN_DoubleMask(nn.Module):
    def __init__(self, cfg, backbone, geo_head_net, neck=None, pnp_net=None):
        super().__init__()
        assert cfg.MODEL.POSE_NET.NAME in ("GDRN_double_mask", "GDRN"), cfg.MODEL.POSE_NET.NAME
        self.backbone = backbone
        self.neck = neck
        self.geo_head_net = geo_head_net
        self.pnp_net = pnp_net
        self.cfg = cfg
        self.double_mask = cfg.MODEL.POSE_NET.NAME == "GDRN_double_mask"
        self.xyz_out_dim, self.mask_out_dim, self.region_out_dim = get_xyz_mask_region_out_dim(cfg, self.double_mask)
        g = cfg.MODEL.POSE_NET.GEO_HEAD
        self.class_aware = bool(g.XYZ_CLASS_AWARE and g.MASK_CLASS_AWARE and g.REGION_CLASS_AWARE)
        # Output-layer slices: one per class for the class-aware heads (the 7 BOP convnext_a6 configs); ONE slice — the whole
        # layer — for the class-agnostic heads (the reference's 162 single-object configs, configs/gdrn/*SO/*, and the base
        # config): the same grouped GEMM / head-tail kernels serve both, with every ROI selecting slice 0.
        agnostic = not (g.XYZ_CLASS_AWARE or g.MASK_CLASS_AWARE or g.REGION_CLASS_AWARE)
        self.slice_classes = cfg.MODEL.POSE_NET.NUM_CLASSES if self.class_aware else (1 if agnostic else None)
        self.exact_reference_order = False
        if self.slice_classes is not None and self.xyz_out_dim == 3:
            self.register_buffer("_cls_rows", geo_head_net.class_channel_index(self.slice_classes), persistent=False)
        self._sliced_w = None  # cache of (weight[C,70,256], bias[C,70]) for eval
        self._sliced_pk = {}    # cache of the packed, 128-row padded slices for the grouped split GEMM (hip_layers.cached)
        self.fused_head_tail = True   # all-NHWC head tail on the HIP path (False: baddbmm + torch ops, for A/B)

    def load_state_dict(self, *args, **kwargs):
        res = super().load_state_dict(*args, **kwargs)
        hip_layers.reset_x3_calibration()     # new weights, new activation scales: the layers look at their inputs again
        return res

    # ------------------------------------------------------------------------------------------
    def _sliced_out_layer(self, feat, roi_classes):
        """Per-ROI 70-channel output layer = the class-aware gather folded into the weights."""
        ol = self.geo_head_net.out_layer
        tag = hip_layers.weight_tag(ol.weight, ol.bias)
        if self._sliced_w is None or self.training or self._sliced_w[0] != tag:
            w = ol.weight.view(ol.out_channels, -1)[self._cls_rows]  # [C,70,256]
            b = ol.bias[self._cls_rows]                               # [C,70]
            self._sliced_w = (tag, w.contiguous(), b.contiguous())
        _, w, b = self._sliced_w
        bs, ch, h, wd = feat.shape
        x = feat.reshape(bs, ch, h * wd)  # NCHW-logical [B,256,4096]; one copy if feat is channels-last
        out = torch.baddbmm(b[roi_classes].unsqueeze(-1), w[roi_classes], x)  # [B,70,4096]
        out = out.view(bs, -1, h, wd)
        k = self.mask_out_dim                      # channels of all masks of one class
        m = k // 2 if self.double_mask else k      # channels per mask (2 for the CE flavour)
        vis = out[:, 0:m]
        full = out[:, m:2 * m] if self.double_mask else None
        return vis, full, out[:, k:k + 1], out[:, k + 1:k + 2], out[:, k + 2:k + 3], out[:, k + 3:]

    def _fused_tail_ok(self, x, feat, roi_classes, coord2d, roi_extents) -> bool:
        """The all-NHWC head tail (grouped class-sliced GEMM -> head_tail kernel -> Patch-PnP on the split convolution) covers
        the GDRNPP BOP configuration: regression xyz, 64 regions, coord2d + region attention into Patch-PnP, no mask attention."""
        net_cfg = self.cfg.MODEL.POSE_NET
        pn = net_cfg.PNP_NET
        return (hip_layers.enabled_for(x) and hip_layers.mlp_gemm() == "split" and self.fused_head_tail
                and self.slice_classes is not None and self.xyz_out_dim == 3 and not self.exact_reference_order and not self.training
                and self.region_out_dim == 65 and self.mask_out_dim == (2 if self.double_mask else 1)
                and pn.WITH_2D_COORD and pn.REGION_ATTENTION and pn.MASK_ATTENTION == "none"
                and roi_classes is not None and coord2d is not None and roi_extents is not None
                and self.pnp_net is not None and hasattr(self.pnp_net, "accepts_prepared_input") and self.pnp_net.accepts_prepared_input()
                and feat.shape[1] % 32 == 0 and (feat.shape[2] * feat.shape[3]) % 256 == 0
                and feat.shape[0] * feat.shape[2] * feat.shape[3] * feat.shape[1] * 4 < (1 << 32))

    def _fused_tail(self, feat, roi_classes, coord2d, roi_extents, pose=None):
        """feat [B,256,64,64] (channels_last) -> Patch-PnP outputs and the maps of out_dict, without leaving NHWC:
        one grouped GEMM launch (each ROI's 4096 rows against the 70-channel weight slice of its class, padded to one 128-wide
        tile), one kernel for [xyz * extent | coord2d | region softmax] + the map planes, Patch-PnP's first convolution with
        Cin padded 69 -> 96 on the implicit-GEMM kernel.  Same arithmetic as the module path (fp32-accurate GEMMs)."""
        ol = self.geo_head_net.out_layer

        def build():
            w = ol.weight.detach().view(ol.out_channels, -1)[self._cls_rows]    # [C,70,256]
            b = ol.bias.detach()[self._cls_rows]                                  # [C,70]
            C, n70, k = w.shape
            w128 = torch.zeros((C, 128, k), dtype=w.dtype, device=w.device)
            w128[:, :n70] = w
            b128 = torch.zeros((C, 128), dtype=b.dtype, device=b.device)
            b128[:, :n70] = b
            return hip_lib.pack_weight_bf16x3(w128.view(C * 128, k).contiguous()), b128.contiguous(), n70

        # (built with torch operators on the filling stream: hip_layers.cached drains it before another stream may hit the entry)
        _, w_pk, b128, n70 = hip_layers.cached(self._sliced_pk, "pk", hip_layers.weight_tag(ol.weight, ol.bias), build, ol.weight)
        bs, ch, h, wd = feat.shape
        feat = feat.contiguous(memory_format=torch.channels_last)
        x2d = feat.permute(0, 2, 3, 1).reshape(bs * h * wd, ch)     # a view of the NHWC memory
        out = hip_lib.linear_f32_split_grouped(x2d, w_pk, b128, roi_classes.to(torch.int32), h * wd, n_store=(n70 + 3) // 4 * 4)
        pnp_in, planes = hip_lib.head_tail_nhwc(out, coord2d.contiguous(), roi_extents.contiguous().float(), self.double_mask)
        x96 = pnp_in.view(bs, h, wd, 96).permute(0, 3, 1, 2)        # [B,96,H,W] channels_last view
        pred_rot_, pred_t_ = self.pnp_net.forward_prepared(x96, pose)
        planes = planes.view(planes.shape[0], bs, 1, h, wd)
        k = 2 if self.double_mask else 1
        maps = {"mask": planes[0], "coor_x": planes[k], "coor_y": planes[k + 1], "coor_z": planes[k + 2],
                "region": out.view(bs, h, wd, out.shape[1])[..., k + 3:n70].permute(0, 3, 1, 2)}
        if self.double_mask:
            maps["full_mask"] = planes[1]
        return pred_rot_, pred_t_, maps

    def forward_maps(self, x, roi_classes=None, roi_coord_2d=None, roi_coord_2d_rel=None, roi_extents=None, pose=None):
        """Everything of ``forward`` up to the Patch-PnP outputs: pure PyTorch (also runs on CPU).  ``pose``: the pose request of
        ``forward`` (hip_layers.pnp_fc_heads): on the HIP path Patch-PnP's last launch also produces the pose."""
        cfg = self.cfg
        net_cfg = cfg.MODEL.POSE_NET
        g_head_cfg = net_cfg.GEO_HEAD
        pnp_net_cfg = net_cfg.PNP_NET
        bs = x.shape[0]
        num_classes = net_cfg.NUM_CLASSES
        out_res = net_cfg.OUTPUT_RES

        conv_feat = self.backbone(x)  # [bs, c, 8, 8]
        if self.neck is not None:
            conv_feat = self.neck(conv_feat)

        full_mask = None
        sliced = False
        if self.slice_classes is not None and self.xyz_out_dim == 3 and not self.exact_reference_order:
            if self.class_aware:
                assert roi_classes is not None
                sel = roi_classes
            else:       # class-agnostic head: every ROI uses the one slice (= the whole output layer)
                sel = torch.zeros((bs,), dtype=torch.long, device=x.device)
            feat = self.geo_head_net.trunk(conv_feat)
            coord2d = roi_coord_2d_rel if pnp_net_cfg.WITH_2D_COORD and pnp_net_cfg.COORD_2D_TYPE == "rel" else roi_coord_2d
            if self._fused_tail_ok(x, feat, sel, coord2d, roi_extents):
                return self._fused_tail(feat, sel, coord2d, roi_extents, pose)
            if hip_layers.enabled_for(x):      # the module-path tail below is PyTorch operators (baddbmm, softmax, cat, gathers)
                hip_layers.note_foreign_launch("GDRN_DoubleMask.forward_maps: head tail on the module path (configuration outside the fused NHWC tail)")
            if self.class_aware:
                vis_mask, full_mask, coor_x, coor_y, coor_z, region = self._sliced_out_layer(feat, sel)
                sliced = True
            else:
                outs = self.geo_head_net.split(self.geo_head_net.out_layer(feat))
        else:
            if hip_layers.enabled_for(x):
                hip_layers.note_foreign_launch("GDRN_DoubleMask.forward_maps: full-width output layer + class gather on the module path")
            outs = self.geo_head_net(conv_feat)
        if not sliced:
            if self.double_mask:
                vis_mask, full_mask, coor_x, coor_y, coor_z, region = outs
            else:
                vis_mask, coor_x, coor_y, coor_z, region = outs
            ar = torch.arange(bs, device=x.device)
            if g_head_cfg.XYZ_CLASS_AWARE:
                assert roi_classes is not None
                coor_x = coor_x.reshape(bs, num_classes, self.xyz_out_dim // 3, out_res, out_res)[ar, roi_classes]
                coor_y = coor_y.reshape(bs, num_classes, self.xyz_out_dim // 3, out_res, out_res)[ar, roi_classes]
                coor_z = coor_z.reshape(bs, num_classes, self.xyz_out_dim // 3, out_res, out_res)[ar, roi_classes]
            if g_head_cfg.MASK_CLASS_AWARE:
                md = self.mask_out_dim // 2 if self.double_mask else self.mask_out_dim
                vis_mask = vis_mask.reshape(bs, num_classes, md, out_res, out_res)[ar, roi_classes]
                if full_mask is not None:
                    full_mask = full_mask.reshape(bs, num_classes, md, out_res, out_res)[ar, roi_classes]
            if g_head_cfg.REGION_CLASS_AWARE:
                region = region.reshape(bs, num_classes, self.region_out_dim, out_res, out_res)[ar, roi_classes]

        # ---- Patch-PnP inputs (GDRN_double_mask.py:128-160) -----------------------------------------
        if coor_x.shape[1] > 1:
            coor_feat = torch.cat([F.softmax(coor_x[:, :-1], dim=1), F.softmax(coor_y[:, :-1], dim=1),
                                   F.softmax(coor_z[:, :-1], dim=1)], dim=1)
        else:
            coor_feat = torch.cat([coor_x, coor_y, coor_z], dim=1)
        if pnp_net_cfg.WITH_2D_COORD:
            if pnp_net_cfg.COORD_2D_TYPE == "rel":
                coor_feat = torch.cat([coor_feat, roi_coord_2d_rel], dim=1)
            else:
                coor_feat = torch.cat([coor_feat, roi_coord_2d], dim=1)
        region_softmax = F.softmax(region[:, 1:], dim=1)  # channel 0 is bg
        mask_atten = None
        if pnp_net_cfg.MASK_ATTENTION != "none":
            mask_atten = get_mask_prob(vis_mask, net_cfg.LOSS_CFG.MASK_LOSS_TYPE)
        region_atten = region_softmax if pnp_net_cfg.REGION_ATTENTION else None
        pred_rot_, pred_t_ = self.pnp_net(coor_feat, region=region_atten, extents=roi_extents,
                                          mask_attention=mask_atten, **({"pose": pose} if pose is not None else {}))

        maps = {"mask": vis_mask, "coor_x": coor_x, "coor_y": coor_y, "coor_z": coor_z, "region": region}
        if full_mask is not None:
            maps["full_mask"] = full_mask
        return pred_rot_, pred_t_, maps

    def forward(self, x, gt_xyz=None, gt_xyz_bin=None, gt_mask_trunc=None, gt_mask_visib=None, gt_mask_obj=None,
                gt_mask_full=None, gt_region=None, gt_ego_rot=None, gt_points=None, sym_infos=None, gt_trans=None,
                gt_trans_ratio=None, roi_classes=None, roi_coord_2d=None, roi_coord_2d_rel=None, roi_cams=None,
                roi_centers=None, roi_whs=None, roi_extents=None, resize_ratios=None, do_loss=False):
        if do_loss:
            raise NotImplementedError("training losses are out of scope of this build (SURVEY.md §2.1)")
        cfg = self.cfg
        pnp_net_cfg = cfg.MODEL.POSE_NET.PNP_NET
        bs = x.shape[0]
        # ---- rot6d -> R, centroid/z -> t, allo -> ego on the device, no host sync: in Patch-PnP's last launch on the HIP path
        # (gdrnpp_pnp_fc_heads_pose), else one kernel of its own (gdrnpp_pose_from_pred) ------------------------------------------
        rot_type = pnp_net_cfg.ROT_TYPE          # get_rot_mat (model_utils.py:347-359) + the three TRANS_TYPE branches (:162-200)
        if rot_type in ("allo_rot6d", "ego_rot6d"):
            rot_mode = "rot6d"
        elif rot_type in ("allo_quat", "ego_quat"):
            rot_mode = "quat"
        elif rot_type in ("allo_log_quat", "ego_log_quat"):
            rot_mode = "log_quat"
        elif rot_type in ("allo_lie_vec", "ego_lie_vec"):
            rot_mode = "lie_vec"
        else:
            raise ValueError(f"Wrong pred_rot type: {rot_type}")      # model_utils.py:358
        trans_type = pnp_net_cfg.TRANS_TYPE
        if trans_type == "centroid_z":
            if pnp_net_cfg.Z_TYPE not in ("REL", "ABS"):
                raise NotImplementedError(f"Z_TYPE={pnp_net_cfg.Z_TYPE}")
            t_mode = "centroid_z_rel" if pnp_net_cfg.Z_TYPE == "REL" else "centroid_z_abs_z"
        elif trans_type in ("centroid_z_abs", "trans"):
            t_mode = trans_type
        else:
            raise ValueError(f"Unknown trans type: {trans_type}")
        need_roi = trans_type == "centroid_z"
        pose = dict(cams=roi_cams.reshape(bs, 9).contiguous().float(), centers=roi_centers.contiguous().float() if need_roi else None,
                    whs=roi_whs.contiguous().float() if need_roi else None,
                    resize_ratios=resize_ratios.reshape(bs).contiguous().float() if need_roi and t_mode == "centroid_z_rel" else None,
                    rot_mode=rot_mode, t_mode=t_mode, is_allo="allo" in rot_type)
        pred_rot_, pred_t_, maps = self.forward_maps(x, roi_classes, roi_coord_2d, roi_coord_2d_rel, roi_extents,
                                                     pose if x.is_cuda else None)
        if "result" in pose:
            pred_ego_rot, pred_trans = pose.pop("result")
        else:
            pred_ego_rot, pred_trans = hip_lib.pose_from_pred(pred_rot_.float().contiguous(), pred_t_.float().contiguous(), **pose)

        out_dict = {"rot": pred_ego_rot, "trans": pred_trans}
        if cfg.TEST.USE_PNP or cfg.TEST.SAVE_RESULTS_ONLY or cfg.TEST.USE_DEPTH_REFINE:
            out_dict.update(maps)
        return out_dict


def build_model_optimizer(cfg, is_test=True, model_cls=None):
    """GDRN_double_mask.py:539-615 — returns (model, optimizer); the optimizer is None (inference build)."""
    if not is_test:
        raise NotImplementedError("training is out of scope of this build (SURVEY.md §2.1)")
    net_cfg = cfg.MODEL.POSE_NET
    init_backbone_args = copy.deepcopy(dict(net_cfg.BACKBONE.INIT_CFG))
    backbone = create_backbone(**init_backbone_args)

    double = net_cfg.NAME == "GDRN_double_mask"
    g = net_cfg.GEO_HEAD
    head_cfg = copy.deepcopy(dict(g.INIT_CFG))
    head_type = head_cfg.pop("type")
    xyz_dim, mask_dim, region_dim = get_xyz_mask_region_out_dim(cfg, double)
    head_cfg.update(
        xyz_num_classes=net_cfg.NUM_CLASSES if g.XYZ_CLASS_AWARE else 1,
        mask_num_classes=net_cfg.NUM_CLASSES if g.MASK_CLASS_AWARE else 1,
        region_num_classes=net_cfg.NUM_CLASSES if g.REGION_CLASS_AWARE else 1,
        xyz_out_dim=xyz_dim, mask_out_dim=mask_dim, region_out_dim=region_dim)
    geo_head = HEADS[head_type](**head_cfg)

    # models/model_utils.py:198-274 (get_pnp_net)
    p = net_cfg.PNP_NET
    xyz_dim1, _, _ = get_xyz_mask_region_out_dim(cfg, False)
    n_in = xyz_dim1 - 3 if net_cfg.LOSS_CFG.XYZ_LOSS_TYPE in ("CE_coor", "CE") else xyz_dim1
    n_in += 2 if p.WITH_2D_COORD else 0
    n_in += g.NUM_REGIONS if p.REGION_ATTENTION else 0
    n_in += 1 if p.MASK_ATTENTION == "concat" else 0
    # model_utils.py:219-230: quaternion 4, log-quaternion / Lie vector 3, else 6
    rot_dim = {"allo_rot6d": 6, "ego_rot6d": 6, "allo_quat": 4, "ego_quat": 4, "allo_log_quat": 3, "ego_log_quat": 3,
               "allo_lie_vec": 3, "ego_lie_vec": 3}[p.ROT_TYPE]
    pnp_cfg = copy.deepcopy(dict(p.INIT_CFG))
    pnp_type = pnp_cfg.pop("type")
    pnp_cfg.update(nIn=n_in, rot_dim=rot_dim, num_regions=g.NUM_REGIONS, mask_attention_type=p.MASK_ATTENTION)
    pnp_net = HEADS[pnp_type](**pnp_cfg)

    model = (model_cls or GDRN_DoubleMask)(cfg, backbone, neck=None, geo_head_net=geo_head, pnp_net=pnp_net)
    model.eval()
    if cfg.MODEL.DEVICE != "cpu" and torch.cuda.is_available():
        model.to(torch.device(cfg.MODEL.DEVICE))
        model.to(memory_format=torch.channels_last)
    return model, None


_ALLOWED_UNEXPECTED = ("num_batches_tracked",)


def load_checkpoint(model, path, strict=True):
    """core/utils/my_checkpoint.py:28-83: ``{"model": state_dict}``, keys may carry a leading ``_module.`` / ``module.``
    prefix (DDP / lightning wrappers) — stripped as a PREFIX only, repeatedly.  A checkpoint that leaves parameters
    uninitialised or brings keys this model does not have is an error unless ``strict=False`` (then it is reported loudly):
    the timm ConvNeXt key naming of ``backbones.py`` is unverified against a real GDRNPP checkpoint, and a silent
    mismatch would leave the backbone random while poses still come out finite."""
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("model", sd)
    clean = {}
    for k, v in sd.items():
        stripped = True
        while stripped:
            stripped = False
            for pre in ("_module.", "module."):
                if k.startswith(pre):
                    k, stripped = k[len(pre):], True
        clean[k] = v
    res = model.load_state_dict(clean, strict=False)
    unexpected = [k for k in res.unexpected_keys if not k.endswith(_ALLOWED_UNEXPECTED)]
    if res.missing_keys or unexpected:
        msg = (f"checkpoint {path}: {len(res.missing_keys)} parameters not found in the file (first: "
               f"{res.missing_keys[:5]}), {len(unexpected)} keys of the file not used (first: {unexpected[:5]})")
        if strict:
            raise RuntimeError(msg)
        import warnings
        warnings.warn(msg)
    return res
