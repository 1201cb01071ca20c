"""Golden vectors of the NETWORK graph from the reference's own modules (authoring container only: needs /root/reference).

What runs here is the reference's code, imported from its files (tests/golden/_refimport.py explains the stand-ins for the
absent third-party packages):

  core/gdrn_modeling/models/GDRN_double_mask.py   build_model_optimizer, GDRN_DoubleMask.forward (class-aware gather,
                                                  Patch-PnP inputs, rot6d -> R, centroid/z -> t, allo -> ego)
  core/gdrn_modeling/models/model_utils.py        get_geo_head, get_pnp_net, get_rot_mat, get_mask_prob, out dims
  core/gdrn_modeling/models/heads/top_down_doublemask_xyz_region_head.py   TopDownDoubleMaskXyzRegionHead
  core/gdrn_modeling/models/heads/conv_pnp_net.py                          ConvPnPNet
  lib/torch_utils/layers/conv_module.py, layer_utils.py                    ConvModule, get_norm, get_nn_act_func
  core/gdrn_modeling/models/pose_from_pred_centroid_z.py, core/utils/utils.py, core/utils/rot_reps.py
  configs/gdrn/{ycbv,tless}/convnext_a6_..._classAware_*.py + configs/_base_/{gdrn_base,common_base}.py   the config values
  core/gdrn_modeling/models/GDRN.py               build_model_optimizer, GDRN.forward (BASELINE configs[0]: the BASE config
                                                  configs/_base_/gdrn_base.py itself — ResNet-34, single mask,
                                                  class-agnostic TopDownMaskXyzRegionHead, Patch-PnP on xyz only, ego_rot6d)
  core/gdrn_modeling/models/heads/top_down_mask_xyz_region_head.py         TopDownMaskXyzRegionHead

with ONE substitution: ``BACKBONES["timm/convnext_base"]`` / ``BACKBONES["timm/resnet34"]`` (timm.create_model — timm 0.6.7
is not installed) build this repo's re-declared ConvNeXt-B / ResNet-34 in their plain-PyTorch form, so the fixture pins everything downstream of the backbone module
boundary against the reference, and the backbone against PyTorch's operators on the same parameters.

Parameters are not stored (head + Patch-PnP + backbone are ~100 M values): every state_dict entry is
``synthetic.seeded_param(key, shape)`` — a counter hash of the key, reproducible anywhere — and the test rebuilds them.
Recorded per config in net_golden_<dataset>.npz: the reference state_dict's key/shape manifest (geo head + Patch-PnP),
the merged config subtree the path reads (JSON), the inputs that are not analytic, and every output of ``forward``.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import _refimport  # noqa: E402

_refimport.install()

from gdrnpp_bop2022_amd import synthetic as S  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.backbones import create_backbone  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.config import Config  # noqa: E402

CONFIGS = {
    "ycbv": "configs/gdrn/ycbv/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_ycbv.py",
    "tless": "configs/gdrn/tless/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_tless.py",
}
# One of the 162 single-object configs (ycbvSO / lmoPbrSO / tlessPbrSO / ...): the same ConvNeXt-B network with a
# class-AGNOSTIC double-mask head (XYZ/MASK/REGION_CLASS_AWARE = False; NUM_CLASSES is set but unused).
SO_CONFIGS = {
    "ycbvso": "configs/gdrn/ycbvSO/convnext_AugCosyAAEGray_DMask_amodalClipBox_ycbv/002_master_chef_can.py",
}
from tests.netgolden import SEED, net_detections, net_detections_b128, net_image, norm_alias  # noqa: E402


def jsonable(o):
    if isinstance(o, dict):
        return {k: jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, (str, int, float, bool)) or o is None:
        return o
    return repr(o)


ALL_CONFIGS = {
    "ycbv": CONFIGS["ycbv"], "tless": CONFIGS["tless"],
    "lmo": "configs/gdrn/lmo_pbr/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_lmo.py",
    "icbin": "configs/gdrn/icbin_pbr/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_icbin.py",
    "hb": "configs/gdrn/hb_pbr/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_hb.py",
    "itodd": "configs/gdrn/itodd_pbr/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_itodd.py",
    "tudl": "configs/gdrn/tudl/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_tudl.py",
}


def write_config_fixture():
    """The merged MODEL / TEST / INPUT / VAL subtrees of the seven BOP convnext_a6 config files -> cfg_golden.json."""
    out = {}
    for ds, path in ALL_CONFIGS.items():
        raw = _refimport.load_ref_config(path)
        out[ds] = jsonable({k: raw[k] for k in ("MODEL", "TEST", "INPUT", "VAL")})
    with open(os.path.join(HERE, "cfg_golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote cfg_golden.json")


def main(configs=None):
    if configs is None:
        write_config_fixture()
        configs = dict(CONFIGS, **SO_CONFIGS)
    torch.set_num_threads(os.cpu_count())
    torch.set_grad_enabled(False)
    hip_layers.set_enabled(False)
    from core.gdrn_modeling.models import GDRN_double_mask as REFM
    from core.gdrn_modeling.models import net_factory

    def backbone_factory(model_name=None, **kw):
        return create_backbone(type="timm/" + model_name, **kw)

    for bname in ("convnext_base",):
        net_factory.BACKBONES[f"timm/{bname}"] = backbone_factory

    for ds, path in configs.items():
        raw = _refimport.load_ref_config(path)
        cfg = Config(raw)
        cfg.MODEL.DEVICE = "cpu"
        cfg.MODEL.POSE_NET.BACKBONE.INIT_CFG.pretrained = False
        cfg.TEST.USE_DEPTH_REFINE = True      # test_gdrn_depth_refine.sh: forward returns the maps
        cfg.SOLVER.BASE_LR = cfg.SOLVER.OPTIMIZER_CFG["lr"]   # main_gdrn.py:103
        model, opt = REFM.build_model_optimizer(cfg, is_test=True)
        assert opt is None and type(model).__module__ == "core.gdrn_modeling.models.GDRN_double_mask"
        model.eval()
        sd = model.state_dict()
        new = S.seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], SEED, alias=norm_alias)
        model.load_state_dict(new, strict=True)

        C = cfg.MODEL.POSE_NET.NUM_CLASSES
        x, det = net_image(), net_detections(C)
        T = torch.from_numpy
        coord2d = S.coord2d_roi(det["roi_center"], det["scale"])
        grab = {}
        model.pnp_net.register_forward_hook(lambda m, i, o: grab.update(pred_rot_=o[0].clone(), pred_t_=o[1].clone()))
        model.backbone.register_forward_hook(lambda m, i, o: grab.update(conv_feat=o[0].clone()))
        out = model(T(x), roi_classes=T(det["roi_cls"]), roi_cams=T(det["roi_cam"]), roi_whs=T(det["roi_wh"]),
                    roi_centers=T(det["roi_center"]), resize_ratios=T(det["resize_ratio"]), roi_coord_2d=T(coord2d),
                    roi_extents=T(det["roi_extent"]), do_loss=False)
        region = out["region"].numpy()
        rec = dict(
            cfg_json=json.dumps(jsonable({k: raw[k] for k in ("MODEL", "TEST", "INPUT")})),
            head_keys=json.dumps([[k, list(v.shape)] for k, v in sd.items() if not k.startswith("backbone.")]),
            roi_cls=det["roi_cls"], roi_cam=det["roi_cam"], roi_wh=det["roi_wh"], roi_center=det["roi_center"],
            resize_ratio=det["resize_ratio"], scale=det["scale"], roi_extent=det["roi_extent"],
            conv_feat_sub=grab["conv_feat"].numpy()[:, ::8], pred_rot_=grab["pred_rot_"].numpy(), pred_t_=grab["pred_t_"].numpy(),
            rot=out["rot"].numpy(), trans=out["trans"].numpy(), mask=out["mask"].numpy(), full_mask=out["full_mask"].numpy(),
            coor_x=out["coor_x"].numpy(), coor_y=out["coor_y"].numpy(), coor_z=out["coor_z"].numpy(),
            region_sub=np.ascontiguousarray(region[:, :, 1::4, 2::4]), region_argmax=region.argmax(1).astype(np.uint8),
            region_absmax=np.float32(np.abs(region).max()))
        for k in ("mask", "coor_x", "rot", "trans", "pred_rot_", "pred_t_"):
            print(ds, k, rec[k].shape, float(np.abs(rec[k]).mean()), float(np.abs(rec[k]).max()))
        np.savez_compressed(os.path.join(HERE, f"net_golden_{ds}.npz"), **rec)
        print("wrote", f"net_golden_{ds}.npz")


def record_b128(datasets=("ycbv", "tless"), b=128):
    """The reference's forward at the BENCHMARK's batch (BASELINE configs[2]: YCB-V, 128 ROIs; configs[3]: one rank's 128-ROI
    shard of T-LESS): every class of the dataset occurs (roi_cls = i mod C, unsorted).  To stay at ~1 MB per fixture the maps are
    stored sub-sampled (every 4th pixel of every ROI; region logits on a 4 x 4 grid + the full arg-max image); R, t and the
    Patch-PnP outputs are stored for all ROIs.  -> net_golden_<ds>_b128.npz"""
    torch.set_num_threads(os.cpu_count())
    torch.set_grad_enabled(False)
    hip_layers.set_enabled(False)
    from core.gdrn_modeling.models import GDRN_double_mask as REFM
    from core.gdrn_modeling.models import net_factory

    net_factory.BACKBONES["timm/convnext_base"] = lambda model_name=None, **kw: create_backbone(type="timm/" + model_name, **kw)
    for ds in datasets:
        raw = _refimport.load_ref_config(CONFIGS[ds])
        cfg = Config(raw)
        cfg.MODEL.DEVICE = "cpu"
        cfg.MODEL.POSE_NET.BACKBONE.INIT_CFG.pretrained = False
        cfg.TEST.USE_DEPTH_REFINE = True
        cfg.SOLVER.BASE_LR = cfg.SOLVER.OPTIMIZER_CFG["lr"]
        model, opt = REFM.build_model_optimizer(cfg, is_test=True)
        assert opt is None and type(model).__module__ == "core.gdrn_modeling.models.GDRN_double_mask"
        model.eval()
        sd = model.state_dict()
        model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], SEED, alias=norm_alias), strict=True)
        C = cfg.MODEL.POSE_NET.NUM_CLASSES
        x, det = net_image(b), net_detections_b128(C, b)
        T = torch.from_numpy
        grab = {}
        model.pnp_net.register_forward_hook(lambda m, i, o: grab.update(pred_rot_=o[0].clone(), pred_t_=o[1].clone()))
        out = model(T(x), roi_classes=T(det["roi_cls"]), roi_cams=T(det["roi_cam"]), roi_whs=T(det["roi_wh"]),
                    roi_centers=T(det["roi_center"]), resize_ratios=T(det["resize_ratio"]),
                    roi_coord_2d=T(S.coord2d_roi(det["roi_center"], det["scale"])), roi_extents=T(det["roi_extent"]), do_loss=False)
        region = out["region"].numpy()
        rec = dict(
            cfg_json=json.dumps(jsonable({k: raw[k] for k in ("MODEL", "TEST", "INPUT")})),
            head_keys=json.dumps([[k, list(v.shape)] for k, v in sd.items() if not k.startswith("backbone.")]),
            roi_cls=det["roi_cls"], roi_cam=det["roi_cam"], roi_wh=det["roi_wh"], roi_center=det["roi_center"],
            resize_ratio=det["resize_ratio"], scale=det["scale"], roi_extent=det["roi_extent"],
            pred_rot_=grab["pred_rot_"].numpy(), pred_t_=grab["pred_t_"].numpy(), rot=out["rot"].numpy(), trans=out["trans"].numpy(),
            region_sub=np.ascontiguousarray(region[:, :, 5::16, 9::16]), region_argmax=region.argmax(1).astype(np.uint8),
            region_absmax=np.float32(np.abs(region).max()))
        for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z"):
            m = out[k].numpy()
            rec[k + "_sub"] = np.ascontiguousarray(m[:, :, ::4, 1::4])
            rec[k + "_absmax"] = np.float32(np.abs(m).max())
        for k in ("mask_sub", "coor_x_sub", "rot", "trans", "pred_rot_", "pred_t_"):
            print(ds, "b128", k, rec[k].shape, float(np.abs(rec[k]).mean()), float(np.abs(rec[k]).max()))
        np.savez_compressed(os.path.join(HERE, f"net_golden_{ds}_b128.npz"), **rec)
        print("wrote", f"net_golden_{ds}_b128.npz")


def record_b128_f64(datasets=("ycbv", "tless"), b=128, chunk=16):
    """The round-4 verdict's question "who is closer to the true value": the SAME reference module, parameters and 128-ROI batch
    as record_b128, evaluated in fp64 (``model.double()``; the fp32 parameters and inputs convert exactly), in chunks of 16 ROIs
    (eval mode: a ROI's outputs do not depend on the batch).  Stored: R, t, the Patch-PnP outputs in fp64, and — from an fp32 run
    of the same chunks, which must reproduce record_b128's fixture — the distance of the reference's own fp32 chain from its fp64
    value per ROI.  -> net_golden_<ds>_b128_f64.npz (1.3 MB: the sub-sampled maps in fp64)"""
    torch.set_num_threads(os.cpu_count())
    torch.set_grad_enabled(False)
    hip_layers.set_enabled(False)
    from core.gdrn_modeling.models import GDRN_double_mask as REFM
    from core.gdrn_modeling.models import net_factory

    net_factory.BACKBONES["timm/convnext_base"] = lambda model_name=None, **kw: create_backbone(type="timm/" + model_name, **kw)
    for ds in datasets:
        raw = _refimport.load_ref_config(CONFIGS[ds])
        cfg = Config(raw)
        cfg.MODEL.DEVICE = "cpu"
        cfg.MODEL.POSE_NET.BACKBONE.INIT_CFG.pretrained = False
        cfg.TEST.USE_DEPTH_REFINE = True
        cfg.SOLVER.BASE_LR = cfg.SOLVER.OPTIMIZER_CFG["lr"]
        model, opt = REFM.build_model_optimizer(cfg, is_test=True)
        assert type(model).__module__ == "core.gdrn_modeling.models.GDRN_double_mask"
        model.eval()
        sd = model.state_dict()
        model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], SEED, alias=norm_alias), strict=True)
        model.double()
        C = cfg.MODEL.POSE_NET.NUM_CLASSES
        x, det = net_image(b), net_detections_b128(C, b)
        coord2d = S.coord2d_roi(det["roi_center"], det["scale"])
        grab = {}
        model.pnp_net.register_forward_hook(lambda m, i, o: grab.update(pred_rot_=o[0].clone(), pred_t_=o[1].clone()))
        acc = {k: [] for k in ("rot", "trans", "pred_rot_", "pred_t_")}
        maps = {k: [] for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z")}
        for s in range(0, b, chunk):
            sl = slice(s, s + chunk)
            D = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).double()   # noqa: E731
            out = model(D(x), roi_classes=torch.from_numpy(det["roi_cls"][sl]), roi_cams=D(det["roi_cam"]), roi_whs=D(det["roi_wh"]),
                        roi_centers=D(det["roi_center"]), resize_ratios=D(det["resize_ratio"]), roi_coord_2d=D(coord2d),
                        roi_extents=D(det["roi_extent"]), do_loss=False)
            assert out["rot"].dtype == torch.float64 and grab["pred_rot_"].dtype == torch.float64
            acc["rot"].append(out["rot"].numpy()); acc["trans"].append(out["trans"].numpy())
            acc["pred_rot_"].append(grab["pred_rot_"].numpy()); acc["pred_t_"].append(grab["pred_t_"].numpy())
            for k in maps:
                maps[k].append(out[k].numpy()[:, :, ::4, 1::4])
            print(ds, "f64 chunk", s, flush=True)
        rec = {k + "_f64": np.concatenate(v) for k, v in acc.items()}
        for k, v in maps.items():
            rec[k + "_sub_f64"] = np.concatenate(v)
        f32 = np.load(os.path.join(HERE, f"net_golden_{ds}_b128.npz"))
        for k in ("rot", "trans", "pred_rot_", "pred_t_"):
            d = np.abs(f32[k].astype(np.float64) - rec[k + "_f64"]).reshape(b, -1).max(1)
            rec["ref_f32_err_" + k] = d
            print(ds, k, "reference fp32 vs its fp64: max", d.max(), "argmax ROI", int(d.argmax()), "ROIs > 5e-5:", np.nonzero(d > 5e-5)[0].tolist())
        np.savez_compressed(os.path.join(HERE, f"net_golden_{ds}_b128_f64.npz"), **rec)
        print("wrote", f"net_golden_{ds}_b128_f64.npz")


def record_large(ds, b, chunk32=32, chunk64=16, seed_shift=1):
    """Parity at the ITERATION sizes BASELINE.json names beyond one 128-ROI step: configs[3]'s 1 024 T-LESS ROIs per iteration and a
    512-ROI YCB-V step (round-5 verdict item 1: the distance between the three- and six-product forms grows with the number of ROIs
    one looks at, so the 128-ROI fixtures do not speak for these sizes).  The same reference module and seeded parameters as
    record_b128 / record_b128_f64, on a NEW batch (images seeded with SEED + 1, detections from their own generator stream, every
    class present), evaluated twice in chunks (eval mode: a ROI's outputs do not depend on the batch): the reference's fp32
    forward, then ``model.double()``.  Only what the north_star's bar speaks about is stored — R, t and the Patch-PnP outputs of
    every ROI in fp32 and fp64 and the per-ROI distance of the reference's own fp32 forward from its fp64 value — so the fixture
    stays at KB size.  -> net_golden_<ds>_b<b>.npz"""
    import time

    torch.set_num_threads(os.cpu_count())
    torch.set_grad_enabled(False)
    hip_layers.set_enabled(False)
    from core.gdrn_modeling.models import GDRN_double_mask as REFM
    from core.gdrn_modeling.models import net_factory
    from tests.netgolden import net_detections_large

    net_factory.BACKBONES["timm/convnext_base"] = lambda model_name=None, **kw: create_backbone(type="timm/" + model_name, **kw)
    raw = _refimport.load_ref_config(CONFIGS[ds])
    cfg = Config(raw)
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.POSE_NET.BACKBONE.INIT_CFG.pretrained = False
    cfg.TEST.USE_DEPTH_REFINE = True
    cfg.SOLVER.BASE_LR = cfg.SOLVER.OPTIMIZER_CFG["lr"]
    model, opt = REFM.build_model_optimizer(cfg, is_test=True)
    assert opt is None and type(model).__module__ == "core.gdrn_modeling.models.GDRN_double_mask"
    model.eval()
    sd = model.state_dict()
    model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], SEED, alias=norm_alias), strict=True)
    C = cfg.MODEL.POSE_NET.NUM_CLASSES
    x, det = net_image(b, SEED + seed_shift), net_detections_large(C, b)
    coord2d = S.coord2d_roi(det["roi_center"], det["scale"])
    grab = {}
    model.pnp_net.register_forward_hook(lambda m, i, o: grab.update(pred_rot_=o[0].clone(), pred_t_=o[1].clone()))

    def sweep(dtype, chunk, tag):
        acc = {k: [] for k in ("rot", "trans", "pred_rot_", "pred_t_")}
        t0 = time.time()
        for s in range(0, b, chunk):
            sl = slice(s, s + chunk)
            D = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).to(dtype)   # noqa: E731
            out = model(D(x), roi_classes=torch.from_numpy(det["roi_cls"][sl]), roi_cams=D(det["roi_cam"]), roi_whs=D(det["roi_wh"]),
                        roi_centers=D(det["roi_center"]), resize_ratios=D(det["resize_ratio"]), roi_coord_2d=D(coord2d),
                        roi_extents=D(det["roi_extent"]), do_loss=False)
            assert out["rot"].dtype == dtype and grab["pred_rot_"].dtype == dtype
            acc["rot"].append(out["rot"].numpy()); acc["trans"].append(out["trans"].numpy())
            acc["pred_rot_"].append(grab["pred_rot_"].numpy()); acc["pred_t_"].append(grab["pred_t_"].numpy())
            print(ds, b, tag, "chunk", s, f"{time.time() - t0:.0f} s", flush=True)
        return {k: np.concatenate(v) for k, v in acc.items()}

    f32 = sweep(torch.float32, chunk32, "f32")
    model.double()
    f64 = sweep(torch.float64, chunk64, "f64")
    rec = dict(
        cfg_json=json.dumps(jsonable({k: raw[k] for k in ("MODEL", "TEST", "INPUT")})),
        head_keys=json.dumps([[k, list(v.shape)] for k, v in sd.items() if not k.startswith("backbone.")]),
        image_seed=np.int64(SEED + seed_shift),
        roi_cls=det["roi_cls"], roi_cam=det["roi_cam"], roi_wh=det["roi_wh"], roi_center=det["roi_center"],
        resize_ratio=det["resize_ratio"], scale=det["scale"], roi_extent=det["roi_extent"])
    for k in ("rot", "trans", "pred_rot_", "pred_t_"):
        rec[k] = f32[k]
        rec[k + "_f64"] = f64[k]
        d = np.abs(f32[k].astype(np.float64) - f64[k]).reshape(b, -1).max(1)
        rec["ref_f32_err_" + k] = d
        print(ds, b, k, "reference fp32 vs its fp64: max", d.max(), "argmax ROI", int(d.argmax()), "ROIs > 5e-5:", np.nonzero(d > 5e-5)[0].tolist())
    np.savez_compressed(os.path.join(HERE, f"net_golden_{ds}_b{b}.npz"), **rec)
    print("wrote", f"net_golden_{ds}_b{b}.npz")


def record_large_alt(ds, b, chunk=32, seed_shift=1):
    """A SECOND fp32 forward of the reference's module on the batch of record_large, on another fp32 backend of the same PyTorch:
    oneDNN switched off (``torch.backends.mkldnn.flags(enabled=False)``: native im2col + BLAS convolutions, other summation
    orders) — the closest this container gets to "the reference's CUDA path vs its CPU path".  Stored beside the first run
    (``*_alt32``) in net_golden_<ds>_b<b>.npz, so that the tests can state how far two fp32 runs of the REFERENCE'S OWN CODE are from
    each other per ROI: the yardstick for what "within 1e-4 of the reference" can mean at ill-conditioned ROIs."""
    import time

    torch.set_num_threads(os.cpu_count())
    torch.set_grad_enabled(False)
    hip_layers.set_enabled(False)
    from core.gdrn_modeling.models import GDRN_double_mask as REFM
    from core.gdrn_modeling.models import net_factory
    from tests.netgolden import net_detections_large

    net_factory.BACKBONES["timm/convnext_base"] = lambda model_name=None, **kw: create_backbone(type="timm/" + model_name, **kw)
    raw = _refimport.load_ref_config(CONFIGS[ds])
    cfg = Config(raw)
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.POSE_NET.BACKBONE.INIT_CFG.pretrained = False
    cfg.TEST.USE_DEPTH_REFINE = True
    cfg.SOLVER.BASE_LR = cfg.SOLVER.OPTIMIZER_CFG["lr"]
    model, _ = REFM.build_model_optimizer(cfg, is_test=True)
    model.eval()
    sd = model.state_dict()
    model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], SEED, alias=norm_alias), strict=True)
    C = cfg.MODEL.POSE_NET.NUM_CLASSES
    path = os.path.join(HERE, f"net_golden_{ds}_b{b}.npz")
    z = np.load(path)
    rec = {k: z[k] for k in z.files}
    x, det = net_image(b, SEED + seed_shift), net_detections_large(C, b)
    assert np.array_equal(det["roi_cls"], rec["roi_cls"]) and np.array_equal(det["roi_center"], rec["roi_center"])
    coord2d = S.coord2d_roi(det["roi_center"], det["scale"])
    grab = {}
    model.pnp_net.register_forward_hook(lambda m, i, o: grab.update(pred_rot_=o[0].clone(), pred_t_=o[1].clone()))
    acc = {k: [] for k in ("rot", "trans", "pred_rot_", "pred_t_")}
    t0 = time.time()
    with torch.backends.mkldnn.flags(enabled=False):
        for s in range(0, b, chunk):
            sl = slice(s, s + chunk)
            D = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl]))   # noqa: E731
            out = model(D(x), roi_classes=torch.from_numpy(det["roi_cls"][sl]), roi_cams=D(det["roi_cam"]), roi_whs=D(det["roi_wh"]),
                        roi_centers=D(det["roi_center"]), resize_ratios=D(det["resize_ratio"]), roi_coord_2d=D(coord2d),
                        roi_extents=D(det["roi_extent"]), do_loss=False)
            acc["rot"].append(out["rot"].numpy()); acc["trans"].append(out["trans"].numpy())
            acc["pred_rot_"].append(grab["pred_rot_"].numpy()); acc["pred_t_"].append(grab["pred_t_"].numpy())
            print(ds, b, "alt fp32 chunk", s, f"{time.time() - t0:.0f} s", flush=True)
    for k, v in acc.items():
        rec[k + "_alt32"] = np.concatenate(v)
        d = np.abs(rec[k + "_alt32"].astype(np.float64) - rec[k].astype(np.float64)).reshape(b, -1).max(1)
        e = np.abs(rec[k + "_alt32"].astype(np.float64) - rec[k + "_f64"]).reshape(b, -1).max(1)
        print(ds, b, k, "two fp32 runs of the reference: max distance", d.max(), "at ROI", int(d.argmax()), "| alt run vs fp64: max", e.max(), "at ROI", int(e.argmax()))
    np.savez_compressed(path, **rec)
    print("updated", path)


def record_resnet34():
    """BASELINE configs[0]: models/GDRN.py built from configs/_base_/gdrn_base.py (NUM_CLASSES=1 for the single LM-O object;
    the base file's 13 gives the same graph — nothing in it is class-aware), 32 ROIs = the batch of configs[0].
    Full maps for the first 4 ROIs, every second pixel for the rest, R / t / Patch-PnP outputs for all."""
    from core.gdrn_modeling.models import GDRN as REFG
    from core.gdrn_modeling.models import net_factory

    net_factory.BACKBONES["timm/resnet34"] = lambda model_name=None, **kw: create_backbone(type="timm/" + model_name, **kw)
    raw = _refimport.load_ref_config("configs/_base_/gdrn_base.py")
    cfg = Config(raw)
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.POSE_NET.NUM_CLASSES = 1
    cfg.MODEL.POSE_NET.BACKBONE.INIT_CFG.pretrained = False
    cfg.TEST.USE_PNP = True               # configs[0] feeds the maps to (uncertainty-)PnP: forward returns them
    cfg.SOLVER.BASE_LR = cfg.SOLVER.OPTIMIZER_CFG["lr"]
    model, opt = REFG.build_model_optimizer(cfg, is_test=True)
    assert opt is None and type(model).__module__ == "core.gdrn_modeling.models.GDRN"
    assert type(model.geo_head_net).__name__ == "TopDownMaskXyzRegionHead"
    model.eval()
    sd = model.state_dict()
    model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], SEED, alias=norm_alias),
                          strict=True)
    b = 32
    x, det = net_image(b), net_detections(1, b)
    T = torch.from_numpy
    grab = {}
    model.pnp_net.register_forward_hook(lambda m, i, o: grab.update(pred_rot_=o[0].clone(), pred_t_=o[1].clone()))
    model.backbone.register_forward_hook(lambda m, i, o: grab.update(conv_feat=o[0].clone()))
    out = model(T(x), roi_classes=T(det["roi_cls"]), roi_cams=T(det["roi_cam"]), roi_whs=T(det["roi_wh"]),
                roi_centers=T(det["roi_center"]), resize_ratios=T(det["resize_ratio"]),
                roi_coord_2d=T(S.coord2d_roi(det["roi_center"], det["scale"])), roi_extents=T(det["roi_extent"]),
                do_loss=False)
    assert "full_mask" not in out
    region = out["region"].numpy()
    raw["MODEL"]["POSE_NET"]["NUM_CLASSES"] = 1
    raw["TEST"]["USE_PNP"] = True
    rec = dict(
        cfg_json=json.dumps(jsonable({k: raw[k] for k in ("MODEL", "TEST", "INPUT")})),
        head_keys=json.dumps([[k, list(v.shape)] for k, v in sd.items() if not k.startswith("backbone.")]),
        backbone_keys=json.dumps([[k, list(v.shape)] for k, v in sd.items() if k.startswith("backbone.")]),
        roi_cls=det["roi_cls"], roi_cam=det["roi_cam"], roi_wh=det["roi_wh"], roi_center=det["roi_center"],
        resize_ratio=det["resize_ratio"], scale=det["scale"], roi_extent=det["roi_extent"],
        conv_feat_sub=grab["conv_feat"].numpy()[:, ::8], pred_rot_=grab["pred_rot_"].numpy(), pred_t_=grab["pred_t_"].numpy(),
        rot=out["rot"].numpy(), trans=out["trans"].numpy(),
        region_sub=np.ascontiguousarray(region[:, :, 1::8, 2::8]), region_argmax=region.argmax(1).astype(np.uint8),
        region_absmax=np.float32(np.abs(region).max()))
    for k in ("mask", "coor_x", "coor_y", "coor_z"):
        m = out[k].numpy()
        rec[k] = np.ascontiguousarray(m[:4])
        rec[k + "_sub"] = np.ascontiguousarray(m[4:, :, ::2, 1::2])
    for k in ("mask", "coor_x", "rot", "trans", "pred_rot_", "pred_t_"):
        print("lmo_resnet34", k, rec[k].shape, float(np.abs(rec[k]).mean()), float(np.abs(rec[k]).max()))
    np.savez_compressed(os.path.join(HERE, "net_golden_lmo_resnet34.npz"), **rec)
    print("wrote net_golden_lmo_resnet34.npz")


def record_resnet34_f64(b=32):
    """BASELINE configs[0] anchored on the true value: the same reference module (models/GDRN.py, ResNet-34), parameters and 32-ROI
    batch as record_resnet34, evaluated in fp64.  Stored: R, t, the Patch-PnP outputs in fp64 and the per-ROI distance of the
    reference's own fp32 forward (the lmo_resnet34 fixture) from them.  -> net_golden_lmo_resnet34_f64.npz"""
    torch.set_num_threads(os.cpu_count())
    torch.set_grad_enabled(False)
    hip_layers.set_enabled(False)
    from core.gdrn_modeling.models import GDRN as REFG
    from core.gdrn_modeling.models import net_factory

    net_factory.BACKBONES["timm/resnet34"] = lambda model_name=None, **kw: create_backbone(type="timm/" + model_name, **kw)
    raw = _refimport.load_ref_config("configs/_base_/gdrn_base.py")
    cfg = Config(raw)
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.POSE_NET.NUM_CLASSES = 1
    cfg.MODEL.POSE_NET.BACKBONE.INIT_CFG.pretrained = False
    cfg.TEST.USE_PNP = True
    cfg.SOLVER.BASE_LR = cfg.SOLVER.OPTIMIZER_CFG["lr"]
    model, _ = REFG.build_model_optimizer(cfg, is_test=True)
    model.eval()
    sd = model.state_dict()
    model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], SEED, alias=norm_alias), strict=True)
    model.double()
    x, det = net_image(b), net_detections(1, b)
    D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()   # noqa: E731
    grab = {}
    model.pnp_net.register_forward_hook(lambda m, i, o: grab.update(pred_rot_=o[0].clone(), pred_t_=o[1].clone()))
    out = model(D(x), roi_classes=torch.from_numpy(det["roi_cls"]), roi_cams=D(det["roi_cam"]), roi_whs=D(det["roi_wh"]),
                roi_centers=D(det["roi_center"]), resize_ratios=D(det["resize_ratio"]),
                roi_coord_2d=D(S.coord2d_roi(det["roi_center"], det["scale"])), roi_extents=D(det["roi_extent"]), do_loss=False)
    assert out["rot"].dtype == torch.float64
    rec = dict(rot_f64=out["rot"].numpy(), trans_f64=out["trans"].numpy(), pred_rot__f64=grab["pred_rot_"].numpy(),
               pred_t__f64=grab["pred_t_"].numpy())
    f32 = np.load(os.path.join(HERE, "net_golden_lmo_resnet34.npz"))
    for k in ("rot", "trans", "pred_rot_", "pred_t_"):
        d = np.abs(f32[k].astype(np.float64) - rec[k + "_f64"]).reshape(b, -1).max(1)
        rec["ref_f32_err_" + k] = d
        print("lmo_resnet34", k, "reference fp32 vs its fp64: max", d.max(), "ROI", int(d.argmax()))
    np.savez_compressed(os.path.join(HERE, "net_golden_lmo_resnet34_f64.npz"), **rec)
    print("wrote net_golden_lmo_resnet34_f64.npz")


if __name__ == "__main__":
    if "--large-alt" in sys.argv:      # --large-alt tless 1024: adds the second fp32 run (oneDNN off) to an existing fixture
        i = sys.argv.index("--large-alt")
        record_large_alt(sys.argv[i + 1], int(sys.argv[i + 2]))
    elif "--large" in sys.argv:        # --large tless 1024 | --large ycbv 512
        i = sys.argv.index("--large")
        record_large(sys.argv[i + 1], int(sys.argv[i + 2]))
    elif "--b128-only" in sys.argv:
        record_b128()
    elif "--b128-f64" in sys.argv:
        record_b128_f64()
    elif "--resnet34-f64" in sys.argv:
        record_resnet34_f64()
    elif "--resnet34-only" in sys.argv:
        torch.set_num_threads(os.cpu_count())
        torch.set_grad_enabled(False)
        hip_layers.set_enabled(False)
        record_resnet34()
    elif "--so-only" in sys.argv:
        main(SO_CONFIGS)
    else:
        main()
        record_resnet34()
        record_b128()
