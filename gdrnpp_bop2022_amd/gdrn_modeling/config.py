"""Config surface kept from the reference (mmcv python-dict configs with ``_base_`` chains,
configs/_base_/gdrn_base.py:1-174, configs/_base_/common_base.py:212-222) — restated as plain
dict builders with attribute access; only keys that drive the inference hot path are carried.
``--opts KEY=VAL`` style overrides: ``merge_opts`` (core/utils/default_args_setup.py:91-96).
"""
from __future__ import annotations

import ast
import copy


class Config(dict):
    """dict with attribute access, recursively (stand-in for mmcv.Config / ConfigDict)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, Config):
            v = Config(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__

    def get_path(self, path):
        cur = self
        for p in path.split("."):
            cur = cur[p]
        return cur

    def set_path(self, path, value):
        cur = self
        parts = path.split(".")
        for p in parts[:-1]:
            cur = cur[p]
        cur[parts[-1]] = value

    def copy(self):
        return Config(copy.deepcopy(dict(self)))


def _merge(base: dict, child: dict) -> dict:
    """mmcv `_base_` merge: dicts merge recursively unless the child carries `_delete_=True`."""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            v = copy.deepcopy(v)
            if isinstance(v, dict):
                v.pop("_delete_", None)
            out[k] = v
    return out


def gdrn_base() -> dict:
    """configs/_base_/gdrn_base.py:5-174 + the TEST/INPUT keys of common_base.py the path reads."""
    return dict(
        INPUT=dict(DZI_PAD_SCALE=1.0, WITH_DEPTH=False, BP_DEPTH=False),   # common_base.py:60
        MODEL=dict(
            DEVICE="cuda", WEIGHTS="", PIXEL_MEAN=[0, 0, 0], PIXEL_STD=[255.0, 255.0, 255.0], LOAD_DETS_TEST=False,
            POSE_NET=dict(
                NAME="GDRN", NUM_CLASSES=13, USE_MTL=False, INPUT_RES=256, OUTPUT_RES=64,
                BACKBONE=dict(FREEZE=False, PRETRAINED="timm",
                              INIT_CFG=dict(type="timm/resnet34", in_chans=3, features_only=True, pretrained=True,
                                            out_indices=(4,))),
                NECK=dict(ENABLED=False),
                GEO_HEAD=dict(
                    FREEZE=False, LR_MULT=1.0,
                    INIT_CFG=dict(type="TopDownMaskXyzRegionHead", in_dim=512,
                                  up_types=("deconv", "bilinear", "bilinear"), deconv_kernel_size=3,
                                  num_conv_per_block=2, feat_dim=256, feat_kernel_size=3, norm="GN", num_gn_groups=32,
                                  act="GELU", out_kernel_size=1, out_layer_shared=True),
                    XYZ_BIN=64, XYZ_CLASS_AWARE=False, MASK_CLASS_AWARE=False, REGION_CLASS_AWARE=False,
                    MASK_THR_TEST=0.5, NUM_REGIONS=64),
                PNP_NET=dict(
                    FREEZE=False, LR_MULT=1.0,
                    INIT_CFG=dict(type="ConvPnPNet", norm="GN", act="relu", num_gn_groups=32, drop_prob=0.0,
                                  denormalize_by_extent=True),
                    WITH_2D_COORD=False, COORD_2D_TYPE="abs", REGION_ATTENTION=False, MASK_ATTENTION="none",
                    ROT_TYPE="ego_rot6d", TRANS_TYPE="centroid_z", Z_TYPE="REL"),
                LOSS_CFG=dict(XYZ_LOSS_TYPE="L1", MASK_LOSS_TYPE="L1", FULL_MASK_LOSS_TYPE="BCE"),
            ),
        ),
        TEST=dict(EVAL_PERIOD=0, VIS=False, TEST_BBOX_TYPE="est", USE_PNP=False, SAVE_RESULTS_ONLY=False,
                  PNP_TYPE="ransac_pnp", USE_DEPTH_REFINE=False, DEPTH_REFINE_ITER=2, DEPTH_REFINE_THRESHOLD=0.8,
                  USE_COOR_Z_REFINE=False, AMP_TEST=False),
        # configs/_base_/common_base.py VAL block (the keys that name the BOP results file, test_utils.py:33-52) and EXP_ID
        # (core/utils/default_args_setup.py derives it from the config file name)
        VAL=dict(DATASET_NAME="lm", SPLIT="test", SPLIT_TYPE="", SAVE_BOP_CSV_ONLY=True),
        EXP_ID="gdrn_hip",
        DIST_PARAMS=dict(backend="nccl"),
    )


def _gdrnpp_convnext(num_classes: int) -> dict:
    """configs/gdrn/ycbv/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_ycbv.py:64-135
    (identical MODEL block in the tless/lmo/... convnext_a6 configs apart from NUM_CLASSES)."""
    return dict(
        INPUT=dict(DZI_PAD_SCALE=1.5),
        MODEL=dict(
            LOAD_DETS_TEST=True, PIXEL_MEAN=[0.0, 0.0, 0.0], PIXEL_STD=[255.0, 255.0, 255.0],
            POSE_NET=dict(
                NAME="GDRN_double_mask", NUM_CLASSES=num_classes,
                BACKBONE=dict(FREEZE=False, PRETRAINED="timm",
                              INIT_CFG=dict(type="timm/convnext_base", pretrained=True, in_chans=3,
                                            features_only=True, out_indices=(3,))),
                GEO_HEAD=dict(FREEZE=False, INIT_CFG=dict(type="TopDownDoubleMaskXyzRegionHead", in_dim=1024),
                              NUM_REGIONS=64, XYZ_CLASS_AWARE=True, MASK_CLASS_AWARE=True, REGION_CLASS_AWARE=True),
                PNP_NET=dict(INIT_CFG=dict(norm="GN", act="gelu"), REGION_ATTENTION=True, WITH_2D_COORD=True,
                             ROT_TYPE="allo_rot6d", TRANS_TYPE="centroid_z"),
                LOSS_CFG=dict(XYZ_LOSS_TYPE="L1", MASK_LOSS_TYPE="L1", FULL_MASK_LOSS_TYPE="L1"),
            ),
        ),
        TEST=dict(EVAL_PERIOD=0, VIS=False, TEST_BBOX_TYPE="est"),
    )


# VAL.DATASET_NAME of the seven convnext_a6 configs (configs/gdrn/<dataset>/convnext_a6_*_classAware_<dataset>.py); all of
# them set SPLIT="test", SPLIT_TYPE=""
_VAL_DATASET_NAME = {"hb": "hbs"}


def get_cfg(name: str = "ycbv_convnext_a6", opts=None) -> Config:
    """Named configs covering BASELINE.json's `configs`."""
    base = gdrn_base()
    num_cls = {"ycbv": 21, "tless": 30, "lmo": 8, "icbin": 2, "hb": 16, "itodd": 28, "tudl": 3}
    if name.endswith("_convnext_a6"):
        ds = name.split("_")[0]
        cfg = _merge(base, _gdrnpp_convnext(num_cls[ds]))
        cfg = _merge(cfg, dict(VAL=dict(DATASET_NAME=_VAL_DATASET_NAME.get(ds, ds), SPLIT="test", SPLIT_TYPE=""), EXP_ID=name))
    elif name.endswith("_convnext_so"):
        # the ten single-object families (configs/gdrn/{ycbv,tless,tudl}SO, {ycbv,lmo,tless,tudl,icbin,itodd,hb}PbrSO:
        # one config per object, e.g. ycbvSO/convnext_AugCosyAAEGray_DMask_amodalClipBox_ycbv/002_master_chef_can.py): the
        # convnext_a6 MODEL block with a class-AGNOSTIC head; NUM_CLASSES keeps gdrn_base.py's 13 and is not used
        ds = name.split("_")[0]
        cfg = _merge(base, _gdrnpp_convnext(13))
        cfg = _merge(cfg, dict(MODEL=dict(POSE_NET=dict(GEO_HEAD=dict(XYZ_CLASS_AWARE=False, MASK_CLASS_AWARE=False,
                                                                     REGION_CLASS_AWARE=False))),
                               VAL=dict(DATASET_NAME=_VAL_DATASET_NAME.get(ds, ds), SPLIT="test", SPLIT_TYPE=""), EXP_ID=name))
    elif name == "lmo_resnet34_ape":
        # BASELINE config 1: base GDRN (ResNet-34, single object, class-agnostic head)
        # = configs/_base_/gdrn_base.py with one class; the ROI padding of every shipped single-object LM-O config
        # (configs/gdrn/lmoPbrSO/convnext_AugCosyAAEGray_DMask_amodalClipBox_lmo/ape.py:6), the base file's is 1.0
        cfg = _merge(base, dict(INPUT=dict(DZI_PAD_SCALE=1.5), MODEL=dict(POSE_NET=dict(NUM_CLASSES=1))))
    else:
        raise KeyError(name)
    cfg = Config(cfg)
    merge_opts(cfg, opts or [])
    return cfg


def merge_opts(cfg: Config, opts) -> Config:
    """`--opts TEST.USE_DEPTH_REFINE=True INPUT.WITH_DEPTH=True` (test_gdrn_depth_refine.sh:22-26)."""
    for o in opts:
        k, v = o.split("=", 1)
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass
        cfg.set_path(k, v)
    return cfg
