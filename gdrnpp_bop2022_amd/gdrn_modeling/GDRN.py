"""``core/gdrn_modeling/models/GDRN.py`` boundary: the reference selects its pose network with
``eval(cfg.MODEL.POSE_NET.NAME)`` over the modules imported in main_gdrn.py:41-48, so a config with ``NAME="GDRN"``
(configs/_base_/gdrn_base.py:18 — BASELINE configs[0]: ResNet-34, single mask, class-agnostic
``TopDownMaskXyzRegionHead``) needs a module ``GDRN`` exposing ``GDRN`` and ``build_model_optimizer``
(GDRN.py:35-205,480-567).  One forward implementation serves both variants (GDRN_double_mask.py here): the single-mask
graph is the double-mask graph without the ``full_mask`` block, exactly as in the reference's two files."""
from .GDRN_double_mask import GDRN_DoubleMask, build_model_optimizer as _build, load_checkpoint  # noqa: F401


class GDRN(GDRN_DoubleMask):
    def __init__(self, cfg, backbone, geo_head_net, neck=None, pnp_net=None):
        assert cfg.MODEL.POSE_NET.NAME == "GDRN", cfg.MODEL.POSE_NET.NAME       # GDRN.py:38
        super().__init__(cfg, backbone, geo_head_net, neck=neck, pnp_net=pnp_net)


def build_model_optimizer(cfg, is_test=True):
    """GDRN.py:480-567 — (model, None) for the inference build."""
    assert cfg.MODEL.POSE_NET.NAME == "GDRN", cfg.MODEL.POSE_NET.NAME
    return _build(cfg, is_test=is_test, model_cls=GDRN)
