"""Output errors of the 128-ROI reference fixtures (tests/golden/net_golden_<ds>_b128.npz) per GEMM engine configuration:
default (fused stage-0 MLP, three products, f16x2 rows) / two launches / fp32 hand-over / six products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import netgolden as NG
from gdrnpp_bop2022_amd import hip_lib as hip
from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

for ds in ("ycbv", "tless"):
    fx = NG.load_fixture(ds + "_b128")
    f64 = np.load(os.path.join(NG.GOLDEN, f"net_golden_{ds}_b128_f64.npz"))     # the reference's module in fp64 (make_golden_net.py record_b128_f64)
    ref_rot, ref_pr = f64["ref_f32_err_rot"], f64["ref_f32_err_pred_rot_"]
    print(f"{ds:6s} reference fp32 vs its own fp64: rot max {ref_rot.max():.3e} (ROI {int(ref_rot.argmax())})  pred_rot_ max {ref_pr.max():.3e}  "
          f"trans max {f64['ref_f32_err_trans'].max():.3e}")
    cfg = get_cfg(NG.cfg_name(ds), opts=["TEST.USE_DEPTH_REFINE=True"])
    model, _ = build_model_optimizer(cfg)
    model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)
    x = torch.from_numpy(NG.net_image(128)).cuda()
    kw = NG.forward_kwargs(fx, "cuda")
    # conditioning of rot6d -> R per ROI: 1 / (|a1| * sin(angle(a1, a2)) * |a2|) style amplification
    p = fx["pred_rot_"].reshape(128, -1)
    for name, setup in (("default", lambda: None), ("two launches", lambda: hip_layers.set_fused_mlp_x3(False)),
                        ("fp32 hand-over", lambda: hip_layers.set_f16x2_rows(False)), ("six products", lambda: hip_layers.set_gemm_products(6))):
        hip_layers.set_fused_mlp_x3(True); hip_layers.set_f16x2_rows(True); hip_layers.set_gemm_products(3)
        setup()
        with torch.no_grad():
            out = model(x, **kw)
            rot_, t_, _ = model.forward_maps(x, kw["roi_classes"], kw["roi_coord_2d"], None, kw["roi_extents"])
        e_rot = np.abs(out["rot"].cpu().numpy() - fx["rot"]).reshape(128, -1).max(1)
        e_pr = np.abs(rot_.cpu().numpy() - fx["pred_rot_"]).reshape(128, -1).max(1)
        e_t = np.abs(out["trans"].cpu().numpy() - fx["trans"]).max()
        o_rot, o_pr = out["rot"].double().cpu().numpy(), rot_.double().cpu().numpy()
        d_rot = np.abs(o_rot - f64["rot_f64"]).reshape(128, -1).max(1)
        d_pr = np.abs(o_pr - f64["pred_rot__f64"]).reshape(128, -1).max(1)
        d_t = np.abs(out["trans"].double().cpu().numpy() - f64["trans_f64"]).max()
        excess = d_rot - ref_rot
        print(f"{ds:6s} {name:15s} vs fp64: rot max {d_rot.max():.3e} (ROI {int(d_rot.argmax())}; reference there {ref_rot[int(d_rot.argmax())]:.3e})  "
              f"pred_rot_ max {d_pr.max():.3e} (reference {ref_pr.max():.3e})  trans {d_t:.3e}  max (ours - reference's) rot error {excess.max():.3e} "
              f"(ROI {int(excess.argmax())})  ROIs where ours > reference's + 2e-5: {np.nonzero(excess > 2e-5)[0].tolist()}  "
              f"ROIs where ours < reference's: {int((excess < 0).sum())}/128")
        worst = int(e_rot.argmax())
        print(f"{ds:6s} {name:15s} rot max {e_rot.max():.3e} (ROI {worst}, its pred_rot_ err {e_pr[worst]:.3e}, |pred_rot_| {np.abs(p[worst]).max():.3f})"
              f"  pred_rot_ max {e_pr.max():.3e}  trans {e_t:.3e}  rot err 2nd worst {np.sort(e_rot)[-2]:.3e} median {np.median(e_rot):.2e}")
    hip_layers.set_fused_mlp_x3(True); hip_layers.set_f16x2_rows(True); hip_layers.set_gemm_products(3)
    print(ds, "pred_rot_ of worst ROI:", p[worst])
