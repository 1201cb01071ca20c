"""-m gpu: the HIP network path (split GEMM / implicit-GEMM convs, NHWC kernels, class-sliced output layer, on-device pose)
against the outputs of the reference's own GDRN_DoubleMask.forward recorded by tests/golden/make_golden_net.py.
Tolerances: maps 1e-4 of their scale, rot / trans 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer
from tests import netgolden as NG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["ycbv", "tless", "ycbvso"])
def case(request, hip):
    ds = request.param
    fx = NG.load_fixture(ds)
    cfg = get_cfg(NG.cfg_name(ds), opts=["TEST.USE_DEPTH_REFINE=True"])
    model, _ = build_model_optimizer(cfg)
    assert next(model.parameters()).is_cuda
    x = torch.from_numpy(NG.net_image()).cuda()
    kw = NG.forward_kwargs(fx, "cuda")
    with torch.no_grad():
        model(x, **kw)                                    # warm the eval-time weight caches with the random init ...
    model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)   # ... then load: caches must refresh
    with torch.no_grad():
        out = model(x, **kw)
    torch.cuda.synchronize()
    return fx, {k: v.cpu().numpy() for k, v in out.items()}


def _err(a, ref, scale=None):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / (np.abs(ref).max() if scale is None else scale)


def test_hip_maps_match_reference_forward(case):
    fx, out = case
    for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z"):
        assert out[k].shape == fx[k].shape
        assert _err(out[k], fx[k]) <= 1e-4, k
    region = out["region"]
    assert _err(region[:, :, 1::4, 2::4], fx["region_sub"], float(fx["region_absmax"])) <= 1e-4
    assert (region.argmax(1) == fx["region_argmax"]).mean() > 0.999


def test_hip_pose_matches_reference_forward(case):
    fx, out = case
    assert np.abs(out["rot"] - fx["rot"]).max() <= 1e-4
    assert np.abs(out["trans"] - fx["trans"]).max() <= 1e-4 * max(1.0, np.abs(fx["trans"]).max())


def test_config_surface_variants_run_through_forward_and_post(hip):
    """MASK_LOSS_TYPE="CE" (engine_utils.py:329-330), ROT_TYPE quaternion, TRANS_TYPE centroid_z_abs / trans
    (GDRN_double_mask.py:162-200): build, forward, post-process — the decoded CE mask equals torch.argmax, poses are finite."""
    from gdrnpp_bop2022_amd.gdrn_modeling.engine import GdrnHipPost

    fx = NG.load_fixture("ycbv")
    x = torch.from_numpy(NG.net_image()).cuda()
    kw = NG.forward_kwargs(fx, "cuda")
    for opts in (["MODEL.POSE_NET.LOSS_CFG.MASK_LOSS_TYPE=CE", "TEST.USE_PNP=True", "TEST.PNP_TYPE=net_iter_pnp"],
                 ["MODEL.POSE_NET.PNP_NET.ROT_TYPE=ego_quat", "MODEL.POSE_NET.PNP_NET.TRANS_TYPE=trans"],
                 ["MODEL.POSE_NET.PNP_NET.ROT_TYPE=allo_quat", "MODEL.POSE_NET.PNP_NET.TRANS_TYPE=centroid_z_abs"]):
        cfg = get_cfg("ycbv_convnext_a6", opts=opts + ["TEST.SAVE_RESULTS_ONLY=True"])
        torch.manual_seed(0)
        model, _ = build_model_optimizer(cfg)
        with torch.no_grad():
            out = model(x, **kw)
        assert torch.isfinite(out["rot"]).all() and torch.isfinite(out["trans"]).all()
        R = out["rot"].double()
        assert (R @ R.transpose(1, 2) - torch.eye(3, device="cuda", dtype=torch.float64)).abs().max().item() < 1e-5
        if "CE" in opts[0]:
            assert out["mask"].shape[1] == 2
            post = GdrnHipPost(cfg)
            batch = dict(roi_cam=kw["roi_cams"], roi_coord_2d=kw["roi_coord_2d"], roi_extent=kw["roi_extents"],
                         im_W=torch.full((NG.B,), 640.0, device="cuda"), im_H=torch.full((NG.B,), 480.0, device="cuda"),
                         roi_cls=kw["roi_classes"])
            count, _, _, _, m = post.process_correspondences(batch, out)
            assert torch.equal(m, torch.argmax(out["mask"], 1, keepdim=True).float())
            rec = post.process(batch, out)
            assert rec.shape == (NG.B, 16) and torch.isfinite(rec).all()


@pytest.mark.parametrize("ds", ["ycbv", "tless", "ycbvso"])
def test_fused_head_tail_matches_module_path(hip, ds):
    """The all-NHWC head tail (grouped class-sliced GEMM, head_tail kernel, Patch-PnP's first convolution with Cin padded to 96)
    against the module path (baddbmm + torch softmax / cat + MIOpen / CK convolution) on the same weights: maps and Patch-PnP
    outputs agree to fp32 rounding; the fused path is really the one taken by default — also for the class-agnostic head of the
    single-object configs ("ycbvso": one slice, every ROI selects it)."""
    fx = NG.load_fixture(ds)
    cfg = get_cfg(NG.cfg_name(ds), opts=["TEST.USE_DEPTH_REFINE=True"])
    model, _ = build_model_optimizer(cfg)
    model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)
    x = torch.from_numpy(NG.net_image()).cuda()
    kw = NG.forward_kwargs(fx, "cuda")
    from gdrnpp_bop2022_amd import hip_lib
    with torch.no_grad():
        assert model.fused_head_tail
        timer = hip_lib.LaunchTimer()
        hip_lib.set_launch_timer(timer)
        try:
            fused = model(x, **kw)
        finally:
            hip_lib.set_launch_timer(None)
        assert sum(1 for r in timer.records if r[0] == "linear_grouped") == 1
        model.fused_head_tail = False
        plain = model(x, **kw)
    assert not fused["region"].is_contiguous() and plain["region"].shape == fused["region"].shape   # NHWC view vs NCHW tensor
    for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z", "region"):
        a, b = fused[k].float(), plain[k].float()
        assert a.shape == b.shape
        assert ((a - b).abs().max() / b.abs().max()).item() < 2e-5, k
    assert (fused["rot"] - plain["rot"]).abs().max().item() < 2e-5
    assert (fused["trans"] - plain["trans"]).abs().max().item() < 2e-5 * max(1.0, plain["trans"].abs().max().item())


# ---- BASELINE configs[0]: models/GDRN.py + TopDownMaskXyzRegionHead + ResNet-34, reference outputs recorded at 32 ROIs ------
@pytest.fixture(scope="module")
def resnet_model(hip):
    from gdrnpp_bop2022_amd.gdrn_modeling import GDRN as G
    fx = NG.load_fixture("lmo_resnet34")
    cfg = get_cfg("lmo_resnet34_ape", opts=["TEST.USE_PNP=True"])
    model, _ = G.build_model_optimizer(cfg)
    assert next(model.parameters()).is_cuda
    x = torch.from_numpy(NG.net_image(32)).cuda()
    kw = NG.forward_kwargs(fx, "cuda")
    with torch.no_grad():
        model(x[:2], **{k: v[:2] for k, v in kw.items()})     # warm the BatchNorm-fold / packed-weight caches on the random init
    model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)   # ... which the load must invalidate
    return fx, model, x, kw


def test_resnet34_hip_path_matches_reference_forward_32_rois(resnet_model):
    """The folded-BatchNorm / bias_act / split implicit-GEMM path of the ResNet-34 configuration (layer 2 on the split
    convolution at 32 ROIs) against GDRN.forward of the reference: maps 1e-4 of scale, R / t 1e-4."""
    from tests.test_net_golden import check_resnet34_outputs
    fx, model, x, kw = resnet_model
    with torch.no_grad():
        out = model(x, **kw)
        rot6, t3, _ = model.forward_maps(x, kw["roi_classes"], kw["roi_coord_2d"], None, kw["roi_extents"])
    torch.cuda.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    check_resnet34_outputs(fx, o, rot6.cpu().numpy(), t3.cpu().numpy(), 1e-4, 1e-4)
    assert np.abs(o["rot"] - fx["rot"]).max() <= 1e-4
    assert np.abs(o["trans"] - fx["trans"]).max() <= 1e-4 * max(1.0, np.abs(fx["trans"]).max())
    # anchored on the reference module's fp64 run of the same 32 ROIs (net_golden_lmo_resnet34_f64.npz): per ROI as close to the
    # true value as the reference's own fp32 forward (+ 2e-5)
    f64 = NG.load_f64_fixture("lmo_resnet34")
    got = {"rot": o["rot"], "trans": o["trans"], "pred_rot_": rot6.cpu().numpy(), "pred_t_": t3.cpu().numpy()}
    for k in ("rot", "trans", "pred_rot_", "pred_t_"):
        ours = np.abs(got[k].astype(np.float64) - f64[k + "_f64"]).reshape(32, -1).max(1)
        worse = np.nonzero(ours > f64["ref_f32_err_" + k] + 2e-5)[0]
        assert worse.size == 0, f"{k}: ROIs {worse.tolist()} {ours[worse]} vs reference {f64['ref_f32_err_' + k][worse]}"


def test_resnet34_hip_path_matches_reference_forward_4_rois(resnet_model):
    """Same at 4 ROIs (eval-mode BatchNorm: a ROI's outputs do not depend on the batch) — below the tile-count thresholds,
    so the small-problem kernels serve the layers the 32-ROI case sends to the 256-row tiles."""
    fx, model, x, kw = resnet_model
    with torch.no_grad():
        out = model(x[:4], **{k: v[:4] for k, v in kw.items()})
    torch.cuda.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    for k in ("mask", "coor_x", "coor_y", "coor_z"):
        scale = max(np.abs(fx[k]).max(), np.abs(fx[k + "_sub"]).max())
        assert _err(o[k], fx[k], scale) <= 1e-4, k
    assert _err(o["region"][:, :, 1::8, 2::8], fx["region_sub"][:4], float(fx["region_absmax"])) <= 1e-4
    assert np.abs(o["rot"] - fx["rot"][:4]).max() <= 1e-4
    assert np.abs(o["trans"] - fx["trans"][:4]).max() <= 1e-4 * max(1.0, np.abs(fx["trans"]).max())


# ---- the benchmark's batch: BASELINE configs[2] (YCB-V, 128 ROIs) and one rank's 128-ROI shard of configs[3] (T-LESS) ----------
@pytest.mark.parametrize("ds", ["ycbv", "tless"])
def test_default_path_matches_reference_forward_at_128_rois(hip, ds):
    """The library's DEFAULT configuration (three-product split GEMMs where the launch is large enough, as bench.py times it)
    against GDRN_DoubleMask.forward of the reference recorded at 128 ROIs with every class of the dataset present
    (net_golden_<ds>_b128.npz, tests/golden/make_golden_net.py record_b128): R / t / Patch-PnP outputs of ALL 128 ROIs and the
    stored sub-sampling of all their maps, tolerances of BASELINE.json's north_star; no layer leaves the three-product range."""
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers

    assert hip_layers.gemm_products() == 3 and hip_layers.mlp_gemm() == "split"
    fx = NG.load_fixture(ds + "_b128")
    b = fx["rot"].shape[0]
    assert b == 128 and len(set(fx["roi_cls"].tolist())) == fx["cfg"]["MODEL"]["POSE_NET"]["NUM_CLASSES"]
    cfg = get_cfg(NG.cfg_name(ds), opts=["TEST.USE_DEPTH_REFINE=True"])
    model, _ = build_model_optimizer(cfg)
    model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)
    x = torch.from_numpy(NG.net_image(b)).cuda()
    kw = NG.forward_kwargs(fx, "cuda")
    timer = hip.LaunchTimer()
    hip.set_launch_timer(timer)
    try:
        with torch.no_grad():
            out = model(x, **kw)
            rot_, t_, _ = model.forward_maps(x, kw["roi_classes"], kw["roi_coord_2d"], None, kw["roi_extents"])
    finally:
        hip.set_launch_timer(None)
    words = hip.split2_range_words()
    kinds = [r[0] for r in timer.records]
    n_lin, n_fused = sum(k == "linear" + hip.X3 for k in kinds), sum(k == "mlp_fused" + hip.X3 for k in kinds)
    assert n_fused == 2 * 6 and n_lin + 2 * n_fused == 2 * 72, "the ConvNeXt MLPs did not run on the three-product kernels (stages 0 and 1 fused)"
    assert words == {}, f"three-product launches left their range: {words}"
    o = {k: v.float().cpu().numpy() for k, v in out.items()}
    for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z"):
        assert _err(o[k][:, :, ::4, 1::4], fx[k + "_sub"], float(fx[k + "_absmax"])) <= 1e-4, k
    assert _err(o["region"][:, :, 5::16, 9::16], fx["region_sub"], float(fx["region_absmax"])) <= 1e-4
    assert (o["region"].argmax(1) == fx["region_argmax"]).mean() > 0.999
    assert np.abs(rot_.cpu().numpy() - fx["pred_rot_"]).max() <= 1e-4 * max(1.0, np.abs(fx["pred_rot_"]).max())
    assert np.abs(t_.cpu().numpy() - fx["pred_t_"]).max() <= 1e-4 * max(1.0, np.abs(fx["pred_t_"]).max())
    # R / t: the plain 1e-4 bar of BASELINE.json's north_star on EVERY ROI — no conditioning clause.  (R is the Gram-Schmidt of
    # the 6-D output; T-LESS ROI 16 amplifies 25x and is still inside: 3.9e-5, profiles/r05a_b128_engine_errors_vs_fp64.txt.)
    e_rot = np.abs(o["rot"] - fx["rot"]).reshape(b, -1).max(1)
    assert (e_rot <= 1e-4).all(), f"ROIs beyond 1e-4: {np.nonzero(e_rot > 1e-4)[0].tolist()} {e_rot.max():.3e}"
    # ... and anchored on the TRUE value of the reference's function: its own module evaluated in fp64 on the same parameters and
    # batch (net_golden_<ds>_b128_f64.npz).  For every ROI this path is as close to the fp64 result as the reference's fp32
    # forward is (+ 2e-5): on 74 of the 128 T-LESS ROIs it is closer.
    f64 = NG.load_f64_fixture(ds)
    got = {"rot": o["rot"], "trans": o["trans"], "pred_rot_": rot_.cpu().numpy(), "pred_t_": t_.cpu().numpy()}
    for k in ("rot", "trans", "pred_rot_", "pred_t_"):
        ours = np.abs(got[k].astype(np.float64) - f64[k + "_f64"]).reshape(b, -1).max(1)
        ref = f64["ref_f32_err_" + k]
        worse = np.nonzero(ours > ref + 2e-5)[0]
        assert worse.size == 0, f"{k}: farther from the fp64 value than the reference's fp32 forward + 2e-5 at ROIs {worse.tolist()} ({ours[worse]}, reference {ref[worse]})"
        assert ours.max() <= 1e-4 * max(1.0, np.abs(f64[k + "_f64"]).max())
    for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z"):       # the maps against their fp64 values, same sub-sampling
        assert _err(o[k][:, :, ::4, 1::4], f64[k + "_sub_f64"], float(fx[k + "_absmax"])) <= 1e-4, k
    assert np.abs(o["trans"] - fx["trans"]).max() <= 1e-4 * max(1.0, np.abs(fx["trans"]).max())


# ---- the ITERATION sizes BASELINE.json names beyond one 128-ROI step: configs[3]'s 1 024 T-LESS ROIs, a 512-ROI YCB-V step ----------
def check_iteration_size_outputs(fx, got, b, tag):
    """Shared by the GPU test below and tools/large_parity_dump.py (which also runs the exact six-product form through it).
    ``got``: rot / trans / pred_rot_ / pred_t_ of one forward.  Returns the report lines; raises AssertionError.

    The NETWORK's outputs (6-D rotation, centroid / z, and the translation formed from them), per ROI, no exemption:
      (a) distance from the fp64 value <= the reference's own fp32 forward's distance + 2e-5;
      (b) within 1e-4 of the reference's fp32 forward;  (c) within 1e-4 of the fp64 value.
    R = the reference's pose function of those outputs (Gram-Schmidt, allocentric -> egocentric), which AMPLIFIES their error
    where the 6-D vectors are short or nearly parallel, or the predicted centroid lies near the principal point (acos / axis
    normalisation; the seeded parameters even put some objects behind the camera).  Its conditioning is computed per ROI from the
    fp64 fixture (tests/netgolden.py ego_rot_sensitivity: |dR/dx_j| for the nine network outputs):
      (d) EVERY ROI: R error <= 1.25 * sum_j |dR/dx_j| * |error of output j| + 3e-6 — the whole R error is explained by the
          fp32-level error of the outputs, held by (a)-(c), times the conditioning of the reference's own function there;
      (e) every WELL-CONDITIONED ROI — 4 sigma of the REFERENCE's own fp32 noise (rms error of its outputs over this fixture)
          times the conditioning stays below 5e-5, so any two forwards at that noise level agree within 1e-4 — : R within 5e-5 of
          the fp64 value and within 1e-4 of the reference's fp32 forward (the north_star's bar);
      (f) at most 4 % of the ROIs are ill-conditioned; each is listed with its conditioning and all three distances.  The
          reference's own fp32 forward exceeds 1e-4 at one of them (T-LESS ROI 30: 1.2e-4, tests/test_net_golden.py)."""
    report = []
    for k in ("pred_rot_", "pred_t_", "trans"):
        f64, ref32, ref_err = fx[k + "_f64"], fx[k].astype(np.float64), fx["ref_f32_err_" + k]
        scale = max(1.0, float(np.abs(f64).max()))
        ours64 = np.abs(got[k].astype(np.float64) - f64).reshape(b, -1).max(1)
        ours32 = np.abs(got[k].astype(np.float64) - ref32).reshape(b, -1).max(1)
        report.append(f"{k}: {tag} vs fp64 max {ours64.max():.2e} mean {ours64.mean():.2e} (reference fp32 vs fp64: max {ref_err.max():.2e} mean {ref_err.mean():.2e}); "
                      f"{tag} vs reference fp32 max {ours32.max():.2e}")
        worse = np.nonzero(ours64 > ref_err + 2e-5)[0]
        assert worse.size == 0, f"{k}: farther from the fp64 value than the reference's fp32 forward + 2e-5 at ROIs {worse.tolist()} ({ours64[worse]}, reference {ref_err[worse]})"
        assert ours32.max() <= 1e-4 * scale, f"{k}: {ours32.max():.3e} from the reference's fp32 forward at ROI {int(ours32.argmax())}"
        assert ours64.max() <= 1e-4 * scale, f"{k}: {ours64.max():.3e} from the fp64 value at ROI {int(ours64.argmax())}"
    S = NG.ego_rot_sensitivity(fx["pred_rot__f64"], fx["pred_t__f64"], fx)                      # [b, 9]
    x64 = np.concatenate([fx["pred_rot__f64"], fx["pred_t__f64"]], 1)
    dx = np.abs(np.concatenate([got["pred_rot_"], got["pred_t_"]], 1).astype(np.float64) - x64)
    ours64 = np.abs(got["rot"].astype(np.float64) - fx["rot_f64"]).reshape(b, -1).max(1)
    ours32 = np.abs(got["rot"].astype(np.float64) - fx["rot"].astype(np.float64)).reshape(b, -1).max(1)
    ref_err = fx["ref_f32_err_rot"]
    bound = 1.25 * (S * dx).sum(1) + 3e-6
    over = np.nonzero(ours64 > bound)[0]                                                          # (d)
    assert over.size == 0, f"rot: error not explained by the error of the network outputs x conditioning at ROIs {over.tolist()} ({ours64[over]} > {bound[over]})"
    ref_dx = np.concatenate([fx["pred_rot_"], fx["pred_t_"]], 1).astype(np.float64) - x64
    sigma = np.concatenate([np.full(6, np.sqrt((ref_dx[:, :6] ** 2).mean())), np.full(3, np.sqrt((ref_dx[:, 6:] ** 2).mean()))])
    cond = 4.0 * np.sqrt(((S * sigma[None]) ** 2).sum(1))
    well = cond <= 5e-5
    ill = np.nonzero(~well)[0]
    report.append(f"rot: {tag} vs fp64 max {ours64.max():.2e} mean {ours64.mean():.2e} (reference fp32 vs fp64: max {ref_err.max():.2e} mean {ref_err.mean():.2e}); "
                  f"well-conditioned ROIs ({int(well.sum())} of {b}): {tag} vs fp64 max {ours64[well].max():.2e}, vs reference fp32 max {ours32[well].max():.2e}, "
                  f"reference fp32 vs fp64 max {ref_err[well].max():.2e}")
    assert ours64[well].max() <= 5e-5, f"rot: {ours64[well].max():.3e} from the fp64 value at well-conditioned ROI {int(np.nonzero(well)[0][ours64[well].argmax()])}"   # (e)
    assert ours32[well].max() <= 1e-4, f"rot: {ours32[well].max():.3e} from the reference's fp32 forward at a well-conditioned ROI"
    assert ill.size <= 0.04 * b, f"{ill.size} of {b} ROIs rated ill-conditioned"                  # (f)
    alt64 = alt32 = None
    if "rot_alt32" in fx:       # the reference's own code on PyTorch's other fp32 backend (oneDNN off): reference vs reference
        alt64 = np.abs(fx["rot_alt32"].astype(np.float64) - fx["rot_f64"]).reshape(b, -1).max(1)
        alt32 = np.abs(fx["rot_alt32"].astype(np.float64) - fx["rot"].astype(np.float64)).reshape(b, -1).max(1)
        a6 = np.abs(fx["pred_rot__alt32"].astype(np.float64) - fx["pred_rot_"].astype(np.float64)).reshape(b, -1).max(1)
        report.append(f"rot: TWO fp32 RUNS OF THE REFERENCE'S OWN CODE (oneDNN vs native kernels) are {alt32.max():.2e} apart at ROI {int(alt32.argmax())} "
                      f"({int((alt32 > 1e-4).sum())} ROIs beyond 1e-4; their 6-D outputs differ by <= {a6.max():.2e}); the second run is {alt64.max():.2e} from fp64 at ROI "
                      f"{int(alt64.argmax())} and violates the per-ROI clause 'distance from fp64 <= first run's + 2e-5' at {int((alt64 > ref_err + 2e-5).sum())} ROIs; "
                      f"on the well-conditioned ROIs the two runs are <= {alt32[well].max():.2e} apart")
    for i in ill:
        report.append(f"rot: ill-conditioned ROI {int(i)}: 4-sigma reference noise x conditioning = {cond[i]:.1e}; {tag} {ours64[i]:.2e} from fp64, "
                      f"{ours32[i]:.2e} from the reference's fp32; the reference's fp32 forward {ref_err[i]:.2e} from its own fp64 value"
                      + (" (> 1e-4)" if ref_err[i] > 1e-4 else "")
                      + (f"; the reference's second fp32 run {alt64[i]:.2e} from fp64, {alt32[i]:.2e} from the first" if alt64 is not None else ""))
    return report


@pytest.mark.parametrize("ds,b", [("tless", 1024), ("ycbv", 512)])
def test_default_path_matches_reference_forward_at_iteration_size(hip, ds, b):
    """Round-5 verdict item 1: the distance between this library's three- and six-product forms grows with the number of ROIs one
    looks at (7.2e-5 over 2 048 ROIs), so the 128-ROI fixtures do not speak for a 1 024-ROI iteration.  Fixture:
    net_golden_<ds>_b<b>.npz (tests/golden/make_golden_net.py record_large) — the reference's own GDRN_DoubleMask on a new seeded
    batch of b ROIs, its fp32 forward AND the same module in fp64, R / t / Patch-PnP outputs of every ROI.  ONE step of b ROIs on
    one GPU, default path; the bars are those of ``check_iteration_size_outputs``.  What the fixtures show (profiles/r06_large_parity.md):
    at this sample size EVERY fp32 forward — the reference's own (1.2e-4 at T-LESS ROI 30), this library's exact six-product form,
    the default three-product form — has a worst ROI around 1e-4 in R, each at a different ROI, all of them ill-conditioned ones;
    in the network's outputs the default path is on average closer to the fp64 value than the reference's fp32 forward."""
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers

    assert hip_layers.gemm_products() == 3 and hip_layers.mlp_gemm() == "split"
    fx = NG.load_fixture(f"{ds}_b{b}")
    assert fx["rot"].shape == (b, 3, 3) and len(set(fx["roi_cls"].tolist())) == fx["cfg"]["MODEL"]["POSE_NET"]["NUM_CLASSES"]
    cfg = get_cfg(NG.cfg_name(ds), opts=["TEST.USE_DEPTH_REFINE=True"])
    model, _ = build_model_optimizer(cfg)
    model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)
    x = torch.from_numpy(NG.net_image(b, int(fx["image_seed"]))).cuda()
    kw = NG.forward_kwargs(fx, "cuda")
    with torch.no_grad():
        out = model(x, **kw)                                  # ONE step of b ROIs
        rot_, t_, _ = model.forward_maps(x, kw["roi_classes"], kw["roi_coord_2d"], None, kw["roi_extents"])
    words = hip.split2_range_words()
    assert words == {}, f"three-product launches left their range: {words}"
    got = {"rot": out["rot"].float().cpu().numpy(), "trans": out["trans"].float().cpu().numpy(),
           "pred_rot_": rot_.cpu().numpy(), "pred_t_": t_.cpu().numpy()}
    del out, x
    torch.cuda.empty_cache()
    report = check_iteration_size_outputs(fx, got, b, "default path")
    text = "\n".join(f"[{ds} b={b}] " + r for r in report)
    print("\n" + text)
    try:                                               # kept next to the run (gpurun_out/ travels back from the GPU box)
        import os
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"large_parity_{ds}_b{b}.txt"), "w") as f:
            f.write(text + "\n")
    except OSError:
        pass
