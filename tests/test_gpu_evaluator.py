"""-m gpu: the evaluator hook end to end.  ``GDRN_Evaluator.process`` (HIP post-processing, one batched pass) must emit the
records the reference's own ``GDRN_Evaluator.process / process_depth_refine`` emitted for the same two images
(tests/golden/make_golden_eval.py): R and score identical, t within 1e-4 m (north_star) — measured ~1e-6."""
import numpy as np
import pytest
import torch

from gdrnpp_bop2022_amd import hip_lib
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
from gdrnpp_bop2022_amd.gdrn_modeling.gdrn_evaluator import GDRN_Evaluator, gdrn_inference_on_dataset
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer
from tests import evalgolden as EG

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _meshes(e):
    m = e["maps"]
    return hip_lib.MeshSet([m["verts"][i] for i in range(3)], [m["faces"][i] for i in range(3)], DEV)


@pytest.mark.parametrize("branch", ["direct", "refine"])
@pytest.mark.parametrize("images_per_call", [1, 2])
def test_process_emits_the_reference_records(hip, tmp_path, branch, images_per_call):
    e = EG.load()
    cfg = get_cfg("ycbv_convnext_a6", opts=[f"TEST.USE_DEPTH_REFINE={branch == 'refine'}", "INPUT.WITH_DEPTH=True"])
    cfg.EXP_ID = e["exp_id"]
    ev = GDRN_Evaluator(cfg, "ycbv_test", False, str(tmp_path), obj_names=e["names"], obj2id=e["obj2id"],
                        meshes=_meshes(e) if branch == "refine" else None)
    ev.reset()
    inputs, od = EG.image_inputs(e, DEV), EG.out_dict(e, DEV)
    if images_per_call == 2:                         # both images in one call (the reference's zoom_K index bug, fixed here)
        ev.process(inputs, [dict(time=float(t)) for t in e["fwd_time"]], od)
    else:
        for k, (lo, hi) in enumerate(e["split"]):
            ev.process([inputs[k]], [dict(time=float(e["fwd_time"][k]))], {key: v[lo:hi] for key, v in od.items()})
    ref = e[f"{branch}_predictions"]
    assert len(ev._predictions) == len(ref) == 5
    for p, r in zip(ev._predictions, ref):
        assert (p["scene_id"], p["im_id"], p["obj_id"]) == (r["scene_id"], r["im_id"], r["obj_id"])
        assert p["score"] == r["score"] and p["R"] == r["R"]            # passed through untouched
        assert np.abs(np.array(p["t"]) - np.array(r["t"])).max() <= 1e-4 * 1000.0     # mm
        assert p["time"] >= [t for t, (lo, hi) in zip(e["fwd_time"], e["split"])][0]
    if branch == "direct":
        assert all(p["t"] == r["t"] for p, r in zip(ev._predictions, ref))
    assert ev.evaluate() == {}
    rows = open(tmp_path / e[f"{branch}_csv_name"]).read().strip().split("\n")
    assert rows[0] == "scene_id,im_id,obj_id,score,R,t,time" and len(rows) == 6


@pytest.mark.parametrize("pnp_type", ["ransac_pnp", "net_iter_pnp", "net_ransac_pnp", "net_ransac_pnp_rot"])
def test_process_emits_the_reference_records_of_the_pnp_branches(hip, tmp_path, pnp_type):
    """TEST.USE_PNP: the records of ``GDRN_Evaluator.process`` against eval_pnp_golden.npz — the reference's own
    process_pnp_ransac / process_net_and_pnp executed on the same two images (correspondence selection, the -100 sentinel / network
    pose below 4 points, the 1 m translation guard, "ransac_rot" running the ITERATIVE solver and keeping the network translation,
    pnp_v2's plumbing are the reference's code; OpenCV's solvers inside are the oracle's restatements).  R within 1e-4, t within
    1e-4 m; the ROI with a one-pixel mask takes the fall-back of its branch exactly."""
    e = EG.load_pnp()
    cfg = get_cfg("ycbv_convnext_a6", opts=["TEST.USE_PNP=True", f"TEST.PNP_TYPE={pnp_type}"])
    cfg.EXP_ID = e["exp_id"]
    ev = GDRN_Evaluator(cfg, "ycbv_test", False, str(tmp_path), obj_names=e["names"], obj2id=e["obj2id"])
    ev.reset()
    od = EG.out_dict(e, DEV)
    od["mask"] = torch.from_numpy(e["pnp_mask"]).to(DEV)
    ev.process(EG.image_inputs_pnp(e, DEV), [dict(time=float(t)) for t in e["fwd_time"]], od)
    ref = e[f"pnp_{pnp_type}_predictions"]
    assert len(ev._predictions) == len(ref) == 5
    for i, (p, r) in enumerate(zip(ev._predictions, ref)):
        assert (p["scene_id"], p["im_id"], p["obj_id"], p["score"]) == (r["scene_id"], r["im_id"], r["obj_id"], r["score"])
        dR, dt = np.abs(np.array(p["R"]) - np.array(r["R"])).max(), np.abs(np.array(p["t"]) - np.array(r["t"])).max()
        if i == 3:                                     # fewer than 4 correspondences: sentinel / network pose, exactly
            assert dR == 0 and dt == 0, (i, dR, dt)
            assert (np.array(p["t"]) == -100000.0).all() if pnp_type == "ransac_pnp" else p["R"] == np.asarray(e["R"][3], np.float32).reshape(-1).tolist()
        else:
            assert dR <= 1e-4 and dt <= 1e-4 * 1000.0, (i, dR, dt)
    if pnp_type.startswith("net_ransac_pnp_rot"):      # the network translation is what comes out
        assert all(np.allclose(np.array(p["t"]) / 1000.0, e["maps"]["t_init"][i], atol=1e-6) for i, p in enumerate(ev._predictions))


def test_train_objs_subset_skips_untrained_classes(hip, tmp_path):
    e = EG.load()
    cfg = get_cfg("ycbv_convnext_a6")
    keep = [n for n in e["names"] if n != e["names"][int(e["roi_cls"][0])]]
    ev = GDRN_Evaluator(cfg, "ycbv_test", False, None, train_objs=keep, obj_names=e["names"], obj2id=e["obj2id"])
    ev.reset()
    ev.process(EG.image_inputs(e, DEV), [dict(time=0.0), dict(time=0.0)], EG.out_dict(e, DEV))
    dropped = int((e["roi_cls"] == e["roi_cls"][0]).sum())
    assert len(ev._predictions) == 5 - dropped
    assert e["obj2id"][e["names"][int(e["roi_cls"][0])]] not in [p["obj_id"] for p in ev._predictions]


def test_inference_loop_protocol(hip, tmp_path):
    """gdrn_inference_on_dataset: warm-up rule, one record per ROI, evaluate() writes the csv."""
    from gdrnpp_bop2022_amd import synthetic as S

    e = EG.load()
    cfg = get_cfg("ycbv_convnext_a6", opts=["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"])
    names = [f"obj_{i:02d}" for i in range(21)]
    rng = np.random.default_rng(3)
    verts, faces, ext = S.make_models(21, rng, 2)
    ev = GDRN_Evaluator(cfg, "ycbv_test", False, str(tmp_path), obj_names=names, obj2id={n: i + 1 for i, n in enumerate(names)},
                        meshes=hip_lib.MeshSet(verts, faces, DEV))
    torch.manual_seed(0)
    model, _ = build_model_optimizer(cfg)
    loader = []
    for im in range(4):
        n = 2 + im % 2
        det = S.make_detections(n, 21, ext, rng)
        T = torch.from_numpy
        loader.append([dict(
            roi_img=torch.rand(n, 3, 256, 256), roi_cls=T(det["roi_cls"]), cam=T(det["roi_cam"]), roi_wh=T(det["roi_wh"]),
            bbox_center=T(det["roi_center"]), resize_ratio=T(det["resize_ratio"]), scale=T(det["scale"]), score=T(det["score"]),
            roi_coord_2d=T(S.coord2d_roi(det["roi_center"], det["scale"])), roi_extent=T(det["roi_extent"]),
            roi_depth=torch.rand(n, 1, 256, 256) + 0.5, scene_im_id=[f"50/{im}"] * n, time=torch.full((n,), 0.01))])
    assert gdrn_inference_on_dataset(cfg, model, loader, ev) == {}
    st = gdrn_inference_on_dataset.last_stats
    assert st["warmup_iters"] == 3 and st["iters"] == 4 and st["rois"] == len(loader[3][0]["roi_cls"])
    assert len(ev._predictions) == sum(len(b[0]["roi_cls"]) for b in loader)
    assert all(np.isfinite(p["t"]).all() and p["time"] > 0.01 for p in ev._predictions)
    assert len(open(tmp_path / "ycbv-convnext-a6-iter0_ycbv-test.csv").read().strip().split("\n")) == 1 + len(ev._predictions)
    # TEST.AMP_TEST (gdrn_evaluator.py:736-747): the forward under torch.autocast(fp16) as the plain module graph, fp32
    # post-processing — same records to mixed-precision accuracy, and the fp32 HIP path is back on afterwards
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
    fp32 = [(p["R"], p["t"]) for p in ev._predictions]
    assert gdrn_inference_on_dataset(cfg, model, loader, ev, amp_test=True) == {}
    assert hip_layers.is_enabled() and len(ev._predictions) == len(fp32)
    for p, (R, t) in zip(ev._predictions, fp32):
        assert np.abs(np.array(p["R"]) - np.array(R)).max() < 5e-2
        assert np.isfinite(p["t"]).all()


def test_inference_loop_with_roi_packing_keeps_records_and_order(hip, tmp_path):
    """gdrn_inference_on_dataset(pack_rois=N): consecutive one-image loader items pooled into steps of >= N ROIs — same records in
    the same order as the reference's one-image-per-forward schedule (R / t within the path's tolerance: other kernels serve the
    larger step), every image of a step charged the step's time, csv written as before."""
    from gdrnpp_bop2022_amd import synthetic as S

    cfg = get_cfg("ycbv_convnext_a6", opts=["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"])
    names = [f"obj_{i:02d}" for i in range(21)]
    rng = np.random.default_rng(4)
    verts, faces, ext = S.make_models(21, rng, 2)
    ev = GDRN_Evaluator(cfg, "ycbv_test", False, str(tmp_path), obj_names=names, obj2id={n: i + 1 for i, n in enumerate(names)},
                        meshes=hip_lib.MeshSet(verts, faces, DEV))
    torch.manual_seed(0)
    model, _ = build_model_optimizer(cfg)
    model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 3), strict=True)
    with torch.no_grad():
        model.pnp_net.fc_t.bias.copy_(torch.tensor([0.0, 0.0, 1.5 * float(S.YCBV_K[0, 0]) * 0.19 / 64.0]))
    loader = []
    for im, n in enumerate([3, 7, 1, 9, 4, 6, 2, 8, 5, 3, 4, 6]):
        det = S.make_detections(n, 21, ext, rng)
        T = torch.from_numpy
        loader.append([dict(
            roi_img=torch.rand(n, 3, 256, 256), roi_cls=T(det["roi_cls"]), cam=T(det["roi_cam"]), roi_wh=T(det["roi_wh"]),
            bbox_center=T(det["roi_center"]), resize_ratio=T(det["resize_ratio"]), scale=T(det["scale"]), score=T(det["score"]),
            roi_coord_2d=T(S.coord2d_roi(det["roi_center"], det["scale"])), roi_extent=T(det["roi_extent"]),
            roi_depth=torch.rand(n, 1, 256, 256) + 0.5, scene_im_id=[f"50/{im}"] * n, time=torch.full((n,), 0.01))])
    assert gdrn_inference_on_dataset(cfg, model, loader, ev) == {}
    plain = [dict(p) for p in ev._predictions]
    assert gdrn_inference_on_dataset(cfg, model, loader, ev, pack_rois=16) == {}
    st = gdrn_inference_on_dataset.last_stats
    assert st["iters"] == 3 and st["warmup_iters"] == 2                      # 58 ROIs in steps of >= 16: 20 + 20 + 18
    packed = ev._predictions
    assert len(packed) == len(plain) == 58
    for p, q in zip(packed, plain):
        assert (p["scene_id"], p["im_id"], p["obj_id"], p["score"]) == (q["scene_id"], q["im_id"], q["obj_id"], q["score"])
        assert np.abs(np.array(p["R"]) - np.array(q["R"])).max() <= 1e-4
        assert np.abs(np.array(p["t"]) - np.array(q["t"])).max() <= 1e-4 * 1000.0          # mm
    by_im = {}
    for p in packed:
        by_im.setdefault(p["im_id"], set()).add(p["time"])
    assert all(len(v) == 1 for v in by_im.values())
    assert len(open(tmp_path / "ycbv-convnext-a6-iter0_ycbv-test.csv").read().strip().split("\n")) == 59
