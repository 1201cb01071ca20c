"""Shared by the evaluator tests: rebuild the two per-image input dicts and the out_dict of tests/golden/eval_golden.npz
(recorded by running the reference's own GDRN_Evaluator, tests/golden/make_golden_eval.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    z = np.load(os.path.join(GOLDEN, "eval_golden.npz"))
    e = {k: z[k] for k in z.files}
    for k in ("names", "obj2id", "val", "direct_predictions", "refine_predictions"):
        e[k] = json.loads(str(e[k]))
    for k in ("exp_id", "direct_csv", "refine_csv", "direct_csv_name", "refine_csv_name"):
        e[k] = str(e[k])
    g = np.load(os.path.join(GOLDEN, "pyref_golden.npz"))
    e["maps"] = {k: g["rf_" + k] for k in ("coor_x", "coor_y", "coor_z", "mask", "roi_depth", "t_init", "verts", "faces")}
    return e


def image_inputs(e, device="cpu"):
    """The list of per-image dicts in the layout of read_data_test (data_loader.py:647-818)."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    out = []
    for k, (lo, hi) in enumerate(e["split"]):
        out.append(dict(roi_img=torch.zeros(hi - lo, 1), cam=T(e["roi_cam"][lo:hi]), roi_cls=T(e["roi_cls"][lo:hi]),
                        score=T(e["score"][lo:hi]), scene_im_id=[f"48/{k + 7}"] * int(hi - lo),
                        roi_depth=T(e["maps"]["roi_depth"][lo:hi]), bbox_center=T(e["roi_center"][lo:hi]),
                        scale=T(e["scale"][lo:hi]), resize_ratio=T(e["resize_ratio"][lo:hi])))
    return out


def out_dict(e, device):
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    m = e["maps"]
    return dict(coor_x=T(m["coor_x"]), coor_y=T(m["coor_y"]), coor_z=T(m["coor_z"]), mask=T(m["mask"]), rot=T(e["R"]),
                trans=T(m["t_init"]))


def load_pnp():
    """eval_pnp_golden.npz: the reference's GDRN_Evaluator.process with TEST.USE_PNP for the four PNP_TYPEs on the same two images
    (tests/golden/make_golden_eval_pnp.py); the mask of ROI 3 holds a single pixel (fewer than 4 correspondences)."""
    e = load()
    z = np.load(os.path.join(GOLDEN, "eval_pnp_golden.npz"))
    for k in z.files:
        e["pnp_" + k] = json.loads(str(z[k])) if k.endswith(("_predictions", "_calls")) else z[k]
    return e


def image_inputs_pnp(e, device="cpu"):
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    out = image_inputs(e, device)
    for k, (lo, hi) in enumerate(e["split"]):
        out[k].update(roi_coord_2d=T(e["pnp_roi_coord_2d"][lo:hi]), roi_extent=T(e["pnp_roi_extent"][lo:hi]), im_H=T(e["pnp_im_H"][lo:hi]),
                      im_W=T(e["pnp_im_W"][lo:hi]))
    return out
