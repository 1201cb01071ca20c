"""Golden records of the EVALUATOR hook from the reference's own class (authoring container only: needs /root/reference).

``core/gdrn_modeling/engine/gdrn_evaluator.py`` is imported from its file (tests/golden/_refimport.py) and
``GDRN_Evaluator.reset / process / process_depth_refine / evaluate / _process_time_of_preds / pose_prediction_to_json`` and
``save_and_eval_results`` (engine/test_utils.py, ``VAL.SAVE_BOP_CSV_ONLY``) run unmodified on two synthetic "images"
(3 + 2 ROIs, three object classes — the same case as make_golden_pyref.py's ``rf_*`` arrays, which hold the maps).
Stand-ins, as in make_golden_pyref.py: the constructor (dataset registry, .ply loading, vispy) is bypassed with
``__new__`` + the attributes it would set; the GL renderer and ``cv2.resize`` are served by the oracle's rasteriser and
its restated INTER_LINEAR x4.  Recorded in eval_golden.npz: the per-ROI scalars of the per-image input dicts, the
``_predictions`` of the direct branch (TEST.USE_DEPTH_REFINE=False) and of the refine branch, and the BOP csv text.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import _refimport  # noqa: E402

_refimport.install()

from gdrnpp_bop2022_amd import synthetic as S  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.config import Config  # noqa: E402
from oracle import postproc as P  # noqa: E402

NAMES = ["obj_a", "obj_b", "obj_c"]
OBJ2ID = {n: i + 1 for i, n in enumerate(NAMES)}
SPLIT = [(0, 3), (3, 5)]
FWD_TIME = [0.25, 0.5]


def synthetic_case():
    """Identical to make_golden_pyref.refine_case (seed 20220925 + 8)."""
    rng = np.random.default_rng(20220925 + 8)
    verts, faces, ext = S.make_models(3, rng, 3)
    det = S.make_detections(5, 3, ext, rng)

    def render_fn(obj, K, R, t, res):
        d, x = zip(*[P.render_depth(verts[obj[i]], faces[obj[i]], K[i], R[i], t[i].astype(np.float64), res_w=res, want_xyz=True)
                     for i in range(len(obj))])
        return np.stack(d), np.stack(x)

    maps = S.make_map_inputs(det, verts, faces, render_fn, rng)
    return verts, faces, det, maps


def main():
    import core.gdrn_modeling.engine.gdrn_evaluator as GE
    import core.gdrn_modeling.engine.test_utils as TU
    import mmcv

    verts, faces, det, maps = synthetic_case()
    g = np.load(os.path.join(HERE, "pyref_golden.npz"))
    for k in ("coor_x", "coor_y", "coor_z", "mask", "roi_depth", "t_init"):
        assert np.array_equal(maps[k], g["rf_" + k]), k          # the maps live in pyref_golden.npz

    class Ren:                                   # stand-in for lib/render_vispy Renderer: GL gets float32 uniforms
        def clear(self): pass
        def set_cam(self, K): self.K = np.asarray(K, np.float32)
        def draw_model(self, model, pose): self.model, self.pose = model, np.asarray(pose, np.float32)
        def finish(self):
            v, f = self.model
            return None, P.render_depth(v, f, self.K, self.pose[:, :3], self.pose[:, 3].astype(np.float64), res_w=64)

    raw = _refimport.load_ref_config("configs/gdrn/ycbv/convnext_a6_AugCosyAAEGray_BG05_mlL1_DMask_amodalClipBox_classAware_ycbv.py")
    cfg = Config(raw)
    cfg.EXP_ID = "convnext_a6_ycbv_test"         # core/utils/default_args_setup.py derives it from the config file name
    cfg.VAL.SAVE_BOP_CSV_ONLY = True
    cfg.VAL.SPLIT = "test"
    GE.cv2 = types.SimpleNamespace(resize=lambda a, size: P.resize_depth_x4_linear(a))
    orig_bdi = GE.batch_data_inference_roi
    GE.batch_data_inference_roi = lambda c, data: orig_bdi(c, data, device="cpu")
    mmcv.mkdir_or_exist = lambda d: os.makedirs(d, exist_ok=True)

    T = torch.from_numpy
    out_all = dict(coor_x=T(maps["coor_x"]), coor_y=T(maps["coor_y"]), coor_z=T(maps["coor_z"]), mask=T(maps["mask"]),
                   rot=T(det["R_gt"]), trans=T(maps["t_init"]))
    rec = {}
    for branch in ("direct", "refine"):
        cfg.TEST.USE_DEPTH_REFINE = branch == "refine"
        cfg.TEST.USE_PNP = False
        ev = GE.GDRN_Evaluator.__new__(GE.GDRN_Evaluator)
        tmp = tempfile.mkdtemp()
        ev.cfg, ev._distributed, ev._output_dir, ev._cpu_device = cfg, False, tmp, torch.device("cpu")
        ev.train_objs, ev.obj_names, ev.obj_ids = None, NAMES, [OBJ2ID[n] for n in NAMES]
        ev.data_ref = types.SimpleNamespace(obj2id=OBJ2ID, objects=NAMES)
        ev.ren, ev.ren_models = Ren(), [(verts[i], faces[i]) for i in range(3)]
        ev.out_res, ev.depth_refine_threshold = 64, cfg.TEST.DEPTH_REFINE_THRESHOLD
        ev.reset()
        # one image per call: the reference indexes zoom_K with the per-image index (gdrn_evaluator.py:493)
        for k, (lo, hi) in enumerate(SPLIT):
            inp = dict(roi_img=torch.zeros(hi - lo, 1), cam=T(det["roi_cam"][lo:hi]), roi_cls=T(det["roi_cls"][lo:hi]),
                       score=T(det["score"][lo:hi]), scene_im_id=[f"48/{k + 7}"] * (hi - lo), roi_depth=T(maps["roi_depth"][lo:hi]),
                       bbox_center=T(det["roi_center"][lo:hi]), scale=T(det["scale"][lo:hi]),
                       resize_ratio=T(det["resize_ratio"][lo:hi]))
            ev.process([inp], [dict(time=FWD_TIME[k])], {key: v[lo:hi] for key, v in out_all.items()})
        preds = [dict(p, score=float(p["score"])) for p in ev._predictions]
        assert len(preds) == 5
        assert ev.evaluate() == {}
        files = os.listdir(tmp)
        assert len(files) == 1
        rec[f"{branch}_predictions"] = json.dumps(preds)
        rec[f"{branch}_csv_name"] = files[0]
        rec[f"{branch}_csv"] = open(os.path.join(tmp, files[0])).read()
        print(branch, files[0], preds[0])
    t_ref = np.array([p["t"] for p in json.loads(rec["refine_predictions"])]) / 1000.0
    assert np.abs(t_ref - g["rf_t_refined"]).max() < 1e-6          # same numbers as the ast-cut recording
    np.savez_compressed(
        os.path.join(HERE, "eval_golden.npz"), names=json.dumps(NAMES), obj2id=json.dumps(OBJ2ID), split=np.array(SPLIT),
        fwd_time=np.array(FWD_TIME), roi_cam=det["roi_cam"], roi_center=det["roi_center"], scale=det["scale"],
        score=det["score"], resize_ratio=det["resize_ratio"], roi_cls=det["roi_cls"], R=det["R_gt"],
        exp_id=cfg.EXP_ID, val=json.dumps({k: cfg.VAL[k] for k in ("DATASET_NAME", "SPLIT", "SPLIT_TYPE")}), **rec)
    print("wrote eval_golden.npz")


if __name__ == "__main__":
    main()
