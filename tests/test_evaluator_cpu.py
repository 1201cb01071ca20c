"""Evaluator hook, host side (no GPU): results-file name and csv text, per-image time rule, batch_data_test — against the
records the reference's own GDRN_Evaluator / save_and_eval_results produced (tests/golden/make_golden_eval.py)."""
import numpy as np
import torch

from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
from gdrnpp_bop2022_amd.gdrn_modeling.gdrn_evaluator import GDRN_Evaluator, batch_data_test, bop_csv_name
from tests import evalgolden as EG


def _evaluator(e, tmp_path, opts=()):
    cfg = get_cfg("ycbv_convnext_a6", opts=list(opts))
    cfg.EXP_ID = e["exp_id"]
    return cfg, GDRN_Evaluator(cfg, "ycbv_test", False, str(tmp_path), obj_names=e["names"], obj2id=e["obj2id"])


def _parse(csv_text):
    rows = [l.split(",") for l in csv_text.strip().split("\n")]
    head, rows = rows[0], rows[1:]
    return head, [(r[0], int(r[1]), int(r[2]), float(r[3]), np.array(r[4].split(), float), np.array(r[5].split(), float),
                   float(r[6])) for r in rows]


def test_csv_name_and_text_like_save_and_eval_results(tmp_path):
    e = EG.load()
    cfg, ev = _evaluator(e, tmp_path)
    assert bop_csv_name(cfg) == e["direct_csv_name"] == e["refine_csv_name"]
    for branch in ("direct", "refine"):
        ev.reset()
        ev._predictions = [dict(p) for p in e[f"{branch}_predictions"]]      # the reference's records in, its csv out
        assert ev.evaluate() == {}
        text = open(tmp_path / e[f"{branch}_csv_name"]).read()
        assert text == e[f"{branch}_csv"]


def test_time_rule_largest_per_image(tmp_path):
    e = EG.load()
    _, ev = _evaluator(e, tmp_path)
    ev._predictions = [dict(scene_id="1", im_id=3, time=0.2), dict(scene_id="1", im_id=3, time=0.5), dict(scene_id="1", im_id=4, time=0.1)]
    ev._process_time_of_preds(ev._predictions)
    assert [p["time"] for p in ev._predictions] == [0.5, 0.5, 0.1]
    # the reference's csv carries one time per image, too
    _, rows = _parse(e["refine_csv"])
    by_im = {}
    for r in rows:
        by_im.setdefault(r[1], set()).add(r[6])
    assert all(len(v) == 1 for v in by_im.values())


def test_batch_data_test_concatenates_per_image_dicts():
    e = EG.load()
    cfg = get_cfg("ycbv_convnext_a6", opts=["INPUT.WITH_DEPTH=True"])
    inputs = EG.image_inputs(e)
    for d in inputs:
        n = len(d["roi_cls"])
        d["roi_wh"] = torch.ones(n, 2)
        d["roi_extent"] = torch.ones(n, 3)
    b = batch_data_test(cfg, inputs, device="cpu")
    assert b["roi_cls"].dtype == torch.long and b["roi_cls"].tolist() == e["roi_cls"].tolist()
    assert b["roi_cam"].shape == (5, 3, 3) and b["roi_center"].shape == (5, 2) and b["roi_depth"].shape == (5, 1, 256, 256)
    assert b["scene_im_id"] == ["48/7"] * 3 + ["48/8"] * 2
    assert torch.equal(b["scale"], torch.from_numpy(e["scale"]))
