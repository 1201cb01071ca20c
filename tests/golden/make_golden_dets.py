"""Golden selection of pre-computed detections (authoring container only: needs /root/reference).

``load_detections_into_dataset`` (core/utils/dataset_utils.py:146-227) is cut out of the reference file with ``ast`` and executed
UNMODIFIED; what it reaches outside itself is served by stand-ins: ``mmcv.load`` returns the detection dict, ``MetadataCatalog.get`` /
``ref`` a dataset with five objects (ids 1, 5, 9, 12, 20 in class order), ``BoxMode.XYWH_ABS`` detectron2's constant 1.  The seeded
detection file holds duplicates with EQUAL scores, objects that are not in the dataset, scores under the threshold, an image without an
entry and one whose entries are all filtered.  Recorded per case (top_k, score_thr, train_objs): for every kept image its key and the
(category_id, bbox_est, score, time) of its annotations in the reference's order.  -> dets_golden.npz"""
import copy
import json
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_pyref import cut  # noqa: E402

OBJ_IDS = [1, 5, 9, 12, 20]
NAMES = [f"obj_{i:02d}" for i in OBJ_IDS]


def make_detections(rng):
    dets, keys = {}, [f"{48 + i // 4}/{i}" for i in range(12)]
    for k in keys:
        if k.endswith("/5"):
            continue                                              # an image the detector returned nothing for
        rows = []
        for _ in range(int(rng.integers(2, 9))):
            oid = int(rng.choice(OBJ_IDS + [99, 3]))              # 99 / 3: not objects of this dataset
            x, y, w, h = [round(float(v), 1) for v in rng.uniform(5, 300, 4)]
            row = {"obj_id": oid, "bbox_est": [x, y, w, h], "score": round(float(rng.uniform(0, 1)), 1)}   # one-decimal scores: ties
            if rng.uniform() < 0.7:
                row["time"] = round(float(rng.uniform(0.01, 0.2)), 3)
            rows.append(row)
        dets[k] = rows
    dets[keys[7]] = [{"obj_id": 99, "bbox_est": [1, 2, 3, 4], "score": 0.9}, {"obj_id": 5, "bbox_est": [4, 3, 2, 1], "score": 0.01}]   # all filtered
    return keys, dets


def main():
    rng = np.random.default_rng(20220925 + 61)
    keys, dets = make_detections(rng)
    data_ref = types.SimpleNamespace(id2obj={i: n for i, n in zip(OBJ_IDS, NAMES)}, get_models_info=lambda: {str(i): {"diameter": float(i)} for i in OBJ_IDS + [3, 99]})
    data_ref.id2obj.update({99: "other_99", 3: "other_03"})
    ns = dict(copy=copy, logger=logging.getLogger("ref"), mmcv=types.SimpleNamespace(load=lambda f: dets),
              MetadataCatalog=types.SimpleNamespace(get=lambda name: types.SimpleNamespace(objs=NAMES, ref_key="syn")),
              ref=types.SimpleNamespace(syn=data_ref), BoxMode=types.SimpleNamespace(XYWH_ABS=1))
    exec(compile(cut("core/utils/dataset_utils.py", "load_detections_into_dataset"), "/root/reference/core/utils/dataset_utils.py", "exec"), ns)
    dataset_dicts = [dict(scene_im_id=k, file_name=f"{k}.png") for k in keys]
    cases = [dict(top_k_per_obj=1, score_thr=0.0, train_objs=None), dict(top_k_per_obj=2, score_thr=0.3, train_objs=None),
             dict(top_k_per_obj=3, score_thr=0.1, train_objs=[NAMES[1], NAMES[3]]), dict(top_k_per_obj=1, score_thr=0.5, train_objs=None)]
    out = []
    for c in cases:
        recs = ns["load_detections_into_dataset"]("syn_test", dataset_dicts, "dets.json", **c)
        out.append(dict(args=c, images=[dict(scene_im_id=r["scene_im_id"],
                                             annotations=[[a["category_id"], a["bbox_est"], a["score"], a["time"]] for a in r["annotations"]])
                                        for r in recs]))
        print(c, len(recs), sum(len(r["annotations"]) for r in recs))
    np.savez_compressed(os.path.join(HERE, "dets_golden.npz"), keys=json.dumps(keys), detections=json.dumps(dets), obj_ids=np.array(OBJ_IDS),
                        names=json.dumps(NAMES), cases=json.dumps(out))
    print("wrote dets_golden.npz")


if __name__ == "__main__":
    main()
