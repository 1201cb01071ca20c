"""Golden vectors of the YOLOX detection post-processing (authoring container only: needs /root/reference).

``postprocess`` (det/yolox/utils/boxes.py:34-74) is executed UNMODIFIED from its source text.  The one third-party call inside it,
``torchvision.ops.nms`` / ``batched_nms`` (torchvision is not installed), is served by the stand-in below, written from torchvision's
published semantics: candidates by descending score (ties by ascending index — torch.sort leaves them unspecified), box j suppressed
when inter / (area_i + area_j - inter) > thr, ``batched_nms`` = the coordinate trick (boxes + class * (max_coordinate + 1)).  So the
fixture pins everything AROUND the NMS primitive to the reference's own code — corner conversion, argmax class, the
obj * class >= conf_thre mask, the (x1, y1, x2, y2, obj, class_conf, class) layout, keep order, the None for an empty image, the
class_agnostic switch — and the primitive itself stays a restatement (oracle/nms_oracle.c says so).  -> yolox_golden.npz"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_golden_pyref import cut  # noqa: E402


def nms(boxes, scores, iou_threshold):
    n = boxes.shape[0]
    order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
    b = boxes.numpy().astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup, keep = np.zeros(n, bool), []
    for a_, i in enumerate(order):
        if sup[i]:
            continue
        keep.append(i)
        for j in order[a_ + 1:]:
            if sup[j]:
                continue
            w = np.float32(max(np.float32(0), min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0])))
            h = np.float32(max(np.float32(0), min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1])))
            inter = np.float32(w * h)
            if inter / np.float32(np.float32(area[i] + area[j]) - inter) > np.float32(iou_threshold):
                sup[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def make_preds(rng, n_img, a, c, clustered):
    """YOLOX head rows (cx, cy, w, h, obj, class scores): boxes in clusters (so that NMS has work), scores with ties."""
    det = np.zeros((n_img, a, 5 + c), np.float32)
    for i in range(n_img):
        centres = rng.uniform(60, 580, (clustered, 2))
        k = rng.integers(0, clustered, a)
        det[i, :, 0:2] = centres[k] + rng.normal(0, 6, (a, 2))
        det[i, :, 2:4] = rng.uniform(30, 120, (a, 2))
        det[i, :, 4] = np.round(rng.uniform(0, 1, a), 2)                   # two-decimal scores: plenty of exact ties
        det[i, :, 5:] = np.round(rng.uniform(0, 1, (a, c)), 2)
    return det


def main():
    ns = {"torch": torch, "torchvision": types.SimpleNamespace(ops=types.SimpleNamespace(nms=nms, batched_nms=batched_nms))}
    exec(compile(cut("det/yolox/utils/boxes.py", "postprocess"), "/root/reference/det/yolox/utils/boxes.py", "exec"), ns)
    rng = np.random.default_rng(20220925 + 41)
    rec = {}
    cases = [("a", make_preds(rng, 3, 200, 21, 9), 21, 0.3, 0.45, False), ("b", make_preds(rng, 2, 150, 21, 5), 21, 0.3, 0.45, True),
             ("c", make_preds(rng, 2, 120, 1, 6), 1, 0.5, 0.65, False), ("d", make_preds(rng, 2, 64, 30, 4), 30, 0.995, 0.45, False)]
    for name, det, c, conf, thr, agn in cases:
        out = ns["postprocess"](torch.from_numpy(det.copy()), c, conf, thr, class_agnostic=agn)    # the reference writes into its input
        rec[f"{name}_det"] = det
        rec[f"{name}_args"] = np.array([c, conf, thr, float(agn)], np.float64)
        rec[f"{name}_count"] = np.array([0 if o is None else o.shape[0] for o in out], np.int64)
        rec[f"{name}_out"] = np.concatenate([np.zeros((0, 7), np.float32)] + [o.numpy() for o in out if o is not None])
        print(name, rec[f"{name}_count"].tolist())
    np.savez_compressed(os.path.join(HERE, "yolox_golden.npz"), **rec)
    print("wrote yolox_golden.npz")


if __name__ == "__main__":
    main()
