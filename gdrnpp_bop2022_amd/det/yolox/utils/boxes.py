"""det/yolox/utils/boxes.py:34-74: ``postprocess(det_preds, num_classes, conf_thre, nms_thre, class_agnostic)``.

Same contract as the reference (list with one ``[n_i, 7]`` tensor per image — x1, y1, x2, y2, obj_conf, class_conf,
class_pred — or ``None`` when nothing survives), served by ``gdrnpp_yolox_postprocess`` for the whole batch in three
launches; the reference loops over images in Python and calls torchvision (not installed on the target).  Unlike the
reference the input tensor is left untouched (it rewrites det_preds[..., :4] to corners in place)."""
from .... import hip_lib


def postprocess(det_preds, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    if not det_preds.is_cuda:
        raise RuntimeError("postprocess: CPU tensors are not supported by this build (no CPU fallback)")
    dets, count = hip_lib.yolox_postprocess(det_preds.contiguous().float(), num_classes, conf_thre, nms_thre, class_agnostic)
    counts = count.tolist()   # one small D2H copy for the whole batch: the list lengths are data dependent
    return [dets[i, :n] if n > 0 else None for i, n in enumerate(counts)]
