"""core/utils/zoom_utils.py:80-96 and core/utils/data_utils.py:65-112 with detectron2's ROIAlign replaced by the
HIP kernel (``gdrnpp_roi_align``) and torchvision's RoIPool (``interpolation="nearest"``) by ``gdrnpp_roi_pool``."""
import numpy as np
import torch

from ... import hip_lib


def batch_crop_resize(x, rois, out_H, out_W, aligned=True, interpolation="bilinear"):
    """x: BCHW (device), rois: Bx5 with rois[:, 0] the index into x."""
    if interpolation == "bilinear":
        return hip_lib.roi_align(x.contiguous(), rois.contiguous().float(), (out_H, out_W), 1.0, 0, aligned)
    if interpolation == "nearest":
        return hip_lib.roi_pool(x.contiguous(), rois.contiguous().float(), (out_H, out_W), 1.0)
    raise ValueError(f"Wrong interpolation type: {interpolation}")


_TO_CHW = {"HW": lambda a: a[None], "HWC": lambda a: np.moveaxis(a, 2, 0), "CHW": lambda a: a}


def crop_resize_by_d2_roialign(img, center, scale, output_size, aligned=True, interpolation="bilinear",
                               in_format="HWC", out_format="HWC", dtype="float32", device="cuda"):
    """Single-image convenience form of ``batch_crop_resize`` with the call surface of the reference helper
    (core/utils/data_utils.py:65-112): a NumPy image in ``in_format``, a box given by its centre and (w, h) extent (a scalar
    means a square), ``output_size`` an int or (w, h) -> the ROIAlign crop as a NumPy array in ``out_format``."""
    out_w, out_h = (output_size, output_size) if np.isscalar(output_size) else output_size
    box_w, box_h = (scale, scale) if np.isscalar(scale) else scale
    if in_format not in _TO_CHW:
        raise ValueError(f"in_format {in_format!r}: expected one of {sorted(_TO_CHW)}")
    chw = np.ascontiguousarray(_TO_CHW[in_format](np.asarray(img)), np.float32)
    half = 0.5 * np.array([box_w, box_h], np.float32)
    c = np.asarray(center, np.float32)
    roi = np.concatenate([[0.0], c - half, c + half]).astype(np.float32)[None]           # (image index, x1, y1, x2, y2)
    crop = batch_crop_resize(torch.from_numpy(chw[None]).to(device), torch.from_numpy(roi).to(device), int(out_h), int(out_w),
                             aligned, interpolation)[0]
    out = crop.cpu().numpy().astype(dtype)
    return np.moveaxis(out, 0, 2) if out_format == "HWC" else out
