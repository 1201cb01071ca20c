"""core/utils/zoom_utils.py:80-96 and core/utils/data_utils.py:65-112 with detectron2's ROIAlign replaced by the
HIP kernel (``gdrnpp_roi_align``).  ``interpolation="nearest"`` (torchvision RoIPool) is not carried."""
import numpy as np
import torch

from ... import hip_lib


def batch_crop_resize(x, rois, out_H, out_W, aligned=True, interpolation="bilinear"):
    """x: BCHW (device), rois: Bx5 with rois[:, 0] the index into x."""
    if interpolation != "bilinear":
        raise NotImplementedError("only the bilinear (ROIAlign) flavour is provided")
    return hip_lib.roi_align(x.contiguous(), rois.contiguous().float(), (out_H, out_W), 1.0, 0, aligned)


def crop_resize_by_d2_roialign(img, center, scale, output_size, aligned=True, interpolation="bilinear",
                               in_format="HWC", out_format="HWC", dtype="float32", device="cuda"):
    """img (np.ndarray) HWC/HW/CHW -> cropped + resized array (same conventions as the reference helper)."""
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    output_size = (output_size[1], output_size[0])  # to (h, w)
    assert in_format in ["HW", "HWC", "CHW"]
    if in_format == "HW":
        img = img[None]
    elif in_format == "HWC":
        img = img.transpose(2, 0, 1)
    img_tensor = torch.as_tensor(np.ascontiguousarray(img[None]).astype("float32")).to(device)
    cx, cy = center
    if isinstance(scale, (int, float)):
        scale = (scale, scale)
    bw, bh = scale
    rois = torch.as_tensor(np.array([0] + [cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], dtype="float32")[None])
    result = batch_crop_resize(img_tensor, rois.to(device), output_size[0], output_size[1], aligned, interpolation)
    result = result[0].cpu().numpy().astype(dtype)
    if out_format == "HWC":
        result = result.transpose(1, 2, 0)
    return result
