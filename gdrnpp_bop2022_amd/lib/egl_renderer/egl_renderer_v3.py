"""Import-compatible stand-in for the reference's ``lib/egl_renderer/egl_renderer_v3.EGLRenderer`` restricted to what
the GDRN pipeline reads from it: the object-space point map (``pc_obj_tensor``, the ``PCObject`` attachment) and the
camera-space point map (``pc_cam_tensor``; depth = ``[..., 2]``), engine_utils.py:131-172, egl_renderer_v3.py:838-1225.

The reference renders through EGL + ``EGL_CUDA_DEVICE_NV`` + CUDA-GL interop (cpp/egl_renderer.cpp:48-96,262-298), none
of which exists on ROCm.  Here one call = one ``gdrnpp_render_depth`` launch per object writing straight into device
memory; several objects are composited by nearest depth.  Pixel (row j, col i) is sampled at (i+0.5, j+0.5) under K,
rows already in image order (the reference flips the GL rows after mapping, :1207,1216)."""
from __future__ import annotations

import numpy as np
import torch

from ... import hip_lib
from ..pysixd.inout import load_ply
from . import CppEGLRenderer

# attachment numbering of the reference's framebuffer (egl_renderer_v3.py:205-262)
COLOR_TEX, COLOR_TEX_2, COLOR_TEX_3, COLOR_TEX_4, COLOR_TEX_5 = 1, 2, 3, 4, 5


class EGLRenderer:
    def __init__(self, model_paths=None, K=None, height=64, width=64, znear=0.25, zfar=6.0, vertex_scale=1.0,
                 models=None, gpu_id=None, device="cuda", **_unused):
        if models is None:
            models = [load_ply(p, vertex_scale=vertex_scale) for p in (model_paths or [])]
        if not models:
            raise ValueError("EGLRenderer needs model_paths or models")
        if height != width or height > 128:
            raise NotImplementedError("the LDS z-buffer rasteriser renders square maps up to 128x128 (GDRN uses 64x64)")
        self.device = torch.device(device if gpu_id is None else f"cuda:{gpu_id}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.r = CppEGLRenderer.CppEGLRenderer(int(width), int(height), self.device.index)      # egl_renderer_v3.py:93-94
        self.r.init()
        self.meshes = hip_lib.MeshSet([np.asarray(m["pts"], np.float32) for m in models],
                                      [np.asarray(m["faces"], np.int32) for m in models], device=self.device)
        self.height = self.width = int(height)
        self.znear, self.zfar = float(znear), float(zfar)
        self.K = None if K is None else np.asarray(K, np.float32).reshape(3, 3)

    def render(self, obj_ids, poses, K=None, pc_obj_tensor=None, pc_cam_tensor=None, seg_tensor=None, rot_type="mat",
               **_ignored):
        """obj_ids: list[int]; poses: list of 3x4 [R|t]; K 3x3.  Fills the given [H,W,4] float tensors in place."""
        if rot_type != "mat":
            raise NotImplementedError("only rot_type='mat' poses are supported")
        if isinstance(obj_ids, int):
            obj_ids = [obj_ids]
        if isinstance(poses, np.ndarray) and poses.ndim == 2:
            poses = [poses]
        K = self.K if K is None else np.asarray(K, np.float32).reshape(3, 3)
        n, res, dev = len(obj_ids), self.height, self.device
        P = np.stack([np.asarray(p, np.float32)[:3, :4] for p in poses])
        Kt = torch.from_numpy(np.repeat(K[None], n, 0)).to(dev)
        depth, xyz = hip_lib.render_depth(
            self.meshes, torch.tensor(obj_ids, dtype=torch.int32, device=dev), Kt.contiguous(),
            torch.from_numpy(np.ascontiguousarray(P[:, :, :3])).to(dev), torch.from_numpy(np.ascontiguousarray(P[:, :, 3])).to(dev),
            res, self.znear, self.zfar, want_xyz=True)
        far = torch.where(depth > 0, depth, torch.full_like(depth, float("inf")))
        zmin, who = far.min(0)                       # nearest object per pixel
        hit = torch.isfinite(zmin)
        z = torch.where(hit, zmin, torch.zeros_like(zmin))
        # the rasteriser fills the attachments; the caller's tensors receive them through the class boundary of the
        # reference — map_tensor, then the row flip of egl_renderer_v3.py:1196-1225
        hitf = hit.float()
        if pc_obj_tensor is not None:
            sel = xyz.gather(0, who[None, :, :, None].expand(1, res, res, 3))[0]
            self.r.write_attachment(COLOR_TEX_4, torch.cat([torch.where(hit[..., None], sel, torch.zeros_like(sel)), hitf[..., None]], 2))
            self.r.map_tensor(COLOR_TEX_4, self.width, self.height, pc_obj_tensor.data_ptr())
            pc_obj_tensor.data = torch.flip(pc_obj_tensor, (0,))
        if pc_cam_tensor is not None:
            jj, ii = torch.meshgrid(torch.arange(res, device=dev, dtype=torch.float32),
                                    torch.arange(res, device=dev, dtype=torch.float32), indexing="ij")
            cam = torch.stack([(ii + 0.5 - float(K[0, 2])) / float(K[0, 0]) * z, (jj + 0.5 - float(K[1, 2])) / float(K[1, 1]) * z,
                               z, hitf], 2)
            self.r.write_attachment(COLOR_TEX_5, cam)
            self.r.map_tensor(COLOR_TEX_5, self.width, self.height, pc_cam_tensor.data_ptr())
            pc_cam_tensor.data = torch.flip(pc_cam_tensor, (0,))
        if seg_tensor is not None:
            ids = torch.tensor(obj_ids, dtype=torch.float32, device=dev)[who] + 1
            seg = torch.zeros((res, res, 4), dtype=torch.float32, device=dev)
            seg[:, :, 0] = torch.where(hit, ids, torch.zeros_like(ids))
            self.r.write_attachment(COLOR_TEX_3, seg)
            self.r.map_tensor(COLOR_TEX_3, self.width, self.height, seg_tensor.data_ptr())
            seg_tensor.data = torch.flip(seg_tensor, (0,))
        return z

    def close(self):
        self.r.release()
