"""``CppEGLRenderer`` — the class boundary of the reference's pybind module of the same name
(lib/egl_renderer/cpp/egl_renderer.cpp:99-311: ``CppEGLRenderer(w, h, device)``, ``init()``, ``query()``,
``map_tensor(tex_id, w, h, dev_ptr)``, ``draw(array)``, ``release()``; imported as ``from . import CppEGLRenderer`` and used
as ``CppEGLRenderer.CppEGLRenderer(...)`` at egl_renderer_v3.py:27,93-94,1185-1225).

The reference class owns an EGL context bound to a CUDA device (``EGL_CUDA_DEVICE_NV``) and copies GL colour attachments into
device tensors through CUDA-GL interop.  Neither exists on ROCm, and nothing needs to: the "attachments" here are plain
[h, w, 4] float32 device buffers that the HIP rasteriser fills (``gdrnpp_render_depth``; ``EGLRenderer.render`` of this
package calls ``write_attachment``), kept in GL ROW ORDER (bottom row first) so that a caller written against the reference —
``map_tensor(...)`` followed by ``torch.flip(t, (0,))`` — gets the same image.  ``map_tensor`` is the same device-to-device
transfer into a raw device pointer.  Texture ids follow the renderer's attachment numbering (egl_renderer_v3.py:205-262):
1 colour, 2 normal, 3 segmentation, 4 object-space points, 5 camera-space points."""
from __future__ import annotations

import torch

from ... import hip_lib

MAX_NUM_RESOURCES = 10       # egl_renderer.cpp:23


class CppEGLRenderer:
    def __init__(self, w: int, h: int, d: int):
        self.m_windowWidth, self.m_windowHeight, self.m_renderDevice = int(w), int(h), int(d)
        self._tex = {}
        self._ready = False

    def init(self) -> int:
        """The reference creates the EGL display / context on the CUDA device here (:121-198); the equivalent is binding
        the HIP device and loading the library — failures raise instead of ``exit(EXIT_FAILURE)``."""
        if not torch.cuda.is_available():
            raise RuntimeError("CppEGLRenderer.init: no HIP device")
        if not 0 <= self.m_renderDevice < torch.cuda.device_count():
            raise RuntimeError(f"CppEGLRenderer.init: device {self.m_renderDevice} of {torch.cuda.device_count()}")
        hip_lib.load()
        self.device = torch.device("cuda", self.m_renderDevice)
        self._ready = True
        return 0

    def query(self) -> None:
        print(f"CppEGLRenderer: HIP device {self.m_renderDevice} ({torch.cuda.get_device_name(self.m_renderDevice)}), "
              f"{self.m_windowWidth}x{self.m_windowHeight}, {len(self._tex)} attachments")

    def write_attachment(self, tex_id: int, image_hw4: torch.Tensor) -> None:
        """(not in the reference class: there GL draws into the FBO.)  Store an [h, w, 4] image given in IMAGE row order
        as attachment ``tex_id``, in GL row order."""
        if not self._ready:
            raise RuntimeError("CppEGLRenderer: init() was not called")
        if not 0 < tex_id < MAX_NUM_RESOURCES:
            raise RuntimeError(f"CppEGLRenderer: texture id {tex_id} outside 1..{MAX_NUM_RESOURCES - 1}")
        if tuple(image_hw4.shape) != (self.m_windowHeight, self.m_windowWidth, 4) or image_hw4.dtype != torch.float32:
            raise RuntimeError("CppEGLRenderer: an attachment is float32 [h, w, 4]")
        self._tex[int(tex_id)] = torch.flip(image_hw4, (0,)).contiguous()

    def map_tensor(self, tex_id: int, width: int, height: int, data: int) -> None:
        """Copy attachment ``tex_id`` ([height, width, 4] float32, GL row order) to the device pointer ``data``."""
        t = self._tex.get(int(tex_id))
        if t is None:
            raise RuntimeError(f"CppEGLRenderer.map_tensor: nothing was rendered into texture {tex_id}")
        if (int(height), int(width)) != (self.m_windowHeight, self.m_windowWidth):
            raise RuntimeError("CppEGLRenderer.map_tensor: size differs from the render target")
        hip_lib.copy_d2d(int(data), t)

    def draw(self, x) -> None:
        """egl_renderer.cpp:238-260 is a binding smoke test that fills the array with 42."""
        x[...] = 42

    draw_py = draw

    def release(self) -> None:
        self._tex.clear()
        self._ready = False
