"""A two-object stand-in for a cffi API-mode module (``ffi``, ``lib``), built on ctypes because cffi is
not installed here.  The reference calls ``ffi.cast("float*", ndarray.ctypes.data)`` and
``lib.<symbol>(ptr, …)`` (core/csrc/fps/fps_utils.py:13-19, uncertainty_pnp/un_pnp_utils.py:58-74)."""
import ctypes

from ... import hip_lib


class _FFI:
    @staticmethod
    def cast(ctype: str, value):
        assert ctype.strip().endswith("*"), ctype
        return ctypes.c_void_p(int(value))


class _Lib:
    def __init__(self, names):
        self._names = set(names)

    def __getattr__(self, name):
        if name.startswith("_") or name not in self._names:
            raise AttributeError(name)
        return getattr(hip_lib.load(), name)  # raises loudly if libgdrnpp_hip.so is missing


def make(names):
    return _FFI(), _Lib(names)
