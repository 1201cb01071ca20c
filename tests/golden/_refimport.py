"""Import the reference's OWN Python modules in the authoring container (needs /root/reference; never used on the GPU box).

The reference cannot be imported as it stands: mmcv, timm, detectron2, cv2, transforms3d, vispy … are not installed and
there is no network.  None of those packages does arithmetic on the network / evaluator path that is pinned here, they
only supply initialisers, registries, loggers and base classes.  This module therefore

* registers a FALLBACK meta-path finder that fabricates inert stand-in modules for exactly the missing third-party
  roots listed in ``STUB_ROOTS`` (any attribute of such a module is an inert object);
* fills in the handful of names the pinned code really calls with their published behaviour:
  ``mmcv.cnn.normal_init / constant_init / kaiming_init`` (thin wrappers over ``torch.nn.init``), the mmcv
  ``CONV_LAYERS`` registry (``Conv2d`` -> ``nn.Conv2d``), ``timm.models.layers.StdConv2d`` (unused unless
  ``use_ws``), ``detectron2.utils.env.TORCH_VERSION``, ``detectron2.evaluation.DatasetEvaluator`` (a plain base class)
  and ``transforms3d.axangles.axangle2mat`` (served by ``scipy.spatial.transform.Rotation``);
* restores the NumPy < 1.24 aliases the reference's ``lib/pysixd`` still uses (``np.float``, ``np.maximum_sctype``) and
  the pre-3.10 ``collections.Sequence`` aliases (``core/utils/data_utils.py:1``).

After ``install()`` the reference's modules import from their files and run unmodified, e.g.
``from core.gdrn_modeling.models.GDRN_double_mask import GDRN_DoubleMask, build_model_optimizer``.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch.nn as nn

REF = "/root/reference"

STUB_ROOTS = {
    "mmcv", "timm", "detectron2", "cv2", "transforms3d", "loguru", "fvcore", "pytorch_lightning", "setproctitle", "vispy",
    "OpenGL", "fairscale", "open3d", "pyrender", "ruamel", "omegaconf", "termcolor", "pycocotools", "imageio", "mmengine",
    "PIL", "matplotlib", "tensorboardX", "imgaug", "plyfile", "png", "glumpy", "thop", "numba", "chardet", "torchvision",
    "ref", "pytorch3d", "kornia", "pyassimp", "skimage", "trimesh", "pyximport", "numba", "seaborn", "dr", "torchcontrib",
    "ranger", "horovod", "apex", "wandb", "tensorboard", "egl_renderer", "gin", "pprofile", "pympler", "petrel_client", "mc",
}


# compiled extension modules that live INSIDE the reference's packages (absent: nothing was built)
STUB_FULL = {"lib.egl_renderer.CppEGLRenderer", "lib.egl_renderer.egl_renderer_v3"}  # the latter binds EGL through ctypes at import


class _Inert:
    """Attribute/call sink for names of absent third-party packages that the pinned code paths never execute."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()

    def __mro_entries__(self, bases):
        return (object,)

    def __iter__(self):
        return iter(())

    def __getitem__(self, k):
        return _Inert()

    def __contains__(self, k):
        return False


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _Inert()
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Fallback (end of sys.meta_path): fabricates modules for the absent third-party roots only."""

    def wants(self, fullname):
        return fullname.split(".")[0] in STUB_ROOTS

    def find_spec(self, fullname, path, target=None):
        if self.wants(fullname):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _FrontFinder(_Finder):
    """Ahead of the path finder: reference modules that exist as files but cannot be executed here."""

    def wants(self, fullname):
        return fullname in STUB_FULL


def _normal_init(module, mean=0, std=1, bias=0):          # mmcv/cnn/utils/weight_init.py
    nn.init.normal_(module.weight, mean, std)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def _constant_init(module, val, bias=0):
    if getattr(module, "weight", None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def _kaiming_init(module, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
    if distribution == "uniform":
        nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


class _Registry(dict):
    def register_module(self, *a, **k):
        def deco(cls):
            self[cls.__name__] = cls
            return cls
        return deco


class DatasetEvaluator:
    """detectron2/evaluation/evaluator.py: the protocol base class (reset / process / evaluate), no behaviour."""

    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        pass


def axangle2mat(axis, angle, is_normalized=False):
    """transforms3d.axangles.axangle2mat, served by an independent implementation of the same rotation."""
    from scipy.spatial.transform import Rotation
    axis = np.asarray(axis, np.float64)
    if not is_normalized:
        axis = axis / np.linalg.norm(axis)
    return Rotation.from_rotvec(axis * angle).as_matrix()


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    if not os.path.isdir(REF):
        raise RuntimeError(f"{REF} is not present: golden vectors are generated in the authoring container only")
    sys.meta_path.append(_Finder())
    sys.meta_path.insert(0, _FrontFinder())
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float
        np.int = int
    import collections
    import collections.abc
    for name in ("Sequence", "Mapping", "Iterable", "MutableMapping"):   # removed from `collections` in Python 3.10
        if not hasattr(collections, name):
            setattr(collections, name, getattr(collections.abc, name))
    if not hasattr(np, "maximum_sctype"):
        np.maximum_sctype = lambda t: np.float64  # noqa: E731

    import mmcv.cnn
    import mmcv.cnn.utils
    import mmcv.cnn.bricks.conv as mconv
    for mod in (mmcv.cnn, mmcv.cnn.utils):
        mod.normal_init, mod.constant_init, mod.kaiming_init = _normal_init, _constant_init, _kaiming_init
    mconv.CONV_LAYERS = _Registry(Conv2d=nn.Conv2d, Conv=nn.Conv2d)
    import detectron2.utils.env as d2env
    d2env.TORCH_VERSION = (2, 10)
    import detectron2.evaluation as d2eval
    d2eval.DatasetEvaluator = DatasetEvaluator
    import timm.models.layers as tl
    tl.StdConv2d = nn.Conv2d
    import transforms3d.axangles as t3a
    t3a.axangle2mat = axangle2mat


def load_ref_config(rel_path):
    """mmcv.Config.fromfile for the reference's python configs: the file is executed, ``_base_`` files are loaded first
    and merged the way mmcv does (dicts merge recursively; a child dict carrying ``_delete_=True`` replaces)."""
    path = os.path.join(REF, rel_path)
    ns = {}
    exec(compile(open(path).read(), path, "exec"), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not isinstance(v, types.ModuleType) and k != "_base_"}
    base = ns.get("_base_", [])
    if isinstance(base, str):
        base = [base]
    merged = {}
    for b in base:
        merged = _merge(merged, load_ref_config(os.path.normpath(os.path.join(os.path.dirname(rel_path), b))))
    return _merge(merged, cfg)


def _merge(base, child):
    import copy
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            v = copy.deepcopy(v)
            if isinstance(v, dict):
                v.pop("_delete_", None)
            out[k] = v
    return out
