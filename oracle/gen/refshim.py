"""Import helper used ONLY by the golden-vector generator scripts in this directory.

TEST INFRASTRUCTURE -- never imported by the product package, never shipped to the GPU box as a
dependency (the reference tree it points at does not exist there).

The reference (`/root/reference/metadrive`) is pure Python, but most modules `import panda3d`,
`gymnasium`, `shapely`, ... at module scope.  Those third-party packages are absent in this
container and stay absent.  To reach the reference's *own* pure arithmetic (lane Frenet math,
block parameter sampling, BIG search, IDM/PID formulas, lidar angular mask, observation
normalisation) this module installs a meta-path finder that fabricates inert placeholder modules
for the absent third-party roots, so that `import` statements at the top of reference files
succeed.  No number in any golden vector flows through a placeholder: every generator script
only records outputs of reference functions whose arithmetic is numpy/math on plain floats, and
asserts that no placeholder object appears in what it records (see `assert_plain`).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from abc import ABCMeta
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get("MD_REFERENCE_ROOT", "/root/reference")

ABSENT_ROOTS = {
    "panda3d", "direct", "gltf", "gymnasium", "gym", "pygame", "shapely", "seaborn", "cv2", "lxml",
    "geopandas", "progressbar", "yapf", "simplepbr", "OpenGL", "cupy", "cuda", "PIL", "matplotlib",
    "tqdm_unused",
}


class _StubMeta(ABCMeta):
    """Metaclass: attribute access on a stub class yields nested stub classes / mocks."""
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name[:1].isupper():
            sub = _StubMeta(name, (Stub, ), {})
            setattr(cls, name, sub)
            return sub
        m = MagicMock(name="%s.%s" % (cls.__name__, name))
        return m

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


class Stub(metaclass=_StubMeta):
    """Inert placeholder object."""
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return MagicMock(name=name)

    def __iter__(self):  # `LPoint3f(*LPoint3f(...))` in block construction only needs an iterable
        return iter(())

    def __len__(self):
        return 0


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name[:1].isupper():
            sub = _StubMeta(name, (Stub, ), {})
        else:
            sub = MagicMock(name="%s.%s" % (self.__name__, name))
        setattr(self, name, sub)
        return sub


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in ABSENT_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Make `import metadrive...` work for the arithmetic-only subset. Idempotent."""
    global _installed
    if _installed:
        return
    sys.dont_write_bytecode = True  # never drop __pycache__ into the reference tree
    sys.meta_path.insert(0, _Finder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def is_placeholder(x):
    return isinstance(x, (MagicMock, Stub)) or isinstance(x, _StubMeta)


def assert_plain(x, where=""):
    """Recursively assert that `x` holds only plain numbers / strings / containers."""
    import numpy as np
    if is_placeholder(x):
        raise AssertionError("placeholder object leaked into a golden vector: %s" % where)
    if isinstance(x, dict):
        for k, v in x.items():
            assert_plain(k, where)
            assert_plain(v, "%s[%r]" % (where, k))
    elif isinstance(x, (list, tuple)):
        for i, v in enumerate(x):
            assert_plain(v, "%s[%d]" % (where, i))
    elif isinstance(x, (int, float, str, bool, type(None), np.ndarray, np.generic)):
        return
    else:
        raise AssertionError("unexpected type %r at %s" % (type(x), where))
