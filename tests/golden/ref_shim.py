"""Import shim for the reference G-VOM under Numba's pure-Python CUDA simulator.

THIS CONTAINER ONLY.  Run with /opt/conda/bin/python3.9 (numba 0.54.1, numpy 1.26.4).
It imports /root/reference/scripts/gvom.py *unmodified* (nothing of the reference is
copied into this repo) so that `make_golden.py` can record inputs + outputs as small
.npz fixtures.  The fixtures (data only) are what travels to the GPU box.

Shim items:
 1. numba 0.54.1 refuses numpy > 1.20 and its `_internal` ufunc C-extension fails to
    initialise against numpy 1.26 -> serve a stub for that one module (never used by
    the simulator) and fake the version string during the import only.
 2. The reference spells local arrays `numba.cuda.local.array(...)`; the simulator only
    provides `cuda.local` inside a running kernel -> provide a module-level object
    whose .array() returns a numpy array of the requested dtype.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

os.environ["NUMBA_ENABLE_CUDASIM"] = "1"
os.environ["NUMBA_DISABLE_JIT"] = "1"
sys.dont_write_bytecode = True

REFERENCE_SCRIPTS = "/root/reference/scripts"


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    names = ("numba.np.ufunc._internal",)

    def find_spec(self, fullname, path, target=None):
        if fullname in self.names:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        # attributes numba.np.ufunc imports from _internal at module import time
        m.PyUFunc_None = -1
        m.PyUFunc_Zero = 0
        m.PyUFunc_One = 1
        m.PyUFunc_ReorderableNone = -2
        m._ufunc = type("_ufunc", (), {})

        class _DUFunc(object):
            def __init__(self, *a, **k):
                pass
        m._DUFunc = _DUFunc
        return m

    def exec_module(self, module):
        pass


def load_reference():
    """Returns the imported reference `gvom` module (simulator-backed)."""
    import numpy as np
    for alias, typ in (("bool", bool), ("int", int), ("float", float), ("complex", complex),
                       ("object", object), ("str", str)):
        if alias not in np.__dict__:
            setattr(np, alias, typ)
    sys.meta_path.insert(0, _StubFinder())
    real_version = np.__version__
    np.__version__ = "1.20.3"
    try:
        import numba
        from numba import cuda
        from numba.np import numpy_support
    finally:
        np.__version__ = real_version

    class _Local(object):
        @staticmethod
        def array(shape, dtype):
            try:
                dt = numpy_support.as_dtype(dtype)
            except Exception:
                dt = np.dtype(dtype)
            return np.empty(shape, dt)

    if not hasattr(numba.cuda, "local") or True:
        numba.cuda.local = _Local()

    if REFERENCE_SCRIPTS not in sys.path:
        sys.path.insert(0, REFERENCE_SCRIPTS)
    import gvom  # the reference, unmodified
    assert os.path.abspath(gvom.__file__).startswith(REFERENCE_SCRIPTS), gvom.__file__
    return gvom
