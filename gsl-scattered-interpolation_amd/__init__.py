"""MI355X-native scattered-data interpolation (GSL-style C API) -- Python plumbing.

The product is the C-ABI shared library ``libgsl_sinterp.so`` (C host code +
hand-written gfx950 HIP kernels; headers in ``include/``).  This package only
binds it with ctypes for tests, ``bench.py`` and the multi-GPU driver; torch is
used for device memory, streams and ``torch.distributed`` (RCCL), nothing else.

There is no fallback: if the library is missing, importing :mod:`capi` raises.
The directory name contains a hyphen, so load it through
``__graft_entry__.load_package()`` (importlib), which registers it as
``gsl_sinterp_amd``.
"""
from . import capi, sharding  # noqa: F401
from .capi import (  # noqa: F401
    GSL_SUCCESS, GSL_EDOM, GSL_EINVAL, GSL_EFAILED, RBF_GAUSSIAN, RBF_TPS, RBF_WENDLAND,
    HipContext, SimplexMesh, SimplexTree, Sinterp, lib, library_path,
)
