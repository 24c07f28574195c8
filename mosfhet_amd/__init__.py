"""mosfhet_amd -- MI355X-native TFHE bootstrap engine behind MOSFHET's API.

The product is the native library ``libmosfhet_hip.so`` (HIP kernels + C ABI, include/mosfhet_hip.h, plus the
MOSFHET-compatible host layer, include/mosfhet_compat.h).  This Python package is a thin ctypes binding used
by the tests and bench.py; PyTorch supplies device memory, streams and torch.distributed only.

There is NO CPU fallback: importing works without a GPU (so the build can be checked), but every compute
call raises when the native library or the device is missing.
"""
from .engine import (Engine, BootstrapKey, KeySwitchKey, MosfhetHipError, lib, lib_path, to_device, to_numpy,  # noqa: F401
                     PARAMS_SET1, PARAMS_LVL2, PARAMS_SET2, PARAMS_SET3)
from . import host  # noqa: F401
