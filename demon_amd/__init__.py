"""demon_amd -- MI355X-native (gfx950) DeMoN inference path behind the reference's Python API.

The numeric work lives in libdemon_hip.so (demon_amd/csrc, C ABI in include/demon_hip.h); this package is
the thin host side: ctypes binding, weight table / IO, and mirrors of the reference interfaces
`depthmotionnet.networks_original` and `lmbspecialops`.
"""
import os as _os

# Hardware queues: importing the package does NOT touch the process environment (round 6; it used to set GPU_MAX_HW_QUEUES here).  The
# module that needs more hardware queues -- demon_amd.lanes, several passes in flight on one GPU -- asks for them when IT is imported
# (lanes.request_hw_queues), says so in every record it produces, and warns when the HIP runtime was already initialised.
from .engine import DemonContext, DemonError  # noqa: F401
from .runtime import get_context, set_default_weights, default_weights  # noqa: F401
from . import weights  # noqa: F401
