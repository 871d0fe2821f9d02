"""demon_amd -- MI355X-native (gfx950) DeMoN inference path behind the reference's Python API.

The numeric work lives in libdemon_hip.so (demon_amd/csrc, C ABI in include/demon_hip.h); this package is
the thin host side: ctypes binding, weight table / IO, and mirrors of the reference interfaces
`depthmotionnet.networks_original` and `lmbspecialops`.
"""
import os as _os

# Hardware queues.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two busy
# lanes (demon_amd/lanes.py) that share one serialise.  Measured on an MI355X at batch 32 (round 5, one box per file):
#   plain process (gpurun_out/r5c/ab.txt):  4 queues -> best 3 lanes, 4 488 - 4 499 pairs/s;  8 queues -> best 4 lanes, 4 622 - 4 640 (+3 %);
#                                           5 / 6 / 10 / 12 / 16 queues and 5 - 6 lanes: no gain
#   under torch.distributed.run with the RCCL communicator alive (gpurun_out/r5h_torchrun_queues.txt): 8 queues -> 3 lanes, 4 452;
#                                           6 / 10 / 12 -> 3 lanes, 4 418 - 4 445;  16 queues -> 4 lanes, 4 646 (four good calibration cells)
# (which cell of the (lanes, placeholder streams) calibration is good moves with every stream alive in the process; the launcher's and
# RCCL's streams shift it).  The runtime reads the variable when it initialises (first HIP call of the process), so it is set here, at
# import, unless the caller already chose a value: 16 under a torch.distributed launcher (LOCAL_RANK / TORCHELASTIC_RUN_ID in the
# environment), 8 otherwise; DEMON_HW_QUEUES=<n> picks another count, DEMON_HW_QUEUES=0 leaves the runtime's default alone.  A C / C++
# host sets GPU_MAX_HW_QUEUES in its own environment.
_q = _os.environ.get("DEMON_HW_QUEUES") or ("16" if ("LOCAL_RANK" in _os.environ or "TORCHELASTIC_RUN_ID" in _os.environ) else "8")
if _q != "0":
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", _q)
del _q

from .engine import DemonContext, DemonError  # noqa: F401,E402
from .runtime import get_context, set_default_weights, default_weights  # noqa: F401,E402
from . import weights  # noqa: F401,E402
