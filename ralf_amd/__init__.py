"""ralf_amd -- MI355X-native (gfx950) implementation of the RALF hot path.

Host side mirrors the reference's operator interface for the path; all device
arithmetic goes through the C ABI of libralf_hip.so (include/ralf_hip.h).
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys
import warnings as _warnings


def _pin_hardware_queues():
    """The captured train step spreads its graph branches over HIP streams that ROCclr maps onto GPU_MAX_HW_QUEUES hardware queues; which
    branch shares a queue with which decides whether a graph edge is an in-queue order or a cross-queue signal.  Measured on MI355X: 4
    queues (ROCclr's default) run the data-parallel three-graph step at 16.2 ms, 8 at 35.6 ms, 6 / 9 / 10 / 12 / 16 run even the plain step
    at 27-28 ms.  The runtime therefore OWNS this setting: it is pinned to 4 when the package is imported, which is
    before the first device touch in every script that goes through the package (the HIP runtime reads it when the device is first
    initialised).  RALF_KEEP_HW_QUEUES=1 leaves a user's value alone."""
    want = "4"
    have = _os.environ.get("GPU_MAX_HW_QUEUES")
    state = {"wanted": want, "found": have, "pinned": False, "hip_initialised_before_import": False}
    torch = _sys.modules.get("torch")
    if torch is not None and getattr(torch, "cuda", None) is not None and torch.cuda.is_initialized():
        state["hip_initialised_before_import"] = True
        if have != want:
            _warnings.warn(f"ralf_amd imported after the GPU was initialised with GPU_MAX_HW_QUEUES={have!r}: the step's graph branches were "
                           f"measured with {want} hardware queues (export GPU_MAX_HW_QUEUES={want} or import ralf_amd before touching the GPU)")
        return state
    if have is not None and have != want and _os.environ.get("RALF_KEEP_HW_QUEUES") == "1":
        return state
    if have != want:
        if have is not None:
            _warnings.warn(f"ralf_amd: GPU_MAX_HW_QUEUES={have} -> {want} (the measured setting of the graph-replayed step; RALF_KEEP_HW_QUEUES=1 keeps yours)")
        _os.environ["GPU_MAX_HW_QUEUES"] = want
    state["pinned"] = True
    return state


HW_QUEUES = _pin_hardware_queues()
