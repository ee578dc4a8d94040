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
    at 27-28 ms (HISTORY.md section 5).  The package therefore sets the variable to 4 when it is UNSET and the GPU has not been initialised
    yet (the HIP runtime reads it at its first device touch).  A value the user exported is RESPECTED (with a warning that the step was
    measured with 4; RALF_FORCE_HW_QUEUES=1 overrides it); nothing is written once the GPU is initialised.  HW_QUEUES reports what is in
    effect as far as this process can know: {"wanted", "found", "effective", "set_by_package", "hip_initialised_before_import"}."""
    want = "4"
    have = _os.environ.get("GPU_MAX_HW_QUEUES")
    state = {"wanted": want, "found": have, "effective": have, "set_by_package": False, "hip_initialised_before_import": False}
    torch = _sys.modules.get("torch")
    if torch is not None and getattr(torch, "cuda", None) is not None and torch.cuda.is_initialized():
        state["hip_initialised_before_import"] = True   # (by torch; a HIP user this process does not know of cannot be seen from here)
        if have != want:
            _warnings.warn(f"ralf_amd imported after the GPU was initialised with GPU_MAX_HW_QUEUES={have!r}: the step's graph branches were "
                           f"measured with {want} hardware queues (export GPU_MAX_HW_QUEUES={want} or import ralf_amd before touching the GPU)")
        return state
    if have is None or (have != want and _os.environ.get("RALF_FORCE_HW_QUEUES") == "1"):
        _os.environ["GPU_MAX_HW_QUEUES"] = want
        state["effective"], state["set_by_package"] = want, True
    elif have != want:
        _warnings.warn(f"ralf_amd: GPU_MAX_HW_QUEUES={have} is kept as exported; the graph-replayed step was measured with {want} "
                       f"(other values ran it 2x slower on MI355X; RALF_FORCE_HW_QUEUES=1 lets the package set {want})")
    return state


HW_QUEUES = _pin_hardware_queues()
