"""iPOKE hot path on MI355X: the second-stage conditional flow and the first-stage video VAE behind the reference's module surface
(DESIGN.md §1), over the C ABI of ``libipoke_hip.so`` (include/ipoke_hip.h).

Importing the package sets ``GPU_MAX_HW_QUEUES=8`` unless the variable is given: the train step keeps four streams busy (chain,
weight gradients, ready / optimizer, encoder prefetch; RCCL adds its own at N > 1) and the HIP runtime deals streams onto its
hardware queues in creation order -- with the default of 4 queues one more stream in the process makes two busy streams share a
queue (+2.5 ms per step).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the
process: import this package (or set the variable) before anything touches the GPU.  ``hw_queue_setting()`` reports what is in
effect; the trainers warn when the runtime was initialised before the variable was set.
"""
import os
import sys
import warnings

_QUEUES_DEFAULT = "8"
_queues_given = "GPU_MAX_HW_QUEUES" in os.environ
_hip_was_live = False
if not _queues_given:
    _t = sys.modules.get("torch")
    try:
        _hip_was_live = bool(_t is not None and _t.cuda.is_initialized())
    except Exception:      # noqa: BLE001 -- a torch build without the cuda module
        _hip_was_live = False
    os.environ["GPU_MAX_HW_QUEUES"] = _QUEUES_DEFAULT


def hw_queue_setting():
    """(value of GPU_MAX_HW_QUEUES, True when it was in place before the HIP runtime initialised)."""
    return os.environ.get("GPU_MAX_HW_QUEUES"), not _hip_was_live


def warn_if_queues_late(who):
    """Called by the trainers: the step time they were tuned at assumes at least as many hardware queues as busy streams."""
    if _hip_was_live:
        warnings.warn(f"{who}: the HIP runtime was initialised before ipoke_amd could set GPU_MAX_HW_QUEUES={_QUEUES_DEFAULT}; with the "
                      "runtime's default of 4 hardware queues two of the step's busy streams may share a queue (measured +2.5 ms per "
                      "second-stage step).  Import ipoke_amd (or export the variable) before the first GPU call.", stacklevel=3)
