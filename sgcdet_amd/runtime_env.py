"""Process-level settings of the HIP runtime that the hot path depends on.

``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0``
    ROCm 7's hipGraph launch has a fast path that records the AQL packets of a graph at instantiation and
    re-submits them as one batch.  While such a batch replays on one stream, kernels launched eagerly on
    ANOTHER stream occasionally observe stale 128-byte lines of a buffer written by the kernel just before
    them (measured on MI355X, tools/race_check.py: 106-116 wrong scenes in 1500 with two scenes in flight,
    always whole cache lines of a freshly written gather output; 0 in 1500 with the knob off, 0 in 1500
    without graphs, 0 on a single stream).  The per-node launch path keeps the same replay throughput
    (323 vs 329 scenes/s on config 2), so the package turns the fast path off -- the variable is read when
    the HIP runtime initialises, hence it is set at import time, before the first HIP call.

``graph_concurrency_safe()`` tells the scene pipeline whether that happened in time; if not (the GPU was
already initialised when the package was imported and the variable was not in the environment) pipelines
with more than one stream launch the neck/head eagerly instead of replaying hipGraphs.
"""
import os
import sys

GRAPH_KNOB = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_state = {"in_time": None}


def _hip_initialised():
    torch = sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:  # pragma: no cover
        return False


def apply():
    """Called once from ``sgcdet_amd/__init__``."""
    if _state["in_time"] is not None:
        return
    preset = os.environ.get(GRAPH_KNOB)
    if preset is not None:
        _state["in_time"] = preset.strip() in ("0", "false", "False")
        return
    if _hip_initialised():
        _state["in_time"] = False
        return
    os.environ[GRAPH_KNOB] = "0"
    _state["in_time"] = True


def graph_concurrency_safe():
    """True when hipGraph replays may run beside eager kernels of another stream (see module docstring)."""
    return bool(_state["in_time"])
