"""mvin_amd -- MI355X (gfx950) native implementation of MVIN's K-hop neighbor-attention
aggregation + embedding-propagation scoring path, behind the reference's MVIN / Aggregator
Python call surface (johnnyjana730/MVIN src/model/MVIN/model.py, aggregators.py).

Host code: Python on PyTorch-ROCm (device memory, streams, torch.distributed only).
Compute: hand-written HIP kernels in libmvin_hip.so, reached through the C ABI declared in
include/mvin_hip.h.  There is no CPU fallback: compute entry points raise if the extension
is missing.
"""
from .config import make_args, parameter_env  # noqa: F401

__all__ = ["make_args", "parameter_env", "MVIN", "SumAggregator_urh_matrix", "Aggregator"]


def __getattr__(name):
    # torch-dependent modules are imported lazily so that host-only utilities (config,
    # synth, params, build) stay importable without initialising torch
    if name == "MVIN":
        from .model import MVIN
        return MVIN
    if name in ("SumAggregator_urh_matrix", "Aggregator"):
        from . import aggregators
        return getattr(aggregators, name)
    raise AttributeError(name)
