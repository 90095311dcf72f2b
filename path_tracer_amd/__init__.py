"""MI355X-native per-pixel render path of Qervas/path_tracer (host-side Python plumbing).

The product is the HIP library behind include/pt.h (path_tracer_amd/lib/libpt_amd.so);
this package only binds it (ctypes), builds scenes in the reference's vocabulary and wires
torch device memory / streams / torch.distributed around it.
"""
from . import _abi as abi  # noqa: F401
from . import scenes  # noqa: F401

__all__ = ["abi", "scenes"]
