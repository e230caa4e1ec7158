"""grad_traj_optimization_amd — MI355X-native batched cost/gradient path of GTOP.

The product is the C-ABI library ``libgtop_hip.so`` (include/gtop.h) built from
csrc/ with hipcc for gfx950.  This package is the thin Python host side:
``_lib`` binds the C-ABI with ctypes, ``GtopContext`` wraps a context, and
``problem`` generates the synthetic inputs of SURVEY.md §8d.  torch is used
only for device memory, streams and torch.distributed.

There is no CPU fallback: importing works anywhere (so the build and the
symbol check can run on a CPU box), but creating a context without a gfx950
device raises.
"""
from ._lib import (GTOP_F32, GTOP_F64, GtopError, GtopParams, OPTI_NODE_PARAMS,
                   GtopContext, GtopGroup, Rendezvous, library_path, load_library)

__all__ = ["GTOP_F32", "GTOP_F64", "GtopError", "GtopParams", "OPTI_NODE_PARAMS",
           "GtopContext", "GtopGroup", "Rendezvous", "library_path", "load_library"]
