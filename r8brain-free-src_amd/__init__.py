"""MI355X-native batched sample-rate converter with the r8brain-free-src front-end.

The directory name carries a hyphen (it is fixed by the task), so import it with

    import importlib; r8b = importlib.import_module("r8brain-free-src_amd")

Public names: BatchResampler, CDSPResampler, CDSPResampler16, CDSPResampler16IR,
CDSPResampler24, DLLResampler (see resampler.py) and `load()` (the ctypes handle of
libr8bsrc_hip.so).  Everything computes on the GPU through the C ABI of include/r8bsrc.h.
"""
from ._capi import load, lib_path, bind, PROTOTYPES  # noqa: F401
from .resampler import (BatchResampler, CDSPResampler, CDSPResampler16, CDSPResampler16IR,  # noqa: F401
                        CDSPResampler24, DLLResampler, fprLinearPhase, fprMinPhase,
                        PCM_F64, PCM_F32, PCM_S16, PCM_S24, PCM_S32)


_SHARDING = ("channel_shard", "scatter_channels", "gather_channels", "ShardedBatchResampler",
             "RootPipeline")


def __getattr__(name):
    # the multi-GPU helpers need torch.distributed; the numpy-only host API must import without it
    if name in _SHARDING:
        from . import sharding
        return getattr(sharding, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
