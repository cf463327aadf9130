"""pogema_amd -- MI355X-native vectorized POGEMA step engine (drop-in for the reference's
grid.py / envs.py reset()/step() hot path).  See DESIGN.md."""
from .grid_config import (GridConfig, Easy8x8, Normal8x8, Hard8x8, Easy16x16, Hard16x16, Easy32x32,
                          Hard32x32, Easy64x64, Hard64x64)

__version__ = "0.1.0"

from .semantics import Semantics

__all__ = ["GridConfig", "Semantics", "release_cached_buffers", "VecPogema", "PipelinedVecPogema", "Pogema", "pogema_v0", "Easy8x8", "Normal8x8", "Hard8x8", "Easy16x16",
           "Hard16x16", "Easy32x32", "Hard32x32", "Easy64x64", "Hard64x64"]


def release_cached_buffers() -> None:
    """Gives back the zone-spread observation buffers that closed environments left mapped for their successors
    (pogema_amd.buffers.ParkedBuffers: at most PGX_POOL_CACHE_MB, default 2.5 GiB) -- for callers who close an
    environment to make room for something else."""
    from .buffers import ParkedBuffers, WalkVerdicts
    ParkedBuffers.clear()
    WalkVerdicts.clear()  # ... and the next engine may walk the device again


def __getattr__(name):
    # engine classes import torch + the HIP library lazily so that `GridConfig` stays importable anywhere
    if name == "VecPogema":
        from .vec_env import VecPogema
        return VecPogema
    if name == "PipelinedVecPogema":
        from .pipeline import PipelinedVecPogema
        return PipelinedVecPogema
    if name in ("Pogema", "pogema_v0", "PogemaParallel"):
        from . import envs
        return getattr(envs, name)
    raise AttributeError(name)
