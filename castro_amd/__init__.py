"""castro_amd -- MI355X-native CTU Godunov hydro advance for Castro (one hot path, C-ABI drop-in).

Layout:
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/castro_hydro_amd.h)
  _lib.py          ctypes binding of libcastro_hydro_amd.so (no CPU fallback)
  hydro.py         HipHydro: the reference's per-FAB hydro interface over the C ABI
  castro.py        Castro: single-level driver (FillPatch halo exchange over RCCL, dt control, retry, gravity)
  amr.py           CastroAmr: coarse level + one refined patch with subcycling, reflux, avgDown
  plotfile.py      Castro plotfile writer / reader
"""
from ._lib import (NUM_STATE, NGDNV, NUM_GROW, URHO, UMX, UMY, UMZ, UEDEN, UEINT, UTEMP, UFS,
                   default_params, make_geom, make_rotation, LIB_PATH)
from .castro import Castro, DistComm, SingleComm, AdvanceFailure, default_grid
from .amr import CastroAmr

__all__ = ["Castro", "CastroAmr", "DistComm", "SingleComm", "AdvanceFailure", "default_grid", "default_params", "make_geom", "make_rotation",
           "NUM_STATE", "NGDNV", "NUM_GROW", "LIB_PATH"]


def HipHydro(*a, **k):
    from .hydro import HipHydro as _H
    return _H(*a, **k)
