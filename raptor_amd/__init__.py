"""raptor_amd -- MI355X (gfx950) implementation of RAPTOR's constraint-evaluation hot path.

The compute lives in ``libraptor_hip.so`` (hand-written HIP, built in-tree from ``csrc/``) behind the
C ABI of ``include/raptor_hip.h``.  This package is the thin host-side mirror of the reference's
classes (``raptor_amd.optimizer``: the Optimizer / TNLP shell; ``raptor_amd._capi``: problem handles) plus the URDF loader.  There is no CPU fallback: importing the API
without the built library raises.
"""
from __future__ import annotations

import os

from . import _capi
from ._capi import (EVAL_G, EVAL_JAC, TIME_CHEBYSHEV, TIME_UNIFORM, TRAJ_ARMOUR_BEZIER, TRAJ_BEZIER,
                    TRAJ_POLYNOMIAL, RaptorError, RaptorInvalidArgument)
from .urdf import RobotModel, load_urdf, lock_joints

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libraptor_hip.so")
_api = None


def api() -> "_capi.CApi":
    """The bound HIP library; fails loudly when it has not been built."""
    global _api
    if _api is None:
        path = os.environ.get("RAPTOR_HIP_LIB", LIB_PATH)  # A/B builds of the same library
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -m raptor_amd.build` "
                "(the hot path has no CPU fallback)")
        _api = _capi.CApi(path, "raptor_")
    return _api
