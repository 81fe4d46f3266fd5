"""The C-ABI library loads without a GPU and exports every symbol include/raptor_hip.h declares."""
import ctypes
import os
import re

import raptor_amd
from raptor_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "raptor_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(raptor_[a-z_]+)\s*\(", text)))


def test_header_symbols_exported():
    syms = declared_symbols()
    assert len(syms) >= 16
    assert os.path.exists(raptor_amd.LIB_PATH), "build with python -m raptor_amd.build"
    lib = ctypes.CDLL(raptor_amd.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted("raptor_" + e for e in _capi.EXPORTS + _capi.DEVICE_ONLY_EXPORTS) == syms


def test_oracle_exports_same_surface(oracle):
    for e in _capi.EXPORTS:
        assert hasattr(oracle.lib, "oracle_" + e)


def test_argument_errors_without_gpu():
    """Host-side validation happens before any HIP call, so it is checkable here."""
    api = raptor_amd.api()
    out = ctypes.c_void_p()
    assert api.fn("model_create")(None, ctypes.byref(out)) == _capi.ERR_INVALID_ARGUMENT
    assert b"null" in api.fn("last_error")()


def test_reset_map_wiring_errors_without_gpu(oracle):
    """The reset-map block needs v_reset / lambda_reset in z and a complete JOINT_MAP: refused the same way by the oracle and by the
    HIP library's host-side validation (no GPU needed: set-up checks run before the first HIP call)."""
    import os

    import pytest

    import helpers as H
    from raptor_amd import robots

    rd = os.path.join(ROOT, "tests", "robots")
    m = robots.talos_model(os.path.join(rd, "talos_reduced_armfixed_floatingbase.urdf"))
    for api in (oracle, raptor_amd.api()):
        keep = []
        mh = api.model_create(m)
        keep.append(mh)
        cfg = robots.talos_config(m)
        cfg.period = robots.talos_period(m)
        cfg.extra_variables = 0
        t = _capi.make_traj_desc(keep, _capi.TRAJ_BEZIER, cfg.T, 4, 12, cfg.degree, _capi.TIME_CHEBYSHEV)
        with pytest.raises(_capi.RaptorInvalidArgument):
            _capi.ProblemHandle.legged(api, mh, t, cfg)
        cfg = robots.talos_config(m)
        cfg.period = dict(robots.talos_period(m))
        del cfg.period["joint_id1"], cfg.period["joint_id2"]
        with pytest.raises(_capi.RaptorInvalidArgument):
            _capi.ProblemHandle.legged(api, mh, t, cfg)
