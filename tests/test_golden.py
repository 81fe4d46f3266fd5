"""Committed fixtures under tests/golden/ (made by scripts/gen_golden.py from the pinned oracle; the reference itself
holds no vectors at fixed z).  CPU: the oracle reproduces them (guards the checker against drift across
compilers / edits).  GPU: the HIP path matches them through the C ABI at the stated tolerance."""
import json
import os

import numpy as np
import pytest

import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-9


def _problem(api, name, keep, kinova):
    if name == "kinova_N4":
        return H.kinova_problem(api, kinova, keep, N=4, boxes=H.reach_boxes())
    if name == "digit_N2":
        return H.digit_problem(api, keep, N=2)[0]
    if name == "digit_full_N2":
        return H.digit_problem(api, keep, N=2, customized=True, periodic=True)[0]
    if name == "talos_full_N2":
        return H.legged_problem(api, "talos", keep, N=2, customized=True, periodic=True)[0]
    return H.legged_problem(api, name.split("_")[0], keep, N=2)[0]


NAMES = ["kinova_N4", "talos_N2", "g1_N2", "digit_N2", "talos_full_N2", "digit_full_N2"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(oracle, kinova, name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    keep = []
    p = _problem(oracle, name, keep, kinova)
    assert list(d["block_offsets"]) == p.block_offsets
    G, J = p.eval_host(d["Z"])
    assert H.rel_err(G, d["G"]) < 1e-11 and H.rel_err(J, d["J"]) < 1e-11


def test_capsule_fixture_is_the_reference_table(oracle):
    cases = json.load(open(os.path.join(GOLD, "capsule_known_answers.json")))
    assert len(cases) == 9
    P = np.array([[c["p11"], c["p12"], c["p21"], c["p22"]] for c in cases], dtype=float)
    R = np.array([[c["r11"], c["r12"], c["r21"], c["r22"]] for c in cases], dtype=float)
    dist, _ = oracle.capsule_distance_host(P, None, R)
    assert np.all(np.abs(dist - np.array([c["distance"] for c in cases])) < 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_golden(hip, kinova, name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    keep = []
    p = _problem(hip, name, keep, kinova)
    assert list(d["block_offsets"]) == p.block_offsets
    G, J = p.eval_host(d["Z"])
    assert H.rel_err(G, d["G"]) <= TOL and H.rel_err(J, d["J"]) <= TOL
    if "q" in d.files:
        nv = d["q"].shape[-1]
        r = p.constrained_id_host(d["Z"], 2, nv, 12, derivatives=False)
        for key in ("q", "tau", "lam"):
            assert H.rel_err(r[key], d[key]) <= TOL


def _hessian_case(api, kinova):
    d = np.load(os.path.join(GOLD, "kinova_hessian_N4.npz"))
    keep = []
    p = H.kinova_problem(api, kinova, keep, N=4, boxes=H.reach_boxes(), gripper=True)
    p.set_objective(d["q_des"], int(d["tplan_n"]))
    F, GR = p.objective_host(d["Z"])
    return d, F, GR, p.eval_h_host(d["Z"], d["lam"], d["obj_factor"])


def test_oracle_reproduces_golden_hessian(oracle, kinova):
    d, F, GR, Hh = _hessian_case(oracle, kinova)
    assert H.rel_err(F, d["F"]) < 1e-12 and H.rel_err(GR, d["GRAD"]) < 1e-12 and H.rel_err(Hh, d["H"]) < 1e-11


@pytest.mark.gpu
def test_hip_matches_golden_hessian(hip, kinova):
    d, F, GR, Hh = _hessian_case(hip, kinova)
    assert H.rel_err(F, d["F"]) <= TOL and H.rel_err(GR, d["GRAD"]) <= TOL and H.rel_err(Hh, d["H"]) <= TOL
