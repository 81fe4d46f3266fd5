"""Text formats of the reference's example programs (raptor_amd/formats.py): settings YAML, matrix / solution / trajectory files."""
import numpy as np
import pytest

from raptor_amd import _capi as A
from raptor_amd import formats as F

import helpers as H

# same keys and value spellings as Examples/Talos/singlestep_optimization_settings.yaml (values changed)
SETTINGS = """
name: TalosSingleStep
N: 18
time_discretization: Uniform # Chebyshev, Uniform
degree: 6
eps_torso_angle: 0.0524
swingfoot_midstep_z_des: 0.15
swingfoot_begin_x_des: -0.22
swingfoot_end_x_des: 0.22
tol: 1e-5
constr_viol_tol: 1e-4
max_wall_time: 100.0
max_iter: 200
obj_scaling_factor: 1e-3
print_level: 5 # detailed
mu_strategy: monotone
linear_solver: ma57
gredient_check: false
"""


def test_settings():
    s = F.load_settings(SETTINGS, is_text=True)
    assert s["N"] == 18 and s["degree"] == 6 and s["time_discretization_code"] == A.TIME_UNIFORM
    assert s["tol"] == 1e-5 and s["obj_scaling_factor"] == 1e-3 and s["max_iter"] == 200 and s["gredient_check"] is False
    assert s["swingfoot_begin_x_des"] == -0.22 and s["swingfoot_begin_y_des"] == 0.0  # absent key: the default
    assert s["mu_strategy"] == "monotone"
    with pytest.raises(ValueError):
        F.load_settings("time_discretization: Gauss", is_text=True)


def test_matrix_and_solution_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    z = rng.uniform(-1, 1, 17) * 10.0 ** rng.integers(-12, 6, 17)
    p = tmp_path / "example_solution.txt"
    F.write_solution(p, z)
    assert np.array_equal(F.read_solution(p), z)  # setprecision(20) round-trips binary64
    assert len(p.read_text().splitlines()) == 17
    (tmp_path / "m.txt").write_text("1 2 3\n4 5 6 extra\n")
    assert np.array_equal(F.read_matrix(tmp_path / "m.txt"), [[1, 2, 3], [4, 5, 6]])
    (tmp_path / "e.txt").write_text("")
    with pytest.raises(RuntimeError):
        F.read_matrix(tmp_path / "e.txt")


def test_single_step_dump(oracle, tmp_path):
    """trajectory-*.txt layout (DigitSingleStep.cpp:213-244): nv rows of q, v, a over the samples, Nact rows of tau, lambda rows."""
    keep = []
    p, m, cfg = H.legged_problem(oracle, "g1", keep, N=4)
    z = 0.1 * np.random.default_rng(1).uniform(-1, 1, p.n)
    r = F.dump_single_step(p, z, 4, m.nv, 12, tmp_path / "example_solution.txt", tmp_path / "trajectory-g1-Bezier-6.txt")
    M = F.read_matrix(tmp_path / "trajectory-g1-Bezier-6.txt")
    assert M.shape == (3 * 18 + 12 + 6, 4)
    assert np.array_equal(M[:18], r["q"][0].T) and np.array_equal(M[54:66], r["tau"][0].T) and np.array_equal(M[66:], r["lam"][0].T)
    assert np.array_equal(F.read_solution(tmp_path / "example_solution.txt"), z)
