"""The text formats either side of the hot path, as the reference's example programs read and write them.

  settings   *_optimization_settings.yaml of the single-step examples (Examples/Digit/singlestep_optimization_settings.yaml,
             parsed at Examples/Digit/DigitSingleStep.cpp:51-66, 95-104; Talos / Unitree-G1 twins)
  matrices   Utils::initializeEigenMatrixFromFile / writeEigenMatrixToFile (Utils/include/Utils.h:209-262): whitespace-separated
             rows, used for initial guesses and solutions
  solution   example_solution.txt: one decision variable per line, setprecision(20) (DigitSingleStep.cpp:204-211)
  trajectory trajectory-<robot>-Bezier-<degree>.txt: NUM_JOINTS rows of q(j)(i) over the samples, then v, a, then
             NUM_INDEPENDENT_JOINTS rows of tau and the rows of lambda (DigitSingleStep.cpp:213-244)
"""
from __future__ import annotations

import numpy as np

from . import _capi as A

# keys read by the examples' mains, with the values the shipped YAML files carry
SETTINGS_DEFAULTS = {
    "N": 14, "time_discretization": "Chebyshev", "degree": 5,
    "eps_torso_angle": 0.02, "swingfoot_midstep_z_des": 0.10,
    "swingfoot_begin_x_des": 0.0, "swingfoot_end_x_des": 0.0, "swingfoot_begin_y_des": 0.0, "swingfoot_end_y_des": 0.0,
    "tol": 1e-5, "constr_viol_tol": 1e-4, "max_wall_time": 100.0, "max_iter": 200, "obj_scaling_factor": 1e-3,
    "print_level": 5, "mu_strategy": "adaptive", "linear_solver": "ma57", "gredient_check": False,
}
_INT_KEYS = ("N", "degree", "max_iter", "print_level")
_STR_KEYS = ("time_discretization", "mu_strategy", "linear_solver", "name")


def load_settings(path_or_text, is_text: bool = False) -> dict:
    """yaml-cpp's `config[key].as<T>()` over the keys above (the reference's spelling `gredient_check` included).  Adds
    `time_discretization_code` (RAPTOR_TIME_*); an unknown name raises like DigitSingleStep.cpp:55-60 falls through to."""
    import yaml

    text = path_or_text if is_text else open(path_or_text).read()
    raw = yaml.safe_load(text) or {}
    if not isinstance(raw, dict):
        raise ValueError("settings: the YAML document is not a map")
    out = dict(SETTINGS_DEFAULTS)
    for k, v in raw.items():
        if k in _INT_KEYS:
            out[k] = int(float(v))       # print_level is read as<double>() and truncated (TalosSingleStep.cpp:87)
        elif k in _STR_KEYS:
            out[k] = str(v)
        elif k == "gredient_check":
            out[k] = v if isinstance(v, bool) else str(v).lower() == "true"
        else:
            out[k] = float(v)            # PyYAML reads `1e-5` (no dot) as a string; yaml-cpp's as<double>() parses it
    td = out["time_discretization"]
    if td == "Uniform":
        out["time_discretization_code"] = A.TIME_UNIFORM
    elif td == "Chebyshev":
        out["time_discretization_code"] = A.TIME_CHEBYSHEV
    else:
        raise ValueError(f"settings: unknown time_discretization {td!r}")
    return out


def gait_from_settings(cfg, settings: dict) -> None:
    """GaitParameters from the settings, into a robots.*_config object (DigitSingleStep.cpp:62-66, TalosSingleStep.cpp:44-47)."""
    g = cfg.gait if getattr(cfg, "gait", None) is not None else {}
    for k in ("eps_torso_angle", "swingfoot_midstep_z_des", "swingfoot_begin_x_des", "swingfoot_end_x_des",
              "swingfoot_begin_y_des", "swingfoot_end_y_des"):
        g[k] = settings[k]
    cfg.gait = g


def read_matrix(path) -> np.ndarray:
    """Utils::initializeEigenMatrixFromFile: rows of whitespace-separated numbers; the first row fixes the column count."""
    rows = []
    with open(path) as f:
        for line in f:
            vals = []
            for tok in line.split():
                try:
                    vals.append(float(tok))
                except ValueError:
                    break                # operator>> stops at the first token that is not a number
            rows.append(vals)
    if not rows:
        raise RuntimeError(f"Empty file: {path}")
    ncol = len(rows[0])
    M = np.zeros((len(rows), ncol))
    for i, r in enumerate(rows):
        if len(r) < ncol:
            raise RuntimeError(f"{path}: row {i} is shorter than the first row")
        M[i] = r[:ncol]
    return M


def _num(x: float) -> str:
    return f"{x:.20g}"  # std::setprecision(20), default float format


def write_solution(path, solution) -> None:
    with open(path, "w") as f:
        for v in np.asarray(solution, dtype=np.float64).reshape(-1):
            f.write(_num(v) + "\n")


def read_solution(path) -> np.ndarray:
    return read_matrix(path).reshape(-1)


def write_trajectory(path, q, v, a, tau, lam) -> None:
    """Arrays are [N][rows] as raptor_constrained_id_eval_host returns them for one z; the file holds one line per joint."""
    with open(path, "w") as f:
        for block in (q, v, a, tau, lam):
            block = np.asarray(block, dtype=np.float64)
            for i in range(block.shape[1]):
                f.write("".join(_num(x) + " " for x in block[:, i]) + "\n")


def dump_single_step(problem: A.ProblemHandle, z, N: int, nv: int, ni: int, solution_path, trajectory_path) -> dict:
    """What the single-step mains write after a solve.  lambda: the stance-foot contact wrench lambda.tail(6) (for Digit the
    reference also prints the 18 closed-loop multipliers, which this library does not expose: DESIGN.md section 5)."""
    r = problem.constrained_id_host(np.asarray(z, dtype=np.float64), N, nv, ni, derivatives=False)
    write_solution(solution_path, z)
    write_trajectory(trajectory_path, r["q"][0], r["v"][0], r["a"][0], r["tau"][0], r["lam"][0])
    return r
