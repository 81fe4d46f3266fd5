"""Robot tables of the reference examples that reach the hot path (the *Constants.h / *DynamicsConstraints.h
files and the model set-up of the example mains), as plain data for `ProblemHandle.legged`."""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np

from .urdf import RobotModel, load_urdf, lock_joints


@dataclasses.dataclass
class LeggedConfig:
    dependent_ids: List[int]
    independent_ids: List[int]
    extra_variables: int
    fk_start_joint: int      # Pinocchio joint id (1-based) of "Rz"
    contact_joint_id: int    # Pinocchio joint id (1-based) of the stance foot joint
    stance_foot_endT_p: Sequence[float]
    stance_foot_T_des_R: np.ndarray
    stance_foot_T_des_p: Sequence[float]
    torque_limits_lower: np.ndarray
    torque_limits_upper: np.ndarray
    joint_limits_lower: np.ndarray
    joint_limits_upper: np.ndarray
    mu: float
    gamma: float
    Lx: float
    Ly: float
    degree: int = 6
    T: float = 0.4
    N: int = 14
    stance_leg: str = "L"
    gait: Optional[dict] = None   # GaitParameters + swing foot (None: customized block left out)
    period: Optional[dict] = None  # reset-map block (None: left out)


# ---- Talos: Examples/Talos/include/TalosConstants.h:7-96, TalosDynamicsConstraints.h:14-36 ----
_TALOS_DEP = ["Px", "Py", "Pz", "Rx", "Ry", "Rz"]
_TALOS_IND = [f"leg_{s}_{k}_joint" for s in ("left", "right") for k in range(1, 7)]
_TALOS_JL_LO = [-1000] * 6 + [-0.349065850399, -0.5236, -2.095, 0, -1.309, -0.5236,
                              -1.57079632679, -0.5236, -2.095, 0, -1.309, -0.5236]
_TALOS_JL_UP = [1000] * 6 + [1.57079632679, 0.5236, 0.7, 2.618, 0.768, 0.5236,
                             0.349065850399, 0.5236, 0.7, 2.618, 0.768, 0.5236]
_TALOS_TL = [100, 160, 160, 300, 160, 100] * 2


def talos_model(urdf_path: str) -> RobotModel:
    """Examples/Talos/TalosSingleStep.cpp:16-25: buildModel, motor dynamics zeroed (gravity stays Pinocchio's -9.81)."""
    m = load_urdf(urdf_path)
    m.armature[:] = 0
    m.damping[:] = 0
    m.friction[:] = 0
    return m


def talos_config(m: RobotModel, stance_leg: str = "L") -> LeggedConfig:
    foot = "leg_left_6_joint" if stance_leg in "Ll" else "leg_right_6_joint"
    return LeggedConfig(
        dependent_ids=[m.joint_id(n) for n in _TALOS_DEP],
        independent_ids=[m.joint_id(n) for n in _TALOS_IND],
        extra_variables=18 + 6,                      # TalosSingleStepOptimizer.cpp:50-51
        fk_start_joint=m.joint_id("Rz") + 1,
        contact_joint_id=m.joint_id(foot) + 1,
        stance_foot_endT_p=[0.0, 0.0, -0.107],       # TalosDynamicsConstraints.cpp:34,38
        stance_foot_T_des_R=np.eye(3), stance_foot_T_des_p=[0.0, 0.0, 0.0],
        torque_limits_lower=-np.array(_TALOS_TL, dtype=float), torque_limits_upper=np.array(_TALOS_TL, dtype=float),
        joint_limits_lower=np.array(_TALOS_JL_LO, dtype=float), joint_limits_upper=np.array(_TALOS_JL_UP, dtype=float),
        mu=0.7, gamma=0.7, Lx=0.1344, Ly=0.2208,     # MU, GAMMA, FOOT_WIDTH, FOOT_LENGTH
        stance_leg="L" if stance_leg in "Ll" else "R",
    )


def talos_gait(m: RobotModel, stance_leg: str = "L") -> dict:
    """TalosCustomizedConstraints.h:12-19 GaitParameters defaults, .cpp:17-18 end transform, :191 torso height."""
    swing = "leg_right_6_joint" if stance_leg in "Ll" else "leg_left_6_joint"
    return dict(swing_joint_id=m.joint_id(swing) + 1, swingfoot_endT_p=(0.0, 0.0, -0.107), eps_torso_angle=np.deg2rad(1.5),
                swingfoot_midstep_z_des=0.10, swingfoot_begin_x_des=0.0, swingfoot_begin_y_des=-0.17, swingfoot_end_x_des=0.0,
                swingfoot_end_y_des=-0.17, torso_height_lb=0.95)


def talos_period(m: RobotModel, stance_leg: str = "L") -> dict:
    """TalosSingleStepPeriodicityConstraints.h:19-32 JOINT_MAP, TalosDynamicsConstraints.cpp:44-62 reinitialize()."""
    swing = "leg_right_6_joint" if stance_leg in "Ll" else "leg_left_6_joint"
    id1 = [m.joint_id(n) for n in _TALOS_IND]
    return dict(swing_contact_joint_id=m.joint_id(swing) + 1, swing_foot_endT_p=(0.0, 0.0, -0.107),
                joint_id1=id1, joint_id2=id1[6:] + id1[:6], mirror_sign=-1.0)


# ---- Unitree-G1: Examples/Unitree-G1/include/G1Constants.h, G1DynamicsConstraints.h:14-36, G1SingleStep.cpp:18-43 ----
_G1_DEP = ["Px_joint", "Py_joint", "Pz_joint", "Rx_joint", "Ry_joint", "Rz_joint"]
_G1_IND = [f"{s}_{n}_joint" for s in ("left", "right")
           for n in ("hip_pitch", "hip_roll", "hip_yaw", "knee", "ankle_pitch", "ankle_roll")]


def g1_model(urdf_path: str) -> RobotModel:
    """G1SingleStep.cpp:18-43.  The ten arm joints (Pinocchio ids 19..28) are locked with buildReducedModel at
    q_armfixed; the example fills q_armfixed(12),(13),(15),(17),(18),(20), but only entries of LOCKED joints
    (indices 18..27) matter, i.e. q(18) = -0.2 and q(20) = 0.3 -- reproduced as the code does it, not as its
    comments say."""
    full = load_urdf(urdf_path)
    q_armfixed = np.zeros(full.nv)
    q_armfixed[[12, 13, 15, 17, 18, 20]] = [0.6, 0.2, 0.3, 0.6, -0.2, 0.3]
    lock = [full.names[i] for i in range(18, 28)]
    m = lock_joints(full, lock, {full.names[i]: float(q_armfixed[i]) for i in range(18, 28)})
    m.armature[:] = 0
    m.damping[:] = 0
    m.friction[:] = 0
    return m


def _g1_constants(path_hint: str = ""):
    # G1Constants.h:12-83.  JOINT_LIMITS_LOWER is missing a comma in the reference ("-0.87267 -0.2618" is one
    # expression), so it has 17 initialisers and the 18th is value-initialised to 0 (SURVEY.md 8a note).
    lo = [-1000] * 6 + [-2.5307, -0.5236, -2.7576, -0.087267, -0.87267 - 0.2618,
                        -2.5307, -2.9671, -2.7576, -0.087267, -0.87267, -0.2618, 0.0]
    up = [1000] * 6 + [2.8798, 2.9671, 2.7576, 2.8798, 0.5236, 0.2618,
                       2.8798, 0.5236, 2.7576, 2.8798, 0.5236, 0.2618]
    tl = [88, 88, 88, 139, 50, 50] * 2
    return lo, up, tl


def g1_config(m: RobotModel, stance_leg: str = "L") -> LeggedConfig:
    lo, up, tl = _g1_constants()
    foot = "left_ankle_roll_joint" if stance_leg in "Ll" else "right_ankle_roll_joint"
    return LeggedConfig(
        dependent_ids=[m.joint_id(n) for n in _G1_DEP],
        independent_ids=[m.joint_id(n) for n in _G1_IND],
        extra_variables=18 + 6,
        fk_start_joint=m.joint_id("Rz_joint") + 1,
        contact_joint_id=m.joint_id(foot) + 1,
        stance_foot_endT_p=[0.035, 0.0, -0.03],      # G1DynamicsConstraints.cpp:34,38
        stance_foot_T_des_R=np.eye(3), stance_foot_T_des_p=[0.0, 0.0, 0.0],
        torque_limits_lower=-np.array(tl, dtype=float), torque_limits_upper=np.array(tl, dtype=float),
        joint_limits_lower=np.array(lo, dtype=float), joint_limits_upper=np.array(up, dtype=float),
        mu=0.7, gamma=0.7, Lx=0.10, Ly=0.03,
        stance_leg="L" if stance_leg in "Ll" else "R",
    )


def g1_gait(m: RobotModel, stance_leg: str = "L") -> dict:
    """G1CustomizedConstraints.h:12-19, .cpp:17-18, :185."""
    swing = "right_ankle_roll_joint" if stance_leg in "Ll" else "left_ankle_roll_joint"
    return dict(swing_joint_id=m.joint_id(swing) + 1, swingfoot_endT_p=(0.035, 0.0, -0.03), eps_torso_angle=np.deg2rad(1.0),
                swingfoot_midstep_z_des=0.10, swingfoot_begin_x_des=0.0, swingfoot_begin_y_des=-0.237013, swingfoot_end_x_des=0.0,
                swingfoot_end_y_des=-0.237013, torso_height_lb=0.55)


def g1_period(m: RobotModel, stance_leg: str = "L") -> dict:
    """G1SingleStepPeriodicityConstraints.h:19-32 JOINT_MAP, G1DynamicsConstraints.cpp reinitialize()."""
    swing = "right_ankle_roll_joint" if stance_leg in "Ll" else "left_ankle_roll_joint"
    id1 = [m.joint_id(n) for n in _G1_IND]
    return dict(swing_contact_joint_id=m.joint_id(swing) + 1, swing_foot_endT_p=(0.035, 0.0, -0.03),
                joint_id1=id1, joint_id2=id1[6:] + id1[:6], mirror_sign=-1.0)


# ---- Digit-v3: Examples/Digit/include/DigitConstants.h, DigitDynamicsConstraints.h:19-57, DigitSingleStep.cpp:16-40 ----
DIGIT_LEG_JOINTS = ["hip_roll", "hip_yaw", "hip_pitch", "achilles_rod", "ach2", "knee", "tarsus", "toe_A", "toe_A_rod",
                    "A2", "toe_B", "toe_B_rod", "B2", "toe_pitch", "toe_roll"]
_DIGIT_ARMATURE = {"hip_roll": 0.173823936, "hip_yaw": 0.067899975, "hip_pitch": 0.1204731904, "knee": 0.1204731904,
                   "toe_A": 0.036089475, "toe_B": 0.036089475}
# degrees (DigitConstants.h:13-91)
_DIGIT_JL_LO = [-1000] * 6 + [-60, -40, -60, -180, -10, -16.5480 - 15, -50.3, -26.8765 - 15, -180, -10, -1.1295 - 15, -180, -10, -44, -37,
                              -60, -40, -90, -180, -10, -58.4, -71.6, -18.2531 - 15, -180, -10, -26.2724 - 15, -180, -10, -34, -33]
_DIGIT_JL_UP = [1000] * 6 + [60, 40, 90, 180, 10, 58.4, 71.6, 0.4032 + 15, 180, 10, 26.2183 + 15, 180, 10, 34, 33,
                             60, 40, 60, 180, 10, 30.7455 + 15, 50.3, 27.0323 + 15, 180, 10, 14.6109 + 15, 180, 10, 44, 37]
_DIGIT_TL_LO = [-113.000, -79.1765, -216.928, -231.317, -41.9759, -41.9759, -126.682, -79.1765, -216.928, -231.317, -41.9759, -41.9759]
_DIGIT_TL_UP = [113.000, 79.1765, 216.928, 231.317, 41.9759, 41.9759, 126.682, 79.1765, 216.928, 231.317, 41.9759, 41.9759]


@dataclasses.dataclass
class DigitConfig:
    base_ids: List[int]
    leg_ids: List[List[int]]
    stance_leg: str
    extra_variables: int
    stance_foot_T_des_R: np.ndarray
    stance_foot_T_des_p: Sequence[float]
    torque_limits_lower: np.ndarray
    torque_limits_upper: np.ndarray
    joint_limits_lower: np.ndarray
    joint_limits_upper: np.ndarray
    mu: float = 0.7
    gamma: float = 0.7
    Lx: float = 0.04     # FOOT_WIDTH
    Ly: float = 0.1175   # FOOT_LENGTH
    degree: int = 5
    T: float = 0.4
    N: int = 14
    gait: Optional[dict] = None   # GaitParameters (None: customized block left out)
    period: Optional[dict] = None  # reset-map block: {} enables it (everything is derived from the stance leg)

    @property
    def independent_ids(self):
        return [self.leg_ids[leg][DIGIT_LEG_JOINTS.index(n)] for leg in range(2)
                for n in ("hip_roll", "hip_yaw", "hip_pitch", "knee", "toe_A", "toe_B")]

    @property
    def dependent_ids(self):
        return list(self.base_ids) + [self.leg_ids[leg][DIGIT_LEG_JOINTS.index(n)] for leg in range(2)
                                      for n in ("achilles_rod", "ach2", "tarsus", "toe_A_rod", "A2", "toe_B_rod", "B2",
                                                "toe_pitch", "toe_roll")]


def digit_model(urdf_path: str) -> RobotModel:
    """DigitSingleStep.cpp:16-40: gravity -9.806, friction zeroed, motor inertias on the twelve actuated joints."""
    m = load_urdf(urdf_path)
    m.gravity = np.array([0.0, 0.0, -9.806])
    m.friction[:] = 0
    for side in ("left", "right"):
        for n, val in _DIGIT_ARMATURE.items():
            m.armature[m.joint_id(f"{side}_{n}")] = val
    return m


def digit_config(m: RobotModel, stance_leg: str = "L") -> DigitConfig:
    c, s = np.cos(-np.pi / 2), np.sin(-np.pi / 2)
    Rdes = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])  # Transform(3, -M_PI_2), DigitSingleStepOptimizer.h:50
    return DigitConfig(
        base_ids=[m.joint_id(n) for n in ("Px", "Py", "Pz", "Rx", "Ry", "Rz")],
        leg_ids=[[m.joint_id(f"{side}_{n}") for n in DIGIT_LEG_JOINTS] for side in ("left", "right")],
        stance_leg=stance_leg,
        extra_variables=36 + 24,  # DigitSingleStepOptimizer.cpp:50-51
        stance_foot_T_des_R=Rdes, stance_foot_T_des_p=[0.0, 0.0, 0.0],
        torque_limits_lower=np.array(_DIGIT_TL_LO), torque_limits_upper=np.array(_DIGIT_TL_UP),
        joint_limits_lower=np.deg2rad(np.array(_DIGIT_JL_LO, dtype=float)),
        joint_limits_upper=np.deg2rad(np.array(_DIGIT_JL_UP, dtype=float)),
    )


def digit_gait() -> dict:
    """singlestep_optimization_settings.yaml:13-22 over the GaitParameters defaults (DigitCustomizedConstraints.h:12-19)."""
    return dict(eps_torso_angle=0.02, swingfoot_midstep_z_des=0.10, swingfoot_begin_x_des=-0.22, swingfoot_begin_y_des=0.0,
                swingfoot_end_x_des=-0.22, swingfoot_end_y_des=0.0)
