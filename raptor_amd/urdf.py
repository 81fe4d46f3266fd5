"""Minimal URDF -> flat rigid-body model loader.

Produces exactly the fields of ``pinocchio::Model`` that the reference hot path reads
(``parents``, ``jointPlacements``, ``inertias``, joint type codes, ``gravity``,
``armature``/``damping``/``friction``; see
/root/reference/KinematicsDynamics/src/CustomizedInverseDynamics.cpp:43-59 and
/root/reference/KinematicsDynamics/src/Transform.cpp:5-54), without Pinocchio.

Conventions reproduced:
  * joint order: depth-first from the root link, children visited in alphabetical order of
    the *joint* name (urdfdom keeps joints in a name-keyed map);
  * fixed joints are merged: the child link's inertia is folded into the supporting joint's
    body and placements are composed;
  * URDF rpy -> rotation goes through a quaternion exactly like urdfdom/Eigen do, so the
    placements agree to the last bit or two with what ``pinocchio::urdf::buildModel`` holds;
  * joint type codes follow ``convertPinocchioJointType``: 1..3 = R{x,y,z}, 4..6 = P{x,y,z},
    a revolute joint about an axis that is not +x/+y/+z ("RevoluteUnaligned") is coded -3,
    as the reference does for Digit-v3.

This is host-side set-up code (runs once per model); nothing here is on the per-sample path.
"""
from __future__ import annotations

import dataclasses
import math
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence

import numpy as np


def rpy_to_matrix(rpy: Sequence[float]) -> np.ndarray:
    """urdfdom ``Rotation::setFromRPY`` followed by Eigen ``Quaternion::toRotationMatrix``."""
    phi, the, psi = rpy[0] / 2.0, rpy[1] / 2.0, rpy[2] / 2.0
    x = math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi)
    y = math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi)
    z = math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi)
    w = math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi)
    s = math.sqrt(x * x + y * y + z * z + w * w)
    if s == 0.0:
        x = y = z = 0.0
        w = 1.0
    else:
        x, y, z, w = x / s, y / s, z / s, w / s
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array(
        [
            [1.0 - (tyy + tzz), txy - twz, txz + twy],
            [txy + twz, 1.0 - (txx + tzz), tyz - twx],
            [txz - twy, tyz + twx, 1.0 - (txx + tyy)],
        ],
        dtype=np.float64,
    )


def _skew(v: np.ndarray) -> np.ndarray:
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


@dataclasses.dataclass
class _Inertia:
    """Spatial inertia as Pinocchio stores it: mass, lever (COM), 3x3 inertia about the COM."""

    mass: float
    lever: np.ndarray
    inertia: np.ndarray

    @staticmethod
    def zero() -> "_Inertia":
        return _Inertia(0.0, np.zeros(3), np.zeros((3, 3)))

    def transformed(self, R: np.ndarray, p: np.ndarray) -> "_Inertia":
        """``SE3(R,p).act(Y)``: express in the parent frame."""
        return _Inertia(self.mass, R @ self.lever + p, R @ self.inertia @ R.T)

    def __add__(self, other: "_Inertia") -> "_Inertia":
        # pinocchio::InertiaTpl::__plus__
        eps = np.finfo(np.float64).eps
        mab = self.mass + other.mass
        mab_inv = 1.0 / max(mab, eps)
        ab = self.lever - other.lever
        sk = _skew(ab)
        return _Inertia(
            mab,
            (self.mass * self.lever + other.mass * other.lever) * mab_inv,
            self.inertia + other.inertia - (self.mass * other.mass * mab_inv) * (sk @ sk),
        )


@dataclasses.dataclass
class RobotModel:
    """Flat model: index i is Pinocchio joint i+1; ``parent[i] == -1`` means the world."""

    nv: int
    names: List[str]
    parent: np.ndarray  # int32 [nv]
    jtype: np.ndarray  # int32 [nv]
    placement_R: np.ndarray  # [nv,3,3] rotation of jointPlacements (child frame -> parent frame)
    placement_p: np.ndarray  # [nv,3]
    mass: np.ndarray  # [nv]
    lever: np.ndarray  # [nv,3]
    inertia: np.ndarray  # [nv,3,3] about the COM, joint frame
    gravity: np.ndarray  # [3] linear gravity, default (0,0,-9.81) as Pinocchio
    armature: np.ndarray  # [nv]
    damping: np.ndarray  # [nv]
    friction: np.ndarray  # [nv]
    # frame placements of every link relative to its supporting joint (for end-effector offsets)
    link_frames: Dict[str, tuple] = dataclasses.field(default_factory=dict)

    def joint_id(self, name: str) -> int:
        """0-based joint index (Pinocchio ``getJointId(name) - 1``)."""
        return self.names.index(name)

    def copy(self) -> "RobotModel":
        return dataclasses.replace(
            self,
            names=list(self.names),
            parent=self.parent.copy(),
            jtype=self.jtype.copy(),
            placement_R=self.placement_R.copy(),
            placement_p=self.placement_p.copy(),
            mass=self.mass.copy(),
            lever=self.lever.copy(),
            inertia=self.inertia.copy(),
            gravity=self.gravity.copy(),
            armature=self.armature.copy(),
            damping=self.damping.copy(),
            friction=self.friction.copy(),
            link_frames=dict(self.link_frames),
        )

    def total_mass(self) -> float:
        return float(self.mass.sum())


def _vec(attr: Optional[str], default=(0.0, 0.0, 0.0)) -> np.ndarray:
    if attr is None:
        return np.array(default, dtype=np.float64)
    return np.array([float(t) for t in attr.split()], dtype=np.float64)


def _axis_to_jtype(kind: str, axis: np.ndarray) -> int:
    base = 0 if kind in ("revolute", "continuous") else 3
    if np.array_equal(axis, [1.0, 0.0, 0.0]):
        return base + 1
    if np.array_equal(axis, [0.0, 1.0, 0.0]):
        return base + 2
    if np.array_equal(axis, [0.0, 0.0, 1.0]):
        return base + 3
    if base == 0 and np.array_equal(axis, [0.0, 0.0, -1.0]):
        # "RevoluteUnaligned" in Pinocchio; Transform.cpp:22-29 codes it -3 (Digit-v3 only)
        return -3
    raise ValueError(f"unsupported joint axis {axis} for a {kind} joint")


def load_urdf(path: str) -> RobotModel:
    root = ET.parse(path).getroot()

    link_inertia: Dict[str, _Inertia] = {}
    for link in root.findall("link"):
        name = link.get("name")
        inertial = link.find("inertial")
        if inertial is None:
            link_inertia[name] = _Inertia.zero()
            continue
        origin = inertial.find("origin")
        xyz = _vec(origin.get("xyz")) if origin is not None else np.zeros(3)
        rpy = _vec(origin.get("rpy")) if origin is not None else np.zeros(3)
        mass_el = inertial.find("mass")
        mass = float(mass_el.get("value")) if mass_el is not None else 0.0
        I_el = inertial.find("inertia")
        I = np.zeros((3, 3))
        if I_el is not None:
            ixx, ixy, ixz = float(I_el.get("ixx", 0)), float(I_el.get("ixy", 0)), float(I_el.get("ixz", 0))
            iyy, iyz, izz = float(I_el.get("iyy", 0)), float(I_el.get("iyz", 0)), float(I_el.get("izz", 0))
            I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        R = rpy_to_matrix(rpy)
        link_inertia[name] = _Inertia(mass, xyz, R @ I @ R.T)

    joints = {}
    children_of: Dict[str, List[str]] = {}
    child_links = set()
    for j in root.findall("joint"):
        if j.get("type") is None:  # e.g. <transmission><joint name=..>
            continue
        name = j.get("name")
        parent = j.find("parent").get("link")
        child = j.find("child").get("link")
        origin = j.find("origin")
        xyz = _vec(origin.get("xyz")) if origin is not None else np.zeros(3)
        rpy = _vec(origin.get("rpy")) if origin is not None else np.zeros(3)
        axis_el = j.find("axis")
        axis = _vec(axis_el.get("xyz")) if axis_el is not None else np.array([1.0, 0.0, 0.0])
        dyn = j.find("dynamics")
        damping = float(dyn.get("damping", 0.0)) if dyn is not None else 0.0
        friction = float(dyn.get("friction", 0.0)) if dyn is not None else 0.0
        joints[name] = dict(
            name=name, type=j.get("type"), parent=parent, child=child,
            R=rpy_to_matrix(rpy), p=xyz, axis=axis, damping=damping, friction=friction,
        )
        children_of.setdefault(parent, []).append(name)
        child_links.add(child)

    roots = [l for l in link_inertia if l not in child_links]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")

    names: List[str] = []
    parent_idx: List[int] = []
    jtype: List[int] = []
    plR: List[np.ndarray] = []
    plp: List[np.ndarray] = []
    body: List[_Inertia] = []
    damping: List[float] = []
    friction: List[float] = []
    link_frames: Dict[str, tuple] = {}

    def visit(link: str, joint_idx: int, R_l: np.ndarray, p_l: np.ndarray) -> None:
        """``link`` is rigidly attached to joint ``joint_idx`` (-1: world) at placement (R_l,p_l)."""
        link_frames[link] = (joint_idx, R_l.copy(), p_l.copy())
        for jn in sorted(children_of.get(link, [])):
            j = joints[jn]
            # placement of the joint frame in the supporting joint's frame
            R_j = R_l @ j["R"]
            p_j = R_l @ j["p"] + p_l
            if j["type"] == "fixed":
                if joint_idx >= 0:
                    body[joint_idx] = body[joint_idx] + link_inertia[j["child"]].transformed(R_j, p_j)
                visit(j["child"], joint_idx, R_j, p_j)
            elif j["type"] in ("revolute", "continuous", "prismatic"):
                idx = len(names)
                names.append(jn)
                parent_idx.append(joint_idx)
                jtype.append(_axis_to_jtype(j["type"], j["axis"]))
                plR.append(R_j)
                plp.append(p_j)
                body.append(link_inertia[j["child"]])
                damping.append(j["damping"])
                friction.append(j["friction"])
                visit(j["child"], idx, np.eye(3), np.zeros(3))
            else:
                raise ValueError(f"unsupported joint type {j['type']} ({jn})")

    visit(roots[0], -1, np.eye(3), np.zeros(3))

    nv = len(names)
    return RobotModel(
        nv=nv,
        names=names,
        parent=np.array(parent_idx, dtype=np.int32),
        jtype=np.array(jtype, dtype=np.int32),
        placement_R=np.array(plR, dtype=np.float64).reshape(nv, 3, 3),
        placement_p=np.array(plp, dtype=np.float64).reshape(nv, 3),
        mass=np.array([b.mass for b in body], dtype=np.float64),
        lever=np.array([b.lever for b in body], dtype=np.float64).reshape(nv, 3),
        inertia=np.array([b.inertia for b in body], dtype=np.float64).reshape(nv, 3, 3),
        gravity=np.array([0.0, 0.0, -9.81]),
        armature=np.zeros(nv),
        damping=np.array(damping, dtype=np.float64),
        friction=np.array(friction, dtype=np.float64),
        link_frames=link_frames,
    )


def jtype_transform(jt: int, q: float):
    """``Transform(jtype, theta)`` of the reference (Transform.cpp:107-159): (R, p)."""
    c, s = math.cos(q), math.sin(q)
    R = np.eye(3)
    p = np.zeros(3)
    a = abs(jt)
    sg = 1.0 if jt > 0 else -1.0
    if a == 1:
        R = np.array([[1, 0, 0], [0, c, -sg * s], [0, sg * s, c]], dtype=np.float64)
    elif a == 2:
        R = np.array([[c, 0, sg * s], [0, 1, 0], [-sg * s, 0, c]], dtype=np.float64)
    elif a == 3:
        R = np.array([[c, -sg * s, 0], [sg * s, c, 0], [0, 0, 1]], dtype=np.float64)
    elif a in (4, 5, 6):
        p[a - 4] = sg * q
    else:
        raise ValueError("jtype 0 has no transform")
    return R, p


def lock_joints(model: RobotModel, lock_names: Sequence[str], q_lock: Dict[str, float]) -> RobotModel:
    """``pinocchio::buildReducedModel``: freeze the named joints at the given positions.

    A locked joint becomes a fixed placement ``jointPlacement * Transform(jtype, q)``; its body
    is folded into the nearest unlocked ancestor (Unitree-G1: G1SingleStep.cpp:18-38).
    """
    lock = set(lock_names)
    nv = model.nv
    new_index = [-1] * nv
    # placement of (old) joint i's frame relative to its supporting *kept* joint frame
    sup = [-1] * nv
    relR = [np.eye(3)] * nv
    relp = [np.zeros(3)] * nv

    names: List[str] = []
    parent: List[int] = []
    jt: List[int] = []
    plR: List[np.ndarray] = []
    plp: List[np.ndarray] = []
    body: List[_Inertia] = []
    arm: List[float] = []
    dmp: List[float] = []
    fric: List[float] = []

    for i in range(nv):
        par = int(model.parent[i])
        if par < 0:
            sR, sp, s = np.eye(3), np.zeros(3), -1
        elif new_index[par] >= 0:
            sR, sp, s = np.eye(3), np.zeros(3), new_index[par]
        else:
            sR, sp, s = relR[par], relp[par], sup[par]
        R_j = sR @ model.placement_R[i]
        p_j = sR @ model.placement_p[i] + sp
        Y = _Inertia(float(model.mass[i]), model.lever[i].copy(), model.inertia[i].copy())
        if model.names[i] in lock:
            Rq, pq = jtype_transform(int(model.jtype[i]), float(q_lock.get(model.names[i], 0.0)))
            R_f = R_j @ Rq
            p_f = R_j @ pq + p_j
            sup[i], relR[i], relp[i] = s, R_f, p_f
            if s >= 0:
                body[s] = body[s] + Y.transformed(R_f, p_f)
        else:
            new_index[i] = len(names)
            names.append(model.names[i])
            parent.append(s)
            jt.append(int(model.jtype[i]))
            plR.append(R_j)
            plp.append(p_j)
            body.append(Y)
            arm.append(float(model.armature[i]))
            dmp.append(float(model.damping[i]))
            fric.append(float(model.friction[i]))

    n = len(names)
    link_frames = {}
    for link, (ji, R_l, p_l) in model.link_frames.items():
        if ji < 0:
            link_frames[link] = (ji, R_l, p_l)
        elif new_index[ji] >= 0:
            link_frames[link] = (new_index[ji], R_l, p_l)
        else:
            link_frames[link] = (sup[ji], relR[ji] @ R_l, relR[ji] @ p_l + relp[ji])
    return RobotModel(
        nv=n,
        names=names,
        parent=np.array(parent, dtype=np.int32),
        jtype=np.array(jt, dtype=np.int32),
        placement_R=np.array(plR).reshape(n, 3, 3),
        placement_p=np.array(plp).reshape(n, 3),
        mass=np.array([b.mass for b in body]),
        lever=np.array([b.lever for b in body]).reshape(n, 3),
        inertia=np.array([b.inertia for b in body]).reshape(n, 3, 3),
        gravity=model.gravity.copy(),
        armature=np.array(arm),
        damping=np.array(dmp),
        friction=np.array(fric),
        link_frames=link_frames,
    )
