"""Generates tests/golden/*.npz: small input/output vectors of the hot path.

The reference holds no golden vectors at fixed z for this path and cannot be built or imported here, so
these are REGRESSION vectors produced by the CPU oracle in this container (fixed seeds), after the oracle was pinned
by tests/test_oracle_pins.py and tests/test_oracle_legged.py.  The only reference-held answers, the nine
TaperedCapsuleCollision cases, are stored verbatim in capsule_known_answers.json.
    python scripts/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import raptor_amd
from raptor_amd import _capi as A
from test_oracle_pins import CAPSULE_KNOWN

OUT = os.path.join(ROOT, "tests", "golden")


def reference_two_step_z():
    """The one data file of the reference that belongs to this path: the two-step Digit decision vector of
    Examples/Digit/data/initial-digit-two-step-5.txt (264 numbers, read as text), stored verbatim with its settings.  It turns out
    to be a converged periodic two-step gait: tests/test_reference_solution_pin.py checks that it satisfies this repository's
    equality rows to the reference solver's tolerance."""
    src = "/root/reference/Examples/Digit/data/initial-digit-two-step-5.txt"
    if not os.path.exists(src):
        return
    z = np.loadtxt(src)
    dst = os.path.join(OUT, "digit_two_step_reference_z.json")
    old = json.load(open(dst)) if os.path.exists(dst) else {}
    old["z"] = [repr(float(x)) for x in z]
    json.dump(old, open(dst, "w"), indent=0)


def kinova_hessian(oracle, kin, keep):
    """KinovaOptimizer objective + Optimizer::eval_h at fixed (z, lambda, obj_factor): regression vectors of the oracle after
    tests/test_oracle_hessian.py pinned it with the reference's finite-difference recipes."""
    p = H.kinova_problem(oracle, kin, keep, N=4, boxes=H.reach_boxes(), gripper=True)
    rng = np.random.default_rng(20250523 + 7)
    q_des = rng.uniform(-3, 3, 7)
    p.set_objective(q_des, 2)
    Z = rng.uniform(-1, 1, (2, 7))
    lam = rng.uniform(-1, 1, (2, p.m))
    of = rng.uniform(0.5, 2.0, 2)
    F, GR = p.objective_host(Z)
    np.savez(os.path.join(OUT, "kinova_hessian_N4.npz"), Z=Z, lam=lam, obj_factor=of, q_des=q_des, tplan_n=np.array(2),
             F=F, GRAD=GR, H=p.eval_h_host(Z, lam, of))


def main():
    os.makedirs(OUT, exist_ok=True)
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    keep = []
    kin = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
    kin.gravity = np.array([0.0, 0.0, -9.81])
    kinova_hessian(oracle, kin, keep)
    if "--hessian-only" in sys.argv:  # added after the other fixtures were committed: leaves them untouched
        return
    p = H.kinova_problem(oracle, kin, keep, N=4, boxes=H.reach_boxes())
    Z = np.random.default_rng(20250523).uniform(-1, 1, (2, 7))
    G, J = p.eval_host(Z)
    np.savez(os.path.join(OUT, "kinova_N4.npz"), Z=Z, G=G, J=J, block_offsets=np.array(p.block_offsets))
    for which in ("talos", "g1"):
        p, m, cfg = H.legged_problem(oracle, which, keep, N=2)
        Z = 0.1 * np.random.default_rng(20250523 + len(which)).uniform(-1, 1, (1, p.n))
        G, J = p.eval_host(Z)
        r = p.constrained_id_host(Z, 2, m.nv, 12)
        np.savez(os.path.join(OUT, f"{which}_N2.npz"), Z=Z, G=G, J=J, q=r["q"], tau=r["tau"], lam=r["lam"],
                 block_offsets=np.array(p.block_offsets))
    p, m, cfg = H.digit_problem(oracle, keep, N=2)
    Z = 0.1 * np.random.default_rng(20250523 + 36).uniform(-1, 1, (1, p.n))
    G, J = p.eval_host(Z)
    r = p.constrained_id_host(Z, 2, 36, 12)
    np.savez(os.path.join(OUT, "digit_N2.npz"), Z=Z, G=G, J=J, q=r["q"], tau=r["tau"], lam=r["lam"],
             block_offsets=np.array(p.block_offsets))
    # the complete single-step problems (customized + reset-map blocks on)
    for which in ("talos", "digit"):
        if which == "digit":
            p, m, cfg = H.digit_problem(oracle, keep, N=2, customized=True, periodic=True)
        else:
            p, m, cfg = H.legged_problem(oracle, which, keep, N=2, customized=True, periodic=True)
        Z = 0.1 * np.random.default_rng(20250523 + 100 + len(which)).uniform(-1, 1, (1, p.n))
        Z[:, -6:] *= 30.0
        G, J = p.eval_host(Z)
        np.savez_compressed(os.path.join(OUT, f"{which}_full_N2.npz"), Z=Z, G=G, J=J, block_offsets=np.array(p.block_offsets))
    json.dump([dict(p11=c[0], p12=c[1], p21=c[2], p22=c[3], r11=c[4], r12=c[5], r21=c[6], r22=c[7], distance=c[8])
               for c in CAPSULE_KNOWN], open(os.path.join(OUT, "capsule_known_answers.json"), "w"), indent=1)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    reference_two_step_z()
    main()
