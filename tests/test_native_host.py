"""A compiled C++ caller of the C ABI (the host shell's native side).

`tests/native/kinova_host_test.cpp` is the adapter of INTEGRATION.md section 2 behind RAPTOR's `Constraints` interface (same member
names, Constraints.h:26-57) with the loops of `Optimizer::eval_g / eval_jac_g` (Optimizer.cpp:361-385, 467-486) on top; Eigen and
Pinocchio are absent from the image, so the members sit on std::vector.  Here: it compiles and links against
`include/raptor_hip.h` / `libraptor_hip.so` with plain g++ (CPU test), and on the GPU its `g` / `values` for several z are compared
bit for bit with the ctypes route, together with the per-block feasibility report under the reference's four block names."""
import os
import struct
import subprocess

import numpy as np
import pytest

import raptor_amd
from raptor_amd import _capi as A

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "kinova_host_test.cpp")
EXE = os.path.join(ROOT, "tests", "native", "kinova_host_test")


def build_native(out=EXE):
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-L", os.path.join(ROOT, "raptor_amd"),
           "-lraptor_hip", "-Wl,-rpath,$ORIGIN/../../raptor_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", out]
    subprocess.run(cmd, check=True, cwd=ROOT)
    return out


def test_native_caller_compiles_and_links(tmp_path):
    assert os.path.exists(raptor_amd.LIB_PATH), "build with python -m raptor_amd.build"
    exe = build_native(str(tmp_path / "kinova_host_test"))
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raptor_amd") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, env=env)  # no arguments: usage, before any HIP call
    assert r.returncode == 2 and "usage" in r.stderr


def _write_input(path, model, T, N, q0, qd0, qdd0, boxes, gripper, buffer, Z):
    c, rpy, size = boxes
    nv = model.nv
    with open(path, "wb") as f:
        f.write(struct.pack("i", nv))
        f.write(np.asarray(model.parent, dtype=np.int32).tobytes())
        f.write(np.asarray(model.jtype, dtype=np.int32).tobytes())
        for a in (model.placement_R, model.placement_p, model.mass, model.lever, model.inertia, model.gravity,
                  model.armature, model.damping, model.friction):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        f.write(struct.pack("d", T))
        f.write(struct.pack("i", N))
        for a in (q0, qd0, qdd0):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        f.write(struct.pack("i", c.shape[0]))
        for a in (c, rpy, size):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        f.write(struct.pack("i", int(gripper)))
        f.write(struct.pack("d", buffer))
        f.write(struct.pack("i", Z.shape[0]))
        f.write(np.ascontiguousarray(Z, dtype=np.float64).tobytes())


@pytest.mark.gpu
def test_native_caller_matches_ctypes(hip, kinova, tmp_path):
    if not os.path.exists(EXE):
        build_native()
    N, T = 16, 1.0
    rng = np.random.default_rng(20250523)
    q0, qd0, qdd0 = rng.uniform(-0.5, 0.5, 7), rng.uniform(-0.3, 0.3, 7), rng.uniform(-0.2, 0.2, 7)
    boxes = H.reach_boxes()
    Z = rng.uniform(-1, 1, (5, 7))
    Z[1] *= 0.01  # a nearly stationary arm: the limit blocks are feasible, the obstacle block need not be
    inp, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write_input(inp, kinova, T, N, q0, qd0, qdd0, boxes, True, 0.02, Z)
    r = subprocess.run([EXE, inp, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    keep = []
    p = H.kinova_problem(hip, kinova, keep, N=N, T=T, q0=q0, qd0=qd0, qdd0=qdd0, boxes=boxes, gripper=True, buffer=0.02)
    raw = open(out, "rb").read()
    m, n = struct.unpack_from("ii", raw, 0)
    assert (m, n) == (p.m, p.n)
    off = 8
    g_l = np.frombuffer(raw, np.float64, m, off); off += 8 * m
    g_u = np.frombuffer(raw, np.float64, m, off); off += 8 * m
    lb, ub = p.bounds()
    assert np.array_equal(g_l, lb) and np.array_equal(g_u, ub)
    Gh, Jh = p.eval_host(Z)
    Gg, _ = p.eval_host(Z, A.EVAL_G)  # eval_g runs the values-only instantiation, like Constraints::compute(z, false)
    names = ["joint limits", "velocity limits", "torque limits", "obstacle avoidance constraints"]  # KinovaOptimizer.cpp:82-109
    offs = list(p.block_offsets) + [m]
    for k in range(Z.shape[0]):
        g = np.frombuffer(raw, np.float64, m, off); off += 8 * m
        vals = np.frombuffer(raw, np.float64, m * n, off).reshape(m, n); off += 8 * m * n
        rep = np.frombuffer(raw, np.int32, 7, off); off += 28
        assert np.array_equal(g, Gg[k]) and np.array_equal(vals, Jh[k])  # same library, same kernels: identical bits
        # Optimizer.cpp:374-378 per block, recomputed here from the ctypes rows
        want = [int((Gg[k, offs[b]:offs[b + 1]] - lb[offs[b]:offs[b + 1]]).min() < -1e-4 or
                    (ub[offs[b]:offs[b + 1]] - Gg[k, offs[b]:offs[b + 1]]).min() < -1e-4) for b in range(4)]
        assert list(rep[1:5]) == want and rep[0] == int(not any(want))
        # one library call for eval_g, one more for eval_jac_g at the same x: the four views share the fused evaluation
        assert rep[5] == 2 * k + 1 and rep[6] == 2 * k + 2
        line = r.stdout.splitlines()[k]
        assert all((f"[{names[b]}]" in line) == bool(want[b]) for b in range(4)), line
    assert off == len(raw)
