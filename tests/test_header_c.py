"""include/raptor_hip.h is a C header: it must compile as C99 on its own, and the descriptor structs the ctypes mirrors in
raptor_amd/_capi.py restate by hand must have the same size and field offsets as the C compiler gives them."""
import ctypes
import os
import shutil
import subprocess

import pytest

from raptor_amd import _capi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRUCTS = {
    "raptor_model_desc": A.ModelDesc, "raptor_traj_desc": A.TrajDesc, "raptor_kinova_desc": A.KinovaDesc,
    "raptor_gait_desc": A.GaitDesc, "raptor_period_desc": A.PeriodDesc, "raptor_legged_desc": A.LeggedDesc,
    "raptor_digit_desc": A.DigitDesc,
}


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_compiles_as_c99_and_layouts_match(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "raptor_hip.h"', "int main(void) {"]
    for cname, cls in STRUCTS.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  printf("MAXB %d\\n", RAPTOR_H_NOALLOC_MAX_B);', "  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        parts = line.split()
        got[tuple(parts[:-1])] = int(parts[-1])
    assert got[("MAXB",)] == 112
    for cname, cls in STRUCTS.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
    # every field the header declares is mirrored (a field missing from the mirror would shift nothing but be lost)
    assert sum(1 for k in got if len(k) == 2 and k[1] != "size") == sum(len(c._fields_) for c in STRUCTS.values())
