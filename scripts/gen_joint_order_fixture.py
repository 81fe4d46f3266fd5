#!/usr/bin/env python3
"""tests/golden/joint_order_tables.json: the joint order of pinocchio::Model as the reference records it.

The reference keeps the order of its models' joints only as index-comment tables beside the joint-limit constants
(`JOINT_LIMITS_LOWER[NUM_JOINTS] = { -1000, // Px ...`).  This script reads the COMMENTS of those tables (data: one joint
name per initialiser line, in index order) and stores them with the file and line range they came from; nothing of the
reference is executed.  Run in the build container only (needs /root/reference):

    python scripts/gen_joint_order_fixture.py
"""
import json
import os
import re

REF = "/root/reference"
TABLES = {
    "talos": ("Examples/Talos/include/TalosConstants.h", "JOINT_LIMITS_LOWER"),
    "g1": ("Examples/Unitree-G1/include/G1Constants.h", "JOINT_LIMITS_LOWER"),
    "digit": ("Examples/Digit/include/DigitConstants.h", "JOINT_LIMITS_LOWER"),
}
out = {}
for robot, (rel, table) in TABLES.items():
    lines = open(os.path.join(REF, rel)).read().split("\n")
    start = next(i for i, l in enumerate(lines) if table + "[" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("};"))
    names = []
    for l in lines[start + 1:end]:
        if "//" not in l:
            continue
        names.append(l.rsplit("//", 1)[1].strip())  # the LAST comment of the line is the joint name (Digit has value comments before it)
    out[robot] = {"source": f"{rel}:{start + 1}-{end + 1}", "table": table, "names": names}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "joint_order_tables.json")
json.dump(out, open(dst, "w"), indent=1)
for k, v in out.items():
    print(k, v["source"], len(v["names"]), v["names"][:8])
