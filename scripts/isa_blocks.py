#!/usr/bin/env python3
"""Static instruction census of one kernel in a gfx950 assembly listing, per basic block.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S raptor_amd/csrc/raptor_hip.hip -o /tmp/raptor.s
  python scripts/isa_blocks.py /tmp/raptor.s kinova_fused_kernelILb1ELb0

Prints, for every label of the kernel, the VALU / fp64 / SALU / SMEM / LDS / VMEM / branch instruction counts and the branch
targets, so that loop bodies (a block that some later branch jumps back to) can be priced by hand with their trip counts.
"""
import re
import sys


def classify(op):
    if op.startswith(("v_cmpx", "v_cmp")):
        return "valu"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
    blocks, cur = [], {"label": "entry", "n": {}, "targets": [], "f64": 0, "trans": 0}
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "n": {}, "targets": [], "f64": 0, "trans": 0}
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        k = classify(op)
        cur["n"][k] = cur["n"].get(k, 0) + 1
        if k == "valu" and "_f64" in op:
            cur["f64"] += 1
        if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div", "v_sin", "v_cos", "v_trig")):
            cur["trans"] += 1
        if k == "branch":
            t = s.split()[-1]
            cur["targets"].append(t)
    blocks.append(cur)
    order = {b["label"]: i for i, b in enumerate(blocks)}
    tot = {}
    print(f"{'block':12s} {'valu':>6s} {'f64':>6s} {'trans':>5s} {'salu':>6s} {'smem':>5s} {'lds':>5s} {'vmem':>5s} {'wait':>5s}  branches")
    for b in blocks:
        n = b["n"]
        for k, v in n.items():
            tot[k] = tot.get(k, 0) + v
        back = [t for t in b["targets"] if t in order and order[t] <= order[b["label"]]]
        print(f"{b['label']:12s} {n.get('valu',0):6d} {b['f64']:6d} {b['trans']:5d} {n.get('salu',0):6d} {n.get('smem',0):5d} {n.get('lds',0):5d} "
              f"{n.get('vmem',0):5d} {n.get('wait',0):5d}  {' '.join(b['targets'])}{'   <-- loop back' if back else ''}")
    print("total", tot)


if __name__ == "__main__":
    main()
