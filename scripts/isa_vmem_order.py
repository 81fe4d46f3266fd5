#!/usr/bin/env python3
"""Order of vector-memory loads, stores and vmcnt waits in one kernel of a gfx950 assembly listing.

  python scripts/isa_vmem_order.py /tmp/raptor.s legged_core_kernelILi32ELb1E

gfx9 retires vector-memory loads and stores through ONE in-order counter, so a load issued behind a store cannot be consumed before
that store's round trip to memory has ended.  The listing shows runs of L (load) / S (store) / scratch traffic and every
`s_waitcnt vmcnt(n)` with the number of stores still outstanding in front of it (stores issued since the last vmcnt(0))."""
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
on = False
seq = []
label = ""
for ln, line in enumerate(open(path), 1):
    if not on:
        if re.match(r"^_Z\d+" + re.escape(kern) + r"[^:]*:", line):
            on = True
        continue
    if "s_endpgm" in line:
        break
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        label = m.group(1)
        continue
    t = line.strip().split()
    if not t:
        continue
    op = t[0]
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        seq.append(("L", label))
    elif op.startswith(("global_store", "buffer_store", "flat_store")):
        seq.append(("S", label))
    elif op.startswith("scratch_load"):
        seq.append(("l", label))
    elif op.startswith("scratch_store"):
        seq.append(("s", label))
    elif op == "s_waitcnt" and "vmcnt" in line:
        n = int(re.search(r"vmcnt\((\d+)\)", line).group(1))
        seq.append(("W%d" % n, label))
out = []
stores_pending = 0
run = None
for kind, lab in seq:
    if kind in ("S", "s"):
        stores_pending += 1
    if kind.startswith("W"):
        tag = "%s[stores in flight <= %d]" % (kind, stores_pending)
        if kind == "W0":
            stores_pending = 0
        out.append((tag, 1, lab))
        continue
    if out and out[-1][0] == kind and out[-1][2] == lab:
        out[-1] = (kind, out[-1][1] + 1, lab)
    else:
        out.append((kind, 1, lab))
for kind, n, lab in out:
    print("%-12s %-40s" % (lab, kind + (" x%d" % n if n > 1 else "")))
