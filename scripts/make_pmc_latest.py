#!/usr/bin/env python3
"""profiles/pmc_latest.json from the pmc_summary.txt files of one profiling round.
usage: python scripts/make_pmc_latest.py <round tag> <workload>=<pmc_summary.txt> ...
Counters are means per dispatch; WRITE_SIZE / FETCH_SIZE are in KB on gfx950 (x 1024 for bytes)."""
import json
import re
import sys

tag = sys.argv[1]
out = {"round": tag,
       "note": "rocprofv3 --pmc passes, one counter set per run of `python3 bench.py --workload <w> --steps 20 --warmup 5 --no-cpu-baseline "
               "--no-secondary --no-latency --no-multistart --no-eval-h --no-spinup` (scripts/profile_round.sh; summaries profiles/<tag>_<w>_pmc_summary.txt). "
               "WRITE_SIZE / FETCH_SIZE are in KB on gfx950: bytes = value x 1024. Mean per dispatch."}
for arg in sys.argv[2:]:
    w, path = arg.split("=")
    for line in open(path):
        m = re.match(r"^(?:void )?(.+?)\s+(\w+)\s+mean\s+([\d.]+)\s+n=(\d+)", line)
        if not m:
            continue
        kern, ctr, val = m.group(1).strip(), m.group(2), float(m.group(3))
        ent = out.setdefault(f"{w}:{kern}", {})
        if ctr in ("WRITE_SIZE", "FETCH_SIZE"):
            ent[ctr + "_bytes"] = val * 1024
        else:
            ent[ctr] = val
# the sources of the kernels these counters describe, by content: bench.py warns when one of them has changed since
import hashlib
import os
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raptor_amd", "csrc")
out["sources_sha256"] = {f: hashlib.sha256(open(os.path.join(src, f), "rb").read()).hexdigest()
                         for f in ("kinova_kernel.h", "legged_kernel.h", "period_kernel.h", "digit_kernel.h", "spatial_dev.h", "kernels.h",
                                   "device_types.h")}
json.dump(out, open("profiles/pmc_latest.json", "w"), indent=1)
print("\n".join(k for k in out if ":" in k))
