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
json.dump(out, open("profiles/pmc_latest.json", "w"), indent=1)
print("\n".join(k for k in out if ":" in k))
