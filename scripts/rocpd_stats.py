#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max (us).

usage: python scripts/rocpd_stats.py <results.db> [out.csv]
Equivalent of the `--stats` kernel table; written so the summaries under profiles/ are plain text.
"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1.0
    out = [("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((name, a[0], f"{a[1]:.1f}", f"{a[1] / a[0]:.2f}", f"{a[2]:.2f}", f"{a[3]:.2f}", f"{100 * a[1] / total:.1f}"))
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
