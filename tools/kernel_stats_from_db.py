#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats writes a rocpd SQLite database (ROCm 7.2 default); this turns its `kernels` view into the
per-kernel summary CSV kept under profiles/ (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs).
Usage: tools/kernel_stats_from_db.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1:3]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by {name}").fetchall()
    total = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, calls, tot, avg, mn, mx in rows:
            w.writerow([n, calls, int(tot), round(avg, 1), round(100.0 * tot / total, 3), int(mn), int(mx)])


if __name__ == "__main__":
    main()
