#!/usr/bin/env python3
"""Kernel statistics (the `--stats` table) from a rocprofv3 rocpd SQLite database.

usage: tools/rocpd_stats.py <results.db> [out.csv]
Columns follow rocprofv3's kernel_stats.csv: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs.
"""
import csv
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    return [(r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / total, 3), r[4], r[5]) for r in rows]


if __name__ == "__main__":
    rows = stats(sys.argv[1])
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    w.writerows(rows)
