#!/usr/bin/env python3
"""Launch sequence of the LAST step of a profiled run, from a rocprofv3 rocpd SQLite database: every kernel in start order
with its duration and the idle gap in front of it.

usage: tools/rocpd_seq.py <results.db> <steps_in_run> [out.txt]
The run must consist of `steps_in_run` identical steps (warm-up included); the last len/steps launches are printed."""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    start = "start" if "start" in cols else "start_timestamp"
    end = "end" if "end" in cols else "end_timestamp"
    rows = c.execute("select name, %s, %s from kernels order by %s" % (start, end, start)).fetchall()
    n = len(rows) // int(sys.argv[2])
    last = rows[-n:]
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    t_prev = last[0][1]
    busy = 0
    for name, s, e in last:
        short = name.split("(")[0].replace("void ", "")
        if len(short) > 90:
            short = short[:87] + "..."
        out.write("%8.1f us  gap %6.1f  %s\n" % ((e - s) / 1e3, (s - t_prev) / 1e3, short))
        t_prev = e
        busy += e - s
    out.write("# %d launches, busy %.1f us, span %.1f us\n" % (n, busy / 1e3, (last[-1][2] - last[0][1]) / 1e3))


if __name__ == "__main__":
    main()
