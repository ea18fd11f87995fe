#!/usr/bin/env python3
"""Per-kernel means of the hardware counters in one or more rocprofv3 rocpd SQLite databases (one per --pmc pass).

usage: tools/rocpd_pmc.py out.json pass1.db [pass2.db ...]
Output: {kernel name (template arguments stripped of the parameter list): {counter: {"dispatches": n, "mean_per_dispatch": v}}}
"""
import json
import sqlite3
import sys


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    res = {}
    for db in dbs:
        c = sqlite3.connect(db)
        # one row per (dispatch, counter): sum over the hardware instances first, then average over the dispatches
        rows = c.execute(
            "select kernel_name, counter_name, count(*), avg(v) from "
            "(select kernel_name, counter_name, dispatch_id, sum(value) as v from counters_collection "
            " group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name").fetchall()
        for name, counter, n, mean in rows:
            key = name.split("(")[0].replace("void ", "")
            res.setdefault(key, {})[counter] = {"dispatches": n, "mean_per_dispatch": mean}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
