#!/usr/bin/env python
"""Per-kernel averages of every counter of a `rocprofv3 --pmc ... --kernel-trace` pass (rocpd sqlite).
usage: pmc_dump.py results.db [name-filter]   -> one line per kernel: dispatches, avg duration, counters (averaged per dispatch)."""
import re
import sqlite3
import sys


def main(db, filt=""):
    cur = sqlite3.connect(db).cursor()
    agg = {}
    for name, counter, n, val, dur in cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(end-start) "
                                                  "from counters_collection group by kernel_name, counter_name"):
        if filt and filt not in name:
            continue
        k = re.sub(r"\(.*$", "", name).replace("void ", "")
        agg.setdefault(k, {"n": n, "us": dur / 1e3})[counter] = val
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"] * kv[1]["n"]):
        cs = "  ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items()) if c not in ("n", "us"))
        print("%-64s n=%4d %9.1f us  %s" % (k[:64], v["n"], v["us"], cs))


if __name__ == "__main__":
    main(*sys.argv[1:3])
