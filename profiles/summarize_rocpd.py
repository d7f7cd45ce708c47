#!/usr/bin/env python
"""Turn a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into the per-kernel stats CSV kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, header):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    with open(out_path, "w") as f:
        f.write("# %s\n" % header)
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    print("total kernel ms %.3f" % (tot / 1e6))
    for r in rows[:14]:
        print("%-70s n=%5d tot=%8.0fus avg=%8.2fus %5.1f%%" % (r[0][:70], r[1], r[2] / 1e3, r[3] / 1e3, 100.0 * r[2] / tot))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
