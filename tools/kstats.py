"""Rows of a rocprofv3 kernel_stats.csv whose kernel name contains a pattern: python tools/kstats.py <dir-or-csv> <pattern> [calls-divisor]"""
import csv
import os
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    if os.path.isdir(path):
        for root, _d, files in os.walk(path):
            for f in files:
                if f.endswith("kernel_stats.csv"):
                    path = os.path.join(root, f)
    tot = 0.0
    for r in csv.DictReader(open(path)):
        if pat in r["Name"]:
            ms = float(r["TotalDurationNs"]) / 1e6 / div
            tot += ms
            print("%-78s calls %6.1f  avg %8.1f us  %7.3f ms" % (r["Name"][:78], int(r["Calls"]) / div, float(r["AverageNs"]) / 1e3, ms))
    print("total %.3f ms" % tot)


if __name__ == "__main__":
    main()
