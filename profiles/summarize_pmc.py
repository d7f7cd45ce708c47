#!/usr/bin/env python
"""Merge two rocprofv3 PMC passes (rocpd sqlite: `--pmc FETCH_SIZE --kernel-trace`, `--pmc WRITE_SIZE --kernel-trace`)
into the per-kernel HBM-traffic JSON kept under profiles/ and read by bench.py.
usage: summarize_pmc.py fetch.db write.db out.json "<source description>"
gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE prices wide (128-B) read requests as
64 B, so coalesced-read bytes = 2 * FETCH_SIZE KiB * 1024; WRITE_SIZE is taken as reported."""
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, avg in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                                    "group by kernel_name", (counter,)):
        out[re.sub(r"\(.*$", "", name).replace("void ", "")] = (n, avg)
    return out


def main(fetch_db, write_db, out_path, source):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("avsr::"):
            continue
        fk, wk = f.get(k, (0, 0.0)), w.get(k, (0, 0.0))
        kernels[k] = {"dispatches": max(fk[0], wk[0]), "FETCH_SIZE_KiB": round(fk[1], 1), "WRITE_SIZE_KiB": round(wk[1], 1),
                      "hbm_bytes_per_dispatch_corrected": int(2 * fk[1] * 1024 + wk[1] * 1024)}
    json.dump({"source": source,
               "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (average over the run). gfx950 correction: fetch bytes = "
                        "2 * FETCH_SIZE * 1024 (wide reads are under-counted by half); WRITE_SIZE as reported.",
               "kernels": kernels}, open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        print("%-60s n=%4d  %10.1f KiB fetch  %10.1f KiB write" % (k[:60], v["dispatches"], v["FETCH_SIZE_KiB"], v["WRITE_SIZE_KiB"]))


if __name__ == "__main__":
    main(*sys.argv[1:5])
