#!/usr/bin/env python
"""Merge two `rocprofv3 --pmc <counter> --kernel-trace` passes (FETCH_SIZE and WRITE_SIZE, collected SEPARATELY, rocpd sqlite output)
into the per-kernel HBM-traffic summary bench.py reads (profiles/r0N_c4_lipcnn_pmc_vM.json).
usage: pmc_json.py fetch.db write.db out.json "<command line that was profiled>"
gfx950 correction (MI355X_MICROARCH.md): fetch bytes = 2 * FETCH_SIZE * 1024 (wide reads are counted by half); WRITE_SIZE as reported."""
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, val in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        out[re.sub(r"\(.*$", "", name).replace("void ", "")] = (n, val)
    return out


def main(fetch_db, write_db, out_json, cmd):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        n = (f.get(k) or w.get(k))[0]
        fk, wk = (f.get(k) or (0, 0.0))[1], (w.get(k) or (0, 0.0))[1]
        kernels[k] = {"dispatches": n, "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                      "hbm_bytes_per_dispatch_corrected": int(2 * fk * 1024 + wk * 1024)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / rocprofv3 --pmc WRITE_SIZE --kernel-trace (separate passes) -- " + cmd,
               "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (average over the run). gfx950 correction: fetch bytes = 2 * FETCH_SIZE * 1024 "
                        "(wide reads are under-counted by half); WRITE_SIZE as reported.", "kernels": kernels}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:5])
