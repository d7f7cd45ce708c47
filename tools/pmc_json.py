#!/usr/bin/env python
"""Merge two `rocprofv3 --pmc <counter> --kernel-trace` passes (FETCH_SIZE and WRITE_SIZE, collected SEPARATELY, rocpd sqlite output)
into the per-kernel HBM-traffic summary bench.py reads (profiles/r0N_c4_lipcnn_pmc_vM.json).
usage: pmc_json.py fetch.db write.db out.json "<command line that was profiled>"
Correction (calibrated on this GPU, profiles/r04_pmc_calibration.json, tools/pmc_calibrate.py): FETCH_SIZE counts L2-miss read requests
at 64 B apiece and a whole 128-byte line is ONE request, so contiguous reads of every width are reported at exactly half their bytes:
fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact."""
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
               "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (average over the run). hbm_bytes_per_dispatch_corrected = 2 * FETCH_SIZE * 1024 + "
                        "WRITE_SIZE * 1024: factors 2.000 / 1.000 calibrated with known-bytes kernels (profiles/r04_pmc_calibration.json).",
               "fetch_factor": 2.0, "write_factor": 1.0, "calibration": "profiles/r04_pmc_calibration.json", "kernels": kernels}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:5])
