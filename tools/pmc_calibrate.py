#!/usr/bin/env python
"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU against kernels that move a KNOWN number of bytes in the access forms
the engine uses (tools/pmc_calib.hip), as MI355X_MICROARCH.md ("HBM") prescribes before trusting an absolute figure.

usage (on the GPU box): python tools/pmc_calibrate.py [out.json]      (default profiles/r04_pmc_calibration.json)

Passes (each its own rocprofv3 run; --pmc never together with a trace domain other than --kernel-trace):
  1. --pmc FETCH_SIZE          2. --pmc WRITE_SIZE
  3. --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum   (request counts by size, if this rocprofv3 knows them)
Output: per kernel the true bytes, the reported KiB and `factor` = true bytes / reported bytes -- the number bench.py and DESIGN.md
multiply a kernel's FETCH_SIZE by (rule: the factor of the access form that dominates the kernel's reads)."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES, SMALL = 1 << 31, 64 << 20
TRUE_READ = {"calib_rd16": BYTES, "calib_rd16b": BYTES, "calib_rd4": BYTES, "calib_rd8": BYTES, "calib_rd16s": BYTES // 2, "calib_rdlds": BYTES,
             "calib_rd16_cached": SMALL}
TRUE_WRITE = {"calib_wr16": BYTES, "calib_wr4": BYTES}
FORM = {"calib_rd16": "global_load_dwordx4, lanes contiguous (1 KiB per wave-instruction)", "calib_rd16b": "buffer_load_dwordx4, lanes contiguous",
        "calib_rd4": "global_load_dword, lanes contiguous (256 B per wave-instruction)", "calib_rd8": "global_load_dwordx2, lanes contiguous",
        "calib_rd16s": "global_load_dwordx4 at a 32-byte lane stride (half of every 64-byte piece requested)",
        "calib_rdlds": "buffer_load_dwordx4 ... lds (LDS-DMA)", "calib_rd16_cached": "global_load_dwordx4 over a 64 MiB window, 4 launches (Infinity-Cache resident after the first)",
        "calib_wr16": "global_store_dwordx4, lanes contiguous", "calib_wr4": "global_store_dword, lanes contiguous"}


def run_pass(exe, counters, tag, work):
    out = os.path.join(work, tag)
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "--", exe]
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if p.returncode != 0:
        return None, (p.stdout + p.stderr)[-1500:]
    rows = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").strip()
            rows.setdefault((name, r.get("Counter_Name")), []).append(float(r.get("Counter_Value", 0.0)))
    return rows, None


def main():
    out_json = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_pmc_calibration.json")
    work = os.path.join(ROOT, "gpurun_out", "pmc_calib")
    os.makedirs(work, exist_ok=True)
    exe = os.path.join(work, "pmc_calib")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(ROOT, "tools", "pmc_calib.hip"), "-o", exe])
    res = {"source": "tools/pmc_calibrate.py: rocprofv3 --pmc <set> --kernel-trace --output-format csv -- pmc_calib (tools/pmc_calib.hip), one pass per counter set; "
                     "per-dispatch values averaged over the dispatches of a kernel", "kernels": {}, "errors": {}}
    fetch, e1 = run_pass(exe, ["FETCH_SIZE"], "fetch", work)
    write, e2 = run_pass(exe, ["WRITE_SIZE"], "write", work)
    req, e3 = run_pass(exe, ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"], "req", work)
    for k, e in (("FETCH_SIZE", e1), ("WRITE_SIZE", e2), ("request counters", e3)):
        if e:
            res["errors"][k] = e
    avg = lambda rows, key: (sum(rows[key]) / len(rows[key])) if rows and key in rows else None
    for k, true in list(TRUE_READ.items()) + list(TRUE_WRITE.items()):
        ent = {"access_form": FORM[k], "true_bytes_per_dispatch": true}
        ctr, rows = ("FETCH_SIZE", fetch) if k in TRUE_READ else ("WRITE_SIZE", write)
        v = avg(rows, (k, ctr))
        if v is not None:
            ent[ctr + "_KiB"] = round(v, 1)
            ent["factor_true_over_reported"] = round(true / (v * 1024.0), 4) if v > 0 else None
            if rows and (k, ctr) in rows:
                ent["per_dispatch_KiB"] = [round(x, 1) for x in rows[(k, ctr)]]
        if req:
            for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"):
                a = avg(req, (k, c))
                if a is not None:
                    ent[c] = round(a, 1)
            r, r32, bub = (avg(req, (k, c)) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"))
            if r is not None and k in TRUE_READ:
                ent["bytes_per_read_request"] = round(true / r, 2) if r else None
        res["kernels"][k] = ent
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
