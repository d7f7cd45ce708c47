#!/usr/bin/env python
"""MFMA-pipe utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
--kernel-trace` pass (rocpd sqlite).  usage: summarize_mfma.py results.db out.json "<source>"
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the chip (MI355X_MICROARCH.md: it counts cycles, 32 per
v_mfma_f32_16x16x4_f32, 64 per 32x32x2_f32); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so clock = GRBM / 8 / duration and
mfma_busy_frac = MFMA_BUSY / (1024 * GRBM / 8)."""
import json
import re
import sqlite3
import sys


def main(db, out_path, source):
    cur = sqlite3.connect(db).cursor()
    agg = {}
    for name, counter, n, val, dur in cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(end-start) "
                                                  "from counters_collection group by kernel_name, counter_name"):
        if "avsr::" not in name:
            continue
        k = re.sub(r"\(.*$", "", name).replace("void ", "")
        agg.setdefault(k, {"dispatches": n, "avg_duration_us": round(dur / 1e3, 2)})[counter] = val
    out = {}
    for k, v in agg.items():
        grbm, busy = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if not grbm:
            continue
        out[k] = {"dispatches": v["dispatches"], "avg_duration_us": v["avg_duration_us"],
                  "clock_ghz": round(grbm / 8 / (v["avg_duration_us"] * 1e3), 3),
                  "mfma_busy_cycles": int(busy), "mfma_busy_frac": round(busy / (1024.0 * grbm / 8.0), 4)}
    json.dump({"source": source, "kernels": out}, open(out_path, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_duration_us"] * kv[1]["dispatches"])[:12]:
        print("%-58s n=%4d %9.1f us  %.2f GHz  MFMA busy %5.1f%%" % (k[:58], v["dispatches"], v["avg_duration_us"], v["clock_ghz"], 100 * v["mfma_busy_frac"]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
