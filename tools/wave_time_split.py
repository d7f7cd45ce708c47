#!/usr/bin/env python
"""Joins the per-kernel counter dumps of tools/pmc_dump.py into one table: where the waves of each kernel spend their time.
usage: wave_time_split.py <dump with SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_INST_LDS> <dump with SQ_VALU_MFMA_BUSY_CYCLES>
                          <dump with SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE> [a second LDS dump: "after"]
Columns (MI355X_MICROARCH.md, counter table): parked = SQ_WAIT_ANY (s_waitcnt / barrier), issue stall = SQ_WAIT_INST_ANY (matrix-pipe or
memory-pipe issue, accumulator dependencies), of which LDS issue = SQ_WAIT_INST_LDS, active = the rest -- shares of SQ_WAVE_CYCLES;
matrix pipe = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x 2.1 GHz); LDS = SQ_LDS_IDX_ACTIVE / 256 CUs / (duration x 2.1 GHz) with the
share of those cycles that are bank-conflict cycles."""
import re
import sys


def load(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+n=\s*(\d+)\s+([\d.]+) us\s+(.*)$", line.rstrip())
        if m:
            out[m.group(1).strip()] = dict(n=int(m.group(2)), us=float(m.group(3)), **{k: float(v) for k, v in (kv.split("=") for kv in m.group(4).split())})
    return out


W, M, L = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
L2 = load(sys.argv[4]) if len(sys.argv) > 4 else {}
GHZ = 2.1
print("%-56s %4s %8s | %6s %6s %6s %6s | %6s | %12s %12s" % ("kernel", "n", "us", "parked", "stall", "(LDS)", "active", "MFMA", "LDS busy/cfl", "after"))
for k, w in sorted(W.items(), key=lambda kv: -kv[1]["us"] * kv[1]["n"]):
    wc = w.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0 or w["us"] * w["n"] < 60:
        continue
    pk, st, sl = w["SQ_WAIT_ANY"] / wc, w["SQ_WAIT_INST_ANY"] / wc, w["SQ_WAIT_INST_LDS"] / wc
    mf = M.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / (w["us"] * 1e3 * GHZ)

    def lds(d):
        x = d.get(k)
        if not x or not x.get("SQ_LDS_IDX_ACTIVE"):
            return "      -     "
        return "%5.0f%% %4.0f%%" % (100 * x["SQ_LDS_IDX_ACTIVE"] / 256 / (x["us"] * 1e3 * GHZ), 100 * x["SQ_LDS_BANK_CONFLICT"] / x["SQ_LDS_IDX_ACTIVE"])
    print("%-56s %4d %8.1f | %5.0f%% %5.0f%% %5.1f%% %5.0f%% | %5.0f%% | %12s %12s" % (k[:56], w["n"], w["us"], 100 * pk, 100 * st, 100 * sl, 100 * max(0.0, 1 - pk - st),
                                                                                    100 * mf, lds(L), lds(L2)))
