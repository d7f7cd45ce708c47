"""Inter-kernel gaps of the REPLAYED train-step graph: rocprofv3 --kernel-trace of `bench.py --steps N` (graph mode), then per replayed step
the sum of kernel durations, the sum of the idle gaps between consecutive kernels, and the gap histogram -- how much of a step is
dependent-launch boundaries rather than kernels.      usage: python tools/graph_gaps.py <rocpd .db or kernel_trace.csv> [steps]"""
import csv
import sqlite3
import sys


def load(path):
    if path.endswith(".db"):
        cur = sqlite3.connect(path).cursor()
        return sorted((s, e, n) for n, s, e in cur.execute("select name, start, end from kernels"))
    rows = list(csv.DictReader(open(path)))
    return sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)


def main():
    ks = load(sys.argv[1])
    # a step = from one adam_kernel to the next; take the LAST complete steps (graph replays)
    idx = [i for i, k in enumerate(ks) if "adam_kernel" in k[2] or "optimiser_kernel" in k[2]]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    idx = idx[-(nsteps + 1):]
    tot_k = tot_g = tot_w = 0.0
    gaps = []
    n_k = 0
    for a, b in zip(idx[:-1], idx[1:]):
        seg = ks[a + 1:b + 1]
        tot_w += (seg[-1][1] - ks[a][1]) / 1e3
        prev_end = ks[a][1]
        for s, e, _n in seg:
            tot_k += (e - s) / 1e3
            g = max(0, s - prev_end) / 1e3
            tot_g += g
            gaps.append(g)
            prev_end = max(prev_end, e)
        n_k += len(seg)
    n = len(idx) - 1
    print("steps %d: %.1f kernels per step, wall %.1f us per step = kernels %.1f us + gaps %.1f us (%.1f %%)" %
          (n, n_k / n, tot_w / n, tot_k / n, tot_g / n, 100.0 * tot_g / tot_w))
    gaps.sort()
    q = lambda p: gaps[min(len(gaps) - 1, int(p * len(gaps)))]
    print("gap per launch: median %.2f us, mean %.2f us, p90 %.2f us, p99 %.2f us, max %.2f us" % (q(0.5), sum(gaps) / len(gaps), q(0.9), q(0.99), gaps[-1]))
    big = sorted(((max(0, ks[i][0] - ks[i - 1][1]) / 1e3, ks[i - 1][2][:60], ks[i][2][:60]) for i in range(idx[-2] + 1, idx[-1] + 1)), reverse=True)[:12]
    for g, a, b in big:
        print("  %.2f us between %s -> %s" % (g, a, b))


if __name__ == "__main__":
    main()
