#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + one --pmc pass per counter) into one per-kernel table.

    python tools/summarize_rocprof.py <stats_dir> [<pmc_dir> ...] > profiles/<name>.md

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced reads (MI355X_MICROARCH.md, HBM section), so "fetch x2" is the corrected upper estimate.
"""
import collections
import csv
import glob
import os
import sys


def kernel_stats(d):
    rows = []
    for f in glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")):
        rows += list(csv.DictReader(open(f)))
    return rows


def pmc(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
    return agg, cnt


def main():
    stats = kernel_stats(sys.argv[1])
    counters = collections.defaultdict(dict)
    for d in sys.argv[2:]:
        agg, cnt = pmc(d)
        for k, v in agg.items():
            for c, s in v.items():
                counters[k][c] = (s, cnt[(k, c)])
    print("| kernel | calls | total ms | avg us | % | FETCH_SIZE GB (x2) | WRITE_SIZE GB | HBM GB per launch (fetch x2 + write) |")
    print("|---|---|---|---|---|---|---|---|")
    for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].split("(")[0]
        c = counters.get(name, {})
        fetch = c.get("FETCH_SIZE", (0, 0))[0] * 1024 / 1e9
        write = c.get("WRITE_SIZE", (0, 0))[0] * 1024 / 1e9
        calls = int(r["Calls"])
        per = (2 * fetch + write) / calls if calls and c else float("nan")
        print("| %s | %d | %.2f | %.1f | %.2f | %.2f (%.2f) | %.2f | %.4f |" % (
            name, calls, float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"]), fetch, 2 * fetch, write, per))


if __name__ == "__main__":
    main()
