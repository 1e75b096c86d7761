#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (one --kernel-trace --stats run + one --pmc pass per counter group) into one table.

    python tools/summarize_rocprof.py <stats_dir> [<pmc_dir> ...] [--json out.json] > profiles/<name>.md

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced
reads (MI355X_MICROARCH.md, HBM section), so reads are doubled before they are compared with byte counts.
SQ_* cycle counters are per-SE quad-cycles as that guide describes; only ratios between them are used here.
"""
import collections, csv, glob, json, os, sys


def short(name):
    """kernel name without return type, anonymous namespace and argument list"""
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    return name[5:] if name.startswith("void ") else name


def files(d, suffix):
    return glob.glob(os.path.join(d, "*" + suffix)) + glob.glob(os.path.join(d, "*", "*" + suffix))


def main():
    args = sys.argv[1:]
    out_json = None
    if "--json" in args:
        k = args.index("--json"); out_json = args[k + 1]; args = args[:k] + args[k + 2:]
    stats = []
    for f in files(args[0], "_kernel_stats.csv"):
        stats += list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in args[1:]:
        for f in files(d, "_counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k][row["Counter_Name"]].add(row["Dispatch_Id"])
    cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
            "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE"]
    print("| kernel | calls | total ms | avg us | % | HBM read GB (FETCH_SIZE x2) | HBM write GB | HBM MB per launch | waiting % of wave cycles | issue-stalled % | MFMA pipes busy % (of 1024 SIMDs) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    summary = {}
    for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"])):
        name = short(r["Name"])
        if "rocclr" in name:
            continue
        c = agg.get(name, {})
        n = {k: max(1, len(disp[name][k])) for k in c}
        calls = int(r["Calls"])
        fetch = 2 * c.get("FETCH_SIZE", 0) * 1024 / n.get("FETCH_SIZE", 1)  # bytes per launch
        write = c.get("WRITE_SIZE", 0) * 1024 / n.get("WRITE_SIZE", 1)
        wc = c.get("SQ_WAVE_CYCLES", 0)
        wait = 100 * c.get("SQ_WAIT_ANY", 0) / wc if wc else float("nan")
        stall = 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else float("nan")
        # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum the 1024 SIMDs (256 CUs x 4)
        mfma = 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) if c.get("GRBM_GUI_ACTIVE") else float("nan")
        print("| %s | %d | %.1f | %.0f | %.1f | %.2f | %.2f | %.1f | %.0f | %.0f | %s |" % (
            name[:44], calls, float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"]),
            fetch * calls / 1e9, write * calls / 1e9, (fetch + write) / 1e6, wait, stall,
            ("%.1f (%.3g MFMA f64 16x16x4)" % (mfma, c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / 4)) if c.get("GRBM_GUI_ACTIVE") else "n/a"))
        summary[name] = {"calls": calls, "avg_us": float(r["AverageNs"]) / 1e3, "hbm_bytes_per_launch": fetch + write,
                         "fetch_bytes_per_launch_x2": fetch, "write_bytes_per_launch": write}
        if c.get("GRBM_GUI_ACTIVE"):
            # (bench.py's roofline.per_kernel quotes these two beside the per-launch bytes)
            summary[name]["mfma_busy_pct"] = mfma
            summary[name]["mfma_f64_16x16x4_per_launch"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / 4 / n.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 1)
        if wc:
            summary[name]["waiting_pct_of_wave_cycles"] = wait
            summary[name]["issue_stalled_pct"] = stall
    if out_json:
        # bench.py quotes these per-launch HBM bytes only for the build they were measured on
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench

        summary["kernel_source_hash"] = bench.kernel_source_hash()
        summary["profile"] = os.environ.get("PROFILE_NAME", os.path.basename(out_json))
        # profiles/hbm_traffic.json is keyed by workload (bench.workload_key): WORKLOAD_KEY set = merge into that file's entry
        key = os.environ.get("WORKLOAD_KEY")
        if key:
            allw = json.load(open(out_json)) if os.path.exists(out_json) else {}
            allw[key] = summary
            json.dump(allw, open(out_json, "w"), indent=1)
        else:
            json.dump(summary, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
