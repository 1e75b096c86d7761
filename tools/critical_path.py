"""What the device was doing during ONE factorisation, from a rocprofv3 --kernel-trace run of bench.py.

For the last factorisation in the trace (one per timed step) prints, per kernel class: launches, summed duration, and the
EXCLUSIVE time (wall time during which only kernels of that class were running) -- the part of the step that class alone
is responsible for -- plus the idle time (no kernel running: launch gaps, host waits).  Classes overlap on purpose (side
streams, look-ahead), so summed durations say little about the step time; exclusive + idle add up to it.
"""
import csv
import sys
from collections import defaultdict

ABBR = [("block_trsv", "solve"), ("block_spmv", "solve"), ("ssssm_dense", "SD"), ("ssssm_tiles", "SD"), ("ssssm_front", "SD"), ("ssssm_sparse", "SS"), ("trsm_dense", "TD"), ("trsm_sparse", "TS"), ("getrf", "GF"),
        ("densify", "dn"), ("sparsify", "sp"), ("half_image", "hi"), ("diag_tile", "iv"), ("flop_count", "fc")]


def cls(name):
    for k, v in ABBR:
        if k in name:
            return v
    return "other"


def main(path, which=-1):
    # factorisations are separated by the un-timed reset of the block values, whose device-to-device copies show up as
    # rocclr copy kernels: a new group starts after every run of them (a fixed idle-time threshold mistook a 2 ms host
    # stall inside a factorisation for the boundary once)
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    groups, cur = [], []
    last_end = None
    for r in rows:
        if "rocclr" in r["Kernel_Name"]:
            if cur:
                groups.append(cur)
                cur = []
            continue
        # (round 4: with the records' snapshot in host memory the reset between two steps is an upload, no copy kernel shows up --
        #  but the device then sits empty for hundreds of milliseconds: a gap of more than 150 ms separates factorisations too)
        if cur and last_end is not None and int(r["Start_Timestamp"]) - last_end > 150_000_000:
            groups.append(cur)
            cur = []
        cur.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"])))
        last_end = max(last_end or 0, int(r["End_Timestamp"]))
    if cur:
        groups.append(cur)
    # factorisations only (the triangular solve at the end of a bench run is a group of its own)
    groups = [g for g in groups if len(g) > 50 and any(e[2] in ("SD", "SS", "GF") for e in g)]
    g = groups[which]
    t0, t1 = g[0][0], max(e[1] for e in g)
    # sweep
    pts = []
    for s, e, c in g:
        pts.append((s, 1, c))
        pts.append((e, -1, c))
    pts.sort()
    active = defaultdict(int)
    excl = defaultdict(int)
    idle = 0
    mixed = 0
    last = t0
    for t, d, c in pts:
        live = [k for k, v in active.items() if v > 0]
        if t > last:
            if not live:
                idle += t - last
            elif len(live) == 1:
                excl[live[0]] += t - last
            else:
                mixed += t - last
        active[c] += d
        last = t
    # idle gaps by (class that ended last -> class that starts next)
    gaps = defaultdict(lambda: [0, 0])
    running_end, last_cls = g[0][1], g[0][2]
    for s_, e_, c_ in g[1:]:
        if s_ > running_end:
            k = "%s->%s" % (last_cls, c_)
            gaps[k][0] += 1
            gaps[k][1] += s_ - running_end
        if e_ > running_end:
            running_end, last_cls = e_, c_
    print("idle gaps (device empty) by transition: " + ", ".join("%s %dx %.0f us avg" % (k, v[0], v[1] / v[0] / 1e3)
                                                                  for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]))
    tot = defaultdict(int)
    cnt = defaultdict(int)
    for s, e, c in g:
        tot[c] += e - s
        cnt[c] += 1
    print("factorisation %d of %d: span %.2f ms, idle %.2f ms, two or more classes at once %.2f ms" % (
        which % len(groups), len(groups), (t1 - t0) / 1e6, idle / 1e6, mixed / 1e6))
    print("| class | launches | summed ms | exclusive ms |\n|---|---|---|---|")
    for c in sorted(tot, key=lambda k: -excl[k]):
        print("| %s | %d | %.2f | %.2f |" % (c, cnt[c], tot[c] / 1e6, excl[c] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -1)
