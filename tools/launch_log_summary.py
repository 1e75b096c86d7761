#!/usr/bin/env python3
"""Summary of a PANGULU_HIP_LAUNCH_LOG file (one line per launch of the profile pass: class, us, workgroups, tasks, live slab
steps): the MFMA update kernel's launches by size, with the rate of the 128 x 128 x 16 slab steps they executed
(2 * 128 * 128 * 16 flops each if every 16 x 16 piece of the slab is live -- exact for dense fronts, an upper bound elsewhere).

    python tools/launch_log_summary.py <launch_log> """
import collections
import sys

rows = [ln.split() for ln in open(sys.argv[1]) if ln.strip()]
dense = [(float(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in rows if r[0] == "5"]
tot_us = sum(d[0] for d in dense)
print("update-kernel launches: %d, %.1f ms, %.2f T slab-step flops -> %.1f TFLOP/s over all launches" % (
    len(dense), tot_us / 1e3, sum(d[3] for d in dense) * 524288 / 1e12, sum(d[3] for d in dense) * 524288 / max(tot_us, 1e-9) / 1e6))
b = collections.defaultdict(lambda: [0, 0.0, 0, 0, 0])
for us, wg, tasks, steps in dense:
    k = 1
    while k < wg:
        k *= 4
    e = b[k]
    e[0] += 1
    e[1] += us
    e[2] += wg
    e[3] += tasks
    e[4] += steps
print("workgroups <= | launches |   total ms | share | avg us | slab steps per workgroup | TFLOP/s (slab steps)")
for k in sorted(b):
    n, us, wg, tasks, steps = b[k]
    print("%13d | %8d | %10.2f | %4.1f%% | %6.0f | %8.1f | %6.1f" % (k, n, us / 1e3, 100 * us / tot_us, us / n, steps / max(wg, 1), steps * 524288 / us / 1e6))
other = collections.defaultdict(float)
for r in rows:
    if r[0] != "5":
        other[r[0]] += float(r[1])
print("other classes (ms):", {k: round(v / 1e3, 2) for k, v in sorted(other.items())})
