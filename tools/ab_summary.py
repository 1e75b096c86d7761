#!/usr/bin/env python3
"""Print the numbers an A/B of two bench.py lines is about: ms per step, the update class's summed kernel time, residuals."""
import json
import sys

for path in sys.argv[1:]:
    line = None
    for ln in open(path):
        if ln.startswith('{"metric"'):
            line = json.loads(ln)
    if line is None:
        print("%-50s no line" % path)
        continue
    k = line.get("kernels", {})
    d = k.get("ssssm_dense_mfma", {})
    print("%-50s %9.2f ms/step  value %s  mfma class %8.2f ms (%s launches, executed %s GFLOP)  getrf %6.2f  trsm %6.2f  densify %6.2f sparsify %6.2f  res %.1e fc %.1e%s" % (
        path.split("/")[-1], line["ms_per_step"], "%.0f" % line["value"] if line["value"] else "NULL", d.get("ms", 0.0), d.get("launches"), d.get("GFLOP_executed"),
        k.get("getrf", {}).get("ms", 0.0), k.get("tstrf", {}).get("ms", 0.0) + k.get("gessm", {}).get("ms", 0.0),
        k.get("densify", {}).get("ms", 0.0), k.get("sparsify", {}).get("ms", 0.0),
        line.get("residual") or -1, line.get("factor_check") or -1, "  PARITY FAILED" if line.get("parity_failed") else ""))
