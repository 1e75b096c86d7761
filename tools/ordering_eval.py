"""F (structural flops, the metric's numerator), fill and analysis time of the built-in ordering with and without mesh
coordinates, from analysis-only handles on the CPU (checker's build of the host; nothing is factorised).

  python tools/ordering_eval.py fem27:40 shell:120 poisson3d:48 [--nb 256]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import pangulu_amd as pa  # noqa: E402
from pangulu_amd import matrices as M  # noqa: E402
from tests.helpers import library_for, oracle_library  # noqa: E402


def build(spec):
    kind, _, arg = spec.partition(":")
    k = int(arg)
    if kind == "fem27":
        return M.fem27(k)
    if kind == "shell":
        return M.shell(k, k)
    if kind == "poisson3d":
        return M.poisson3d(k)
    if kind == "elastic3d":
        return M.elastic3d(k)
    if kind == "kkt":
        return M.kkt(k)
    raise SystemExit("unknown matrix " + spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="+")
    ap.add_argument("--nb", type=int, default=256)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    os.environ["PANGULU_AMD_ANALYSIS_ONLY"] = "1"
    lib = library_for(oracle_library("r64"))
    for spec in a.specs:
        n, cp, ri, va, co = build(spec)
        row = {}
        for label, coords in (("coords", co), ("graph", None)):
            t0 = time.time()
            h = pa.pangulu_init(n, len(va), cp, ri, va, nb=a.nb, ordering="nd", coords=coords, lib=lib, nthread=a.threads)
            info = h.info()
            row[label] = (float(info["flop"]), int(info["symbolic_nnz"]), info["time_reorder"], time.time() - t0, int(info["n_padded"]))
            pa.pangulu_finalize(h)
        fc, fg = row["coords"][0], row["graph"][0]
        print("%-14s n=%8d  coords: F=%.3e fill=%.1fM reorder %.2fs pad %d | graph: F=%.3e fill=%.1fM reorder %.2fs pad %d | F(graph)/F(coords)=%.2f"
              % (spec, n, fc, row["coords"][1] / 1e6, row["coords"][2], row["coords"][4] - n, fg, row["graph"][1] / 1e6, row["graph"][2],
                 row["graph"][4] - n, fg / fc), flush=True)


if __name__ == "__main__":
    main()
