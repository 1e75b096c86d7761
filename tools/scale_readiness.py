#!/usr/bin/env python3
"""What can be said about 2 / 4 / 8 MI355X without a second GPU (VERDICT r4 next #3a): the structure-only model of a workload for
N = 1, 2, 4, 8 ranks -- T*(N), link term, latency-aware critical path, bytes on the links, HBM of the fullest rank (records owned,
records received, dense mirrors) -- from an ANALYSIS-ONLY handle on the host (PANGULU_AMD_ANALYSIS_ONLY=1: ordering, symbolic
factorisation, block pattern, mapping, models; no records, no device).

    python tools/scale_readiness.py elastic3d 77            # R64, nb = 256
    python tools/scale_readiness.py kkt 120
    python tools/scale_readiness.py poisson 96 --cr64       # complex Poisson (BASELINE configs[4] class), nb = 128
    python tools/scale_readiness.py --poisson-fit 48 64 80 96 --cr64   # ... and the largest N^3 that fits 8 x 288 GB, by a power-law fit

Prints one table per workload and one JSON line (kept under profiles/)."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PANGULU_AMD_ANALYSIS_ONLY"] = "1"

import numpy as np  # noqa: E402

import pangulu_amd as pa  # noqa: E402
from pangulu_amd import matrices as M  # noqa: E402
from tests.helpers import library_for, oracle_library  # noqa: E402

HBM_PER_GPU = 288e9


def matrix(workload, size):
    if workload == "elastic3d":
        return M.elastic3d(size), "elastic3d(%d)" % size
    if workload == "fem27":
        return M.fem27(size), "fem27(%d)" % size
    if workload == "kkt":
        return M.kkt(size), "kkt(%d)" % size
    if workload == "shell":
        return M.shell(size, size), "shell(%d,%d)" % (size, size)
    if workload == "poisson":
        return M.poisson3d(size), "poisson3d(%d)" % size
    raise SystemExit("unknown workload %r" % workload)


def analyse(workload, size, vtype, nb, threads):
    if vtype.startswith("c") and workload == "poisson":
        mat, label = M.poisson3d(size, dtype=np.complex128, shift=0.5j), "poisson3d(%d), diagonal 6 + 0.5i" % size  # 
    else:
        mat, label = matrix(workload, size)
    n, cp, ri, va, co = mat
    lib = library_for(oracle_library(vtype), vtype)
    t0 = time.time()
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype=vtype, ordering="nd", coords=co, lib=lib, nthread=threads)
    info = h.info()
    rows = {}
    for N in (1, 2, 4, 8):
        rows[N] = pa.model_for_ranks(h, N)
    pa.pangulu_finalize(h)
    out = {"workload": label, "value_type": vtype, "nb": nb, "n": int(info["n"]), "nnz": int(info["nnz"]), "flop": int(info["flop"]),
           "symbolic_nnz": int(info["symbolic_nnz"]), "blocks": int(info["nblocks_nondiag"]), "analysis_s": round(time.time() - t0, 1), "ranks": rows}
    return out


def table(r):
    print("%s  %s nb = %d: n = %d, %d entries, F = %.3e, symbolic nnz %.3e, %d off-diagonal blocks (analysis %.0f s)" % (
        r["workload"], r["value_type"].upper(), r["nb"], r["n"], r["nnz"], r["flop"], r["symbolic_nnz"], r["blocks"], r["analysis_s"]))
    print("  N | T*(N) ms | link ms | latency-aware chain ms | bound max(T*, chain) ms | sent GB | flop share | HBM fullest rank GB = records + received + mirrors | fits 288 GB")
    for N, m in r["ranks"].items():
        tot = m["hbm_bytes_fullest_rank"]
        print("  %d | %8.1f | %7.2f | %22.1f | %22.1f | %7.1f | %10.3f | %7.1f = %.1f + %.1f + %.1f | %s" % (
            N, 1e3 * m["T_star_s"], 1e3 * m["link_term_s_max"], 1e3 * m["latency_chain_s"], 1e3 * max(m["T_star_s"], m["latency_chain_s"]),
            m["sent_bytes"] / 1e9, m["rank_flop_share"], tot / 1e9, m["hbm_records_owned"] / 1e9, m["hbm_records_received"] / 1e9,
            m["hbm_dense_mirrors"] / 1e9, "yes" if tot < 0.9 * HBM_PER_GPU else ("tight" if tot < HBM_PER_GPU else "NO")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?")
    ap.add_argument("size", nargs="?", type=int)
    ap.add_argument("--cr64", action="store_true")
    ap.add_argument("--nb", type=int, default=0)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--poisson-fit", type=int, nargs="+", default=None)
    a = ap.parse_args()
    vtype = "cr64" if a.cr64 else "r64"
    nb = a.nb or (128 if a.cr64 else 256)
    if a.poisson_fit:
        rs = [analyse("poisson", s, vtype, nb, a.threads) for s in a.poisson_fit]
        for r in rs:
            table(r)
        # HBM of the fullest of 8 ranks ~ c N^e: least squares on the logarithms; the largest N below 90 % of 288 GB
        xs = np.log([float(s) for s in a.poisson_fit])
        ys = np.log([r["ranks"][8]["hbm_bytes_fullest_rank"] for r in rs])
        e, logc = np.polyfit(xs, ys, 1)
        nmax = math.exp((math.log(0.9 * HBM_PER_GPU) - logc) / e)
        f_e, f_c = np.polyfit(xs, np.log([float(r["flop"]) for r in rs]), 1)
        fit = {"hbm_fullest_of_8_ranks_bytes": "%.3g * N^%.2f" % (math.exp(logc), e), "largest_N_within_90pct_of_288GB": int(nmax),
               "hbm_at_256_GB_per_rank": math.exp(logc) * 256.0 ** e / 1e9, "flop": "%.3g * N^%.2f" % (math.exp(f_c), f_e),
               "flop_at_256": math.exp(f_c) * 256.0 ** f_e, "flop_at_largest_N": math.exp(f_c) * nmax ** f_e}
        print("fit over N = %s: HBM of the fullest of 8 ranks = %s bytes -> poisson3d(256) would need %.0f GB per rank; the largest N^3 within 90 %% of "
              "288 GB per rank: N = %d (F = %.2e)" % (a.poisson_fit, fit["hbm_fullest_of_8_ranks_bytes"], fit["hbm_at_256_GB_per_rank"], int(nmax), fit["flop_at_largest_N"]))
        print(json.dumps({"poisson_fit": fit, "cases": rs}))
        return
    r = analyse(a.workload, a.size, vtype, nb, a.threads)
    table(r)
    print(json.dumps(r))


if __name__ == "__main__":
    main()
