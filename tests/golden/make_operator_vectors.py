#!/usr/bin/env python3
"""Per-operator before/after vectors of the oracle's 0100000 operators on hand-built slots ("Next #1").

    python tests/golden/make_operator_vectors.py                     # rewrites tests/golden/operator_vectors.json, KEEPING its permutation
    python tests/golden/make_operator_vectors.py --new-permutation   # ... with the permutation today's nested dissection gives

The fixture CARRIES its permutation (`case.perm`, perm[new] = old): the records are built with it as a user ordering, so a change
of the ordering code no longer changes the vectors (they had to be regenerated two rounds running for that reason).

The case is small on purpose (poisson3d(5), nb = 16, R64: 8 block rows): the whole factorisation is run one operator call per
task in the reference's serial right-looking order (src/pangulu_kernel_interface.c:190-337); for the FIRST task of each kind
(GETRF, TSTRF, GESSM, SSSSM) the destination's values before and after the call are stored in full, for every task the sum
and the sum of squares of the destination afterwards.  tests/test_oracle_golden.py replays the oracle against the file
(pins the oracle and the slot builder), tests/test_gpu_operators.py the HIP operators (1e-12).
The oracle itself is pinned by the known-answer and reference-pin tests (tests/test_oracle_golden.py,
tests/test_reference_pin.py); these vectors freeze what it produces per operator."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from pangulu_amd import matrices as M  # noqa: E402
from tests import slots as S  # noqa: E402
from tests.helpers import oracle_library  # noqa: E402

CASE = {"generator": "poisson3d(5)", "nb": 16, "vtype": "r64", "ordering": "frozen permutation (case.perm in the fixture)"}
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "operator_vectors.json")
KIND = {S.GETRF: "getrf", S.TSTRF: "tstrf", S.GESSM: "gessm", S.SSSSM: "ssssm"}


def case_matrix():
    return M.poisson3d(5)


def frozen_perm():
    """The permutation the committed fixture was made with."""
    return np.array(json.load(open(FIXTURE))["perm"], dtype=np.uint32)


def case_records(perm=None):
    """The block records of the case under the fixture's permutation (or `perm`)."""
    return S.exported_records(case_matrix(), CASE["nb"], CASE["vtype"], user_perm=frozen_perm() if perm is None else perm)


def replay(call, bm, record_full=True):
    """Runs the serial task list through `call(kind, nb, dst, a, b)`; returns the per-task trace."""
    trace, seen = [], set()
    for kid, dst, a, b in bm.tasks():
        halves = [dst] if kid != S.GETRF else [dst, bm.blocks[(dst.brow, dst.bcol, 0 if dst.is_upper else 1)]]
        before = [np.array(h.values, dtype=np.float64).copy() for h in halves]
        call(kid, bm.nb, dst, a, b)
        after = [np.array(h.values, dtype=np.float64).copy() for h in halves]
        e = {"kind": KIND[kid], "dst": [int(dst.brow), int(dst.bcol), int(dst.is_upper)],
             "sum": float(sum(x.sum() for x in after)), "sumsq": float(sum((x * x).sum() for x in after))}
        if a is not None:
            e["op1"] = [int(a.brow), int(a.bcol), int(a.is_upper)]
        if b is not None:
            e["op2"] = [int(b.brow), int(b.bcol), int(b.is_upper)]
        if record_full and kid not in seen:
            seen.add(kid)
            e["before"] = [x.tolist() for x in before]
            e["after"] = [x.tolist() for x in after]
        trace.append(e)
    return trace


def oracle_call(fo):
    def call(kid, nb, dst, a, b):
        if kid == S.GETRF:
            fo("getrf")(nb, dst.ref(), 0)
        elif kid == S.TSTRF:
            fo("tstrf")(nb, dst.ref(), a.ref(), 0)
        elif kid == S.GESSM:
            fo("gessm")(nb, dst.ref(), a.ref(), 0)
        else:
            fo("ssssm")(nb, dst.ref(), a.ref(), b.ref(), 0)
    return call


def main():
    if "--new-permutation" in sys.argv or not os.path.exists(FIXTURE) or "perm" not in json.load(open(FIXTURE)):
        import pangulu_amd as pa
        from tests.helpers import library_for

        n, cp, ri, va, coords = case_matrix()
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=CASE["nb"], vtype=CASE["vtype"], ordering="nd", coords=coords,
                            lib=library_for(oracle_library(CASE["vtype"]), CASE["vtype"]))
        perm = pa.permutation(h)
        pa.pangulu_finalize(h)
        assert len(perm) == n, "the case's dissection must not pad (user permutations have the matrix's order)"
    else:
        perm = frozen_perm()
    recs = case_records(perm)
    bm = S.BlockMatrix(recs, CASE["nb"], np.float64, None)
    fo = S.declare_platform(ctypes.CDLL(oracle_library(CASE["vtype"])), "0100000")
    trace = replay(oracle_call(fo), bm)
    out = {"case": CASE, "perm": [int(x) for x in perm], "blocks": len(bm.blocks), "tasks": trace}
    path = FIXTURE
    with open(path, "w") as f:
        json.dump(out, f)
    kinds = {}
    for e in trace:
        kinds[e["kind"]] = kinds.get(e["kind"], 0) + 1
    print("wrote", path, os.path.getsize(path), "bytes;", len(trace), "tasks", kinds)


if __name__ == "__main__":
    main()
