"""One rank of a ThreadSanitizer run over the HOST code (tools/tsan_host.sh): the scheduler's thread, the launcher thread and the
transport's sender thread of a multi-rank factorisation on the oracle's CPU operators.  No torch, no gloo: the solver's own process
group (TCP) is all the ranks share.  argv: rank world base_port spec nb"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import pangulu_amd as pa  # noqa: E402
from pangulu_amd import _lib  # noqa: E402
from pangulu_amd import matrices as M  # noqa: E402
from tests.helpers import oracle_library  # noqa: E402

_lib.test_library_path = lambda v="r64": os.environ.get("PG_TSAN_LIBRARY", "/tmp/tsan/libpangulu_amd_test_r64.so")


def main():
    rank, world, base_port, spec, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    lib = _lib.load("r64", test_hooks=True)
    assert lib.pangulu_amd_use_platform_library(oracle_library("r64").encode(), _lib.PLATFORM_CPU_NAIVE) == 0
    if world > 1:
        assert lib.pangulu_amd_comm_init(rank, world, b"127.0.0.1", base_port, _lib.TRANSPORT_HOST, None) == 0
    kind, size = spec.split("_")
    n, cp, ri, va, co = {"fem27": M.fem27, "kkt": M.kkt, "elastic3d": M.elastic3d, "poisson": M.poisson3d}[kind](int(size))
    if rank == 0:
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, ordering="nd", coords=co, lib=lib, nthread=2)
    else:
        h = pa.pangulu_init(0, 0, None, None, None, nb=nb, ordering="nd", lib=lib, nthread=2)
    assert lib.pangulu_amd_snapshot(h.ref) == 0
    pa.pangulu_gstrf(h)
    assert lib.pangulu_amd_reset_numeric(h.ref) == 0
    pa.pangulu_gstrf(h)
    b = M.rhs_of_ones(n, cp, ri, va) if rank == 0 else None
    x = pa.pangulu_gstrs(h, b)
    if rank == 0:
        res = M.relative_residual(n, cp, ri, va, x, b)
        assert res < 1e-11, res
        print("rank 0: %s nb %d world %d residual %.2e" % (spec, nb, world, res), flush=True)
    pa.pangulu_finalize(h)
    if world > 1:
        lib.pangulu_amd_comm_finalize()


if __name__ == "__main__":
    main()
