import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangulu_amd import _lib
_lib.test_library_path = lambda v="r64": "/tmp/asan/libpangulu_amd_test_r64.so"
import numpy as np
import pangulu_amd as pa
from pangulu_amd import matrices as M
from tests.helpers import factorize, oracle_library, library_for, max_rel_diff
# 1. orderings (graph-only + geometric) on several classes, analysis only
os.environ["PANGULU_AMD_ANALYSIS_ONLY"] = "1"
lib = library_for(oracle_library("r64"))
for name, mat in (("fem27_20", M.fem27(20)), ("shell_40", M.shell(40, 40)), ("poisson_24", M.poisson3d(24)), ("kkt_10", M.kkt(10)), ("elastic_10", M.elastic3d(10)),
                  ("random", M.random_pattern(400, 0.02, 3)), ("trefethen", M.trefethen())):
    n, cp, ri, va, co = mat
    for coords in (co, None):
        for thr in (1, 6):
            h = pa.pangulu_init(n, len(va), cp, ri, va, nb=64, ordering="nd", coords=coords, lib=lib, nthread=thr)
            i = h.info(); perm = pa.permutation(h)
            assert sorted(perm.tolist()) == list(range(len(perm)))
            pa.pangulu_finalize(h)
    print("ordering ok", name, flush=True)
os.environ.pop("PANGULU_AMD_ANALYSIS_ONLY")
# 2. whole factorisations on the oracle through the sanitised host (scheduler, records, solve)
for name, mat, nb in (("fem27_8", M.fem27(8), 32), ("elastic_6", M.elastic3d(6), 48), ("kkt_6", M.kkt(6), 16)):
    n, cp, ri, va, co = mat
    for coords in (co, None):
        r = factorize((n, cp, ri, va, coords), nb, oracle_library("r64"))
        assert r["residual"] < 1e-12, r["residual"]
    print("factorize ok", name, flush=True)
print("ASAN RUN DONE")
