"""Round 5: a look-ahead call of the scheduler leaves destinations with fewer than PANGULU_AMD_LOOKAHEAD_MIN_QUEUE queued updates alone
(pg_numeric.cpp; by default only from PANGULU_AMD_LOOKAHEAD_DEFER_FROM = 8192 queued updates on, which no test matrix reaches).  The rule
changes WHEN an update is launched, never the order of the updates of one destination: forced from the first update on, the factors
must be the oracle's own -- on the CPU platform, where every update is one call of the restated reference kernel, bit for bit."""
import os

import numpy as np
import pytest

from pangulu_amd import matrices as M
from tests.helpers import factorize, max_rel_diff, oracle_library


def run(mat, nb, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return factorize(mat, nb, oracle_library("r64"), ordering="nd")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name,mat,nb", [("fem27_7", M.fem27(7), 16), ("shell_24x20", M.shell(24, 20), 24), ("kkt7", M.kkt(7), 16)])
@pytest.mark.parametrize("min_queue", ["2", "3", "1000"])
def test_deferral_of_shallow_queues_gives_the_same_factors(name, mat, nb, min_queue):
    ref = run(mat, nb, {"PANGULU_AMD_LOOKAHEAD_MIN_QUEUE": "1"})
    got = run(mat, nb, {"PANGULU_AMD_LOOKAHEAD_MIN_QUEUE": min_queue, "PANGULU_AMD_LOOKAHEAD_DEFER_FROM": "0"})
    assert ref["info"]["deferred_queues"] == 0 and got["info"]["deferred_queues"] > 0  # (the rule was in force, and only there)
    assert got["info"]["flop"] == ref["info"]["flop"]
    assert got["info"]["ntask_ssssm"] == ref["info"]["ntask_ssssm"]
    # same updates in the same order per destination: identical bits
    assert (got["L"] != ref["L"]).nnz == 0 and (got["U"] != ref["U"]).nnz == 0
    assert got["residual"] < 1e-12 and got["factor_check"] < 1e-12
    assert max_rel_diff(got["L"], ref["L"]) == 0.0
