"""CPU check of the per-operator test plumbing (tests/slots.py): hand-built slots driven through the ORACLE's single-task
operators in the reference's serial order must reproduce the factors the native scheduler produces on the same platform."""
import ctypes

import numpy as np
import pytest

import pangulu_amd as pa
from pangulu_amd import _lib
from pangulu_amd import matrices as M

from . import slots as S
from .helpers import library_for, oracle_library


@pytest.mark.parametrize("vtype", ["r64", "cr64"])
@pytest.mark.parametrize("name,gen,nb", [("fem27_5", lambda dt: M.fem27(5, dtype=dt), 32), ("kkt3", lambda dt: M.kkt(3, dtype=dt), 16),
                                         ("shell_8x6", lambda dt: M.shell(8, 6, dtype=dt), 64)])
def test_hand_built_slots_reproduce_the_scheduler_result(name, gen, nb, vtype):
    dt = _lib.VALUE_TYPES[vtype][0]
    mat = gen(dt)
    recs = S.exported_records(mat, nb, vtype)
    ora = ctypes.CDLL(oracle_library(vtype))
    fo = S.declare_platform(ora, "0100000")
    bm = S.BlockMatrix(recs, nb, dt, None)
    tasks = bm.tasks()
    arr = bm.task_array(tasks)
    fo("hybrid_batched")(nb, len(tasks), arr)
    # the same matrix through the scheduler
    n, cp, ri, va, co = mat
    lib = library_for(oracle_library(vtype), vtype)
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype=vtype, coords=co, lib=lib)
    pa.pangulu_gstrf(h)
    done = {(br, bc, up): v for br, bc, up, _, _, v in pa.owned_blocks(h)}
    pa.pangulu_finalize(h)
    scale = max(np.abs(v).max() for v in done.values() if len(v))
    for key, b in bm.blocks.items():
        if b.nnz:
            assert np.abs(b.values - done[key]).max() <= 1e-13 * scale, key
