"""Multi-GPU partitioning (reference rule src/pangulu.c:83-90, src/pangulu_common.h:135) and the structure-only
model behind it (pg_model.cpp), on the CPU: the mapping and the model depend on the rank COUNT only, so they are evaluated
for 2, 4 and 8 ranks in one process through an analysis-only handle (oracle/pangulu_amd_test_hooks.h)."""
import ctypes
import os

import numpy as np
import pytest

import pangulu_amd as pa
from pangulu_amd import matrices as M

from .helpers import library_for, oracle_library


@pytest.fixture
def tlib():
    lib = library_for(oracle_library("r64"))
    yield lib
    lib.pangulu_amd_test_set_analysis_ranks(1)
    os.environ.pop("PANGULU_AMD_ANALYSIS_ONLY", None)
    os.environ.pop("PANGULU_AMD_SEPARATOR_MAP", None)


def analysis_handle(lib, mat, nb, nranks):
    n, cp, ri, va, co = mat
    os.environ["PANGULU_AMD_ANALYSIS_ONLY"] = "1"
    lib.pangulu_amd_test_set_analysis_ranks(nranks)
    return pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, ordering="nd", coords=co, lib=lib, nthread=4)


def test_structure_model_equals_the_model_from_the_records(tlib):
    """T* from the symbolic pattern alone (every rank, any rank count) == T* from the block records (single-rank handles):
    same bytes, same flops, same split into HBM-bound and MFMA-bound tasks; the flops sum to F."""
    for mat, nb in ((M.fem27(12), 32), (M.shell(30, 30), 48), (M.poisson3d(14), 32), (M.kkt(8), 32)):
        n, cp, ri, va, co = mat
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, ordering="nd", coords=co, lib=tlib, nthread=4)
        a = h.info()
        tlib.pangulu_amd_model_roofline(h.ref, 8000.0, 78.6)
        b = h.info()
        assert a["model_ranks_bytes_total"] == b["model_bytes_total"]
        assert abs(a["model_ranks_tstar_hbm"] - b["model_tmin_hbm_bound"]) <= 1e-12 * b["model_tmin_hbm_bound"]
        assert abs(a["model_ranks_tstar_fp"] - b["model_tmin_fp_bound"]) <= 1e-12 * max(b["model_tmin_fp_bound"], 1e-30)
        f = (ctypes.c_double * 1)()
        assert tlib.pangulu_amd_rank_model(h.ref, None, f, None) == 1
        assert f[0] == float(a["flop"]) == b["model_flop_total"]
        assert a["model_rank_flop_share"] == 1.0 and a["model_critical_path_tasks"] >= 3
        assert 0 < a["model_critical_path"] <= a["model_ranks_tstar_sum"]
        pa.pangulu_finalize(h)


@pytest.mark.parametrize("name,nb,limit", [("fem27_24", 64, 1.25), ("fem27_20", 32, 1.25), ("shell_60", 64, 1.30), ("poisson_20", 32, 1.25)])
def test_flop_weighted_proportional_mapping_balances_the_ranks(tlib, name, nb, limit):
    """max over ranks of the structural flops a rank executes <= limit x the mean, at 2, 4 and 8 ranks; the heavy
    separators are shared 2D block-cyclic over rank groups that shrink down the tree, subtrees live on single ranks."""
    mat = {"fem27_24": lambda: M.fem27(24), "fem27_20": lambda: M.fem27(20), "shell_60": lambda: M.shell(60, 60),
           "poisson_20": lambda: M.poisson3d(20)}[name]()
    for nranks in (2, 4, 8):
        h = analysis_handle(tlib, mat, nb, nranks)
        info = h.info()
        flop = (ctypes.c_double * nranks)()
        tstar = (ctypes.c_double * nranks)()
        assert tlib.pangulu_amd_rank_model(h.ref, tstar, flop, None) == nranks
        fl = np.array(flop[:])
        assert abs(fl.sum() - info["flop"]) <= 1e-9 * info["flop"]
        assert fl.max() / fl.mean() <= limit, (name, nranks, fl / fl.mean())
        assert abs(info["model_rank_flop_share"] - fl.max() / fl.mean()) < 1e-12
        # T*(N) is below the single-rank T* and not below its N-th part
        assert info["model_ranks_tstar_sum"] / nranks <= info["model_ranks_tstar_max"] < info["model_ranks_tstar_sum"]
        assert info["model_sent_bytes_total"] > 0
        # structure of the map: some block column is shared by several ranks (a distributed separator) and, at the leaves,
        # some column lives on one rank whole
        nbk = int(info["block_length"])
        top_owners = {tlib.pangulu_amd_block_owner(h.ref, i, nbk - 1) for i in range(max(0, nbk - 6), nbk)} | \
                     {tlib.pangulu_amd_block_owner(h.ref, nbk - 1, j) for j in range(max(0, nbk - 6), nbk)}
        assert len(top_owners) > 1, "the top separator sits on one rank"
        assert all(0 <= o < nranks for o in top_owners)
        pa.pangulu_finalize(h)


def test_legacy_separator_maps_still_available(tlib):
    mat = M.fem27(16)
    shares = {}
    for mode in ("group", "path", "cyclic", "rank0"):
        os.environ["PANGULU_AMD_SEPARATOR_MAP"] = mode
        h = analysis_handle(tlib, mat, 32, 4)
        shares[mode] = h.info()["model_rank_flop_share"]
        pa.pangulu_finalize(h)
    # the reference's rule applied to every separator balances; one rank for all separators does not
    assert shares["group"] < 1.5 and shares["cyclic"] < 1.5 and shares["rank0"] > shares["group"] and shares["path"] > shares["group"]
