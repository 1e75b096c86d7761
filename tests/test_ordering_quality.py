"""Quality of the built-in nested dissection (pg_ordering.cpp; the reference hands the graph to METIS_NodeND,
src/pangulu_reordering.c:1065-1089).  F = sum_k (c_k + 2 c_k^2) is the numerator of the headline metric and the work of the
factorisation, so the ordering may not depend on the caller knowing mesh coordinates: the graph-only dissection (multilevel vertex
separators) has to stay within 1.3x of the geometric one, and the geometric one may not lose to the graph's (7-point meshes: the
smallest separators are diagonal, not axis planes).  Analysis-only handles on the CPU, nb = 256 as in the bench."""
import os

import pytest

import pangulu_amd as pa
from pangulu_amd import matrices as M

from .helpers import library_for, oracle_library


@pytest.fixture
def tlib():
    lib = library_for(oracle_library("r64"))
    os.environ["PANGULU_AMD_ANALYSIS_ONLY"] = "1"
    yield lib
    os.environ.pop("PANGULU_AMD_ANALYSIS_ONLY", None)


def analysis(lib, mat, coords):
    n, cp, ri, va, co = mat
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=256, ordering="nd", coords=co if coords else None, lib=lib, nthread=4)
    info = h.info()
    perm = pa.permutation(h)
    pa.pangulu_finalize(h)
    assert sorted(perm.tolist()) == list(range(len(perm)))
    return float(info["flop"]), int(info["symbolic_nnz"])


# (matrix, F(graph) / F(coords) at most, F(coords) / F(graph) at most, F(coords) at most: the values of the round this test was written in, +3 %)
CASES = [("fem27_40", lambda: M.fem27(40), 1.3, 1.1, 5.4e10), ("shell_120", lambda: M.shell(120, 120), 1.3, 1.1, 1.37e10),
         ("poisson3d_48", lambda: M.poisson3d(48), 1.3, 1.1, 5.6e10), ("kkt_16", lambda: M.kkt(16), 1.3, 1.1, 7.2e7)]


@pytest.mark.parametrize("name,gen,graph_over_coords,coords_over_graph,f_coords_max", CASES, ids=[c[0] for c in CASES])
def test_graph_only_ordering_is_close_to_the_geometric_one(tlib, name, gen, graph_over_coords, coords_over_graph, f_coords_max):
    mat = gen()
    f_c, fill_c = analysis(tlib, mat, True)
    f_g, fill_g = analysis(tlib, mat, False)
    assert f_g <= graph_over_coords * f_c, (name, f_g, f_c)
    assert f_c <= coords_over_graph * f_g, (name, f_c, f_g)
    assert f_c <= f_coords_max, (name, f_c)


def test_the_ordering_is_a_function_of_the_matrix_alone(tlib):
    """Same matrix, different thread counts: the same permutation (the regions' random choices are seeded from their vertices,
    never from the schedule of the OpenMP tasks that dissect them)."""
    n, cp, ri, va, co = M.fem27(30)  # (large enough for the halves to be dissected by concurrent tasks)
    perms = []
    for threads in (1, 4):
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=64, ordering="nd", coords=None, lib=tlib, nthread=threads)
        perms.append(pa.permutation(h).tolist())
        pa.pangulu_finalize(h)
    assert perms[0] == perms[1]


def test_elastic3d_has_serenas_row_length():
    """The default bench matrix: 3 unknowns per node, 15-point node connectivity, a full 3 x 3 block per node pair -- 45 entries per
    interior row (Serena: 46.1), strictly diagonally dominant rows (no pivoting needed), coordinates per unknown."""
    import numpy as np

    n, cp, ri, va, co = M.elastic3d(9)
    assert n == 3 * 9 ** 3 and co.shape == (n, 3)
    A = M.to_scipy(n, cp, ri, va).tocsr()
    row_len = np.diff(A.indptr)
    assert row_len.max() == 45 and (row_len == 45).sum() == 3 * 7 ** 3  # interior nodes
    d = A.diagonal()
    off = np.asarray(abs(A).sum(axis=1)).ravel() - abs(d)
    assert (d > off).all()
    # (values are not symmetric by construction; the pattern is)
    P = (A != 0).astype(np.int8)
    assert (P != P.T).nnz == 0


def test_unknowns_of_one_node_are_dissected_together(tlib):
    """Several unknowns per mesh node (elastic3d: 3, every unknown of a node coupled to the same neighbours): the graph-only dissection
    collapses indistinguishable vertices before it looks for separators (METIS_NodeND's `compress`), so that a refinement move
    carries a whole node; without it single-vertex FM is stuck, the more so the larger the graph.  F(graph) / F(coords) with / without
    (`PANGULU_AMD_ND_COMPRESS=0`; tools/ordering_eval.py): elastic3d(32) 0.99 / 1.05, (48) 1.01 / 1.08, (77) -- the bench matrix -- 1.08 / 1.26;
    shell(120), six unknowns per node: 1.03 / 1.14."""
    mat = M.elastic3d(32)
    f_c, _ = analysis(tlib, mat, True)
    f_g, _ = analysis(tlib, mat, False)
    assert f_g <= 1.02 * f_c, (f_g, f_c)


def test_constraint_rows_are_eliminated_ahead_of_their_unknowns(tlib):
    """nlpkkt-class stand-in [[H, J^T], [J, -d I]]: a row of J couples two unknowns that H couples as well -- a simplicial vertex of degree two.
    Such vertices are taken out before the dissection and ordered right ahead of their first neighbour, where they create no fill: the
    factorisation then costs what H's own costs (kkt(24): F = 7.9e8, was 9.6e8 with the constraint rows inside the dissection; the graph-only
    dissection 1.06x that, was 1.09-1.14x -- and 2.0x / 3.6x on the node / edge ways of the multilevel search alone)."""
    f_h, _ = analysis(tlib, M.poisson3d(24, shift=2.0), True)
    mat = M.kkt(24)
    f_c, fill_c = analysis(tlib, mat, True)
    f_g, fill_g = analysis(tlib, mat, False)
    assert f_c <= 1.02 * f_h, (f_c, f_h)
    assert f_g <= 1.1 * f_c, (f_g, f_c)


@pytest.mark.parametrize("name,gen", [("elastic3d_20", lambda: M.elastic3d(20)), ("kkt_20", lambda: M.kkt(20))], ids=["elastic3d_20", "kkt_20"])
def test_compressed_and_pre_eliminated_orderings_do_not_depend_on_the_thread_count(tlib, name, gen):
    """The paths of the end of round 4 -- indistinguishable vertices collapsed, constraint rows taken out, the attempts near the root as
    concurrent tasks from seeds drawn in order -- give the same permutation with one thread and with four."""
    n, cp, ri, va, co = gen()
    perms = []
    for threads in (1, 4):
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=64, ordering="nd", coords=None, lib=tlib, nthread=threads)
        perms.append(pa.permutation(h).tolist())
        pa.pangulu_finalize(h)
    assert perms[0] == perms[1]
