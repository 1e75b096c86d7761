"""Deterministic synthetic matrices and MatrixMarket input for tests and bench.py.

The SuiteSparse matrices BASELINE.json names (ldoor, Serena, nlpkkt120) are not in the image and there is no
network, so bench.py uses stand-ins of the same class and size unless a MatrixMarket path is
given.  All stand-ins are strictly diagonally dominant: the factorisation has no pivoting.
Every generator returns ``(n, colptr[u64], rowidx[u32], values, coords or None)`` in CSC with sorted columns.
"""
import numpy as np
import scipy.sparse as sp


def _finish(A, dtype, coords=None):
    A = sp.csc_matrix(A, dtype=dtype)
    A.sum_duplicates()
    A.sort_indices()
    return (A.shape[0], A.indptr.astype(np.uint64), A.indices.astype(np.uint32), A.data.astype(dtype), coords)


def to_scipy(n, colptr, rowidx, values):
    return sp.csc_matrix((values, rowidx.astype(np.int64), colptr.astype(np.int64)), shape=(n, n))


def _grid_coords(nx, ny, nz):
    x, y, z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    return np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1).astype(np.float64)


def _stencil(nx, ny, nz, offsets):
    """Adjacency (no diagonal) of a 3D grid under the given neighbour offsets; vertex id = (x*ny + y)*nz + z."""
    x, y, z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    x, y, z = x.ravel(), y.ravel(), z.ravel()
    rows, cols = [], []
    for dx, dy, dz in offsets:
        ok = (x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (z + dz >= 0) & (z + dz < nz)
        src = ((x * ny + y) * nz + z)[ok]
        dst = (((x + dx) * ny + (y + dy)) * nz + (z + dz))[ok]
        rows.append(src)
        cols.append(dst)
    n = nx * ny * nz
    r, c = np.concatenate(rows), np.concatenate(cols)
    return sp.csc_matrix((np.ones(len(r)), (r, c)), shape=(n, n))


def poisson3d(nx, ny=None, nz=None, dtype=np.float64, shift=0.0):
    """7-point Laplacian, diagonal 6 (+ shift), off-diagonals -1.  With a complex dtype and shift=0.5j this is
    BASELINE.json's "3D 7-point Poisson, complex path" (diagonal 6+0.5i)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    off = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    G = _stencil(nx, ny, nz, off)
    n = nx * ny * nz
    A = -G.astype(dtype) + sp.identity(n, dtype=dtype, format="csc") * (6.0 + shift)
    return _finish(A, dtype, _grid_coords(nx, ny, nz))


def fem27(nx, ny=None, nz=None, dtype=np.float64):
    """27-point (trilinear-FEM-like) stencil: off-diagonals -1, diagonal = degree + 1 (Serena class)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    off = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]
    G = _stencil(nx, ny, nz, off)
    deg = np.asarray(G.sum(axis=1)).ravel()
    A = -G.astype(dtype) + sp.diags(deg + 1.0, format="csc").astype(dtype)
    return _finish(A, dtype, _grid_coords(nx, ny, nz))


def shell(nx, ny, layers=2, dofs=3, dtype=np.float64):
    """ldoor-class stand-in: a thin structural shell.  `layers` sheets of an nx x ny node mesh with 27-point node
    connectivity and `dofs` unknowns per node, every node pair coupled by a full dofs x dofs block
    (n = nx*ny*layers*dofs; about 18*dofs entries per row for 2 layers).  Off-diagonal entries are
    -(1 + 0.25*((i*7 + j*13) mod 5)) so the blocks are not rank-one, the diagonal makes every row strictly
    diagonally dominant.  ldoor itself: n = 952 203, 44.6 entries per row."""
    off = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]
    G = _stencil(nx, ny, layers, off) + sp.identity(nx * ny * layers, format="csc")
    K = sp.kron(G, np.ones((dofs, dofs)), format="coo")
    i, j = K.row.astype(np.int64), K.col.astype(np.int64)
    offd = i != j
    i, j = i[offd], j[offd]
    v = -(1.0 + 0.25 * ((i * 7 + j * 13) % 5))
    n = nx * ny * layers * dofs
    A = sp.csc_matrix((v, (i, j)), shape=(n, n))
    rowsum = np.asarray(abs(A).sum(axis=1)).ravel()
    A = A + sp.diags(rowsum + 1.0, format="csc")
    coords = np.repeat(_grid_coords(nx, ny, layers), dofs, axis=0)
    return _finish(A.astype(dtype), dtype, coords)


def elastic3d(nx, ny=None, nz=None, dofs=3, dtype=np.float64):
    """Serena-class stand-in with Serena's ROW LENGTH: a 3D solid with `dofs` unknowns per node and the 15-point node connectivity of
    a tetrahedral mesh (the node itself, its 6 face and its 8 corner neighbours), every node pair coupled by a full dofs x dofs
    block: 15 * dofs = 45 entries per row.  Serena itself (gas-reservoir geomechanics): n = 1 391 349, 46.1 entries per row.
    elastic3d(77): n = 1 369 599, 61.2 M entries.  Values as in shell(): off-diagonal blocks that are not rank-one, rows strictly
    diagonally dominant (no pivoting needed)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    off = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if abs(a) + abs(b) + abs(c) in (1, 3)]
    G = _stencil(nx, ny, nz, off) + sp.identity(nx * ny * nz, format="csc")
    K = sp.kron(G, np.ones((dofs, dofs)), format="coo")
    i, j = K.row.astype(np.int64), K.col.astype(np.int64)
    offd = i != j
    i, j = i[offd], j[offd]
    v = -(1.0 + 0.25 * ((i * 7 + j * 13) % 5))
    n = nx * ny * nz * dofs
    A = sp.csc_matrix((v, (i, j)), shape=(n, n))
    rowsum = np.asarray(abs(A).sum(axis=1)).ravel()
    A = A + sp.diags(rowsum + 1.0, format="csc")
    coords = np.repeat(_grid_coords(nx, ny, nz), dofs, axis=0)
    return _finish(A.astype(dtype), dtype, coords)


def kkt(nx, dtype=np.float64, delta=1e-2, dominant=False):
    """nlpkkt-class stand-in (BASELINE configs[3]): the 2 x 2 block KKT system [[H, J^T], [J, -delta I]] of SURVEY.md §8d with
    delta = 1e-2 -- H a 7-point operator on nx^3 (symmetric positive definite), J a one-sided difference.  The matrix is symmetric
    QUASI-DEFINITE: it has an L D L^T (hence L U) factorisation without pivoting under ANY symmetric permutation (Vanderbei 1995), with
    pivots of both signs down to -delta and element growth of order |J|^2 / delta -- the class the reference treats with MC64 and
    inserted 1e-8 diagonals (src/pangulu_reordering.c:149-681, 715-796).  `dominant=True` is rounds 1-4's matrix, whose (2,2) block was
    -(delta + 4) I: diagonally dominant, an easy stand-in that is not the class (VERDICT r4 missing #3); `kkt_dominant(nx)` names it."""
    n1, cp, ri, va, coords = poisson3d(nx, dtype=dtype, shift=2.0)
    H = to_scipy(n1, cp, ri, va)
    J = (sp.identity(n1, format="csc") - _stencil(nx, nx, nx, [(1, 0, 0)])).astype(dtype) * 0.5
    A = sp.bmat([[H, J.T], [J, -(delta + (4.0 if dominant else 0.0)) * sp.identity(n1, dtype=dtype)]], format="csc")
    return _finish(A, dtype, np.concatenate([coords, coords + 0.25], axis=0))


def kkt_dominant(nx, dtype=np.float64, delta=1e-2):
    """Rounds 1-4's `kkt`: the same blocks with a diagonally dominant (2,2) block -(delta + 4) I."""
    return kkt(nx, dtype=dtype, delta=delta, dominant=True)


def trefethen(size=20, drop_first=True, dtype=np.float64):
    """Trefethen's prime matrix: primes on the diagonal, ones where |i-j| is a power of two.  size=20 with the
    first row/column dropped is the reference's only fixture, examples/Trefethen_20b.mtx (19 x 19, 147 entries)."""
    primes = []
    c = 2
    while len(primes) < size:
        if all(c % p for p in primes):
            primes.append(c)
        c += 1
    A = sp.lil_matrix((size, size))
    for i in range(size):
        A[i, i] = primes[i]
        d = 1
        while i + d < size:
            A[i, i + d] = 1.0
            A[i + d, i] = 1.0
            d *= 2
    A = A.tocsc()
    if drop_first:
        A = A[1:, 1:]
    return _finish(A, dtype, None)


def random_pattern(n, density, seed, dtype=np.float64, symmetric_pattern=True):
    """Random sparse matrix made diagonally dominant; for kernel-level parity tests."""
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=density, random_state=rng, format="csc", data_rvs=lambda k: rng.uniform(-1, 1, k))
    if symmetric_pattern:
        A = A + sp.csc_matrix((rng.uniform(-1, 1, A.nnz), A.T.tocsc().indices, A.T.tocsc().indptr), shape=(n, n))
    A = A.astype(dtype)
    if np.issubdtype(dtype, np.complexfloating):
        A = A + 1j * sp.csc_matrix((rng.uniform(-1, 1, A.nnz), A.indices, A.indptr), shape=(n, n))
    rowsum = np.asarray(abs(A).sum(axis=1)).ravel()
    A = A + sp.diags(rowsum + 1.0, format="csc").astype(dtype)
    return _finish(A, dtype, None)


def read_mtx(path, dtype=np.float64):
    """MatrixMarket reader (symmetric/hermitian storage is expanded, like examples/mmio_highlevel.h)."""
    import scipy.io

    A = scipy.io.mmread(path)
    return _finish(sp.csc_matrix(A), dtype, None)


def read_lid(path, dtype=np.float64):
    """The reference's binary matrix format (examples/example.c:112-163): `m, n` as u32, `nnz` as u64, then a CSR
    matrix -- row pointer u64 x (n+1), column index u32 x nnz, values x nnz in the library's value type."""
    import os

    dtype = np.dtype(dtype)
    with open(path, "rb") as f:
        m, n = np.fromfile(f, dtype=np.uint32, count=2)
        nnz = int(np.fromfile(f, dtype=np.uint64, count=1)[0])
        rowptr = np.fromfile(f, dtype=np.uint64, count=int(n) + 1)
        colidx = np.fromfile(f, dtype=np.uint32, count=nnz)
        values = np.fromfile(f, dtype=dtype, count=nnz)
    expected = 16 + 8 * (int(n) + 1) + (4 + dtype.itemsize) * nnz  # (the format does not name its value type: the size must)
    if len(rowptr) != int(n) + 1 or len(colidx) != nnz or len(values) != nnz or int(rowptr[-1]) != nnz or os.path.getsize(path) != expected:
        raise ValueError("%s is not a .lid file for value type %s (truncated, or written for another type)" % (path, dtype))
    A = sp.csr_matrix((values, colidx.astype(np.int64), rowptr.astype(np.int64)), shape=(int(m), int(n)))
    return _finish(A.tocsc(), dtype, None)


def write_lid(path, n, colptr, rowidx, values):
    """Writes a CSC matrix in the reference's .lid layout (CSR on disk)."""
    A = to_scipy(n, colptr, rowidx, values).tocsr()
    A.sort_indices()
    with open(path, "wb") as f:
        np.array([n, n], dtype=np.uint32).tofile(f)
        np.array([A.nnz], dtype=np.uint64).tofile(f)
        A.indptr.astype(np.uint64).tofile(f)
        A.indices.astype(np.uint32).tofile(f)
        A.data.astype(values.dtype).tofile(f)


def read_matrix(path, dtype=np.float64):
    """By the last letter of the name, as examples/example.c:100-163 does: ...x -> MatrixMarket, ...d -> .lid."""
    if path.endswith("d"):
        return read_lid(path, dtype)
    return read_mtx(path, dtype)


def read_rhs(path, n, dtype=np.float64):
    """Right-hand side file of examples/example.c:167-243: '%' comment lines, the length, then one value per line
    (two numbers, real and imaginary part, for complex types)."""
    dtype = np.dtype(dtype)
    with open(path) as f:
        lines = [ln for ln in f if ln.strip() and not ln.lstrip().startswith("%")]
    if not lines:
        raise ValueError("%s contains only comments or is empty" % path)
    length = int(lines[0].split()[0])
    if length != n:
        raise ValueError("vector dimension mismatch - expected %d, got %d" % (n, length))
    tok = " ".join(lines[1:]).split()
    per = 2 if np.issubdtype(dtype, np.complexfloating) else 1
    if len(tok) < per * n:
        raise ValueError("failed to read vector element %d from %s" % (len(tok) // per, path))
    v = np.array(tok[:per * n], dtype=np.float64)
    if per == 2:
        v = v[0::2] + 1j * v[1::2]
    return v.astype(dtype)


def rhs_of_ones(n, colptr, rowidx, values):
    """b = A * 1, the right-hand side examples/example.c:252-264 builds."""
    return np.asarray(to_scipy(n, colptr, rowidx, values).sum(axis=1)).ravel().astype(values.dtype)


def relative_residual(n, colptr, rowidx, values, x, b):
    """|| A x - b ||_2 / || b ||_2 (examples/example.c:304-364)."""
    A = to_scipy(n, colptr, rowidx, values)
    r = A @ x - b
    return float(np.linalg.norm(r) / np.linalg.norm(b))
