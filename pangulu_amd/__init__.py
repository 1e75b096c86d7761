"""MI355X-native sparse LU numeric factorisation (PanguLU-compatible API).

The compute path lives in ``pangulu_amd/lib/libpangulu_amd_<type>.so`` (C++ host scheduler + hand-written HIP kernels
for gfx950, built by ``__graft_entry__.build()``); this package is the ctypes binding, the synthetic matrix generators
and the MatrixMarket reader around it.
"""
from . import _lib, matrices  # noqa: F401
from .solver import (  # noqa: F401
    Handle, factor_check, factor_check_vectors, factors_as_scipy, hip_memory, hip_stats, model_for_ranks, owned_blocks, pangulu_finalize, pangulu_gssv, pangulu_gstrf, pangulu_gstrs,
    pangulu_init, permutation, update_values,
)
