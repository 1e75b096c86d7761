"""N > 1 on CPU: 2D block-cyclic ownership, point-to-point block exchange, slot recycling, distributed SpTRSV.

Ranks are separate processes (rendezvous through torch.distributed / gloo on 127.0.0.1); kernels are the oracle's.
The sum of all ranks' factor blocks must equal the single-rank factors to rounding (the reference itself differs
between 1 and N ranks in the last bits), and ||Ax-b||/||b|| must stay at rounding level."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch.distributed  # noqa: F401  (pages the library in once, here, instead of in every rank under a timeout)

from pangulu_amd import matrices as M

from .helpers import ROOT, factorize, max_rel_diff, oracle_library


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(world, spec, nb, out_path, vtype="r64", platform="oracle", transport="host", repeat=False, separators="cyclic"):
    """`separators`: PANGULU_AMD_SEPARATOR_MAP of the run.  The product's default is "path" (a separator follows its heaviest
    child: on the small test matrices that often leaves nothing to exchange); the transport tests use the reference's 2D
    block-cyclic separators, test_separator_maps covers the others."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", PANGULU_TEST_TRANSPORT=transport, HSA_ENABLE_IPC_MODE_LEGACY="0",
                   PANGULU_TEST_REPEAT="1" if repeat else "0", PANGULU_AMD_SEPARATOR_MAP=separators)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker.py"), spec, str(nb), out_path, vtype, platform],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=int(os.environ.get("PANGULU_TEST_RANK_TIMEOUT", "420")))  # (the first `import torch` on a fresh box can take minutes)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    failed = [r for r, p in enumerate(procs) if p.returncode != 0]
    assert not failed, "ranks %s failed:\n%s" % (failed, "\n".join("--- rank %d ---\n%s" % (r, outs[r][-3000:]) for r in range(world)))


GENS = {"fem27_6": lambda: M.fem27(6), "poisson8": lambda: M.poisson3d(8), "shell_8x7": lambda: M.shell(8, 7),
        "trefethen": lambda: M.trefethen(), "random200": lambda: M.random_pattern(200, 0.03, 5), "fem27_9": lambda: M.fem27(9),
        "shell_20x16": lambda: M.shell(20, 16), "kkt6": lambda: M.kkt(6), "shell_40x40": lambda: M.shell(40, 40)}


@pytest.mark.parametrize("world,spec,nb", [(2, "fem27_6", 32), (2, "trefethen", 4), (4, "poisson8", 32), (4, "trefethen", 4),
                                           (3, "shell_8x7", 24), (2, "random200", 16),
                                           # subtree-to-rank mapping takes blocks out of the 2D block-cyclic process rows
                                           (4, "shell_20x16", 24), (4, "fem27_9", 16)])
def test_multirank_matches_single_rank(tmp_path, world, spec, nb):
    out = str(tmp_path / "out.npz")
    run_ranks(world, spec, nb, out)
    z = np.load(out)
    mat = GENS[spec]()
    n = len(z["L_ptr"]) - 1  # n_padded: a block-aligned dissection adds isolated unit rows
    ordering = "identity" if spec == "trefethen" else "nd"
    ref = factorize(mat, nb, oracle_library("r64"), ordering=ordering)
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    U = sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))
    assert int(z["flop"]) == ref["info"]["flop"]
    assert max_rel_diff(L, ref["L"]) < 1e-12 and max_rel_diff(U, ref["U"]) < 1e-12
    assert float(z["residual"]) < 1e-13
    # every rank took part and the bytes sent equal the bytes received
    assert sum(z["sent"]) == sum(z["recv"]) and sum(z["recv"]) > 0
    assert sum(z["tasks"]) == ref["info"]["ntask_ssssm"]
    if spec == "trefethen":
        assert float(z["residual"]) < 4e-16  # the reference printed 1.4e-16 at 2 and 4 ranks, nb=4


@pytest.mark.parametrize("separators", ["path", "rank0"])
@pytest.mark.parametrize("world,spec,nb", [(2, "shell_20x16", 24), (4, "shell_40x40", 32), (3, "fem27_9", 16), (4, "kkt6", 16)])
def test_separator_maps(tmp_path, world, spec, nb, separators):
    """The separators above the mapped subtrees on the rank of their heaviest child (the default) / all on rank 0: same factors
    as one rank, bytes sent = bytes received, every update ran somewhere."""
    out = str(tmp_path / "out.npz")
    run_ranks(world, spec, nb, out, separators=separators)
    z = np.load(out)
    mat = GENS[spec]()
    n = len(z["L_ptr"]) - 1
    ref = factorize(mat, nb, oracle_library("r64"))
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    U = sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))
    assert int(z["flop"]) == ref["info"]["flop"]
    assert max_rel_diff(L, ref["L"]) < 1e-12 and max_rel_diff(U, ref["U"]) < 1e-12
    assert float(z["residual"]) < 1e-12
    assert sum(z["sent"]) == sum(z["recv"])
    assert sum(z["tasks"]) == ref["info"]["ntask_ssssm"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,spec,nb", [(2, "fem27_6", 32), (4, "poisson8", 32), (3, "shell_8x7", 24)])
def test_multirank_on_the_gpu_host_staged(tmp_path, world, spec, nb):
    """Same check with the HIP back-end: all ranks share the box's single GPU, blocks travel host-staged
    (D2H -> TCP -> H2D).  Exercises the receive thread's uploads next to the compute thread's launches."""
    out = str(tmp_path / "out.npz")
    run_ranks(world, spec, nb, out, platform="hip")
    z = np.load(out)
    mat = GENS[spec]()
    n = len(z["L_ptr"]) - 1  # n_padded: a block-aligned dissection adds isolated unit rows
    ref = factorize(mat, nb, oracle_library("r64"), ordering="nd")
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    U = sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))
    assert max_rel_diff(L, ref["L"]) < 1e-12 and max_rel_diff(U, ref["U"]) < 1e-12
    assert float(z["residual"]) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("separators", ["path", "rank0"])
@pytest.mark.parametrize("world,spec,nb", [(2, "shell_40x40", 128), (4, "shell_40x40", 256), (3, "fem27_9", 128)])
def test_separator_maps_on_the_gpu(tmp_path, world, spec, nb, separators):
    """The default separator mapping (and "rank0") with the HIP back-end and the peer-copy transport, dense paths engaged."""
    out = str(tmp_path / "out.npz")
    run_ranks(world, spec, nb, out, platform="hip", transport="ipc", separators=separators)
    z = np.load(out)
    mat = GENS[spec]()
    n = len(z["L_ptr"]) - 1
    ref = factorize(mat, nb, oracle_library("r64"), ordering="nd")
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    U = sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))
    assert max_rel_diff(L, ref["L"]) < 1e-12 and max_rel_diff(U, ref["U"]) < 1e-12
    assert float(z["residual"]) < 1e-12
    assert sum(z["sent"]) == sum(z["recv"])
    assert sum(z["tasks"]) == ref["info"]["ntask_ssssm"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,spec,nb", [(2, "fem27_6", 32), (3, "shell_8x7", 24), (2, "fem27_9", 128), (4, "shell_20x16", 128)])
def test_multirank_on_the_gpu_peer_copies(tmp_path, world, spec, nb):
    """The one-node transport: every rank maps its peers' HBM arenas (HIP IPC) and pulls announced records with one
    device-to-device copy.  On the single-GPU test box all ranks share the device, which exercises the mapping,
    the announcements and the pulls (not the xGMI links).  The nb = 128 cases run the dense (MFMA) paths, including
    solves against diagonal blocks another rank factorised (LU images rebuilt from the received halves)."""
    from pangulu_amd import _lib

    out = str(tmp_path / "out.npz")
    run_ranks(world, spec, nb, out, platform="hip", transport="ipc")
    z = np.load(out)
    assert int(z["transport"]) == _lib.TRANSPORT_IPC, "peer-copy transport fell back to host staging"
    mat = GENS[spec]()
    n = len(z["L_ptr"]) - 1
    ref = factorize(mat, nb, oracle_library("r64"), ordering="nd")
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    U = sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))
    assert max_rel_diff(L, ref["L"]) < 1e-12 and max_rel_diff(U, ref["U"]) < 1e-12
    assert float(z["residual"]) < 1e-13
    assert sum(z["sent"]) == sum(z["recv"]) and sum(z["recv"]) > 0


def test_ipc_transport_falls_back_on_a_host_memory_platform(tmp_path):
    """Requesting peer copies where there is no device arena (the CPU oracle platform) must degrade to host staging on
    all ranks together, not fail."""
    from pangulu_amd import _lib

    out = str(tmp_path / "out.npz")
    run_ranks(2, "fem27_6", 32, out, transport="ipc")
    z = np.load(out)
    assert int(z["transport"]) == _lib.TRANSPORT_HOST
    assert float(z["residual"]) < 1e-13


@pytest.mark.parametrize("world,spec,nb", [(2, "fem27_6", 32), (4, "shell_20x16", 24), (2, "kkt6", 16)])
def test_multirank_snapshot_reset_and_second_factorisation(tmp_path, world, spec, nb):
    """bench.py's multi-rank sequence (snapshot, gstrf, reset_numeric, gstrf) gives the same factors twice."""
    out = str(tmp_path / "out.npz")
    run_ranks(world, spec, nb, out, repeat=True)
    z = np.load(out)
    ref = factorize(GENS[spec](), nb, oracle_library("r64"), ordering="nd")
    n = len(z["L_ptr"]) - 1
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    U = sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))
    assert max_rel_diff(L, ref["L"]) < 1e-12 and max_rel_diff(U, ref["U"]) < 1e-12
    assert float(z["residual"]) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("world,spec,nb,transport", [(2, "fem27_9", 128, "ipc"), (4, "shell_40x40", 256, "ipc"), (2, "kkt6", 64, "host")])
def test_multirank_on_the_gpu_snapshot_reset_and_second_factorisation(tmp_path, world, spec, nb, transport):
    out = str(tmp_path / "out.npz")
    run_ranks(world, spec, nb, out, platform="hip", transport=transport, repeat=True)
    z = np.load(out)
    ref = factorize(GENS[spec](), nb, oracle_library("r64"), ordering="nd")
    n = len(z["L_ptr"]) - 1
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    U = sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))
    assert max_rel_diff(L, ref["L"]) < 1e-12 and max_rel_diff(U, ref["U"]) < 1e-12
    assert float(z["residual"]) < 1e-12


@pytest.mark.gpu
def test_multirank_cr64_on_the_gpu_peer_copies(tmp_path):
    """BASELINE config 5 class (complex Poisson, CR64) on 2 ranks with the one-node transport."""
    from pangulu_amd import _lib

    out = str(tmp_path / "out.npz")
    run_ranks(2, "poisson12c", 128, out, vtype="cr64", platform="hip", transport="ipc")
    z = np.load(out)
    assert int(z["transport"]) == _lib.TRANSPORT_IPC
    mat = M.poisson3d(12, dtype=np.complex128, shift=0.5j)
    ref = factorize(mat, 128, oracle_library("cr64"), vtype="cr64", ordering="nd")
    n = len(z["L_ptr"]) - 1
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    U = sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))
    assert max_rel_diff(L, ref["L"]) < 1e-12 and max_rel_diff(U, ref["U"]) < 1e-12
    assert float(z["residual"]) < 1e-13


def test_rccl_transport_falls_back_on_a_host_memory_platform(tmp_path):
    """Requesting RCCL where there is no device (the CPU oracle platform): all ranks agree over the control plane and
    degrade to host staging together."""
    from pangulu_amd import _lib

    out = str(tmp_path / "out.npz")
    run_ranks(3, "fem27_6", 32, out, transport="rccl")
    z = np.load(out)
    assert int(z["transport"]) == _lib.TRANSPORT_HOST
    assert float(z["residual"]) < 1e-13


@pytest.mark.gpu
def test_rccl_transport_on_a_shared_gpu_degrades_without_hanging(tmp_path):
    """On the one-GPU test box two ranks share the device, which RCCL refuses ("Duplicate GPU detected"): the ids are
    exchanged, the helper thread's communicator creation fails or times out, every rank falls back and the factorisation is
    still right.  (The working RCCL path needs two devices: the driver's multi-GPU bench reports config.transport.)"""
    from pangulu_amd import _lib

    out = str(tmp_path / "out.npz")
    os.environ["PANGULU_AMD_RCCL_TIMEOUT_S"] = "30"
    try:
        run_ranks(2, "fem27_6", 32, out, platform="hip", transport="rccl")
    finally:
        del os.environ["PANGULU_AMD_RCCL_TIMEOUT_S"]
    z = np.load(out)
    assert int(z["transport"]) in (_lib.TRANSPORT_HOST, _lib.TRANSPORT_RCCL)
    ref = factorize(GENS["fem27_6"](), 32, oracle_library("r64"), ordering="nd")
    n = len(z["L_ptr"]) - 1
    L = sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n))
    assert max_rel_diff(L, ref["L"]) < 1e-12 and float(z["residual"]) < 1e-13
