"""One rank of a multi-process CPU run (launched by tests/test_multirank.py).

Rendezvous and result gathering use torch.distributed with the gloo backend; the solver's own process group
(pangulu_amd_comm_init: TCP control plane + host-staged block records) is what is under test.  The numeric kernels
are the oracle's CPU platform -- this checks the distributed scheduler, not the GPU.
"""
import faulthandler
import os
import sys
import time

faulthandler.enable()  # a rank that dies on a signal says where (round 5 lost a rank of an 8-process case without a line of output)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get("PANGULU_TEST_NATIVE_BACKTRACE"):  # the native frames of a crash inside the solver library (tests/native_backtrace.c)
    import ctypes
    import subprocess

    _so = "/tmp/pg_native_backtrace_%d.so" % os.getpid()
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", _so, os.path.join(ROOT, "tests", "native_backtrace.c")])
    ctypes.CDLL(_so).pg_test_install_native_backtrace()

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import pangulu_amd as pa  # noqa: E402
from pangulu_amd import _lib  # noqa: E402
from pangulu_amd import matrices as M  # noqa: E402
from tests.helpers import oracle_library  # noqa: E402


_T0 = time.time()


def stage(what):
    """One line per stage on stderr (shown by the harness only when a rank fails): the last one printed is where the rank was."""
    sys.stderr.write("[mp_worker rank %s pid %d +%.2fs] %s\n" % (os.environ.get("RANK"), os.getpid(), time.time() - _T0, what))
    sys.stderr.flush()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    stage("imports done")
    spec, nb, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    vtype = sys.argv[4] if len(sys.argv) > 4 else "r64"
    platform = sys.argv[5] if len(sys.argv) > 5 else "oracle"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if platform == "oracle":
        lib = _lib.load(vtype, test_hooks=True)  # the checker's build of the host, operators routed to the CPU restatement
        assert lib.pangulu_amd_use_platform_library(oracle_library(vtype).encode(), _lib.PLATFORM_CPU_NAIVE) == 0
    else:
        lib = _lib.load(vtype)  # the product; every rank on the one GPU of the test box (LOCAL_RANK % device count)
    # the solver's own listeners (base_port + rank): a block below the ephemeral range, where the rendezvous port and
    # gloo's pair sockets live (a number taken by one of those would be dialled by mistake)
    base_port = 20000 + (int(os.environ["MASTER_PORT"]) * 7) % 8000
    transport = {"host": _lib.TRANSPORT_HOST, "ipc": _lib.TRANSPORT_IPC, "rccl": _lib.TRANSPORT_RCCL}[os.environ.get("PANGULU_TEST_TRANSPORT", "host")]
    stage("library loaded, gloo up")
    assert lib.pangulu_amd_comm_init(rank, world, b"127.0.0.1", base_port, transport, None) == 0
    stage("comm_init done")
    dtype = _lib.VALUE_TYPES[vtype][0]
    gen = {"fem27_6": lambda: M.fem27(6, dtype=dtype), "poisson8": lambda: M.poisson3d(8, dtype=dtype),
           "shell_8x7": lambda: M.shell(8, 7, dtype=dtype), "trefethen": lambda: M.trefethen(dtype=dtype),
           "random200": lambda: M.random_pattern(200, 0.03, 5, dtype=dtype), "fem27_9": lambda: M.fem27(9, dtype=dtype),
           "shell_20x16": lambda: M.shell(20, 16, dtype=dtype),
           "poisson12c": lambda: M.poisson3d(12, dtype=dtype, shift=0.5j if np.issubdtype(dtype, np.complexfloating) else 0.0),
           "kkt6": lambda: M.kkt(6, dtype=dtype), "shell_40x40": lambda: M.shell(40, 40, dtype=dtype),
           "kkt8": lambda: M.kkt(8, dtype=dtype), "kkt10": lambda: M.kkt(10, dtype=dtype),
           # (the opt-in at-size cases of test_multirank.py: "elastic3d_<m>", "kkt_<m>", "cpoisson_<m>")
           **({spec: lambda: M.elastic3d(int(spec.split("_")[1]))} if spec.startswith("elastic3d_") else {}),
           **({spec: lambda: M.kkt(int(spec.split("_")[1]), dtype=dtype)} if spec.startswith("kkt_") else {}),
           **({spec: lambda: M.poisson3d(int(spec.split("_")[1]), dtype=dtype, shift=0.5j)} if spec.startswith("cpoisson_") else {})}[spec]
    n, cp, ri, va, co = gen()
    ordering = "identity" if spec == "trefethen" else "nd"
    if rank == 0:
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype=vtype, ordering=ordering, coords=co if ordering == "nd" else None, lib=lib)
    else:
        h = pa.pangulu_init(0, 0, None, None, None, nb=nb, vtype=vtype, ordering=ordering, lib=lib)  # rank 0 broadcasts the matrix
    stage("pangulu_init done")
    repeat = os.environ.get("PANGULU_TEST_REPEAT") == "1"
    if repeat:  # bench.py's sequence: snapshot, gstrf, reset_numeric, gstrf again
        assert lib.pangulu_amd_snapshot(h.ref) == 0
    pa.pangulu_gstrf(h)
    replayed = [int(h.info()["replayed"])]
    if repeat:
        L1, U1 = pa.factors_as_scipy(h)
        scale = max(abs(L1).max(), abs(U1).max())
        for again in range(int(os.environ.get("PANGULU_TEST_REPEATS", "1"))):
            assert lib.pangulu_amd_reset_numeric(h.ref) == 0
            pa.pangulu_gstrf(h)
            replayed.append(int(h.info()["replayed"]))
            L2, U2 = pa.factors_as_scipy(h)
            assert abs(L1 - L2).max() <= 1e-12 * scale and abs(U1 - U2).max() <= 1e-12 * scale, "factorisation %d differs" % (again + 2)
    stage("gstrf done")
    info = h.info()
    info["replayed_flags"] = replayed
    L, U = pa.factors_as_scipy(h)  # this rank's blocks only
    b = M.rhs_of_ones(n, cp, ri, va) if rank == 0 else None
    x = pa.pangulu_gstrs(h, b)
    parts = [None] * world if rank == 0 else None
    dist.gather_object((L, U, info), parts, dst=0)
    if rank == 0:
        Ls = sum(p[0] for p in parts)
        Us = sum(p[1] for p in parts)
        import scipy.sparse as sp

        Ls = Ls - (world - 1) * sp.identity(Ls.shape[0], format="csc", dtype=dtype)  # every rank added the unit diagonal
        res = M.relative_residual(n, cp, ri, va, x, b)
        np.savez(out_path, L_data=Ls.tocsc().data, L_ind=Ls.tocsc().indices, L_ptr=Ls.tocsc().indptr,
                 U_data=Us.tocsc().data, U_ind=Us.tocsc().indices, U_ptr=Us.tocsc().indptr, residual=res,
                 flop=parts[0][2]["flop"], sent=[p[2]["sent_bytes"] for p in parts], recv=[p[2]["recv_bytes"] for p in parts],
                 recv_blocks=[p[2]["recv_blocks"] for p in parts], tasks=[p[2]["ntask_ssssm"] for p in parts],
                 transport=int(lib.pangulu_amd_comm_transport()), replayed=[p[2]["replayed_flags"] for p in parts])
    stage("results gathered")
    pa.pangulu_finalize(h)
    lib.pangulu_amd_comm_finalize()
    dist.barrier()
    dist.destroy_process_group()
    stage("end")


if __name__ == "__main__":
    main()
