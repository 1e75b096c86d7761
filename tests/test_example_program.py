"""examples/solve_mtx.c -- a C user program against include/pangulu.h (the reference's counterpart: examples/example.c:282-364).
CPU: built against the checker's build of the host, with a constructor that routes the operators to the oracle (the program itself
calls nothing but the public API).  GPU: built against the product library."""
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

from pangulu_amd import _lib
from pangulu_amd import matrices as M

from .helpers import ROOT, oracle_library

SRC = os.path.join(ROOT, "examples", "solve_mtx.c")
TREFETHEN = os.path.join(ROOT, "tests", "golden", "Trefethen_20b.mtx")


def build(tmp_path, test_platform):
    exe = str(tmp_path / "solve_mtx")
    cmd = ["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-DCALCULATE_TYPE_R64", "-I", os.path.join(ROOT, "include"), SRC]
    if test_platform:
        lib = _lib.test_library_path("r64")
        shim = tmp_path / "route_to_oracle.c"
        shim.write_text('int pangulu_amd_use_platform_library(const char *, unsigned int);\n'
                        '__attribute__((constructor)) static void route(void) { if (pangulu_amd_use_platform_library("%s", 0x%x)) __builtin_trap(); }\n'
                        % (oracle_library("r64"), _lib.PLATFORM_CPU_NAIVE))
        cmd.append(str(shim))
    else:
        lib = os.path.join(ROOT, "pangulu_amd", "lib", "libpangulu_amd_r64.so")
    cmd += ["-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib), "-lm"]
    subprocess.run(cmd, check=True)
    return exe


def run(exe, *args, expect_rc=0):
    out = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=600)
    assert out.returncode == expect_rc, out.stdout + out.stderr
    return out.stdout + out.stderr


def residual_of(text):
    m = re.search(r"\|\| Ax - b \|\| / \|\| b \|\| = ([0-9.eE+-]+)", text)
    assert m, text
    return float(m.group(1))


def write_cases(tmp_path):
    """A general file with an explicit right-hand side, and a symmetric one stored as its lower triangle."""
    n, cp, ri, va, _ = M.shell(6, 5)
    A = M.to_scipy(n, cp, ri, va).tocoo()
    general = str(tmp_path / "shell_general.mtx")
    scipy.io.mmwrite(general, A, symmetry="general")
    rng = np.random.default_rng(7)
    b = rng.standard_normal(n)
    rhs = str(tmp_path / "rhs.txt")
    with open(rhs, "w") as f:
        f.write("%% a right-hand side\n%d\n" % n)
        f.writelines("%.17g\n" % v for v in b)
    n2, cp2, ri2, va2, _ = M.poisson3d(6)
    S = M.to_scipy(n2, cp2, ri2, va2)
    assert abs(S - S.T).max() == 0
    symmetric = str(tmp_path / "poisson_symmetric.mtx")
    scipy.io.mmwrite(symmetric, sp.coo_matrix(S), symmetry="symmetric")
    M.write_lid(str(tmp_path / "shell.lid"), n, cp, ri, va)  # (the reference's binary layout of the same matrix as `general`)
    return general, rhs, A.tocsr(), b, symmetric


def test_example_program_on_the_oracle(tmp_path):
    exe = build(tmp_path, test_platform=True)
    text = run(exe, "-f", TREFETHEN, "-n", "8")
    assert "n = 19, 147 entries" in text and residual_of(text) < 1e-13  # (19 diagonal + 2 x 64 mirrored entries)
    general, rhs, A, b, symmetric = write_cases(tmp_path)
    assert residual_of(run(exe, "-f", general, "-n", "24", "-r", rhs)) < 1e-12
    assert residual_of(run(exe, "-f", symmetric, "-n", "32")) < 1e-12
    lid = run(exe, "-f", str(tmp_path / "shell.lid"), "-n", "24", "-r", rhs)
    assert residual_of(lid) < 1e-12 and "n = %d, %d entries" % (A.shape[0], A.nnz) in lid
    assert "usage" in run(exe, expect_rc=1)
    assert "cannot open" in run(exe, "-f", str(tmp_path / "missing.mtx"), expect_rc=1)


def test_example_program_has_no_cpu_fallback(tmp_path):
    """Against the product library and without a GPU the program ends with the library's message (reference behaviour on misuse:
    message + exit(1))."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    exe = build(tmp_path, test_platform=False)
    assert "no HIP device" in run(exe, "-f", TREFETHEN, "-n", "8", expect_rc=1)


@pytest.mark.gpu
def test_example_program_on_the_gpu(tmp_path):
    exe = build(tmp_path, test_platform=False)
    text = run(exe, "-f", TREFETHEN, "-n", "8")
    assert "n = 19, 147 entries" in text and residual_of(text) < 1e-13
    general, rhs, A, b, symmetric = write_cases(tmp_path)
    assert residual_of(run(exe, "-f", general, "-n", "128", "-r", rhs)) < 1e-12
    assert residual_of(run(exe, "-f", symmetric, "-n", "64")) < 1e-12
    assert residual_of(run(exe, "-f", str(tmp_path / "shell.lid"), "-n", "128")) < 1e-12


def test_example_program_at_two_ranks_on_the_oracle(tmp_path):
    """RANK / WORLD_SIZE / MASTER_* in the environment (what torch.distributed.run exports): both processes call
    pangulu_amd_comm_init, rank 0 reads the file and prints.  (The RCCL transport the program asks for falls back to host staging on a
    host-memory platform, on both ranks together.)"""
    from .test_multirank import free_port

    exe = build(tmp_path, test_platform=True)
    general, rhs, A, b, symmetric = write_cases(tmp_path)
    port = free_port()
    procs = [subprocess.Popen([exe, "-f", general, "-n", "24", "-r", rhs], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert residual_of(outs[0]) < 1e-12 and "2 rank(s)" in outs[0]
    assert "Ax - b" not in outs[1]


@pytest.mark.gpu
def test_example_program_at_two_ranks_on_the_gpu(tmp_path):
    """Two processes of the program sharing the box's GPU (the transport it asks for, RCCL, verifies itself at start-up and falls back
    on both ranks together when two ranks sit on one device)."""
    from .test_multirank import free_port

    exe = build(tmp_path, test_platform=False)
    general, rhs, A, b, symmetric = write_cases(tmp_path)
    port = free_port()
    procs = [subprocess.Popen([exe, "-f", general, "-n", "128", "-r", rhs], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                       HSA_ENABLE_IPC_MODE_LEGACY="0"))
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert residual_of(outs[0]) < 1e-12 and "2 rank(s)" in outs[0]
