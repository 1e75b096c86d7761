#!/usr/bin/env python3
"""bench.py -- numeric factorisation GFLOP/s (pangulu_gstrf, R64) on N MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no RANK in the environment this process starts N rank processes itself -- fresh children of
`python -m torch.distributed.run`, created before anything here has touched a GPU -- relays rank 0's JSON line and exits
non-zero if any rank did.  Started by a launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) it is one rank:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one complete pangulu_gstrf of the workload matrix.  Between steps the block records are restored from a
device-side snapshot (pangulu_amd_reset_numeric, un-timed), so every timed step starts with its inputs resident in
HBM; each step is bracketed by a barrier + device synchronise on both sides and the slowest rank's time counts.
value = F / t with F = sum_k (c_k + 2 c_k^2) the reference's structural flop count (src/pangulu_kernel_interface.c:4-176,
computed once from the symbolic pattern outside the timed region).  The SAME matrix at every N: strong scaling.

Workload: BASELINE.json's north-star matrix is SuiteSparse Serena (n = 1 391 349, nnz = 64.1 M, 46 entries per row: a 3D geomechanics
model with 3 unknowns per node; R64, nb = 256).  It is not in the image and there is no network, so unless --mtx points at a
MatrixMarket / .lid file the run uses the deterministic stand-in pangulu_amd.matrices.elastic3d(77): a 77^3 node mesh, 3 unknowns
per node, the 15-point node connectivity of a tetrahedral mesh with a full 3 x 3 block per node pair -- n = 1 369 599, 60.0 M
entries, 45 per row, diagonally dominant: Serena's order, entry count AND row length (round 3's stand-in fem27(112) had the order
and 27 entries per row; it stays in the line as a `secondary` workload, next to shell(398,398), the ldoor-class matrix of BASELINE
configs[1] and default of rounds 1-2).  `--workload fem27|shell|poisson|kkt` select the other classes.
Ordering: built-in nested dissection (geometric with the generators' coordinates, multilevel graph-only with --no-coords or a
matrix file; stated in the JSON line: F depends on it).  The device-side snapshot moves to host memory when it would not fit.

The line's residual and factor_check (the reference's two criteria, examples/example.c:304-364 and
src/pangulu_numeric.c:1082-1341) are taken from the factors of the LAST TIMED step, in the timed configuration; kernel
times come from one extra, un-timed factorisation with every launch on one stream between two events (no queueing).
cpu_baseline legs (the oracle = CPU restatement of the reference's CPU platform, sampled) run in child processes after
the GPU steps, so nothing they do can take the GPU result with it.

Process layout: every rank process is a SUPERVISOR that never touches the GPU.  It starts the GPU worker (this file with
--gpu-worker) as a fresh child, and if that fails -- a transport that passes its start-up self-test and then stalls or
crashes: the RCCL plane has never run on real links in the builder's hands -- every rank's supervisor starts a new worker
with the next transport of the order rccl -> ipc -> host (workers end behind a common barrier, so the ranks agree on
success).  Then the cpu_baseline children, then rank 0 prints the one JSON line.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak (v_mfma_f64_16x16x4: 64 cycles per 2048 flop per SIMD)
XGMI_LINK_GBS = 153.0     # one xGMI link per GPU pair
PARITY_TOL = 1e-10        # north star: ||Ax - b|| / ||b|| within 1e-10 of the CPU reference; the factor check is held to the same bound
CHECK_VECTORS = 8         # factor check on the all-ones vector + 7 seeded random +-1 vectors (pangulu_amd_factor_check_vectors)
RC_PARITY_FAILED = 4      # exit code of a run whose factors fail the gate: the line is printed with "value": null, "parity_failed": true
GENERAL_KERNEL_ROCPROF_NAME = "ssssm_tilesv_f64_kernel"  # the general MFMA update kernel as rocprofv3 names it (profiles/hbm_traffic.json)
CPU_GFLOPS_GUESS = 22.0   # one core of the oracle with OpenBLAS inside SSSSM (measured: 22-24 on the bench hosts), for sizing the sample only


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="elastic3d", choices=["shell", "fem27", "poisson", "kkt", "elastic3d"])
    ap.add_argument("--size", type=int, nargs="*", default=None, help="generator size arguments (fem27: n [ny nz]; shell: nx ny)")
    ap.add_argument("--mtx", default=None, help="matrix file to factorise instead of the synthetic stand-in: MatrixMarket (.mtx) or the "
                                                "reference's binary .lid (examples/example.c:112-163)")
    ap.add_argument("--rhs", default=None, help="right-hand side file (examples/example.c:167-243); default b = A*1")
    ap.add_argument("--nb", type=int, default=256)
    ap.add_argument("--ordering", default="nd", choices=["nd", "identity"])
    ap.add_argument("--no-coords", action="store_true", help="withhold the generator's mesh coordinates from the ordering: the graph-only "
                    "nested dissection (multilevel vertex separators) that a matrix file without coordinates gets")
    ap.add_argument("--host-threads", type=int, default=0, help="threads for the analysis phase (0: all cores / ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sched-steps", action="store_true", help="skip the three scheduler-in-the-loop steps behind the timed ones (profiling runs)")
    ap.add_argument("--multi-replay", action="store_true", help="(the default since round 5; kept for old command lines) N > 1: every rank logs its "
                    "first factorisation (a warm-up step) and replays the log afterwards; needs --warmup >= 1 and a transport that defers sends "
                    "(rccl or ipc) -- otherwise the ranks fall back to the scheduler by themselves")
    ap.add_argument("--no-multi-replay", action="store_true", help="N > 1: the scheduler in the loop in every step (PANGULU_AMD_MULTI_REPLAY=0)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads of the default run (the ldoor-class matrix of "
                    "BASELINE configs[1] and round 3's headline matrix, 3 steps each): their lines ride in the JSON line as `secondary`")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--cpu-sample-stride", type=int, default=0,
                    help="CPU baseline: execute every k-th task of each kernel class of the SAME factorisation (0: sized for about "
                         "12 s of one core from the structural flop count)")
    ap.add_argument("--cpu-leg-timeout", type=int, default=0, help="seconds a cpu_baseline child may take before it is given up "
                    "(0: 600 at --gpus 1, 150 at --gpus > 1; never more than what is left of --total-budget)")
    ap.add_argument("--cpu-ranks-leg", action="store_true", help="--gpus N > 1: also run the N ranks x 1 thread cpu_baseline leg (all blocks "
                    "exchanged host-staged: 119 GB on the default matrix at N = 8, never timed at full size).  Off by default: the contract asks "
                    "for the CPU baseline at N = 1 only, and an N > 1 run has to finish inside the driver's limit whatever happens")
    ap.add_argument("--total-budget", type=int, default=1500, help="seconds the whole run may take (the driver kills bench.py at 1800): every "
                    "phase -- transport attempts, the transport A/B, the cpu_baseline children -- gets what is left of it at most")
    ap.add_argument("--no-transport-ab", action="store_true", help="--gpus N > 1: skip the three steps on the OTHER device transport (ipc if rccl "
                    "ran, rccl if ipc ran) that are reported as `transport_ab`")
    ap.add_argument("--transport", default=os.environ.get("PANGULU_AMD_TRANSPORT", "auto"), choices=["auto", "host", "rccl", "ipc"],
                    help="block exchange for --gpus > 1: auto = rccl (ncclSend/ncclRecv per ordered pair over xGMI), else ipc (the "
                         "consumer pulls each record out of the owner's HBM arena with one peer copy), else host-staged TCP: each is "
                         "verified by a self-test at start-up and all ranks fall back together; the line says what ran")
    ap.add_argument("--worker-timeout", type=int, default=0, help="seconds a GPU worker (one transport attempt) may take before its supervisor "
                    "gives it up (0: 1100 at --gpus 1, where the worker also runs the secondary workloads; 400 at --gpus > 1, so that "
                    "three attempts + the cpu_baseline child fit --total-budget)")
    # internal: the GPU part of one rank (started by the supervisor below)
    ap.add_argument("--gpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--attempt", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--last-attempt", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--ab-run", action="store_true", help=argparse.SUPPRESS)  # the transport A/B worker: steps only, no profile pass
    # internal: one rank of a cpu_baseline leg (started by run_cpu_leg below)
    ap.add_argument("--cpu-leg", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-leg-port", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-leg-stride", type=int, default=1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def default_sizes(args):
    if args.size:
        return list(args.size)
    return {"shell": [398, 398], "fem27": [112], "poisson": [64], "kkt": [40], "elastic3d": [77]}[args.workload]


def make_matrix(args, M):
    if args.mtx:
        n, cp, ri, va, co = M.read_matrix(args.mtx)
        return (n, cp, ri, va, co), "file:%s" % os.path.basename(args.mtx)
    size = default_sizes(args)
    if args.workload == "shell":
        nx, ny = (size + [None])[:2]
        ny = ny or nx
        return M.shell(nx, ny), "ldoor-class stand-in: shell(%d,%d) 2 layers x 3 dofs" % (nx, ny)
    if args.workload == "fem27":
        return M.fem27(*size), "Serena-class stand-in: fem27(%s)" % ",".join(map(str, size))
    if args.workload == "poisson":
        return M.poisson3d(*size), "poisson3d(%s)" % ",".join(map(str, size))
    if args.workload == "elastic3d":
        return M.elastic3d(*size), "Serena-class stand-in with Serena's row length: elastic3d(%s), 3 dofs x 15-point node stencil" % ",".join(map(str, size))
    return M.kkt(size[0]), "nlpkkt-class stand-in: kkt(%d)" % size[0]


def workload_key(args):
    if args.mtx:
        return "file:%s" % os.path.basename(args.mtx)
    return "%s(%s) nb=%d %s%s" % (args.workload, ",".join(map(str, default_sizes(args))), args.nb, args.ordering, " no-coords" if args.no_coords else "")


def kernel_source_hash():
    """sha256 over the native sources (kernels and host): a committed PMC pass is only quoted for the build it profiled."""
    import hashlib

    hsh = hashlib.sha256()
    # (the host sources too: ordering, mapping and the scheduler decide which launches a workload is made of)
    for sub in ("platform", "host"):
        d = os.path.join(ROOT, "pangulu_amd", "csrc", sub)
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h", ".cpp")):
                hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def grid(world):
    """p x q process grid, p the largest divisor of the rank count not above its square root (src/pangulu.c:83-90)."""
    p = int(np.sqrt(world))
    while world % p:
        p -= 1
    return p, world // p


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def find_openblas():
    """The CPU baseline's SSSSM uses OpenBLAS dgemm like the reference (…0100000.c:317-327) when scipy's bundled
    library is present; otherwise the oracle's own triple loop."""
    try:
        import scipy

        d = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
        for f in sorted(os.listdir(d)):
            if "openblas" in f and f.endswith(".so") and "64_" not in f:
                return os.path.join(d, f)
    except Exception:
        pass
    return None


def free_port(lo=20000, hi=28000):
    """A port in a range the solver's own listeners (base + rank, + 64 per transport attempt) can share, free right now."""
    import random
    import socket

    rng = random.Random(os.getpid() * 7919 + int(time.time()))
    for _ in range(200):
        p = rng.randrange(lo, hi, 1024 // 4)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
            return p
        except OSError:
            continue
        finally:
            s.close()
    return lo + 1234


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: N fresh rank processes before this process has made any GPU call
# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("PG_BENCH_T0", repr(time.time()))  # the budget's clock starts here on every rank
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    if rc != 0:
        return rc
    return 0 if line else 1


# ---------------------------------------------------------------------------------------------------------------------
# cpu_baseline: the oracle on a bounded sample of the SAME factorisation, in child processes
# ---------------------------------------------------------------------------------------------------------------------
def cpu_leg_main(args):
    import faulthandler

    faulthandler.enable()
    return _cpu_leg_main(args)


def _cpu_leg_main(args):
    """One rank of a cpu_baseline leg (child process; never touches the GPU): the checker's build of the host routed to the
    oracle's CPU operators (oracle/pangulu_oracle.c, OpenBLAS dgemm inside SSSSM like the reference), `--cpu-leg` ranks x 1
    compute thread over the host-staged transport (the reference example's configuration, examples/example.c:284).  Every
    k-th task of each kernel class is executed in the scheduler's order, the others are only released: a kernel's time
    depends on the patterns of its operands, not on their values, so the sample is a 1/k cut through all levels of the
    elimination tree.  Rank 0 prints one JSON object."""
    import pangulu_amd as pa
    from pangulu_amd import _lib
    from pangulu_amd import matrices as M
    from tests.helpers import library_for, oracle_library

    R = args.cpu_leg
    rank = int(os.environ.get("PG_CPU_LEG_RANK", "0"))
    blas = find_openblas()
    if blas:
        os.environ["PANGULU_ORACLE_BLAS"] = blas
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ.setdefault("PANGULU_AMD_RECV_BUDGET_GB", "16")  # receive bins are host memory here
    tlib = library_for(oracle_library("r64"))  # the checker's build of the host, routed to the CPU restatement
    tlib.pangulu_amd_test_set_task_sampling.argtypes = [ctypes.c_int]
    if R > 1:
        assert tlib.pangulu_amd_comm_init(rank, R, b"127.0.0.1", args.cpu_leg_port, _lib.TRANSPORT_HOST, None) == 0
    if rank == 0:
        mat, workload = make_matrix(args, M)
        n, cp, ri, va, co = mat
    else:
        n, cp, ri, va, co, workload = 0, None, None, None, None, ""
    stride = max(1, args.cpu_leg_stride)
    tlib.pangulu_amd_test_set_task_sampling(stride)
    nthreads = max(1, (os.cpu_count() or 1) // R)
    h = pa.pangulu_init(n, len(va) if va is not None else 0, cp, ri, va, nb=args.nb, ordering=args.ordering,
                        coords=co if args.ordering == "nd" and not args.no_coords else None, nthread=nthreads, lib=tlib)
    t0 = time.time()
    pa.pangulu_gstrf(h)
    dt = time.time() - t0
    info = h.info()
    # sums / maxima over the ranks through the library's own reductions
    v = np.array([dt, info["time_numeric_platform"]], dtype=np.float64)
    if R > 1:
        tlib.pangulu_amd_comm_allreduce_max_f64(v.ctypes.data_as(ctypes.c_void_p), 2)
    fl = np.array([info["sampled_flop"] if stride > 1 else 0.0, float(info["sampled_tasks"])], dtype=np.float64)
    ntask = info["ntask_getrf"] + info["ntask_tstrf"] + info["ntask_gessm"] + info["ntask_ssssm"]
    tot = np.array([float(ntask)], dtype=np.float64)
    if R > 1:
        # (no sum reduction in the C API: gather through max of one-hot slots)
        slots = np.zeros(3 * R, dtype=np.float64)
        slots[3 * rank:3 * rank + 3] = [fl[0], fl[1], tot[0]]
        tlib.pangulu_amd_comm_allreduce_max_f64(slots.ctypes.data_as(ctypes.c_void_p), 3 * R)
        fl = np.array([slots[0::3].sum(), slots[1::3].sum()])
        tot = np.array([slots[2::3].sum()])
    pa.pangulu_finalize(h)
    if R > 1:
        tlib.pangulu_amd_comm_finalize()
    if rank == 0:
        fsample = fl[0] if stride > 1 else float(info["flop"])
        print(json.dumps({"cpu_leg": R, "stride": stride, "wall_s": float(v[0]), "platform_s": float(v[1]), "sampled_flop": fsample,
                          "sampled_tasks": int(fl[1]) if stride > 1 else int(tot[0]), "tasks": int(tot[0]), "flop": float(info["flop"]),
                          "blas": "OpenBLAS (scipy bundle)" if blas else "oracle triple loop", "workload": workload}), flush=True)


def run_cpu_leg(args, R, rank, port, stride):
    """Start this rank's child of an R-rank cpu_baseline leg; returns the Popen (rank 0's stdout carries the result)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", str(R), "--cpu-leg-port", str(port), "--cpu-leg-stride", str(stride),
           "--workload", args.workload, "--nb", str(args.nb), "--ordering", args.ordering]
    if args.no_coords:
        cmd.append("--no-coords")
    if args.size:
        cmd += ["--size"] + [str(s) for s in args.size]
    if args.mtx:
        cmd += ["--mtx", args.mtx]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PANGULU_AMD_HOST_THREADS")}
    env["PG_CPU_LEG_RANK"] = str(rank)
    env["HIP_VISIBLE_DEVICES"] = ""  # the child is CPU only
    env["ROCR_VISIBLE_DEVICES"] = ""
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)


def finish_cpu_leg(proc, timeout, want_result):
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"error": "cpu_baseline child exceeded %d s" % timeout}
    if proc.returncode != 0:
        return {"error": "cpu_baseline child failed (%d): %s" % (proc.returncode, (err or "")[-300:])}
    if not want_result:
        return None
    for ln in reversed(out.strip().splitlines()):
        if ln.startswith('{"cpu_leg"'):
            return json.loads(ln)
    return {"error": "cpu_baseline child printed no result"}


def leg_summary(res, cores, workload):
    """value = structural flops of the executed tasks / the time spent inside the platform calls that ran them (max over the
    ranks); the wall time of the sampled run -- scheduling and release of the skipped tasks and, for R > 1, the exchange of
    ALL blocks included -- is stated beside it."""
    if res is None or "error" in res:
        return {"value": None, "unit": "GFLOP/s", "cores": cores, "kind": "port", "sample": (res or {}).get("error", "not run")}
    k = res["stride"]
    return {
        "value": res["sampled_flop"] / max(res["platform_s"], 1e-9) / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
        "wall_value": res["sampled_flop"] / max(res["wall_s"], 1e-9) / 1e9,
        "sample": "same matrix, ordering and nb as the GPU line (%s): every %d%s task of each kernel class, %d of %d tasks, %.3e of %.3e "
                  "structural flops; %d rank(s) x 1 compute thread (examples/example.c:284), host-staged exchange; time base = seconds "
                  "inside the operator calls of the executed tasks, max over ranks (%.1f s; whole sampled run incl. release of the "
                  "skipped tasks%s: %.1f s); SSSSM GEMM: %s" % (
                      workload, k, "th" if k > 3 else ("st", "nd", "rd")[k - 1], res["sampled_tasks"], res["tasks"], res["sampled_flop"],
                      res["flop"], cores, res["platform_s"], " and exchange of all blocks" if cores > 1 else "", res["wall_s"], res["blas"]),
    }


# ---------------------------------------------------------------------------------------------------------------------
# supervisor: one per rank, never touches the GPU
# ---------------------------------------------------------------------------------------------------------------------
DONE_MARK = "__PG_WORKER_DONE__"
RC_TRANSPORT_UNAVAILABLE = 3


def parity_ok(*values):
    """True when every criterion is a number <= PARITY_TOL (a NaN or a missing value fails)."""
    return all(v is not None and v <= PARITY_TOL for v in values)


def passthrough_args(args):
    out = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--workload", args.workload, "--nb", str(args.nb),
           "--ordering", args.ordering, "--host-threads", str(args.host_threads)]
    if args.size:
        out += ["--size"] + [str(x) for x in args.size]
    if args.mtx:
        out += ["--mtx", args.mtx]
    if args.rhs:
        out += ["--rhs", args.rhs]
    if args.no_profile_pass:
        out.append("--no-profile-pass")
    if args.no_secondary:
        out.append("--no-secondary")
    if args.multi_replay:
        out.append("--multi-replay")
    if args.no_multi_replay:
        out.append("--no-multi-replay")
    if args.no_sched_steps:
        out.append("--no-sched-steps")
    if args.no_coords:
        out.append("--no-coords")
    return out


def run_worker(args, rank, transport, attempt, timeout, last=False, extra=()):
    """One GPU worker (a fresh child: the only process of this rank that touches the GPU) with one transport; returns what it printed."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpu-worker", "--transport", transport if transport != "none" else "auto",
           "--attempt", str(attempt)] + passthrough_args(args) + list(extra)
    if last:
        cmd.append("--last-attempt")
    t0 = time.time()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    done, line, meta = False, None, None
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        sys.stderr.write("[bench.py] rank %d: GPU worker with transport %s exceeded %d s\n" % (rank, transport, timeout))
    for ln in (out or "").splitlines():
        if ln.startswith('{"metric"'):
            line = json.loads(ln)
        elif ln.startswith('{"pg_worker"'):
            meta = json.loads(ln)["pg_worker"]
        elif ln.strip() == DONE_MARK:
            done = True
        elif ln.strip():
            sys.stderr.write(ln + "\n")
    return {"done": done, "line": line, "meta": meta, "rc": proc.returncode, "s": round(time.time() - t0, 1)}


def supervisor(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    args.gpus = world
    # THE TIME BUDGET (three attempts of 2400 s + CPU legs of 900 s against a driver that kills the command at 1800 s).
    # Everything below gets min(its own cap, what is left); the cpu_baseline child's share is set aside before the attempts start.
    t_begin = float(os.environ.get("PG_BENCH_T0", "0") or 0) or time.time()
    left = lambda: args.total_budget - (time.time() - t_begin)  # noqa: E731
    worker_cap = args.worker_timeout or (1100 if world == 1 else 400)
    cpu_cap = args.cpu_leg_timeout or (600 if world == 1 else 150)
    reserve = 0 if args.no_cpu_baseline else min(cpu_cap, 300 if world == 1 else cpu_cap) + 10
    order = {"auto": ["rccl", "ipc", "host"], "rccl": ["rccl"], "ipc": ["ipc"], "host": ["host"]}[args.transport] if world > 1 else ["none"]
    line, meta, attempts, ran = None, None, [], None
    for k, tr in enumerate(order):
        cap = int(min(worker_cap, max(30, left() - reserve)))  # (an attempt always gets its own cap at most, and 30 s at least of a spent budget)
        w = run_worker(args, rank, tr, k, cap, last=(k + 1 == len(order)))
        line, meta = w["line"], w["meta"]
        attempts.append({"transport": tr, "rc": w["rc"], "done": w["done"], "s": w["s"], "cap_s": cap})
        if w["done"] and (rank != 0 or line is not None):
            ran = tr
            break
        sys.stderr.write("[bench.py] rank %d: GPU worker with transport %s did not finish (rc %s)%s\n" % (
            rank, tr, w["rc"], "; trying the next transport" if k + 1 < len(order) else ""))
        line = None
    if line is None and rank == 0 or meta is None:
        sys.exit(1)

    # The OTHER device transport, three steps (VERDICT r5 next #2c): one SCALE run then says which data plane to keep.  Only when the
    # FIRST attempt succeeded -- every rank knows that (workers end behind a common barrier), so all supervisors take the same branch
    # without talking to each other -- and its failure changes nothing in the line but this object.
    ab = None
    if world > 1 and not args.no_transport_ab and len(attempts) == 1 and ran in ("rccl", "ipc") and args.steps > 0:
        other = "ipc" if ran == "rccl" else "rccl"
        cap = int(min(300, left() - reserve))
        if cap >= 60:
            w = run_worker(args, rank, other, 3, cap, last=False, extra=["--ab-run"])
            if rank == 0:
                if w["done"] and w["line"] is not None:
                    L = w["line"]
                    ab = {"transport": L["config"]["transport"], "steps": L["steps"], "warmup": L["warmup"], "ms_per_step": L["ms_per_step"],
                          "step_ms": L["step_ms"], "value": L["value"], "residual": L["residual"], "factor_check": L["factor_check"],
                          "headline_transport": ran, "headline_ms_per_step": line["ms_per_step"], "s": w["s"]}
                else:
                    ab = {"transport": other, "error": "worker did not finish (rc %s, %s s of %d)" % (w["rc"], w["s"], cap)}
        elif rank == 0:
            ab = {"transport": other, "error": "skipped: %d s left of the budget" % int(left())}

    # cpu_baseline LAST, in child processes (measured before the GPU steps, ten seconds of host-only work left them 15 %
    # slower).  One rank x one thread on rank 0 -- the contract's object (N = 1 only; at N > 1 it is reported for reference).
    # --cpu-ranks-leg adds R = N ranks x one thread (every rank starts its own child; SURVEY §8d, examples/example.c:284).
    cpu = None
    if not args.no_cpu_baseline:
        flop = float(meta["flop"])
        stride = args.cpu_sample_stride or max(1, int(round(flop / (CPU_GFLOPS_GUESS * 1e9 * 12.0))))
        legs = {}
        if world > 1 and args.cpu_ranks_leg:
            p = run_cpu_leg(args, world, rank, int(meta["base_port"]) + 700, stride)
            res = finish_cpu_leg(p, int(max(20, min(cpu_cap, left() - cpu_cap - 10))), rank == 0)
            if rank == 0:
                legs["ranks_x_1"] = leg_summary(res, world, meta["workload"])
        if rank == 0:
            p = run_cpu_leg(args, 1, 0, 0, stride)
            legs["1_x_1"] = leg_summary(finish_cpu_leg(p, int(max(20, min(cpu_cap, left() - 5))), True), 1, meta["workload"])
            # the contract's object = the leg with as many ranks as GPUs when it ran; the other one beside it
            cpu = dict(legs.get("ranks_x_1") or legs["1_x_1"])
            cpu["cpu_model"] = cpu_model_name()
            cpu["host_cores"] = os.cpu_count()
            if "ranks_x_1" in legs:
                cpu["one_rank_x_one_thread"] = legs["1_x_1"]
    if rank == 0:
        line["cpu_baseline"] = cpu
        line["config"]["worker_attempts"] = attempts
        if world > 1:
            line["transport_ab"] = ab
        line["bench_wall_s"] = round(time.time() - t_begin, 1)
        line["time_budget_s"] = args.total_budget
        print(json.dumps(line), flush=True)
        if line.get("parity_failed"):
            sys.stderr.write("[bench.py] the factors failed the parity gate (residual / factor_check > %g): no value is reported\n" % PARITY_TOL)
            sys.exit(RC_PARITY_FAILED)


def main():
    args = parse_args()
    if args.cpu_leg:
        return cpu_leg_main(args)
    if args.gpu_worker:
        return gpu_worker_main(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    return supervisor(args)


# ---------------------------------------------------------------------------------------------------------------------
# the GPU part of one rank
# ---------------------------------------------------------------------------------------------------------------------
def gpu_worker_main(args):
    import faulthandler

    faulthandler.enable()  # a rank that dies on a signal says where (a rank lost without a line of output)
    if args.ab_run:
        # the transport A/B: a few steps on the other data plane, nothing else
        args.steps, args.warmup = min(args.steps, 3), 2
        args.no_profile_pass = args.no_secondary = args.no_sched_steps = True
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus and "RANK" in os.environ:
        args.gpus = world
    fake = os.environ.get("PANGULU_BENCH_TEST_FAKE_WORKER")
    if fake:
        # tests/test_bench_supervisor.py (CPU): the supervisors' walk through the transports, their time budget and the transport A/B
        # with workers that touch no GPU -- "<transport>=hang" never finishes, "<transport>=fail" exits at once, anything else prints
        # a canned line for that transport
        behaviour = dict(kv.split("=") for kv in fake.split(",") if "=" in kv).get(args.transport, "ok")
        if behaviour == "hang":
            time.sleep(10 ** 6)
        if behaviour == "fail":
            sys.exit(RC_TRANSPORT_UNAVAILABLE)
        if rank == 0:
            print(json.dumps({"metric": "numeric factorisation GFLOP/s (pangulu_gstrf, R64)", "value": 1.0, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": 1.0, "step_ms": [1.0] * args.steps, "residual": 0.0, "factor_check": 0.0, "parity_failed": False,
                              "config": {"workload": "fake", "transport": args.transport}, "cpu_baseline": None}), flush=True)
        print(json.dumps({"pg_worker": {"flop": 1e9, "base_port": 20000, "workload": "fake"}}), flush=True)
        print(DONE_MARK, flush=True)
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # completion signals by polling instead of interrupts (must be set before the runtime starts): the scheduler and launcher
    # threads wait on hundreds of short events per factorisation; measured 44.2 ms (all 40 steps within 43.8-44.8) against
    # 44.6 ms (44.0-48.0) on one box.  A user of the library sets it the same way.
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
    os.environ["LOCAL_RANK"] = str(local_rank)
    if args.no_multi_replay:
        os.environ["PANGULU_AMD_MULTI_REPLAY"] = "0"
    elif args.multi_replay:
        os.environ["PANGULU_AMD_MULTI_REPLAY"] = "1"

    import torch  # device selection + the synchronise the contract asks for; not on the compute path

    import pangulu_amd as pa
    from pangulu_amd import _lib
    from pangulu_amd import matrices as M

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the numeric factorisation has no CPU fallback")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    lib = _lib.load("r64")
    nthreads = args.host_threads or max(1, (os.cpu_count() or 1) // max(1, world))
    os.environ["PANGULU_AMD_HOST_THREADS"] = str(nthreads)

    tried, comm_init_s, base_port = [], 0.0, 0
    if world > 1:
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        base_port = int(os.environ.get("MASTER_PORT", "29500")) + 23
        if base_port + 1024 >= 32768:  # keep the solver's listeners (base_port + rank, + 64 per attempt) out of the ephemeral port range
            base_port = 20000 + (base_port * 7) % 8000
        # one transport per worker: the supervisors walk the order rccl (north star: MPI point-to-point -> RCCL send/recv over
        # xGMI) -> ipc -> host.  Each device transport self-tests on every pair at start-up and all ranks agree on the outcome;
        # if it is not there the library falls back to host staging on all ranks together -- this worker then ends (all of
        # them do) and the supervisors start the next one, unless this was the last.
        codes = {"host": _lib.TRANSPORT_HOST, "rccl": _lib.TRANSPORT_RCCL, "ipc": _lib.TRANSPORT_IPC}
        name = args.transport if args.transport != "auto" else "rccl"
        t_comm = time.time()
        base_port += 128 * args.attempt
        rc = lib.pangulu_amd_comm_init(rank, world, addr.encode(), base_port, codes[name], None)
        assert rc == 0
        got = {0: "host", 1: "rccl", 2: "ipc"}[lib.pangulu_amd_comm_transport()]
        tried.append("%s->%s" % (name, got))
        if got != name and not args.last_attempt:
            lib.pangulu_amd_comm_barrier()
            sys.stderr.write("[bench.py] rank %d: transport %s is not available here (self-test failed on some rank)\n" % (rank, name))
            sys.stderr.flush()
            os._exit(RC_TRANSPORT_UNAVAILABLE)
        comm_init_s = time.time() - t_comm
        if os.environ.get("PANGULU_BENCH_TEST_HANG_TRANSPORT") == name:
            # (tests/test_gpu_smoke_bench.py: a transport that passes its self-test and then never finishes a step -- the supervisors
            #  have to give it up inside their budget and walk on)
            time.sleep(10 ** 6)

    if rank == 0:
        mat, workload = make_matrix(args, M)
        n, cp, ri, va, coords = mat
    else:
        mat, n, cp, ri, va, coords, workload = None, 0, None, None, None, None, ""
    # structural flop counting of MFMA-path updates costs an extra pass per task: off in the timed steps (F comes from
    # the symbolic pattern), on in the profile pass below.  Set BEFORE pangulu_init: on one rank the launch schedule is recorded
    # there, under the options in force, and a factorisation under other options runs the scheduler again.
    lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 0)
    t0 = time.time()
    h = pa.pangulu_init(n, len(va) if va is not None else 0, cp, ri, va, nb=args.nb, ordering=args.ordering,
                        coords=coords if args.ordering == "nd" and not args.no_coords else None, nthread=nthreads)
    t_init = time.time() - t0
    # what is really in use: ipc / rccl fall back to host staging on all ranks together when their self-test fails
    effective_transport = {0: "host", 1: "rccl", 2: "ipc"}[lib.pangulu_amd_comm_transport()] if world > 1 else "none"
    rccl_nranks = int(lib.pangulu_amd_comm_rccl_ranks()) if world > 1 else 0
    assert lib.pangulu_amd_snapshot(h.ref) == 0
    info0 = h.info()
    flop = float(info0["flop"])

    def one_step(handle=None):
        lib.pangulu_amd_comm_barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        pa.pangulu_gstrf(handle or h)  # ends with a stream synchronise + barrier inside the library
        torch.cuda.synchronize()
        lib.pangulu_amd_comm_barrier()
        return time.perf_counter() - t

    # (COUNT_FLOPS is off since before pangulu_init; tests/test_gpu_env_switches.py runs the parity cases in this
    # configuration too, and the line's residual / factor_check come from the last TIMED step)
    first_ms = None
    for w_ in range(args.warmup):
        t_ = one_step()
        if w_ == 0:
            first_ms = 1e3 * t_  # N > 1: the scheduler in the loop + the log the later steps replay; N = 1: already a replay (recorded in pangulu_init)
        lib.pangulu_amd_reset_numeric(h.ref)
    pa.hip_stats(lib, reset=True)
    times = []
    for s in range(args.steps):
        times.append(one_step())
        if s + 1 < args.steps:
            lib.pangulu_amd_reset_numeric(h.ref)
    # max over ranks of the summed step time
    tsum = np.array([sum(times)], dtype=np.float64)
    if world > 1:
        lib.pangulu_amd_comm_allreduce_max_f64(tsum.ctypes.data_as(ctypes.c_void_p), 1)
    ms_per_step = float(tsum[0]) / max(1, args.steps) * 1e3
    info = h.info()
    used = ctypes.c_size_t(0)
    lib.pangulu_platform_0201001_get_device_memory_usage(ctypes.byref(used))  # records + receive bins + mirror pool + snapshot
    mem = pa.hip_memory(lib)

    # the two correctness criteria of the reference on the factors the last timed step left on the device(s)
    # ||L(U x) - A x|| / ||A x||: x = 1 (src/pangulu_numeric.c:1082-1341) and CHECK_VECTORS - 1 random +-1 vectors, the worst of them
    factor_check = pa.factor_check_vectors(h, CHECK_VECTORS) if args.steps > 0 else None
    residual = None
    if args.steps > 0:
        if rank == 0:
            b = M.read_rhs(args.rhs, n) if args.rhs else M.rhs_of_ones(n, cp, ri, va)
        else:
            b = None
        lib.pangulu_amd_comm_barrier()
        t_solve0 = time.perf_counter()
        x = pa.pangulu_gstrs(h, b)                                  # ||Ax - b|| / ||b||, b = A*1 (examples/example.c:252-264,304-364)
        gstrs_s = time.perf_counter() - t_solve0
        if rank == 0:
            va_check = va
            if os.environ.get("PANGULU_BENCH_TEST_BREAK_FACTORS"):
                # (tests/test_gpu_smoke_bench.py: the gate itself under test -- the residual is taken against a matrix whose largest
                #  entry was changed behind the factorisation's back, i.e. the factors are those of the wrong matrix)
                va_check = np.array(va, copy=True)
                va_check[int(np.argmax(np.abs(va_check)))] *= 1.5
            residual = M.relative_residual(n, cp, ri, va_check, x, b)
    pa.hip_stats(lib, reset=True)

    # One rank replays its recorded launch schedule; N > 1 ranks have to run the scheduler beside the device (arrival order is
    # dynamic).  So that a scaling curve compares like with like, a few un-timed-for-the-metric steps go through the scheduler
    # here as well, and the line carries both numbers.
    ms_scheduler_in_loop, sched_step_ms = None, None
    if world == 1 and args.steps > 0 and info.get("replayed") and not args.no_sched_steps:
        before = lib.pangulu_amd_set_replay(0)
        ts = []
        for s in range(4):
            lib.pangulu_amd_reset_numeric(h.ref)
            ts.append(one_step())
        lib.pangulu_amd_set_replay(before)
        # the same statistic as ms_per_step (mean of the timed steps); the first step is this mode's warm-up (it re-creates the
        # launcher thread's buffers)
        sched_step_ms = [round(1e3 * t, 2) for t in ts[1:]]
        ms_scheduler_in_loop = 1e3 * sum(ts[1:]) / len(ts[1:])
        pa.hip_stats(lib, reset=True)  # (the profile pass below counts its own launches only)

    # one extra, un-timed factorisation with per-launch hipEvents to attribute time to kernels: every launch on the ONE main
    # stream (side streams and the records stream off), so that an event pair brackets its kernel and nothing else
    roofline = None
    kernels = {}
    if not args.no_profile_pass and args.steps > 0:
        lib.pangulu_amd_reset_numeric(h.ref)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_PROFILE, 1)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 1)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_TWO_STREAMS, 0)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_RECORDS_STREAM, 0)
        one_step()
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_PROFILE, 0)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 0)  # (back to the timed configuration: the secondary workload records under it)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_TWO_STREAMS, 1)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_RECORDS_STREAM, 1)
        st = pa.hip_stats(lib, reset=True)
        for name, v in st.items():
            if v["launches"]:
                kernels[name] = {
                    "launches": v["launches"], "tasks": v["tasks"], "ms": round(v["elapsed_ms"], 3),
                    "avg_launch_us": round(1e3 * v["elapsed_ms"] / v["launches"], 2),
                    "alg_GB": round(v["alg_bytes"] / 1e9, 4), "GFLOP": round(v["flops"] / 1e9, 4),
                }
                if name == "ssssm_dense_mfma":
                    kernels[name]["GFLOP_executed"] = round(v["mfma_flops_executed"] / 1e9, 2)
                    kernels[name]["workgroups"] = {"dense_front_kernel": v["front_workgroups"], "general_kernel": v["general_workgroups"]}
        task_classes = ("getrf", "tstrf", "gessm", "ssssm_sparse", "ssssm_dense_mfma")
        if any(k in kernels for k in task_classes):
            # the dominant kernel among the task classes (mirror maintenance -- densify, sparsify, remote LU images -- is listed
            # in `kernels` and counted in the denominator of share_of_kernel_time, but has no algorithmic bytes or flops)
            dom = max((k for k in kernels if k in task_classes), key=lambda k: kernels[k]["ms"])
            v = st[dom]
            sec = v["elapsed_ms"] / 1e3
            if dom == "ssssm_dense_mfma":
                ach = v["flops"] / sec / 1e12
                roofline = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / FP64_PEAK_TFLOPS, "traffic": None,
                            # `achieved` counts the structural (algorithmic) flops of the tasks; the matrix cores execute
                            # whole 16x16x16 tile products wherever both operand tiles hold pattern entries, at this rate:
                            "mfma_executed_tflops": v["mfma_flops_executed"] / sec / 1e12}
            else:
                ach = v["alg_bytes"] / sec / 1e9
                roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "traffic": None}
            roofline["this_rank_only"] = world > 1
            tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            # HBM traffic per launch of that kernel: PMC counters cannot be read from inside this process, so the value comes
            # from a committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass (tools/profile_recipe.sh; FETCH_SIZE doubled as
            # MI355X_MICROARCH.md prescribes for gfx950) -- but ONLY if that pass profiled this build's kernels on this
            # workload: the file records the workload and a hash of the kernel sources, anything else leaves `traffic` null
            # (a launch of the MFMA update class is one launch of the general kernel plus, from 8192 dense-front workgroups on,
            #  one of the dense-front kernel: their bytes are added up per launch of the class)
            rocprof_names = {"ssssm_dense_mfma": [GENERAL_KERNEL_ROCPROF_NAME, "ssssm_front_f64_kernel<2, true>"], "getrf": ["getrf_pipe_f64_kernel<16>"],
                             "tstrf": ["trsm_dense_ring_f64_kernel<16>"], "gessm": ["trsm_dense_ring_f64_kernel<16>"],
                             "ssssm_sparse": ["ssssm_sparse_kernel<false>"]}.get(dom)
            if world == 1 and rocprof_names and os.path.exists(tfile):
                tj = json.load(open(tfile)).get(workload_key(args))
                if tj and tj.get("kernel_source_hash") == kernel_source_hash() and tj.get(rocprof_names[0]):
                    total = sum(tj[k]["hbm_bytes_per_launch"] * tj[k]["calls"] for k in rocprof_names if tj.get(k))
                    roofline["traffic"] = total / tj[rocprof_names[0]]["calls"]
                    roofline["traffic_unit"] = "HBM bytes per launch of the class (rocprofv3 PMC passes of this build on this workload, %s; FETCH_SIZE doubled per MI355X_MICROARCH.md)" % tj.get("profile", "profiles/")
                    roofline["traffic_over_algorithmic"] = roofline["traffic"] / (v["alg_bytes"] / max(1, v["launches"]))
                else:
                    roofline["traffic_note"] = "no PMC pass of this build on this workload committed (profiles/hbm_traffic.json)"
            # class by class and, inside the MFMA update class, kernel by kernel.  `structural` = the reference's flop
            # count of the tasks, `executed` = 16 x 16 x 16 products issued x 8192 (device counters of this pass); busy % and bytes per
            # launch ride along from the committed PMC pass when it is of this build and workload.
            tj_all = {}
            if world == 1 and os.path.exists(tfile):
                tj_all = json.load(open(tfile)).get(workload_key(args)) or {}
                if tj_all.get("kernel_source_hash") != kernel_source_hash():
                    tj_all = {}

            def pmc(name):
                e = tj_all.get(name) or {}
                return {"rocprof_kernel": name, "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "mfma_busy_pct": e.get("mfma_busy_pct"),
                        "rocprof_avg_us": e.get("avg_us"), "rocprof_calls": e.get("calls")}

            def rate(fl, ms):
                return fl / (ms / 1e3) / 1e12 if ms else None

            per = {}
            d5 = st["ssssm_dense_mfma"]
            if d5["launches"]:
                fr_ms, ge_ms = d5["front_kernel_ms"], d5["general_kernel_ms"]
                fr_fl = d5["front_flops_executed"]
                ge_fl = d5["mfma_flops_executed"] - fr_fl
                per["ssssm_front"] = dict(ms=round(fr_ms, 3), executed_TFLOPs=rate(fr_fl, fr_ms), executed_frac_of_peak=(rate(fr_fl, fr_ms) or 0) / FP64_PEAK_TFLOPS,
                                          workgroups=d5["front_workgroups"], **pmc("ssssm_front_f64_kernel<2, true>"))
                per["ssssm_general"] = dict(ms=round(ge_ms, 3), executed_TFLOPs=rate(ge_fl, ge_ms), executed_frac_of_peak=(rate(ge_fl, ge_ms) or 0) / FP64_PEAK_TFLOPS,
                                            workgroups=d5["general_workgroups"], **pmc(GENERAL_KERNEL_ROCPROF_NAME))
                per["ssssm_class"] = dict(ms=round(d5["elapsed_ms"], 3), structural_TFLOPs=rate(d5["flops"], d5["elapsed_ms"]),
                                          executed_TFLOPs=rate(d5["mfma_flops_executed"], d5["elapsed_ms"]),
                                          executed_over_structural=d5["mfma_flops_executed"] / d5["flops"] if d5["flops"] else None)
            tr_ms = st["tstrf"]["elapsed_ms"] + st["gessm"]["elapsed_ms"]
            if tr_ms:
                tr_fl = st["tstrf"]["flops"] + st["gessm"]["flops"]
                per["trsm"] = dict(ms=round(tr_ms, 3), structural_TFLOPs=rate(tr_fl, tr_ms), structural_frac_of_peak=(rate(tr_fl, tr_ms) or 0) / FP64_PEAK_TFLOPS,
                                   alg_GBs=(st["tstrf"]["alg_bytes"] + st["gessm"]["alg_bytes"]) / (tr_ms / 1e3) / 1e9,
                                   launches=max(st["tstrf"]["launches"], st["gessm"]["launches"]), **pmc("trsm_dense_ring_f64_kernel<16>"))
            if st["getrf"]["elapsed_ms"]:
                g = st["getrf"]
                per["getrf"] = dict(ms=round(g["elapsed_ms"], 3), structural_TFLOPs=rate(g["flops"], g["elapsed_ms"]),
                                    structural_frac_of_peak=(rate(g["flops"], g["elapsed_ms"]) or 0) / FP64_PEAK_TFLOPS,
                                    avg_launch_us=round(1e3 * g["elapsed_ms"] / g["launches"], 2), **pmc("getrf_pipe_f64_kernel<16>"))
            roofline["per_kernel"] = per
            roofline["per_kernel_note"] = ("ms: hipEvent pairs of the profile pass (one stream); a solve launch carries TSTRF and GESSM tasks together and is "
                                           "listed once here, split by algorithmic bytes in `kernels`; PMC fields null = no pass of this build committed")
            roofline["avg_launch_us"] = kernels[dom]["avg_launch_us"]
            roofline["share_of_kernel_time"] = kernels[dom]["ms"] / sum(k["ms"] for k in kernels.values())
            roofline["kernel_times"] = "hipEvent pairs around each launch with every launch on one stream (no queueing inside a pair)"
    # Whole-factorisation bound of SURVEY.md §8d for THIS rank count: T*_r = sum over the tasks rank r runs of max(bytes_t / 8 TB/s,
    # flop_t / 78.6 TF), from the symbolic pattern alone (pg_model.cpp, evaluated by every rank at init); T*(N) = max_r (T*_r +
    # bytes rank r sends to its busiest peer / 153 GB/s).  At N = 1 this is round 2's model_T_star.
    model = {
        "T_star_ms": 1e3 * info0["model_ranks_tstar_max"], "T_star_over_t_gstrf": 1e3 * info0["model_ranks_tstar_max"] / ms_per_step if ms_per_step else None,
        "sum_over_ranks_ms": 1e3 * info0["model_ranks_tstar_sum"],
        "split_ms": {"hbm_bound_tasks": 1e3 * info0["model_ranks_tstar_hbm"], "mfma_bound_tasks": 1e3 * info0["model_ranks_tstar_fp"]},
        "alg_GB": info0["model_ranks_bytes_total"] / 1e9,
        "rank_flop_share_max_over_mean": info0["model_rank_flop_share"], "rank_T_star_share_max_over_mean": info0["model_rank_time_share"],
        "link_term_ms_max": 1e3 * info0["model_comm_seconds_max"], "sent_GB": info0["model_sent_bytes_total"] / 1e9,
        "critical_path_ms": 1e3 * info0["model_critical_path"], "critical_path_tasks": int(info0["model_critical_path_tasks"]),
        "peaks": {"hbm_GBs": HBM_PEAK_GBS, "fp64_TFLOPs": FP64_PEAK_TFLOPS, "xgmi_link_GBs": XGMI_LINK_GBS},
    }
    # What the structure says about 1 / 2 / 4 / 8 ranks (pangulu_amd_model_for_ranks: the mapping, the consumer sets and the rank
    # model for each count, on this handle's replicated pattern): T*(N), the link term, the LATENCY-AWARE chain (every task at
    # max(T*_t, the measured floor of a lone launch of its class), a hop per operand from another rank) and the HBM of the fullest
    # rank.  `bound_ms` = max(T*(N), chain): no schedule is faster; `calibrated_ms` = max(T*(N) / (T*(1) / t measured here), chain).
    model["latency_chain_ms"] = 1e3 * info0.get("model_critical_path_latency", 0.0)
    model["hbm_fullest_rank_GB"] = {"total": info0.get("model_rank_hbm_bytes_max", 0.0) / 1e9, "records_owned": info0.get("model_rank_hbm_records", 0.0) / 1e9,
                                    "records_received": info0.get("model_rank_hbm_received", 0.0) / 1e9, "dense_mirrors": info0.get("model_rank_hbm_mirrors", 0.0) / 1e9}
    if rank == 0 and world == 1:
        pred = {}
        eff1 = (1e3 * info0["model_ranks_tstar_max"] / ms_per_step) if ms_per_step else None
        for N in (1, 2, 4, 8):
            m = pa.model_for_ranks(h, N)
            if m is None:
                continue
            bound = 1e3 * max(m["T_star_s"], m["latency_chain_s"])
            pred[str(N)] = {"T_star_ms": 1e3 * m["T_star_s"], "link_term_ms": 1e3 * m["link_term_s_max"], "latency_chain_ms": 1e3 * m["latency_chain_s"],
                            "bound_ms": bound, "calibrated_ms": max(1e3 * m["T_star_s"] / eff1, 1e3 * m["latency_chain_s"]) if eff1 else None,
                            "sent_GB": m["sent_bytes"] / 1e9, "rank_flop_share": m["rank_flop_share"],
                            "hbm_fullest_rank_GB": m["hbm_bytes_fullest_rank"] / 1e9}
        model["scaling_prediction"] = pred
        model["scaling_prediction_note"] = ("structure only; launch floors GETRF 138 us, dense panel solve 47 us, update launch 25 us at nb = 256 (measured lone launches of this build's kernels, DESIGN.md §4.4-4.5), "
                                            "20 us per hop between ranks (an ASSUMPTION until a run on real links calibrates it)")
    if roofline is not None:
        roofline["model_T_star_ms"] = model["T_star_ms"]
        roofline["model_T_star_over_t_gstrf"] = model["T_star_over_t_gstrf"]
        roofline["model_split_ms"] = model["split_ms"]
        roofline["model_alg_GB"] = model["alg_GB"]

    pa.pangulu_finalize(h)

    # Secondary workloads of the DEFAULT run (one rank, default matrix), 3 timed steps behind 1 warm-up each, residual and factor check
    # from the last one: (1) BASELINE configs[1]'s class -- ldoor, n = 952 K, one GPU -- on its stand-in shell(398,398): a latency-bound
    # matrix (thin shell: small fronts, long chains near the root) beside the MFMA-bound headline one; (2) fem27(112), the headline
    # matrix of round 3 (27 entries per row where Serena has 46), for continuity between the rounds' lines.
    secondary = None
    if world == 1 and not args.no_secondary and not args.mtx and not args.size and args.workload == "elastic3d" and not args.no_coords and args.steps > 0:
        secondary = []
        for label, gen in (("ldoor-class stand-in: shell(398,398) 2 layers x 3 dofs", lambda: M.shell(398, 398)),
                           ("Serena-class stand-in of round 3: fem27(112)", lambda: M.fem27(112))):
            n2, cp2, ri2, va2, co2 = gen()
            t0 = time.time()
            h2 = pa.pangulu_init(n2, len(va2), cp2, ri2, va2, nb=args.nb, ordering="nd", coords=co2, nthread=nthreads)
            t_init2 = time.time() - t0
            assert lib.pangulu_amd_snapshot(h2.ref) == 0
            ts2 = []
            for s2 in range(4):
                ts2.append(one_step(h2))  # (the same bracket as the headline steps: barrier + synchronise on both sides)
                if s2 < 3:
                    lib.pangulu_amd_reset_numeric(h2.ref)
            info2 = h2.info()
            fc2 = pa.factor_check_vectors(h2, CHECK_VECTORS)
            b2 = M.rhs_of_ones(n2, cp2, ri2, va2)
            t = time.perf_counter()
            x2 = pa.pangulu_gstrs(h2, b2)
            gstrs2 = time.perf_counter() - t
            ms2 = 1e3 * sum(ts2[1:]) / 3
            res2 = M.relative_residual(n2, cp2, ri2, va2, x2, b2)
            ok2 = parity_ok(res2, fc2)
            # ... and with the scheduler in the loop, like the headline: one warm-up + three steps
            sched2 = None
            if info2.get("replayed") and not args.no_sched_steps:
                before2 = lib.pangulu_amd_set_replay(0)
                tq = []
                for _ in range(4):
                    lib.pangulu_amd_reset_numeric(h2.ref)
                    tq.append(one_step(h2))
                lib.pangulu_amd_set_replay(before2)
                sched2 = 1e3 * sum(tq[1:]) / 3
            tstar2 = 1e3 * info2["model_ranks_tstar_max"]
            secondary.append({"workload": label, "n": int(info2["n"]), "nnz": int(info2["nnz"]), "nb": int(info2["nb"]),
                              "flop": int(info2["flop"]), "steps": 3, "warmup": 1, "ms_per_step": ms2, "step_ms": [round(1e3 * t_, 2) for t_ in ts2[1:]],
                              "value": float(info2["flop"]) / (ms2 / 1e3) / 1e9 if ok2 else None, "unit": "GFLOP/s", "residual": res2,
                              "factor_check": fc2, "parity_failed": not ok2, "gstrs_s": gstrs2, "init_s": round(t_init2, 2),
                              "static_schedule_replayed": bool(info2["replayed"]), "ms_per_step_scheduler_in_loop": sched2,
                              "model_T_star_ms": tstar2, "model_T_star_over_t_gstrf": tstar2 / ms2 if ms2 else None,
                              "latency_chain_ms": 1e3 * info2.get("model_critical_path_latency", 0.0)})
            pa.pangulu_finalize(h2)
            del n2, cp2, ri2, va2, co2, b2, x2
    if world > 1:
        lib.pangulu_amd_comm_barrier()
        lib.pangulu_amd_comm_finalize()

    if rank == 0:
        # THE GATE: a factorisation whose factors fail either criterion of the reference -- on the headline matrix or on a secondary
        # one -- publishes no number (a silent wrong-result bug lived behind printed-but-unchecked residuals)
        failed = args.steps > 0 and (not parity_ok(residual, factor_check) or any(w["parity_failed"] for w in (secondary or [])))
        value = flop / (ms_per_step / 1e3) / 1e9 if ms_per_step and not failed else (None if failed else 0.0)
        sep_map = os.environ.get("PANGULU_AMD_SEPARATOR_MAP", "group")
        line = {
            "metric": "numeric factorisation GFLOP/s (pangulu_gstrf, R64)",
            "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "parity_failed": bool(failed), "parity_tol": PARITY_TOL,
            "ms_per_step": ms_per_step,
            # the handle's FIRST pangulu_gstrf (the first warm-up step; null with --warmup 0).  N > 1: the host scheduler runs beside the
            # devices and logs what the later steps replay -- what a user who factorises once gets.  N = 1: the launch schedule was
            # recorded inside pangulu_init (schedule_record_s), so this step replays too; ms_per_step_scheduler_in_loop is its counterpart
            "ms_per_step_first_factorisation": first_ms,
            "ms_per_step_scheduler_in_loop": ms_scheduler_in_loop, "step_ms_scheduler_in_loop": sched_step_ms, "gstrs_s": gstrs_s if args.steps > 0 else None,
            "step_ms": [round(1e3 * t, 2) for t in times], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not args.mtx else "file",
            "config": {
                "workload": workload, "n": int(info["n"]), "nnz": int(info["nnz"]), "nb": int(info["nb"]),
                "ordering": "identity" if args.ordering != "nd" else
                            ("built-in nested dissection (geometric: median cuts along 13 lattice directions, FM-refined)" if coords is not None and not args.no_coords else
                             "built-in nested dissection (graph only: multilevel vertex separators, no coordinates)"),
                "symbolic_nnz": int(info["symbolic_nnz"]), "flop": int(info["flop"]),
                "parallelism": ("one rank") if world == 1 else
                               {"group": "proportional mapping of the block elimination tree with rank groups that shrink down the tree: subtrees whose group "
                                         "is one rank live on it whole, heavy separators 2D block-cyclic over their group's p x q grid (all %d ranks: %dx%d), "
                                         "light ones on the least loaded rank of their group" % ((world,) + grid(world)),
                                "path": "subtrees on single ranks; separators on the rank of their heaviest child", "rank0": "subtrees on single ranks; separators on rank 0",
                                "cyclic": "subtrees on single ranks; separators 2D block-cyclic %dx%d" % grid(world)}.get(sep_map, sep_map),
                # what is really in use: ipc / rccl fall back to host staging on all ranks when their self-test fails
                "transport": effective_transport, "transport_tried": tried,
                "rccl_nranks": rccl_nranks,   # ranks whose RCCL communicators passed the start-up self-test (from the library)
                "visible_gpus": ndev, "comm_init_s": round(comm_init_s, 2),
                "blocks": int(info["nblocks_nondiag"]),
                "rank_flop_share": model["rank_flop_share_max_over_mean"],
                "tasks_rank0": {"getrf": int(info["ntask_getrf"]), "tstrf": int(info["ntask_tstrf"]), "gessm": int(info["ntask_gessm"]),
                                "ssssm": int(info["ntask_ssssm"])},
            },
            "residual": residual, "factor_check": factor_check,
            "checked": "residual and factor_check are from the factors of the last timed step (timed configuration); factor_check = the worst of "
                       "%d vectors (all ones + seeded random +-1); either above %g => value null, parity_failed, exit code %d" % (CHECK_VECTORS, PARITY_TOL, RC_PARITY_FAILED),
            "factor_check_vectors": CHECK_VECTORS,
            "init_s": round(t_init, 2),
            "hbm_used_GB": round(used.value / 1e9, 2), "owned_records_GB": round(info["owned_bytes"] / 1e9, 2),
            # where the memory in use is: the records (authoritative form of every block), bench.py's own device-side snapshot of
            # them (restored between steps; a user has none), the dense mirrors, the recorded schedule's descriptors
            "hbm_breakdown_GB": {"records": round(info["owned_bytes"] / 1e9, 2), "bench_snapshot_of_records": round(info["snapshot_device_bytes"] / 1e9, 2),
                                 "dense_mirror_pool": round(mem["mirror_pool_bytes"] / 1e9, 2), "dense_mode_blocks": mem["dense_mode_blocks"],
                                 "schedule_descriptors": round(mem["schedule_descriptor_bytes"] / 1e9, 2),
                                 "getrf_scratch": round(mem["getrf_scratch_bytes"] / 1e9, 2)},
            "host_sched_s_last_step": round(info["time_numeric_host_sched"], 4),
            # one rank: the first pangulu_gstrf of the handle (a warm-up step) recorded its launches, the timed steps replay the list
            "static_schedule_replayed": bool(info["replayed"]), "schedule_record_s": round(info["time_schedule_record"], 2),
            "batches_per_step": int(info["batches"]),
            # destinations a background update call left alone because their queues were shallow (pg_numeric.cpp, round 5), in the run that scheduled
            "deferred_queues_per_step": int(info.get("deferred_queues", 0)),
            "roofline": roofline,
            "model": model,
            "kernels": kernels,
            "secondary": secondary,
            "cpu_baseline": None,  # (filled in by the supervisor)
        }
        print(json.dumps(line), flush=True)
    # every rank: what the supervisor needs for the cpu_baseline legs, and the mark that this worker ended behind the final barrier
    print(json.dumps({"pg_worker": {"flop": flop, "base_port": base_port, "workload": workload}}), flush=True)
    print(DONE_MARK, flush=True)


if __name__ == "__main__":
    main()
