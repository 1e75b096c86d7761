#!/usr/bin/env python3
"""bench.py -- numeric factorisation GFLOP/s (pangulu_gstrf, R64) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one complete pangulu_gstrf of the workload matrix.  Between steps the block records are restored from a
device-side snapshot (pangulu_amd_reset_numeric, un-timed), so every timed step starts with its inputs resident in
HBM; each step is bracketed by a barrier + device synchronise on both sides and the slowest rank's time counts.
value = F / t with F = sum_k (c_k + 2 c_k^2) the reference's structural flop count (src/pangulu_kernel_interface.c:4-176,
computed once from the symbolic pattern outside the timed region).

Workload (BASELINE.json configs[1]): "SuiteSparse ldoor (n=952K, nnz=42M) R64, nb=256".  ldoor is not in the image and
there is no network, so unless --mtx points at a MatrixMarket file the run uses the deterministic stand-in
pangulu_amd.matrices.shell(398, 398): a two-layer structural shell with 3 unknowns per node, n = 950 424,
~50 M entries, diagonally dominant -- the same class (thin-walled structure, ~45-55 entries per row) and size.
Ordering: built-in geometric nested dissection (stated in the JSON line; F depends on it).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak (v_mfma_f64_16x16x4: 64 cycles per 2048 flop per SIMD)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="shell", choices=["shell", "fem27", "poisson", "kkt"])
    ap.add_argument("--size", type=int, nargs="*", default=None, help="generator size arguments (shell: nx ny)")
    ap.add_argument("--mtx", default=None, help="matrix file to factorise instead of the synthetic stand-in: MatrixMarket (.mtx) or the "
                                                "reference's binary .lid (examples/example.c:112-163)")
    ap.add_argument("--rhs", default=None, help="right-hand side file (examples/example.c:167-243); default b = A*1")
    ap.add_argument("--nb", type=int, default=256)
    ap.add_argument("--ordering", default="nd", choices=["nd", "identity"])
    ap.add_argument("--host-threads", type=int, default=0, help="threads for the analysis phase (0: all cores / ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--cpu-sample-stride", type=int, default=6,
                    help="CPU baseline: execute every k-th task of each kernel class of the SAME factorisation (about 15 s of one core at 6)")
    ap.add_argument("--transport", default=os.environ.get("PANGULU_AMD_TRANSPORT", "auto"), choices=["auto", "host", "rccl", "ipc"],
                    help="block exchange for --gpus > 1: auto = rccl (ncclSend/ncclRecv per ordered pair over xGMI), else ipc (the "
                         "consumer pulls each record out of the owner's HBM arena with one peer copy), else host-staged TCP: each is "
                         "verified by a self-test at start-up and all ranks fall back together; the line says what ran")
    return ap.parse_args()


def make_matrix(args, M):
    if args.mtx:
        n, cp, ri, va, co = M.read_matrix(args.mtx)
        return (n, cp, ri, va, co), "file:%s" % os.path.basename(args.mtx)
    size = args.size
    if args.workload == "shell":
        nx, ny = (size + [None, None])[:2] if size else (398, 398)
        ny = ny or nx
        return M.shell(nx, ny), "ldoor-class stand-in: shell(%d,%d) 2 layers x 3 dofs" % (nx, ny)
    if args.workload == "fem27":
        s = size or [64]
        return M.fem27(*s), "Serena-class stand-in: fem27(%s)" % ",".join(map(str, s))
    if args.workload == "poisson":
        s = size or [64]
        return M.poisson3d(*s), "poisson3d(%s)" % ",".join(map(str, s))
    s = size or [40]
    return M.kkt(s[0]), "nlpkkt-class stand-in: kkt(%d)" % s[0]


def kernel_source_hash():
    """sha256 over the HIP kernel sources: a committed PMC pass is only quoted for the build it profiled."""
    import hashlib

    d = os.path.join(ROOT, "pangulu_amd", "csrc", "platform")
    hsh = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def grid(world):
    """p x q process grid, p the largest divisor of the rank count not above its square root (src/pangulu.c:83-90)."""
    p = int(np.sqrt(world))
    while world % p:
        p -= 1
    return p, world // p


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


def cpu_baseline(args, pa, M, mat, workload):
    """Oracle (CPU restatement of the reference's CPU platform, OpenBLAS dgemm inside SSSSM like the reference) timed on one
    host core on a bounded sample of the SAME factorisation: same matrix, ordering and nb; every k-th task of each kernel
    class is executed (in the scheduler's order), the others are only released.  A kernel's time depends on the patterns
    of its operands, not on their values, so the sample is a 1/k cut through all levels of the elimination tree.
    value = structural flops of the executed tasks / their time.  Reported beside the GPU number; not a target."""
    from tests.helpers import library_for, oracle_library

    blas = find_openblas()
    if blas:
        os.environ["PANGULU_ORACLE_BLAS"] = blas
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    n, cp, ri, va, co = mat
    tlib = library_for(oracle_library("r64"))  # the checker's build of the host, routed to the CPU restatement
    tlib.pangulu_amd_test_set_task_sampling.argtypes = [ctypes.c_int]
    stride = max(1, args.cpu_sample_stride)
    tlib.pangulu_amd_test_set_task_sampling(stride)
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=args.nb, ordering=args.ordering, coords=co if args.ordering == "nd" else None,
                        nthread=max(1, os.cpu_count() or 1), lib=tlib)
    t0 = time.time()
    pa.pangulu_gstrf(h)
    dt = time.time() - t0
    info = h.info()
    pa.pangulu_finalize(h)
    tlib.pangulu_amd_test_set_task_sampling(1)
    ntask = info["ntask_getrf"] + info["ntask_tstrf"] + info["ntask_gessm"] + info["ntask_ssssm"]
    fsample = info["sampled_flop"] if stride > 1 else float(info["flop"])
    return {
        "value": fsample / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
        "sample": "same matrix, ordering and nb as the GPU line (%s): every %d%s task of each kernel class, %d of %d tasks, "
                  "%.3e of %.3e structural flops, %.1f s; 1 rank x 1 compute thread (= R x 1 at R = 1 GPU), SSSSM GEMM: %s" % (
                      workload, stride, "th" if stride > 3 else ("st", "nd", "rd")[stride - 1] if stride <= 3 else "th",
                      info["sampled_tasks"] if stride > 1 else ntask, ntask, fsample, float(info["flop"]), dt,
                      "OpenBLAS (scipy bundle)" if blas else "oracle triple loop"),
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus and "RANK" in os.environ:
        args.gpus = world
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit("for --gpus > 1 launch through torch.distributed.run (one process per GPU)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # completion signals by polling instead of interrupts (must be set before the runtime starts): the scheduler and launcher
    # threads wait on hundreds of short events per factorisation; measured 44.2 ms (all 40 steps within 43.8-44.8) against
    # 44.6 ms (44.0-48.0) on one box.  A user of the library sets it the same way.
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
    os.environ["LOCAL_RANK"] = str(local_rank)

    import torch  # device selection + the synchronise the contract asks for; not on the compute path

    import pangulu_amd as pa
    from pangulu_amd import _lib
    from pangulu_amd import matrices as M

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the numeric factorisation has no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    lib = _lib.load("r64")
    nthreads = args.host_threads or max(1, (os.cpu_count() or 1) // max(1, world))
    os.environ["PANGULU_AMD_HOST_THREADS"] = str(nthreads)

    if world > 1:
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        base_port = int(os.environ.get("MASTER_PORT", "29500")) + 23
        if base_port + world >= 32768:  # keep the solver's listeners (base_port + rank) out of the ephemeral port range
            base_port = 20000 + (base_port * 7) % 8000
        # auto: rccl first (north star: MPI point-to-point -> RCCL send/recv over xGMI); it self-tests on every pair and all
        # ranks agree on the outcome; if it is not there the same ranks try peer copies, then host staging
        order = {"auto": ["rccl", "ipc", "host"], "rccl": ["rccl"], "ipc": ["ipc"], "host": ["host"]}[args.transport]
        codes = {"host": _lib.TRANSPORT_HOST, "rccl": _lib.TRANSPORT_RCCL, "ipc": _lib.TRANSPORT_IPC}
        t_comm = time.time()
        tried = []
        for k, name in enumerate(order):
            rc = lib.pangulu_amd_comm_init(rank, world, addr.encode(), base_port + 64 * k, codes[name], None)
            assert rc == 0
            got = {0: "host", 1: "rccl", 2: "ipc"}[lib.pangulu_amd_comm_transport()]
            tried.append("%s->%s" % (name, got))
            if got == name or k + 1 == len(order):
                break
            lib.pangulu_amd_comm_finalize()  # (all ranks saw the same fall-back: they all move on to the next one)
        comm_init_s = time.time() - t_comm

    if world == 1:
        tried, comm_init_s = [], 0.0
    if rank == 0:
        mat, workload = make_matrix(args, M)
        n, cp, ri, va, coords = mat
    else:
        n, cp, ri, va, coords, workload = 0, None, None, None, None, ""
    t0 = time.time()
    h = pa.pangulu_init(n, len(va) if va is not None else 0, cp, ri, va, nb=args.nb, ordering=args.ordering,
                        coords=coords if args.ordering == "nd" else None, nthread=nthreads)
    t_init = time.time() - t0
    # what is really in use: ipc / rccl fall back to host staging on all ranks together when their self-test fails
    effective_transport = {0: "host", 1: "rccl", 2: "ipc"}[lib.pangulu_amd_comm_transport()] if world > 1 else "none"
    assert lib.pangulu_amd_snapshot(h.ref) == 0
    info0 = h.info()
    flop = float(info0["flop"])

    def one_step():
        lib.pangulu_amd_comm_barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        pa.pangulu_gstrf(h)            # ends with a stream synchronise + barrier inside the library
        torch.cuda.synchronize()
        lib.pangulu_amd_comm_barrier()
        return time.perf_counter() - t

    # structural flop counting of MFMA-path updates costs an extra pass per task: off in the timed steps (F comes from
    # the symbolic pattern), on in the profile pass below
    lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 0)
    for _ in range(args.warmup):
        one_step()
        lib.pangulu_amd_reset_numeric(h.ref)
    pa.hip_stats(lib, reset=True)
    times = []
    for s in range(args.steps):
        times.append(one_step())
        if s + 1 < args.steps or not args.no_profile_pass:
            lib.pangulu_amd_reset_numeric(h.ref)
    # max over ranks of the summed step time
    tsum = np.array([sum(times)], dtype=np.float64)
    if world > 1:
        lib.pangulu_amd_comm_allreduce_max_f64(tsum.ctypes.data_as(ctypes.c_void_p), 1)
    ms_per_step = float(tsum[0]) / max(1, args.steps) * 1e3
    info = h.info()
    used = ctypes.c_size_t(0)
    lib.pangulu_platform_0201001_get_device_memory_usage(ctypes.byref(used))  # records + receive bins + mirror pool + snapshot
    stats_timed = pa.hip_stats(lib, reset=True)

    default_workload = (not args.mtx and args.workload == "shell" and not args.size and args.nb == 256 and args.ordering == "nd")
    # one extra, un-timed factorisation with per-launch hipEvents to attribute time to kernels
    roofline = None
    kernels = {}
    if not args.no_profile_pass:
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_PROFILE, 1)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 1)
        one_step()
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_PROFILE, 0)
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
        task_classes = ("getrf", "tstrf", "gessm", "ssssm_sparse", "ssssm_dense_mfma")
        if kernels:
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
            # HBM traffic per launch of that kernel: PMC counters cannot be read from inside this process, so the value comes
            # from a committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass (tools/profile_recipe.sh; FETCH_SIZE doubled as
            # MI355X_MICROARCH.md prescribes for gfx950) -- but ONLY if that pass profiled this build's kernels: the file
            # records a hash of the kernel sources, anything else leaves `traffic` null
            rocprof_name = {"ssssm_dense_mfma": "ssssm_dense_f64_kernel", "getrf": "getrf_tiled_f64_kernel",
                            "tstrf": "trsm_dense_direct_f64_kernel<16>", "gessm": "trsm_dense_direct_f64_kernel<16>",
                            "ssssm_sparse": "ssssm_sparse_kernel<false>"}.get(dom)
            tfile = os.path.join(ROOT, "profiles", "hbm_traffic_default_workload.json")
            if world == 1 and default_workload and rocprof_name and os.path.exists(tfile):
                tj = json.load(open(tfile))
                if tj.get("kernel_source_hash") == kernel_source_hash() and tj.get(rocprof_name):
                    roofline["traffic"] = tj[rocprof_name]["hbm_bytes_per_launch"]
                    roofline["traffic_unit"] = "bytes per launch (rocprofv3 PMC pass of this build, %s)" % tj.get("profile", "profiles/")
                else:
                    roofline["traffic_note"] = "no PMC pass of this build committed (kernel sources changed since %s)" % tj.get("profile", "the last one")
            roofline["avg_launch_us"] = kernels[dom]["avg_launch_us"]
            roofline["share_of_kernel_time"] = kernels[dom]["ms"] / sum(k["ms"] for k in kernels.values())
            # whole-factorisation bound of SURVEY.md §8d: T* = sum over tasks of max(bytes_t / 8 TB/s, flop_t / 78.6 TF), from the
            # task list's structure alone (pg_model.cpp); single-rank handles
            if world == 1:
                lib.pangulu_amd_model_roofline(h.ref, HBM_PEAK_GBS, FP64_PEAK_TFLOPS)
                mi = h.info()
                t_star = mi["model_tmin_hbm_bound"] + mi["model_tmin_fp_bound"]
                roofline["model_T_star_ms"] = 1e3 * t_star
                roofline["model_T_star_over_t_gstrf"] = 1e3 * t_star / ms_per_step
                roofline["model_split_ms"] = {"hbm_bound_tasks": 1e3 * mi["model_tmin_hbm_bound"], "mfma_bound_tasks": 1e3 * mi["model_tmin_fp_bound"]}
                roofline["model_alg_GB"] = mi["model_bytes_total"] / 1e9

    # end-to-end check of the last factorisation: ||Ax-b||/||b|| with b = A*1 (examples/example.c:252-264,304-364)
    residual = None
    if rank == 0:
        b = M.read_rhs(args.rhs, n) if args.rhs else M.rhs_of_ones(n, cp, ri, va)
    else:
        b = None
    x = pa.pangulu_gstrs(h, b)
    if rank == 0:
        residual = M.relative_residual(n, cp, ri, va, x, b)
    pa.pangulu_finalize(h)
    if world > 1:
        lib.pangulu_amd_comm_finalize()

    # the CPU baseline runs LAST: measured before the GPU steps it left them 15 % slower (72 instead of 61 ms per step
    # after ten seconds of host-only work with the device idle, whatever the warm-up count; cause not isolated)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, pa, M, mat, workload)

    if rank == 0:
        value = flop / (ms_per_step / 1e3) / 1e9
        line = {
            "metric": "numeric factorisation GFLOP/s (pangulu_gstrf, R64)",
            "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "step_ms": [round(1e3 * t, 2) for t in times], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not args.mtx else "file",
            "config": {
                "workload": workload, "n": int(info["n"]), "nnz": int(info["nnz"]), "nb": int(info["nb"]),
                "ordering": "built-in nested dissection (geometric)" if args.ordering == "nd" else "identity",
                "symbolic_nnz": int(info["symbolic_nnz"]), "flop": int(info["flop"]),
                "parallelism": ("2D block-cyclic %dx%d" % grid(world)) if world == 1 else
                               ("subtrees of the block elimination tree mapped to single ranks (proportional mapping); separators above them: %s"
                                % {"path": "on the rank of their heaviest child", "rank0": "on rank 0",
                                   "cyclic": "2D block-cyclic %dx%d" % grid(world)}[os.environ.get("PANGULU_AMD_SEPARATOR_MAP", "path")]),
                # what is really in use: ipc / rccl fall back to host staging on all ranks when their self-test fails
                "transport": effective_transport,
                "transport_tried": tried, "comm_nranks": world if world > 1 else 0, "comm_init_s": round(comm_init_s, 2),
                "blocks": int(info["nblocks_nondiag"]),
                "tasks": {"getrf": int(info["ntask_getrf"]), "tstrf": int(info["ntask_tstrf"]), "gessm": int(info["ntask_gessm"]),
                          "ssssm": int(info["ntask_ssssm"])},
            },
            "residual": residual,
            "init_s": round(t_init, 2),
            "hbm_used_GB": round(used.value / 1e9, 2), "owned_records_GB": round(info["owned_bytes"] / 1e9, 2),
            "host_sched_s_last_step": round(info["time_numeric_host_sched"], 4),
            "batches_per_step": int(info["batches"]),
            "roofline": roofline,
            "kernels": kernels,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
