"""ctypes binding of libpangulu_amd_<type>.so (host scheduler + HIP back-end, one shared object per value type).

The library is the product: there is no Python or CPU fallback.  If it has not been built, loading raises with
the build command; if no GPU is visible, ``pangulu_init`` aborts inside the library.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
LIB_DIR = os.environ.get("PANGULU_AMD_LIB_DIR") or os.path.join(_HERE, "lib")  # (the override: A/B runs of two builds on one box)

# value type tag -> (numpy dtype, sizeof, is_complex)
VALUE_TYPES = {
    "r64": (np.float64, 8, 0),
    "r32": (np.float32, 4, 0),
    "cr64": (np.complex128, 16, 1),
    "cr32": (np.complex64, 8, 1),
}


class InitOptions(ctypes.Structure):
    """pangulu_init_options (include/pangulu.h; reference include/pangulu_interface_common.h:3-12)."""

    _fields_ = [
        ("nthread", ctypes.c_int),
        ("nb", ctypes.c_int),
        ("gpu_kernel_warp_per_block", ctypes.c_int),
        ("gpu_data_move_warp_per_block", ctypes.c_int),
        ("sizeof_value", ctypes.c_int),
        ("is_complex_matrix", ctypes.c_int),
        ("mpi_recv_buffer_level", ctypes.c_float),
    ]


class GstrfOptions(ctypes.Structure):
    _fields_ = [("reserved_", ctypes.c_char)]


class GstrsOptions(ctypes.Structure):
    _fields_ = [("reserved_", ctypes.c_char)]


class Info(ctypes.Structure):
    """pangulu_amd_info_t (include/pangulu_amd_ext.h)."""

    _fields_ = [
        ("n", ctypes.c_ulonglong),
        ("nnz", ctypes.c_ulonglong),
        ("nb", ctypes.c_ulonglong),
        ("block_length", ctypes.c_ulonglong),
        ("n_padded", ctypes.c_ulonglong),
        ("symbolic_nnz", ctypes.c_ulonglong),
        ("flop", ctypes.c_longlong),
        ("nblocks_nondiag", ctypes.c_ulonglong),
        ("nblocks_owned", ctypes.c_ulonglong),
        ("ntask_getrf", ctypes.c_ulonglong),
        ("ntask_tstrf", ctypes.c_ulonglong),
        ("ntask_gessm", ctypes.c_ulonglong),
        ("ntask_ssssm", ctypes.c_ulonglong),
        ("owned_bytes", ctypes.c_ulonglong),
        ("recv_blocks", ctypes.c_ulonglong),
        ("sent_bytes", ctypes.c_ulonglong),
        ("recv_bytes", ctypes.c_ulonglong),
        ("time_reorder", ctypes.c_double),
        ("time_symbolic", ctypes.c_double),
        ("time_preprocess", ctypes.c_double),
        ("time_numeric", ctypes.c_double),
        ("time_solve", ctypes.c_double),
        ("time_numeric_host_sched", ctypes.c_double),
        ("model_bytes_total", ctypes.c_double),
        ("model_flop_total", ctypes.c_double),
        ("model_tmin_hbm_bound", ctypes.c_double),
        ("model_tmin_fp_bound", ctypes.c_double),
        ("batches", ctypes.c_ulonglong),
        ("sampled_flop", ctypes.c_double),
        ("sampled_tasks", ctypes.c_ulonglong),
        ("time_numeric_platform", ctypes.c_double),
        ("replayed", ctypes.c_ulonglong),
        ("time_schedule_record", ctypes.c_double),
        ("model_ranks_tstar_max", ctypes.c_double),
        ("model_ranks_tstar_sum", ctypes.c_double),
        ("model_ranks_tstar_hbm", ctypes.c_double),
        ("model_ranks_tstar_fp", ctypes.c_double),
        ("model_ranks_bytes_total", ctypes.c_double),
        ("model_rank_flop_share", ctypes.c_double),
        ("model_rank_time_share", ctypes.c_double),
        ("model_comm_seconds_max", ctypes.c_double),
        ("model_sent_bytes_total", ctypes.c_double),
        ("model_critical_path", ctypes.c_double),
        ("model_critical_path_tasks", ctypes.c_ulonglong),
        ("snapshot_device_bytes", ctypes.c_ulonglong),
        ("model_critical_path_latency", ctypes.c_double),
        ("model_rank_hbm_bytes_max", ctypes.c_double),
        ("model_rank_hbm_records", ctypes.c_double),
        ("model_rank_hbm_received", ctypes.c_double),
        ("model_rank_hbm_mirrors", ctypes.c_double),
        ("inserted_diagonals", ctypes.c_ulonglong),
        ("deferred_queues", ctypes.c_ulonglong),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class HipStats(ctypes.Structure):
    """pangulu_hip_stats_t (include/pangulu_platform.h); class index 1 GETRF, 2 TSTRF, 3 GESSM, 4 SSSSM sparse, 5 SSSSM dense,
    6 densify, 7 sparsify, 8 LU images of remote diagonal blocks."""

    _fields_ = [
        ("launches", ctypes.c_ulonglong * 9),
        ("tasks", ctypes.c_ulonglong * 9),
        ("alg_bytes", ctypes.c_double * 9),
        ("flops", ctypes.c_double * 9),
        ("elapsed_ms", ctypes.c_double * 9),
        ("mfma_flops_executed", ctypes.c_double),
        ("trsm_dense_tasks", ctypes.c_ulonglong),
        ("ssssm_front_workgroups", ctypes.c_ulonglong),
        ("ssssm_general_workgroups", ctypes.c_ulonglong),
        ("chase_launches", ctypes.c_ulonglong),
        ("chase_solves", ctypes.c_ulonglong),
        ("ssssm_kernel_ms", ctypes.c_double * 2),
        ("ssssm_front_flops_executed", ctypes.c_double),
    ]


KERNEL_CLASSES = {1: "getrf", 2: "tstrf", 3: "gessm", 4: "ssssm_sparse", 5: "ssssm_dense_mfma",
                  6: "densify", 7: "sparsify", 8: "remote_lu_image"}  # 6..8: mirror maintenance, no task of the reference's model

_cache = {}


def library_path(vtype="r64"):
    return os.path.join(LIB_DIR, "libpangulu_amd_%s.so" % vtype)


def test_library_path(vtype="r64"):
    """The checker's build of the host (oracle/pangulu_amd_test_hooks.h): same sources + a platform loader.
    Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it."""
    return os.path.join(REPO_ROOT, "oracle", "_build", "libpangulu_amd_test_%s.so" % vtype)


def load(vtype="r64", test_hooks=False):
    """Load (once) the shared object for a value type and declare the signatures used from Python.

    test_hooks=True loads the checker's variant instead (it can run the scheduler on the oracle's CPU operators)."""
    vtype = vtype.lower()
    key = (vtype, bool(test_hooks))
    if key in _cache:
        return _cache[key]
    path = test_library_path(vtype) if test_hooks else library_path(vtype)
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pangulu_amd/csrc TYPE=%s`); there is no fallback implementation" % (path, vtype.upper())
        )
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    vp, vpp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)
    lib.pangulu_init.argtypes = [ctypes.c_uint32, ctypes.c_uint64, vp, vp, vp, ctypes.POINTER(InitOptions), vpp]
    lib.pangulu_init.restype = None
    lib.pangulu_gstrf.argtypes = [ctypes.POINTER(GstrfOptions), vpp]
    lib.pangulu_gstrf.restype = None
    lib.pangulu_gstrs.argtypes = [vp, ctypes.POINTER(GstrsOptions), vpp]
    lib.pangulu_gstrs.restype = None
    lib.pangulu_gssv.argtypes = [vp, ctypes.POINTER(GstrfOptions), ctypes.POINTER(GstrsOptions), vpp]
    lib.pangulu_gssv.restype = None
    lib.pangulu_finalize.argtypes = [vpp]
    lib.pangulu_finalize.restype = None

    lib.pangulu_amd_comm_init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, vp]
    lib.pangulu_amd_comm_init.restype = ctypes.c_int
    lib.pangulu_amd_comm_barrier.restype = None
    lib.pangulu_amd_comm_allreduce_max_f64.argtypes = [vp, ctypes.c_int]
    lib.pangulu_amd_comm_allreduce_max_f64.restype = None
    lib.pangulu_amd_comm_finalize.restype = None
    lib.pangulu_amd_comm_transport.restype = ctypes.c_int
    lib.pangulu_amd_comm_rccl_ranks.restype = ctypes.c_int
    lib.pangulu_amd_comm_rank.restype = ctypes.c_int
    lib.pangulu_amd_comm_size.restype = ctypes.c_int
    if test_hooks:
        lib.pangulu_amd_use_platform_library.argtypes = [ctypes.c_char_p, ctypes.c_uint]
        lib.pangulu_amd_use_platform_library.restype = ctypes.c_int
    lib.pangulu_amd_active_platform.restype = ctypes.c_uint
    lib.pangulu_amd_use_builtin_platform.restype = None
    lib.pangulu_amd_set_ordering.argtypes = [ctypes.c_int]
    lib.pangulu_amd_set_ordering.restype = None
    lib.pangulu_amd_set_user_perm.argtypes = [vp, ctypes.c_uint32]
    lib.pangulu_amd_set_user_perm.restype = None
    lib.pangulu_amd_set_coordinates.argtypes = [vp, ctypes.c_uint32, ctypes.c_int]
    lib.pangulu_amd_set_coordinates.restype = None
    lib.pangulu_amd_set_eager_host_mirror.argtypes = [ctypes.c_int]
    lib.pangulu_amd_set_eager_host_mirror.restype = None
    lib.pangulu_amd_set_scaling.argtypes = [ctypes.c_int]
    lib.pangulu_amd_set_scaling.restype = None
    lib.pangulu_amd_reset_options.restype = None
    lib.pangulu_amd_get_info.argtypes = [vpp, ctypes.POINTER(Info)]
    lib.pangulu_amd_get_info.restype = None
    lib.pangulu_amd_model_roofline.argtypes = [vpp, ctypes.c_double, ctypes.c_double]
    lib.pangulu_amd_model_roofline.restype = None
    lib.pangulu_amd_block_owner.argtypes = [vpp, ctypes.c_uint32, ctypes.c_uint32]
    lib.pangulu_amd_block_owner.restype = ctypes.c_int
    lib.pangulu_amd_rank_model.argtypes = [vpp, vp, vp, vp]
    lib.pangulu_amd_rank_model.restype = ctypes.c_int
    lib.pangulu_amd_owned_block_count.argtypes = [vpp]
    lib.pangulu_amd_owned_block_count.restype = ctypes.c_longlong
    lib.pangulu_amd_owned_block.argtypes = [
        vpp, ctypes.c_longlong, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong), vpp, vpp, vpp,
    ]
    lib.pangulu_amd_owned_block.restype = ctypes.c_int
    lib.pangulu_amd_update_values.argtypes = [vpp, vp]
    lib.pangulu_amd_update_values.restype = ctypes.c_int
    lib.pangulu_amd_snapshot.argtypes = [vpp]
    lib.pangulu_amd_snapshot.restype = ctypes.c_int
    lib.pangulu_amd_reset_numeric.argtypes = [vpp]
    lib.pangulu_amd_reset_numeric.restype = ctypes.c_int
    lib.pangulu_amd_set_replay.argtypes = [ctypes.c_int]
    lib.pangulu_amd_set_replay.restype = ctypes.c_int
    lib.pangulu_amd_get_perm.argtypes = [vpp]
    lib.pangulu_amd_get_perm.restype = ctypes.POINTER(ctypes.c_uint32)
    lib.pangulu_amd_apply_lu.argtypes = [vpp, vp, vp]
    lib.pangulu_amd_apply_lu.restype = ctypes.c_int
    lib.pangulu_amd_factor_check.argtypes = [vpp, ctypes.POINTER(ctypes.c_double)]
    lib.pangulu_amd_factor_check.restype = ctypes.c_int
    lib.pangulu_amd_factor_check_vectors.argtypes = [vpp, ctypes.c_int, ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_double)]
    lib.pangulu_amd_factor_check_vectors.restype = ctypes.c_int
    lib.pangulu_amd_model_for_ranks.argtypes = [vpp, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    lib.pangulu_amd_model_for_ranks.restype = ctypes.c_int

    lib.pangulu_platform_0201001_set_option.argtypes = [ctypes.c_int, ctypes.c_longlong]
    lib.pangulu_platform_0201001_set_option.restype = ctypes.c_int
    lib.pangulu_platform_0201001_get_stats.argtypes = [ctypes.POINTER(HipStats), ctypes.c_int]
    lib.pangulu_platform_0201001_get_stats.restype = None
    lib.pangulu_platform_0201001_get_stream.restype = ctypes.c_void_p
    _cache[key] = lib
    return lib


PLATFORM_SYMBOLS = [
    "malloc", "malloc_pinned", "synchronize", "memset", "create_stream", "memcpy", "memcpy_async", "free",
    "get_device_num", "set_default_device", "get_device_name", "get_device_memory_usage",
    "getrf", "tstrf", "gessm", "ssssm", "ssssm_batched", "hybrid_batched", "spmv", "vecadd", "sptrsv",
]

HIP_OPT_HOST_MIRROR = 1
HIP_OPT_DENSE_THRESHOLD_PERMILLE = 2
HIP_OPT_PROFILE = 3
HIP_OPT_ASSUME_INDEPENDENT = 4
HIP_OPT_GETRF_STRICT_ORDER = 5
HIP_OPT_COUNT_FLOPS = 6
HIP_OPT_RESET_BLOCK_STATE = 7
HIP_OPT_SSSSM_GROUP_CHUNK = 8
HIP_OPT_TRSM_DENSE_PERMILLE = 9
HIP_OPT_TWO_STREAMS = 10
HIP_OPT_SMALL_LAUNCH_TASKS = 11
HIP_OPT_XCD_SWIZZLE = 12
HIP_OPT_RECORDS_STREAM = 13
HIP_OPT_BACKGROUND_UPDATES = 14
HIP_OPT_FRONT_STAGES = 15
HIP_OPT_TILES_STAGES = 16

ORDER_IDENTITY, ORDER_ND, ORDER_USER = 0, 1, 2
TRANSPORT_HOST, TRANSPORT_RCCL, TRANSPORT_IPC = 0, 1, 2
PLATFORM_CPU_NAIVE = 0x0100000
PLATFORM_GPU_HIP = 0x0201001
