// pg_api.cpp -- public entry points (include/pangulu.h, include/pangulu_amd_ext.h) and the platform table.
//
// pangulu_init / gstrf / gstrs / gssv / finalize keep the reference's contract (src/pangulu.c:11-345):
// void returns, opaque handle, message + exit(1) on misuse.
#include <cmath>
#include <cstdarg>
#include <dlfcn.h>
#include <omp.h>
#include <sys/time.h>

#include <sys/mman.h>
#include "pg_host.h"

namespace pg
{
#ifdef PANGULU_COMPLEX
static inline val_t make_value(double re) { return val_t{(decltype(val_t{}.re))re, 0}; }
#else
static inline val_t make_value(double re) { return (val_t)re; }
#endif


void fatal(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[PanguLU-AMD ERROR] ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    fflush(stderr);
    exit(1);
}

double wall_seconds()
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

// ---------------------------------------------------------------------------------------------------------
// platform table
// ---------------------------------------------------------------------------------------------------------
static Platform g_platform;
static bool g_platform_ready = false;
static bool g_platform_builtin = true;

void bind_builtin_hip(Platform &p)
{
    p = Platform();
    p.id = PANGULU_PLATFORM_GPU_HIP;
    p.host_memory = false;
    p.malloc_ = pangulu_platform_0201001_malloc;
    p.malloc_pinned = pangulu_platform_0201001_malloc_pinned;
    p.synchronize = pangulu_platform_0201001_synchronize;
    p.memset_ = pangulu_platform_0201001_memset;
    p.create_stream = pangulu_platform_0201001_create_stream;
    p.memcpy_ = pangulu_platform_0201001_memcpy;
    p.memcpy_async = pangulu_platform_0201001_memcpy_async;
    p.free_ = pangulu_platform_0201001_free;
    p.get_device_num = pangulu_platform_0201001_get_device_num;
    p.set_default_device = pangulu_platform_0201001_set_default_device;
    p.get_device_name = pangulu_platform_0201001_get_device_name;
    p.get_device_memory_usage = pangulu_platform_0201001_get_device_memory_usage;
    p.getrf = pangulu_platform_0201001_getrf;
    p.tstrf = pangulu_platform_0201001_tstrf;
    p.gessm = pangulu_platform_0201001_gessm;
    p.ssssm = pangulu_platform_0201001_ssssm;
    p.ssssm_batched = pangulu_platform_0201001_ssssm_batched;
    p.hybrid_batched = pangulu_platform_0201001_hybrid_batched;
    p.spmv = pangulu_platform_0201001_spmv;
    p.vecadd = pangulu_platform_0201001_vecadd;
    p.sptrsv = pangulu_platform_0201001_sptrsv;
    p.set_option = pangulu_platform_0201001_set_option;
    p.prepare_diag = pangulu_platform_0201001_prepare_diag;
    p.prepare_blocks = pangulu_platform_0201001_prepare_blocks;
    p.marker_record = pangulu_platform_0201001_marker_record;
    p.marker_done = pangulu_platform_0201001_marker_done;
    p.marker_wait = pangulu_platform_0201001_marker_wait;
    p.block_trsv = pangulu_platform_0201001_block_trsv;
    p.block_spmv_add = pangulu_platform_0201001_block_spmv_add;
    p.schedule = pangulu_platform_0201001_schedule;
    p.schedule_range = pangulu_platform_0201001_schedule_range;
    p.marker_record_replay = pangulu_platform_0201001_marker_record_replay;
    p.bind_near_device = pangulu_platform_0201001_bind_near_device;
}

Platform &active_platform()
{
    if (!g_platform_ready)
    {
        bind_builtin_hip(g_platform);
        g_platform_ready = true;
        g_platform_builtin = true;
    }
    return g_platform;
}

bool platform_is_builtin_hip() { return g_platform_builtin; }

static Options g_options;
Options &pending_options() { return g_options; }

} // namespace pg

using namespace pg;

namespace
{
// the calling thread stays next to the device for the duration of an API call (pangulu_platform_0201001_bind_near_device)
struct NearDevice
{
    Platform &plat;
    bool bound = false;
    explicit NearDevice(Platform &p) : plat(p)
    {
        if (!plat.host_memory && plat.bind_near_device)
        {
            // Threads created while the mask is narrowed keep it for good.  The scheduler's launcher thread is meant to (it ends
            // with the call); OpenMP pool workers are not -- they outlive the call and serve the host application's own parallel
            // regions.  Make sure the pool exists, with the caller's mask, before the mask changes.
#pragma omp parallel
            {
            }
            bound = plat.bind_near_device(1) == 0;
        }
    }
    ~NearDevice()
    {
        if (bound)
            plat.bind_near_device(0);
    }
};
} // namespace

extern "C"
{

#ifdef PANGULU_AMD_TEST_HOOKS
    // TEST BUILD ONLY (oracle/_build/libpangulu_amd_test_*.so, oracle/pangulu_amd_test_hooks.h): the shipped library
    // binds platform 0201001 statically and has no way to route the operators anywhere else.
    int pangulu_amd_use_platform_library(const char *so_path, unsigned int platform_id)
    {
        void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
        if (!h)
        {
            fprintf(stderr, "[PanguLU-AMD] cannot load platform library %s: %s\n", so_path, dlerror());
            return 1;
        }
        Platform p;
        p.id = platform_id;
        p.dl_handle = h;
        p.host_memory = (platform_id >> 20) == 0x01; // 01_SHAREDMEM family
        char sym[128];
        int missing = 0;
        auto get = [&](const char *name) -> void *
        {
            snprintf(sym, sizeof(sym), "pangulu_platform_%07x_%s", platform_id, name);
            void *f = dlsym(h, sym);
            if (!f)
            {
                fprintf(stderr, "[PanguLU-AMD] %s lacks %s\n", so_path, sym);
                missing++;
            }
            return f;
        };
        p.malloc_ = (void (*)(void **, size_t))get("malloc");
        p.malloc_pinned = (void (*)(void **, size_t))get("malloc_pinned");
        p.synchronize = (void (*)())get("synchronize");
        p.memset_ = (void (*)(void *, int, size_t))get("memset");
        p.create_stream = (void (*)(void **))get("create_stream");
        p.memcpy_ = (void (*)(void *, const void *, size_t, unsigned))get("memcpy");
        p.memcpy_async = (void (*)(void *, const void *, size_t, unsigned, void *))get("memcpy_async");
        p.free_ = (void (*)(void *))get("free");
        p.get_device_num = (void (*)(int *))get("get_device_num");
        p.set_default_device = (void (*)(int))get("set_default_device");
        p.get_device_name = (void (*)(char *, int))get("get_device_name");
        p.get_device_memory_usage = (void (*)(size_t *))get("get_device_memory_usage");
        p.getrf = (void (*)(pangulu_inblock_idx, slot_t *, int))get("getrf");
        p.tstrf = (void (*)(pangulu_inblock_idx, slot_t *, slot_t *, int))get("tstrf");
        p.gessm = (void (*)(pangulu_inblock_idx, slot_t *, slot_t *, int))get("gessm");
        p.ssssm = (void (*)(pangulu_inblock_idx, slot_t *, slot_t *, slot_t *, int))get("ssssm");
        p.ssssm_batched = (void (*)(pangulu_inblock_idx, pangulu_uint64_t, task_t *))get("ssssm_batched");
        p.hybrid_batched = (void (*)(pangulu_inblock_idx, pangulu_uint64_t, task_t *))get("hybrid_batched");
        p.spmv = (void (*)(pangulu_inblock_idx, slot_t *, val_t *, val_t *))get("spmv");
        p.vecadd = (void (*)(pangulu_int64_t, val_t *, val_t *))get("vecadd");
        p.sptrsv = (void (*)(pangulu_inblock_idx, slot_t *, val_t *, pangulu_int64_t))get("sptrsv");
        if (missing)
            return 2;
        int (*sz)() = (int (*)())dlsym(h, "pangulu_oracle_sizeof_value");
        if (sz && sz() != (int)sizeof(val_t))
        {
            fprintf(stderr, "[PanguLU-AMD] %s was built for another value type\n", so_path);
            return 3;
        }
        g_platform = p;
        g_platform_ready = true;
        g_platform_builtin = false;
        return 0;
    }

    // Mapping and models for `size` ranks evaluated in ONE process: the next pangulu_init must run with
    // PANGULU_AMD_ANALYSIS_ONLY=1 (no records, no exchange); size <= 1 restores the single-rank world.
    void pangulu_amd_test_set_analysis_ranks(int size) { set_fake_world(size); }

    // bench.py's cpu_baseline leg: run every stride-th task of each kernel class only (pg_numeric.cpp)
    void pangulu_amd_test_set_task_sampling(int stride) { g_task_sample_stride = stride < 1 ? 1 : stride; }

    // The host's own priority heap driven by a script (script[i] >= 0: push tasks[script[i]], -1: pop); the popped tasks
    // go to `out`.  tests/test_reference_pin.py runs the same script through the reference's src/pangulu_task.c.
    long long pangulu_amd_test_heap_script(long long nscript, const long long *script, const pangulu_task_t *tasks, pangulu_task_t *out)
    {
        TaskHeap heap;
        long long npop = 0;
        for (long long i = 0; i < nscript; i++)
        {
            if (script[i] >= 0)
                heap.push(tasks[script[i]]);
            else if (!heap.pop(out[npop++]))
                return -2;
        }
        return npop;
    }

    // maximum-product matching + scaling on a bare CSC matrix (pg_scaling.cpp); returns 0 on success
    int pangulu_amd_test_matching(sparse_index_t n, const sparse_pointer_t *colptr, const sparse_index_t *rowidx, const sparse_value_t *value,
                                  sparse_index_t *col_of_row, double *dr, double *dc)
    {
        CscMatrix A;
        A.n = n;
        A.colptr.assign(colptr, colptr + n + 1);
        A.rowidx.assign(rowidx, rowidx + colptr[n]);
        A.value.assign((const val_t *)value, (const val_t *)value + colptr[n]);
        std::vector<u32> q;
        std::vector<double> r, c;
        if (!max_product_matching(A, q, r, c))
            return 1;
        std::copy(q.begin(), q.end(), col_of_row);
        std::copy(r.begin(), r.end(), dr);
        std::copy(c.begin(), c.end(), dc);
        return 0;
    }

    // The host's symbolic phase on a CSC pattern (already ordered): lower fill pattern incl. diagonal per column, the
    // reference's symbolic_nnz and the structural flop count.  Arrays are malloc'ed; free with free().
    int pangulu_amd_test_symbolic(sparse_index_t n, const sparse_pointer_t *colptr, const sparse_index_t *rowidx,
                                  sparse_pointer_t **out_ptr, sparse_index_t **out_idx, unsigned long long *out_symbolic_nnz, long long *out_flop)
    {
        CscMatrix A;
        A.n = n;
        A.colptr.assign(colptr, colptr + n + 1);
        A.rowidx.assign(rowidx, rowidx + colptr[n]);
        A.value.assign((size_t)colptr[n], val_t());
        Symbolic sym;
        symbolic_factorize(A, sym);
        *out_ptr = (sparse_pointer_t *)malloc(sizeof(sparse_pointer_t) * ((size_t)n + 1));
        *out_idx = (sparse_index_t *)malloc(sizeof(sparse_index_t) * std::max<size_t>(1, sym.idx.size()));
        std::copy(sym.ptr.begin(), sym.ptr.end(), *out_ptr);
        std::copy(sym.idx.begin(), sym.idx.end(), *out_idx);
        *out_symbolic_nnz = sym.symbolic_nnz;
        *out_flop = sym.flop;
        return 0;
    }
#endif

    void pangulu_amd_use_builtin_platform(void)
    {
        bind_builtin_hip(g_platform);
        g_platform_ready = true;
        g_platform_builtin = true;
    }

    unsigned int pangulu_amd_active_platform(void) { return active_platform().id; }

    int pangulu_amd_comm_init(int rank, int size, const char *addr, int base_port, int transport, const void *nccl_unique_id)
    {
        if (size <= 1)
        {
            set_world(nullptr);
            return 0;
        }
        if (size > 64)
        {
            // consumer sets and forwarding masks are one bit per rank in a 64-bit word (Solver::consumers, pg_model.cpp); the
            // scope is one node of 8 GPUs (the reference's grids: 1x1 ... 2x4, src/pangulu.c:83-90)
            fprintf(stderr, "[PanguLU-AMD ERROR] %d ranks requested: at most 64 are supported\n", size);
            return 1;
        }
        // the device transports capture the calling thread's current device: select this rank's GPU first, by the same
        // rule pangulu_init uses (LOCAL_RANK, else the rank, modulo the visible devices)
        Platform &plat = active_platform();
        if (!plat.host_memory && plat.get_device_num && plat.set_default_device)
        {
            int ndev = 0;
            plat.get_device_num(&ndev);
            if (ndev > 0)
            {
                const char *lr = getenv("LOCAL_RANK");
                plat.set_default_device(lr ? atoi(lr) % ndev : rank % ndev);
            }
        }
        Comm *c = make_socket_comm(rank, size, addr ? addr : "127.0.0.1", base_port, transport, nccl_unique_id);
        set_world(c);
        return 0;
    }
    int pangulu_amd_set_replay(int on)
    {
        const int before = g_replay_enabled < 0 ? !(getenv("PANGULU_AMD_REPLAY") && atoi(getenv("PANGULU_AMD_REPLAY")) == 0) : g_replay_enabled;
        g_replay_enabled = on ? 1 : 0;
        return before;
    }
    void pangulu_amd_comm_barrier(void) { world()->barrier(); }
    void pangulu_amd_comm_allreduce_max_f64(double *values, int count) { world()->allreduce_max_f64(values, count); }
    void pangulu_amd_comm_finalize(void) { set_world(nullptr); }
    int pangulu_amd_rccl_unique_id(void *out128) { return rccl_make_unique_id(out128); }
    int pangulu_amd_comm_transport(void) { return world()->transport; }
    int pangulu_amd_comm_rccl_ranks(void) { return world()->rccl_ranks(); }
    int pangulu_amd_comm_rank(void) { return world()->rank; }
    int pangulu_amd_comm_size(void) { return world()->size; }

    void pangulu_amd_set_ordering(int kind) { pending_options().ordering = kind; }
    void pangulu_amd_set_user_perm(const sparse_index_t *perm, sparse_index_t n)
    {
        pending_options().user_perm.assign(perm, perm + n);
        pending_options().ordering = PANGULU_AMD_ORDER_USER;
    }
    void pangulu_amd_set_coordinates(const double *xyz, sparse_index_t n, int dim)
    {
        pending_options().coords.assign(xyz, xyz + (size_t)n * dim);
        pending_options().coord_dim = dim;
    }
    void pangulu_amd_set_eager_host_mirror(int on) { pending_options().eager_host_mirror = on != 0; }
    void pangulu_amd_set_scaling(int on) { pending_options().scaling = on != 0; }
    void pangulu_amd_reset_options(void) { pending_options() = Options(); }

    // ----------------------------------------------------------------------------------------------------
    void pangulu_init(sparse_index_t pangulu_n, sparse_pointer_t pangulu_nnz, sparse_pointer_t *csc_colptr,
                      sparse_index_t *csc_rowidx, sparse_value_t *csc_value, pangulu_init_options *init_options,
                      void **pangulu_handle)
    {
        Comm *comm = world();
        int rank = comm->rank, size = comm->size;
        if (init_options == nullptr)
        {
            if (rank == 0)
                printf("[PanguLU ERROR] Invalid input parameter. Option struct pointer is NULL. Exit.\n");
            exit(1);
        }
#ifdef PANGULU_COMPLEX
        const int lib_complex = 1;
#else
        const int lib_complex = 0;
#endif
        if ((init_options->is_complex_matrix != 0) != (lib_complex != 0))
        {
            if (rank == 0)
                printf("[PanguLU ERROR] Complex/real mismatch between init_options.is_complex_matrix and the library's value type. Exit.\n");
            exit(1);
        }
        if (init_options->sizeof_value != (int)sizeof(calculate_type))
        {
            if (rank == 0)
                printf("[PanguLU ERROR] init_options.sizeof_value (%d) differs from the library's sizeof(value) (%d). Exit.\n",
                       init_options->sizeof_value, (int)sizeof(calculate_type));
            exit(1);
        }
        if (init_options->nb == 0)
        {
            if (rank == 0)
                printf("[PanguLU ERROR] Invalid input parameter. nb is zero. Exit.\n");
            exit(1);
        }
        Platform &plat = active_platform();
        if (!plat.host_memory)
        {
            // the product path: a device is mandatory, there is no CPU fallback
            int ndev = 0;
            plat.get_device_num(&ndev);
            if (ndev <= 0)
                fatal("no HIP device visible: the numeric factorisation runs on the GPU only");
            const char *lr = getenv("LOCAL_RANK");
            int dev = lr ? atoi(lr) % ndev : rank % ndev;
            plat.set_default_device(dev);
        }
        // (no NUMA binding here: the analysis runs on all cores with OpenMP, and pool workers created under a narrowed mask
        // would keep it for good -- the host application's own OpenMP regions included.  Only the latency-critical threads of
        // pangulu_gstrf / pangulu_gstrs, scheduler and launcher, are kept next to the device, for the duration of the call.)

        Solver *S = new Solver();
        memset(&S->info, 0, sizeof(S->info));
        S->rank = rank;
        S->nproc = size;
        int p = (int)std::sqrt((double)size);
        while (size % p)
            p--;
        S->p = p;
        S->q = size / p;
        S->nb = init_options->nb < 0 ? 256u : (u32)init_options->nb;
        if (S->nb > 65535)
            fatal("nb = %u exceeds the 16-bit in-block index range", S->nb);
        S->recv_buffer_level = init_options->mpi_recv_buffer_level;
        int nthread = init_options->nthread <= 0 ? 1 : init_options->nthread;
        const char *ht = getenv("PANGULU_AMD_HOST_THREADS");
        if (ht)
            nthread = atoi(ht);
        omp_set_num_threads(nthread);
        if (init_options->gpu_kernel_warp_per_block > 0)
            pangulu_gpu_kernel_warp_per_block = init_options->gpu_kernel_warp_per_block;
        if (init_options->gpu_data_move_warp_per_block > 0)
            pangulu_gpu_data_move_warp_per_block = init_options->gpu_data_move_warp_per_block;

        // the matrix lives on rank 0 (examples/example.c reads it there); everyone gets a copy
        CscMatrix A;
        u64 dims[2] = {pangulu_n, pangulu_nnz};
        comm->bcast(dims, sizeof(dims), 0);
        A.n = (u32)dims[0];
        if (A.n == 0)
        {
            if (rank == 0)
                printf("[PanguLU ERROR] Invalid input parameter. Matrix order is zero. Exit.\n");
            exit(1);
        }
        A.colptr.resize((size_t)A.n + 1);
        A.rowidx.resize(dims[1]);
        A.value.resize(dims[1]);
        if (rank == 0)
        {
            std::copy(csc_colptr, csc_colptr + A.n + 1, A.colptr.begin());
            std::copy(csc_rowidx, csc_rowidx + dims[1], A.rowidx.begin());
            std::copy(csc_value, csc_value + dims[1], A.value.begin());
        }
        comm->bcast(A.colptr.data(), sizeof(u64) * A.colptr.size(), 0);
        comm->bcast(A.rowidx.data(), sizeof(u32) * A.rowidx.size(), 0);
        comm->bcast(A.value.data(), sizeof(val_t) * A.value.size(), 0);
        S->n_user = A.n;
        S->user_colptr = A.colptr; // (pattern of the matrix as the user passed it: pangulu_amd_update_values maps new values through it)
        S->user_rowidx = A.rowidx;
        S->info.n = A.n;
        S->info.nnz = A.nnz();
        S->info.nb = S->nb;

        Options &opt = pending_options();
        S->eager_host_mirror = opt.eager_host_mirror;
        double t0 = wall_seconds();
        // Optional first step, like the reference's when built with its MC64 port (src/pangulu_reordering.c:1150-1173): put
        // large entries on the diagonal by a column permutation and scale rows and columns so that they are 1 and nothing
        // is larger.  A failure (structurally singular input) leaves the matrix as it is, with a warning, as there.
        {
            const char *se = getenv("PANGULU_AMD_SCALING");
            const bool want = se ? atoi(se) != 0 : opt.scaling;
            u64 applied = 0;
            if (want && rank == 0)
            {
                if (max_product_matching(A, S->match_col, S->scale_row, S->scale_col))
                {
                    CscMatrix A1;
                    apply_matching(A, S->match_col, S->scale_row, S->scale_col, A1);
                    // columns of A1 are not sorted by row any more only if A's were not; keep CSC order per column
                    A = std::move(A1);
                    applied = 1;
                }
                else
                {
                    printf("[PanguLU-AMD WARNING] maximum-product matching failed (structurally singular?): continuing without scaling\n");
                    S->match_col.clear();
                    S->scale_row.clear();
                    S->scale_col.clear();
                }
            }
            comm->bcast(&applied, sizeof(applied), 0);
            if (applied && rank != 0)
            {
                // (only rank 0 applies b and x transformations in gstrs; the others need the scaled matrix only)
            }
            if (applied)
            {
                comm->bcast(A.colptr.data(), sizeof(u64) * A.colptr.size(), 0);
                comm->bcast(A.rowidx.data(), sizeof(u32) * A.rowidx.size(), 0);
                comm->bcast(A.value.data(), sizeof(val_t) * A.value.size(), 0);
            }
        }
        // The reference's zero-diagonal rule (pangulu_add_diagonal_element_csc, src/pangulu_reordering.c:715-796, called on its METIS
        // path, :959): a column without a stored diagonal entry gets one, with the value 1e-8, at its sorted position -- a static
        // perturbation instead of a pivot search (matrix files of KKT systems have structurally empty diagonal blocks; with the matching
        // on, the permuted diagonal is full and nothing is inserted).  Here on the nested-dissection path like there;
        // PANGULU_AMD_ZERO_DIAGONAL=0 leaves such entries structurally present and numerically zero (GETRF's pivot clamp then sees
        // them), any other value is used instead of 1e-8.  Every rank holds the matrix: every rank inserts.
        {
            double zero_element = 1e-8;
            if (const char *e = getenv("PANGULU_AMD_ZERO_DIAGONAL"))
                zero_element = atof(e);
            if (opt.ordering == PANGULU_AMD_ORDER_ND && zero_element != 0.0)
            {
                u64 missing = 0;
                for (u32 j = 0; j < A.n; j++)
                {
                    bool has = false;
                    for (u64 p_ = A.colptr[j]; p_ < A.colptr[j + 1] && !has; p_++)
                        has = A.rowidx[p_] == j;
                    missing += has ? 0 : 1;
                }
                if (missing)
                {
                    CscMatrix B;
                    B.n = A.n;
                    B.colptr.resize((size_t)A.n + 1);
                    B.rowidx.reserve(A.rowidx.size() + missing);
                    B.value.reserve(A.value.size() + missing);
                    B.colptr[0] = 0;
                    for (u32 j = 0; j < A.n; j++)
                    {
                        bool has = false, placed = false;
                        for (u64 p_ = A.colptr[j]; p_ < A.colptr[j + 1] && !has; p_++)
                            has = A.rowidx[p_] == j;
                        for (u64 p_ = A.colptr[j]; p_ < A.colptr[j + 1]; p_++)
                        {
                            if (!has && !placed && A.rowidx[p_] > j)
                            {
                                B.rowidx.push_back(j);
                                B.value.push_back(make_value(zero_element));
                                placed = true;
                            }
                            B.rowidx.push_back(A.rowidx[p_]);
                            B.value.push_back(A.value[p_]);
                        }
                        if (!has && !placed)
                        {
                            B.rowidx.push_back(j);
                            B.value.push_back(make_value(zero_element));
                        }
                        B.colptr[(size_t)j + 1] = B.rowidx.size();
                    }
                    A = std::move(B);
                    if (rank == 0 && getenv("PANGULU_AMD_TRACE"))
                        fprintf(stderr, "[pangulu_amd trace] init: %llu columns without a diagonal entry got one (%g)\n", (unsigned long long)missing, zero_element);
                }
                S->info.inserted_diagonals = missing;
            }
        }
        if (rank == 0)
        {
            if (opt.ordering == PANGULU_AMD_ORDER_USER)
            {
                if (opt.user_perm.size() != A.n)
                    fatal("user permutation has length %zu, matrix order is %u", opt.user_perm.size(), A.n);
                S->perm = opt.user_perm;
            }
            else if (opt.ordering == PANGULU_AMD_ORDER_ND)
            {
                const double *xyz = (opt.coord_dim > 0 && opt.coords.size() == (size_t)A.n * opt.coord_dim) ? opt.coords.data() : nullptr;
                const char *al = getenv("PANGULU_AMD_ND_ALIGN");
                u32 align = (al && atoi(al) == 0) ? 0u : S->nb; // subtrees start on block boundaries (pg_analysis.cpp)
                order_nested_dissection(A, xyz, opt.coord_dim, align, S->perm);
            }
            else
            {
                order_identity(A.n, S->perm);
            }
        }
        // an aligned dissection is longer than the matrix: positions whose "old" index is >= n_user are padding,
        // realised as isolated rows/columns with a unit diagonal (x = b = 1 there, no coupling, no flops)
        u64 np = S->perm.size();
        comm->bcast(&np, sizeof(np), 0);
        S->perm.resize(np);
        comm->bcast(S->perm.data(), sizeof(u32) * np, 0);
        // one-shot: the data that belongs to ONE matrix -- a stale user permutation or coordinate array of the same length
        // would otherwise be applied silently to the next one (ORDER_USER falls back to the default ordering with them).
        // Sticky until changed: the choices -- ordering kind, scaling, eager host mirror (pangulu_amd_reset_options()
        // restores the defaults).
        opt.coords.clear();
        opt.coord_dim = 0;
        opt.user_perm.clear();
        if (opt.ordering == PANGULU_AMD_ORDER_USER)
            opt.ordering = PANGULU_AMD_ORDER_ND;
        S->n = (u32)np;
        S->nbk = (S->n + S->nb - 1) / S->nb;
        S->info.n_padded = S->n;
        S->info.block_length = S->nbk;
        if (S->n > A.n)
        {
            u32 extra = S->n - A.n;
            u64 base = A.nnz();
            A.colptr.resize((size_t)S->n + 1);
            A.rowidx.resize(base + extra);
            A.value.resize(base + extra);
            for (u32 d = 0; d < extra; d++)
            {
                A.rowidx[base + d] = A.n + d;
#ifdef PANGULU_COMPLEX
                A.value[base + d] = val_t{1, 0};
#else
                A.value[base + d] = 1;
#endif
                A.colptr[(size_t)A.n + d + 1] = base + d + 1;
            }
            A.n = S->n;
        }
        S->iperm.resize(S->n);
        for (u32 i = 0; i < S->n; i++)
            S->iperm[S->perm[i]] = i;
        permute_symmetric(A, S->perm, S->Aperm);
        A = CscMatrix();
        S->info.time_reorder = wall_seconds() - t0;

        t0 = wall_seconds();
        symbolic_factorize(S->Aperm, S->sym);
        S->info.symbolic_nnz = S->sym.symbolic_nnz;
        S->info.flop = S->sym.flop;
        S->info.time_symbolic = wall_seconds() - t0;

        t0 = wall_seconds();
        build_block_pattern(S->sym, S->nb, S->pat);
        S->info.nblocks_nondiag = S->pat.colptr[S->nbk];
        {
            const char *ao = getenv("PANGULU_AMD_ANALYSIS_ONLY");
            S->analysis_only = ao && atoi(ao) != 0;
        }
        const bool trace_init = getenv("PANGULU_AMD_TRACE") != nullptr && rank == 0;
        double t1 = wall_seconds();
        if (trace_init)
            fprintf(stderr, "[pangulu_amd trace] init: reorder %.2f s, symbolic %.2f s, block pattern %.2f s (%d threads)\n", S->info.time_reorder,
                    S->info.time_symbolic, t1 - t0, nthread);
        build_structure_model(*S); // weights for the mapping, from the symbolic pattern (every rank the same)
        double t2 = wall_seconds();
        preprocess(*S, S->Aperm);
        double t3 = wall_seconds();
        compute_rank_model(*S);    // per-rank T*, flop shares, link term, critical path under the mapping just made
        if (trace_init)
            fprintf(stderr, "[pangulu_amd trace] init: structure model %.2f s, records + counters + upload %.2f s, rank model + critical path %.2f s\n",
                    t2 - t1, t3 - t2, wall_seconds() - t3);
        // the element-level pattern is only needed to build the records
        S->sym.idx = std::vector<u32>();
        S->sym.ptr = std::vector<u64>();
        comm->barrier();
        S->info.time_preprocess = wall_seconds() - t0;
        record_schedule(*S); // (one rank on the device: the first pangulu_gstrf will already replay)
        *pangulu_handle = (void *)S;
    }

    void pangulu_gstrf(pangulu_gstrf_options *gstrf_options, void **pangulu_handle)
    {
        if (gstrf_options == nullptr)
        {
            if (world()->rank == 0)
                printf("[PanguLU ERROR] Invalid input parameter. gstrf option struct pointer is NULL. Exit.\n");
            exit(1);
        }
        Solver *S = (Solver *)*pangulu_handle;
        if (S->analysis_only)
            fatal("PANGULU_AMD_ANALYSIS_ONLY handle: pattern, mapping and models only, nothing to factorise");
        NearDevice near(active_platform());
        numeric_factorize(*S);
    }

    void pangulu_gstrs(sparse_value_t *rhs, pangulu_gstrs_options *gstrs_options, void **pangulu_handle)
    {
        if (gstrs_options == nullptr)
        {
            if (world()->rank == 0)
                printf("[PanguLU ERROR] Invalid input parameter. gstrs option struct pointer is NULL. Exit.\n");
            exit(1);
        }
        Solver *S = (Solver *)*pangulu_handle;
        NearDevice near(active_platform());
        Comm *comm = world();
        std::vector<val_t> b(S->n);
        const bool scaled = !S->scale_row.empty();
        auto times = [](val_t v, double s) -> val_t
        {
#ifdef PANGULU_COMPLEX
            return val_t{(calculate_real_type)(v.re * s), (calculate_real_type)(v.im * s)};
#else
            return (val_t)(v * s);
#endif
        };
        if (comm->rank == 0)
        {
            for (u32 i = 0; i < S->n; i++)
            {
                if (S->perm[i] < S->n_user)
                    b[i] = scaled ? times(rhs[S->perm[i]], S->scale_row[S->perm[i]]) : rhs[S->perm[i]]; // A1 y = Dr b
                else
                    memset(&b[i], 0, sizeof(val_t)); // padding row: 1 * x = 0
            }
        }
        comm->bcast(b.data(), sizeof(val_t) * S->n, 0);
        comm->barrier();
        double t0 = wall_seconds();
        triangular_solve(*S, b.data());
        S->info.time_solve = wall_seconds() - t0;
        if (comm->rank == 0)
        {
            if (!scaled)
            {
                for (u32 i = 0; i < S->n; i++)
                    if (S->perm[i] < S->n_user)
                        rhs[S->perm[i]] = b[i];
            }
            else
            {
                // y (in the scaled matrix's column order) -> x[q(i)] = dc[q(i)] y[i]
                std::vector<val_t> y(S->n_user);
                for (u32 i = 0; i < S->n; i++)
                    if (S->perm[i] < S->n_user)
                        y[S->perm[i]] = b[i];
                for (u32 i = 0; i < S->n_user; i++)
                    rhs[S->match_col[i]] = times(y[i], S->scale_col[S->match_col[i]]);
            }
        }
    }

    void pangulu_gssv(sparse_value_t *rhs, pangulu_gstrf_options *gstrf_options, pangulu_gstrs_options *gstrs_options, void **pangulu_handle)
    {
        pangulu_gstrf(gstrf_options, pangulu_handle);
        pangulu_gstrs(rhs, gstrs_options, pangulu_handle);
    }

    void pangulu_finalize(void **pangulu_handle)
    {
        Solver *S = (Solver *)*pangulu_handle;
        delete S;
        *pangulu_handle = nullptr;
    }

    // ----------------------------------------------------------------------------------------------------
    void pangulu_amd_get_info(void **pangulu_handle, pangulu_amd_info_t *out)
    {
        Solver *S = (Solver *)*pangulu_handle;
        *out = S->info;
    }

    long long pangulu_amd_owned_block_count(void **pangulu_handle)
    {
        Solver *S = (Solver *)*pangulu_handle;
        return (long long)S->storage.owned.size();
    }

    int pangulu_amd_owned_block(void **pangulu_handle, long long idx, sparse_index_t *brow, sparse_index_t *bcol, int *is_upper,
                                unsigned long long *nnz, const pangulu_inblock_ptr **colptr, const pangulu_inblock_idx **rowidx,
                                const calculate_type **value)
    {
        Solver *S = (Solver *)*pangulu_handle;
        if (idx < 0 || (size_t)idx >= S->storage.owned.size())
            return 1;
        download_factors(*S);
        slot_t &s = S->storage.owned[(size_t)idx];
        *brow = s.brow_pos;
        *bcol = s.bcol_pos;
        *is_upper = s.is_upper;
        *nnz = s.columnpointer[S->nb];
        *colptr = s.columnpointer;
        *rowidx = s.rowindex;
        *value = s.value;
        return 0;
    }

    const sparse_index_t *pangulu_amd_get_perm(void **pangulu_handle)
    {
        Solver *S = (Solver *)*pangulu_handle;
        return S->perm.data();
    }

    int pangulu_amd_apply_lu(void **pangulu_handle, const calculate_type *x, calculate_type *y)
    {
        Solver *S = (Solver *)*pangulu_handle;
        if (S->nproc != 1)
            return 1;
        download_factors(*S);
        u32 nb = S->nb, n = S->n;
        std::vector<long double> t((size_t)S->nbk * nb, 0.0L), r((size_t)S->nbk * nb, 0.0L);
#ifdef PANGULU_COMPLEX
        (void)x;
        (void)y;
        (void)n;
        return 2; // complex factor check is done in Python from the exported blocks
#else
        // t = U x
        for (auto &s : S->storage.owned)
        {
            u32 r0 = s.brow_pos * nb, c0 = s.bcol_pos * nb;
            if (s.brow_pos == s.bcol_pos && s.is_upper)
            {
                for (u32 row = 0; row < nb; row++)
                    for (u32 p = s.columnpointer[row]; p < s.columnpointer[row + 1]; p++)
                        t[r0 + row] += (long double)s.value[p] * (long double)x[c0 + s.rowindex[p]];
            }
            else if (s.brow_pos < s.bcol_pos)
            {
                for (u32 c = 0; c < nb; c++)
                    for (u32 p = s.columnpointer[c]; p < s.columnpointer[c + 1]; p++)
                        t[r0 + s.rowindex[p]] += (long double)s.value[p] * (long double)x[c0 + c];
            }
        }
        // y = L t, L unit lower
        for (size_t i = 0; i < r.size(); i++)
            r[i] = t[i];
        for (auto &s : S->storage.owned)
        {
            u32 r0 = s.brow_pos * nb, c0 = s.bcol_pos * nb;
            bool lower_half = (s.brow_pos == s.bcol_pos && !s.is_upper) || s.brow_pos > s.bcol_pos;
            if (!lower_half)
                continue;
            for (u32 c = 0; c < nb; c++)
                for (u32 p = s.columnpointer[c]; p < s.columnpointer[c + 1]; p++)
                    r[r0 + s.rowindex[p]] += (long double)s.value[p] * t[c0 + c];
        }
        for (u32 i = 0; i < n; i++)
            y[i] = (calculate_type)r[i];
        return 0;
#endif
    }

    int pangulu_amd_factor_check(void **pangulu_handle, double *relative_error)
    {
        Solver *S = (Solver *)*pangulu_handle;
        if (!S->factored)
            return 1;
        NearDevice near(active_platform());
        *relative_error = factor_check(*S);
        return 0;
    }

    int pangulu_amd_factor_check_vectors(void **pangulu_handle, int nvec, unsigned long long seed, double *worst_relative_error)
    {
        Solver *S = (Solver *)*pangulu_handle;
        if (!S->factored)
            return 1;
        NearDevice near(active_platform());
        *worst_relative_error = factor_check_vectors(*S, nvec, seed);
        return 0;
    }

    int pangulu_amd_update_values(void **pangulu_handle, const sparse_value_t *csc_value)
    {
        Solver *S = (Solver *)*pangulu_handle;
        Comm *comm = world();
        if (!S->scale_row.empty() || S->analysis_only)
            return 1; // (with scaling on, new values would need a new matching: call pangulu_init)
        // the user's entry p of column j sits at (iperm[i], iperm[j]) of the permuted matrix; padding rows keep their unit diagonal
        const u64 nnz = S->info.nnz;
        std::vector<val_t> vals(nnz);
        if (comm->rank == 0)
            std::copy((const val_t *)csc_value, (const val_t *)csc_value + nnz, vals.begin());
        comm->bcast(vals.data(), sizeof(val_t) * nnz, 0);
        // Aperm holds the pattern; rebuild its values through the same permutation: entry order inside a permuted column is
        // by row, so locate each entry by binary search
        if (S->user_colptr.size() != (size_t)S->n_user + 1)
            return 2;
        CscMatrix &B = S->Aperm;
#pragma omp parallel for schedule(dynamic, 512)
        for (i64 j_ = 0; j_ < (i64)S->n_user; j_++)
        {
            const u32 j = (u32)j_, pj = S->iperm[j];
            for (u64 p = S->user_colptr[j]; p < S->user_colptr[j + 1]; p++)
            {
                const u32 pi = S->iperm[S->user_rowidx[p]];
                const u32 *b0 = B.rowidx.data() + B.colptr[pj], *b1 = B.rowidx.data() + B.colptr[pj + 1];
                const u32 *hit = std::lower_bound(b0, b1, pi);
                if (hit == b1 || *hit != pi)
                    fatal("pangulu_amd_update_values: entry (%u,%u) is not in the pattern the handle was initialised with", S->user_rowidx[p], j);
                B.value[hit - B.rowidx.data()] = vals[p];
            }
        }
        NearDevice near(active_platform());
        reload_values(*S, B);
        if (S->arena_snapshot)
            pangulu_amd_snapshot(pangulu_handle); // (bench-style resets restore the NEW values)
        return 0;
    }

    // a large host buffer on transparent huge pages where the system has them (see the host arena, pg_preprocess.cpp): free() releases it
    static char *big_host_alloc(size_t bytes)
    {
        void *p = nullptr;
        if (posix_memalign(&p, bytes >= ((size_t)2 << 20) ? ((size_t)2 << 20) : 64, std::max<size_t>(bytes, 64)) != 0)
            return nullptr;
#ifdef MADV_HUGEPAGE
        if (bytes >= ((size_t)2 << 20) && !getenv("PANGULU_AMD_NO_HUGEPAGES"))
            (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
        return (char *)p;
    }

    int pangulu_amd_snapshot(void **pangulu_handle)
    {
        Solver *S = (Solver *)*pangulu_handle;
        Platform &plat = active_platform();
        if (S->factored)
            return 1;
        if (!S->arena_snapshot)
        {
            if (plat.host_memory)
                S->arena_snapshot = big_host_alloc(S->storage.arena_bytes);
            else
            {
                // Where the copy lives (PANGULU_AMD_SNAPSHOT=device|host|auto): on the device a reset is one pass over HBM, but it
                // doubles the footprint of the records -- with `auto` (default) only while three times the records still fit what
                // is free now (records + copy + room for the dense mirrors) -- or, when the handle's schedule has been recorded and its
                // mirrors and descriptors are therefore allocated already, while the copy fits with 16 GB to spare; otherwise in host
                // memory (a reset is then an upload).
                const char *mode = getenv("PANGULU_AMD_SNAPSHOT");
                bool on_host = mode && strcmp(mode, "host") == 0;
                if (!mode || strcmp(mode, "auto") == 0)
                {
                    // (free and total memory of THIS device or partition: ADVICE r4 -- a 288 GB constant picked a device copy that
                    //  cannot fit on anything smaller, and the allocation aborts instead of failing)
                    size_t used = 0;
                    plat.get_device_memory_usage(&used);
                    const long long free_mib = plat.set_option ? (long long)plat.set_option(PANGULU_HIP_OPT_QUERY_FREE_MIB, 0) : -1;
                    const size_t free_now = free_mib > 0 ? (size_t)free_mib << 20 : 0;
                    const size_t total = used + free_now;
                    if (S->schedule_recorded) // (the dry run at init has already taken the mirrors and descriptors this handle needs)
                        on_host = free_now < S->storage.arena_bytes + ((size_t)16 << 30);
                    else
                        on_host = used + 3 * S->storage.arena_bytes > total;
                }
                S->snapshot_on_host = on_host;
                if (on_host)
                    S->arena_snapshot = big_host_alloc(S->storage.arena_bytes);
                else
                    plat.malloc_((void **)&S->arena_snapshot, S->storage.arena_bytes);
                if (!S->arena_snapshot)
                    return 1;
            }
        }
        if (plat.host_memory)
            plat.memcpy_(S->arena_snapshot, S->storage.darena, S->storage.arena_bytes, 2);
        else
            for (size_t c = 0; c < S->storage.dchunks.size(); c++)
                plat.memcpy_(S->arena_snapshot + c * S->storage.dchunk_bytes, S->storage.dchunks[c], S->storage.chunk_len(c), S->snapshot_on_host ? 1 : 2);
        plat.synchronize();
        S->info.snapshot_device_bytes = (plat.host_memory || S->snapshot_on_host) ? 0 : S->storage.arena_bytes;
        return 0;
    }

    int pangulu_amd_reset_numeric(void **pangulu_handle)
    {
        Solver *S = (Solver *)*pangulu_handle;
        Platform &plat = active_platform();
        if (!S->arena_snapshot)
            return 1;
        plat.synchronize();
        if (plat.host_memory)
            plat.memcpy_(S->storage.darena, S->arena_snapshot, S->storage.arena_bytes, 2);
        else
            for (size_t c = 0; c < S->storage.dchunks.size(); c++)
                plat.memcpy_(S->storage.dchunks[c], S->arena_snapshot + c * S->storage.dchunk_bytes, S->storage.chunk_len(c), S->snapshot_on_host ? 0 : 2);
        // Test mode (tests/test_multirank.py): the values of every record another rank sent in the logged factorisation are
        // overwritten with NaNs.  A replayed launch that read a receive slot before its block had arrived again would otherwise see the
        // previous factorisation's bit-identical record there, and a comparison of the factors could not tell.
        if (!plat.host_memory && getenv("PANGULU_AMD_POISON_RECV") && atoi(getenv("PANGULU_AMD_POISON_RECV")) != 0)
            for (const Solver::RankLog::Arrival &a : S->rank_log.arrivals)
                if (a.slot && a.slot->d_value && a.nnz)
                    plat.memset_(a.slot->d_value, 0xFF, (size_t)a.nnz * sizeof(val_t));
        plat.synchronize();
        S->remain = S->remain0;
        S->remain_diag = S->remain_diag0;
        S->rank_remain_task = S->rank_remain_task0;
        S->rank_remain_recv = S->rank_remain_recv0;
        for (auto &s : S->storage.owned)
            s.data_status = PANGULU_DATA_PREPARING;
        for (auto &q : S->pending)
            q.clear();
        S->pending_dirty.clear();
        S->pending_total = 0;
        S->factored = false;
        S->host_values_current = plat.host_memory;
        return 0;
    }

    int pangulu_amd_block_owner(void **pangulu_handle, sparse_index_t brow, sparse_index_t bcol)
    {
        Solver *S = (Solver *)*pangulu_handle;
        if (brow >= S->nbk || bcol >= S->nbk)
            return -1;
        return S->owner(brow, bcol);
    }

    int pangulu_amd_rank_model(void **pangulu_handle, double *tstar_seconds, double *flop, double *comm_seconds)
    {
        Solver *S = (Solver *)*pangulu_handle;
        const StructureModel &M = S->smodel;
        const int np = (int)M.rank_flop.size();
        for (int r = 0; r < np; r++)
        {
            if (tstar_seconds)
                tstar_seconds[r] = M.rank_time_hbm[(size_t)r] + M.rank_time_fp[(size_t)r];
            if (flop)
                flop[r] = M.rank_flop[(size_t)r];
            if (comm_seconds)
                comm_seconds[r] = M.rank_comm_s[(size_t)r];
        }
        return np;
    }

    int pangulu_amd_model_for_ranks(void **pangulu_handle, int nranks, double *out12)
    {
        Solver *S = (Solver *)*pangulu_handle;
        evaluate_model_for_ranks(*S, nranks, out12);
        return out12[0] < 0 ? 1 : 0;
    }

    void pangulu_amd_model_roofline(void **pangulu_handle, double hbm_gbytes_per_s, double fp_tflops)
    {
        Solver *S = (Solver *)*pangulu_handle;
        compute_task_model(*S, hbm_gbytes_per_s * 1e9, fp_tflops * 1e12);
    }
}
