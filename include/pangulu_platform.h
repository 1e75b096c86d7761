/*
 * pangulu_platform.h -- C-ABI of the MI355X (gfx950) HIP back-end, platform id 0201001 "GPU_HIP".
 *
 * This is the drop-in boundary of the build.  The 21 entry points below are exactly the
 * operator table the reference generates from build_helper.py:8-32 and dispatches through
 * src/pangulu_platform_helper.c:7-27; the reference's CUDA implementation of the same table is
 * src/platforms/02_NONSHAREDMEM/01_GPU/000_CUDA/pangulu_platform_0201000.cu:52-979 and its CPU
 * implementation src/platforms/01_SHAREDMEM/00_CPU/000_CPU/pangulu_platform_0100000.c:14-506.
 *
 * The two descriptor structs are ABI-identical to the reference's (compiled with -DGPU_OPEN):
 *   pangulu_storage_slot_t  <- src/pangulu_common.h:207-231   (144 bytes)
 *   pangulu_task_t          <- src/pangulu_common.h:233-243   ( 48 bytes)
 * so a reference host built with PANGULU_DEFAULT_PLATFORM = PANGULU_PLATFORM_GPU_HIP can link this library
 * unchanged (INTEGRATION.md shows the three lines a maintainer adds).
 *
 * Value type is fixed at compile time like the reference (src/pangulu_common.h:11-33):
 *   -DCALCULATE_TYPE_R64 (default) | _R32 | _CR64 | _CR32
 */
#ifndef PANGULU_PLATFORM_H
#define PANGULU_PLATFORM_H

#include <stddef.h>
#include <stdint.h>

/* ---- scalar types (src/pangulu_common.h:35-70) ------------------------------------------------------- */
typedef long long int pangulu_int64_t;
typedef unsigned long long int pangulu_uint64_t;
typedef int pangulu_int32_t;
typedef unsigned int pangulu_uint32_t;
typedef short int pangulu_int16_t;
typedef unsigned short int pangulu_uint16_t;

typedef pangulu_uint64_t pangulu_exblock_ptr; /* pointer into the global (whole-matrix) CSC        */
typedef pangulu_uint32_t pangulu_exblock_idx; /* global row/column or block-row/block-column index */
typedef pangulu_uint32_t pangulu_inblock_ptr; /* pointer inside one nb x nb block                  */
typedef pangulu_uint16_t pangulu_inblock_idx; /* row/column inside one block (nb <= 65535)         */

#if defined(CALCULATE_TYPE_CR64)
#define PANGULU_COMPLEX 1
typedef double calculate_real_type;
#elif defined(CALCULATE_TYPE_CR32)
#define PANGULU_COMPLEX 1
typedef float calculate_real_type;
#elif defined(CALCULATE_TYPE_R32)
typedef float calculate_real_type;
#else
#ifndef CALCULATE_TYPE_R64
#define CALCULATE_TYPE_R64
#endif
typedef double calculate_real_type;
#endif

#ifdef PANGULU_COMPLEX
#if defined(__cplusplus)
/* layout-compatible with C99 `T _Complex` (re, im adjacent) */
typedef struct pangulu_complex_t
{
    calculate_real_type re, im;
} calculate_type;
#elif defined(CALCULATE_TYPE_CR64)
typedef double _Complex calculate_type;
#else
typedef float _Complex calculate_type;
#endif
#else
typedef calculate_real_type calculate_type;
#endif

/* ---- task ids, data states, tolerances (src/pangulu_common.h:124-134) --------------------------------- */
#define PANGULU_TASK_GETRF 1
#define PANGULU_TASK_TSTRF 2
#define PANGULU_TASK_GESSM 3
#define PANGULU_TASK_SSSSM 4
#define PANGULU_LOWER 0
#define PANGULU_UPPER 1
#define PANGULU_DATA_INVALID 0
#define PANGULU_DATA_PREPARING 1
#define PANGULU_DATA_READY 2
#define PANGULU_TOL 1e-16
#define PANGULU_SPTRSV_TOL 1e-16

/* platform ids: <shared/nonshared 2 digits><device class 2><implementation 3>, build_helper.py:60-87 */
#define PANGULU_PLATFORM_CPU_NAIVE 0x0100000
#define PANGULU_PLATFORM_GPU_CUDA 0x0201000
#define PANGULU_PLATFORM_GPU_HIP 0x0201001

#ifdef __cplusplus
extern "C"
{
#endif

    typedef struct pangulu_aggregate_queue_t
    {
        unsigned long long capacity;
        unsigned long long length;
        void *task_descriptors;
    } pangulu_aggregate_queue_t;

    /*
     * Block descriptor.  Host fields describe the host copy of the block record, d_* fields the device
     * copy.  Conventions the kernels rely on (src/pangulu_storage.c:247-421, pangulu_communication.c:1805-1895):
     *   off-diagonal block : CSC  -> columnpointer / rowindex / value          (d_columnpointer / d_rowindex / d_value)
     *   lower block only   : +CSR view -> rowpointer / columnindex / idx_of_csc_value_for_csr (and d_*)
     *   diagonal, lower    : strictly-lower CSC, is_upper = 0, same fields as an off-diagonal block
     *   diagonal, upper    : upper-incl-diagonal CSR (diagonal entry first in each row), is_upper = 1;
     *                        host fields are still named columnpointer/rowindex, the DEVICE fields are
     *                        d_rowpointer / d_columnindex (d_columnpointer is unset)
     *   related_block links the two halves of a diagonal block.
     * `value` points 32 bytes into the contiguous record (header: u64 nnz @-32, u32 brow @-24, u32 bcol @-20,
     * u32 is_upper @-16).
     */
    typedef struct pangulu_storage_slot_t
    {
        pangulu_exblock_idx brow_pos;
        pangulu_exblock_idx bcol_pos;
        pangulu_inblock_ptr *columnpointer;
        pangulu_inblock_idx *rowindex;
        calculate_type *value;
        pangulu_inblock_ptr *rowpointer;
        pangulu_inblock_idx *columnindex;
        pangulu_inblock_ptr *idx_of_csc_value_for_csr;
        volatile char data_status;
        struct pangulu_storage_slot_t *related_block;
        pangulu_int32_t is_upper;
        pangulu_int32_t bin_id;
        pangulu_int32_t slot_idx;
        pangulu_aggregate_queue_t *task_queue;
        pangulu_inblock_ptr *d_columnpointer;
        pangulu_inblock_idx *d_rowindex;
        calculate_type *d_value;
        pangulu_inblock_ptr *d_rowpointer;
        pangulu_inblock_idx *d_columnindex;
        pangulu_inblock_ptr *d_idx_of_csc_value_for_csr;
    } pangulu_storage_slot_t;

    typedef struct pangulu_task_t
    {
        pangulu_exblock_idx row;
        pangulu_exblock_idx col;
        pangulu_int16_t kernel_id;
        pangulu_exblock_idx task_level;
        pangulu_int64_t compare_flag;
        pangulu_storage_slot_t *opdst;
        pangulu_storage_slot_t *op1;
        pangulu_storage_slot_t *op2;
    } pangulu_task_t;

#if defined(__cplusplus)
    static_assert(sizeof(pangulu_storage_slot_t) == 144, "slot ABI must match src/pangulu_common.h:207-231 (GPU_OPEN)");
    static_assert(offsetof(pangulu_storage_slot_t, value) == 24, "slot ABI");
    static_assert(offsetof(pangulu_storage_slot_t, related_block) == 64, "slot ABI");
    static_assert(offsetof(pangulu_storage_slot_t, d_columnpointer) == 96, "slot ABI");
    static_assert(offsetof(pangulu_storage_slot_t, d_value) == 112, "slot ABI");
    static_assert(sizeof(pangulu_task_t) == 48, "task ABI must match src/pangulu_common.h:233-243");
    static_assert(offsetof(pangulu_task_t, opdst) == 24, "task ABI");
#endif

    /* Globals the reference host defines (src/pangulu.c:7-9) and its GPU back-end reads
     * (pangulu_platform_0201000.cu:7-9).  This library defines them WEAK so it links both against the
     * reference host (whose strong definitions win) and stand-alone.  The HIP kernels size their
     * workgroups for 64-wide wavefronts themselves; the two "warp_per_block" knobs are accepted and used as
     * wavefronts-per-workgroup hints. set_default_device() writes pangulu_gpu_shared_mem_size. */
    extern int pangulu_gpu_kernel_warp_per_block;
    extern int pangulu_gpu_data_move_warp_per_block;
    extern int pangulu_gpu_shared_mem_size;

    /* ---- the 21 operators, one per row of build_helper.py:8-32 ------------------------------------- */
    /* runtime shims: replaces pangulu_platform_0201000.cu:52-163 */
    void pangulu_platform_0201001_malloc(void **platform_address, size_t size);
    void pangulu_platform_0201001_malloc_pinned(void **platform_address, size_t size);
    void pangulu_platform_0201001_synchronize(void);
    void pangulu_platform_0201001_memset(void *s, int c, size_t n);
    void pangulu_platform_0201001_create_stream(void **stream);
    /* kind: 0 = host->device, 1 = device->host, 2 = device->device (…0201000.cu:82-103).
     * memcpy, memcpy_async and synchronize may be called from a second thread (the reference's receive thread does,
     * src/pangulu_communication.c:1850,1880) while another is inside hybrid_batched: they order themselves behind everything queued so
     * far under the back-end's lock; memcpy returns when ITS copy is complete. */
    void pangulu_platform_0201001_memcpy(void *dst, const void *src, size_t count, unsigned int kind);
    void pangulu_platform_0201001_memcpy_async(void *dst, const void *src, size_t count, unsigned int kind, void *stream);
    void pangulu_platform_0201001_free(void *devptr);
    void pangulu_platform_0201001_get_device_num(int *device_num);
    void pangulu_platform_0201001_set_default_device(int device_num);
    void pangulu_platform_0201001_get_device_name(char *name, int device_num);
    void pangulu_platform_0201001_get_device_memory_usage(size_t *used_byte);

    /* numeric kernels: replaces …0201000.cu:547-909 (CPU semantics: …0100000.c:57-431) */
    void pangulu_platform_0201001_getrf(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, int tid);
    void pangulu_platform_0201001_tstrf(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *opdiag, int tid);
    void pangulu_platform_0201001_gessm(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *opdiag, int tid);
    void pangulu_platform_0201001_ssssm(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *op1, pangulu_storage_slot_t *op2, int tid);
    void pangulu_platform_0201001_ssssm_batched(pangulu_inblock_idx nb, pangulu_uint64_t ntask, pangulu_task_t *tasks);
    void pangulu_platform_0201001_hybrid_batched(pangulu_inblock_idx nb, pangulu_uint64_t ntask, pangulu_task_t *tasks);

    /* solve-side kernels: the reference's GPU versions are empty stubs (…0201000.cu:958-979); these are
     * real device kernels with the CPU semantics of …0100000.c:435-506 (x, y are DEVICE pointers). */
    void pangulu_platform_0201001_spmv(pangulu_inblock_idx nb, pangulu_storage_slot_t *a, calculate_type *x, calculate_type *y);
    void pangulu_platform_0201001_vecadd(pangulu_int64_t length, calculate_type *bval, calculate_type *xval);
    void pangulu_platform_0201001_sptrsv(pangulu_inblock_idx nb, pangulu_storage_slot_t *s, calculate_type *xval, pangulu_int64_t uplo);

    /* ---- extensions (not in the reference table; safe to ignore) ----------------------------------- */
    /* Behaviour switches of the back-end.
     *   PANGULU_HIP_OPT_HOST_MIRROR (default 1): after GETRF/TSTRF/GESSM copy the block's values back into
     *     slot->value like …0201000.cu:639-640,680,714 does (the reference host's MPI send and SpTRSV read
     *     host memory).  The native host keeps factors device-resident and sets 0.
     *   PANGULU_HIP_OPT_DENSE_THRESHOLD_PERMILLE (default 2): an update C -= A*B whose operands have a
     *     geometric-mean fill sqrt(dA*dB) of at least this many per mille runs on the f64 MFMA kernel on dense
     *     mirrors of the three blocks (kept in HBM, built once per block); a destination that has a mirror
     *     accumulates all its updates there.  1000 = only completely full operands (the reference's
     *     cuBLAS-direct rule, …0201000.cu:827); 1001 disables the dense path.
     */
#define PANGULU_HIP_OPT_HOST_MIRROR 1
#define PANGULU_HIP_OPT_DENSE_THRESHOLD_PERMILLE 2
    /*   PANGULU_HIP_OPT_PROFILE (default 0): time every kernel launch with hipEvents (see get_stats).
     *   PANGULU_HIP_OPT_ASSUME_INDEPENDENT (default 0): the caller guarantees that the tasks of one
     *     hybrid_batched call do not depend on each other (other than SSSSM tasks sharing a destination, which
     *     must then be adjacent in the array); skips the hazard scan that otherwise splits the array into
     *     dependent phases.  The native scheduler sets it. */
#define PANGULU_HIP_OPT_PROFILE 3
#define PANGULU_HIP_OPT_ASSUME_INDEPENDENT 4
    /*   PANGULU_HIP_OPT_GETRF_STRICT_ORDER (default 0): 1 = reproducible operation order everywhere: the
     *     pattern-driven GETRF kernel for every block (the only one for value types other than R64) and the
     *     one-entry-at-a-time sparse SSSSM kernel (fused multiply-adds in ascending pivot order, no LDS atomics);
     *     0 lets R64 diagonal blocks use the LDS-blocked MFMA kernel and sparse updates run four op2 entries at a
     *     time with LDS floating-point atomics. */
#define PANGULU_HIP_OPT_GETRF_STRICT_ORDER 5
    /*   PANGULU_HIP_OPT_COUNT_FLOPS (default 1): also count the structural flops of updates that run on the dense
     *     MFMA kernel (one extra pass over op2's pattern per such task, like the reference's PERF counters
     *     src/pangulu_kernel_interface.c:161-176); 0 skips that pass (flops[5] then stays 0).
     *   PANGULU_HIP_OPT_RESET_BLOCK_STATE: forget all dense mirrors' contents (value ignored).  A host that restores
     *     block values behind the back-end's back (bench.py's reset between repeated factorisations) must call it. */
#define PANGULU_HIP_OPT_COUNT_FLOPS 6
#define PANGULU_HIP_OPT_RESET_BLOCK_STATE 7
    /*   PANGULU_HIP_OPT_SSSSM_GROUP_CHUNK (default 8): updates queued on one destination run as one sequential pass
     *     per destination column (deterministic, no atomics) while there are at most this many; longer queues are
     *     cut into chunks of this size that run concurrently and add their partial sums with floating-point
     *     atomics (results then vary in the last bits from run to run).  0 = never split. */
#define PANGULU_HIP_OPT_SSSSM_GROUP_CHUNK 8
    /*   PANGULU_HIP_OPT_TRSM_DENSE_PERMILLE (default 5): TSTRF/GESSM on a block with at least this fill (or whose
     *     updates already live in a dense mirror) run as panel-wise dense solves on the matrix cores against the dense
     *     LU image GETRF leaves behind; 1001 keeps every solve on the sparse kernel. */
#define PANGULU_HIP_OPT_TRSM_DENSE_PERMILLE 9
    /*   PANGULU_HIP_OPT_TWO_STREAMS (default 1): run the MFMA update kernel of a batch on a side stream beside the LDS
     *     update kernel (fork/join with events inside the call; everything else stays on the one in-order stream). */
#define PANGULU_HIP_OPT_TWO_STREAMS 10
    /*   PANGULU_HIP_OPT_SMALL_LAUNCH_TASKS (default 2048): an update launch with at most this many tasks gives every
     *     update its own workgroups (chunk 1) - near the root of the elimination tree latency matters, not traffic. */
#define PANGULU_HIP_OPT_SMALL_LAUNCH_TASKS 11
    /*   PANGULU_HIP_OPT_XCD_SWIZZLE (default 1): map workgroup ids so that the workgroups of one update queue / one
     *     destination / one solve run on the same XCD and share its L2 (0: hardware round-robin order). */
#define PANGULU_HIP_OPT_XCD_SWIZZLE 12
    /*   PANGULU_HIP_OPT_RECORDS_STREAM (default 1; environment PANGULU_HIP_RECORDS_STREAM at start-up): the sparsify jobs
     *     that bring the sparse records of finished blocks up to date run on their own stream beside the next kernels;
     *     0 keeps them on the main stream.  bench.py's profile pass turns it (and TWO_STREAMS) off so that the hipEvent
     *     pair around a launch brackets that kernel and nothing else. */
#define PANGULU_HIP_OPT_RECORDS_STREAM 13
    /*   PANGULU_HIP_OPT_BACKGROUND_UPDATES (default 1; environment PANGULU_HIP_BACKGROUND_UPDATES at start-up): in a
     *     dependency-free call (ASSUME_INDEPENDENT) that carries diagonal factorisations and updates but no panel solves --
     *     the native scheduler's look-ahead -- the update kernels run on a background stream that the main stream does not
     *     join at the end of the call; a later call that touches one of their destinations waits for them first (the native scheduler
     *     issues the panel-tile updates ahead of such a call, PANGULU_AMD_PANEL_FIRST).  0: GETRFs on a side stream, everything
     *     joined at the end of the call. */
#define PANGULU_HIP_OPT_BACKGROUND_UPDATES 14
    /*   PANGULU_HIP_OPT_FRONT_STAGES (default 2; environment PANGULU_HIP_FRONT_STAGES at start-up): (destination, 128 x 128
     *     tile) pairs whose queued updates are all dense-front products -- every 16 x 16 piece of both operands that meets
     *     the tile holds pattern entries -- need no occupancy bookkeeping.  1: they run inside the general launch on its
     *     no-step-list path (one launch, one tail; needs TILES_STAGES != 0); 2, 3 or 4: on the dense-front kernel of their
     *     own (operand slabs by LDS-DMA, that many LDS stages) when a launch has at least PANGULU_HIP_FRONT_MIN_WGS (8192) of
     *     them, inside the general launch otherwise; 0: treated like any other tile. */
#define PANGULU_HIP_OPT_FRONT_STAGES 15
    /*   PANGULU_HIP_OPT_TILES_STAGES (default 2; environment PANGULU_HIP_TILES_STAGES at start-up): the general MFMA update
     *     kernel.  Non-zero: operand slabs by LDS-DMA in two LDS stages, strided piece ownership of the wavefronts, step records
     *     fetched a step ahead, the DMA instructions of the next slab issued behind the first products of the current one
     *     (ssssm_tilesv_f64_kernel); 0: round 2's kernel (register staging, contiguous 64 x 32 sub-tiles).  The values 1, 3, 4
     *     and 5 used to select earlier / experimental kernels (ssssm_tiles_f64_kernel<STAGES>, ssssm_tilesp_f64_kernel): those
     *     are in tools/experiments/ since round 6 and the values now mean the default kernel. */
#define PANGULU_HIP_OPT_TILES_STAGES 16
    /*   PANGULU_HIP_OPT_QUERY_FREE_MIB: a QUERY, nothing is set (value ignored): returns the free memory of the back-end's device in MiB
     *     (hipMemGetInfo), so that a host decides placements (pangulu_amd_snapshot) by what THIS device or partition has, not by a
     *     288 GB constant. */
#define PANGULU_HIP_OPT_QUERY_FREE_MIB 17
    int pangulu_platform_0201001_set_option(int option, long long value);
    /* Optional: build, ahead of the numeric phase, the by-column view of a diagonal block's upper (CSR) half
     * that SSSSM updates INTO that block need (it is built lazily on first use otherwise, which costs an
     * allocation and three small copies inside the factorisation).  `diag` may be either half. */
    void pangulu_platform_0201001_prepare_diag(pangulu_inblock_idx nb, pangulu_storage_slot_t *diag);
    /* Once per owned off-diagonal block at preprocessing (host pattern arrays valid): the back-end summarises which
     * 16 x 16 tiles of the symbolic pattern hold entries, so that its launches can leave out workgroups (128 x 128
     * update tiles, 64-wide solve slabs) that would find nothing to do.  Optional; without it every workgroup is launched. */
    void pangulu_platform_0201001_prepare_blocks(pangulu_inblock_idx nb, pangulu_uint64_t nslot, pangulu_storage_slot_t **slots);
    /* Optional: a waitable handle for "everything queued on the back-end so far" (a hipEvent_t on the back-end's
     * stream, owned by the back-end).  The native multi-rank scheduler uses it to announce finished blocks and to
     * recycle receive slots without draining the device after every batch. */
    void *pangulu_platform_0201001_marker_record(void);
    int pangulu_platform_0201001_marker_done(void *marker);  /* 1 when everything before the marker has completed */
    void pangulu_platform_0201001_marker_wait(void *marker);
    /* Optional: level-scheduled block triangular solve for pangulu_gstrs on one rank, on the device-resident factors (no
     * download): the reference makes one spmv / sptrsv platform call per block, on its CPU platform
     * (src/pangulu_sptrsv.c:62,94,126,159); this call sweeps all levels of the block dependency graph, one launch per
     * level.  `x` is a HOST vector of `xlen` values, overwritten with the solution of the sweep (upper = 0: L, unit
     * diagonal; 1: U).  rows[level_ptr[l] .. level_ptr[l+1]) are the block rows of level l, `diag` their diagonal half
     * (lower / upper), blk_slots / blk_bcol[first .. first + nblk) their off-diagonal blocks on the sweep's side. */
    typedef struct pangulu_hip_solve_row_t
    {
        pangulu_exblock_idx brow, nblk;
        pangulu_uint64_t first;
        pangulu_storage_slot_t *diag;
    } pangulu_hip_solve_row_t;
    void pangulu_platform_0201001_block_trsv(pangulu_inblock_idx nb, int upper, pangulu_uint64_t nlevel, const pangulu_uint64_t *level_ptr,
                                             const pangulu_hip_solve_row_t *rows, pangulu_storage_slot_t *const *blk_slots,
                                             const pangulu_exblock_idx *blk_bcol, calculate_type *x, pangulu_uint64_t xlen);
    /* Optional: y[dst segment] += A_blk * x[src segment] for a list of device-resident block records, in one launch (one
     * workgroup per block, floating-point atomics on y).  `x` and `y` are HOST vectors of `xlen` values (nb per block
     * row / column); y is read, updated and written back.  csr[i] != 0: the record is an upper diagonal half (CSR, diagonal
     * first).  The native host builds the reference's factor check ||L(U.1) - A.1|| / ||A.1|| from it
     * (src/pangulu_numeric.c:1082-1341) without downloading the factors. */
    void pangulu_platform_0201001_block_spmv_add(pangulu_inblock_idx nb, pangulu_uint64_t nblk, pangulu_storage_slot_t *const *slots,
                                                 const pangulu_exblock_idx *src_seg, const pangulu_exblock_idx *dst_seg, const int *csr,
                                                 const calculate_type *x, calculate_type *y, pangulu_uint64_t xlen);
    /* Optional: keep the calling thread (and threads it creates afterwards) on the CPUs of the NUMA node the device hangs off
     * while enable = 1, restore its previous affinity mask with enable = 0 (the reference pins its threads as well,
     * src/pangulu_thread.c:3-12).  The native host calls it around pangulu_init / gstrf / gstrs.  Returns 0 when the mask
     * was changed / restored, non-zero when there was nothing to do (no NUMA information, PANGULU_AMD_BIND_NUMA=0). */
    int pangulu_platform_0201001_bind_near_device(int enable);
    /* Optional: the static schedule of a factorisation.  For a host whose call sequence is a pure function of the block
     * pattern (one rank, dependency-free batches in a fixed order -- the native scheduler's) the launches of one
     * factorisation can be recorded once and replayed for every later factorisation of the same pattern, with no host
     * work per task.  cmd 1: start recording everything hybrid_batched issues, for `owner` (an opaque token); 4: like 1,
     * but RECORD ONLY -- the calls that follow go through the back-end's launch code and build their descriptors, nothing is
     * launched and no block is touched (the native host's dry run of its scheduler at pangulu_init); 2: stop (returns the
     * number of recorded operations); 3: replay (0 = replayed, 1 = nothing valid: other owner, other options, nothing
     * recorded, or a resource of the back-end the recorded launches point into -- GETRF scratch, mirror pool -- has been
     * re-allocated since, e.g. by a handle with another block order); 0: drop (the owner's blocks are about to be freed).
     * Profiled runs and the eager host mirror are not recorded (cmd 1 / 4 return -1).  The host-side counters of get_stats
     * (launches, tasks, alg_bytes, workgroup counts) count a dry run not at all and every replay once. */
    long long pangulu_platform_0201001_schedule(int cmd, const void *owner);
    /* Multi-rank hosts (round 4): cmd 5 = record like cmd 1 with the descriptor segments packed (thousands of small batches);
     * cmd 6 = number of operations recorded so far; cmd 7 / 8 = a replay in RANGES begins (same validity rule as cmd 3; 0 = go) /
     * ends; schedule_range replays the operations [first, last) in between -- the host waits for the blocks of other ranks
     * between its batches --; marker_record_replay records a marker at the current point of the main stream and nothing else
     * (the stream joins a marker needs are operations of the recorded list). */
    int pangulu_platform_0201001_schedule_range(const void *owner, long long first, long long last);
    void *pangulu_platform_0201001_marker_record_replay(void);
    /* stream all numeric kernels are launched on (a hipStream_t); for event timing in bench.py */
    void *pangulu_platform_0201001_get_stream(void);
    /* Cumulative per-kernel-class counters since the last reset.  Classes: 1 GETRF, 2 TSTRF, 3 GESSM,
     * 4 SSSSM (sparse LDS-accumulator kernel), 5 SSSSM (dense MFMA kernel); index 0 unused.
     * Classes 6..8 are the MIRROR MAINTENANCE of the dense-mode blocks -- no task of the reference's model, so no
     * algorithmic bytes or flops, but real device time and HBM traffic: 6 densify (record -> dense mirror), 7 sparsify
     * (mirror -> record), 8 LU images of remote diagonal blocks (half_image + tile inversion).  For them `tasks` counts
     * blocks and `alg_bytes` the bytes the jobs move by construction (record read + image written, or the reverse).
     *   alg_bytes : algorithmic HBM bytes of the launched tasks (SURVEY.md §8d formulas, from nnz only)
     *   flops     : structural flops the kernels executed (counted on the device for the sparse kernels,
     *               2*nb^3 per task for the dense kernel)
     *   elapsed_ms: sum of launch durations from hipEvents on the back-end stream; only collected while
     *               PANGULU_HIP_OPT_PROFILE is 1 (it adds two event records per launch).  One solve launch serves TSTRF and GESSM
     *               tasks together: it counts as a launch of each class it carries tasks of, and its duration is split between
     *               classes 2 and 3 in proportion to their algorithmic bytes (round 6; it used to go to the larger class whole) */
#define PANGULU_HIP_STAT_CLASSES 9
    typedef struct pangulu_hip_stats_t
    {
        unsigned long long launches[PANGULU_HIP_STAT_CLASSES];
        unsigned long long tasks[PANGULU_HIP_STAT_CLASSES];
        double alg_bytes[PANGULU_HIP_STAT_CLASSES];
        double flops[PANGULU_HIP_STAT_CLASSES];
        double elapsed_ms[PANGULU_HIP_STAT_CLASSES];
        double mfma_flops_executed; /* class 5: flops the matrix cores actually executed (16x16x16 products issued x 8192;
                                     *  structurally empty tiles are skipped); counted while COUNT_FLOPS is on */
        unsigned long long trsm_dense_tasks; /* TSTRF/GESSM tasks that took the dense MFMA path */
        /* class 5: workgroups (destination tile x update queue) launched on the dense-front kernel / on the general MFMA kernel */
        unsigned long long ssssm_front_workgroups, ssssm_general_workgroups;
        /* GETRF -> dense-solve chase: launches that carried a level's factorisations AND its dense solves, and the solves in them */
        unsigned long long chase_launches, chase_solves;
        /* class 5 by kernel (round 6; appended, older callers' layout is a prefix): the two MFMA update kernels of a class-5 launch
         * timed on their own while PANGULU_HIP_OPT_PROFILE is 1 ([0] dense-front kernel, [1] general kernel), and the flops the
         * dense-front kernel's products executed (the general kernel's = mfma_flops_executed - this), counted under COUNT_FLOPS */
        double ssssm_kernel_ms[2];
        double ssssm_front_flops_executed;
    } pangulu_hip_stats_t;
    void pangulu_platform_0201001_get_stats(pangulu_hip_stats_t *out, int reset);
    /* Device memory the BACK-END holds for itself, in bytes (the host's records, receive bins and snapshots are the host's
     * allocations): [0] the dense-mirror pool (chunks allocated so far), [1] HBM twins of the descriptor segments of a recorded
     * schedule, [2] GETRF scratch images, [3] the most mirrors (blocks in dense mode) in use at once. */
    void pangulu_platform_0201001_get_memory(unsigned long long out[4]);

#ifdef __cplusplus
}
#endif

#endif /* PANGULU_PLATFORM_H */
