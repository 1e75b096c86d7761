// pg_host.h -- internal declarations of the native host (analysis, storage, scheduler, transport).
//
// Restates, MI355X-first, the host side of the reference's hot path:
//   block records        src/pangulu_communication.c:1290-1393, src/pangulu_storage.c:199-426
//   dependency counters  src/pangulu_preprocessing.c:132-315, 443-556
//   heap + aggregator    src/pangulu_task.c:7-472
//   scheduler            src/pangulu_numeric.c:6-1080
//   p2p block exchange   src/pangulu_communication.c:93-105, 1786-1944
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <deque>
#include <vector>

#include "../../../include/pangulu_amd_ext.h"

namespace pg
{

using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;
using i64 = int64_t;
using val_t = calculate_type;
using slot_t = pangulu_storage_slot_t;
using task_t = pangulu_task_t;

[[noreturn]] void fatal(const char *fmt, ...);
double wall_seconds();

// ---------------------------------------------------------------------------------------------------------
// platform operator table (the reference's generated switch, src/pangulu_platform_helper.c:7-27)
// ---------------------------------------------------------------------------------------------------------
struct Platform
{
    unsigned id = 0;
    bool host_memory = false; // true: "device" pointers are host pointers (CPU platform)
    void *dl_handle = nullptr;
    void (*malloc_)(void **, size_t) = nullptr;
    void (*malloc_pinned)(void **, size_t) = nullptr;
    void (*synchronize)() = nullptr;
    void (*memset_)(void *, int, size_t) = nullptr;
    void (*create_stream)(void **) = nullptr;
    void (*memcpy_)(void *, const void *, size_t, unsigned) = nullptr;
    void (*memcpy_async)(void *, const void *, size_t, unsigned, void *) = nullptr;
    void (*free_)(void *) = nullptr;
    void (*get_device_num)(int *) = nullptr;
    void (*set_default_device)(int) = nullptr;
    void (*get_device_name)(char *, int) = nullptr;
    void (*get_device_memory_usage)(size_t *) = nullptr;
    void (*getrf)(pangulu_inblock_idx, slot_t *, int) = nullptr;
    void (*tstrf)(pangulu_inblock_idx, slot_t *, slot_t *, int) = nullptr;
    void (*gessm)(pangulu_inblock_idx, slot_t *, slot_t *, int) = nullptr;
    void (*ssssm)(pangulu_inblock_idx, slot_t *, slot_t *, slot_t *, int) = nullptr;
    void (*ssssm_batched)(pangulu_inblock_idx, pangulu_uint64_t, task_t *) = nullptr;
    void (*hybrid_batched)(pangulu_inblock_idx, pangulu_uint64_t, task_t *) = nullptr;
    void (*spmv)(pangulu_inblock_idx, slot_t *, val_t *, val_t *) = nullptr;
    void (*vecadd)(pangulu_int64_t, val_t *, val_t *) = nullptr;
    void (*sptrsv)(pangulu_inblock_idx, slot_t *, val_t *, pangulu_int64_t) = nullptr;
    // extension of the HIP back-end (nullptr elsewhere)
    int (*set_option)(int, long long) = nullptr;
    void (*prepare_diag)(pangulu_inblock_idx, slot_t *) = nullptr;
    void (*prepare_blocks)(pangulu_inblock_idx, pangulu_uint64_t, slot_t **) = nullptr;
    void *(*marker_record)() = nullptr;
    int (*marker_done)(void *) = nullptr;
    void (*marker_wait)(void *) = nullptr;
    int (*bind_near_device)(int) = nullptr;
    void (*block_trsv)(pangulu_inblock_idx, int, pangulu_uint64_t, const pangulu_uint64_t *, const pangulu_hip_solve_row_t *, slot_t *const *,
                       const pangulu_exblock_idx *, val_t *, pangulu_uint64_t) = nullptr;
    long long (*schedule)(int, const void *) = nullptr;
    int (*schedule_range)(const void *, long long, long long) = nullptr; // ranged replay of a recorded list (multi-rank)
    void *(*marker_record_replay)() = nullptr;
    void (*block_spmv_add)(pangulu_inblock_idx, pangulu_uint64_t, slot_t *const *, const pangulu_exblock_idx *, const pangulu_exblock_idx *, const int *,
                           const val_t *, val_t *, pangulu_uint64_t) = nullptr;
};
Platform &active_platform();               // built-in HIP unless the test hook replaced it
bool platform_is_builtin_hip();

// ---------------------------------------------------------------------------------------------------------
// analysis
// ---------------------------------------------------------------------------------------------------------
struct CscMatrix
{
    u32 n = 0;
    std::vector<u64> colptr;
    std::vector<u32> rowidx;
    std::vector<val_t> value;
    u64 nnz() const { return colptr.empty() ? 0 : colptr.back(); }
};

// perm[new] = old
void order_identity(u32 n, std::vector<u32> &perm);
// `align` > 0: pad so that large subtrees start on multiples of `align`; perm may then be longer than A.n, entries >= A.n
// are padding positions (isolated identity rows)
void order_nested_dissection(const CscMatrix &A, const double *coords, int dim, u32 align, std::vector<u32> &perm);
// B = P A P^T with sorted columns
void permute_symmetric(const CscMatrix &A, const std::vector<u32> &perm, CscMatrix &B);

// Fill pattern of the LU factors of a matrix with the symmetrised pattern of A (the reference's
// pangulu_symbolic_symmetric, src/pangulu_symbolic.c:132-247): lower triangle incl. diagonal, CSC, sorted.
struct Symbolic
{
    u32 n = 0;
    std::vector<u64> ptr;
    std::vector<u32> idx;
    u64 symbolic_nnz = 0; // 2*nnz(L) - n
    i64 flop = 0;         // sum_k (c_k + 2 c_k^2)
};
void symbolic_factorize(const CscMatrix &A, Symbolic &S);

// Whole-matrix block structure (every rank holds it; it is small: a few words per non-empty block).
struct BlockPattern
{
    u32 nb = 0, nbk = 0, n = 0;
    // lower block pattern incl. diagonal blocks, block-CSC: block rows >= bc, ascending
    std::vector<u64> lcolptr;
    std::vector<u32> lrowidx;
    std::vector<u32> lnnz; // nnz of each lower block (diagonal block: strictly-lower + diagonal entries)
    // all non-diagonal blocks, block-CSC (U blocks br<bc first, then L blocks br>bc) and block-CSR
    std::vector<u64> colptr, first_after_diag;
    std::vector<u32> rowidx;
    std::vector<u64> rowptr, first_after_diag_csr, csr_to_csc;
    std::vector<u32> colidx;
    std::vector<u32> nnz;              // per non-diagonal block (block-CSC index)
    std::vector<u32> diag_lower_nnz;   // strictly lower entries of diagonal block k
    std::vector<u32> diag_upper_nnz;   // upper incl. diagonal
    // locate block (br, bc), br != bc, in block-CSC order; returns ~0ull when structurally empty
    u64 find(u32 br, u32 bc) const;
};
void build_block_pattern(const Symbolic &S, u32 nb, BlockPattern &P);

// ---------------------------------------------------------------------------------------------------------
// storage: owned block records + size-classed receive bins (src/pangulu_storage.c)
// ---------------------------------------------------------------------------------------------------------
size_t record_bytes(u32 nb, u64 nnz, bool lower_offdiag); // incl. 32-byte header, 8-byte padded
// lay the slot's host/device pointers over a record that starts at (hrec, drec)
void bind_record(slot_t &s, u32 nb, u64 nnz, char *hrec, char *drec, bool lower_offdiag, bool diag_upper);

struct RecvBin
{
    size_t slot_capacity = 0;
    std::vector<slot_t> slots;
    std::deque<i32> free_list; // FIFO (round 4): a recycled slot goes to the back, so that a bin provisioned for every block the rank
                               // receives hands out every slot ONCE per factorisation (a rank's log can only be replayed then: slot
                               // addresses are in the recorded descriptors)
    char *hbuf = nullptr, *dbuf = nullptr;
};

struct Storage
{
    u32 nb = 0;
    std::vector<slot_t> owned;        // bin 0: off-diagonal owned blocks first, then diagonal halves (lower, upper)
    size_t n_owned_nondiag = 0;
    char *harena = nullptr, *darena = nullptr; // darena: host-memory platforms only (== harena)
    // device arena in separately allocated chunks (records never straddle a chunk): every chunk stays far below
    // 2 GiB, above which mapping another process's allocation (pg_comm_ipc.cpp) was seen to block forever
    std::vector<char *> dchunks;
    size_t dchunk_bytes = 0;
    char *device_ptr(size_t off) const { return dchunks.empty() ? darena + off : dchunks[off / dchunk_bytes] + off % dchunk_bytes; }
    size_t chunk_len(size_t c) const { return std::min(dchunk_bytes, arena_bytes - c * dchunk_bytes); }
    size_t arena_bytes = 0;
    std::vector<RecvBin> bins;        // bins[0] unused; 1..6 receive classes
    std::mutex mutex;
    slot_t *allocate(size_t bytes);   // first bin with capacity >= bytes and a free slot; nullptr if none
    void recycle(slot_t *s);
};

// ---------------------------------------------------------------------------------------------------------
// priority heap of panel tasks + per-destination SSSSM aggregation (src/pangulu_task.c)
// ---------------------------------------------------------------------------------------------------------
struct TaskHeap
{
    std::vector<task_t> store;
    std::vector<i64> heap;
    std::vector<i64> free_ids;
    std::mutex mutex;
    void reserve(size_t cap);
    void clear();
    void push(const task_t &t);
    bool pop(task_t &out);
    bool empty();
    size_t size();
    static bool before(const task_t &a, const task_t &b); // strategy 0 of pangulu_task_compare
};

// ---------------------------------------------------------------------------------------------------------
// transport
// ---------------------------------------------------------------------------------------------------------
struct BlockHeader // the 32-byte record header as it travels (src/pangulu_communication.c:1929-1942)
{
    u64 nnz;
    u32 brow, bcol;
    u32 is_upper;
    u32 bytes_lo; // total record bytes (ours: the reference derives it from MPI_Get_count)
    u64 reserved;
};
static_assert(sizeof(BlockHeader) == 32, "record header is 32 bytes");

// "Everything queued on the back-end up to a point of the scheduler's program order" as a handle that can be handed out
// BEFORE the platform has recorded it: with the launcher thread (pg_numeric.cpp) the platform calls trail the scheduler,
// and the launcher records the event (Platform::marker_record) when it gets there, in launch order.  `ev` is null until then.
struct Marker
{
    std::atomic<void *> ev{nullptr};
};

class Comm
{
public:
    int rank = 0, size = 1;
    int transport = PANGULU_AMD_TRANSPORT_HOST;
    virtual ~Comm() {}
    virtual void barrier() = 0;
    virtual void bcast(void *buf, size_t bytes, int root) = 0;
    virtual void allreduce_sum_i64(i64 *v, int count) = 0;
    virtual void allreduce_max_f64(double *v, int count) = 0;
    // small tagged host messages (SpTRSV partial sums), blocking
    virtual void send_bytes(int dst, int tag, const void *buf, size_t bytes) = 0;
    virtual void recv_bytes(int src, int tag, void *buf, size_t bytes) = 0;
    // block records: post a send of the record behind `s` (header + payload) to dst; returns at once
    virtual void isend_block(slot_t *s, const BlockHeader &h, int dst) = 0;
    // poll for an incoming block header; true when one is available (then recv_block must follow)
    virtual bool probe_block(BlockHeader &h, int &src) = 0;
    // receive the announced record into the slot's buffers (host mirror and/or device)
    virtual void recv_block(slot_t *s, const BlockHeader &h, int src) = 0;
    // pipelined variant for transports whose receives are asynchronous device copies: begin() may return before the
    // data is there, finish() completes everything begun since the last finish (defaults: synchronous receive)
    virtual void recv_block_begin(slot_t *s, const BlockHeader &h, int src) { recv_block(s, h, src); }
    virtual void recv_blocks_finish() {}
    // make sure everything posted has left (called before the slot memory may be reused / at the end)
    virtual void flush_sends() = 0;
    // Blocks posted from now on may only leave once `marker` has been recorded (by the launcher thread, possibly later) and
    // has completed; nullptr = at once.  Returns false when the transport cannot defer (host staging copies the values at
    // post time): the caller then has to drain the launcher and the device itself before posting.
    virtual bool set_send_gate(Marker * /*marker*/) { return false; }
    // collective: the device arena holding every record this rank owns (nullptr: records live in host memory);
    // transports that copy straight between arenas map their peers' here (pg_comm_ipc.cpp)
    virtual void register_arena(char *const * /*chunks*/, size_t /*nchunks*/, size_t /*chunk_bytes*/, size_t /*total_bytes*/) {}
    u64 sent_bytes = 0, recv_bytes_total = 0;
    // ranks whose RCCL communicators passed the start-up self-test (0: the data plane is not RCCL)
    virtual int rccl_ranks() const { return 0; }
    // true: a helper thread of this object was abandoned inside a library call (RCCL start-up that never returned) and may
    // still touch the object: set_world() then leaks it instead of deleting it
    bool abandoned = false;
};
Comm *world();             // never null: a 1-rank loopback by default
void set_world(Comm *c);   // takes ownership
void set_fake_world(int size); // checker's build: analysis-only handles for a rank count without the ranks
Comm *make_socket_comm(int rank, int size, const char *addr, int base_port, int transport, const void *nccl_id);
int rccl_make_unique_id(void *out128); // 0 on success; librccl.so is loaded lazily
Comm *make_rccl_comm(int rank, int size, const char *addr, int base_port, const void *nccl_id);
Comm *make_ipc_comm(int rank, int size, const char *addr, int base_port);

// ---------------------------------------------------------------------------------------------------------
// the solver instance behind the opaque handle
// ---------------------------------------------------------------------------------------------------------
// pg_scaling.cpp: maximum-product matching + scaling (the reference's MC64 step, src/pangulu_reordering.c:149-681)
bool max_product_matching(const CscMatrix &A, std::vector<u32> &col_of_row, std::vector<double> &dr, std::vector<double> &dc);
void apply_matching(const CscMatrix &A, const std::vector<u32> &col_of_row, const std::vector<double> &dr, const std::vector<double> &dc, CscMatrix &out);

struct Options
{
    bool scaling = false; // PANGULU_AMD_SCALING / pangulu_amd_set_scaling
    int ordering = PANGULU_AMD_ORDER_ND;
    std::vector<u32> user_perm;
    std::vector<double> coords;
    int coord_dim = 0;
    bool eager_host_mirror = false;
};
Options &pending_options();

struct TaskModel // SURVEY.md §8d algorithmic bytes / flops of the rank's tasks
{
    double bytes[5] = {0, 0, 0, 0, 0};
    double flop[5] = {0, 0, 0, 0, 0};
    u64 count[5] = {0, 0, 0, 0, 0};
};

// Structure-only model of the whole factorisation, evaluated by every rank at pangulu_init from the replicated symbolic
// pattern (pg_model.cpp): per-task algorithmic bytes / structural flops of SURVEY.md §8d for ALL tasks, whoever runs them.
// It weighs the block elimination tree for the subtree-to-rank mapping and gives T* per rank for any number of ranks.
struct StructureModel
{
    double hbm_bytes_per_s = 8e12, fp_flops_per_s = 78.6e12, link_bytes_per_s = 153e9;
    std::vector<u16> lcount;            // per lower block (BlockPattern::lcolptr order) and in-block column: entries
    // per block column c: T* seconds / structural flops of the tasks that EXECUTE in column c under a column-wise mapping:
    // GETRF(c), the panel solves of column c and row c, and every update whose destination (i, j) has min(i, j) = c
    std::vector<double> col_time, col_flop;
    // after the mapping: per rank
    std::vector<double> rank_time_hbm, rank_time_fp, rank_flop, rank_bytes, rank_comm_s;
    std::vector<double> sent_bytes;     // [from * nproc + to]
    std::vector<double> rank_mem_records, rank_mem_recv, rank_mem_mirrors; // HBM demand per rank under the mapping (bytes)
    double critical_path_s = 0;
    double critical_path_latency_s = 0; // the same chain with every task at max(T*_t, launch floor of its class) and hops between ranks
    u64 critical_path_tasks = 0;
};

struct Solver
{
    // configuration
    u32 n = 0, nb = 0, nbk = 0; // n includes the padding rows of an aligned ordering
    u32 n_user = 0;             // order of the matrix the user passed
    int rank = 0, nproc = 1, p = 1, q = 1;
    float recv_buffer_level = 0.5f;
    bool eager_host_mirror = false;
    // analysis products
    std::vector<u32> match_col;            // scaling on: column of the user's matrix that sits in column i of the scaled one
    std::vector<double> scale_row, scale_col; // scaling on: Dr, Dc (user's index space); empty otherwise
    std::vector<u32> perm, iperm;
    Symbolic sym;
    BlockPattern pat;
    CscMatrix Aperm;                       // kept for the factor check and residuals (rank 0 / all ranks)
    std::vector<u64> user_colptr;          // pattern of the user's matrix (for pangulu_amd_update_values)
    std::vector<u32> user_rowidx;
    // numeric state
    Storage storage;
    std::vector<slot_t *> slot_of;         // per non-diagonal block: owned / received slot, nullptr otherwise
    std::vector<slot_t *> diag_lower, diag_upper;
    std::vector<i32> remain;               // per non-diagonal block (reference: nondiag_remain_task_count)
    // per non-diagonal block: ranks (bit r) that run at least one update with this block as an operand.  A finished block is
    // forwarded to exactly those ranks (nproc <= 64; empty = the reference's rule: every owner in the process row / column)
    std::vector<u64> consumers;
    std::vector<i32> remain_diag;          // per level
    std::vector<i32> remain0, remain_diag0; // pristine copies (gstrf may be called once per init, kept for checks)
    i64 rank_remain_task = 0, rank_remain_task0 = 0;
    i64 rank_remain_recv = 0, rank_remain_recv0 = 0;
    TaskHeap heap;
    std::mutex info_mutex;
    // per-destination pending SSSSM tasks (the aggregator); index = owned slot index of the destination
    std::vector<std::vector<task_t>> pending;
    std::vector<u32> pending_dirty;        // owned slot indices with non-empty queues, in first-touch order
    u64 pending_total = 0;
    bool factored = false, host_values_current = true;
    bool schedule_recorded = false;        // the back-end holds the launch list of this handle's factorisation (one rank)
    // Multi-rank replay (round 4, PANGULU_AMD_MULTI_REPLAY=1): what THIS rank did in its first factorisation, in order -- the ranges
    // of the back-end's recorded operations its platform calls produced, the markers between them with the blocks that were
    // announced behind each, and which blocks of other ranks had arrived (into which receive slot) before each call was made.
    // A later factorisation replays the log: no task release, no descriptor building; it waits for arrivals where the first run
    // had them.  The per-rank orders are what HAPPENED in one run, so together with the messages they are acyclic: no deadlock.
    struct RankLog
    {
        struct Entry
        {
            long long op_end = 0;  // operations [previous entry's op_end, op_end) of the back-end's list belong to this entry
            int marker = -1;       // >= 0: record marker `marker` behind them (and post the sends gated by it)
            u32 need = 0;          // arrivals[0 .. need) of the first run must have arrived before the operations go out
        };
        struct Send
        {
            int marker;
            slot_t *slot;
            BlockHeader h;
            int dst;
        };
        struct Arrival
        {
            u32 brow, bcol, is_upper;
            slot_t *slot;
            u64 nnz; // (of the record that landed there: the poisoning of pangulu_amd_reset_numeric's test mode needs the extent of its values)
        };
        std::vector<Entry> entries;
        std::vector<Send> sends;        // in posting order (grouped by marker)
        std::vector<Arrival> arrivals;  // in the first run's order
        int nmarkers = 0;
        bool valid = false, slot_reused = false;
        bool unusable = false; // a first run has been logged and cannot be replayed (slot reuse, host staging): do not record again
    };
    RankLog rank_log;
    char *arena_snapshot = nullptr;        // pristine copy of the owned records, see pangulu_amd_snapshot
    bool snapshot_on_host = false;         // ... kept in host memory (PANGULU_AMD_SNAPSHOT, or too little HBM for a device copy)
    // statistics
    pangulu_amd_info_t info;
    TaskModel model;
    StructureModel smodel;
    // Ownership.  The reference is purely 2D block-cyclic (src/pangulu.c:83-90, owner = (br mod p) q + (bc mod q)).  Here
    // the panels (diagonal block, L column, U row) of every block column inside a SUBTREE of the block elimination
    // tree that was given to one rank belong to that rank (home[min(br,bc)] >= 0): disjoint subtrees are independent
    // computations, so the bottom of the tree -- where most levels are -- runs without any exchange and in lock-step
    // batches per rank; only the top of the tree (home < 0) is 2D block-cyclic.  Filled by assign_subtrees().
    std::vector<int> home;
    // ... or, home < 0 and grp non-empty: the column's blocks are 2D block-cyclic over the p x q grid of a rank GROUP
    // [lo, lo + p q) (proportional mapping: the groups shrink down the tree); grp empty: over the grid of all ranks
    struct Group
    {
        unsigned short lo, p, q;
    };
    std::vector<Group> grp;
    bool analysis_only = false; // PANGULU_AMD_ANALYSIS_ONLY=1: pattern, mapping, counters and models only -- no records, no gstrf
    int owner(u32 br, u32 bc) const
    {
        if (!home.empty())
        {
            const u32 c = br < bc ? br : bc;
            const int h = home[c];
            if (h >= 0)
                return h;
            if (!grp.empty())
            {
                const Group &g = grp[c];
                return (int)(g.lo + (br % g.p) * g.q + (bc % g.q));
            }
        }
        return (int)((br % (u32)p) * (u32)q + (bc % (u32)q));
    }
    ~Solver();
};

void assign_subtrees(Solver &S);                          // subtree-to-rank mapping (multi-rank runs)
void preprocess(Solver &S, const CscMatrix &Aperm);       // records, counters, bins, upload
void scatter_values(Solver &S, const CscMatrix &Aperm);   // values of the permuted matrix into the owned records (host side)
void reload_values(Solver &S, const CscMatrix &Aperm);    // new values on the same pattern: refill + upload + re-arm the counters
void numeric_factorize(Solver &S);                        // the hot path
void record_schedule(Solver &S);                          // pangulu_init, one rank: dry run of the scheduler, the back-end records the launch list
void download_factors(Solver &S);                         // device -> host mirror of owned values
void triangular_solve(Solver &S, val_t *rhs_permuted);    // forward + backward block sweeps (host kernels)
double factor_check(Solver &S);                           // ||L(U 1) - A 1|| / ||A 1|| on the factors where they are (pg_check.cpp; collective)
double factor_check_vectors(Solver &S, int nvec, unsigned long long seed); // the same on the ones vector + nvec - 1 random +-1 vectors: the worst quotient
void compute_task_model(Solver &S, double hbm_bytes_per_s, double fp_flops_per_s); // pg_model.cpp: T* of SURVEY.md §8d
void build_structure_model(Solver &S);   // pg_model.cpp: column counts per lower block + per-column work (before the mapping)
void evaluate_model_for_ranks(Solver &S, int nranks, double *out12); // pg_model.cpp: the same figures for any rank count
void compute_rank_model(Solver &S);      // pg_model.cpp: per-rank T*, flop shares, link term, critical path (after preprocess)
double task_structural_flop(u32 nb, const task_t &t);
// Checker's build only (oracle/pangulu_amd_test_hooks.h): execute every stride-th task of each kernel class and skip the
// rest -- a bounded, representative sample of the SAME factorisation for bench.py's cpu_baseline leg.  1 = everything.
extern int g_task_sample_stride;
extern int g_replay_enabled; // pangulu_amd_set_replay (-1: the environment decides)

} // namespace pg
