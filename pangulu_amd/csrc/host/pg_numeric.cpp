// pg_numeric.cpp -- the synchronisation-free numeric factorisation (the hot path's host side).
//
// Same dependency-counter scheme as the reference (src/pangulu_numeric.c:6-1080, src/pangulu_task.c:7-472,
// counters described in SURVEY.md §3.4), re-designed around an asynchronous device:
//   * panel tasks (GETRF / TSTRF / GESSM) go through the priority heap (strategy 0 of pangulu_task_compare,
//     src/pangulu_task.c:268-281) and are drained into ONE platform hybrid_batched call per drain, GETRFs
//     included (the reference runs every GETRF alone, src/pangulu_numeric.c:987-991);
//   * SSSSM updates never enter the heap.  The reference pushes each one twice (heap + per-tile aggregator,
//     src/pangulu_task.c:373-385) and the heap copy's only effect when popped is "if the tile's counter is 1,
//     push its panel task" (src/pangulu_numeric.c:602-650).  Counters only decrease, so that test is made
//     right when the update is queued: identical task graph, no heap traffic for the most numerous task;
//   * queued updates of a tile are flushed, as one batch together with those of every other tile of the
//     drain, right before the tile's panel task (src/pangulu_numeric.c:311) or when the rank is idle
//     (src/pangulu_task.c:93-177);
//   * single-rank runs never synchronise with the device inside the loop: every batch goes to the back-end's
//     one in-order stream, so program order is dependency order and the host simply runs ahead.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <unistd.h>

#include "pg_host.h"

namespace pg
{

int g_task_sample_stride = 1;

// ---------------------------------------------------------------------------------------------------------
// heap
// ---------------------------------------------------------------------------------------------------------
bool TaskHeap::before(const task_t &a, const task_t &b)
{
    if (a.compare_flag != b.compare_flag)
        return a.compare_flag < b.compare_flag;
    i64 ka = (i64)a.row + (i64)a.col - a.compare_flag;
    i64 kb = (i64)b.row + (i64)b.col - b.compare_flag;
    return ka < kb;
}

void TaskHeap::reserve(size_t cap)
{
    store.reserve(cap);
    heap.reserve(cap);
}

void TaskHeap::clear()
{
    std::lock_guard<std::mutex> g(mutex);
    store.clear();
    heap.clear();
    free_ids.clear();
}

void TaskHeap::push(const task_t &t)
{
    std::lock_guard<std::mutex> g(mutex);
    i64 id;
    if (!free_ids.empty())
    {
        id = free_ids.back();
        free_ids.pop_back();
        store[(size_t)id] = t;
    }
    else
    {
        id = (i64)store.size();
        store.push_back(t);
    }
    size_t son = heap.size();
    heap.push_back(id);
    while (son > 0)
    {
        size_t parent = (son - 1) / 2;
        if (!before(store[(size_t)heap[son]], store[(size_t)heap[parent]]))
            break;
        std::swap(heap[son], heap[parent]);
        son = parent;
    }
}

bool TaskHeap::pop(task_t &out)
{
    std::lock_guard<std::mutex> g(mutex);
    if (heap.empty())
        return false;
    i64 top = heap[0];
    out = store[(size_t)top];
    free_ids.push_back(top);
    heap[0] = heap.back();
    heap.pop_back();
    size_t len = heap.size(), cur = 0;
    while (true)
    {
        size_t l = 2 * cur + 1, r = l + 1, best = cur;
        if (l < len && before(store[(size_t)heap[l]], store[(size_t)heap[best]]))
            best = l;
        if (r < len && before(store[(size_t)heap[r]], store[(size_t)heap[best]]))
            best = r;
        if (best == cur)
            break;
        std::swap(heap[cur], heap[best]);
        cur = best;
    }
    return true;
}

bool TaskHeap::empty()
{
    std::lock_guard<std::mutex> g(mutex);
    return heap.empty();
}

size_t TaskHeap::size()
{
    std::lock_guard<std::mutex> g(mutex);
    return heap.size();
}

// ---------------------------------------------------------------------------------------------------------
// scheduler
// ---------------------------------------------------------------------------------------------------------
namespace
{

struct Sched
{
    Solver &S;
    const BlockPattern &P;
    Platform &plat;
    Comm *comm;
    std::vector<char> sent_flag;
    std::vector<task_t> batch, ssssm_batch, combined;
    size_t lookahead_max_getrf = 32;  // PANGULU_AMD_LOOKAHEAD_MAX_GETRF (0 disables; 128 until round 3's sweep on replayed runs: fem27(112) 836.1 / 845.0 / 902.9 ms at 32 / 128 / 1024)
    bool panel_lookahead_on = true;   // PANGULU_AMD_PANEL_LOOKAHEAD (0 disables)
    // Round 5: a look-ahead call used to take EVERY queued update.  A destination tile is read and written once per launch it is part of
    // whatever the depth of its queue, and a work item's fixed cost is 20-48 % of its life: on elastic3d(77) the queues
    // were two deep on average.  Now a destination waits until MIN_QUEUE updates are queued on it -- or until everything it will ever
    // receive is (queue_is_complete) --, provided there is enough queued work for the device anyway (DEFER_FROM updates in all).
    // elastic3d(77): 1 596 -> 1 543 ms, update class 1 545 -> 1 436 ms (profiles/r05a{h,i,j,k}_*: 2 / 3 / 4 / 5 / 6 / 8 / 16 = 1 564 / 1 543 /
    // 1 548 / 1 552 / 1 555 / 1 564 / 1 572 ms; everything deferred to the destination's own panel task: 1 633).
    size_t lookahead_min_queue = 3;     // PANGULU_AMD_LOOKAHEAD_MIN_QUEUE (1: round 4's behaviour)
    size_t lookahead_defer_from = 8192; // PANGULU_AMD_LOOKAHEAD_DEFER_FROM
    size_t lookahead_min_tasks = 0;     // PANGULU_AMD_LOOKAHEAD_MIN_TASKS: a floor on the size of a look-ahead call with deferral (measured: no gain)
    bool panel_first_on = true;       // PANGULU_AMD_PANEL_FIRST (0: panel-tile updates inside the look-ahead call, round 2's order)
    // multi-rank batching patience (PANGULU_AMD_GATHER_MIN_BATCH / _MAX_US / _QUIET_US)
    size_t gather_min_batch = 256;
    double gather_max_s = 600e-6, gather_quiet_s = 60e-6, t_gather = 0, t_idle = 0, t_work = 0;
    // ... but only while blocks ARE arriving: the receive thread counts arrivals and pulls in flight; a drain that
    // follows a dispatch with neither (the subtree phase of the factorisation exchanges nothing) goes out at once
    std::atomic<u64> arrivals{0};
    std::atomic<int> recv_inflight{0};
    u64 arrivals_seen = 0;
    double t_last_progress = 0, t_last_arrival = 0, stall_limit_s = 120; // PANGULU_AMD_STALL_S
    // multi-rank without draining the device after every batch: receive slots are handed back, and finished blocks
    // announced, once a marker recorded behind the kernels that use / produce them has completed
    bool use_markers = false;
    Marker *last_marker = nullptr; // marker behind the most recent platform call (created on demand)
    int last_marker_id = -1;
    std::deque<std::unique_ptr<Marker>> markers; // all markers of this factorisation (the transport's sender thread keeps pointers)
    struct Retired
    {
        slot_t *s;
        Marker *marker;
    };
    std::deque<Retired> retired;
    double t_platform = 0, t_sched = 0;
    u64 batches = 0;
    u64 deferred = 0; // queues a look-ahead call left alone (pangulu_amd_info_t::deferred_queues)
    bool multi;
    // Runs on a device: the platform calls (descriptor building and launches) cost about as much host time
    // as the scheduling itself, and nothing the scheduler does next depends on their return -- the back-end's streams
    // keep program order.  A launcher thread makes the calls in batch order while this thread releases successors and
    // drains the next batch.  Round 3: multi-rank runs too (one rank through the multi-rank loop took 70 ms against 44 ms
    // for the bench matrix, and every rank of an N-GPU run paid that): what the scheduler needs from the device there are
    // MARKERS -- "the blocks finished so far may be sent", "this receive slot is free again" -- and a marker is now a
    // handle (Marker) the scheduler hands out at once while the launcher records the event behind it when it gets to that
    // point of the launch order; the transport's sender thread and poll_retired() look at it only once it is there.  On by default (PANGULU_AMD_ASYNC_LAUNCH=0 turns it off): in round 1 the one-thread loop still
    // stayed ahead of the device; with the faster GETRF the leaf levels of the bench matrix became host-bound (the
    // device sat empty 10-19 ms of a 57 ms factorisation, mostly before the densify launch that follows a level's
    // GETRF) and the launcher thread brought 59.7-70.7 ms down to 52.0-52.4 ms on the same box.
    bool async_launch = false;
    std::thread launcher;
    std::mutex lq_mutex;
    std::condition_variable lq_cv, lq_idle_cv;
    struct LItem
    {
        std::vector<task_t> tasks;
        Marker *marker = nullptr; // non-null: record a platform marker here instead of running tasks
        int marker_id = -1;       // (recording a rank log: the marker's number)
        u32 need = 0;             // (recording: arrivals handled when the batch was made)
    };
    // Recording of this rank's log for later replays (Solver::RankLog): the launcher thread appends an entry behind every platform
    // call, the scheduler thread the sends, the receive thread the arrivals.
    bool rec_multi = false, rec_gates_ok = true;
    int gate_marker_id = -1;
    std::atomic<u32> arrivals_handled{0};
    std::unordered_set<const slot_t *> rec_used_slots; // receive slots taken so far (a second use makes the log useless: slot addresses are in the descriptors)
    void log_entry(int marker_id, u32 need)
    {
        Solver::RankLog::Entry e;
        e.op_end = plat.schedule(6, &S);
        e.marker = marker_id;
        e.need = need;
        S.rank_log.entries.push_back(e);
    }
    std::deque<LItem> lq;
    std::vector<std::vector<task_t>> lq_free;
    bool forced_multi = false; // PANGULU_AMD_FORCE_MULTI_LOOP=1: one rank through the multi-rank loop (measurement aid)
    bool lq_stop = false, lq_busy = false;

    bool dry_run = false; // the scheduler's dry run at pangulu_init: platform calls record, nothing is launched (no launcher thread)
    explicit Sched(Solver &s, bool dry = false) : S(s), P(s.pat), plat(active_platform()), comm(world()), sent_flag((size_t)s.nproc, 0), multi(s.nproc > 1), dry_run(dry)
    {
        if (const char *e = getenv("PANGULU_AMD_FORCE_MULTI_LOOP"))
            forced_multi = atoi(e) != 0 && s.nproc == 1;
        multi = multi || forced_multi;
        if (const char *e = getenv("PANGULU_AMD_LOOKAHEAD_MAX_GETRF"))
            lookahead_max_getrf = (size_t)atol(e);
        if (const char *e = getenv("PANGULU_AMD_PANEL_LOOKAHEAD"))
            panel_lookahead_on = atoi(e) != 0;
        if (const char *e = getenv("PANGULU_AMD_LOOKAHEAD_MIN_QUEUE"))
            lookahead_min_queue = (size_t)std::max(1L, atol(e));
        if (const char *e = getenv("PANGULU_AMD_LOOKAHEAD_DEFER_FROM"))
            lookahead_defer_from = (size_t)std::max(0L, atol(e));
        if (const char *e = getenv("PANGULU_AMD_LOOKAHEAD_MIN_TASKS"))
            lookahead_min_tasks = (size_t)std::max(0L, atol(e));
        if (const char *e = getenv("PANGULU_AMD_PANEL_FIRST"))
            panel_first_on = atoi(e) != 0;
        if (const char *e = getenv("PANGULU_AMD_GATHER_MIN_BATCH"))
            gather_min_batch = (size_t)atol(e);
        if (const char *e = getenv("PANGULU_AMD_GATHER_MAX_US"))
            gather_max_s = 1e-6 * atof(e);
        if (const char *e = getenv("PANGULU_AMD_GATHER_QUIET_US"))
            gather_quiet_s = 1e-6 * atof(e);
        if (const char *e = getenv("PANGULU_AMD_STALL_S"))
            stall_limit_s = atof(e);
        use_markers = multi && !plat.host_memory && plat.marker_record && plat.marker_done && plat.marker_wait && !getenv("PANGULU_AMD_SYNC_EVERY_BATCH");
        const char *al = getenv("PANGULU_AMD_ASYNC_LAUNCH");
        async_launch = !plat.host_memory && !(al && atoi(al) == 0) && (!multi || use_markers) && !dry_run;
        if (async_launch)
            launcher = std::thread([this]()
                                   { launcher_loop(); });
    }

    ~Sched()
    {
        if (launcher.joinable())
        {
            {
                std::lock_guard<std::mutex> g(lq_mutex);
                lq_stop = true;
            }
            lq_cv.notify_all();
            launcher.join();
        }
    }

    void launcher_loop()
    {
        std::unique_lock<std::mutex> lk(lq_mutex);
        while (true)
        {
            lq_cv.wait(lk, [this]()
                       { return lq_stop || !lq.empty(); });
            if (lq.empty())
                return; // (stop requested and nothing left)
            LItem item = std::move(lq.front());
            lq.pop_front();
            lq_busy = true;
            lk.unlock();
            double dt = 0;
            if (item.marker)
            {
                item.marker->ev.store(plat.marker_record(), std::memory_order_release);
                if (rec_multi)
                    log_entry(item.marker_id, 0);
            }
            else
            {
                double t0 = wall_seconds();
                plat.hybrid_batched((pangulu_inblock_idx)S.nb, item.tasks.size(), item.tasks.data());
                dt = wall_seconds() - t0;
                if (rec_multi)
                    log_entry(-1, item.need);
            }
            lk.lock();
            t_platform += dt;
            lq_busy = false;
            if (!item.marker)
            {
                item.tasks.clear();
                lq_free.push_back(std::move(item.tasks));
            }
            if (lq.empty())
                lq_idle_cv.notify_all();
        }
    }

    // every platform call handed to the launcher has returned (their kernels are queued on the device)
    void drain_launcher()
    {
        if (!async_launch)
            return;
        std::unique_lock<std::mutex> lk(lq_mutex);
        lq_idle_cv.wait(lk, [this]()
                        { return lq.empty() && !lq_busy; });
    }

    // ---- task creation -------------------------------------------------------------------------------
    void push_panel(u32 row, u32 col, u32 level, int kernel, slot_t *dst, slot_t *op1)
    {
        task_t t;
        memset(&t, 0, sizeof(t));
        t.row = row;
        t.col = col;
        t.kernel_id = (pangulu_int16_t)kernel;
        t.task_level = level;
        t.compare_flag = level;
        t.opdst = dst;
        t.op1 = op1;
        t.op2 = nullptr;
        S.heap.push(t);
    }

    static u32 tile_index(const slot_t *dst)
    {
        // canonical owned-slot index of a destination tile (diagonal tiles: their lower half)
        if (dst->brow_pos == dst->bcol_pos && dst->is_upper)
            return (u32)dst->related_block->slot_idx;
        return (u32)dst->slot_idx;
    }

    // Queue C(row,col) -= op1 * op2 on its destination tile, then run the check the reference performs when
    // the heap copy of an SSSSM task is popped (src/pangulu_numeric.c:602-650).  Caller holds info_mutex and
    // has already decremented the destination's counter.
    void queue_update(u32 row, u32 col, u32 level, slot_t *dst, slot_t *op1, slot_t *op2, u64 dst_bidx)
    {
        task_t t;
        memset(&t, 0, sizeof(t));
        t.row = row;
        t.col = col;
        t.kernel_id = PANGULU_TASK_SSSSM;
        t.task_level = level;
        t.compare_flag = level;
        t.opdst = dst;
        t.op1 = op1;
        t.op2 = op2;
        u32 ti = tile_index(dst);
        auto &q = S.pending[ti];
        if (q.empty())
        {
            // one entry per tile (a deferred tile whose queue flush (1) or panel-first had emptied was pushed a second time,
            // both entries were kept and counted while shallow, and the list grew)
            if (in_dirty.size() < S.pending.size())
                in_dirty.resize(S.pending.size(), 0);
            if (!in_dirty[ti])
            {
                in_dirty[ti] = 1;
                S.pending_dirty.push_back(ti);
            }
            if (lookahead_min_queue > 1)
            {
                if (pending_bidx.size() < S.pending.size())
                    pending_bidx.resize(S.pending.size(), ~0ull);
                pending_bidx[ti] = dst_bidx;
            }
        }
        q.push_back(t);
        S.pending_total++;

        if (row == col)
        {
            if (S.remain_diag[row] == 1)
            {
                S.remain_diag[row]--;
                push_panel(row, col, row, PANGULU_TASK_GETRF, dst, nullptr);
            }
        }
        else if (S.remain[dst_bidx] == 1)
        {
            u32 lv = std::min(row, col);
            if (row < col)
            {
                slot_t *dl = S.diag_lower[lv];
                if (dl && dl->data_status == PANGULU_DATA_READY)
                {
                    S.remain[dst_bidx]--;
                    push_panel(row, col, lv, PANGULU_TASK_GESSM, dst, dl);
                }
            }
            else
            {
                slot_t *du = S.diag_upper[lv];
                if (du && du->data_status == PANGULU_DATA_READY)
                {
                    S.remain[dst_bidx]--;
                    push_panel(row, col, lv, PANGULU_TASK_TSTRF, dst, du);
                }
            }
        }
    }

    void send_once(slot_t *s, int target)
    {
        if (sent_flag[(size_t)target])
            return;
        sent_flag[(size_t)target] = 1;
        BlockHeader h;
        memset(&h, 0, sizeof(h));
        h.nnz = s->columnpointer[S.nb];
        h.brow = s->brow_pos;
        h.bcol = s->bcol_pos;
        h.is_upper = (u32)s->is_upper;
        size_t bytes = record_bytes(S.nb, h.nnz, s->brow_pos > s->bcol_pos);
        h.bytes_lo = (u32)bytes;
        if (rec_multi)
            S.rank_log.sends.push_back(Solver::RankLog::Send{gate_marker_id, s, h, target});
        comm->isend_block(s, h, target);
    }

    // ---- successor release (all under info_mutex) ---------------------------------------------------------
    // a factorised diagonal's U half is available (local GETRF or arrival): release the TSTRFs below it
    void release_after_diag_upper(u32 level, slot_t *upper, bool do_sends)
    {
        if (do_sends)
            std::fill(sent_flag.begin(), sent_flag.end(), 0);
        for (u64 b = P.first_after_diag[level]; b < P.colptr[level + 1]; b++)
        {
            u32 brow = P.rowidx[b];
            int target = S.owner(brow, level);
            if (target == S.rank)
            {
                if (S.remain[b] == 1)
                {
                    S.remain[b]--;
                    push_panel(brow, level, level, PANGULU_TASK_TSTRF, S.slot_of[b], upper);
                }
            }
            else if (do_sends)
            {
                send_once(upper, target);
            }
        }
    }

    void release_after_diag_lower(u32 level, slot_t *lower, bool do_sends)
    {
        if (do_sends)
            std::fill(sent_flag.begin(), sent_flag.end(), 0);
        for (u64 r = P.first_after_diag_csr[level]; r < P.rowptr[level + 1]; r++)
        {
            u32 bcol = P.colidx[r];
            u64 b = P.csr_to_csc[r];
            int target = S.owner(level, bcol);
            if (target == S.rank)
            {
                if (S.remain[b] == 1)
                {
                    S.remain[b]--;
                    push_panel(level, bcol, level, PANGULU_TASK_GESSM, S.slot_of[b], lower);
                }
            }
            else if (do_sends)
            {
                send_once(lower, target);
            }
        }
    }

    // a finished L block L(brow, level) is available: queue C(brow, bcol) -= L(brow,level) * U(level,bcol) for
    // every bcol > level whose U(level,bcol) is already final, forward the block to the other owners in its
    // process row and to the owner of diagonal (brow,brow) (src/pangulu_numeric.c:436-518, 84-138)
    void release_after_L(slot_t *L, u32 brow, u32 level, bool do_sends)
    {
        if (do_sends)
            std::fill(sent_flag.begin(), sent_flag.end(), 0);
        u64 rb = P.rowptr[brow], re = P.rowptr[brow + 1];
        // position of (brow, level) in block row brow
        u64 pos = (u64)(std::lower_bound(P.colidx.begin() + (i64)rb, P.colidx.begin() + (i64)re, level) - P.colidx.begin());
        u64 ub = P.first_after_diag_csr[level], ue = P.rowptr[level + 1]; // U blocks of block row `level`
        slot_t *u_for_diag = nullptr;
        u64 u = ub;
        // ranks that run an update with this block (pg_preprocess.cpp, forwarding rule); all ones = the reference's rule
        const u64 wanted = S.consumers.empty() ? ~0ull : S.consumers[P.csr_to_csc[pos]];
        for (u64 r = pos + 1; r < re; r++)
        {
            u32 bcol = P.colidx[r];
            int target = S.owner(brow, bcol);
            if (target == S.rank)
            {
                while (u < ue && P.colidx[u] < bcol)
                    u++;
                if (u < ue && P.colidx[u] == bcol)
                {
                    slot_t *U = S.slot_of[P.csr_to_csc[u]];
                    if (U && U->data_status == PANGULU_DATA_READY)
                    {
                        u64 d = P.csr_to_csc[r];
                        S.remain[d]--;
                        queue_update(brow, bcol, level, S.slot_of[d], L, U, d);
                    }
                }
            }
            else if (do_sends && ((wanted >> target) & 1ull))
            {
                send_once(L, target);
            }
        }
        // diagonal destination (brow, brow) needs U(level, brow)
        {
            u64 hit = (u64)(std::lower_bound(P.colidx.begin() + (i64)ub, P.colidx.begin() + (i64)ue, brow) - P.colidx.begin());
            if (hit < ue && P.colidx[hit] == brow)
                u_for_diag = S.slot_of[P.csr_to_csc[hit]];
            int target = S.owner(brow, brow);
            if (target == S.rank)
            {
                if (u_for_diag && u_for_diag->data_status == PANGULU_DATA_READY)
                {
                    S.remain_diag[brow]--;
                    queue_update(brow, brow, level, S.diag_lower[brow], L, u_for_diag, ~0ull);
                }
            }
            else if (do_sends && ((wanted >> target) & 1ull))
            {
                send_once(L, target);
            }
        }
    }

    // mirror image for a finished U block U(level, bcol) (src/pangulu_numeric.c:519-601, 139-193)
    void release_after_U(slot_t *U, u32 level, u32 bcol, bool do_sends)
    {
        if (do_sends)
            std::fill(sent_flag.begin(), sent_flag.end(), 0);
        u64 cb = P.colptr[bcol], ce = P.colptr[bcol + 1];
        u64 pos = (u64)(std::lower_bound(P.rowidx.begin() + (i64)cb, P.rowidx.begin() + (i64)ce, level) - P.rowidx.begin());
        u64 lb = P.first_after_diag[level], le = P.colptr[level + 1]; // L blocks of block column `level`
        u64 l = lb;
        const u64 wanted = S.consumers.empty() ? ~0ull : S.consumers[pos];
        for (u64 c = pos + 1; c < ce; c++)
        {
            u32 brow = P.rowidx[c];
            int target = S.owner(brow, bcol);
            if (target == S.rank)
            {
                while (l < le && P.rowidx[l] < brow)
                    l++;
                if (l < le && P.rowidx[l] == brow)
                {
                    slot_t *L = S.slot_of[l];
                    if (L && L->data_status == PANGULU_DATA_READY)
                    {
                        S.remain[c]--;
                        queue_update(brow, bcol, level, S.slot_of[c], L, U, c);
                    }
                }
            }
            else if (do_sends && ((wanted >> target) & 1ull))
            {
                send_once(U, target);
            }
        }
        {
            slot_t *l_for_diag = nullptr;
            u64 hit = (u64)(std::lower_bound(P.rowidx.begin() + (i64)lb, P.rowidx.begin() + (i64)le, bcol) - P.rowidx.begin());
            if (hit < le && P.rowidx[hit] == bcol)
                l_for_diag = S.slot_of[hit];
            int target = S.owner(bcol, bcol);
            if (target == S.rank)
            {
                if (l_for_diag && l_for_diag->data_status == PANGULU_DATA_READY)
                {
                    S.remain_diag[bcol]--;
                    queue_update(bcol, bcol, level, S.diag_lower[bcol], l_for_diag, U, ~0ull);
                }
            }
            else if (do_sends && ((wanted >> target) & 1ull))
            {
                send_once(U, target);
            }
        }
    }

    // a remote diagonal is no longer needed once all my TSTRF/GESSM tasks of that level ran
    void consume_remote_diag(u32 level)
    {
        if (S.owner(level, level) == S.rank)
            return;
        if (--S.remain_diag[level] == 0)
        {
            retire(S.diag_upper[level]);
            S.diag_upper[level] = nullptr;
            retire(S.diag_lower[level]);
            S.diag_lower[level] = nullptr;
        }
    }

    // ---- execution -------------------------------------------------------------------------------------
    u64 sample_seen[5] = {0, 0, 0, 0, 0};
    void run_platform_batch(std::vector<task_t> &tasks)
    {
        if (tasks.empty())
            return;
        if (g_task_sample_stride > 1)
        {
            // cpu_baseline leg of bench.py (checker's build): every stride-th task of each class runs, the others are only
            // released (the caller walks `tasks` afterwards: it stays whole) -- the time a kernel takes depends on the
            // patterns, not on the values the skipped tasks would have left behind
            static thread_local std::vector<task_t> kept;
            kept.clear();
            for (const task_t &t : tasks)
                if (sample_seen[t.kernel_id]++ % (u64)g_task_sample_stride == 0)
                {
                    S.info.sampled_flop += task_structural_flop(S.nb, t);
                    S.info.sampled_tasks++;
                    kept.push_back(t);
                }
            if (kept.empty())
                return;
            batches++;
            double t0 = wall_seconds();
            plat.hybrid_batched((pangulu_inblock_idx)S.nb, kept.size(), kept.data());
            t_platform += wall_seconds() - t0;
            last_marker = nullptr;
            return;
        }
        batches++;
        if (async_launch)
        {
            std::vector<task_t> copy;
            {
                std::lock_guard<std::mutex> g(lq_mutex);
                if (!lq_free.empty())
                {
                    copy = std::move(lq_free.back());
                    lq_free.pop_back();
                }
            }
            copy.assign(tasks.begin(), tasks.end());
            {
                std::lock_guard<std::mutex> g(lq_mutex);
                LItem item;
                item.tasks = std::move(copy);
                item.need = arrivals_handled.load(std::memory_order_acquire); // (everything that released these tasks has been handled)
                lq.push_back(std::move(item));
            }
            lq_cv.notify_one();
            last_marker = nullptr;
            return;
        }
        const u32 need = arrivals_handled.load(std::memory_order_acquire);
        double t0 = wall_seconds();
        plat.hybrid_batched((pangulu_inblock_idx)S.nb, tasks.size(), tasks.data());
        t_platform += wall_seconds() - t0;
        if (rec_multi)
            log_entry(-1, need);
        last_marker = nullptr;
    }

    // a marker behind every platform call issued so far (in the scheduler's order); with the launcher thread the event
    // behind it is recorded when the launcher gets there
    Marker *current_marker()
    {
        if (last_marker)
            return last_marker;
        markers.emplace_back(new Marker());
        last_marker = markers.back().get();
        last_marker_id = (int)markers.size() - 1;
        if (async_launch)
        {
            {
                std::lock_guard<std::mutex> g(lq_mutex);
                LItem item;
                item.marker = last_marker;
                item.marker_id = last_marker_id;
                lq.push_back(std::move(item));
            }
            lq_cv.notify_one();
        }
        else
        {
            last_marker->ev.store(plat.marker_record(), std::memory_order_release);
            if (rec_multi)
                log_entry(last_marker_id, 0);
        }
        return last_marker;
    }

    // the device has finished everything the scheduler has issued (launcher included)
    void drain_device()
    {
        drain_launcher();
        plat.synchronize();
    }

    // a receive slot whose last consumer has been queued on the device: back to its bin once that work is done
    void retire(slot_t *s)
    {
        if (!s)
            return;
        if (use_markers)
            retired.push_back(Retired{s, current_marker()});
        else
            S.storage.recycle(s);
    }

    void poll_retired()
    {
        while (!retired.empty())
        {
            void *ev = retired.front().marker->ev.load(std::memory_order_acquire);
            if (!ev || !plat.marker_done(ev))
                break;
            S.storage.recycle(retired.front().s);
            retired.pop_front();
        }
    }

    // every update the destination will ever receive is in its queue (only its own panel task is left on its counter): waiting longer
    // cannot make the queue any deeper
    std::vector<unsigned char> in_dirty; // tile has an entry in S.pending_dirty
    void clear_dirty_list()
    {
        for (u32 tile : S.pending_dirty)
            if (tile < in_dirty.size())
                in_dirty[tile] = 0;
        S.pending_dirty.clear();
    }
    std::vector<u64> pending_bidx; // destination block of a tile's queue (~0: a diagonal block, counted per block row)
    bool queue_is_complete(u32 tile) const
    {
        const auto &q = S.pending[tile];
        if (q.empty() || tile >= pending_bidx.size())
            return false;
        const u64 b = pending_bidx[tile];
        return b == ~0ull ? S.remain_diag[q.front().row] <= 1 : S.remain[b] <= 1;
    }

    // move the queued updates of the given tiles into ssssm_batch (grouped by tile, queue order kept)
    void take_pending(u32 tile)
    {
        auto &q = S.pending[tile];
        if (q.empty())
            return;
        ssssm_batch.insert(ssssm_batch.end(), q.begin(), q.end());
        S.pending_total -= q.size();
        q.clear();
    }

    void run_updates_and_release_operands()
    {
        if (ssssm_batch.empty())
            return;
        run_platform_batch(ssssm_batch);
        release_update_operands();
    }

    void release_update_operands()
    {
        if (multi)
        {
            // operands received from other ranks are dropped once their last consumer has run
            // (src/pangulu_numeric.c:226-251); the device must be done with them before the slot is reused
            bool synced = use_markers;
            std::lock_guard<std::mutex> g(S.info_mutex);
            for (auto &t : ssssm_batch)
            {
                slot_t *ops[2] = {t.op1, t.op2};
                for (slot_t *op : ops)
                {
                    if (S.owner(op->brow_pos, op->bcol_pos) == S.rank)
                        continue;
                    u64 b = P.find(op->brow_pos, op->bcol_pos);
                    if (--S.remain[b] == 0)
                    {
                        if (!synced)
                        {
                            drain_device();
                            synced = true;
                        }
                        retire(S.slot_of[b]);
                        S.slot_of[b] = nullptr;
                    }
                }
            }
        }
        ssssm_batch.clear();
    }

    void rebuild_dirty_list()
    {
        size_t w = 0;
        for (size_t i = 0; i < S.pending_dirty.size(); i++)
            if (!S.pending[S.pending_dirty[i]].empty())
                S.pending_dirty[w++] = S.pending_dirty[i];
            else
                in_dirty[S.pending_dirty[i]] = 0;
        S.pending_dirty.resize(w);
    }

    double t_sec[5] = {0, 0, 0, 0, 0}; // (trace) work_batched: queued updates, panel call, send gate, successor release
    void work_batched()
    {
        double t_s = wall_seconds();
#define SEC(i_)                            \
    {                                      \
        const double now_ = wall_seconds(); \
        t_sec[i_] += now_ - t_s;           \
        t_s = now_;                        \
    }
        // A few diagonal factorisations on their own leave the device almost idle (one workgroup each): every update
        // queued anywhere else goes into the same call (2), the back-end runs the two kinds side by side ("look-ahead")
        bool all_getrf = true;
        for (auto &t : batch)
            all_getrf = all_getrf && t.kernel_id == PANGULU_TASK_GETRF;
        const bool lookahead_shape = all_getrf && batch.size() <= lookahead_max_getrf;
        // the panel tiles of the levels being factorised: the blocks below and right of each diagonal block
        auto for_panel_tiles = [&](auto &&fn)
        {
            for (auto &t : batch)
            {
                const u32 k = t.task_level;
                for (u64 b = P.first_after_diag[k]; b < P.colptr[k + 1]; b++)
                    if (S.slot_of[b] && S.owner(P.rowidx[b], k) == S.rank)
                        fn(tile_index(S.slot_of[b]));
                for (u64 r = P.first_after_diag_csr[k]; r < P.rowptr[k + 1]; r++)
                {
                    const u64 b = P.csr_to_csc[r];
                    if (S.slot_of[b] && S.owner(k, P.colidx[r]) == S.rank)
                        fn(tile_index(S.slot_of[b]));
                }
            }
        };
        // (1) every update queued on a tile of this drain runs first, as one batch.  With a look-ahead call coming, the
        // updates queued on the PANEL tiles of these levels go here too (round 3): they are what the panel solves of the next
        // drain wait for, and keeping them out of the big look-ahead call lets the back-end run that one in the background,
        // beside the factorisations AND the solves that follow (pangulu_platform.h, PANGULU_HIP_OPT_BACKGROUND_UPDATES).
        {
            std::lock_guard<std::mutex> g(S.info_mutex);
            for (auto &t : batch)
                take_pending(tile_index(t.opdst));
            if (lookahead_shape && panel_first_on && S.pending_total != 0)
                for_panel_tiles([&](u32 tile)
                                { take_pending(tile); });
            if (S.pending_dirty.size() > 4096)
                rebuild_dirty_list();
        }
        run_updates_and_release_operands();
        SEC(0)
        // (2) the panel tasks themselves
        const bool lookahead = lookahead_shape && S.pending_total != 0;
        // A larger batch of diagonal factorisations: the updates queued on the PANEL tiles of these levels (the blocks
        // below and right of each diagonal block) do not depend on the factorisations either -- every one of them was
        // queued before the diagonal block became ready (same descendants, src/pangulu_numeric.c:436-601) -- and the
        // panel solves are what follows.  They run beside the GETRFs instead of between them and the solves; unlike the
        // flush above this creates no extra pass over any tile.
        const bool panel_lookahead = all_getrf && !lookahead_shape && panel_lookahead_on && S.pending_total != 0;
        if (lookahead || panel_lookahead)
        {
            {
                std::lock_guard<std::mutex> g(S.info_mutex);
                if (lookahead)
                {
                    if (lookahead_min_queue <= 1 || S.pending_total < lookahead_defer_from)
                    {
                        for (u32 tile : S.pending_dirty)
                            take_pending(tile);
                        clear_dirty_list();
                    }
                    else
                    {
                        // shallow queues wait: a destination tile is read and written once per launch it is part of, whatever the
                        // number of updates in its queue (its own panel task takes whatever is left, flush (1) above)
                        size_t w = 0;
                        for (size_t i = 0; i < S.pending_dirty.size(); i++)
                        {
                            const u32 tile = S.pending_dirty[i];
                            if (S.pending[tile].size() >= lookahead_min_queue || queue_is_complete(tile))
                                take_pending(tile);
                            else if (!S.pending[tile].empty())
                            {
                                S.pending_dirty[w++] = tile;
                                deferred++;
                                continue;
                            }
                            in_dirty[tile] = 0;
                        }
                        S.pending_dirty.resize(w);
                        // ... unless the call would then be too small to keep the device busy beside the factorisations and the
                        // solves that follow them: shallow queues after all, oldest first, up to lookahead_min_tasks updates
                        if (ssssm_batch.size() < lookahead_min_tasks)
                        {
                            size_t i = 0;
                            for (; i < S.pending_dirty.size() && ssssm_batch.size() < lookahead_min_tasks; i++)
                            {
                                take_pending(S.pending_dirty[i]);
                                in_dirty[S.pending_dirty[i]] = 0;
                            }
                            S.pending_dirty.erase(S.pending_dirty.begin(), S.pending_dirty.begin() + (long)i);
                        }
                    }
                }
                else
                    for_panel_tiles([&](u32 tile)
                                    { take_pending(tile); });
            }
            combined.assign(batch.begin(), batch.end());
            combined.insert(combined.end(), ssssm_batch.begin(), ssssm_batch.end());
            run_platform_batch(combined);
            release_update_operands();
        }
        else
            run_platform_batch(batch);
        SEC(1)
        if (multi)
        {
            // finished blocks are about to be announced: either the transport holds the announcements back until a
            // marker behind this batch has completed, or the device is drained here
            if (use_markers && comm->set_send_gate(current_marker()))
                gate_marker_id = last_marker_id;
            else
            {
                drain_device();
                rec_gates_ok = false; // (a transport that copies at post time: its sends cannot be replayed behind a marker)
            }
        }
        SEC(2)
        // (3) successor release
        std::lock_guard<std::mutex> g(S.info_mutex);
        for (auto &t : batch)
        {
            u32 level = t.task_level;
            if (t.kernel_id == PANGULU_TASK_GETRF)
            {
                slot_t *up = t.opdst->is_upper ? t.opdst : t.opdst->related_block;
                slot_t *lo = up->related_block;
                up->data_status = PANGULU_DATA_READY;
                lo->data_status = PANGULU_DATA_READY;
                release_after_diag_upper(level, up, multi);
                release_after_diag_lower(level, lo, multi);
            }
            else if (t.kernel_id == PANGULU_TASK_TSTRF)
            {
                t.opdst->data_status = PANGULU_DATA_READY;
                consume_remote_diag(level);
                release_after_L(t.opdst, t.row, level, multi);
            }
            else if (t.kernel_id == PANGULU_TASK_GESSM)
            {
                t.opdst->data_status = PANGULU_DATA_READY;
                consume_remote_diag(level);
                release_after_U(t.opdst, level, t.col, multi);
            }
        }
        SEC(3)
#undef SEC
    }

    // nothing runnable: use the time for queued updates (src/pangulu_task.c:93-177)
    bool idle_flush()
    {
        {
            std::lock_guard<std::mutex> g(S.info_mutex);
            if (S.pending_total == 0)
                return false;
            for (u32 tile : S.pending_dirty)
                take_pending(tile);
            clear_dirty_list();
        }
        run_updates_and_release_operands();
        return true;
    }

    void compute_loop()
    {
        if (!plat.host_memory && plat.set_default_device)
        {
            int ndev = 1;
            plat.get_device_num(&ndev);
            (void)ndev; // the device was selected at init; HIP's current device is per thread
        }
        static const bool trace = getenv("PANGULU_AMD_TRACE") != nullptr;
        double t_trace = wall_seconds();
        const double t_loop_begin = t_trace;
        while (S.rank_remain_task != 0)
        {
            if (trace && wall_seconds() - t_trace > 2.0)
            {
                t_trace = wall_seconds();
                fprintf(stderr, "[pangulu_amd trace] rank %d: %lld panel tasks, %lld receives outstanding, %llu updates queued, %llu batches so far\n",
                        S.rank, (long long)S.rank_remain_task, (long long)S.rank_remain_recv, (unsigned long long)S.pending_total,
                        (unsigned long long)batches);
            }
            if (use_markers)
                poll_retired();
            batch.clear();
            task_t t;
            while (S.heap.pop(t))
                batch.push_back(t);
            const bool expecting = multi && (recv_inflight.load(std::memory_order_relaxed) > 0 || arrivals.load(std::memory_order_relaxed) != arrivals_seen);
            if (expecting && (i64)batch.size() < S.rank_remain_task && batch.size() < gather_min_batch)
            {
                // Blocks from other ranks arrive one by one, and every launch costs its latency whatever it carries (a
                // GETRF launch takes as long for 1 block as for 256): dispatching each arrival on its own degenerates
                // into thousands of single-task launches.  Keep collecting while arrivals keep coming, for a bounded time.
                const double t_begin = wall_seconds();
                double t_last = t_begin;
                size_t seen = batch.size();
                while (true)
                {
                    const double now = wall_seconds();
                    if (now - t_begin > gather_max_s || now - t_last > gather_quiet_s)
                        break;
                    while (S.heap.pop(t))
                        batch.push_back(t);
                    if (use_markers)
                        poll_retired();
                    if (batch.size() != seen)
                    {
                        seen = batch.size();
                        t_last = now;
                        if (seen >= gather_min_batch || (i64)seen >= S.rank_remain_task)
                            break;
                    }
                    else
                        usleep(5);
                }
                t_gather += wall_seconds() - t_begin;
            }
            if (multi && !batch.empty())
                arrivals_seen = arrivals.load(std::memory_order_relaxed);
            if (batch.empty())
            {
                if (!idle_flush())
                {
                    if (!multi)
                        fatal("scheduler stalled with %lld panel tasks left and nothing runnable", (long long)S.rank_remain_task);
                    // waiting for blocks of other ranks: fail loudly instead of hanging if nothing moves for a long time
                    const double now = wall_seconds();
                    if (t_last_progress == 0)
                        t_last_progress = now;
                    if (now - t_last_progress > stall_limit_s)
                        fatal("rank %d: no runnable task for %.0f s with %lld panel tasks and %lld block receives outstanding "
                              "(%llu updates queued): a block that was announced never arrived, or a dependency cycle",
                              S.rank, now - t_last_progress, (long long)S.rank_remain_task, (long long)S.rank_remain_recv,
                              (unsigned long long)S.pending_total);
                    const double t_idle0 = wall_seconds();
                    usleep(20);
                    t_idle += wall_seconds() - t_idle0;
                }
                else
                    t_last_progress = 0;
                continue;
            }
            t_last_progress = 0;
            S.rank_remain_task -= (i64)batch.size();
            const double t_work0 = wall_seconds();
            work_batched();
            t_work += wall_seconds() - t_work0;
        }
        if (trace)
            fprintf(stderr, "[pangulu_amd trace] rank %d: scheduler loop %.1f ms: waiting for arrivals (nothing runnable) %.1f, gathering small batches %.1f, "
                            "dispatch + release %.1f (platform calls %.1f; queued updates %.1f, panel call %.1f, send gate %.1f, successor release %.1f), %llu batches\n",
                    S.rank, 1e3 * (wall_seconds() - t_loop_begin), 1e3 * t_idle, 1e3 * t_gather, 1e3 * t_work, 1e3 * t_platform,
                    1e3 * t_sec[0], 1e3 * t_sec[1], 1e3 * t_sec[2], 1e3 * t_sec[3], (unsigned long long)batches);
        // updates into tiles are always flushed by the tile's own panel task, so nothing can be left
        if (S.pending_total != 0)
            fatal("scheduler finished with %llu queued updates", (unsigned long long)S.pending_total);
        t_sched = wall_seconds() - t_loop_begin;
        drain_launcher();
    }

    // ---- arrivals (receive thread) -------------------------------------------------------------------------
    void handle_arrival(slot_t *s, const BlockHeader &h)
    {
        std::lock_guard<std::mutex> g(S.info_mutex);
        s->brow_pos = h.brow;
        s->bcol_pos = h.bcol;
        s->is_upper = (i32)h.is_upper;
        s->data_status = PANGULU_DATA_READY;
        struct Handled // (counted when the successors have been released, whichever way this function returns)
        {
            std::atomic<u32> &c;
            ~Handled() { c.fetch_add(1, std::memory_order_release); }
        } handled{arrivals_handled};
        if (rec_multi)
            S.rank_log.arrivals.push_back(Solver::RankLog::Arrival{h.brow, h.bcol, h.is_upper, s, h.nnz});
        if (h.brow == h.bcol)
        {
            u32 level = h.brow;
            if (h.is_upper)
            {
                S.diag_upper[level] = s;
                s->related_block = S.diag_lower[level];
                if (S.diag_lower[level])
                    S.diag_lower[level]->related_block = s;
                release_after_diag_upper(level, s, false);
            }
            else
            {
                S.diag_lower[level] = s;
                s->related_block = S.diag_upper[level];
                if (S.diag_upper[level])
                    S.diag_upper[level]->related_block = s;
                release_after_diag_lower(level, s, false);
            }
            return;
        }
        u64 b = P.find(h.brow, h.bcol);
        if (b == ~0ull)
            fatal("received block (%u,%u) that is not in the block pattern", h.brow, h.bcol);
        S.slot_of[b] = s;
        if (S.remain[b] == 0)
        {
            // forwarded by the owner's rule, but none of my updates pairs it with an existing partner block
            S.slot_of[b] = nullptr;
            S.storage.recycle(s);
            return;
        }
        if (h.brow > h.bcol)
            release_after_L(s, h.brow, h.bcol, false);
        else
            release_after_U(s, h.brow, h.bcol, false);
    }

    void receive_loop()
    {
        struct Begun
        {
            slot_t *s;
            BlockHeader h;
        };
        std::vector<Begun> begun; // receives started but not completed (pipelined transports)
        auto finish = [&]()
        {
            if (begun.empty())
                return;
            comm->recv_blocks_finish();
            for (auto &p : begun)
            {
                S.rank_remain_recv--;
                handle_arrival(p.s, p.h);
                arrivals.fetch_add(1, std::memory_order_relaxed);
            }
            begun.clear();
            recv_inflight.store(0, std::memory_order_relaxed);
        };
        while (S.rank_remain_recv != 0)
        {
            BlockHeader h;
            int src = -1;
            if ((i64)begun.size() == S.rank_remain_recv || !comm->probe_block(h, src))
            {
                // nothing more waiting right now: complete what was started, then idle
                if (!begun.empty())
                    finish();
                else
                {
                    const double now = wall_seconds();
                    if (t_last_arrival == 0)
                        t_last_arrival = now;
                    if (now - t_last_arrival > stall_limit_s)
                        fatal("rank %d: no block arrived for %.0f s with %lld receives outstanding: a peer stopped sending (see its output)",
                              S.rank, now - t_last_arrival, (long long)S.rank_remain_recv);
                    usleep(10);
                }
                continue;
            }
            t_last_arrival = 0;
            size_t bytes = h.bytes_lo;
            slot_t *s = nullptr;
            int spins = 0;
            while (!(s = S.storage.allocate(bytes)))
            {
                // all slots of every fitting class are in use: wait for the compute thread to retire consumers
                finish(); // (the blocks already on their way may be what those consumers wait for)
                if (++spins == 1)
                    fprintf(stderr, "[pangulu_amd] rank %d: receive buffers exhausted, waiting (raise PANGULU_AMD_RECV_BUDGET_GB or mpi_recv_buffer_level)\n", S.rank);
                usleep(200);
                if (spins > 300000)
                    fatal("receive buffers exhausted for 60 s: dependency deadlock");
            }
            if (rec_multi)
            {
                if (!rec_used_slots.insert(s).second)
                    S.rank_log.slot_reused = true;
            }
            bind_record(*s, S.nb, h.nnz, (char *)s->value - 32, (char *)s->d_value - 32, h.brow > h.bcol, h.brow == h.bcol && h.is_upper);
            s->brow_pos = h.brow;
            s->bcol_pos = h.bcol;
            s->is_upper = (i32)h.is_upper;
            comm->recv_block_begin(s, h, src);
            begun.push_back(Begun{s, h});
            recv_inflight.store((int)begun.size(), std::memory_order_relaxed);
            if (begun.size() >= 128)
                finish();
        }
    }
};

} // namespace

int g_replay_enabled = -1; // pangulu_amd_set_replay; -1: not set, PANGULU_AMD_REPLAY decides (default on)

static bool schedule_possible(const Solver &S, Platform &plat)
{
    static const bool replay_env = !(getenv("PANGULU_AMD_REPLAY") && atoi(getenv("PANGULU_AMD_REPLAY")) == 0);
    const bool replay_on = g_replay_enabled < 0 ? replay_env : g_replay_enabled != 0;
    return replay_on && S.nproc == 1 && !plat.host_memory && plat.schedule && g_task_sample_stride <= 1 && !S.eager_host_mirror && !S.analysis_only &&
           !getenv("PANGULU_AMD_FORCE_MULTI_LOOP");
}

// pangulu_init, one rank on the device: run the scheduler once with the back-end in record-only mode -- every batch goes
// through the same host code as in a factorisation (mirror bookkeeping, descriptors, work lists), nothing is launched -- so that
// already the FIRST pangulu_gstrf of the handle replays a launch list instead of scheduling 2.8 million tasks beside the device.
// PANGULU_AMD_RECORD_AT_INIT=0 leaves the recording to the first pangulu_gstrf (which then runs the scheduler).
void record_schedule(Solver &S)
{
    Platform &plat = active_platform();
    static const bool at_init = !(getenv("PANGULU_AMD_RECORD_AT_INIT") && atoi(getenv("PANGULU_AMD_RECORD_AT_INIT")) == 0);
    if (!at_init || !schedule_possible(S, plat))
        return;
    const double t0 = wall_seconds();
    plat.set_option(PANGULU_HIP_OPT_HOST_MIRROR, 0);
    plat.set_option(PANGULU_HIP_OPT_ASSUME_INDEPENDENT, 1);
    plat.set_option(PANGULU_HIP_OPT_RESET_BLOCK_STATE, 0);
    if (plat.schedule(4, &S) != 0)
        return;
    {
        Sched sch(S, true);
        S.heap.clear();
        S.pending_total = 0;
        S.pending_dirty.clear();
        {
            std::lock_guard<std::mutex> g(S.info_mutex);
            for (u32 level = 0; level < S.nbk; level++)
                if (S.remain_diag[level] == 1)
                {
                    S.remain_diag[level]--;
                    sch.push_panel(level, level, level, PANGULU_TASK_GETRF, S.diag_lower[level], nullptr);
                }
        }
        sch.compute_loop();
        S.info.batches = sch.batches;
        S.info.deferred_queues = sch.deferred;
    }
    S.schedule_recorded = plat.schedule(2, &S) > 0;
    // re-arm: nothing ran on the device, the block values are untouched
    S.remain = S.remain0;
    S.remain_diag = S.remain_diag0;
    S.rank_remain_task = S.rank_remain_task0;
    for (auto &sl : S.storage.owned)
        sl.data_status = PANGULU_DATA_PREPARING;
    for (auto &q : S.pending)
        q.clear();
    S.pending_dirty.clear();
    S.pending_total = 0;
    S.heap.clear();
    plat.set_option(PANGULU_HIP_OPT_RESET_BLOCK_STATE, 0);
    S.info.time_schedule_record = wall_seconds() - t0;
    if (getenv("PANGULU_AMD_TRACE"))
        fprintf(stderr, "[pangulu_amd trace] init: launch schedule recorded by a dry run of the scheduler in %.2f s (%s)\n", S.info.time_schedule_record,
                S.schedule_recorded ? "ok" : "nothing recorded");
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-rank replay (round 4; the default since round 5, PANGULU_AMD_MULTI_REPLAY=0 turns it off).  A rank's FIRST factorisation runs the scheduler and records -- the
// back-end its launches (schedule cmd 5: like a single-rank recording, descriptor segments packed), the scheduler this rank's log
// (Solver::RankLog): the operation range of every platform call, the markers between them and the blocks announced behind each,
// and the blocks of other ranks in the order they were handled, with the receive slot each one landed in and how many had been
// handled when each platform call was made.  Later factorisations replay the log: the receive thread puts every announced block
// into the slot it had (the descriptors hold slot addresses), the compute thread walks the entries -- wait until the blocks the
// entry needed have arrived, replay its operations, record its marker, post its sends behind it.  No heap, no counters, no
// descriptor building.  Every rank's order is what HAPPENED on it in one run, and so are the messages: the union is acyclic, a
// replay cannot deadlock, whatever the other ranks do (they may replay or schedule).  Not replayable, and then simply not
// replayed: a first run in which a receive slot was used twice (recycling: its address would have to be re-timed), or whose
// transport copies at post time (host staging).  Cost of the approach: the batch boundaries of the first run, arrival-driven
// fragmentation included, are frozen.
// ---------------------------------------------------------------------------------------------------------------------
static bool multi_replay_enabled()
{
    // ON by default since round 5 (PANGULU_AMD_MULTI_REPLAY=0 turns it off): N = 1 and N > 1 then run the same kind of loop -- the
    // first factorisation of a handle schedules and records, later ones replay -- and a rank whose log cannot be replayed (a receive
    // slot used twice, a transport that copies at post time, options changed since) falls back to the scheduler by itself, before
    // its start barrier, whatever the other ranks do.
    static const bool on = !getenv("PANGULU_AMD_MULTI_REPLAY") || atoi(getenv("PANGULU_AMD_MULTI_REPLAY")) != 0;
    return on && g_replay_enabled != 0;
}

static bool replay_rank_log(Solver &S)
{
    Platform &plat = active_platform();
    Comm *comm = world();
    Solver::RankLog &L = S.rank_log;
    if (!L.valid || !plat.schedule || !plat.schedule_range || !plat.marker_record_replay)
        return false;
    // The decision to replay is taken BEFORE the start barrier: a rank that falls through to the scheduler makes that
    // path's one start barrier, a rank that replays makes this one -- every rank exactly one, whatever the others decided.  (Checked
    // behind the barrier, a rank whose list had gone stale made three barriers per factorisation against its peers' two and the
    // run hung until the stall limit.)  cmd 9 is the validity rule of cmd 7 as a pure query.
    if (plat.schedule(9, &S) != 0)
    {
        L.valid = false; // (options or back-end resources changed since the recording: schedule again, and record again)
        return false;
    }
    comm->barrier();
    const double t0 = wall_seconds();
    if (plat.schedule(7, &S) != 0)
        fatal("rank %d (replay): the recorded launch list went stale between the validity query and the start of the replay", S.rank);
    const size_t narr = L.arrivals.size();
    std::unique_ptr<std::atomic<unsigned char>[]> arrived(new std::atomic<unsigned char>[narr + 1]);
    std::unordered_map<u64, u32> index; // (brow, bcol, is_upper) -> position in the first run's arrival order
    index.reserve(narr * 2 + 16);
    auto key_of = [](u32 brow, u32 bcol, u32 up) -> u64
    { return ((u64)brow << 33) | ((u64)bcol << 1) | (u64)(up & 1u); };
    for (size_t i = 0; i < narr; i++)
    {
        arrived[i].store(0, std::memory_order_relaxed);
        index[key_of(L.arrivals[i].brow, L.arrivals[i].bcol, L.arrivals[i].is_upper)] = (u32)i;
    }
    double stall_limit_s = 120;
    if (const char *e = getenv("PANGULU_AMD_STALL_S"))
        stall_limit_s = atof(e);
    std::vector<std::unique_ptr<Marker>> markers((size_t)L.nmarkers);
    for (auto &m : markers)
        m.reset(new Marker());

    std::thread worker([&]()
                       {
                           size_t have = 0, send_at = 0;
                           long long op = 0;
                           for (const Solver::RankLog::Entry &e : L.entries)
                           {
                               double t_wait = 0;
                               while (have < e.need)
                               {
                                   if (arrived[have].load(std::memory_order_acquire))
                                   {
                                       have++;
                                       t_wait = 0;
                                       continue;
                                   }
                                   const double now = wall_seconds();
                                   if (t_wait == 0)
                                       t_wait = now;
                                   if (now - t_wait > stall_limit_s)
                                       fatal("rank %d (replay): block (%u,%u) of the first run's arrival order has not come for %.0f s", S.rank,
                                             L.arrivals[have].brow, L.arrivals[have].bcol, now - t_wait);
                                   usleep(5);
                               }
                               if (e.op_end > op)
                               {
                                   if (plat.schedule_range(&S, op, e.op_end) != 0)
                                       fatal("rank %d (replay): the recorded launch list is gone", S.rank);
                                   op = e.op_end;
                               }
                               if (e.marker >= 0)
                               {
                                   Marker *m = markers[(size_t)e.marker].get();
                                   m->ev.store(plat.marker_record_replay(), std::memory_order_release);
                                   if (send_at < L.sends.size() && L.sends[send_at].marker == e.marker)
                                   {
                                       if (!comm->set_send_gate(m))
                                           fatal("rank %d (replay): the transport no longer defers sends", S.rank);
                                       while (send_at < L.sends.size() && L.sends[send_at].marker == e.marker)
                                       {
                                           const Solver::RankLog::Send &sd = L.sends[send_at++];
                                           comm->isend_block(sd.slot, sd.h, sd.dst);
                                       }
                                   }
                               }
                           }
                           if (send_at != L.sends.size())
                               fatal("rank %d (replay): %zu recorded sends were never posted", S.rank, L.sends.size() - send_at);
                           // (operations recorded behind the last logged call: the stream joins of the first run's final synchronise)
                           const long long total = plat.schedule(6, &S);
                           if (total > op)
                               plat.schedule_range(&S, op, total);
                       });
    // receive thread: every announced block into the slot it had in the first run
    {
        std::vector<u32> begun;
        size_t done = 0;
        double t_last = 0;
        auto finish = [&]()
        {
            if (begun.empty())
                return;
            comm->recv_blocks_finish();
            for (u32 i : begun)
                arrived[i].store(1, std::memory_order_release);
            done += begun.size();
            begun.clear();
        };
        while (done + begun.size() < narr)
        {
            BlockHeader h;
            int src = -1;
            if (!comm->probe_block(h, src))
            {
                if (!begun.empty())
                    finish();
                else
                {
                    const double now = wall_seconds();
                    if (t_last == 0)
                        t_last = now;
                    if (now - t_last > stall_limit_s)
                        fatal("rank %d (replay): no block arrived for %.0f s with %zu receives outstanding", S.rank, now - t_last, narr - done);
                    usleep(10);
                }
                continue;
            }
            t_last = 0;
            auto it = index.find(key_of(h.brow, h.bcol, h.is_upper));
            if (it == index.end())
                fatal("rank %d (replay): block (%u,%u) was not received in the recorded run", S.rank, h.brow, h.bcol);
            slot_t *s = L.arrivals[it->second].slot;
            bind_record(*s, S.nb, h.nnz, (char *)s->value - 32, (char *)s->d_value - 32, h.brow > h.bcol, h.brow == h.bcol && h.is_upper);
            s->brow_pos = h.brow;
            s->bcol_pos = h.bcol;
            s->is_upper = (i32)h.is_upper;
            comm->recv_block_begin(s, h, src);
            begun.push_back(it->second);
            if (begun.size() >= 128)
                finish();
        }
        finish();
    }
    worker.join();
    plat.schedule(8, &S);
    plat.synchronize();
    comm->flush_sends();
    comm->set_send_gate(nullptr);
    comm->barrier();
    S.info.time_numeric = wall_seconds() - t0;
    S.info.time_numeric_host_sched = S.info.time_numeric_platform = 0;
    S.info.replayed = 1;
    S.info.sent_bytes = comm->sent_bytes;
    S.info.recv_bytes = comm->recv_bytes_total;
    for (auto &sl : S.storage.owned)
        sl.data_status = PANGULU_DATA_READY;
    S.rank_remain_task = 0;
    S.rank_remain_recv = 0;
    S.factored = true;
    S.host_values_current = false;
    return true;
}

void numeric_factorize(Solver &S)
{
    if (S.factored)
        fatal("pangulu_gstrf called twice on one handle (call pangulu_init again)");
    if (getenv("PANGULU_AMD_TRACE"))
        fprintf(stderr, "[pangulu_amd trace] rank %d: numeric factorisation starts\n", S.rank);
    Comm *comm = world();
    Platform &plat = active_platform();
    // Static schedule (one rank on the device; PANGULU_AMD_REPLAY=0 turns it off): the launches of a factorisation depend on
    // the block pattern and the back-end's options only, so the first pangulu_gstrf of a handle records them in the back-end
    // (pangulu_platform_0201001_schedule) and every later one replays the list -- no scheduler, no descriptor building.
    const bool can_schedule = schedule_possible(S, plat);
    if (plat.set_option)
    {
        plat.set_option(PANGULU_HIP_OPT_HOST_MIRROR, S.eager_host_mirror ? 1 : 0);
        plat.set_option(PANGULU_HIP_OPT_ASSUME_INDEPENDENT, 1); // batches of this scheduler are dependency-free
    }
    if (can_schedule && S.schedule_recorded)
    {
        comm->barrier();
        const double t0 = wall_seconds();
        if (plat.schedule(3, &S) == 0)
        {
            const double t_issued = wall_seconds() - t0; // (the host's share of a replay: issuing the recorded launches)
            plat.synchronize();
            S.info.time_numeric = wall_seconds() - t0;
            S.info.time_numeric_host_sched = 0;
            S.info.time_numeric_platform = t_issued;
            if (getenv("PANGULU_AMD_TRACE"))
                fprintf(stderr, "[pangulu_amd trace] replay: launches issued in %.2f ms, factorisation done after %.2f ms\n", 1e3 * t_issued, 1e3 * S.info.time_numeric);
            S.info.replayed = 1;
            for (auto &sl : S.storage.owned)
                sl.data_status = PANGULU_DATA_READY;
            S.rank_remain_task = 0;
            S.factored = true;
            S.host_values_current = false;
            return;
        }
        S.schedule_recorded = false; // (other options than at recording time, or another handle recorded since: a normal run)
    }
    const bool multi_replay = S.nproc > 1 && !plat.host_memory && multi_replay_enabled() && plat.schedule && plat.schedule_range && g_task_sample_stride <= 1 &&
                              !S.eager_host_mirror;
    if (multi_replay && S.rank_log.valid && replay_rank_log(S))
        return;
    S.info.replayed = 0;
    Sched sch(S);
    S.info.sampled_flop = 0;
    S.info.sampled_tasks = 0;
    if (plat.set_option)
        plat.set_option(PANGULU_HIP_OPT_RESET_BLOCK_STATE, 0);  // block values were (re)loaded behind the back-end's back
    const bool recording = can_schedule && plat.schedule(1, &S) == 0;
    if (multi_replay && sch.use_markers && !S.rank_log.unusable)
    {
        S.rank_log = Solver::RankLog();
        sch.rec_multi = plat.schedule(5, &S) == 0; // (drops an older recording of this back-end first)
    }
    S.heap.clear();
    S.pending_total = 0;
    S.pending_dirty.clear();
    comm->barrier();
    double t0 = wall_seconds();
    {
        std::lock_guard<std::mutex> g(S.info_mutex);
        for (u32 level = 0; level < S.nbk; level++)
        {
            if (S.owner(level, level) == S.rank && S.remain_diag[level] == 1)
            {
                S.remain_diag[level]--;
                sch.push_panel(level, level, level, PANGULU_TASK_GETRF, S.diag_lower[level], nullptr);
            }
        }
    }
    S.host_values_current = plat.host_memory;
    if (S.nproc > 1 || sch.forced_multi)
    {
        std::thread worker([&]()
                           { sch.compute_loop(); });
        sch.receive_loop();
        worker.join();
        plat.synchronize();
        sch.poll_retired(); // (everything has completed: all of them go back)
        comm->flush_sends();
        comm->set_send_gate(nullptr);
        if (sch.rec_multi)
        {
            Solver::RankLog &L = S.rank_log;
            L.nmarkers = (int)sch.markers.size();
            const bool listed = plat.schedule(2, &S) > 0;
            L.valid = listed && sch.rec_gates_ok && !L.slot_reused && (i64)L.arrivals.size() == S.rank_remain_recv0;
            L.unusable = !L.valid && listed; // (for a structural reason: the next factorisation would log the same thing)
            if (!L.valid && listed)
                plat.schedule(0, &S); // (drop the list: nobody will replay it)
            S.schedule_recorded = L.valid; // (the back-end holds a list of this handle: dropped with the handle)
            if (getenv("PANGULU_AMD_TRACE"))
                fprintf(stderr, "[pangulu_amd trace] rank %d: log of this factorisation: %zu platform calls and markers, %zu sends, %zu arrivals -- %s\n", S.rank,
                        L.entries.size(), L.sends.size(), L.arrivals.size(),
                        L.valid ? "later factorisations replay it" : (L.slot_reused ? "a receive slot was used twice: not replayable" : "not replayable"));
        }
    }
    else
    {
        sch.compute_loop();
        if (recording)
            S.schedule_recorded = plat.schedule(2, &S) > 0;
        plat.synchronize();
    }
    if (getenv("PANGULU_AMD_TRACE"))
        fprintf(stderr, "[pangulu_amd trace] rank %d: compute and receive loops done, entering the final barrier\n", S.rank);
    comm->barrier();
    S.info.time_numeric = wall_seconds() - t0;
    // host time the scheduler itself needed: with the launcher thread, the time until the last batch was handed over
    S.info.time_numeric_host_sched = sch.async_launch ? sch.t_sched : S.info.time_numeric - sch.t_platform;
    S.info.batches = sch.batches;
    S.info.deferred_queues = sch.deferred;
    S.info.time_numeric_platform = sch.t_platform;
    S.info.sent_bytes = comm->sent_bytes;
    S.info.recv_bytes = comm->recv_bytes_total;
    S.factored = true;
    if (S.eager_host_mirror && !plat.host_memory)
        S.host_values_current = true;
}

} // namespace pg
