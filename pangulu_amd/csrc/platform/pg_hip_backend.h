// pg_hip_backend.h -- host-side state of the back-end: streams and options (Backend), the static-schedule recorder, descriptor
// segments, launch timers, slot helpers.  Included inside the anonymous namespace of pg_hip_platform.hip, before the launch code.
#pragma once

struct DiagAux // column view of a diagonal block's upper (CSR) half, built on first use
{
    u32 *d_cp = nullptr;
    u16 *d_ri = nullptr;
    u32 *d_vi = nullptr;
    u32 nnz = 0;
    u32 brow = 0;
};

struct Ring // descriptor staging in pinned host memory that the kernels read in place, reused segment by segment
{
    static const int NSEG = 32;
    std::vector<int> pending; // segments handed to kernels since the last event record
    size_t seg_bytes = 0;
    char *h = nullptr, *d = nullptr;
    hipEvent_t ev[NSEG];
    bool used[NSEG];
    int cur = 0;
};

struct EventPair
{
    hipEvent_t a, b;
    int cls;          // 100 + k: kernel k of a class-5 launch (Backend stats ssssm_kernel_ms[k]), not a class of its own
    int split_cls;    // > 0: `split_frac` of the duration goes to this class instead (a solve launch with TSTRF and GESSM tasks)
    double split_frac;
    unsigned long long tag[3]; // per-launch log (PANGULU_HIP_LAUNCH_LOG): workgroups, tasks, live 128 x 128 x 16 slab steps
};

#define PG_FLOP_WORDS 48 // device words behind Backend::d_flops: counters, debug stamps, the pipe GETRF's phase stamps

struct Backend
{
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;
    bool bulk_streams_masked = false; // stream / stream2 leave PANGULU_HIP_RESERVED_CUS CUs to the GETRF stream
    hipStream_t stream2 = nullptr; // side stream: the MFMA update kernel runs beside the LDS update kernel
    hipStream_t stream3 = nullptr; // second side stream: GETRFs of a batch run beside its TSTRF/GESSM solves
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork3 = nullptr, ev_join3 = nullptr;
    // the dense-front launch of an update call on a stream of its own, beside the general launch (PANGULU_HIP_FRONT_FORK=1; see launch_ssssm)
    hipStream_t stream_front = nullptr;
    hipEvent_t ev_front_fork = nullptr, ev_front_join = nullptr;
    // Round 4 measured 0.3-0.8 % for it (profiles/r04aa_front_fork_ab.log) and left it OFF because the two launches then overlap in a kernel
    // trace and rocprofv3's per-kernel durations no longer add up to the update class's launch time.  Round 6 measured it again --
    // elastic3d(77) 1 509.5 against 1 520.1 ms, fem27(112) 613.8 / 617.9, kkt(120) 406.0 / 408.0, shell(398) 33.6 / 33.5, two pairs each on
    // one box (profiles/r06w_front_fork_ab.txt) -- and turned it ON: bench.py's per-kernel times come from its one-stream profile pass
    // anyway (the fork is off there: !opt_profile), and tools/profile_recipe.sh runs the rocprofv3 passes with PANGULU_HIP_FRONT_FORK=0
    // so that their per-kernel durations stay additive and comparable with that pass.
    long long opt_front_fork = 1;
    bool getrf_join_pending = false;
    // Records stream: the sparse record stays the authoritative form of every finished block, but the dense kernels
    // of the following steps read mirrors and LU images only.  The sparsify jobs behind the dense solves and behind
    // the blocked GETRF run here, beside whatever comes next; everything that reads or rewrites sparse records (the
    // LDS update kernel, sparse solves, densify, copies to the host, markers, synchronize) joins it first.
    hipStream_t stream_rec = nullptr;
    hipEvent_t ev_rec_fork = nullptr, ev_rec = nullptr;
    std::atomic<bool> rec_dirty{false};
    // Background stream (round 3): in a call that carries diagonal factorisations AND updates (the scheduler's look-ahead:
    // the GETRFs of the next level(s) together with every update queued anywhere), the updates are the trailing-matrix
    // work of the previous level and nothing on the critical path -- next panel's updates, GETRF, panel solves -- depends
    // on them.  They go to this stream and the main stream does NOT join at the end of the call: the solves of the next
    // panel (the following call) run beside them instead of behind them (fem27(112): the dense solves ran ALONE on the
    // device for 65 of 974 ms).  The destinations of the launches in flight are remembered; the first later call that
    // touches one of them -- as destination or operand -- makes the main stream wait first (join_background).
    // MEASURED (fem27(112), one box each): the overlap is there -- dense solves exclusive 65 -> 17 ms, GETRF 13 -> 4, two or more
    // classes at once 81 -> 245 ms.  With the scheduler in the loop the factorisation did not get faster (954.7 against 940.7 ms:
    // the extra call per level cost the host-bound run more than the overlap returned); replayed from the static schedule it
    // does: 873.1 against 887.6 ms, shell(398) 39.05 against 39.72.  On by default since then.
    hipStream_t stream_bg = nullptr;
    hipEvent_t ev_bg_fork = nullptr, ev_bg_done = nullptr;
    // Early stream (round 4, replayed schedules only): the FIRST densify of a block whose record no kernel has written yet depends on
    // nothing -- the recording moves such jobs out of the launch order into a prologue on this stream, which the replay starts before
    // anything else; the main stream waits for a chunk's event where its densify launch used to be (pg_hip_dense_host.h, flush_early_jobs).
    hipStream_t stream_early = nullptr;
    bool bg_active = false;
    std::unordered_set<const void *> bg_tiles;
    long long opt_background_updates = 1; // PANGULU_HIP_BACKGROUND_UPDATES=0 / option 14
    // dense-front kernel (pg_hip_front.h) for the (destination, tile) pairs all of whose queued updates have every 16 x 16
    // piece live: LDS stages of its operand pipeline (2, 3 or 4; 0 = off, everything through the general kernel)
    long long opt_front_stages = 2; // PANGULU_HIP_FRONT_STAGES / option 15: 1 = inside the general launch (no step list), 2..4 = own kernel
    long long opt_front_min_wgs = 2048; // PANGULU_HIP_FRONT_MIN_WGS: ... from this many qualifying workgroups in a launch on (round 3 sweep, fem27(112): 842.8 / 845.0 / 849.0 ms at 8192 / 2048 / never; round 4, with the destination preloaded: 656.9 / 655.4 / 653.6 / 659.6 ms at 8192 / 2048 / 512 / 64)
    long long opt_front_unit = 1;   // PANGULU_HIP_FRONT_UNIT: consecutive destinations of the front launch that share an XCD
    // general MFMA update kernel: 0 = round 2's (register staging, contiguous sub-tiles; pg_hip_dense.h), any other value = the
    // two-stage LDS-DMA pipeline with strided piece ownership (ssssm_tilesv_f64_kernel, pg_hip_front.h).  (The values 1, 3, 4, 5
    // selected round 3's ssssm_tiles_f64_kernel<STAGES> and round 5's ssssm_tilesp_f64_kernel: tools/experiments/ since round 6.)
    long long opt_tiles_stages = 2; // PANGULU_HIP_TILES_STAGES / option 16
    long long opt_tiles_unit = 1;   // PANGULU_HIP_TILES_UNIT: consecutive destinations of the general launch that share an XCD
    unsigned long long front_workgroups = 0, general_workgroups = 0;
    long long opt_records_stream = 1; // PANGULU_HIP_RECORDS_STREAM=0: sparsify on the main stream as before
    int nb_cfg = 0;
    // Bumped whenever a process-global resource that recorded launches point into is freed or re-assigned (the GETRF scratch,
    // the mirror pool, the chase's progress words): a recorded schedule is only replayed under the generation it ended in.
    unsigned long long generation = 0;
    // options
    long long opt_host_mirror = 1;
    long long opt_dense_permille = 2; // (10 until the end of round 2, 5 until round 3's sweep on replayed runs: fem27(112) 887.8 / 892.1 / 906.8 ms at 2 / 5 / 10, shell(398) 37.9 / 38.5 / 39.2)
    long long opt_profile = 0;
    long long opt_assume_independent = 0;
    long long opt_getrf_strict = 0;
    long long opt_count_flops = 1;
    long long opt_group_chunk = 8;
    long long opt_small_launch_tasks = 2048;
    long long opt_trsm_dense_permille = 5; // (round 3 sweep: shell(398) 38.2 / 38.5 / 39.4 ms at 5 / 10 / 30, fem27(112) indifferent)
    long long opt_two_streams = 1;
    double mfma_flops_executed = 0;
    // resources
    Ring ring;
    unsigned *d_progress = nullptr;        // progress words of the GETRF -> dense-solve chase (one per held factorisation task)
    size_t progress_next = 0;
    unsigned long long chase_launches = 0, chase_solves = 0;
    unsigned long long zgetrf_tasks = 0; // complex types: diagonal blocks factorised in their mirrors
    unsigned long long *d_flops = nullptr; // [PG_FLOP_WORDS]
    val_t *getrf_scratch = nullptr;
    int getrf_scratch_slots = 0;
    std::unordered_map<const void *, DiagAux> diag_aux;
    // stats
    pangulu_hip_stats_t stats;
    std::vector<EventPair> pending_events;
    std::vector<hipEvent_t> event_pool;
    std::mutex mutex;
};

Backend B;

// ---------------------------------------------------------------------------------------------------------------
// Static schedule (round 3).  For one rank the sequence of launches of a factorisation -- kernels, grids, descriptor
// contents, stream forks and joins -- is a pure function of the block pattern and the options: nothing in it depends
// on values or on timing (one launcher thread issues everything in the scheduler's order).  The first pangulu_gstrf on a
// handle therefore RECORDS every launch and stream operation it issues (a closure each; the descriptor segments they
// read are kept instead of recycled), and every later pangulu_gstrf on that handle with the same options REPLAYS the
// list: no scheduler, no descriptor building, no host work per task -- about three thousand closures for the
// Serena-class matrix instead of 2.8 million tasks.  pangulu_platform_0201001_schedule() is the control call.
// ---------------------------------------------------------------------------------------------------------------
struct Recorder
{
    int mode = 0; // 1: recording while executing; 2: recording only (dry run of the scheduler at pangulu_init: nothing is launched)
    bool valid = false;
    const void *owner = nullptr;
    unsigned long long signature = 0;
    std::vector<std::function<void()>> ops;
    std::vector<std::function<void()>> prologue; // launched by a replay before `ops` (early densify jobs on their own stream)
    bool in_prologue = false;                    // launches and event records issued now go to `prologue`
    std::vector<hipEvent_t> early_events;        // one per prologue chunk, owned by the recording
    // Descriptor segments of the recorded launches.  While recording, the kernels read them in place from pinned host memory
    // (h, device-visible at d) like every other run; the REPLAYS read a copy in HBM (twin), made once when the recording ends:
    // a workgroup's first two dependent reads -- its work item, its task descriptors -- then cost an L2/HBM round trip instead
    // of two trips to host memory.  The closures are built with the twin addresses (rec_xl), the pinned originals are freed.
    struct Seg
    {
        char *h, *d, *twin;
        size_t cap;
    };
    std::vector<Seg> segs;
    size_t descriptor_bytes = 0;
    // Packed recording (multi-rank runs record thousands of small batches: a fresh 8 MB segment + twin per platform call would be tens
    // of GB): a call gets the free TAIL of a recorded segment with at least 2 MB left and hands back what it did not use at commit.
    // Calls nest (a launch function flushes mirror jobs while it holds its own segment), so a tail in use is out of the list.
    bool pack = false;
    struct Tail
    {
        int seg;
        size_t used;
    };
    std::vector<Tail> tails;
    // what else the closures depend on: the block order and the generation of the back-end's shared resources when the list
    // was complete (B.generation)
    int nb = 0;
    unsigned long long generation = 0;
    // host-side counters of ONE factorisation (launches, tasks, algorithmic bytes, workgroup counts): taken as the difference
    // over the recording, added by every replay; a dry run (mode 2) launched nothing and leaves the live counters as they were
    pangulu_hip_stats_t stats_before, stats_delta;
    unsigned long long wgs_before[4] = {0, 0, 0, 0}, wgs_delta[4] = {0, 0, 0, 0}; // front, general, chase launches, chase solves
};
Recorder REC;

// GETRF -> dense-solve chase (recorded schedules only).  A launch of the tiled GETRF on the main stream is HELD until the next
// platform call: if that call is the level's dense TSTRF/GESSM against exactly these diagonal blocks, both go out as ONE launch
// (getrf_trsm_chase_kernel: the solves of panel p start when the factorisation has published panel p); anything else launches
// the held factorisation first, as it was.  `hold` keeps the preparatory launches of launch_trsm (densify of the panel blocks:
// independent of the factorisation) from doing that.
#define PROGRESS_WORDS 8192
struct PendingGetrf
{
    bool active = false, hold = false;
    int nb = 0;
    size_t take = 0;
    const void *d_tasks = nullptr;       // GetrfTaskD * (device view)
    unsigned *d_progress = nullptr;      // one word per task
    std::vector<const double *> images;  // LU images the held factorisation will leave, in task order
    std::function<void()> plain;         // the launch as it would have been
    std::function<void()> post;          // what follows the launch (record-stream fork, deferred sparsify jobs, statistics)
};
PendingGetrf PEND;
inline void flush_pending_getrf()
{
    if (!PEND.active || PEND.hold)
        return;
    PEND.active = false;
    PEND.plain();
    PEND.post();
    PEND.plain = nullptr;
    PEND.post = nullptr;
}

// PEND is back-end state like everything else: entry points that do not hold B.mutex anyway take it for the flush
inline void flush_pending_getrf_locked();

// a kernel argument as the replay will pass it: pointers into a recorded descriptor segment move to the segment's HBM twin
template <class T>
inline T rec_xl(T v)
{
    if constexpr (std::is_pointer<T>::value)
    {
        const char *p = reinterpret_cast<const char *>(v);
        for (const Recorder::Seg &sg : REC.segs)
            if (p >= sg.d && p < sg.d + sg.cap)
                return reinterpret_cast<T>(const_cast<char *>(sg.twin + (p - sg.d)));
    }
    return v;
}

template <class K, class... A>
inline void pg_launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, A... args)
{
    flush_pending_getrf();
    if (REC.mode != 0)
    {
        auto targs = std::make_tuple(rec_xl(args)...);
        (REC.in_prologue ? REC.prologue : REC.ops).emplace_back([=]()
                             { std::apply([&](auto... a)
                                          { hipLaunchKernelGGL(kernel, grid, block, (unsigned)shmem, st, a...); },
                                          targs); });
        if (REC.mode == 2)
            return;
    }
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)shmem, st, args...);
}
#define PG_LAUNCH(kernel_, grid_, block_, shmem_, stream_, ...) pg_launch(kernel_, grid_, block_, shmem_, stream_, __VA_ARGS__)

inline void pg_event_record(hipEvent_t e, hipStream_t s)
{
    flush_pending_getrf();
    if (REC.mode != 0)
        (REC.in_prologue ? REC.prologue : REC.ops).emplace_back([e, s]() { HIP_CHECK(hipEventRecord(e, s)); });
    if (REC.mode != 2)
        HIP_CHECK(hipEventRecord(e, s));
}
inline void pg_stream_wait(hipStream_t s, hipEvent_t e)
{
    flush_pending_getrf();
    if (REC.mode != 0)
        REC.ops.emplace_back([e, s]() { HIP_CHECK(hipStreamWaitEvent(s, e, 0)); });
    if (REC.mode != 2)
        HIP_CHECK(hipStreamWaitEvent(s, e, 0));
}

inline void flush_pending_getrf_locked()
{
    if (!PEND.active) // (only ever set under the mutex by the thread that launches; a stale read here just skips a no-op)
        return;
    std::lock_guard<std::mutex> g(B.mutex);
    flush_pending_getrf();
}

void ensure_ready()
{
    if (B.ready)
        return;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
    {
        fprintf(stderr, "[PanguLU-AMD ERROR] no HIP device available (%s); the GPU_HIP platform has no CPU fallback\n",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        exit(EXIT_FAILURE);
    }
    HIP_CHECK(hipSetDevice(B.device));
    // Optional (PANGULU_HIP_RESERVED_CUS=n, default 0 = off): the bulk streams (updates, solves, mirror maintenance) leave n
    // CUs alone -- mask bit i is CU i / 8 of XCD i % 8 (tools/experiments/cu_mask_probe.hip) -- and the GETRF stream
    // (stream3) sees all of them.  A GETRF workgroup needs 139 KB of LDS, i.e. a CU to itself, and an update launch that is
    // still handing out workgroups never leaves one empty: the factorisations of the upper tree levels took 240-470 us
    // beside such a launch against 205 us alone (tools/launch_size_histogram.py).  Measured with n = 8: GETRF time 15.3 ->
    // 13.4 ms (bench matrix) and 58 -> 25 ms (fem27(80)), but the update kernel lost more than the 3 % of CUs it gave up
    // (fem27(80): 126 -> 137 ms) and the factorisation did not get faster (47.3 vs 45.4-47.7 ms; 187 vs 175 ms): off.
    {
        long reserved = 0;
        if (const char *e = getenv("PANGULU_HIP_RESERVED_CUS"))
            reserved = atol(e);
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, B.device));
        const int ncu = prop.multiProcessorCount;
        if (reserved > 0 && reserved < ncu / 2 && ncu % 32 == 0)
        {
            std::vector<uint32_t> mask((size_t)ncu / 32, 0xFFFFFFFFu);
            for (int i = ncu - (int)reserved; i < ncu; i++)
                mask[(size_t)i / 32] &= ~(1u << (i % 32));
            HIP_CHECK(hipExtStreamCreateWithCUMask(&B.stream, (uint32_t)mask.size(), mask.data()));
            HIP_CHECK(hipExtStreamCreateWithCUMask(&B.stream2, (uint32_t)mask.size(), mask.data()));
            B.bulk_streams_masked = true;
        }
        else
        {
            HIP_CHECK(hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking));
            HIP_CHECK(hipStreamCreateWithFlags(&B.stream2, hipStreamNonBlocking));
        }
    }
    HIP_CHECK(hipStreamCreateWithFlags(&B.stream_front, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_front_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_front_join, hipEventDisableTiming));
    if (const char *e = getenv("PANGULU_HIP_FRONT_FORK"))
        B.opt_front_fork = atol(e);
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_join, hipEventDisableTiming));
    HIP_CHECK(hipStreamCreateWithFlags(&B.stream3, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_fork3, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_join3, hipEventDisableTiming));
    HIP_CHECK(hipStreamCreateWithFlags(&B.stream_rec, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_rec_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_rec, hipEventDisableTiming));
    if (const char *e = getenv("PANGULU_HIP_RECORDS_STREAM"))
        B.opt_records_stream = atol(e);
    HIP_CHECK(hipStreamCreateWithFlags(&B.stream_early, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&B.stream_bg, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_bg_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_bg_done, hipEventDisableTiming));
    if (const char *e = getenv("PANGULU_HIP_BACKGROUND_UPDATES"))
        B.opt_background_updates = atol(e);
    if (const char *e = getenv("PANGULU_HIP_FRONT_STAGES"))
        B.opt_front_stages = atol(e);
    if (const char *e = getenv("PANGULU_HIP_FRONT_UNIT"))
        B.opt_front_unit = atol(e);
    if (const char *e = getenv("PANGULU_HIP_FRONT_MIN_WGS"))
        B.opt_front_min_wgs = atol(e);
    if (const char *e = getenv("PANGULU_HIP_TILES_STAGES"))
        B.opt_tiles_stages = atol(e);
    if (const char *e = getenv("PANGULU_HIP_GROUP_CHUNK"))
        B.opt_group_chunk = atol(e);
    if (const char *e = getenv("PANGULU_HIP_TILES_UNIT"))
        B.opt_tiles_unit = atol(e);
    if (const char *e = getenv("PANGULU_HIP_DENSE_PERMILLE"))
        B.opt_dense_permille = atol(e);
    if (const char *e = getenv("PANGULU_HIP_TRSM_DENSE_PERMILLE"))
        B.opt_trsm_dense_permille = atol(e);
    if (const char *e = getenv("PANGULU_HIP_SMALL_LAUNCH_TASKS"))
        B.opt_small_launch_tasks = atol(e);
    // Descriptors are written once by the host and read once per workgroup: the kernels read them straight from
    // pinned host memory (non-coherent, so the device L2 may cache them) instead of waiting for a staging copy per
    // launch (rocprofv3 showed ~1900 blit dispatches, ~50 ms, per factorisation of the bench matrix).
    B.ring.seg_bytes = (size_t)8 << 20;
    HIP_CHECK(hipHostMalloc((void **)&B.ring.h, B.ring.seg_bytes * Ring::NSEG, hipHostMallocNonCoherent | hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&B.ring.d, B.ring.h, 0));
    for (int i = 0; i < Ring::NSEG; i++)
    {
        HIP_CHECK(hipEventCreateWithFlags(&B.ring.ev[i], hipEventDisableTiming));
        B.ring.used[i] = false;
    }
    // [0..7] flop counters, [8..15] debug stamps, [16..47] the phase stamps of getrf_pipe_f64_kernel (its slots 16..20 and 24..29 count from
    // d_flops + 8: ADVICE r5 found them 22 words past a 16-word allocation)
    HIP_CHECK(hipMalloc((void **)&B.d_flops, sizeof(unsigned long long) * PG_FLOP_WORDS));
    HIP_CHECK(hipMemset(B.d_flops, 0, sizeof(unsigned long long) * PG_FLOP_WORDS));
    memset(&B.stats, 0, sizeof(B.stats));
    B.ready = true;
}

// a staging segment: host pointer to fill, device pointer the kernels will read after commit()
struct Segment
{
    char *h, *d;
    size_t cap, used;
    int index;       // ring segment, or -1: a recorded segment of its own, -2: the free tail of a recorded segment (packed recording)
    int rec_idx = -1; // index -2: which recorded segment
    size_t rec_off = 0; // ... and where this part starts
    template <typename T>
    T *alloc(size_t count, T **dev)
    {
        size_t off = (used + 15) & ~(size_t)15;
        if (off + sizeof(T) * count > cap)
            return nullptr;
        used = off + sizeof(T) * count;
        *dev = reinterpret_cast<T *>(d + off);
        return reinterpret_cast<T *>(h + off);
    }
};

// host-side cost of preparing launches (PANGULU_HIP_HOST_TIMING=1 prints it with every get_stats(reset))
double g_host_seconds[6] = {0, 0, 0, 0, 0, 0}; // 0 ssssm, 1 trsm, 2 getrf, 3 mirror jobs, 4 waiting for a staging segment, 5 whole calls
struct HostTimer
{
    int k;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { g_host_seconds[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// Record, behind everything launched so far, that the committed segments may be reused.  Must be called AFTER the
// kernels reading those segments have been launched (an event recorded earlier would let the host overwrite a
// segment a queued kernel has yet to read).
void release_pending_segments(hipStream_t on = nullptr)
{
    Ring &r = B.ring;
    for (int i : r.pending)
    {
        HIP_CHECK(hipEventRecord(r.ev[i], on ? on : B.stream));
        r.used[i] = true;
    }
    r.pending.clear();
}

// sparse records are about to be read or rewritten on stream s: wait for the sparsify jobs of the records stream
void join_records(hipStream_t s)
{
    if (!B.rec_dirty.load(std::memory_order_acquire))
        return;
    pg_stream_wait(s, B.ev_rec);
    if (s == B.stream)
        B.rec_dirty.store(false, std::memory_order_release);
}

// stream s is about to touch blocks that update launches on the background stream may still be writing (or: everything
// queued so far has to be complete behind s)
void join_background(hipStream_t s)
{
    if (!B.bg_active)
        return;
    pg_stream_wait(s, B.ev_bg_done);
    if (s == B.stream)
    {
        B.bg_active = false;
        B.bg_tiles.clear();
    }
}

// `want`: bytes the caller is going to fill (0: unknown -- it sizes its launch by what it gets).  Only a RECORDING cares: the
// descriptors of a recorded schedule stay for the life of the handle, in a pinned host segment while recording and in an HBM twin
// afterwards.  Until round 5 every platform call of a single-rank recording took a segment of 8 MB of its own whatever it wrote:
// 13.2 GB of HBM on elastic3d(77) for 2-3 GB of descriptors.  Now every recording packs: a call takes the free tail of a recorded
// segment that has room for what it wants (an update call states its task count: it is not cut into more launches than before),
// or a new segment just large enough, and hands back what it did not use.
Segment acquire_segment(size_t want = 0)
{
    Ring &r = B.ring;
    if (REC.mode != 0)
    {
        const size_t min_room = want ? std::min(r.seg_bytes, (want + 4096 + 65535) & ~(size_t)65535) : ((size_t)2 << 20);
        // recording: the launches will be replayed, their descriptors have to stay
        for (size_t k = REC.tails.size(); k-- > 0;)
        {
            const Recorder::Tail t = REC.tails[k];
            const Recorder::Seg &sg = REC.segs[(size_t)t.seg];
            if (sg.cap - t.used < min_room)
                continue;
            REC.tails.erase(REC.tails.begin() + (long)k);
            Segment s;
            s.h = sg.h + t.used;
            s.d = sg.d + t.used;
            s.cap = sg.cap - t.used;
            s.used = 0;
            s.index = -2;
            s.rec_idx = t.seg;
            s.rec_off = t.used;
            return s;
        }
        // (a new segment: what is wanted, or a chunk of 4 MB whose tail serves the small calls that follow)
        const size_t bytes = std::min(r.seg_bytes, std::max(min_room, (size_t)4 << 20));
        char *h = nullptr, *d = nullptr, *twin = nullptr;
        HIP_CHECK(hipHostMalloc((void **)&h, bytes, hipHostMallocNonCoherent | hipHostMallocMapped));
        HIP_CHECK(hipHostGetDevicePointer((void **)&d, h, 0));
        HIP_CHECK(hipMalloc((void **)&twin, bytes));
        REC.segs.push_back(Recorder::Seg{h, d, twin, bytes});
        REC.descriptor_bytes += bytes;
        Segment s;
        s.h = h;
        s.d = d;
        s.cap = bytes;
        s.used = 0;
        s.index = -2;
        s.rec_idx = (int)REC.segs.size() - 1;
        s.rec_off = 0;
        return s;
    }
    int i = r.cur;
    r.cur = (r.cur + 1) % Ring::NSEG;
    if (r.used[i])
    {
        HostTimer ht(4);
        HIP_CHECK(hipEventSynchronize(r.ev[i])); // the kernels that last read this segment are done
    }
    Segment s;
    s.h = r.h + (size_t)i * r.seg_bytes;
    s.d = r.d + (size_t)i * r.seg_bytes;
    s.cap = r.seg_bytes;
    s.used = 0;
    s.index = i;
    return s;
}

// the segment is complete: kernels launched from now on may read it (in place, see ensure_ready)
void commit_segment(Segment &s)
{
    if (s.index >= 0)
        B.ring.pending.push_back(s.index);
    else if (s.index == -2 && REC.mode != 0)
    {
        // what the call did not use goes back, if it is worth a call
        const size_t used = s.rec_off + ((s.used + 255) & ~(size_t)255);
        if (used + ((size_t)64 << 10) <= REC.segs[(size_t)s.rec_idx].cap)
            REC.tails.push_back(Recorder::Tail{s.rec_idx, used});
    }
}

hipEvent_t take_event()
{
    if (!B.event_pool.empty())
    {
        hipEvent_t e = B.event_pool.back();
        B.event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
}

struct LaunchTimer
{
    int cls;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    unsigned long long tag[3] = {0, 0, 0};
    int split_cls = 0;
    double split_frac = 0.0;
    explicit LaunchTimer(int c, hipStream_t stream = nullptr) : cls(c), st(stream ? stream : B.stream)
    {
        if (B.opt_profile)
        {
            a = take_event();
            b = take_event();
            HIP_CHECK(hipEventRecord(a, st));
        }
    }
    ~LaunchTimer()
    {
        if (B.opt_profile)
        {
            HIP_CHECK(hipEventRecord(b, st));
            B.pending_events.push_back(EventPair{a, b, cls, split_cls, split_frac, {tag[0], tag[1], tag[2]}});
        }
    }
};

void harvest_events()
{
    // PANGULU_HIP_LAUNCH_LOG=<file> (with PROFILE on): one line per launch -- class, microseconds, workgroups, tasks, live slab
    // steps -- for tuning the update kernel by launch shape (tools/launch_log_summary.py)
    static FILE *launch_log = getenv("PANGULU_HIP_LAUNCH_LOG") ? fopen(getenv("PANGULU_HIP_LAUNCH_LOG"), "w") : nullptr;
    for (auto &p : B.pending_events)
    {
        HIP_CHECK(hipEventSynchronize(p.b));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        if (p.cls >= 100)
        {
            if (p.cls - 100 < 2)
                B.stats.ssssm_kernel_ms[p.cls - 100] += ms;
            B.event_pool.push_back(p.a);
            B.event_pool.push_back(p.b);
            continue;
        }
        if (p.split_cls > 0)
        {
            B.stats.elapsed_ms[p.split_cls] += ms * p.split_frac;
            B.stats.elapsed_ms[p.cls] += ms * (1.0 - p.split_frac);
        }
        else
            B.stats.elapsed_ms[p.cls] += ms;
        if (launch_log)
            fprintf(launch_log, "%d %.2f %llu %llu %llu\n", p.cls, 1e3 * ms, p.tag[0], p.tag[1], p.tag[2]);
        B.event_pool.push_back(p.a);
        B.event_pool.push_back(p.b);
    }
    B.pending_events.clear();
    if (launch_log)
        fflush(launch_log);
}

inline u32 host_nnz(const slot_t *s, int nb) { return s->columnpointer[nb]; }

// both halves of a diagonal block are one destination: name it by its lower half
inline slot_t *canon_dst(slot_t *s)
{
    if (s->brow_pos == s->bcol_pos && s->is_upper && s->related_block)
        return s->related_block;
    return s;
}

// identity of a block for the background-stream bookkeeping (both halves of a diagonal block are one block)
inline const void *block_key_any(const slot_t *s)
{
    if (s->brow_pos == s->bcol_pos && s->is_upper && s->related_block)
        return (const void *)s->related_block->d_value;
    return (const void *)s->d_value;
}

// Building the descriptors of a task touches, per operand, the slot struct, the last entry of its pattern pointer array and
// its block-table entry -- nine cache misses per update, and the leaf levels of the bench matrix (8000 updates + 4000
// solves per level) were bound by this thread, not by the device.  Two-stage software prefetch, a fixed distance ahead in
// the task list: the slot structs first, then what their fields point to.
inline void prefetch_task_slots(const task_t *t)
{
    for (const slot_t *s : {t->op1, t->op2, t->opdst})
        if (s)
        {
            __builtin_prefetch(s);
            __builtin_prefetch(reinterpret_cast<const char *>(s) + 64);
            __builtin_prefetch(reinterpret_cast<const char *>(s) + 128);
        }
}
void prefetch_task_details(const task_t *t, int nb); // (needs the block table: defined after pg_hip_dense_host.h)

inline void diag_halves(slot_t *any, slot_t **upper, slot_t **lower)
{
    if (any->is_upper)
    {
        *upper = any;
        *lower = any->related_block;
    }
    else
    {
        *upper = any->related_block;
        *lower = any;
    }
    if (!*upper || !*lower)
    {
        fprintf(stderr, "[PanguLU-AMD ERROR] diagonal block (%u,%u) is missing its other half\n", any->brow_pos, any->bcol_pos);
        exit(EXIT_FAILURE);
    }
}

// column view of the upper half of a diagonal block (needed when it is an SSSSM destination: the update runs
// column by column, the half is stored by rows)
const DiagAux &get_diag_aux(slot_t *upper, int nb)
{
    auto it = B.diag_aux.find((const void *)upper->d_value);
    u32 nnz = host_nnz(upper, nb);
    if (it != B.diag_aux.end() && it->second.nnz == nnz && it->second.brow == upper->brow_pos)
        return it->second;
    DiagAux aux;
    aux.nnz = nnz;
    aux.brow = upper->brow_pos;
    const u32 *rp = upper->columnpointer; // CSR row pointer (host naming, see pangulu_platform.h)
    const u16 *ci = upper->rowindex;
    std::vector<u32> cp(nb + 1, 0), vi(nnz);
    std::vector<u16> ri(nnz);
    for (u32 p = 0; p < nnz; p++)
        cp[ci[p] + 1]++;
    for (int c = 0; c < nb; c++)
        cp[c + 1] += cp[c];
    std::vector<u32> cur(cp.begin(), cp.end() - 1);
    for (int r = 0; r < nb; r++)
        for (u32 p = rp[r]; p < rp[r + 1]; p++)
        {
            u32 o = cur[ci[p]]++;
            ri[o] = (u16)r;
            vi[o] = p;
        }
    size_t bytes_cp = sizeof(u32) * (nb + 1), bytes_vi = sizeof(u32) * nnz, bytes_ri = sizeof(u16) * nnz;
    char *d = nullptr;
    size_t off_vi = (bytes_cp + 15) & ~(size_t)15, off_ri = (off_vi + bytes_vi + 15) & ~(size_t)15;
    HIP_CHECK(hipMalloc((void **)&d, off_ri + bytes_ri + 16));
    HIP_CHECK(hipMemcpy(d, cp.data(), bytes_cp, hipMemcpyHostToDevice));
    if (nnz)
    {
        HIP_CHECK(hipMemcpy(d + off_vi, vi.data(), bytes_vi, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d + off_ri, ri.data(), bytes_ri, hipMemcpyHostToDevice));
    }
    aux.d_cp = (u32 *)d;
    aux.d_vi = (u32 *)(d + off_vi);
    aux.d_ri = (u16 *)(d + off_ri);
    if (it != B.diag_aux.end())
    {
        HIP_CHECK(hipFree(it->second.d_cp));
        it->second = aux;
        return it->second;
    }
    return B.diag_aux.emplace((const void *)upper->d_value, aux).first->second;
}

void mirror_to_host(slot_t *s, int nb)
{
    size_t bytes = sizeof(val_t) * (size_t)host_nnz(s, nb);
    join_records(B.stream);
    join_background(B.stream);
    if (bytes)
        HIP_CHECK(hipMemcpyAsync(s->value, s->d_value, bytes, hipMemcpyDeviceToHost, B.stream));
}
