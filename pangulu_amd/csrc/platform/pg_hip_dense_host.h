// pg_hip_dense_host.h -- host-side bookkeeping of dense-mode blocks (included inside the anonymous namespace of
// pg_hip_platform.hip, after Backend / Segment / host_nnz / diag_halves).
#pragma once

#if defined(PG_DENSE_UPDATES)

struct BlockState
{
    double *mirror = nullptr;
    bool mirror_current = false; // mirror holds the block's current values
    bool sparse_current = true;  // the sparse record holds the block's current values
    bool lu_image = false;       // diagonal block: the mirror holds L\\U with inverted diagonal tiles (pg_hip_trsm_dense.h)
    bool lu_map = false;         // ... and the occupancy map behind it describes the factorised block (tiled GETRF)
    bool written = false;        // a kernel has written the block's SPARSE RECORD in this factorisation (sparse update, sparse solve, GETRF)
    bool densified_once = false; // the block has had a densify job in this factorisation
    unsigned char image_halves = 0; // ... of which triangles: 1 = strictly lower (L), 2 = upper (U)
    u32 brow = 0, bcol = 0, nnz = 0;
    // Host-side occupancy summary of an OWNED off-diagonal block's pattern (pangulu_platform_0201001_prepare_blocks, once
    // per block at preprocessing; the pattern is symbolic and never changes; survives reset_block_states):
    //   occ_a[t] bit s: column slab s (16 wide) has entries in the rows of 128-row half t   (the block as left operand)
    //   occ_b[t] bit s: row slab s has entries in the columns of 128-column half t          (the block as right operand)
    //   occ_rows bit r / occ_cols bit c: row slab r / column slab c has entries at all          (the block in a dense solve)
    // The launch code uses it to leave out workgroups that would find nothing to do.
    bool occ_valid = false;
    unsigned short occ_a[2] = {0, 0}, occ_b[2] = {0, 0}, occ_rows = 0, occ_cols = 0;
    // ... and the full 16 x 16 occupancy map, as densify writes it behind the mirror (word c: row slabs with entries in column
    // slab c) and transposed (word r: column slabs with entries in row slab r).  Update tasks carry them to the MFMA kernel,
    // which then needs no second memory round trip (descriptor -> mirror -> map) before it knows what to fetch.
    unsigned short occ_map[16] = {0}, occ_map_t[16] = {0};
};

// Open-addressing table block key -> BlockState.  Looked up three times per update task (destination and both operands)
// and emptied at every new factorisation: no node allocations, clear() keeps the capacity.
struct BlockTable
{
    struct Entry
    {
        const void *key = nullptr;
        BlockState st;
    };
    std::vector<Entry> tab;
    size_t mask = 0, count = 0;
    static size_t hash(const void *k)
    {
        unsigned long long x = (unsigned long long)(uintptr_t)k >> 3;
        x *= 0x9E3779B97F4A7C15ull;
        return (size_t)(x >> 24);
    }
    void prefetch(const void *k) const
    {
        if (count)
            __builtin_prefetch(&tab[hash(k) & mask]);
    }
    BlockState *find(const void *k)
    {
        if (count == 0)
            return nullptr;
        for (size_t i = hash(k) & mask;; i = (i + 1) & mask)
        {
            if (tab[i].key == k)
                return &tab[i].st;
            if (!tab[i].key)
                return nullptr;
        }
    }
    // (the reference is valid until the next insertion)
    BlockState &operator[](const void *k)
    {
        if ((count + 1) * 2 > tab.size())
            grow();
        for (size_t i = hash(k) & mask;; i = (i + 1) & mask)
        {
            if (tab[i].key == k)
                return tab[i].st;
            if (!tab[i].key)
            {
                tab[i].key = k;
                tab[i].st = BlockState();
                count++;
                return tab[i].st;
            }
        }
    }
    void grow()
    {
        std::vector<Entry> old;
        old.swap(tab);
        tab.resize(old.empty() ? (size_t)1 << 16 : old.size() * 2);
        mask = tab.size() - 1;
        count = 0;
        for (const Entry &e : old)
            if (e.key)
                (*this)[e.key] = e.st;
    }
    void clear()
    {
        for (Entry &e : tab)
            e.key = nullptr;
        count = 0;
    }
    template <class F>
    void for_each(F f)
    {
        for (Entry &e : tab)
            if (e.key)
                f(e.st);
    }
};

struct MirrorPool
{
    std::vector<char *> chunks;
    size_t chunk_bytes = 0, cursor = 0, mirror_bytes = 0; // cursor counts mirrors handed out since the last reset
    size_t limit_mirrors = 0, peak = 0; // peak: most mirrors in use at once since the pool was (re)started
    BlockTable blocks; // key: d_value of the (lower half of the) block
    std::vector<MirrorJobD> to_densify, to_sparsify;
    // early densify (recorded schedules): jobs moved out of the launch order since the last wait point, and the descriptor
    // segment the prologue's launches read them from
    std::vector<MirrorJobD> early;
    char *early_h = nullptr, *early_d = nullptr;
    size_t early_cap = 0, early_used = 0;
    unsigned long long early_jobs = 0, early_chunks = 0;
};
MirrorPool MP;

inline bool dense_mode_available(int nb)
{
    // 128 or 256: the work lists, the per-task live-slab tables and the (group, tile) encoding of a launch are built for at
    // most 2 x 2 tiles of 128 and 16 K-slabs; larger block orders stay on the pattern-driven kernels (found by
    // test_block_orders_beyond_the_tuned_ones: nb = 384 used to overrun those tables)
    return (nb % DG_TILE == 0) && nb <= 256 && B.opt_dense_permille <= 1000;
}

// Is C -= A*B worth the matrix cores?  The LDS kernel's time grows with the structural flops, about 2*nnzA*nnzB/nb for
// evenly spread patterns, at well under 1 TFLOP/s; the MFMA kernel spends 2*nb^3 flops whatever the fill, at tens of
// TFLOP/s.  The threshold t (per mille, PANGULU_HIP_OPT_DENSE_THRESHOLD_PERMILLE) is the geometric-mean fill of the
// two operands above which the update goes dense: sqrt(dA*dB) >= t/1000.
inline bool is_heavy_update(u32 nnz_a, u32 nnz_b, int nb)
{
    const double full = (double)nb * (double)nb;
    const double t = (double)B.opt_dense_permille / 1000.0;
    return (double)nnz_a * (double)nnz_b >= t * t * full * full;
}

inline const void *block_key(slot_t *s)
{
    if (s->brow_pos == s->bcol_pos && s->is_upper && s->related_block)
        return (const void *)s->related_block->d_value;
    return (const void *)s->d_value;
}

BlockState &block_state(slot_t *s, int nb)
{
    s = canon_dst(s); // a diagonal block is known by its lower half
    BlockState &st = MP.blocks[block_key(s)];
    u32 nnz = host_nnz(s, nb);
    if (st.brow != s->brow_pos || st.bcol != s->bcol_pos || st.nnz != nnz)
    {
        // first sight, or a receive slot that now holds another block: forget everything but the memory
        // (the occupancy summary goes too: it described the previous block)
        double *keep = st.mirror;
        st = BlockState();
        st.mirror = keep;
        st.brow = s->brow_pos;
        st.bcol = s->bcol_pos;
        st.nnz = nnz;
    }
    return st;
}

// a mirror for the block, or nullptr when the pool's budget is exhausted (callers then stay on the sparse path)
double *obtain_mirror(BlockState &st, int nb)
{
    if (st.mirror)
        return st.mirror;
    // values + occupancy map (pg_hip_dense.h) + the nb/16 diagonal tiles a blocked GETRF saves before they are inverted;
    // CR64: two such planes (real, imaginary)
    size_t mb = PG_PLANES * (sizeof(double) * (size_t)nb * nb + MIRROR_MAP_BYTES + sizeof(double) * 16 * (size_t)nb);
    if (MP.mirror_bytes != mb)
    {
        // block order changed (or first use): start over (a recorded schedule's launches point into the old chunks)
        B.generation++;
        for (char *c : MP.chunks)
            HIP_CHECK(hipFree(c));
        MP.chunks.clear();
        MP.blocks.for_each([](BlockState &st)
                           { st.mirror = nullptr; });
        MP.mirror_bytes = mb;
        MP.cursor = 0;
        MP.peak = 0;
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const char *frac_env = getenv("PANGULU_HIP_MIRROR_FRACTION");
        double frac = frac_env ? atof(frac_env) : 0.85; // of what is free now (0.6 until round 4): a LIMIT -- chunks are allocated as mirrors are handed out --; the rest stays for descriptor twins, scratch, receive slots
        MP.limit_mirrors = (size_t)(frac * (double)free_b / (double)mb);
        MP.chunk_bytes = mb * 1024; // 512 MiB at nb = 256
    }
    if (MP.cursor >= MP.limit_mirrors)
        return nullptr;
    size_t per_chunk = MP.chunk_bytes / mb;
    size_t chunk = MP.cursor / per_chunk, slot = MP.cursor % per_chunk;
    if (chunk >= MP.chunks.size())
    {
        char *c = nullptr;
        if (hipMalloc((void **)&c, MP.chunk_bytes) != hipSuccess)
        {
            (void)hipGetLastError();
            MP.limit_mirrors = MP.cursor;
            return nullptr;
        }
        MP.chunks.push_back(c);
    }
    MP.cursor++;
    MP.peak = std::max(MP.peak, MP.cursor);
    st.mirror = reinterpret_cast<double *>(MP.chunks[chunk] + slot * mb);
    return st.mirror;
}

MirrorJobD mirror_job(slot_t *s, double *dense, int nb)
{
    MirrorJobD J;
    memset(&J, 0, sizeof(J));
    if (s->brow_pos == s->bcol_pos)
    {
        slot_t *up, *lo;
        diag_halves(s, &up, &lo);
        J.lo = BlkView{lo->d_columnpointer, lo->d_rowindex, lo->d_value};
        const DiagAux &aux = get_diag_aux(up, nb);
        J.ucp = aux.d_cp;
        J.uri = aux.d_ri;
        J.uvi = aux.d_vi;
        J.uval = up->d_value;
    }
    else
    {
        J.lo = BlkView{s->d_columnpointer, s->d_rowindex, s->d_value};
    }
    J.dense = dense;
    {
        // what the job moves by construction: every pattern entry once as (index, value) in the record and once as a
        // value in the image, plus the record's pointer array(s)
        double nnz = host_nnz(canon_dst(s), nb);
        double ptrs = 4.0 * (nb + 1);
        if (s->brow_pos == s->bcol_pos)
        {
            slot_t *up, *lo;
            diag_halves(s, &up, &lo);
            nnz = (double)host_nnz(lo, nb) + host_nnz(up, nb);
            ptrs *= 2;
        }
        J.move_bytes = (unsigned long long)(nnz * (sizeof(double) + 2.0 + sizeof(double)) + ptrs);
    }
    return J;
}

// the block's mirror holds newer values than its sparse record (updates have been accumulating there)
bool mirror_is_ahead(slot_t *s)
{
    s = canon_dst(s);
    const BlockState *found = MP.blocks.find(block_key(s));
    if (!found)
        return false;
    const BlockState &st = *found;
    return st.brow == s->brow_pos && st.bcol == s->bcol_pos && st.mirror && !st.sparse_current;
}

#if defined(PG_DENSE_PANELS) || defined(PG_COMPLEX_PANELS)
// dense LU image of the diagonal block `half` belongs to, if GETRF left one
const double *lu_image_of(slot_t *half)
{
    slot_t *lo = canon_dst(half);
    const BlockState *found = MP.blocks.find(block_key(lo));
    if (!found)
        return nullptr;
    const BlockState &st = *found;
    if (st.brow != lo->brow_pos || st.bcol != lo->bcol_pos || !st.lu_image || !(st.image_halves & (half->is_upper ? 2 : 1)))
        return nullptr;
    return st.mirror;
}
bool lu_image_has_map(slot_t *half)
{
    slot_t *lo = canon_dst(half);
    const BlockState *found = MP.blocks.find(block_key(lo));
    return found && found->lu_image && found->lu_map;
}
#endif
#if defined(PG_DENSE_PANELS)

// A diagonal block factorised on another rank has no image here: queue one to be built from the halves that have
// arrived (pg_hip_trsm_dense.h, half_image_kernel).  Returns the mirror, or nullptr when the pool is exhausted.
std::vector<HalfImageJobD> g_half_image_jobs;
const double *request_half_image(slot_t *half, int nb)
{
    slot_t *key_slot = canon_dst(half);
    BlockState &st = block_state(key_slot, nb);
    double *m = obtain_mirror(st, nb);
    if (!m)
        return nullptr;
    slot_t *lower = half->is_upper ? half->related_block : half;
    slot_t *upper = half->is_upper ? half : half->related_block;
    HalfImageJobD J;
    memset(&J, 0, sizeof(J));
    unsigned char halves = 0;
    if (lower && lower->d_columnpointer)
    {
        J.lcp = lower->d_columnpointer;
        J.lri = lower->d_rowindex;
        J.lval = lower->d_value;
        halves |= 1;
    }
    if (upper && upper->d_rowpointer)
    {
        J.urp = upper->d_rowpointer;
        J.uci = upper->d_columnindex;
        J.uval = upper->d_value;
        halves |= 2;
    }
    J.dense = m;
    g_half_image_jobs.push_back(J);
    st.lu_image = true;
    st.lu_map = false; // (half_image_kernel writes no map)
    st.image_halves = halves;
    st.mirror_current = false; // (the mirror is an image now, not the block)
    st.sparse_current = true;
    return m;
}

#endif // R64: LU images

// Early densify (PANGULU_HIP_EARLY_DENSIFY=1; OFF by default) applies to recordings made by a dry run (nothing executes while
// recording, so moving a launch is free of consequences for the recording run itself).
// MEASURED (profiles/r04k_early_densify_ab.log, one box): correct -- same residuals and factor checks, 43 676 jobs in 53 chunks on
// shell(398), 164 889 in 140 on fem27(112) -- and NOT faster: shell(398) 38.0-38.4 ms against 37.6, fem27(112) 653.1 against 650.5,
// poisson3d(64) 26.3 against 25.2.  The 4.4 ms of exclusive densify time a kernel trace showed on the shell were largely the
// tracer's launch gaps; untraced, the jobs cost less where they were than their 21 GB of mirror writes cost the latency-bound leaf
// levels they now run beside.
inline bool early_densify_on()
{
    static const bool on = getenv("PANGULU_HIP_EARLY_DENSIFY") && atoi(getenv("PANGULU_HIP_EARLY_DENSIFY")) != 0;
    return on && REC.mode == 2;
}

// make sure the block has a mirror holding its current values; queues a densify job if it has to be (re)built.
// Returns nullptr when no mirror can be had.
double *current_mirror(slot_t *s, int nb)
{
    BlockState &st = block_state(s, nb);
    double *m = obtain_mirror(st, nb);
    if (!m)
        return nullptr;
    if (!st.mirror_current)
    {
        // (sparse_current holds whenever mirror_current does not)
        // The first densify of a block whose record no kernel has written depends on nothing that happens in the factorisation: a
        // dry-run recording moves it into the prologue (flush_early_jobs)
        if (early_densify_on() && !st.written && !st.densified_once)
            MP.early.push_back(mirror_job(s, m, nb));
        else
            MP.to_densify.push_back(mirror_job(s, m, nb));
        st.densified_once = true;
        st.mirror_current = true;
    }
    return m;
}

// the block's sparse values are about to be read or overwritten: bring them up to date first
void require_sparse(slot_t *s, int nb)
{
    s = canon_dst(s);
    BlockState *found = MP.blocks.find(block_key(s));
    if (!found)
        return;
    BlockState &st = *found;
    if (st.brow != s->brow_pos || st.bcol != s->bcol_pos)
        return;
    if (!st.sparse_current && st.mirror)
    {
        MP.to_sparsify.push_back(mirror_job(s, st.mirror, nb));
        st.sparse_current = true;
    }
}

// `on_records_stream` (sparsify jobs of blocks that are FINISHED, whose mirrors nothing will write again): the launch
// goes to the records stream, ordered behind everything queued on `after` so far, and the main stream does not wait for it
// (`fork_recorded`: the caller has already recorded B.ev_rec_fork at the point the jobs depend on)
void flush_mirror_jobs(int nb, std::vector<MirrorJobD> &jobs, bool densify, bool on_records_stream = false, hipStream_t after = nullptr,
                       bool fork_recorded = false)
{
    HostTimer ht(3);
    hipStream_t st = B.stream;
    if (on_records_stream && !densify && B.opt_records_stream && !jobs.empty())
    {
        st = B.stream_rec;
        if (!fork_recorded)
            pg_event_record(B.ev_rec_fork, after ? after : B.stream);
        pg_stream_wait(st, B.ev_rec_fork);
    }
    // (main-stream jobs need no join with the records stream: its jobs in flight write the records of FINISHED blocks
    // whose mirrors stay current; densify reads, and sparsify on the main stream rewrites, records of unfinished ones)
    size_t i = 0;
    while (i < jobs.size())
    {
        Segment seg = acquire_segment();
        size_t take = std::min(jobs.size() - i, seg.cap / (sizeof(MirrorJobD) + 16));
        MirrorJobD *d_jobs;
        MirrorJobD *h = seg.alloc<MirrorJobD>(take, &d_jobs);
        memcpy(h, jobs.data() + i, sizeof(MirrorJobD) * take);
        commit_segment(seg);
        // few jobs: cut every block into column runs so the launch still covers the chip
        // (enough workgroups to hide the per-workgroup chain of dependent loads: aim at ~16k)
        static const long target_env = getenv("PANGULU_HIP_MIRROR_JOB_WGS") ? atol(getenv("PANGULU_HIP_MIRROR_JOB_WGS")) : 16384;
        unsigned slices = 16;
        while (slices > 1 && (size_t)slices * take > (size_t)target_env)
            slices >>= 1;
        slices = std::min<unsigned>(slices, (unsigned)std::max(1, nb / 16)); // whole 16-column slabs per workgroup
        {
            LaunchTimer lt(densify ? 6 : 7, st);
            if (densify)
                PG_LAUNCH(densify_kernel, dim3((unsigned)take, slices), dim3(256), 0, st, d_jobs, nb);
            else
                PG_LAUNCH(sparsify_kernel, dim3((unsigned)take, slices), dim3(256), 0, st, d_jobs, nb);
        }
        HIP_CHECK(hipGetLastError());
        B.stats.launches[densify ? 6 : 7]++;
        B.stats.tasks[densify ? 6 : 7] += take;
        for (size_t k = 0; k < take; k++)
            B.stats.alg_bytes[densify ? 6 : 7] += jobs[i + k].move_bytes;
        release_pending_segments(st); // (callers commit their own segment only after this returns)
        i += take;
    }
    if (st == B.stream_rec)
    {
        pg_event_record(B.ev_rec, st);
        B.rec_dirty.store(true, std::memory_order_release);
    }
    jobs.clear();
}

// The jobs moved out of the launch order since the last wait point become ONE chunk of the prologue -- a densify launch on the
// early stream, an event behind it -- and the main stream waits for that event HERE, where the jobs' launch used to be: everything
// that reads these mirrors is ordered behind this point of the main stream (side streams fork from it afterwards).  In a replay
// the prologue's chunks run from the first moment on, in the order the factorisation consumes them, beside the latency-bound leaf
// levels; by the time the main stream gets to a chunk's wait it has long completed (shell(398): densify was 4.4 of 37.5 ms
// exclusive, tools/critical_path.py).
void flush_early_jobs(int nb)
{
    if (MP.early.empty())
        return;
    HostTimer ht(3);
    const size_t take = MP.early.size();
    const size_t need = sizeof(MirrorJobD) * take + 64;
    if (MP.early_used + need > MP.early_cap)
    {
        Segment seg = acquire_segment(); // (recording: a segment of its own, kept with the recording and twinned in HBM)
        if (need > seg.cap)
        {
            // (more first-touch jobs at one point than a segment holds: leave them in the launch order)
            MP.to_densify.insert(MP.to_densify.end(), MP.early.begin(), MP.early.end());
            MP.early.clear();
            return;
        }
        MP.early_h = seg.h;
        MP.early_d = seg.d;
        MP.early_cap = seg.cap;
        MP.early_used = 0;
    }
    const size_t off = (MP.early_used + 15) & ~(size_t)15;
    memcpy(MP.early_h + off, MP.early.data(), sizeof(MirrorJobD) * take);
    const MirrorJobD *d_jobs = reinterpret_cast<const MirrorJobD *>(MP.early_d + off);
    MP.early_used = off + sizeof(MirrorJobD) * take;
    static const long target_env = getenv("PANGULU_HIP_MIRROR_JOB_WGS") ? atol(getenv("PANGULU_HIP_MIRROR_JOB_WGS")) : 16384;
    unsigned slices = 16;
    while (slices > 1 && (size_t)slices * take > (size_t)target_env)
        slices >>= 1;
    slices = std::min<unsigned>(slices, (unsigned)std::max(1, nb / 16));
    hipEvent_t ev;
    HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    REC.early_events.push_back(ev);
    REC.in_prologue = true;
    PG_LAUNCH(densify_kernel, dim3((unsigned)take, slices), dim3(256), 0, B.stream_early, d_jobs, nb);
    pg_event_record(ev, B.stream_early);
    REC.in_prologue = false;
    pg_stream_wait(B.stream, ev);
    B.stats.launches[6]++;
    B.stats.tasks[6] += take;
    for (const MirrorJobD &J : MP.early)
        B.stats.alg_bytes[6] += J.move_bytes;
    MP.early_jobs += take;
    MP.early_chunks++;
    MP.early.clear();
}

void reset_block_states()
{
    // everything about values and mirrors is forgotten; the identity of owned blocks and their occupancy summaries stay
    MP.blocks.for_each([](BlockState &st)
                       {
                           BlockState fresh;
                           fresh.brow = st.brow;
                           fresh.bcol = st.bcol;
                           fresh.nnz = st.nnz;
                           fresh.occ_valid = st.occ_valid;
                           fresh.occ_a[0] = st.occ_a[0];
                           fresh.occ_a[1] = st.occ_a[1];
                           fresh.occ_b[0] = st.occ_b[0];
                           fresh.occ_b[1] = st.occ_b[1];
                           fresh.occ_rows = st.occ_rows;
                           fresh.occ_cols = st.occ_cols;
                           memcpy(fresh.occ_map, st.occ_map, sizeof(fresh.occ_map));
                           memcpy(fresh.occ_map_t, st.occ_map_t, sizeof(fresh.occ_map_t));
                           st = fresh; });
    MP.cursor = 0;
    MP.to_densify.clear();
    MP.to_sparsify.clear();
    MP.early.clear();
}

#else // other value types have no dense mode

inline bool dense_mode_available(int) { return false; }
inline void require_sparse(slot_t *, int) {}
inline void reset_block_states() {}

#endif
