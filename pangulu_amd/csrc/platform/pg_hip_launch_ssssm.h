// pg_hip_launch_ssssm.h -- host side of the update launches: dense / sparse decision per task, work lists, descriptors,
// the MFMA and LDS update kernels' launches.  Included inside the anonymous namespace of pg_hip_platform.hip.
#pragma once

// ---- SSSSM -----------------------------------------------------------------------------------------------------
// Tasks arrive grouped by destination.  Per group the destination is either dense-mode (updates accumulate in its
// mirror) or sparse; per task the update runs on the matrix cores when destination and both operands have mirrors,
// on the LDS-accumulator kernel otherwise.
#define DG_TILE_HOST 128 // = DG_TILE of pg_hip_dense.h (R64 only; harmless elsewhere)
// Tasks per launch (PANGULU_HIP_LAUNCH_CHUNK).  The host builds the descriptors of a launch before it can start: a leaf level
// of the bench matrix has 8000 updates and 4000 solves, and the device sat idle for 260 us per level while their mirror
// jobs and descriptors were written.  Cut into chunks, the first kernels run while the rest is being prepared.
size_t launch_chunk_tasks()
{
    static const size_t chunk = []()
    {
        const char *e = getenv("PANGULU_HIP_LAUNCH_CHUNK");
        long v = e ? atol(e) : 0;
        return v > 0 ? (size_t)v : ~(size_t)0;
    }();
    return chunk;
}

// `background`: the update kernels of this call go to the background stream (see Backend::stream_bg); their mirror jobs
// stay on the main stream, in front of the fork
void launch_ssssm(int nb, task_t **list, size_t n, bool background = false)
{
    if (n == 0)
        return;
    HostTimer ht(0);
    hipStream_t const ms = background ? B.stream_bg : B.stream; // where the update kernels of this call run
    const bool dense_ok = dense_mode_available(nb);
    // A call that does not fit one descriptor segment is cut into several launches.  On the background stream the mirror jobs
    // of a later launch -- queued on the MAIN stream -- are not ordered behind the kernels of the earlier ones: a destination
    // whose first updates went to its sparse record there and whose mirror is built here would be read before they have
    // landed (found at the end of round 4: poisson3d(48) with a dense threshold of 100 per mille, factor check 4e-3).  The
    // destinations of the earlier launches of this call are kept, and the main stream waits for them when a job touches one.
    static std::unordered_set<const void *> bg_earlier;
    bg_earlier.clear();
    size_t i = 0;
    while (i < n)
    {
        bool bg_hazard = false;
        const size_t chunk_begin = i;
        // worst case per task: one group + one task descriptor in each class; fill until the segment is full
        // (per update: a task descriptor in each class -- PG_PLANES^2 real products on the MFMA side --, a group in each, four
        // work items per MFMA group; the K-split of very small launches multiplies groups and work items of <= 64 tasks by four)
        const size_t per_task = sizeof(SsssmTaskD) * (1 + PG_PLANES * PG_PLANES) + sizeof(SsssmGroupD) * (1 + PG_PLANES) +
                                sizeof(SsssmWorkD) * 8 * PG_PLANES;
        const size_t fixed_part = 64 * 4 * PG_PLANES * (sizeof(SsssmGroupD) + 4 * sizeof(SsssmWorkD)) + 4096;
        Segment seg = acquire_segment(std::min(n - i, launch_chunk_tasks()) * per_task + fixed_part + 65536);
        size_t max_tasks = (seg.cap - fixed_part) / per_task;
        size_t take = std::min(n - i, std::min(max_tasks, launch_chunk_tasks()));
        SsssmTaskD *d_tasks_s, *d_tasks_d;
        SsssmGroupD *d_groups_s, *d_groups_d;
        SsssmTaskD *tasks_s = seg.alloc<SsssmTaskD>(take, &d_tasks_s);
        SsssmTaskD *tasks_d = seg.alloc<SsssmTaskD>(take * PG_PLANES * PG_PLANES, &d_tasks_d); // (CR64: four real products per update)
        const int tiles_per_dim = nb >= DG_TILE_HOST ? nb / DG_TILE_HOST : 1;
        const unsigned ksplit = (nb <= 256 && nb % 64 == 0 && take * (size_t)(tiles_per_dim * tiles_per_dim) <= 64) ? 4u : 1u;
        SsssmGroupD *groups_s = seg.alloc<SsssmGroupD>(take, &d_groups_s);
        SsssmGroupD *groups_d = seg.alloc<SsssmGroupD>(take * ksplit * PG_PLANES, &d_groups_d);
        static std::vector<unsigned short> live_k; // per dense task and tile: K-slabs in which both operands have entries
        live_k.assign(take * 4 * PG_PLANES * PG_PLANES, 0);
        static std::vector<unsigned char> full_t; // per dense task: tiles on which the update is a dense-front product
        full_t.assign(take * PG_PLANES * PG_PLANES, 0);
        SsssmWorkD *d_work, *d_work_f;
        SsssmWorkD *work = seg.alloc<SsssmWorkD>(take * ksplit * 4 * PG_PLANES, &d_work); // every workgroup of the MFMA launch ...
        SsssmWorkD *work_f = seg.alloc<SsssmWorkD>(take * 4 * PG_PLANES, &d_work_f);      // ... and of the dense-front launch
        if (!tasks_s || !tasks_d || !groups_s || !groups_d || !work || !work_f)
        {
            fprintf(stderr, "[PanguLU-AMD ERROR] descriptor staging segment too small\n");
            exit(EXIT_FAILURE);
        }
        size_t ns = 0, nd = 0, gs = 0, gd = 0, nd_updates = 0; // nd: real MFMA tasks; nd_updates: the updates they stand for
        double bytes_s = 0, bytes_d = 0;
        size_t end = i + take;
        while (i < end)
        {
            slot_t *dst = canon_dst(list[i]->opdst);
            size_t j = i;
            while (j < end && canon_dst(list[j]->opdst) == dst)
                j++;
            const bool diag = dst->brow_pos == dst->bcol_pos;
            SsssmGroupD G;
            memset(&G, 0, sizeof(G));
            u32 nnz_c;
            slot_t *up = nullptr, *lo = dst;
            if (diag)
            {
                diag_halves(dst, &up, &lo);
                nnz_c = host_nnz(lo, nb) + host_nnz(up, nb);
            }
            else
            {
                nnz_c = host_nnz(dst, nb);
            }
#if defined(PG_DENSE_UPDATES)
            // The destination works on its dense mirror when the mirror is already ahead of the sparse record, or
            // when at least one update of the group is heavy enough for the matrix cores.
            double *cm = nullptr;
            const size_t jobs_before = MP.to_densify.size() + MP.to_sparsify.size() + MP.early.size();
            if (dense_ok)
            {
                bool want = mirror_is_ahead(dst);
                for (size_t t = i; t < j && !want; t++)
                    want = is_heavy_update(host_nnz(list[t]->op1, nb), host_nnz(list[t]->op2, nb), nb);
                if (want)
                    cm = current_mirror(dst, nb);
                if (cm)
                {
                    block_state(dst, nb).sparse_current = false; // from now on the mirror is ahead of the record
                    G.cdense = reinterpret_cast<val_t *>(cm);
                }
            }
            if (!cm)
            {
                require_sparse(dst, nb);
                block_state(dst, nb).written = true; // (the sparse kernel updates the record: its first densify is no longer free to move)
            }
            if (background && !bg_earlier.empty() && MP.to_densify.size() + MP.to_sparsify.size() + MP.early.size() != jobs_before &&
                bg_earlier.count(block_key_any(dst)))
                bg_hazard = true;
#endif
            if (!G.cdense)
            {
                G.c = BlkView{lo->d_columnpointer, lo->d_rowindex, lo->d_value};
                if (diag)
                {
                    const DiagAux &aux = get_diag_aux(up, nb);
                    G.ucp = aux.d_cp;
                    G.uri = aux.d_ri;
                    G.uvi = aux.d_vi;
                    G.uval = up->d_value;
                }
            }
            size_t s0 = ns;
#if defined(PG_DENSE_UPDATES)
            // updates of this destination that go to the matrix cores: (operand mirrors, live K-slabs per tile)
            struct Heavy
            {
                SsssmTaskD T;
                unsigned short live[4];
                unsigned char full; // bit tl: every 16 x 16 piece of both operands that meets tile tl is live (dense front)
            };
            static thread_local std::vector<Heavy> heavy;
            heavy.clear();
#endif
            for (size_t t = i; t < j; t++)
            {
                if (t + PREFETCH_SLOTS_AHEAD < n)
                    prefetch_task_slots(list[t + PREFETCH_SLOTS_AHEAD]);
                if (t + PREFETCH_DETAILS_AHEAD < n)
                    prefetch_task_details(list[t + PREFETCH_DETAILS_AHEAD], nb);
                slot_t *a = list[t]->op1, *b = list[t]->op2;
                SsssmTaskD T;
                memset(&T, 0, sizeof(T));
                T.a = BlkView{a->d_columnpointer, a->d_rowindex, a->d_value};
                T.b = BlkView{b->d_columnpointer, b->d_rowindex, b->d_value};
                T.sign = 1.0;
                T.count = 1;
                u32 na = host_nnz(a, nb), nbz = host_nnz(b, nb);
                double by = (SV + 2) * ((double)na + nbz) + (2 * SV + 2) * (double)nnz_c + 12.0 * (nb + 1);
                bool on_mfma = false;
#if defined(PG_DENSE_UPDATES)
                if (G.cdense && is_heavy_update(na, nbz, nb))
                {
                    double *am = current_mirror(a, nb);
                    double *bm = am ? current_mirror(b, nb) : nullptr;
                    if (am && bm)
                    {
                        T.a.val = reinterpret_cast<val_t *>(am); // the pattern pointers stay: the flop counter reads them
                        T.b.val = reinterpret_cast<val_t *>(bm);
                        on_mfma = true;
                    }
                }
                if (on_mfma)
                {
                    // tiles of the destination this update can reach (tile = tm + tiles * tn), per K-slab
                    Heavy H;
                    H.T = T;
                    const BlockState *sa = MP.blocks.find(block_key(a)), *sb = MP.blocks.find(block_key(b));
                    for (int tl = 0; tl < 4; tl++)
                        H.live[tl] = tl >= tiles_per_dim * tiles_per_dim ? (unsigned short)0
                                     : (sa && sb && sa->occ_valid && sb->occ_valid)
                                         ? (unsigned short)(sa->occ_a[tl % tiles_per_dim] & sb->occ_b[tl / tiles_per_dim])
                                         : (unsigned short)0xFFFF;
                    H.full = 0;
                    if (sa && sb && sa->occ_valid && sb->occ_valid)
                    {
                        H.T.has_map = 1;
                        memcpy(H.T.amap, sa->occ_map, sizeof(H.T.amap));
                        memcpy(H.T.bmap_t, sb->occ_map_t, sizeof(H.T.bmap_t));
                        const int nslab = nb / 16;
                        const unsigned pm = nb >= 128 ? 0xFFu : ((1u << nslab) - 1u);
                        for (int tl = 0; tl < tiles_per_dim * tiles_per_dim; tl++)
                        {
                            const int tm = tl % tiles_per_dim, tn = tl / tiles_per_dim;
                            bool all = true;
                            for (int sl = 0; sl < nslab && all; sl++)
                                all = (((unsigned)H.T.amap[sl] >> (8 * tm)) & pm) == pm && (((unsigned)H.T.bmap_t[sl] >> (8 * tn)) & pm) == pm;
                            if (all)
                                H.full |= (unsigned char)(1u << tl);
                        }
                    }
                    heavy.push_back(H);
                    bytes_d += by;
                    nd_updates++;
                }
#endif
                if (!on_mfma)
                {
                    tasks_s[ns++] = T;
                    bytes_s += by;
                }
            }
            // cut long queues into chunks that run concurrently and merge with atomics
            // (a launch with few updates cannot fill the chip with whole queues: one update per workgroup then)
            size_t chunk = (size_t)(B.opt_group_chunk > 0 ? B.opt_group_chunk : 1 << 30);
            if (B.opt_group_chunk > 0 && take <= (size_t)B.opt_small_launch_tasks)
                chunk = 1;
            size_t nheavy = 0;
#if defined(PG_DENSE_UPDATES)
            nheavy = heavy.size();
#endif
            // ... and a destination updated by both kernels at once (they run side by side on two streams) must take
            // atomics from both
            const bool split = (ns - s0) > chunk || nheavy > chunk || ((ns > s0) && nheavy && B.opt_two_streams);
            for (size_t c = s0; c < ns; c += chunk)
            {
                G.task_begin = (u32)c;
                G.task_end = (u32)std::min(ns, c + chunk);
                G.atomic = split ? 1u : 0u;
                groups_s[gs++] = G;
            }
#if defined(PG_DENSE_UPDATES)
            // R64: one task per update.  CR64: per destination plane the two real products of every update, consecutive, so
            // that one accumulator pass serves both (C_re -= A_re B_re - A_im B_im;  C_im -= A_re B_im + A_im B_re)
            for (int plane = 0; plane < PG_PLANES && nheavy; plane++)
            {
                const size_t d0 = nd;
                for (const Heavy &H : heavy)
                    for (int term = 0; term < PG_PLANES; term++)
                    {
                        SsssmTaskD T = H.T;
#if PG_PLANES > 1
                        const size_t ps = mirror_plane_stride(nb);
                        double *am = reinterpret_cast<double *>(H.T.a.val), *bm = reinterpret_cast<double *>(H.T.b.val);
                        // plane 0 (real):  + A_re B_re  - A_im B_im      plane 1 (imaginary):  + A_re B_im  + A_im B_re
                        const int a_im = term, b_im = plane ^ term;
                        T.a.val = reinterpret_cast<val_t *>(am + (a_im ? ps : 0));
                        T.b.val = reinterpret_cast<val_t *>(bm + (b_im ? ps : 0));
                        T.sign = (plane == 0 && term == 1) ? -1.0 : 1.0;
                        T.count = (plane == 0 && term == 0) ? 1u : 0u;
#endif
                        for (int tl = 0; tl < 4; tl++)
                            live_k[nd * 4 + tl] = H.live[tl];
                        full_t[nd] = H.full;
                        tasks_d[nd++] = T;
                    }
                SsssmGroupD GP = G;
#if PG_PLANES > 1
                GP.cdense = reinterpret_cast<val_t *>(reinterpret_cast<double *>(G.cdense) + (size_t)plane * mirror_plane_stride(nb));
#endif
                const size_t dchunk = chunk >= ((size_t)1 << 28) ? chunk : chunk * PG_PLANES;
                for (size_t c = d0; c < nd; c += dchunk)
                {
                    GP.task_begin = (u32)c;
                    GP.task_end = (u32)std::min(nd, c + dchunk);
                    GP.atomic = (split || ksplit > 1) ? 1u : 0u;
                    const unsigned slabs = (unsigned)nb / 16u, per = slabs / ksplit;
                    for (unsigned q = 0; q < ksplit; q++)
                    {
                        GP.slab_mask = ksplit > 1 ? (((1u << per) - 1u) << (q * per)) : 0u;
                        const unsigned kmask = GP.slab_mask ? GP.slab_mask : 0xFFFFu;
                        GP.live_tiles = 0;
                        for (u32 t = GP.task_begin; t < GP.task_end; t++)
                            for (int tl = 0; tl < tiles_per_dim * tiles_per_dim; tl++)
                                if (live_k[(size_t)t * 4 + tl] & kmask)
                                    GP.live_tiles |= 1u << tl;
                        groups_d[gd++] = GP;
                    }
                }
            }
#endif
            G.slab_mask = 0;
            G.live_tiles = 0;
            i = j;
        }
#if defined(PG_DENSE_UPDATES)
        if (bg_hazard)
        {
            pg_event_record(B.ev_bg_done, ms);
            pg_stream_wait(B.stream, B.ev_bg_done);
        }
        // mirrors that have to be (re)built for this launch, and sparse records that must catch up first
        if (!MP.to_sparsify.empty())
            flush_mirror_jobs(nb, MP.to_sparsify, false);
        flush_early_jobs(nb);
        if (!MP.to_densify.empty())
            flush_mirror_jobs(nb, MP.to_densify, true);
#endif
        // longest queues first: workgroups are dispatched in grid order, so the big groups start at once and the small
        // ones fill the tail of the launch
        auto by_size = [](const SsssmGroupD &x, const SsssmGroupD &y)
        { return (x.task_end - x.task_begin) > (y.task_end - y.task_begin); };
        std::stable_sort(groups_s, groups_s + gs, by_size);
        std::stable_sort(groups_d, groups_d + gd, by_size);
        commit_segment(seg);
        if (background)
        {
            // mirrors are current and the operands final from here on (main stream); the kernels run on the background stream
            pg_event_record(B.ev_bg_fork, B.stream);
            pg_stream_wait(ms, B.ev_bg_fork);
        }
        if (gs && gd && B.opt_two_streams && !background)
            pg_event_record(B.ev_fork, B.stream); // mirrors are current from here on
        if (gs)
        {
            join_records(ms); // operands and destinations of the LDS kernel are sparse records
            LaunchTimer lt(4, ms);
            // columns per wavefront: 1 unless the grid would exceed 2^20 workgroups (more parallel waves beat fewer launches:
            // measured 176 ms vs 181 ms per factorisation of the bench matrix with an 8k-workgroup target)
            int cpw = 1;
            while (cpw < 64 && gs * (size_t)((nb + SSSSM_WAVES * cpw - 1) / (SSSSM_WAVES * cpw)) > ((size_t)1 << 20))
                cpw *= 2;
            int colblocks = (nb + SSSSM_WAVES * cpw - 1) / (SSSSM_WAVES * cpw);
            size_t lds = sizeof(val_t) * (size_t)nb * SSSSM_WAVES;
            if (B.opt_getrf_strict)
                PG_LAUNCH(ssssm_sparse_kernel<true>, dim3((unsigned)(gs * colblocks)), dim3(SSSSM_WAVES * 64), lds, ms,
                                   d_groups_s, d_tasks_s, nb, cpw, B.d_flops + 4);
            else
                PG_LAUNCH(ssssm_sparse_kernel<false>, dim3((unsigned)(gs * colblocks)), dim3(SSSSM_WAVES * 64), lds, ms,
                                   d_groups_s, d_tasks_s, nb, cpw, B.d_flops + 4);
            B.stats.launches[4]++;
            B.stats.tasks[4] += ns;
            B.stats.alg_bytes[4] += bytes_s;
        }
#if defined(PG_DENSE_UPDATES)
        if (gd)
        {
            hipStream_t ds = ms;
            const bool side = B.opt_two_streams && gs && !background;
            if (side)
            {
                // fork: the MFMA kernel starts as soon as the mirrors are ready and runs beside the LDS kernel (both
                // are bound by memory latency and launch tails, not by a shared resource)
                ds = B.stream2;
                pg_stream_wait(ds, B.ev_fork);
            }
            {
                // one workgroup per (group, tile) some update of the group can reach.  Pairs whose whole queue is dense-front
                // products (every 16 x 16 piece of every operand live, no K-split) go to the front kernel's list
                int tiles = nb / DG_TILE;
                size_t nw = 0, nf = 0, nfm = 0;
                const bool front_on = B.opt_front_stages >= 1 && (nb == 128 || nb == 256) && (B.opt_front_stages >= 2 || B.opt_tiles_stages >= 1);
                // (first pass: which pairs qualify, and how many -- a front launch of its own pays from a few thousand workgroups
                //  on: fem27(112) 883.8 ms with it against 892.1 with the pairs inside the general launch, shell(398) 39.2 against 38.5)
                static std::vector<unsigned char> full_g;
                full_g.assign(gd, 0);
                size_t nfull = 0;
                for (size_t gi = 0; gi < gd && front_on; gi++)
                {
                    const SsssmGroupD &Gd = groups_d[gi];
                    unsigned all_full = Gd.slab_mask ? 0u : 0xFu;
                    for (u32 t = Gd.task_begin; t < Gd.task_end && all_full; t++)
                        all_full &= full_t[t];
                    all_full &= Gd.live_tiles;
                    full_g[gi] = (unsigned char)all_full;
                    nfull += (size_t)__builtin_popcount(all_full);
                }
                const bool own_launch = B.opt_front_stages >= 2 && (B.opt_tiles_stages < 1 || nfull >= (size_t)B.opt_front_min_wgs);
                // Longest queues first (PANGULU_HIP_HEAVY_FIRST): a launch ends with its last workgroup, and a queue of 128 live slab
                // steps that starts when the others are done is a tail of its own length.  Classes by the live steps of a group's
                // busiest tile -- sixteen of 16 steps each (2, the default), or four (1) --, the scheduler's order kept inside a class
                // (neighbours share operands: L2).  fem27(112), one box: 810.3-811.6 / 813.3 / 815.4 ms with 2 / 1 / 0
                // (profiles/r03ak_heavy_first.log); shell(398) indifferent.
                static const int heavy_mode = getenv("PANGULU_HIP_HEAVY_FIRST") ? atoi(getenv("PANGULU_HIP_HEAVY_FIRST")) : 2;
                static const bool heavy_first = heavy_mode != 0;
                static std::vector<u32> g_order;
                g_order.resize(gd);
                if (heavy_first && gd > 1)
                {
                    static std::vector<unsigned char> g_class;
                    g_class.resize(gd);
                    size_t count[16] = {0};
                    for (size_t gi = 0; gi < gd; gi++)
                    {
                        const SsssmGroupD &Gd = groups_d[gi];
                        const unsigned kmask = Gd.slab_mask ? Gd.slab_mask : 0xFFFFu;
                        unsigned steps[4] = {0, 0, 0, 0};
                        for (u32 t = Gd.task_begin; t < Gd.task_end; t++)
                            for (int tl = 0; tl < tiles * tiles; tl++)
                                steps[tl] += (unsigned)__builtin_popcount(live_k[(size_t)t * 4 + tl] & kmask);
                        const unsigned most = std::max(std::max(steps[0], steps[1]), std::max(steps[2], steps[3]));
                        if (heavy_mode == 2)
                            g_class[gi] = (unsigned char)(15 - std::min(15u, most / 16u));
                        else
                            g_class[gi] = most >= 96 ? 0 : most >= 48 ? 1 : most >= 24 ? 2 : 3;
                        count[g_class[gi]]++;
                    }
                    size_t at[16];
                    at[0] = 0;
                    for (int c = 1; c < 16; c++)
                        at[c] = at[c - 1] + count[c - 1];
                    for (size_t gi = 0; gi < gd; gi++)
                        g_order[at[g_class[gi]]++] = (u32)gi;
                }
                else
                    for (size_t gi = 0; gi < gd; gi++)
                        g_order[gi] = (u32)gi;
                for (size_t go = 0; go < gd; go++)
                {
                    const size_t gi = g_order[go];
                    const SsssmGroupD &Gd = groups_d[gi];
                    const unsigned all_full = full_g[gi];
                    for (int tl = 0; tl < tiles * tiles; tl++)
                        if ((Gd.live_tiles >> tl) & 1u)
                        {
                            SsssmWorkD item{Gd.cdense, Gd.task_begin, Gd.task_end, Gd.atomic, Gd.slab_mask, (u32)tl, 0u};
                            if (!((all_full >> tl) & 1u))
                                work[nw++] = item;
                            else if (own_launch)
                                work_f[nf++] = item;
                            else
                            {
                                // same launch as the partly filled tiles: one launch, one tail; the kernel skips the step list
                                item.pad_ = 1u;
                                work[nw++] = item;
                                nfm++;
                            }
                        }
                }
                LaunchTimer lt(5, ds);
                if (B.opt_profile)
                {
                    unsigned long long steps = 0;
                    for (size_t gi = 0; gi < gd; gi++)
                    {
                        const SsssmGroupD &Gd = groups_d[gi];
                        const unsigned kmask = Gd.slab_mask ? Gd.slab_mask : 0xFFFFu;
                        for (u32 t = Gd.task_begin; t < Gd.task_end; t++)
                            for (int tl = 0; tl < tiles * tiles; tl++)
                                steps += (unsigned long long)__builtin_popcount(live_k[(size_t)t * 4 + tl] & kmask);
                    }
                    lt.tag[0] = nw + nf;
                    lt.tag[1] = nd;
                    lt.tag[2] = steps;
                }
                B.front_workgroups += nf + nfm;
                B.general_workgroups += nw - nfm;
                static const bool debug_ssssm = getenv("PANGULU_HIP_DEBUG_SSSSM") != nullptr; // (stamps share the GETRF debug slots)
                unsigned long long *pc = B.opt_count_flops ? B.d_flops + 6 : nullptr;
                unsigned long long *pc_front = B.opt_count_flops ? B.d_flops + 7 : nullptr; // (the dense-front kernel's products on their own)
                // PANGULU_HIP_FRONT_FORK (on by default since round 6, see Backend::opt_front_fork): the two launches of a call on two streams, so that workgroups of both are resident at once
                // (one bound by the matrix pipes, the other by its per-step latencies) instead of one launch behind the other's tail.
                const bool fork_front = nf && nw && B.opt_front_fork && !B.opt_profile;
                hipStream_t fs = ds;
                if (fork_front)
                {
                    fs = B.stream_front;
                    pg_event_record(B.ev_front_fork, ds);
                    pg_stream_wait(fs, B.ev_front_fork);
                }
                if (nf)
                {
                    // the longest-running workgroups first: the front launch, then the general one fills in behind it
                    LaunchTimer lk(100, fs);
                    const unsigned unit = (unsigned)(tiles * tiles) * (unsigned)std::max<long long>(1, B.opt_front_unit);
                    if (B.opt_front_stages >= 4)
                        PG_LAUNCH((ssssm_front_f64_kernel<4, true>), dim3((unsigned)nf), dim3(FR_THREADS), 0, fs, d_tasks_d, nb, d_work_f, pc_front, unit);
                    else if (B.opt_front_stages == 3)
                        PG_LAUNCH((ssssm_front_f64_kernel<3, true>), dim3((unsigned)nf), dim3(FR_THREADS), 0, fs, d_tasks_d, nb, d_work_f, pc_front, unit);
                    else
                        PG_LAUNCH((ssssm_front_f64_kernel<2, true>), dim3((unsigned)nf), dim3(FR_THREADS), 0, fs, d_tasks_d, nb, d_work_f, pc_front, unit);
                }
                if (fork_front)
                    pg_event_record(B.ev_front_join, fs);
                LaunchTimer lk_general(nw ? 101 : 102, ds); // (102: no general launch in this call -- an empty pair, dropped at harvest)
                if (nw && B.opt_tiles_stages >= 1)
                {
                    // the general MFMA update kernel: LDS-DMA pipeline of two stages, strided piece ownership, step records a step ahead,
                    // DMA issue behind the first products (pg_hip_front.h).  (Round 3's ssssm_tiles_f64_kernel<STAGES> and round 5's
                    // piece-indexed ssssm_tilesp_f64_kernel -- option values 1, 3, 4, 5 -- live in tools/experiments/ since round 6:
                    // neither won an in-situ comparison, profiles/r03*, r05[a-h]_pieces_*.)
                    const unsigned unit = (unsigned)(tiles * tiles) * (unsigned)std::max<long long>(1, B.opt_tiles_unit);
                    PG_LAUNCH(ssssm_tilesv_f64_kernel, dim3((unsigned)nw), dim3(FR_THREADS), 0, ds, d_tasks_d, nb, d_work, pc, unit);
                }
                else if (nw)
                    PG_LAUNCH(ssssm_dense_f64_kernel, dim3((unsigned)nw), dim3(DG_THREADS), 0, ds, d_tasks_d, nb, pc,
                                       debug_ssssm ? B.d_flops + 8 : nullptr, d_work);
                if (fork_front)
                    pg_stream_wait(ds, B.ev_front_join);
            }
            if (B.opt_count_flops)
                PG_LAUNCH(ssssm_flop_count_kernel, dim3((unsigned)nd), dim3(256), 0, ds, d_tasks_d, nb, B.d_flops + 5);
            if (side)
            {
                pg_event_record(B.ev_join, ds);
                pg_stream_wait(B.stream, B.ev_join); // join before anything later on the main stream
            }
            B.stats.launches[5]++;
            B.stats.tasks[5] += nd_updates;
            B.stats.alg_bytes[5] += bytes_d;
        }
#endif
        HIP_CHECK(hipGetLastError());
        release_pending_segments(ms);
        if (background && i < n)
            for (size_t t = chunk_begin; t < i; t++)
                bg_earlier.insert(block_key_any(list[t]->opdst));
    }
    if (background)
    {
        pg_event_record(B.ev_bg_done, ms);
        B.bg_active = true;
        for (size_t t = 0; t < n; t++)
            B.bg_tiles.insert(block_key_any(list[t]->opdst));
    }
}
