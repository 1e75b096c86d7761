// pg_hip_launch_getrf.h -- host side of the diagonal-block factorisations: kernel choice, scratch images, mirror jobs behind
// them.  Included inside the anonymous namespace of pg_hip_platform.hip.
#pragma once

// ---- GETRF -------------------------------------------------------------------------------------------------------
// `gs`: stream the factorisation kernels go to (the main stream, or a side stream that has already been made to wait
// for everything these blocks depend on; the caller joins it back)
// (rounds 1-2's kernels -- getrf_blocked / getrf_lookahead, PANGULU_HIP_GETRF_TILED=0 -- moved to tools/experiments/ in round 6:
//  the tiled kernel covers the same blocks (R64, nb % 16 == 0, nb <= 256) and won every comparison since round 2)
#define GETRF_DENSE_MAX_NB 256 // dense-mode factorisation: one 16 x 16 tile grid of at most 16 x 16 tiles

// round 5's kernel (pg_hip_getrf_pipe.h) for blocks factorised in their mirrors; PANGULU_HIP_GETRF_PIPE=0 keeps the tiled kernel
inline bool getrf_pipe_selected()
{
    static const bool on = !(getenv("PANGULU_HIP_GETRF_PIPE") && atoi(getenv("PANGULU_HIP_GETRF_PIPE")) == 0);
    return on;
}

void launch_getrf(int nb, task_t **list, size_t n, hipStream_t gs, bool defer_join)
{
    HostTimer ht(2);
    const int max_slots = 256;
    if (!B.getrf_scratch || B.nb_cfg != nb)
    {
        B.generation++; // (recorded launches point into the scratch)
        if (B.getrf_scratch)
        {
            HIP_CHECK(hipDeviceSynchronize()); // (factorisations run on side streams too)
            HIP_CHECK(hipFree(B.getrf_scratch));
        }
        HIP_CHECK(hipMalloc((void **)&B.getrf_scratch, std::max(sizeof(val_t), sizeof(double)) * (size_t)nb * nb * max_slots)); // (a slot holds a double image)
        B.getrf_scratch_slots = max_slots;
        B.nb_cfg = nb;
    }
    bool blocked_kernel = false;
#if defined(PG_DENSE_PANELS)
    blocked_kernel = !B.opt_getrf_strict && (nb % 16 == 0) && nb <= GETRF_DENSE_MAX_NB;
#endif
    size_t i = 0;
    while (i < n)
    {
        Segment seg = acquire_segment();
        size_t take = std::min(n - i, (size_t)B.getrf_scratch_slots);
        GetrfTaskD *d_tasks;
        GetrfTaskD *tasks = seg.alloc<GetrfTaskD>(take, &d_tasks);
#if defined(PG_DENSE_PANELS)
        std::vector<double *> lu_images; // dense images that will hold L\\U after this launch
        std::vector<MirrorJobD> deferred; // their sparse records are written by sparsify jobs on the records stream
        bool held = false;                // the launch waits for the next platform call (PendingGetrf)
#endif
#if defined(PG_COMPLEX_PANELS)
        // complex types: diagonal blocks that have a mirror are factorised THERE (zgetrf_planes_kernel), the others by the
        // pattern-driven kernel; PANGULU_HIP_COMPLEX_PANELS=0: all of them by the pattern-driven kernel
        static const bool zpanels_on = !(getenv("PANGULU_HIP_COMPLEX_PANELS") && atoi(getenv("PANGULU_HIP_COMPLEX_PANELS")) == 0);
        std::vector<ZGetrfTaskD> ztasks;
        std::vector<GetrfTaskD> zcount; // their pattern views, for the structural flop count
        std::vector<MirrorJobD> deferred; // sparse records of the blocks factorised in their mirrors: sparsify jobs behind the kernel
#endif
        size_t nsp = 0; // tasks of the pattern-driven / blocked launch
        double by = 0;
        for (size_t k = 0; k < take; k++)
        {
            slot_t *up, *lo;
            diag_halves(list[i + k]->opdst, &up, &lo);
            GetrfTaskD T;
            T.lcp = lo->d_columnpointer;
            T.lri = lo->d_rowindex;
            T.lval = lo->d_value;
            T.urp = up->d_rowpointer;
            T.uci = up->d_columnindex;
            T.uval = up->d_value;
            T.dense = reinterpret_cast<val_t *>(reinterpret_cast<char *>(B.getrf_scratch) + std::max(sizeof(val_t), sizeof(double)) * (size_t)k * nb * nb);
            T.preloaded = 0;
            T.defer_gather = 0;
            T.invert_tiles = 0;
#if defined(PG_COMPLEX_PANELS)
            {
                BlockState &st = block_state(lo, nb);
                double *m = (zpanels_on && !B.opt_getrf_strict && !B.opt_host_mirror && (nb == 128 || nb == 256) && dense_mode_available(nb)) ? obtain_mirror(st, nb) : nullptr;
                if (m)
                {
                    // in the mirror: bring it up to date if the record is ahead, factorise it there, and let a sparsify job write
                    // the record behind the kernel; the image serves the dense solves of this level (ztrsm_planes_kernel)
                    if (!st.mirror_current)
                    {
                        MP.to_densify.push_back(mirror_job(lo, m, nb));
                        st.mirror_current = true;
                    }
                    ztasks.push_back(ZGetrfTaskD{m});
                    zcount.push_back(T);
                    deferred.push_back(mirror_job(lo, m, nb));
                    st.sparse_current = true; // (once the deferred job has run: everything that reads the record waits for it)
                    st.lu_image = true;
                    st.lu_map = false;
                    st.image_halves = 3;
                    by += (2 * SV + 2) * ((double)host_nnz(lo, nb) + host_nnz(up, nb)) + 8.0 * (nb + 1);
                    continue;
                }
                // (no mirror to be had) updates may have accumulated in the block's mirror: the record catches up first,
                // and the mirror is stale once the block is factorised
                if (!st.sparse_current && st.mirror)
                    MP.to_sparsify.push_back(mirror_job(lo, st.mirror, nb));
                st.sparse_current = true;
                st.mirror_current = false;
                st.lu_image = false;
            }
#endif
#if defined(PG_DENSE_PANELS)
            {
                // work on the block's own mirror whenever the pool has one: it may already hold the block (updates
                // accumulated there), and the dense LU it is left with serves the dense TSTRF/GESSM of this level
                BlockState &st = block_state(lo, nb);
                if (blocked_kernel)
                {
                    double *m = dense_mode_available(nb) ? obtain_mirror(st, nb) : nullptr;
                    if (m)
                    {
                        T.dense = reinterpret_cast<val_t *>(m);
                        T.preloaded = (st.mirror_current && !st.sparse_current) ? 1u : 0u;
                        T.invert_tiles = 1;
                        if (B.opt_records_stream && nb <= 256)
                        {
                            T.defer_gather = 1;
                            MirrorJobD J = mirror_job(lo, m, nb);
                            J.diag_tiles = m + (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double);
                            deferred.push_back(J);
                        }
                        lu_images.push_back(m);
                        st.lu_image = true;
                        st.lu_map = true;
                        st.image_halves = 3;
                    }
                    else if (!st.sparse_current && st.mirror)
                    {
                        MP.to_sparsify.push_back(mirror_job(lo, st.mirror, nb));
                    }
                }
                else if (!st.sparse_current && st.mirror)
                {
                    MP.to_sparsify.push_back(mirror_job(lo, st.mirror, nb));
                }
                st.sparse_current = true;
                st.mirror_current = false;
                st.written = true;
            }
#endif
            tasks[nsp++] = T;
            by += (2 * SV + 2) * ((double)host_nnz(lo, nb) + host_nnz(up, nb)) + 8.0 * (nb + 1);
        }
        hipStream_t ks = gs;
#if defined(PG_DENSE_UPDATES)
        if (!MP.to_sparsify.empty())
        {
            flush_mirror_jobs(nb, MP.to_sparsify, false); // (main stream) these blocks must see it: stay on the main stream
            ks = B.stream;
        }
#endif
#if defined(PG_COMPLEX_PANELS)
        ZGetrfTaskD *d_ztasks = nullptr;
        GetrfTaskD *d_zcount = nullptr;
        if (!ztasks.empty())
        {
            if (!MP.to_densify.empty() || !B.opt_records_stream)
                ks = B.stream; // (mirror jobs run on the main stream: the factorisation follows them there)
            flush_early_jobs(nb);
            if (!MP.to_densify.empty())
                flush_mirror_jobs(nb, MP.to_densify, true);
            ZGetrfTaskD *hz = seg.alloc<ZGetrfTaskD>(ztasks.size(), &d_ztasks);
            GetrfTaskD *hc = seg.alloc<GetrfTaskD>(zcount.size(), &d_zcount);
            if (hc)
                memcpy(hc, zcount.data(), sizeof(GetrfTaskD) * zcount.size());
            if (!hz || !hc)
            {
                fprintf(stderr, "[PanguLU-AMD ERROR] descriptor staging segment too small\n");
                exit(EXIT_FAILURE);
            }
            memcpy(hz, ztasks.data(), sizeof(ZGetrfTaskD) * ztasks.size());
        }
#endif
        commit_segment(seg);
        // (no join with the records stream: its jobs in flight write the records of blocks that are finished, these
        // kernels touch the records of the blocks they factorise)
        {
            LaunchTimer lt(1, ks);
            bool blocked = blocked_kernel;
#if defined(PG_DENSE_PANELS)
            if (blocked)
            {
                static const bool debug_stamps = getenv("PANGULU_HIP_DEBUG_GETRF") != nullptr;
                {
                    // static tile ownership + a dedicated factorisation wavefront (pg_hip_getrf_tiled.h)
                    const size_t lds_t = gt_lds_bytes(nb);
                    static size_t t_allowed = 0;
                    if (lds_t > t_allowed)
                    {
                        HIP_CHECK(hipFuncSetAttribute((const void *)getrf_tiled_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                        t_allowed = lds_t;
                    }
                    // (held for the chase when it could serve the dense solves of its level: see PendingGetrf)
                    static const bool chase_on = getenv("PANGULU_HIP_CHASE") && atoi(getenv("PANGULU_HIP_CHASE")) != 0; // (off by default: see getrf_trsm_chase_kernel)
                    bool all_images = !lu_images.empty() && lu_images.size() == take;
                    for (size_t k = 0; k < take && all_images; k++)
                        all_images = tasks[k].invert_tiles && tasks[k].defer_gather;
                    // (near the root only: a level with many diagonal blocks is bound by throughput, and there the two-in-one launch costs
                    //  more than the chain it removes -- PANGULU_HIP_CHASE_MAX_GETRF)
                    static const size_t chase_max = getenv("PANGULU_HIP_CHASE_MAX_GETRF") ? (size_t)atol(getenv("PANGULU_HIP_CHASE_MAX_GETRF")) : 4;
                    if (chase_on && REC.mode != 0 && ks == B.stream && i == 0 && take == n && take <= chase_max && all_images && !debug_stamps && !B.opt_profile &&
                        !B.opt_host_mirror && (nb == 128 || nb == 256))
                    {
                        held = true;
                        PEND.nb = nb;
                        PEND.take = take;
                        PEND.d_tasks = d_tasks;
                        PEND.images.assign(lu_images.begin(), lu_images.end());
                        if (!B.d_progress)
                        {
                            B.generation++;
                            HIP_CHECK(hipMalloc((void **)&B.d_progress, sizeof(unsigned) * PROGRESS_WORDS));
                            HIP_CHECK(hipMemset(B.d_progress, 0, sizeof(unsigned) * PROGRESS_WORDS));
                        }
                        if (B.progress_next + take > PROGRESS_WORDS)
                            B.progress_next = 0;
                        PEND.d_progress = B.d_progress + B.progress_next;
                        B.progress_next += take;
                        unsigned long long *fc = B.d_flops + 1;
                        const unsigned ntake = (unsigned)take;
                        PEND.plain = [=]()
                        { PG_LAUNCH(getrf_tiled_f64_kernel, dim3(ntake), dim3(GT_THREADS), lds_t, ks, d_tasks, nb, fc, (unsigned long long *)nullptr); };
                    }
                    else if (getrf_pipe_selected() && all_images && (nb == 128 || nb == 256))
                    {
                        // round 5: the trailing block resident in registers, panels finished on the matrix cores (pg_hip_getrf_pipe.h)
                        const size_t lds_p = gp_lds_bytes(nb);
                        static size_t p_allowed = 0;
                        if (lds_p > p_allowed)
                        {
                            HIP_CHECK(hipFuncSetAttribute((const void *)getrf_pipe_f64_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gp_lds_bytes(256)));
                            HIP_CHECK(hipFuncSetAttribute((const void *)getrf_pipe_f64_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gp_lds_bytes(256)));
                            p_allowed = gp_lds_bytes(256);
                        }
                        if (nb == 256)
                            PG_LAUNCH(getrf_pipe_f64_kernel<16>, dim3((unsigned)take), dim3(GP_THREADS), lds_p, ks, d_tasks, nb, B.d_flops + 1,
                                      debug_stamps ? B.d_flops + 8 : nullptr);
                        else
                            PG_LAUNCH(getrf_pipe_f64_kernel<8>, dim3((unsigned)take), dim3(GP_THREADS), lds_p, ks, d_tasks, nb, B.d_flops + 1,
                                      debug_stamps ? B.d_flops + 8 : nullptr);
                    }
                    else
                        PG_LAUNCH(getrf_tiled_f64_kernel, dim3((unsigned)take), dim3(GT_THREADS), lds_t, ks, d_tasks, nb, B.d_flops + 1,
                                           debug_stamps ? B.d_flops + 8 : nullptr);
                }
            }
#endif
            if (!blocked && nsp)
            {
                size_t lds = (sizeof(val_t) * 2 + sizeof(u16) * 2) * (size_t)nb;
                PG_LAUNCH(getrf_kernel, dim3((unsigned)nsp), dim3(GETRF_THREADS), lds, ks, d_tasks, nb, B.d_flops + 1);
            }
#if defined(PG_COMPLEX_PANELS)
            if (!ztasks.empty())
            {
                const size_t lds_z = sizeof(double) * 4 * ZP_PANEL * (size_t)nb;
                static size_t z_allowed = 0;
                if (lds_z > z_allowed)
                {
                    HIP_CHECK(hipFuncSetAttribute((const void *)zgetrf_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_z));
                    z_allowed = lds_z;
                }
                PG_LAUNCH(zgetrf_planes_kernel, dim3((unsigned)ztasks.size()), dim3(ZG_THREADS), lds_z, ks, (const ZGetrfTaskD *)d_ztasks, nb);
                if (ztrsm_direct_selected() && nb == 128) // the solves of this level multiply with the inverted diagonal tiles (left behind the planes)
                    PG_LAUNCH(zdiag_tile_inverse_kernel, dim3((unsigned)(ztasks.size() * (size_t)(nb / 16))), dim3(64), 0, ks, (const ZGetrfTaskD *)d_ztasks, nb);
                if (B.opt_count_flops)
                    PG_LAUNCH(getrf_flop_count_kernel, dim3((unsigned)ztasks.size()), dim3(256), 0, ks, (const GetrfTaskD *)d_zcount, nb, B.d_flops + 1);
                B.zgetrf_tasks += ztasks.size();
            }
#endif
            HIP_CHECK(hipGetLastError());
        }
#if defined(PG_COMPLEX_PANELS)
        if (!deferred.empty())
            pg_event_record(B.ev_rec_fork, ks); // behind the factorisation
#endif
#if defined(PG_DENSE_PANELS)
        if (held)
        {
            // (everything that follows the launch follows it when it is made: PendingGetrf)
            PEND.post = [=]() mutable
            {
                if (!deferred.empty())
                    pg_event_record(B.ev_rec_fork, ks); // behind the factorisation
                release_pending_segments(ks);
                if (!deferred.empty())
                    flush_mirror_jobs(nb, deferred, false, true, nullptr, true);
                B.stats.launches[1]++;
                B.stats.tasks[1] += take;
                B.stats.alg_bytes[1] += by;
            };
            PEND.active = true;
            return; // (take == n)
        }
        if (!deferred.empty())
            pg_event_record(B.ev_rec_fork, ks); // behind the factorisation
#endif
        if (ks != B.stream)
        {
            pg_event_record(B.ev_join3, ks);
            if (defer_join && !B.opt_host_mirror)
                B.getrf_join_pending = true; // the caller makes the main stream wait once its own kernels are queued
            else
                pg_stream_wait(B.stream, B.ev_join3);
        }
        release_pending_segments(ks); // (the descriptors are read on ks, which the main stream may not have joined yet)
#if defined(PG_DENSE_PANELS) || defined(PG_COMPLEX_PANELS)
        if (!deferred.empty())
            flush_mirror_jobs(nb, deferred, false, true, nullptr, true);
#endif
        B.stats.launches[1]++;
        B.stats.tasks[1] += take;
        B.stats.alg_bytes[1] += by;
        if (B.opt_host_mirror)
            for (size_t k = 0; k < take; k++)
            {
                slot_t *up, *lo;
                diag_halves(list[i + k]->opdst, &up, &lo);
                mirror_to_host(up, nb);
                mirror_to_host(lo, nb);
            }
        i += take;
    }
}
