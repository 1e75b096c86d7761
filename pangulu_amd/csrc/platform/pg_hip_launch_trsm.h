// pg_hip_launch_trsm.h -- host side of the TSTRF / GESSM launches (dense solves against LU images, sparse solves, complex planes,
// the GETRF -> dense-solve chase).  Included inside the anonymous namespace of pg_hip_platform.hip.
#pragma once

// ---- TSTRF / GESSM -----------------------------------------------------------------------------------------------
// complex types: TSTRF / GESSM on the matrix cores (ztrsm_direct_kernel; PANGULU_HIP_ZTRSM_DIRECT=0: the vector-unit kernel)
// round 5's kernel (pg_hip_trsm_ring.h) for the TSTRF tasks of a dense-solve launch; PANGULU_HIP_TRSM_RING=0 keeps the direct kernel
inline bool trsm_ring_selected()
{
    static const bool on = !(getenv("PANGULU_HIP_TRSM_RING") && atoi(getenv("PANGULU_HIP_TRSM_RING")) == 0);
    return on;
}

inline bool ztrsm_direct_selected()
{
    static const bool on = !(getenv("PANGULU_HIP_ZTRSM_DIRECT") && atoi(getenv("PANGULU_HIP_ZTRSM_DIRECT")) == 0);
    return on;
}

void launch_trsm(int nb, task_t **list, size_t n)
{
    HostTimer ht(1);
    size_t i = 0;
    PEND.hold = PEND.active; // (a held factorisation waits until this call knows whether its solves can chase it)
    while (i < n)
    {
        const size_t per_solve = sizeof(TrsmTaskD) + sizeof(TrsmTaskD) + 64 + 80 + 4 * sizeof(u32); // (+80: a remote-diagonal image job per task at worst)
        Segment seg = acquire_segment(std::min(n - i, launch_chunk_tasks()) * per_solve + 65536);
        size_t take = std::min(n - i, seg.cap / per_solve);
        take = std::min(take, launch_chunk_tasks());
        {
            // PANGULU_HIP_TRSM_CHUNK: solves per launch (0 = all).  The leaf levels of a large problem bring tens of thousands of
            // solves in one call, and the device sits empty while their mirror jobs and descriptors are written
            static const size_t trsm_chunk = []()
            {
                const char *e = getenv("PANGULU_HIP_TRSM_CHUNK");
                const long v = e ? atol(e) : 0;
                return v > 0 ? (size_t)v : ~(size_t)0;
            }();
            take = std::min(take, trsm_chunk);
        }
        TrsmTaskD *d_tasks, *d_ftasks;
        TrsmTaskD *tasks = seg.alloc<TrsmTaskD>(take, &d_tasks);
        TrsmTaskD *ftasks = seg.alloc<TrsmTaskD>(take, &d_ftasks); // sparse views of the dense-path tasks (flop counting)
        double by_t = 0, by_g = 0;
        size_t nt = 0, ng = 0, nsparse = 0, ndense = 0;
#if defined(PG_DENSE_PANELS)
        TrsmDenseTaskD *d_dtasks;
        TrsmDenseTaskD *dtasks = seg.alloc<TrsmDenseTaskD>(take, &d_dtasks);
        u32 *d_dwork;
        u32 *dwork = seg.alloc<u32>(take * 4, &d_dwork); // (task, 64-wide slab) of every workgroup of the dense-solve launch
        static std::vector<unsigned short> dlive;        // per dense task: which 16-wide strips of the block hold entries
        dlive.assign(take, 0);
        std::vector<slot_t *> solved_dense;
        const bool dense_ok = dense_mode_available(nb);
#endif
#if defined(PG_COMPLEX_PANELS)
        static const bool zpanels_on = !(getenv("PANGULU_HIP_COMPLEX_PANELS") && atoi(getenv("PANGULU_HIP_COMPLEX_PANELS")) == 0);
        std::vector<ZTrsmTaskD> zt; // (block, 64-wide slab) items of the solves that run on mirrors (ztrsm_planes_kernel)
        std::vector<slot_t *> solved_dense;
        const bool dense_ok = dense_mode_available(nb);
#endif
        for (size_t k = 0; k < take; k++)
        {
            if (i + k + PREFETCH_SLOTS_AHEAD < n)
                prefetch_task_slots(list[i + k + PREFETCH_SLOTS_AHEAD]);
            if (i + k + PREFETCH_DETAILS_AHEAD < n)
                prefetch_task_details(list[i + k + PREFETCH_DETAILS_AHEAD], nb);
            task_t *t = list[i + k];
            slot_t *dst = t->opdst, *diag = t->op1;
            // opdiag may be either half (…0100000.c:143-145,184-186); only the half the solve reads has to exist
            // (a rank that received a remote diagonal for its TSTRFs only may never get the L half)
            const bool want_upper = t->kernel_id == PANGULU_TASK_TSTRF;
            slot_t *half = ((diag->is_upper != 0) == want_upper) ? diag : diag->related_block;
            if (!half)
            {
                fprintf(stderr, "[PanguLU-AMD ERROR] %s on block (%u,%u): the %s half of diagonal %u is not available\n",
                        want_upper ? "TSTRF" : "GESSM", dst->brow_pos, dst->bcol_pos, want_upper ? "upper" : "lower", diag->brow_pos);
                exit(EXIT_FAILURE);
            }
            slot_t *up = half, *lo = half;
            TrsmTaskD T;
            memset(&T, 0, sizeof(T));
            u32 nnz_b = host_nnz(dst, nb);
            if (t->kernel_id == PANGULU_TASK_TSTRF)
            {
                T.vptr = dst->d_rowpointer;
                T.vidx = dst->d_columnindex;
                T.vmap = dst->d_idx_of_csc_value_for_csr;
                T.bval = dst->d_value;
                T.tptr = up->d_rowpointer;
                T.tidx = up->d_columnindex;
                T.tval = up->d_value;
                T.is_tstrf = 1;
                by_t += (2 * SV + 6) * (double)nnz_b + 4.0 * (nb + 1) + (SV + 2) * (double)host_nnz(up, nb) + 4.0 * (nb + 1);
                nt++;
            }
            else
            {
                T.vptr = dst->d_columnpointer;
                T.vidx = dst->d_rowindex;
                T.vmap = nullptr;
                T.bval = dst->d_value;
                T.tptr = lo->d_columnpointer;
                T.tidx = lo->d_rowindex;
                T.tval = lo->d_value;
                T.is_tstrf = 0;
                by_g += (2 * SV + 2) * (double)nnz_b + 4.0 * (nb + 1) + (SV + 2) * (double)host_nnz(lo, nb) + 4.0 * (nb + 1);
                ng++;
            }
            bool dense = false;
#if defined(PG_DENSE_PANELS)
            // dense path: the diagonal block left a dense LU image with inverted diagonal tiles (launch_getrf) and the
            // block being solved is well filled or already lives in its mirror
            if (dense_ok && (nb == 128 || nb == 256) && B.opt_trsm_dense_permille <= 1000)
            {
                const double *lu = lu_image_of(half);
                const bool filled = (u64)nnz_b * 1000ull >= (u64)B.opt_trsm_dense_permille * (u64)nb * (u64)nb;
                if (!lu && (filled || mirror_is_ahead(dst)))
                    lu = request_half_image(half, nb); // a diagonal block another rank factorised
                if (lu && (filled || mirror_is_ahead(dst)))
                {
                    double *bm = current_mirror(dst, nb);
                    if (bm)
                    {
                        TrsmDenseTaskD D;
                        D.b = bm;
                        D.lu = lu;
                        D.is_tstrf = T.is_tstrf;
                        D.lu_map = lu_image_has_map(half) ? 1u : 0u;
                        {
                            // strips of the solve = row slabs (TSTRF) / column slabs (GESSM) of the block
                            const BlockState *sd = MP.blocks.find(block_key(dst));
                            dlive[ndense] = (sd && sd->occ_valid) ? (T.is_tstrf ? sd->occ_rows : sd->occ_cols) : (unsigned short)0xFFFF;
                        }
                        dtasks[ndense] = D;
                        // the flop counter wants the CSC view of the block in both cases
                        T.vptr = dst->d_columnpointer;
                        T.vidx = dst->d_rowindex;
                        ftasks[ndense++] = T;
                        solved_dense.push_back(dst);
                        dense = true;
                    }
                }
            }
#endif
#if defined(PG_COMPLEX_PANELS)
            // complex types: the diagonal block was factorised in its mirror (launch_getrf) and the block being solved is well filled
            // or already lives in its mirror: solve it there, one workgroup per 64-wide slab that holds pattern entries
            if (zpanels_on && dense_ok && (nb == 128 || nb == 256) && B.opt_trsm_dense_permille <= 1000 && !B.opt_host_mirror)
            {
                const double *lu = lu_image_of(half);
                const bool filled = (u64)nnz_b * 1000ull >= (u64)B.opt_trsm_dense_permille * (u64)nb * (u64)nb;
                if (lu && (filled || mirror_is_ahead(dst)))
                {
                    double *bm = current_mirror(dst, nb);
                    if (bm)
                    {
                        const BlockState *sd = MP.blocks.find(block_key(dst));
                        const unsigned live = (sd && sd->occ_valid) ? (T.is_tstrf ? sd->occ_rows : sd->occ_cols) : 0xFFFFu;
                        for (int w = 0; w < nb / 64; w++)
                            if ((live >> (4 * w)) & 0xFu)
                                zt.push_back(ZTrsmTaskD{bm, lu, (u32)T.is_tstrf, (u32)w});
                        // the flop counter wants the CSC view of the block in both cases
                        T.vptr = dst->d_columnpointer;
                        T.vidx = dst->d_rowindex;
                        ftasks[ndense++] = T;
                        solved_dense.push_back(dst);
                        dense = true;
                    }
                }
            }
#endif
            if (!dense)
            {
                require_sparse(dst, nb); // updates may have been accumulating in the block's mirror
                tasks[nsparse++] = T;
#if defined(PG_DENSE_UPDATES)
                if (BlockState *found = MP.blocks.find(block_key(dst)))
                    found->mirror_current = false; // the sparse solve rewrites the record
                block_state(dst, nb).written = true;
#endif
            }
        }
#if defined(PG_DENSE_UPDATES)
        if (!MP.to_sparsify.empty())
            flush_mirror_jobs(nb, MP.to_sparsify, false);
        flush_early_jobs(nb);
        if (!MP.to_densify.empty())
            flush_mirror_jobs(nb, MP.to_densify, true);
#endif
#if defined(PG_DENSE_PANELS)
        if (!g_half_image_jobs.empty())
        {
            // images of remote diagonal blocks: build, then invert their diagonal tiles (main stream, before the solves)
            const size_t nj = g_half_image_jobs.size();
            HalfImageJobD *d_jobs;
            HalfImageJobD *hj = seg.alloc<HalfImageJobD>(nj, &d_jobs);
            double **d_imgs;
            double **imgs = seg.alloc<double *>(nj, &d_imgs);
            if (!hj || !imgs)
            {
                fprintf(stderr, "[PanguLU-AMD ERROR] descriptor staging segment too small\n");
                exit(EXIT_FAILURE);
            }
            for (size_t q = 0; q < nj; q++)
            {
                hj[q] = g_half_image_jobs[q];
                imgs[q] = g_half_image_jobs[q].dense;
            }
            g_half_image_jobs.clear();
            {
                LaunchTimer lt(8);
                PG_LAUNCH(half_image_kernel, dim3((unsigned)nj), dim3(1024), sizeof(u32) * (size_t)(nb + 1), B.stream, d_jobs, nb);
                PG_LAUNCH(diag_tile_inverse_kernel, dim3((unsigned)(nj * (nb / 16))), dim3(64), 0, B.stream, d_imgs, nb);
            }
            B.stats.launches[8]++;
            B.stats.tasks[8] += nj;
            B.stats.alg_bytes[8] += (double)nj * sizeof(double) * nb * nb;
            HIP_CHECK(hipGetLastError());
        }
#endif
#if defined(PG_COMPLEX_PANELS)
        ZTrsmTaskD *d_zt = nullptr;
        if (!zt.empty())
        {
            ZTrsmTaskD *hz = seg.alloc<ZTrsmTaskD>(zt.size(), &d_zt);
            if (!hz)
            {
                fprintf(stderr, "[PanguLU-AMD ERROR] descriptor staging segment too small\n");
                exit(EXIT_FAILURE);
            }
            memcpy(hz, zt.data(), sizeof(ZTrsmTaskD) * zt.size());
        }
#endif
        commit_segment(seg);
#if defined(PG_DENSE_PANELS)
        // chase: every solve of this call is a dense one against an image the held factorisation is going to leave
        static const bool direct_solves = getenv("PANGULU_HIP_TRSM_DIRECT") ? atoi(getenv("PANGULU_HIP_TRSM_DIRECT")) != 0 : true;
        bool chase = PEND.active && PEND.hold && i == 0 && take == n && ndense > 0 && nsparse == 0 && direct_solves && PEND.nb == nb;
        for (size_t t = 0; t < ndense && chase; t++)
        {
            size_t at = 0;
            while (at < PEND.images.size() && PEND.images[at] != dtasks[t].lu)
                at++;
            chase = at < PEND.images.size();
            if (chase)
                dtasks[t].progress = PEND.d_progress + at;
        }
        PEND.hold = false;
        if (!chase)
            flush_pending_getrf(); // (as it was: the factorisation, then this call's kernels)
        if (ndense && nsparse && B.opt_two_streams)
            pg_event_record(B.ev_fork, B.stream); // mirrors and sparse records are current from here on
#endif
        {
            LaunchTimer lt(nt >= ng ? 2 : 3);
            if (nt && ng)
            {
                // one launch, two classes: its duration is shared in proportion to the algorithmic bytes of each
                lt.split_cls = nt >= ng ? 3 : 2;
                lt.split_frac = (nt >= ng ? by_g : by_t) / std::max(1.0, by_t + by_g);
            }
            if (nsparse)
            {
                join_records(B.stream); // the sparse solves read the diagonal halves' records (behind the fork: the dense solves do not wait)
                int vblocks = (nb + TRSM_WAVES - 1) / TRSM_WAVES;
                size_t lds = sizeof(val_t) * (size_t)nb * TRSM_WAVES;
                PG_LAUNCH(trsm_sparse_kernel, dim3((unsigned)(nsparse * vblocks)), dim3(TRSM_WAVES * 64), lds, B.stream, d_tasks,
                                   nb, B.d_flops + 2, B.d_flops + 3);
            }
#if defined(PG_COMPLEX_PANELS)
            if (!zt.empty())
            {
                const size_t lds_z = sizeof(double) * 2 * ZP_PANEL * (size_t)nb;
                static size_t zt_allowed = 0;
                if (lds_z > zt_allowed)
                {
                    HIP_CHECK(hipFuncSetAttribute((const void *)ztrsm_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_z));
                    zt_allowed = lds_z;
                }
                // (nb = 256: sixteen solution tiles on two planes do not fit a wavefront's registers -- 1110 spilled -- and the complex
                //  types' block order is 128, DESIGN.md 7: the vector-unit kernel stays there)
                if (ztrsm_direct_selected() && nb == 128)
                    PG_LAUNCH(ztrsm_direct_kernel<8>, dim3((unsigned)zt.size()), dim3(256), 0, B.stream, (const ZTrsmTaskD *)d_zt);
                else
                    PG_LAUNCH(ztrsm_planes_kernel, dim3((unsigned)zt.size()), dim3(ZT_THREADS), lds_z, B.stream, (const ZTrsmTaskD *)d_zt, nb);
            }
#endif
#if defined(PG_DENSE_PANELS)
            if (ndense)
            {
                // the dense solves run beside the sparse ones (other blocks, same diagonal operands)
                hipStream_t ds = (B.opt_two_streams && nsparse) ? B.stream2 : B.stream;
                if (ds != B.stream)
                    pg_stream_wait(ds, B.ev_fork);
                static const bool debug_trsm = getenv("PANGULU_HIP_DEBUG_TRSM") != nullptr; // (stamps share the GETRF debug slots)
                // barrier-free kernel by default (PANGULU_HIP_TRSM_DIRECT=0: the LDS-staged one)
                static const bool direct = getenv("PANGULU_HIP_TRSM_DIRECT") ? atoi(getenv("PANGULU_HIP_TRSM_DIRECT")) != 0 : true;
                unsigned long long *dbg = debug_trsm ? B.d_flops + 8 : nullptr;
                // one workgroup per (task, 64-wide slab) that holds pattern entries
                size_t nw = 0;
                for (size_t t = 0; t < ndense; t++)
                    for (int w = 0; w < nb / 64; w++)
                        if ((dlive[t] >> (4 * w)) & 0xFu)
                            dwork[nw++] = (u32)(t << 2) | (u32)w;
                if (chase)
                {
                    // one launch: the held factorisation's workgroups first, then two (task, slab) items per workgroup
                    const size_t lds_t = gt_lds_bytes(nb);
                    static size_t c_allowed = 0;
                    if (lds_t > c_allowed)
                    {
                        HIP_CHECK(hipFuncSetAttribute((const void *)getrf_trsm_chase_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                        HIP_CHECK(hipFuncSetAttribute((const void *)getrf_trsm_chase_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                        c_allowed = lds_t;
                    }
                    PendingGetrf P = std::move(PEND);
                    PEND = PendingGetrf();
                    const unsigned ng_ = (unsigned)P.take, nwg = ng_ + (unsigned)((nw + 1) / 2);
                    const GetrfTaskD *gt_ = static_cast<const GetrfTaskD *>(P.d_tasks);
                    PG_LAUNCH(zero_words_kernel, dim3(1), dim3(256), 0, ds, P.d_progress, ng_);
                    if (nb == 256)
                        PG_LAUNCH(getrf_trsm_chase_kernel<16>, dim3(nwg), dim3(GT_THREADS), lds_t, ds, gt_, ng_, P.d_progress, B.d_flops + 1, (const TrsmDenseTaskD *)d_dtasks,
                                  (const u32 *)d_dwork, (unsigned)nw);
                    else
                        PG_LAUNCH(getrf_trsm_chase_kernel<8>, dim3(nwg), dim3(GT_THREADS), lds_t, ds, gt_, ng_, P.d_progress, B.d_flops + 1, (const TrsmDenseTaskD *)d_dtasks,
                                  (const u32 *)d_dwork, (unsigned)nw);
                    P.post();
                    B.chase_launches++;
                    B.chase_solves += ndense;
                }
                else if (!nw)
                    ;
                else if (direct && trsm_ring_selected())
                {
                    // round 5: TSTRF tasks with their factor tiles requested ahead through a ring in LDS, GESSM tasks on the direct
                    // body (pg_hip_trsm_ring.h); PANGULU_HIP_TRSM_RING=0 goes back to the direct kernel for both
                    static size_t r_allowed = 0;
                    const size_t lds_r = tr_lds_bytes(256);
                    if (r_allowed < lds_r)
                    {
                        HIP_CHECK(hipFuncSetAttribute((const void *)trsm_dense_ring_f64_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
                        HIP_CHECK(hipFuncSetAttribute((const void *)trsm_dense_ring_f64_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
                        r_allowed = lds_r;
                    }
                    if (nb == 256)
                        PG_LAUNCH(trsm_dense_ring_f64_kernel<16>, dim3((unsigned)nw), dim3(256), tr_lds_bytes(nb), ds, d_dtasks, d_dwork);
                    else
                        PG_LAUNCH(trsm_dense_ring_f64_kernel<8>, dim3((unsigned)nw), dim3(256), tr_lds_bytes(nb), ds, d_dtasks, d_dwork);
                }
                else if (direct && nb == 256)
                    PG_LAUNCH(trsm_dense_direct_f64_kernel<16>, dim3((unsigned)nw), dim3(256), 0, ds, d_dtasks, d_dwork);
                else if (direct)
                    PG_LAUNCH(trsm_dense_direct_f64_kernel<8>, dim3((unsigned)nw), dim3(256), 0, ds, d_dtasks, d_dwork);
                else if (nb == 256)
                    PG_LAUNCH(trsm_dense_f64_kernel<16>, dim3((unsigned)nw), dim3(256), 0, ds, d_dtasks, dbg, d_dwork);
                else
                    PG_LAUNCH(trsm_dense_f64_kernel<8>, dim3((unsigned)nw), dim3(256), 0, ds, d_dtasks, dbg, d_dwork);
                if (ds != B.stream)
                {
                    pg_event_record(B.ev_join, ds);
                    pg_stream_wait(B.stream, B.ev_join);
                }
            }
#endif
            HIP_CHECK(hipGetLastError());
        }
#if defined(PG_DENSE_PANELS) || defined(PG_COMPLEX_PANELS)
        if (ndense)
        {
            if (B.opt_count_flops)
                PG_LAUNCH(trsm_flop_count_kernel, dim3((unsigned)ndense), dim3(256), 0, B.stream, d_ftasks, nb, B.d_flops + 2,
                                   B.d_flops + 3);
            B.stats.trsm_dense_tasks += ndense;
        }
#endif
        release_pending_segments();
#if defined(PG_DENSE_PANELS) || defined(PG_COMPLEX_PANELS)
        // the solutions live in the mirrors: bring the sparse records (the authoritative form of a finished block) up
        // to date at once; the mirrors stay valid as MFMA operands
        for (slot_t *s : solved_dense)
        {
            BlockState &st = block_state(s, nb);
            st.mirror_current = true;
            MP.to_sparsify.push_back(mirror_job(s, st.mirror, nb));
            st.sparse_current = true;
        }
        if (!MP.to_sparsify.empty())
            flush_mirror_jobs(nb, MP.to_sparsify, false, true);
#endif
        // one launch serves both kinds: a launch of every class it carries tasks of (booked under the larger class
        // alone, GESSM's tasks, bytes and flops fell out of bench.py's `kernels`, which lists classes with launches)
        B.stats.launches[2] += nt ? 1 : 0;
        B.stats.launches[3] += ng ? 1 : 0;
        B.stats.tasks[2] += nt;
        B.stats.tasks[3] += ng;
        B.stats.alg_bytes[2] += by_t;
        B.stats.alg_bytes[3] += by_g;
        if (B.opt_host_mirror)
            for (size_t k = 0; k < take; k++)
                mirror_to_host(list[i + k]->opdst, nb);
        i += take;
    }
    PEND.hold = false;
    flush_pending_getrf(); // (nothing stays held past the call that could have used it)
}
