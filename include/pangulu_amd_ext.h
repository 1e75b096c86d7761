/*
 * pangulu_amd_ext.h -- additions of the MI355X-native build around the reference API (include/pangulu.h).
 *
 * Nothing here exists in the reference; these calls replace what the reference gets from MPI_COMM_WORLD
 * (rank/size/transport), expose what it only prints under -DPANGULU_PERF (structural flop count, phase times,
 * src/pangulu.c:241-262), and give tests and bench.py access to the factor blocks and to the roofline model
 * of SURVEY.md §8d.
 */
#ifndef PANGULU_AMD_EXT_H
#define PANGULU_AMD_EXT_H

#include "pangulu.h"

#ifdef __cplusplus
extern "C"
{
#endif

    /* ---- process group ---------------------------------------------------------------------------- */
#define PANGULU_AMD_TRANSPORT_HOST 0 /* block records staged through host memory over TCP (127.0.0.1) */
#define PANGULU_AMD_TRANSPORT_RCCL 1 /* device-to-device ncclSend/ncclRecv over xGMI, TCP control plane */
#define PANGULU_AMD_TRANSPORT_IPC 2  /* one node: the consumer pulls the record out of the owner's HBM arena (HIP IPC mapping,
                                       * one peer copy over the pair's xGMI link), TCP control plane */
    /* One process per GPU.  `base_port + rank` is the TCP port this rank listens on for the control plane
     * (headers, barriers, broadcasts).  For RCCL, `nccl_unique_id` is the 128-byte ncclUniqueId rank 0 made
     * (pangulu_amd_rccl_unique_id) and the launcher distributed.  Returns 0 on success. */
    int pangulu_amd_comm_init(int rank, int size, const char *addr, int base_port, int transport,
                              const void *nccl_unique_id);
    int pangulu_amd_rccl_unique_id(void *out128);
    void pangulu_amd_comm_barrier(void);
    void pangulu_amd_comm_allreduce_max_f64(double *values, int count);
    void pangulu_amd_comm_finalize(void);
    /* transport in effect (RCCL falls back to HOST on all ranks together when its self-test fails) */
    int pangulu_amd_comm_transport(void);
    /* ranks whose RCCL communicators were created and passed the start-up self-test (a 1 MiB pattern over every directed
     * pair); 0 when the data plane in effect is not RCCL */
    int pangulu_amd_comm_rccl_ranks(void);
    int pangulu_amd_comm_rank(void);
    int pangulu_amd_comm_size(void);

    /* ---- back-end selection ------------------------------------------------------------------------- */
    /* The product always runs on the built-in HIP platform (PANGULU_PLATFORM_GPU_HIP), bound statically, and aborts
     * when no device is present; this library has no platform loader.  (The checker's test build,
     * oracle/_build/libpangulu_amd_test_*.so, adds one: oracle/pangulu_amd_test_hooks.h.) */
    /* (re)binds the built-in HIP platform; a no-op in this library, kept for callers written against the test build */
    void pangulu_amd_use_builtin_platform(void);
    unsigned int pangulu_amd_active_platform(void);

    /* ---- analysis options (set before pangulu_init) -------------------------------------------------- */
#define PANGULU_AMD_ORDER_IDENTITY 0 /* what the reference does without METIS/MC64 */
#define PANGULU_AMD_ORDER_ND 1       /* built-in nested dissection on the graph of A+A^T (default)   */
#define PANGULU_AMD_ORDER_USER 2     /* permutation supplied with pangulu_amd_set_user_perm           */
    void pangulu_amd_set_ordering(int kind);
    /* perm[new] = old, length n; copied */
    void pangulu_amd_set_user_perm(const sparse_index_t *perm, sparse_index_t n);
    /* optional vertex coordinates (dim = 2 or 3, n*dim doubles, vertex-major) turn ORDER_ND into a geometric
     * dissection; copied.  What belongs to ONE matrix is one-shot: the next pangulu_init consumes the user permutation and
     * the coordinates (ORDER_USER then falls back to ORDER_ND).  The choices -- ordering kind, scaling, eager host mirror --
     * stay as set until changed; pangulu_amd_reset_options() restores all defaults */
    void pangulu_amd_set_coordinates(const double *xyz, sparse_index_t n, int dim);
    /* 0 (default): device-resident numeric phase, factors downloaded once when gstrs / block export needs
     * them.  1: reference behaviour, every finished panel block is copied back to the host at once. */
    void pangulu_amd_set_eager_host_mirror(int on);
    /* 1: before the fill-reducing ordering, permute columns and scale rows and columns so that the diagonal entries have
     * modulus 1 and no entry is larger (maximum-product matching; what the reference does when built with its MC64 port,
     * src/pangulu_reordering.c:1150-1173) -- for matrices with small or zero diagonal entries (KKT systems), which a
     * factorisation without pivoting cannot take as they are.  pangulu_gstrs applies the scalings and the permutation
     * to b and x.  Default 0 (the reference's default build has no MC64 either); environment PANGULU_AMD_SCALING
     * overrides.  A structurally singular input leaves the matrix unscaled, with a warning. */
    void pangulu_amd_set_scaling(int on);
    void pangulu_amd_reset_options(void);

    /* ---- introspection -------------------------------------------------------------------------------- */
    typedef struct pangulu_amd_info_t
    {
        unsigned long long n, nnz, nb, block_length;
        unsigned long long n_padded;          /* n plus the padding rows of a block-aligned ordering (== n otherwise)  */
        unsigned long long symbolic_nnz;      /* nnz(L+U) incl. diagonal once, as src/pangulu_symbolic.c:242          */
        long long flop;                       /* structural flop count F = sum_k (c_k + 2 c_k^2), SURVEY.md §8a a9     */
        unsigned long long nblocks_nondiag;   /* non-empty off-diagonal blocks, whole matrix                          */
        unsigned long long nblocks_owned;     /* records this rank owns (diagonal halves count once each)             */
        unsigned long long ntask_getrf, ntask_tstrf, ntask_gessm, ntask_ssssm; /* this rank                           */
        unsigned long long owned_bytes;       /* bytes of block records this rank owns                                */
        unsigned long long recv_blocks;       /* remote blocks this rank receives during gstrf                        */
        unsigned long long sent_bytes, recv_bytes;
        double time_reorder, time_symbolic, time_preprocess, time_numeric, time_solve;
        double time_numeric_host_sched;       /* seconds the compute thread spent outside platform calls              */
        /* roofline model of SURVEY.md §8d for this rank's tasks: sum over tasks of algorithmic bytes / flops,
         * split by which bound is larger per task at (hbm_gbs, fp_tflops) given to pangulu_amd_model_roofline */
        double model_bytes_total, model_flop_total;
        double model_tmin_hbm_bound, model_tmin_fp_bound; /* seconds */
        unsigned long long batches;           /* platform hybrid_batched calls issued by the scheduler                */
        /* checker's build with task sampling on (oracle/pangulu_amd_test_hooks.h); 0 in the product */
        double sampled_flop;                  /* structural flops of the tasks that were executed                      */
        unsigned long long sampled_tasks;
        double time_numeric_platform;         /* seconds inside the platform's hybrid_batched calls (compute thread or launcher) */
        unsigned long long replayed;          /* 1: the last pangulu_gstrf replayed the handle's recorded launch schedule        */
        double time_schedule_record;          /* seconds pangulu_init spent recording the launch schedule (dry run of the scheduler) */
        /* Structure-only model of the WHOLE factorisation for this handle's rank count, evaluated by every rank at
         * pangulu_init from the replicated symbolic pattern (HBM 8 TB/s, 78.6 TFLOP/s, 153 GB/s per xGMI link unless
         * PANGULU_AMD_MODEL_HBM_GBS / _FP_TFLOPS / _LINK_GBS say otherwise):
         *   T*_r = sum over the tasks rank r runs of max(bytes_t / BW, flop_t / P);  comm_r = max over peers of the bytes
         *   rank r sends to that peer / link rate (one xGMI link per pair);  T*(N) = max_r (T*_r + comm_r). */
        double model_ranks_tstar_max;         /* T*(N), seconds                                                         */
        double model_ranks_tstar_sum;         /* sum_r T*_r (= the single-rank T*)                                      */
        double model_ranks_tstar_hbm, model_ranks_tstar_fp; /* the sum split into HBM-bound and MFMA-bound tasks         */
        double model_ranks_bytes_total;       /* algorithmic bytes of all tasks                                         */
        double model_rank_flop_share;         /* max_r flop_r / mean_r flop_r (1 = perfectly balanced)                  */
        double model_rank_time_share;         /* max_r T*_r / mean_r T*_r                                               */
        double model_comm_seconds_max;        /* max_r comm_r                                                           */
        double model_sent_bytes_total;        /* bytes of block records forwarded between ranks                         */
        double model_critical_path;           /* longest dependent chain of tasks, each at its own T*_t, seconds        */
        unsigned long long model_critical_path_tasks; /* ... and its length in tasks                                    */
        unsigned long long snapshot_device_bytes;     /* bytes of device memory pangulu_amd_snapshot holds (0: none, or kept on the host) */
        /* (round 5) the same chain with every task at max(T*_t, the measured floor of a lone launch of its class) and a hop
         * (PANGULU_AMD_MODEL_HOP_US + record bytes over one link) wherever an operand comes from another rank: what bounds
         * strong scaling; seconds */
        double model_critical_path_latency;
        /* HBM demand of the fullest rank under the mapping, bytes: all of it, the records it owns, the records it receives
         * (bins provisioned for every received block once), dense mirrors of both kinds from the dense threshold on */
        double model_rank_hbm_bytes_max, model_rank_hbm_records, model_rank_hbm_received, model_rank_hbm_mirrors;
        /* columns of the input that had no stored diagonal entry and got one (1e-8) by the reference's zero-diagonal rule
         * (src/pangulu_reordering.c:715-796; nested-dissection path; PANGULU_AMD_ZERO_DIAGONAL) */
        unsigned long long inserted_diagonals;
        /* (round 5) times a look-ahead call left a destination's queued updates alone because the queue was shallower than
         * PANGULU_AMD_LOOKAHEAD_MIN_QUEUE and not yet complete -- in the run that SCHEDULED: the last pangulu_gstrf, or the dry run
         * at pangulu_init whose schedule it replayed */
        unsigned long long deferred_queues;
    } pangulu_amd_info_t;
    void pangulu_amd_get_info(void **pangulu_handle, pangulu_amd_info_t *out);
    /* evaluate T* = sum_t max(bytes_t / BW, flop_t / P) over this rank's task list (structure only) */
    void pangulu_amd_model_roofline(void **pangulu_handle, double hbm_gbytes_per_s, double fp_tflops);
    /* rank that owns block (brow, bcol) under the handle's mapping (the reference: (brow mod p) q + (bcol mod q),
     * src/pangulu.c:83-90; here subtrees on single ranks, heavy separators block-cyclic over their rank group); -1 outside
     * the block grid */
    int pangulu_amd_block_owner(void **pangulu_handle, sparse_index_t brow, sparse_index_t bcol);
    /* per-rank figures of the structure-only model above: arrays of info.nproc entries (any may be NULL);
     * returns the number of ranks */
    int pangulu_amd_rank_model(void **pangulu_handle, double *tstar_seconds, double *flop, double *comm_seconds);
    /* The structure-only model for ANY rank count on this handle (1 <= nranks <= 64): out12 =
     * { T*(N) incl. link term, sum_r T*_r, max link term, bytes sent, critical path at T*_t, latency-aware critical path, max / mean
     * flop share, max / mean T* share, HBM of the fullest rank, its records owned, its records received, its dense mirrors }
     * (seconds, bytes).  Local (no communication): the pattern and the weights are replicated.  Returns 0, or 1 when the model is
     * not available (nb > 65535).  The handle's state is the same after the call as before it, but DURING the call its mapping,
     * consumer sets and model figures are those of `nranks`: not thread-safe, exclusive with every other use of the handle
     * (pangulu_gstrf, pangulu_amd_get_info, pangulu_amd_rank_model). */
    int pangulu_amd_model_for_ranks(void **pangulu_handle, int nranks, double *out12);

    /* ---- repeated factorisations (bench.py) ---------------------------------------------------------- */
    /* gstrf overwrites the matrix with its factors.  snapshot() keeps a pristine device-side copy of this rank's
     * block records (call it after pangulu_init, before the first gstrf); reset_numeric() restores the records
     * from it with one device-to-device copy and re-arms the dependency counters, so the next pangulu_gstrf
     * factorises the same matrix again with its inputs already resident in HBM.  Both return 0 on success. */
    /* New values on the pattern the handle was initialised with (csc_value in the order of the csc_rowidx given to
     * pangulu_init, on rank 0): the block records are refilled and uploaded, ordering, symbolic factorisation, records,
     * mapping -- and, on one rank, the recorded launch schedule -- are kept, and the next pangulu_gstrf factorises the new
     * matrix (time stepping / Newton iterations on a fixed mesh).  Collective.  Returns 0 on success, 1 when the handle was
     * built with scaling on (the matching depends on the values: call pangulu_init). */
    int pangulu_amd_update_values(void **pangulu_handle, const sparse_value_t *csc_value);
    int pangulu_amd_snapshot(void **pangulu_handle);
    int pangulu_amd_reset_numeric(void **pangulu_handle);
    /* One rank: pangulu_gstrf replays the handle's recorded launch schedule (default, environment PANGULU_AMD_REPLAY) or runs
     * the scheduler beside the device like a multi-rank run has to (0).  A recording stays valid while replay is off.
     * bench.py times a few steps either way so that the N = 1 line can be compared with N > 1 like for like.  Returns the
     * previous setting. */
    int pangulu_amd_set_replay(int on);

    /* ---- factor access (tests) ------------------------------------------------------------------------ */
    /* Block records this rank owns, in storage order.  Pointers are host pointers into the record and stay
     * valid until pangulu_finalize.  After pangulu_gstrf the values are the factors (downloaded on demand).
     * For is_upper==1 diagonal halves colptr/rowidx are the CSR row pointer / column index. */
    long long pangulu_amd_owned_block_count(void **pangulu_handle);
    int pangulu_amd_owned_block(void **pangulu_handle, long long idx, sparse_index_t *brow, sparse_index_t *bcol,
                                int *is_upper, unsigned long long *nnz, const pangulu_inblock_ptr **colptr,
                                const pangulu_inblock_idx **rowidx, const calculate_type **value);
    /* the symmetric permutation used: perm[new] = old, length info.n_padded; entries >= n are padding rows */
    const sparse_index_t *pangulu_amd_get_perm(void **pangulu_handle);
    /* y = L*(U*x) with the (downloaded) factors of a single-rank run, in the permuted ordering; used for the
     * reference's factor check ||L(U.1) - A.1|| / ||A.1|| (src/pangulu_numeric.c:1082-1341) */
    int pangulu_amd_apply_lu(void **pangulu_handle, const calculate_type *x, calculate_type *y);
    /* The reference's numeric check after pangulu_gstrf, || L (U 1) - A 1 ||_2 / || A 1 ||_2 with A the reordered (and, with
     * scaling on, scaled) matrix (pangulu_numeric_check, src/pangulu_numeric.c:1082-1341), evaluated on the factors where
     * they are: on the device-resident records (no download), any number of ranks (collective: every rank calls it, every
     * rank gets the value), all four value types.  Returns 0 on success, 1 if the handle has not been factorised. */
    int pangulu_amd_factor_check(void **pangulu_handle, double *relative_error);
    /* The same criterion on `nvec` vectors -- the all-ones vector of the reference first, then nvec - 1 random +-1 vectors from a
     * seeded generator (identical on every rank) -- and the LARGEST quotient of them.  One vector probes one direction; bench.py
     * gates its line on nvec = 8 at full size.  Collective like pangulu_amd_factor_check; same return codes. */
    int pangulu_amd_factor_check_vectors(void **pangulu_handle, int nvec, unsigned long long seed, double *worst_relative_error);

#ifdef __cplusplus
}
#endif

#endif /* PANGULU_AMD_EXT_H */
