/*
 * TEST INFRASTRUCTURE, NOT PRODUCT API.
 *
 * oracle/_build/libpangulu_amd_test_<type>.so is the native host (scheduler, preprocessing, transports) compiled from the
 * same sources as the product with -DPANGULU_AMD_TEST_HOOKS, which adds exactly one entry point: a loader that routes
 * the 21 platform operators to another shared object.  tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it to run the scheduler on the CPU restatement of the reference's CPU platform (libpangulu_oracle_*.so, id
 * 0x0100000).  The shipped libpangulu_amd_<type>.so does not export it.
 */
#ifndef PANGULU_AMD_TEST_HOOKS_H
#define PANGULU_AMD_TEST_HOOKS_H
#ifdef __cplusplus
extern "C"
{
#endif
    /* so_path must export pangulu_platform_<7-digit id>_<name> for the 21 names of build_helper.py:8-32.  0 on success. */
    int pangulu_amd_use_platform_library(const char *so_path, unsigned int platform_id);
    /* the host's maximum-product matching + scaling (pg_scaling.cpp) on a bare CSC matrix: column matched to every row, Dr, Dc */
    int pangulu_amd_test_matching(unsigned int n, const unsigned long long *colptr, const unsigned int *rowidx, const void *value,
                                  unsigned int *col_of_row, double *dr, double *dc);
    /* bench.py's cpu_baseline leg: execute every stride-th task of each kernel class of a factorisation and only release
     * the others (a bounded sample of the same matrix / ordering / nb); pangulu_amd_info_t.sampled_flop / sampled_tasks
     * say what ran.  1 = everything (default). */
    void pangulu_amd_test_set_task_sampling(int stride);
    /* the subtree-to-rank mapping and the per-rank structure model for `size` ranks evaluated in one process: the next
     * pangulu_init must run with PANGULU_AMD_ANALYSIS_ONLY=1 (no block records, nothing to factorise); <= 1 restores
     * the single-rank world.  tests/test_mapping.py checks the per-rank flop shares with it. */
    void pangulu_amd_test_set_analysis_ranks(int size);
    /* the host's priority heap driven by a push/pop script, and its symbolic phase on a bare pattern: compared with the
     * reference's own src/pangulu_task.c / src/pangulu_symbolic.c (oracle/ref/ref_pin.c) in tests/test_reference_pin.py */
    long long pangulu_amd_test_heap_script(long long nscript, const long long *script, const void *tasks, void *out);
    int pangulu_amd_test_symbolic(unsigned int n, const unsigned long long *colptr, const unsigned int *rowidx,
                                  unsigned long long **out_ptr, unsigned int **out_idx, unsigned long long *out_symbolic_nnz, long long *out_flop);
#ifdef __cplusplus
}
#endif
#endif
