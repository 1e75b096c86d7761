/*
 * pangulu.h -- public solver API of the MI355X-native build.
 *
 * Same five entry points, argument meaning and option structs as the reference's include/pangulu.h:11-15 and
 * include/pangulu_interface_common.h:3-20 (implemented there in src/pangulu.c:11-345), so a user program
 * written against the reference (examples/example.c:282-300) recompiles against this header unchanged:
 *
 *     pangulu_init(n, nnz, colptr, rowidx, value, &init_options, &handle);   // reorder + symbolic + distribute
 *     pangulu_gstrf(&gstrf_options, &handle);                                // numeric LU (the hot path)
 *     pangulu_gstrs(rhs, &gstrs_options, &handle);                           // triangular solves, rhs <- x
 *     pangulu_finalize(&handle);
 *
 * Input is CSC with 64-bit column pointers and 32-bit row indices (src/pangulu_common.h:67-70); the value
 * type is fixed when the library is built (-DCALCULATE_TYPE_{R64,R32,CR64,CR32}) and checked at run time
 * through init_options.sizeof_value / is_complex_matrix exactly like src/pangulu.c:28-36.
 * Errors follow the reference: a message on stdout/stderr and exit(1).
 *
 * Multi-process runs use one process per GPU.  The reference takes rank/size from MPI_COMM_WORLD; this
 * build takes them from pangulu_amd_comm_init() (see pangulu_amd_ext.h), which bench.py / the tests call
 * with the RANK / WORLD_SIZE that torch.distributed.run exports.  Without that call the library runs as a
 * single rank.
 */
#ifndef PANGULU_H
#define PANGULU_H

#include "pangulu_platform.h"

typedef pangulu_exblock_ptr sparse_pointer_t;
typedef pangulu_exblock_idx sparse_index_t;
typedef calculate_type sparse_value_t;
typedef calculate_real_type sparse_value_real_t;

#ifdef __cplusplus
extern "C"
{
#endif

    typedef struct pangulu_init_options
    {
        int nthread;                      /* host threads for the analysis phase (0 -> 1)                       */
        int nb;                           /* block order; <= 0 -> 256                                            */
        int gpu_kernel_warp_per_block;    /* wavefronts per workgroup hint for the numeric kernels (0 -> 4)      */
        int gpu_data_move_warp_per_block; /* wavefronts per workgroup hint for densify/scatter kernels (0 -> 4)  */
        int sizeof_value;                 /* must equal sizeof(sparse_value_t) of the library                    */
        int is_complex_matrix;            /* must match the library's value type                                 */
        float mpi_recv_buffer_level;      /* scales the number of receive slots per size class                   */
    } pangulu_init_options;

    typedef struct pangulu_gstrf_options
    {
        char reserved_; /* the reference's struct is empty (a GNU C extension); one byte keeps it valid C++ */
    } pangulu_gstrf_options;

    typedef struct pangulu_gstrs_options
    {
        char reserved_;
    } pangulu_gstrs_options;

    void pangulu_init(sparse_index_t pangulu_n, sparse_pointer_t pangulu_nnz, sparse_pointer_t *csc_colptr,
                      sparse_index_t *csc_rowidx, sparse_value_t *csc_value, pangulu_init_options *init_options,
                      void **pangulu_handle);
    void pangulu_gstrf(pangulu_gstrf_options *gstrf_options, void **pangulu_handle);
    void pangulu_gstrs(sparse_value_t *rhs, pangulu_gstrs_options *gstrs_options, void **pangulu_handle);
    void pangulu_gssv(sparse_value_t *rhs, pangulu_gstrf_options *gstrf_options, pangulu_gstrs_options *gstrs_options,
                      void **pangulu_handle);
    void pangulu_finalize(void **pangulu_handle);

#ifdef __cplusplus
}
#endif

#endif /* PANGULU_H */
