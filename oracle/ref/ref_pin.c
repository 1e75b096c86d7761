/*
 * ref_pin.c -- TEST INFRASTRUCTURE.  Thin call-through into REFERENCE translation units compiled from where they lie
 * under /root/reference (never copied), so that the oracle's and the product's INTEGER outputs can be pinned bit-exactly
 * against the reference itself:
 *
 *   reference TU (compiled with the reference's own -DPANGULU_PLATFORM_ENV switch, src/pangulu_common.h:81-103,
 *   which leaves out mpi.h / cblas.h / metis.h)                        what is pinned
 *   ---------------------------------------------------------------   ---------------------------------------------
 *   src/pangulu_kernel_interface.c:4-176  (-DPANGULU_PERF)            structural flop counters of the four kernels
 *   src/pangulu_symbolic.c:3-277                                      A+A^T and the symbolic fill pattern / nnz
 *   src/pangulu_task.c:179-472                                        priority heap (strategy 0) pop order
 *   src/pangulu_memory.c, src/pangulu_thread.c                        what the above call
 *
 * Only the functions reachable from the pg_ref_* entry points below are kept (-ffunction-sections + --gc-sections +
 * version script): everything in those files that would need MPI, CBLAS or the generated platform helper is
 * discarded by the linker, nothing is stubbed.  The one definition this file adds is the storage of `global_stat`,
 * which the reference defines in src/pangulu.c (a file that needs mpi.h+cblas.h and is not compiled here); the type
 * comes from the reference's own header.
 *
 * The floating-point kernels (src/platforms/.../pangulu_platform_0100000.c) include cblas.h unconditionally and the
 * image has none: they stay unbuildable, and the FP parity of the oracle stays "pinned by known answers + residual
 * criteria" (oracle/pangulu_oracle.c header).
 */
#include "pangulu_common.h" /* the reference's, via -I/root/reference/src */

pangulu_stat_t global_stat; /* src/pangulu.c defines it in a full build */

/* ---- flop counters: src/pangulu_kernel_interface.c:4-176 ----------------------------------------------------------
 * `slot` pointers are pangulu_storage_slot_t as the reference lays it out WITHOUT GPU_OPEN (96 bytes); the repo's
 * 144-byte GPU_OPEN layout has the same first 96 bytes, so tests pass their slots as they are. */
void pangulu_getrf_flop(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, int tid);
void pangulu_tstrf_flop(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *opdiag, int tid);
void pangulu_gessm_flop(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *opdiag, int tid);
void pangulu_ssssm_flop(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *op1, pangulu_storage_slot_t *op2, int tid);

long long pg_ref_task_flop(int kernel_id, int nb, void *opdst, void *op1, void *op2)
{
    global_stat.flop = 0;
    switch (kernel_id)
    {
    case PANGULU_TASK_GETRF:
        pangulu_getrf_flop((pangulu_inblock_idx)nb, (pangulu_storage_slot_t *)opdst, 0);
        break;
    case PANGULU_TASK_TSTRF:
        pangulu_tstrf_flop((pangulu_inblock_idx)nb, (pangulu_storage_slot_t *)opdst, (pangulu_storage_slot_t *)op1, 0);
        break;
    case PANGULU_TASK_GESSM:
        pangulu_gessm_flop((pangulu_inblock_idx)nb, (pangulu_storage_slot_t *)opdst, (pangulu_storage_slot_t *)op1, 0);
        break;
    case PANGULU_TASK_SSSSM:
        pangulu_ssssm_flop((pangulu_inblock_idx)nb, (pangulu_storage_slot_t *)opdst, (pangulu_storage_slot_t *)op1, (pangulu_storage_slot_t *)op2, 0);
        break;
    default:
        return -1;
    }
    return global_stat.flop;
}

int pg_ref_sizeof_slot(void) { return (int)sizeof(pangulu_storage_slot_t); }

/* ---- symbolic factorisation: src/pangulu_symbolic.c:249-277 (pangulu_symbolic) --------------------------------------
 * in : CSC pattern of the (already reordered) matrix;  out: malloc'ed column pointer / row index of the lower fill
 * pattern including the diagonal, and the reference's symbolic_nnz.  Caller frees with pg_ref_free. */
void pangulu_symbolic(pangulu_block_common *block_common, pangulu_block_smatrix *block_smatrix, pangulu_origin_smatrix *reorder_matrix);

int pg_ref_symbolic(unsigned int n, unsigned long long nnz, unsigned long long *colptr, unsigned int *rowidx, int nb,
                    unsigned long long **out_ptr, unsigned int **out_idx, unsigned long long *out_symbolic_nnz)
{
    pangulu_block_common bc;
    pangulu_block_smatrix bs;
    pangulu_origin_smatrix A;
    memset(&bc, 0, sizeof(bc));
    memset(&bs, 0, sizeof(bs));
    memset(&A, 0, sizeof(A));
    bc.n = n;
    bc.nb = (pangulu_inblock_idx)nb;
    bc.block_length = (n + nb - 1) / nb;
    A.row = n;
    A.column = n;
    A.nnz = nnz;
    A.columnpointer = colptr;
    A.rowindex = rowidx;
    pangulu_symbolic(&bc, &bs, &A);
    *out_ptr = bs.symbolic_rowpointer;
    *out_idx = bs.symbolic_columnindex;
    *out_symbolic_nnz = bs.symbolic_nnz;
    return 0;
}

void pg_ref_free(void *p) { pangulu_free(__FILE__, __LINE__, p); }

/* ---- priority heap: src/pangulu_task.c:204-472 ----------------------------------------------------------------------
 * script[i] >= 0: push tasks[script[i]];  script[i] == -1: pop.  Returns the popped tasks in order (out, count).
 * Panel tasks only (kernel ids 1..3): an SSSSM push also feeds the per-tile aggregator, which belongs to the numeric
 * loop and needs the storage module. */
void pangulu_task_queue_init(pangulu_task_queue_t *heap, pangulu_int64_t capacity);
void pangulu_task_queue_push(pangulu_task_queue_t *heap, pangulu_int64_t row, pangulu_int64_t col, pangulu_int64_t task_level,
                             pangulu_int64_t kernel_id, pangulu_int64_t compare_flag, pangulu_storage_slot_t *opdst,
                             pangulu_storage_slot_t *op1, pangulu_storage_slot_t *op2, pangulu_int64_t block_length, const char *file, int line);
pangulu_task_t pangulu_task_queue_pop(pangulu_task_queue_t *heap);

long long pg_ref_heap_script(long long nscript, const long long *script, const pangulu_task_t *tasks, long long capacity, pangulu_task_t *out)
{
    pangulu_task_queue_t *heap = (pangulu_task_queue_t *)pangulu_malloc(__FILE__, __LINE__, sizeof(pangulu_task_queue_t));
    pangulu_task_queue_init(heap, capacity);
    long long npop = 0;
    for (long long i = 0; i < nscript; i++)
    {
        if (script[i] >= 0)
        {
            const pangulu_task_t *t = &tasks[script[i]];
            if (t->kernel_id == PANGULU_TASK_SSSSM)
                return -1;
            pangulu_task_queue_push(heap, t->row, t->col, t->task_level, t->kernel_id, t->compare_flag, t->opdst, t->op1, t->op2, 0, __FILE__, __LINE__);
        }
        else
        {
            if (heap->length == 0)
                return -2;
            out[npop++] = pangulu_task_queue_pop(heap);
        }
    }
    return npop;
}

int pg_ref_sizeof_task(void) { return (int)sizeof(pangulu_task_t); }
