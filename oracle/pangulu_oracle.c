/*
 * pangulu_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's CPU platform (platform id 0100000, "CPU_NAIVE") for the numeric
 * factorisation hot path, written from the reference's behaviour:
 *     /root/reference/src/platforms/01_SHAREDMEM/00_CPU/000_CPU/pangulu_platform_0100000.c
 * and of its structural flop counters  /root/reference/src/pangulu_kernel_interface.c:4-176.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (libpangulu_amd_*.so) never links or calls it.
 *
 * Pinning status: the FLOATING-POINT kernels below are "parity unpinned" by reference code -- the reference's CPU
 * platform file includes cblas.h unconditionally, the image has none, and writing one would be a stand-in, so no
 * reference binary backs GETRF/TSTRF/GESSM/SSSSM.  The INTEGER outputs (the four structural flop counters at the end
 * of this file) ARE pinned bit for bit against the reference's own src/pangulu_kernel_interface.c:4-176, compiled from
 * /root/reference under its PANGULU_PLATFORM_ENV switch into oracle/_ref/ (oracle/ref/ref_pin.c,
 * tests/test_reference_pin.py).  The floating-point part is pinned against the known answers SURVEY.md §8c / BASELINE.md §2 record from the reference's
 * own run on its only fixture (examples/Trefethen_20b.mtx: symbolic nnz 285, structural flop 2491,
 * ||Ax-b||/||b|| ~ 2e-16) and on Poisson 24^3 (symbolic nnz 15 302 062, flop 8 686 870 069), and against
 * the reference's own correctness criteria (||L(U.1) - A.1|| / ||A.1||, src/pangulu_numeric.c:1082-1341,
 * and ||Ax-b||/||b||, examples/example.c:304-364) -- see tests/test_oracle_*.py.  Per-kernel bit-level
 * vectors from a reference build do not exist: for those kernels parity is "pinned by known answers +
 * residual criteria", not by golden vectors.
 *
 * The GEMM inside SSSSM lives in a third-party library in the reference (OpenBLAS 0.3.26 via cblas_?gemm,
 * call sites ...0100000.c:317-327).  Here it is the textbook column-major triple loop; when the
 * environment variable PANGULU_ORACLE_BLAS names a shared object exporting (scipy_)cblas_dgemm it is
 * dlopen'ed and used instead for R64 (that is how the CPU baseline in bench.py gets an OpenBLAS-backed
 * SSSSM like the reference's).
 *
 * Build variants: -DCALCULATE_TYPE_{R64,R32,CR64,CR32}; -DPG_ORACLE_FMA (real types only) rounds every
 * `a -= b*c` update as one fused multiply-add, which is what the GPU does, to allow bit-exact comparison of
 * GETRF/TSTRF/GESSM whose operation order is fixed.
 */
#include <complex.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pangulu_platform.h"

typedef calculate_type val_t;
typedef pangulu_storage_slot_t slot_t;

/* a -= b*c */
#if defined(PG_ORACLE_FMA) && !defined(PANGULU_COMPLEX)
#if defined(CALCULATE_TYPE_R32)
#define SUBMUL(a, b, c) ((a) = fmaf(-(b), (c), (a)))
#else
#define SUBMUL(a, b, c) ((a) = fma(-(b), (c), (a)))
#endif
#else
#define SUBMUL(a, b, c) ((a) -= (b) * (c))
#endif

/* The reference tests `fabs(pivot) < PANGULU_TOL` (...0100000.c:80,153); for _Complex operands the implicit
 * conversion to double keeps only the real part.  Reproduced as is. */
static inline int pivot_is_tiny(val_t v)
{
#ifdef PANGULU_COMPLEX
    return fabs((double)creal(v)) < PANGULU_TOL;
#else
    return fabs((double)v) < PANGULU_TOL;
#endif
}

static inline val_t clamp_pivot(val_t v)
{
    if (pivot_is_tiny(v))
    {
        return (val_t)PANGULU_TOL;
    }
    return v;
}

/* consumers of a block's colptr treat entry 0 as 0 (...0100000.c:259,267,283,304,383) */
static inline pangulu_int32_t ptr_at(const pangulu_inblock_ptr *ptr, pangulu_int32_t i)
{
    return i == 0 ? 0 : (pangulu_int32_t)ptr[i];
}

/* dst[q] -= mul * src[p] for every index common to the two sorted index ranges (both ascending).
 * Same updates, in the same ascending order, as the reference's three-loop merges
 * (...0100000.c:97-109, 119-131, 160-172, 194-206). */
static inline void merge_submul(
    const pangulu_inblock_idx *src_idx, const val_t *src_val, pangulu_int32_t p, pangulu_int32_t p_end,
    const pangulu_inblock_idx *dst_idx, val_t *dst_val, pangulu_int32_t q, pangulu_int32_t q_end,
    val_t mul)
{
    while (p < p_end && q < q_end)
    {
        pangulu_inblock_idx a = src_idx[p], b = dst_idx[q];
        if (a == b)
        {
            SUBMUL(dst_val[q], mul, src_val[p]);
            p++;
            q++;
        }
        else if (a < b)
        {
            p++;
        }
        else
        {
            q++;
        }
    }
}

static inline void split_diag(slot_t *any_half, slot_t **upper, slot_t **lower)
{
    if (any_half->is_upper)
    {
        *upper = any_half;
        *lower = any_half->related_block;
    }
    else
    {
        *upper = any_half->related_block;
        *lower = any_half;
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* runtime shims (...0100000.c:14-55): host memory                                                       */
/* ---------------------------------------------------------------------------------------------------- */
void pangulu_platform_0100000_malloc(void **platform_address, size_t size)
{
    *platform_address = malloc(size ? size : 1);
    if (!*platform_address)
    {
        fprintf(stderr, "[pangulu oracle] malloc(%zu) failed\n", size);
        exit(1);
    }
}
void pangulu_platform_0100000_malloc_pinned(void **platform_address, size_t size)
{
    pangulu_platform_0100000_malloc(platform_address, size);
}
void pangulu_platform_0100000_synchronize(void) {}
void pangulu_platform_0100000_memset(void *s, int c, size_t n) { memset(s, c, n); }
void pangulu_platform_0100000_create_stream(void **stream) { (void)stream; }
void pangulu_platform_0100000_memcpy(void *dst, const void *src, size_t count, unsigned int kind)
{
    (void)kind;
    if (dst != src)
        memcpy(dst, src, count);
}
void pangulu_platform_0100000_memcpy_async(void *dst, const void *src, size_t count, unsigned int kind, void *stream)
{
    (void)kind;
    (void)stream;
    if (dst != src)
        memcpy(dst, src, count);
}
void pangulu_platform_0100000_free(void *devptr) { free(devptr); }
void pangulu_platform_0100000_get_device_num(int *device_num) { *device_num = 1; }
void pangulu_platform_0100000_set_default_device(int device_num) { (void)device_num; }
void pangulu_platform_0100000_get_device_name(char *name, int device_num)
{
    (void)device_num;
    strcpy(name, "CPU");
}
void pangulu_platform_0100000_get_device_memory_usage(size_t *used_byte) { *used_byte = 0; }

/* ---------------------------------------------------------------------------------------------------- */
/* GETRF (...0100000.c:57-135): in-place sparse LU of one diagonal block on its fixed pattern, no pivoting */
/* ---------------------------------------------------------------------------------------------------- */
void pangulu_platform_0100000_getrf(pangulu_inblock_idx nb, slot_t *opdst, int tid)
{
    (void)tid;
    slot_t *U, *L;
    split_diag(opdst, &U, &L);
    const pangulu_inblock_ptr *urp = U->columnpointer; /* upper half is CSR: row pointer     */
    const pangulu_inblock_idx *uci = U->rowindex;      /*                    column index    */
    val_t *uv = U->value;
    const pangulu_inblock_ptr *lcp = L->columnpointer; /* lower half is strictly-lower CSC   */
    const pangulu_inblock_idx *lri = L->rowindex;
    val_t *lv = L->value;

    for (pangulu_int32_t k = 0; k < nb; k++)
    {
        if (urp[k] == urp[k + 1])
        {
            continue; /* structurally empty row: nothing to eliminate (padding rows of the last block) */
        }
        val_t pivot = clamp_pivot(uv[urp[k]]);
        /* (1) scale L(:,k) */
        for (pangulu_int32_t p = lcp[k]; p < (pangulu_int32_t)lcp[k + 1]; p++)
        {
            lv[p] /= pivot;
        }
        /* (2) U(r,:) -= L(r,k) * U(k,:) for every r in L(:,k); the merge starts at U(k,k), which never
         *     matches because row r starts at column r > k */
        for (pangulu_int32_t p = lcp[k]; p < (pangulu_int32_t)lcp[k + 1]; p++)
        {
            pangulu_int32_t r = lri[p];
            merge_submul(uci, uv, urp[k], urp[k + 1], uci, uv, urp[r], urp[r + 1], lv[p]);
        }
        /* (3) L(:,c) -= L(:,k) * U(k,c) for every c > k in U(k,:) */
        for (pangulu_int32_t p = urp[k] + 1; p < (pangulu_int32_t)urp[k + 1]; p++)
        {
            pangulu_int32_t c = uci[p];
            merge_submul(lri, lv, lcp[k], lcp[k + 1], lri, lv, lcp[c], lcp[c + 1], uv[p]);
        }
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* TSTRF (...0100000.c:137-175): B <- B * U^{-1}; B is a lower block walked row by row through its CSR view */
/* ---------------------------------------------------------------------------------------------------- */
void pangulu_platform_0100000_tstrf(pangulu_inblock_idx nb, slot_t *opdst, slot_t *opdiag, int tid)
{
    (void)tid;
    if (opdiag->is_upper == 0)
    {
        opdiag = opdiag->related_block;
    }
    const pangulu_inblock_ptr *urp = opdiag->columnpointer;
    const pangulu_inblock_idx *uci = opdiag->rowindex;
    const val_t *uv = opdiag->value;
    const pangulu_inblock_ptr *brp = opdst->rowpointer;
    const pangulu_inblock_idx *bci = opdst->columnindex;
    const pangulu_inblock_ptr *map = opdst->idx_of_csc_value_for_csr;
    val_t *bv = opdst->value;

    for (pangulu_int32_t row = 0; row < nb; row++)
    {
        pangulu_int32_t row_end = brp[row + 1];
        for (pangulu_int32_t p = brp[row]; p < row_end; p++)
        {
            pangulu_int32_t c = bci[p];
            pangulu_int32_t d = urp[c]; /* U(c,c) is the first entry of CSR row c */
            val_t x = bv[map[p]] / clamp_pivot(uv[d]);
            bv[map[p]] = x;
            /* later entries of this row: B(row,c') -= x * U(c,c') */
            pangulu_int32_t q = p + 1, u = d, u_end = urp[c + 1];
            while (q < row_end && u < u_end)
            {
                pangulu_inblock_idx a = uci[u], b = bci[q];
                if (a == b)
                {
                    SUBMUL(bv[map[q]], x, uv[u]);
                    u++;
                    q++;
                }
                else if (a < b)
                {
                    u++;
                }
                else
                {
                    q++;
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* GESSM (...0100000.c:178-209): B <- L^{-1} * B, L unit lower; B is an upper block walked column by column */
/* ---------------------------------------------------------------------------------------------------- */
void pangulu_platform_0100000_gessm(pangulu_inblock_idx nb, slot_t *opdst, slot_t *opdiag, int tid)
{
    (void)tid;
    if (opdiag->is_upper == 1)
    {
        opdiag = opdiag->related_block;
    }
    const pangulu_inblock_ptr *lcp = opdiag->columnpointer;
    const pangulu_inblock_idx *lri = opdiag->rowindex;
    const val_t *lv = opdiag->value;
    const pangulu_inblock_ptr *bcp = opdst->columnpointer;
    const pangulu_inblock_idx *bri = opdst->rowindex;
    val_t *bv = opdst->value;

    for (pangulu_int32_t col = 0; col < nb; col++)
    {
        pangulu_int32_t col_end = bcp[col + 1];
        for (pangulu_int32_t p = bcp[col]; p < col_end; p++)
        {
            pangulu_int32_t r = bri[p];
            merge_submul(lri, lv, lcp[r], lcp[r + 1], bri, bv, p + 1, col_end, bv[p]);
        }
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* SSSSM (...0100000.c:211-397): C <- C - A*B on C's pattern via gather -> GEMM -> scatter                */
/* ---------------------------------------------------------------------------------------------------- */
typedef void (*dgemm_fn)(int order, int ta, int tb, int m, int n, int k, double alpha, const double *a, int lda,
                         const double *b, int ldb, double beta, double *c, int ldc);
static dgemm_fn ext_dgemm = NULL;
static int ext_dgemm_probed = 0;

static void probe_external_blas(void)
{
    ext_dgemm_probed = 1;
#if defined(CALCULATE_TYPE_R64)
    const char *path = getenv("PANGULU_ORACLE_BLAS");
    if (!path || !*path)
    {
        return;
    }
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h)
    {
        fprintf(stderr, "[pangulu oracle] PANGULU_ORACLE_BLAS=%s could not be loaded (%s); using the built-in GEMM\n", path, dlerror());
        return;
    }
    void *sym = dlsym(h, "scipy_cblas_dgemm");
    if (!sym)
        sym = dlsym(h, "cblas_dgemm");
    if (!sym)
        sym = dlsym(h, "scipy_cblas_dgemm64_");
    ext_dgemm = (dgemm_fn)sym;
    /* one thread, like the reference's openblas_set_num_threads(1) (src/pangulu.c:117-119) */
    void (*set_threads)(int) = (void (*)(int))dlsym(h, "scipy_openblas_set_num_threads");
    if (!set_threads)
        set_threads = (void (*)(int))dlsym(h, "openblas_set_num_threads");
    if (set_threads)
        set_threads(1);
#endif
}

int pangulu_oracle_uses_external_blas(void)
{
    if (!ext_dgemm_probed)
        probe_external_blas();
    return ext_dgemm != NULL;
}

/* T(m x n) = A(m x k) * B(k x n), all column-major, leading dimensions m, k, m */
static void panel_gemm(int m, int n, int k, const val_t *A, const val_t *B, val_t *T)
{
    if (!ext_dgemm_probed)
        probe_external_blas();
#if defined(CALCULATE_TYPE_R64)
    if (ext_dgemm && m > 0 && n > 0 && k > 0)
    {
        ext_dgemm(102 /*ColMajor*/, 111 /*NoTrans*/, 111, m, n, k, 1.0, A, m, B, k, 0.0, T, m);
        return;
    }
#endif
    for (int j = 0; j < n; j++)
    {
        val_t *t = T + (size_t)j * m;
        for (int i = 0; i < m; i++)
            t[i] = 0;
        for (int l = 0; l < k; l++)
        {
            val_t b = B[(size_t)j * k + l];
            const val_t *a = A + (size_t)l * m;
            for (int i = 0; i < m; i++)
            {
                t[i] += a[i] * b;
            }
        }
    }
}

typedef struct
{
    int nb;
    pangulu_int32_t *kmap;    /* op1 column -> dense k index (non-empty columns, ascending)        */
    pangulu_int32_t *rowmap;  /* op1 row -> dense m index, first-touch order; -1 = untouched        */
    pangulu_int32_t *rowinv;  /* dense m index -> row                                               */
    pangulu_int32_t *colmap;  /* op2 column -> dense n index; -1 = empty column                     */
    pangulu_int32_t *colinv;  /* dense n index -> column                                            */
    val_t *lpanel, *upanel, *tpanel;
} ssssm_scratch_t;
static __thread ssssm_scratch_t S = {0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL};

static void ssssm_scratch_reserve(int nb)
{
    if (S.nb >= nb)
        return;
    free(S.kmap);
    free(S.rowmap);
    free(S.rowinv);
    free(S.colmap);
    free(S.colinv);
    free(S.lpanel);
    free(S.upanel);
    free(S.tpanel);
    S.nb = nb;
    S.kmap = (pangulu_int32_t *)malloc(sizeof(pangulu_int32_t) * nb);
    S.rowmap = (pangulu_int32_t *)malloc(sizeof(pangulu_int32_t) * nb);
    S.rowinv = (pangulu_int32_t *)malloc(sizeof(pangulu_int32_t) * nb);
    S.colmap = (pangulu_int32_t *)malloc(sizeof(pangulu_int32_t) * nb);
    S.colinv = (pangulu_int32_t *)malloc(sizeof(pangulu_int32_t) * nb);
    S.lpanel = (val_t *)calloc((size_t)nb * nb, sizeof(val_t));
    S.upanel = (val_t *)calloc((size_t)nb * nb, sizeof(val_t));
    S.tpanel = (val_t *)malloc(sizeof(val_t) * (size_t)nb * nb);
}

#if defined(PG_ORACLE_FMA) && !defined(PANGULU_COMPLEX)
/* FMA build only: the same update in the operation order of the GPU's sparse kernel -- every destination entry
 * takes its terms one by one, in ascending k, each as one fused multiply-add -- so that a whole factorisation on
 * the sparse path can be compared bit for bit.  (The plain build below keeps the reference's gather/GEMM/scatter
 * order; tests/test_oracle_* checks that the two agree to rounding.) */
static void ssssm_in_place_fused(pangulu_inblock_idx nb, slot_t *opdst, slot_t *op1, slot_t *op2)
{
    ssssm_scratch_reserve(nb);
    val_t *acc = S.tpanel; /* dense destination column */
    slot_t *U = NULL, *L = opdst;
    if (opdst->brow_pos == opdst->bcol_pos)
    {
        split_diag(opdst, &U, &L);
    }
    /* by-column positions of the upper (CSR) half of a diagonal destination */
    pangulu_int32_t *ucnt = S.kmap;
    for (int j = 0; j < nb; j++)
    {
        pangulu_int32_t b0 = ptr_at(op2->columnpointer, j), b1 = op2->columnpointer[j + 1];
        if (b0 == b1)
            continue;
        for (pangulu_int32_t p = ptr_at(L->columnpointer, j); p < (pangulu_int32_t)L->columnpointer[j + 1]; p++)
            acc[L->rowindex[p]] = L->value[p];
        if (U)
        {
            for (int r = 0; r <= j; r++)
            {
                ucnt[r] = -1;
                for (pangulu_int32_t p = U->columnpointer[r]; p < (pangulu_int32_t)U->columnpointer[r + 1]; p++)
                    if (U->rowindex[p] == j)
                    {
                        ucnt[r] = p;
                        acc[r] = U->value[p];
                    }
            }
        }
        for (pangulu_int32_t q = b0; q < b1; q++)
        {
            int k = op2->rowindex[q];
            val_t b = op2->value[q];
            for (pangulu_int32_t p = ptr_at(op1->columnpointer, k); p < (pangulu_int32_t)op1->columnpointer[k + 1]; p++)
                SUBMUL(acc[op1->rowindex[p]], op1->value[p], b);
        }
        for (pangulu_int32_t p = ptr_at(L->columnpointer, j); p < (pangulu_int32_t)L->columnpointer[j + 1]; p++)
            L->value[p] = acc[L->rowindex[p]];
        if (U)
        {
            for (int r = 0; r <= j; r++)
                if (ucnt[r] >= 0)
                    U->value[ucnt[r]] = acc[r];
        }
    }
}
#endif

void pangulu_platform_0100000_ssssm(pangulu_inblock_idx nb, slot_t *opdst, slot_t *op1, slot_t *op2, int tid)
{
    (void)tid;
#if defined(PG_ORACLE_FMA) && !defined(PANGULU_COMPLEX)
    ssssm_in_place_fused(nb, opdst, op1, op2);
    return;
#endif
    ssssm_scratch_reserve(nb);
    int m = 0, n = 0, k = 0;
    for (int i = 0; i < nb; i++)
    {
        S.rowmap[i] = -1;
        S.colmap[i] = -1;
        S.kmap[i] = -1;
    }
    /* k index: non-empty columns of op1, ascending (...0100000.c:257-264) */
    for (int c = 0; c < nb; c++)
    {
        if ((pangulu_int32_t)op1->columnpointer[c + 1] > ptr_at(op1->columnpointer, c))
        {
            S.kmap[c] = k++;
        }
    }
    /* m index: rows of op1 in first-touch order scanning column by column (...0100000.c:265-279) */
    for (int c = 0; c < nb; c++)
    {
        for (pangulu_int32_t p = ptr_at(op1->columnpointer, c); p < (pangulu_int32_t)op1->columnpointer[c + 1]; p++)
        {
            int r = op1->rowindex[p];
            if (S.rowmap[r] == -1)
            {
                S.rowinv[m] = r;
                S.rowmap[r] = m++;
            }
        }
    }
    /* gather op2 into a k x n panel; entries whose row meets an empty op1 column are dropped
     * (...0100000.c:281-301) */
    for (int c = 0; c < nb; c++)
    {
        pangulu_int32_t beg = ptr_at(op2->columnpointer, c), end = op2->columnpointer[c + 1];
        if (end > beg)
        {
            val_t *ucol = S.upanel + (size_t)n * k;
            for (pangulu_int32_t p = beg; p < end; p++)
            {
                int r = op2->rowindex[p];
                if (S.kmap[r] >= 0)
                {
                    ucol[S.kmap[r]] = op2->value[p];
                }
            }
            S.colinv[n] = c;
            S.colmap[c] = n++;
        }
    }
    /* gather op1 into an m x k panel (...0100000.c:302-311) */
    for (int c = 0; c < nb; c++)
    {
        if (S.kmap[c] < 0)
            continue;
        val_t *lcol = S.lpanel + (size_t)m * S.kmap[c];
        for (pangulu_int32_t p = ptr_at(op1->columnpointer, c); p < (pangulu_int32_t)op1->columnpointer[c + 1]; p++)
        {
            lcol[S.rowmap[op1->rowindex[p]]] = op1->value[p];
        }
    }

    panel_gemm(m, n, k, S.lpanel, S.upanel, S.tpanel);

    memset(S.lpanel, 0, sizeof(val_t) * (size_t)m * k);
    memset(S.upanel, 0, sizeof(val_t) * (size_t)k * n);

    if (opdst->brow_pos == opdst->bcol_pos)
    {
        /* diagonal destination: lower CSC half then upper CSR half (...0100000.c:332-375) */
        slot_t *U, *L;
        split_diag(opdst, &U, &L);
        for (int j = 0; j < n; j++)
        {
            int c = S.colinv[j];
            const val_t *t = S.tpanel + (size_t)j * m;
            for (pangulu_int32_t p = L->columnpointer[c]; p < (pangulu_int32_t)L->columnpointer[c + 1]; p++)
            {
                int mi = S.rowmap[L->rowindex[p]];
                if (mi != -1)
                {
                    L->value[p] -= t[mi];
                }
            }
        }
        for (int i = 0; i < m; i++)
        {
            int r = S.rowinv[i];
            for (pangulu_int32_t p = U->columnpointer[r]; p < (pangulu_int32_t)U->columnpointer[r + 1]; p++)
            {
                int nj = S.colmap[U->rowindex[p]];
                if (nj != -1)
                {
                    U->value[p] -= S.tpanel[(size_t)nj * m + i];
                }
            }
        }
    }
    else
    {
        /* (...0100000.c:376-396) */
        for (int j = 0; j < n; j++)
        {
            int c = S.colinv[j];
            const val_t *t = S.tpanel + (size_t)j * m;
            for (pangulu_int32_t p = ptr_at(opdst->columnpointer, c); p < (pangulu_int32_t)opdst->columnpointer[c + 1]; p++)
            {
                int mi = S.rowmap[opdst->rowindex[p]];
                if (mi != -1)
                {
                    opdst->value[p] -= t[mi];
                }
            }
        }
    }
}

/* serial loops, as the reference (...0100000.c:399-431) */
void pangulu_platform_0100000_ssssm_batched(pangulu_inblock_idx nb, pangulu_uint64_t ntask, pangulu_task_t *tasks)
{
    for (pangulu_uint64_t i = 0; i < ntask; i++)
    {
        pangulu_platform_0100000_ssssm(nb, tasks[i].opdst, tasks[i].op1, tasks[i].op2, 0);
    }
}

void pangulu_platform_0100000_hybrid_batched(pangulu_inblock_idx nb, pangulu_uint64_t ntask, pangulu_task_t *tasks)
{
    for (pangulu_uint64_t i = 0; i < ntask; i++)
    {
        pangulu_task_t *t = &tasks[i];
        switch (t->kernel_id)
        {
        case PANGULU_TASK_GETRF:
            pangulu_platform_0100000_getrf(nb, t->opdst, 0);
            break;
        case PANGULU_TASK_TSTRF:
            pangulu_platform_0100000_tstrf(nb, t->opdst, t->op1, 0);
            break;
        case PANGULU_TASK_GESSM:
            pangulu_platform_0100000_gessm(nb, t->opdst, t->op1, 0);
            break;
        case PANGULU_TASK_SSSSM:
            pangulu_platform_0100000_ssssm(nb, t->opdst, t->op1, t->op2, 0);
            break;
        default:
            break;
        }
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* solve-side kernels (...0100000.c:435-506)                                                              */
/* ---------------------------------------------------------------------------------------------------- */
void pangulu_platform_0100000_spmv(pangulu_inblock_idx nb, slot_t *a, val_t *x, val_t *y)
{
    /* y -= A x, A in CSC */
    for (int c = 0; c < nb; c++)
    {
        val_t xc = x[c];
        for (pangulu_int32_t p = ptr_at(a->columnpointer, c); p < (pangulu_int32_t)a->columnpointer[c + 1]; p++)
        {
            y[a->rowindex[p]] -= a->value[p] * xc;
        }
    }
}

void pangulu_platform_0100000_vecadd(pangulu_int64_t length, val_t *bval, val_t *xval)
{
    for (pangulu_int64_t i = 0; i < length; i++)
    {
        bval[i] += xval[i];
    }
}

void pangulu_platform_0100000_sptrsv(pangulu_inblock_idx nb, slot_t *s, val_t *x, pangulu_int64_t uplo)
{
    if (uplo == PANGULU_LOWER)
    {
        /* unit-lower, strictly-lower CSC: forward substitution by columns */
        for (int c = 0; c < nb; c++)
        {
            val_t xc = x[c];
            for (pangulu_int32_t p = s->columnpointer[c]; p < (pangulu_int32_t)s->columnpointer[c + 1]; p++)
            {
                x[s->rowindex[p]] -= s->value[p] * xc;
            }
        }
    }
    else
    {
        /* upper CSR with the diagonal first in each row: backward substitution by rows */
        for (int r = nb - 1; r >= 0; r--)
        {
            pangulu_int32_t beg = s->columnpointer[r], end = s->columnpointer[r + 1];
            if (beg == end)
                continue;
            val_t acc = x[r];
            for (pangulu_int32_t p = beg + 1; p < end; p++)
            {
                acc -= s->value[p] * x[s->rowindex[p]];
            }
            val_t d = s->value[beg];
#ifdef PANGULU_COMPLEX
            int tiny = !(fabs((double)creal(d)) > PANGULU_SPTRSV_TOL);
#else
            int tiny = !(fabs((double)d) > PANGULU_SPTRSV_TOL);
#endif
            x[r] = tiny ? acc / (val_t)PANGULU_SPTRSV_TOL : acc / d;
        }
    }
}

/* ---------------------------------------------------------------------------------------------------- */
/* structural flop counters (src/pangulu_kernel_interface.c:4-176)                                       */
/* ---------------------------------------------------------------------------------------------------- */
static long long count_matches(const pangulu_inblock_idx *a, pangulu_int32_t p, pangulu_int32_t p_end,
                               const pangulu_inblock_idx *b, pangulu_int32_t q, pangulu_int32_t q_end)
{
    long long hits = 0;
    while (p < p_end && q < q_end)
    {
        if (a[p] == b[q])
        {
            hits++;
            p++;
            q++;
        }
        else if (a[p] < b[q])
        {
            p++;
        }
        else
        {
            q++;
        }
    }
    return hits;
}

long long pangulu_oracle_getrf_flop(pangulu_inblock_idx nb, slot_t *opdst)
{
    slot_t *U, *L;
    split_diag(opdst, &U, &L);
    long long flop = 0;
    for (int k = 0; k < nb; k++)
    {
        if (U->columnpointer[k] == U->columnpointer[k + 1])
            continue;
        flop += L->columnpointer[k + 1] - L->columnpointer[k];
        for (pangulu_int32_t p = L->columnpointer[k]; p < (pangulu_int32_t)L->columnpointer[k + 1]; p++)
        {
            int r = L->rowindex[p];
            flop += 2 * count_matches(U->rowindex, U->columnpointer[k], U->columnpointer[k + 1],
                                      U->rowindex, U->columnpointer[r], U->columnpointer[r + 1]);
        }
        for (pangulu_int32_t p = U->columnpointer[k] + 1; p < (pangulu_int32_t)U->columnpointer[k + 1]; p++)
        {
            int c = U->rowindex[p];
            flop += 2 * count_matches(L->rowindex, L->columnpointer[k], L->columnpointer[k + 1],
                                      L->rowindex, L->columnpointer[c], L->columnpointer[c + 1]);
        }
    }
    return flop;
}

long long pangulu_oracle_tstrf_flop(pangulu_inblock_idx nb, slot_t *opdst, slot_t *opdiag)
{
    if (opdiag->is_upper == 0)
        opdiag = opdiag->related_block;
    long long flop = 0;
    for (int row = 0; row < nb; row++)
    {
        pangulu_int32_t row_end = opdst->rowpointer[row + 1];
        for (pangulu_int32_t p = opdst->rowpointer[row]; p < row_end; p++)
        {
            int c = opdst->columnindex[p];
            flop += 1;
            flop += 2 * count_matches(opdiag->rowindex, opdiag->columnpointer[c], opdiag->columnpointer[c + 1],
                                      opdst->columnindex, p + 1, row_end);
        }
    }
    return flop;
}

long long pangulu_oracle_gessm_flop(pangulu_inblock_idx nb, slot_t *opdst, slot_t *opdiag)
{
    if (opdiag->is_upper == 1)
        opdiag = opdiag->related_block;
    long long flop = 0;
    for (int col = 0; col < nb; col++)
    {
        pangulu_int32_t col_end = opdst->columnpointer[col + 1];
        for (pangulu_int32_t p = opdst->columnpointer[col]; p < col_end; p++)
        {
            int r = opdst->rowindex[p];
            flop += 2 * count_matches(opdiag->rowindex, opdiag->columnpointer[r], opdiag->columnpointer[r + 1],
                                      opdst->rowindex, p + 1, col_end);
        }
    }
    return flop;
}

long long pangulu_oracle_ssssm_flop(pangulu_inblock_idx nb, slot_t *op1, slot_t *op2)
{
    long long flop = 0;
    for (int c = 0; c < nb; c++)
    {
        for (pangulu_int32_t p = op2->columnpointer[c]; p < (pangulu_int32_t)op2->columnpointer[c + 1]; p++)
        {
            int r = op2->rowindex[p];
            flop += 2LL * (op1->columnpointer[r + 1] - op1->columnpointer[r]);
        }
    }
    return flop;
}

long long pangulu_oracle_task_flop(pangulu_inblock_idx nb, pangulu_task_t *t)
{
    switch (t->kernel_id)
    {
    case PANGULU_TASK_GETRF:
        return pangulu_oracle_getrf_flop(nb, t->opdst);
    case PANGULU_TASK_TSTRF:
        return pangulu_oracle_tstrf_flop(nb, t->opdst, t->op1);
    case PANGULU_TASK_GESSM:
        return pangulu_oracle_gessm_flop(nb, t->opdst, t->op1);
    case PANGULU_TASK_SSSSM:
        return pangulu_oracle_ssssm_flop(nb, t->op1, t->op2);
    default:
        return 0;
    }
}

int pangulu_oracle_sizeof_value(void) { return (int)sizeof(val_t); }
