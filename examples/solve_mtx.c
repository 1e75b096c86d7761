/*
 * solve_mtx.c -- a user program against include/pangulu.h, the way a program written for the reference is
 * (the reference ships examples/example.c:282-300: init, gstrf, gstrs, finalize, then ||Ax - b|| / ||b||).
 *
 *   solve_mtx -f matrix.mtx|matrix.lid [-n block_order] [-r rhs.txt]
 *
 * Reads a Matrix Market coordinate file (real / integer / pattern, general / symmetric / skew-symmetric) or -- by the last letter of
 * the name, as the reference's example chooses (examples/example.c:100-163) -- its binary .lid layout (u32 rows, u32 columns, u64
 * entries, then the CSR arrays: u64 row pointers, u32 column indices, values), solves A x = b on the GPU of this process and prints
 * the relative residual.  Without -r the right-hand side is b = A * 1 (the reference's choice,
 * examples/example.c:245-266); with it, a text file holding the length on the first non-comment line and one value per line.
 *
 * One process per GPU: started under a launcher that exports RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run does)
 * every process calls pangulu_amd_comm_init() first -- the place of MPI_Init_thread in the reference's example -- and only rank 0
 * reads the file.  Built for R64 (link against libpangulu_amd_r64.so):
 *
 *   gcc -O2 -DCALCULATE_TYPE_R64 -Iinclude examples/solve_mtx.c -o solve_mtx -Lpangulu_amd/lib -lpangulu_amd_r64 -Wl,-rpath,$PWD/pangulu_amd/lib -lm
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pangulu.h"
#include "pangulu_amd_ext.h"

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void die(const char *what, const char *arg)
{
    fprintf(stderr, "solve_mtx: %s%s%s\n", what, arg ? ": " : "", arg ? arg : "");
    exit(1);
}

typedef struct
{
    sparse_index_t n;
    sparse_pointer_t nnz;
    sparse_pointer_t *colptr;
    sparse_index_t *rowidx;
    sparse_value_t *value;
} csc_t;

/* coordinate entries -> CSC by a counting sort on the column; duplicates are summed (Matrix Market allows them in assembled form) */
static csc_t read_matrix_market(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f)
        die("cannot open", path);
    char line[1024], object[64], format[64], field[64], symmetry[64];
    if (!fgets(line, sizeof line, f) || sscanf(line, "%%%%MatrixMarket %63s %63s %63s %63s", object, format, field, symmetry) != 4)
        die("not a Matrix Market file", path);
    for (char *p = field; *p; p++)
        *p = (char)tolower((unsigned char)*p);
    for (char *p = symmetry; *p; p++)
        *p = (char)tolower((unsigned char)*p);
    if (strcmp(format, "coordinate") != 0 || strcmp(field, "complex") == 0)
        die("only real / integer / pattern coordinate files (this build is R64)", path);
    const int pattern = strcmp(field, "pattern") == 0;
    const int mirror = strcmp(symmetry, "general") != 0;
    const double mirror_sign = strcmp(symmetry, "skew-symmetric") == 0 ? -1.0 : 1.0;
    do
    {
        if (!fgets(line, sizeof line, f))
            die("no size line", path);
    } while (line[0] == '%' || line[0] == '\n');
    long long rows, cols, entries;
    if (sscanf(line, "%lld %lld %lld", &rows, &cols, &entries) != 3 || rows != cols || rows <= 0)
        die("the matrix has to be square", path);
    const size_t cap = (size_t)entries * (mirror ? 2 : 1);
    sparse_index_t *ti = malloc(sizeof *ti * cap), *tj = malloc(sizeof *tj * cap);
    double *tv = malloc(sizeof *tv * cap);
    if (!ti || !tj || !tv)
        die("out of memory", NULL);
    size_t m = 0;
    for (long long e = 0; e < entries; e++)
    {
        long long i, j;
        double v = 1.0;
        if (!fgets(line, sizeof line, f))
            die("file ends before its last entry", path);
        if ((pattern ? sscanf(line, "%lld %lld", &i, &j) != 2 : sscanf(line, "%lld %lld %lf", &i, &j, &v) != 3) || i < 1 || j < 1 || i > rows || j > cols)
            die("bad entry line", line);
        ti[m] = (sparse_index_t)(i - 1), tj[m] = (sparse_index_t)(j - 1), tv[m++] = v;
        if (mirror && i != j)
            ti[m] = (sparse_index_t)(j - 1), tj[m] = (sparse_index_t)(i - 1), tv[m++] = mirror_sign * v;
    }
    fclose(f);
    csc_t A;
    A.n = (sparse_index_t)rows;
    A.colptr = calloc((size_t)rows + 1, sizeof *A.colptr);
    A.rowidx = malloc(sizeof *A.rowidx * (m ? m : 1));
    A.value = malloc(sizeof *A.value * (m ? m : 1));
    sparse_pointer_t *fill = malloc(sizeof *fill * (size_t)rows);
    if (!A.colptr || !A.rowidx || !A.value || !fill)
        die("out of memory", NULL);
    for (size_t k = 0; k < m; k++)
        A.colptr[tj[k] + 1]++;
    for (long long c = 0; c < rows; c++)
        A.colptr[c + 1] += A.colptr[c], fill[c] = A.colptr[c];
    for (size_t k = 0; k < m; k++)
    {
        sparse_pointer_t at = fill[tj[k]]++;
        A.rowidx[at] = ti[k], A.value[at] = tv[k];
    }
    /* rows ascending inside every column (insertion sort: columns are short), equal rows merged */
    sparse_pointer_t out = 0;
    for (long long c = 0; c < rows; c++)
    {
        const sparse_pointer_t lo = A.colptr[c], hi = fill[c];
        for (sparse_pointer_t a = lo + 1; a < hi; a++)
        {
            sparse_index_t r = A.rowidx[a];
            sparse_value_t v = A.value[a];
            sparse_pointer_t b = a;
            for (; b > lo && A.rowidx[b - 1] > r; b--)
                A.rowidx[b] = A.rowidx[b - 1], A.value[b] = A.value[b - 1];
            A.rowidx[b] = r, A.value[b] = v;
        }
        A.colptr[c] = out;
        for (sparse_pointer_t a = lo; a < hi; a++)
            if (out > A.colptr[c] && A.rowidx[out - 1] == A.rowidx[a])
                A.value[out - 1] += A.value[a];
            else
                A.rowidx[out] = A.rowidx[a], A.value[out++] = A.value[a];
    }
    A.colptr[rows] = out;
    A.nnz = out;
    free(ti), free(tj), free(tv), free(fill);
    return A;
}

/* the reference's binary layout: CSR on disk, transposed into CSC here (rows stay ascending inside a column) */
static csc_t read_lid(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        die("cannot open", path);
    unsigned int dims[2];
    unsigned long long entries;
    if (fread(dims, sizeof dims[0], 2, f) != 2 || fread(&entries, sizeof entries, 1, f) != 1 || dims[0] != dims[1] || dims[0] == 0)
        die("not a square .lid file", path);
    const size_t n = dims[0], m = (size_t)entries;
    unsigned long long *rowptr = malloc(sizeof *rowptr * (n + 1));
    unsigned int *colidx = malloc(sizeof *colidx * (m ? m : 1));
    sparse_value_t *val = malloc(sizeof *val * (m ? m : 1));
    if (!rowptr || !colidx || !val)
        die("out of memory", NULL);
    if (fread(rowptr, sizeof *rowptr, n + 1, f) != n + 1 || fread(colidx, sizeof *colidx, m, f) != m || fread(val, sizeof *val, m, f) != m ||
        rowptr[n] != entries)
        die("the .lid file is short or inconsistent", path);
    fclose(f);
    csc_t A;
    A.n = (sparse_index_t)n, A.nnz = (sparse_pointer_t)m;
    A.colptr = calloc(n + 1, sizeof *A.colptr);
    A.rowidx = malloc(sizeof *A.rowidx * (m ? m : 1));
    A.value = malloc(sizeof *A.value * (m ? m : 1));
    sparse_pointer_t *fill = malloc(sizeof *fill * n);
    if (!A.colptr || !A.rowidx || !A.value || !fill)
        die("out of memory", NULL);
    for (size_t k = 0; k < m; k++)
    {
        if (colidx[k] >= n)
            die("column index out of range", path);
        A.colptr[colidx[k] + 1]++;
    }
    for (size_t c = 0; c < n; c++)
        A.colptr[c + 1] += A.colptr[c], fill[c] = A.colptr[c];
    for (size_t r = 0; r < n; r++)
        for (unsigned long long k = rowptr[r]; k < rowptr[r + 1]; k++)
        {
            sparse_pointer_t at = fill[colidx[k]]++;
            A.rowidx[at] = (sparse_index_t)r, A.value[at] = val[k];
        }
    free(rowptr), free(colidx), free(val), free(fill);
    return A;
}

static void multiply(const csc_t *A, const sparse_value_t *x, sparse_value_t *y)
{
    memset(y, 0, sizeof *y * A->n);
    for (sparse_index_t c = 0; c < A->n; c++)
        for (sparse_pointer_t p = A->colptr[c]; p < A->colptr[c + 1]; p++)
            y[A->rowidx[p]] += A->value[p] * x[c];
}

static void read_vector(const char *path, sparse_index_t n, sparse_value_t *b)
{
    FILE *f = fopen(path, "r");
    if (!f)
        die("cannot open", path);
    char line[256];
    long long len = -1, got = 0;
    while (fgets(line, sizeof line, f))
    {
        if (line[0] == '%' || line[0] == '#' || line[0] == '\n')
            continue;
        if (len < 0)
        {
            if (sscanf(line, "%lld", &len) != 1 || len != (long long)n)
                die("the right-hand side's length does not match the matrix", path);
        }
        else if (got < len && sscanf(line, "%lf", &b[got]) == 1)
            got++;
    }
    fclose(f);
    if (got != (long long)n)
        die("the right-hand side file is short", path);
}

int main(int argc, char **argv)
{
    const char *mtx = NULL, *rhs = NULL;
    int nb = 256;
    for (int a = 1; a < argc; a++)
        if (!strcmp(argv[a], "-f") && a + 1 < argc)
            mtx = argv[++a];
        else if (!strcmp(argv[a], "-r") && a + 1 < argc)
            rhs = argv[++a];
        else if ((!strcmp(argv[a], "-n") || !strcmp(argv[a], "-nb")) && a + 1 < argc)
            nb = atoi(argv[++a]);
        else
            die("usage: solve_mtx -f matrix.mtx|matrix.lid [-n block_order] [-r rhs.txt]", NULL);
    if (!mtx || nb <= 0)
        die("usage: solve_mtx -f matrix.mtx|matrix.lid [-n block_order] [-r rhs.txt]", NULL);

    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
    const int size = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
    if (size > 1)
    {
        const char *addr = getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1";
        const int port = (getenv("MASTER_PORT") ? atoi(getenv("MASTER_PORT")) : 29500) + 23;
        if (pangulu_amd_comm_init(rank, size, addr, port, PANGULU_AMD_TRANSPORT_RCCL, NULL) != 0)
            die("pangulu_amd_comm_init failed", NULL);
    }

    csc_t A;
    memset(&A, 0, sizeof A);
    sparse_value_t *b = NULL, *x = NULL;
    if (rank == 0)
    {
        double t = now_s();
        A = mtx[strlen(mtx) - 1] == 'd' ? read_lid(mtx) : read_matrix_market(mtx);
        printf("%s: n = %u, %llu entries, read in %.2f s\n", mtx, (unsigned)A.n, (unsigned long long)A.nnz, now_s() - t);
        b = malloc(sizeof *b * A.n), x = malloc(sizeof *x * A.n);
        if (!b || !x)
            die("out of memory", NULL);
        if (rhs)
            read_vector(rhs, A.n, b);
        else
        {
            for (sparse_index_t i = 0; i < A.n; i++)
                x[i] = 1.0;
            multiply(&A, x, b);
        }
        memcpy(x, b, sizeof *x * A.n);
    }

    pangulu_init_options init_options;
    memset(&init_options, 0, sizeof init_options);
    init_options.nthread = 8;
    init_options.nb = nb;
    init_options.sizeof_value = (int)sizeof(sparse_value_t);
    init_options.is_complex_matrix = 0;
    init_options.mpi_recv_buffer_level = 1.0f;
    pangulu_gstrf_options gstrf_options;
    pangulu_gstrs_options gstrs_options;
    memset(&gstrf_options, 0, sizeof gstrf_options);
    memset(&gstrs_options, 0, sizeof gstrs_options);
    void *handle = NULL;

    double t0 = now_s();
    pangulu_init(A.n, A.nnz, A.colptr, A.rowidx, A.value, &init_options, &handle); /* (ranks other than 0 pass an empty matrix) */
    double t1 = now_s();
    pangulu_gstrf(&gstrf_options, &handle);
    double t2 = now_s();
    pangulu_gstrs(x, &gstrs_options, &handle);
    double t3 = now_s();

    if (rank == 0)
    {
        pangulu_amd_info_t info;
        pangulu_amd_get_info(&handle, &info);
        sparse_value_t *ax = malloc(sizeof *ax * A.n);
        multiply(&A, x, ax);
        double rr = 0.0, bb = 0.0;
        for (sparse_index_t i = 0; i < A.n; i++)
            rr += (ax[i] - b[i]) * (ax[i] - b[i]), bb += b[i] * b[i];
        printf("preprocess %.3f s, numeric factorisation %.3f s (%.1f GFLOP/s), solve %.3f s, %d rank(s)\n", t1 - t0, t2 - t1,
               (double)info.flop / (t2 - t1) / 1e9, t3 - t2, size);
        printf("|| Ax - b || / || b || = %le\n", sqrt(rr) / sqrt(bb));
        free(ax);
    }
    pangulu_finalize(&handle);
    if (size > 1)
        pangulu_amd_comm_finalize();
    free(A.colptr), free(A.rowidx), free(A.value), free(b), free(x);
    return 0;
}
