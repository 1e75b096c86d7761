// pg_model.cpp -- structure-only roofline model of the factorisation's task list.
//
//   T* = sum over tasks t of max(bytes_t / BW_HBM, flop_t / P_fp),   reported as T*/t_gstrf, split by which bound applies.
//
// flop_t are the reference's structural counts (src/pangulu_kernel_interface.c:4-176) evaluated in closed form on the
// (closed) symbolic pattern -- an update the merges of the reference would look for always finds its target there:
//   GETRF  sum_k nl_k (1 + 2 (nu_k - 1))              nl_k / nu_k = entries of L column k / U row k (diagonal included in nu)
//   TSTRF  sum_c colcount_B(c) (1 + 2 (nu_c - 1))     B = the L block being solved
//   GESSM  sum_r rowcount_B(r) 2 nl_r                 B = the U block being solved
//   SSSSM  sum_k rowcount_op2(k) 2 colcount_op1(k)
// and bytes_t are §8d's: every operand record streamed once + destination values read and written once.
// (tests/test_reference_pin.py pins the same four counts of the oracle against the reference's own counters; the sum over
// all tasks equals info.flop.)  Single-rank handles only: the patterns of all blocks must be local.
#include <omp.h>

#include "pg_host.h"

namespace pg
{

namespace
{

inline u32 ptr_at(const pangulu_inblock_ptr *p, u32 i) { return i == 0 ? 0u : p[i]; }

struct Acc
{
    double bytes[5] = {0, 0, 0, 0, 0}, flop[5] = {0, 0, 0, 0, 0};
    u64 count[5] = {0, 0, 0, 0, 0};
    double t_hbm = 0, t_fp = 0; // seconds of the tasks whose larger term is the HBM / the floating-point one
    void add(int cls, double by, double fl, double bw, double peak)
    {
        bytes[cls] += by;
        flop[cls] += fl;
        count[cls]++;
        const double a = by / bw, b = fl / peak;
        if (a >= b)
            t_hbm += a;
        else
            t_fp += b;
    }
};

} // namespace

// structural flops of ONE task from the host patterns of its operands (same closed forms as below)
double task_structural_flop(u32 nb, const task_t &t)
{
    auto counts = [&](const slot_t *s, u32 c) -> double
    { return (double)(s->columnpointer[c + 1] - ptr_at(s->columnpointer, c)); };
    const slot_t *dst = t.opdst;
    double fl = 0;
    switch (t.kernel_id)
    {
    case PANGULU_TASK_GETRF:
    {
        const slot_t *up = dst->is_upper ? dst : dst->related_block, *lo = dst->is_upper ? dst->related_block : dst;
        for (u32 c = 0; c < nb; c++)
            if (counts(up, c) > 0)
                fl += counts(lo, c) * (1.0 + 2.0 * (counts(up, c) - 1.0));
        break;
    }
    case PANGULU_TASK_TSTRF:
    {
        const slot_t *up = t.op1->is_upper ? t.op1 : t.op1->related_block;
        for (u32 c = 0; c < nb; c++)
            if (counts(dst, c) > 0)
                fl += counts(dst, c) * (1.0 + 2.0 * std::max(0.0, counts(up, c) - 1.0));
        break;
    }
    case PANGULU_TASK_GESSM:
    {
        const slot_t *lo = t.op1->is_upper ? t.op1->related_block : t.op1;
        const u32 nnz = dst->columnpointer[nb];
        for (u32 p = 0; p < nnz; p++)
            fl += 2.0 * counts(lo, dst->rowindex[p]);
        break;
    }
    case PANGULU_TASK_SSSSM:
    {
        const u32 nnz = t.op2->columnpointer[nb];
        for (u32 p = 0; p < nnz; p++)
            fl += 2.0 * counts(t.op1, t.op2->rowindex[p]);
        break;
    }
    default:
        break;
    }
    return fl;
}

void compute_task_model(Solver &S, double hbm_bytes_per_s, double fp_flops_per_s)
{
    S.model = TaskModel();
    S.info.model_bytes_total = S.info.model_flop_total = S.info.model_tmin_hbm_bound = S.info.model_tmin_fp_bound = 0;
    if (S.nproc != 1)
        return;
    const BlockPattern &P = S.pat;
    const u32 nb = S.nb, nbk = S.nbk;
    const double sv = (double)sizeof(val_t);
    const int nthr = omp_get_max_threads();
    std::vector<Acc> acc((size_t)nthr);
#pragma omp parallel
    {
        Acc &A = acc[(size_t)omp_get_thread_num()];
        std::vector<u32> rowcount(nb);
#pragma omp for schedule(dynamic, 4)
        for (i64 kk = 0; kk < (i64)nbk; kk++)
        {
            const u32 k = (u32)kk;
            const slot_t *lo = S.diag_lower[k], *up = S.diag_upper[k];
            if (!lo || !up)
                continue;
            const pangulu_inblock_ptr *lcp = lo->columnpointer, *urp = up->columnpointer; // (upper half: CSR row pointer)
            const double nnzL = lcp[nb], nnzU = urp[nb];
            {
                double fl = 0;
                for (u32 c = 0; c < nb; c++)
                {
                    const double nl = lcp[c + 1] - ptr_at(lcp, c), nu = urp[c + 1] - ptr_at(urp, c);
                    if (nu > 0)
                        fl += nl * (1.0 + 2.0 * (nu - 1.0));
                }
                A.add(PANGULU_TASK_GETRF, (2 * sv + 2) * (nnzL + nnzU) + 8.0 * (nb + 1), fl, hbm_bytes_per_s, fp_flops_per_s);
            }
            // TSTRF: L blocks (i, k), i > k
            for (u64 b = P.first_after_diag[k]; b < P.colptr[k + 1]; b++)
            {
                const slot_t *B = S.slot_of[b];
                if (!B)
                    continue;
                const pangulu_inblock_ptr *cp = B->columnpointer;
                double fl = 0;
                for (u32 c = 0; c < nb; c++)
                {
                    const double cnt = cp[c + 1] - ptr_at(cp, c), nu = urp[c + 1] - ptr_at(urp, c);
                    if (cnt > 0)
                        fl += cnt * (1.0 + 2.0 * std::max(0.0, nu - 1.0));
                }
                const double nnzB = cp[nb];
                A.add(PANGULU_TASK_TSTRF, (2 * sv + 6) * nnzB + (sv + 2) * nnzU + 8.0 * (nb + 1), fl, hbm_bytes_per_s, fp_flops_per_s);
            }
            // GESSM: U blocks (k, j), j > k; and the SSSSM tasks they take part in
            for (u64 q = P.first_after_diag_csr[k]; q < P.rowptr[k + 1]; q++)
            {
                const u64 bu = P.csr_to_csc[q];
                const u32 j = P.colidx[q];
                const slot_t *U = S.slot_of[bu];
                if (!U)
                    continue;
                const pangulu_inblock_ptr *ucp = U->columnpointer;
                const pangulu_inblock_idx *uri = U->rowindex;
                const u32 nnz_u = ucp[nb];
                std::fill(rowcount.begin(), rowcount.end(), 0u);
                for (u32 p = 0; p < nnz_u; p++)
                    rowcount[uri[p]]++;
                {
                    double fl = 0;
                    for (u32 r = 0; r < nb; r++)
                        if (rowcount[r])
                            fl += 2.0 * rowcount[r] * (double)(lcp[r + 1] - ptr_at(lcp, r));
                    A.add(PANGULU_TASK_GESSM, (2 * sv + 2) * (double)nnz_u + (sv + 2) * nnzL + 8.0 * (nb + 1), fl, hbm_bytes_per_s, fp_flops_per_s);
                }
                for (u64 b = P.first_after_diag[k]; b < P.colptr[k + 1]; b++)
                {
                    const u32 i = P.rowidx[b];
                    const slot_t *L = S.slot_of[b];
                    if (!L)
                        continue;
                    double nnz_c;
                    if (i == j)
                        nnz_c = (double)S.diag_lower[i]->columnpointer[nb] + S.diag_upper[i]->columnpointer[nb];
                    else
                    {
                        const u64 bd = P.find(i, j);
                        if (bd == ~0ull || !S.slot_of[bd])
                            continue;
                        nnz_c = S.slot_of[bd]->columnpointer[nb];
                    }
                    const pangulu_inblock_ptr *cp = L->columnpointer;
                    double fl = 0;
                    for (u32 c = 0; c < nb; c++)
                        if (rowcount[c])
                            fl += 2.0 * rowcount[c] * (double)(cp[c + 1] - ptr_at(cp, c));
                    A.add(PANGULU_TASK_SSSSM, (sv + 2) * ((double)cp[nb] + nnz_u) + (2 * sv + 2) * nnz_c + 12.0 * (nb + 1), fl, hbm_bytes_per_s,
                          fp_flops_per_s);
                }
            }
        }
    }
    for (const Acc &A : acc)
    {
        for (int c = 1; c <= 4; c++)
        {
            S.model.bytes[c] += A.bytes[c];
            S.model.flop[c] += A.flop[c];
            S.model.count[c] += A.count[c];
            S.info.model_bytes_total += A.bytes[c];
            S.info.model_flop_total += A.flop[c];
        }
        S.info.model_tmin_hbm_bound += A.t_hbm;
        S.info.model_tmin_fp_bound += A.t_fp;
    }
}

} // namespace pg
