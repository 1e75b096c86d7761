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
#include <cmath>
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

// ---------------------------------------------------------------------------------------------------------
// The same model for ALL tasks of the factorisation, from the replicated symbolic pattern alone (any number of ranks).
//
// The symbolic pattern is symmetric, so the pattern of U(k, j) is the transpose of L(j, k) and every closed form above
// needs nothing but the column counts of the lower blocks of block column k:
//   cnt_B[c] = entries of column c of lower block B = L(i, k)   (diagonal block: diagonal entry included)
//   nl_c = cnt_diag[c] - 1  (strictly lower entries of column c of the diagonal block)  = nu_c - 1
//   GETRF(k)            sum_c nl_c (1 + 2 nl_c)
//   TSTRF  L(i,k)       sum_c cnt_B[c] (1 + 2 nl_c)
//   GESSM  U(k,j)       sum_c 2 cnt_{L(j,k)}[c] nl_c
//   SSSSM  (i,j) from k 2 sum_c cnt_{L(i,k)}[c] cnt_{L(j,k)}[c]
// (tests/test_abi_and_host.py checks these against compute_task_model on single-rank handles: identical sums.)
// ---------------------------------------------------------------------------------------------------------
namespace
{

struct TaskCost
{
    double bytes, flop;
};

// walks every task of the factorisation once, level by level; fn(cls, level, brow, bcol, cost, slot) is called from an OpenMP
// region (parallel over levels) -- callers accumulate with atomics or per-thread state.  `slot`: for an update, its
// position bi * nl + ai among the nl x nl pairs of its level (a, b = positions of L(i,k), L(j,k) in the level's L column)
template <typename F>
void for_every_task(const Solver &S, F &&fn)
{
    const BlockPattern &P = S.pat;
    const u32 nb = S.nb, nbk = S.nbk;
    const double sv = (double)sizeof(val_t);
    const u16 *lcount = S.smodel.lcount.data();
#pragma omp parallel for schedule(dynamic, 1)
    for (i64 kk = 0; kk < (i64)nbk; kk++)
    {
        const u32 k = (u32)kk;
        const u64 l0 = P.lcolptr[k], l1 = P.lcolptr[k + 1];
        // the diagonal block is the first lower block of its column
        const u16 *cd = lcount + (size_t)l0 * nb;
        const double nnzL = P.diag_lower_nnz[k], nnzU = P.diag_upper_nnz[k];
        {
            double fl = 0;
            for (u32 c = 0; c < nb; c++)
                if (cd[c])
                {
                    const double nl = (double)cd[c] - 1.0;
                    fl += nl * (1.0 + 2.0 * nl);
                }
            fn(PANGULU_TASK_GETRF, k, k, k, TaskCost{(2 * sv + 2) * (nnzL + nnzU) + 8.0 * (nb + 1), fl}, (u64)0);
        }
        for (u64 a = l0 + 1; a < l1; a++)
        {
            const u32 i = P.lrowidx[a];
            const u16 *ca = lcount + (size_t)a * nb;
            const double nnzB = P.lnnz[a];
            double ft = 0, fg = 0;
            for (u32 c = 0; c < nb; c++)
                if (ca[c])
                {
                    const double nl = cd[c] ? (double)cd[c] - 1.0 : 0.0;
                    ft += (double)ca[c] * (1.0 + 2.0 * nl);
                    fg += 2.0 * (double)ca[c] * nl;
                }
            fn(PANGULU_TASK_TSTRF, k, i, k, TaskCost{(2 * sv + 6) * nnzB + (sv + 2) * nnzU + 8.0 * (nb + 1), ft}, (u64)0);
            fn(PANGULU_TASK_GESSM, k, k, i, TaskCost{(2 * sv + 2) * nnzB + (sv + 2) * nnzL + 8.0 * (nb + 1), fg}, (u64)0);
        }
        // updates C(i, j) -= L(i, k) U(k, j), i and j over the off-diagonal lower blocks of column k
        for (u64 b = l0 + 1; b < l1; b++)
        {
            const u32 j = P.lrowidx[b];
            const u16 *cb = lcount + (size_t)b * nb;
            for (u64 a = l0 + 1; a < l1; a++)
            {
                const u32 i = P.lrowidx[a];
                double nnz_c;
                if (i == j)
                    nnz_c = (double)P.diag_lower_nnz[i] + P.diag_upper_nnz[i];
                else
                {
                    const u64 bd = P.find(i, j);
                    if (bd == ~0ull)
                        continue;
                    nnz_c = P.nnz[bd];
                }
                const u16 *ca = lcount + (size_t)a * nb;
                double fl = 0;
                for (u32 c = 0; c < nb; c++)
                    fl += (double)((u32)ca[c] * (u32)cb[c]);
                fn(PANGULU_TASK_SSSSM, k, i, j, TaskCost{(sv + 2) * ((double)P.lnnz[a] + P.lnnz[b]) + (2 * sv + 2) * nnz_c + 12.0 * (nb + 1), 2.0 * fl},
                   (b - (l0 + 1)) * (l1 - l0 - 1) + (a - (l0 + 1)));
            }
        }
    }
}

} // namespace

void build_structure_model(Solver &S)
{
    StructureModel &M = S.smodel;
    if (const char *e = getenv("PANGULU_AMD_MODEL_HBM_GBS"))
        M.hbm_bytes_per_s = 1e9 * atof(e);
    if (const char *e = getenv("PANGULU_AMD_MODEL_FP_TFLOPS"))
        M.fp_flops_per_s = 1e12 * atof(e);
    if (const char *e = getenv("PANGULU_AMD_MODEL_LINK_GBS"))
        M.link_bytes_per_s = 1e9 * atof(e);
    const BlockPattern &P = S.pat;
    const Symbolic &sym = S.sym;
    const u32 nb = S.nb, nbk = S.nbk, n = S.n;
    if (nb > 65535)
        return;
    M.lcount.assign((size_t)P.lcolptr[nbk] * nb, 0);
#pragma omp parallel
    {
        std::vector<i64> local_of(nbk, -1);
#pragma omp for schedule(dynamic, 2)
        for (i64 bc_ = 0; bc_ < (i64)nbk; bc_++)
        {
            const u32 bc = (u32)bc_;
            const u64 l0 = P.lcolptr[bc], l1 = P.lcolptr[bc + 1];
            for (u64 t = l0; t < l1; t++)
                local_of[P.lrowidx[t]] = (i64)t;
            const u32 j0 = bc * nb, j1 = std::min(n, j0 + nb);
            for (u32 j = j0; j < j1; j++)
                for (u64 p = sym.ptr[j]; p < sym.ptr[j + 1]; p++)
                    M.lcount[(size_t)local_of[sym.idx[p] / nb] * nb + (j - j0)]++;
            for (u64 t = l0; t < l1; t++)
                local_of[P.lrowidx[t]] = -1;
        }
    }
    M.col_time.assign(nbk, 0.0);
    M.col_flop.assign(nbk, 0.0);
    const double bw = M.hbm_bytes_per_s, peak = M.fp_flops_per_s;
    // per-thread sums, reduced afterwards: the columns of a top separator collect the updates of every level below, and
    // hundreds of threads adding to the same words with compare-and-swap loops do not scale
    // ... in INTEGERS (femtoseconds, flops): the mapping compares these weights and every rank must come to the same
    // map, whatever its thread count and scheduling -- integer sums do not depend on the order of the terms
    const int nthr = omp_get_max_threads();
    std::vector<std::vector<i64>> part((size_t)nthr);
    for_every_task(S, [&](int, u32, u32 brow, u32 bcol, const TaskCost &c, u64)
                   {
                       std::vector<i64> &mine = part[(size_t)omp_get_thread_num()];
                       if (mine.empty())
                           mine.assign((size_t)nbk * 2, 0);
                       const u32 col = std::min(brow, bcol);
                       mine[2 * (size_t)col] += (i64)std::llround(std::max(c.bytes / bw, c.flop / peak) * 1e15);
                       mine[2 * (size_t)col + 1] += (i64)std::llround(c.flop); });
    std::vector<i64> tot((size_t)nbk * 2, 0);
    for (const auto &mine : part)
        if (!mine.empty())
            for (size_t k = 0; k < (size_t)nbk * 2; k++)
                tot[k] += mine[k];
    for (u32 k = 0; k < nbk; k++)
    {
        M.col_time[k] = 1e-15 * (double)tot[2 * (size_t)k];
        M.col_flop[k] = (double)tot[2 * (size_t)k + 1];
    }
}

#ifdef PANGULU_COMPLEX
#define MODEL_MIRROR_PLANES 2.0 // (a complex block's mirror is two real planes)
#else
#define MODEL_MIRROR_PLANES 1.0
#endif
void compute_rank_model(Solver &S)
{
    StructureModel &M = S.smodel;
    const BlockPattern &P = S.pat;
    const u32 nb = S.nb, nbk = S.nbk;
    const int np = S.nproc;
    if (M.lcount.empty())
        return;
    M.rank_time_hbm.assign((size_t)np, 0.0);
    M.rank_time_fp.assign((size_t)np, 0.0);
    M.rank_flop.assign((size_t)np, 0.0);
    M.rank_bytes.assign((size_t)np, 0.0);
    M.rank_comm_s.assign((size_t)np, 0.0);
    M.sent_bytes.assign((size_t)np * np, 0.0);
    const double bw = M.hbm_bytes_per_s, peak = M.fp_flops_per_s;
    // per-task times kept for the critical path below: panel tasks per block, updates per (level, pair)
    const u64 nblk = P.colptr[nbk];
    std::vector<float> t_panel(nblk, 0.f), t_getrf(nbk, 0.f);
    std::vector<u64> upd_off((size_t)nbk + 1, 0);
    for (u32 k = 0; k < nbk; k++)
    {
        const u64 nl = P.lcolptr[k + 1] - P.lcolptr[k] - 1;
        upd_off[k + 1] = upd_off[k] + nl * nl;
    }
    std::vector<float> t_upd(upd_off[nbk], -1.f); // (< 0: the pair has no destination block)
    // (per-thread sums: with one rank every task of the factorisation would add to the same four words)
    const int nthr = omp_get_max_threads();
    std::vector<double> acc((size_t)nthr * np * 4, 0.0);
    for_every_task(S, [&](int cls, u32 level, u32 brow, u32 bcol, const TaskCost &c, u64 slot)
                   {
                       const int r = S.owner(brow, bcol);
                       const double a = c.bytes / bw, b = c.flop / peak;
                       double *mine = acc.data() + ((size_t)omp_get_thread_num() * np + r) * 4;
                       mine[a >= b ? 0 : 1] += std::max(a, b);
                       mine[2] += c.flop;
                       mine[3] += c.bytes;
                       if (cls == PANGULU_TASK_GETRF)
                           t_getrf[brow] = (float)std::max(a, b);
                       else if (cls == PANGULU_TASK_SSSSM)
                           t_upd[upd_off[level] + slot] = (float)std::max(a, b);
                       else
                           t_panel[P.find(brow, bcol)] = (float)std::max(a, b); });
    for (int t = 0; t < nthr; t++)
        for (int r = 0; r < np; r++)
        {
            const double *mine = acc.data() + ((size_t)t * np + r) * 4;
            M.rank_time_hbm[(size_t)r] += mine[0];
            M.rank_time_fp[(size_t)r] += mine[1];
            M.rank_flop[(size_t)r] += mine[2];
            M.rank_bytes[(size_t)r] += mine[3];
        }
    // bytes forwarded between ranks: every finished block goes once to each rank that runs an update with it
    // (Solver::consumers), a diagonal block's halves to the ranks that run panel solves against them
    if (np > 1)
    {
        for (u32 bc = 0; bc < nbk; bc++)
            for (u64 t = P.colptr[bc]; t < P.colptr[bc + 1]; t++)
            {
                const u32 br = P.rowidx[t];
                const int o = S.owner(br, bc);
                const double bytes = (double)record_bytes(nb, P.nnz[t], br > bc);
                u64 to = S.consumers.empty() ? 0ull : S.consumers[t];
                if (S.consumers.empty())
                {
                    // the reference's rule: the whole process row / column that owns blocks behind it
                    if (br > bc)
                        for (u64 r = P.rowptr[br]; r < P.rowptr[br + 1]; r++)
                        {
                            if (P.colidx[r] > bc)
                                to |= 1ull << S.owner(br, P.colidx[r]);
                        }
                    else
                        for (u64 c = P.colptr[bc]; c < P.colptr[bc + 1]; c++)
                            if (P.rowidx[c] > br)
                                to |= 1ull << S.owner(P.rowidx[c], bc);
                }
                to &= ~(1ull << o);
                for (int r = 0; r < np && r < 64; r++)
                    if ((to >> r) & 1ull)
                        M.sent_bytes[(size_t)o * np + r] += bytes;
            }
        for (u32 k = 0; k < nbk; k++)
        {
            const int o = S.owner(k, k);
            u64 to_u = 0, to_l = 0;
            for (u64 t = P.first_after_diag[k]; t < P.colptr[k + 1]; t++)
                to_u |= 1ull << S.owner(P.rowidx[t], k);
            for (u64 t = P.first_after_diag_csr[k]; t < P.rowptr[k + 1]; t++)
                to_l |= 1ull << S.owner(k, P.colidx[t]);
            for (int r = 0; r < np && r < 64; r++)
                if (r != o)
                {
                    if ((to_u >> r) & 1ull)
                        M.sent_bytes[(size_t)o * np + r] += (double)record_bytes(nb, P.diag_upper_nnz[k], false);
                    if ((to_l >> r) & 1ull)
                        M.sent_bytes[(size_t)o * np + r] += (double)record_bytes(nb, P.diag_lower_nnz[k], false);
                }
        }
    }
    // HBM a rank needs under this mapping: the records it owns, the records it receives (receive bins provisioned for every block
    // once, as the multi-rank replay wants them), and a dense mirror for every block of either kind whose fill reaches the dense
    // threshold (2 per mille unless PANGULU_HIP_DENSE_PERMILLE says otherwise; nb = 128 or 256 only) -- the pool's DEMAND: when it
    // is not met the blocks stay on the sparse kernels.  bench.py's snapshot of the records and the recorded schedule's descriptors
    // come on top (see hbm_breakdown_GB of a measured line).
    {
        M.rank_mem_records.assign((size_t)np, 0.0);
        M.rank_mem_recv.assign((size_t)np, 0.0);
        M.rank_mem_mirrors.assign((size_t)np, 0.0);
        double permille = 2.0;
        if (const char *e = getenv("PANGULU_HIP_DENSE_PERMILLE"))
            permille = atof(e);
        const bool dense_mode = nb == 128 || nb == 256;
        const double mirror_bytes = dense_mode ? 8.0 * ((double)nb * nb + 8.0 + 16.0 * nb) * MODEL_MIRROR_PLANES : 0.0;
        const double thr = permille * 1e-3 * (double)nb * nb;
        for (u32 bc = 0; bc < nbk; bc++)
            for (u64 t = P.colptr[bc]; t < P.colptr[bc + 1]; t++)
            {
                const u32 br = P.rowidx[t];
                const int o = S.owner(br, bc);
                const double bytes = (double)record_bytes(nb, P.nnz[t], br > bc);
                const double mir = (double)P.nnz[t] >= thr ? mirror_bytes : 0.0;
                M.rank_mem_records[(size_t)o] += bytes;
                M.rank_mem_mirrors[(size_t)o] += mir;
                u64 to = S.consumers.empty() ? 0ull : S.consumers[t];
                to &= ~(1ull << o);
                for (int r = 0; r < np && r < 64; r++)
                    if ((to >> r) & 1ull)
                    {
                        M.rank_mem_recv[(size_t)r] += bytes;
                        M.rank_mem_mirrors[(size_t)r] += mir;
                    }
            }
        for (u32 k = 0; k < nbk; k++)
        {
            const int o = S.owner(k, k);
            M.rank_mem_records[(size_t)o] += (double)record_bytes(nb, P.diag_upper_nnz[k], false) + (double)record_bytes(nb, P.diag_lower_nnz[k], false);
            M.rank_mem_mirrors[(size_t)o] += ((double)P.diag_upper_nnz[k] + P.diag_lower_nnz[k] >= thr) ? mirror_bytes : 0.0;
        }
        double worst = 0;
        int at = 0;
        for (int r = 0; r < np; r++)
        {
            const double tot = M.rank_mem_records[(size_t)r] + M.rank_mem_recv[(size_t)r] + M.rank_mem_mirrors[(size_t)r];
            if (tot > worst)
            {
                worst = tot;
                at = r;
            }
        }
        S.info.model_rank_hbm_bytes_max = worst;
        S.info.model_rank_hbm_records = M.rank_mem_records[(size_t)at];
        S.info.model_rank_hbm_received = M.rank_mem_recv[(size_t)at];
        S.info.model_rank_hbm_mirrors = M.rank_mem_mirrors[(size_t)at];
    }
    double tsum = 0, tmax = 0, tworst = 0, fsum = 0, fmax = 0, cmax = 0, sent = 0, thbm = 0, tfp = 0, bsum = 0;
    for (int r = 0; r < np; r++)
    {
        double c = 0;
        for (int q = 0; q < np; q++)
        {
            c = std::max(c, M.sent_bytes[(size_t)r * np + q] / M.link_bytes_per_s);
            sent += M.sent_bytes[(size_t)r * np + q];
        }
        M.rank_comm_s[(size_t)r] = c;
        const double t = M.rank_time_hbm[(size_t)r] + M.rank_time_fp[(size_t)r];
        tsum += t;
        thbm += M.rank_time_hbm[(size_t)r];
        tfp += M.rank_time_fp[(size_t)r];
        bsum += M.rank_bytes[(size_t)r];
        tmax = std::max(tmax, t);
        tworst = std::max(tworst, t + c);
        fsum += M.rank_flop[(size_t)r];
        fmax = std::max(fmax, M.rank_flop[(size_t)r]);
        cmax = std::max(cmax, c);
    }
    S.info.model_ranks_tstar_max = tworst;
    S.info.model_ranks_tstar_sum = tsum;
    S.info.model_ranks_tstar_hbm = thbm;
    S.info.model_ranks_tstar_fp = tfp;
    S.info.model_ranks_bytes_total = bsum;
    S.info.model_rank_flop_share = fsum > 0 ? fmax / (fsum / np) : 1.0;
    S.info.model_rank_time_share = tsum > 0 ? tmax / (tsum / np) : 1.0;
    S.info.model_comm_seconds_max = cmax;
    S.info.model_sent_bytes_total = sent;

    // Critical path of the block task graph (levels ascending; a block's updates may run concurrently once both operands are final,
    // its panel task starts when the last of them is done; serial: a few operations per task), twice:
    //  * with every task at its own T*_t -- a lower bound on any schedule's makespan however many devices there are, but a useless
    //    predictor: 0.24 us per task on the chain where a lone kernel launch has a floor of tens of microseconds;
    //  * LATENCY-AWARE: a task takes max(T*_t, the measured floor of a lone launch of its class) -- GETRF 138 us, a dense panel
    //    solve 47 us (round 5's kernels: getrf_pipe, trsm_dense_ring), an update launch 25 us at nb = 256, scaled with nb / 256 for the panel classes (16 dependent panel steps per
    //    256 columns; DESIGN.md, kernel table; PANGULU_AMD_MODEL_FLOOR_{GETRF,PANEL,UPDATE}_US) -- and an operand that comes from
    //    another rank arrives a hop later: PANGULU_AMD_MODEL_HOP_US (default 20: marker, announcement, start of the copy; an
    //    assumption until a run on real links calibrates it) + record bytes over one link.  This is what bounds strong scaling.
    {
        const double scale_nb = (double)nb / 256.0;
        auto env_us = [](const char *name, double dflt)
        {
            const char *e = getenv(name);
            return 1e-6 * (e ? atof(e) : dflt);
        };
        const float floor_getrf = (float)env_us("PANGULU_AMD_MODEL_FLOOR_GETRF_US", 138.0 * scale_nb);
        const float floor_panel = (float)env_us("PANGULU_AMD_MODEL_FLOOR_PANEL_US", 47.0 * scale_nb);
        const float floor_update = (float)env_us("PANGULU_AMD_MODEL_FLOOR_UPDATE_US", 25.0);
        const float hop = (float)env_us("PANGULU_AMD_MODEL_HOP_US", 20.0);
        const float inv_link = (float)(1.0 / M.link_bytes_per_s);
        auto chain = [&](bool lat, float &cp_out, u32 &depth_out)
        {
            std::vector<float> ready(nblk, 0.f), fin(nblk, 0.f), ready_d(nbk, 0.f), fin_d(nbk, 0.f);
            std::vector<u32> depth(nblk, 0), depth_rd(nbk, 0), depth_r(nblk, 0), depth_d(nbk, 0);
            std::vector<u64> bl_of, bu_of;
            std::vector<int> own_l, own_u;
            // when a block finished on rank `from` is usable on rank `to`
            auto arrive = [&](float t_done, int from, int to, double bytes) -> float
            { return (!lat || from == to) ? t_done : t_done + hop + (float)bytes * inv_link; };
            for (u32 k = 0; k < nbk; k++)
            {
                const int od = S.owner(k, k);
                fin_d[k] = ready_d[k] + (lat ? std::max(t_getrf[k], floor_getrf) : t_getrf[k]);
                depth_d[k] = depth_rd[k] + 1;
                const u64 l0 = P.lcolptr[k], l1 = P.lcolptr[k + 1];
                const u64 nl = l1 - l0 - 1;
                bl_of.resize(nl);
                bu_of.resize(nl);
                own_l.resize(nl);
                own_u.resize(nl);
                const double diag_bytes = (double)record_bytes(nb, P.diag_upper_nnz[k], false);
                // panel solves of column k and row k
                for (u64 a = 0; a < nl; a++)
                {
                    const u32 i = P.lrowidx[l0 + 1 + a];
                    const u64 bl = P.find(i, k), bu = P.find(k, i);
                    bl_of[a] = bl;
                    bu_of[a] = bu;
                    own_l[a] = S.owner(i, k);
                    own_u[a] = S.owner(k, i);
                    fin[bl] = std::max(ready[bl], arrive(fin_d[k], od, own_l[a], diag_bytes)) + (lat ? std::max(t_panel[bl], floor_panel) : t_panel[bl]);
                    depth[bl] = std::max(depth_r[bl], depth_d[k]) + 1;
                    fin[bu] = std::max(ready[bu], arrive(fin_d[k], od, own_u[a], diag_bytes)) + (lat ? std::max(t_panel[bu], floor_panel) : t_panel[bu]);
                    depth[bu] = std::max(depth_r[bu], depth_d[k]) + 1;
                }
                // updates generated by level k
                for (u64 b = 0; b < nl; b++)
                {
                    const u32 j = P.lrowidx[l0 + 1 + b];
                    const u64 bu = bu_of[b];
                    const double bu_bytes = lat ? (double)record_bytes(nb, P.nnz[bu], false) : 0.0;
                    // destinations (i, j), i over the L column: walk block column j of the pattern alongside
                    u64 cur = P.colptr[j];
                    const u64 cend = P.colptr[j + 1];
                    for (u64 a = 0; a < nl; a++)
                    {
                        const float t = t_upd[upd_off[k] + b * nl + a];
                        if (t < 0)
                            continue;
                        const u32 i = P.lrowidx[l0 + 1 + a];
                        const u64 bl = bl_of[a];
                        float in_l = fin[bl], in_u = fin[bu];
                        if (lat)
                        {
                            const int o = S.owner(i, j);
                            in_l = arrive(in_l, own_l[a], o, (double)record_bytes(nb, P.nnz[bl], true));
                            in_u = arrive(in_u, own_u[b], o, bu_bytes);
                        }
                        const float done = std::max(in_l, in_u) + (lat ? std::max(t, floor_update) : t);
                        const u32 dp = std::max(depth[bl], depth[bu]) + 1;
                        if (i == j)
                        {
                            ready_d[i] = std::max(ready_d[i], done);
                            depth_rd[i] = std::max(depth_rd[i], dp);
                        }
                        else
                        {
                            while (cur < cend && P.rowidx[cur] < i)
                                cur++;
                            // (t >= 0 means the destination exists; i ascends with a, but U blocks (i < j) come first in the column)
                            const u64 bd = (cur < cend && P.rowidx[cur] == i) ? cur : P.find(i, j);
                            ready[bd] = std::max(ready[bd], done);
                            depth_r[bd] = std::max(depth_r[bd], dp);
                        }
                    }
                }
            }
            float cp = 0;
            u32 dmax = 0;
            for (u32 k = 0; k < nbk; k++)
            {
                cp = std::max(cp, fin_d[k]);
                dmax = std::max(dmax, depth_d[k]);
            }
            for (u64 b = 0; b < nblk; b++)
            {
                cp = std::max(cp, fin[b]);
                dmax = std::max(dmax, depth[b]);
            }
            cp_out = cp;
            depth_out = dmax;
        };
        float cp = 0, cp_lat = 0;
        u32 dmax = 0, dlat = 0;
        chain(false, cp, dmax);
        chain(true, cp_lat, dlat);
        M.critical_path_s = cp;
        M.critical_path_tasks = dmax;
        M.critical_path_latency_s = cp_lat;
        S.info.model_critical_path = cp;
        S.info.model_critical_path_tasks = dmax;
        S.info.model_critical_path_latency = cp_lat;
    }
    // (M.lcount stays: 51 MB of host memory for the Serena-class matrix, and evaluate_model_for_ranks() needs it again)
}

// The structure-only model of this handle's factorisation on `nranks` ranks -- what pangulu_init evaluates for the handle's own
// rank count -- for ANY rank count, on a handle of any rank count (the pattern and the per-column weights are replicated): the
// mapping is made for nranks, the sets of consuming ranks are derived from it, the rank model runs, and the handle's own mapping,
// consumer sets and model figures are put back.  out[0..11]: T*(N) incl. the link term, sum_r T*_r, max link term, bytes sent in
// all, critical path at T*_t, LATENCY-AWARE critical path, max / mean flop share, max / mean T* share, HBM of the fullest rank and
// its three parts (records owned, records received, dense mirrors), all in seconds / bytes.
void evaluate_model_for_ranks(Solver &S, int nranks, double *out)
{
    if (nranks < 1 || nranks > 64 || S.smodel.lcount.empty())
    {
        for (int i = 0; i < 12; i++)
            out[i] = -1.0;
        return;
    }
    const int nproc0 = S.nproc, p0 = S.p, q0 = S.q, rank0 = S.rank;
    std::vector<int> home0;
    std::vector<Solver::Group> grp0;
    std::vector<u64> cons0;
    home0.swap(S.home);
    grp0.swap(S.grp);
    cons0.swap(S.consumers);
    // (the figures only: lcount -- 51 MB for the Serena-class matrix, read-only in compute_rank_model -- is parked beside the copy)
    std::vector<u16> lcount_parked;
    lcount_parked.swap(S.smodel.lcount);
    const StructureModel model0 = S.smodel;
    S.smodel.lcount.swap(lcount_parked);
    const pangulu_amd_info_t info0 = S.info;
    S.nproc = nranks;
    int p = (int)std::sqrt((double)nranks);
    while (nranks % p)
        p--;
    S.p = p;
    S.q = nranks / p;
    S.rank = 0;
    assign_subtrees(S);
    // who consumes which block under that mapping (the rule of preprocess(): the owner of every update's destination)
    const BlockPattern &P = S.pat;
    const u32 nbk = S.nbk;
    if (nranks > 1 && !getenv("PANGULU_AMD_REFERENCE_FORWARDING"))
    {
        S.consumers.assign(P.colptr[nbk], 0);
#pragma omp parallel
        {
            std::vector<i64> pos(nbk, -1);
#pragma omp for schedule(dynamic, 2)
            for (i64 b_ = 0; b_ < (i64)nbk; b_++)
            {
                const u32 b = (u32)b_;
                for (u64 t = P.colptr[b]; t < P.colptr[b + 1]; t++)
                    pos[P.rowidx[t]] = (i64)t;
                for (u64 t = P.colptr[b]; t < P.first_after_diag[b]; t++)
                {
                    const u32 k = P.rowidx[t];
                    for (u64 la = P.first_after_diag[k]; la < P.colptr[k + 1]; la++)
                    {
                        const u32 a = P.rowidx[la];
                        if (a != b && pos[a] < 0)
                            continue;
                        const int od = S.owner(a, b);
                        if (S.owner(a, k) != od)
                        {
#pragma omp atomic
                            S.consumers[la] |= 1ull << od;
                        }
                        if (S.owner(k, b) != od)
                        {
#pragma omp atomic
                            S.consumers[t] |= 1ull << od;
                        }
                    }
                }
                for (u64 t = P.colptr[b]; t < P.colptr[b + 1]; t++)
                    pos[P.rowidx[t]] = -1;
            }
        }
    }
    compute_rank_model(S);
    out[0] = S.info.model_ranks_tstar_max;
    out[1] = S.info.model_ranks_tstar_sum;
    out[2] = S.info.model_comm_seconds_max;
    out[3] = S.info.model_sent_bytes_total;
    out[4] = S.info.model_critical_path;
    out[5] = S.info.model_critical_path_latency;
    out[6] = S.info.model_rank_flop_share;
    out[7] = S.info.model_rank_time_share;
    out[8] = S.info.model_rank_hbm_bytes_max;
    out[9] = S.info.model_rank_hbm_records;
    out[10] = S.info.model_rank_hbm_received;
    out[11] = S.info.model_rank_hbm_mirrors;
    // ... and everything back
    S.nproc = nproc0;
    S.p = p0;
    S.q = q0;
    S.rank = rank0;
    S.home.swap(home0);
    S.grp.swap(grp0);
    S.consumers.swap(cons0);
    lcount_parked.swap(S.smodel.lcount);
    S.smodel = model0;
    S.smodel.lcount.swap(lcount_parked);
    S.info = info0;
}

} // namespace pg
