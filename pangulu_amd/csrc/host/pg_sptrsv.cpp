// pg_sptrsv.cpp -- block forward/backward substitution for pangulu_gstrs ("next" row f1 of SURVEY.md §8).
//
// Same sweep as the reference (src/pangulu_sptrsv.c:24-191): block row by block row, every rank that owns
// blocks of the row (the diagonal owner's process row, or any rank under the subtree mapping) adds their products, the diagonal owner sums the partial
// vectors, solves with its diagonal half and broadcasts the finished segment.  Like the reference (which pins
// the solve to PANGULU_PLATFORM_CPU_NAIVE, src/pangulu_sptrsv.c:62,94,126,159) it runs on the host copies of
// the factors; the in-block kernels follow ...0100000.c:435-506.
#include <cmath>

#include "pg_host.h"

namespace pg
{

namespace
{

#ifdef PANGULU_COMPLEX
inline val_t vmul(val_t a, val_t b) { return val_t{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline val_t vsub(val_t a, val_t b) { return val_t{a.re - b.re, a.im - b.im}; }
inline val_t vadd(val_t a, val_t b) { return val_t{a.re + b.re, a.im + b.im}; }
inline val_t vdiv(val_t a, val_t b)
{
    calculate_real_type d = b.re * b.re + b.im * b.im;
    return val_t{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
inline double vreal(val_t a) { return (double)a.re; }
inline val_t vmake(double r) { return val_t{(calculate_real_type)r, 0}; }
#else
inline val_t vmul(val_t a, val_t b) { return a * b; }
inline val_t vsub(val_t a, val_t b) { return a - b; }
inline val_t vadd(val_t a, val_t b) { return a + b; }
inline val_t vdiv(val_t a, val_t b) { return a / b; }
inline double vreal(val_t a) { return (double)a; }
inline val_t vmake(double r) { return (val_t)r; }
#endif

// y -= A x for a CSC block
void block_spmv(u32 nb, const slot_t *a, const val_t *x, val_t *y)
{
    for (u32 c = 0; c < nb; c++)
    {
        val_t xc = x[c];
        for (u32 p = a->columnpointer[c]; p < a->columnpointer[c + 1]; p++)
            y[a->rowindex[p]] = vsub(y[a->rowindex[p]], vmul(a->value[p], xc));
    }
}

void block_lower_solve(u32 nb, const slot_t *l, val_t *x)
{
    for (u32 c = 0; c < nb; c++)
    {
        val_t xc = x[c];
        for (u32 p = l->columnpointer[c]; p < l->columnpointer[c + 1]; p++)
            x[l->rowindex[p]] = vsub(x[l->rowindex[p]], vmul(l->value[p], xc));
    }
}

void block_upper_solve(u32 nb, const slot_t *u, val_t *x)
{
    for (i64 r = (i64)nb - 1; r >= 0; r--)
    {
        u32 b = u->columnpointer[r], e = u->columnpointer[r + 1];
        if (b == e)
            continue;
        val_t acc = x[r];
        for (u32 p = b + 1; p < e; p++)
            acc = vsub(acc, vmul(u->value[p], x[u->rowindex[p]]));
        val_t d = u->value[b];
        x[r] = (std::fabs(vreal(d)) > PANGULU_SPTRSV_TOL) ? vdiv(acc, d) : vdiv(acc, vmake(PANGULU_SPTRSV_TOL));
    }
}

} // namespace

// Single rank on a device: both sweeps on the device-resident factors (no download of the factors), level by level of the
// block dependency graph -- pangulu_platform_0201001_block_trsv.  PANGULU_AMD_DEVICE_SOLVE=0 keeps the host sweep.
static bool device_solve(Solver &S, val_t *rhs)
{
    Platform &plat = active_platform();
    const char *e = getenv("PANGULU_AMD_DEVICE_SOLVE");
    if (S.nproc != 1 || plat.host_memory || !plat.block_trsv || (e && atoi(e) == 0))
        return false;
    const BlockPattern &P = S.pat;
    const u32 nb = S.nb, nbk = S.nbk;
    std::vector<val_t> x((size_t)nbk * nb, vmake(0));
    std::copy(rhs, rhs + S.n, x.begin());
    for (int pass = 0; pass < 2; pass++)
    {
        const bool lower = pass == 0;
        // level of a block row = 1 + the highest level among the rows its off-diagonal blocks (on the sweep's side) read
        std::vector<u32> level(nbk, 0);
        u32 nlevel = 0;
        for (u32 step = 0; step < nbk; step++)
        {
            const u32 brow = lower ? step : nbk - 1 - step;
            const u64 rb = lower ? P.rowptr[brow] : P.first_after_diag_csr[brow];
            const u64 re = lower ? P.first_after_diag_csr[brow] : P.rowptr[brow + 1];
            u32 lv = 0;
            for (u64 r = rb; r < re; r++)
                lv = std::max(lv, level[P.colidx[r]] + 1);
            level[brow] = lv;
            nlevel = std::max(nlevel, lv + 1);
        }
        std::vector<pangulu_uint64_t> level_ptr((size_t)nlevel + 1, 0);
        for (u32 k = 0; k < nbk; k++)
            level_ptr[level[k] + 1]++;
        for (u32 l = 0; l < nlevel; l++)
            level_ptr[l + 1] += level_ptr[l];
        std::vector<pangulu_hip_solve_row_t> rows(nbk);
        std::vector<slot_t *> blk_slots;
        std::vector<pangulu_exblock_idx> blk_bcol;
        std::vector<pangulu_uint64_t> cur(level_ptr.begin(), level_ptr.end() - 1);
        for (u32 brow = 0; brow < nbk; brow++)
        {
            pangulu_hip_solve_row_t R;
            R.brow = brow;
            R.first = blk_slots.size();
            R.diag = lower ? S.diag_lower[brow] : S.diag_upper[brow];
            const u64 rb = lower ? P.rowptr[brow] : P.first_after_diag_csr[brow];
            const u64 re = lower ? P.first_after_diag_csr[brow] : P.rowptr[brow + 1];
            for (u64 r = rb; r < re; r++)
            {
                blk_slots.push_back(S.slot_of[P.csr_to_csc[r]]);
                blk_bcol.push_back(P.colidx[r]);
            }
            R.nblk = (pangulu_exblock_idx)(blk_slots.size() - R.first);
            rows[cur[level[brow]]++] = R;
        }
        if (blk_slots.empty())
        {
            blk_slots.push_back(nullptr);
            blk_bcol.push_back(0);
        }
        plat.block_trsv((pangulu_inblock_idx)nb, lower ? 0 : 1, nlevel, level_ptr.data(), rows.data(), blk_slots.data(), blk_bcol.data(), x.data(),
                        (pangulu_uint64_t)x.size());
    }
    std::copy(x.begin(), x.begin() + S.n, rhs);
    return true;
}

void triangular_solve(Solver &S, val_t *rhs)
{
    if (device_solve(S, rhs))
        return;
    download_factors(S);
    Comm *comm = world();
    const BlockPattern &P = S.pat;
    u32 nb = S.nb, nbk = S.nbk;
    int me = S.rank;
    std::vector<char> contributes((size_t)S.nproc, 0);
    std::vector<val_t> x((size_t)nbk * nb, vmake(0)), acc(nb), tmp(nb);
    std::copy(rhs, rhs + S.n, x.begin());
    const int TAG_PART = 0x100000;

    for (int pass = 0; pass < 2; pass++)
    {
        bool lower = pass == 0;
        for (u32 step = 0; step < nbk; step++)
        {
            u32 brow = lower ? step : nbk - 1 - step;
            val_t *seg = x.data() + (size_t)brow * nb;
            int diag_rank = S.owner(brow, brow);
            // the ranks that hold blocks of block row brow on the relevant side of the diagonal (with the subtree
            // mapping these are not confined to the diagonal owner's process row): each adds the products of its
            // own blocks, the diagonal owner sums the partial vectors
            u64 rb = lower ? P.rowptr[brow] : P.first_after_diag_csr[brow];
            u64 re = lower ? P.first_after_diag_csr[brow] : P.rowptr[brow + 1];
            std::fill(contributes.begin(), contributes.end(), 0);
            for (u64 r = rb; r < re; r++)
                contributes[(size_t)S.owner(brow, P.colidx[r])] = 1;
            if (contributes[(size_t)me] || me == diag_rank)
            {
                std::fill(acc.begin(), acc.end(), vmake(0));
                for (u64 r = rb; r < re; r++)
                {
                    u32 bcol = P.colidx[r];
                    if (S.owner(brow, bcol) != me)
                        continue;
                    slot_t *blk = S.slot_of[P.csr_to_csc[r]];
                    block_spmv(nb, blk, x.data() + (size_t)bcol * nb, acc.data());
                }
                if (me == diag_rank)
                {
                    for (int r = 0; r < S.nproc; r++)
                    {
                        if (r == me || !contributes[(size_t)r])
                            continue;
                        comm->recv_bytes(r, TAG_PART + (int)(brow & 0xfffff), tmp.data(), sizeof(val_t) * nb);
                        for (u32 i = 0; i < nb; i++)
                            acc[i] = vadd(acc[i], tmp[i]);
                    }
                    for (u32 i = 0; i < nb; i++)
                        seg[i] = vadd(seg[i], acc[i]);
                    if (lower)
                        block_lower_solve(nb, S.diag_lower[brow], seg);
                    else
                        block_upper_solve(nb, S.diag_upper[brow], seg);
                }
                else
                {
                    comm->send_bytes(diag_rank, TAG_PART + (int)(brow & 0xfffff), acc.data(), sizeof(val_t) * nb);
                }
            }
            comm->bcast(seg, sizeof(val_t) * nb, diag_rank);
        }
        comm->barrier();
    }
    std::copy(x.begin(), x.begin() + S.n, rhs);
}

} // namespace pg
