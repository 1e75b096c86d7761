// pg_check.cpp -- the reference's factor check on the factors where they are.
//
// pangulu_numeric_check (src/pangulu_numeric.c:1082-1341) computes, after pangulu_gstrf,
//     || L (U 1) - A 1 ||_2 / || A 1 ||_2
// with A the reordered matrix: every rank multiplies the blocks it owns (U blocks and upper diagonal halves first, then the
// L blocks, unit diagonal), the partial vectors are summed over the ranks, rank 0 prints the quotient.  Same here, but the
// block products run on the DEVICE-resident records (pangulu_platform_0201001_block_spmv_add: one launch per sweep, no
// download of the factors -- 46 GB for the Serena-class matrix), on the host copies for host-memory platforms.
#include <cmath>

#include "pg_host.h"

namespace pg
{

namespace
{

#ifdef PANGULU_COMPLEX
inline val_t cmul(val_t a, val_t b) { return val_t{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline void cacc(val_t &d, val_t v)
{
    d.re += v.re;
    d.im += v.im;
}
inline double cabs2(val_t a) { return (double)a.re * a.re + (double)a.im * a.im; }
inline val_t csub(val_t a, val_t b) { return val_t{a.re - b.re, a.im - b.im}; }
inline val_t cone() { return val_t{1, 0}; }
inline val_t czero() { return val_t{0, 0}; }
#else
inline val_t cmul(val_t a, val_t b) { return a * b; }
inline void cacc(val_t &d, val_t v) { d += v; }
inline double cabs2(val_t a) { return (double)a * (double)a; }
inline val_t csub(val_t a, val_t b) { return a - b; }
inline val_t cone() { return (val_t)1; }
inline val_t czero() { return (val_t)0; }
#endif

// y[dst] += A x[src] over the listed blocks, wherever the records live
void apply_blocks(Solver &S, const std::vector<slot_t *> &slots, const std::vector<pangulu_exblock_idx> &src, const std::vector<pangulu_exblock_idx> &dst,
                  const std::vector<int> &csr, const std::vector<val_t> &x, std::vector<val_t> &y)
{
    Platform &plat = active_platform();
    const u32 nb = S.nb;
    if (!plat.host_memory && plat.block_spmv_add)
    {
        plat.block_spmv_add((pangulu_inblock_idx)nb, slots.size(), slots.data(), src.data(), dst.data(), csr.data(), x.data(), y.data(), x.size());
        return;
    }
    download_factors(S);
    for (size_t i = 0; i < slots.size(); i++)
    {
        const slot_t *s = slots[i];
        const val_t *xs = x.data() + (size_t)src[i] * nb;
        val_t *yd = y.data() + (size_t)dst[i] * nb;
        for (u32 c = 0; c < nb; c++)
            for (u32 p = s->columnpointer[c]; p < s->columnpointer[c + 1]; p++)
            {
                if (csr[i])
                    cacc(yd[c], cmul(s->value[p], xs[s->rowindex[p]]));
                else
                    cacc(yd[s->rowindex[p]], cmul(s->value[p], xs[c]));
            }
    }
}

// element-wise sum of a vector over all ranks, result everywhere (partial vectors to rank 0, broadcast back)
void sum_over_ranks(std::vector<val_t> &v)
{
    Comm *comm = world();
    if (comm->size <= 1)
        return;
    const int TAG = 0x200000;
    if (comm->rank == 0)
    {
        std::vector<val_t> tmp(v.size());
        for (int r = 1; r < comm->size; r++)
        {
            comm->recv_bytes(r, TAG, tmp.data(), sizeof(val_t) * tmp.size());
            for (size_t i = 0; i < v.size(); i++)
                cacc(v[i], tmp[i]);
        }
    }
    else
        comm->send_bytes(0, TAG, v.data(), sizeof(val_t) * v.size());
    comm->bcast(v.data(), sizeof(val_t) * v.size(), 0);
}

} // namespace

double factor_check(Solver &S)
{
    const u32 nb = S.nb, nbk = S.nbk, n = S.n;
    const size_t len = (size_t)nbk * nb;
    std::vector<slot_t *> slots;
    std::vector<pangulu_exblock_idx> src, dst;
    std::vector<int> csr;
    auto add = [&](slot_t *s, u32 from, u32 to, int is_csr)
    {
        slots.push_back(s);
        src.push_back(from);
        dst.push_back(to);
        csr.push_back(is_csr);
    };
    // t = U 1: this rank's upper blocks and upper diagonal halves
    for (auto &s : S.storage.owned)
    {
        if (s.brow_pos == s.bcol_pos)
        {
            if (s.is_upper)
                add(&s, s.bcol_pos, s.brow_pos, 1);
        }
        else if (s.brow_pos < s.bcol_pos)
            add(&s, s.bcol_pos, s.brow_pos, 0);
    }
    std::vector<val_t> ones(len, czero()), t(len, czero());
    for (u32 i = 0; i < n; i++)
        ones[i] = cone();
    apply_blocks(S, slots, src, dst, csr, ones, t);
    sum_over_ranks(t);
    // y = L t, L unit lower: y = t + (strictly lower part) t
    slots.clear();
    src.clear();
    dst.clear();
    csr.clear();
    for (auto &s : S.storage.owned)
    {
        if (s.brow_pos == s.bcol_pos)
        {
            if (!s.is_upper)
                add(&s, s.bcol_pos, s.brow_pos, 0);
        }
        else if (s.brow_pos > s.bcol_pos)
            add(&s, s.bcol_pos, s.brow_pos, 0);
    }
    std::vector<val_t> y(len, czero());
    apply_blocks(S, slots, src, dst, csr, t, y);
    sum_over_ranks(y);
    // A 1 from the reordered matrix every rank holds (row sums), the two norms
    std::vector<val_t> a1(len, czero());
    const CscMatrix &A = S.Aperm;
    for (u32 j = 0; j < A.n; j++)
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            cacc(a1[A.rowidx[p]], A.value[p]);
    double num = 0, den = 0;
    for (u32 i = 0; i < n; i++)
    {
        val_t lu = y[i];
        cacc(lu, t[i]);
        num += cabs2(csub(lu, a1[i]));
        den += cabs2(a1[i]);
    }
    return den > 0 ? std::sqrt(num / den) : std::sqrt(num);
}

} // namespace pg
