// pg_check.cpp -- the reference's factor check on the factors where they are.
//
// pangulu_numeric_check (src/pangulu_numeric.c:1082-1341) computes, after pangulu_gstrf,
//     || L (U 1) - A 1 ||_2 / || A 1 ||_2
// with A the reordered matrix: every rank multiplies the blocks it owns (U blocks and upper diagonal halves first, then the
// L blocks, unit diagonal), the partial vectors are summed over the ranks, rank 0 prints the quotient.  Same here, but the
// block products run on the DEVICE-resident records (pangulu_platform_0201001_block_spmv_add: one launch per sweep, no
// download of the factors -- 46 GB for the Serena-class matrix), on the host copies for host-memory platforms.
#include <algorithm>
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

namespace
{

// the block lists of the two sweeps (built once per call) and || L (U x) - A x ||_2 / || A x ||_2 for one vector x
struct CheckLists
{
    std::vector<slot_t *> u_slots, l_slots;
    std::vector<pangulu_exblock_idx> u_src, u_dst, l_src, l_dst;
    std::vector<int> u_csr, l_csr;
};

CheckLists check_lists(Solver &S)
{
    CheckLists L;
    for (auto &s : S.storage.owned)
    {
        // t = U x: this rank's upper blocks and upper diagonal halves (CSR); y = L t, L unit lower: y = t + (strictly lower part) t
        const bool diag = s.brow_pos == s.bcol_pos;
        if ((diag && s.is_upper) || (!diag && s.brow_pos < s.bcol_pos))
        {
            L.u_slots.push_back(&s);
            L.u_src.push_back(s.bcol_pos);
            L.u_dst.push_back(s.brow_pos);
            L.u_csr.push_back(diag ? 1 : 0);
        }
        else
        {
            L.l_slots.push_back(&s);
            L.l_src.push_back(s.bcol_pos);
            L.l_dst.push_back(s.brow_pos);
            L.l_csr.push_back(0);
        }
    }
    return L;
}

double check_one_vector(Solver &S, const CheckLists &L, const std::vector<val_t> &x)
{
    const u32 n = S.n;
    const size_t len = x.size();
    std::vector<val_t> t(len, czero()), y(len, czero()), ax(len, czero());
    apply_blocks(S, L.u_slots, L.u_src, L.u_dst, L.u_csr, x, t);
    sum_over_ranks(t);
    apply_blocks(S, L.l_slots, L.l_src, L.l_dst, L.l_csr, t, y);
    sum_over_ranks(y);
    // A x from the reordered matrix every rank holds, the two norms
    const CscMatrix &A = S.Aperm;
    for (u32 j = 0; j < A.n; j++)
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            cacc(ax[A.rowidx[p]], cmul(A.value[p], x[j]));
    double num = 0, den = 0;
    for (u32 i = 0; i < n; i++)
    {
        val_t lu = y[i];
        cacc(lu, t[i]);
        num += cabs2(csub(lu, ax[i]));
        den += cabs2(ax[i]);
    }
    return den > 0 ? std::sqrt(num / den) : std::sqrt(num);
}

} // namespace

double factor_check(Solver &S)
{
    const size_t len = (size_t)S.nbk * S.nb;
    std::vector<val_t> ones(len, czero());
    for (u32 i = 0; i < S.n; i++)
        ones[i] = cone();
    return check_one_vector(S, check_lists(S), ones);
}

// The same criterion on `nvec` vectors: the reference's all-ones vector first (src/pangulu_numeric.c:1082-1341 is the one-vector
// form), then random +-1 vectors from a seeded generator (the same on every rank) -- the largest quotient.  One vector probes one
// direction: an error confined to entries whose columns cancel under x = 1 does not show; eight independent sign patterns leave
// 2^-7 of that room per entry pair.
double factor_check_vectors(Solver &S, int nvec, unsigned long long seed)
{
    const size_t len = (size_t)S.nbk * S.nb;
    const CheckLists L = check_lists(S);
    std::vector<val_t> x(len, czero());
    double worst = 0;
    unsigned long long st = seed ? seed : 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < std::max(1, nvec); k++)
    {
        for (u32 i = 0; i < S.n; i++)
        {
            if (k == 0)
            {
                x[i] = cone();
                continue;
            }
            st += 0x9E3779B97F4A7C15ull; // splitmix64
            unsigned long long z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            x[i] = (z >> 63) ? cone() : csub(czero(), cone());
        }
        const double q = check_one_vector(S, L, x);
        if (std::isnan(q) || (!std::isnan(worst) && q > worst)) // a NaN on ANY vector sticks (`!(q <= worst)` let the next finite q overwrite it)
            worst = q;
    }
    return worst;
}

} // namespace pg
