// pg_analysis.cpp -- symbolic fill and block structure (the ordering lives in pg_ordering.cpp).
//
// Reference counterparts (re-designed, not translated):
//   symbolic      src/pangulu_symbolic.c:3-271 (pattern of A+A^T, column merge over the elimination tree)
//   block pattern src/pangulu_communication.c:792-1100 (block-CSC/CSR of the filled matrix)
#include <algorithm>
#include <cstring>
#include <cmath>
#include <numeric>
#include <omp.h>

#include "pg_host.h"

namespace pg
{

void permute_symmetric(const CscMatrix &A, const std::vector<u32> &perm, CscMatrix &B)
{
    u32 n = A.n;
    std::vector<u32> iperm(n);
    for (u32 i = 0; i < n; i++)
        iperm[perm[i]] = i;
    B.n = n;
    B.colptr.assign(n + 1, 0);
    for (u32 jn = 0; jn < n; jn++)
    {
        u32 jo = perm[jn];
        B.colptr[jn + 1] = B.colptr[jn] + (A.colptr[jo + 1] - A.colptr[jo]);
    }
    B.rowidx.resize(A.nnz());
    B.value.resize(A.nnz());
#pragma omp parallel
    {
        std::vector<std::pair<u32, val_t>> col;
#pragma omp for schedule(dynamic, 512)
        for (i64 jn = 0; jn < (i64)n; jn++)
        {
            u32 jo = perm[jn];
            col.clear();
            for (u64 p = A.colptr[jo]; p < A.colptr[jo + 1]; p++)
                col.emplace_back(iperm[A.rowidx[p]], A.value[p]);
            std::sort(col.begin(), col.end(), [](const std::pair<u32, val_t> &x, const std::pair<u32, val_t> &y)
                      { return x.first < y.first; });
            u64 o = B.colptr[jn];
            for (auto &e : col)
            {
                B.rowidx[o] = e.first;
                B.value[o] = e.second;
                o++;
            }
        }
    }
}

void symbolic_factorize(const CscMatrix &A, Symbolic &S)
{
    u32 n = A.n;
    S.n = n;
    // lower triangle (incl. diagonal, always present) of the pattern of A + A^T, CSC, sorted
    std::vector<u64> sptr(n + 1, 0);
    for (u32 j = 0; j < n; j++)
    {
        sptr[j + 1]++; // diagonal
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
        {
            u32 i = A.rowidx[p];
            if (i != j)
                sptr[std::min(i, j) + 1]++;
        }
    }
    for (u32 j = 0; j < n; j++)
        sptr[j + 1] += sptr[j];
    std::vector<u32> sidx(sptr[n]);
    {
        std::vector<u64> cur(sptr.begin(), sptr.end() - 1);
        for (u32 j = 0; j < n; j++)
        {
            sidx[cur[j]++] = j;
            for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            {
                u32 i = A.rowidx[p];
                if (i != j)
                {
                    u32 c = std::min(i, j), r = std::max(i, j);
                    sidx[cur[c]++] = r;
                }
            }
        }
    }
    std::vector<u64> slen(n);
#pragma omp parallel for schedule(dynamic, 1024)
    for (i64 j = 0; j < (i64)n; j++)
    {
        u32 *b = sidx.data() + sptr[j], *e = sidx.data() + sptr[j + 1];
        std::sort(b, e);
        slen[j] = (u64)(std::unique(b, e) - b);
    }

    // column merge over the elimination tree: struct(L_j) = struct(S_j) U (U_{children c} struct(L_c) \ {c})
    S.ptr.assign(n + 1, 0);
    S.idx.clear();
    S.idx.reserve((size_t)sptr[n] * 4);
    std::vector<u32> mark(n, 0xFFFFFFFFu), first_child(n, 0xFFFFFFFFu), next_sib(n, 0xFFFFFFFFu);
    i64 flop = 0;
    for (u32 j = 0; j < n; j++)
    {
        size_t base = S.idx.size();
        mark[j] = j;
        S.idx.push_back(j);
        for (u64 p = sptr[j]; p < sptr[j] + slen[j]; p++)
        {
            u32 i = sidx[p];
            if (i > j && mark[i] != j)
            {
                mark[i] = j;
                S.idx.push_back(i);
            }
        }
        for (u32 c = first_child[j]; c != 0xFFFFFFFFu; c = next_sib[c])
        {
            for (u64 p = S.ptr[c] + 1; p < S.ptr[c + 1]; p++) // skip c itself (first, columns are sorted)
            {
                u32 i = S.idx[p];
                if (i > j && mark[i] != j)
                {
                    mark[i] = j;
                    S.idx.push_back(i);
                }
            }
        }
        std::sort(S.idx.begin() + base, S.idx.end());
        S.ptr[j + 1] = S.idx.size();
        u64 cj = S.idx.size() - base - 1;
        flop += (i64)cj + 2 * (i64)cj * (i64)cj;
        if (cj > 0)
        {
            u32 parent = S.idx[base + 1];
            next_sib[j] = first_child[parent];
            first_child[parent] = j;
        }
    }
    S.symbolic_nnz = 2 * (u64)S.idx.size() - n;
    S.flop = flop;
}

u64 BlockPattern::find(u32 br, u32 bc) const
{
    u64 lo = colptr[bc], hi = colptr[bc + 1];
    while (lo < hi)
    {
        u64 mid = (lo + hi) >> 1;
        u32 v = rowidx[mid];
        if (v == br)
            return mid;
        if (v < br)
            lo = mid + 1;
        else
            hi = mid;
    }
    return ~0ull;
}

void build_block_pattern(const Symbolic &S, u32 nb, BlockPattern &P)
{
    u32 n = S.n;
    u32 nbk = (n + nb - 1) / nb;
    P.nb = nb;
    P.nbk = nbk;
    P.n = n;
    // lower blocks per block column
    std::vector<std::vector<std::pair<u32, u32>>> cols(nbk);
#pragma omp parallel
    {
        std::vector<u32> cnt(nbk, 0), touched;
#pragma omp for schedule(dynamic, 4)
        for (i64 bc = 0; bc < (i64)nbk; bc++)
        {
            touched.clear();
            u32 j0 = (u32)bc * nb, j1 = std::min(n, j0 + nb);
            for (u32 j = j0; j < j1; j++)
            {
                for (u64 p = S.ptr[j]; p < S.ptr[j + 1]; p++)
                {
                    u32 br = S.idx[p] / nb;
                    if (cnt[br]++ == 0)
                        touched.push_back(br);
                }
            }
            std::sort(touched.begin(), touched.end());
            auto &out = cols[bc];
            out.reserve(touched.size());
            for (u32 br : touched)
            {
                out.emplace_back(br, cnt[br]);
                cnt[br] = 0;
            }
        }
    }
    P.lcolptr.assign(nbk + 1, 0);
    for (u32 bc = 0; bc < nbk; bc++)
        P.lcolptr[bc + 1] = P.lcolptr[bc] + cols[bc].size();
    P.lrowidx.resize(P.lcolptr[nbk]);
    P.lnnz.resize(P.lcolptr[nbk]);
    for (u32 bc = 0; bc < nbk; bc++)
    {
        u64 o = P.lcolptr[bc];
        for (auto &e : cols[bc])
        {
            P.lrowidx[o] = e.first;
            P.lnnz[o] = e.second;
            o++;
        }
    }
    cols.clear();
    cols.shrink_to_fit();

    // transpose of the strictly-lower block pattern: for block row b, the block columns k < b with L(b,k)
    std::vector<u64> lrowptr(nbk + 1, 0);
    for (u32 bc = 0; bc < nbk; bc++)
        for (u64 p = P.lcolptr[bc]; p < P.lcolptr[bc + 1]; p++)
            if (P.lrowidx[p] != bc)
                lrowptr[P.lrowidx[p] + 1]++;
    for (u32 b = 0; b < nbk; b++)
        lrowptr[b + 1] += lrowptr[b];
    std::vector<u32> lcolidx(lrowptr[nbk]), lrow_nnz(lrowptr[nbk]);
    {
        std::vector<u64> cur(lrowptr.begin(), lrowptr.end() - 1);
        for (u32 bc = 0; bc < nbk; bc++)
            for (u64 p = P.lcolptr[bc]; p < P.lcolptr[bc + 1]; p++)
            {
                u32 br = P.lrowidx[p];
                if (br != bc)
                {
                    lcolidx[cur[br]] = bc;
                    lrow_nnz[cur[br]] = P.lnnz[p];
                    cur[br]++;
                }
            }
    }

    // all non-diagonal blocks, block-CSC: column bc = U blocks (br < bc, mirror of row bc of L) then L blocks
    P.colptr.assign(nbk + 1, 0);
    P.first_after_diag.assign(nbk, 0);
    for (u32 bc = 0; bc < nbk; bc++)
    {
        u64 nu = lrowptr[bc + 1] - lrowptr[bc];
        u64 nl = (P.lcolptr[bc + 1] - P.lcolptr[bc]) - 1; // minus the diagonal block
        P.first_after_diag[bc] = P.colptr[bc] + nu;
        P.colptr[bc + 1] = P.colptr[bc] + nu + nl;
    }
    u64 nblk = P.colptr[nbk];
    P.rowidx.resize(nblk);
    P.nnz.resize(nblk);
    P.diag_lower_nnz.assign(nbk, 0);
    P.diag_upper_nnz.assign(nbk, 0);
    for (u32 bc = 0; bc < nbk; bc++)
    {
        u64 o = P.colptr[bc];
        for (u64 p = lrowptr[bc]; p < lrowptr[bc + 1]; p++)
        {
            P.rowidx[o] = lcolidx[p];
            P.nnz[o] = lrow_nnz[p];
            o++;
        }
        for (u64 p = P.lcolptr[bc]; p < P.lcolptr[bc + 1]; p++)
        {
            if (P.lrowidx[p] == bc)
            {
                u32 ndiag = std::min(nb, n - bc * nb);
                P.diag_lower_nnz[bc] = P.lnnz[p] - ndiag;
                P.diag_upper_nnz[bc] = P.lnnz[p];
                continue;
            }
            P.rowidx[o] = P.lrowidx[p];
            P.nnz[o] = P.lnnz[p];
            o++;
        }
    }
    // block-CSR with the map back into block-CSC order
    P.rowptr.assign(nbk + 1, 0);
    for (u64 b = 0; b < nblk; b++)
        P.rowptr[P.rowidx[b] + 1]++;
    for (u32 b = 0; b < nbk; b++)
        P.rowptr[b + 1] += P.rowptr[b];
    P.colidx.resize(nblk);
    P.csr_to_csc.resize(nblk);
    P.first_after_diag_csr.assign(nbk, 0);
    {
        std::vector<u64> cur(P.rowptr.begin(), P.rowptr.end() - 1);
        for (u32 bc = 0; bc < nbk; bc++)
            for (u64 b = P.colptr[bc]; b < P.colptr[bc + 1]; b++)
            {
                u32 br = P.rowidx[b];
                P.colidx[cur[br]] = bc;
                P.csr_to_csc[cur[br]] = b;
                cur[br]++;
            }
    }
    for (u32 br = 0; br < nbk; br++)
    {
        u64 f = P.rowptr[br];
        while (f < P.rowptr[br + 1] && P.colidx[f] < br)
            f++;
        P.first_after_diag_csr[br] = f;
    }
}

} // namespace pg
