// pg_analysis.cpp -- symbolic fill and block structure (the ordering lives in pg_ordering.cpp).
//
// Reference counterparts (re-designed, not translated):
//   symbolic      src/pangulu_symbolic.c:3-271 (pattern of A+A^T, column merge over the elimination tree)
//   block pattern src/pangulu_communication.c:792-1100 (block-CSC/CSR of the filled matrix)
#include <algorithm>
#include <cstring>
#include <cmath>
#include <memory>
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
    const double t_begin = wall_seconds();
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

    const double t_pattern = wall_seconds();
    // Column merge over the elimination tree: struct(L_j) = struct(S_j) U (U_{children c} struct(L_c) \ {c}).
    // Inside a supernode nothing is merged: column j has ONE child c = the previous column of the chain, and the matrix's own entries
    // of column j are in the child's structure already -- struct(L_j) is the child's sorted list minus its first entry, a SUFFIX of a
    // list that exists.  Such columns get a view (pointer + length) in O(own entries x log) and are materialised at the end, all
    // columns in parallel; only the heads of chains merge (marker array, sorted tail merged into the longest child's list).  Round 4:
    // the sequential merge-and-sort of every column was 25 of the 60 s pangulu_init took on the default bench matrix (3.4e9 entries).
    struct View
    {
        const u32 *p;
        u64 len;
    };
    std::vector<View> view(n);
    std::vector<std::unique_ptr<std::vector<u32>>> heads;
    std::vector<u32> mark(n, 0xFFFFFFFFu), first_child(n, 0xFFFFFFFFu), next_sib(n, 0xFFFFFFFFu), scratch;
    S.ptr.assign(n + 1, 0);
    i64 flop = 0;
    for (u32 j = 0; j < n; j++)
    {
        const u32 *own = sidx.data() + sptr[j]; // sorted, unique, starts with j (the diagonal is always there)
        const u64 nown = slen[j];
        const u32 c0 = first_child[j];
        bool chained = false;
        if (c0 != 0xFFFFFFFFu && next_sib[c0] == 0xFFFFFFFFu)
        {
            // one child: its list minus the child starts with j (the child's parent is its smallest off-diagonal row)
            const View suf{view[c0].p + 1, view[c0].len - 1};
            chained = true;
            for (u64 t = 0; t < nown && chained; t++)
                if (own[t] > j)
                    chained = std::binary_search(suf.p, suf.p + suf.len, own[t]);
            if (chained)
                view[j] = suf;
        }
        if (!chained)
        {
            std::unique_ptr<std::vector<u32>> list(new std::vector<u32>());
            std::vector<u32> &L = *list;
            mark[j] = j;
            L.push_back(j);
            // the longest child's list first: it is sorted; what the others and the matrix add is sorted by itself and merged in
            u32 longest = 0xFFFFFFFFu;
            for (u32 c = c0; c != 0xFFFFFFFFu; c = next_sib[c])
                if (longest == 0xFFFFFFFFu || view[c].len > view[longest].len)
                    longest = c;
            if (longest != 0xFFFFFFFFu)
            {
                L.reserve(view[longest].len + nown);
                for (u64 t = 1; t < view[longest].len; t++)
                {
                    const u32 i = view[longest].p[t];
                    if (i > j)
                    {
                        mark[i] = j;
                        L.push_back(i);
                    }
                }
            }
            const size_t mid = L.size();
            for (u64 t = 0; t < nown; t++)
                if (own[t] > j && mark[own[t]] != j)
                {
                    mark[own[t]] = j;
                    L.push_back(own[t]);
                }
            const size_t own_end = L.size();
            for (u32 c = c0; c != 0xFFFFFFFFu; c = next_sib[c])
            {
                if (c == longest)
                    continue;
                for (u64 t = 1; t < view[c].len; t++)
                {
                    const u32 i = view[c].p[t];
                    if (i > j && mark[i] != j)
                    {
                        mark[i] = j;
                        L.push_back(i);
                    }
                }
            }
            if (L.size() > mid)
            {
                if (L.size() > own_end) // (the matrix's own entries come sorted; other children's do not)
                    std::sort(L.begin() + mid, L.end());
                if (mid > 1)
                    std::inplace_merge(L.begin() + 1, L.begin() + mid, L.end());
            }
            view[j] = View{L.data(), (u64)L.size()};
            heads.push_back(std::move(list));
        }
        const u64 cj = view[j].len - 1;
        S.ptr[j + 1] = S.ptr[j] + view[j].len;
        flop += (i64)cj + 2 * (i64)cj * (i64)cj;
        if (cj > 0)
        {
            const u32 parent = view[j].p[1];
            next_sib[j] = first_child[parent];
            first_child[parent] = j;
        }
    }
    const double t_merge = wall_seconds();
    S.idx.resize(S.ptr[n]);
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 j = 0; j < (i64)n; j++)
        std::copy(view[j].p, view[j].p + view[j].len, S.idx.begin() + (i64)S.ptr[j]);
    const size_t nheads = heads.size();
    heads.clear();
    S.symbolic_nnz = 2 * (u64)S.idx.size() - n;
    S.flop = flop;
    if (getenv("PANGULU_AMD_TRACE"))
        fprintf(stderr, "[pangulu_amd trace] symbolic: pattern of A + A^T %.2f s, merges of %zu chain heads %.2f s, columns written %.2f s (%.1f M entries)\n",
                t_pattern - t_begin, nheads, t_merge - t_pattern, wall_seconds() - t_merge, 1e-6 * (double)S.idx.size());
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
