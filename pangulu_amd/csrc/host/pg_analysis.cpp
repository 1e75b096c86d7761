// pg_analysis.cpp -- ordering, symbolic fill and block structure.
//
// Reference counterparts (re-designed, not translated):
//   ordering      src/pangulu_reordering.c:1130-1272 (METIS/MC64 driver; identity when neither is compiled in)
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

void order_identity(u32 n, std::vector<u32> &perm)
{
    perm.resize(n);
    std::iota(perm.begin(), perm.end(), 0u);
}

namespace
{

// adjacency of A + A^T without the diagonal
struct Graph
{
    u32 n = 0;
    std::vector<u64> ptr;
    std::vector<u32> adj;
};

void build_graph(const CscMatrix &A, Graph &G)
{
    u32 n = A.n;
    G.n = n;
    std::vector<u64> cnt(n + 1, 0);
    for (u32 j = 0; j < n; j++)
    {
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
        {
            u32 i = A.rowidx[p];
            if (i != j)
            {
                cnt[i + 1]++;
                cnt[j + 1]++;
            }
        }
    }
    for (u32 i = 0; i < n; i++)
        cnt[i + 1] += cnt[i];
    std::vector<u32> raw(cnt[n]);
    std::vector<u64> cur(cnt.begin(), cnt.end() - 1);
    for (u32 j = 0; j < n; j++)
    {
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
        {
            u32 i = A.rowidx[p];
            if (i != j)
            {
                raw[cur[i]++] = j;
                raw[cur[j]++] = i;
            }
        }
    }
    // sort + unique every list
    G.ptr.assign(n + 1, 0);
#pragma omp parallel for schedule(dynamic, 1024)
    for (i64 v = 0; v < (i64)n; v++)
    {
        u32 *b = raw.data() + cnt[v], *e = raw.data() + cnt[v + 1];
        std::sort(b, e);
        G.ptr[v + 1] = (u64)(std::unique(b, e) - b);
    }
    for (u32 v = 0; v < n; v++)
        G.ptr[v + 1] += G.ptr[v];
    G.adj.resize(G.ptr[n]);
#pragma omp parallel for schedule(dynamic, 1024)
    for (i64 v = 0; v < (i64)n; v++)
    {
        u64 len = G.ptr[v + 1] - G.ptr[v];
        std::copy(raw.data() + cnt[v], raw.data() + cnt[v] + len, G.adj.data() + G.ptr[v]);
    }
}

// Nested dissection with vertex separators taken from a bisection's boundary.  A region is split into
// [left | right | separator]; the separator is ordered last so that its fill stays at the end of the region.
//
// Block alignment (the MI355X-first part): the solver tiles the matrix in regular nb x nb blocks, and a block that
// straddles two sibling subtrees chains them together -- the block-level task graph of an unaligned dissection is
// nearly one long chain of diagonal blocks, which starves a GPU.  With `align` > 0 the start of the right child of
// every large split is moved up to the next multiple of `align` by inserting padding positions (kNoVertex in the
// output; the caller turns them into isolated identity rows).  Large regions then start on block boundaries by
// induction, sibling subtrees share no block, and their panels can be batched into the same launches.
const u32 kNoVertex = 0xFFFFFFFFu;

struct Dissector
{
    const Graph &G;
    const double *xyz;
    int dim;
    u32 leaf;
    u32 align, align_min; // pad to `align` when both children have at least `align_min` vertices
    std::vector<u32> out;        // ordering being built: out[new] = old, or kNoVertex for padding
    std::vector<u32> region;     // region id of every vertex (which live region currently owns it)
    std::vector<u32> level;      // BFS scratch
    std::vector<u32> queue, seen, side;
    u32 next_region = 1, stamp = 0;

    Dissector(const Graph &g, const double *c, int d, u32 leaf_size, u32 align_, u32 align_min_)
        : G(g), xyz(c), dim(d), leaf(leaf_size), align(align_), align_min(align_min_), region(g.n, 0), level(g.n, 0), seen(g.n, 0), side(g.n, 0)
    {
        queue.reserve(g.n);
        out.reserve(g.n + g.n / 8);
    }

    // BFS inside region `rid` from `start`; fills queue (visit order) and level[]; returns number of levels
    u32 bfs(u32 start, u32 rid)
    {
        stamp++;
        queue.clear();
        queue.push_back(start);
        seen[start] = stamp;
        level[start] = 0;
        size_t head = 0;
        u32 maxl = 0;
        while (head < queue.size())
        {
            u32 v = queue[head++];
            for (u64 p = G.ptr[v]; p < G.ptr[v + 1]; p++)
            {
                u32 w = G.adj[p];
                if (region[w] == rid && seen[w] != stamp)
                {
                    seen[w] = stamp;
                    level[w] = level[v] + 1;
                    maxl = std::max(maxl, level[w]);
                    queue.push_back(w);
                }
            }
        }
        return maxl + 1;
    }

    void emit(const std::vector<u32> &vs) { out.insert(out.end(), vs.begin(), vs.end()); }

    // A separator goes out in k-d order when there are coordinates (PANGULU_AMD_SEPARATOR_ORDER=natural: as it came, i.e. in the
    // mesh's own numbering): halve it along its widest axis at the median, recursively, down to runs of at most 16.  The
    // rows a descendant region touches in this separator are (nearly) an axis-aligned box; in the mesh's lexicographic
    // numbering a box is one short run per mesh line -- most 16-row pieces of the factor blocks below then hold a few live
    // rows --, in k-d order it is a few long runs.  Same fill, same flops by the reference's count, fewer and fuller pieces.
    void emit_separator(std::vector<u32> &S)
    {
        static const bool kd = !(getenv("PANGULU_AMD_SEPARATOR_ORDER") && strcmp(getenv("PANGULU_AMD_SEPARATOR_ORDER"), "natural") == 0);
        if (!kd || !xyz || dim <= 1 || S.size() <= 16)
        {
            emit(S);
            return;
        }
        // (only separators that are surfaces: a line of vertices -- the separators of a shell or a 2D mesh -- is in the best order
        //  as it comes; second-largest extent of the bounding box at least a sixteenth of the largest)
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (u32 v : S)
            for (int d = 0; d < dim; d++)
            {
                const double c = xyz[(size_t)v * dim + d];
                lo[d] = std::min(lo[d], c);
                hi[d] = std::max(hi[d], c);
            }
        double ext[3] = {0, 0, 0};
        for (int d = 0; d < dim; d++)
            ext[d] = hi[d] - lo[d];
        std::sort(ext, ext + dim);
        if (!(ext[dim - 2] * 16.0 >= ext[dim - 1]) || !(ext[dim - 1] > 0))
        {
            emit(S);
            return;
        }
        kd_order(S.data(), S.size());
        emit(S);
    }
    void kd_order(u32 *v, size_t m)
    {
        if (m <= 16)
            return;
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (size_t i = 0; i < m; i++)
            for (int d = 0; d < dim; d++)
            {
                const double c = xyz[(size_t)v[i] * dim + d];
                lo[d] = std::min(lo[d], c);
                hi[d] = std::max(hi[d], c);
            }
        int ax = 0;
        for (int d = 1; d < dim; d++)
            if (hi[d] - lo[d] > hi[ax] - lo[ax])
                ax = d;
        if (!(hi[ax] > lo[ax]))
            return;
        // (ties broken by the other coordinates, then by vertex id: the order is a function of the geometry alone)
        const size_t half = m / 2;
        std::nth_element(v, v + half, v + m, [&](u32 a, u32 b)
                         {
                             for (int k = 0; k < dim; k++)
                             {
                                 const int d = (ax + k) % dim;
                                 const double ca = xyz[(size_t)a * dim + d], cb = xyz[(size_t)b * dim + d];
                                 if (ca != cb)
                                     return ca < cb;
                             }
                             return a < b; });
        kd_order(v, half);
        kd_order(v + half, m - half);
    }

    void relabel(const std::vector<u32> &vs, u32 rid)
    {
        for (u32 v : vs)
            region[v] = rid;
    }

    // orders the vertices of `vs` (all labelled `rid`) behind what is already in `out`
    void order(std::vector<u32> &vs, u32 rid)
    {
        const u32 m = (u32)vs.size();
        if (m <= leaf)
        {
            emit(vs);
            return;
        }
        bool have_sides = false;
        if (xyz && dim > 0)
        {
            // geometric: cut the widest axis at the median coordinate
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (u32 v : vs)
                for (int d = 0; d < dim; d++)
                {
                    double c = xyz[(size_t)v * dim + d];
                    lo[d] = std::min(lo[d], c);
                    hi[d] = std::max(hi[d], c);
                }
            int ax = 0;
            for (int d = 1; d < dim; d++)
                if (hi[d] - lo[d] > hi[ax] - lo[ax])
                    ax = d;
            if (hi[ax] > lo[ax])
            {
                std::vector<u32> tmp(vs);
                std::nth_element(tmp.begin(), tmp.begin() + m / 2, tmp.end(), [&](u32 a, u32 b)
                                 { return xyz[(size_t)a * dim + ax] < xyz[(size_t)b * dim + ax]; });
                double cut = xyz[(size_t)tmp[m / 2] * dim + ax];
                if (cut <= lo[ax])
                    cut = std::nextafter(lo[ax], hi[ax]); // many ties at the low end: cut just above them
                for (u32 v : vs)
                    side[v] = xyz[(size_t)v * dim + ax] < cut ? 0 : 1;
                have_sides = true;
            }
        }
        if (!have_sides)
        {
            // graph: level structure rooted at a pseudo-peripheral vertex
            u32 nl = bfs(vs[0], rid);
            for (int it = 0; it < 2; it++)
            {
                u32 nl2 = bfs(queue.back(), rid);
                bool better = nl2 > nl;
                nl = nl2;
                if (!better)
                    break;
            }
            if ((u32)queue.size() < m)
            {
                // disconnected region: the component just found and the rest are independent, no separator
                std::vector<u32> comp(queue), rest;
                rest.reserve(m - comp.size());
                u32 rc = next_region++, rr = next_region++;
                relabel(comp, rc);
                for (u32 v : vs)
                    if (region[v] == rid)
                        rest.push_back(v);
                relabel(rest, rr);
                std::vector<u32>().swap(vs);
                order(comp, rc);
                maybe_align(comp.size(), rest.size());
                order(rest, rr);
                return;
            }
            if (nl < 3)
            {
                emit(vs); // clique-like: nothing to gain
                return;
            }
            std::vector<u32> lcount(nl, 0);
            for (u32 v : queue)
                lcount[level[v]]++;
            u32 acc = 0, cutl = 1;
            for (u32 l = 0; l < nl; l++)
            {
                acc += lcount[l];
                if (acc * 2 >= m)
                {
                    cutl = l;
                    break;
                }
            }
            cutl = std::min(std::max(cutl, 1u), nl - 2);
            for (u32 v : vs)
                side[v] = level[v] < cutl ? 0 : 1;
        }
        // vertex separator = boundary of the left side or of the right side, whichever is smaller
        u32 bl = 0, br = 0;
        for (u32 v : vs)
        {
            bool touches = false;
            for (u64 p = G.ptr[v]; p < G.ptr[v + 1] && !touches; p++)
            {
                u32 w = G.adj[p];
                touches = (region[w] == rid) && ((side[w] & 1u) != (side[v] & 1u));
            }
            if (touches)
            {
                if ((side[v] & 1u) == 0)
                {
                    bl++;
                    side[v] = 4; // left boundary (bit 0 clear)
                }
                else
                {
                    br++;
                    side[v] = 5; // right boundary (bit 0 set)
                }
            }
        }
        const u32 sepmark = (bl <= br) ? 4u : 5u;
        std::vector<u32> L, R, S;
        for (u32 v : vs)
        {
            u32 s = side[v];
            if (s == sepmark)
                S.push_back(v);
            else if ((s & 1u) == 0)
                L.push_back(v);
            else
                R.push_back(v);
        }
        if (S.empty() || L.empty() || R.empty())
        {
            emit(vs); // no proper 3-way split (e.g. the boundary swallowed a side)
            return;
        }
        std::vector<u32>().swap(vs);
        u32 rl = next_region++, rr = next_region++, rs = next_region++;
        relabel(L, rl);
        relabel(R, rr);
        relabel(S, rs); // separators are final
        size_t nl_ = L.size(), nr_ = R.size();
        order(L, rl);
        maybe_align(nl_, nr_);
        order(R, rr);
        emit_separator(S);
    }

    void maybe_align(size_t left, size_t right)
    {
        if (align == 0 || left < align_min || right < align_min)
            return;
        while (out.size() % align)
            out.push_back(kNoVertex);
    }
};

} // namespace

void order_nested_dissection(const CscMatrix &A, const double *coords, int dim, u32 align, std::vector<u32> &perm)
{
    Graph G;
    build_graph(A, G);
    const char *leaf_env = getenv("PANGULU_AMD_ND_LEAF");
    u32 leaf = leaf_env ? (u32)atoi(leaf_env) : 96u;
    if (leaf < 4)
        leaf = 4;
    const char *amin_env = getenv("PANGULU_AMD_ND_ALIGN_MIN_BLOCKS");
    u32 align_min = align * (amin_env ? (u32)atoi(amin_env) : 8u);
    Dissector D(G, coords, dim, leaf, align, align_min);
    std::vector<u32> all(A.n);
    std::iota(all.begin(), all.end(), 0u);
    D.order(all, 0);
    // padding positions become fresh vertex ids n, n+1, ... (isolated identity rows added by the caller)
    perm = std::move(D.out);
    u32 next = A.n;
    for (u32 &v : perm)
        if (v == kNoVertex)
            v = next++;
}

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
