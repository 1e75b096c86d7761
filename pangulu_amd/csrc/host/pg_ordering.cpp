// pg_ordering.cpp -- fill-reducing ordering: nested dissection with multilevel vertex separators.
//
// Reference counterpart (re-designed, not translated): src/pangulu_reordering.c:1065-1089 hands the graph of A + A^T to
// METIS_NodeND and :1130-1272 drives it.  There is no METIS in this stack; this file is the build's own dissection:
//   * WITHOUT coordinates (the SuiteSparse matrices of BASELINE.json arrive that way): multilevel vertex bisection --
//     heavy-edge matching down to a few hundred vertices, graph-growing separators from several seeds at the coarsest level,
//     two-sided vertex-separator FM refinement (hill climbing with rollback) at every level on the way up;
//   * WITH mesh coordinates: median cuts along 13 lattice directions (axes, face and space diagonals, in units of the mesh
//     spacing), the direction with the smallest boundary wins (axis planes on 27-point meshes, x + y + z = c on 7-point
//     ones, where a plane holds 4/3 of the diagonal's vertices), polished by the same FM refinement;
//   * separators are ordered so that the rows a descendant region touches are few long runs (k-d order on the coordinates,
//     or on pseudo-coordinates = positions of a vertex's neighbours in the two halves' own orderings), DESIGN.md 3.1;
//   * block alignment: the right child of every large split starts on a multiple of nb (padding positions), see below.
// The recursion builds a tree (OpenMP tasks), the emission pass walks it in order.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <numeric>
#include <queue>
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

const u32 NONE = 0xFFFFFFFFu;
const u32 kNoVertex = 0xFFFFFFFFu;

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

// deterministic generator (the ordering must not depend on the thread schedule: every region seeds its own from its vertices)
struct Rng
{
    u64 s;
    explicit Rng(u64 seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull) {}
    u32 next()
    {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (u32)(s >> 16);
    }
    u32 below(u32 m) { return m ? next() % m : 0; }
};

// ------------------------------------------------------------------------------------------------------------------
// graph of one region in local numbering, with the weights the coarsening accumulates
// ------------------------------------------------------------------------------------------------------------------
struct LGraph
{
    u32 n = 0;
    std::vector<u32> xadj, adj, adjw, vw;
    u64 total = 0;
};

// One level of heavy-edge matching: visit the vertices in a random order, light ones first; an unmatched vertex takes the
// unmatched neighbour behind its heaviest edge (a heavy edge stands for many fine edges: collapsing it keeps them out of every
// coarser cut).  Returns the coarse graph and the map fine -> coarse.
void coarsen(const LGraph &g, LGraph &c, std::vector<u32> &cmap, u32 maxvw, Rng &rng)
{
    const u32 n = g.n;
    std::vector<u32> order(n);
    {
        // random permutation, then a counting sort by capped degree (stable: random inside a bucket)
        std::vector<u32> rnd(n);
        std::iota(rnd.begin(), rnd.end(), 0u);
        for (u32 i = n; i > 1; i--)
            std::swap(rnd[i - 1], rnd[rng.below(i)]);
        const u32 cap = 64;
        std::vector<u32> cnt(cap + 2, 0);
        for (u32 v = 0; v < n; v++)
            cnt[std::min(g.xadj[v + 1] - g.xadj[v], cap) + 1]++;
        for (u32 d = 0; d <= cap; d++)
            cnt[d + 1] += cnt[d];
        for (u32 i = 0; i < n; i++)
        {
            const u32 v = rnd[i];
            order[cnt[std::min(g.xadj[v + 1] - g.xadj[v], cap)]++] = v;
        }
    }
    std::vector<u32> match(n, NONE), first;
    first.reserve(n / 2 + 16);
    cmap.assign(n, NONE);
    for (u32 i = 0; i < n; i++)
    {
        const u32 v = order[i];
        if (match[v] != NONE)
            continue;
        u32 best = NONE, bw = 0;
        for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
        {
            const u32 u = g.adj[p];
            if (match[u] == NONE && u != v && g.vw[v] + g.vw[u] <= maxvw && (best == NONE || g.adjw[p] > bw))
            {
                best = u;
                bw = g.adjw[p];
            }
        }
        if (best == NONE)
            match[v] = v;
        else
        {
            match[v] = best;
            match[best] = v;
        }
        cmap[v] = (u32)first.size();
        if (best != NONE)
            cmap[best] = (u32)first.size();
        first.push_back(v);
    }
    const u32 cn = (u32)first.size();
    c.n = cn;
    c.vw.assign(cn, 0);
    c.xadj.assign((size_t)cn + 1, 0);
    c.adj.clear();
    c.adjw.clear();
    c.adj.reserve(g.adj.size() / 2 + 16);
    c.adjw.reserve(g.adj.size() / 2 + 16);
    c.total = g.total;
    std::vector<u32> mark(cn, NONE); // position of a coarse neighbour in the row being built
    for (u32 cu = 0; cu < cn; cu++)
    {
        const u32 row = (u32)c.adj.size();
        const u32 pair[2] = {first[cu], match[first[cu]]};
        for (int t = 0; t < (pair[1] != pair[0] ? 2 : 1); t++)
        {
            const u32 w = pair[t];
            c.vw[cu] += g.vw[w];
            for (u32 p = g.xadj[w]; p < g.xadj[w + 1]; p++)
            {
                const u32 cx = cmap[g.adj[p]];
                if (cx == cu)
                    continue;
                const u32 m = mark[cx];
                if (m != NONE && m >= row && m < c.adj.size())
                    c.adjw[m] += g.adjw[p];
                else
                {
                    mark[cx] = (u32)c.adj.size();
                    c.adj.push_back(cx);
                    c.adjw.push_back(g.adjw[p]);
                }
            }
        }
        c.xadj[cu + 1] = (u32)c.adj.size();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Indistinguishable vertices -- equal CLOSED neighbourhoods inside the region: the unknowns of one mesh node in a problem with
// several degrees of freedom per node -- collapsed into one weighted vertex each (what METIS_NodeND does before it dissects,
// `compress`).  Some minimum separator keeps such a group on one side, so nothing is lost; what is gained is that a refinement
// move carries the whole node (moving one of three unknowns never has a positive gain: single-vertex FM is stuck in every
// local minimum of such a graph) and a graph a third the size.  elastic3d(77) without coordinates, 3 unknowns per node:
// root separator 20 493 -> see DESIGN.md 3.2.  Returns false (c untouched) when fewer than a fifth of the vertices would go.
// ------------------------------------------------------------------------------------------------------------------
bool compress_indistinguishable(const LGraph &g, LGraph &c, std::vector<u32> &cmap)
{
    const u32 n = g.n;
    auto mix = [](u64 x)
    {
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    std::vector<u64> h(n);
    for (u32 v = 0; v < n; v++)
    {
        u64 a = mix(v);
        for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
            a += mix(g.adj[p]);
        h[v] = a;
    }
    std::vector<u32> ord(n);
    std::iota(ord.begin(), ord.end(), 0u);
    auto deg = [&](u32 v)
    { return g.xadj[v + 1] - g.xadj[v]; };
    std::sort(ord.begin(), ord.end(), [&](u32 a, u32 b)
              {
                  if (h[a] != h[b])
                      return h[a] < h[b];
                  if (deg(a) != deg(b))
                      return deg(a) < deg(b);
                  return a < b; });
    std::vector<u32> leader(n);
    std::iota(leader.begin(), leader.end(), 0u);
    std::vector<u32> la, lb, heads;
    auto closed = [&](u32 v, std::vector<u32> &out)
    {
        out.assign(g.adj.begin() + g.xadj[v], g.adj.begin() + g.xadj[v + 1]);
        out.push_back(v);
        std::sort(out.begin(), out.end());
    };
    u32 merged = 0;
    for (u32 b = 0; b < n;)
    {
        u32 e = b + 1;
        while (e < n && h[ord[e]] == h[ord[b]] && deg(ord[e]) == deg(ord[b]))
            e++;
        if (e - b >= 2)
        {
            // (equal sums can collide: every member is compared with the heads of the classes found in the run so far)
            heads.assign(1, ord[b]);
            for (u32 k = b + 1; k < e; k++)
            {
                const u32 v = ord[k];
                closed(v, la);
                bool found = false;
                for (u32 hd : heads)
                {
                    closed(hd, lb);
                    if (la == lb)
                    {
                        leader[v] = hd; // (hd < v: the run is sorted by index)
                        merged++;
                        found = true;
                        break;
                    }
                }
                if (!found)
                    heads.push_back(v);
            }
        }
        b = e;
    }
    if ((u64)merged * 5 < (u64)n)
        return false;
    cmap.assign(n, NONE);
    u32 cn = 0;
    for (u32 v = 0; v < n; v++)
        cmap[v] = leader[v] == v ? cn++ : cmap[leader[v]];
    // members of every group, grouped (counting sort by group)
    std::vector<u32> start((size_t)cn + 1, 0), members(n);
    for (u32 v = 0; v < n; v++)
        start[cmap[v] + 1]++;
    for (u32 k = 0; k < cn; k++)
        start[k + 1] += start[k];
    {
        std::vector<u32> cur(start.begin(), start.end() - 1);
        for (u32 v = 0; v < n; v++)
            members[cur[cmap[v]]++] = v;
    }
    c.n = cn;
    c.vw.assign(cn, 0);
    c.xadj.assign((size_t)cn + 1, 0);
    c.adj.clear();
    c.adjw.clear();
    c.adj.reserve(g.adj.size() / 4 + 16);
    c.adjw.reserve(g.adj.size() / 4 + 16);
    c.total = g.total;
    std::vector<u32> mark(cn, NONE);
    for (u32 cu = 0; cu < cn; cu++)
    {
        const u32 row = (u32)c.adj.size();
        for (u32 k = start[cu]; k < start[cu + 1]; k++)
        {
            const u32 w = members[k];
            c.vw[cu] += g.vw[w];
            for (u32 p = g.xadj[w]; p < g.xadj[w + 1]; p++)
            {
                const u32 cx = cmap[g.adj[p]];
                if (cx == cu)
                    continue;
                const u32 m = mark[cx];
                if (m != NONE && m >= row && m < c.adj.size())
                    c.adjw[m] += g.adjw[p];
                else
                {
                    mark[cx] = (u32)c.adj.size();
                    c.adj.push_back(cx);
                    c.adjw.push_back(g.adjw[p]);
                }
            }
        }
        c.xadj[cu + 1] = (u32)c.adj.size();
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// Two-sided vertex-separator refinement (Fiduccia-Mattheyses on separators, as in multilevel nested dissection codes).
// where[v]: 0 / 1 = the sides, 2 = separator; invariant: no edge joins side 0 and side 1.  Moving a separator vertex v to
// side s pulls its neighbours on the other side into the separator: gain_s(v) = w(v) - sum of their weights.  A pass moves
// vertices by best gain (negative ones too), every vertex at most once, remembers the best state and rolls back to it.
// ------------------------------------------------------------------------------------------------------------------
struct FmScratch
{
    std::vector<i64> gain[2];
    std::vector<u32> lock; // pass stamp
    std::vector<u32> pulled;
    struct Move
    {
        u32 v, to, first_pulled;
    };
    std::vector<Move> log;
    u32 stamp = 0;
    void size(u32 n)
    {
        if (gain[0].size() < n)
        {
            gain[0].resize(n);
            gain[1].resize(n);
            lock.assign(n, 0);
            stamp = 0;
        }
    }
};

inline void weights_of(const LGraph &g, const std::vector<unsigned char> &where, u64 pw[3])
{
    pw[0] = pw[1] = pw[2] = 0;
    for (u32 v = 0; v < g.n; v++)
        pw[where[v]] += g.vw[v];
}

void node_fm(const LGraph &g, std::vector<unsigned char> &where, u64 pw[3], u64 maxpw, int passes, FmScratch &S)
{
    typedef std::pair<i64, u32> Entry; // (gain, vertex): the larger vertex id wins ties -- any fixed rule will do
    const u32 n = g.n;
    S.size(n);
    const u32 limit = (u32)std::min<u64>(std::max<u64>(n / 100, 30), 400);
    for (int pass = 0; pass < passes; pass++)
    {
        if (++S.stamp == 0)
        {
            std::fill(S.lock.begin(), S.lock.end(), 0u);
            S.stamp = 1;
        }
        const u32 stamp = S.stamp;
        std::priority_queue<Entry> Q[2];
        auto gains_of = [&](u32 v)
        {
            i64 g0 = g.vw[v], g1 = g.vw[v];
            for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
            {
                const u32 x = g.adj[p];
                if (where[x] == 1)
                    g0 -= g.vw[x];
                else if (where[x] == 0)
                    g1 -= g.vw[x];
            }
            S.gain[0][v] = g0;
            S.gain[1][v] = g1;
            Q[0].push(Entry(g0, v));
            Q[1].push(Entry(g1, v));
        };
        for (u32 v = 0; v < n; v++)
            if (where[v] == 2)
                gains_of(v);
        S.log.clear();
        S.pulled.clear();
        const u64 sep0 = pw[2];
        u64 best_sep = pw[2];
        u64 best_diff = pw[0] > pw[1] ? pw[0] - pw[1] : pw[1] - pw[0];
        size_t best_at = 0;
        auto valid_top = [&](int s) -> bool
        {
            while (!Q[s].empty())
            {
                const Entry e = Q[s].top();
                if (where[e.second] == 2 && S.lock[e.second] != stamp && S.gain[s][e.second] == e.first)
                    return true;
                Q[s].pop();
            }
            return false;
        };
        while (true)
        {
            const bool h0 = valid_top(0), h1 = valid_top(1);
            if (!h0 && !h1)
                break;
            int to;
            if (h0 && h1)
            {
                const i64 g0 = Q[0].top().first, g1 = Q[1].top().first;
                to = g0 > g1 ? 0 : (g1 > g0 ? 1 : (pw[0] <= pw[1] ? 0 : 1));
                if (pw[to] + g.vw[Q[to].top().second] > maxpw)
                    to = 1 - to;
            }
            else
                to = h0 ? 0 : 1;
            const u32 v = Q[to].top().second;
            Q[to].pop();
            if (pw[to] + g.vw[v] > maxpw)
            {
                // does not fit on this side (it stays available for the other one)
                if (!(h0 && h1))
                    continue;
                // both tops blocked: neither side takes its best vertex any more
                if (pw[1 - to] + g.vw[Q[1 - to].top().second] > maxpw)
                    Q[1 - to].pop();
                continue;
            }
            S.lock[v] = stamp;
            where[v] = (unsigned char)to;
            pw[2] -= g.vw[v];
            pw[to] += g.vw[v];
            S.log.push_back(FmScratch::Move{v, (u32)to, (u32)S.pulled.size()});
            for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
            {
                const u32 u = g.adj[p];
                if (where[u] == 1 - to)
                {
                    where[u] = 2;
                    pw[1 - to] -= g.vw[u];
                    pw[2] += g.vw[u];
                    S.pulled.push_back(u);
                    if (S.lock[u] != stamp)
                        gains_of(u);
                    for (u32 q = g.xadj[u]; q < g.xadj[u + 1]; q++)
                    {
                        const u32 x = g.adj[q];
                        if (where[x] == 2 && x != u && S.lock[x] != stamp)
                        {
                            // (x's gain towards `to` counted u on the other side; u is in the separator now.  A vertex pulled
                            //  earlier in this loop computed its gains before u changed: same correction.)
                            S.gain[to][x] += g.vw[u];
                            Q[to].push(Entry(S.gain[to][x], x));
                        }
                    }
                }
                else if (where[u] == 2 && S.lock[u] != stamp)
                {
                    S.gain[1 - to][u] -= g.vw[v]; // moving u to the other side would now pull v back
                    Q[1 - to].push(Entry(S.gain[1 - to][u], u));
                }
            }
            const u64 diff = pw[0] > pw[1] ? pw[0] - pw[1] : pw[1] - pw[0];
            if (pw[2] < best_sep || (pw[2] == best_sep && diff < best_diff))
            {
                best_sep = pw[2];
                best_diff = diff;
                best_at = S.log.size();
            }
            else if (S.log.size() - best_at > limit)
                break;
        }
        // roll back to the best state
        for (size_t i = S.log.size(); i > best_at; i--)
        {
            const FmScratch::Move &m = S.log[i - 1];
            const size_t pe = i < S.log.size() ? S.log[i].first_pulled : S.pulled.size();
            for (size_t k = m.first_pulled; k < pe; k++)
            {
                const u32 u = S.pulled[k];
                where[u] = (unsigned char)(1 - m.to);
                pw[2] -= g.vw[u];
                pw[1 - m.to] += g.vw[u];
            }
            where[m.v] = 2;
            pw[m.to] -= g.vw[m.v];
            pw[2] += g.vw[m.v];
        }
        (void)sep0;
        if (best_at == 0)
            break; // nothing gained in this pass
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Edge bisection of the coarse levels.  A coarse vertex is a clump of fine vertices and a coarse edge's weight is the number
// of fine edges between two clumps: the weighted edge cut of a coarse bisection IS the edge cut of the fine one, and the
// fine vertex separator is about cut / (edges a separator vertex has across).  The weight of a one-clump-thick vertex
// separator of a coarse graph says little (a fifth of all vertices at 150 clumps), so the coarse levels minimise the edge cut
// (2-way Fiduccia-Mattheyses with rollback) and only the finest levels work on the vertex separator itself.
// ------------------------------------------------------------------------------------------------------------------
i64 edge_cut(const LGraph &g, const std::vector<unsigned char> &side)
{
    i64 cut = 0;
    for (u32 v = 0; v < g.n; v++)
        for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
            if (side[g.adj[p]] != side[v])
                cut += g.adjw[p];
    return cut / 2;
}

void edge_fm(const LGraph &g, std::vector<unsigned char> &side, u64 pw[2], u64 maxpw, int passes, FmScratch &S)
{
    typedef std::pair<i64, u32> Entry;
    const u32 n = g.n;
    S.size(n);
    std::vector<i64> &gain = S.gain[0];
    const u32 limit = (u32)std::min<u64>(std::max<u64>(n / 50, 40), 1000);
    for (int pass = 0; pass < passes; pass++)
    {
        if (++S.stamp == 0)
        {
            std::fill(S.lock.begin(), S.lock.end(), 0u);
            S.stamp = 1;
        }
        const u32 stamp = S.stamp;
        std::priority_queue<Entry> Q[2]; // by the side a vertex is ON
        for (u32 v = 0; v < n; v++)
        {
            i64 ext = 0, in = 0;
            for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
                (side[g.adj[p]] != side[v] ? ext : in) += g.adjw[p];
            gain[v] = ext - in;
            if (ext > 0 || g.xadj[v] == g.xadj[v + 1])
                Q[side[v]].push(Entry(gain[v], v)); // boundary vertices (and isolated ones: free to balance with)
        }
        S.log.clear();
        i64 cur = 0, best = 0; // change of the cut since the start of the pass
        u64 best_diff = pw[0] > pw[1] ? pw[0] - pw[1] : pw[1] - pw[0];
        size_t best_at = 0;
        auto valid_top = [&](int s) -> bool
        {
            while (!Q[s].empty())
            {
                const Entry e = Q[s].top();
                if (side[e.second] == s && S.lock[e.second] != stamp && gain[e.second] == e.first)
                    return true;
                Q[s].pop();
            }
            return false;
        };
        while (true)
        {
            const bool h0 = valid_top(0), h1 = valid_top(1);
            if (!h0 && !h1)
                break;
            // from which side: the best gain whose move keeps the target side within its limit; an overweight side gives first
            int from;
            if (pw[0] > maxpw && h0)
                from = 0;
            else if (pw[1] > maxpw && h1)
                from = 1;
            else if (h0 && h1)
            {
                const i64 g0 = Q[0].top().first, g1 = Q[1].top().first;
                from = g0 > g1 ? 0 : (g1 > g0 ? 1 : (pw[0] >= pw[1] ? 0 : 1));
                if (pw[1 - from] + g.vw[Q[from].top().second] > maxpw)
                    from = 1 - from;
            }
            else
                from = h0 ? 0 : 1;
            if (!(from == 0 ? h0 : h1))
                break;
            const u32 v = Q[from].top().second;
            Q[from].pop();
            if (pw[1 - from] + g.vw[v] > maxpw && !(pw[from] > maxpw))
                continue; // does not fit over there
            S.lock[v] = stamp;
            side[v] = (unsigned char)(1 - from);
            pw[from] -= g.vw[v];
            pw[1 - from] += g.vw[v];
            cur -= gain[v];
            gain[v] = -gain[v];
            S.log.push_back(FmScratch::Move{v, (u32)from, 0});
            for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
            {
                const u32 u = g.adj[p];
                // u on v's old side: the edge became external; on the new side: internal
                gain[u] += side[u] == from ? 2 * (i64)g.adjw[p] : -2 * (i64)g.adjw[p];
                if (S.lock[u] != stamp)
                    Q[side[u]].push(Entry(gain[u], u));
            }
            const u64 diff = pw[0] > pw[1] ? pw[0] - pw[1] : pw[1] - pw[0];
            const bool feasible = std::max(pw[0], pw[1]) <= maxpw;
            if ((cur < best && (feasible || diff < best_diff)) || (cur == best && diff < best_diff))
            {
                best = cur;
                best_diff = diff;
                best_at = S.log.size();
            }
            else if (S.log.size() - best_at > limit)
                break;
        }
        for (size_t i = S.log.size(); i > best_at; i--)
        {
            const FmScratch::Move &m = S.log[i - 1];
            side[m.v] = (unsigned char)m.to; // (`to` holds the side the vertex came from)
            pw[1 - m.to] -= g.vw[m.v];
            pw[m.to] += g.vw[m.v];
        }
        if (best_at == 0)
            break;
    }
}

// bisection of a small graph: grow a region from a seed until it holds half the weight, refine the cut; best of several seeds
void initial_bisection(const LGraph &g, std::vector<unsigned char> &side, u64 pw[2], u64 maxpw, Rng &rng, FmScratch &S)
{
    const u32 n = g.n;
    std::vector<unsigned char> trial(n), best, seen(n);
    std::vector<u32> queue;
    u64 best_pw[2] = {0, 0};
    i64 best_cut = 0;
    bool have = false;
    static const int trials_env = getenv("PANGULU_AMD_ND_TRIALS") ? atoi(getenv("PANGULU_AMD_ND_TRIALS")) : 16;
    const int trials = n <= 8 ? 1 : std::min<int>(trials_env, (int)n);
    for (int t = 0; t < trials; t++)
    {
        std::fill(trial.begin(), trial.end(), (unsigned char)1);
        std::fill(seen.begin(), seen.end(), (unsigned char)0);
        queue.clear();
        u64 grown = 0;
        size_t head = 0;
        u32 scan = 0;
        u32 seed = rng.below(n);
        while (grown * 2 < g.total)
        {
            if (head == queue.size())
            {
                if (seen[seed])
                {
                    while (scan < n && seen[scan])
                        scan++;
                    if (scan == n)
                        break;
                    seed = scan;
                }
                seen[seed] = 1;
                queue.push_back(seed);
            }
            const u32 v = queue[head++];
            trial[v] = 0;
            grown += g.vw[v];
            for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
            {
                const u32 u = g.adj[p];
                if (!seen[u])
                {
                    seen[u] = 1;
                    queue.push_back(u);
                }
            }
        }
        u64 tpw[2] = {0, 0};
        for (u32 v = 0; v < n; v++)
            tpw[trial[v]] += g.vw[v];
        edge_fm(g, trial, tpw, maxpw, 8, S);
        const i64 cut = edge_cut(g, trial);
        const u64 big = std::max(tpw[0], tpw[1]);
        const bool ok = tpw[0] > 0 && tpw[1] > 0;
        const i64 pen = !ok ? (i64)1 << 60 : (big > maxpw ? (i64)(big - maxpw) * 64 : 0);
        if (!have || cut + pen < best_cut)
        {
            best = trial;
            best_cut = cut + pen;
            best_pw[0] = tpw[0];
            best_pw[1] = tpw[1];
            have = true;
        }
    }
    side = best;
    pw[0] = best_pw[0];
    pw[1] = best_pw[1];
}

// separator of a small graph: grow a region from a seed until it holds half the weight, take the lighter of the two
// boundaries, refine; best of several seeds
void initial_separator(const LGraph &g, std::vector<unsigned char> &where, u64 pw[3], u64 maxpw, Rng &rng, FmScratch &S)
{
    const u32 n = g.n;
    std::vector<unsigned char> trial(n), best;
    std::vector<u32> queue;
    std::vector<unsigned char> seen(n);
    u64 best_pw[3] = {0, 0, 0};
    bool have = false;
    static const int trials_env = getenv("PANGULU_AMD_ND_TRIALS") ? atoi(getenv("PANGULU_AMD_ND_TRIALS")) : 10;
    const int trials = n <= 8 ? 1 : std::min<int>(trials_env, (int)n);
    for (int t = 0; t < trials; t++)
    {
        std::fill(trial.begin(), trial.end(), (unsigned char)1);
        std::fill(seen.begin(), seen.end(), (unsigned char)0);
        queue.clear();
        u64 grown = 0;
        size_t head = 0;
        u32 scan = 0;
        u32 seed = rng.below(n);
        while (grown * 2 < g.total)
        {
            if (head == queue.size())
            {
                // first seed, or the component is exhausted: continue from an unseen vertex
                if (seen[seed])
                {
                    while (scan < n && seen[scan])
                        scan++;
                    if (scan == n)
                        break;
                    seed = scan;
                }
                seen[seed] = 1;
                queue.push_back(seed);
            }
            const u32 v = queue[head++];
            trial[v] = 0;
            grown += g.vw[v];
            for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
            {
                const u32 u = g.adj[p];
                if (!seen[u])
                {
                    seen[u] = 1;
                    queue.push_back(u);
                }
            }
        }
        // boundaries of the two sides
        u64 b0 = 0, b1 = 0;
        for (u32 v = 0; v < n; v++)
        {
            bool touches = false;
            for (u32 p = g.xadj[v]; p < g.xadj[v + 1] && !touches; p++)
                touches = (trial[g.adj[p]] & 1) != (trial[v] & 1);
            if (touches)
            {
                (trial[v] & 1 ? b1 : b0) += g.vw[v];
                trial[v] |= 4;
            }
        }
        const unsigned char sepmark = b0 <= b1 ? 4 : 5;
        for (u32 v = 0; v < n; v++)
            trial[v] = trial[v] == sepmark ? 2 : (trial[v] & 1);
        u64 tpw[3];
        weights_of(g, trial, tpw);
        node_fm(g, trial, tpw, maxpw, 4, S);
        auto score = [&](const u64 w[3]) -> std::pair<u64, u64>
        {
            const u64 big = std::max(w[0], w[1]);
            // (an empty side is no bisection; an overweight side counts against the candidate)
            const u64 pen = (w[0] == 0 || w[1] == 0) ? g.total : (big > maxpw ? big - maxpw : 0);
            return std::make_pair(w[2] + 4 * pen, big - std::min(w[0], w[1]));
        };
        if (!have || score(tpw) < score(best_pw))
        {
            best = trial;
            memcpy(best_pw, tpw, sizeof(tpw));
            have = true;
        }
    }
    where = best;
    memcpy(pw, best_pw, sizeof(best_pw));
}

// sides 0 / 1 without a separator -> the lighter boundary becomes the separator
void boundary_separator(const LGraph &g, std::vector<unsigned char> &where)
{
    u64 b0 = 0, b1 = 0;
    for (u32 v = 0; v < g.n; v++)
    {
        bool touches = false;
        for (u32 p = g.xadj[v]; p < g.xadj[v + 1] && !touches; p++)
            touches = (where[g.adj[p]] & 1) != (where[v] & 1);
        if (touches)
        {
            (where[v] & 1 ? b1 : b0) += g.vw[v];
            where[v] |= 4;
        }
    }
    const unsigned char sepmark = b0 <= b1 ? 4 : 5;
    for (u32 v = 0; v < g.n; v++)
        where[v] = where[v] == sepmark ? 2 : (where[v] & 1);
}

// rounds 1-3's bisection: level structure rooted at a pseudo-peripheral vertex, cut at the median level.  On 7-point meshes the
// level sets from a corner are the planes x + y + z = c, the smallest separators there are.
bool levelset_sides(const LGraph &g, std::vector<unsigned char> &where)
{
    const u32 m = g.n;
    std::vector<u32> level(m, NONE), queue;
    auto bfs = [&](u32 start) -> u32
    {
        std::fill(level.begin(), level.end(), NONE);
        queue.assign(1, start);
        level[start] = 0;
        u32 maxl = 0;
        for (size_t head = 0; head < queue.size(); head++)
        {
            const u32 v = queue[head];
            for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
                if (level[g.adj[p]] == NONE)
                {
                    level[g.adj[p]] = level[v] + 1;
                    maxl = std::max(maxl, level[v] + 1);
                    queue.push_back(g.adj[p]);
                }
        }
        return maxl + 1;
    };
    u32 nl = bfs(0);
    for (int it = 0; it < 2; it++)
    {
        const u32 nl2 = bfs(queue.back());
        const bool better = nl2 > nl;
        nl = nl2;
        if (!better)
            break;
    }
    if (nl < 3 || queue.size() < m)
        return false; // clique-like: nothing to gain (or not connected: the caller splits components first)
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
    where.resize(m);
    for (u32 v = 0; v < m; v++)
        where[v] = level[v] < cutl ? 0 : 1;
    return true;
}

// a < b: the better separator -- lighter; an empty side is no bisection, an overweight side counts against it
bool better_separator(const u64 a[3], const u64 b[3], u64 maxpw)
{
    auto cost = [&](const u64 w[3]) -> double
    {
        if (w[0] == 0 || w[1] == 0 || w[2] == 0)
            return 1e300;
        const u64 big = std::max(w[0], w[1]);
        return (double)w[2] + (big > maxpw ? 4.0 * (double)(big - maxpw) : 0.0);
    };
    const double ca = cost(a), cb = cost(b);
    if (ca != cb)
        return ca < cb;
    return std::max(a[0], a[1]) < std::max(b[0], b[1]);
}

// Multilevel vertex bisection of g0; where[] gets 0 / 1 / 2.  ONE coarsening hierarchy, a portfolio of ways up (the smallest
// separator wins):
//   edge : the coarse levels minimise the weighted edge cut, the finest level(s) turn the cut into a vertex separator and
//          refine that -- finds the planes of 27-point / finite-element meshes, where cut and separator go together;
//   node : vertex-separator refinement on every level -- finds what the edge cut cannot see (on a 7-point mesh the plane has
//          the fewest cut edges and 4/3 of the diagonal's vertices);
//   level: the level-structure bisection of rounds 1-3 on the fine graph, refined.
void multilevel_separator(const LGraph &g0, std::vector<unsigned char> &where, u64 pw[3], double ub, Rng &rng, int attempts)
{
    static const u32 coarsen_to = getenv("PANGULU_AMD_ND_COARSEN_TO") ? (u32)atoi(getenv("PANGULU_AMD_ND_COARSEN_TO")) : 160u;
    // levels (counted from the finest) on which the `edge` way refines the vertex separator itself
    static const size_t node_levels = getenv("PANGULU_AMD_ND_NODE_LEVELS") ? (size_t)atoi(getenv("PANGULU_AMD_ND_NODE_LEVELS")) : 1;
    static const double ub_edge = getenv("PANGULU_AMD_ND_UB_EDGE") ? atof(getenv("PANGULU_AMD_ND_UB_EDGE")) : 1.05;
    static const char *ways_env = getenv("PANGULU_AMD_ND_WAYS"); // any of "e", "n", "l" (default all three)
    static const bool way_edge = !ways_env || strchr(ways_env, 'e'), way_node = !ways_env || strchr(ways_env, 'n'),
                      way_level = !ways_env || strchr(ways_env, 'l');
    static const bool trace = getenv("PANGULU_AMD_ND_TRACE") && atoi(getenv("PANGULU_AMD_ND_TRACE")) >= 2;
    const bool tr = trace && g0.n >= 20000;
    const u64 maxpw_node = (u64)(ub * 0.5 * (double)g0.total) + 1;
    FmScratch S;
    u64 best_pw[3] = {0, 0, 0};
    std::vector<unsigned char> best;
    auto offer = [&](std::vector<unsigned char> &w, const u64 wpw[3], const char *name)
    {
        if (tr)
            fprintf(stderr, "[nd]   way %-5s: sides %llu %llu separator %llu\n", name, (unsigned long long)wpw[0], (unsigned long long)wpw[1],
                    (unsigned long long)wpw[2]);
        if (best.empty() || better_separator(wpw, best_pw, maxpw_node))
        {
            best.swap(w);
            memcpy(best_pw, wpw, sizeof(best_pw));
        }
    };
    // The attempts are independent: each coarsens anew from its own random stream (seeds drawn here, in order, so the result does
    // not depend on which thread runs which attempt) and they run as OpenMP tasks -- at the root of the dissection, where they are
    // asked for, nothing else is there to run beside them.  Their separators are compared afterwards, in attempt order.
    struct Attempt
    {
        std::vector<unsigned char> w_edge, w_node;
        u64 pw_edge[3] = {0, 0, 0}, pw_node[3] = {0, 0, 0};
        bool has_edge = false, has_node = false;
    };
    const int nattempts = std::max(1, attempts);
    std::vector<Attempt> results((size_t)nattempts);
    std::vector<u64> seeds((size_t)nattempts);
    for (int a = 0; a < nattempts; a++)
        seeds[a] = ((u64)rng.next() << 32) | rng.next();
    auto run_attempt = [&](int attempt)
    {
        Rng rng(seeds[attempt]);
        FmScratch S;
        Attempt &out = results[attempt];
        // (every attempt coarsens anew: the random matchings are where the variance between attempts comes from)
        std::vector<std::unique_ptr<LGraph>> levels;
        std::vector<std::vector<u32>> cmaps;
        const LGraph *cur = &g0;
        const u32 maxvw = (u32)std::max<u64>(1, (u64)(1.5 * (double)g0.total / coarsen_to));
        while (cur->n > coarsen_to)
        {
            std::unique_ptr<LGraph> c(new LGraph());
            std::vector<u32> cmap;
            coarsen(*cur, *c, cmap, maxvw, rng);
            if (c->n >= cur->n)
                break;
            const bool stalled = c->n > cur->n - cur->n / 20; // (nearly nothing matched: stars, or everything at the weight cap)
            levels.push_back(std::move(c));
            cmaps.push_back(std::move(cmap));
            cur = levels.back().get();
            if (stalled)
                break;
        }
        auto fine_of = [&](size_t l) -> const LGraph & { return l >= 2 ? *levels[l - 2] : g0; }; // the graph level l - 1 projects to
        if (way_edge)
        {
            std::vector<unsigned char> w;
            u64 epw[2], npw[3] = {0, 0, 0};
            auto edge_limit = [&](const LGraph &g) -> u64
            {
                u32 heaviest = 0;
                for (u32 v = 0; v < g.n; v++)
                    heaviest = std::max(heaviest, g.vw[v]);
                return std::max<u64>((u64)(ub_edge * 0.5 * (double)g.total) + 1, (g.total + 1) / 2 + heaviest);
            };
            initial_bisection(*cur, w, epw, edge_limit(*cur), rng, S);
            bool have_sep = false;
            if (levels.size() < node_levels)
            {
                boundary_separator(*cur, w);
                weights_of(*cur, w, npw);
                node_fm(*cur, w, npw, maxpw_node, 6, S);
                have_sep = true;
            }
            for (size_t l = levels.size(); l > 0; l--)
            {
                const LGraph &fine = fine_of(l);
                const std::vector<u32> &cmap = cmaps[l - 1];
                std::vector<unsigned char> wf(fine.n);
                for (u32 v = 0; v < fine.n; v++)
                    wf[v] = w[cmap[v]];
                w.swap(wf);
                if (!have_sep)
                {
                    epw[0] = epw[1] = 0;
                    for (u32 v = 0; v < fine.n; v++)
                        epw[w[v]] += fine.vw[v];
                    edge_fm(fine, w, epw, edge_limit(fine), 6, S);
                    if (l - 1 >= node_levels)
                        continue;
                    boundary_separator(fine, w);
                    have_sep = true;
                }
                weights_of(fine, w, npw);
                node_fm(fine, w, npw, maxpw_node, 6, S);
            }
            if (!have_sep)
            {
                boundary_separator(g0, w);
                weights_of(g0, w, npw);
                node_fm(g0, w, npw, maxpw_node, 6, S);
            }
            out.w_edge.swap(w);
            memcpy(out.pw_edge, npw, sizeof(npw));
            out.has_edge = true;
        }
        if (way_node)
        {
            std::vector<unsigned char> w;
            u64 npw[3];
            initial_separator(*cur, w, npw, maxpw_node, rng, S);
            for (size_t l = levels.size(); l > 0; l--)
            {
                const LGraph &fine = fine_of(l);
                const std::vector<u32> &cmap = cmaps[l - 1];
                std::vector<unsigned char> wf(fine.n);
                for (u32 v = 0; v < fine.n; v++)
                    wf[v] = w[cmap[v]];
                w.swap(wf);
                weights_of(fine, w, npw);
                node_fm(fine, w, npw, maxpw_node, 6, S);
            }
            out.w_node.swap(w);
            memcpy(out.pw_node, npw, sizeof(npw));
            out.has_node = true;
        }
    };
    const bool parallel_attempts = nattempts > 1 && g0.n >= 20000;
#pragma omp taskgroup
    {
        for (int attempt = 0; attempt < nattempts; attempt++)
        {
#pragma omp task default(shared) firstprivate(attempt) if (parallel_attempts)
            run_attempt(attempt);
        }
    }
    for (int attempt = 0; attempt < nattempts; attempt++)
    {
        if (results[attempt].has_edge)
            offer(results[attempt].w_edge, results[attempt].pw_edge, "edge");
        if (results[attempt].has_node)
            offer(results[attempt].w_node, results[attempt].pw_node, "node");
    }
    if (way_level || best.empty())
    {
        std::vector<unsigned char> w;
        u64 npw[3];
        if (levelset_sides(g0, w))
        {
            boundary_separator(g0, w);
            weights_of(g0, w, npw);
            node_fm(g0, w, npw, maxpw_node, 4, S);
            offer(w, npw, "level");
        }
    }
    where.swap(best);
    memcpy(pw, best_pw, sizeof(best_pw));
}

// ------------------------------------------------------------------------------------------------------------------
// The dissection.  A region is split into [left | right | separator]; the separator is ordered last so that its fill stays
// at the end of the region.
//
// Block alignment (the MI355X-first part): the solver tiles the matrix in regular nb x nb blocks, and a block that
// straddles two sibling subtrees chains them together -- the block-level task graph of an unaligned dissection is
// nearly one long chain of diagonal blocks, which starves a GPU.  With `align` > 0 the start of the right child of
// every large split is moved up to the next multiple of `align` by inserting padding positions (kNoVertex in the
// output; the caller turns them into isolated identity rows).  Large regions then start on block boundaries by
// induction, sibling subtrees share no block, and their panels can be batched into the same launches.
// ------------------------------------------------------------------------------------------------------------------
struct Node
{
    std::unique_ptr<Node> left, right;
    std::vector<u32> verts; // leaf: its vertices; otherwise the separator
    size_t nleft = 0, nright = 0;
    bool leaf = true;
};

struct Dissector
{
    const Graph &G;
    const double *xyz;
    int dim;
    u32 leaf;
    u32 align, align_min; // pad to `align` when both children have at least `align_min` vertices
    bool multilevel = true, kd = true, polish = true, diagonals = true, compare = false;
    double ub = 1.2;
    std::vector<u32> region; // id of the live region that owns a vertex (relaxed atomics: tasks read their neighbours' labels)
    std::vector<u32> local;  // local index of a vertex inside the region being split
    u32 next_region = 1;

    Dissector(const Graph &g, const double *c, int d, u32 leaf_size, u32 align_, u32 align_min_)
        : G(g), xyz(c), dim(d), leaf(leaf_size), align(align_), align_min(align_min_), region(g.n, 0), local(g.n, 0)
    {
        auto flag = [](const char *name, bool def)
        {
            const char *e = getenv(name);
            return e ? atoi(e) != 0 : def;
        };
        const char *m = getenv("PANGULU_AMD_ND_METHOD"); // multilevel (default) | levelset (rounds 1-3: BFS level structure, no refinement)
        multilevel = !(m && strcmp(m, "levelset") == 0);
        kd = !(getenv("PANGULU_AMD_SEPARATOR_ORDER") && strcmp(getenv("PANGULU_AMD_SEPARATOR_ORDER"), "natural") == 0);
        polish = flag("PANGULU_AMD_ND_POLISH", true);
        diagonals = flag("PANGULU_AMD_ND_DIAGONALS", true);
        compare = flag("PANGULU_AMD_ND_COMPARE", false); // with coordinates: also run the multilevel bisection, keep the smaller separator
        if (const char *e = getenv("PANGULU_AMD_ND_UB"))
            ub = std::max(1.01, atof(e));
    }

    // The separators near the root carry the large fronts: a region of at least an eighth of the graph gets several attempts
    // (other random matchings, run as concurrent tasks), the best separator is kept (PANGULU_AMD_ND_ATTEMPTS, default 8 -- 3 until the end of
    // round 4: elastic3d(77) without coordinates F 7.98e13 -> 7.65e13 --; 1 below that size).
    int attempts_for(u32 m) const
    {
        static const int top = getenv("PANGULU_AMD_ND_ATTEMPTS") ? atoi(getenv("PANGULU_AMD_ND_ATTEMPTS")) : 8;
        // ... and a region of at least a sixty-fourth of it three (PANGULU_AMD_ND_ATTEMPTS_MID): elastic3d(77) without coordinates F 7.64e13 -> 7.49e13,
        // fem27(112) 2.76e13 -> 2.69e13 (1.01x the geometric ordering's, both), no longer to compute: these regions are dissected by concurrent tasks anyway
        static const int mid = getenv("PANGULU_AMD_ND_ATTEMPTS_MID") ? atoi(getenv("PANGULU_AMD_ND_ATTEMPTS_MID")) : 3;
        return (u64)m * 8 >= (u64)G.n ? top : ((u64)m * 64 >= (u64)G.n ? mid : 1);
    }
    u32 new_region()
    {
        return __atomic_fetch_add(&next_region, 1u, __ATOMIC_RELAXED);
    }
    u32 region_of(u32 v) const { return __atomic_load_n(&region[v], __ATOMIC_RELAXED); }
    void relabel(const std::vector<u32> &vs, u32 rid)
    {
        for (u32 v : vs)
            __atomic_store_n(&region[v], rid, __ATOMIC_RELAXED);
    }

    // the region's graph in local numbering (vertex i = vs[i]); unit weights
    void extract(const std::vector<u32> &vs, u32 rid, LGraph &g)
    {
        const u32 m = (u32)vs.size();
        for (u32 i = 0; i < m; i++)
            local[vs[i]] = i;
        g.n = m;
        g.xadj.assign((size_t)m + 1, 0);
        g.vw.assign(m, 1);
        g.total = m;
        size_t cap = 0;
        for (u32 i = 0; i < m; i++)
            cap += (size_t)(G.ptr[vs[i] + 1] - G.ptr[vs[i]]);
        g.adj.clear();
        g.adj.reserve(cap);
        for (u32 i = 0; i < m; i++)
        {
            const u32 v = vs[i];
            for (u64 p = G.ptr[v]; p < G.ptr[v + 1]; p++)
            {
                const u32 w = G.adj[p];
                if (region_of(w) == rid)
                    g.adj.push_back(local[w]);
            }
            g.xadj[i + 1] = (u32)g.adj.size();
        }
        g.adjw.assign(g.adj.size(), 1);
    }

    // connected components of the region's graph: comp[i] for local vertex i, returns their number
    static u32 components(const LGraph &g, std::vector<u32> &comp)
    {
        comp.assign(g.n, NONE);
        std::vector<u32> stack;
        u32 nc = 0;
        for (u32 s = 0; s < g.n; s++)
        {
            if (comp[s] != NONE)
                continue;
            comp[s] = nc;
            stack.assign(1, s);
            while (!stack.empty())
            {
                const u32 v = stack.back();
                stack.pop_back();
                for (u32 p = g.xadj[v]; p < g.xadj[v + 1]; p++)
                    if (comp[g.adj[p]] == NONE)
                    {
                        comp[g.adj[p]] = nc;
                        stack.push_back(g.adj[p]);
                    }
            }
            nc++;
        }
        return nc;
    }

    // Geometric candidates: median cuts along lattice directions, in units of the mesh spacing along each axis (the largest
    // coordinate difference over an edge).  Only vertices within one edge length of the cut can be boundary vertices, so a
    // candidate costs a projection per vertex and an adjacency scan of a thin layer.  Returns false when the region has no extent.
    bool geometric_sides(const std::vector<u32> &vs, const LGraph &g, std::vector<unsigned char> &where)
    {
        const u32 m = (u32)vs.size();
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, h[3] = {0, 0, 0};
        for (u32 v : vs)
            for (int d = 0; d < dim; d++)
            {
                const double c = xyz[(size_t)v * dim + d];
                lo[d] = std::min(lo[d], c);
                hi[d] = std::max(hi[d], c);
            }
        int widest = 0;
        for (int d = 1; d < dim; d++)
            if (hi[d] - lo[d] > hi[widest] - lo[widest])
                widest = d;
        if (!(hi[widest] > lo[widest]))
            return false;
        for (u32 i = 0; i < m; i++)
            for (u32 p = g.xadj[i]; p < g.xadj[i + 1]; p++)
                for (int d = 0; d < dim; d++)
                    h[d] = std::max(h[d], std::fabs(xyz[(size_t)vs[i] * dim + d] - xyz[(size_t)vs[g.adj[p]] * dim + d]));
        // directions: the widest axis first (ties between candidates go to the earlier one), then the other axes, then diagonals
        int dirs[13][3];
        int nd = 0;
        for (int k = 0; k < dim; k++)
        {
            const int d = (widest + k) % dim;
            if (hi[d] > lo[d])
            {
                dirs[nd][0] = dirs[nd][1] = dirs[nd][2] = 0;
                dirs[nd][d] = 1;
                nd++;
            }
        }
        if (diagonals && dim >= 2)
        {
            static const int diag[10][3] = {{1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1}, {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {-1, 1, 1}};
            for (int k = 0; k < 10; k++)
            {
                bool ok = true;
                for (int d = 0; d < 3; d++)
                    if (diag[k][d] != 0 && (d >= dim || !(h[d] > 0) || !(hi[d] > lo[d])))
                        ok = false;
                if (ok)
                {
                    memcpy(dirs[nd], diag[k], sizeof(dirs[nd]));
                    nd++;
                }
            }
        }
        std::vector<double> proj(m), tmp(m);
        std::vector<unsigned char> trial(m);
        u32 best_size = NONE;
        for (int k = 0; k < nd; k++)
        {
            double reach = 0; // an edge moves the projection by at most this much
            double w[3] = {0, 0, 0};
            for (int d = 0; d < dim; d++)
                if (dirs[k][d] != 0)
                {
                    w[d] = h[d] > 0 ? dirs[k][d] / h[d] : (double)dirs[k][d];
                    reach += h[d] > 0 ? 1.0 : 0.0;
                }
            for (u32 i = 0; i < m; i++)
            {
                double s = 0;
                for (int d = 0; d < dim; d++)
                    s += w[d] * xyz[(size_t)vs[i] * dim + d];
                proj[i] = s;
            }
            tmp = proj;
            std::nth_element(tmp.begin(), tmp.begin() + m / 2, tmp.end());
            double cut = tmp[m / 2];
            const double pmin = *std::min_element(proj.begin(), proj.end());
            if (cut <= pmin)
            {
                // many ties at the low end: cut just above them
                double next = 1e300;
                for (double x : proj)
                    if (x > pmin)
                        next = std::min(next, x);
                if (next == 1e300)
                    continue;
                cut = next;
            }
            u32 b0 = 0, b1 = 0, n0 = 0;
            const double win = reach * 1.0001 + 1e-12;
            for (u32 i = 0; i < m; i++)
            {
                const unsigned char s = proj[i] < cut ? 0 : 1;
                trial[i] = s;
                n0 += s == 0;
            }
            if (n0 == 0 || n0 == m)
                continue;
            for (u32 i = 0; i < m; i++)
            {
                if (proj[i] < cut - win || proj[i] >= cut + win)
                    continue;
                bool touches = false;
                for (u32 p = g.xadj[i]; p < g.xadj[i + 1] && !touches; p++)
                    touches = trial[g.adj[p]] != trial[i];
                if (touches)
                    (trial[i] ? b1 : b0)++;
            }
            const u32 size = std::min(b0, b1);
            if (size > 0 && (best_size == NONE || size < best_size))
            {
                best_size = size;
                where = trial;
            }
            else if (size == 0 && best_size == NONE)
            {
                // (the cut falls between components: no separator needed -- cannot happen on a connected region)
            }
        }
        return best_size != NONE;
    }

    std::unique_ptr<Node> make_leaf(std::vector<u32> &vs)
    {
        std::unique_ptr<Node> nd(new Node());
        nd->verts.swap(vs);
        return nd;
    }

    // splits `vs` (all labelled `rid`) and recurses
    std::unique_ptr<Node> build(std::vector<u32> &vs, u32 rid)
    {
        const u32 m = (u32)vs.size();
        if (m <= leaf)
            return make_leaf(vs);
        LGraph g;
        extract(vs, rid, g);
        std::vector<unsigned char> where;
        std::vector<u32> L, R, S;
        {
            std::vector<u32> comp;
            const u32 nc = components(g, comp);
            if (nc > 1)
            {
                // disconnected: no separator.  Components to two groups, largest first to the lighter group.
                std::vector<u32> size(nc, 0), ord(nc);
                for (u32 i = 0; i < m; i++)
                    size[comp[i]]++;
                std::iota(ord.begin(), ord.end(), 0u);
                std::stable_sort(ord.begin(), ord.end(), [&](u32 a, u32 b)
                                 { return size[a] > size[b]; });
                std::vector<unsigned char> grp(nc, 0);
                u64 w0 = 0, w1 = 0;
                for (u32 c : ord)
                {
                    if (w0 <= w1)
                    {
                        grp[c] = 0;
                        w0 += size[c];
                    }
                    else
                    {
                        grp[c] = 1;
                        w1 += size[c];
                    }
                }
                for (u32 i = 0; i < m; i++)
                    (grp[comp[i]] ? R : L).push_back(vs[i]);
            }
        }
        if (L.empty())
        {
            Rng rng((u64)vs[0] * 2654435761u + m);
            bool have = false;
            u64 pw[3] = {0, 0, 0};
            if (xyz && dim > 0 && geometric_sides(vs, g, where))
            {
                boundary_separator(g, where);
                weights_of(g, where, pw);
                if (polish)
                {
                    FmScratch fs;
                    node_fm(g, where, pw, (u64)(ub * 0.5 * (double)g.total) + 1, 3, fs);
                }
                have = pw[0] > 0 && pw[1] > 0;
                if (have && compare && multilevel)
                {
                    std::vector<unsigned char> w2;
                    u64 pw2[3];
                    multilevel_separator(g, w2, pw2, ub, rng, attempts_for(m));
                    if (pw2[0] > 0 && pw2[1] > 0 && pw2[2] < pw[2])
                    {
                        where.swap(w2);
                        memcpy(pw, pw2, sizeof(pw));
                    }
                }
            }
            if (!have && multilevel)
            {
                static const bool compress_on = !(getenv("PANGULU_AMD_ND_COMPRESS") && atoi(getenv("PANGULU_AMD_ND_COMPRESS")) == 0);
                LGraph gc;
                std::vector<u32> cm;
                if (compress_on && m >= 1024 && compress_indistinguishable(g, gc, cm))
                {
                    // the separator of the compressed graph, carried back: a node's unknowns stay together
                    std::vector<unsigned char> wc;
                    multilevel_separator(gc, wc, pw, ub, rng, attempts_for(m));
                    where.resize(m);
                    for (u32 i = 0; i < m; i++)
                        where[i] = wc[cm[i]];
                }
                else
                    multilevel_separator(g, where, pw, ub, rng, attempts_for(m));
                have = pw[0] > 0 && pw[1] > 0 && pw[2] > 0;
            }
            if (!have && !multilevel && levelset_sides(g, where))
            {
                boundary_separator(g, where);
                weights_of(g, where, pw);
                have = pw[0] > 0 && pw[1] > 0 && pw[2] > 0;
            }
            if (!have)
                return make_leaf(vs); // clique-like, or no proper 3-way split: nothing to gain
            for (u32 i = 0; i < m; i++)
                (where[i] == 2 ? S : (where[i] == 0 ? L : R)).push_back(vs[i]);
        }
        static const bool trace = getenv("PANGULU_AMD_ND_TRACE") != nullptr;
        if (trace && m >= 4096)
            fprintf(stderr, "[nd] region of %u: left %zu right %zu separator %zu\n", m, L.size(), R.size(), S.size());
        // (scratch of this level goes before the recursion)
        g = LGraph();
        std::vector<unsigned char>().swap(where);
        std::vector<u32>().swap(vs);
        std::unique_ptr<Node> nd(new Node());
        nd->leaf = false;
        nd->nleft = L.size();
        nd->nright = R.size();
        const u32 rl = new_region(), rr = new_region(), rs = new_region();
        relabel(L, rl);
        relabel(R, rr);
        relabel(S, rs); // separators are final
        nd->verts.swap(S);
        Node *raw = nd.get();
        const bool spawn = std::min(L.size(), R.size()) >= 4096;
#pragma omp task default(shared) firstprivate(raw, rl) if (spawn)
        {
            raw->left = build(L, rl);
        }
        raw->right = build(R, rr);
#pragma omp taskwait
        return nd;
    }

    // ---- emission ----------------------------------------------------------------------------------------------
    std::vector<u32> out; // ordering being built: out[new] = old, or kNoVertex for padding
    std::vector<u32> pos; // position of an emitted vertex

    // Vertices taken out before the dissection (order_nested_dissection: simplicial vertices of degree one or two) go out right
    // before the first of their neighbours: ahead of ALL their neighbours, where their elimination creates no fill.
    const std::vector<unsigned char> *hanging = nullptr;
    double emitted_per_dissected = 1.0; // vertices of the matrix per vertex the dissection sees
    void emit_plain(const std::vector<u32> &vs)
    {
        for (u32 v : vs)
        {
            if (hanging)
                for (u64 p = G.ptr[v]; p < G.ptr[v + 1]; p++)
                {
                    const u32 w = G.adj[p];
                    if ((*hanging)[w] && pos[w] == NONE)
                    {
                        pos[w] = (u32)out.size();
                        out.push_back(w);
                    }
                }
            pos[v] = (u32)out.size();
            out.push_back(v);
        }
    }

    void maybe_align(size_t left, size_t right)
    {
        if (align == 0 || left < align_min || right < align_min)
            return;
        while (out.size() % align)
            out.push_back(kNoVertex);
    }

    // halve along the widest axis at the median, recursively, down to runs of at most 16 (keys: kdim doubles per entry)
    static void kd_order(u32 *v, const double *key, int kdim, std::vector<u32> &idx, size_t b, size_t e)
    {
        const size_t m = e - b;
        if (m <= 16)
            return;
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (size_t i = b; i < e; i++)
            for (int d = 0; d < kdim; d++)
            {
                const double c = key[(size_t)idx[i] * kdim + d];
                lo[d] = std::min(lo[d], c);
                hi[d] = std::max(hi[d], c);
            }
        int ax = 0;
        for (int d = 1; d < kdim; d++)
            if (hi[d] - lo[d] > hi[ax] - lo[ax])
                ax = d;
        if (!(hi[ax] > lo[ax]))
            return;
        // (ties broken by the other coordinates, then by vertex id: the order is a function of the keys alone)
        const size_t half = m / 2;
        std::nth_element(idx.begin() + b, idx.begin() + b + half, idx.begin() + e, [&](u32 a, u32 c)
                         {
                             for (int k = 0; k < kdim; k++)
                             {
                                 const int d = (ax + k) % kdim;
                                 const double ca = key[(size_t)a * kdim + d], cb = key[(size_t)c * kdim + d];
                                 if (ca != cb)
                                     return ca < cb;
                             }
                             return v[a] < v[c]; });
        kd_order(v, key, kdim, idx, b, b + half);
        kd_order(v, key, kdim, idx, b + half, e);
    }

    // A separator goes out in k-d order (PANGULU_AMD_SEPARATOR_ORDER=natural: as it came): the rows a descendant region touches
    // in this separator are (nearly) a box of its surface; in the mesh's lexicographic numbering a box is one short run per
    // mesh line -- most 16-row pieces of the factor blocks below then hold a few live rows --, in k-d order it is a few long
    // runs.  Same fill, same flops by the reference's count, fewer and fuller pieces.  With coordinates the keys are the
    // coordinates; without, the positions of the vertex's first neighbours in the left and in the right half's own ordering
    // (those orderings are dissections themselves: a descendant region is a run in them).
    void emit_separator(std::vector<u32> &S, size_t lbeg, size_t lend, size_t rbeg, size_t rend)
    {
        const size_t m = S.size();
        if (!kd || m <= 16)
        {
            emit_plain(S);
            return;
        }
        std::vector<double> key;
        int kdim = 0;
        if (xyz && dim > 1)
        {
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
                emit_plain(S);
                return;
            }
            kdim = dim;
            key.resize(m * (size_t)dim);
            for (size_t i = 0; i < m; i++)
                for (int d = 0; d < dim; d++)
                    key[i * dim + d] = xyz[(size_t)S[i] * dim + d];
        }
        else if (!xyz)
        {
            // PANGULU_AMD_SEPARATOR_ORDER_GRAPH: "surface" (default) only separators that are large for their region (|S|^2 >= 12
            // |region|: the surfaces of 3D regions, not the lines of shells and 2D meshes -- the counterpart of the bounding-box test
            // above), "kd" every separator, "natural" none.  Measured (profiles/r04b_orderings.log, graph-only orderings):
            // fem27(112) 750.9 ms against 901.1 in the matrix's own numbering, poisson3d(80) 63.5 against 74.9; shell(300) 27.2
            // against 25.7 -- lines stay as they come.
            static const char *mode_env = getenv("PANGULU_AMD_SEPARATOR_ORDER_GRAPH");
            static const int mode = !mode_env ? 1 : (strcmp(mode_env, "kd") == 0 ? 0 : (strcmp(mode_env, "natural") == 0 ? 2 : 1));
            const double region_size = (double)(lend - lbeg) + (double)(rend - rbeg) + (double)m;
            if (mode == 2 || (mode == 1 && (double)m * (double)m < 12.0 * region_size))
            {
                emit_plain(S);
                return;
            }
            kdim = 2;
            key.resize(m * 2);
            for (size_t i = 0; i < m; i++)
            {
                const u32 v = S[i];
                double kl = -1, kr = -1;
                for (u64 p = G.ptr[v]; p < G.ptr[v + 1]; p++)
                {
                    const u32 w = G.adj[p];
                    const size_t q = pos[w];
                    if (q == NONE)
                        continue;
                    if (q >= lbeg && q < lend)
                        kl = kl < 0 ? (double)q : std::min(kl, (double)q);
                    else if (q >= rbeg && q < rend)
                        kr = kr < 0 ? (double)q : std::min(kr, (double)q);
                }
                // scaled to the halves' lengths; a vertex that touches one half only sits where that half puts it
                const double sl = lend > lbeg ? (kl - (double)lbeg) / (double)(lend - lbeg) : 0.0;
                const double sr = rend > rbeg ? (kr - (double)rbeg) / (double)(rend - rbeg) : 0.0;
                key[2 * i] = kl < 0 ? (kr < 0 ? 0.0 : sr) : sl;
                key[2 * i + 1] = kr < 0 ? (kl < 0 ? 0.0 : sl) : sr;
            }
        }
        else
        {
            emit_plain(S);
            return;
        }
        std::vector<u32> idx(m);
        std::iota(idx.begin(), idx.end(), 0u);
        kd_order(S.data(), key.data(), kdim, idx, 0, m);
        std::vector<u32> T(m);
        for (size_t i = 0; i < m; i++)
            T[i] = S[idx[i]];
        emit_plain(T);
    }

    void emit(Node *nd)
    {
        if (nd->leaf)
        {
            emit_plain(nd->verts);
            return;
        }
        const size_t lbeg = out.size();
        emit(nd->left.get());
        const size_t lend = out.size();
        // (sizes as emitted: vertices taken out before the dissection go out with their first neighbour -- the left half's count is
        //  known, the right half's is scaled by the graph's ratio)
        maybe_align(lend - lbeg, (size_t)((double)nd->nright * emitted_per_dissected));
        const size_t rbeg = out.size();
        emit(nd->right.get());
        const size_t rend = out.size();
        emit_separator(nd->verts, lbeg, lend, rbeg, rend);
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
    Dissector D(G, coords, dim > 3 ? 3 : dim, leaf, align, align_min);
    // Simplicial vertices of degree one or two -- a vertex whose neighbours are adjacent to each other, e.g. a
    // constraint row of a KKT system that couples two neighbouring unknowns -- are taken out of the graph first (an independent set
    // of them: no two adjacent) and ordered ahead of their first neighbour: eliminating them there creates no fill at all, and the
    // dissection sees the graph that matters.  The multilevel separator search does badly with them in (on the nlpkkt-class
    // stand-in the clumps of the coarse levels straddle the two fields: kkt(48) 1.14x the geometric ordering's flops, and only
    // because the level-set way rescues it -- 2.0x / 3.6x for the node / edge ways alone; 1.01x without them).  With coordinates
    // the geometric cuts then work on the mesh of the unknowns that carry the fill.  PANGULU_AMD_ND_PREELIMINATE=0: off.
    std::vector<unsigned char> hanging;
    std::vector<u32> all;
    static const bool preeliminate = !(getenv("PANGULU_AMD_ND_PREELIMINATE") && atoi(getenv("PANGULU_AMD_ND_PREELIMINATE")) == 0);
    if (preeliminate)
    {
        hanging.assign(A.n, 0);
        u32 taken = 0;
        for (u32 v = 0; v < A.n; v++)
        {
            const u64 d = G.ptr[v + 1] - G.ptr[v];
            if (d == 0 || d > 2)
                continue;
            const u32 a = G.adj[G.ptr[v]], b = d == 2 ? G.adj[G.ptr[v] + 1] : a;
            if (hanging[a] || hanging[b])
                continue;
            if (d == 2 && !std::binary_search(G.adj.begin() + G.ptr[a], G.adj.begin() + G.ptr[a + 1], b))
                continue;
            hanging[v] = 1;
            taken++;
        }
        if (taken && taken < A.n)
        {
            all.reserve(A.n - taken);
            for (u32 v = 0; v < A.n; v++)
                if (!hanging[v])
                    all.push_back(v);
                else
                    D.region[v] = NONE; // (never a region's label: the dissection does not see the vertex)
            D.hanging = &hanging;
            D.emitted_per_dissected = (double)A.n / (double)(A.n - taken);
        }
    }
    if (all.empty())
    {
        all.resize(A.n);
        std::iota(all.begin(), all.end(), 0u);
    }
    std::unique_ptr<Node> root;
#pragma omp parallel
    {
#pragma omp single
        {
            root = D.build(all, 0);
        }
    }
    D.out.reserve((size_t)A.n + A.n / 8);
    D.pos.assign(A.n, NONE);
    D.emit(root.get());
    // padding positions become fresh vertex ids n, n+1, ... (isolated identity rows added by the caller)
    perm = std::move(D.out);
    u32 next = A.n;
    for (u32 &v : perm)
        if (v == kNoVertex)
            v = next++;
}

} // namespace pg
