// pg_scaling.cpp -- maximum-product matching with row / column scaling (the job the reference gives to its MC64 port,
// src/pangulu_reordering.c:149-681, driver :1130-1272).
//
// Without pivoting the factorisation needs large entries on the diagonal.  For A with entries a_ij this finds a
// column permutation Q and positive diagonal scalings Dr, Dc such that A1 = Dr A Dc Q has |a1_ii| = 1 and |a1_ij| <= 1
// (Duff & Koster, "On algorithms for permuting large entries to the diagonal of a sparse matrix", 2001: the objective of
// MC64 job 5).  With costs c_ij = log(max_i |a_ij|) - log |a_ij| >= 0 this is a minimum-sum assignment; it is solved by
// shortest augmenting paths on the sparse graph (Dijkstra with dual variables u_i, v_j: reduced costs c_ij - u_i - v_j
// stay non-negative, matched entries are tight), after a greedy start on tight entries.  The duals give the scalings:
// dr_i = exp(u_i), dc_j = exp(v_j) / max_i |a_ij|.  Own implementation of the published algorithm; the reference's file
// is a translation of the HSL routine and was not consulted for the code.
//
// A x = b then becomes A1 y = Dr b with x[q(i)] = dc[q(i)] y[i]  (q(i) = the column matched to row i).
#include <cmath>
#include <limits>
#include <queue>

#include "pg_host.h"

namespace pg
{

namespace
{
inline double absval(const val_t &v)
{
#ifdef PANGULU_COMPLEX
    return std::hypot((double)v.re, (double)v.im);
#else
    return std::fabs((double)v);
#endif
}
} // namespace

bool max_product_matching(const CscMatrix &A, std::vector<u32> &col_of_row, std::vector<double> &dr, std::vector<double> &dc)
{
    const u32 n = A.n;
    const u64 nnz = A.nnz();
    const double INF = std::numeric_limits<double>::infinity();
    const u32 NONE = ~0u;
    std::vector<double> cost(nnz), colmax(n, 0.0), u(n, INF), v(n, INF);
    for (u32 j = 0; j < n; j++)
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            colmax[j] = std::max(colmax[j], absval(A.value[p]));
    for (u32 j = 0; j < n; j++)
    {
        if (!(colmax[j] > 0))
            return false; // an empty (or all-zero) column: numerically singular
        const double lm = std::log(colmax[j]);
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
        {
            const double a = absval(A.value[p]);
            cost[p] = a > 0 ? lm - std::log(a) : INF;
            if (cost[p] < u[A.rowidx[p]])
                u[A.rowidx[p]] = cost[p];
        }
    }
    for (u32 i = 0; i < n; i++)
        if (u[i] == INF)
            return false; // an all-zero row
    for (u32 j = 0; j < n; j++)
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            if (cost[p] < INF)
                v[j] = std::min(v[j], cost[p] - u[A.rowidx[p]]);
    // greedy start: a column takes a free row along a tight entry (the diagonal first, so that matrices that need no
    // permutation keep theirs)
    std::vector<u32> match_col(n, NONE), match_row(n, NONE); // column -> row, row -> column
    const double tight = 1e-12;
    for (int pass = 0; pass < 2; pass++)
        for (u32 j = 0; j < n; j++)
        {
            if (match_col[j] != NONE)
                continue;
            for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            {
                const u32 i = A.rowidx[p];
                if (pass == 0 && i != j)
                    continue;
                if (match_row[i] == NONE && cost[p] < INF && cost[p] - u[i] - v[j] <= tight)
                {
                    match_col[j] = i;
                    match_row[i] = j;
                    break;
                }
            }
        }
    // shortest augmenting paths for the columns still free
    std::vector<double> d(n, INF);
    std::vector<u32> pred(n, NONE), touched, scanned;
    std::vector<char> done(n, 0);
    typedef std::pair<double, u32> HeapItem;
    std::priority_queue<HeapItem, std::vector<HeapItem>, std::greater<HeapItem>> heap;
    for (u32 j0 = 0; j0 < n; j0++)
    {
        if (match_col[j0] != NONE)
            continue;
        touched.clear();
        scanned.clear();
        while (!heap.empty())
            heap.pop();
        u32 j = j0, end_row = NONE;
        double dj = 0, delta = 0;
        for (;;)
        {
            for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++)
            {
                const u32 i = A.rowidx[p];
                if (done[i] || !(cost[p] < INF))
                    continue;
                const double nd = dj + std::max(0.0, cost[p] - u[i] - v[j]);
                if (nd < d[i])
                {
                    if (d[i] == INF)
                        touched.push_back(i);
                    d[i] = nd;
                    pred[i] = j;
                    heap.push(HeapItem(nd, i));
                }
            }
            u32 i = NONE;
            while (!heap.empty())
            {
                const HeapItem t = heap.top();
                heap.pop();
                if (!done[t.second] && t.first == d[t.second])
                {
                    i = t.second;
                    break;
                }
            }
            if (i == NONE)
                break; // no augmenting path: structurally singular
            if (match_row[i] == NONE)
            {
                end_row = i;
                delta = d[i];
                break;
            }
            done[i] = 1;
            scanned.push_back(i);
            j = match_row[i];
            dj = d[i];
        }
        if (end_row == NONE)
            return false;
        // duals: scanned rows and the columns of the search tree move so that the path becomes tight and every reduced
        // cost stays non-negative
        for (u32 i : scanned)
        {
            const double t = delta - d[i];
            u[i] -= t;
            v[match_row[i]] += t;
        }
        v[j0] += delta;
        // augment along the predecessors
        for (u32 i = end_row;;)
        {
            const u32 jc = pred[i], next = match_col[jc];
            match_row[i] = jc;
            match_col[jc] = i;
            if (jc == j0)
                break;
            i = next;
        }
        for (u32 i : touched)
        {
            d[i] = INF;
            pred[i] = NONE;
            done[i] = 0;
        }
    }
    col_of_row.resize(n);
    dr.resize(n);
    dc.resize(n);
    for (u32 i = 0; i < n; i++)
    {
        col_of_row[i] = match_row[i];
        dr[i] = std::exp(u[i]);
        dc[i] = std::exp(v[i]) / colmax[i];
        if (!std::isfinite(dr[i]) || !std::isfinite(dc[i]) || !(dr[i] > 0) || !(dc[i] > 0))
            return false;
    }
    return true;
}

// A1 = Dr A Dc Q: column i of A1 is column col_of_row[i] of A, scaled
void apply_matching(const CscMatrix &A, const std::vector<u32> &col_of_row, const std::vector<double> &dr, const std::vector<double> &dc,
                    CscMatrix &out)
{
    const u32 n = A.n;
    out.n = n;
    out.colptr.assign((size_t)n + 1, 0);
    for (u32 i = 0; i < n; i++)
        out.colptr[i + 1] = out.colptr[i] + (A.colptr[col_of_row[i] + 1] - A.colptr[col_of_row[i]]);
    out.rowidx.resize(A.nnz());
    out.value.resize(A.nnz());
    for (u32 i = 0; i < n; i++)
    {
        const u32 j = col_of_row[i];
        u64 o = out.colptr[i];
        for (u64 p = A.colptr[j]; p < A.colptr[j + 1]; p++, o++)
        {
            const u32 r = A.rowidx[p];
            const double s = dr[r] * dc[j];
            out.rowidx[o] = r;
#ifdef PANGULU_COMPLEX
            out.value[o] = val_t{(calculate_real_type)(A.value[p].re * s), (calculate_real_type)(A.value[p].im * s)};
#else
            out.value[o] = (val_t)(A.value[p] * s);
#endif
        }
    }
}

} // namespace pg
