/* Built-in fill-reducing ordering: nested dissection by breadth-first level
 * structures.
 *
 * The reference orders with AMD / COLAMD / METIS / its own NESDIS
 * (Cholesky/cholmod_analyze.c:569-804, Partition/cholmod_nesdis.c); none of
 * those packages is part of this build.  So that cholmod_l_analyze is usable
 * without a UserPerm, this file provides a self-contained ordering of the same
 * family as cholmod_nested_dissection: recursive graph bisection with the
 * separator ordered last.  It is NOT a restatement of any reference routine
 * and makes no claim of matching the reference's permutation; the supernodal
 * maps are a function of (A, Perm), and parity is tested with the final Perm
 * handed to the oracle as CHOLMOD_GIVEN.
 *
 * Algorithm, for a vertex subset S of the graph of A+A':
 *   - split S into connected components; each is ordered on its own;
 *   - |S| <= leaf (96): exact minimum degree on the piece (bit-mask elimination
 *     graph); a larger piece that cannot be cut (a clique): reverse Cuthill-McKee;
 *   - else: pseudo-peripheral root by repeated BFS (George-Liu), level
 *     structure L_0..L_h; the separator is the smallest level among those that
 *     leave 30..70 % of the vertices below it (or the median level); separator
 *     vertices without a neighbour in the next level move to the lower part;
 *     recurse on the lower part, then the upper part, separator last.
 * O(nnz) per recursion depth. */

#include "host_internal.h"
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>

#define ND_LEAF 96

typedef struct
{
    Int n ;
    const Int *Gp, *Gi ;    /* symmetric adjacency, no diagonal */
    Int *mark ;             /* mark [v] == tag: v belongs to the subset with that tag; -1: ordered */
    Int *level ;            /* BFS level */
    Int *visit ;            /* BFS stamps (globally unique), scratch class store */
    Int *queue ;            /* BFS orders: a subset uses queue [start, start+count) */
    Int *list ;             /* a subset's vertices: list [start, start+count) */
    Int *tmp ;              /* scratch, same slicing */
    Int *Perm ;             /* a subset's slot: Perm [start, start+count) */
    Int next_tag, next_stamp ;
    int failed ;
} ND ;

static Int nd_new_stamp (ND *g) { return __atomic_add_fetch (&g->next_stamp, 1, __ATOMIC_RELAXED) ; }
static Int nd_new_tag (ND *g) { return __atomic_add_fetch (&g->next_tag, 1, __ATOMIC_RELAXED) ; }

/* Minimum-degree order of a small connected piece (count <= ND_LEAF <= 128
 * vertices) on its induced subgraph: the elimination graph is kept exactly, as
 * one 128-bit adjacency mask per vertex (a pivot's neighbours become a clique).
 * Ties go to the vertex met first.  verts [] is overwritten with the order. */
static void nd_leaf_min_degree (ND *g, Int *verts, Int count, Int tag, Int *out)
{
    unsigned long long lo [ND_LEAF], hi [ND_LEAF] ;
    for (Int k = 0 ; k < count ; k++) g->level [verts [k]] = k ;       /* local ids (we own these vertices) */
    for (Int k = 0 ; k < count ; k++)
    {
        Int v = verts [k] ;
        unsigned long long a = 0, b = 0 ;
        for (Int p = g->Gp [v] ; p < g->Gp [v+1] ; p++)
        {
            Int w = g->Gi [p] ;
            if (g->mark [w] != tag) continue ;
            Int id = g->level [w] ;
            if (id < 64) a |= 1ull << id ; else b |= 1ull << (id - 64) ;
        }
        lo [k] = a ; hi [k] = b ;
    }
    unsigned long long alive_lo = count >= 64 ? ~0ull : ((1ull << count) - 1) ;
    unsigned long long alive_hi = count > 64 ? (count >= 128 ? ~0ull : ((1ull << (count - 64)) - 1)) : 0 ;
    for (Int step = 0 ; step < count ; step++)
    {
        int best = -1, bd = 1 << 30 ;
        for (int k = 0 ; k < (int) count ; k++)
        {
            int al = k < 64 ? (int) ((alive_lo >> k) & 1) : (int) ((alive_hi >> (k - 64)) & 1) ;
            if (!al) continue ;
            int d = __builtin_popcountll (lo [k] & alive_lo) + __builtin_popcountll (hi [k] & alive_hi) ;
            if (d < bd) { bd = d ; best = k ; }
        }
        out [step] = verts [best] ;
        if (best < 64) alive_lo &= ~(1ull << best) ; else alive_hi &= ~(1ull << (best - 64)) ;
        unsigned long long nl = lo [best] & alive_lo, nh = hi [best] & alive_hi ;
        for (unsigned long long m = nl ; m ; m &= m - 1)
        {
            int u = __builtin_ctzll (m) ;
            lo [u] = (lo [u] | nl) & ~(1ull << u) ; hi [u] |= nh ;
        }
        for (unsigned long long m = nh ; m ; m &= m - 1)
        {
            int u = 64 + __builtin_ctzll (m) ;
            lo [u] |= nl ; hi [u] = (hi [u] | nh) & ~(1ull << (u - 64)) ;
        }
    }
}

static void nd_process (ND *g, Int start, Int count) ;

/* hand a subset on: an OpenMP task (big subsets) or a plain call */
static void nd_push (ND *g, Int start, Int count)
{
    if (count <= 0) return ;
    #pragma omp task firstprivate (g, start, count) if (count > 4096)
    nd_process (g, start, count) ;
}

/* BFS from root inside the vertices with mark == tag; queue [0..count) in BFS
 * order, *height = last level; returns the number of vertices reached */
static Int nd_bfs (ND *g, Int *queue, Int root, Int tag, Int stamp, Int *height)
{
    Int head = 0, tail = 0 ;
    Int *visit = g->visit, *level = g->level ;
    queue [tail++] = root ; visit [root] = stamp ; level [root] = 0 ;
    while (head < tail)
    {
        Int v = queue [head++] ;
        for (Int p = g->Gp [v] ; p < g->Gp [v+1] ; p++)
        {
            Int w = g->Gi [p] ;
            if (g->mark [w] != tag || visit [w] == stamp) continue ;
            visit [w] = stamp ; level [w] = level [v] + 1 ;
            queue [tail++] = w ;
        }
    }
    *height = level [queue [tail-1]] ;
    return tail ;
}

static Int nd_degree_in (const ND *g, Int v, Int tag)
{
    Int d = 0 ;
    for (Int p = g->Gp [v] ; p < g->Gp [v+1] ; p++) if (g->mark [g->Gi [p]] == tag) d++ ;
    return d ;
}

/* Order one subset: either completely (leaf, small components) or by cutting it
 * and pushing the parts.  Touches only its own vertices and its own slices of the
 * position-indexed arrays, so disjoint subsets can be processed concurrently. */
static void nd_process (ND *g, Int start, Int count)
{
    const Int *Gp = g->Gp, *Gi = g->Gi ;
    Int *S = g->list + start, *queue = g->queue + start, *tmp = g->tmp + start, *Perm = g->Perm + start ;
    Int *visit = g->visit ;
    Int tag = nd_new_tag (g) ;
    for (Int k = 0 ; k < count ; k++) g->mark [S [k]] = tag ;
    /* connected components, all in one pass (a diagonal matrix has n of them):
     * small ones are ordered at once (reverse BFS order), big ones are pushed */
    Int height = 0 ;
    Int nc = nd_bfs (g, queue, S [0], tag, nd_new_stamp (g), &height) ;
    if (nc < count)
    {
        Int a = 0 ;
        Int stamp = nd_new_stamp (g) ;
        for (Int k = 0 ; k < count ; k++)
        {
            Int v = S [k] ;
            if (visit [v] == stamp) continue ;
            Int h = 0 ;
            Int c = nd_bfs (g, tmp + a, v, tag, stamp, &h) ;
            if (c <= ND_LEAF)
            {
                nd_leaf_min_degree (g, tmp + a, c, tag, Perm + a) ;
                for (Int q = 0 ; q < c ; q++) g->mark [tmp [a + q]] = -1 ;
            }
            a += c ;
        }
        /* list = components in discovery order; the big ones become subsets of their own */
        for (Int k = 0 ; k < count ; k++) S [k] = tmp [k] ;
        for (Int k = 0 ; k < count ; )
        {
            Int v = S [k] ;
            if (g->mark [v] == -1) { k++ ; continue ; }
            /* a big component occupies a contiguous run of not-yet-ordered vertices that
             * ends where the next component (ordered or not) begins: recover its length
             * from the BFS levels -- a component starts at level 0 */
            Int e = k + 1 ;
            while (e < count && g->mark [S [e]] != -1 && g->level [S [e]] != 0) e++ ;
            nd_push (g, start + k, e - k) ;
            k = e ;
        }
        return ;
    }
    /* connected subset: pseudo-peripheral root (George-Liu): restart from a
     * minimum-degree vertex of the last level while the structure gets deeper */
    if (count > ND_LEAF)
        for (int sweep = 0 ; sweep < 4 ; sweep++)
        {
            Int cand = queue [nc-1], cd = nd_degree_in (g, cand, tag) ;
            for (Int k = nc - 1 ; k >= 0 && g->level [queue [k]] == height ; k--)
            {
                Int d = nd_degree_in (g, queue [k], tag) ;
                if (d < cd) { cd = d ; cand = queue [k] ; }
            }
            Int h2 = 0 ;
            nd_bfs (g, queue, cand, tag, nd_new_stamp (g), &h2) ;
            if (h2 <= height) { height = h2 ; break ; }
            height = h2 ;
        }
    if (count <= ND_LEAF || height < 2)
    {
        if (count <= ND_LEAF)
        {
            /* leaf: exact minimum degree on the piece */
            nd_leaf_min_degree (g, queue, count, tag, Perm) ;
            for (Int k = 0 ; k < count ; k++) g->mark [S [k]] = -1 ;
            return ;
        }
        /* too "round" to cut (e.g. a clique): reverse Cuthill-McKee from a
         * pseudo-peripheral vertex */
        Int root = queue [nc-1] ;
        nd_bfs (g, queue, root, tag, nd_new_stamp (g), &height) ;
        for (Int k = 0 ; k < count ; k++) Perm [k] = queue [count - 1 - k] ;
        for (Int k = 0 ; k < count ; k++) g->mark [S [k]] = -1 ;
        return ;
    }
    /* level sizes (queue is in level order) */
    Int *lsize = tmp ;                  /* height+1 <= count entries */
    for (Int l = 0 ; l <= height ; l++) lsize [l] = 0 ;
    for (Int k = 0 ; k < count ; k++) lsize [g->level [queue [k]]]++ ;
    Int best = -1, bsz = 0 ;
    {
        Int cum = 0 ;
        for (Int l = 1 ; l < height ; l++)
        {
            cum += lsize [l-1] ;
            double frac = (double) cum / (double) count ;
            if (frac >= 0.3 && frac <= 0.7 && (best < 0 || lsize [l] < bsz)) { best = l ; bsz = lsize [l] ; }
        }
        if (best < 0)
        {
            /* no level in the balanced window: the level holding the median vertex */
            cum = 0 ;
            for (Int l = 1 ; l < height ; l++)
            {
                cum += lsize [l-1] ;
                best = l ;
                if (2 * (cum + lsize [l]) >= count) break ;
            }
        }
    }
    /* classify: 0 lower (levels < best, and separator vertices with no neighbour in
     * level best+1), 1 upper, 2 separator */
    Int nlow = 0, nup = 0, nsep = 0 ;
    for (Int k = 0 ; k < count ; k++)
    {
        Int v = queue [k], l = g->level [v] ;
        Int cls = l < best ? 0 : (l > best ? 1 : 2) ;
        if (cls == 2)
        {
            int touches = 0 ;
            for (Int p = Gp [v] ; p < Gp [v+1] && !touches ; p++)
            {
                Int w = Gi [p] ;
                if (g->mark [w] == tag && g->level [w] == best + 1) touches = 1 ;
            }
            if (!touches) cls = 0 ;
        }
        visit [v] = -(cls + 1) ;        /* class store: -1, -2, -3 (never a stamp) */
        if (cls == 0) nlow++ ; else if (cls == 1) nup++ ; else nsep++ ;
    }
    /* list = [lower | upper | separator]; the separator is ordered last in the slot */
    {
        Int a = 0, b = nlow, c = nlow + nup ;
        for (Int k = 0 ; k < count ; k++)
        {
            Int v = queue [k] ;
            Int cls = -visit [v] - 1 ;
            if (cls == 0) tmp [a++] = v ; else if (cls == 1) tmp [b++] = v ; else tmp [c++] = v ;
        }
        for (Int k = 0 ; k < count ; k++) { S [k] = tmp [k] ; visit [S [k]] = 0 ; }
    }
    for (Int k = 0 ; k < nsep ; k++)
    {
        Perm [nlow + nup + k] = S [nlow + nup + k] ;
        g->mark [S [nlow + nup + k]] = -1 ;
    }
    nd_push (g, start + nlow, nup) ;
    nd_push (g, start, nlow) ;
}

int ssamd_nested_dissection (Int n, const Int *Ap, const Int *Ai, Int *Perm, cholmod_common *Common)
{
    if (n == 0) return TRUE ;
    double t_graph = omp_get_wtime () ;
    /* symmetric adjacency of the stored triangle */
    Int nz = Ap [n] ;
    Int *Gp = cholmod_l_calloc ((size_t) n + 2, sizeof (Int), Common) ;
    Int *Gi = cholmod_l_malloc ((size_t) (2 * nz + 1), sizeof (Int), Common) ;
    Int *iw = cholmod_l_malloc ((size_t) (8 * n + 8), sizeof (Int), Common) ;
    if (!Gp || !Gi || !iw)
    {
        if (Gp) cholmod_l_free ((size_t) n + 2, sizeof (Int), Gp, Common) ;
        if (Gi) cholmod_l_free ((size_t) (2 * nz + 1), sizeof (Int), Gi, Common) ;
        if (iw) cholmod_l_free ((size_t) (8 * n + 8), sizeof (Int), iw, Common) ;
        return FALSE ;
    }
    for (Int j = 0 ; j < n ; j++)
        for (Int p = Ap [j] ; p < Ap [j+1] ; p++)
        {
            Int i = Ai [p] ;
            if (i != j && i >= 0 && i < n) { Gp [i+1]++ ; Gp [j+1]++ ; }
        }
    for (Int j = 0 ; j < n ; j++) Gp [j+1] += Gp [j] ;
    {
        Int *pos = iw ;
        for (Int j = 0 ; j < n ; j++) pos [j] = Gp [j] ;
        for (Int j = 0 ; j < n ; j++)
            for (Int p = Ap [j] ; p < Ap [j+1] ; p++)
            {
                Int i = Ai [p] ;
                if (i != j && i >= 0 && i < n) { Gi [pos [i]++] = j ; Gi [pos [j]++] = i ; }
            }
    }
    ND g ;
    g.n = n ; g.Gp = Gp ; g.Gi = Gi ;
    g.mark = iw ; g.level = iw + n ; g.queue = iw + 2 * n ; g.visit = iw + 3 * n ;
    g.list = iw + 4 * n ; g.tmp = iw + 5 * n ; g.Perm = Perm ;
    g.next_tag = 0 ; g.next_stamp = 0 ; g.failed = 0 ;
    for (Int v = 0 ; v < n ; v++) { g.mark [v] = 0 ; g.visit [v] = 0 ; g.list [v] = v ; }
    /* Disjoint subsets are independent: every cut spawns its two parts as OpenMP
     * tasks (small ones run inline).  The result does not depend on the schedule:
     * a subset's order is a function of its own vertex list.  The team is kept
     * small; the top of the recursion is serial anyway. */
    double t_start = omp_get_wtime () ;
    int nth = omp_get_max_threads () ;
    if (nth > 16) nth = 16 ;
    if (n < 20000) nth = 1 ;
    #pragma omp parallel num_threads (nth)
    {
        #pragma omp single
        nd_process (&g, 0, n) ;
    }
    if (getenv ("CHOLMOD_ORDER_TIMING"))
        fprintf (stderr, "ssamd_nested_dissection: n %ld, %d threads, graph %.3f s, dissection %.3f s\n", (long) n, nth,
            t_start - t_graph, omp_get_wtime () - t_start) ;
    int ok = !g.failed ;
    cholmod_l_free ((size_t) n + 2, sizeof (Int), Gp, Common) ;
    cholmod_l_free ((size_t) (2 * nz + 1), sizeof (Int), Gi, Common) ;
    cholmod_l_free ((size_t) (8 * n + 8), sizeof (Int), iw, Common) ;
    return ok ;
}
