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
 *   - |S| <= leaf: reverse Cuthill-McKee order inside the leaf;
 *   - else: pseudo-peripheral root by repeated BFS (George-Liu), level
 *     structure L_0..L_h; the separator is the smallest level among those that
 *     leave 30..70 % of the vertices below it (or the median level); separator
 *     vertices without a neighbour in the next level move to the lower part;
 *     recurse on the lower part, then the upper part, separator last.
 * O(nnz) per recursion depth. */

#include "host_internal.h"

#define ND_LEAF 96

typedef struct
{
    Int n ;
    const Int *Gp, *Gi ;    /* symmetric adjacency, no diagonal */
    Int *mark ;             /* mark [v] == tag: v belongs to the current subset */
    Int *level ;            /* BFS level (valid for the current component) */
    Int *queue ;            /* BFS order */
    Int *out ;              /* permutation under construction */
    Int nout ;
} ND ;

/* BFS inside the vertices with mark == tag and visit != stamp; returns count,
 * queue [0..count) in BFS order, *height = last level */
static Int nd_bfs (ND *g, Int root, Int tag, Int *visit, Int stamp, Int *height)
{
    Int head = 0, tail = 0 ;
    g->queue [tail++] = root ; visit [root] = stamp ; g->level [root] = 0 ;
    while (head < tail)
    {
        Int v = g->queue [head++] ;
        for (Int p = g->Gp [v] ; p < g->Gp [v+1] ; p++)
        {
            Int w = g->Gi [p] ;
            if (g->mark [w] != tag || visit [w] == stamp) continue ;
            visit [w] = stamp ; g->level [w] = g->level [v] + 1 ;
            g->queue [tail++] = w ;
        }
    }
    *height = g->level [g->queue [tail-1]] ;
    return tail ;
}

static Int nd_degree_in (const ND *g, Int v, Int tag)
{
    Int d = 0 ;
    for (Int p = g->Gp [v] ; p < g->Gp [v+1] ; p++) if (g->mark [g->Gi [p]] == tag) d++ ;
    return d ;
}

int ssamd_nested_dissection (Int n, const Int *Ap, const Int *Ai, Int *Perm, cholmod_common *Common)
{
    if (n == 0) return TRUE ;
    /* symmetric adjacency of the stored triangle */
    Int nz = Ap [n] ;
    Int *Gp = cholmod_l_calloc ((size_t) n + 2, sizeof (Int), Common) ;
    Int *Gi = cholmod_l_malloc ((size_t) (2 * nz + 1), sizeof (Int), Common) ;
    Int *iw = cholmod_l_malloc ((size_t) (6 * n + 6), sizeof (Int), Common) ;
    if (!Gp || !Gi || !iw)
    {
        if (Gp) cholmod_l_free ((size_t) n + 2, sizeof (Int), Gp, Common) ;
        if (Gi) cholmod_l_free ((size_t) (2 * nz + 1), sizeof (Int), Gi, Common) ;
        if (iw) cholmod_l_free ((size_t) (6 * n + 6), sizeof (Int), iw, Common) ;
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
    g.mark = iw ; g.level = iw + n ; g.queue = iw + 2 * n ; g.out = Perm ; g.nout = 0 ;
    Int *visit = iw + 3 * n ;           /* BFS stamps */
    Int *list = iw + 4 * n ;            /* vertex lists of the pending subsets, stack-allocated */
    Int *tmp = iw + 5 * n ;
    /* work stack of subsets: (start, count, tag) over `list`; tags are unique */
    /* (every pop pushes at most two subsets and a leaf pushes none: never more than
     * n + 1 pending) */
    Int cap = n + 2, top = 0 ;
    Int *stk = cholmod_l_malloc ((size_t) (3 * cap), sizeof (Int), Common) ;
    int ok = stk != NULL ;
    Int next_tag = 1, stamp = 0 ;
    for (Int v = 0 ; v < n ; v++) { g.mark [v] = 0 ; visit [v] = 0 ; list [v] = v ; }
    /* Every subset owns the slot [start, start+count) of both `list` and Perm: its
     * separator is written to the END of the slot at once, the two parts keep the
     * front of it and are pushed for later. */
    if (ok)
    {
        stk [0] = 0 ; stk [1] = n ; stk [2] = 0 ;      /* start in list, count, tag 0 (= everything) */
        top = 1 ;
    }
    /* slot base of a subset == its start in `list` (lists are permuted in place so
     * that a subset's vertices always occupy list [start, start+count) and Perm
     * [start, start+count) is its slot) */
    while (ok && top > 0)
    {
        top-- ;
        Int start = stk [3*top], count = stk [3*top+1], tag = stk [3*top+2] ;
        Int *S = list + start ;
        if (count <= 0) continue ;
        /* (re)mark the subset with its tag */
        for (Int k = 0 ; k < count ; k++) g.mark [S [k]] = tag ;
        /* connected components, all in one pass (a diagonal matrix has n of them):
         * small ones are ordered at once (reverse BFS order), big ones are pushed */
        stamp++ ;
        Int height = 0 ;
        Int nc = nd_bfs (&g, S [0], tag, visit, stamp, &height) ;
        if (nc < count)
        {
            Int *qsave = g.queue ;
            Int a = 0 ;
            stamp++ ;
            for (Int k = 0 ; k < count ; k++)
            {
                Int v = S [k] ;
                if (visit [v] == stamp) continue ;
                g.queue = tmp + a ;
                Int h = 0 ;
                Int c = nd_bfs (&g, v, tag, visit, stamp, &h) ;
                if (c <= ND_LEAF)
                {
                    for (Int q = 0 ; q < c ; q++) { Perm [start + a + q] = tmp [a + c - 1 - q] ; g.mark [tmp [a + q]] = -1 ; }
                }
                else
                {
                    if (top + 1 > cap) { ok = FALSE ; break ; }
                    stk [3*top] = start + a ; stk [3*top+1] = c ; stk [3*top+2] = -1 ; top++ ;   /* tag assigned below */
                }
                a += c ;
            }
            g.queue = qsave ;
            if (!ok) break ;
            for (Int k = 0 ; k < count ; k++) S [k] = tmp [k] ;
            /* fresh tags for the pushed components (top-down over the entries just pushed) */
            for (Int q = top - 1 ; q >= 0 && stk [3*q+2] == -1 ; q--) stk [3*q+2] = next_tag++ ;
            continue ;
        }
        /* connected subset: pseudo-peripheral root (George-Liu): restart from a
         * minimum-degree vertex of the last level while the structure gets deeper */
        if (count > ND_LEAF)
            for (int sweep = 0 ; sweep < 4 ; sweep++)
            {
                Int cand = g.queue [nc-1], cd = nd_degree_in (&g, cand, tag) ;
                for (Int k = nc - 1 ; k >= 0 && g.level [g.queue [k]] == height ; k--)
                {
                    Int d = nd_degree_in (&g, g.queue [k], tag) ;
                    if (d < cd) { cd = d ; cand = g.queue [k] ; }
                }
                Int h2 = 0 ;
                stamp++ ;
                nd_bfs (&g, cand, tag, visit, stamp, &h2) ;
                if (h2 <= height) { height = h2 ; break ; }
                height = h2 ;
            }
        if (count <= ND_LEAF || height < 2)
        {
            /* leaf (or too "round" to cut, e.g. a clique): reverse Cuthill-McKee from a
             * pseudo-peripheral vertex of the leaf */
            Int root = g.queue [nc-1] ;
            stamp++ ;
            nd_bfs (&g, root, tag, visit, stamp, &height) ;
            for (Int k = 0 ; k < count ; k++) Perm [start + k] = g.queue [count - 1 - k] ;
            for (Int k = 0 ; k < count ; k++) g.mark [S [k]] = -1 ;    /* done */
            continue ;
        }
        /* level sizes (queue is in level order) */
        Int *lsize = tmp ;                  /* height+1 entries */
        for (Int l = 0 ; l <= height ; l++) lsize [l] = 0 ;
        for (Int k = 0 ; k < count ; k++) lsize [g.level [g.queue [k]]]++ ;
        Int best = -1, below = 0, bsz = 0 ;
        {
            Int cum = 0 ;
            for (Int l = 1 ; l < height ; l++)
            {
                cum += lsize [l-1] ;
                double frac = (double) cum / (double) count ;
                if (frac >= 0.3 && frac <= 0.7 && (best < 0 || lsize [l] < bsz)) { best = l ; bsz = lsize [l] ; below = cum ; }
            }
            if (best < 0)
            {
                /* no level in the balanced window: the level holding the median vertex */
                cum = 0 ;
                for (Int l = 1 ; l < height ; l++)
                {
                    cum += lsize [l-1] ;
                    best = l ; below = cum ;
                    if (2 * (cum + lsize [l]) >= count) break ;
                }
            }
        }
        (void) below ;
        /* classify: 0 lower (levels < best, and separator vertices with no neighbour
         * in level best+1), 1 upper, 2 separator */
        Int nlow = 0, nup = 0, nsep = 0 ;
        for (Int k = 0 ; k < count ; k++)
        {
            Int v = g.queue [k], l = g.level [v] ;
            Int cls = l < best ? 0 : (l > best ? 1 : 2) ;
            if (cls == 2)
            {
                int touches = 0 ;
                for (Int p = Gp [v] ; p < Gp [v+1] && !touches ; p++)
                {
                    Int w = Gi [p] ;
                    if (g.mark [w] == tag && g.level [w] == best + 1) touches = 1 ;
                }
                if (!touches) cls = 0 ;
            }
            visit [v] = -(cls + 1) ;        /* reuse visit as class store: -1, -2, -3 */
            if (cls == 0) nlow++ ; else if (cls == 1) nup++ ; else nsep++ ;
        }
        /* list = [lower | upper | separator]; the separator goes to the end of the slot */
        {
            Int a = 0, b = nlow, c = nlow + nup ;
            for (Int k = 0 ; k < count ; k++)
            {
                Int v = g.queue [k] ;
                Int cls = -visit [v] - 1 ;
                if (cls == 0) tmp [a++] = v ; else if (cls == 1) tmp [b++] = v ; else tmp [c++] = v ;
            }
            /* tmp also held lsize: it is no longer needed */
            for (Int k = 0 ; k < count ; k++) { S [k] = tmp [k] ; visit [S [k]] = 0 ; }
        }
        for (Int k = 0 ; k < nsep ; k++)
        {
            Perm [start + nlow + nup + k] = S [nlow + nup + k] ;
            g.mark [S [nlow + nup + k]] = -1 ;
        }
        if (top + 2 > cap) { ok = FALSE ; break ; }
        stk [3*top] = start + nlow ; stk [3*top+1] = nup ; stk [3*top+2] = next_tag++ ; top++ ;
        stk [3*top] = start ; stk [3*top+1] = nlow ; stk [3*top+2] = next_tag++ ; top++ ;
    }
    if (stk) cholmod_l_free ((size_t) (3 * cap), sizeof (Int), stk, Common) ;
    cholmod_l_free ((size_t) n + 2, sizeof (Int), Gp, Common) ;
    cholmod_l_free ((size_t) (2 * nz + 1), sizeof (Int), Gi, Common) ;
    cholmod_l_free ((size_t) (6 * n + 6), sizeof (Int), iw, Common) ;
    return ok ;
}
