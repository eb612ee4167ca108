/* analyze.c -- symbolic analysis of the host layer: elimination tree, weighted
 * postorder, column counts, cholmod_l_analyze[_p|_p2] and the supernodal
 * symbolic factorization.  Integer-only host code; the index maps it produces
 * (super / pi / px / s / maxcsize / maxesize) are contractual: bit-exact with
 * the reference's useGPU==0 partition (SURVEY.md finding 3 -- no device-buffer
 * driven supernode splits are applied here, unlike the reference's GPU branch
 * at CHOLMOD/Supernodal/cholmod_super_symbolic.c:422-429, :582-592).
 *
 * Reference files: CHOLMOD/Cholesky/cholmod_etree.c, cholmod_postorder.c,
 * cholmod_rowcolcounts.c, cholmod_analyze.c; CHOLMOD/Supernodal/
 * cholmod_super_symbolic.c. */
#include "host_internal.h"
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- elimination tree ------------------------------------------------------------ */

/* Liu's algorithm with path compression over an upper-stored pattern
 * (reference Cholesky/cholmod_etree.c:81-223, stype > 0 branch). */
int ssamd_etree_upper (Int n, const Int *Up, const Int *Ui, Int *Parent)
{
    Int *anc = SuiteSparse_malloc ((size_t) (n > 0 ? n : 1), sizeof (Int)) ;
    if (!anc) return FALSE ;
    for (Int j = 0 ; j < n ; j++) { Parent [j] = EMPTY ; anc [j] = EMPTY ; }
    for (Int j = 0 ; j < n ; j++)
    {
        for (Int p = Up [j] ; p < Up [j+1] ; p++)
        {
            Int i = Ui [p] ;
            /* climb from i towards the root, hanging every visited root under j */
            while (i < j)
            {
                Int up = anc [i] ;
                anc [i] = j ;
                if (up == EMPTY) { Parent [i] = j ; break ; }
                if (up == j) break ;
                i = up ;
            }
        }
    }
    SuiteSparse_free (anc) ;
    return TRUE ;
}

int cholmod_l_etree (cholmod_sparse *A, SuiteSparse_long *Parent, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (A, FALSE) ;
    RETURN_IF_NULL (Parent, FALSE) ;
    Common->status = CHOLMOD_OK ;
    if (A->stype < 0) { ERROR (CHOLMOD_INVALID, "symmetric lower not supported") ; return FALSE ; }
    if (A->stype == 0) { ERROR (CHOLMOD_NOT_INSTALLED, "etree(A'A) not built") ; return FALSE ; }
    if (!A->packed) { ERROR (CHOLMOD_NOT_INSTALLED, "unpacked etree input not built") ; return FALSE ; }
    if (!ssamd_etree_upper ((Int) A->ncol, A->p, A->i, Parent))
    { ERROR (CHOLMOD_OUT_OF_MEMORY, "out of memory") ; return FALSE ; }
    return TRUE ;
}

/* ---- postorder --------------------------------------------------------------------- */

/* Children are visited in increasing weight, ties in increasing node number
 * (no weights: increasing node number); roots in increasing node number.
 * This is the visiting order the reference obtains from its bucket lists
 * (Cholesky/cholmod_postorder.c:185-260) and non-recursive dfs (:60-94).
 * work3n: 3n Ints. */
Int ssamd_postorder (Int n, const Int *Parent, const Int *Weight, Int *Post, Int *work)
{
    Int *first_child = work, *sibling = work + n, *stack = work + 2*n ;
    for (Int j = 0 ; j < n ; j++) first_child [j] = EMPTY ;
    if (!Weight)
    {
        for (Int j = n - 1 ; j >= 0 ; j--)
        {
            Int p = Parent [j] ;
            if (p >= 0 && p < n) { sibling [j] = first_child [p] ; first_child [p] = j ; }
        }
    }
    else
    {
        /* counting sort of the nodes by clamped weight (stable in node number),
         * then push them on their parents' lists from heaviest to lightest so
         * that every list ends up lightest-first */
        Int *bucket = stack ;                   /* n counters, reused as stack later */
        Int *order = SuiteSparse_malloc ((size_t) (n > 0 ? n : 1), sizeof (Int)) ;
        if (!order) return EMPTY ;
        for (Int w = 0 ; w < n ; w++) bucket [w] = 0 ;
        for (Int j = 0 ; j < n ; j++)
        {
            Int w = Weight [j] ; if (w < 0) w = 0 ; if (w > n - 1) w = n - 1 ;
            bucket [w]++ ;
        }
        Int run = 0 ;
        for (Int w = 0 ; w < n ; w++) { Int c = bucket [w] ; bucket [w] = run ; run += c ; }
        for (Int j = 0 ; j < n ; j++)
        {
            Int w = Weight [j] ; if (w < 0) w = 0 ; if (w > n - 1) w = n - 1 ;
            order [bucket [w]++] = j ;
        }
        for (Int q = n - 1 ; q >= 0 ; q--)
        {
            Int j = order [q] ;
            Int p = Parent [j] ;
            if (p >= 0 && p < n) { sibling [j] = first_child [p] ; first_child [p] = j ; }
        }
        SuiteSparse_free (order) ;
    }
    Int k = 0 ;
    for (Int r = 0 ; r < n ; r++)
    {
        if (Parent [r] != EMPTY) continue ;
        Int top = 0 ;
        stack [0] = r ;
        while (top >= 0)
        {
            Int node = stack [top] ;
            Int c = first_child [node] ;
            if (c == EMPTY) { Post [k++] = node ; top-- ; }
            else { first_child [node] = sibling [c] ; stack [++top] = c ; }
        }
    }
    return k ;
}

SuiteSparse_long cholmod_l_postorder (SuiteSparse_long *Parent, size_t n,
    SuiteSparse_long *Weight, SuiteSparse_long *Post, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (EMPTY) ;
    RETURN_IF_NULL (Parent, EMPTY) ;
    RETURN_IF_NULL (Post, EMPTY) ;
    Common->status = CHOLMOD_OK ;
    Int *work = cholmod_l_malloc (3 * n + 1, sizeof (Int), Common) ;
    if (!work) return EMPTY ;
    Int k = ssamd_postorder ((Int) n, Parent, Weight, Post, work) ;
    cholmod_l_free (3 * n + 1, sizeof (Int), work, Common) ;
    if (k == EMPTY) ERROR (CHOLMOD_OUT_OF_MEMORY, "out of memory") ;
    return k ;
}

/* ---- column counts ------------------------------------------------------------------ */

/* Gilbert-Ng-Peyton skeleton algorithm on a lower-stored pattern (column j
 * holds the rows i >= j of the symmetric matrix).  Produces the counts the
 * reference computes at Cholesky/cholmod_rowcolcounts.c:184-533 (diagonal
 * included).  work5n: 5n Ints. */
void ssamd_colcounts (Int n, const Int *Lp, const Int *Li, const Int *Parent, const Int *Post,
    Int *ColCount, Int *work)
{
    Int *first = work, *maxfirst = work + n, *prevleaf = work + 2*n, *setroot = work + 3*n ;
    Int *delta = ColCount ;
    for (Int j = 0 ; j < n ; j++) { first [j] = EMPTY ; maxfirst [j] = EMPTY ; prevleaf [j] = EMPTY ; setroot [j] = j ; }
    for (Int k = 0 ; k < n ; k++)
    {
        Int j = Post [k] ;
        delta [j] = (first [j] == EMPTY) ? 1 : 0 ;         /* 1 for a leaf of the etree */
        for ( ; j != EMPTY && first [j] == EMPTY ; j = Parent [j]) first [j] = k ;
    }
    for (Int k = 0 ; k < n ; k++)
    {
        Int j = Post [k] ;
        if (Parent [j] != EMPTY) delta [Parent [j]]-- ;
        for (Int p = Lp [j] ; p < Lp [j+1] ; p++)
        {
            Int i = Li [p] ;
            if (i <= j || first [j] <= maxfirst [i]) continue ;  /* j not a leaf of row subtree i */
            maxfirst [i] = first [j] ;
            Int jprev = prevleaf [i] ;
            prevleaf [i] = j ;
            delta [j]++ ;
            if (jprev != EMPTY)
            {
                /* least common ancestor of the previous leaf and j */
                Int q = jprev ;
                while (q != setroot [q]) q = setroot [q] ;
                for (Int s = jprev ; s != q ; ) { Int nx = setroot [s] ; setroot [s] = q ; s = nx ; }
                delta [q]-- ;
            }
        }
        if (Parent [j] != EMPTY) setroot [j] = Parent [j] ;
    }
    for (Int j = 0 ; j < n ; j++)
        if (Parent [j] != EMPTY) ColCount [Parent [j]] += ColCount [j] ;
}

int cholmod_l_rowcolcounts (cholmod_sparse *A, SuiteSparse_long *fset, size_t fsize,
    SuiteSparse_long *Parent, SuiteSparse_long *Post, SuiteSparse_long *RowCount,
    SuiteSparse_long *ColCount, SuiteSparse_long *First, SuiteSparse_long *Level,
    cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (A, FALSE) ;
    RETURN_IF_NULL (Parent, FALSE) ;
    RETURN_IF_NULL (Post, FALSE) ;
    RETURN_IF_NULL (ColCount, FALSE) ;
    (void) fset ; (void) fsize ;
    Common->status = CHOLMOD_OK ;
    if (A->stype > 0) { ERROR (CHOLMOD_INVALID, "symmetric upper not supported") ; return FALSE ; }
    if (A->stype == 0 || !A->packed) { ERROR (CHOLMOD_NOT_INSTALLED, "A*A' counts not built") ; return FALSE ; }
    Int n = (Int) A->nrow ;
    Int *work = cholmod_l_malloc (5 * (size_t) n + 1, sizeof (Int), Common) ;
    if (!work) return FALSE ;
    ssamd_colcounts (n, A->p, A->i, Parent, Post, ColCount, work) ;
    if (First) for (Int j = 0 ; j < n ; j++) First [j] = work [j] ;
    if (Level)
    {
        for (Int k = n - 1 ; k >= 0 ; k--)
        {
            Int j = Post [k] ;
            Level [j] = (Parent [j] == EMPTY) ? 0 : Level [Parent [j]] + 1 ;
        }
    }
    if (RowCount)
    {
        /* nnz in row i of L = size of the row subtree; recount with marking
         * (only used for statistics, O(nnz(L))) */
        Int *mark = work ;
        for (Int i = 0 ; i < n ; i++) { RowCount [i] = 1 ; mark [i] = EMPTY ; }
        cholmod_sparse *U = cholmod_l_ptranspose (A, 0, NULL, NULL, 0, Common) ;
        if (U)
        {
            Int *Up = U->p, *Ui = U->i ;
            for (Int i = 0 ; i < n ; i++)
            {
                mark [i] = i ;
                for (Int p = Up [i] ; p < Up [i+1] ; p++)
                    for (Int k = Ui [p] ; k < i && mark [k] != i ; k = Parent [k]) { RowCount [i]++ ; mark [k] = i ; }
            }
            cholmod_l_free_sparse (&U, Common) ;
        }
    }
    double fl = 0, lnz = 0 ;
    for (Int j = 0 ; j < n ; j++) { double c = (double) ColCount [j] ; fl += c * c ; lnz += c ; }
    Common->fl = fl ; Common->lnz = lnz ;       /* rowcolcounts.c:517-528 */
    cholmod_l_free (5 * (size_t) n + 1, sizeof (Int), work, Common) ;
    return TRUE ;
}

/* ---- supernodal symbolic ---------------------------------------------------------------- */

/* resolve Common->useGPU == EMPTY from CHOLMOD_USE_GPU exactly as the reference
 * does (Supernodal/cholmod_super_symbolic.c:257-296): the variable set to a
 * non-zero number selects the GPU, unset (or 0) the CPU. */
int ssamd_resolve_use_gpu (cholmod_common *Common)
{
    if (Common->useGPU == EMPTY)
    {
        const char *e = getenv ("CHOLMOD_USE_GPU") ;
        Common->useGPU = (e && atoi (e) != 0) ? 1 : 0 ;
        const char *b = getenv ("CHOLMOD_GPU_MEM_BYTES") ;
        if (b) Common->maxGpuMemBytes = (size_t) strtoull (b, NULL, 10) ;
        const char *f = getenv ("CHOLMOD_GPU_MEM_FRACTION") ;
        if (f) Common->maxGpuMemFraction = atof (f) ;
    }
    return Common->useGPU ;
}

typedef struct
{
    Int first ;     /* leading column */
    Int ncols ;
    Int lead_nz ;   /* entries in the leading column (= rows of the supernode) */
    Int zeros ;     /* explicit zeros accumulated by amalgamation */
    Int into ;      /* supernode this one has been merged into, or EMPTY */
    Int tparent ;   /* parent in the fundamental supernodal etree */
} fsnode ;

static Int live_ancestor (fsnode *F, Int s)
{
    Int r = F [s].tparent ;
    while (F [r].into != EMPTY) r = F [r].into ;
    /* path compression: everything on the way now points at the live node */
    for (Int q = F [s].tparent ; F [q].into != EMPTY ; )
    {
        Int nx = F [q].into ;
        F [q].into = r ;
        q = nx ;
    }
    return r ;
}

int cholmod_l_super_symbolic2 (int for_whom, cholmod_sparse *A, cholmod_sparse *Fm,
    SuiteSparse_long *Parent, cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (A, FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    RETURN_IF_NULL (Parent, FALSE) ;
    if (A->stype < 0) { ERROR (CHOLMOD_INVALID, "symmetric lower not supported") ; return FALSE ; }
    if (A->stype == 0)
    {
        /* A*F, F = A(:,f)' (cholmod_super_symbolic.c:167-181 requires F): the upper pattern of the product, formed, takes
         * the symmetric route */
        if (!Fm) { ERROR (CHOLMOD_INVALID, "F is required for the unsymmetric case") ; return FALSE ; }
        cholmod_sparse *C = ssamd_aat (A, Fm, 0, FALSE, Common) ;
        if (!C) return FALSE ;
        int okc = cholmod_l_super_symbolic2 (for_whom, C, NULL, Parent, L, Common) ;
        cholmod_l_free_sparse (&C, Common) ;
        return okc ;
    }
    if (L->is_super || L->xtype != CHOLMOD_PATTERN)
    { ERROR (CHOLMOD_INVALID, "L must be symbolic on input") ; return FALSE ; }
    if (!A->packed) { ERROR (CHOLMOD_NOT_INSTALLED, "unpacked input not built") ; return FALSE ; }
    Common->status = CHOLMOD_OK ;
    Int n = (Int) A->nrow ;
    const Int *Up = A->p, *Ui = A->i ;
    const Int *ColCount = L->ColCount ;

    /* GPU selection only (no memory pools are cut here; the engine reserves HBM
     * when the plan is created at the first numeric factorization) */
    if (for_whom == CHOLMOD_ANALYZE_FOR_CHOLESKY && ssamd_resolve_use_gpu (Common) == 1)
        L->useGPU = cholmod_l_gpu_probe (Common) ;
    else
        L->useGPU = 0 ;

    double zr [3] ;
    for (int t = 0 ; t < 3 ; t++) zr [t] = isnan (Common->zrelax [t]) ? 0 : Common->zrelax [t] ;
    Int nr0 = (Int) Common->nrelax [0], nr1 = (Int) Common->nrelax [1], nr2 = (Int) Common->nrelax [2] ;

    fsnode *F = cholmod_l_malloc (n + 1, sizeof (fsnode), Common) ;
    Int *col2s = cholmod_l_malloc (n + 1, sizeof (Int), Common) ;
    Int *kids = cholmod_l_calloc (n + 1, sizeof (Int), Common) ;
    if (!F || !col2s || !kids)
    {
        if (F) cholmod_l_free (n + 1, sizeof (fsnode), F, Common) ;
        if (col2s) cholmod_l_free (n + 1, sizeof (Int), col2s, Common) ;
        if (kids) cholmod_l_free (n + 1, sizeof (Int), kids, Common) ;
        return FALSE ;
    }
    /* fundamental supernodes: column j continues the supernode of j-1 iff j is
     * the only child of... precisely the three tests of :416-435 */
    for (Int j = 0 ; j < n ; j++) if (Parent [j] != EMPTY) kids [Parent [j]]++ ;
    Int nf = 0 ;
    for (Int j = 0 ; j < n ; j++)
    {
        int chain = j > 0 && Parent [j-1] == j && ColCount [j-1] == ColCount [j] + 1 && kids [j] <= 1 ;
        if (!chain)
        {
            F [nf].first = j ; F [nf].ncols = 0 ; F [nf].lead_nz = ColCount [j] ;
            F [nf].zeros = 0 ; F [nf].into = EMPTY ;
            nf++ ;
        }
        F [nf-1].ncols++ ;
        col2s [j] = nf - 1 ;
    }
    for (Int s = 0 ; s < nf ; s++)
    {
        Int last = F [s].first + F [s].ncols - 1 ;
        F [s].tparent = (Parent [last] == EMPTY) ? EMPTY : col2s [Parent [last]] ;
    }
    /* relaxed amalgamation, right to left (:478-602) */
    for (Int s = nf - 2 ; s >= 0 ; s--)
    {
        if (F [s].tparent == EMPTY) continue ;
        if (live_ancestor (F, s) != s + 1) continue ;
        fsnode *a = &F [s], *b = &F [s+1] ;
        Int ns = a->ncols + b->ncols ;
        Int zeros = b->zeros ;
        int merge ;
        if (ns <= nr0)
        {
            merge = TRUE ;      /* tiny: merged without counting its new zeros (:530-534) */
        }
        else
        {
            double lnz0 = (double) a->lead_nz, lnz1 = (double) b->lead_nz ;
            double xnew = a->ncols * (lnz1 + a->ncols - lnz0) ;
            if (xnew == 0)
            {
                merge = TRUE ;
            }
            else
            {
                double xns = (double) ns ;
                double xsize = (xns * (xns + 1) / 2) + xns * (lnz1 - b->ncols) ;
                double z = (((double) zeros) + xnew) / xsize ;
                zeros += a->ncols * (b->lead_nz + a->ncols - a->lead_nz) ;
                merge = ((ns <= nr1 && z < zr [0]) || (ns <= nr2 && z < zr [1]) || (z < zr [2]))
                    && (xsize < (double) INT64_MAX / sizeof (double)) ;
            }
        }
        if (merge)
        {
            a->zeros = zeros ;
            b->into = s ;
            a->lead_nz = a->ncols + b->lead_nz ;
            a->ncols = ns ;
        }
    }
    /* relaxed supernodes and their sizes (:612-699) */
    Int nsuper = 0 ;
    for (Int s = 0 ; s < nf ; s++) if (F [s].into == EMPTY) nsuper++ ;
    Int *Super = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *Lpi = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *Lpx = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *Sparent = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *fill = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *seen = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *Ls = NULL ;
    Int ssize = 0, xsize = 0 ;
    /* for_whom (:155-160, :662-663, :749-771, :912-913): the numeric sizes (xsize,
     * px, maxcsize, maxesize) exist for Cholesky and for GPU-accelerated SPQR;
     * plain SPQR gets the row structure only and px [0] = 123456 as the marker */
    const int want_px = (for_whom == CHOLMOD_ANALYZE_FOR_CHOLESKY || for_whom == CHOLMOD_ANALYZE_FOR_SPQRGPU) ;
    int ok = Super && Lpi && Lpx && Sparent && fill && seen ;
    if (ok)
    {
        double xx = 0 ;
        Int q = 0 ;
        for (Int s = 0 ; s < nf ; s++)
        {
            if (F [s].into != EMPTY) continue ;
            Super [q] = F [s].first ;
            Lpi [q] = ssize ; Lpx [q] = xsize ;
            ssize += F [s].lead_nz ;
            if (want_px)
            {
                xsize += F [s].ncols * F [s].lead_nz ;
                xx += (double) F [s].ncols * (double) F [s].lead_nz ;
            }
            if (ssize < 0 || xx > (double) INT64_MAX)
            {
                ERROR (CHOLMOD_TOO_LARGE, "problem too large") ;
                ok = FALSE ;
                break ;
            }
            q++ ;
        }
        if (ok)
        {
            Super [nsuper] = n ; Lpi [nsuper] = ssize ; Lpx [nsuper] = xsize ;
            if (!want_px) Lpx [0] = 123456 ;
            for (Int s = 0 ; s < nsuper ; s++)
                for (Int k = Super [s] ; k < Super [s+1] ; k++) col2s [k] = s ;
            for (Int s = 0 ; s < nsuper ; s++)
            {
                Int par = Parent [Super [s+1] - 1] ;
                Sparent [s] = (par == EMPTY) ? EMPTY : col2s [par] ;
            }
            Ls = cholmod_l_malloc (ssize > 1 ? ssize : 1, sizeof (Int), Common) ;
            ok = Ls != NULL ;
        }
    }
    if (ok)
    {
        /* row structure: own columns first, then for every column k the
         * supernodes on the etree paths from the supernodes of A(0:k1-1,k) up
         * to s receive row k (:786-835).  Lists come out strictly ascending. */
        Ls [0] = 0 ;
        for (Int s = 0 ; s < nsuper ; s++) { fill [s] = Lpi [s] ; seen [s] = EMPTY ; }
        /* One supernode's share of the traversal (:786-835): its own columns first,
         * then column k is appended to every not yet stamped supernode on the etree
         * paths from the supernodes of A(0:k1-1,k) up towards s.  Everything it
         * touches lies in the subtree of s. */
#define SSAMD_LS_OF(s) do { \
            Int k1_ = Super [s], k2_ = Super [(s)+1] ; \
            for (Int k = k1_ ; k < k2_ ; k++) Ls [fill [s]++] = k ; \
            for (Int k = k1_ ; k < k2_ ; k++) \
            { \
                seen [s] = k ;          /* stamps are column numbers: unique per k */ \
                for (Int p = Up [k] ; p < Up [k+1] ; p++) \
                { \
                    Int i = Ui [p] ; \
                    if (i >= k1_) { if (A->sorted) break ; else continue ; } \
                    for (Int t = col2s [i] ; seen [t] != k ; t = Sparent [t]) \
                    { \
                        Ls [fill [t]++] = k ; \
                        seen [t] = k ; \
                    } \
                } \
            } } while (0)
        /* Disjoint subtrees of the supernodal etree never touch each other's lists:
         * cut the tree where a subtree holds less than 1/(8 threads) of all rows,
         * run the subtrees in parallel (members in ascending order inside each),
         * then the few supernodes above the cut one after the other.  A list still
         * receives its rows in ascending order: the columns of an ancestor are
         * larger than those of any descendant. */
        Int *wsub = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
        Int *owner = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
        Int *mlist = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
        Int *mptr = cholmod_l_malloc (nsuper + 2, sizeof (Int), Common) ;
        if (wsub && owner && mlist && mptr)
        {
            int nth = 1 ;
#ifdef _OPENMP
            nth = ssamd_host_threads () ;
#endif
            for (Int s = 0 ; s < nsuper ; s++) wsub [s] = Lpi [s+1] - Lpi [s] ;
            for (Int s = 0 ; s < nsuper ; s++) if (Sparent [s] != EMPTY) wsub [Sparent [s]] += wsub [s] ;
            Int thr = (nth > 1) ? ssize / (8 * (Int) nth) : ssize + 1 ;
            /* owner: EMPTY above the cut, else the root of the subtree (parent > child,
             * so a descending sweep sees parents first) */
            Int nroots = 0 ;
            for (Int s = nsuper - 1 ; s >= 0 ; s--)
            {
                Int par = Sparent [s] ;
                if (wsub [s] > thr) owner [s] = EMPTY ;
                else if (par == EMPTY || owner [par] == EMPTY) { owner [s] = s ; nroots++ ; }
                else owner [s] = owner [par] ;
            }
            /* members of every subtree in ascending order (counting sort by root) */
            Int *rootid = wsub ;                    /* reuse: root supernode -> 0 .. nroots-1 */
            Int nr = 0 ;
            for (Int s = 0 ; s < nsuper ; s++) if (owner [s] == s) rootid [s] = nr++ ;
            for (Int r = 0 ; r <= nroots ; r++) mptr [r] = 0 ;
            for (Int s = 0 ; s < nsuper ; s++) if (owner [s] != EMPTY) mptr [rootid [owner [s]] + 1]++ ;
            for (Int r = 0 ; r < nroots ; r++) mptr [r+1] += mptr [r] ;
            {
                Int *pos = cholmod_l_malloc (nroots + 1, sizeof (Int), Common) ;
                if (pos)
                {
                    for (Int r = 0 ; r < nroots ; r++) pos [r] = mptr [r] ;
                    for (Int s = 0 ; s < nsuper ; s++) if (owner [s] != EMPTY) mlist [pos [rootid [owner [s]]]++] = s ;
                    cholmod_l_free (nroots + 1, sizeof (Int), pos, Common) ;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth)
                    for (Int r = 0 ; r < nroots ; r++)
                        for (Int q = mptr [r] ; q < mptr [r+1] ; q++) { Int s = mlist [q] ; SSAMD_LS_OF (s) ; }
                    for (Int s = 0 ; s < nsuper ; s++) if (owner [s] == EMPTY) SSAMD_LS_OF (s) ;
                }
                else ok = FALSE ;
            }
        }
        else ok = FALSE ;
        if (wsub) cholmod_l_free (nsuper + 1, sizeof (Int), wsub, Common) ;
        if (owner) cholmod_l_free (nsuper + 1, sizeof (Int), owner, Common) ;
        if (mlist) cholmod_l_free (nsuper + 1, sizeof (Int), mlist, Common) ;
        if (mptr) cholmod_l_free (nsuper + 2, sizeof (Int), mptr, Common) ;
#undef SSAMD_LS_OF
        for (Int s = 0 ; s < nsuper && ok ; s++) if (fill [s] != Lpi [s+1]) ok = FALSE ;
        /* (a failed allocation above has already said CHOLMOD_OUT_OF_MEMORY: that is not a structure error) */
        if (!ok && Common->status == CHOLMOD_OK) ERROR (CHOLMOD_INVALID, "invalid symbolic structure (ColCount/Parent mismatch)") ;
    }
    Int maxcsize = 1, maxesize = 1 ;
    if (ok && want_px)
    {
        /* largest update matrix / largest set of rows below a diagonal block
         * (:907-948): runs of rows belonging to one ancestor supernode */
#pragma omp parallel for schedule(dynamic, 1024) reduction(max:maxcsize) reduction(max:maxesize) num_threads(ssamd_host_threads ())
        for (Int d = 0 ; d < nsuper ; d++)
        {
            Int nscol = Super [d+1] - Super [d] ;
            Int p = Lpi [d] + nscol, pend = Lpi [d+1] ;
            if (pend - p > maxesize) maxesize = pend - p ;
            while (p < pend)
            {
                Int t = col2s [Ls [p]] ;
                Int q = p ;
                while (q < pend && col2s [Ls [q]] == t) q++ ;
                Int csize = (pend - p) * (q - p) ;
                if (csize > maxcsize) maxcsize = csize ;
                p = q ;
            }
        }
    }
    if (ok)
    {
        L->nsuper = nsuper ;
        L->ssize = ssize > 1 ? ssize : 1 ;
        L->xsize = xsize > 1 ? xsize : 1 ;
        L->maxcsize = maxcsize ; L->maxesize = maxesize ;
        L->super = Super ; L->pi = Lpi ; L->px = Lpx ; L->s = Ls ;
        L->is_super = TRUE ; L->is_ll = TRUE ; L->xtype = CHOLMOD_PATTERN ;
        L->minor = n ;
    }
    else
    {
        if (Super) cholmod_l_free (nsuper + 1, sizeof (Int), Super, Common) ;
        if (Lpi) cholmod_l_free (nsuper + 1, sizeof (Int), Lpi, Common) ;
        if (Lpx) cholmod_l_free (nsuper + 1, sizeof (Int), Lpx, Common) ;
        if (Ls) cholmod_l_free (ssize > 1 ? ssize : 1, sizeof (Int), Ls, Common) ;
    }
    if (Sparent) cholmod_l_free (nsuper + 1, sizeof (Int), Sparent, Common) ;
    if (fill) cholmod_l_free (nsuper + 1, sizeof (Int), fill, Common) ;
    if (seen) cholmod_l_free (nsuper + 1, sizeof (Int), seen, Common) ;
    cholmod_l_free (n + 1, sizeof (fsnode), F, Common) ;
    cholmod_l_free (n + 1, sizeof (Int), col2s, Common) ;
    cholmod_l_free (n + 1, sizeof (Int), kids, Common) ;
    return ok ;
}

int cholmod_l_super_symbolic (cholmod_sparse *A, cholmod_sparse *F, SuiteSparse_long *Parent,
    cholmod_factor *L, cholmod_common *Common)
{
    return cholmod_l_super_symbolic2 (CHOLMOD_ANALYZE_FOR_CHOLESKY, A, F, Parent, L, Common) ;
}

/* ---- analyze -------------------------------------------------------------------------------- */

static cholmod_factor *new_symbolic_factor (Int n, cholmod_common *Common)
{
    cholmod_factor *L = cholmod_l_calloc (1, sizeof (cholmod_factor), Common) ;
    if (!L) return NULL ;
    L->n = n ; L->minor = n ;
    L->is_ll = FALSE ; L->is_super = FALSE ; L->is_monotonic = TRUE ;
    L->itype = CHOLMOD_LONG ; L->xtype = CHOLMOD_PATTERN ; L->dtype = CHOLMOD_DOUBLE ;
    L->ordering = CHOLMOD_NATURAL ;
    L->Perm = cholmod_l_malloc (n, sizeof (Int), Common) ;
    L->ColCount = cholmod_l_malloc (n, sizeof (Int), Common) ;
    if (!L->Perm || !L->ColCount) { cholmod_l_free_factor (&L, Common) ; return NULL ; }
    Int *P = L->Perm, *C = L->ColCount ;
    for (Int j = 0 ; j < n ; j++) { P [j] = j ; C [j] = 1 ; }
    return L ;
}

/* upper- and lower-stored patterns of P A P' */
static int permuted_patterns (cholmod_sparse *A, Int *Perm, cholmod_sparse **U, cholmod_sparse **Lw,
    cholmod_common *Common)
{
    *U = NULL ; *Lw = NULL ;
    if (A->stype > 0)
    {
        *Lw = cholmod_l_ptranspose (A, 0, Perm, NULL, 0, Common) ;
        if (*Lw) *U = cholmod_l_ptranspose (*Lw, 0, NULL, NULL, 0, Common) ;
    }
    else
    {
        *U = cholmod_l_ptranspose (A, 0, Perm, NULL, 0, Common) ;
        if (*U) *Lw = cholmod_l_ptranspose (*U, 0, NULL, NULL, 0, Common) ;
    }
    if (!*U || !*Lw)
    {
        cholmod_l_free_sparse (U, Common) ;
        cholmod_l_free_sparse (Lw, Common) ;
        return FALSE ;
    }
    return TRUE ;
}

/* reference: Cholesky/cholmod_analyze.c:401-935.  Orderings available in this
 * build: the user's permutation (CHOLMOD_GIVEN), the natural ordering, and a
 * built-in nested dissection (order.c) that stands in for the ordering packages
 * (AMD, COLAMD, METIS/NESDIS are not part of this build); the "try several
 * methods, keep the sparsest" loop (:569-804) reduces to the one candidate at
 * hand.  Then etree, column counts, weighted postorder composed
 * into L->Perm (:855-906) and the supernodal symbolic factorization (:913-931). */
static double ssamd_now (void)
{
    struct timespec ts ;
    clock_gettime (CLOCK_MONOTONIC, &ts) ;
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec ;
}

cholmod_factor *cholmod_l_analyze_p2 (int for_whom, cholmod_sparse *A, SuiteSparse_long *UserPerm,
    SuiteSparse_long *fset, size_t fsize, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (A, NULL) ;
    Common->status = CHOLMOD_OK ;
    const int timing = getenv ("CHOLMOD_ANALYZE_TIMING") != NULL ;
    double tt [8] ; tt [0] = ssamd_now () ;
    if (A->stype == 0)
    {
        /* unsymmetric A: analyse A*A', or A(:,f)*A(:,f)' for a column subset f (cholmod_analyze.c:402-418 orders and counts
         * it).  Here its pattern is formed (core.c: ssamd_column_subset, ssamd_aat) and analysed as the symmetric matrix
         * it is. */
        cholmod_sparse *Af = fset ? ssamd_column_subset (A, fset, fsize, 0, Common) : NULL ;
        if (fset && !Af) return NULL ;
        cholmod_sparse *C = ssamd_aat (Af ? Af : A, NULL, 0, TRUE, Common) ;
        if (Af) cholmod_l_free_sparse (&Af, Common) ;
        if (!C) return NULL ;
        cholmod_factor *LC = cholmod_l_analyze_p2 (for_whom, C, UserPerm, NULL, 0, Common) ;
        cholmod_l_free_sparse (&C, Common) ;
        return LC ;
    }
    if (A->nrow != A->ncol) { ERROR (CHOLMOD_INVALID, "matrix invalid") ; return NULL ; }
    if (Common->supernodal == CHOLMOD_SIMPLICIAL)
    {
        ERROR (CHOLMOD_NOT_INSTALLED, "simplicial factorization not built (supernodal only)") ;
        return NULL ;
    }
    Int n = (Int) A->nrow ;
    cholmod_sparse *Apk = NULL ;
    if (!A->packed) { Apk = cholmod_l_copy_sparse (A, Common) ; }
    if (!A->packed)
    {
        /* pack a private copy so the transposes can assume packed input */
        if (!Apk) return NULL ;
        Int *p = Apk->p, *nzc = Apk->nz, *ii = Apk->i ; double *xx = Apk->x ;
        Int dst = 0 ;
        for (Int j = 0 ; j < n ; j++)
        {
            Int s = p [j], e = s + nzc [j] ;
            p [j] = dst ;
            for (Int q = s ; q < e ; q++) { ii [dst] = ii [q] ; if (xx) xx [dst] = xx [q] ; dst++ ; }
        }
        p [n] = dst ;
        cholmod_l_free (Apk->ncol > 0 ? Apk->ncol : 1, sizeof (Int), Apk->nz, Common) ;
        Apk->nz = NULL ; Apk->packed = TRUE ;
        A = Apk ;
    }
    cholmod_factor *L = new_symbolic_factor (n, Common) ;
    Int *Parent = cholmod_l_malloc (n + 1, sizeof (Int), Common) ;
    Int *Post = cholmod_l_malloc (n + 1, sizeof (Int), Common) ;
    Int *work = cholmod_l_malloc (5 * (size_t) n + 1, sizeof (Int), Common) ;
    cholmod_sparse *U = NULL, *Lw = NULL ;
    int ok = L && Parent && Post && work ;
    if (ok)
    {
        Int *Perm = L->Perm ;
        if (UserPerm)
        {
            /* validate (reference cholmod_analyze.c:617-635 via check_perm) */
            for (Int k = 0 ; k < n ; k++) work [k] = 0 ;
            for (Int k = 0 ; k < n && ok ; k++)
            {
                Int j = UserPerm [k] ;
                if (j < 0 || j >= n || work [j]) ok = FALSE ; else work [j] = 1 ;
            }
            if (!ok) ERROR (CHOLMOD_INVALID, "invalid UserPerm") ;
            else for (Int k = 0 ; k < n ; k++) Perm [k] = UserPerm [k] ;
            L->ordering = CHOLMOD_GIVEN ;
        }
        else
        {
            /* no UserPerm.  Default strategy (nmethods == 0; the reference would try
             * AMD and METIS, cholmod_analyze.c:569-804) and any explicit request for
             * an ordering package: the built-in nested dissection (order.c).
             * method [0].ordering == CHOLMOD_NATURAL (or GIVEN without a
             * permutation) keeps the natural order. */
            int want = (Common->nmethods >= 1) ? Common->method [0].ordering : CHOLMOD_NESDIS ;
            if (want == CHOLMOD_NATURAL || want == CHOLMOD_GIVEN) L->ordering = CHOLMOD_NATURAL ;
            else
            {
                ok = ssamd_nested_dissection (n, A->p, A->i, Perm, Common) ;
                L->ordering = CHOLMOD_NESDIS ;
                if (!ok && Common->status == CHOLMOD_OK) ERROR (CHOLMOD_OUT_OF_MEMORY, "ordering failed") ;
            }
        }
    }
    tt [1] = ssamd_now () ;
    ok = ok && permuted_patterns (A, L->Perm, &U, &Lw, Common) ;
    tt [2] = ssamd_now () ;
    if (ok)
    {
        Int *Perm = L->Perm, *ColCount = L->ColCount ;
        double t3a = ssamd_now () ;
        ok = ssamd_etree_upper (n, U->p, U->i, Parent) ;
        double t3b = ssamd_now () ;
        ok = ok && ssamd_postorder (n, Parent, NULL, Post, work) == n ;
        double t3c = ssamd_now () ;
        if (ok) ssamd_colcounts (n, Lw->p, Lw->i, Parent, Post, ColCount, work) ;
        if (timing) fprintf (stderr, "cholmod_l_analyze:   etree %.3f s, postorder %.3f s, colcounts %.3f s\n",
            t3b - t3a, t3c - t3b, ssamd_now () - t3c) ;
        if (ok)
        {
            double fl = 0, lnz = 0 ;
            for (Int j = 0 ; j < n ; j++) { double c = (double) ColCount [j] ; fl += c * c ; lnz += c ; }
            Common->fl = fl ; Common->lnz = lnz ;
            Common->anz = (double) ((Int *) U->p) [n] ;
            Common->method [0].fl = fl ; Common->method [0].lnz = lnz ;
            Common->selected = 0 ;
            /* default strategy (nmethods == 0: the method table is the library's own):
             * the selected method names the ordering that was actually used */
            if (Common->nmethods == 0 && L->ordering != CHOLMOD_POSTORDERED) Common->method [0].ordering = L->ordering ;
        }
        if (ok && Common->postorder)
        {
            double t4a = ssamd_now () ;
            /* (EMPTY: its workspace could not be allocated -- an error, not a reason to skip the postorder silently;
             * found by the fault loop of tests/test_memory_faults.py) */
            const Int npost = ssamd_postorder (n, Parent, ColCount, Post, work) ;
            if (npost == EMPTY) ok = FALSE ;
            else if (npost == n)
            {
                double t4b = ssamd_now () ;
                Int *tmp = work, *inv = work + n ;
                for (Int k = 0 ; k < n ; k++) tmp [k] = Perm [Post [k]] ;
                for (Int k = 0 ; k < n ; k++) Perm [k] = tmp [k] ;
                for (Int k = 0 ; k < n ; k++) tmp [k] = ColCount [Post [k]] ;
                for (Int k = 0 ; k < n ; k++) ColCount [k] = tmp [k] ;
                for (Int k = 0 ; k < n ; k++) inv [Post [k]] = k ;
                for (Int c = 0 ; c < n ; c++)
                {
                    Int op = Parent [Post [c]] ;
                    tmp [c] = (op == EMPTY) ? EMPTY : inv [op] ;
                }
                for (Int k = 0 ; k < n ; k++) Parent [k] = tmp [k] ;
                if (L->ordering == CHOLMOD_NATURAL) L->ordering = CHOLMOD_POSTORDERED ;
                /* only the upper pattern of the final P A P' is needed from here on
                 * (supernodal row structure); one permuted transpose for a lower-stored A */
                cholmod_l_free_sparse (&U, Common) ;
                cholmod_l_free_sparse (&Lw, Common) ;
                if (A->stype < 0) U = cholmod_l_ptranspose (A, 0, Perm, NULL, 0, Common) ;
                else
                {
                    Lw = cholmod_l_ptranspose (A, 0, Perm, NULL, 0, Common) ;
                    if (Lw) U = cholmod_l_ptranspose (Lw, 0, NULL, NULL, 0, Common) ;
                }
                ok = (U != NULL) ;
                if (timing) fprintf (stderr, "cholmod_l_analyze:   weighted postorder %.3f s, compose + permuted pattern %.3f s\n",
                    t4b - t4a, ssamd_now () - t4b) ;
            }
        }
    }
    tt [3] = ssamd_now () ;
    if (ok) ok = cholmod_l_super_symbolic2 (for_whom, U, NULL, Parent, L, Common) ;
    tt [4] = ssamd_now () ;
    if (timing)
        fprintf (stderr, "cholmod_l_analyze: ordering %.3f s, permute %.3f s, etree+colcounts+postorder(+re-permute) %.3f s, super_symbolic %.3f s\n",
            tt [1] - tt [0], tt [2] - tt [1], tt [3] - tt [2], tt [4] - tt [3]) ;
    cholmod_l_free_sparse (&U, Common) ;
    cholmod_l_free_sparse (&Lw, Common) ;
    if (Parent) cholmod_l_free (n + 1, sizeof (Int), Parent, Common) ;
    if (Post) cholmod_l_free (n + 1, sizeof (Int), Post, Common) ;
    if (work) cholmod_l_free (5 * (size_t) n + 1, sizeof (Int), work, Common) ;
    if (Apk) cholmod_l_free_sparse (&Apk, Common) ;
    if (!ok)
    {
        if (Common->status == CHOLMOD_OK) ERROR (CHOLMOD_OUT_OF_MEMORY, "analyze failed") ;
        cholmod_l_free_factor (&L, Common) ;
        return NULL ;
    }
    /* the engine's plan, as the reference cuts its device pools inside the analysis (cholmod_super_symbolic.c:243-327);
     * real matrices: a complex one is factorized through a twin factor that owns the plan (complex.c) */
    if (for_whom == CHOLMOD_ANALYZE_FOR_CHOLESKY && A->xtype == CHOLMOD_REAL && !Common->hip_lazy_plan)
    {
        double tp = ssamd_now () ;
        ssamd_plan_ahead (L, Common) ;
        if (timing) fprintf (stderr, "cholmod_l_analyze: engine plan (schedule, maps, HBM reservation) %.3f s%s\n", ssamd_now () - tp,
            L->hip_plan ? "" : " -- none built") ;
    }
    return L ;
}

cholmod_factor *cholmod_l_analyze_p (cholmod_sparse *A, SuiteSparse_long *UserPerm,
    SuiteSparse_long *fset, size_t fsize, cholmod_common *Common)
{
    return cholmod_l_analyze_p2 (CHOLMOD_ANALYZE_FOR_CHOLESKY, A, UserPerm, fset, fsize, Common) ;
}

cholmod_factor *cholmod_l_analyze (cholmod_sparse *A, cholmod_common *Common)
{
    return cholmod_l_analyze_p2 (CHOLMOD_ANALYZE_FOR_CHOLESKY, A, NULL, NULL, 0, Common) ;
}
