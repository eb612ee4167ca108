/* subset_solve.c -- cholmod_l_solve2 with a sparse right-hand side (Bset): the entries of x = A\b (or of one of the other
 * systems) on the pattern Bset reaches in the elimination tree, in time proportional to the entries of L on that pattern.
 *
 * Reference: Cholesky/cholmod_solve.c:1146-1520 (the Bset branch of solve2), Cholesky/cholmod_rowfac.c:359-545
 * (cholmod_lsolve_pattern: the reach, in topological order), and the simplicial LL' solves it ends in
 * (cholmod_solve.c:196-330 -> t_cholmod_lsolve.c / t_cholmod_ltsolve.c).  The reference first converts a supernodal L into
 * a simplicial one -- column j = the rows of its supernode from its own row down, with every entry of the supernode, zero or
 * not (Core/t_cholmod_change_factor.c:444-514) -- and leaves L in that form.  Here the same columns are read in place from
 * the supernodal L (same pattern, hence the same reach; same operations per entry) and L STAYS supernodal: the simplicial
 * numeric factorization a later cholmod_l_factorize of a converted L would need is outside this library (DESIGN.md
 * section 8).  The values are read on the host: a factor that lives in HBM only is downloaded once (L->x, kept until the
 * next factorization).
 */
#include <string.h>
#include "host_internal.h"

int cholmod_l_factor_to_host (cholmod_factor *L, cholmod_common *Common) ;

/* per-factor workspace of the subset solves, built by the first one (as the reference builds L->IPerm then,
 * cholmod_solve.c:1196-1218): column -> supernode [0, n), Flag [n, 2n), the current mark [2n] */
static Int *subset_work (cholmod_factor *L, cholmod_common *Common)
{
    if (L->bset_work) return (Int *) L->bset_work ;
    const Int n = (Int) L->n, nsuper = (Int) L->nsuper ;
    Int *W = cholmod_l_malloc (2 * (size_t) n + 1, sizeof (Int), Common) ;
    if (!W) return NULL ;
    const Int *Super = L->super ;
    for (Int s = 0 ; s < nsuper ; s++)
        for (Int k = Super [s] ; k < Super [s+1] ; k++) W [k] = s ;
    for (Int k = 0 ; k <= n ; k++) W [n + k] = 0 ;
    L->bset_work = W ;
    return W ;
}

/* the first row below the diagonal in column j of L: its parent in the elimination tree of the factor as stored
 * (cholmod_rowfac.c:518, PARENT) */
static inline Int parent_of (Int j, const Int *Super, const Int *Lpi, const Int *Ls, const Int *col_super)
{
    const Int s = col_super [j] ;
    if (j + 1 < Super [s+1]) return j + 1 ;
    const Int nscol = Super [s+1] - Super [s], nsrow = Lpi [s+1] - Lpi [s] ;
    return nsrow > nscol ? Ls [Lpi [s] + nscol] : EMPTY ;
}

int ssamd_solve_subset (int sys, cholmod_factor *L, cholmod_dense *B, cholmod_sparse *Bset, cholmod_dense *X,
    cholmod_sparse **Xset_Handle, cholmod_dense **Y_Handle, cholmod_common *Common)
{
    const Int n = (Int) L->n ;
    if (!L->is_super || !L->is_ll || (L->xtype != CHOLMOD_REAL && L->xtype != CHOLMOD_COMPLEX))
    { ERROR (CHOLMOD_INVALID, "L must be a numeric supernodal factor") ; return FALSE ; }
    if (!Xset_Handle) { ERROR (CHOLMOD_INVALID, "argument missing") ; return FALSE ; }
    if (Bset->nrow != (size_t) n || Bset->ncol != 1 || Bset->stype != 0 || Bset->itype != CHOLMOD_LONG)
    { ERROR (CHOLMOD_INVALID, "Bset must be an n-by-1 unsymmetric sparse column") ; return FALSE ; }
    const int cplx = (L->xtype == CHOLMOD_COMPLEX) ;
    const int need_values = !(sys == CHOLMOD_P || sys == CHOLMOD_Pt || sys == CHOLMOD_D) ;
    if (need_values && !(L->x && (!L->hip_on_device || L->hip_host_valid)) && !cholmod_l_factor_to_host (L, Common)) return FALSE ;

    /* Perm only for the systems that permute, and not for the natural ordering; IPerm for x = A\b and x = Pb
     * (cholmod_solve.c:1101-1108, :1196-1226) */
    const Int *Perm = ((sys == CHOLMOD_P || sys == CHOLMOD_Pt || sys == CHOLMOD_A) && L->ordering != CHOLMOD_NATURAL) ? L->Perm : NULL ;
    const Int *IPerm = NULL ;
    if ((sys == CHOLMOD_A || sys == CHOLMOD_P) && Perm)
    {
        if (!L->IPerm)
        {
            Int *ip = cholmod_l_malloc ((size_t) n, sizeof (Int), Common) ;
            if (!ip) return FALSE ;
            for (Int k = 0 ; k < n ; k++) ip [Perm [k]] = k ;
            L->IPerm = ip ;
        }
        IPerm = L->IPerm ;
    }
    if (sys == CHOLMOD_P) Perm = NULL ;             /* (no P' at the end) */

    /* Xset: n-by-1, pattern only, packed, unsorted, room for n entries (cholmod_solve.c:1236-1252) */
    cholmod_sparse *Xset = *Xset_Handle ;
    if (!Xset || Xset->nrow != (size_t) n || Xset->ncol != 1 || Xset->nzmax < (size_t) n || Xset->xtype != CHOLMOD_PATTERN)
    {
        cholmod_l_free_sparse (Xset_Handle, Common) ;
        Xset = cholmod_l_allocate_sparse ((size_t) n, 1, (size_t) n, FALSE, TRUE, 0, CHOLMOD_PATTERN, Common) ;
        if (!Xset) return FALSE ;
        *Xset_Handle = Xset ;
    }
    Xset->sorted = FALSE ; Xset->stype = 0 ;
    Int *Xseti = Xset->i, *Xsetp = Xset->p ;

    Int *W = subset_work (L, Common) ;
    if (!W) return FALSE ;
    const Int *col_super = W ;
    Int *Flag = W + n ;
    /* (a fresh mark per call; the flags are cleared when it would wrap, cholmod_clear_flag) */
    if (W [2*n] >= (Int) 0x7ffffffffffffff0ll) { for (Int k = 0 ; k < n ; k++) Flag [k] = 0 ; W [2*n] = 0 ; }
    const Int mark = ++W [2*n] ;

    /* Y: the reference keeps a 1-by-n workspace in *Y_Handle between calls (cholmod_solve.c:1185) */
    cholmod_dense *Y = Y_Handle ? *Y_Handle : NULL, *Ytmp = NULL ;
    if (!Y || Y->nzmax < (size_t) n || Y->xtype != L->xtype)
    {
        if (Y_Handle) cholmod_l_free_dense (Y_Handle, Common) ;
        Y = cholmod_l_allocate_dense (1, (size_t) n, 1, L->xtype, Common) ;
        if (!Y) return FALSE ;
        if (Y_Handle) *Y_Handle = Y ; else Ytmp = Y ;
    }
    double *Yx = Y->x ;

    const Int *Bsetp = Bset->p, *Bseti = Bset->i, *Bsetnz = Bset->nz ;
    const Int blen = Bset->packed ? Bsetp [1] : Bsetnz [0] ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s ;
    for (Int p = 0 ; p < blen ; p++)
        if (Bseti [p] < 0 || Bseti [p] >= n)
        {
            ERROR (CHOLMOD_INVALID, "Bset holds an index outside 0 .. n-1") ;
            if (Ytmp) cholmod_l_free_dense (&Ytmp, Common) ;
            return FALSE ;
        }

    /* ---- Yset: the pattern of L \ (P Bset) in topological order, or P Bset itself (:1338-1356).  It is built in Xseti --
     * the stack grows down from n, every path root-last, then the whole stack moves to the front (rowfac.c:101-118, :521-545) */
    Int *Yseti = Xseti ;
    Int ysetlen ;
    if (!need_values)
    {
        /* (an index listed twice enters once, as in the reach below: Xset has n slots, a Bset with duplicates may be longer) */
        ysetlen = 0 ;
        for (Int p = 0 ; p < blen ; p++)
        {
            const Int i = IPerm ? IPerm [Bseti [p]] : Bseti [p] ;
            if (Flag [i] == mark) continue ;
            Flag [i] = mark ;
            Yseti [ysetlen++] = i ;
        }
    }
    else
    {
        Int top = n ;
        /* (the walk of one entry is collected at the low end of the same array: len <= top always holds, because every
         * index enters exactly once) */
        for (Int p = 0 ; p < blen ; p++)
        {
            Int i = IPerm ? IPerm [Bseti [p]] : Bseti [p], len = 0 ;
            for ( ; i != EMPTY && Flag [i] != mark ; i = parent_of (i, Super, Lpi, Ls, col_super))
            {
                Yseti [len++] = i ;
                Flag [i] = mark ;
            }
            while (len > 0) Yseti [--top] = Yseti [--len] ;
        }
        ysetlen = n - top ;
        for (Int k = 0 ; k < ysetlen ; k++) Yseti [k] = Yseti [top + k] ;
    }

    /* ---- Y (Yset) = 0, Y (P Bset) = B (Bset)  (:1362-1436) */
    const double *Bx = B->x ;
    if (cplx) for (Int k = 0 ; k < ysetlen ; k++) { Yx [2 * Yseti [k]] = 0 ; Yx [2 * Yseti [k] + 1] = 0 ; }
    else for (Int k = 0 ; k < ysetlen ; k++) Yx [Yseti [k]] = 0 ;
    for (Int p = 0 ; p < blen ; p++)
    {
        const Int iold = Bseti [p], inew = IPerm ? IPerm [iold] : iold ;
        if (cplx) { Yx [2*inew] = Bx [2*iold] ; Yx [2*inew+1] = Bx [2*iold+1] ; }
        else Yx [inew] = Bx [iold] ;
    }

    /* ---- the solves over the columns of Yset (LL': D = I, so LD = L and DLt = Lt; cholmod_solve.c:196-330) */
    const double *Lx = L->x ;
    const int fwd = need_values && (sys == CHOLMOD_A || sys == CHOLMOD_LDLt || sys == CHOLMOD_L || sys == CHOLMOD_LD) ;
    const int bwd = need_values && (sys == CHOLMOD_A || sys == CHOLMOD_LDLt || sys == CHOLMOD_Lt || sys == CHOLMOD_DLt) ;
    if (fwd)
        for (Int k = 0 ; k < ysetlen ; k++)
        {
            /* y (j) /= L (j, j) ; y (i) -= L (i, j) y (j) for the rows below */
            const Int j = Yseti [k], s = col_super [j], jj = j - Super [s], nsrow = Lpi [s+1] - Lpi [s] ;
            const Int *rows = Ls + Lpi [s] ;
            if (cplx)
            {
                const double *col = Lx + 2 * (Lpx [s] + jj * nsrow) ;
                const double d = col [2*jj] ;                                   /* (the diagonal of L is real) */
                const double yr = Yx [2*j] / d, yi = Yx [2*j+1] / d ;
                Yx [2*j] = yr ; Yx [2*j+1] = yi ;
                for (Int ii = jj + 1 ; ii < nsrow ; ii++)
                {
                    const Int i = rows [ii] ;
                    const double lr = col [2*ii], li = col [2*ii+1] ;
                    Yx [2*i] -= lr * yr - li * yi ;
                    Yx [2*i+1] -= lr * yi + li * yr ;
                }
            }
            else
            {
                const double *col = Lx + Lpx [s] + jj * nsrow ;
                const double y = Yx [j] / col [jj] ;
                Yx [j] = y ;
                for (Int ii = jj + 1 ; ii < nsrow ; ii++) Yx [rows [ii]] -= col [ii] * y ;
            }
        }
    if (bwd)
        for (Int k = ysetlen - 1 ; k >= 0 ; k--)
        {
            /* y (j) = (y (j) - sum_i conj (L (i, j)) y (i)) / L (j, j): every row i of column j is an ancestor of j, in Yset */
            const Int j = Yseti [k], s = col_super [j], jj = j - Super [s], nsrow = Lpi [s+1] - Lpi [s] ;
            const Int *rows = Ls + Lpi [s] ;
            if (cplx)
            {
                const double *col = Lx + 2 * (Lpx [s] + jj * nsrow) ;
                double yr = Yx [2*j], yi = Yx [2*j+1] ;
                for (Int ii = jj + 1 ; ii < nsrow ; ii++)
                {
                    const Int i = rows [ii] ;
                    const double lr = col [2*ii], li = col [2*ii+1] ;
                    yr -= lr * Yx [2*i] + li * Yx [2*i+1] ;
                    yi -= lr * Yx [2*i+1] - li * Yx [2*i] ;
                }
                const double d = col [2*jj] ;
                Yx [2*j] = yr / d ; Yx [2*j+1] = yi / d ;
            }
            else
            {
                const double *col = Lx + Lpx [s] + jj * nsrow ;
                double y = Yx [j] ;
                for (Int ii = jj + 1 ; ii < nsrow ; ii++) y -= col [ii] * Yx [rows [ii]] ;
                Yx [j] = y / col [jj] ;
            }
        }

    /* ---- X (Perm (Yset)) = Y (Yset), Xset = Perm (Yset)  (:1469-1517); the other entries of X are not touched */
    double *Xx = X->x, *Xz = X->z ;
    for (Int k = 0 ; k < ysetlen ; k++)
    {
        const Int inew = Yseti [k], iold = Perm ? Perm [inew] : inew ;
        if (!cplx) Xx [iold] = Yx [inew] ;
        else if (X->xtype == CHOLMOD_COMPLEX) { Xx [2*iold] = Yx [2*inew] ; Xx [2*iold+1] = Yx [2*inew+1] ; }
        else { Xx [iold] = Yx [2*inew] ; Xz [iold] = Yx [2*inew+1] ; }
        Xseti [k] = iold ;
    }
    Xsetp [0] = 0 ; Xsetp [1] = ysetlen ;
    if (Ytmp) cholmod_l_free_dense (&Ytmp, Common) ;
    return TRUE ;
}
