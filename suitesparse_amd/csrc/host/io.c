/* io.c -- harness-level utilities the demo path needs: triplet / Matrix-Market
 * reader, factor invariants, sparse*dense product, norms, statistics printer.
 * Reference files: CHOLMOD/Check/cholmod_read.c, cholmod_check.c;
 * CHOLMOD/MatrixOps/cholmod_sdmult.c, cholmod_norm.c. */
#include "host_internal.h"
#include <ctype.h>

/* ---- reader (format: reference Check/cholmod_read.c:14-110) ------------------------- */

static int next_data_line (FILE *f, char *buf, size_t cap, char *mm_sym)
{
    while (fgets (buf, (int) cap, f))
    {
        char *p = buf ;
        while (*p && isspace ((unsigned char) *p)) p++ ;
        if (*p == '\0') continue ;
        if (*p == '%')
        {
            if (mm_sym && strncasecmp (p, "%%MatrixMarket", 14) == 0)
            {
                /* %%MatrixMarket matrix <fmt> <type> <storage> */
                char w [5][64] = {{0}} ;
                sscanf (p, "%63s %63s %63s %63s %63s", w [0], w [1], w [2], w [3], w [4]) ;
                char c0 = (char) tolower ((unsigned char) w [4][0]) ;
                char c1 = (char) tolower ((unsigned char) w [4][1]) ;
                *mm_sym = (c0 == 's' && c1 == 'k') ? 'k' : c0 ;
            }
            continue ;
        }
        return TRUE ;
    }
    return FALSE ;
}

/* Returns a real sparse matrix; symmetric inputs come back upper-stored when
 * Common->prefer_upper (the default), as the reference (cholmod_read.c:1135-1190). */
cholmod_sparse *cholmod_l_read_sparse (FILE *f, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (f, NULL) ;
    Common->status = CHOLMOD_OK ;
    char buf [1024] ;
    char mm = 0 ;
    if (!next_data_line (f, buf, sizeof buf, &mm)) { ERROR (CHOLMOD_INVALID, "premature EOF") ; return NULL ; }
    double h [4] = {0, 0, 0, 0} ;
    int nh = sscanf (buf, "%lg %lg %lg %lg", &h [0], &h [1], &h [2], &h [3]) ;
    if (nh < 3) { ERROR (CHOLMOD_INVALID, "invalid header (dense 'array' files are not built)") ; return NULL ; }
    Int nrow = (Int) h [0], ncol = (Int) h [1], nnz = (Int) h [2] ;
    int stype_known = 0, stype = 0 ;
    if (nh >= 4) { stype = (int) h [3] ; stype_known = 1 ; }
    else if (mm) { stype = (mm == 's' || mm == 'h') ? -1 : 0 ; stype_known = 1 ; }
    if (nrow < 0 || ncol < 0 || nnz < 0) { ERROR (CHOLMOD_INVALID, "invalid header") ; return NULL ; }
    cholmod_triplet *T = cholmod_l_allocate_triplet (nrow, ncol, nnz, 0, CHOLMOD_REAL, Common) ;
    if (!T) return NULL ;
    Int *Ti = T->i, *Tj = T->j ;
    double *Tx = T->x ;
    int one_based = TRUE, pattern = FALSE ;
    Int k ;
    for (k = 0 ; k < nnz ; k++)
    {
        if (!next_data_line (f, buf, sizeof buf, NULL)) { ERROR (CHOLMOD_INVALID, "premature EOF") ; break ; }
        double a = 0, b = 0, v = 1, w = 0 ;
        int nt = sscanf (buf, "%lg %lg %lg %lg", &a, &b, &v, &w) ;
        if (nt < 2) { ERROR (CHOLMOD_INVALID, "invalid matrix file") ; break ; }
        if (nt >= 4) { ERROR (CHOLMOD_NOT_INSTALLED, "complex matrices not built") ; break ; }
        if (nt == 2) { pattern = TRUE ; v = 1 ; }
        Ti [k] = (Int) a ; Tj [k] = (Int) b ; Tx [k] = v ;
        if (Ti [k] == 0 || Tj [k] == 0) one_based = FALSE ;
    }
    if (k < nnz) { cholmod_l_free_triplet (&T, Common) ; return NULL ; }
    T->nnz = nnz ;
    if (one_based) for (k = 0 ; k < nnz ; k++) { Ti [k]-- ; Tj [k]-- ; }
    if (!stype_known)
    {
        int lo = FALSE, up = FALSE ;
        for (k = 0 ; k < nnz ; k++) { if (Ti [k] > Tj [k]) lo = TRUE ; if (Ti [k] < Tj [k]) up = TRUE ; }
        stype = (nrow != ncol || (lo && up)) ? 0 : (up ? 1 : -1) ;
    }
    if (mm == 'k') stype = 0 ;      /* skew-symmetric: returned unsymmetric in the reference */
    T->stype = stype ;
    if (pattern && stype != 0)
    {
        /* symmetric pattern: diagonal = degree+1, off-diagonals -1 (reader notes :106-110) */
        Int *deg = cholmod_l_calloc (nrow + 1, sizeof (Int), Common) ;
        if (deg)
        {
            for (k = 0 ; k < nnz ; k++) if (Ti [k] != Tj [k]) { deg [Ti [k]]++ ; deg [Tj [k]]++ ; }
            for (k = 0 ; k < nnz ; k++) Tx [k] = (Ti [k] == Tj [k]) ? (double) (deg [Ti [k]] + 1) : -1.0 ;
            cholmod_l_free (nrow + 1, sizeof (Int), deg, Common) ;
        }
    }
    cholmod_sparse *A = cholmod_l_triplet_to_sparse (T, 0, Common) ;
    cholmod_l_free_triplet (&T, Common) ;
    if (A && A->stype < 0 && Common->prefer_upper)
    {
        cholmod_sparse *A2 = cholmod_l_ptranspose (A, 2, NULL, NULL, 0, Common) ;
        cholmod_l_free_sparse (&A, Common) ;
        A = A2 ;
    }
    return A ;
}

/* ---- invariants (reference Check/cholmod_check.c:1823-2000, :1618-1637) ------------ */

int cholmod_l_check_factor (cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    Common->status = CHOLMOD_OK ;
#define BAD(msg) do { ERROR (CHOLMOD_INVALID, msg) ; return FALSE ; } while (0)
    Int n = (Int) L->n ;
    if (L->itype != CHOLMOD_LONG || L->dtype != CHOLMOD_DOUBLE) BAD ("invalid itype/dtype") ;
    const Int *Perm = L->Perm, *CC = L->ColCount ;
    if (!Perm || !CC) BAD ("Perm or ColCount missing") ;
    Int *seen = cholmod_l_calloc (n + 1, sizeof (Int), Common) ;
    if (!seen) return FALSE ;
    int okp = TRUE ;
    for (Int k = 0 ; k < n && okp ; k++)
    {
        Int j = Perm [k] ;
        if (j < 0 || j >= n || seen [j]) okp = FALSE ; else seen [j] = 1 ;
        if (CC [k] < 0 || CC [k] > n - k) okp = FALSE ;
    }
    cholmod_l_free (n + 1, sizeof (Int), seen, Common) ;
    if (!okp) BAD ("invalid permutation or column count") ;
    if (!L->is_super) return TRUE ;
    const Int *Sup = L->super, *pi = L->pi, *px = L->px, *s = L->s ;
    if (!Sup || !pi || !px || !s) BAD ("supernodal arrays missing") ;
    Int nsuper = (Int) L->nsuper ;
    if (s [0] == EMPTY) BAD ("supernodes not defined") ;
    if (pi [0] != 0 || (pi [nsuper] > 1 ? pi [nsuper] : 1) != (Int) L->ssize) BAD ("invalid pi") ;
    /* px [0] == 123456: a factor analysed for SPQR without the GPU carries no px
     * (CHOLMOD/Supernodal/cholmod_super_symbolic.c:769, Check/cholmod_check.c:1880) */
    const int check_px = (px [0] != 123456) ;
    if (check_px && (px [0] != 0 || (px [nsuper] > 1 ? px [nsuper] : 1) != (Int) L->xsize)) BAD ("invalid px") ;
    for (Int q = 0 ; q < nsuper ; q++)
    {
        Int k1 = Sup [q], k2 = Sup [q+1] ;
        Int nscol = k2 - k1, nsrow = pi [q+1] - pi [q] ;
        if (k1 > k2 || k1 < 0 || k2 > n || nsrow < nscol) BAD ("invalid supernode") ;
        if (check_px && px [q+1] - px [q] != nsrow * nscol) BAD ("invalid supernode") ;
        for (Int r = 0 ; r < nsrow ; r++)
        {
            Int i = s [pi [q] + r] ;
            if (r < nscol) { if (i != k1 + r) BAD ("row index invalid") ; }
            else if (i <= s [pi [q] + r - 1] || i >= n) BAD ("row index out of range") ;
        }
    }
#undef BAD
    return TRUE ;
}

int cholmod_l_check_sparse (cholmod_sparse *A, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (A, FALSE) ;
    Common->status = CHOLMOD_OK ;
    const Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    if (!Ap || !Ai || (!A->packed && !Anz)) { ERROR (CHOLMOD_INVALID, "invalid") ; return FALSE ; }
    if (A->stype != 0 && A->nrow != A->ncol) { ERROR (CHOLMOD_INVALID, "invalid") ; return FALSE ; }
    for (size_t j = 0 ; j < A->ncol ; j++)
    {
        Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j] ;
        if (p < 0 || pend > (Int) A->nzmax || p > pend) { ERROR (CHOLMOD_INVALID, "invalid") ; return FALSE ; }
        for ( ; p < pend ; p++)
            if (Ai [p] < 0 || Ai [p] >= (Int) A->nrow) { ERROR (CHOLMOD_INVALID, "invalid") ; return FALSE ; }
    }
    return TRUE ;
}

/* reference: Check/cholmod_check.c:604-649 */
int cholmod_l_gpu_stats (cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    if (Common->print < 2) return TRUE ;
    printf ("\nCHOLMOD HIP engine statistics (MI355X / gfx950):\n") ;
    printf ("device time    %12.4f s   kernel launches %d\n", Common->gpuKernelTime, Common->gpuNumKernelLaunches) ;
    printf ("executed flops %12.4e     -> %.2f GFLOP/s on the device\n", (double) Common->gpuFlops,
        Common->gpuKernelTime > 0 ? 1e-9 * (double) Common->gpuFlops / Common->gpuKernelTime : 0.0) ;
    printf ("dense updates  %12.4f s   (%lu launches; syrk+gemm of the reference)\n",
        Common->cholmod_gpu_syrk_time, (unsigned long) Common->cholmod_gpu_syrk_calls) ;
    printf ("potrf          %12.4f s\ntrsm           %12.4f s\n", Common->cholmod_gpu_potrf_time,
        Common->cholmod_gpu_trsm_time) ;
    printf ("assemble A     %12.4f s\nextend-add     %12.4f s\n", Common->cholmod_assemble_time,
        Common->cholmod_assemble_time2) ;
    printf ("(per-class times are collected only with Common->hip_profile)\n") ;
    return TRUE ;
}

/* ---- sdmult / norms (reference MatrixOps/cholmod_sdmult.c:54, cholmod_norm.c:66,249) - */

int cholmod_l_sdmult (cholmod_sparse *A, int transpose, double alpha [2], double beta [2],
    cholmod_dense *X, cholmod_dense *Y, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (A, FALSE) ;
    RETURN_IF_NULL (X, FALSE) ;
    RETURN_IF_NULL (Y, FALSE) ;
    if (A->xtype != CHOLMOD_REAL || X->xtype != CHOLMOD_REAL || Y->xtype != CHOLMOD_REAL)
    { ERROR (CHOLMOD_INVALID, "real matrices only") ; return FALSE ; }
    Int ny = transpose ? (Int) A->ncol : (Int) A->nrow ;
    Int nx = transpose ? (Int) A->nrow : (Int) A->ncol ;
    if (X->nrow != (size_t) nx || X->ncol != Y->ncol || Y->nrow != (size_t) ny)
    { ERROR (CHOLMOD_INVALID, "X and/or Y have wrong dimensions") ; return FALSE ; }
    Common->status = CHOLMOD_OK ;
    const Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    const double *Ax = A->x ;
    double al = alpha ? alpha [0] : 1.0, be = beta ? beta [0] : 0.0 ;
    for (size_t r = 0 ; r < X->ncol ; r++)
    {
        const double *x = (double *) X->x + r * X->d ;
        double *y = (double *) Y->x + r * Y->d ;
        if (be == 0) for (Int i = 0 ; i < ny ; i++) y [i] = 0 ;
        else if (be != 1) for (Int i = 0 ; i < ny ; i++) y [i] *= be ;
        for (Int j = 0 ; j < (Int) A->ncol ; j++)
        {
            Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j] ;
            for ( ; p < pend ; p++)
            {
                Int i = Ai [p] ;
                double a = Ax [p] ;
                if (A->stype == 0)
                {
                    if (transpose) y [j] += al * a * x [i] ; else y [i] += al * a * x [j] ;
                }
                else
                {
                    if ((A->stype > 0 && i > j) || (A->stype < 0 && i < j)) continue ;
                    y [i] += al * a * x [j] ;
                    if (i != j) y [j] += al * a * x [i] ;
                }
            }
        }
    }
    return TRUE ;
}

double cholmod_l_norm_dense (cholmod_dense *X, int norm, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (EMPTY) ;
    RETURN_IF_NULL (X, EMPTY) ;
    if (norm < 0 || norm > 2 || (norm == 2 && X->ncol > 1)) { ERROR (CHOLMOD_INVALID, "invalid norm") ; return EMPTY ; }
    Common->status = CHOLMOD_OK ;
    /* |x| of an entry in each of the three numeric layouts (reference: abs_value,
     * CHOLMOD/MatrixOps/cholmod_norm.c:34-60): real, complex (interleaved), zomplex (x and z) */
    if (X->xtype < CHOLMOD_REAL || X->xtype > CHOLMOD_ZOMPLEX || !X->x || (X->xtype == CHOLMOD_ZOMPLEX && !X->z))
    { ERROR (CHOLMOD_INVALID, "invalid xtype") ; return EMPTY ; }
    const double *x = X->x, *z = X->z ;
    const int xt = X->xtype ;
#define ABS_AT(p_) (xt == CHOLMOD_REAL ? fabs (x [p_]) : xt == CHOLMOD_COMPLEX ? hypot (x [2 * (p_)], x [2 * (p_) + 1]) \
    : hypot (x [p_], z [p_]))
    double res = 0 ;
    if (norm == 0)
    {
        for (size_t i = 0 ; i < X->nrow ; i++)
        {
            double s = 0 ;
            for (size_t j = 0 ; j < X->ncol ; j++) s += ABS_AT (i + j * X->d) ;
            if (s > res || s != s) res = s ;
        }
    }
    else if (norm == 1)
    {
        for (size_t j = 0 ; j < X->ncol ; j++)
        {
            double s = 0 ;
            for (size_t i = 0 ; i < X->nrow ; i++) s += ABS_AT (i + j * X->d) ;
            if (s > res || s != s) res = s ;
        }
    }
    else
    {
        /* (cholmod_norm.c:171-204: sum of x^2 (+ z^2) over the column, then the root) */
        for (size_t i = 0 ; i < X->nrow ; i++)
        {
            if (xt == CHOLMOD_REAL) res += x [i] * x [i] ;
            else if (xt == CHOLMOD_COMPLEX) res += x [2 * i] * x [2 * i] + x [2 * i + 1] * x [2 * i + 1] ;
            else res += x [i] * x [i] + z [i] * z [i] ;
        }
        res = sqrt (res) ;
    }
#undef ABS_AT
    return res ;
}

double cholmod_l_norm_sparse (cholmod_sparse *A, int norm, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (EMPTY) ;
    RETURN_IF_NULL (A, EMPTY) ;
    if (norm < 0 || norm > 1) { ERROR (CHOLMOD_INVALID, "invalid norm") ; return EMPTY ; }
    if (A->xtype != CHOLMOD_REAL) { ERROR (CHOLMOD_INVALID, "real matrices only") ; return EMPTY ; }
    Common->status = CHOLMOD_OK ;
    const Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    const double *Ax = A->x ;
    Int nr = (Int) A->nrow, nc = (Int) A->ncol ;
    Int nw = nr > nc ? nr : nc ;
    double *w = cholmod_l_calloc (nw + 1, sizeof (double), Common) ;
    if (!w) return EMPTY ;
    /* for symmetric storage the row sums equal the column sums */
    for (Int j = 0 ; j < nc ; j++)
    {
        Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j] ;
        for ( ; p < pend ; p++)
        {
            Int i = Ai [p] ;
            double a = fabs (Ax [p]) ;
            if (A->stype == 0) { if (norm == 0) w [i] += a ; else w [j] += a ; }
            else
            {
                if ((A->stype > 0 && i > j) || (A->stype < 0 && i < j)) continue ;
                w [i] += a ;
                if (i != j) w [j] += a ;
            }
        }
    }
    double res = 0 ;
    Int lim = (A->stype != 0) ? nr : (norm == 0 ? nr : nc) ;
    for (Int i = 0 ; i < lim ; i++) if (w [i] > res || w [i] != w [i]) res = w [i] ;
    cholmod_l_free (nw + 1, sizeof (double), w, Common) ;
    return res ;
}
