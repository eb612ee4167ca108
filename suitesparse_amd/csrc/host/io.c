/* io.c -- harness-level utilities the demo path needs: triplet / Matrix-Market
 * reader, factor invariants, sparse*dense product, norms, statistics printer.
 * Reference files: CHOLMOD/Check/cholmod_read.c, cholmod_check.c;
 * CHOLMOD/MatrixOps/cholmod_sdmult.c, cholmod_norm.c. */
#include "host_internal.h"
#include <ctype.h>

/* ---- reader (format: reference Check/cholmod_read.c:9-140) ---------------------------
 * One pass over the lines of the file: '%' lines are comments (the first one may be the
 * Matrix Market banner), blank lines are skipped, the first data line is the header
 * (2 numbers: a dense "array"; 3 or 4: triplets, the 4th the stype), every later data line
 * one entry whose token count gives the type (coordinate 2 pattern / 3 real / 4 complex;
 * array 1 real / 2 complex). */

#define RD_LINE 1030            /* the reference's MAXLINE (cholmod_read.c:154) + slack */
enum { RD_GENERAL = 0, RD_LOWER = -1, RD_UPPER = 1, RD_SKEW = -2, RD_CSYM = -3, RD_UNKNOWN = 999 } ;

typedef struct
{
    int have_banner ;       /* %%MatrixMarket seen on the first line */
    int array ;             /* banner says "array" (dense) */
    int stype ;             /* RD_* */
    double h [4] ;          /* header numbers */
    int nh ;
} rd_head ;

static int rd_blank (const char *p)
{
    if (*p == '%') return TRUE ;
    while (*p) { if (!isspace ((unsigned char) *p)) return FALSE ; p++ ; }
    return TRUE ;
}

/* values at or beyond 1e308 stand for +-Inf in these files (cholmod_read.c:173-185) */
static double rd_value (double x) { return (x >= 1e308 || x <= -1e308) ? 2 * x : x ; }

static int rd_header (FILE *f, char *buf, rd_head *H)
{
    int first = TRUE ;
    memset (H, 0, sizeof (*H)) ;
    H->stype = RD_UNKNOWN ;
    while (fgets (buf, RD_LINE, f))
    {
        if (first && strncasecmp (buf, "%%MatrixMarket", 14) == 0)
        {
            /* %%MatrixMarket matrix <fmt> <type> <storage>: first letters only */
            char w [5][64] = {{0}} ;
            if (sscanf (buf, "%63s %63s %63s %63s %63s", w [0], w [1], w [2], w [3], w [4]) < 5) return FALSE ;
            int fmt = tolower ((unsigned char) w [2][0]), typ = tolower ((unsigned char) w [3][0]) ;
            int s0 = tolower ((unsigned char) w [4][0]), s1 = tolower ((unsigned char) w [4][1]) ;
            if (tolower ((unsigned char) w [1][0]) != 'm' || (fmt != 'c' && fmt != 'a')) return FALSE ;
            if (typ != 'r' && typ != 'c' && typ != 'p' && typ != 'i') return FALSE ;
            H->have_banner = TRUE ;
            H->array = (fmt == 'a') ;
            if (s0 == 'g') H->stype = RD_GENERAL ;
            else if (s0 == 's' && s1 == 'k') H->stype = RD_SKEW ;
            else if (s0 == 's') H->stype = (typ == 'c') ? RD_CSYM : RD_LOWER ;
            else if (s0 == 'h') H->stype = RD_LOWER ;
            else return FALSE ;
            first = FALSE ;
            continue ;
        }
        first = FALSE ;
        if (rd_blank (buf)) continue ;
        H->h [0] = H->h [1] = -1 ;
        H->nh = sscanf (buf, "%lg %lg %lg %lg", &H->h [0], &H->h [1], &H->h [2], &H->h [3]) ;
        if (H->nh < 2 || H->nh > 4 || H->h [0] < 0 || H->h [1] < 0 || H->h [0] > 9.2e18 || H->h [1] > 9.2e18) return FALSE ;
        if (H->nh == 2 && !H->have_banner) { H->array = TRUE ; H->stype = RD_GENERAL ; }
        if (H->nh == 2 && !H->array) return FALSE ;
        if (H->nh >= 3 && !H->have_banner) H->array = FALSE ;
        if (H->nh >= 3 && H->h [2] < 0) return FALSE ;
        if (H->nh == 4) H->stype = H->h [3] < 0 ? RD_LOWER : H->h [3] > 0 ? RD_UPPER : RD_GENERAL ;
        if (H->h [0] != H->h [1]) H->stype = RD_GENERAL ;         /* rectangular: unsymmetric */
        return TRUE ;
    }
    return FALSE ;
}

static cholmod_triplet *rd_triplets (FILE *f, char *buf, const rd_head *H, int prefer_unsym, cholmod_common *Common)
{
    size_t nrow = (size_t) H->h [0], ncol = (size_t) H->h [1], nnz = (size_t) H->h [2] ;
    int stype = H->stype ;
    if (nrow == 0 || ncol == 0 || nnz == 0) return cholmod_l_allocate_triplet (nrow, ncol, 0, 0, CHOLMOD_REAL, Common) ;
    const int unknown = (stype == RD_UNKNOWN), skew = (stype == RD_SKEW), csym = (stype == RD_CSYM) ;
    /* skew-symmetric and complex symmetric files come back with both triangles (stype 0); so does everything when the
     * caller prefers unsymmetric, and an unknown stype may turn out to be (cholmod_read.c:531-545) */
    size_t extra = (stype < RD_LOWER || unknown || (prefer_unsym && stype != RD_GENERAL)) ? nnz : 0 ;
    if (extra) stype = unknown ? RD_UNKNOWN : RD_GENERAL ;
    if (nnz > (size_t) 1 << 60 || nrow > (size_t) 1 << 60 || ncol > (size_t) 1 << 60)
    { ERROR (CHOLMOD_TOO_LARGE, "problem too large") ; return NULL ; }
    cholmod_triplet *T = NULL ;
    Int *Ti = NULL, *Tj = NULL ;
    double *Tx = NULL ;
    int xtype = CHOLMOD_PATTERN, ntok = 0, lower_only = TRUE, upper_only = TRUE, one_based = TRUE ;
    Int imax = 0, jmax = 0 ;
    for (size_t k = 0 ; k < nnz ; k++)
    {
        double a = -1, b = -1, x = 0, z = 0 ;
        int nt = 0 ;
        for ( ; ; )
        {
            if (!fgets (buf, RD_LINE, f))
            {
                cholmod_l_free_triplet (&T, Common) ;
                ERROR (CHOLMOD_INVALID, "premature EOF") ;
                return NULL ;
            }
            if (rd_blank (buf)) continue ;
            nt = sscanf (buf, "%lg %lg %lg %lg", &a, &b, &x, &z) ;
            break ;
        }
        if (nt == EOF) nt = 0 ;
        if (k == 0)
        {
            if (nt < 2 || nt > 4) { ERROR (CHOLMOD_INVALID, "invalid format") ; return NULL ; }
            ntok = nt ;
            xtype = nt == 2 ? CHOLMOD_PATTERN : nt == 3 ? CHOLMOD_REAL : CHOLMOD_COMPLEX ;
            T = cholmod_l_allocate_triplet (nrow, ncol, nnz + extra, stype == RD_UNKNOWN ? 0 : stype,
                xtype == CHOLMOD_PATTERN ? CHOLMOD_REAL : xtype, Common) ;
            if (!T) return NULL ;
            Ti = T->i ; Tj = T->j ; Tx = T->x ;
        }
        if (nt != ntok || a < 0 || b < 0 || a > 9.2e18 || b > 9.2e18)
        {
            cholmod_l_free_triplet (&T, Common) ;
            ERROR (CHOLMOD_INVALID, "invalid matrix file") ;
            return NULL ;
        }
        Int i = (Int) a, j = (Int) b ;
        Ti [k] = i ; Tj [k] = j ;
        if (i < j) lower_only = FALSE ;
        if (i > j) upper_only = FALSE ;
        if (xtype == CHOLMOD_REAL) Tx [k] = rd_value (x) ;
        else if (xtype == CHOLMOD_COMPLEX) { Tx [2*k] = rd_value (x) ; Tx [2*k+1] = rd_value (z) ; }
        if (i == 0 || j == 0) one_based = FALSE ;
        if (i > imax) imax = i ;
        if (j > jmax) jmax = j ;
    }
    if (one_based) for (size_t k = 0 ; k < nnz ; k++) { Ti [k]-- ; Tj [k]-- ; }
    if (one_based ? (imax > (Int) nrow || jmax > (Int) ncol) : (imax >= (Int) nrow || jmax >= (Int) ncol))
    {
        cholmod_l_free_triplet (&T, Common) ;
        ERROR (CHOLMOD_INVALID, "indices out of range") ;
        return NULL ;
    }
    if (unknown)
    {
        /* only one triangle present: symmetric with that triangle stored (a diagonal matrix: upper); else unsymmetric */
        if (lower_only && !upper_only) { stype = RD_LOWER ; }
        else if (upper_only) { stype = RD_UPPER ; }
        else { stype = RD_GENERAL ; extra = 0 ; }
        if (prefer_unsym && stype != RD_GENERAL) stype = RD_GENERAL ; else if (stype != RD_GENERAL) extra = 0 ;
        if (stype == RD_GENERAL && !prefer_unsym) extra = 0 ;
    }
    if (extra > 0)
    {
        size_t p = nnz ;
        for (size_t k = 0 ; k < nnz ; k++)
        {
            if (Ti [k] == Tj [k]) continue ;
            Ti [p] = Tj [k] ; Tj [p] = Ti [k] ;
            if (xtype == CHOLMOD_REAL) Tx [p] = skew ? -Tx [k] : Tx [k] ;
            else if (xtype == CHOLMOD_COMPLEX)
            {
                Tx [2*p]   = skew ? -Tx [2*k] : Tx [2*k] ;
                Tx [2*p+1] = csym ? Tx [2*k+1] : -Tx [2*k+1] ;      /* skew: -(x); Hermitian: conj (x) */
            }
            p++ ;
        }
        nnz = p ;
    }
    T->nnz = nnz ;
    T->stype = stype ;
    if (xtype == CHOLMOD_PATTERN)
    {
        if (stype == RD_GENERAL || Common->prefer_binary)
        {
            for (size_t k = 0 ; k < nnz ; k++) Tx [k] = 1 ;
        }
        else
        {
            /* a symmetric pattern becomes positive definite: diagonal = 1 + degree, off-diagonals -1 (:819-857) */
            Int *deg = cholmod_l_calloc (nrow + 1, sizeof (Int), Common) ;
            if (!deg) { cholmod_l_free_triplet (&T, Common) ; return NULL ; }
            for (size_t k = 0 ; k < nnz ; k++)
                if ((stype < 0 && Ti [k] > Tj [k]) || (stype > 0 && Ti [k] < Tj [k])) { deg [Ti [k]]++ ; deg [Tj [k]]++ ; }
            for (size_t k = 0 ; k < nnz ; k++) Tx [k] = (Ti [k] == Tj [k]) ? (double) (1 + deg [Ti [k]]) : -1.0 ;
            cholmod_l_free (nrow + 1, sizeof (Int), deg, Common) ;
        }
    }
    return T ;
}

/* Matrix Market "array": column-major, one entry per line; symmetric / Hermitian / skew files hold the lower triangle
 * (skew: strictly lower) and come back full (cholmod_read.c:128-150, :880-1060). */
static cholmod_dense *rd_dense (FILE *f, char *buf, const rd_head *H, cholmod_common *Common)
{
    size_t nrow = (size_t) H->h [0], ncol = (size_t) H->h [1] ;
    int stype = H->stype ;
    if (nrow == 0 || ncol == 0) return cholmod_l_zeros (nrow, ncol, CHOLMOD_REAL, Common) ;
    cholmod_dense *X = NULL ;
    double *Xx = NULL ;
    int ntok = 0, xtype = CHOLMOD_REAL, firstent = TRUE ;
    for (size_t j = 0 ; j < ncol ; j++)
    {
        size_t i0 = stype == RD_GENERAL ? 0 : stype == RD_SKEW ? j + 1 : j ;
        for (size_t i = i0 ; i < nrow ; i++)
        {
            double x = 0, z = 0 ;
            int nt = 0 ;
            for ( ; ; )
            {
                if (!fgets (buf, RD_LINE, f))
                {
                    cholmod_l_free_dense (&X, Common) ;
                    ERROR (CHOLMOD_INVALID, "premature EOF") ;
                    return NULL ;
                }
                if (rd_blank (buf)) continue ;
                nt = sscanf (buf, "%lg %lg", &x, &z) ;
                break ;
            }
            if (nt == EOF) nt = 0 ;
            if (firstent)
            {
                if (nt < 1 || nt > 2) { ERROR (CHOLMOD_INVALID, "invalid matrix file") ; return NULL ; }
                ntok = nt ;
                xtype = nt == 2 ? CHOLMOD_COMPLEX : CHOLMOD_REAL ;
                X = cholmod_l_zeros (nrow, ncol, xtype, Common) ;
                if (!X) return NULL ;
                Xx = X->x ;
                firstent = FALSE ;
            }
            if (nt != ntok)
            {
                cholmod_l_free_dense (&X, Common) ;
                ERROR (CHOLMOD_INVALID, "invalid matrix file") ;
                return NULL ;
            }
            x = rd_value (x) ; z = rd_value (z) ;
            size_t p = i + j * nrow, q = j + i * nrow ;
            if (xtype == CHOLMOD_REAL)
            {
                Xx [p] = x ;
                if (p != q && stype == RD_LOWER) Xx [q] = x ;
                if (p != q && stype == RD_SKEW) Xx [q] = -x ;
            }
            else
            {
                Xx [2*p] = x ; Xx [2*p+1] = z ;
                if (p != q && stype == RD_LOWER) { Xx [2*q] = x ; Xx [2*q+1] = -z ; }      /* Hermitian */
                if (p != q && stype == RD_SKEW) { Xx [2*q] = -x ; Xx [2*q+1] = -z ; }
                if (p != q && stype == RD_CSYM) { Xx [2*q] = x ; Xx [2*q+1] = z ; }
            }
        }
    }
    return X ;
}

cholmod_triplet *cholmod_l_read_triplet (FILE *f, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (f, NULL) ;
    Common->status = CHOLMOD_OK ;
    char buf [RD_LINE + 2] ;
    rd_head H ;
    if (!rd_header (f, buf, &H) || H.array) { ERROR (CHOLMOD_INVALID, "invalid format") ; return NULL ; }
    return rd_triplets (f, buf, &H, FALSE, Common) ;
}

/* Symmetric inputs come back upper-stored when Common->prefer_upper (the default), as the
 * reference (cholmod_read.c:1135-1190). */
cholmod_sparse *cholmod_l_read_sparse (FILE *f, cholmod_common *Common)
{
    cholmod_triplet *T = cholmod_l_read_triplet (f, Common) ;
    if (!T) return NULL ;
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

cholmod_dense *cholmod_l_read_dense (FILE *f, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (f, NULL) ;
    Common->status = CHOLMOD_OK ;
    char buf [RD_LINE + 2] ;
    rd_head H ;
    if (!rd_header (f, buf, &H) || !H.array) { ERROR (CHOLMOD_INVALID, "invalid format") ; return NULL ; }
    return rd_dense (f, buf, &H, Common) ;
}

/* Either kind (cholmod_read.c:1236-1330): *mtype says what came back.  prefer: 0 as stored (triplet), 1 sparse with
 * both triangles of a symmetric file (stype 0), 2 sparse, symmetric files stored symmetric. */
void *cholmod_l_read_matrix (FILE *f, int prefer, int *mtype, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (f, NULL) ;
    RETURN_IF_NULL (mtype, NULL) ;
    Common->status = CHOLMOD_OK ;
    char buf [RD_LINE + 2] ;
    rd_head H ;
    if (!rd_header (f, buf, &H)) { ERROR (CHOLMOD_INVALID, "invalid format") ; return NULL ; }
    if (H.array)
    {
        *mtype = CHOLMOD_DENSE ;
        return rd_dense (f, buf, &H, Common) ;
    }
    cholmod_triplet *T = rd_triplets (f, buf, &H, prefer == 1, Common) ;
    *mtype = CHOLMOD_TRIPLET ;
    if (prefer == 0 || !T) return T ;
    cholmod_sparse *A = cholmod_l_triplet_to_sparse (T, 0, Common) ;
    cholmod_l_free_triplet (&T, Common) ;
    if (A && prefer == 2 && A->stype < 0)
    {
        cholmod_sparse *A2 = cholmod_l_ptranspose (A, 2, NULL, NULL, 0, Common) ;
        cholmod_l_free_sparse (&A, Common) ;
        A = A2 ;
    }
    *mtype = CHOLMOD_SPARSE ;
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

/* reference: Check/cholmod_check.c:652-960 (check_sparse), the conditions only -- nothing is printed here */
int cholmod_l_check_sparse (cholmod_sparse *A, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    Common->status = CHOLMOD_OK ;
#define BAD(msg) { ERROR (CHOLMOD_INVALID, msg) ; free (Wi) ; return FALSE ; }
    Int *Wi = NULL ;
    if (!A) BAD ("null") ;
    const Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    const Int nrow = (Int) A->nrow, ncol = (Int) A->ncol, nzmax = (Int) A->nzmax ;
    if (A->itype != CHOLMOD_LONG) BAD ("integer type must match routine") ;
    if (A->xtype < CHOLMOD_PATTERN || A->xtype > CHOLMOD_ZOMPLEX) BAD ("unknown xtype") ;
    if (A->dtype != CHOLMOD_DOUBLE) BAD ("real type must match routine") ;
    if (A->stype != 0 && nrow != ncol) BAD ("symmetric but not square") ;
    if (!Ap) BAD ("p array not present") ;
    if (!Ai) BAD ("i array not present") ;
    if (!A->packed && !Anz) BAD ("nz array not present") ;
    if (A->xtype != CHOLMOD_PATTERN && !A->x) BAD ("x array not present") ;
    if (A->xtype == CHOLMOD_ZOMPLEX && !A->z) BAD ("z array not present") ;
    if (A->packed && Ap [0] != 0) BAD ("p [0] must be zero") ;
    if (A->packed && (Ap [ncol] < Ap [0] || Ap [ncol] > nzmax)) BAD ("p [ncol] invalid") ;
    if (!A->sorted && nrow > 0)
    {
        // (unsorted columns: duplicates are found with a mark per row, as the reference does with Common->Iwork)
        Wi = malloc ((size_t) nrow * sizeof (Int)) ;
        if (!Wi) { ERROR (CHOLMOD_OUT_OF_MEMORY, "out of memory") ; return FALSE ; }
        for (Int i = 0 ; i < nrow ; i++) Wi [i] = EMPTY ;
    }
    for (Int j = 0 ; j < ncol ; j++)
    {
        Int p = Ap [j], pend, nz ;
        if (A->packed) { pend = Ap [j+1] ; nz = pend - p ; }
        else { nz = Anz [j] < 0 ? 0 : Anz [j] ; pend = p + nz ; }       // (Anz [j] < 0 is treated as zero)
        if (p < 0 || pend > nzmax) BAD ("pointer invalid") ;
        if (nz < 0 || nz > nrow) BAD ("nz invalid") ;
        Int ilast = EMPTY ;
        for ( ; p < pend ; p++)
        {
            Int i = Ai [p] ;
            if (i < 0 || i >= nrow) BAD ("row index out of range") ;
            if (A->sorted && i <= ilast) BAD ("row indices out of order") ;
            if (!A->sorted && Wi [i] == j) BAD ("duplicate row index") ;
            ilast = i ;
            if (!A->sorted) Wi [i] = j ;
        }
    }
    free (Wi) ;
#undef BAD
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
