/* complex.c -- complex and zomplex input of cholmod_l_super_numeric and its callers.
 *
 * Reference: Supernodal/t_cholmod_super_numeric.c:41-83 (the COMPLEX and ZOMPLEX
 * templates: "A and F are complex or zomplex, L and C are complex"), called from
 * Supernodal/cholmod_super_numeric.c:286-300; same index maps, zherk / zgemm /
 * zpotrf / ztrsm instead of the real BLAS.
 *
 * The gfx950 matrix cores multiply real fp64 tiles (v_mfma_f64_16x16x4), so the
 * complex arithmetic is carried by the real embedding
 *
 *      phi (a + i b) = [ a  -b ]        rows / columns 2k, 2k+1 <-> re, im of k
 *                      [ b   a ]
 *
 * phi is a ring homomorphism with phi (Z^H) = phi (Z)', it maps a lower triangular
 * matrix with a real positive diagonal onto a lower triangular matrix with a positive
 * diagonal, and the Cholesky factor is unique: for Hermitian positive definite A,
 * chol (phi (A)) = phi (chol (A)).  So the complex factorization IS the real
 * factorization of the 2n x 2n matrix phi (S), S = tril (P A P^H), on a supernodal
 * structure with every supernode doubled (column k -> 2k, 2k+1; row r -> 2r, 2r+1),
 * and an interleaved complex column k of L is literally the EVEN column 2k of the
 * real factor (rows 2r, 2r+1 = re, im of L(r,k)).  A first failing pivot of the
 * complex matrix is the first failing pivot of phi (A) (column 2 minor), the
 * triangular solves with L and L^H are the real solves with phi (L) and phi (L)'
 * on the interleaved right-hand side (the same bytes).
 *
 * That is what this file does: it keeps, beside the complex cholmod_factor L (the
 * reference's maps, bit-exact, L->x complex interleaved), a real "twin" factor of
 * the doubled structure (L->cx_twin) and hands the twin to the real path -- the HIP
 * engine with all of its kernels (thin fronts, matrix-core updates, device solves,
 * several GPUs) when Common->useGPU == 1, the CPU path otherwise.  Cost: the
 * embedding holds every complex product twice, i.e. it executes 2x the real flops
 * (8/3 n^3 against 4/3 n^3 for a dense block) and 2x the bytes of a native
 * zherk/zgemm tile kernel; see DESIGN.md section 7e. */
#include "host_internal.h"

/* Which form the twin takes on the engine: 2 = complex storage (the engine keeps only the even
 * columns of every front, i.e. the interleaved complex factor itself, 2 xsize doubles:
 * CHOLMOD_HIP_CX_STORAGE), 1 = the full twin (4 xsize doubles).  The CPU path and several ranks
 * use the full twin; CHOLMOD_HIP_CX_TWIN=1 (or CHOLMOD_HIP_TWIN_FULL_K=1) asks for it on one GPU. */
static int twin_kind_wanted (cholmod_common *Common)
{
    /* (a silent degradation to the CPU path, if asked for, needs the full twin as well) */
    if (ssamd_resolve_use_gpu (Common) != 1 || Common->hip_world > 1 || Common->hip_cpu_fallback) return 1 ;
    const char *e = getenv ("CHOLMOD_HIP_CX_TWIN"), *f = getenv ("CHOLMOD_HIP_TWIN_FULL_K") ;
    if ((e && atoi (e) != 0) || (f && atoi (f) != 0)) return 1 ;
    return 2 ;
}

/* the twin of a complex factor: real supernodal symbolic factor of the doubled structure */
cholmod_factor *ssamd_complex_twin (cholmod_factor *L, cholmod_common *Common)
{
    const int kind = twin_kind_wanted (Common) ;
    if (L->cx_twin)
    {
        cholmod_factor *T0 = (cholmod_factor *) L->cx_twin ;
        /* the other form is wanted (the caller switched between the engine and the CPU path, or between
         * one rank and several): the twin is rebuilt -- its values are about to be recomputed anyway */
        if (T0->hip_is_twin == kind) return T0 ;
        cholmod_l_free_factor ((cholmod_factor **) &L->cx_twin, Common) ;
        L->hip_on_device = FALSE ;
    }
    /* a plan built by the analysis of a REAL matrix of this pattern and never used: the twin brings its own */
    if (L->hip_plan && L->hip_plan_ahead && !L->hip_on_device)
    {
        cholmod_hip_plan_destroy ((cholmod_hip_plan *) L->hip_plan) ;
        L->hip_plan = NULL ; L->hip_plan_ahead = 0 ;
    }
    size_t n = L->n, nsuper = L->nsuper ;
    cholmod_factor *T = cholmod_l_calloc (1, sizeof (cholmod_factor), Common) ;
    if (!T) return NULL ;
    T->n = 2 * n ; T->minor = 2 * n ;
    T->nsuper = nsuper ; T->ssize = 2 * L->ssize ; T->xsize = (kind == 2 ? 2 : 4) * L->xsize ;
    T->maxcsize = 4 * L->maxcsize ; T->maxesize = 2 * L->maxesize ;
    T->ordering = CHOLMOD_NATURAL ; T->is_ll = TRUE ; T->is_super = TRUE ; T->is_monotonic = TRUE ;
    T->itype = CHOLMOD_LONG ; T->xtype = CHOLMOD_PATTERN ; T->dtype = CHOLMOD_DOUBLE ;
    T->useGPU = L->useGPU ;
    T->Perm = cholmod_l_malloc (2 * n, sizeof (Int), Common) ;
    T->ColCount = cholmod_l_malloc (2 * n, sizeof (Int), Common) ;
    T->super = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    T->pi = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    T->px = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    T->s = cholmod_l_malloc (T->ssize, sizeof (Int), Common) ;
    if (!T->Perm || !T->ColCount || !T->super || !T->pi || !T->px || !T->s)
    {
        cholmod_l_free_factor (&T, Common) ;
        return NULL ;
    }
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s, *Cc = L->ColCount ;
    Int *TP = T->Perm, *TC = T->ColCount, *Tsuper = T->super, *Tpi = T->pi, *Tpx = T->px, *Ts = T->s ;
    for (size_t k = 0 ; k < n ; k++)
    {
        TP [2*k] = (Int) (2*k) ; TP [2*k+1] = (Int) (2*k+1) ;
        TC [2*k] = 2 * Cc [k] ; TC [2*k+1] = 2 * Cc [k] - 1 ;
    }
    for (size_t s = 0 ; s <= nsuper ; s++)
    {
        Tsuper [s] = 2 * Super [s] ; Tpi [s] = 2 * Lpi [s] ;
        Tpx [s] = (Lpx [0] == 123456) ? Lpx [s] : (kind == 2 ? 2 : 4) * Lpx [s] ;
    }
    for (size_t p = 0 ; p < L->ssize ; p++) { Ts [2*p] = 2 * Ls [p] ; Ts [2*p+1] = 2 * Ls [p] + 1 ; }
    T->hip_is_twin = kind ;
    L->cx_twin = T ;
    return T ;
}

/* phi (S): S is the lower-stored (stype < 0) complex or zomplex matrix handed to
 * cholmod_l_super_numeric; entries in the ignored triangle are skipped, imaginary
 * parts of the diagonal are ignored (zpotrf reads the real part only). */
static cholmod_sparse *embed_lower (cholmod_sparse *A, cholmod_common *Common)
{
    Int n = (Int) A->nrow ;
    const Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    const double *Ax = A->x, *Az = A->z ;
    const int zomplex = (A->xtype == CHOLMOD_ZOMPLEX) ;
    const int nth = ssamd_host_threads () ;
    Int *cnt = cholmod_l_malloc (n + 1, sizeof (Int), Common) ;
    if (!cnt) return NULL ;
#pragma omp parallel for schedule(static) num_threads(nth)
    for (Int j = 0 ; j < n ; j++)
    {
        Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j], c = 0 ;
        for ( ; p < pend ; p++) { Int i = Ai [p] ; if (i >= j && i < n) c += (i == j) ? 1 : 2 ; }
        cnt [j] = c ;           /* entries of column 2j; column 2j+1 has as many */
    }
    Int nz = 0 ;
    for (Int j = 0 ; j < n ; j++) { Int c = cnt [j] ; cnt [j] = nz ; nz += 2 * c ; }
    cnt [n] = nz ;
    cholmod_sparse *S = cholmod_l_allocate_sparse (2 * n, 2 * n, nz, A->sorted, TRUE, -1, CHOLMOD_REAL, Common) ;
    if (S)
    {
        Int *Sp = S->p, *Si = S->i ;
        double *Sx = S->x ;
#pragma omp parallel for schedule(static) num_threads(nth)
        for (Int j = 0 ; j < n ; j++)
        {
            Int q0 = cnt [j], q1 = q0 + (cnt [j+1] - cnt [j]) / 2 ;
            Sp [2*j] = q0 ; Sp [2*j+1] = q1 ;
            Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j] ;
            for ( ; p < pend ; p++)
            {
                Int i = Ai [p] ;
                if (i < j || i >= n) continue ;
                double re = zomplex ? Ax [p] : Ax [2*p], im = zomplex ? Az [p] : Ax [2*p+1] ;
                if (i == j)
                {
                    Si [q0] = 2*j ; Sx [q0++] = re ;
                    Si [q1] = 2*j+1 ; Sx [q1++] = re ;
                }
                else
                {
                    Si [q0] = 2*i ; Sx [q0++] = re ; Si [q0] = 2*i+1 ; Sx [q0++] = im ;
                    Si [q1] = 2*i ; Sx [q1++] = -im ; Si [q1] = 2*i+1 ; Sx [q1++] = re ;
                }
            }
        }
        Sp [2*n] = nz ;
    }
    cholmod_l_free (n + 1, sizeof (Int), cnt, Common) ;
    return S ;
}

/* L->x (complex, interleaved) = the even columns of the twin's host values */
static int gather_even_columns (cholmod_factor *L, cholmod_factor *T, cholmod_common *Common)
{
    if (!L->x) L->x = cholmod_l_malloc (L->xsize, 2 * sizeof (double), Common) ;
    if (!L->x) return FALSE ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px ;
    double *Lx = L->x ;
    const double *Tx = T->x ;
    const int nth = ssamd_host_threads () ;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nth)
    for (Int s = 0 ; s < (Int) L->nsuper ; s++)
    {
        Int nscol = Super [s+1] - Super [s], nsrow = Lpi [s+1] - Lpi [s] ;
        for (Int j = 0 ; j < nscol ; j++)
        {
            memcpy (Lx + 2 * (Lpx [s] + j * nsrow), Tx + 4 * Lpx [s] + (2*j) * (2*nsrow),
                (size_t) (2 * nsrow) * sizeof (double)) ;
            /* entry (2j+1, 2j) of the real factor is zero in exact arithmetic and a
             * rounding residue here; zpotrf leaves an exactly real diagonal */
            Lx [2 * (Lpx [s] + j * nsrow + j) + 1] = 0.0 ;
        }
    }
    return TRUE ;
}

/* bring the complex L->x up to date with the twin (downloads the twin from HBM if
 * its values live there only) */
int ssamd_complex_sync_host (cholmod_factor *L, cholmod_common *Common)
{
    cholmod_factor *T = (cholmod_factor *) L->cx_twin ;
    if (!T) { ERROR (CHOLMOD_INVALID, "no numeric factor") ; return FALSE ; }
    if (L->hip_host_valid) return TRUE ;
    if (T->hip_plan && T->hip_on_device && !T->hip_host_valid)
    {
        /* gathered on the device: half the bytes cross PCIe, no host copy of the twin */
        if (!L->x) L->x = cholmod_l_malloc (L->xsize, 2 * sizeof (double), Common) ;
        if (!L->x) return FALSE ;
        if (cholmod_hip_download_even_columns ((cholmod_hip_plan *) T->hip_plan, L->x) == CHOLMOD_HIP_OK)
        {
            L->hip_host_valid = TRUE ;
            return TRUE ;
        }
    }
    if (T->hip_is_twin == 2) { ERROR (CHOLMOD_GPU_PROBLEM, "complex factor: download from the engine failed") ; return FALSE ; }
    if (!cholmod_l_factor_to_host (T, Common)) return FALSE ;
    if (!gather_even_columns (L, T, Common)) return FALSE ;
    L->hip_host_valid = TRUE ;
    return TRUE ;
}

int ssamd_complex_super_numeric (cholmod_sparse *A, double beta, cholmod_factor *L, cholmod_common *Common)
{
    cholmod_factor *T = ssamd_complex_twin (L, Common) ;
    if (!T) return FALSE ;
    T->useGPU = L->useGPU ;
    cholmod_sparse *S2 = embed_lower (A, Common) ;
    if (!S2) return FALSE ;
    double b [2] = {beta, 0} ;
    /* on the engine the twin stays in HBM (the complex L->x is gathered there, below);
     * on the CPU path its host values are the stepping stone to the complex L->x */
    const int keep = Common->hip_factor_on_device ;
    Common->hip_factor_on_device = TRUE ;
    int ok = cholmod_l_super_numeric (S2, NULL, b, T, Common) ;
    Common->hip_factor_on_device = keep ;
    cholmod_l_free_sparse (&S2, Common) ;
    if (!ok || T->xtype != CHOLMOD_REAL)
    {
        if (L->x) { cholmod_l_free (L->xsize, 2 * sizeof (double), L->x, Common) ; L->x = NULL ; }
        L->xtype = CHOLMOD_PATTERN ;
        return FALSE ;
    }
    int status = Common->status ;           /* CHOLMOD_OK or CHOLMOD_NOT_POSDEF */
    L->xtype = CHOLMOD_COMPLEX ; L->dtype = CHOLMOD_DOUBLE ; L->is_ll = TRUE ;
    L->minor = (T->minor >= T->n) ? L->n : T->minor / 2 ;
    L->useGPU = T->useGPU ;
    L->hip_on_device = T->hip_on_device ;
    L->hip_host_valid = FALSE ;
    if (T->hip_on_device && !keep)
    {
        if (Common->hip_world > 1 && !cholmod_l_gather_factor (T, Common)) return FALSE ;
        if (!ssamd_complex_sync_host (L, Common)) return FALSE ;
    }
    else if (T->x && (T->hip_host_valid || !T->hip_on_device))
    {
        if (!gather_even_columns (L, T, Common)) return FALSE ;
        L->hip_host_valid = TRUE ;
        if (T->hip_on_device)
        {
            /* the engine keeps the factor in HBM for the solves: drop the twin's host copy */
            cholmod_l_free (T->xsize, sizeof (double), T->x, Common) ;
            T->x = NULL ; T->hip_host_valid = FALSE ;
        }
    }
    Common->status = status ;
    return TRUE ;
}
