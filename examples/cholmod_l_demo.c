/* cholmod_l_demo.c -- the sequence of the reference's demo driver (CHOLMOD/Demo/cholmod_l_demo.c:52-733, BASELINE.json
 * configs[0]) on this library, written against include/cholmod.h:
 *
 *   read a matrix (stdin or a file); unsymmetric: A*A'+beta*I is factorized   (:133-178 of the reference driver)
 *   norms of A, the right-hand side b(i) = 1 + i/n       (:180-243)
 *   cholmod_l_analyze, timed                             (:249-277)
 *   cholmod_l_factorize, timed                           (:279-297)
 *   integers / doubles held by L, cholmod_l_rcond        (:300-331)
 *   method 0: one cholmod_l_solve                        (:355-361)
 *   method 1: NTRIALS x cholmod_l_solve, b tweaked       (:362-374)
 *   method 2: NTRIALS x cholmod_l_solve2, workspace kept (:375-390)
 *   (method 3, solves with a sparse Bset, needs the simplicial factor form: not built here, reported as skipped)
 *   the residual |Ax-b|_inf / (|A|_inf |x|_inf + |b|_inf) of every method   (:585-594)
 *   one step of iterative refinement and its residual    (:605-631)
 *   the summary block: ordering, flops, times and rates, residuals, rcond   (:637-716)
 *   cholmod_l_gpu_stats                                  (:718)
 *
 *   gcc -O2 -I include examples/cholmod_l_demo.c -L suitesparse_amd/lib -lcholmod_amd \
 *       -Wl,-rpath,$PWD/suitesparse_amd/lib -lm -o cholmod_l_demo
 *   ./cholmod_l_demo [-cpu] [-perm perm.txt] [matrix-file] < tests/golden/bcsstk01.tri
 *
 * -cpu selects Common->useGPU = 0 (the CPU supernodal path: "plumbing, no GPU"); the default is the HIP engine.  -perm
 * names a file with n integers, the fill-reducing permutation (CHOLMOD_GIVEN); without it the library's default
 * strategy orders the matrix.  Any CHOLMOD error ends the program through the error handler, as in the reference demo. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "cholmod.h"

#define NTRIALS 100

static double wall (void)
{
    struct timespec ts ;
    clock_gettime (CLOCK_MONOTONIC, &ts) ;
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec ;
}

/* warnings (status > 0: not positive definite) are reported and the demo goes on; errors stop it */
static void demo_handler (int status, const char *file, int line, const char *message)
{
    printf ("cholmod %s: file: %s line: %d status: %d: %s\n", status < 0 ? "error" : "warning", file, line, status, message) ;
    if (status < 0) exit (1) ;
}

static const char *ordering_name (int ordering)
{
    switch (ordering)
    {
        case CHOLMOD_NATURAL: return "natural" ;
        case CHOLMOD_GIVEN: return "user" ;
        case CHOLMOD_AMD: return "AMD" ;
        case CHOLMOD_METIS: return "METIS" ;
        case CHOLMOD_NESDIS: return "NESDIS (built-in nested dissection)" ;
        case CHOLMOD_POSTORDERED: return "postordered" ;
        default: return "?" ;
    }
}

/* |b - A x|_inf / (|A|_inf |x|_inf + |b|_inf); *Rout (if not NULL) receives R = b - A x */
static double residual (cholmod_sparse *A, cholmod_dense *X, cholmod_dense *B, double anorm, double bnorm,
    cholmod_dense **Rout, cholmod_common *cm)
{
    double one [2] = {1, 0}, minusone [2] = {-1, 0}, zero [2] = {0, 0} ;
    cholmod_dense *R = cholmod_l_copy_dense (B, cm) ;
    if (A->stype == 0)
    {
        /* (A*A' + beta*I) x = b was solved (:537-575): W = A' x, R = b - beta x - A W */
        cholmod_dense *W = cholmod_l_allocate_dense (A->ncol, 1, A->ncol, CHOLMOD_REAL, cm) ;
        cholmod_l_sdmult (A, 1, one, zero, X, W, cm) ;
        double *Rx = R->x, *Xx = X->x ;
        for (size_t i = 0 ; i < A->nrow ; i++) Rx [i] -= 1e-6 * Xx [i] ;
        cholmod_l_sdmult (A, 0, minusone, one, W, R, cm) ;
        cholmod_l_free_dense (&W, cm) ;
    }
    else cholmod_l_sdmult (A, 0, minusone, one, X, R, cm) ;
    double rnorm = cholmod_l_norm_dense (R, 0, cm), xnorm = cholmod_l_norm_dense (X, 0, cm) ;
    double scale = anorm * xnorm + bnorm + ((A->nrow == 0) ? 1 : 0) ;
    if (Rout) *Rout = R ; else cholmod_l_free_dense (&R, cm) ;
    return rnorm / scale ;
}

int main (int argc, char **argv)
{
    int use_gpu = 1 ;
    const char *perm_file = NULL, *matrix_file = NULL ;
    for (int a = 1 ; a < argc ; a++)
    {
        if (!strcmp (argv [a], "-cpu")) use_gpu = 0 ;
        else if (!strcmp (argv [a], "-perm") && a + 1 < argc) perm_file = argv [++a] ;
        else matrix_file = argv [a] ;
    }
    FILE *f = stdin ;
    if (matrix_file && !(f = fopen (matrix_file, "r"))) { printf ("unable to open %s\n", matrix_file) ; return 1 ; }

    /* ---- start, parameters */
    cholmod_common Common, *cm = &Common ;
    cholmod_l_start (cm) ;
    cm->error_handler = demo_handler ;
    cm->supernodal = CHOLMOD_SUPERNODAL ;           /* the supernodal LL' path is what this library builds */
    cm->useGPU = use_gpu ;
    printf ("---------------------------------- cholmod_l_demo (%s):\n", use_gpu ? "HIP engine" : "CPU supernodal path") ;

    /* ---- the matrix */
    cholmod_sparse *A = cholmod_l_read_sparse (f, cm) ;
    if (matrix_file) fclose (f) ;
    if (!A) { printf ("no matrix read\n") ; return 1 ; }
    /* an unsymmetric matrix: A*A' + beta*I is factorized, beta = 1e-6 (:124-127, :280-286 of the reference driver) */
    const int aat = (A->stype == 0) ;
    double beta [2] = {1e-6, 0} ;
    const size_t n = A->nrow ;
    const int xtype = A->xtype ;
    double anorm = cholmod_l_norm_sparse (A, 0, cm) ;
    printf ("A: %zu-by-%zu, nnz stored %ld, stype %d, %s\n", n, A->ncol, (long) cholmod_l_nnz (A, cm), A->stype,
        xtype == CHOLMOD_REAL ? "real" : "complex") ;
    printf ("norm (A,inf) = %g\nnorm (A,1)   = %g\n", anorm, cholmod_l_norm_sparse (A, 1, cm)) ;
    SuiteSparse_long *perm = NULL ;
    if (perm_file)
    {
        FILE *pf = fopen (perm_file, "r") ;
        if (!pf) { printf ("unable to open %s\n", perm_file) ; return 1 ; }
        perm = malloc ((n ? n : 1) * sizeof (SuiteSparse_long)) ;
        for (size_t k = 0 ; k < n ; k++) { long v ; if (fscanf (pf, "%ld", &v) != 1) { printf ("short permutation\n") ; return 1 ; } perm [k] = v ; }
        fclose (pf) ;
    }

    /* ---- right-hand side: b(i) = 1 + i/n (:231-239) */
    cholmod_dense *B = cholmod_l_zeros (n, 1, xtype, cm) ;
    double *Bx = B->x ;
    for (size_t i = 0 ; i < n ; i++)
    {
        if (xtype == CHOLMOD_REAL) Bx [i] = 1 + i / (double) n ;
        else { Bx [2*i] = 1 + i / (double) n ; Bx [2*i+1] = ((double) n / 2 - (double) i) / (3 * (double) n) ; }
    }
    double bnorm = cholmod_l_norm_dense (B, 0, cm) ;
    printf ("bnorm %g\n", bnorm) ;

    /* ---- analyze */
    double t = wall () ;
    cholmod_factor *L = perm ? cholmod_l_analyze_p (A, perm, NULL, 0, cm) : cholmod_l_analyze (A, cm) ;
    double ta = wall () - t ;
    printf ("Analyze: flop %g lnz %g\n", cm->fl, cm->lnz) ;
    printf ("L: supernodal symbolic, %zu supernodes, ssize %zu, xsize %zu, maxcsize %zu, maxesize %zu, ordering %d, useGPU %d\n",
        L->nsuper, L->ssize, L->xsize, L->maxcsize, L->maxesize, L->ordering, L->useGPU) ;

    /* ---- factorize */
    printf (aat ? "Factorizing A*A'+beta*I\n" : "Factorizing A\n") ;
    t = wall () ;
    if (aat) cholmod_l_factorize_p (A, beta, NULL, 0, L, cm) ; else cholmod_l_factorize (A, L, cm) ;
    double tf = wall () - t ;
    printf ("L: supernodal numeric LL', minor %zu, status %d\n", L->minor, cm->status) ;
    /* integers and doubles of the supernodal L (:300-315) */
    double isize = (double) n + (double) n + 3.0 * (double) (L->nsuper + 1) + (double) L->ssize ;
    double xsize = (double) L->xsize ;
    double rcond = cholmod_l_rcond (L, cm) ;

    /* ---- solve, three ways */
    const int nmethods = 2 ;
    double ts [3] = {0, 0, 0}, resid [4] = {-1, -1, -1, -1} ;
    cholmod_dense *X = NULL ;
    for (int method = 0 ; method <= nmethods ; method++)
    {
        const double x = (double) n ;
        if (method == 0)
        {
            t = wall () ;
            X = cholmod_l_solve (CHOLMOD_A, L, B, cm) ;
            ts [0] = wall () - t ;
        }
        else if (method == 1)
        {
            /* many solves, b tweaked every time; the last one is kept */
            t = wall () ;
            for (int trial = 0 ; trial < NTRIALS ; trial++)
            {
                cholmod_l_free_dense (&X, cm) ;
                Bx [0] = 1 + trial / x ;
                X = cholmod_l_solve (CHOLMOD_A, L, B, cm) ;
            }
            ts [1] = (wall () - t) / NTRIALS ;
        }
        else
        {
            /* the same with the solution and the workspaces reused from call to call */
            cholmod_dense *Ywork = NULL, *Ework = NULL ;
            cholmod_l_free_dense (&X, cm) ;
            t = wall () ;
            for (int trial = 0 ; trial < NTRIALS ; trial++)
            {
                Bx [0] = 1 + trial / x ;
                cholmod_l_solve2 (CHOLMOD_A, L, B, NULL, &X, NULL, &Ywork, &Ework, cm) ;
            }
            ts [2] = (wall () - t) / NTRIALS ;
            cholmod_l_free_dense (&Ywork, cm) ;
            cholmod_l_free_dense (&Ework, cm) ;
        }
        resid [method] = residual (A, X, B, anorm, bnorm, NULL, cm) ;
    }
    printf ("method 3 (solve2 with a sparse Bset) skipped: it needs the simplicial form of L, which this library does not build\n") ;

    /* ---- one step of iterative refinement (real symmetric case, :605-631): X += A \\ (B - A X) */
    double resid2 = -1 ;
    if (xtype == CHOLMOD_REAL && !aat)
    {
        cholmod_dense *R = NULL ;
        (void) residual (A, X, B, anorm, bnorm, &R, cm) ;
        cholmod_dense *R2 = cholmod_l_solve (CHOLMOD_A, L, R, cm) ;
        double *Xx = X->x, *Rx = R2->x ;
        const double xnorm_before = cholmod_l_norm_dense (X, 0, cm) ;
        for (size_t i = 0 ; i < n ; i++) Xx [i] += Rx [i] ;
        cholmod_l_free_dense (&R2, cm) ;
        cholmod_l_free_dense (&R, cm) ;
        /* (the reference keeps the scale of the unrefined solution, :589 and :627) */
        double one [2] = {1, 0}, minusone [2] = {-1, 0} ;
        R = cholmod_l_copy_dense (B, cm) ;
        cholmod_l_sdmult (A, 0, minusone, one, X, R, cm) ;
        resid2 = cholmod_l_norm_dense (R, 0, cm) / (anorm * xnorm_before + bnorm + ((n == 0) ? 1 : 0)) ;
        cholmod_l_free_dense (&R, cm) ;
    }

    /* ---- results */
    for (int i = 0 ; i < CHOLMOD_MAXMETHODS ; i++)
    {
        double fl = cm->method [i].fl, xlnz = cm->method [i].lnz ;
        if (fl < 0) continue ;
        printf ("Ordering: %-8s", ordering_name (cm->method [i].ordering)) ;
        if (xlnz > 0) printf (" fl/lnz %10.1f", fl / xlnz) ;
        if (cm->anz > 0) printf ("  lnz/anz %10.1f", xlnz / cm->anz) ;
        printf ("\n") ;
    }
    const double tot = ta + tf + ts [0] ;
    printf ("ints in L: %15.0f, doubles in L: %15.0f\n", isize, xsize) ;
    printf ("factor flops %g nnz(L) %15.0f (w/no amalgamation)\n", cm->fl, cm->lnz) ;
    printf (aat ? "nnz(A*A'): %15.0f\n" : "nnz(A):    %15.0f\n", cm->anz) ;
    if (cm->lnz > 0) printf ("flops / nnz(L):  %8.1f\n", cm->fl / cm->lnz) ;
    if (cm->anz > 0) printf ("nnz(L) / nnz(A): %8.1f\n", cm->lnz / cm->anz) ;
    printf ("analyze walltime: %12.4f\n", ta) ;
    printf ("factor  walltime: %12.4f mflop: %8.1f\n", tf, tf > 0 ? 1e-6 * cm->fl / tf : 0) ;
    printf ("solve   walltime: %12.4f mflop: %8.1f\n", ts [0], ts [0] > 0 ? 1e-6 * 4 * cm->lnz / ts [0] : 0) ;
    printf ("overall walltime: %12.4f mflop: %8.1f\n", tot, tot > 0 ? 1e-6 * (cm->fl + 4 * cm->lnz) / tot : 0) ;
    printf ("solve   walltime: %12.4f mflop: %8.1f (%d trials)\n", ts [1], ts [1] > 0 ? 1e-6 * 4 * cm->lnz / ts [1] : 0, NTRIALS) ;
    printf ("solve2  walltime: %12.4f mflop: %8.1f (%d trials)\n", ts [2], ts [2] > 0 ? 1e-6 * 4 * cm->lnz / ts [2] : 0, NTRIALS) ;
    printf ("peak memory usage: %12.0f (MB)\n", (double) cm->memory_usage / 1048576.) ;
    printf ("residual (|Ax-b|/(|A||x|+|b|)): ") ;
    for (int method = 0 ; method <= nmethods ; method++) printf ("%8.2e ", resid [method]) ;
    printf ("\n") ;
    if (resid2 >= 0) printf ("residual %8.1e (|Ax-b|/(|A||x|+|b|)) after iterative refinement\n", resid2) ;
    printf ("rcond    %8.1e\n\n", rcond) ;
    cholmod_l_gpu_stats (cm) ;

    /* ---- free everything; nothing may be left */
    cholmod_l_free_factor (&L, cm) ;
    cholmod_l_free_dense (&X, cm) ;
    cholmod_l_free_sparse (&A, cm) ;
    cholmod_l_free_dense (&B, cm) ;
    cholmod_l_finish (cm) ;
    free (perm) ;
    printf ("malloc_count %zu memory_inuse %zu (both must be 0)\n", cm->malloc_count, cm->memory_inuse) ;
    int bad = 0 ;
    for (int method = 0 ; method <= nmethods ; method++) if (!(resid [method] >= 0 && resid [method] < 1e-9)) bad = 1 ;
    return (cm->malloc_count == 0 && cm->memory_inuse == 0 && !bad) ? 0 : 2 ;
}
