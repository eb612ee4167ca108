/* cholmod_l_demo.c -- the reference demo's flow (CHOLMOD/Demo/cholmod_l_demo.c:
 * 52-733) on this library: read a symmetric matrix from stdin, b(i) = 1+i/n,
 * analyze (supernodal forced, BASELINE.json config #1), factorize on the HIP
 * engine, solve, print the residual the reference prints
 * (|Ax-b|_inf / (|A|_inf |x|_inf + |b|_inf), :585-594) and the 2-norm form.
 *
 *   gcc -O2 -I include examples/cholmod_l_demo.c -L suitesparse_amd/lib \
 *       -lcholmod_amd -Wl,-rpath,$PWD/suitesparse_amd/lib -lm -o cholmod_l_demo
 *   ./cholmod_l_demo [perm.txt [cpu]] < tests/golden/bcsstk01.tri
 * ("cpu" selects Common->useGPU = 0: the CPU supernodal path, BASELINE.json configs[0])
 *
 * An optional file with n integers supplies the fill-reducing permutation
 * (the ordering packages are out of scope; default is the natural ordering
 * followed by the weighted postorder). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "cholmod.h"

int main (int argc, char **argv)
{
    cholmod_common Common, *cm = &Common ;
    cholmod_l_start (cm) ;
    cm->supernodal = CHOLMOD_SUPERNODAL ;        /* SURVEY.md finding 1 */
    cm->useGPU = (argc > 2 && !strcmp (argv [2], "cpu")) ? 0 : 1 ;    /* "cpu": BASELINE.json configs[0] */
    cholmod_sparse *A = cholmod_l_read_sparse (stdin, cm) ;
    if (!A) { printf ("read failed, status %d\n", cm->status) ; return 1 ; }
    if (A->stype == 0 || A->nrow != A->ncol) { printf ("matrix must be symmetric\n") ; return 1 ; }
    size_t n = A->nrow ;
    printf ("cholmod_l_demo: n %zu nnz %ld stype %d\n", n, (long) cholmod_l_nnz (A, cm), A->stype) ;
    SuiteSparse_long *perm = NULL ;
    if (argc > 1)
    {
        FILE *pf = fopen (argv [1], "r") ;
        perm = malloc ((n ? n : 1) * sizeof (SuiteSparse_long)) ;
        for (size_t k = 0 ; pf && k < n ; k++) { long v ; if (fscanf (pf, "%ld", &v) != 1) return 1 ; perm [k] = v ; }
        if (pf) fclose (pf) ;
    }
    cholmod_dense *B = cholmod_l_zeros (n, 1, CHOLMOD_REAL, cm) ;
    for (size_t i = 0 ; i < n ; i++) ((double *) B->x) [i] = 1 + i / (double) n ;
    cholmod_factor *L = perm ? cholmod_l_analyze_p (A, perm, NULL, 0, cm) : cholmod_l_analyze (A, cm) ;
    if (!L) { printf ("analyze failed, status %d\n", cm->status) ; return 1 ; }
    printf ("analyze: fl %g lnz %g nsuper %zu ssize %zu xsize %zu maxcsize %zu maxesize %zu ordering %d\n",
        cm->fl, cm->lnz, L->nsuper, L->ssize, L->xsize, L->maxcsize, L->maxesize, L->ordering) ;
    if (!cholmod_l_factorize (A, L, cm)) { printf ("factorize failed, status %d\n", cm->status) ; return 1 ; }
    printf ("factorize: status %d minor %zu device time %.6f s  %.2f GFLOP/s (fl/t)\n", cm->status, L->minor,
        cm->gpuKernelTime, cm->gpuKernelTime > 0 ? 1e-9 * cm->fl / cm->gpuKernelTime : 0.0) ;
    cholmod_dense *X = cholmod_l_solve (CHOLMOD_A, L, B, cm) ;
    if (!X) { printf ("solve failed, status %d\n", cm->status) ; return 1 ; }
    /* R = B - A*X */
    cholmod_dense *R = cholmod_l_copy_dense (B, cm) ;
    double one [2] = {1, 0}, minusone [2] = {-1, 0} ;
    cholmod_l_sdmult (A, 0, minusone, one, X, R, cm) ;
    double rnorm = cholmod_l_norm_dense (R, 0, cm), xnorm = cholmod_l_norm_dense (X, 0, cm) ;
    double bnorm = cholmod_l_norm_dense (B, 0, cm), anorm = cholmod_l_norm_sparse (A, 0, cm) ;
    double r2 = cholmod_l_norm_dense (R, 2, cm), b2 = cholmod_l_norm_dense (B, 2, cm) ;
    printf ("residual %8.1e (|Ax-b|/(|A||x|+|b|))   %8.1e (2-norm relative)\n",
        rnorm / (anorm * xnorm + bnorm), r2 / b2) ;
    cholmod_l_gpu_stats (cm) ;
    cholmod_l_free_dense (&R, cm) ; cholmod_l_free_dense (&X, cm) ; cholmod_l_free_dense (&B, cm) ;
    cholmod_l_free_factor (&L, cm) ; cholmod_l_free_sparse (&A, cm) ;
    cholmod_l_finish (cm) ;
    printf ("malloc_count %zu memory_inuse %zu (both must be 0)\n", cm->malloc_count, cm->memory_inuse) ;
    free (perm) ;
    return (cm->malloc_count == 0) ? 0 : 2 ;
}
