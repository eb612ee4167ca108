/* cpu_numeric.c -- the CPU supernodal numeric factorization and triangular
 * solves of the host layer: what runs when Common->useGPU == 0 (the reference's
 * default when CHOLMOD_USE_GPU is unset, Supernodal/cholmod_super_symbolic.c:
 * 286-291) and what a GPU request degrades to when no device can be used
 * (Supernodal/t_cholmod_super_numeric.c:183-192).
 *
 * Algorithm: the reference's left-looking supernodal loop
 * (t_cholmod_super_numeric.c:279-1048) -- assemble A into the supernode, pull one
 * dense update  C = L_d(rows >= s) * L_d(rows in s)'  per pending descendant d and
 * scatter it through the row map, factor the diagonal block, solve the rows
 * below -- with the pending descendants kept in per-supernode lists that are
 * re-linked as each descendant's row cursor moves up the elimination tree.
 * Dense arithmetic: an LP64 BLAS/LAPACK found at run time (CHOLMOD_BLAS_LIBRARY,
 * then the usual sonames), else the blocked C kernels below.
 *
 * Product code: nothing here uses oracle/ (which is test infrastructure). */
#include "host_internal.h"
#include <dlfcn.h>
#include <pthread.h>
#include <time.h>
#include <errno.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/wait.h>

/* ---- dense kernels --------------------------------------------------------------- */

typedef void (*dgemm_fn) (const char *, const char *, const int *, const int *, const int *, const double *,
    const double *, const int *, const double *, const int *, const double *, double *, const int *) ;
typedef void (*dsyrk_fn) (const char *, const char *, const int *, const int *, const double *, const double *,
    const int *, const double *, double *, const int *) ;
typedef void (*dtrsm_fn) (const char *, const char *, const char *, const char *, const int *, const int *,
    const double *, const double *, const int *, double *, const int *) ;
typedef void (*dpotrf_fn) (const char *, const int *, double *, const int *, int *) ;

typedef void (*set_threads_fn) (int) ;
static struct
{
    int tried ;
    void *handle ;
    dgemm_fn gemm ; dsyrk_fn syrk ; dtrsm_fn trsm ; dpotrf_fn potrf ;
    set_threads_fn set_threads ;    /* openblas_set_num_threads / MKL_Set_Num_Threads / bli_thread_set_num_threads, or NULL */
    int max_threads, cur_threads ;
    double grain ;                  /* flops of a dense call per BLAS thread */
    char name [256] ;
} g_blas ;

/* Threads for ONE dense call, by its flop count.  The left-looking loop issues its BLAS calls one after the other, and
 * almost all of them are tiny (Poisson 100^3 under AMD: 97 % of 601 338 updates have <= 16 columns): a threaded
 * BLAS that wakes 64 threads for each of them spends its time in the wake-up, not in the arithmetic -- rounds 3-4 measured
 * 36 GFLOP/s at 64 threads against 216 at 16.  So a call gets one thread per ~128 Mflop, at most what the caller allows
 * (OMP_NUM_THREADS): small calls run on the calling thread, the few big ones (which hold the flops) on all of them, and
 * more threads never cost.  Without a thread-control entry point in the bound library the BLAS decides for itself. */
static void blas_threads_for (double flops)
{
    if (!g_blas.set_threads) return ;
    /* (one thread per 128 Mflop -- a few milliseconds of dgemm per thread.  Poisson 100^3 on 2 x EPYC 9575F with scipy's
     * OpenBLAS, GFLOP/s at 16 / 32 / 64 threads: 542 / 286 / 149 with 4 Mflop per thread, 466 / 304 / 179 with 32, 480 / 330 /
     * 221 with 128; rounds 3-4, every call on all threads: 216 at 16, 36 at 64.  What is left of the drop beyond 16 threads
     * is the BLAS's own scaling on the few top fronts.) */
    const double grain = g_blas.grain ;          /* (CHOLMOD_CPU_MFLOP_PER_THREAD, read when the BLAS is bound) */
    int t = flops < grain ? 1 : (int) (flops / grain) + 1 ;
    if (t > g_blas.max_threads) t = g_blas.max_threads ;
    if (t != g_blas.cur_threads) { g_blas.set_threads (t) ; g_blas.cur_threads = t ; }
}

static void *sym2 (void *h, const char *prefix, const char *name)
{
    char buf [96] ;
    snprintf (buf, sizeof (buf), "%s%s", prefix, name) ;
    return dlsym (h, buf) ;
}

/* 3 x 3 dpotrf and a 3 x 2 dgemm with known answers; 1 = the library computes them */
static int lp64_probe (dgemm_fn g, dpotrf_fn p)
{
    double M [9] = {4, 2, 2,  0, 5, 3,  0, 0, 6} ;          /* lower triangle, column-major */
    int n3 [2] = {3, -1}, info [2] = {-1, -1} ;
    p ("L", n3, M, n3, info) ;
    double Aa [6] = {1, 2, 3,  4, 5, 6}, Bb [4] = {1, 0,  0, 1}, Cc [6] = {0, 0, 0, 0, 0, 0} ;
    int m3 [2] = {3, -1}, n2 [2] = {2, -1}, k2 [2] = {2, -1} ;
    double one = 1.0, zero = 0.0 ;
    g ("N", "N", m3, n2, k2, &one, Aa, m3, Bb, k2, &zero, Cc, m3) ;
    int okp = (info [0] == 0 && fabs (M [0] - 2.0) < 1e-14 && fabs (M [1] - 1.0) < 1e-14 && fabs (M [4] - 2.0) < 1e-14
        && fabs (M [5] - 1.0) < 1e-14 && fabs (M [8] - 2.0) < 1e-14) ;
    int okg = 1 ;
    for (int q = 0 ; q < 6 ; q++) if (Cc [q] != Aa [q]) okg = 0 ;
    return okp && okg ;
}

/* In this process, behind the negative second words (round-4 advisor item: no fork () inside a library -- the host process
 * may be multi-threaded with the HIP runtime, RCCL and OpenMP loaded, and a child that runs BLAS code after fork can
 * deadlock on a lock whose owner no longer exists).  An ILP64 library reads (n, -1) as a negative 64-bit dimension and
 * leaves through its argument check. */
static int lp64_self_check (dgemm_fn g, dpotrf_fn p)
{
    return lp64_probe (g, p) ;
}

static int try_blas (const char *path, const char *prefix)
{
    void *h = dlopen (path, RTLD_NOW | RTLD_LOCAL) ;
    if (!h) return 0 ;
    dgemm_fn g = (dgemm_fn) sym2 (h, prefix, "dgemm_") ;
    dsyrk_fn s = (dsyrk_fn) sym2 (h, prefix, "dsyrk_") ;
    dtrsm_fn t = (dtrsm_fn) sym2 (h, prefix, "dtrsm_") ;
    dpotrf_fn p = (dpotrf_fn) sym2 (h, prefix, "dpotrf_") ;
    if (!g || !s || !t || !p) { dlclose (h) ; return 0 ; }
    /* self-check before trusting it: the arguments are 32-bit ints (LP64 interface); an ILP64 build
     * found under the same soname reads 8 bytes from each.  Every size therefore sits in a two-word
     * array whose second word is -1: an ILP64 library sees a NEGATIVE dimension and leaves through
     * its argument check (info < 0 / xerbla) instead of running over the 9- and 6-element arrays
     * with whatever the stack held next to a lone int (lp64_self_check). */
    if (!lp64_self_check (g, p))
    {
        fprintf (stderr, "cholmod (CPU path): %s fails the LP64 self-check (an ILP64 build?): not used\n", path) ;
        dlclose (h) ;
        return 0 ;
    }
    g_blas.gemm = g ; g_blas.syrk = s ; g_blas.trsm = t ; g_blas.potrf = p ;
    /* per-call thread control, where the library offers it (blas_threads_for) */
    g_blas.set_threads = (set_threads_fn) sym2 (h, prefix, "openblas_set_num_threads") ;
    if (!g_blas.set_threads) g_blas.set_threads = (set_threads_fn) dlsym (h, "openblas_set_num_threads") ;
    if (!g_blas.set_threads) g_blas.set_threads = (set_threads_fn) dlsym (h, "MKL_Set_Num_Threads") ;
    if (!g_blas.set_threads) g_blas.set_threads = (set_threads_fn) dlsym (h, "bli_thread_set_num_threads") ;
    g_blas.max_threads = ssamd_host_threads_uncapped () ;
    { const char *ge = getenv ("CHOLMOD_CPU_MFLOP_PER_THREAD") ; g_blas.grain = (ge && atof (ge) > 0) ? 1e6 * atof (ge) : 128e6 ; }
    g_blas.cur_threads = -1 ;
    snprintf (g_blas.name, sizeof (g_blas.name), "%s%s%s", path, prefix [0] ? " prefix " : "", prefix) ;
    g_blas.handle = h ;         /* (last: everything above is in place when a reader sees the handle) */
    return 1 ;
}

/* CHOLMOD_BLAS_LIBRARY = path[:symbol-prefix] ; "none" forces the built-in kernels */
/* (bound exactly once per process, whatever the number of threads that enter the CPU path
 * at the same time with their own Common: pthread_once publishes the finished binding) */
static void bind_blas_impl (void)
{
    g_blas.tried = 1 ;
    const char *e = getenv ("CHOLMOD_BLAS_LIBRARY") ;
    if (e && e [0])
    {
        if (!strcmp (e, "none")) return ;
        char path [512] ;
        snprintf (path, sizeof (path), "%s", e) ;
        char *c = strchr (path, ':') ;
        const char *prefix = "" ;
        if (c) { *c = '\0' ; prefix = c + 1 ; }
        if (try_blas (path, prefix)) return ;
    }
    static const char *names [] = {"libopenblas.so.0", "libopenblas.so", "libmkl_rt.so", "libblis.so.4", "libblis.so",
        "libblas.so.3", NULL} ;
    for (int q = 0 ; names [q] ; q++)
    {
        /* (a reference BLAS without LAPACK fails the dpotrf_ lookup and is skipped) */
        if (try_blas (names [q], "")) return ;
    }
}

static pthread_once_t g_blas_once = PTHREAD_ONCE_INIT ;
static void bind_blas_once (void) { (void) pthread_once (&g_blas_once, bind_blas_impl) ; }

const char *ssamd_cpu_blas_name (void)
{
    bind_blas_once () ;
    return g_blas.handle ? g_blas.name : "built-in blocked C kernels" ;
}

/* C (m x n, ldc) -= A (m x k, lda) * B (n x k, ldb)' ; tri: only the entries i >= j
 * are touched (the strictly upper part of a supernode's diagonal block is dead
 * space that must keep its zeros, as with dsyrk "L") */
static void k_gemm_nt (Int m, Int n, Int k, const double *A, Int lda, const double *B, Int ldb,
    double *C, Int ldc, int tri)
{
    const Int JB = 32, IB = 128 ;
#pragma omp parallel for schedule(dynamic, 1) if (m * n * k > 200000)
    for (Int j0 = 0 ; j0 < n ; j0 += JB)
    {
        Int jn = (n - j0 < JB) ? n - j0 : JB ;
        for (Int i0 = tri ? j0 : 0 ; i0 < m ; i0 += IB)
        {
            Int in = (m - i0 < IB) ? m - i0 : IB ;
            int diag = tri && i0 < j0 + jn ;            /* this row block crosses the diagonal */
            for (Int j = 0 ; j < jn ; j += 2)
            {
                int two = (j + 1 < jn) && !diag ;
                if (diag)
                {
                    /* one column at a time, from its diagonal entry down */
                    for (Int jj = j ; jj < j + 2 && jj < jn ; jj++)
                    {
                        Int is = (j0 + jj > i0) ? j0 + jj - i0 : 0 ;
                        double *c0 = C + i0 + (j0 + jj) * ldc ;
                        const double *b0 = B + (j0 + jj) ;
                        for (Int p = 0 ; p < k ; p++)
                        {
                            double x0 = b0 [p * ldb] ;
                            const double *a = A + i0 + p * lda ;
                            for (Int i = is ; i < in ; i++) c0 [i] -= a [i] * x0 ;
                        }
                    }
                    continue ;
                }
                double *c0 = C + i0 + (j0 + j) * ldc, *c1 = c0 + (two ? ldc : 0) ;
                const double *b0 = B + (j0 + j), *b1 = b0 + (two ? 1 : 0) ;
                for (Int p = 0 ; p < k ; p++)
                {
                    double x0 = b0 [p * ldb], x1 = two ? b1 [p * ldb] : 0.0 ;
                    const double *a = A + i0 + p * lda ;
                    if (two) for (Int i = 0 ; i < in ; i++) { double v = a [i] ; c0 [i] -= v * x0 ; c1 [i] -= v * x1 ; }
                    else for (Int i = 0 ; i < in ; i++) c0 [i] -= a [i] * x0 ;
                }
            }
        }
    }
}

/* in-place lower Cholesky of the n x n block; returns LAPACK's info */
static Int k_potrf (Int n, double *A, Int lda)
{
    const Int NBK = 48 ;
    for (Int j0 = 0 ; j0 < n ; j0 += NBK)
    {
        Int jb = (n - j0 < NBK) ? n - j0 : NBK ;
        /* diagonal block, unblocked */
        for (Int j = j0 ; j < j0 + jb ; j++)
        {
            double d = A [j + j * lda] ;
            for (Int p = j0 ; p < j ; p++) d -= A [j + p * lda] * A [j + p * lda] ;
            if (d <= 0.0) return j + 1 ;                /* NaN does not trip, as dpotrf */
            d = sqrt (d) ;
            A [j + j * lda] = d ;
            for (Int i = j + 1 ; i < n ; i++)
            {
                double v = A [i + j * lda] ;
                for (Int p = j0 ; p < j ; p++) v -= A [i + p * lda] * A [j + p * lda] ;
                A [i + j * lda] = v / d ;
            }
        }
        /* trailing update with the finished block column */
        Int r = j0 + jb ;
        if (r < n) k_gemm_nt (n - r, n - r, jb, A + r + j0 * lda, lda, A + r + j0 * lda, lda, A + r + r * lda, lda, 1) ;
    }
    return 0 ;
}

/* B (m x n, ldb) := B * inv(L)' , L n x n lower (ldl) */
static void k_trsm (Int m, Int n, const double *L, Int ldl, double *B, Int ldb)
{
#pragma omp parallel for schedule(static) if (m * n * n > 200000)
    for (Int i0 = 0 ; i0 < m ; i0 += 64)
    {
        Int in = (m - i0 < 64) ? m - i0 : 64 ;
        for (Int j = 0 ; j < n ; j++)
        {
            double *bj = B + i0 + j * ldb ;
            for (Int p = 0 ; p < j ; p++)
            {
                double l = L [j + p * ldl] ;
                const double *bp = B + i0 + p * ldb ;
                for (Int i = 0 ; i < in ; i++) bj [i] -= bp [i] * l ;
            }
            double d = 1.0 / L [j + j * ldl] ;
            for (Int i = 0 ; i < in ; i++) bj [i] *= d ;
        }
    }
}

static double now_s (void)
{
    struct timespec ts ;
    clock_gettime (CLOCK_MONOTONIC, &ts) ;
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec ;
}

/* ---- numeric factorization --------------------------------------------------------- */

/* A: lower-stored permuted matrix (packed or not, sorted or not).  L: supernodal
 * symbolic or numeric factor with L->x allocated (xsize doubles).  Returns TRUE
 * (also when not positive definite: status / L->minor tell), FALSE on failure. */
int ssamd_cpu_super_numeric (cholmod_sparse *A, double beta, cholmod_factor *L, cholmod_common *Common)
{
    bind_blas_once () ;
    const int have_blas = g_blas.handle != NULL ;
    const Int n = (Int) L->n, nsuper = (Int) L->nsuper ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s ;
    const Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    const double *Ax = A->x ;
    double *Lx = L->x ;
    const int packed = A->packed ;

    /* workspace: position of a global row in the current supernode; the supernode
     * of every column; per descendant the cursor into its row list and the link
     * of the pending list it currently sits on */
    Int *where = cholmod_l_malloc (n > 0 ? n : 1, sizeof (Int), Common) ;
    Int *col2s = cholmod_l_malloc (n > 0 ? n : 1, sizeof (Int), Common) ;
    Int *cursor = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *link = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *pending = cholmod_l_malloc (nsuper + 1, sizeof (Int), Common) ;
    Int *relpos = cholmod_l_malloc (L->maxesize + 1, sizeof (Int), Common) ;
    double *C = cholmod_l_malloc (L->maxcsize > 0 ? L->maxcsize : 1, sizeof (double), Common) ;
    int ok = where && col2s && cursor && link && pending && relpos && C ;
    Int sfail = EMPTY, info_fail = 0 ;
    double t_syrk = 0, t_gemm = 0, t_potrf = 0, t_trsm = 0, t_asm = 0 ;
    size_t n_syrk = 0, n_gemm = 0, n_potrf = 0, n_trsm = 0 ;
    if (ok)
    {
        for (Int s = 0 ; s < nsuper ; s++)
        {
            for (Int k = Super [s] ; k < Super [s+1] ; k++) col2s [k] = s ;
            pending [s] = EMPTY ; link [s] = EMPTY ; cursor [s] = 0 ;
        }
        for (Int i = 0 ; i < n ; i++) where [i] = EMPTY ;
    }
    for (Int s = 0 ; ok && s < nsuper ; s++)
    {
        const Int k1 = Super [s], k2 = Super [s+1], nscol = k2 - k1 ;
        const Int psi = Lpi [s], nsrow = Lpi [s+1] - psi, psx = Lpx [s] ;
        double *Fs = Lx + psx ;
        double t0 ;
        for (Int r = 0 ; r < nsrow ; r++) where [Ls [psi + r]] = r ;
        Int good = nscol ;
        /* pass 0: the supernode as the reference builds it; pass 1 only after a
         * failed pivot: the same again, to factor the columns before that pivot
         * (the reference's "repeat supernode", t_cholmod_super_numeric.c:883-968;
         * the pending lists are only advanced once the supernode is done) */
        for (int pass = 0 ; pass < 2 ; pass++)
        {
            t0 = now_s () ;
            memset (Fs, 0, (size_t) nsrow * (size_t) nscol * sizeof (double)) ;
            /* A(:, k1:k2-1), lower part, into the supernode (entries outside the
             * symbolic pattern are dropped, t_cholmod_super_numeric.c:377-378) */
            for (Int k = k1 ; k < k2 ; k++)
            {
                Int p = Ap [k], pend = packed ? Ap [k+1] : p + Anz [k] ;
                double *col = Fs + (k - k1) * nsrow ;
                for ( ; p < pend ; p++)
                {
                    Int i = Ai [p] ;
                    if (i < k) continue ;
                    Int r = where [i] ;
                    if (r >= 0 && r < nsrow && Ls [psi + r] == i) col [r] = Ax [p] ;
                }
                col [k - k1] += beta ;
            }
            t_asm += now_s () - t0 ;
            /* updates from the descendants waiting on s */
            for (Int d = pending [s] ; d != EMPTY ; d = link [d])
            {
                const Int dk = Super [d+1] - Super [d] ;               /* columns of d */
                const Int dpi = Lpi [d], drows = Lpi [d+1] - dpi ;
                const Int q1 = cursor [d] ;                             /* first row of d inside s */
                Int q2 = q1 ;
                while (q2 < drows && Ls [dpi + q2] < k2) q2++ ;
                const Int n1 = q2 - q1, n2 = drows - q1 ;               /* rows inside s / from there down */
                const double *Ld = Lx + Lpx [d] + q1 ;                  /* ld = drows */
                for (Int r = 0 ; r < n2 ; r++) relpos [r] = where [Ls [dpi + q1 + r]] ;
                if (have_blas)
                {
                    const double one = 1.0, zero = 0.0 ;
                    int in1 = (int) n1, idk = (int) dk, ild = (int) drows, ildc = (int) n2, in3 = (int) (n2 - n1) ;
                    t0 = now_s () ;
                    blas_threads_for ((double) n1 * (double) n2 * (double) dk * 2.0) ;     /* (syrk + gemm of this descendant) */
                    g_blas.syrk ("L", "N", &in1, &idk, &one, Ld, &ild, &zero, C, &ildc) ;
                    t_syrk += now_s () - t0 ; n_syrk++ ;
                    if (in3 > 0)
                    {
                        t0 = now_s () ;
                        g_blas.gemm ("N", "C", &in3, &in1, &idk, &one, Ld + n1, &ild, Ld, &ild, &zero, C + n1, &ildc) ;
                        t_gemm += now_s () - t0 ; n_gemm++ ;
                    }
                    for (Int j = 0 ; j < n1 ; j++)
                    {
                        double *dst = Fs + relpos [j] * nsrow ;
                        const double *cj = C + j * n2 ;
                        for (Int i = j ; i < n2 ; i++) dst [relpos [i]] -= cj [i] ;
                    }
                }
                else
                {
                    /* built-in: the scratch receives 0 - L_d L_d' (lower trapezoid) */
                    t0 = now_s () ;
                    for (Int j = 0 ; j < n1 ; j++) memset (C + j * n2 + j, 0, (size_t) (n2 - j) * sizeof (double)) ;
                    k_gemm_nt (n2, n1, dk, Ld, drows, Ld, drows, C, n2, 1) ;
                    t_syrk += now_s () - t0 ; n_syrk++ ;
                    for (Int j = 0 ; j < n1 ; j++)
                    {
                        double *dst = Fs + relpos [j] * nsrow ;
                        const double *cj = C + j * n2 ;
                        for (Int i = j ; i < n2 ; i++) dst [relpos [i]] += cj [i] ;
                    }
                }
            }
            /* diagonal block: the first `good` columns */
            Int info = 0 ;
            t0 = now_s () ;
            if (have_blas)
            {
                int in = (int) good, ild = (int) nsrow, iinfo = 0 ;
                blas_threads_for ((double) good * (double) good * (double) good / 3.0) ;
                g_blas.potrf ("L", &in, Fs, &ild, &iinfo) ;
                info = iinfo ;
            }
            else info = k_potrf (good, Fs, nsrow) ;
            t_potrf += now_s () - t0 ; n_potrf++ ;
            if (info > 0 && pass == 0)
            {
                sfail = s ; info_fail = info ;
                good = (info == 1 || Common->quick_return_if_not_posdef) ? 0 : info - 1 ;
                if (good > 0) continue ;                /* redo with the leading columns only */
            }
            break ;
        }
        if (good > 0 && nsrow > good)
        {
            /* rows below the factored block (after a failed pivot that includes the
             * remaining rows of the diagonal block, as in the reference) */
            t0 = now_s () ;
            if (have_blas)
            {
                const double one = 1.0 ;
                int im = (int) (nsrow - good), in = (int) good, ild = (int) nsrow ;
                blas_threads_for ((double) (nsrow - good) * (double) good * (double) good) ;
                g_blas.trsm ("R", "L", "C", "N", &im, &in, &one, Fs, &ild, Fs + good, &ild) ;
            }
            else k_trsm (nsrow - good, good, Fs, nsrow, Fs + good, nsrow) ;
            t_trsm += now_s () - t0 ; n_trsm++ ;
        }
        if (sfail != EMPTY)
        {
            /* zero the columns from the failed pivot on, and every later supernode */
            memset (Fs + good * nsrow, 0, (size_t) nsrow * (size_t) (nscol - good) * sizeof (double)) ;
            if (Lpx [s+1] < (Int) L->xsize)
                memset (Lx + Lpx [s+1], 0, (size_t) ((Int) L->xsize - Lpx [s+1]) * sizeof (double)) ;
            break ;
        }
        /* the descendants move on to the supernode of their next row, if any */
        for (Int d = pending [s] ; d != EMPTY ; )
        {
            Int dnext = link [d] ;
            const Int dpi = Lpi [d], drows = Lpi [d+1] - dpi ;
            Int q2 = cursor [d] ;
            while (q2 < drows && Ls [dpi + q2] < k2) q2++ ;
            cursor [d] = q2 ;
            if (q2 < drows)
            {
                Int t = col2s [Ls [dpi + q2]] ;
                link [d] = pending [t] ; pending [t] = d ;
            }
            d = dnext ;
        }
        pending [s] = EMPTY ;
        /* s becomes a pending descendant of the supernode of its first row below */
        cursor [s] = nscol ;
        if (nsrow > nscol)
        {
            Int t = col2s [Ls [psi + nscol]] ;
            link [s] = pending [t] ; pending [t] = s ;
        }
    }
    if (where) cholmod_l_free (n > 0 ? n : 1, sizeof (Int), where, Common) ;
    if (col2s) cholmod_l_free (n > 0 ? n : 1, sizeof (Int), col2s, Common) ;
    if (cursor) cholmod_l_free (nsuper + 1, sizeof (Int), cursor, Common) ;
    if (link) cholmod_l_free (nsuper + 1, sizeof (Int), link, Common) ;
    if (pending) cholmod_l_free (nsuper + 1, sizeof (Int), pending, Common) ;
    if (relpos) cholmod_l_free (L->maxesize + 1, sizeof (Int), relpos, Common) ;
    if (C) cholmod_l_free (L->maxcsize > 0 ? L->maxcsize : 1, sizeof (double), C, Common) ;
    if (!ok) return FALSE ;
    /* the reference's counters (cholmod_core.h:1004-1024) */
    Common->cholmod_cpu_syrk_time = t_syrk ; Common->cholmod_cpu_gemm_time = t_gemm ;
    Common->cholmod_cpu_potrf_time = t_potrf ; Common->cholmod_cpu_trsm_time = t_trsm ;
    Common->cholmod_cpu_syrk_calls = n_syrk ; Common->cholmod_cpu_gemm_calls = n_gemm ;
    Common->cholmod_cpu_potrf_calls = n_potrf ; Common->cholmod_cpu_trsm_calls = n_trsm ;
    Common->cholmod_gpu_syrk_time = Common->cholmod_gpu_gemm_time = 0 ;
    Common->cholmod_gpu_potrf_time = Common->cholmod_gpu_trsm_time = 0 ;
    Common->cholmod_gpu_syrk_calls = Common->cholmod_gpu_gemm_calls = 0 ;
    Common->cholmod_gpu_potrf_calls = Common->cholmod_gpu_trsm_calls = 0 ;
    Common->cholmod_assemble_time = t_asm ; Common->cholmod_assemble_time2 = 0 ;
    L->minor = (size_t) n ;
    if (sfail != EMPTY)
    {
        L->minor = (size_t) (Super [sfail] + info_fail - 1) ;
        ERROR (CHOLMOD_NOT_POSDEF, "matrix not positive definite") ;
    }
    return TRUE ;
}

/* ---- triangular solves on the host factor ------------------------------------------------ */

/* which: 1 = L x = b, 2 = L' x = b, 0 = both; X n-by-nrhs in place, leading dim ldx
 * (reference t_cholmod_super_solve.c:14-411, restated with plain loops) */
void ssamd_cpu_super_solve (int which, const cholmod_factor *L, double *X, Int nrhs, Int ldx)
{
    const Int nsuper = (Int) L->nsuper ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s ;
    const double *Lx = L->x ;
#pragma omp parallel for schedule(static) if (nrhs > 1)
    for (Int r = 0 ; r < nrhs ; r++)
    {
        double *x = X + r * ldx ;
        if (which == 0 || which == 1)
        {
            for (Int s = 0 ; s < nsuper ; s++)
            {
                const Int k1 = Super [s], nscol = Super [s+1] - k1, psi = Lpi [s], nsrow = Lpi [s+1] - psi ;
                const double *F = Lx + Lpx [s] ;
                for (Int j = 0 ; j < nscol ; j++)
                {
                    double xj = x [k1 + j] / F [j + j * nsrow] ;
                    x [k1 + j] = xj ;
                    const double *col = F + j * nsrow ;
                    for (Int i = j + 1 ; i < nscol ; i++) x [k1 + i] -= col [i] * xj ;
                    for (Int i = nscol ; i < nsrow ; i++) x [Ls [psi + i]] -= col [i] * xj ;
                }
            }
        }
        if (which == 0 || which == 2)
        {
            for (Int s = nsuper - 1 ; s >= 0 ; s--)
            {
                const Int k1 = Super [s], nscol = Super [s+1] - k1, psi = Lpi [s], nsrow = Lpi [s+1] - psi ;
                const double *F = Lx + Lpx [s] ;
                for (Int j = nscol - 1 ; j >= 0 ; j--)
                {
                    const double *col = F + j * nsrow ;
                    double v = x [k1 + j] ;
                    for (Int i = j + 1 ; i < nscol ; i++) v -= col [i] * x [k1 + i] ;
                    for (Int i = nscol ; i < nsrow ; i++) v -= col [i] * x [Ls [psi + i]] ;
                    x [k1 + j] = v / col [j] ;
                }
            }
        }
    }
}
