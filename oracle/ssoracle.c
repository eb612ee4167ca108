/* ssoracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the CHOLMOD supernodal Cholesky
 * path (analyze -> super_symbolic -> super_numeric -> super_lsolve/ltsolve) of
 * the reference tree (sergiud/SuiteSparse, CHOLMOD 3.0.14).  It is the checker
 * for the HIP engine: only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py may load it.  The product library never links or calls it.
 *
 * Pinning: CHOLMOD itself cannot be compiled in this image under the project
 * rules (it needs the CMake-generated cholmod_export.h / cholmod_config.h and an
 * external BLAS/LAPACK).  This restatement is pinned
 *  - against REFERENCE CODE COMPILED HERE: the reference tree's CSparse
 *    (oracle/Makefile target `ref`, oracle/csref.py): etree and column counts
 *    identical, the pattern of L contained in / equal to the supernodal structure,
 *    the values of L to 1e-12 (tests/test_csparse_reference.py);
 *  - against the reference-HELD record LDL/Demo/ldlmain.out (48 pairs "Nz in L /
 *    Flop count": lnz, fl; tests/test_ldl_recorded.py);
 *  - against the outputs of the compiled CHOLMOD recorded in SURVEY.md 8c/8d
 *    (bcsstk01 maps, BLAS call counts, ND Poisson nsuper / update counts;
 *    tests/golden/reference_recorded.json, tests/test_oracle_golden.py) for what
 *    CSparse has no notion of: the relaxed supernode partition, maxcsize /
 *    maxesize, the descendant lists.
 * See DESIGN.md section 5 for what each of them constrains.
 *
 * Every routine cites the reference file:line whose behaviour it follows
 * (paths relative to the reference root, CHOLMOD/...).  Index type is int64
 * (the reference's cholmod_l_* / DLONG build).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <dlfcn.h>

typedef int64_t Int;
#define EMPTY (-1)
#define ORC_OK 0
#define ORC_NOT_POSDEF 1
#define ORC_OUT_OF_MEMORY (-2)
#define ORC_TOO_LARGE (-3)
#define ORC_INVALID (-4)

typedef struct orc_factor
{
    Int n, nsuper, ssize, xsize, maxcsize, maxesize, minor ;
    Int *Perm, *ColCount, *Parent ;     /* size n, in the final ordering */
    Int *super, *pi, *px, *s ;          /* supernodal maps */
    double *x ;                         /* xsize, NULL until factorized */
    double fl, lnz, anz ;
    int status ;
    Int calls [4] ;                     /* syrk, gemm, potrf, trsm */
    double exec_flops ;                 /* flops actually executed by numeric */
} orc_factor ;

/* ------------------------------------------------------------------------ */
/* optional external BLAS (dlopen) for the cpu_baseline leg                  */
/* ------------------------------------------------------------------------ */

typedef void (*dgemm_f) (const char *, const char *, const int *, const int *,
    const int *, const double *, const double *, const int *, const double *,
    const int *, const double *, double *, const int *) ;
typedef void (*dsyrk_f) (const char *, const char *, const int *, const int *,
    const double *, const double *, const int *, const double *, double *,
    const int *) ;
typedef void (*dtrsm_f) (const char *, const char *, const char *, const char *,
    const int *, const int *, const double *, const double *, const int *,
    double *, const int *) ;
typedef void (*dpotrf_f) (const char *, const int *, double *, const int *,
    int *) ;

static dgemm_f  x_dgemm  = NULL ;
static dsyrk_f  x_dsyrk  = NULL ;
static dtrsm_f  x_dtrsm  = NULL ;
static dpotrf_f x_dpotrf = NULL ;

/* Returns 1 if an LP64 Fortran BLAS/LAPACK was bound from `path` using symbol
 * prefix `prefix` (e.g. "" for MKL/OpenBLAS, "scipy_" for scipy's bundled
 * OpenBLAS).  Mirrors the reference's BLAS_dsyrk/dgemm/dtrsm + LAPACK_dpotrf
 * macros (Include/cholmod_blas.h:172-381). */
int orc_bind_blas (const char *path, const char *prefix)
{
    char name [128] ;
    void *h = dlopen (path, RTLD_NOW | RTLD_GLOBAL) ;
    if (!h) return 0 ;
    snprintf (name, sizeof name, "%sdgemm_", prefix) ;
    dgemm_f g = (dgemm_f) dlsym (h, name) ;
    snprintf (name, sizeof name, "%sdsyrk_", prefix) ;
    dsyrk_f s = (dsyrk_f) dlsym (h, name) ;
    snprintf (name, sizeof name, "%sdtrsm_", prefix) ;
    dtrsm_f t = (dtrsm_f) dlsym (h, name) ;
    snprintf (name, sizeof name, "%sdpotrf_", prefix) ;
    dpotrf_f p = (dpotrf_f) dlsym (h, name) ;
    if (!g || !s || !t || !p) return 0 ;
    x_dgemm = g ; x_dsyrk = s ; x_dtrsm = t ; x_dpotrf = p ;
    return 1 ;
}

void orc_unbind_blas (void)
{
    x_dgemm = NULL ; x_dsyrk = NULL ; x_dtrsm = NULL ; x_dpotrf = NULL ;
}

/* ------------------------------------------------------------------------ */
/* built-in dense kernels (column-major, Fortran BLAS semantics)             */
/* ------------------------------------------------------------------------ */

#define CLONES __attribute__((target_clones("avx2,fma","default")))

/* C(m,n) = alpha_sign * A(m,k) * B(n,k)' + (acc ? C : 0).  sign=+1 or -1.
 * The reference calls dgemm("N","C",...) at t_cholmod_super_numeric.c:708-717. */
CLONES static void k_gemm_nt (Int m, Int n, Int k, const double *A, Int lda,
    const double *B, Int ldb, double *C, Int ldc, int acc, double sign)
{
    Int i, j, l ;
    for (j = 0 ; j + 4 <= n ; j += 4)
    {
        for (i = 0 ; i + 8 <= m ; i += 8)
        {
            double c [4][8] ;
            for (int jj = 0 ; jj < 4 ; jj++)
                for (int ii = 0 ; ii < 8 ; ii++) c [jj][ii] = 0 ;
            for (l = 0 ; l < k ; l++)
            {
                const double *a = A + i + l*lda ;
                double b0 = B [j + l*ldb], b1 = B [j+1 + l*ldb] ;
                double b2 = B [j+2 + l*ldb], b3 = B [j+3 + l*ldb] ;
                for (int ii = 0 ; ii < 8 ; ii++)
                {
                    double av = a [ii] ;
                    c [0][ii] += av * b0 ; c [1][ii] += av * b1 ;
                    c [2][ii] += av * b2 ; c [3][ii] += av * b3 ;
                }
            }
            for (int jj = 0 ; jj < 4 ; jj++)
            {
                double *cp = C + i + (j+jj)*ldc ;
                if (acc) for (int ii = 0 ; ii < 8 ; ii++) cp [ii] += sign * c [jj][ii] ;
                else     for (int ii = 0 ; ii < 8 ; ii++) cp [ii]  = sign * c [jj][ii] ;
            }
        }
        for ( ; i < m ; i++)
        {
            double c0 = 0, c1 = 0, c2 = 0, c3 = 0 ;
            for (l = 0 ; l < k ; l++)
            {
                double av = A [i + l*lda] ;
                c0 += av * B [j + l*ldb] ;   c1 += av * B [j+1 + l*ldb] ;
                c2 += av * B [j+2 + l*ldb] ; c3 += av * B [j+3 + l*ldb] ;
            }
            if (acc)
            {
                C [i + j*ldc] += sign*c0 ;     C [i + (j+1)*ldc] += sign*c1 ;
                C [i + (j+2)*ldc] += sign*c2 ; C [i + (j+3)*ldc] += sign*c3 ;
            }
            else
            {
                C [i + j*ldc] = sign*c0 ;     C [i + (j+1)*ldc] = sign*c1 ;
                C [i + (j+2)*ldc] = sign*c2 ; C [i + (j+3)*ldc] = sign*c3 ;
            }
        }
    }
    for ( ; j < n ; j++)
    {
        for (i = 0 ; i < m ; i++)
        {
            double c0 = 0 ;
            for (l = 0 ; l < k ; l++) c0 += A [i + l*lda] * B [j + l*ldb] ;
            if (acc) C [i + j*ldc] += sign*c0 ; else C [i + j*ldc] = sign*c0 ;
        }
    }
}

/* lower triangle of C(n,n) = sign * A(n,k)*A(n,k)' (+ C if acc), blocked so
 * that only tiles touching the lower triangle are computed.  The strictly
 * upper part inside diagonal tiles is also written (dsyrk leaves it alone, but
 * the reference never reads it: t_cholmod_super_numeric.c:756-772 uses i>=j). */
static void k_syrk_ln (Int n, Int k, const double *A, Int lda, double *C,
    Int ldc, int acc, double sign)
{
    const Int nb = 64 ;
    for (Int j = 0 ; j < n ; j += nb)
    {
        Int jb = (n - j < nb) ? (n - j) : nb ;
        k_gemm_nt (n - j, jb, k, A + j, lda, A + j, lda, C + j + j*ldc, ldc,
            acc, sign) ;
    }
}

/* unblocked lower Cholesky, LAPACK dpotf2 semantics: returns info (1-based
 * index of the first non-positive pivot, 0 if ok).  NaN pivots do not trip
 * (reference comment t_cholmod_super_numeric.c:907-908). */
static Int k_potf2 (Int n, double *A, Int lda)
{
    for (Int j = 0 ; j < n ; j++)
    {
        double ajj = A [j + j*lda] ;
        for (Int l = 0 ; l < j ; l++) ajj -= A [j + l*lda] * A [j + l*lda] ;
        if (ajj <= 0.0) { A [j + j*lda] = ajj ; return j + 1 ; }
        ajj = sqrt (ajj) ;
        A [j + j*lda] = ajj ;
        for (Int i = j + 1 ; i < n ; i++)
        {
            double v = A [i + j*lda] ;
            for (Int l = 0 ; l < j ; l++) v -= A [i + l*lda] * A [j + l*lda] ;
            A [i + j*lda] = v / ajj ;
        }
    }
    return 0 ;
}

/* B(m,n) := B * inv(L)' with L(n,n) lower, non-unit: dtrsm("R","L","C","N").
 * reference call: t_cholmod_super_numeric.c:997-1002 */
static void k_trsm_rltn (Int m, Int n, const double *L, Int ldl, double *B,
    Int ldb)
{
    const Int nb = 32 ;
    for (Int j = 0 ; j < n ; j += nb)
    {
        Int jb = (n - j < nb) ? (n - j) : nb ;
        if (j > 0)
        {
            /* B(:, j:j+jb) -= B(:, 0:j) * L(j:j+jb, 0:j)' */
            k_gemm_nt (m, jb, j, B, ldb, L + j, ldl, B + j*ldb, ldb, 1, -1.0) ;
        }
        for (Int jj = j ; jj < j + jb ; jj++)
        {
            double *bj = B + jj*ldb ;
            for (Int l = j ; l < jj ; l++)
            {
                double f = L [jj + l*ldl] ;
                const double *bl = B + l*ldb ;
                for (Int i = 0 ; i < m ; i++) bj [i] -= bl [i] * f ;
            }
            double d = L [jj + jj*ldl] ;
            for (Int i = 0 ; i < m ; i++) bj [i] /= d ;
        }
    }
}

/* blocked lower Cholesky: LAPACK dpotrf("L") semantics (info as dpotf2).
 * reference call: t_cholmod_super_numeric.c:864-867 */
static Int k_potrf (Int n, double *A, Int lda)
{
    const Int nb = 64 ;
    if (n <= nb) return k_potf2 (n, A, lda) ;
    for (Int j = 0 ; j < n ; j += nb)
    {
        Int jb = (n - j < nb) ? (n - j) : nb ;
        /* A(j:,j:j+jb) -= A(j:,0:j) A(j:j+jb,0:j)' (left-looking, as dpotrf) */
        if (j > 0)
        {
            k_gemm_nt (n - j, jb, j, A + j, lda, A + j, lda, A + j + j*lda,
                lda, 1, -1.0) ;
        }
        Int info = k_potf2 (jb, A + j + j*lda, lda) ;
        if (info) return j + info ;
        if (j + jb < n)
        {
            k_trsm_rltn (n - j - jb, jb, A + j + j*lda, lda,
                A + (j + jb) + j*lda, lda) ;
        }
    }
    return 0 ;
}

/* ---- dispatchers: external BLAS when bound, built-in otherwise ---------- */

static void d_syrk (Int n, Int k, const double *A, Int lda, double *C, Int ldc)
{
    if (x_dsyrk)
    {
        int N = (int) n, K = (int) k, LDA = (int) lda, LDC = (int) ldc ;
        double one = 1, zero = 0 ;
        x_dsyrk ("L", "N", &N, &K, &one, A, &LDA, &zero, C, &LDC) ;
    }
    else k_syrk_ln (n, k, A, lda, C, ldc, 0, 1.0) ;
}

static void d_gemm (Int m, Int n, Int k, const double *A, Int lda,
    const double *B, Int ldb, double *C, Int ldc)
{
    if (x_dgemm)
    {
        int M = (int) m, N = (int) n, K = (int) k, LDA = (int) lda,
            LDB = (int) ldb, LDC = (int) ldc ;
        double one = 1, zero = 0 ;
        x_dgemm ("N", "C", &M, &N, &K, &one, A, &LDA, B, &LDB, &zero, C, &LDC) ;
    }
    else k_gemm_nt (m, n, k, A, lda, B, ldb, C, ldc, 0, 1.0) ;
}

static Int d_potrf (Int n, double *A, Int lda)
{
    if (x_dpotrf)
    {
        int N = (int) n, LDA = (int) lda, info = 0 ;
        x_dpotrf ("L", &N, A, &LDA, &info) ;
        return info ;
    }
    return k_potrf (n, A, lda) ;
}

static void d_trsm (Int m, Int n, const double *L, Int ldl, double *B, Int ldb)
{
    if (x_dtrsm)
    {
        int M = (int) m, N = (int) n, LDL = (int) ldl, LDB = (int) ldb ;
        double one = 1 ;
        x_dtrsm ("R", "L", "C", "N", &M, &N, &one, L, &LDL, B, &LDB) ;
    }
    else k_trsm_rltn (m, n, L, ldl, B, ldb) ;
}

/* ------------------------------------------------------------------------ */
/* symmetric permutation (the role of cholmod_ptranspose in the callers)     */
/* ------------------------------------------------------------------------ */

/* Build T = triu(P A P') (upper != 0) or tril(P A P') (upper == 0) in packed
 * CSC with sorted columns, from a symmetric A of which only the `stype`
 * triangle is stored (entries in the other triangle are ignored, as
 * Core/cholmod_transpose.c:871 `ptranspose` does for stype != 0).  Perm may be
 * NULL (identity).  Tx may be NULL (pattern only).  Caller frees Tp/Ti/Tx.
 * Follows the permuted-transpose used at Cholesky/cholmod_factorize.c:225-244
 * and Cholesky/cholmod_analyze.c:174-298 (permute_matrices). */
static int sym_permute_e (Int n, const Int *Ap, const Int *Ai, const double *Ax,
    int stype, const Int *Perm, int upper, Int **Tp_out, Int **Ti_out,
    double **Tx_out, int E) ;
static int sym_permute (Int n, const Int *Ap, const Int *Ai, const double *Ax,
    int stype, const Int *Perm, int upper, Int **Tp_out, Int **Ti_out,
    double **Tx_out)
{
    return sym_permute_e (n, Ap, Ai, Ax, stype, Perm, upper, Tp_out, Ti_out, Tx_out, 1) ;
}
/* E = 1: real values; E = 2: complex (interleaved) values of a Hermitian matrix --
 * an entry that lands in the other triangle is conjugated (the conjugate permuted
 * transpose, `values = 2`, of Cholesky/cholmod_factorize.c:225-244). */
static int sym_permute_e (Int n, const Int *Ap, const Int *Ai, const double *Ax,
    int stype, const Int *Perm, int upper, Int **Tp_out, Int **Ti_out,
    double **Tx_out, int E)
{
    Int *Pinv = malloc ((n+1) * sizeof (Int)) ;
    Int *cnt = calloc (n+1, sizeof (Int)) ;
    Int *Tp = malloc ((n+1) * sizeof (Int)) ;
    if (!Pinv || !cnt || !Tp) return 0 ;
    for (Int k = 0 ; k < n ; k++) Pinv [Perm ? Perm [k] : k] = k ;
    Int nz = 0 ;
    for (Int j = 0 ; j < n ; j++)
    {
        for (Int p = Ap [j] ; p < Ap [j+1] ; p++)
        {
            Int i = Ai [p] ;
            if ((stype < 0 && i < j) || (stype > 0 && i > j)) continue ;
            Int r = Pinv [i], c = Pinv [j] ;
            Int lo = r < c ? r : c, hi = r < c ? c : r ;
            cnt [upper ? hi : lo]++ ;
            nz++ ;
        }
    }
    Tp [0] = 0 ;
    for (Int j = 0 ; j < n ; j++) Tp [j+1] = Tp [j] + cnt [j] ;
    Int *Ti = malloc ((nz > 0 ? nz : 1) * sizeof (Int)) ;
    double *Tx = Ax ? malloc ((nz > 0 ? nz : 1) * E * sizeof (double)) : NULL ;
    /* two-pass bucket sort by row to get sorted columns: first bucket the
     * entries by their row index, then sweep rows in order */
    Int *rp = calloc (n+2, sizeof (Int)) ;
    Int *ecol = malloc ((nz > 0 ? nz : 1) * sizeof (Int)) ;
    Int *erow = malloc ((nz > 0 ? nz : 1) * sizeof (Int)) ;
    double *eval = Ax ? malloc ((nz > 0 ? nz : 1) * E * sizeof (double)) : NULL ;
    if (!Ti || !rp || !ecol || !erow || (Ax && (!Tx || !eval))) return 0 ;
    for (Int j = 0 ; j < n ; j++)
        for (Int p = Ap [j] ; p < Ap [j+1] ; p++)
        {
            Int i = Ai [p] ;
            if ((stype < 0 && i < j) || (stype > 0 && i > j)) continue ;
            Int r = Pinv [i], c = Pinv [j] ;
            Int lo = r < c ? r : c, hi = r < c ? c : r ;
            rp [(upper ? lo : hi) + 1]++ ;
        }
    for (Int i = 0 ; i < n ; i++) rp [i+1] += rp [i] ;
    for (Int j = 0 ; j < n ; j++)
        for (Int p = Ap [j] ; p < Ap [j+1] ; p++)
        {
            Int i = Ai [p] ;
            if ((stype < 0 && i < j) || (stype > 0 && i > j)) continue ;
            Int r = Pinv [i], c = Pinv [j] ;
            Int lo = r < c ? r : c, hi = r < c ? c : r ;
            Int row = upper ? lo : hi, col = upper ? hi : lo ;
            Int q = rp [row]++ ;
            erow [q] = row ; ecol [q] = col ;
            if (Ax && E == 1) eval [q] = Ax [p] ;
            else if (Ax)
            {
                eval [2*q] = Ax [2*p] ;
                eval [2*q+1] = (row == r) ? Ax [2*p+1] : -Ax [2*p+1] ;
            }
        }
    memset (cnt, 0, (n+1) * sizeof (Int)) ;
    for (Int q = 0 ; q < nz ; q++)
    {
        Int col = ecol [q] ;
        Int dst = Tp [col] + cnt [col]++ ;
        Ti [dst] = erow [q] ;
        if (Ax) for (int e = 0 ; e < E ; e++) Tx [E*dst+e] = eval [E*q+e] ;
    }
    free (rp) ; free (ecol) ; free (erow) ; free (eval) ;
    free (Pinv) ; free (cnt) ;
    *Tp_out = Tp ; *Ti_out = Ti ;
    if (Tx_out) *Tx_out = Tx ; else free (Tx) ;
    return 1 ;
}

/* ------------------------------------------------------------------------ */
/* etree / postorder / column counts                                          */
/* ------------------------------------------------------------------------ */

/* Elimination tree of an upper-stored symmetric pattern.
 * reference: Cholesky/cholmod_etree.c:81-223 (stype>0 branch + update_etree) */
static void etree_upper (Int n, const Int *Up, const Int *Ui, Int *Parent)
{
    Int *Anc = malloc ((n > 0 ? n : 1) * sizeof (Int)) ;
    for (Int j = 0 ; j < n ; j++) { Parent [j] = EMPTY ; Anc [j] = EMPTY ; }
    for (Int j = 0 ; j < n ; j++)
    {
        for (Int p = Up [j] ; p < Up [j+1] ; p++)
        {
            Int i = Ui [p] ;
            if (i >= j) continue ;
            for (;;)
            {
                Int a = Anc [i] ;
                if (a == j) break ;
                Anc [i] = j ;
                if (a == EMPTY) { Parent [i] = j ; break ; }
                i = a ;
            }
        }
    }
    free (Anc) ;
}

/* Postorder of a forest; children visited in increasing Weight (ties: by
 * increasing node number), or by node number if Weight is NULL.
 * reference: Cholesky/cholmod_postorder.c:138-290 (bucket lists + dfs :60-94) */
static Int postorder (Int n, const Int *Parent, const Int *Weight, Int *Post)
{
    Int *Head = malloc ((n+1) * sizeof (Int)) ;
    Int *Next = malloc ((n+1) * sizeof (Int)) ;
    Int *Stack = malloc ((n+1) * sizeof (Int)) ;
    for (Int j = 0 ; j < n ; j++) Head [j] = EMPTY ;
    if (!Weight)
    {
        for (Int j = n-1 ; j >= 0 ; j--)
        {
            Int p = Parent [j] ;
            if (p >= 0 && p < n) { Next [j] = Head [p] ; Head [p] = j ; }
        }
    }
    else
    {
        Int *Whead = Stack ;
        for (Int w = 0 ; w < n ; w++) Whead [w] = EMPTY ;
        for (Int j = 0 ; j < n ; j++)
        {
            Int p = Parent [j] ;
            if (p >= 0 && p < n)
            {
                Int w = Weight [j] ;
                if (w < 0) w = 0 ;
                if (w > n-1) w = n-1 ;
                Next [j] = Whead [w] ; Whead [w] = j ;
            }
        }
        for (Int w = n-1 ; w >= 0 ; w--)
        {
            Int nextj ;
            for (Int j = Whead [w] ; j != EMPTY ; j = nextj)
            {
                nextj = Next [j] ;
                Int p = Parent [j] ;
                Next [j] = Head [p] ; Head [p] = j ;
            }
        }
    }
    Int k = 0 ;
    for (Int r = 0 ; r < n ; r++)
    {
        if (Parent [r] != EMPTY) continue ;
        Int top = 0 ;
        Stack [0] = r ;
        while (top >= 0)
        {
            Int p = Stack [top] ;
            Int j = Head [p] ;
            if (j == EMPTY) { top-- ; Post [k++] = p ; }
            else { Head [p] = Next [j] ; Stack [++top] = j ; }
        }
    }
    free (Head) ; free (Next) ; free (Stack) ;
    return k ;
}

/* Column counts of L (diagonal included) by explicit row-subtree marking:
 * row i of L is the union of the etree paths from every k in A(0:i-1,i) up to
 * (not including) i.  O(nnz(L)) time -- deliberately the textbook definition
 * rather than the reference's skeleton/LCA algorithm; the result is the unique
 * ColCount the reference computes at Cholesky/cholmod_rowcolcounts.c:184-533,
 * with fl = sum cc^2 and lnz = sum cc as at :517-528. */
static void colcounts (Int n, const Int *Up, const Int *Ui, const Int *Parent,
    Int *ColCount, double *fl, double *lnz)
{
    Int *mark = malloc ((n > 0 ? n : 1) * sizeof (Int)) ;
    for (Int j = 0 ; j < n ; j++) { ColCount [j] = 1 ; mark [j] = EMPTY ; }
    for (Int i = 0 ; i < n ; i++)
    {
        mark [i] = i ;
        for (Int p = Up [i] ; p < Up [i+1] ; p++)
        {
            Int k = Ui [p] ;
            if (k >= i) continue ;
            while (mark [k] != i)
            {
                ColCount [k]++ ;
                mark [k] = i ;
                k = Parent [k] ;
            }
        }
    }
    double f = 0, l = 0 ;
    for (Int j = 0 ; j < n ; j++)
    {
        double c = (double) ColCount [j] ;
        f += c*c ; l += c ;
    }
    *fl = f ; *lnz = l ;
    free (mark) ;
}

/* ------------------------------------------------------------------------ */
/* supernodal symbolic analysis                                               */
/* ------------------------------------------------------------------------ */

/* reference: Supernodal/cholmod_super_symbolic.c:155-958 with useGPU==0 and
 * for_whom==CHOLMOD_ANALYZE_FOR_CHOLESKY.  U is the upper-stored permuted
 * pattern, Parent/ColCount in the same ordering. */
static int super_symbolic (orc_factor *L, const Int *Up, const Int *Ui,
    const Int nrelax [3], const double zrelax_in [3])
{
    Int n = L->n ;
    const Int *Parent = L->Parent, *ColCount = L->ColCount ;
    double zrelax [3] ;
    for (int t = 0 ; t < 3 ; t++)           /* NaN -> 0, :370-372 */
        zrelax [t] = (zrelax_in [t] != zrelax_in [t]) ? 0 : zrelax_in [t] ;

    Int *Wi = calloc (n+1, sizeof (Int)) ;
    Int *Super = malloc ((n+2) * sizeof (Int)) ;
    Int *SuperMap = malloc ((n+1) * sizeof (Int)) ;
    Int *Sparent = malloc ((n+1) * sizeof (Int)) ;
    Int *Snz = malloc ((n+1) * sizeof (Int)) ;
    Int *Merged = malloc ((n+1) * sizeof (Int)) ;
    Int *Zeros = malloc ((n+1) * sizeof (Int)) ;
    Int *Nscol = malloc ((n+1) * sizeof (Int)) ;
    if (!Wi || !Super || !SuperMap || !Sparent || !Snz || !Merged || !Zeros
        || !Nscol) return ORC_OUT_OF_MEMORY ;

    /* fundamental supernodes, :397-436 */
    for (Int j = 0 ; j < n ; j++)
        if (Parent [j] != EMPTY) Wi [Parent [j]]++ ;
    Int nfsuper = (n == 0) ? 0 : 1 ;
    Super [0] = 0 ;
    for (Int j = 1 ; j < n ; j++)
    {
        if (Parent [j-1] != j || ColCount [j-1] != ColCount [j] + 1
            || Wi [j] > 1)
        {
            Super [nfsuper++] = j ;
        }
    }
    Super [nfsuper] = n ;
    for (Int s = 0 ; s < nfsuper ; s++)
        for (Int k = Super [s] ; k < Super [s+1] ; k++) SuperMap [k] = s ;
    /* fundamental supernodal etree, :461-467 */
    for (Int s = 0 ; s < nfsuper ; s++)
    {
        Int parent = Parent [Super [s+1] - 1] ;
        Sparent [s] = (parent == EMPTY) ? EMPTY : SuperMap [parent] ;
    }
    /* relaxed amalgamation, :478-602 */
    for (Int s = 0 ; s < nfsuper ; s++)
    {
        Merged [s] = EMPTY ;
        Nscol [s] = Super [s+1] - Super [s] ;
        Zeros [s] = 0 ;
        Snz [s] = ColCount [Super [s]] ;
    }
    for (Int s = nfsuper - 2 ; s >= 0 ; s--)
    {
        Int ss = Sparent [s] ;
        if (ss == EMPTY) continue ;
        for (ss = Sparent [s] ; Merged [ss] != EMPTY ; ss = Merged [ss]) ;
        Int sparent = ss ;
        Int snext ;
        for (ss = Sparent [s] ; Merged [ss] != EMPTY ; ss = snext)
        {
            snext = Merged [ss] ;
            Merged [ss] = sparent ;
        }
        if (sparent != s+1) continue ;
        Int nscol0 = Nscol [s], nscol1 = Nscol [s+1] ;
        Int ns = nscol0 + nscol1 ;
        Int totzeros = Zeros [s+1] ;
        double lnz1 = (double) Snz [s+1] ;
        int merge ;
        if (ns <= nrelax [0])
        {
            merge = 1 ;             /* tiny: merge, zeros NOT counted (:530) */
        }
        else
        {
            double lnz0 = (double) Snz [s] ;
            double xnewzeros = nscol0 * (lnz1 + nscol0 - lnz0) ;
            Int newzeros = nscol0 * (Snz [s+1] + nscol0 - Snz [s]) ;
            if (xnewzeros == 0)
            {
                merge = 1 ;
            }
            else
            {
                double xtotzeros = ((double) totzeros) + xnewzeros ;
                double xns = (double) ns ;
                double xtotsize = (xns * (xns+1) / 2) + xns * (lnz1 - nscol1) ;
                double z = xtotzeros / xtotsize ;
                totzeros += newzeros ;
                merge = ((ns <= nrelax [1] && z < zrelax [0]) ||
                         (ns <= nrelax [2] && z < zrelax [1]) ||
                         (z < zrelax [2])) &&
                        (xtotsize < (double) INT64_MAX / sizeof (double)) ;
            }
        }
        if (merge)
        {
            Zeros [s] = totzeros ;
            Merged [s+1] = s ;
            Snz [s] = nscol0 + Snz [s+1] ;
            Nscol [s] += Nscol [s+1] ;
        }
    }
    /* relaxed supernode list, :612-653 */
    Int nsuper = 0 ;
    for (Int s = 0 ; s < nfsuper ; s++)
    {
        if (Merged [s] == EMPTY)
        {
            Super [nsuper] = Super [s] ;
            Snz [nsuper] = Snz [s] ;
            nsuper++ ;
        }
    }
    Super [nsuper] = n ;
    for (Int s = 0 ; s < nsuper ; s++)
        for (Int k = Super [s] ; k < Super [s+1] ; k++) SuperMap [k] = s ;
    for (Int s = 0 ; s < nsuper ; s++)
    {
        Int parent = Parent [Super [s+1] - 1] ;
        Sparent [s] = (parent == EMPTY) ? EMPTY : SuperMap [parent] ;
    }
    /* sizes, :659-699 */
    Int ssize = 0, xsize = 0 ;
    double xxsize = 0 ;
    for (Int s = 0 ; s < nsuper ; s++)
    {
        Int nscol = Super [s+1] - Super [s], nsrow = Snz [s] ;
        ssize += nsrow ;
        xsize += nscol * nsrow ;
        xxsize += ((double) nscol) * ((double) nsrow) ;
        if (ssize < 0 || xxsize > (double) INT64_MAX) return ORC_TOO_LARGE ;
    }
    if (xsize < 1) xsize = 1 ;
    if (ssize < 1) ssize = 1 ;
    L->nsuper = nsuper ; L->ssize = ssize ; L->xsize = xsize ;
    L->super = malloc ((nsuper+1) * sizeof (Int)) ;
    L->pi = malloc ((nsuper+1) * sizeof (Int)) ;
    L->px = malloc ((nsuper+1) * sizeof (Int)) ;
    L->s = malloc (ssize * sizeof (Int)) ;
    if (!L->super || !L->pi || !L->px || !L->s) return ORC_OUT_OF_MEMORY ;
    L->s [0] = 0 ;                                          /* :717 */
    for (Int s = 0 ; s <= nsuper ; s++) L->super [s] = Super [s] ;
    /* pi, px prefix sums, :734-761 */
    Int p = 0 ;
    for (Int s = 0 ; s < nsuper ; s++) { L->pi [s] = p ; p += Snz [s] ; }
    L->pi [nsuper] = p ;
    p = 0 ;
    for (Int s = 0 ; s < nsuper ; s++)
    {
        L->px [s] = p ;
        p += (Super [s+1] - Super [s]) * Snz [s] ;
    }
    L->px [nsuper] = p ;
    /* row structure via supernodal row-subtree walks, :786-835 + subtree :82 */
    Int *Lpi2 = Wi, *Flag = Zeros ;
    for (Int s = 0 ; s < nsuper ; s++) { Lpi2 [s] = L->pi [s] ; Flag [s] = EMPTY ; }
    Int mark = 0 ;
    Int *Ls = L->s ;
    for (Int s = 0 ; s < nsuper ; s++)
    {
        Int k1 = Super [s], k2 = Super [s+1] ;
        for (Int k = k1 ; k < k2 ; k++) Ls [Lpi2 [s]++] = k ;
        for (Int k = k1 ; k < k2 ; k++)
        {
            mark++ ;
            Flag [s] = mark ;
            for (Int q = Up [k] ; q < Up [k+1] ; q++)
            {
                Int i = Ui [q] ;
                if (i >= k1) continue ;
                for (Int si = SuperMap [i] ; Flag [si] < mark ; si = Sparent [si])
                {
                    Ls [Lpi2 [si]++] = k ;
                    Flag [si] = mark ;
                }
            }
        }
    }
    for (Int s = 0 ; s < nsuper ; s++)
        if (Lpi2 [s] != L->pi [s+1]) return ORC_INVALID ;  /* :840 */
    /* maxcsize / maxesize, :907-948 */
    Int maxcsize = 1, maxesize = 1 ;
    for (Int d = 0 ; d < nsuper ; d++)
    {
        Int nscol = Super [d+1] - Super [d] ;
        Int pp = L->pi [d] + nscol ;
        Int plast = pp, pend = L->pi [d+1] ;
        Int esize = pend - pp ;
        if (esize > maxesize) maxesize = esize ;
        Int slast = (pp == pend) ? EMPTY : SuperMap [Ls [pp]] ;
        for ( ; pp <= pend ; pp++)
        {
            Int s = (pp == pend) ? EMPTY : SuperMap [Ls [pp]] ;
            if (s != slast)
            {
                Int ndrow1 = pp - plast, ndrow2 = pend - plast ;
                Int csize = ndrow2 * ndrow1 ;
                if (csize > maxcsize) maxcsize = csize ;
                plast = pp ;
                slast = s ;
            }
        }
    }
    L->maxcsize = maxcsize ; L->maxesize = maxesize ;
    free (Wi) ; free (Super) ; free (SuperMap) ; free (Sparent) ; free (Snz) ;
    free (Merged) ; free (Zeros) ; free (Nscol) ;
    return ORC_OK ;
}

/* ------------------------------------------------------------------------ */
/* public: analyze                                                            */
/* ------------------------------------------------------------------------ */

void orc_free (orc_factor *L)
{
    if (!L) return ;
    free (L->Perm) ; free (L->ColCount) ; free (L->Parent) ;
    free (L->super) ; free (L->pi) ; free (L->px) ; free (L->s) ; free (L->x) ;
    free (L) ;
}

/* Ordering GIVEN (UserPerm) or NATURAL (NULL), etree, counts, optional
 * weighted postorder composed into Perm, then supernodal symbolic.
 * reference: Cholesky/cholmod_analyze.c:401-935 restricted to nmethods=1,
 * method[0].ordering in {CHOLMOD_GIVEN, CHOLMOD_NATURAL}, supernodal forced
 * (Common->supernodal = CHOLMOD_SUPERNODAL); analyze_ordering :312-372;
 * postorder composition :855-906. */
orc_factor *orc_analyze (Int n, const Int *Ap, const Int *Ai, int stype,
    const Int *UserPerm, int do_postorder, const Int *nrelax,
    const double *zrelax)
{
    static const Int nrelax_def [3] = {4, 16, 48} ;     /* cholmod_common.c:270 */
    static const double zrelax_def [3] = {0.8, 0.1, 0.05} ;
    if (!nrelax) nrelax = nrelax_def ;
    if (!zrelax) zrelax = zrelax_def ;
    if (stype == 0) return NULL ;
    orc_factor *L = calloc (1, sizeof (orc_factor)) ;
    L->n = n ; L->minor = n ;
    L->Perm = malloc ((n+1) * sizeof (Int)) ;
    L->ColCount = malloc ((n+1) * sizeof (Int)) ;
    L->Parent = malloc ((n+1) * sizeof (Int)) ;
    for (Int k = 0 ; k < n ; k++) L->Perm [k] = UserPerm ? UserPerm [k] : k ;
    Int *Up = NULL, *Ui = NULL ;
    if (!sym_permute (n, Ap, Ai, NULL, stype, L->Perm, 1, &Up, &Ui, NULL))
    { orc_free (L) ; return NULL ; }
    L->anz = (double) Up [n] ;
    etree_upper (n, Up, Ui, L->Parent) ;
    colcounts (n, Up, Ui, L->Parent, L->ColCount, &L->fl, &L->lnz) ;
    if (do_postorder)
    {
        Int *Post = malloc ((n+1) * sizeof (Int)) ;
        Int *W = malloc ((n+1) * sizeof (Int)) ;
        Int *InvPost = malloc ((n+1) * sizeof (Int)) ;
        if (postorder (n, L->Parent, L->ColCount, Post) == n)
        {
            for (Int k = 0 ; k < n ; k++) W [k] = L->Perm [Post [k]] ;
            memcpy (L->Perm, W, n * sizeof (Int)) ;
            for (Int k = 0 ; k < n ; k++) W [k] = L->ColCount [Post [k]] ;
            memcpy (L->ColCount, W, n * sizeof (Int)) ;
            for (Int k = 0 ; k < n ; k++) InvPost [Post [k]] = k ;
            for (Int c = 0 ; c < n ; c++)
            {
                Int op = L->Parent [Post [c]] ;
                W [c] = (op == EMPTY) ? EMPTY : InvPost [op] ;
            }
            memcpy (L->Parent, W, n * sizeof (Int)) ;
        }
        free (Post) ; free (W) ; free (InvPost) ;
        free (Up) ; free (Ui) ;
        if (!sym_permute (n, Ap, Ai, NULL, stype, L->Perm, 1, &Up, &Ui, NULL))
        { orc_free (L) ; return NULL ; }
    }
    L->status = super_symbolic (L, Up, Ui, nrelax, zrelax) ;
    free (Up) ; free (Ui) ;
    return L ;
}

/* ------------------------------------------------------------------------ */
/* public: numeric factorization (left-looking, as the reference)            */
/* ------------------------------------------------------------------------ */

/* reference: Supernodal/cholmod_super_numeric.c:97-308 (wrapper) and
 * Supernodal/t_cholmod_super_numeric.c:93-1079 (loop), real case, stype<0
 * input after the caller's permuted transpose (cholmod_factorize.c:225-244).
 * A is the ORIGINAL symmetric matrix (stype triangle stored); the permutation
 * by L->Perm is applied here.  Returns status (0 ok, 1 not posdef, <0 error);
 * like the reference, not-posdef is a successful return with L->minor set. */
int orc_factorize (orc_factor *L, const Int *Ap, const Int *Ai,
    const double *Ax, int stype, double beta, int quick_return_if_not_posdef)
{
    Int n = L->n, nsuper = L->nsuper ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s ;
    Int *Sp = NULL, *Si = NULL ;
    double *Sx = NULL ;
    if (!sym_permute (n, Ap, Ai, Ax, stype, L->Perm, 0, &Sp, &Si, &Sx))
        return (L->status = ORC_OUT_OF_MEMORY) ;
    if (!L->x) L->x = malloc (L->xsize * sizeof (double)) ;
    double *C = malloc ((L->maxcsize > 0 ? L->maxcsize : 1) * sizeof (double)) ;
    Int *SuperMap = malloc ((n+1) * sizeof (Int)) ;
    Int *RelativeMap = malloc ((n+1) * sizeof (Int)) ;
    Int *Map = malloc ((n+1) * sizeof (Int)) ;
    Int *Next = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Lpos = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Next_save = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Lpos_save = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Head = malloc ((nsuper+1) * sizeof (Int)) ;
    if (!L->x || !C || !SuperMap || !RelativeMap || !Map || !Next || !Lpos
        || !Next_save || !Lpos_save || !Head)
        return (L->status = ORC_OUT_OF_MEMORY) ;
    double *Lx = L->x ;
    L->minor = n ;
    L->status = ORC_OK ;
    L->exec_flops = 0 ;
    for (int t = 0 ; t < 4 ; t++) L->calls [t] = 0 ;
    for (Int s = 0 ; s < nsuper ; s++)                      /* wrapper :266 */
        for (Int k = Super [s] ; k < Super [s+1] ; k++) SuperMap [k] = s ;
    for (Int s = 0 ; s < nsuper ; s++) Head [s] = EMPTY ;
    for (Int i = 0 ; i < n ; i++) Map [i] = EMPTY ;

    int repeat_supernode = 0 ;
    Int nscol_new = 0 ;
    for (Int s = 0 ; s < nsuper ; s++)                      /* loop :279 */
    {
        Int k1 = Super [s], k2 = Super [s+1] ;
        Int nscol = k2 - k1 ;
        Int psi = Lpi [s], psend = Lpi [s+1], psx = Lpx [s] ;
        Int nsrow = psend - psi ;
        Int pend = psx + nsrow * nscol ;
        for (Int p = psx ; p < pend ; p++) Lx [p] = 0 ;     /* :305-317 */
        for (Int k = 0 ; k < nsrow ; k++) Map [Ls [psi + k]] = k ; /* :326 */
        /* assemble A(:,k1:k2-1), lower part, ASSIGN semantics, :353-418 */
        for (Int k = k1 ; k < k2 ; k++)
        {
            for (Int p = Sp [k] ; p < Sp [k+1] ; p++)
            {
                Int i = Si [p] ;
                if (i >= k)
                {
                    Int imap = Map [i] ;
                    if (imap >= 0 && imap < nsrow)
                        Lx [imap + (psx + (k-k1)*nsrow)] = Sx [p] ;
                }
            }
        }
        if (beta != 0.0)                                    /* :421-431 */
        {
            Int pk = psx ;
            for (Int k = k1 ; k < k2 ; k++) { Lx [pk] += beta ; pk += nsrow + 1 ; }
        }
        /* save/restore the descendant lists for a repeated supernode :442-460 */
        if (!repeat_supernode)
        {
            for (Int d = Head [s] ; d != EMPTY ; d = Next [d])
            {
                Lpos_save [d] = Lpos [d] ;
                Next_save [d] = Next [d] ;
            }
        }
        else
        {
            for (Int d = Head [s] ; d != EMPTY ; d = Next [d])
            {
                Lpos [d] = Lpos_save [d] ;
                Next [d] = Next_save [d] ;
            }
        }
        /* descendant updates, :498-810 */
        Int dnext ;
        for (Int d = Head [s] ; d != EMPTY ; d = dnext)
        {
            Int kd1 = Super [d], kd2 = Super [d+1] ;
            Int ndcol = kd2 - kd1 ;
            Int pdi = Lpi [d], pdend = Lpi [d+1], pdx = Lpx [d] ;
            Int ndrow = pdend - pdi ;
            Int p = Lpos [d] ;
            Int pdi1 = pdi + p ;
            Int pdx1 = pdx + p ;
            Int pdi2 ;
            for (pdi2 = pdi1 ; pdi2 < pdend && Ls [pdi2] < k2 ; pdi2++) ;
            Int ndrow1 = pdi2 - pdi1 ;
            Int ndrow2 = pdend - pdi1 ;
            Int ndrow3 = ndrow2 - ndrow1 ;
            dnext = Next [d] ;
            /* C1 = L1*L1' (lower), C2 = L2*L1' ; :682-717 */
            d_syrk (ndrow1, ndcol, Lx + pdx1, ndrow, C, ndrow2) ;
            L->calls [0]++ ;
            L->exec_flops += (double) ndrow1 * ndrow1 * ndcol ;
            if (ndrow3 > 0)
            {
                d_gemm (ndrow3, ndrow1, ndcol, Lx + pdx1 + ndrow1, ndrow,
                    Lx + pdx1, ndrow, C + ndrow1, ndrow2) ;
                L->calls [1]++ ;
                L->exec_flops += 2.0 * ndrow3 * ndrow1 * ndcol ;
            }
            for (Int i = 0 ; i < ndrow2 ; i++)              /* :743-750 */
                RelativeMap [i] = Map [Ls [pdi1 + i]] ;
            for (Int j = 0 ; j < ndrow1 ; j++)              /* :756-772 */
            {
                Int px = psx + RelativeMap [j] * nsrow ;
                for (Int i = j ; i < ndrow2 ; i++)
                    Lx [px + RelativeMap [i]] -= C [i + ndrow2*j] ;
            }
            /* advance d to its next ancestor, :791-808 */
            Lpos [d] = pdi2 - pdi ;
            if (Lpos [d] < ndrow)
            {
                Int dancestor = SuperMap [Ls [pdi2]] ;
                Next [d] = Head [dancestor] ;
                Head [dancestor] = d ;
            }
        }
        /* factorize the diagonal block, :825-867 */
        Int nscol2 = repeat_supernode ? nscol_new : nscol ;
        Int info = d_potrf (nscol2, Lx + psx, nsrow) ;
        L->calls [2]++ ;
        L->exec_flops += (double) nscol2 * nscol2 * nscol2 / 3.0 ;
        if (repeat_supernode)                               /* :883-896 */
        {
            info = 0 ;
            Int p = psx + nsrow * nscol_new ;
            for ( ; p < psx + nsrow * nscol ; p++) Lx [p] = 0 ;
        }
        if (info != 0)                                      /* :905-968 */
        {
            L->status = ORC_NOT_POSDEF ;
            L->minor = k1 + info - 1 ;
            for (Int ss = s+1 ; ss < nsuper ; ss++) Head [ss] = EMPTY ;
            for (Int p = psx ; p < L->xsize ; p++) Lx [p] = 0 ;
            if (info == 1 || quick_return_if_not_posdef)
            {
                Head [s] = EMPTY ;
                goto done ;
            }
            repeat_supernode = 1 ;
            nscol_new = info - 1 ;
            s-- ;
            continue ;
        }
        /* L2 = S2 / L1', :997-1002 */
        Int nsrow2 = nsrow - nscol2 ;
        if (nsrow2 > 0)
        {
            d_trsm (nsrow2, nscol2, Lx + psx, nsrow, Lx + psx + nscol2, nsrow) ;
            L->calls [3]++ ;
            L->exec_flops += (double) nsrow2 * nscol2 * nscol2 ;
            /* with nscol2 < nscol (repeat) the solve covers rows nscol2..nsrow-1
             * of the leading nscol2 columns, as the reference does */
            if (!repeat_supernode)                          /* :1021-1034 */
            {
                Lpos [s] = nscol ;
                Int sparent = SuperMap [Ls [psi + nscol]] ;
                Next [s] = Head [sparent] ;
                Head [sparent] = s ;
            }
        }
        Head [s] = EMPTY ;
        if (repeat_supernode) goto done ;                   /* :1052-1064 */
    }
done:
    free (C) ; free (SuperMap) ; free (RelativeMap) ; free (Map) ; free (Next) ;
    free (Lpos) ; free (Next_save) ; free (Lpos_save) ; free (Head) ;
    free (Sp) ; free (Si) ; free (Sx) ;
    return L->status ;
}

/* ------------------------------------------------------------------------ */
/* public: triangular solves                                                  */
/* ------------------------------------------------------------------------ */

/* Forward solve L y = b in place on X (n-by-nrhs, leading dimension n).
 * reference: Supernodal/t_cholmod_super_solve.c:14-220 (gather E = X[Ls],
 * dtrsv/dgemv for nrhs==1 :89-102, dtrsm/dgemm otherwise :164-181, scatter). */
void orc_lsolve (const orc_factor *L, double *X, Int nrhs)
{
    Int n = L->n ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s ;
    const double *Lx = L->x ;
    double *E = malloc ((L->maxesize > 0 ? L->maxesize : 1) * sizeof (double)) ;
    for (Int r = 0 ; r < nrhs ; r++)
    {
        double *x = X + r*n ;
        for (Int s = 0 ; s < L->nsuper ; s++)
        {
            Int k1 = Super [s], nscol = Super [s+1] - k1 ;
            Int psi = Lpi [s], nsrow = Lpi [s+1] - psi, psx = Lpx [s] ;
            Int nsrow2 = nsrow - nscol, ps2 = psi + nscol ;
            for (Int ii = 0 ; ii < nsrow2 ; ii++) E [ii] = x [Ls [ps2 + ii]] ;
            for (Int j = 0 ; j < nscol ; j++)           /* dtrsv L,N,N */
            {
                double v = x [k1 + j] / Lx [psx + j + j*nsrow] ;
                x [k1 + j] = v ;
                for (Int i = j+1 ; i < nscol ; i++)
                    x [k1 + i] -= Lx [psx + i + j*nsrow] * v ;
            }
            for (Int j = 0 ; j < nscol ; j++)           /* dgemv N, alpha=-1 */
            {
                double v = x [k1 + j] ;
                const double *col = Lx + psx + nscol + j*nsrow ;
                for (Int ii = 0 ; ii < nsrow2 ; ii++) E [ii] -= col [ii] * v ;
            }
            for (Int ii = 0 ; ii < nsrow2 ; ii++) x [Ls [ps2 + ii]] = E [ii] ;
        }
    }
    free (E) ;
}

/* Backward solve L' x = y in place.
 * reference: Supernodal/t_cholmod_super_solve.c:222-411 (dgemv "C" then
 * dtrsv "L","C","N" :297-310; reverse supernode order). */
void orc_ltsolve (const orc_factor *L, double *X, Int nrhs)
{
    Int n = L->n ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s ;
    const double *Lx = L->x ;
    double *E = malloc ((L->maxesize > 0 ? L->maxesize : 1) * sizeof (double)) ;
    for (Int r = 0 ; r < nrhs ; r++)
    {
        double *x = X + r*n ;
        for (Int s = L->nsuper - 1 ; s >= 0 ; s--)
        {
            Int k1 = Super [s], nscol = Super [s+1] - k1 ;
            Int psi = Lpi [s], nsrow = Lpi [s+1] - psi, psx = Lpx [s] ;
            Int nsrow2 = nsrow - nscol, ps2 = psi + nscol ;
            for (Int ii = 0 ; ii < nsrow2 ; ii++) E [ii] = x [Ls [ps2 + ii]] ;
            for (Int j = 0 ; j < nscol ; j++)           /* x1 -= L2' E */
            {
                const double *col = Lx + psx + nscol + j*nsrow ;
                double acc = 0 ;
                for (Int ii = 0 ; ii < nsrow2 ; ii++) acc += col [ii] * E [ii] ;
                x [k1 + j] -= acc ;
            }
            for (Int j = nscol - 1 ; j >= 0 ; j--)      /* dtrsv L,C,N */
            {
                double v = x [k1 + j] ;
                for (Int i = j+1 ; i < nscol ; i++)
                    v -= Lx [psx + i + j*nsrow] * x [k1 + i] ;
                x [k1 + j] = v / Lx [psx + j + j*nsrow] ;
            }
        }
    }
    free (E) ;
}

/* Solve A x = b with the supernodal factor: x = P' L'^-1 L^-1 P b.
 * reference: Cholesky/cholmod_solve.c:1541-1580 (perm :105, iperm :322). */
void orc_solve (const orc_factor *L, const double *B, double *X, Int nrhs)
{
    Int n = L->n ;
    double *Y = malloc ((n*nrhs > 0 ? n*nrhs : 1) * sizeof (double)) ;
    for (Int r = 0 ; r < nrhs ; r++)
        for (Int k = 0 ; k < n ; k++) Y [k + r*n] = B [L->Perm [k] + r*n] ;
    orc_lsolve (L, Y, nrhs) ;
    orc_ltsolve (L, Y, nrhs) ;
    for (Int r = 0 ; r < nrhs ; r++)
        for (Int k = 0 ; k < n ; k++) X [L->Perm [k] + r*n] = Y [k + r*n] ;
    free (Y) ;
}

/* ------------------------------------------------------------------------ */
/* complex / zomplex input: L and C are complex (interleaved)                 */
/* ------------------------------------------------------------------------ */
/* reference: Supernodal/t_cholmod_super_numeric.c:41-83 (L_ENTRY = 2, the
 * L_ASSIGN / L_ASSEMBLE / L_ASSEMBLESUB macros of the complex and zomplex
 * templates), BLAS_zherk / BLAS_zgemm :682-717, LAPACK_zpotrf :864-867,
 * BLAS_ztrsm :997-1002; Supernodal/t_cholmod_super_solve.c with ztrsv/zgemv. */

typedef void (*zgemm_f) (const char *, const char *, const int *, const int *,
    const int *, const double *, const double *, const int *, const double *,
    const int *, const double *, double *, const int *) ;
typedef void (*zherk_f) (const char *, const char *, const int *, const int *,
    const double *, const double *, const int *, const double *, double *,
    const int *) ;
typedef void (*ztrsm_f) (const char *, const char *, const char *, const char *,
    const int *, const int *, const double *, const double *, const int *,
    double *, const int *) ;
typedef void (*zpotrf_f) (const char *, const int *, double *, const int *, int *) ;
static zgemm_f  x_zgemm  = NULL ;
static zherk_f  x_zherk  = NULL ;
static ztrsm_f  x_ztrsm  = NULL ;
static zpotrf_f x_zpotrf = NULL ;

int orc_bind_blas_complex (const char *path, const char *prefix)
{
    void *h = dlopen (path, RTLD_NOW | RTLD_LOCAL) ;
    if (!h) return 0 ;
    char name [256] ;
    snprintf (name, sizeof name, "%szgemm_", prefix) ;
    zgemm_f g = (zgemm_f) dlsym (h, name) ;
    snprintf (name, sizeof name, "%szherk_", prefix) ;
    zherk_f s = (zherk_f) dlsym (h, name) ;
    snprintf (name, sizeof name, "%sztrsm_", prefix) ;
    ztrsm_f t = (ztrsm_f) dlsym (h, name) ;
    snprintf (name, sizeof name, "%szpotrf_", prefix) ;
    zpotrf_f q = (zpotrf_f) dlsym (h, name) ;
    if (!g || !s || !t || !q) return 0 ;
    x_zgemm = g ; x_zherk = s ; x_ztrsm = t ; x_zpotrf = q ;
    return 1 ;
}

/* C(m,n) = A(m,k) * B(n,k)^H, complex interleaved (zgemm "N","C") */
static void z_gemm (Int m, Int n, Int k, const double *A, Int lda,
    const double *B, Int ldb, double *C, Int ldc)
{
    if (x_zgemm)
    {
        int M = (int) m, N = (int) n, K = (int) k, LDA = (int) lda,
            LDB = (int) ldb, LDC = (int) ldc ;
        double one [2] = {1, 0}, zero [2] = {0, 0} ;
        x_zgemm ("N", "C", &M, &N, &K, one, A, &LDA, B, &LDB, zero, C, &LDC) ;
        return ;
    }
    for (Int j = 0 ; j < n ; j++)
        for (Int i = 0 ; i < m ; i++)
        {
            double cr = 0, ci = 0 ;
            for (Int l = 0 ; l < k ; l++)
            {
                double ar = A [2*(i + l*lda)], ai = A [2*(i + l*lda)+1] ;
                double br = B [2*(j + l*ldb)], bi = B [2*(j + l*ldb)+1] ;
                cr += ar * br + ai * bi ;
                ci += ai * br - ar * bi ;
            }
            C [2*(i + j*ldc)] = cr ; C [2*(i + j*ldc)+1] = ci ;
        }
}

/* lower triangle of C(n,n) = A(n,k) * A(n,k)^H (zherk "L","N") */
static void z_herk (Int n, Int k, const double *A, Int lda, double *C, Int ldc)
{
    if (x_zherk)
    {
        int N = (int) n, K = (int) k, LDA = (int) lda, LDC = (int) ldc ;
        double one = 1, zero = 0 ;
        x_zherk ("L", "N", &N, &K, &one, A, &LDA, &zero, C, &LDC) ;
        return ;
    }
    for (Int j = 0 ; j < n ; j++)
        z_gemm (n - j, 1, k, A + 2*j, lda, A + 2*j, lda, C + 2*(j + j*ldc), ldc) ;
    for (Int j = 0 ; j < n ; j++) C [2*(j + j*ldc)+1] = 0 ;
}

/* zpotrf "L": info as LAPACK (first pivot <= 0, imaginary parts of the diagonal ignored) */
static Int z_potrf (Int n, double *A, Int lda)
{
    if (x_zpotrf)
    {
        int N = (int) n, LDA = (int) lda, info = 0 ;
        x_zpotrf ("L", &N, A, &LDA, &info) ;
        return info ;
    }
    for (Int j = 0 ; j < n ; j++)
    {
        double ajj = A [2*(j + j*lda)] ;
        for (Int l = 0 ; l < j ; l++)
        {
            double r = A [2*(j + l*lda)], im = A [2*(j + l*lda)+1] ;
            ajj -= r * r + im * im ;
        }
        if (ajj <= 0.0) { A [2*(j + j*lda)] = ajj ; return j + 1 ; }
        ajj = sqrt (ajj) ;
        A [2*(j + j*lda)] = ajj ; A [2*(j + j*lda)+1] = 0 ;
        for (Int i = j + 1 ; i < n ; i++)
        {
            double vr = A [2*(i + j*lda)], vi = A [2*(i + j*lda)+1] ;
            for (Int l = 0 ; l < j ; l++)
            {
                double ar = A [2*(i + l*lda)], ai = A [2*(i + l*lda)+1] ;
                double br = A [2*(j + l*lda)], bi = A [2*(j + l*lda)+1] ;
                vr -= ar * br + ai * bi ;
                vi -= ai * br - ar * bi ;
            }
            A [2*(i + j*lda)] = vr / ajj ; A [2*(i + j*lda)+1] = vi / ajj ;
        }
    }
    return 0 ;
}

/* B(m,n) := B * inv(L)^H, L(n,n) lower non-unit (ztrsm "R","L","C","N") */
static void z_trsm (Int m, Int n, const double *L, Int ldl, double *B, Int ldb)
{
    if (x_ztrsm)
    {
        int M = (int) m, N = (int) n, LDL = (int) ldl, LDB = (int) ldb ;
        double one [2] = {1, 0} ;
        x_ztrsm ("R", "L", "C", "N", &M, &N, one, L, &LDL, B, &LDB) ;
        return ;
    }
    for (Int jj = 0 ; jj < n ; jj++)
    {
        double *bj = B + 2*jj*ldb ;
        for (Int l = 0 ; l < jj ; l++)
        {
            double fr = L [2*(jj + l*ldl)], fi = -L [2*(jj + l*ldl)+1] ;   /* conj (L(jj,l)) */
            const double *bl = B + 2*l*ldb ;
            for (Int i = 0 ; i < m ; i++)
            {
                bj [2*i]   -= bl [2*i] * fr - bl [2*i+1] * fi ;
                bj [2*i+1] -= bl [2*i] * fi + bl [2*i+1] * fr ;
            }
        }
        double d = L [2*(jj + jj*ldl)] ;
        for (Int i = 0 ; i < m ; i++) { bj [2*i] /= d ; bj [2*i+1] /= d ; }
    }
}

/* As orc_factorize, for a Hermitian A with complex (Az == NULL: Ax interleaved)
 * or zomplex (Ax real parts, Az imaginary parts) values; L->x is complex
 * interleaved, 2*xsize doubles (t_cholmod_super_numeric.c:41-83: "A and F are
 * complex or zomplex, L and C are complex").  beta is real (:421-431, b[0]). */
int orc_factorize_complex (orc_factor *L, const Int *Ap, const Int *Ai,
    const double *Ax, const double *Az, int stype, double beta,
    int quick_return_if_not_posdef)
{
    Int n = L->n, nsuper = L->nsuper ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s ;
    Int *Sp = NULL, *Si = NULL ;
    double *Sx = NULL, *Axz = NULL ;
    if (Az)
    {
        Int nz = Ap [n] ;
        Axz = malloc ((nz > 0 ? nz : 1) * 2 * sizeof (double)) ;
        if (!Axz) return (L->status = ORC_OUT_OF_MEMORY) ;
        for (Int p = 0 ; p < nz ; p++) { Axz [2*p] = Ax [p] ; Axz [2*p+1] = Az [p] ; }
        Ax = Axz ;
    }
    if (!sym_permute_e (n, Ap, Ai, Ax, stype, L->Perm, 0, &Sp, &Si, &Sx, 2))
        return (L->status = ORC_OUT_OF_MEMORY) ;
    free (Axz) ;
    free (L->x) ;
    L->x = malloc ((L->xsize > 0 ? L->xsize : 1) * 2 * sizeof (double)) ;
    double *C = malloc ((L->maxcsize > 0 ? L->maxcsize : 1) * 2 * sizeof (double)) ;
    Int *SuperMap = malloc ((n+1) * sizeof (Int)) ;
    Int *RelativeMap = malloc ((n+1) * sizeof (Int)) ;
    Int *Map = malloc ((n+1) * sizeof (Int)) ;
    Int *Next = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Lpos = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Next_save = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Lpos_save = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Head = malloc ((nsuper+1) * sizeof (Int)) ;
    if (!L->x || !C || !SuperMap || !RelativeMap || !Map || !Next || !Lpos
        || !Next_save || !Lpos_save || !Head)
        return (L->status = ORC_OUT_OF_MEMORY) ;
    double *Lx = L->x ;
    L->minor = n ;
    L->status = ORC_OK ;
    L->exec_flops = 0 ;
    for (int t = 0 ; t < 4 ; t++) L->calls [t] = 0 ;
    for (Int s = 0 ; s < nsuper ; s++)
        for (Int k = Super [s] ; k < Super [s+1] ; k++) SuperMap [k] = s ;
    for (Int s = 0 ; s < nsuper ; s++) Head [s] = EMPTY ;
    for (Int i = 0 ; i < n ; i++) Map [i] = EMPTY ;

    int repeat_supernode = 0 ;
    Int nscol_new = 0 ;
    for (Int s = 0 ; s < nsuper ; s++)
    {
        Int k1 = Super [s], k2 = Super [s+1] ;
        Int nscol = k2 - k1 ;
        Int psi = Lpi [s], psend = Lpi [s+1], psx = Lpx [s] ;
        Int nsrow = psend - psi ;
        Int pend = psx + nsrow * nscol ;
        for (Int p = 2*psx ; p < 2*pend ; p++) Lx [p] = 0 ;             /* L_CLEAR */
        for (Int k = 0 ; k < nsrow ; k++) Map [Ls [psi + k]] = k ;
        for (Int k = k1 ; k < k2 ; k++)
            for (Int p = Sp [k] ; p < Sp [k+1] ; p++)
            {
                Int i = Si [p] ;
                if (i >= k)
                {
                    Int imap = Map [i] ;
                    if (imap >= 0 && imap < nsrow)
                    {
                        Int q = imap + (psx + (k-k1)*nsrow) ;            /* L_ASSIGN */
                        Lx [2*q] = Sx [2*p] ; Lx [2*q+1] = Sx [2*p+1] ;
                    }
                }
            }
        if (beta != 0.0)
        {
            Int pk = psx ;
            for (Int k = k1 ; k < k2 ; k++) { Lx [2*pk] += beta ; pk += nsrow + 1 ; }   /* L_ASSEMBLE */
        }
        if (!repeat_supernode)
            for (Int d = Head [s] ; d != EMPTY ; d = Next [d])
            { Lpos_save [d] = Lpos [d] ; Next_save [d] = Next [d] ; }
        else
            for (Int d = Head [s] ; d != EMPTY ; d = Next [d])
            { Lpos [d] = Lpos_save [d] ; Next [d] = Next_save [d] ; }
        Int dnext ;
        for (Int d = Head [s] ; d != EMPTY ; d = dnext)
        {
            Int kd1 = Super [d], kd2 = Super [d+1] ;
            Int ndcol = kd2 - kd1 ;
            Int pdi = Lpi [d], pdend = Lpi [d+1], pdx = Lpx [d] ;
            Int ndrow = pdend - pdi ;
            Int p = Lpos [d] ;
            Int pdi1 = pdi + p ;
            Int pdx1 = pdx + p ;
            Int pdi2 ;
            for (pdi2 = pdi1 ; pdi2 < pdend && Ls [pdi2] < k2 ; pdi2++) ;
            Int ndrow1 = pdi2 - pdi1 ;
            Int ndrow2 = pdend - pdi1 ;
            Int ndrow3 = ndrow2 - ndrow1 ;
            dnext = Next [d] ;
            z_herk (ndrow1, ndcol, Lx + 2*pdx1, ndrow, C, ndrow2) ;
            L->calls [0]++ ;
            L->exec_flops += 4.0 * ndrow1 * ndrow1 * ndcol ;
            if (ndrow3 > 0)
            {
                z_gemm (ndrow3, ndrow1, ndcol, Lx + 2*(pdx1 + ndrow1), ndrow,
                    Lx + 2*pdx1, ndrow, C + 2*ndrow1, ndrow2) ;
                L->calls [1]++ ;
                L->exec_flops += 8.0 * ndrow3 * ndrow1 * ndcol ;
            }
            for (Int i = 0 ; i < ndrow2 ; i++)
                RelativeMap [i] = Map [Ls [pdi1 + i]] ;
            for (Int j = 0 ; j < ndrow1 ; j++)
            {
                Int px = psx + RelativeMap [j] * nsrow ;
                for (Int i = j ; i < ndrow2 ; i++)
                {
                    Int q = px + RelativeMap [i] ;                       /* L_ASSEMBLESUB */
                    Lx [2*q]   -= C [2*(i + ndrow2*j)] ;
                    Lx [2*q+1] -= C [2*(i + ndrow2*j)+1] ;
                }
            }
            Lpos [d] = pdi2 - pdi ;
            if (Lpos [d] < ndrow)
            {
                Int dancestor = SuperMap [Ls [pdi2]] ;
                Next [d] = Head [dancestor] ;
                Head [dancestor] = d ;
            }
        }
        Int nscol2 = repeat_supernode ? nscol_new : nscol ;
        Int info = z_potrf (nscol2, Lx + 2*psx, nsrow) ;
        L->calls [2]++ ;
        L->exec_flops += 4.0 * nscol2 * nscol2 * nscol2 / 3.0 ;
        if (repeat_supernode)
        {
            info = 0 ;
            for (Int p = 2*(psx + nsrow * nscol_new) ; p < 2*(psx + nsrow * nscol) ; p++) Lx [p] = 0 ;
        }
        if (info != 0)
        {
            L->status = ORC_NOT_POSDEF ;
            L->minor = k1 + info - 1 ;
            for (Int ss = s+1 ; ss < nsuper ; ss++) Head [ss] = EMPTY ;
            for (Int p = 2*psx ; p < 2*L->xsize ; p++) Lx [p] = 0 ;
            if (info == 1 || quick_return_if_not_posdef) { Head [s] = EMPTY ; goto done ; }
            repeat_supernode = 1 ;
            nscol_new = info - 1 ;
            s-- ;
            continue ;
        }
        Int nsrow2 = nsrow - nscol2 ;
        if (nsrow2 > 0)
        {
            z_trsm (nsrow2, nscol2, Lx + 2*psx, nsrow, Lx + 2*(psx + nscol2), nsrow) ;
            L->calls [3]++ ;
            L->exec_flops += 4.0 * nsrow2 * nscol2 * nscol2 ;
            if (!repeat_supernode)
            {
                Lpos [s] = nscol ;
                Int sparent = SuperMap [Ls [psi + nscol]] ;
                Next [s] = Head [sparent] ;
                Head [sparent] = s ;
            }
        }
        Head [s] = EMPTY ;
        if (repeat_supernode) goto done ;
    }
done:
    free (C) ; free (SuperMap) ; free (RelativeMap) ; free (Map) ; free (Next) ;
    free (Lpos) ; free (Next_save) ; free (Lpos_save) ; free (Head) ;
    free (Sp) ; free (Si) ; free (Sx) ;
    return L->status ;
}

/* x = P^T L^-H L^-1 P b, complex interleaved B and X (n-by-nrhs, ld n).
 * reference: Supernodal/t_cholmod_super_solve.c (complex template: ztrsv
 * "L","N","N" + zgemv "N" forward, zgemv "C" + ztrsv "L","C","N" backward). */
void orc_solve_complex (const orc_factor *L, const double *B, double *X, Int nrhs)
{
    Int n = L->n ;
    const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px, *Ls = L->s ;
    const double *Lx = L->x ;
    double *Y = malloc ((n > 0 ? n : 1) * 2 * sizeof (double)) ;
    for (Int r = 0 ; r < nrhs ; r++)
    {
        for (Int k = 0 ; k < n ; k++)
        { Y [2*k] = B [2*(L->Perm [k] + r*n)] ; Y [2*k+1] = B [2*(L->Perm [k] + r*n)+1] ; }
        for (Int s = 0 ; s < L->nsuper ; s++)
        {
            Int k1 = Super [s], nscol = Super [s+1] - k1 ;
            Int psi = Lpi [s], nsrow = Lpi [s+1] - psi, psx = Lpx [s] ;
            for (Int j = 0 ; j < nscol ; j++)
            {
                double d = Lx [2*(psx + j + j*nsrow)] ;
                double vr = Y [2*(k1+j)] / d, vi = Y [2*(k1+j)+1] / d ;
                Y [2*(k1+j)] = vr ; Y [2*(k1+j)+1] = vi ;
                for (Int i = j+1 ; i < nsrow ; i++)
                {
                    double lr = Lx [2*(psx + i + j*nsrow)], li = Lx [2*(psx + i + j*nsrow)+1] ;
                    Int row = Ls [psi + i] ;
                    Y [2*row]   -= lr * vr - li * vi ;
                    Y [2*row+1] -= lr * vi + li * vr ;
                }
            }
        }
        for (Int s = L->nsuper - 1 ; s >= 0 ; s--)
        {
            Int k1 = Super [s], nscol = Super [s+1] - k1 ;
            Int psi = Lpi [s], nsrow = Lpi [s+1] - psi, psx = Lpx [s] ;
            for (Int j = nscol - 1 ; j >= 0 ; j--)
            {
                double vr = Y [2*(k1+j)], vi = Y [2*(k1+j)+1] ;
                for (Int i = j+1 ; i < nsrow ; i++)
                {
                    double lr = Lx [2*(psx + i + j*nsrow)], li = -Lx [2*(psx + i + j*nsrow)+1] ;
                    Int row = Ls [psi + i] ;
                    vr -= lr * Y [2*row] - li * Y [2*row+1] ;
                    vi -= lr * Y [2*row+1] + li * Y [2*row] ;
                }
                double d = Lx [2*(psx + j + j*nsrow)] ;
                Y [2*(k1+j)] = vr / d ; Y [2*(k1+j)+1] = vi / d ;
            }
        }
        for (Int k = 0 ; k < n ; k++)
        { X [2*(L->Perm [k] + r*n)] = Y [2*k] ; X [2*(L->Perm [k] + r*n)+1] = Y [2*k+1] ; }
    }
    free (Y) ;
}

/* ------------------------------------------------------------------------ */
/* public: derived maps and statistics                                        */
/* ------------------------------------------------------------------------ */

/* Supernodal etree parent: Sparent[s] = SuperMap[Ls[pi[s]+nscol]] or EMPTY,
 * exactly what the reference recomputes at t_cholmod_super_numeric.c:1025. */
void orc_sparent (const orc_factor *L, Int *Sparent)
{
    Int *SuperMap = malloc ((L->n + 1) * sizeof (Int)) ;
    for (Int s = 0 ; s < L->nsuper ; s++)
        for (Int k = L->super [s] ; k < L->super [s+1] ; k++) SuperMap [k] = s ;
    for (Int s = 0 ; s < L->nsuper ; s++)
    {
        Int nscol = L->super [s+1] - L->super [s] ;
        Int nsrow = L->pi [s+1] - L->pi [s] ;
        Sparent [s] = (nsrow > nscol) ? SuperMap [L->s [L->pi [s] + nscol]] : EMPTY ;
    }
    free (SuperMap) ;
}

/* Relative row map of every supernode d against its parent p: for the rows
 * below d's diagonal block, RelMap[pi[d]-Super[d] + i] ... stored compactly:
 * out[ off[d] + i ] = Map_p[ Ls[pi[d]+nscol_d+i] ], off[d] = pi[d]-super[d].
 * This is the reference's RelativeMap (t_cholmod_super_numeric.c:743-750) at
 * the first update of d (target = parent, Lpos[d]=nscol, ndrow2 = all rows
 * below the diagonal block).  Output size ssize - n. */
void orc_relmap_to_parent (const orc_factor *L, Int *out)
{
    Int n = L->n ;
    Int *Map = malloc ((n+1) * sizeof (Int)) ;
    Int *Sparent = malloc ((L->nsuper + 1) * sizeof (Int)) ;
    orc_sparent (L, Sparent) ;
    /* process children grouped by parent so Map is built once per parent */
    Int *head = malloc ((L->nsuper + 1) * sizeof (Int)) ;
    Int *next = malloc ((L->nsuper + 1) * sizeof (Int)) ;
    for (Int s = 0 ; s < L->nsuper ; s++) head [s] = EMPTY ;
    for (Int s = L->nsuper - 1 ; s >= 0 ; s--)
        if (Sparent [s] != EMPTY) { next [s] = head [Sparent [s]] ; head [Sparent [s]] = s ; }
    for (Int p = 0 ; p < L->nsuper ; p++)
    {
        if (head [p] == EMPTY) continue ;
        Int psi = L->pi [p], nsrow = L->pi [p+1] - psi ;
        for (Int k = 0 ; k < nsrow ; k++) Map [L->s [psi + k]] = k ;
        for (Int d = head [p] ; d != EMPTY ; d = next [d])
        {
            Int nscol = L->super [d+1] - L->super [d] ;
            Int pd = L->pi [d] + nscol, pe = L->pi [d+1] ;
            Int off = L->pi [d] - L->super [d] ;
            for (Int q = pd ; q < pe ; q++) out [off + (q - pd)] = Map [L->s [q]] ;
        }
    }
    free (Map) ; free (Sparent) ; free (head) ; free (next) ;
}

/* Update statistics per SURVEY.md appendix D: number of (d,s) updates, update
 * flops, scatter elements, panel-read elements, panel (potrf+trsm) flops. */
void orc_update_stats (const orc_factor *L, double *out5)
{
    Int *SuperMap = malloc ((L->n + 1) * sizeof (Int)) ;
    for (Int s = 0 ; s < L->nsuper ; s++)
        for (Int k = L->super [s] ; k < L->super [s+1] ; k++) SuperMap [k] = s ;
    double nupd = 0, uflops = 0, scat = 0, pread = 0, pflops = 0 ;
    for (Int d = 0 ; d < L->nsuper ; d++)
    {
        double ndcol = (double) (L->super [d+1] - L->super [d]) ;
        Int p0 = L->pi [d] + (Int) ndcol, pend = L->pi [d+1] ;
        double nsrow = (double) (pend - L->pi [d]) ;
        pflops += ndcol*ndcol*ndcol/3.0 + (nsrow - ndcol)*ndcol*ndcol ;
        Int p = p0 ;
        while (p < pend)
        {
            Int s = SuperMap [L->s [p]] ;
            Int q = p ;
            while (q < pend && SuperMap [L->s [q]] == s) q++ ;
            double n1 = (double) (q - p), n2 = (double) (pend - p) ;
            nupd += 1 ;
            uflops += n1*n1*ndcol + 2.0*(n2 - n1)*n1*ndcol ;
            scat += n1*n2 - n1*(n1 - 1)/2 ;
            pread += n2*ndcol ;
            p = q ;
        }
    }
    out5 [0] = nupd ; out5 [1] = uflops ; out5 [2] = scat ; out5 [3] = pread ;
    out5 [4] = pflops ;
    free (SuperMap) ;
}

/* ---- plain getters for ctypes ------------------------------------------ */
Int orc_n (const orc_factor *L) { return L->n ; }
Int orc_nsuper (const orc_factor *L) { return L->nsuper ; }
Int orc_ssize (const orc_factor *L) { return L->ssize ; }
Int orc_xsize (const orc_factor *L) { return L->xsize ; }
Int orc_maxcsize (const orc_factor *L) { return L->maxcsize ; }
Int orc_maxesize (const orc_factor *L) { return L->maxesize ; }
Int orc_minor (const orc_factor *L) { return L->minor ; }
int orc_status (const orc_factor *L) { return L->status ; }
double orc_fl (const orc_factor *L) { return L->fl ; }
double orc_lnz (const orc_factor *L) { return L->lnz ; }
double orc_exec_flops (const orc_factor *L) { return L->exec_flops ; }
const Int *orc_perm (const orc_factor *L) { return L->Perm ; }
const Int *orc_colcount (const orc_factor *L) { return L->ColCount ; }
const Int *orc_parent (const orc_factor *L) { return L->Parent ; }
const Int *orc_super (const orc_factor *L) { return L->super ; }
const Int *orc_pi (const orc_factor *L) { return L->pi ; }
const Int *orc_px (const orc_factor *L) { return L->px ; }
const Int *orc_s (const orc_factor *L) { return L->s ; }
const double *orc_x (const orc_factor *L) { return L->x ; }
const Int *orc_calls (const orc_factor *L) { return L->calls ; }
