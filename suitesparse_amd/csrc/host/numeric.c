/* numeric.c -- cholmod_l_factorize / cholmod_l_super_numeric / cholmod_l_solve
 * and the cholmod_l_gpu_* entry points of the host layer.  With Common->useGPU
 * the arithmetic is done by the HIP engine behind include/cholmod_hip.h; this
 * file checks arguments, permutes the input (ptranspose), moves vectors and
 * maps engine return codes to Common->status.  With Common->useGPU == 0, or when
 * no device can be used, the CPU path of cpu_numeric.c runs instead, as in the
 * reference (t_cholmod_super_numeric.c:183-192).
 *
 * Reference files: CHOLMOD/Cholesky/cholmod_factorize.c, cholmod_solve.c;
 * CHOLMOD/Supernodal/cholmod_super_numeric.c, cholmod_super_solve.c;
 * CHOLMOD/GPU/cholmod_gpu.c. */
#include "host_internal.h"
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_thread_num (void) { return 0 ; }
#endif
#include <time.h>

/* ---- GPU entry points (reference CHOLMOD/GPU/cholmod_gpu.c:71-486) ---------------- */

int cholmod_l_gpu_memorysize (size_t *total_mem, size_t *available_mem, cholmod_common *Common)
{
    if (total_mem) *total_mem = 0 ;
    if (available_mem) *available_mem = 0 ;
    if (!Common) return 1 ;
    return cholmod_hip_memorysize (total_mem, available_mem) ;
}

int cholmod_l_gpu_probe (cholmod_common *Common)
{
    if (!Common) return 0 ;
    if (!cholmod_hip_probe ()) return 0 ;
    size_t t = 0, a = 0 ;
    if (cholmod_hip_memorysize (&t, &a) == 0) Common->gpuMemorySize = a ;
    return 1 ;
}

/* The reference carves Common->dev_mempool / host_pinned_mempool here
 * (cholmod_gpu.c:364-486).  The HIP engine keeps L itself resident and sizes
 * its HBM reservation per symbolic factor (cholmod_hip_plan_create), so these
 * two calls only validate that a device exists.  0 = ok, as in the reference. */
/* int-flavour symbols: present and inert in the reference (#ifndef DLONG return 0,
 * CHOLMOD/GPU/cholmod_gpu.c:84-86, :176, :216, :262, :372); kept so that code linking
 * both flavours resolves. */
int cholmod_gpu_memorysize (size_t *total_mem, size_t *available_mem, cholmod_common *Common)
{
    (void) Common ;
    if (total_mem) *total_mem = 0 ;
    if (available_mem) *available_mem = 0 ;
    return 0 ;
}
int cholmod_gpu_probe (cholmod_common *Common) { (void) Common ; return 0 ; }
int cholmod_gpu_deallocate (cholmod_common *Common) { (void) Common ; return 0 ; }
void cholmod_gpu_end (cholmod_common *Common) { (void) Common ; }
int cholmod_gpu_allocate (cholmod_common *Common) { (void) Common ; return 0 ; }

int cholmod_l_gpu_allocate (cholmod_common *Common)
{
    if (!Common) return 1 ;
    return cholmod_hip_probe () ? 0 : 1 ;
}

int cholmod_l_gpu_deallocate (cholmod_common *Common)
{
    if (!Common) return 1 ;
    return 0 ;
}

void cholmod_l_gpu_end (cholmod_common *Common)
{
    (void) Common ;
}

/* ---- plan management ------------------------------------------------------------------ */

static int map_hip_status (int rc, cholmod_common *Common, const char *what)
{
    switch (rc)
    {
        case CHOLMOD_HIP_OK: return TRUE ;
        case CHOLMOD_HIP_NOT_POSDEF: return TRUE ;
        case CHOLMOD_HIP_OUT_OF_MEMORY: ERROR (CHOLMOD_OUT_OF_MEMORY, what) ; return FALSE ;
        case CHOLMOD_HIP_INVALID: ERROR (CHOLMOD_INVALID, what) ; return FALSE ;
        case CHOLMOD_HIP_TOO_LARGE: ERROR (CHOLMOD_TOO_LARGE, "problem too large for the HIP engine (32-bit maps)") ; return FALSE ;
        case CHOLMOD_HIP_NO_DEVICE:
            ERROR (CHOLMOD_GPU_PROBLEM, "no usable HIP device: the numeric factorization of this "
                "library runs only on the GPU engine") ;
            return FALSE ;
        default: ERROR (CHOLMOD_GPU_PROBLEM, what) ; return FALSE ;
    }
}

/* A factor analysed with CHOLMOD_ANALYZE_FOR_SPQR carries the row structure only:
 * px [0] = 123456, the other px zero, xsize = 1 (cholmod_super_symbolic.c:662-663, :749-771).
 * Nothing numeric may be built on it. */
int ssamd_factor_has_cholesky_sizes (const cholmod_factor *L)
{
    if (!L || !L->is_super || !L->px || !L->super || !L->pi) return FALSE ;
    const Int *px = L->px ;
    if (L->nsuper > 0 && px [0] == 123456) return FALSE ;
    return (size_t) px [L->nsuper] <= L->xsize ;
}

/* the plan flags a factorization of L would ask for now */
static int plan_flags_of (const cholmod_factor *L, const cholmod_common *Common)
{
    int flags = Common->hip_flags ;
    /* the real twin of a complex factor (complex.c): the update kernels contract over the even
     * panel columns only -- the complex multiply-add as four real ones (zherk / zgemm,
     * t_cholmod_super_numeric.c:41-83) instead of the eight of the plain embedding */
    if (L->hip_is_twin == 2) flags |= CHOLMOD_HIP_CX_STORAGE ;      /* the complex factor in its own storage */
    else if (L->hip_is_twin)
    {
        const char *e = getenv ("CHOLMOD_HIP_TWIN_FULL_K") ;
        if (!(e && atoi (e) != 0)) flags |= CHOLMOD_HIP_PHI_TWIN ;
    }
    return flags ;
}

int ssamd_ensure_plan (cholmod_factor *L, cholmod_common *Common)
{
    if (!ssamd_factor_has_cholesky_sizes (L))
    { ERROR (CHOLMOD_INVALID, "L was analysed for SPQR (no Cholesky sizes)") ; return FALSE ; }
    int st = 0 ;
    int world = Common->hip_world > 1 ? Common->hip_world : 1 ;
    const int flags = plan_flags_of (L, Common) ;
    if (L->hip_plan && L->hip_plan_ahead && !L->hip_on_device && (L->hip_plan_ahead - 1 != flags || world > 1))
    {
        /* a plan built inside the analysis, and Common->hip_flags / hip_world have changed since: build it again */
        cholmod_hip_plan_destroy ((cholmod_hip_plan *) L->hip_plan) ;
        L->hip_plan = NULL ;
    }
    L->hip_plan_ahead = 0 ;
    if (L->hip_plan) return TRUE ;
    const double t_plan = omp_get_wtime () ;
    cholmod_hip_plan *P = cholmod_hip_plan_create_dist ((int64_t) L->n, (int64_t) L->nsuper,
        L->super, L->pi, L->px, L->s, flags, world > 1 ? Common->hip_rank : 0, world, &st) ;
    Common->hip_plan_seconds = omp_get_wtime () - t_plan ;
    if (!P) return map_hip_status (st ? st : CHOLMOD_HIP_GPU_PROBLEM, Common, "HIP plan creation failed") ;
    /* (several ranks need an exchange: the Common->hip_allreduce callback, or the
     * native RCCL path attached to L->hip_plan with cholmod_hip_rccl_attach after
     * cholmod_l_hip_prepare; without either the factorization returns CHOLMOD_INVALID) */
    if (Common->hip_allreduce)
        cholmod_hip_set_allreduce (P, Common->hip_allreduce, Common->hip_allreduce_user) ;
    L->hip_plan = P ;
    return TRUE ;
}

/* The plan at the end of cholmod_l_analyze (Common->hip_lazy_plan == 0; one GPU -- the ranks of a multi-GPU run attach their
 * exchange to the plan after cholmod_l_hip_prepare).  Not an error of the analysis when it cannot be built: L stays without
 * a plan, the first factorization tries again and reports. */
void ssamd_plan_ahead (cholmod_factor *L, cholmod_common *Common)
{
    if (!L || !L->is_super || !L->useGPU || L->hip_plan || Common->useGPU != 1) return ;
    if (Common->hip_world > 1 || Common->hip_allreduce) return ;
    if (!ssamd_factor_has_cholesky_sizes (L) || !cholmod_hip_probe ()) return ;
    const int st = Common->status, tc = Common->try_catch ;
    Common->try_catch = TRUE ;
    if (ssamd_ensure_plan (L, Common)) L->hip_plan_ahead = plan_flags_of (L, Common) + 1 ;
    Common->try_catch = tc ; Common->status = st ;
}

static void absorb_stats (cholmod_factor *L, cholmod_common *Common)
{
    double s [CHOLMOD_HIP_NSTATS] ;
    if (cholmod_hip_get_stats ((cholmod_hip_plan *) L->hip_plan, s) != CHOLMOD_HIP_OK) return ;
    /* keep the reference's counters meaningful (cholmod_core.h:1004-1024): the
     * engine's dense updates play the role of the dsyrk+dgemm calls */
    Common->gpuKernelTime = s [0] ;
    Common->gpuFlops = (SuiteSparse_long) s [1] ;
    Common->gpuNumKernelLaunches = (int) s [2] ;
    Common->cholmod_gpu_syrk_time = s [6] + s [14] + s [27] + s [32] ;
    Common->cholmod_gpu_syrk_calls = (size_t) (s [7] + s [26] + s [33]) ;
    Common->cholmod_gpu_gemm_time = 0 ; Common->cholmod_gpu_gemm_calls = 0 ;
    Common->cholmod_gpu_potrf_time = s [11] ;
    Common->cholmod_gpu_trsm_time = s [12] ;
    Common->cholmod_assemble_time = s [13] ;
    Common->cholmod_assemble_time2 = s [9] ;
    Common->cholmod_cpu_gemm_time = Common->cholmod_cpu_syrk_time = 0 ;
    Common->cholmod_cpu_trsm_time = Common->cholmod_cpu_potrf_time = 0 ;
    Common->cholmod_cpu_gemm_calls = Common->cholmod_cpu_syrk_calls = 0 ;
    Common->cholmod_cpu_trsm_calls = Common->cholmod_cpu_potrf_calls = 0 ;
}

/* ---- cholmod_l_super_numeric ------------------------------------------------------------ */

/* a factorization that cannot hand its result over leaves L symbolic, as the reference does when it runs out of memory on
 * a symbolic L (cholmod_super_numeric.c:235-248): the values on the device are forgotten, the plan stays */
static int back_to_symbolic (cholmod_factor *L)
{
    L->xtype = CHOLMOD_PATTERN ; L->minor = L->n ;
    L->hip_on_device = FALSE ; L->hip_host_valid = FALSE ;
    return FALSE ;
}

static int finish_numeric (int rc, int64_t minor, cholmod_factor *L, cholmod_common *Common)
{
    const int was_symbolic = (L->xtype == CHOLMOD_PATTERN && !L->x) ;
    if (rc < 0)
    {
        /* the device copy is void whatever L was on entry (run_factorize clears it first): a numeric L keeps the values
         * it has on the host, if it has them (the reference leaves the old values of a numeric L in place); otherwise L is
         * symbolic again -- never a "numeric" L whose device factor is a zeroed or partial one */
        if (was_symbolic || !(L->x && L->hip_host_valid)) back_to_symbolic (L) ;
        else L->hip_on_device = FALSE ;
        return map_hip_status (rc, Common, "HIP factorization failed") ;
    }
    L->xtype = CHOLMOD_REAL ;
    L->dtype = CHOLMOD_DOUBLE ;
    L->is_ll = TRUE ;
    L->minor = (size_t) minor ;
    L->hip_on_device = TRUE ;
    L->hip_host_valid = FALSE ;
    if (!Common->hip_factor_on_device)
    {
        if (Common->hip_world > 1)
        {
            int rg = cholmod_hip_gather_factor ((cholmod_hip_plan *) L->hip_plan) ;
            if (rg != CHOLMOD_HIP_OK) { if (was_symbolic) back_to_symbolic (L) ; return map_hip_status (rg, Common, "factor gather failed") ; }
        }
        if (!L->x) L->x = cholmod_l_malloc (L->xsize, sizeof (double), Common) ;
        if (!L->x) return was_symbolic ? back_to_symbolic (L) : FALSE ;     /* (found by the fault loop, tests/test_memory_faults.py) */
        int r2 = cholmod_hip_download_factor ((cholmod_hip_plan *) L->hip_plan, L->x) ;
        if (r2 != CHOLMOD_HIP_OK)
        {
            if (was_symbolic) { L->x = cholmod_l_free (L->xsize, sizeof (double), L->x, Common) ; back_to_symbolic (L) ; }
            return map_hip_status (r2, Common, "factor download failed") ;
        }
        L->hip_host_valid = TRUE ;
    }
    absorb_stats (L, Common) ;
    if (rc == CHOLMOD_HIP_NOT_POSDEF)
        ERROR (CHOLMOD_NOT_POSDEF, "matrix not positive definite") ;
    return TRUE ;     /* TRUE also when not positive definite, as the reference */
}

/* reference: Supernodal/cholmod_super_numeric.c:97-308.  A must be the
 * permuted matrix in symmetric-lower form (stype < 0). */
int cholmod_l_super_numeric (cholmod_sparse *A, cholmod_sparse *F, double beta [2],
    cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    RETURN_IF_NULL (A, FALSE) ;
    if (A->xtype < CHOLMOD_REAL || A->xtype > CHOLMOD_ZOMPLEX) { ERROR (CHOLMOD_INVALID, "A must be numeric") ; return FALSE ; }
    if (A->stype == 0)
    {
        /* L L' = A*F + beta*I with F = A(:,f)' (t_cholmod_super_numeric.c:223-237, :385-418: the reference assembles every
         * column of A*F on the fly): tril (A*F) is formed on the host (core.c: ssamd_aat) and factorized as the symmetric
         * matrix it is */
        if (!F) { ERROR (CHOLMOD_INVALID, "F is required for the unsymmetric case") ; return FALSE ; }
        if (F->xtype != A->xtype) { ERROR (CHOLMOD_INVALID, "A and F must have the same xtype") ; return FALSE ; }
        if (A->nrow != L->n) { ERROR (CHOLMOD_INVALID, "invalid dimensions") ; return FALSE ; }
        cholmod_sparse *C = ssamd_aat (A, F, 1, TRUE, Common) ;
        if (!C) return FALSE ;
        int okc = cholmod_l_super_numeric (C, NULL, beta, L, Common) ;
        cholmod_l_free_sparse (&C, Common) ;
        return okc ;
    }
    /* a numeric L keeps its kind: real for real A, complex for complex or zomplex A
     * (reference cholmod_super_numeric.c:160-175) */
    const int want = (A->xtype == CHOLMOD_REAL) ? CHOLMOD_REAL : CHOLMOD_COMPLEX ;
    if (L->xtype != CHOLMOD_PATTERN && L->xtype != want)
    { ERROR (CHOLMOD_INVALID, "complex type mismatch") ; return FALSE ; }
    if (A->stype > 0) { ERROR (CHOLMOD_INVALID, "symmetric upper case not supported") ; return FALSE ; }
    if (A->nrow != A->ncol || A->nrow != L->n) { ERROR (CHOLMOD_INVALID, "invalid dimensions") ; return FALSE ; }
    if (!L->is_super) { ERROR (CHOLMOD_INVALID, "L not supernodal") ; return FALSE ; }
    if (!ssamd_factor_has_cholesky_sizes (L))
    { ERROR (CHOLMOD_INVALID, "L was analysed for SPQR (no Cholesky sizes): it cannot be factorized") ; return FALSE ; }
    Common->status = CHOLMOD_OK ;
    double b = beta ? beta [0] : 0.0 ;
    /* complex / zomplex A: the real factorization of the embedded matrix (complex.c) */
    if (want == CHOLMOD_COMPLEX) return ssamd_complex_super_numeric (A, b, L, Common) ;
    /* GPU or CPU (reference t_cholmod_super_numeric.c:183-192: the GPU is used only
     * if Common->useGPU == 1 and the analysis found one, L->useGPU; a device that
     * cannot be initialised degrades to the CPU path).  A factor that already lives
     * on the engine stays there. */
    int want_gpu = (ssamd_resolve_use_gpu (Common) == 1) ;
    if (want_gpu && !(L->useGPU || L->hip_plan) && Common->hip_cpu_fallback) want_gpu = FALSE ;
    if (want_gpu)
    {
        int was_ok = Common->status ;
        void (*eh) (int, const char *, int, const char *) = Common->error_handler ;
        Common->error_handler = NULL ;                  /* a failed device probe is not an error yet */
        int have = ssamd_ensure_plan (L, Common) ;
        Common->error_handler = eh ;
        if (have)
        {
            int64_t minor = (int64_t) L->n ;
            int rc = cholmod_hip_factorize ((cholmod_hip_plan *) L->hip_plan, A->p, A->i,
                A->packed ? NULL : A->nz, A->x, b, Common->quick_return_if_not_posdef, NULL, &minor) ;
            return finish_numeric (rc, minor, L, Common) ;
        }
        /* no device / no HBM for this factor: loud by default; degrade to the CPU path
         * only if the caller asked for the reference's behaviour (several ranks
         * cannot degrade on their own) */
        if (!Common->hip_cpu_fallback || Common->hip_world > 1 || Common->status == CHOLMOD_INVALID
            || Common->status == CHOLMOD_TOO_LARGE)
        {
            int st = Common->status ;
            Common->status = was_ok ;
            ERROR (st, "HIP plan creation failed") ;
            return FALSE ;
        }
        Common->status = CHOLMOD_OK ;
        L->useGPU = 0 ;
    }
    /* ---- CPU path */
    if (L->hip_is_twin == 2) { ERROR (CHOLMOD_GPU_PROBLEM, "a complex factor in engine storage has no CPU form") ; return FALSE ; }
    int was_symbolic = (L->x == NULL) ;
    if (!L->x) L->x = cholmod_l_malloc (L->xsize, sizeof (double), Common) ;
    if (!L->x) return FALSE ;           /* out of memory: L is returned symbolic (cholmod_super_numeric.c:235-248) */
    (void) was_symbolic ;
    L->xtype = CHOLMOD_REAL ; L->dtype = CHOLMOD_DOUBLE ; L->is_ll = TRUE ;
    if (!ssamd_cpu_super_numeric (A, b, L, Common))
    {
        if (was_symbolic) { cholmod_l_free (L->xsize, sizeof (double), L->x, Common) ; L->x = NULL ; L->xtype = CHOLMOD_PATTERN ; }
        return FALSE ;
    }
    L->hip_on_device = FALSE ;
    L->hip_host_valid = TRUE ;
    Common->gpuKernelTime = 0 ; Common->gpuFlops = 0 ; Common->gpuNumKernelLaunches = 0 ;
    return TRUE ;
}

int cholmod_l_refactorize_resident (double beta [2], cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    if (L->cx_twin)
    {
        cholmod_factor *T = L->cx_twin ;
        const int keep = Common->hip_factor_on_device ;
        Common->hip_factor_on_device = TRUE ;       /* the complex L->x is gathered on the device */
        int ok = cholmod_l_refactorize_resident (beta, T, Common) ;
        Common->hip_factor_on_device = keep ;
        if (!ok) return FALSE ;
        int status = Common->status ;
        L->minor = (T->minor >= T->n) ? L->n : T->minor / 2 ;
        L->hip_on_device = T->hip_on_device ; L->hip_host_valid = FALSE ;
        if (!keep && !ssamd_complex_sync_host (L, Common)) return FALSE ;
        Common->status = status ;
        return TRUE ;
    }
    if (!L->hip_plan) { ERROR (CHOLMOD_INVALID, "no resident matrix") ; return FALSE ; }
    Common->status = CHOLMOD_OK ;
    int64_t minor = (int64_t) L->n ;
    int rc = cholmod_hip_factorize_resident ((cholmod_hip_plan *) L->hip_plan, beta ? beta [0] : 0.0,
        Common->quick_return_if_not_posdef, &minor) ;
    return finish_numeric (rc, minor, L, Common) ;
}

/* ---- cholmod_l_factorize ---------------------------------------------------------------- */

static double api_now (void)
{
    struct timespec ts ;
    clock_gettime (CLOCK_MONOTONIC, &ts) ;
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec ;
}

/* 64-bit hash of a packed pattern (dimensions, stype, p, i): the key under which the
 * engine's value map of a matrix is remembered */
/* Two independent 64-bit fingerprints of the pattern (p and i arrays): the values-only fast
 * path of cholmod_l_factorize trusts them as proof that the resident S and its value map still
 * fit A.  Each is a sum of position-keyed, avalanche-mixed words (order-independent, hence
 * parallel and deterministic); the two use different keys, different mixers (the splitmix64
 * and murmur3 finalisers) and are folded differently, so a pattern change must defeat 128
 * bits, not 64 (round-2 review: one 64-bit sum folded by addition was thin). */
/* Threads for the two short host loops of every cholmod_l_factorize call (staging copy of A->x, pattern hash: a few hundred
 * microseconds each).  Measured on the 2 x 64-core box (2D 1259^2, wall around the call, resident step 5.26 ms): a team of 32
 * 7.4 / 8.5 / 10.6 ms in three runs, 16 threads 6.14 / 6.15 / 6.26 ms, 8 threads 6.83 ms -- waking a wide team costs more
 * than its bandwidth returns.  CHOLMOD_API_THREADS overrides. */
static int api_threads (void)
{
    const char *e = getenv ("CHOLMOD_API_THREADS") ;
    if (e && atoi (e) > 0) return atoi (e) ;
    const int nt = ssamd_host_threads () ;
    return nt > 16 ? 16 : nt ;
}

typedef struct { uint64_t hp, gp, hi, gi ; } pat_sums ;

/* the column pointers j0 .. j1-1 / the row indices p0 .. p1-1 into the four sums (order-independent: any split will do) */
static void pattern_hash_p (const Int *Ap, Int j0, Int j1, pat_sums *S)
{
    uint64_t hp = 0, gp = 0 ;
    for (Int j = j0 ; j < j1 ; j++)
    {
        uint64_t x = (uint64_t) Ap [j] + 0x9E3779B97F4A7C15ull * (uint64_t) (j + 1) ;
        x ^= x >> 30 ; x *= 0xbf58476d1ce4e5b9ull ; x ^= x >> 27 ; x *= 0x94d049bb133111ebull ; x ^= x >> 31 ;
        hp += x ;
        uint64_t y = ((uint64_t) Ap [j] ^ 0xA24BAED4963EE407ull) * (2 * (uint64_t) j + 0x632BE59BD9B4E019ull) ;
        y ^= y >> 33 ; y *= 0xff51afd7ed558ccdull ; y ^= y >> 33 ; y *= 0xc4ceb9fe1a85ec53ull ; y ^= y >> 33 ;
        gp += y ;
    }
    S->hp += hp ; S->gp += gp ;
}

static void pattern_hash_i (const Int *Ai, Int p0, Int p1, pat_sums *S)
{
    uint64_t hi = 0, gi = 0 ;
    for (Int p = p0 ; p < p1 ; p++)
    {
        uint64_t x = (uint64_t) Ai [p] + 0xD1B54A32D192ED03ull * (uint64_t) (p + 1) ;
        x ^= x >> 30 ; x *= 0xbf58476d1ce4e5b9ull ; x ^= x >> 27 ; x *= 0x94d049bb133111ebull ; x ^= x >> 31 ;
        hi += x ;
        uint64_t y = ((uint64_t) Ai [p] ^ 0x8EBC6AF09C88C6E3ull) * (2 * (uint64_t) p + 0x589965CC75374CC3ull) ;
        y ^= y >> 33 ; y *= 0xff51afd7ed558ccdull ; y ^= y >> 33 ; y *= 0xc4ceb9fe1a85ec53ull ; y ^= y >> 33 ;
        gi += y ;
    }
    S->hi += hi ; S->gi += gi ;
}

static uint64_t pattern_hash_fold (cholmod_sparse *A, const pat_sums *S, uint64_t *second)
{
    const Int ncol = (Int) A->ncol, nz = ((Int *) A->p) [ncol] ;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t) A->nrow * 0xff51afd7ed558ccdull) ^ ((uint64_t) (A->stype + 2) << 56) ;
    if (second)
    {
        uint64_t g = S->gp ^ ((S->gi << 23) | (S->gi >> 41)) ^ ((uint64_t) nz * 0x9FB21C651E98DF25ull) ;
        g ^= g >> 32 ; g *= 0xd6e8feb86659fd93ull ; g ^= g >> 32 ;
        *second = g ^ ((uint64_t) ncol << 17) ;
    }
    return h ^ S->hp ^ (S->hi * 0x2545F4914F6CDD1Dull) ;
}

static uint64_t pattern_hash (cholmod_sparse *A, uint64_t *second)
{
    const Int *Ap = A->p, *Ai = A->i ;
    const Int ncol = (Int) A->ncol, nz = Ap [ncol] ;
    const int nth = api_threads () ;
    uint64_t hp = 0, hi = 0, gp = 0, gi = 0 ;
    const Int PIECE = (Int) 1 << 16 ;
#pragma omp parallel for schedule(static) num_threads(nth) reduction(+:hp,gp)
    for (Int j = 0 ; j <= ncol ; j += PIECE)
    {
        pat_sums T = {0, 0, 0, 0} ;
        pattern_hash_p (Ap, j, (j + PIECE < ncol + 1) ? j + PIECE : ncol + 1, &T) ;
        hp += T.hp ; gp += T.gp ;
    }
#pragma omp parallel for schedule(static) num_threads(nth) reduction(+:hi,gi)
    for (Int p = 0 ; p < nz ; p += PIECE)
    {
        pat_sums T = {0, 0, 0, 0} ;
        pattern_hash_i (Ai, p, (p + PIECE < nz) ? p + PIECE : nz, &T) ;
        hi += T.hi ; gi += T.gi ;
    }
    pat_sums S = {hp, gp, hi, gi} ;
    return pattern_hash_fold (A, &S, second) ;
}

/* reference: Cholesky/cholmod_factorize.c:97-300, supernodal symmetric branch
 * (:177-288): S = tril(P A P') by one permuted transpose when A is stored
 * upper (:225-232) or two when it is stored lower (:233-244). */
int cholmod_l_factorize_p (cholmod_sparse *A, double beta [2], SuiteSparse_long *fset, size_t fsize,
    cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (A, FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    if (A->xtype < CHOLMOD_REAL || A->xtype > CHOLMOD_ZOMPLEX) { ERROR (CHOLMOD_INVALID, "A must be numeric") ; return FALSE ; }
    if (A->nrow != L->n) { ERROR (CHOLMOD_INVALID, "A and L dimensions do not match") ; return FALSE ; }
    if (!L->is_super) { ERROR (CHOLMOD_NOT_INSTALLED, "simplicial factorization not built") ; return FALSE ; }
    if (A->stype == 0)
    {
        /* factorize A*A' + beta*I (cholmod_factorize.c:197-224: S = A(p,f), F = S', super_numeric (S, F, beta)): tril (A*A')
         * is formed on the host and takes the symmetric branch below -- permutation, upload and, from the second call with
         * the same pattern on, the values-only path included.  A column subset f: A(:,f)*A(:,f)', the columns cut out first. */
        cholmod_sparse *Af = fset ? ssamd_column_subset (A, fset, fsize, 1, Common) : NULL ;
        if (fset && !Af) return FALSE ;
        cholmod_sparse *C = ssamd_aat (Af ? Af : A, NULL, 1, TRUE, Common) ;
        if (Af) cholmod_l_free_sparse (&Af, Common) ;
        if (!C) return FALSE ;
        int okc = cholmod_l_factorize_p (C, beta, NULL, 0, L, Common) ;
        cholmod_l_free_sparse (&C, Common) ;
        return okc ;
    }
    if (A->nrow != A->ncol) { ERROR (CHOLMOD_INVALID, "A and L dimensions do not match") ; return FALSE ; }
    Common->status = CHOLMOD_OK ;
    double zero [2] = {0, 0} ;
    /* Same pattern as the matrix the engine already holds (hash of p / i, nnz): only the
     * values travel -- H2D of A->x and a gather into the resident S on the device; the
     * host-side permutation, the pattern upload and the assembly search are skipped. */
    const int vmap_ok = (A->xtype == CHOLMOD_REAL && A->packed && L->hip_plan && Common->hip_world <= 1
        && ssamd_resolve_use_gpu (Common) == 1) ;
    uint64_t hash2 = 0, hash = 0 ;
    int hashed = FALSE ;
    size_t annz = vmap_ok ? (size_t) ((Int *) A->p) [A->ncol] : 0 ;
    if (vmap_ok && L->hip_apat_valid && L->hip_apat_nnz == annz && (L->xtype == CHOLMOD_REAL || L->xtype == CHOLMOD_PATTERN))
    {
        /* The values travel first, the proof that they belong there is computed while they are on their way: A->x goes
         * chunk by chunk into the plan's pinned staging buffer (all host threads) and every chunk leaves by DMA as soon as
         * it is filled; the hash of the pattern is taken while the last chunks are in flight; then commit (gather on the
         * device, the factorization's assembly waits for it -- its clearing of L runs beside the upload) or cancel. */
        cholmod_hip_plan *plan = (cholmod_hip_plan *) L->hip_plan ;
        double *stage = NULL ;
        int64_t cap = 0 ;
        const int timing = getenv ("CHOLMOD_API_TIMING") != NULL ;
        double tq0 = timing ? api_now () : 0, tq1 = 0, tq2 = 0 ;
        int rc = cholmod_hip_values_staging (plan, &stage, &cap) ;
        const int nth = api_threads () ;
        const int overlap = 1 ;
        const int64_t *gidx = NULL ;
        int64_t glen = 0, gcount = 0 ;
        if (rc == CHOLMOD_HIP_OK && (size_t) cap == annz && nth >= 3 && overlap
            && cholmod_hip_values_gather_index (plan, &gidx, &glen, &gcount) == CHOLMOD_HIP_OK && gidx && glen > 0 && gcount <= cap
            && cholmod_hip_values_begin (plan) == CHOLMOD_HIP_OK)
        {
            /* Round 6, second step: the values in the order the factorization needs them.  The entries of S are staged
             * sorted by the batch of their front (stage [k] = A->x [index [k]]: the permutation to S's order, formerly a
             * gather kernel on the device, happens in the staging copy) and pushed chunk by chunk; the factorization --
             * already enqueueing on thread 0, its clearing of L first -- waits before every batch for that batch's chunks
             * only.  Thread 0: the factorization; thread 1: pushes chunks as they are complete; the others: stage, then
             * take the pattern hash.  The hash is checked after the fact, as below. */
            const double *Ax = A->x ;
            const Int *Ap = A->p, *Ai = A->i ;
            const Int ncol = (Int) A->ncol ;
            const int64_t snz = gcount ;                    /* (one staged value per entry of S) */
            const int64_t CH = glen, PIECE = 16384 ;        /* (glen is a multiple of PIECE: 2^20) */
            const int64_t nchunk = (snz + CH - 1) / CH, npiece = (snz + PIECE - 1) / PIECE ;
            const int64_t HP = (int64_t) 1 << 16 ;
            const int64_t nhp = ((int64_t) ncol + 1 + HP - 1) / HP, nhi = ((int64_t) annz + HP - 1) / HP ;
            int64_t *chunk_done = cholmod_l_calloc ((size_t) (nchunk > 0 ? nchunk : 1), sizeof (int64_t), Common) ;
            int64_t next_piece = 0, next_hash = 0 ;
            pat_sums tot = {0, 0, 0, 0} ;
            int64_t minor = (int64_t) L->n ;
            int rcf = CHOLMOD_HIP_OK ;
            if (!chunk_done)
            {
                /* (the prologue is enqueued and a factorization is expected to follow: let it fail cleanly) */
                (void) cholmod_hip_values_push_chunk (plan, -1) ;
                (void) cholmod_hip_factorize_resident (plan, 0.0, 0, &minor) ;
                return finish_numeric (CHOLMOD_HIP_OUT_OF_MEMORY, minor, L, Common) ;
            }
#pragma omp parallel num_threads(nth)
            {
                const int me = omp_get_thread_num (), team = omp_get_num_threads () ;
                /* (a team smaller than asked for -- a thread limit, a call from inside a parallel region -- must not leave
                 * the factorization waiting for chunks nobody pushes: thread 0 then stages and pushes everything first) */
                const int stager = (team >= 3) ? (me >= 2) : (me == 0), pusher = (team >= 3) ? (me == 1) : (me == 0) ;
                if (stager)
                    for ( ; ; )
                    {
                        const int64_t k = __atomic_fetch_add (&next_piece, 1, __ATOMIC_RELAXED) ;
                        if (k >= npiece) break ;
                        const int64_t q0 = k * PIECE, q1 = (q0 + PIECE < snz) ? q0 + PIECE : snz ;
                        for (int64_t q = q0 ; q < q1 ; q++)
                        {
                            /* (the index rises inside a batch: mostly the next cache lines, the prefetch covers the gaps) */
                            if (q + 24 < q1) __builtin_prefetch (Ax + gidx [q + 24], 0, 0) ;
                            stage [q] = Ax [gidx [q]] ;
                        }
                        __atomic_fetch_add (&chunk_done [q0 / CH], 1, __ATOMIC_RELEASE) ;
                    }
                if (pusher)
                {
                    for (int64_t c = 0 ; c < nchunk ; c++)
                    {
                        const int64_t o = c * CH, cnt = (snz - o < CH) ? snz - o : CH ;
                        const int64_t need = (cnt + PIECE - 1) / PIECE ;
                        while (__atomic_load_n (&chunk_done [c], __ATOMIC_ACQUIRE) < need) { /* (microseconds) */ }
                        if (cholmod_hip_values_push_chunk (plan, c) != CHOLMOD_HIP_OK) break ;
                    }
                    if (timing) tq1 = api_now () ;
                }
                if (me == 0)
                    rcf = cholmod_hip_factorize_resident (plan, beta ? beta [0] : 0.0, Common->quick_return_if_not_posdef, &minor) ;
                if (me >= 1 || team == 1)
                {
                    pat_sums mine = {0, 0, 0, 0} ;
                    for ( ; ; )
                    {
                        const int64_t k = __atomic_fetch_add (&next_hash, 1, __ATOMIC_RELAXED) ;
                        if (k >= nhp + nhi) break ;
                        if (k < nhp) pattern_hash_p (Ap, (Int) (k * HP), (Int) (((k + 1) * HP < ncol + 1) ? (k + 1) * HP : ncol + 1), &mine) ;
                        else pattern_hash_i (Ai, (Int) ((k - nhp) * HP), (Int) (((k - nhp + 1) * HP < (int64_t) annz) ? (k - nhp + 1) * HP : (int64_t) annz), &mine) ;
                    }
#pragma omp critical (ssamd_pattern_hash)
                    { tot.hp += mine.hp ; tot.gp += mine.gp ; tot.hi += mine.hi ; tot.gi += mine.gi ; }
                }
            }
            cholmod_l_free ((size_t) (nchunk > 0 ? nchunk : 1), sizeof (int64_t), chunk_done, Common) ;
            hash = pattern_hash_fold (A, &tot, &hash2) ;
            hashed = TRUE ;
            if (timing) tq2 = api_now () ;
            if (L->hip_apat_hash == hash && L->hip_apat_hash2 == hash2)
            {
                if (timing) fprintf (stderr, "cholmod_l_factorize (values only, in batch order): last push at %.3f ms, factorization + hash done at %.3f ms\n",
                    1e3 * (tq1 - tq0), 1e3 * (tq2 - tq0)) ;
                return finish_numeric (rcf, minor, L, Common) ;
            }
            /* (another pattern after all: the device's S, map and factor are void; the long way rebuilds all three) */
        }
        else if (rc == CHOLMOD_HIP_OK && (size_t) cap == annz && nth >= 2 && overlap)
        {
            /* Round 6: nothing but the DMA itself stands between the call and the factorization.  Thread 0 pushes every chunk
             * as soon as the others have staged it, commits and ENQUEUES THE FACTORIZATION AT ONCE; the other threads stage
             * the chunks (A->x into the pinned buffer) and then take the pattern hash while the device already works.  The
             * proof that the resident S and its value map fit A therefore arrives after the fact: should it fail -- another
             * pattern with the same number of entries -- what the device computed is discarded and the call takes the long
             * way (permutation, full upload, factorization), which rebuilds S, map and factor from scratch.  (Round 5: stage
             * 1.0 ms, then the hash 1.25 ms, then the factorization: 2.7 ms on top of the nd24k stand-in's 27.9 ms step.) */
            const double *Ax = A->x ;
            const Int *Ap = A->p, *Ai = A->i ;
            const Int ncol = (Int) A->ncol ;
            const int64_t CH = (int64_t) 1 << 20 ;          /* 8 MB per push */
            const int64_t PIECE = 32768 ;                   /* 256 KB per staging turn */
            const int64_t nchunk = (cap + CH - 1) / CH ;
            const int64_t npiece = (cap + PIECE - 1) / PIECE ;
            const int64_t HP = (int64_t) 1 << 16 ;          /* entries per hash turn */
            const int64_t nhp = ((int64_t) ncol + 1 + HP - 1) / HP, nhi = ((int64_t) annz + HP - 1) / HP ;
            int64_t *chunk_done = cholmod_l_calloc ((size_t) (nchunk > 0 ? nchunk : 1), sizeof (int64_t), Common) ;
            if (!chunk_done) return FALSE ;
            int64_t next_piece = 0, next_hash = 0 ;
            pat_sums tot = {0, 0, 0, 0} ;
            int64_t minor = (int64_t) L->n ;
            int rcf = CHOLMOD_HIP_OK, rcp = CHOLMOD_HIP_OK ;
#pragma omp parallel num_threads(nth)
            {
                if (omp_get_thread_num () == 0)
                {
                    for (int64_t c = 0 ; c < nchunk && rcp == CHOLMOD_HIP_OK ; c++)
                    {
                        const int64_t o = c * CH, cnt = (cap - o < CH) ? cap - o : CH ;
                        const int64_t need = (cnt + PIECE - 1) / PIECE ;
                        while (__atomic_load_n (&chunk_done [c], __ATOMIC_ACQUIRE) < need) { /* (a few microseconds) */ }
                        rcp = cholmod_hip_values_push (plan, o, cnt) ;
                    }
                    if (timing) tq1 = api_now () ;
                    if (rcp == CHOLMOD_HIP_OK && cholmod_hip_values_commit (plan, 1) == CHOLMOD_HIP_OK)
                        rcf = cholmod_hip_factorize_resident (plan, beta ? beta [0] : 0.0, Common->quick_return_if_not_posdef, &minor) ;
                    else
                    {
                        (void) cholmod_hip_values_commit (plan, 0) ;
                        rcp = CHOLMOD_HIP_INVALID ;
                    }
                }
                else
                {
                    for ( ; ; )
                    {
                        const int64_t k = __atomic_fetch_add (&next_piece, 1, __ATOMIC_RELAXED) ;
                        if (k >= npiece) break ;
                        const int64_t q = k * PIECE, len = (cap - q < PIECE) ? cap - q : PIECE ;
                        memcpy (stage + q, Ax + q, (size_t) len * sizeof (double)) ;
                        __atomic_fetch_add (&chunk_done [q / CH], 1, __ATOMIC_RELEASE) ;
                    }
                    pat_sums mine = {0, 0, 0, 0} ;
                    for ( ; ; )
                    {
                        const int64_t k = __atomic_fetch_add (&next_hash, 1, __ATOMIC_RELAXED) ;
                        if (k >= nhp + nhi) break ;
                        if (k < nhp) pattern_hash_p (Ap, (Int) (k * HP), (Int) (((k + 1) * HP < ncol + 1) ? (k + 1) * HP : ncol + 1), &mine) ;
                        else pattern_hash_i (Ai, (Int) ((k - nhp) * HP), (Int) (((k - nhp + 1) * HP < (int64_t) annz) ? (k - nhp + 1) * HP : (int64_t) annz), &mine) ;
                    }
#pragma omp critical (ssamd_pattern_hash)
                    { tot.hp += mine.hp ; tot.gp += mine.gp ; tot.hi += mine.hi ; tot.gi += mine.gi ; }
                }
            }
            cholmod_l_free ((size_t) (nchunk > 0 ? nchunk : 1), sizeof (int64_t), chunk_done, Common) ;
            hash = pattern_hash_fold (A, &tot, &hash2) ;
            hashed = TRUE ;
            if (timing) tq2 = api_now () ;
            if (rcp == CHOLMOD_HIP_OK && L->hip_apat_hash == hash && L->hip_apat_hash2 == hash2)
            {
                if (timing) fprintf (stderr, "cholmod_l_factorize (values only, overlapped): last push at %.3f ms, factorization + hash done at %.3f ms\n",
                    1e3 * (tq1 - tq0), 1e3 * (tq2 - tq0)) ;
                return finish_numeric (rcf, minor, L, Common) ;
            }
            /* (another pattern after all: the device's S, map and factor are void; the long way rebuilds all three) */
        }
        else if (rc == CHOLMOD_HIP_OK && (size_t) cap == annz)
        {
            const double *Ax = A->x ;
            const int64_t CH = (int64_t) 1 << 20 ;          /* 8 MB per push */
            for (int64_t o = 0 ; o < cap && rc == CHOLMOD_HIP_OK ; o += CH)
            {
                const int64_t cnt = (cap - o < CH) ? cap - o : CH ;
                const int64_t PIECE = 32768 ;               /* 256 KB per thread and turn */
#pragma omp parallel for schedule(static) num_threads(nth) if (cnt > 4 * PIECE)
                for (int64_t q = 0 ; q < cnt ; q += PIECE)
                    memcpy (stage + o + q, Ax + o + q, (size_t) ((cnt - q < PIECE) ? cnt - q : PIECE) * sizeof (double)) ;
                rc = cholmod_hip_values_push (plan, o, cnt) ;
            }
            if (timing) tq1 = api_now () ;
            hash = pattern_hash (A, &hash2) ;
            hashed = TRUE ;
            if (timing) tq2 = api_now () ;
            const int same = (rc == CHOLMOD_HIP_OK && L->hip_apat_hash == hash && L->hip_apat_hash2 == hash2) ;
            if (cholmod_hip_values_commit (plan, same) == CHOLMOD_HIP_OK && same)
            {
                int64_t minor = (int64_t) L->n ;
                rc = cholmod_hip_factorize_resident (plan, beta ? beta [0] : 0.0, Common->quick_return_if_not_posdef, &minor) ;
                if (timing) fprintf (stderr, "cholmod_l_factorize (values only): stage + push %.3f ms, pattern hash %.3f ms, commit + factorization %.3f ms\n",
                    1e3 * (tq1 - tq0), 1e3 * (tq2 - tq1), 1e3 * (api_now () - tq2)) ;
                return finish_numeric (rc, minor, L, Common) ;
            }
        }
        L->hip_apat_valid = FALSE ;         /* no usable map after all: the long way */
    }
    const int ltiming = getenv ("CHOLMOD_API_TIMING") != NULL ;
    double tl0 = ltiming ? api_now () : 0, tl1 = 0, tl2 = 0, tl3 = 0 ;
    if (vmap_ok && !hashed) hash = pattern_hash (A, &hash2) ;
    cholmod_sparse *S = NULL ;
    Int *src = NULL ;
    int natural = (L->ordering == CHOLMOD_NATURAL) ;
    Int *Perm = natural ? NULL : (Int *) L->Perm ;
    if (natural && A->packed && A->stype < 0)
    {
        S = A ;
    }
    else
    {
        /* S = tril (P A P'): upper-stored A by the conjugate permuted transpose of the
         * reference (:225-232); lower-stored A: the reference transposes twice (:233-244),
         * here one symmetric permutation lands in the lower triangle directly */
        S = ssamd_sym_permute_src (A, 2, Perm, FALSE, vmap_ok || (A->xtype == CHOLMOD_REAL && A->packed) ? &src : NULL, Common) ;
    }
    if (!S) return FALSE ;
    L->hip_apat_valid = FALSE ;
    if (ltiming) tl1 = api_now () ;
    int ok = cholmod_l_super_numeric (S, NULL, beta ? beta : zero, L, Common) ;
    if (ltiming) tl2 = api_now () ;
    /* the engine now holds S: tell it where S's values come from in A */
    if (ok && L->hip_plan && L->hip_on_device && A->xtype == CHOLMOD_REAL && A->packed && Common->hip_world <= 1
        && (S == A || src))
    {
        size_t snz = (size_t) ((Int *) S->p) [S->ncol] ;
        size_t an = (size_t) ((Int *) A->p) [A->ncol] ;
        Int *id = NULL ;
        if (S == A)
        {
            id = cholmod_l_malloc (snz > 0 ? snz : 1, sizeof (Int), Common) ;
            if (id) for (size_t q = 0 ; q < snz ; q++) id [q] = (Int) q ;
        }
        if ((S == A ? id : src) && cholmod_hip_set_value_map ((cholmod_hip_plan *) L->hip_plan,
                S == A ? id : src, (int64_t) snz, (int64_t) an) == CHOLMOD_HIP_OK)
        {
            if (!vmap_ok) hash = pattern_hash (A, &hash2) ;
            L->hip_apat_hash = hash ; L->hip_apat_hash2 = hash2 ;
            L->hip_apat_nnz = an ;
            L->hip_apat_valid = TRUE ;
        }
        if (id) cholmod_l_free (snz > 0 ? snz : 1, sizeof (Int), id, Common) ;
    }
    if (src) cholmod_l_free (((Int *) S->p) [S->ncol] > 0 ? ((Int *) S->p) [S->ncol] : 1, sizeof (Int), src, Common) ;
    if (S != A) cholmod_l_free_sparse (&S, Common) ;
    if (ltiming)
    {
        tl3 = api_now () ;
        fprintf (stderr, "cholmod_l_factorize (long way): hash + permutation %.3f s, plan + upload + factorization %.3f s, value map %.3f s\n",
            tl1 - tl0, tl2 - tl1, tl3 - tl2) ;
    }
    return ok ;
}

int cholmod_l_factorize (cholmod_sparse *A, cholmod_factor *L, cholmod_common *Common)
{
    double zero [2] = {0, 0} ;
    return cholmod_l_factorize_p (A, zero, NULL, 0, L, Common) ;
}

int cholmod_l_factor_to_host (cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    if (L->hip_host_valid) return TRUE ;
    if (L->cx_twin) return ssamd_complex_sync_host (L, Common) ;
    if (!L->hip_plan || !L->hip_on_device) { ERROR (CHOLMOD_INVALID, "no numeric factor") ; return FALSE ; }
    if (!L->x) L->x = cholmod_l_malloc (L->xsize, sizeof (double), Common) ;
    if (!L->x) return FALSE ;
    int rc = cholmod_hip_download_factor ((cholmod_hip_plan *) L->hip_plan, L->x) ;
    if (rc != CHOLMOD_HIP_OK) return map_hip_status (rc, Common, "factor download failed") ;
    L->hip_host_valid = TRUE ;
    return TRUE ;
}

/* ---- cholmod_l_rcond -------------------------------------------------------------------------- */

/* reference: Cholesky/cholmod_rcond.c:64-161.  (min L_jj / max L_jj)^2 over the diagonal of the supernodal LL' factor:
 * -1 on error, 1 for a 0-by-0 matrix, 0 if the factorization failed (L->minor < n) or a diagonal entry is NaN.  A factor
 * that lives in HBM is not downloaded for this: one pass over its n diagonal entries on the device
 * (cholmod_hip_diag_minmax) returns the two extremes. */
double cholmod_l_rcond (cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (EMPTY) ;
    RETURN_IF_NULL (L, EMPTY) ;
    if (L->xtype < CHOLMOD_REAL || L->xtype > CHOLMOD_ZOMPLEX) { ERROR (CHOLMOD_INVALID, "invalid xtype") ; return EMPTY ; }
    Common->status = CHOLMOD_OK ;
    if (L->n == 0) return 1 ;
    if (L->minor < L->n) return 0 ;
    if (!L->is_super) { ERROR (CHOLMOD_NOT_INSTALLED, "simplicial factors not built") ; return EMPTY ; }
    double lmin, lmax ;
    cholmod_factor *D = L->cx_twin ? (cholmod_factor *) L->cx_twin : L ;      /* (complex L: the engine's factor) */
    if (D->hip_plan && D->hip_on_device && !L->hip_host_valid)
    {
        if (Common->hip_world > 1)
        {
            int rg = cholmod_hip_gather_factor ((cholmod_hip_plan *) D->hip_plan) ;
            if (rg != CHOLMOD_HIP_OK) { map_hip_status (rg, Common, "factor gather failed") ; return EMPTY ; }
        }
        double out [3] ;
        int rc = cholmod_hip_diag_minmax ((cholmod_hip_plan *) D->hip_plan, out) ;
        if (rc != CHOLMOD_HIP_OK) { map_hip_status (rc, Common, "diagonal scan failed") ; return EMPTY ; }
        if (out [2] > 0) return 0 ;             /* a NaN on the diagonal (:31-41) */
        lmin = out [0] ; lmax = out [1] ;
    }
    else
    {
        if (!L->x) { ERROR (CHOLMOD_INVALID, "no numeric values") ; return EMPTY ; }
        const Int *Super = L->super, *Lpi = L->pi, *Lpx = L->px ;
        const double *Lx = L->x ;
        const Int e = (L->xtype == CHOLMOD_COMPLEX) ? 2 : 1 ;
        lmin = lmax = Lx [0] ;
        if (lmin != lmin) return 0 ;
        for (Int s = 0 ; s < (Int) L->nsuper ; s++)
        {
            const Int nscol = Super [s+1] - Super [s], nsrow = Lpi [s+1] - Lpi [s], psx = Lpx [s] ;
            for (Int jj = 0 ; jj < nscol ; jj++)
            {
                const double ljj = Lx [e * (psx + jj + jj * nsrow)] ;
                if (ljj != ljj) return 0 ;
                if (ljj < lmin) lmin = ljj ; else if (ljj > lmax) lmax = ljj ;
            }
        }
    }
    double rcond = lmin / lmax ;
    if (L->is_ll) rcond = rcond * rcond ;
    return rcond ;
}

/* ---- cholmod_l_change_factor ---------------------------------------------------------------------- */

/* reference: Core/cholmod_change_factor.c:1005-1230, the supernodal conversions (the simplicial forms are not built):
 *   supernodal numeric  -> supernodal symbolic  (ll_super_to_super_symbolic, :373-393): the values are discarded -- L->x
 *                                                freed, the engine's resident copy forgotten (the plan, a function of the
 *                                                symbolic factor, stays) --, xtype = PATTERN, minor = n;
 *   supernodal symbolic -> supernodal numeric   (super_symbolic_to_ll_super, :946-989): L->x allocated, contents
 *                                                undefined, xtype = to_xtype, minor = n -- what cholmod_l_super_numeric
 *                                                does on entry (cholmod_super_numeric.c:212-223);
 *   supernodal numeric  -> supernodal numeric   nothing to do.
 * to_ll / to_packed / to_monotonic only concern simplicial factors.  A failed conversion leaves L as it was. */
int cholmod_l_change_factor (int to_xtype, int to_ll, int to_super, int to_packed, int to_monotonic,
    cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    (void) to_ll ; (void) to_packed ; (void) to_monotonic ;
    if (L->xtype < CHOLMOD_PATTERN || L->xtype > CHOLMOD_ZOMPLEX) { ERROR (CHOLMOD_INVALID, "invalid xtype") ; return FALSE ; }
    if (to_xtype < CHOLMOD_PATTERN || to_xtype > CHOLMOD_ZOMPLEX) { ERROR (CHOLMOD_INVALID, "xtype invalid") ; return FALSE ; }
    Common->status = CHOLMOD_OK ;
    if (to_super && to_xtype == CHOLMOD_ZOMPLEX) { ERROR (CHOLMOD_INVALID, "supernodal zomplex L not supported") ; return FALSE ; }
    if (!to_super || !L->is_super)
    {
        /* any conversion from or to a simplicial factor (:1053-1062, :1138-1224) */
        ERROR (CHOLMOD_NOT_INSTALLED, "simplicial factors are not built: only supernodal symbolic <-> numeric") ;
        return FALSE ;
    }
    if (to_xtype == CHOLMOD_PATTERN)
    {
        if (L->xtype == CHOLMOD_PATTERN) return TRUE ;
        const size_t w = (L->xtype == CHOLMOD_COMPLEX) ? 2 : 1 ;
        if (L->x) L->x = cholmod_l_free (L->xsize, w * sizeof (double), L->x, Common) ;
        if (L->cx_twin) cholmod_l_free_factor ((cholmod_factor **) &L->cx_twin, Common) ;
        L->xtype = CHOLMOD_PATTERN ; L->dtype = CHOLMOD_DOUBLE ;
        L->minor = L->n ; L->is_ll = TRUE ;
        L->hip_on_device = FALSE ; L->hip_host_valid = FALSE ; L->hip_apat_valid = FALSE ;
        return TRUE ;
    }
    if (L->xtype != CHOLMOD_PATTERN) return TRUE ;       /* already numeric (:1129-1136: nothing to do) */
    if (!ssamd_factor_has_cholesky_sizes (L))
    { ERROR (CHOLMOD_INVALID, "L was analysed for SPQR (no Cholesky sizes)") ; return FALSE ; }
    const size_t w = (to_xtype == CHOLMOD_REAL) ? 1 : 2 ;
    double *Lx = cholmod_l_malloc (L->xsize, w * sizeof (double), Common) ;
    if (!Lx) return FALSE ;                                 /* out of memory: L unchanged */
    if (L->xsize == 1) { Lx [0] = 0 ; if (w == 2) Lx [1] = 0 ; }
    L->x = Lx ; L->xtype = to_xtype ; L->dtype = CHOLMOD_DOUBLE ; L->minor = L->n ; L->is_ll = TRUE ;
    L->hip_on_device = FALSE ; L->hip_host_valid = TRUE ;   /* (what the caller sees is L->x) */
    return TRUE ;
}

int cholmod_l_hip_prepare (cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    if (!L->is_super) { ERROR (CHOLMOD_INVALID, "L not supernodal") ; return FALSE ; }
    Common->status = CHOLMOD_OK ;
    if (L->cx_twin) return ssamd_ensure_plan ((cholmod_factor *) L->cx_twin, Common) ;
    return ssamd_ensure_plan (L, Common) ;
}

int cholmod_l_gather_factor (cholmod_factor *L, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    if (L->cx_twin) return cholmod_l_gather_factor ((cholmod_factor *) L->cx_twin, Common) ;
    if (!L->hip_plan || !L->hip_on_device) { ERROR (CHOLMOD_INVALID, "no numeric factor") ; return FALSE ; }
    int rc = cholmod_hip_gather_factor ((cholmod_hip_plan *) L->hip_plan) ;
    if (rc != CHOLMOD_HIP_OK) return map_hip_status (rc, Common, "factor gather failed") ;
    return TRUE ;
}

int cholmod_l_hip_stats (cholmod_factor *L, double *stats, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    RETURN_IF_NULL (stats, FALSE) ;
    if (L->cx_twin) L = (cholmod_factor *) L->cx_twin ;
    if (!L->hip_plan) { ERROR (CHOLMOD_INVALID, "no plan") ; return FALSE ; }
    return cholmod_hip_get_stats ((cholmod_hip_plan *) L->hip_plan, stats) == CHOLMOD_HIP_OK ;
}

/* ---- triangular solves ---------------------------------------------------------------------- */

/* make sure the device holds the numeric values the caller sees in L->x */
static int factor_on_device (cholmod_factor *L, cholmod_common *Common)
{
    if (L->xtype != CHOLMOD_REAL || !L->is_super)
    { ERROR (CHOLMOD_INVALID, "L must be a numeric supernodal factor") ; return FALSE ; }
    if (!ssamd_ensure_plan (L, Common)) return FALSE ;
    if (!L->hip_on_device)
    {
        if (!L->x) { ERROR (CHOLMOD_INVALID, "L has no values") ; return FALSE ; }
        int rc = cholmod_hip_upload_factor ((cholmod_hip_plan *) L->hip_plan, L->x) ;
        if (rc != CHOLMOD_HIP_OK) return map_hip_status (rc, Common, "factor upload failed") ;
        L->hip_on_device = TRUE ;
    }
    return TRUE ;
}

/* the triangular solves run where the factor is: on the host when the values are
 * in L->x and the engine is not to be used (Common->useGPU == 0, no device, or a
 * factor the CPU path computed) */
static int solve_on_host (cholmod_factor *L, cholmod_common *Common)
{
    if (L->xtype != CHOLMOD_REAL || !L->is_super || !L->x) return FALSE ;
    if (L->hip_on_device && !L->hip_host_valid) return FALSE ;      /* current values only in HBM */
    if (L->hip_on_device) return FALSE ;                            /* resident copy: use it */
    if (ssamd_resolve_use_gpu (Common) != 1) return TRUE ;
    /* a GPU was asked for: the host solve is only the opted-in degradation */
    return Common->hip_cpu_fallback && (!L->useGPU || !cholmod_hip_probe ()) ;
}

static int super_solve (int which, cholmod_factor *L, cholmod_dense *X, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    RETURN_IF_NULL (X, FALSE) ;
    /* L and X must be of the same kind (reference cholmod_super_solve.c:70-75); a complex
     * X against a complex L is the real system of the twin on the same bytes (complex.c) */
    Int ldx = (Int) X->d ;
    if (L->xtype == CHOLMOD_COMPLEX && L->cx_twin)
    {
        if (X->xtype != CHOLMOD_COMPLEX) { ERROR (CHOLMOD_INVALID, "L and X must both be complex") ; return FALSE ; }
        if (X->nrow != L->n || X->d < X->nrow) { ERROR (CHOLMOD_INVALID, "X and L dimensions must match") ; return FALSE ; }
        L = (cholmod_factor *) L->cx_twin ;
        ldx *= 2 ;
    }
    else
    {
        if (X->xtype != CHOLMOD_REAL) { ERROR (CHOLMOD_INVALID, "X must be real") ; return FALSE ; }
        if (X->nrow != L->n || X->d < X->nrow) { ERROR (CHOLMOD_INVALID, "X and L dimensions must match") ; return FALSE ; }
    }
    Common->status = CHOLMOD_OK ;
    if (solve_on_host (L, Common))
    {
        ssamd_cpu_super_solve (which, L, X->x, (Int) X->ncol, ldx) ;
        return Common->blas_ok ;
    }
    if (!factor_on_device (L, Common)) return FALSE ;
    int rc = cholmod_hip_solve ((cholmod_hip_plan *) L->hip_plan, which, X->x, (int64_t) X->ncol, (int64_t) ldx) ;
    if (rc != CHOLMOD_HIP_OK) return map_hip_status (rc, Common, "HIP solve failed") ;
    return Common->blas_ok ;
}

/* reference: Supernodal/cholmod_super_solve.c:43-134 / :136-231.  E (workspace
 * in the reference) is accepted and ignored. */
int cholmod_l_super_lsolve (cholmod_factor *L, cholmod_dense *X, cholmod_dense *E, cholmod_common *Common)
{
    (void) E ;
    return super_solve (1, L, X, Common) ;
}

int cholmod_l_super_ltsolve (cholmod_factor *L, cholmod_dense *X, cholmod_dense *E, cholmod_common *Common)
{
    (void) E ;
    return super_solve (2, L, X, Common) ;
}

/* reference: Cholesky/cholmod_solve.c:946-1030 and the supernodal branch of
 * solve2 (:1541-1580): Y = P B, L solves, X = P' Y.  LL' factors only, so the
 * LDL' system codes collapse as in the reference (D = I). */
cholmod_dense *cholmod_l_solve (int sys, cholmod_factor *L, cholmod_dense *B, cholmod_common *Common)
{
    cholmod_dense *X = NULL ;
    if (!cholmod_l_solve2 (sys, L, B, NULL, &X, NULL, NULL, NULL, Common))
        cholmod_l_free_dense (&X, Common) ;
    return X ;
}

/* entry (k, r) of a dense matrix of any numeric kind */
static inline void dense_get (const cholmod_dense *B, Int k, Int r, double *re, double *im)
{
    size_t q = (size_t) k + (size_t) r * B->d ;
    const double *x = B->x ;
    if (B->xtype == CHOLMOD_COMPLEX) { *re = x [2*q] ; *im = x [2*q+1] ; }
    else { *re = x [q] ; *im = (B->xtype == CHOLMOD_ZOMPLEX) ? ((const double *) B->z) [q] : 0.0 ; }
}

static inline void dense_put (cholmod_dense *X, Int k, Int r, double re, double im)
{
    size_t q = (size_t) k + (size_t) r * X->d ;
    double *x = X->x ;
    if (X->xtype == CHOLMOD_COMPLEX) { x [2*q] = re ; x [2*q+1] = im ; }
    else { x [q] = re ; if (X->xtype == CHOLMOD_ZOMPLEX) ((double *) X->z) [q] = im ; }
}

int cholmod_l_solve2 (int sys, cholmod_factor *L, cholmod_dense *B, cholmod_sparse *Bset,
    cholmod_dense **X_Handle, cholmod_sparse **Xset_Handle, cholmod_dense **Y_Handle,
    cholmod_dense **E_Handle, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    RETURN_IF_NULL (L, FALSE) ;
    RETURN_IF_NULL (B, FALSE) ;
    RETURN_IF_NULL (X_Handle, FALSE) ;
    (void) E_Handle ;
    if (sys < CHOLMOD_A || sys > CHOLMOD_Pt) { ERROR (CHOLMOD_INVALID, "invalid system") ; return FALSE ; }
    if (B->xtype < CHOLMOD_REAL || B->xtype > CHOLMOD_ZOMPLEX) { ERROR (CHOLMOD_INVALID, "B must be numeric") ; return FALSE ; }
    if (B->d < L->n || B->nrow != L->n) { ERROR (CHOLMOD_INVALID, "dimensions of L and B do not match") ; return FALSE ; }
    if (Bset)
    {
        /* (reference cholmod_solve.c:1081-1094) */
        if (B->ncol != 1) { ERROR (CHOLMOD_INVALID, "Bset requires a single right-hand side") ; return FALSE ; }
        if (L->xtype != B->xtype) { ERROR (CHOLMOD_INVALID, "Bset requires xtype of L and B to match") ; return FALSE ; }
    }
    Common->status = CHOLMOD_OK ;
    Int n = (Int) L->n, nrhs = (Int) B->ncol ;
    const int Lcomplex = (L->xtype == CHOLMOD_COMPLEX) ;
    const int Bcomplex = (B->xtype != CHOLMOD_REAL) ;
    /* kind of X (reference cholmod_solve.c:1112-1134): real if L and B are real (P, Pt:
     * if B is real), else the preferred complex kind */
    const int ctype = Common->prefer_zomplex ? CHOLMOD_ZOMPLEX : CHOLMOD_COMPLEX ;
    int xtype = (sys == CHOLMOD_P || sys == CHOLMOD_Pt) ? (Bcomplex ? ctype : CHOLMOD_REAL)
              : (!Lcomplex && !Bcomplex) ? CHOLMOD_REAL : ctype ;
    cholmod_dense *X = *X_Handle ;
    if (!X || X->nrow != (size_t) n || X->ncol != (size_t) nrhs || X->xtype != xtype)
    {
        cholmod_l_free_dense (X_Handle, Common) ;
        X = cholmod_l_allocate_dense (n, nrhs, n, xtype, Common) ;
        if (!X) return FALSE ;
        *X_Handle = X ;
    }
    /* a sparse right-hand side: the entries of X on the pattern Bset reaches, subset_solve.c */
    if (Bset) return ssamd_solve_subset (sys, L, B, Bset, X, Xset_Handle, Y_Handle, Common) ;
    const Int *Perm = L->Perm ;
    double re, im ;
    if (sys == CHOLMOD_P || sys == CHOLMOD_Pt || sys == CHOLMOD_D)
    {
        for (Int r = 0 ; r < nrhs ; r++)
            for (Int k = 0 ; k < n ; k++)
            {
                dense_get (B, sys == CHOLMOD_P ? Perm [k] : k, r, &re, &im) ;
                dense_put (X, sys == CHOLMOD_Pt ? Perm [k] : k, r, re, im) ;
            }
        return TRUE ;
    }
    int which = (sys == CHOLMOD_A || sys == CHOLMOD_LDLt) ? 0
              : (sys == CHOLMOD_L || sys == CHOLMOD_LD) ? 1 : 2 ;
    /* the system the engine sees: a complex L is the real twin of order 2n acting on
     * interleaved vectors (complex.c); a real L against a complex B solves the real and
     * the imaginary parts as 2 nrhs real right-hand sides (the reference's "dual" Y,
     * cholmod_solve.c:1553) */
    cholmod_factor *Lr = L ;
    if (Lcomplex)
    {
        if (!L->cx_twin) { ERROR (CHOLMOD_INVALID, "complex L without its engine factor") ; return FALSE ; }
        Lr = (cholmod_factor *) L->cx_twin ;
    }
    const Int yn = Lcomplex ? 2 * n : n ;
    const Int ycols = (!Lcomplex && Bcomplex) ? 2 * nrhs : nrhs ;
    int on_host = solve_on_host (Lr, Common) ;
    if (!on_host && !factor_on_device (Lr, Common)) return FALSE ;
    cholmod_dense *Y = cholmod_l_allocate_dense (yn, ycols, yn, CHOLMOD_REAL, Common) ;
    if (!Y) return FALSE ;
    double *Yx = Y->x ;
    const int nth = ssamd_host_threads () ;
    for (Int r = 0 ; r < nrhs ; r++)
    {
#pragma omp parallel for schedule(static) num_threads(nth) if (n > 100000)
        for (Int k = 0 ; k < n ; k++)
        {
            double re, im ;
            dense_get (B, sys == CHOLMOD_A ? Perm [k] : k, r, &re, &im) ;
            if (Lcomplex) { Yx [2*k + r*yn] = re ; Yx [2*k+1 + r*yn] = im ; }
            else { Yx [k + r*yn] = re ; if (Bcomplex) Yx [k + (nrhs + r)*yn] = im ; }
        }
    }
    int ok = TRUE ;
    if (on_host) ssamd_cpu_super_solve (which, Lr, Yx, ycols, yn) ;
    else
    {
        int rc = cholmod_hip_solve ((cholmod_hip_plan *) Lr->hip_plan, which, Yx, ycols, yn) ;
        ok = (rc == CHOLMOD_HIP_OK) ? TRUE : map_hip_status (rc, Common, "HIP solve failed") ;
    }
    if (ok)
        for (Int r = 0 ; r < nrhs ; r++)
        {
#pragma omp parallel for schedule(static) num_threads(nth) if (n > 100000)
            for (Int k = 0 ; k < n ; k++)
            {
                double re, im ;
                if (Lcomplex) { re = Yx [2*k + r*yn] ; im = Yx [2*k+1 + r*yn] ; }
                else { re = Yx [k + r*yn] ; im = Bcomplex ? Yx [k + (nrhs + r)*yn] : 0.0 ; }
                dense_put (X, sys == CHOLMOD_A ? Perm [k] : k, r, re, im) ;
            }
        }
    cholmod_l_free_dense (&Y, Common) ;
    return ok ;
}
