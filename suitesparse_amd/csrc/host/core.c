/* core.c -- objects, memory and permuted transposes of the host layer.
 * Mirrors the part of the reference's Core module that the supernodal path
 * uses (reference files: CHOLMOD/Core/cholmod_common.c, cholmod_memory.c,
 * cholmod_sparse.c, cholmod_dense.c, cholmod_triplet.c, cholmod_transpose.c,
 * cholmod_factor.c, cholmod_error.c).  64-bit indices, real double only. */
#include "host_internal.h"
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- error / common ----------------------------------------------------------- */

/* reference: Core/cholmod_error.c:31-81 */
int cholmod_l_error (int status, const char *file, int line, const char *message,
    cholmod_common *Common)
{
    if (!Common) return FALSE ;
    Common->status = status ;
    if (!Common->try_catch)
    {
        if (Common->print > 1 && status != CHOLMOD_OK)
        {
            fprintf (stderr, "CHOLMOD %s: %s (file %s line %d)\n",
                status < 0 ? "error" : "warning", message ? message : "", file ? file : "?", line) ;
        }
        if (Common->error_handler) Common->error_handler (status, file, line, message) ;
    }
    return TRUE ;
}

/* reference: Core/cholmod_common.c:244-330 */
int cholmod_l_defaults (cholmod_common *Common)
{
    if (!Common) return FALSE ;
    Common->final_asis = TRUE ; Common->final_super = TRUE ; Common->final_ll = FALSE ;
    Common->final_pack = TRUE ; Common->final_monotonic = TRUE ; Common->final_resymbol = FALSE ;
    Common->supernodal = CHOLMOD_AUTO ;
    Common->supernodal_switch = 40 ;
    Common->nrelax [0] = 4 ; Common->nrelax [1] = 16 ; Common->nrelax [2] = 48 ;
    Common->zrelax [0] = 0.8 ; Common->zrelax [1] = 0.1 ; Common->zrelax [2] = 0.05 ;
    Common->prefer_upper = TRUE ;
    Common->quick_return_if_not_posdef = FALSE ;
    Common->print = 3 ;
    Common->nmethods = 0 ;
    Common->current = 0 ; Common->selected = 0 ;
    for (int i = 0 ; i <= CHOLMOD_MAXMETHODS ; i++)
    {
        Common->method [i].ordering = CHOLMOD_AMD ;
        Common->method [i].fl = EMPTY ;
        Common->method [i].lnz = EMPTY ;
    }
    Common->method [0].ordering = CHOLMOD_GIVEN ;
    Common->method [1].ordering = CHOLMOD_AMD ;
    Common->method [2].ordering = CHOLMOD_METIS ;
    Common->postorder = TRUE ;
    Common->useGPU = EMPTY ;
    return TRUE ;
}

/* reference: Core/cholmod_common.c:58-242 (useGPU = EMPTY for the long build :371) */
int cholmod_l_start (cholmod_common *Common)
{
    if (!Common) return FALSE ;
    memset (Common, 0, sizeof (cholmod_common)) ;
    Common->itype = CHOLMOD_LONG ;
    Common->dtype = CHOLMOD_DOUBLE ;
    cholmod_l_defaults (Common) ;
    Common->try_catch = FALSE ;
    Common->error_handler = NULL ;
    Common->status = CHOLMOD_OK ;
    Common->blas_ok = TRUE ;
    Common->fl = EMPTY ; Common->lnz = EMPTY ; Common->anz = EMPTY ; Common->modfl = EMPTY ;
    Common->maxGpuMemFraction = 0.0 ;
    Common->maxGpuMemBytes = 0 ;
    {
        const char *e = getenv ("CHOLMOD_HIP_CPU_FALLBACK") ;
        Common->hip_cpu_fallback = (e && atoi (e) != 0) ? 1 : 0 ;
    }
    return TRUE ;
}

/* reference: Core/cholmod_common.c:415-420 (frees workspace; none is cached here) */
int cholmod_l_finish (cholmod_common *Common)
{
    if (!Common) return FALSE ;
    cholmod_l_gpu_deallocate (Common) ;
    return TRUE ;
}

/* ---- SuiteSparse_config (reference: SuiteSparse_config/SuiteSparse_config.c:57-330) --- */

static int ssc_divcomplex (double ar, double ai, double br, double bi, double *cr, double *ci)
{
    /* (ar + i ai) / (br + i bi), Smith's method as the reference (SuiteSparse_config.c:455-527); 1 if the divisor is zero */
    double tr, ti, r, den ;
    if (fabs (br) >= fabs (bi))
    {
        r = bi / br ; den = br + r * bi ;
        tr = (ar + ai * r) / den ; ti = (ai - ar * r) / den ;
    }
    else
    {
        r = br / bi ; den = r * br + bi ;
        tr = (ar * r + ai) / den ; ti = (ai * r - ar) / den ;
    }
    *cr = tr ; *ci = ti ;
    return den == 0.0 ;
}

struct SuiteSparse_config_struct SuiteSparse_config = { malloc, calloc, realloc, free, printf, hypot, ssc_divcomplex } ;

void SuiteSparse_start (void)
{
    SuiteSparse_config.malloc_func = malloc ; SuiteSparse_config.calloc_func = calloc ;
    SuiteSparse_config.realloc_func = realloc ; SuiteSparse_config.free_func = free ;
    SuiteSparse_config.printf_func = printf ; SuiteSparse_config.hypot_func = hypot ;
    SuiteSparse_config.divcomplex_func = ssc_divcomplex ;
}
void SuiteSparse_finish (void) { }

static int ssc_too_large (size_t nitems, size_t size)
{
    return size == 0 || nitems >= SIZE_MAX / size || nitems >= (size_t) INT64_MAX / size ;
}
void *SuiteSparse_malloc (size_t nitems, size_t size_of_item)
{
    if (nitems < 1) nitems = 1 ;
    if (size_of_item < 1) size_of_item = 1 ;
    if (ssc_too_large (nitems, size_of_item)) return NULL ;
    return SuiteSparse_config.malloc_func (nitems * size_of_item) ;
}
void *SuiteSparse_calloc (size_t nitems, size_t size_of_item)
{
    if (nitems < 1) nitems = 1 ;
    if (size_of_item < 1) size_of_item = 1 ;
    if (ssc_too_large (nitems, size_of_item)) return NULL ;
    return SuiteSparse_config.calloc_func (nitems, size_of_item) ;
}
void *SuiteSparse_realloc (size_t nitems_new, size_t nitems_old, size_t size_of_item, void *p, int *ok)
{
    if (nitems_new < 1) nitems_new = 1 ;
    if (nitems_old < 1) nitems_old = 1 ;
    if (size_of_item < 1) size_of_item = 1 ;
    if (ssc_too_large (nitems_new, size_of_item)) { *ok = 0 ; return p ; }
    if (!p)
    {
        p = SuiteSparse_malloc (nitems_new, size_of_item) ;
        *ok = p != NULL ;
        return p ;
    }
    if (nitems_old == nitems_new) { *ok = 1 ; return p ; }
    void *pnew = SuiteSparse_config.realloc_func (p, nitems_new * size_of_item) ;
    if (!pnew)
    {
        /* a failed shrink leaves the (larger) old block in place and counts as done (SuiteSparse_config.c:266-278) */
        *ok = nitems_new < nitems_old ;
        return p ;
    }
    *ok = 1 ;
    return pnew ;
}
void *SuiteSparse_free (void *p)
{
    if (p) SuiteSparse_config.free_func (p) ;
    return NULL ;
}
double SuiteSparse_hypot (double x, double y) { return SuiteSparse_config.hypot_func (x, y) ; }
int SuiteSparse_divcomplex (double ar, double ai, double br, double bi, double *cr, double *ci)
{
    return SuiteSparse_config.divcomplex_func (ar, ai, br, bi, cr, ci) ;
}

/* ---- memory (reference: Core/cholmod_memory.c:111-230) ------------------------- */

void *cholmod_l_malloc (size_t n, size_t size, cholmod_common *Common)
{
    if (!Common) return NULL ;
    if (size == 0) { ERROR (CHOLMOD_INVALID, "sizeof(item) must be > 0") ; return NULL ; }
    if (n >= (SIZE_MAX / size) || n >= (size_t) INT64_MAX / size)
    {
        ERROR (CHOLMOD_TOO_LARGE, "problem too large") ;
        return NULL ;
    }
    void *p = SuiteSparse_malloc (n, size) ;
    if (!p) { ERROR (CHOLMOD_OUT_OF_MEMORY, "out of memory") ; return NULL ; }
    Common->malloc_count++ ;
    Common->memory_inuse += n * size ;
    if (Common->memory_inuse > Common->memory_usage) Common->memory_usage = Common->memory_inuse ;
    return p ;
}

void *cholmod_l_calloc (size_t n, size_t size, cholmod_common *Common)
{
    if (!Common) return NULL ;
    if (size == 0) { ERROR (CHOLMOD_INVALID, "sizeof(item) must be > 0") ; return NULL ; }
    if (n >= (SIZE_MAX / size) || n >= (size_t) INT64_MAX / size)
    {
        ERROR (CHOLMOD_TOO_LARGE, "problem too large") ;
        return NULL ;
    }
    void *p = SuiteSparse_calloc (n, size) ;
    if (!p) { ERROR (CHOLMOD_OUT_OF_MEMORY, "out of memory") ; return NULL ; }
    Common->malloc_count++ ;
    Common->memory_inuse += n * size ;
    if (Common->memory_inuse > Common->memory_usage) Common->memory_usage = Common->memory_inuse ;
    return p ;
}

/* reference: Core/cholmod_memory.c:232-330.  *n is the current size on input and the new size on output (unchanged on
 * failure, when the old block is returned intact). */
void *cholmod_l_realloc (size_t nnew, size_t size, void *p, size_t *n, cholmod_common *Common)
{
    if (!Common) return NULL ;
    if (size == 0) { ERROR (CHOLMOD_INVALID, "sizeof(item) must be > 0") ; return p ; }
    if (!p)
    {
        p = cholmod_l_malloc (nnew, size, Common) ;
        *n = p ? nnew : 0 ;
        return p ;
    }
    if (*n == nnew) return p ;
    if (nnew >= (SIZE_MAX / size) || nnew >= (size_t) INT64_MAX / size)
    {
        ERROR (CHOLMOD_TOO_LARGE, "problem too large") ;
        return p ;
    }
    int ok = 1 ;
    void *pnew = SuiteSparse_realloc (nnew, *n, size, p, &ok) ;
    if (!ok) { ERROR (CHOLMOD_OUT_OF_MEMORY, "out of memory") ; return p ; }
    Common->memory_inuse += (nnew - *n) * size ;
    if (Common->memory_inuse > Common->memory_usage) Common->memory_usage = Common->memory_inuse ;
    *n = nnew ;
    return pnew ;
}

void *cholmod_l_free (size_t n, size_t size, void *p, cholmod_common *Common)
{
    if (!Common) return NULL ;
    if (p)
    {
        SuiteSparse_free (p) ;
        Common->malloc_count-- ;
        Common->memory_inuse -= n * size ;
    }
    return NULL ;
}

/* ---- sparse (reference: Core/cholmod_sparse.c) --------------------------------- */

cholmod_sparse *cholmod_l_allocate_sparse (size_t nrow, size_t ncol, size_t nzmax,
    int sorted, int packed, int stype, int xtype, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    if (stype != 0 && nrow != ncol)
    {
        ERROR (CHOLMOD_INVALID, "rectangular matrix with stype != 0 invalid") ;
        return NULL ;
    }
    if (xtype < CHOLMOD_PATTERN || xtype > CHOLMOD_ZOMPLEX)
    {
        ERROR (CHOLMOD_INVALID, "xtype invalid") ;
        return NULL ;
    }
    Common->status = CHOLMOD_OK ;
    cholmod_sparse *A = cholmod_l_calloc (1, sizeof (cholmod_sparse), Common) ;
    if (!A) return NULL ;
    nzmax = nzmax > 1 ? nzmax : 1 ;
    A->nrow = nrow ; A->ncol = ncol ; A->nzmax = nzmax ;
    A->packed = packed ; A->stype = stype ;
    A->itype = CHOLMOD_LONG ; A->xtype = xtype ; A->dtype = CHOLMOD_DOUBLE ;
    A->sorted = (nrow <= 1) ? TRUE : sorted ;
    A->p = cholmod_l_calloc (ncol + 1, sizeof (Int), Common) ;
    if (!packed) A->nz = cholmod_l_calloc (ncol > 0 ? ncol : 1, sizeof (Int), Common) ;
    A->i = cholmod_l_malloc (nzmax, sizeof (Int), Common) ;
    /* complex: interleaved (re, im) pairs in x; zomplex: real parts in x, imaginary in z
     * (reference Core/cholmod_complex.c) */
    if (xtype != CHOLMOD_PATTERN) A->x = cholmod_l_malloc (nzmax, SSAMD_XENT (xtype) * sizeof (double), Common) ;
    if (xtype == CHOLMOD_ZOMPLEX) A->z = cholmod_l_malloc (nzmax, sizeof (double), Common) ;
    if (Common->status < CHOLMOD_OK) { cholmod_l_free_sparse (&A, Common) ; return NULL ; }
    return A ;
}

int cholmod_l_free_sparse (cholmod_sparse **AH, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    if (!AH || !*AH) return TRUE ;
    cholmod_sparse *A = *AH ;
    cholmod_l_free (A->ncol + 1, sizeof (Int), A->p, Common) ;
    if (A->nz) cholmod_l_free (A->ncol > 0 ? A->ncol : 1, sizeof (Int), A->nz, Common) ;
    cholmod_l_free (A->nzmax, sizeof (Int), A->i, Common) ;
    if (A->x) cholmod_l_free (A->nzmax, SSAMD_XENT (A->xtype) * sizeof (double), A->x, Common) ;
    if (A->z) cholmod_l_free (A->nzmax, sizeof (double), A->z, Common) ;
    cholmod_l_free (1, sizeof (cholmod_sparse), A, Common) ;
    *AH = NULL ;
    return TRUE ;
}

SuiteSparse_long cholmod_l_nnz (cholmod_sparse *A, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (EMPTY) ;
    RETURN_IF_NULL (A, EMPTY) ;
    Int *Ap = A->p, *Anz = A->nz ;
    if (A->packed) return Ap [A->ncol] ;
    Int nz = 0 ;
    for (size_t j = 0 ; j < A->ncol ; j++) nz += Anz [j] > 0 ? Anz [j] : 0 ;
    return nz ;
}

cholmod_sparse *cholmod_l_copy_sparse (cholmod_sparse *A, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (A, NULL) ;
    cholmod_sparse *C = cholmod_l_allocate_sparse (A->nrow, A->ncol, A->nzmax, A->sorted,
        A->packed, A->stype, A->xtype, Common) ;
    if (!C) return NULL ;
    memcpy (C->p, A->p, (A->ncol + 1) * sizeof (Int)) ;
    if (!A->packed) memcpy (C->nz, A->nz, A->ncol * sizeof (Int)) ;
    Int *Ap = A->p, *Anz = A->nz ;
    for (size_t j = 0 ; j < A->ncol ; j++)
    {
        Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j] ;
        if (pend > p)
        {
            memcpy ((Int *) C->i + p, (Int *) A->i + p, (pend - p) * sizeof (Int)) ;
            size_t e = SSAMD_XENT (A->xtype) ;
            if (A->x) memcpy ((double *) C->x + e * p, (double *) A->x + e * p, (pend - p) * e * sizeof (double)) ;
            if (A->z) memcpy ((double *) C->z + p, (double *) A->z + p, (pend - p) * sizeof (double)) ;
        }
    }
    return C ;
}

/* ---- transposes (reference: Core/cholmod_transpose.c:871-1139) ----------------- */

/* Threads for the memory-bound analysis loops (scatter with atomics, sorts of
 * short columns): they stop scaling at a few dozen and lose beyond a socket
 * (EPYC 9575F x 2, 256 hardware threads: 4x slower than with 32), so the OpenMP
 * default is capped; CHOLMOD_HOST_THREADS overrides. */
/* CPUs' worth of time the container may use (cgroup v2 cpu.max, v1 cfs quota), 0 = no quota.  The GPU boxes of this
 * project show 256 hardware threads and a quota of 16: every thread beyond the quota is a thread that gets throttled --
 * which is why, for five rounds, every host loop of this library measured best at 16 threads and worse beyond (round 6). */
int ssamd_cpu_quota (void)
{
    static int cached = -1 ;
    if (cached >= 0) return cached ;
    int q = 0 ;
    FILE *f = fopen ("/sys/fs/cgroup/cpu.max", "r") ;
    if (f)
    {
        char a [64] = {0} ; double per = 0 ;
        if (fscanf (f, "%63s %lf", a, &per) == 2 && strcmp (a, "max") != 0 && per > 0) q = (int) ceil (atof (a) / per) ;
        fclose (f) ;
    }
    else
    {
        double qu = -1, pe = 0 ;
        FILE *g = fopen ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *h = fopen ("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r") ;
        if (g && h && fscanf (g, "%lf", &qu) == 1 && fscanf (h, "%lf", &pe) == 1 && qu > 0 && pe > 0) q = (int) ceil (qu / pe) ;
        if (g) fclose (g) ;
        if (h) fclose (h) ;
    }
    cached = q > 0 ? q : 0 ;
    return cached ;
}

/* the OpenMP default, not beyond the container's CPU quota unless the caller asked for a number (OMP_NUM_THREADS) */
static int omp_threads_within_quota (void)
{
    int nt = 1 ;
#ifdef _OPENMP
    nt = omp_get_max_threads () ;
#endif
    const int q = ssamd_cpu_quota () ;
    if (q > 0 && nt > q && !getenv ("OMP_NUM_THREADS")) nt = q ;
    return nt > 0 ? nt : 1 ;
}

int ssamd_host_threads (void)
{
    const char *e = getenv ("CHOLMOD_HOST_THREADS") ;
    if (e && atoi (e) > 0) return atoi (e) ;
    const int nt = omp_threads_within_quota () ;
    return nt > 32 ? 32 : nt ;
}

/* threads the caller allows (OMP_NUM_THREADS / the OpenMP default), not capped: the dense calls of the CPU path */
int ssamd_host_threads_uncapped (void)
{
    return omp_threads_within_quota () ;
}

/* Symmetric permutation C = P A P' (Perm may be NULL), upper_out selects the
 * triangle C is stored in -- independent of the triangle A is stored in, so a
 * lower-stored A goes to a lower-stored permuted S in one pass (the reference
 * takes two transposes for that, Cholesky/cholmod_factorize.c:233-244). */
cholmod_sparse *ssamd_sym_permute (cholmod_sparse *A, int values, SuiteSparse_long *Perm, int upper_out,
    cholmod_common *Common)
{
    return ssamd_sym_permute_src (A, values, Perm, upper_out, NULL, Common) ;
}

/* The same; *src_out (if not NULL) receives, for every entry q of C, the position in A it
 * came from (nnz(C) entries, cholmod_l_malloc'ed, the caller frees): the value map that lets
 * a later factorization of a matrix with the same pattern skip this permutation
 * (cholmod_l_factorize_p, cholmod_hip_set_value_map). */
cholmod_sparse *ssamd_sym_permute_src (cholmod_sparse *A, int values, SuiteSparse_long *Perm, int upper_out,
    SuiteSparse_long **src_out, cholmod_common *Common)
{
    /* C = P A P' of a symmetric A with the wanted triangle stored.  Three parallel passes
     * (count, scatter with atomic cursors, per-column sort by (row, source
     * position)), so the output is deterministic and its columns are sorted. */
    Int n = (Int) A->nrow ;
    Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    double *Ax = A->x, *Az = A->z ;
    int xtype = values ? A->xtype : CHOLMOD_PATTERN ;
    const double csign = (values == 2) ? -1.0 : 1.0 ;     /* 2: conjugate transpose, 1: array transpose */
    int upper_in = A->stype > 0 ;
    const int packed = A->packed ;
    const int nth = ssamd_host_threads () ;
    Int *Pinv = NULL ;
    if (Perm)
    {
        Pinv = cholmod_l_malloc (n > 0 ? n : 1, sizeof (Int), Common) ;
        if (!Pinv) return NULL ;
        int bad = 0 ;
#pragma omp parallel for schedule(static) num_threads(nth)
        for (Int k = 0 ; k < n ; k++) Pinv [k] = EMPTY ;
#pragma omp parallel for schedule(static) num_threads(nth) reduction(|:bad)
        for (Int k = 0 ; k < n ; k++)
        {
            Int j = Perm [k] ;
            if (j < 0 || j >= n || !__sync_bool_compare_and_swap (&Pinv [j], (Int) EMPTY, k)) bad |= 1 ;
        }
        if (bad)
        {
            cholmod_l_free (n > 0 ? n : 1, sizeof (Int), Pinv, Common) ;
            ERROR (CHOLMOD_INVALID, "invalid permutation") ;
            return NULL ;
        }
    }
    Int *cursor = cholmod_l_malloc (n + 1, sizeof (Int), Common) ;
    if (!cursor) { if (Pinv) cholmod_l_free (n > 0 ? n : 1, sizeof (Int), Pinv, Common) ; return NULL ; }
#pragma omp parallel for schedule(static) num_threads(nth)
    for (Int j = 0 ; j <= n ; j++) cursor [j] = 0 ;
    /* pass 1: entries per output column */
#pragma omp parallel for schedule(static) num_threads(nth)
    for (Int j = 0 ; j < n ; j++)
    {
        Int p = Ap [j], pend = packed ? Ap [j+1] : p + Anz [j] ;
        Int c = Pinv ? Pinv [j] : j ;
        for ( ; p < pend ; p++)
        {
            Int i = Ai [p] ;
            if (upper_in ? (i > j) : (i < j)) continue ;    /* ignored triangle */
            Int r = Pinv ? Pinv [i] : i ;
            Int lo = r < c ? r : c, hi = r < c ? c : r ;
            Int col = upper_out ? hi : lo ;
#pragma omp atomic
            cursor [col]++ ;
        }
    }
    Int nz = 0 ;
    for (Int j = 0 ; j < n ; j++) { Int c = cursor [j] ; cursor [j] = nz ; nz += c ; }
    cursor [n] = nz ;
    cholmod_sparse *C = cholmod_l_allocate_sparse (n, n, nz, TRUE, TRUE, upper_out ? 1 : -1, xtype, Common) ;
    Int *src = C ? cholmod_l_malloc (nz > 0 ? nz : 1, sizeof (Int), Common) : NULL ;
    if (C && src)
    {
        Int *Cp = C->p, *Ci = C->i ;
        double *Cx = C->x ;
#pragma omp parallel for schedule(static) num_threads(nth)
        for (Int j = 0 ; j <= n ; j++) Cp [j] = cursor [j] ;
        /* pass 2: scatter (row, source position) */
#pragma omp parallel for schedule(static) num_threads(nth)
        for (Int j = 0 ; j < n ; j++)
        {
            Int p = Ap [j], pend = packed ? Ap [j+1] : p + Anz [j] ;
            Int c = Pinv ? Pinv [j] : j ;
            for ( ; p < pend ; p++)
            {
                Int i = Ai [p] ;
                if (upper_in ? (i > j) : (i < j)) continue ;
                Int r = Pinv ? Pinv [i] : i ;
                Int lo = r < c ? r : c, hi = r < c ? c : r ;
                Int row = upper_out ? lo : hi, col = upper_out ? hi : lo ;
                Int q ;
#pragma omp atomic capture
                q = cursor [col]++ ;
                /* (source position, moved to the other triangle = transposed; the unpermuted transpose of the reference
                 * conjugates the diagonal too, the permuted one does not: Core/t_cholmod_transpose.c:199-215, :268-277) */
                Ci [q] = row ; src [q] = (p << 1) | (row != r || (!Pinv && r == c && upper_in != upper_out)) ;
            }
        }
        /* pass 3: every column sorted by (row, source position), values gathered */
#pragma omp parallel for schedule(dynamic, 4096) num_threads(nth)
        for (Int j = 0 ; j < n ; j++)
        {
            Int b0 = Cp [j], e0 = Cp [j+1] ;
            if (e0 - b0 > 48)
            {
                /* long column (dense stencils: hundreds of entries): heap sort in place */
                Int len = e0 - b0 ;
                Int *R = Ci + b0, *Sq = src + b0 ;
#define SSAMD_LESS(x, y) (R [x] < R [y] || (R [x] == R [y] && Sq [x] < Sq [y]))
#define SSAMD_SWAP(x, y) do { Int t_ = R [x] ; R [x] = R [y] ; R [y] = t_ ; t_ = Sq [x] ; Sq [x] = Sq [y] ; Sq [y] = t_ ; } while (0)
                for (Int start = len / 2 - 1 ; start >= 0 ; start--)
                    for (Int root = start ; ; )
                    {
                        Int ch = 2 * root + 1 ;
                        if (ch >= len) break ;
                        if (ch + 1 < len && SSAMD_LESS (ch, ch + 1)) ch++ ;
                        if (!SSAMD_LESS (root, ch)) break ;
                        SSAMD_SWAP (root, ch) ; root = ch ;
                    }
                for (Int end = len - 1 ; end > 0 ; end--)
                {
                    SSAMD_SWAP (0, end) ;
                    for (Int root = 0 ; ; )
                    {
                        Int ch = 2 * root + 1 ;
                        if (ch >= end) break ;
                        if (ch + 1 < end && SSAMD_LESS (ch, ch + 1)) ch++ ;
                        if (!SSAMD_LESS (root, ch)) break ;
                        SSAMD_SWAP (root, ch) ; root = ch ;
                    }
                }
#undef SSAMD_LESS
#undef SSAMD_SWAP
            }
            else for (Int q = b0 + 1 ; q < e0 ; q++)
            {
                Int r = Ci [q], sp = src [q], t = q ;
                while (t > b0 && (Ci [t-1] > r || (Ci [t-1] == r && src [t-1] > sp)))
                { Ci [t] = Ci [t-1] ; src [t] = src [t-1] ; t-- ; }
                Ci [t] = r ; src [t] = sp ;
            }
            if (xtype == CHOLMOD_REAL) for (Int q = b0 ; q < e0 ; q++) Cx [q] = Ax [src [q] >> 1] ;
            else if (xtype == CHOLMOD_COMPLEX)
                for (Int q = b0 ; q < e0 ; q++)
                {
                    Int p = src [q] >> 1 ;
                    Cx [2*q] = Ax [2*p] ;
                    Cx [2*q+1] = (src [q] & 1) ? csign * Ax [2*p+1] : Ax [2*p+1] ;
                }
            else if (xtype == CHOLMOD_ZOMPLEX)
                for (Int q = b0 ; q < e0 ; q++)
                {
                    Int p = src [q] >> 1 ;
                    Cx [q] = Ax [p] ;
                    ((double *) C->z) [q] = (src [q] & 1) ? csign * Az [p] : Az [p] ;
                }
        }
    }
    else if (C) cholmod_l_free_sparse (&C, Common) ;
    if (src_out) *src_out = NULL ;
    if (src && C && src_out)
    {
#pragma omp parallel for schedule(static) num_threads(nth)
        for (Int q = 0 ; q < nz ; q++) src [q] >>= 1 ;      /* drop the transposed-entry flag */
        *src_out = src ;
        src = NULL ;
    }
    if (src) cholmod_l_free (nz > 0 ? nz : 1, sizeof (Int), src, Common) ;
    cholmod_l_free (n + 1, sizeof (Int), cursor, Common) ;
    if (Pinv) cholmod_l_free (n > 0 ? n : 1, sizeof (Int), Pinv, Common) ;
    return C ;
}

/* reference: Core/cholmod_transpose.c:871.  Symmetric case: C = A(p,p)' with the
 * other triangle stored (stype flips), columns sorted.  Unsymmetric case: plain
 * transpose (Perm/fset are not supported for stype == 0 in this build). */
cholmod_sparse *cholmod_l_ptranspose (cholmod_sparse *A, int values, SuiteSparse_long *Perm,
    SuiteSparse_long *fset, size_t fsize, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (A, NULL) ;
    (void) fsize ;
    Common->status = CHOLMOD_OK ;
    Int n = (Int) A->nrow, ncol = (Int) A->ncol ;
    Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    double *Ax = A->x, *Az = A->z ;
    int xtype = values ? A->xtype : CHOLMOD_PATTERN ;
    const double csign = (values == 2) ? -1.0 : 1.0 ;
    if (A->stype == 0)
    {
        if (Perm || fset)
        {
            ERROR (CHOLMOD_NOT_INSTALLED, "permuted unsymmetric transpose not built") ;
            return NULL ;
        }
        Int nz = cholmod_l_nnz (A, Common) ;
        cholmod_sparse *C = cholmod_l_allocate_sparse (ncol, n, nz, TRUE, TRUE, 0, xtype, Common) ;
        if (!C) return NULL ;
        Int *Cp = C->p, *Ci = C->i ;
        double *Cx = C->x ;
        Int *w = cholmod_l_calloc (n + 1, sizeof (Int), Common) ;
        if (!w) { cholmod_l_free_sparse (&C, Common) ; return NULL ; }
        for (Int j = 0 ; j < ncol ; j++)
        {
            Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j] ;
            for ( ; p < pend ; p++) w [Ai [p]]++ ;
        }
        Cp [0] = 0 ;
        for (Int i = 0 ; i < n ; i++) { Cp [i+1] = Cp [i] + w [i] ; w [i] = Cp [i] ; }
        for (Int j = 0 ; j < ncol ; j++)
        {
            Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j] ;
            for ( ; p < pend ; p++)
            {
                Int q = w [Ai [p]]++ ;
                Ci [q] = j ;
                if (xtype == CHOLMOD_REAL) Cx [q] = Ax [p] ;
                else if (xtype == CHOLMOD_COMPLEX) { Cx [2*q] = Ax [2*p] ; Cx [2*q+1] = csign * Ax [2*p+1] ; }
                else if (xtype == CHOLMOD_ZOMPLEX) { Cx [q] = Ax [p] ; ((double *) C->z) [q] = csign * Az [p] ; }
            }
        }
        cholmod_l_free (n + 1, sizeof (Int), w, Common) ;
        return C ;
    }
    return ssamd_sym_permute (A, values, Perm, !(A->stype > 0), Common) ;
}

/* A (:, f): the columns fset [0 .. fsize-1] of A, in that order (values = 0: pattern only).  The reference hands fset to its
 * transposes and its A*A' assembly (CHOLMOD/Cholesky/cholmod_analyze.c:402-418, cholmod_factorize.c:197-224: F = A(p,f)');
 * here the subset is cut out once and A(:,f)*A(:,f)' formed from it.  An index outside 0 .. ncol-1 or listed twice is
 * CHOLMOD_INVALID, as in the reference's subset check (Core/cholmod_transpose.c:520-560). */
cholmod_sparse *ssamd_column_subset (cholmod_sparse *A, const SuiteSparse_long *fset, size_t fsize, int values, cholmod_common *Common)
{
    const Int ncol = (Int) A->ncol ;
    const Int *Ap = A->p, *Ai = A->i, *Anz = A->nz ;
    const double *Ax = A->x ;
    char *seen = cholmod_l_calloc (ncol > 0 ? ncol : 1, 1, Common) ;
    if (!seen) return NULL ;
    Int nz = 0 ;
    int ok = TRUE ;
    for (size_t k = 0 ; k < fsize && ok ; k++)
    {
        const Int j = fset [k] ;
        if (j < 0 || j >= ncol || seen [j]) { ok = FALSE ; break ; }
        seen [j] = 1 ;
        nz += A->packed ? Ap [j+1] - Ap [j] : Anz [j] ;
    }
    cholmod_l_free (ncol > 0 ? ncol : 1, 1, seen, Common) ;
    if (!ok) { ERROR (CHOLMOD_INVALID, "invalid fset") ; return NULL ; }
    const int av = values && A->xtype != CHOLMOD_PATTERN && Ax ;
    cholmod_sparse *S = cholmod_l_allocate_sparse (A->nrow, fsize, nz, A->sorted, TRUE, 0, av ? A->xtype : CHOLMOD_PATTERN, Common) ;
    if (!S) return NULL ;
    Int *Sp = S->p, *Si = S->i ;
    double *Sx = S->x, *Sz = S->z ;
    const double *Az = A->z ;
    const int cx = av && A->xtype == CHOLMOD_COMPLEX, zx = av && A->xtype == CHOLMOD_ZOMPLEX ;
    Int dst = 0 ;
    for (size_t k = 0 ; k < fsize ; k++)
    {
        const Int j = fset [k] ;
        Sp [k] = dst ;
        Int p = Ap [j], pend = A->packed ? Ap [j+1] : p + Anz [j] ;
        for ( ; p < pend ; p++)
        {
            Si [dst] = Ai [p] ;
            if (cx) { Sx [2*dst] = Ax [2*p] ; Sx [2*dst+1] = Ax [2*p+1] ; }
            else if (av) { Sx [dst] = Ax [p] ; if (zx) Sz [dst] = Az [p] ; }
            dst++ ;
        }
    }
    Sp [fsize] = dst ;
    return S ;
}

/* C = A * F as a symmetric matrix, one triangle stored (lower: stype -1, else upper: stype 1), columns sorted; F = A' (or
 * A(:,f)') is taken if given, else formed.  values = 0: pattern only.  Real matrices.
 *
 * This is how this build serves the reference's unsymmetric input form (A->stype == 0: "factorize A*A' + beta*I",
 * CHOLMOD/Cholesky/cholmod_factorize.c:197-224, CHOLMOD/Supernodal/t_cholmod_super_numeric.c:223-237 and :385-418, where
 * every column of A*F is assembled on the fly): the product is FORMED once on the host (Gustavson's row-merge, one dense
 * accumulator) and then takes the symmetric path unchanged -- the same L, at the price of holding tril (A*A').
 * Reference for the product itself: CHOLMOD/Core/cholmod_aat.c. */
cholmod_sparse *ssamd_aat (cholmod_sparse *A, cholmod_sparse *F, int values, int lower, cholmod_common *Common)
{
    const Int m = (Int) A->nrow, ncol = (Int) A->ncol ;
    cholmod_sparse *Fown = NULL ;
    if (F && values && F->xtype != A->xtype) { ERROR (CHOLMOD_INVALID, "A and F must have the same xtype") ; return NULL ; }
    if (!F)
    {
        Fown = cholmod_l_ptranspose (A, values ? (A->xtype == CHOLMOD_REAL ? 1 : 2) : 0, NULL, NULL, 0, Common) ;
        if (!Fown) return NULL ;
        F = Fown ;
    }
    if ((Int) F->nrow != ncol || (Int) F->ncol != m)
    {
        if (Fown) cholmod_l_free_sparse (&Fown, Common) ;
        ERROR (CHOLMOD_INVALID, "A and F dimensions do not match") ;
        return NULL ;
    }
    const Int *Ap = A->p, *Ai = A->i, *Anz = A->nz, *Fp = F->p, *Fi = F->i, *Fnz = F->nz ;
    const double *Ax = A->x, *Fx = F->x, *Az = A->z, *Fz = F->z ;
    const int av = values && Ax && Fx ;
    const int cx = av && A->xtype == CHOLMOD_COMPLEX, zx = av && A->xtype == CHOLMOD_ZOMPLEX ;
    const size_t we = (cx || zx) ? 2 : 1 ;
    Int *mark = cholmod_l_malloc (m > 0 ? m : 1, sizeof (Int), Common) ;
    Int *Cp = cholmod_l_malloc (m + 1, sizeof (Int), Common) ;
    double *w = av ? cholmod_l_calloc (m > 0 ? m : 1, we * sizeof (double), Common) : NULL ;
    cholmod_sparse *C = NULL ;
    int ok = mark && Cp && (!av || w) ;
    if (ok)
    {
        /* pass 1: entries per column of the stored triangle */
        for (Int i = 0 ; i < m ; i++) mark [i] = EMPTY ;
        Int nz = 0 ;
        for (Int j = 0 ; j < m ; j++)
        {
            Cp [j] = nz ;
            Int p = Fp [j], pend = F->packed ? Fp [j+1] : p + Fnz [j] ;
            for ( ; p < pend ; p++)
            {
                Int k = Fi [p] ;
                Int q = Ap [k], qend = A->packed ? Ap [k+1] : q + Anz [k] ;
                for ( ; q < qend ; q++)
                {
                    Int i = Ai [q] ;
                    if ((lower ? i < j : i > j) || mark [i] == j) continue ;
                    mark [i] = j ; nz++ ;
                }
            }
        }
        Cp [m] = nz ;
        C = cholmod_l_allocate_sparse (m, m, nz, TRUE, TRUE, lower ? -1 : 1,
            !av ? CHOLMOD_PATTERN : (cx || zx) ? CHOLMOD_COMPLEX : CHOLMOD_REAL, Common) ;
        ok = (C != NULL) ;
    }
    if (ok)
    {
        Int *Cpp = C->p, *Ci = C->i ;
        double *Cx = C->x ;
        for (Int j = 0 ; j <= m ; j++) Cpp [j] = Cp [j] ;
        for (Int i = 0 ; i < m ; i++) mark [i] = EMPTY ;
        for (Int j = 0 ; j < m ; j++)
        {
            Int dst = Cp [j] ;
            Int p = Fp [j], pend = F->packed ? Fp [j+1] : p + Fnz [j] ;
            for ( ; p < pend ; p++)
            {
                Int k = Fi [p] ;
                const double fkj = !av ? 0.0 : cx ? Fx [2*p] : Fx [p] ;
                const double fkjz = cx ? Fx [2*p+1] : zx ? Fz [p] : 0.0 ;
                Int q = Ap [k], qend = A->packed ? Ap [k+1] : q + Anz [k] ;
                for ( ; q < qend ; q++)
                {
                    Int i = Ai [q] ;
                    if (lower ? i < j : i > j) continue ;
                    if (mark [i] != j) { mark [i] = j ; Ci [dst++] = i ; }
                    if (cx || zx)
                    {
                        const double ar = cx ? Ax [2*q] : Ax [q], ai = cx ? Ax [2*q+1] : Az [q] ;
                        w [2*i] += ar * fkj - ai * fkjz ;
                        w [2*i+1] += ar * fkjz + ai * fkj ;
                    }
                    else if (av) w [i] += Ax [q] * fkj ;
                }
            }
            /* sort the column (short lists: insertion sort; long ones: by a counting pass over the marks would need O(m)) */
            Int len = dst - Cp [j] ;
            Int *col = Ci + Cp [j] ;
            if (len > 64)
            {
                /* heap sort, in place */
                for (Int start = len / 2 - 1 ; start >= 0 ; start--)
                    for (Int r = start ; ; )
                    {
                        Int c = 2 * r + 1 ; if (c >= len) break ;
                        if (c + 1 < len && col [c + 1] > col [c]) c++ ;
                        if (col [r] >= col [c]) break ;
                        Int t = col [r] ; col [r] = col [c] ; col [c] = t ; r = c ;
                    }
                for (Int end = len - 1 ; end > 0 ; end--)
                {
                    Int t = col [0] ; col [0] = col [end] ; col [end] = t ;
                    for (Int r = 0 ; ; )
                    {
                        Int c = 2 * r + 1 ; if (c >= end) break ;
                        if (c + 1 < end && col [c + 1] > col [c]) c++ ;
                        if (col [r] >= col [c]) break ;
                        Int t2 = col [r] ; col [r] = col [c] ; col [c] = t2 ; r = c ;
                    }
                }
            }
            else
                for (Int a = 1 ; a < len ; a++)
                {
                    Int v = col [a], b = a - 1 ;
                    while (b >= 0 && col [b] > v) { col [b + 1] = col [b] ; b-- ; }
                    col [b + 1] = v ;
                }
            if (cx || zx)
                for (Int a = 0 ; a < len ; a++)
                {
                    Cx [2 * (Cp [j] + a)] = w [2 * col [a]] ; Cx [2 * (Cp [j] + a) + 1] = w [2 * col [a] + 1] ;
                    w [2 * col [a]] = w [2 * col [a] + 1] = 0.0 ;
                }
            else if (av) for (Int a = 0 ; a < len ; a++) { Cx [Cp [j] + a] = w [col [a]] ; w [col [a]] = 0.0 ; }
        }
    }
    if (mark) cholmod_l_free (m > 0 ? m : 1, sizeof (Int), mark, Common) ;
    if (Cp) cholmod_l_free (m + 1, sizeof (Int), Cp, Common) ;
    if (w) cholmod_l_free (m > 0 ? m : 1, we * sizeof (double), w, Common) ;
    if (Fown) cholmod_l_free_sparse (&Fown, Common) ;
    if (!ok && C) cholmod_l_free_sparse (&C, Common) ;
    return ok ? C : NULL ;
}

cholmod_sparse *cholmod_l_transpose (cholmod_sparse *A, int values, cholmod_common *Common)
{
    return cholmod_l_ptranspose (A, values, NULL, NULL, 0, Common) ;
}

/* ---- triplet (reference: Core/cholmod_triplet.c) ------------------------------- */

cholmod_triplet *cholmod_l_allocate_triplet (size_t nrow, size_t ncol, size_t nzmax,
    int stype, int xtype, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    if (xtype != CHOLMOD_PATTERN && xtype != CHOLMOD_REAL && xtype != CHOLMOD_COMPLEX)
    {
        ERROR (CHOLMOD_INVALID, "xtype invalid") ;      /* (zomplex triplets: not built -- the reader returns complex) */
        return NULL ;
    }
    Common->status = CHOLMOD_OK ;
    cholmod_triplet *T = cholmod_l_calloc (1, sizeof (cholmod_triplet), Common) ;
    if (!T) return NULL ;
    nzmax = nzmax > 1 ? nzmax : 1 ;
    T->nrow = nrow ; T->ncol = ncol ; T->nzmax = nzmax ; T->nnz = 0 ;
    T->stype = stype ; T->itype = CHOLMOD_LONG ; T->xtype = xtype ; T->dtype = CHOLMOD_DOUBLE ;
    T->i = cholmod_l_malloc (nzmax, sizeof (Int), Common) ;
    T->j = cholmod_l_malloc (nzmax, sizeof (Int), Common) ;
    if (xtype != CHOLMOD_PATTERN) T->x = cholmod_l_malloc (nzmax, SSAMD_XENT (xtype) * sizeof (double), Common) ;
    if (Common->status < CHOLMOD_OK) { cholmod_l_free_triplet (&T, Common) ; return NULL ; }
    return T ;
}

int cholmod_l_free_triplet (cholmod_triplet **TH, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    if (!TH || !*TH) return TRUE ;
    cholmod_triplet *T = *TH ;
    cholmod_l_free (T->nzmax, sizeof (Int), T->i, Common) ;
    cholmod_l_free (T->nzmax, sizeof (Int), T->j, Common) ;
    if (T->x) cholmod_l_free (T->nzmax, SSAMD_XENT (T->xtype) * sizeof (double), T->x, Common) ;
    cholmod_l_free (1, sizeof (cholmod_triplet), T, Common) ;
    *TH = NULL ;
    return TRUE ;
}

/* Duplicates are summed; for stype != 0 entries in the ignored triangle are
 * transposed into the stored one (reference Core/cholmod_triplet.c:266-270:
 * "entries in the wrong triangle are transposed" -- the value moves as it is, complex ones
 * are not conjugated: t_cholmod_triplet.c:62-100).  Output columns sorted. */
cholmod_sparse *cholmod_l_triplet_to_sparse (cholmod_triplet *T, size_t nzmax,
    cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (T, NULL) ;
    Common->status = CHOLMOD_OK ;
    Int nrow = (Int) T->nrow, ncol = (Int) T->ncol, nz = (Int) T->nnz ;
    Int *Ti = T->i, *Tj = T->j ;
    double *Tx = T->x ;
    if (T->stype != 0 && nrow != ncol) { ERROR (CHOLMOD_INVALID, "matrix invalid") ; return NULL ; }
    for (Int k = 0 ; k < nz ; k++)
        if (Ti [k] < 0 || Ti [k] >= nrow || Tj [k] < 0 || Tj [k] >= ncol)
        { ERROR (CHOLMOD_INVALID, "index out of range") ; return NULL ; }
    /* sort by (col,row) with two stable bucket passes */
    Int *r1 = cholmod_l_malloc (nz > 0 ? nz : 1, sizeof (Int), Common) ;   /* order after row pass */
    Int *cnt = cholmod_l_calloc ((nrow > ncol ? nrow : ncol) + 2, sizeof (Int), Common) ;
    Int *r2 = cholmod_l_malloc (nz > 0 ? nz : 1, sizeof (Int), Common) ;
    if (!r1 || !cnt || !r2)
    {
        if (r1) cholmod_l_free (nz > 0 ? nz : 1, sizeof (Int), r1, Common) ;
        if (r2) cholmod_l_free (nz > 0 ? nz : 1, sizeof (Int), r2, Common) ;
        if (cnt) cholmod_l_free ((nrow > ncol ? nrow : ncol) + 2, sizeof (Int), cnt, Common) ;
        return NULL ;
    }
#define TROW(k) ((T->stype > 0 && Ti [k] > Tj [k]) || (T->stype < 0 && Ti [k] < Tj [k]) ? Tj [k] : Ti [k])
#define TCOL(k) ((T->stype > 0 && Ti [k] > Tj [k]) || (T->stype < 0 && Ti [k] < Tj [k]) ? Ti [k] : Tj [k])
    for (Int k = 0 ; k < nz ; k++) cnt [TROW (k) + 1]++ ;
    for (Int i = 0 ; i < nrow ; i++) cnt [i+1] += cnt [i] ;
    for (Int k = 0 ; k < nz ; k++) r1 [cnt [TROW (k)]++] = k ;
    memset (cnt, 0, ((nrow > ncol ? nrow : ncol) + 2) * sizeof (Int)) ;
    for (Int k = 0 ; k < nz ; k++) cnt [TCOL (k) + 1]++ ;
    for (Int j = 0 ; j < ncol ; j++) cnt [j+1] += cnt [j] ;
    for (Int q = 0 ; q < nz ; q++) { Int k = r1 [q] ; r2 [cnt [TCOL (k)]++] = k ; }
    /* count distinct */
    Int nd = 0 ;
    for (Int q = 0 ; q < nz ; q++)
    {
        Int k = r2 [q] ;
        if (q == 0 || TROW (k) != TROW (r2 [q-1]) || TCOL (k) != TCOL (r2 [q-1])) nd++ ;
    }
    size_t cap = (size_t) nd > nzmax ? (size_t) nd : nzmax ;
    cholmod_sparse *A = cholmod_l_allocate_sparse (nrow, ncol, cap, TRUE, TRUE, T->stype,
        T->xtype, Common) ;
    if (A)
    {
        Int *Ap = A->p, *Ai = A->i ;
        double *Ax = A->x ;
        const int cx = (T->xtype == CHOLMOD_COMPLEX) ;
        Int dst = -1 ;
        for (Int j = 0 ; j <= ncol ; j++) Ap [j] = 0 ;
        for (Int q = 0 ; q < nz ; q++)
        {
            Int k = r2 [q] ;
            Int r = TROW (k), c = TCOL (k) ;
            if (q == 0 || r != TROW (r2 [q-1]) || c != TCOL (r2 [q-1]))
            {
                dst++ ;
                Ai [dst] = r ;
                if (Ax && !cx) Ax [dst] = 0 ;
                if (Ax && cx) Ax [2*dst] = Ax [2*dst+1] = 0 ;
                Ap [c+1]++ ;
            }
            if (Ax && !cx) Ax [dst] += Tx [k] ;
            if (Ax && cx) { Ax [2*dst] += Tx [2*k] ; Ax [2*dst+1] += Tx [2*k+1] ; }
        }
        for (Int j = 0 ; j < ncol ; j++) Ap [j+1] += Ap [j] ;
    }
#undef TROW
#undef TCOL
    cholmod_l_free (nz > 0 ? nz : 1, sizeof (Int), r1, Common) ;
    cholmod_l_free (nz > 0 ? nz : 1, sizeof (Int), r2, Common) ;
    cholmod_l_free ((nrow > ncol ? nrow : ncol) + 2, sizeof (Int), cnt, Common) ;
    return A ;
}

/* ---- dense (reference: Core/cholmod_dense.c) ----------------------------------- */

cholmod_dense *cholmod_l_allocate_dense (size_t nrow, size_t ncol, size_t d, int xtype,
    cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    if (d < nrow) { ERROR (CHOLMOD_INVALID, "leading dimension invalid") ; return NULL ; }
    if (xtype < CHOLMOD_REAL || xtype > CHOLMOD_ZOMPLEX) { ERROR (CHOLMOD_INVALID, "xtype invalid") ; return NULL ; }
    Common->status = CHOLMOD_OK ;
    cholmod_dense *X = cholmod_l_calloc (1, sizeof (cholmod_dense), Common) ;
    if (!X) return NULL ;
    size_t nzmax = d * ncol ; nzmax = nzmax > 1 ? nzmax : 1 ;
    X->nrow = nrow ; X->ncol = ncol ; X->nzmax = nzmax ; X->d = d ;
    X->xtype = xtype ; X->dtype = CHOLMOD_DOUBLE ;
    X->x = cholmod_l_malloc (nzmax, SSAMD_XENT (xtype) * sizeof (double), Common) ;
    if (X->x && xtype == CHOLMOD_ZOMPLEX) X->z = cholmod_l_malloc (nzmax, sizeof (double), Common) ;
    if (!X->x || (xtype == CHOLMOD_ZOMPLEX && !X->z)) { cholmod_l_free_dense (&X, Common) ; return NULL ; }
    return X ;
}

cholmod_dense *cholmod_l_zeros (size_t nrow, size_t ncol, int xtype, cholmod_common *Common)
{
    cholmod_dense *X = cholmod_l_allocate_dense (nrow, ncol, nrow, xtype, Common) ;
    if (X) memset (X->x, 0, X->nzmax * SSAMD_XENT (xtype) * sizeof (double)) ;
    if (X && X->z) memset (X->z, 0, X->nzmax * sizeof (double)) ;
    return X ;
}

cholmod_dense *cholmod_l_ones (size_t nrow, size_t ncol, int xtype, cholmod_common *Common)
{
    cholmod_dense *X = cholmod_l_allocate_dense (nrow, ncol, nrow, xtype, Common) ;
    if (X)
    {
        double *x = X->x ;
        if (xtype == CHOLMOD_COMPLEX) for (size_t k = 0 ; k < X->nzmax ; k++) { x [2*k] = 1.0 ; x [2*k+1] = 0.0 ; }
        else for (size_t k = 0 ; k < X->nzmax ; k++) x [k] = 1.0 ;
        if (X->z) memset (X->z, 0, X->nzmax * sizeof (double)) ;
    }
    return X ;
}

cholmod_dense *cholmod_l_copy_dense (cholmod_dense *X, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (NULL) ;
    RETURN_IF_NULL (X, NULL) ;
    cholmod_dense *Y = cholmod_l_allocate_dense (X->nrow, X->ncol, X->d, X->xtype, Common) ;
    if (Y) memcpy (Y->x, X->x, X->d * X->ncol * SSAMD_XENT (X->xtype) * sizeof (double)) ;
    if (Y && X->z) memcpy (Y->z, X->z, X->d * X->ncol * sizeof (double)) ;
    return Y ;
}

int cholmod_l_free_dense (cholmod_dense **XH, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    if (!XH || !*XH) return TRUE ;
    cholmod_dense *X = *XH ;
    if (X->x) cholmod_l_free (X->nzmax, SSAMD_XENT (X->xtype) * sizeof (double), X->x, Common) ;
    if (X->z) cholmod_l_free (X->nzmax, sizeof (double), X->z, Common) ;
    cholmod_l_free (1, sizeof (cholmod_dense), X, Common) ;
    *XH = NULL ;
    return TRUE ;
}

/* ---- factor (reference: Core/cholmod_factor.c:160-228 free_factor) ------------- */

int cholmod_l_free_factor (cholmod_factor **LH, cholmod_common *Common)
{
    RETURN_IF_NULL_COMMON (FALSE) ;
    if (!LH || !*LH) return TRUE ;
    cholmod_factor *L = *LH ;
    size_t n = L->n ;
    if (L->hip_plan) { cholmod_hip_plan_destroy ((cholmod_hip_plan *) L->hip_plan) ; L->hip_plan = NULL ; }
    cholmod_l_free (n, sizeof (Int), L->Perm, Common) ;
    cholmod_l_free (n, sizeof (Int), L->ColCount, Common) ;
    if (L->IPerm) cholmod_l_free (n, sizeof (Int), L->IPerm, Common) ;
    if (L->bset_work) cholmod_l_free (2 * n + 1, sizeof (Int), L->bset_work, Common) ;
    if (L->super) cholmod_l_free (L->nsuper + 1, sizeof (Int), L->super, Common) ;
    if (L->pi) cholmod_l_free (L->nsuper + 1, sizeof (Int), L->pi, Common) ;
    if (L->px) cholmod_l_free (L->nsuper + 1, sizeof (Int), L->px, Common) ;
    if (L->s) cholmod_l_free (L->ssize, sizeof (Int), L->s, Common) ;
    if (L->cx_twin) cholmod_l_free_factor ((cholmod_factor **) &L->cx_twin, Common) ;
    if (L->x) cholmod_l_free (L->xsize, SSAMD_XENT (L->xtype) * sizeof (double), L->x, Common) ;
    cholmod_l_free (1, sizeof (cholmod_factor), L, Common) ;
    *LH = NULL ;
    return TRUE ;
}
