/* cholmod.h -- host-side C API of the MI355X supernodal Cholesky build.
 *
 * Mirrors the subset of the reference's cholmod_l_* API that lies on the
 * supernodal path analyze -> factorize -> solve, with
 * the same names, argument meaning and error behaviour, so a user of
 *     cholmod_l_start / cholmod_l_analyze[_p] / cholmod_l_factorize /
 *     cholmod_l_solve / cholmod_l_free_* / cholmod_l_finish
 * with Common->useGPU can switch to this library.  Reference declarations:
 *     CHOLMOD/Include/cholmod_core.h      (objects, :416 common, :1243 sparse,
 *                                          :1673 factor, :1976 dense)
 *     CHOLMOD/Include/cholmod_cholesky.h  (:89 analyze, :113 analyze_p,
 *                                          :160 factorize, :183 factorize_p,
 *                                          :216 solve, :281 etree, :311
 *                                          rowcolcounts, :634 postorder)
 *     CHOLMOD/Include/cholmod_supernodal.h(:73 super_symbolic, :100
 *                                          super_symbolic2, :126 super_numeric,
 *                                          :151 super_lsolve, :176 super_ltsolve)
 *     CHOLMOD/Include/cholmod_gpu.h       (:60-93 gpu_memorysize/probe/
 *                                          allocate/deallocate/end)
 *     CHOLMOD/Include/cholmod_check.h     (:110 gpu_stats), cholmod_matrixops.h
 *
 * Only the 64-bit integer flavour exists (the reference's GPU path is
 * cholmod_l_* only, CHOLMOD/Include/cholmod_internal.h:250-251); double
 * precision; real, complex and zomplex matrices (complex input is carried by the
 * real embedding, csrc/host/complex.c).  Common->useGPU selects the numeric path
 * as in the reference: 1 = the HIP engine (include/cholmod_hip.h), 0 = the CPU
 * supernodal path (csrc/host/cpu_numeric.c), -1 = decided by CHOLMOD_USE_GPU
 * (unset: CPU).  The struct members carry the reference's names; the structs are
 * this library's own (compile against this header).
 *
 * Where this library deliberately differs from the reference:
 *  - Common->supernodal = CHOLMOD_AUTO always picks the supernodal path (the
 *    simplicial branch is not built; CHOLMOD_SIMPLICIAL: CHOLMOD_NOT_INSTALLED);
 *  - orderings: UserPerm, natural, or the built-in nested dissection (any request
 *    for AMD / METIS / NESDIS / COLAMD maps to it, L->ordering = CHOLMOD_NESDIS);
 *  - a GPU request that cannot be served fails loudly (CHOLMOD_GPU_PROBLEM /
 *    CHOLMOD_OUT_OF_MEMORY) unless Common->hip_cpu_fallback asks for the
 *    reference's silent degradation to the CPU path;
 *  - the supernode partition is the CPU one on both paths (no devBuffSize splits);
 *  - an unsymmetric A (stype 0: factorize A*A' + beta*I) is served by forming tril (A*A') on
 *    the host and taking the symmetric path (real A; a column subset fset: A(:,f)*A(:,f)', the columns cut out first);
 *  - cholmod_l_solve2 with a sparse right-hand side (Bset) reads the columns the reference's supernodal -> simplicial
 *    conversion would produce in place (same pattern, same reach, same Xset order) and leaves L supernodal
 *    (host/subset_solve.c; the reference leaves it simplicial, Cholesky/cholmod_solve.c:1158-1180);
 *  - update/downdate and every simplicial form of L, complex triplets and
 *    complex Matrix Market files: CHOLMOD_NOT_INSTALLED / CHOLMOD_INVALID.
 */
#ifndef CHOLMOD_AMD_H
#define CHOLMOD_AMD_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t SuiteSparse_long ;

#define CHOLMOD_MAIN_VERSION 3
#define CHOLMOD_SUB_VERSION 0
#define CHOLMOD_SUBSUB_VERSION 14

#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif

/* itype / dtype / xtype (cholmod_core.h:310-333) */
#define CHOLMOD_INT 0
#define CHOLMOD_INTLONG 1
#define CHOLMOD_LONG 2
#define CHOLMOD_DOUBLE 0
#define CHOLMOD_SINGLE 1
#define CHOLMOD_PATTERN 0
#define CHOLMOD_REAL 1
#define CHOLMOD_COMPLEX 2
#define CHOLMOD_ZOMPLEX 3

/* Common->status (cholmod_core.h:386-393) */
#define CHOLMOD_OK 0
#define CHOLMOD_NOT_INSTALLED (-1)
#define CHOLMOD_OUT_OF_MEMORY (-2)
#define CHOLMOD_TOO_LARGE (-3)
#define CHOLMOD_INVALID (-4)
#define CHOLMOD_GPU_PROBLEM (-5)
#define CHOLMOD_NOT_POSDEF (1)
#define CHOLMOD_DSMALL (2)

/* orderings (cholmod_core.h:397-409) */
#define CHOLMOD_NATURAL 0
#define CHOLMOD_GIVEN 1
#define CHOLMOD_AMD 2
#define CHOLMOD_METIS 3
#define CHOLMOD_NESDIS 4
#define CHOLMOD_COLAMD 5
#define CHOLMOD_POSTORDERED 6

/* Common->supernodal (cholmod_core.h:412-414) */
#define CHOLMOD_SIMPLICIAL 0
#define CHOLMOD_AUTO 1
#define CHOLMOD_SUPERNODAL 2

/* cholmod_l_solve systems (cholmod_cholesky.h:194-202) */
#define CHOLMOD_A 0
#define CHOLMOD_LDLt 1
#define CHOLMOD_LD 2
#define CHOLMOD_DLt 3
#define CHOLMOD_L 4
#define CHOLMOD_Lt 5
#define CHOLMOD_D 6
#define CHOLMOD_P 7
#define CHOLMOD_Pt 8

#define CHOLMOD_ANALYZE_FOR_SPQR 0
#define CHOLMOD_ANALYZE_FOR_CHOLESKY 1
#define CHOLMOD_ANALYZE_FOR_SPQRGPU 2

#define CHOLMOD_MAXMETHODS 9

typedef struct cholmod_method_struct
{
    double lnz ;            /* nnz(L) found by this method */
    double fl ;             /* flop count found by this method */
    int ordering ;
} cholmod_method ;

typedef struct cholmod_common_struct
{
    /* parameters (defaults: reference Core/cholmod_common.c:244-330) */
    int supernodal ;                /* CHOLMOD_AUTO by default */
    double supernodal_switch ;      /* 40 */
    int final_asis, final_super, final_ll, final_pack, final_monotonic,
        final_resymbol ;
    double zrelax [3] ;             /* 0.8, 0.1, 0.05 */
    size_t nrelax [3] ;             /* 4, 16, 48 */
    int prefer_upper ;
    int quick_return_if_not_posdef ;
    int print ;
    int nmethods ;
    int current, selected ;
    cholmod_method method [CHOLMOD_MAXMETHODS + 1] ;
    int postorder ;                 /* TRUE */
    int try_catch ;
    void (*error_handler) (int status, const char *file, int line,
        const char *message) ;
    int itype, dtype ;
    int status ;
    /* statistics */
    double fl, lnz, anz, modfl ;
    size_t malloc_count, memory_usage, memory_inuse ;
    double nrealloc_col, nrealloc_factor, ndbounds_hit ;
    double rowfacfl, aatfl ;
    int called_nd ;
    int blas_ok ;
    /* GPU control (cholmod_core.h:958-1000) */
    int useGPU ;                    /* -1: decide from CHOLMOD_USE_GPU, 0, 1 */
    size_t maxGpuMemBytes ;
    double maxGpuMemFraction ;
    size_t gpuMemorySize ;
    double gpuKernelTime ;
    SuiteSparse_long gpuFlops ;
    int gpuNumKernelLaunches ;
    size_t devBuffSize ;
    int ibuffer ;
    double syrkStart ;
    /* timers / counters printed by cholmod_l_gpu_stats (cholmod_core.h:1004) */
    double cholmod_cpu_gemm_time, cholmod_cpu_syrk_time, cholmod_cpu_trsm_time,
        cholmod_cpu_potrf_time ;
    double cholmod_gpu_gemm_time, cholmod_gpu_syrk_time, cholmod_gpu_trsm_time,
        cholmod_gpu_potrf_time ;
    double cholmod_assemble_time, cholmod_assemble_time2 ;
    size_t cholmod_cpu_gemm_calls, cholmod_cpu_syrk_calls, cholmod_cpu_trsm_calls,
        cholmod_cpu_potrf_calls ;
    size_t cholmod_gpu_gemm_calls, cholmod_gpu_syrk_calls, cholmod_gpu_trsm_calls,
        cholmod_gpu_potrf_calls ;
    /* this library: keep the numeric factor in HBM only (L->x stays NULL until
     * cholmod_l_factor_to_host is called); solves then run on the device */
    int hip_factor_on_device ;
    int hip_flags ;                 /* CHOLMOD_HIP_* plan flags */
    int hip_profile ;               /* collect per-kernel-class device times */
    /* multi-GPU, one process per GPU (see cholmod_hip.h): rank / world size of
     * this process and the host-provided sum all-reduce on device memory */
    int hip_rank, hip_world ;
    int (*hip_allreduce) (void *dev_ptr, int64_t count_doubles, int group_first,
        int group_size, void *user) ;
    void *hip_allreduce_user ;
    /* Common->useGPU == 1 but no usable device (none present, or no HBM for this
     * factor): 0 (default) = fail loudly with CHOLMOD_GPU_PROBLEM / CHOLMOD_OUT_OF_MEMORY;
     * 1 = degrade to the CPU path with status CHOLMOD_OK, as the reference does
     * (CHOLMOD/Supernodal/t_cholmod_super_numeric.c:183-192).  The environment
     * variable CHOLMOD_HIP_CPU_FALLBACK=1 sets it in cholmod_l_start. */
    int hip_cpu_fallback ;
    int prefer_zomplex ;            /* X of cholmod_l_solve: zomplex instead of complex
                                     * (reference cholmod_core.h, Cholesky/cholmod_solve.c:1112) */
    int prefer_binary ;             /* cholmod_l_read_*: a symmetric pattern-only file keeps all-one values instead of
                                     * diagonal = 1 + degree, off-diagonal = -1 (cholmod_core.h:545-560) */
    int hip_lazy_plan ;             /* 0 (default): cholmod_l_analyze of a real matrix with Common->useGPU on builds the
                                     * engine's plan -- schedule, maps, the HBM reservation for L and the contribution
                                     * blocks -- as the reference cuts its device pools inside the analysis
                                     * (cholmod_super_symbolic.c:243-327); a failure there is not an error of the analysis,
                                     * the first factorization tries again and reports.  1: at the first factorization
                                     * (or cholmod_l_hip_prepare), as until round 5 -- for callers that analyse more
                                     * patterns than the device could hold factors of */
    double hip_plan_seconds ;       /* out: what the last creation of an engine plan took (inside cholmod_l_analyze, the
                                     * first factorization or cholmod_l_hip_prepare) */
} cholmod_common ;

typedef struct cholmod_sparse_struct
{
    size_t nrow, ncol, nzmax ;
    void *p, *i, *nz, *x, *z ;
    int stype ;     /* 0 unsymmetric, >0 upper stored, <0 lower stored */
    int itype, xtype, dtype ;
    int sorted, packed ;
} cholmod_sparse ;

typedef struct cholmod_dense_struct
{
    size_t nrow, ncol, nzmax, d ;
    void *x, *z ;
    int xtype, dtype ;
} cholmod_dense ;

typedef struct cholmod_triplet_struct
{
    size_t nrow, ncol, nzmax, nnz ;
    void *i, *j, *x, *z ;
    int stype, itype, xtype, dtype ;
} cholmod_triplet ;

typedef struct cholmod_factor_struct
{
    size_t n, minor ;
    void *Perm, *ColCount, *IPerm ;
    size_t nzmax ;
    void *p, *i, *x, *z, *nz, *next, *prev ;      /* simplicial part: unused */
    size_t nsuper, ssize, xsize, maxcsize, maxesize ;
    void *super, *pi, *px, *s ;
    int ordering, is_ll, is_super, is_monotonic ;
    int itype, xtype, dtype ;
    int useGPU ;
    /* this library: the device plan + resident numeric factor */
    void *hip_plan ;
    int hip_on_device ;     /* numeric values currently valid in HBM */
    int hip_host_valid ;    /* L->x holds the current numeric values */
    /* complex factors (L->xtype == CHOLMOD_COMPLEX): the real supernodal factor of the
     * 2n x 2n embedding [re -im ; im re] the engine actually computes; the complex
     * L->x is its even columns */
    void *cx_twin ;
    /* pattern of the last matrix cholmod_l_factorize handed to the engine (hash of its
     * p / i arrays, its nnz): a call with the same pattern only refreshes the values of
     * the resident matrix (cholmod_hip_refresh_values) */
    uint64_t hip_apat_hash ;
    size_t hip_apat_nnz ;
    int hip_apat_valid ;
    uint64_t hip_apat_hash2 ;   /* second, independent fingerprint of the same pattern */
    int hip_is_twin ;           /* this factor IS the real twin of a complex factor (its owner's cx_twin): 1 = the
                                 * full twin (x: 4 xsize doubles), 2 = engine-only, complex storage (px = 2 x the
                                 * complex px; CHOLMOD_HIP_CX_STORAGE) */
    void *bset_work ;           /* cholmod_l_solve2 with Bset: column -> supernode, flags (2n + 1 integers, built by the first call) */
    int hip_plan_ahead ;        /* != 0: hip_plan was built inside cholmod_l_analyze and no factorization has used it yet (plan flags + 1) */
} cholmod_factor ;

/* ---- Core ---------------------------------------------------------------- */
int cholmod_l_start (cholmod_common *Common) ;
int cholmod_l_finish (cholmod_common *Common) ;
int cholmod_l_defaults (cholmod_common *Common) ;
int cholmod_l_error (int status, const char *file, int line, const char *message,
    cholmod_common *Common) ;
void *cholmod_l_malloc (size_t n, size_t size, cholmod_common *Common) ;
void *cholmod_l_calloc (size_t n, size_t size, cholmod_common *Common) ;
void *cholmod_l_free (size_t n, size_t size, void *p, cholmod_common *Common) ;
/* CHOLMOD/Core/cholmod_memory.c:232-330; *n: current size in, new size out (unchanged on failure, old block returned) */
void *cholmod_l_realloc (size_t nnew, size_t size, void *p, size_t *n, cholmod_common *Common) ;

cholmod_sparse *cholmod_l_allocate_sparse (size_t nrow, size_t ncol, size_t nzmax,
    int sorted, int packed, int stype, int xtype, cholmod_common *Common) ;
int cholmod_l_free_sparse (cholmod_sparse **A, cholmod_common *Common) ;
cholmod_sparse *cholmod_l_copy_sparse (cholmod_sparse *A, cholmod_common *Common) ;
SuiteSparse_long cholmod_l_nnz (cholmod_sparse *A, cholmod_common *Common) ;
/* values: 0 pattern, 1 array transpose, 2 complex conjugate (== 1 for real) */
cholmod_sparse *cholmod_l_ptranspose (cholmod_sparse *A, int values,
    SuiteSparse_long *Perm, SuiteSparse_long *fset, size_t fsize,
    cholmod_common *Common) ;
cholmod_sparse *cholmod_l_transpose (cholmod_sparse *A, int values,
    cholmod_common *Common) ;

cholmod_triplet *cholmod_l_allocate_triplet (size_t nrow, size_t ncol, size_t nzmax,
    int stype, int xtype, cholmod_common *Common) ;
int cholmod_l_free_triplet (cholmod_triplet **T, cholmod_common *Common) ;
cholmod_sparse *cholmod_l_triplet_to_sparse (cholmod_triplet *T, size_t nzmax,
    cholmod_common *Common) ;

cholmod_dense *cholmod_l_allocate_dense (size_t nrow, size_t ncol, size_t d,
    int xtype, cholmod_common *Common) ;
cholmod_dense *cholmod_l_zeros (size_t nrow, size_t ncol, int xtype, cholmod_common *Common) ;
cholmod_dense *cholmod_l_ones (size_t nrow, size_t ncol, int xtype, cholmod_common *Common) ;
cholmod_dense *cholmod_l_copy_dense (cholmod_dense *X, cholmod_common *Common) ;
int cholmod_l_free_dense (cholmod_dense **X, cholmod_common *Common) ;

int cholmod_l_free_factor (cholmod_factor **L, cholmod_common *Common) ;

/* ---- Check / IO ------------------------------------------------------------ */
/* Check/cholmod_read.c (cholmod_check.h:251-330): triplet ("coordinate") and dense ("array") files, Matrix Market banner
 * optional, real / complex / pattern */
#define CHOLMOD_SPARSE 1            /* *mtype of cholmod_l_read_matrix (cholmod_core.h:300-303) */
#define CHOLMOD_DENSE 3
#define CHOLMOD_TRIPLET 4
cholmod_sparse *cholmod_l_read_sparse (FILE *f, cholmod_common *Common) ;
cholmod_triplet *cholmod_l_read_triplet (FILE *f, cholmod_common *Common) ;
cholmod_dense *cholmod_l_read_dense (FILE *f, cholmod_common *Common) ;
void *cholmod_l_read_matrix (FILE *f, int prefer, int *mtype, cholmod_common *Common) ;
int cholmod_l_check_factor (cholmod_factor *L, cholmod_common *Common) ;
int cholmod_l_check_sparse (cholmod_sparse *A, cholmod_common *Common) ;
int cholmod_l_gpu_stats (cholmod_common *Common) ;

/* ---- MatrixOps --------------------------------------------------------------- */
int cholmod_l_sdmult (cholmod_sparse *A, int transpose, double alpha [2],
    double beta [2], cholmod_dense *X, cholmod_dense *Y, cholmod_common *Common) ;
double cholmod_l_norm_dense (cholmod_dense *X, int norm, cholmod_common *Common) ;
double cholmod_l_norm_sparse (cholmod_sparse *A, int norm, cholmod_common *Common) ;

/* ---- Cholesky ---------------------------------------------------------------- */
cholmod_factor *cholmod_l_analyze (cholmod_sparse *A, cholmod_common *Common) ;
cholmod_factor *cholmod_l_analyze_p (cholmod_sparse *A, SuiteSparse_long *UserPerm,
    SuiteSparse_long *fset, size_t fsize, cholmod_common *Common) ;
cholmod_factor *cholmod_l_analyze_p2 (int for_whom, cholmod_sparse *A,
    SuiteSparse_long *UserPerm, SuiteSparse_long *fset, size_t fsize,
    cholmod_common *Common) ;
int cholmod_l_factorize (cholmod_sparse *A, cholmod_factor *L, cholmod_common *Common) ;
int cholmod_l_factorize_p (cholmod_sparse *A, double beta [2], SuiteSparse_long *fset,
    size_t fsize, cholmod_factor *L, cholmod_common *Common) ;
cholmod_dense *cholmod_l_solve (int sys, cholmod_factor *L, cholmod_dense *B,
    cholmod_common *Common) ;
int cholmod_l_solve2 (int sys, cholmod_factor *L, cholmod_dense *B, cholmod_sparse *Bset,
    cholmod_dense **X_Handle, cholmod_sparse **Xset_Handle, cholmod_dense **Y_Handle,
    cholmod_dense **E_Handle, cholmod_common *Common) ;
/* CHOLMOD/Include/cholmod_cholesky.h:601-614: (min diag L / max diag L)^2; -1 error, 0 singular / NaN / failed factor */
double cholmod_l_rcond (cholmod_factor *L, cholmod_common *Common) ;
/* CHOLMOD/Include/cholmod_core.h:1852-1872; here: supernodal symbolic <-> supernodal numeric (the simplicial forms:
 * CHOLMOD_NOT_INSTALLED) */
int cholmod_l_change_factor (int to_xtype, int to_ll, int to_super, int to_packed, int to_monotonic,
    cholmod_factor *L, cholmod_common *Common) ;
int cholmod_l_etree (cholmod_sparse *A, SuiteSparse_long *Parent, cholmod_common *Common) ;
SuiteSparse_long cholmod_l_postorder (SuiteSparse_long *Parent, size_t n,
    SuiteSparse_long *Weight, SuiteSparse_long *Post, cholmod_common *Common) ;
int cholmod_l_rowcolcounts (cholmod_sparse *A, SuiteSparse_long *fset, size_t fsize,
    SuiteSparse_long *Parent, SuiteSparse_long *Post, SuiteSparse_long *RowCount,
    SuiteSparse_long *ColCount, SuiteSparse_long *First, SuiteSparse_long *Level,
    cholmod_common *Common) ;

/* ---- Supernodal -------------------------------------------------------------- */
int cholmod_l_super_symbolic (cholmod_sparse *A, cholmod_sparse *F,
    SuiteSparse_long *Parent, cholmod_factor *L, cholmod_common *Common) ;
int cholmod_l_super_symbolic2 (int for_whom, cholmod_sparse *A, cholmod_sparse *F,
    SuiteSparse_long *Parent, cholmod_factor *L, cholmod_common *Common) ;
int cholmod_l_super_numeric (cholmod_sparse *A, cholmod_sparse *F, double beta [2],
    cholmod_factor *L, cholmod_common *Common) ;
int cholmod_l_super_lsolve (cholmod_factor *L, cholmod_dense *X, cholmod_dense *E,
    cholmod_common *Common) ;
int cholmod_l_super_ltsolve (cholmod_factor *L, cholmod_dense *X, cholmod_dense *E,
    cholmod_common *Common) ;

/* ---- GPU --------------------------------------------------------------------- */
int cholmod_l_gpu_memorysize (size_t *total_mem, size_t *available_mem,
    cholmod_common *Common) ;
int cholmod_l_gpu_probe (cholmod_common *Common) ;
int cholmod_l_gpu_deallocate (cholmod_common *Common) ;
void cholmod_l_gpu_end (cholmod_common *Common) ;
int cholmod_l_gpu_allocate (cholmod_common *Common) ;
/* int-flavour counterparts: inert, as in the reference (cholmod_gpu.c:84-86) */
int cholmod_gpu_memorysize (size_t *total_mem, size_t *available_mem, cholmod_common *Common) ;
int cholmod_gpu_probe (cholmod_common *Common) ;
int cholmod_gpu_deallocate (cholmod_common *Common) ;
void cholmod_gpu_end (cholmod_common *Common) ;
int cholmod_gpu_allocate (cholmod_common *Common) ;

/* ---- this library only --------------------------------------------------------- */
/* Materialise L->x on the host from the device-resident factor. */
int cholmod_l_factor_to_host (cholmod_factor *L, cholmod_common *Common) ;
/* Device time and per-class statistics of the last factorization
 * (CHOLMOD_HIP_NSTATS doubles, see cholmod_hip.h). */
int cholmod_l_hip_stats (cholmod_factor *L, double *stats, cholmod_common *Common) ;
/* Create the device plan of a symbolic factor now (HBM reservation for L and the
 * contribution blocks) instead of at the first numeric factorization; FALSE with
 * CHOLMOD_OUT_OF_MEMORY if the device cannot hold it. */
int cholmod_l_hip_prepare (cholmod_factor *L, cholmod_common *Common) ;
/* Multi-GPU: complete the factor on every rank after a distributed
 * factorization (needed before cholmod_l_solve / cholmod_l_factor_to_host). */
int cholmod_l_gather_factor (cholmod_factor *L, cholmod_common *Common) ;
/* Re-run the numeric factorization on the matrix already resident in HBM. */
int cholmod_l_refactorize_resident (double beta [2], cholmod_factor *L,
    cholmod_common *Common) ;

#ifdef __cplusplus
}
#endif
#endif
