/* host_internal.h -- shared declarations of the host C layer. */
#ifndef SSAMD_HOST_INTERNAL_H
#define SSAMD_HOST_INTERNAL_H

#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/SuiteSparse_config.h"
#include "../../../include/cholmod.h"
#include "../../../include/cholmod_hip.h"

typedef SuiteSparse_long Int ;
#define EMPTY (-1)

/* doubles per entry of an x array (complex: interleaved pairs; zomplex keeps its
 * imaginary parts in a z array of its own) */
#define SSAMD_XENT(xtype) ((size_t) ((xtype) == CHOLMOD_COMPLEX ? 2 : 1))

#define ERROR(status, msg) cholmod_l_error (status, __FILE__, __LINE__, msg, Common)

#define RETURN_IF_NULL_COMMON(result) \
    do { if (Common == NULL) return (result) ; \
         if (Common->itype != CHOLMOD_LONG || Common->dtype != CHOLMOD_DOUBLE) \
         { Common->status = CHOLMOD_INVALID ; return (result) ; } } while (0)

#define RETURN_IF_NULL(A, result) \
    do { if ((A) == NULL) { \
             if (Common->status != CHOLMOD_OUT_OF_MEMORY) { ERROR (CHOLMOD_INVALID, "argument missing") ; } \
             return (result) ; } } while (0)

/* core.c */
int ssamd_host_threads (void) ;
int ssamd_host_threads_uncapped (void) ;
cholmod_sparse *ssamd_aat (cholmod_sparse *A, cholmod_sparse *F, int values, int lower, cholmod_common *Common) ;
cholmod_sparse *ssamd_column_subset (cholmod_sparse *A, const SuiteSparse_long *fset, size_t fsize, int values, cholmod_common *Common) ;
cholmod_sparse *ssamd_sym_permute (cholmod_sparse *A, int values, SuiteSparse_long *Perm, int upper_out,
    cholmod_common *Common) ;
cholmod_sparse *ssamd_sym_permute_src (cholmod_sparse *A, int values, SuiteSparse_long *Perm, int upper_out,
    SuiteSparse_long **src_out, cholmod_common *Common) ;

/* analyze.c */
int ssamd_etree_upper (Int n, const Int *Up, const Int *Ui, Int *Parent) ;
Int ssamd_postorder (Int n, const Int *Parent, const Int *Weight, Int *Post, Int *work3n) ;
void ssamd_colcounts (Int n, const Int *Lp, const Int *Li, const Int *Parent, const Int *Post,
    Int *ColCount, Int *work5n) ;

/* numeric.c */
int ssamd_nested_dissection (Int n, const Int *Ap, const Int *Ai, Int *Perm, cholmod_common *Common) ;
int ssamd_resolve_use_gpu (cholmod_common *Common) ;
int ssamd_ensure_plan (cholmod_factor *L, cholmod_common *Common) ;
void ssamd_plan_ahead (cholmod_factor *L, cholmod_common *Common) ;

/* complex.c: complex / zomplex input through the real embedding */
int ssamd_complex_super_numeric (cholmod_sparse *A, double beta, cholmod_factor *L, cholmod_common *Common) ;
int ssamd_complex_sync_host (cholmod_factor *L, cholmod_common *Common) ;
cholmod_factor *ssamd_complex_twin (cholmod_factor *L, cholmod_common *Common) ;

/* cpu_numeric.c */
int ssamd_cpu_super_numeric (cholmod_sparse *A, double beta, cholmod_factor *L, cholmod_common *Common) ;
void ssamd_cpu_super_solve (int which, const cholmod_factor *L, double *X, Int nrhs, Int ldx) ;
const char *ssamd_cpu_blas_name (void) ;
int ssamd_cpu_max_threads (void) ;
int ssamd_cpu_quota (void) ;
int ssamd_factor_has_cholesky_sizes (const cholmod_factor *L) ;
/* subset_solve.c: cholmod_l_solve2 with Bset */
int ssamd_solve_subset (int sys, cholmod_factor *L, cholmod_dense *B, cholmod_sparse *Bset, cholmod_dense *X,
    cholmod_sparse **Xset_Handle, cholmod_dense **Y_Handle, cholmod_common *Common) ;

#endif
