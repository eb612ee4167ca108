/* SuiteSparse_config.h -- the process-wide memory / printf hooks of the reference, for the host layer of this build.
 *
 * Reference: SuiteSparse_config/SuiteSparse_config.h:75-98 (struct SuiteSparse_config_struct, the extern object),
 * :100-141 (SuiteSparse_start / _finish / _malloc / _calloc / _realloc / _free), SuiteSparse_config.c:57-330.
 * Every allocation of the host C layer goes through SuiteSparse_config.malloc_func / calloc_func / realloc_func /
 * free_func (cholmod_l_malloc, cholmod_l_calloc, cholmod_l_realloc, cholmod_l_free: CHOLMOD/Core/cholmod_memory.c:111-230),
 * so an application -- or a test in the shape of CHOLMOD/Tcov/memory.c:126-190 -- can replace them, e.g. with an allocator
 * that fails its k-th call.  As in the reference the object is process-global and is meant to be set before threads start.
 * (Allocations inside the HIP engine's plan builder are C++ containers; their failures are caught at the C boundary and
 * reported as CHOLMOD_OUT_OF_MEMORY, but they do not pass through these hooks.) */
#ifndef SUITESPARSE_CONFIG_AMD_H
#define SUITESPARSE_CONFIG_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

struct SuiteSparse_config_struct
{
    void *(*malloc_func) (size_t) ;             /* pointer to malloc */
    void *(*calloc_func) (size_t, size_t) ;     /* pointer to calloc */
    void *(*realloc_func) (void *, size_t) ;    /* pointer to realloc */
    void (*free_func) (void *) ;                /* pointer to free */
    int (*printf_func) (const char *, ...) ;    /* pointer to printf */
    double (*hypot_func) (double, double) ;     /* pointer to hypot */
    int (*divcomplex_func) (double, double, double, double, double *, double *) ;
} ;

extern struct SuiteSparse_config_struct SuiteSparse_config ;

void SuiteSparse_start (void) ;     /* resets the hooks to malloc / calloc / realloc / free / printf */
void SuiteSparse_finish (void) ;
void *SuiteSparse_malloc (size_t nitems, size_t size_of_item) ;
void *SuiteSparse_calloc (size_t nitems, size_t size_of_item) ;
/* returns the new block, or the ORIGINAL block and *ok = 0 if the reallocation failed */
void *SuiteSparse_realloc (size_t nitems_new, size_t nitems_old, size_t size_of_item, void *p, int *ok) ;
void *SuiteSparse_free (void *p) ;  /* always returns NULL */
double SuiteSparse_hypot (double x, double y) ;
int SuiteSparse_divcomplex (double ar, double ai, double br, double bi, double *cr, double *ci) ;

#ifdef __cplusplus
}
#endif
#endif
