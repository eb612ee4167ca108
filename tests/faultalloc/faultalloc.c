/* faultalloc.c -- TEST INFRASTRUCTURE: an allocator that fails its k-th call, to be installed in SuiteSparse_config
 * (include/SuiteSparse_config.h) the way CHOLMOD/Tcov/memory.c:60-125 installs my_malloc2 / my_calloc2 / my_realloc2 /
 * my_free2.  fa_arm (k): from now on the k-th allocation call (malloc, calloc or realloc; 0-based, counted over all
 * threads) returns NULL, every other call goes to libc; k < 0: never fail.  fa_calls (): allocation calls since fa_arm. */
#include <stdlib.h>
#include <stdatomic.h>
#include <stdio.h>
#include <execinfo.h>

static atomic_long g_calls ;
static atomic_long g_fail_at = -1 ;
static atomic_long g_failed ;

void fa_arm (long k) { atomic_store (&g_calls, 0) ; atomic_store (&g_failed, 0) ; atomic_store (&g_fail_at, k) ; }
long fa_calls (void) { return atomic_load (&g_calls) ; }
long fa_failed (void) { return atomic_load (&g_failed) ; }

static int fails_now (void)
{
    long c = atomic_fetch_add (&g_calls, 1) ;
    if (c == atomic_load (&g_fail_at))
    {
        atomic_fetch_add (&g_failed, 1) ;
        if (getenv ("FA_TRACE"))        /* who asked: the call chain of the allocation that is being refused */
        {
            void *bt [16] ;
            int nb = backtrace (bt, 16) ;
            backtrace_symbols_fd (bt, nb, 2) ;
        }
        return 1 ;
    }
    return 0 ;
}

void *fa_malloc (size_t n) { return fails_now () ? NULL : malloc (n) ; }
void *fa_calloc (size_t n, size_t s) { return fails_now () ? NULL : calloc (n, s) ; }
void *fa_realloc (void *p, size_t n) { return fails_now () ? NULL : realloc (p, n) ; }
void fa_free (void *p) { free (p) ; }
