/* cholmod_hip.h -- thin C-ABI shim of the MI355X (gfx950) supernodal Cholesky
 * engine.  Plain pointers and sizes only; no CHOLMOD structs, no torch types.
 *
 * This is the boundary a CHOLMOD maintainer binds: each entry point names the
 * reference interface it replaces (paths relative to the reference root).
 * The host layer in include/cholmod.h (cholmod_l_* mirror) is built on exactly
 * these calls; INTEGRATION.md shows the same calls made from the reference's
 * own cholmod_super_numeric.c.
 *
 * Index type is int64 (the reference's GPU path exists only in the
 * cholmod_l_* / DLONG build: CHOLMOD/Include/cholmod_internal.h:250-251).
 * All functions return 0 on success unless stated otherwise.
 */
#ifndef CHOLMOD_HIP_H
#define CHOLMOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes (engine level; the host layer maps them to Common->status) */
#define CHOLMOD_HIP_OK            0
#define CHOLMOD_HIP_NOT_POSDEF    1     /* success, but minor < n            */
#define CHOLMOD_HIP_NO_DEVICE   (-1)    /* no usable gfx950 device / runtime */
#define CHOLMOD_HIP_OUT_OF_MEMORY (-2)
#define CHOLMOD_HIP_TOO_LARGE   (-3)    /* n or nsuper beyond the engine's 32-bit maps */
#define CHOLMOD_HIP_INVALID     (-4)
#define CHOLMOD_HIP_GPU_PROBLEM (-5)    /* a HIP call or kernel failed       */

/* plan flags */
#define CHOLMOD_HIP_PLAN_DEFAULT   0
#define CHOLMOD_HIP_TILE128         4    /* tuning: 128x128 update tiles on big regions
                                           (default: 64x64 everywhere, which fills the
                                           256 CUs on mid-size fronts)               */
#define CHOLMOD_HIP_NO_SMALL_FRONTS 16   /* tuning: no fused LDS-resident kernel for
                                           thin fronts (generic kernels everywhere)  */
#define CHOLMOD_HIP_NO_XCD_SWIZZLE 32    /* tuning: plain block -> tile order          */
#define CHOLMOD_HIP_FIXED_OB       64    /* tuning: 512-column outer blocks everywhere  */
#define CHOLMOD_HIP_WIDE_OB       128    /* tests: 2048-column outer blocks everywhere  */
#define CHOLMOD_HIP_NO_CB_ASSIGN   2048   /* tuning: zero-fill every contribution block and
                                           extend-add before the dense phase           */
#define CHOLMOD_HIP_NO_FUSED_POTRF 512   /* tuning: every diagonal block through a k_potrf_mfma
                                         * launch of its own (no k_update2f)                 */
#define CHOLMOD_HIP_NO_FUSED_TRSM 1024   /* tuning: the K = 64 steps of the panel chain as separate
                                         * solve and update launches (no k_trsm_upd)               */
#define CHOLMOD_HIP_NO_LEAF_PAIRS 4096    /* tuning: leaf fronts one per wave like every other thin
                                         * front (no k_leaf_pair)                                */
#define CHOLMOD_HIP_NO_EXCHANGE_LOOKAHEAD 256 /* multi-GPU: all-reduce a block column only
                                         * when it is due (no overlap with updates)  */
#define CHOLMOD_HIP_CHAIN256     8192    /* tuning / tests: the panel chain in 256-column sub-blocks -- one launch per sub-block
                                         * (k_chainf: the diagonal sub-block spread over four workgroups that hand their row
                                         * block of L to the row workgroups through flags; CHOLMOD_HIP_NO_CHAINF=1: the two
                                         * kernels of round 3, k_diag + k_rowsolve) -- instead of the 64-column chain
                                         * (k_potrf_mfma, k_trsm_mfma, k_trsm_upd, k_update2f).  On one GPU within +-2 % of the
                                         * 64-column chain; the default for the batches of a multi-GPU
                                         * plan that hold a front shared between ranks */
#define CHOLMOD_HIP_PHI_TWIN    16384    /* the structure is the real twin of a complex factor (every supernode, row and
                                         * column doubled, host/complex.c; checked at plan creation): the update kernels
                                         * contract over the even panel columns only and rebuild the 2 x 2 blocks of the
                                         * embedding in the lanes -- a complex multiply-add as 4 real ones, not 8
                                         * (reference zherk / zgemm, t_cholmod_super_numeric.c:41-83, :682-717).  Set by
                                         * cholmod_l_super_numeric for complex / zomplex A; CHOLMOD_HIP_TWIN_FULL_K=1
                                         * in the environment keeps the plain embedding (A/B timing, tests) */
#define CHOLMOD_HIP_CX_STORAGE   32768    /* a complex factor in its own storage: super / pi / s are those of the twin (as
                                         * CHOLMOD_HIP_PHI_TWIN, implied), px [s] = 2 x the complex factor's px [s]: of
                                         * every front only the even twin columns exist -- nsrow_twin x nscol_twin / 2
                                         * doubles, i.e. the interleaved complex panel of the reference's L->x
                                         * (t_cholmod_super_numeric.c:41-83, L_ENTRY 2): 2 xsize doubles of HBM instead of
                                         * the twin's 4 xsize, contribution blocks likewise.  Kernels rebuild the odd
                                         * columns (rotations of the even ones) on the way into LDS / registers.  One
                                         * rank, generic kernels only (no thin-front kernels).  cholmod_hip_download_factor
                                         * then returns the complex factor itself */
#define CHOLMOD_HIP_PLAN_HOST_ONLY 2    /* build the schedule only, touch no device
                                           (CPU-side tests of the host logic)       */

typedef struct cholmod_hip_plan cholmod_hip_plan ;  /* opaque, one per symbolic L */

/* replaces cholmod_l_gpu_probe (CHOLMOD/GPU/cholmod_gpu.c:170-205):
 * returns 1 if a device is usable, 0 otherwise. */
int cholmod_hip_probe (void) ;

/* replaces cholmod_l_gpu_memorysize (CHOLMOD/GPU/cholmod_gpu.c:71-160):
 * returns 0 and fills total/available bytes, or 1 if there is no device. */
int cholmod_hip_memorysize (size_t *total_mem, size_t *available_mem) ;

/* Select the device for this process (one process per GPU). */
int cholmod_hip_set_device (int device) ;
/* Devices this process can see (a launcher checks it before it starts one rank per GPU);
 * no counterpart in the reference, which drives device 0 only (CHOLMOD/GPU/cholmod_gpu.c:160-164). */
int cholmod_hip_device_count (int *count) ;

/* Build the device plan of a supernodal symbolic factor: uploads the index
 * maps (super/pi/px/s exactly as in cholmod_factor, CHOLMOD/Include/
 * cholmod_core.h:1673-1798), derives the supernodal etree, its level sets, the
 * child->parent relative row maps (the reference's RelativeMap,
 * CHOLMOD/Supernodal/t_cholmod_super_numeric.c:743-750, computed on the device)
 * and the batched launch schedule.  Replaces r_cholmod_l_gpu_init
 * (CHOLMOD/GPU/t_cholmod_gpu.c:79-205) and cholmod_l_gpu_allocate
 * (CHOLMOD/GPU/cholmod_gpu.c:364-486): all device memory for L (xsize doubles)
 * and for the contribution-block arena is reserved here.
 * On failure returns NULL and stores a CHOLMOD_HIP_* code in *status. */
cholmod_hip_plan *cholmod_hip_plan_create (int64_t n, int64_t nsuper,
    const int64_t *super, const int64_t *pi, const int64_t *px,
    const int64_t *s, int flags, int *status) ;

void cholmod_hip_plan_destroy (cholmod_hip_plan *plan) ;

/* ---- multi-GPU (one process per GPU; no counterpart in the reference, which is
 * single-GPU: CHOLMOD/GPU/cholmod_gpu.c:160-164) -------------------------------
 * Every rank builds the same plan from the same symbolic factor and passes its
 * rank / world size.  The supernodal etree is mapped proportionally: the root's
 * front is *shared* by all ranks, the heavy children of a shared front split
 * its rank group [first, first+size) between them where their weights allow,
 * and the light subtrees below are dealt to the ranks of the group they hang
 * off (largest first).  A shared front lives on every rank of its group as a
 * partial sum; the group factors its panels redundantly and deals the tiles of
 * its trailing updates round-robin.  The only exchange is an in-place sum
 * all-reduce, over the front's group, of a 512-column block column right
 * before it is factored; the engine asks the host for it through this callback
 * (bench.py / the tests implement it with torch.distributed: RCCL over xGMI,
 * or gloo in CPU tests).  group_first/group_size name the contiguous rank range
 * that takes part (every rank of the range makes the same call, in the same
 * order; ranks outside do not call).  The callback is entered with the data
 * ready and must return only when the result is visible to later device work.
 * Returns 0 on success. */
typedef int (*cholmod_hip_allreduce_fn) (void *dev_ptr, int64_t count_doubles,
    int group_first, int group_size, void *user) ;
cholmod_hip_plan *cholmod_hip_plan_create_dist (int64_t n, int64_t nsuper,
    const int64_t *super, const int64_t *pi, const int64_t *px,
    const int64_t *s, int flags, int rank, int world, int *status) ;
int cholmod_hip_set_allreduce (cholmod_hip_plan *plan,
    cholmod_hip_allreduce_fn fn, void *user) ;
/* Native exchange: instead of the host callback the engine calls RCCL itself
 * (librccl bound with dlopen), stream-ordered on its own streams -- no host
 * synchronisation per block column, usable from a plain C caller.  One rank
 * obtains a 128-byte id with cholmod_hip_rccl_unique_id and hands it to the
 * others by any means (MPI_Bcast, a file, torch.distributed); every rank then
 * calls cholmod_hip_rccl_attach on its plan: ncclCommInitRank over the world and
 * one ncclCommSplit per rank group the plan shares fronts over (collective:
 * all ranks must call it).  The callback, if set, is then no longer used. */
int cholmod_hip_rccl_unique_id (void *id128) ;
int cholmod_hip_rccl_attach (cholmod_hip_plan *plan, const void *id128) ;
/* back to the callback (destroys the plan's communicators) */
int cholmod_hip_rccl_detach (cholmod_hip_plan *plan) ;
/* test hook (several ranks): the (contributor d, shared ancestor a) pairs whose contributions this rank routes straight into a's
 * windows, and per front the block [cb_lo, cb_hi) of contribution-block columns the rank stores (-1: not distributed).
 * Returns the number of pairs, fills at most cap. */
int64_t cholmod_hip_debug_routing (cholmod_hip_plan *plan, int64_t cap, int64_t *pair_d, int64_t *pair_a,
    int64_t *cb_lo, int64_t *cb_hi) ;
/* test hook: fingerprint (16 words) of the rank's plan -- fronts, routing, layout, every group array, the launch list */
int cholmod_hip_debug_schedule_hash (cholmod_hip_plan *plan, uint64_t *out16) ;
/* Progress of the factorization that is running (or ran last) on this plan, for a watchdog thread of the caller (bench.py
 * --gpus N: a hung collective must end in an error line, not in the driver's kill).  cholmod_hip_progress_enable (plan, 1)
 * allocates two words of pinned host memory the device marks, in stream order, around every block-column exchange.
 * cholmod_hip_progress: out [12] = [0] factorizations started, [1] launches of the schedule enqueued by the host in the current
 * one, [2] launches in the schedule, [3] exchanges enqueued, [4] exchanges in the schedule, [5] / [6] exchange the DEVICE has
 * entered / left (-1 without markers), and of the exchange entered and not left: [7] kind (7 = reduce-scatter + broadcast of
 * the diagonal block, 11 = all-gather), [8] first rank and [9] size of its rank group, [10] columns of the block column,
 * [11] rows below its diagonal block.  May be called from another thread while a factorization runs. */
int cholmod_hip_progress_enable (cholmod_hip_plan *plan, int enable) ;
int cholmod_hip_progress (cholmod_hip_plan *plan, int64_t *out12) ;
/* The batch order every rank of a partition must derive alike (the collectives of a group are issued in batch order):
 * batch_of[s] = index of the batch front s is factored in, counted over ALL fronts, *global_arena = length (doubles) of the
 * contribution-block arena laid out over all fronts, from which the memory-aware split was chosen; returns the number of
 * subtrees swept one after the other (1 = plain level order).  Either pointer may be NULL. */
int64_t cholmod_hip_get_batches (cholmod_hip_plan *plan, int64_t *batch_of, int64_t *global_arena) ;
/* owner[s] = rank that factors supernode s, -1 for the shared fronts */
int cholmod_hip_get_partition (cholmod_hip_plan *plan, int64_t *owner) ;
/* rank group of every supernode: ranks [first[s], first[s]+size[s]) hold it
 * (size 1 = private to owner[s]) */
int cholmod_hip_get_groups (cholmod_hip_plan *plan, int64_t *first, int64_t *size) ;
/* Several ranks: a rank allocates L only for the fronts it holds (its subtrees and the shared
 * fronts of its groups, packed; cholmod_hip_get_stats [36] bytes against [5] for the whole
 * factor).  This call builds the COMPLETE factor in the reference layout on every rank (a second,
 * full-size array: each front written by the first rank of its group, one sum over all ranks);
 * solves, downloads and checks of a multi-rank plan need it.  If the full array does not fit
 * next to the rank's part, the contribution-block arena makes room and is allocated again by the
 * next factorization.  The next factorization invalidates the gathered copy. */
int cholmod_hip_gather_factor (cholmod_hip_plan *plan) ;

/* Numeric factorization  L L' = S + beta*I  of the already permuted,
 * lower-stored matrix S = tril(P A P') (packed or unpacked CSC on the host:
 * Snz may be NULL), the whole of cholmod_l_super_numeric's loop
 * (CHOLMOD/Supernodal/t_cholmod_super_numeric.c:93-1079: assemble, descendant
 * updates, dpotrf, dtrsm, not-positive-definite protocol :883-968).
 * The factor stays resident in HBM; if Lx_host != NULL the packed Lx array
 * (xsize doubles, reference layout) is also copied back.
 * *minor receives L->minor (== n when positive definite).
 * Returns CHOLMOD_HIP_OK, CHOLMOD_HIP_NOT_POSDEF, or a negative error. */
int cholmod_hip_factorize (cholmod_hip_plan *plan, const int64_t *Sp,
    const int64_t *Si, const int64_t *Snz, const double *Sx, double beta,
    int quick_return_if_not_posdef, double *Lx_host, int64_t *minor) ;

/* The same in two steps, so that a caller (bench.py) can time the factorization
 * with the input already resident in HBM: upload S once, refactorize often. */
int cholmod_hip_upload_matrix (cholmod_hip_plan *plan, const int64_t *Sp,
    const int64_t *Si, const int64_t *Snz, const double *Sx) ;
int cholmod_hip_factorize_resident (cholmod_hip_plan *plan, double beta,
    int quick_return_if_not_posdef, int64_t *minor) ;

/* New values for the resident S when only the VALUES of the caller's matrix changed
 * (cholmod_l_factorize called again with the same pattern: the common case of a
 * nonlinear or time-stepping loop).  cholmod_hip_set_value_map hands over, once per
 * upload, where every entry of the resident packed S came from in the caller's value
 * array: src [q] in [0, nvalues).  cholmod_hip_refresh_values then copies the caller's
 * nvalues doubles to the device and gathers Sx [q] = values [src [q]] there -- no
 * host-side permutation, no pattern upload, the assembly map stays valid.
 * Both return CHOLMOD_HIP_INVALID without a resident packed S of snz entries. */
int cholmod_hip_set_value_map (cholmod_hip_plan *plan, const int64_t *src, int64_t snz,
    int64_t nvalues) ;
int cholmod_hip_refresh_values (cholmod_hip_plan *plan, const double *values, int64_t nvalues) ;
/* The same upload as a pipeline the caller drives (round 5): cholmod_hip_values_staging hands out a PINNED host buffer of
 * nvalues doubles owned by the plan; the caller fills it chunk by chunk (with its own threads) and calls
 * cholmod_hip_values_push (offset, count) after every chunk -- an asynchronous DMA, no host wait; cholmod_hip_values_commit
 * (plan, 1) then enqueues the gather into the resident S and an event the ASSEMBLY of the next cholmod_hip_factorize_resident
 * waits for (the clearing of L runs beside the upload); commit = 0 abandons what was pushed (the pattern changed after all)
 * and waits until nothing is in flight.  The caller's own array may be reused as soon as it has been copied out. */
int cholmod_hip_values_staging (cholmod_hip_plan *plan, double **host_buffer, int64_t *nvalues) ;
int cholmod_hip_values_push (cholmod_hip_plan *plan, int64_t offset, int64_t count) ;
int cholmod_hip_values_commit (cholmod_hip_plan *plan, int commit) ;
/* Round 6: the same upload in the order the factorization needs the values, chunk by chunk, so that a batch of fronts
 * waits for its own entries only (cholmod_l_factorize drives it from three kinds of host threads; engine.hip has the
 * protocol).  cholmod_hip_values_gather_index: *index = where staged position k comes from in the caller's value array, or
 * NULL when this plan has no batch order (several ranks, no assembly map yet) -- then staging / push / commit above apply;
 * *count = staged positions (the entries of the resident S), *chunk_len = positions per chunk. */
int cholmod_hip_values_gather_index (cholmod_hip_plan *plan, const int64_t **index, int64_t *chunk_len, int64_t *count) ;
int cholmod_hip_values_begin (cholmod_hip_plan *plan) ;
int cholmod_hip_values_push_chunk (cholmod_hip_plan *plan, int64_t chunk) ;

/* Copy the device-resident packed Lx (xsize doubles) to the host. */
int cholmod_hip_download_factor (cholmod_hip_plan *plan, double *Lx_host) ;
/* Even columns only, packed (xsize / 2 doubles): for a plan built on the doubled
 * structure of a complex factor (the real embedding, csrc/host/complex.c) these are the
 * interleaved complex columns of L, i.e. the reference's complex L->x, gathered on the
 * device. */
int cholmod_hip_download_even_columns (cholmod_hip_plan *plan, double *out_host) ;
/* Replace the device-resident Lx by host values (e.g. a factor computed
 * elsewhere), so the device solves can be used with it. */
int cholmod_hip_upload_factor (cholmod_hip_plan *plan, const double *Lx_host) ;

/* Supernodal triangular solves on the device-resident factor, in place on the
 * host array X (n-by-nrhs, leading dimension ldx), in the permuted ordering:
 * replace cholmod_l_super_lsolve / cholmod_l_super_ltsolve
 * (CHOLMOD/Supernodal/t_cholmod_super_solve.c:14-220, :222-411).
 * which: 0 = L then L' (both), 1 = L only, 2 = L' only. */
int cholmod_hip_solve (cholmod_hip_plan *plan, int which, double *X,
    int64_t nrhs, int64_t ldx) ;

/* Parity hooks: copy derived integer maps back to the host.
 *  sparent  [nsuper]     supernodal etree (reference :1025)
 *  level    [nsuper]     height of s in that tree (leaves 0)
 *  relmap   [ssize - n]  relative row maps, relmap[pi[d]-super[d] + i] = local
 *                        row in the parent of row i below d's diagonal block */
int cholmod_hip_get_maps (cholmod_hip_plan *plan, int64_t *sparent,
    int64_t *level, int64_t *relmap) ;

/* out3 = {min L_jj, max L_jj, number of NaN or negative diagonal entries} of the resident factor (one pass over the n
 * diagonal entries on the device): what cholmod_l_rcond needs (CHOLMOD/Cholesky/cholmod_rcond.c:64-161).  Several ranks:
 * after cholmod_hip_gather_factor. */
int cholmod_hip_diag_minmax (cholmod_hip_plan *plan, double *out3) ;
/* Size-independent invariants of the device-resident factor, one pass over Lx
 * (the checks CHOLMOD/Check/cholmod_check.c:1823-2000 cannot do on values, at
 * sizes no CPU oracle reaches):  out5[0] = sum_j log L(j,j)  (= logdet(A)/2,
 * known in closed form for the Poisson grids);  out5[1] = entries != 0 in the
 * dead strictly-upper triangles of the diagonal blocks;  out5[2] = non-finite
 * entries of the lower trapezoids;  out5[3] = ||L||_F^2 over the lower
 * trapezoids;  out5[4] = diagonal entries <= 0. */
int cholmod_hip_factor_checks (cholmod_hip_plan *plan, double *out5) ;
/* The same five numbers over the fronts this rank answers for (the first rank of a front's group),
 * from the rank's own part of a distributed factor: their sums over the ranks are the invariants
 * of the complete factor, no gathered copy needed. */
int cholmod_hip_factor_checks_local (cholmod_hip_plan *plan, double *out5) ;

/* Statistics of the last factorization / of the plan (doubles):
 *  [0] device seconds, whole factorization (HIP events on the engine stream)
 *  [1] executed flops (updates + panel factorizations, as SURVEY.md 8d)
 *  [2] kernel launches   [3] levels   [4] arena bytes   [5] Lx bytes
 *  [6] seconds in the 64x64 dense-update kernel   [7] its launches
 *  [8] its algorithmic flops (2*k per updated lower-trapezoid entry)
 *  [14] seconds in the 128x128 dense-update kernel (CHOLMOD_HIP_TILE128 only)
 *  [15] its algorithmic flops
 *  [16] algorithmic bytes of the 64x64 update launches: 16 B read-modify-write
 *       per updated entry + 8 B per operand entry (each panel entry once per
 *       update region)     [17] all-reduce calls   [18] all-reduce bytes
 *  [19] seconds in the fused small-front kernel  [20] its algorithmic HBM bytes
 *       (A entries aside: children CBs in, panel + CB out)   [21] fronts it handled
 *  [23] the part of [6] spent in K < 512 (panel-level) update launches
 *  [22] subtrees the schedule sweeps one after the other to fit the CB arena
 *       next to L (1 = plain level order)
 *  [9] seconds in extend-add kernels    [10] algorithmic bytes of extend-add
 *  [11] seconds in potrf kernels        [12] seconds in trsm kernels
 *  [13] seconds in assemble (memset + A scatter)
 *  [24] device seconds of the last cholmod_hip_solve (its kernels, without the
 *       copies of the right-hand side)
 *  [25] bytes of the all-gathers of [18] (as segments sent)   [39] those of them the main stream waits for at once: the
 *       near-row chunks of every block column and the far-row chunks of the last block column of an outer block (the other
 *       far-row gathers run on the exchange stream beside the chain of the following block columns)
 *  [26] trailing-update launches that also factor the next diagonal block (k_update2f: the
 *       K < 512 updates of the panel chain; NOT counted in [6]-[8], [16], [23])
 *  [27] their seconds   [28] their flops   [29] their algorithmic bytes
 *  [30] seconds of the fused solve + K = 64 update + factorization launches (k_trsm_upd)
 *  [31] their number
 *  [32] seconds in the one-wave-per-tile dense-update kernel (k_update3: the regions with
 *       >= 2048 tiles)   [33] its launches   [34] its algorithmic flops   [35] its algorithmic bytes
 *       (as [16]); the regions below that size stay with [6]-[8]
 *  [36] bytes of L this rank allocates: the fronts it holds -- of a shared front the column slabs it owns -- packed,
 *       plus the windows of the shared fronts (= [5] with one rank)
 *  [37] block columns of distributed fronts opened into their windows   [38] those whose window has a negative
 *       virtual base (the window is addressed as if the whole front were there: tests make sure both signs occur)
 * Per-class seconds are only collected when profiling is enabled with
 * cholmod_hip_set_profiling(plan, 1) (it serialises the stream with events). */
#define CHOLMOD_HIP_NSTATS 40
int cholmod_hip_get_stats (cholmod_hip_plan *plan, double *stats) ;
int cholmod_hip_set_profiling (cholmod_hip_plan *plan, int on) ;
/* The launch list of the plan and, after a factorization with profiling on, the
 * device milliseconds of every launch (tuning; tools/launch_profile.py).
 * kind: 0 zero, 1 extend-add, 2 potrf, 3 trsm, 4 update(128), 5 update(64),
 * 7 all-reduce, 8 thin fronts, 9 update + factorization of the next diagonal block,
 * 10 solve + K = 64 update + factorization of the next diagonal block, 11 all-gather of a shared block column,
 * 12 update (one wave per tile, k_update3), 13 / 14 the 256-column chain as two kernels (k_diag, k_rowsolve), 15 window moves of
 * a distributed front (k_win_move), 16 the 256-column chain in one launch (k_chainf).  Fills at most cap entries of the arrays that are
 * not NULL, returns the number of launches. */
int64_t cholmod_hip_get_launch_profile (cholmod_hip_plan *plan, int64_t cap, int32_t *kind,
    int32_t *grid, int32_t *aux, double *ms, double *flops, double *bytes) ;

/* Tuning probe (plans created with CHOLMOD_HIP_THIN_TIMING set): shader cycles one
 * front of thin-front launch `launch` spent per phase: [0] requests + zero, [1] A,
 * [2] children, [3] panel chain, [4] publish + row solves, [5] store + barrier,
 * [6] trailing update / contribution block. */
int cholmod_hip_debug_thin_cycles (cholmod_hip_plan *plan, int64_t launch, long long *out10) ;
/* tuning: the update regions of one update launch, 12 numbers each (m, n, k, tri, c_in_cb, lda,
 * ldc, ntiles, nblk, front, assign, swz); returns the number of regions (tools/launch_profile.py) */
int64_t cholmod_hip_debug_launch_regions (cholmod_hip_plan *plan, int64_t launch, int64_t cap, int64_t *out) ;

/* Test hook: run the engine's dense partial factorization on ONE dense front
 * given on the host (column-major nsrow-by-nsrow, lower; the first nscol
 * columns are eliminated; on return F holds [L11; L21] in the first nscol
 * columns and the Schur complement in the rest).  Exercises potrf/trsm/update
 * kernels without any sparse structure. */
int cholmod_hip_dense_partial_factor (double *F, int64_t nsrow, int64_t nscol,
    int flags, int64_t *info) ;

const char *cholmod_hip_version (void) ;

#ifdef __cplusplus
}
#endif
#endif
