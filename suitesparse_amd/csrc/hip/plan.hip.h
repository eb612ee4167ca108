// plan.hip.h -- the types the host-side translation units of the engine share: the launch list (Launch, Schedule), the
// contribution-block arena allocator, the RCCL entry points bound at run time, and the plan itself (cholmod_hip_plan:
// one symbolic factor prepared for one rank).  plan_build.hip derives the plan (etree, ownership, layout, batches),
// schedule_dense.hip the launches of a batch of fronts, engine.hip uploads and runs it.
#pragma once
#include "descriptors.hip.h"
#include "../../../include/cholmod_hip.h"

#include <rccl/rccl.h>      // types only: the library itself is bound with dlopen (cholmod_hip_rccl_attach)

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <functional>
#include <memory>
#include <new>
#include <queue>
#include <vector>
#include <atomic>
#include <chrono>

// Test hooks (CHOLMOD_HIP_TEST_*: stream jitter, poisoned arena, dropped waits, injected failures, a hung exchange) exist
// only in the library built with -DCHOLMOD_HIP_TEST_HOOKS (lib/libcholmod_amd_testhooks.so, loaded by the tests that need
// them); in the product library the names do not even appear as strings: no environment variable can make it compute a
// wrong factor or fail on purpose.
#ifdef CHOLMOD_HIP_TEST_HOOKS
#define TEST_ENV(name) getenv (name)
#else
#define TEST_ENV(name) ((const char *) nullptr)
#endif

namespace sship {

constexpr int NB = PF_NB ;      // inner panel width (potrf / trsm block)
constexpr int MB = 512 ;        // mid block: inner (K = 64) updates stay inside MB columns
// outer block (contraction length of the big trailing updates): MB for small
// fronts, up to 2048 for the largest ones -- the update kernel reaches 52.8 /
// 61.4 TFLOP/s at K = 512 / 2048 on a 16k x 16k region (the 16 B read-modify-
// write of C is amortised over 4x more flops), at the price of OB/nsrow of the
// flops moving to K = MB mid-level updates
struct ObThresholds { int t1, t2, t3 ; } ;
// read at every plan build (tests change the thresholds between plans)
static inline ObThresholds outer_block_thresholds ()
{
    const char *e1 = getenv ("CHOLMOD_HIP_OB1024_ROWS"), *e2 = getenv ("CHOLMOD_HIP_OB2048_ROWS") ;
    const char *e3 = getenv ("CHOLMOD_HIP_OB4096_ROWS") ;
    return ObThresholds {e1 ? atoi (e1) : 4000, e2 ? atoi (e2) : 8000, e3 ? atoi (e3) : 24000} ;
}
static inline int outer_block (int maxrows, const ObThresholds &t)
{
    return maxrows >= t.t3 ? 4096 : maxrows >= t.t2 ? 2048 : maxrows >= t.t1 ? 1024 : MB ;
}
constexpr int BIG = 128, SMALL = 64, BKK = 16 ;
// Several GPUs: the panel of a shared front is stored by slabs of OWN_W columns, slab t on member t % g of
// its group (owner-computes: the outer trailing updates of a slab run on its owner).  Narrower slabs balance
// the members better (a member's columns are all own_w (g - 1) rows taller than the last member's), wider
// ones make fewer, larger update regions.  CHOLMOD_HIP_OWN_W overrides (a multiple of 64 dividing 512).
static inline int own_width ()
{
    const char *e = getenv ("CHOLMOD_HIP_OWN_W") ;
    int w = e ? atoi (e) : 128 ;
    return (w == 64 || w == 128 || w == 256 || w == 512) ? w : 128 ;
}
static inline int front_ob (const FrontD &f, int flags, const ObThresholds &t)
{
    return (flags & CHOLMOD_HIP_FIXED_OB) ? MB : (flags & CHOLMOD_HIP_WIDE_OB) ? 2048 : outer_block (f.nsrow, t) ;
}
// doubles of ONE window buffer of a distributed front (it has two when it has more than one outer block)
static inline i64 window_len (const FrontD &f, int ob) { return (i64) f.nsrow * std::min (ob, (int) f.nscol) ; }
static inline int window_count (const FrontD &f, int ob) { return f.nscol > ob ? 2 : 1 ; }

// K_XCHG_RS / K_XCHG_AG: the exchange of a shared front's block column (multi-GPU): reduce-scatter of
// the partial sums by row chunks before its panel chain, all-gathers of the solved chunks after it (near rows in line,
// far rows on the exchange stream, awaited by a K_JOIN ahead of the outer update)
enum Kind { K_ZERO = 0, K_EA, K_POTRF, K_TRSM, K_UPD_BIG, K_UPD_SMALL, K_JOIN, K_XCHG_RS, K_SMALL, K_UPD_PF, K_TRSM_UPD, K_XCHG_AG, K_UPD_W, K_DIAG, K_ROWSOLVE, K_WIN, K_CHAINF, K_NKIND } ;

struct Launch {
    int kind ;
    int grid ;
    int ng ;
    size_t goff ;       // first group (index into the kind's group array)
    double flops ;      // algorithmic flops (dense kinds)
    double bytes ;      // algorithmic bytes (extend-add / zero)
    int stream = 0 ;    // 0 = main, 1 = exchange stream (window open, pack, collective ahead of time)
    int wait_ev = -1 ;  // event this launch's stream waits for first
    int rec_ev = -1 ;   // event recorded on its stream right after it
    XchgD xd = {0, 0, 0, 0, 0, 1, 0, 0, 0, 0} ;     // K_XCHG_RS / K_XCHG_AG: the block column and its row chunks
    int far = 0 ;                           // K_XCHG_AG: 0 = the near chunks (in line), 1 = the far chunks (exchange stream)
    int ar_g0 = 0, ar_gn = 1 ;              // ... exchanged over the ranks [ar_g0, ar_g0+ar_gn)
    int aux = 0 ;                   // K_TRSM: widest panel of the launch (LDS sizing)
    int leaf_T = 0 ;                // K_SMALL, leaf_pw: doubles of LDS per front (its panel columns, packed)
    int leaf_pw = 0 ;               // K_SMALL: every front is a leaf of <= 32 rows and <= leaf_pw (4/8/12/16) columns: two per wave (k_leaf_pair)
    int ndiag = 0 ;                 // K_CHAINF: diagonal workgroups of the launch (they come first in the grid)
    int half = 0 ;                  // K_UPD_W: two waves per 64 x 64 tile, 64 x 32 each (k_update3<..., HALF>): launches of 512 .. 10 240 tiles
} ;

#define HIPCHK(call) do { hipError_t e_ = (call) ; if (e_ != hipSuccess) { \
    fprintf (stderr, "cholmod_hip: %s failed: %s (%s:%d)\n", #call, \
        hipGetErrorString (e_), __FILE__, __LINE__) ; return CHOLMOD_HIP_GPU_PROBLEM ; } } while (0)

// best-fit free-list allocator for the contribution-block arena (plan time)
struct Arena {
    std::map<i64, i64> free_by_off ;            // off -> len
    std::multimap<i64, i64> free_by_len ;       // len -> off
    i64 top = 0 ;
    void erase_len (i64 len, i64 off)
    {
        auto r = free_by_len.equal_range (len) ;
        for (auto it = r.first ; it != r.second ; ++it)
            if (it->second == off) { free_by_len.erase (it) ; return ; }
    }
    i64 alloc (i64 len)
    {
        if (len == 0) return 0 ;
        auto it = free_by_len.lower_bound (len) ;
        if (it != free_by_len.end ())
        {
            i64 blen = it->first, off = it->second ;
            free_by_len.erase (it) ;
            free_by_off.erase (off) ;
            if (blen > len)
            {
                free_by_off [off + len] = blen - len ;
                free_by_len.insert ({blen - len, off + len}) ;
            }
            return off ;
        }
        // grow: merge with a free block that touches the top, if any
        i64 off = top ;
        if (!free_by_off.empty ())
        {
            auto last = std::prev (free_by_off.end ()) ;
            if (last->first + last->second == top)
            {
                off = last->first ;
                erase_len (last->second, last->first) ;
                free_by_off.erase (last) ;
            }
        }
        top = off + len ;
        return off ;
    }
    void release (i64 off, i64 len)
    {
        if (len == 0) return ;
        auto nx = free_by_off.lower_bound (off) ;
        if (nx != free_by_off.end () && off + len == nx->first)
        {
            len += nx->second ;
            erase_len (nx->second, nx->first) ;
            nx = free_by_off.erase (nx) ;
        }
        if (nx != free_by_off.begin ())
        {
            auto pv = std::prev (nx) ;
            if (pv->first + pv->second == off)
            {
                off = pv->first ;
                len += pv->second ;
                erase_len (pv->second, pv->first) ;
                free_by_off.erase (pv) ;
            }
        }
        free_by_off [off] = len ;
        free_by_len.insert ({len, off}) ;
    }
} ;

struct Schedule {
    std::vector<ZeroGroup> zg ;
    std::vector<EaGroup> eg ;
    std::vector<PfGroup> pg ;
    std::vector<TrGroup> tg ;
    std::vector<GemmGroup> gg ;
    std::vector<DgGroup> dg ;       // k_diag: diagonal sub-blocks (256-column panel chain)
    std::vector<RsGroup> rg ;       // k_rowsolve: the rows below them
    std::vector<WinD> wg ;          // k_win_move: block columns of distributed fronts into / out of their windows
    std::vector<CfGroup> cg ;       // k_chainf: the 256-column chain in one launch (diagonal + row workgroups, flags)
    int ncflags = 0 ;               // flag slots (one per front and sub-block column of the whole schedule)
    int max_dinv_slots = 0 ;        // most diagonal sub-blocks in one launch (size of the inverse buffer)
    std::vector<i32> sm ;           // front ids handled by the fused small-front kernel
    std::vector<Launch> launches ;
    int nevents = 0 ;
} ;

template <typename T> static T *dupload (const std::vector<T> &v, hipError_t &err)
{
    T *d = nullptr ;
    size_t bytes = std::max<size_t> (v.size (), 1) * sizeof (T) ;
    err = hipMalloc ((void **) &d, bytes) ;
    if (err != hipSuccess) return nullptr ;
    if (!v.empty ()) err = hipMemcpy (d, v.data (), v.size () * sizeof (T), hipMemcpyHostToDevice) ;
    return d ;
}

// ---- RCCL, bound at run time (no link-time dependency: the library also serves
// single-GPU callers and CPU-only hosts) ------------------------------------------
struct RcclApi {
    void *h = nullptr ;
    ncclResult_t (*GetUniqueId) (ncclUniqueId *) = nullptr ;
    ncclResult_t (*CommInitRank) (ncclComm_t *, int, ncclUniqueId, int) = nullptr ;
    ncclResult_t (*CommSplit) (ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *) = nullptr ;
    ncclResult_t (*AllReduce) (const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr ;
    ncclResult_t (*ReduceScatter) (const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr ;
    ncclResult_t (*AllGather) (const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr ;
    ncclResult_t (*Broadcast) (const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr ;
    ncclResult_t (*CommDestroy) (ncclComm_t) = nullptr ;
    const char *(*GetErrorString) (ncclResult_t) = nullptr ;
} ;
RcclApi *rccl_api () ;      // (engine.hip)

} // namespace sship

using namespace sship ;

struct cholmod_hip_plan {
    i64 n = 0, nsuper = 0, ssize = 0, xsize = 0 ;
    int flags = 0 ;
    bool host_only = false ;
    std::vector<i64> super, pi, px, Ls ;
    std::vector<FrontD> fr ;
    std::vector<i32> level, child, supermap ;
    std::vector<i32> lvl_ptr, lvl_list ;        // fronts by level
    i64 relsize = 0, arena = 0 ;
    i64 arena_budget = 0 ;                  // bytes the CB arena may take (0 = no limit)
    i64 global_arena = 0 ;                  // arena of the layout over ALL fronts (doubles): what the batch split was chosen by
    std::vector<i32> batch_of ;             // global batch index of every front (the same on every rank)
    int nsplit = 1 ;                        // subtrees swept one after the other (memory)
    int nlevels = 0 ;
    // multi-GPU: one process per GPU; owner[s] = rank that factors front s, or
    // -1 for the shared top fronts every rank holds as partial sums
    int rank = 0, world = 1 ;
    bool force_shared = false ;     // single-rank self test of the exchange path
    std::vector<i32> owner ;
    std::vector<i32> grp0, grpn ;   // ranks [grp0, grp0+grpn) hold front s (grpn == 1: solo)
    std::vector<char> assign_cb ;   // front's CB is written (not updated) by its first trailing update
    std::vector<i32> my_lvl_ptr, my_lvl_list ;  // this rank's fronts by level
    cholmod_hip_allreduce_fn ar_fn = nullptr ;
    void *ar_user = nullptr ;
    // native exchange: communicator of the world and one per rank group of the plan
    // ((first << 16) | size -> communicator); stream-ordered ncclAllReduce calls
    int jitter_us = 0 ; unsigned long long jitter_state = 0 ;     // test hook CHOLMOD_HIP_TEST_JITTER (run_launch)
    int test_drop_waits = 0 ;          // test hook CHOLMOD_HIP_TEST_DROP_WAITS (read per plan, upload_plan)
    int test_hang_rank = -1 ; long test_hang_xchg = -1, test_hang_fact = 2 ;     // test hook CHOLMOD_HIP_TEST_HANG_EXCHANGE=rank:seq (bench.py's watchdog)
    bool upd3_wg4 = false ;             // k_update3 with four tiles per workgroup (CHOLMOD_HIP_UPD3_WG4)
    ncclComm_t nccl_world = nullptr ;
    std::map<i64, ncclComm_t> nccl_group ;
    hipEvent_t ar_done = nullptr ;          // all-reduce on the second stream finished
    double *d_xchg = nullptr ;
    double *d_stage = nullptr ;             // the g segments of a block column (reduce-scatter, in place)
    double *d_ag = nullptr ;                // the g solved near-row chunks of a block column (all-gather, in place)
    double *d_agf = nullptr ;               // ... its far-row chunks (a gather of its own, on the exchange stream)
    i64 stage_len = 0, ag_len = 0, agf_len = 0 ;
    // triangular solves: per level, the supernodes one workgroup handles whole
    // and the big ones walked in SOLVE_SB-column blocks by many workgroups (k_solve_*_blk)
    std::vector<SolveTask> sv_tasks ;       // [whole-supernode tasks by level | block tasks]
    std::vector<i32> sv_ptr ;               // level -> range of whole-supernode tasks
    std::vector<std::vector<i32>> sv_big ;  // level -> big supernodes
    SolveTask *d_sv = nullptr ;
    // explicit inverses of the 64x64 diagonal blocks of the big supernodes (solve
    // only; built lazily after each factorization, see k_diag_inv64)
    std::vector<InvTask> inv_tasks ;        // all blocks, grouped by supernode
    std::vector<i64> inv_first ;            // supernode -> index of its first block
    InvTask *d_inv_tasks = nullptr ;
    double *d_winv = nullptr ;
    bool winv_valid = false ;
    double *d_solved = nullptr ; i64 solved_cap = 0 ;     // side vector Y of the forward walk
    double *d_sv_acc = nullptr ; i64 sv_acc_cap = 0 ;
    unsigned int *d_ticket = nullptr ;
    // the walk, batched over the big supernodes of a level: step b of a level = one
    // launch holding block b of every big supernode of the level that has one
    struct SbLaunch { i32 level, first, ntasks, grid ; } ;
    std::vector<SolveBlk> sb_tasks, sb_commit ;     // [block tasks by launch], [one commit task per big supernode]
    std::vector<SbLaunch> sb_launch, sb_commit_launch ;
    std::vector<i32> sb_lvl_ptr ;                   // level -> range of sb_launch
    SolveBlk *d_sb_tasks = nullptr, *d_sb_commit = nullptr ;
    int sb_max_tasks = 0 ;
    long long *d_thin_tim = nullptr ;       // CHOLMOD_HIP_THIN_TIMING: 10 cycle counters per launch
    CheckTask *d_chk = nullptr ; i64 nchk = 0 ;     // cholmod_hip_factor_checks task list (lazy)
    double *d_chk_out = nullptr ;
    Schedule sch ;
    double exec_flops = 0 ;
    // device
    hipStream_t stream = nullptr ;          // main stream
    hipStream_t stream2 = nullptr ;         // exchange stream (exchange look-ahead of a shared front's next block column)
    std::vector<hipEvent_t> sync_ev ;       // schedule events (no timing)
    i64 *d_Ls = nullptr ;
    FrontD *d_fr = nullptr ;
    i32 *d_supermap = nullptr, *d_child = nullptr, *d_relmap = nullptr, *d_info = nullptr ;
    i32 *d_lvl_list = nullptr ;
    double *d_Lx = nullptr, *d_cb = nullptr ;
    ZeroGroup *d_zg = nullptr ; EaGroup *d_eg = nullptr ; PfGroup *d_pg = nullptr ;
    TrGroup *d_tg = nullptr ; GemmGroup *d_gg = nullptr ; i32 *d_sm = nullptr ;
    DgGroup *d_dg = nullptr ; RsGroup *d_rg = nullptr ; double *d_dinv = nullptr ;     // 256-column panel chain
    // multi-GPU: a rank allocates L only for the fronts it holds (its own subtrees and the shared
    // fronts of its groups), packed in supernode order: lpx [s] = offset of front s in the rank's
    // d_Lx (-1: not held), lx_local = its length.  FrontD.psx is that LOCAL offset, so every kernel
    // of the factorization works on the compact array unchanged.  The complete factor in the
    // reference layout (L->px) exists on a rank only after cholmod_hip_gather_factor: d_Lx_full /
    // d_fr_full (descriptors with the global offsets), what solves, downloads and checks use.
    // One rank: lpx = px, the local array IS the factor.
    std::vector<i64> lpx ;
    i64 lx_local = 0 ;
    // ... and of a SHARED front only the column slabs it owns (FrontD::own_w / own_g / own_r: slab t of own_w columns on
    // member t % own_g of the front's group).  The outer block column a group is factoring lives in windows at the
    // tail of d_Lx (win_off [s], -1: none; schedule_dense: psx_at); lx_fronts = doubles of d_Lx before the windows.
    std::vector<i64> win_off ;
    i64 lx_fronts = 0 ;
    // contributions routed past the contribution blocks of shared fronts (build_host: passthru): per entry of the child
    // lists the offset of that pair's relative map, the (contributor, ancestor) pairs whose maps are computed next to the
    // child -> parent ones, the total length of the map array
    std::vector<i64> crel ;
    std::vector<RelPair> relpairs ;
    i64 relsize_all = 0 ;
    bool passthru = false ;
    i64 *d_crel = nullptr ; RelPair *d_relpairs = nullptr ;
    WinD *d_wg = nullptr ;
    CfGroup *d_cg = nullptr ; int *d_cflags = nullptr ;     // k_chainf groups; its flags ([4 slot + row block]) and, last, the error word
    double *d_Lx_full = nullptr ; FrontD *d_fr_full = nullptr ;
    bool full_valid = false ;
    FrontD *d_smd = nullptr ; i64 *d_sp01 = nullptr ;   // thin launches: descriptor and range of S of every front, in block order (as d_sm)
    ChildD *d_cdesc = nullptr ;      // per entry of the child lists: (cb, rel, ncb, cbp) of that child
    i32 *d_tu_cnt = nullptr ;       // k_trsm_upd: per group, workgroups that have read the rows workgroup 0 overwrites
    double cur_beta = 0 ;
    // resident input matrix
    i64 *d_Sp = nullptr, *d_Si = nullptr, *d_Snz = nullptr ; double *d_Sx = nullptr ;
    i64 s_nz = 0 ; bool s_unpacked = false ;
    int *d_first_fail = nullptr ;           // k_first_fail result
    i64 *d_vsrc = nullptr ; double *d_vals = nullptr ;      // value map of the resident S (cholmod_hip_set_value_map)
    i64 vsrc_nz = 0, vals_n = 0, s_cur_nz = 0 ;
    double *h_vals = nullptr ; i64 h_vals_n = 0 ;           // pinned staging of the value upload (cholmod_hip_values_staging)
    hipEvent_t values_ev = nullptr ; bool values_pending = false ;     // ... and the event behind its gather into S
    // round 6: the same upload in the order the factorization needs it -- the entries of S sorted by the batch of their
    // column's front (cholmod_hip_set_value_map), staged / pushed / applied chunk by chunk; a batch waits for its own chunks only
    std::vector<i64> h_Sp ;                         // host copy of S's column pointers (cholmod_hip_upload_matrix)
    std::vector<int64_t> h_vgather ;                    // staged position -> position in the caller's value array
    i64 *d_vorder = nullptr ;                       // staged position -> entry of S
    std::vector<i64> batch_entries_end ;            // staged entries of the batches 0 .. b
    std::vector<size_t> batch_launch0 ;             // first launch of batch b (build_host)
    std::vector<i32> launch_need_chunks ;           // chunks that must have been applied before launch q
    std::vector<hipEvent_t> chunk_ev ;
    i64 chunk_len = (i64) 1 << 20 ;                 // entries per chunk (8 MB)
    long nchunks = 0, chunks_applied = 0 ;
    std::atomic<long> chunks_pushed {0} ;           // ... by the caller's pushing thread (-1: it failed)
    bool values_chunked = false, prologue_done = false ;
    int cur_mapped = 0 ;
    i64 *d_amap = nullptr ; bool amap_valid = false ;    // S entry -> index in Lx (or -1), built by the first assembly of a resident S
    // solve workspace
    double *d_X = nullptr, *d_Y = nullptr ; i64 x_cap = 0 ;
    i64 *d_perm = nullptr ;
    // progress of the running factorization, readable from another host thread (cholmod_hip_progress): the host side
    // counts what it has enqueued; with markers enabled the device writes, in stream order, the sequence number of the
    // exchange it has entered / left into pinned host memory (prog_dev [0] / [1])
    volatile long long prog_fact = 0, prog_launch = 0, prog_xchg_enq = 0 ;
    long long *prog_dev = nullptr ;
    // stats
    bool profiling = false ;
    double stats [CHOLMOD_HIP_NSTATS] = {0} ;
    double solve_seconds = 0 ;              // device time of the last cholmod_hip_solve (kernels only)
    std::vector<float> launch_ms ;          // per-launch device time of the last profiled factorization
    hipEvent_t ev0 = nullptr, ev1 = nullptr ;
    std::vector<hipEvent_t> evpool ;
} ;

namespace sship {

// plan_build.hip: everything a rank derives from the symbolic factor on the host (supernodal etree, levels, ownership,
// the rank's layout of L and of the arena, batches, launch list)
int build_host (cholmod_hip_plan *P) ;
// schedule_dense.hip: the launches of the dense partial factorization of one batch of fronts
void schedule_dense (const std::vector<FrontD> &fr, const i32 *ids, int nf,
    Schedule &S, int flags, const i32 *owner, const i32 *grp0, const i32 *grpn, int rank, int world,
    const char *assign_cb = nullptr, const i64 *win = nullptr, const i32 *child = nullptr, bool allow_half = false) ;

} // namespace sship
