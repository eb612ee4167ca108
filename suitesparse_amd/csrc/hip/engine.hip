// engine.hip -- host side of the MI355X supernodal Cholesky engine: plan
// (supernodal etree, level sets, contribution-block arena, batched launch
// schedule), the runner, and the extern "C" shim declared in
// include/cholmod_hip.h.  One process drives one GPU; everything runs on one
// HIP stream owned by the plan.
#include "kernels.hip.h"
#include "../../../include/cholmod_hip.h"

#include <rccl/rccl.h>      // types only: the library itself is bound with dlopen (cholmod_hip_rccl_attach)
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
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
#include <chrono>

using namespace sship ;

// Test hooks (CHOLMOD_HIP_TEST_*: stream jitter, poisoned arena, dropped waits, injected failures, a hung exchange) exist
// only in the library built with -DCHOLMOD_HIP_TEST_HOOKS (lib/libcholmod_amd_testhooks.so, loaded by the tests that need
// them); in the product library the names do not even appear as strings: no environment variable can make it compute a
// wrong factor or fail on purpose.
#ifdef CHOLMOD_HIP_TEST_HOOKS
#define TEST_ENV(name) getenv (name)
#else
#define TEST_ENV(name) ((const char *) nullptr)
#endif

namespace {

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
// the partial sums by row chunks before its panel chain, all-gather of the solved chunks after it
enum Kind { K_ZERO = 0, K_EA, K_POTRF, K_TRSM, K_UPD_BIG, K_UPD_SMALL, K_JOIN, K_XCHG_RS, K_SMALL, K_UPD_PF, K_TRSM_UPD, K_XCHG_AG, K_UPD_W, K_DIAG, K_ROWSOLVE, K_WIN, K_CHAINF, K_NKIND } ;

struct Launch {
    int kind ;
    int grid ;
    int ng ;
    size_t goff ;       // first group (index into the kind's group array)
    double flops ;      // algorithmic flops (dense kinds)
    double bytes ;      // algorithmic bytes (extend-add / zero)
    int stream = 0 ;    // 0 = main, 1 = look-ahead (panel) stream
    int wait_ev = -1 ;  // event this launch's stream waits for first
    int rec_ev = -1 ;   // event recorded on its stream right after it
    XchgD xd = {0, 0, 0, 0, 0, 1, 0} ;     // K_XCHG_RS / K_XCHG_AG: the block column and its row chunks
    int ar_g0 = 0, ar_gn = 1 ;              // ... exchanged over the ranks [ar_g0, ar_g0+ar_gn)
    int aux = 0 ;                   // K_TRSM: widest panel of the launch (LDS sizing)
    int leaf_T = 0 ;                // K_SMALL, leaf_pw: doubles of LDS per front (its panel columns, packed)
    int leaf_pw = 0 ;               // K_SMALL: every front is a leaf of <= 32 rows and <= leaf_pw (4/8/12/16) columns: two per wave (k_leaf_pair)
    int ndiag = 0 ;                 // K_CHAINF: diagonal workgroups of the launch (they come first in the grid)
    int pcnt = -1 ;                 // K_UPD_W: >= 0: persistent form (k_update3p), its block of eight tile counters
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
    int npcnt = 0 ;                 // persistent update launches (eight counters each)
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

} // namespace

// ---- RCCL, bound at run time (no link-time dependency: the library also serves
// single-GPU callers and CPU-only hosts) ------------------------------------------
namespace {
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
static RcclApi *rccl_api ()
{
    static RcclApi api ;
    static bool tried = false ;
    if (tried) return api.h ? &api : nullptr ;
    tried = true ;
    // CHOLMOD_HIP_RCCL_LIBRARY names the collective library to bind instead of the system's RCCL
    // (any library exporting the nccl* entry points below; tests/standin_rccl lets several ranks
    // share one GPU, which RCCL itself refuses).  No fallback to RCCL when it is set and missing.
    const char *over = getenv ("CHOLMOD_HIP_RCCL_LIBRARY") ;
    const char *names [] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr} ;
    void *h = nullptr ;
    if (over && *over)
    {
        h = dlopen (over, RTLD_NOW | RTLD_LOCAL) ;
        if (!h) fprintf (stderr, "cholmod_hip: CHOLMOD_HIP_RCCL_LIBRARY=%s: %s\n", over, dlerror ()) ;
    }
    else for (int q = 0 ; names [q] && !h ; q++) h = dlopen (names [q], RTLD_NOW | RTLD_LOCAL) ;
    if (!h) return nullptr ;
    api.GetUniqueId = (decltype (api.GetUniqueId)) dlsym (h, "ncclGetUniqueId") ;
    api.CommInitRank = (decltype (api.CommInitRank)) dlsym (h, "ncclCommInitRank") ;
    api.CommSplit = (decltype (api.CommSplit)) dlsym (h, "ncclCommSplit") ;
    api.AllReduce = (decltype (api.AllReduce)) dlsym (h, "ncclAllReduce") ;
    api.ReduceScatter = (decltype (api.ReduceScatter)) dlsym (h, "ncclReduceScatter") ;
    api.AllGather = (decltype (api.AllGather)) dlsym (h, "ncclAllGather") ;
    api.Broadcast = (decltype (api.Broadcast)) dlsym (h, "ncclBroadcast") ;
    api.CommDestroy = (decltype (api.CommDestroy)) dlsym (h, "ncclCommDestroy") ;
    api.GetErrorString = (decltype (api.GetErrorString)) dlsym (h, "ncclGetErrorString") ;
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommSplit || !api.AllReduce || !api.ReduceScatter || !api.AllGather || !api.Broadcast || !api.CommDestroy) return nullptr ;
    api.h = h ;
    return &api ;
}
}
#define COMMA ,
// (solves, checks) K<true> for a complex factor in its own storage; needs `cxs` in scope
#define CXS_LAUNCH(K, ...) do { if (cxs) hipLaunchKernelGGL (K<true>, __VA_ARGS__) ; else hipLaunchKernelGGL (K<false>, __VA_ARGS__) ; } while (0)
#define RCCLCHK(call) do { ncclResult_t r_ = (call) ; if (r_ != ncclSuccess) { \
    fprintf (stderr, "cholmod_hip: %s failed: %s (%s:%d)\n", #call, \
        (rccl_api () && rccl_api ()->GetErrorString) ? rccl_api ()->GetErrorString (r_) : "?", __FILE__, __LINE__) ; \
    return CHOLMOD_HIP_GPU_PROBLEM ; } } while (0)

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
    bool test_drop_waits = false ;          // test hook CHOLMOD_HIP_TEST_DROP_WAITS (read per plan, upload_plan)
    int test_hang_rank = -1 ; long test_hang_xchg = -1 ;     // test hook CHOLMOD_HIP_TEST_HANG_EXCHANGE=rank:seq (bench.py's watchdog)
    int la_reserve_cu = 0 ;             // CHOLMOD_HIP_LA_RESERVE_CU (tuning, read per plan): CUs with cu_id below it stay free of persistent update waves
    int ncu = 256, la_reserve = 64 ;    // compute units of the device; workgroup slots a persistent update leaves to the panel chain
    int *d_pcnt = nullptr ;             // tile counters of the persistent update launches (8 per launch, zeroed per factorization)
    bool upd3_wg4 = false ;             // k_update3 with four tiles per workgroup (CHOLMOD_HIP_UPD3_WG4)
    ncclComm_t nccl_world = nullptr ;
    std::map<i64, ncclComm_t> nccl_group ;
    hipEvent_t ar_done = nullptr ;          // all-reduce on the second stream finished
    double *d_xchg = nullptr ;
    double *d_stage = nullptr ;             // the g segments of a block column (reduce-scatter, in place)
    double *d_ag = nullptr ;                // the g solved row chunks of a block column (all-gather, in place)
    i64 stage_len = 0, ag_len = 0 ;
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
    hipStream_t stream2 = nullptr ;         // look-ahead (panel) stream
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

namespace {

// Append the launches that perform the dense partial factorization of a batch
// of fronts (all of one etree level): two-level blocked right-looking Cholesky
// of the first nscol columns of every front [panel | CB].
// CHOLMOD_HIP_LOOKAHEAD=1: panel look-ahead on plans of one rank (schedule_dense); read per plan
static bool lookahead_enabled ()
{
    const char *e = getenv ("CHOLMOD_HIP_LOOKAHEAD") ;
    return e && atoi (e) != 0 ;
}

static void schedule_dense (const std::vector<FrontD> &fr, const i32 *ids, int nf,
    Schedule &S, int flags, const i32 *owner, const i32 *grp0, const i32 *grpn, int rank, int world,
    const char *assign_cb = nullptr, const i64 *win = nullptr, const i32 *child = nullptr)
{
    // The real twin of a complex factor (phi embedding, host/complex.c): every row / column pair
    // (2i, 2i+1) is (re, im) of one complex row, the odd columns of a panel are the rotations of the
    // even ones.  The update kernels then contract over the EVEN columns only -- column stride
    // 2 nsrow, K / 2 -- and rebuild the 2 x 2 blocks from the four real products in the lanes
    // (kernels.hip.h: update_tile / update_tile_w, TW): half the flops of the embedding.
    // A complex factor in its own storage (CHOLMOD_HIP_CX_STORAGE; kernels.hip.h: ldcx / stcx): the
    // index space is still the twin's, but only its even columns exist -- column c of a front or
    // of a contribution block lives at (c >> 1) ld, the panels ARE their even columns (operand
    // stride ld, K / 2 contraction steps).
    const bool cx = (flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;
    const bool twin = (flags & CHOLMOD_HIP_PHI_TWIN) != 0 || cx ;
    bool use_big = (flags & CHOLMOD_HIP_TILE128) != 0 && !twin ;
    auto co = [cx] (int c, i64 ld) -> i64 { return cx ? (i64) (c >> 1) * ld : (i64) c * ld ; } ;
    auto twin_operands = [cx] (GemmGroup &G, int origin, int kc)
    {
        // (everything is even in a doubled structure; a plan that claims to be a twin and is not
        // would silently drop a column)
        if ((origin | kc | G.k | G.m | G.n | G.lda | G.ldc) & 1) { fprintf (stderr, "cholmod_hip: twin plan with an odd region\n") ; abort () ; }
        if (!cx) G.lda *= 2 ;
        G.k /= 2 ;
    } ;
    int maxnscol = 0, maxrows = 0 ;
    for (int q = 0 ; q < nf ; q++)
    {
        maxnscol = std::max (maxnscol, fr [ids [q]].nscol) ;
        maxrows = std::max (maxrows, fr [ids [q]].nsrow) ;
    }
    // Outer block width: a property of the FRONT (its row count), not of the batch --
    // the ranks of a multi-GPU group see different batches around the same shared
    // front and must cut its updates into the same regions.
    const ObThresholds obt = outer_block_thresholds () ;
    auto ob_of = [&] (const FrontD &f) -> int { return front_ob (f, flags, obt) ; } ;
    // A distributed front (several GPUs: win [fid] >= 0) has its panel stored by column slabs on their
    // owners; the outer block column being factored lives in a window of nsrow x OB doubles (two of them,
    // used alternately), addressed as if the whole front were there: psx_at (fid, c) is the base to use
    // for anything that touches column c of the front during the panel chain of c's outer block.
    auto windowed = [&] (int fid) { return win && win [fid] >= 0 ; } ;
    auto psx_at = [&] (int fid, int col) -> i64
    {
        const FrontD &f = fr [fid] ;
        if (!windowed (fid)) return f.psx ;
        int OBq = ob_of (f), ob = col / OBq ;
        return win [fid] + (i64) (ob & 1) * window_len (f, OBq) - (i64) ob * OBq * f.nsrow ;
    } ;
    (void) maxrows ;
    std::vector<GemmGroup> pfv ;        // narrow updates whose first tile is factored on the spot (k_update2f)
    std::vector<GemmGroup> wav ;        // regions big enough for one wave per 64 x 64 tile (k_update3)
    // A region goes to k_update3 (one wave per tile: 75 TFLOP/s at K = 4096 against 64 for the
    // four-wave k_update2, 58 against 49 at K = 512; measured, tools/upd3.py) when it has enough
    // tiles to put two waves on every SIMD; below that the four waves per tile of k_update2
    // fill the chip better.  CHOLMOD_HIP_UPD3_MIN_TILES overrides (0 = never).
    // (read at every plan build: tests change it between plans)
    const i64 w_min_tiles = [] () { const char *e = getenv ("CHOLMOD_HIP_UPD3_MIN_TILES") ; return e ? (i64) atoll (e) : (i64) 2048 ; } () ;
    // (likewise read per plan: an in-process A/B that toggles it between plans gets what it asks for)
    const bool by_launch = [] () { const char *e = getenv ("CHOLMOD_HIP_UPD3_BY_LAUNCH") ; return !(e && atoi (e) == 0) ; } () ;
    auto region_tiles = [] (const GemmGroup &G) -> i64
    {
        i64 mt = (G.m + SMALL - 1) / SMALL, nt = (G.n + SMALL - 1) / SMALL ;
        return G.tri ? nt * (nt + 1) / 2 + (mt - nt) * nt : mt * nt ;
    } ;
    // Panel look-ahead (one GPU; CHOLMOD_HIP_LOOKAHEAD): la_on = this batch runs its panel chain on the second stream beside
    // the rest of the previous outer update (decided below, before the first launch of the batch).  tag_chain marks a launch
    // as part of the chain: second stream, behind the event the chain is waiting for (if any).
    bool la_on = false ;
    const bool la_persistent = !getenv ("CHOLMOD_HIP_LA_NO_PERSISTENT") ;
    int la_wait = -1 ;              // event the next chain launch has to wait for (the update that completed its block column)
    long la_last = -1 ;             // index of the last chain launch
    auto tag_chain = [&] (Launch &L)
    {
        if (!la_on) return ;
        L.stream = 1 ;
        if (la_wait >= 0) { L.wait_ev = la_wait ; la_wait = -1 ; }
        la_last = (long) S.launches.size () ;       // (the caller pushes it next)
    } ;
    auto flush_updates = [&] (std::vector<GemmGroup> &big, std::vector<GemmGroup> &small, int on_stream = 0)
    {
        if (w_min_tiles > 0 && !use_big)
        {
            // (a factored-first update of a big region: the diagonal block is factored by a
            // separate launch instead -- 17 us next to milliseconds)
            // ... and what fills the chip is the LAUNCH, not the region: the regions of many mid-size fronts of one
            // level (a few hundred tiles each, K >= 256) together are tens of thousands of tiles -- they go with the
            // big ones when their sum reaches the threshold (CHOLMOD_HIP_UPD3_BY_LAUNCH=0: by region only).  Below
            // K = 256 the update is bound by the read-modify-write of C and the two kernels are on par.
            i64 pooled = 0 ;
            if (by_launch) for (auto &G : small) if (G.k >= 256 || region_tiles (G) >= w_min_tiles) pooled += region_tiles (G) ;
            std::vector<GemmGroup> keep ;
            for (auto &G : small)
            {
                const bool w = region_tiles (G) >= w_min_tiles || (by_launch && pooled >= w_min_tiles && G.k >= 256) ;
                if (w) wav.push_back (G) ; else keep.push_back (G) ;
            }
            small.swap (keep) ;
        }
        for (int pass = 0 ; pass < 4 ; pass++)
        {
            std::vector<GemmGroup> &v = pass == 3 ? wav : pass == 2 ? pfv : pass ? small : big ;
            if (v.empty ()) continue ;
            int T = pass ? SMALL : BIG ;
            Launch L {pass == 3 ? K_UPD_W : pass == 2 ? K_UPD_PF : pass ? K_UPD_SMALL : K_UPD_BIG, 0, (int) v.size (), S.gg.size (), 0, 0} ;
            i64 tiles = 0 ;
            // tuning (CHOLMOD_HIP_UPDW_ONE_REGION=1): every region of a k_update3 launch as a launch of
            // its own, so that tools/launch_profile.py times the regions one by one
            static const bool one_region = getenv ("CHOLMOD_HIP_UPDW_ONE_REGION") != nullptr ;
            auto close_launch = [&] ()
            {
                L.ng = (int) (S.gg.size () - L.goff) ;
                L.grid = (int) tiles ;
                if (L.ng) { if (on_stream == 1) tag_chain (L) ; S.launches.push_back (L) ; }
                L = Launch {L.kind, 0, 0, S.gg.size (), 0, 0} ;
                tiles = 0 ;
            } ;
            for (auto &G : v)
            {
                if (one_region && pass == 3 && S.gg.size () > L.goff) close_launch () ;
                G.mt = (G.m + T - 1) / T ; G.nt = (G.n + T - 1) / T ;
                i64 cnt = G.tri ? (i64) G.nt * (G.nt + 1) / 2 + (i64) (G.mt - G.nt) * G.nt
                                : (i64) G.mt * G.nt ;
                G.ntiles = (i32) cnt ;
                // blocks this rank spends on the group (see decode_tile)
                i64 mine ;
                G.swz = 0 ;
                if (G.tile_mul == 1 && cnt < 1024 && !G.tile_cnt) mine = cnt ;
                else
                {
                    i64 nch = (cnt + 63) / 64 ;
                    i64 mych = nch > G.tile_add ? (nch - G.tile_add + G.tile_mul - 1) / G.tile_mul : 0 ;
                    if (G.tile_cnt) mych = std::min<i64> (G.tile_cnt, nch > G.tile_add ? nch - G.tile_add : 0) ;     // a range of chunks
                    mine = mych * 64 ;
                    G.swz = !(flags & CHOLMOD_HIP_NO_XCD_SWIZZLE) && mine >= 1024 ;
                    // one wave per tile: an XCD runs 256 tiles at a time, so a 16 x 16 super-tile (32 operand
                    // panels per 256 tiles) is one XCD's load.  Opt-in (CHOLMOD_HIP_SWZ16=1): standalone on a
                    // triangular 49 152^2 region, K = 4096, it is 74.6 against 74.4 TFLOP/s and 200 against
                    // 220 GB fetched, inside the 200^3 factorization 105.4 against 105.1 ms per launch and
                    // 478 against 462 GB (same box, top-48 launches): the tiles of an XCD drift apart in k
                    // either way, and the wider strip only widens what they drift over.
                    if (G.swz && pass == 3 && G.tile_mul == 1 && !G.tile_cnt && cnt >= 8192 && getenv ("CHOLMOD_HIP_SWZ16"))
                    {
                        G.swz = 2 ;
                        mine = (cnt + 255) / 256 * 256 ;
                    }
                    if (G.swz) tiles = (tiles + 7) / 8 * 8 ;     // keep block % 8 == XCD aligned
                    else if (G.tile_mul == 1 && !G.tile_cnt) mine = cnt ;
                }
                if (mine == 0) continue ;
                G.nblk = (i32) mine ;
                G.tile_start = (i32) tiles ;
                tiles += mine ;
                double elems = G.tri ? (double) G.n * (G.n + 1) / 2 + (double) (G.m - G.n) * G.n
                                     : (double) G.m * G.n ;
                double share = (G.tile_mul == 1 && !G.tile_cnt) ? 1.0 : std::min (1.0, (double) mine / (double) cnt) ;
                L.flops += 2.0 * elems * G.k * share ;
                L.aux = std::max (L.aux, (int) G.k) ;
                L.bytes += ((G.assign ? 8.0 : 16.0) * elems + 8.0 * ((double) G.m + G.n) * G.k) * share ;
                S.gg.push_back (G) ;
            }
            close_launch () ;
            v.clear () ;
        }
    } ;
    // (owner [] < 0 only occurs with world > 1, or in the single-rank self test
    // CHOLMOD_HIP_SHARE_AS_WORLD that drives the exchange path with one rank)
    auto is_shared = [&] (int fid) { return owner && owner [fid] < 0 ; } ;
    auto add_update = [&] (std::vector<GemmGroup> &big, std::vector<GemmGroup> &small,
        const FrontD &f, int fid, int r0, int kc, int kk, int m, int ncols, bool to_cb,
        bool split = false, bool factor_first = false)
    {
        // target region: rows r0.., cols r0.. of the front (starts on the diagonal)
        if (m <= 0 || ncols <= 0 || kk <= 0) return ;
        GemmGroup G ;
        memset (&G, 0, sizeof (G)) ;
        G.a_off = psx_at (fid, kc) + r0 + co (kc, f.nsrow) ;
        G.b_off = G.a_off ;
        G.lda = f.nsrow ;
        if (to_cb) { G.c_off = f.cb ; G.ldc = f.ncb ; G.c_in_cb = 1 ; }
        else { G.c_off = psx_at (fid, r0) + r0 + co (r0, f.nsrow) ; G.ldc = f.nsrow ; }
        G.m = m ; G.n = ncols ; G.k = kk ; G.tri = 1 ; G.front = fid ;
        if (twin) twin_operands (G, r0, kc) ;
        G.tile_mul = 1 ; G.tile_add = 0 ;
        // first update of a contribution block nobody zeroed: C = -A*B'
        G.assign = (to_cb && kc == 0 && assign_cb && assign_cb [fid]) ? 1 : 0 ;
        if (split && is_shared (fid)) { G.tile_mul = grpn [fid] ; G.tile_add = rank - grp0 [fid] ; }
        if (factor_first) { G.pf_next = 1 ; G.pf_col0 = r0 ; pfv.push_back (G) ; return ; }

        bool isbig = use_big && ncols >= BIG && m >= 2 * BIG ;
        (isbig ? big : small).push_back (G) ;
    } ;
    // The outer update (K = OB, operands in the window of outer block kc / OB) of the in-front columns
    // [ca, cb) of a distributed front: owner-computes -- this rank updates the slabs it stores, in place,
    // every one a region of its own that starts on the diagonal.
    auto add_outer_slabs = [&] (std::vector<GemmGroup> &small, const FrontD &f, int fid, int kc, int kk, int ca, int cb)
    {
        if (kk <= 0) return ;
        for (int c0 = (ca / f.own_w) * f.own_w ; c0 < cb ; c0 += f.own_w)
        {
            if (!col_owned (f, c0)) continue ;
            int a = std::max (c0, ca), b = std::min ({c0 + f.own_w, cb, (int) f.nscol}) ;
            if (b <= a) continue ;
            GemmGroup G ;
            memset (&G, 0, sizeof (G)) ;
            G.a_off = psx_at (fid, kc) + a + co (kc, f.nsrow) ;
            G.b_off = G.a_off ;
            G.lda = f.nsrow ;
            G.c_off = f.psx + a + (i64) col_local (f, a) * f.nsrow ; G.ldc = f.nsrow ;
            G.m = f.nsrow - a ; G.n = b - a ; G.k = kk ; G.tri = 1 ; G.front = fid ;
            if (twin) twin_operands (G, a, kc) ;
            G.tile_mul = 1 ; G.tile_add = 0 ;
            small.push_back (G) ;
        }
    } ;
    std::vector<GemmGroup> big, small ;
    auto record_last = [&] () -> int
    {
        if (S.launches.size () == 0) return -1 ;
        if (S.launches.back ().rec_ev < 0) S.launches.back ().rec_ev = S.nevents++ ;
        return S.launches.back ().rec_ev ;
    } ;
    // Exchange look-ahead (multi-GPU): the update that completes the NEXT 512-column
    // block column of a shared front is issued first (U_next), the rest of the
    // trailing update (U_rest) right behind it, and the block column's all-reduce
    // then runs -- host-driven, staged on the second stream -- while U_rest keeps
    // the chip busy.  early [q] = block column of front q already summed this way.
    const bool xla = !(flags & CHOLMOD_HIP_NO_EXCHANGE_LOOKAHEAD) ;
    std::vector<int> early (nf, -1) ;
    // (not with distributed fronts: their block columns live in windows, see the 64-column chain below)
    bool chain256 = (flags & CHOLMOD_HIP_CHAIN256) != 0 && !cx ;
    // CHOLMOD_HIP_CHAINF_AUTO=1 (tuning): the fused 256-column chain (k_chainf) for the batches it is measured to win on --
    // fronts of at least 192 columns and at most 16 384 rows (one round of row workgroups), none shared between ranks
    if (!chain256 && !cx && !twin && getenv ("CHOLMOD_HIP_CHAINF_AUTO") && atoi (getenv ("CHOLMOD_HIP_CHAINF_AUTO")) != 0 && !(flags & CHOLMOD_HIP_NO_FUSED_POTRF))
    {
        bool any_shared = false ;
        for (int q = 0 ; q < nf ; q++) if (is_shared (ids [q])) any_shared = true ;
        const char *e1 = getenv ("CHOLMOD_HIP_CHAINF_MIN_COLS"), *e2 = getenv ("CHOLMOD_HIP_CHAINF_MAX_ROWS") ;
        chain256 = !any_shared && maxnscol >= (e1 ? atoi (e1) : 192) && maxrows <= (e2 ? atoi (e2) : 16384) ;
    }
    // A batch that holds a front shared between ranks takes the fused 256-column chain by default: the chain of a shared
    // front is the part of a rank's work that does not shrink with the number of ranks, and its 64-column form has no fused
    // kernels there (the diagonal blocks are replicated, the rows dealt by chunks: dpotrf, dtrsm and the narrow updates are
    // separate launches, ~45 us per 64 columns against ~28 in k_chainf).  CHOLMOD_HIP_SHARED_CHAIN64=1: the 64-column chain.
    {
        bool any_win = false ;
        for (int q = 0 ; q < nf ; q++) if (windowed (ids [q])) any_win = true ;
        if (any_win)
        {
            const bool c64 = getenv ("CHOLMOD_HIP_SHARED_CHAIN64") || getenv ("CHOLMOD_HIP_NO_CHAINF") || cx || (flags & CHOLMOD_HIP_NO_FUSED_POTRF) ;
            chain256 = !c64 ;
        }
    }
    const bool fuse_potrf = !(flags & CHOLMOD_HIP_NO_FUSED_POTRF) && !chain256 ;     // (the 256-column chain has no separate dpotrf launches to fuse)
    const bool fuse_trsm = fuse_potrf && !(flags & CHOLMOD_HIP_NO_FUSED_TRSM) ;
    std::vector<int> pf_done (nf, -1) ;     // column whose diagonal block a fused update has factored
    // The exchange of the block column [c0, c1) of shared front q: geometry of its row chunks
    // (kernels.hip.h: XchgD).  This rank keeps the diagonal block and rows [own_lo, own_hi).
    auto xchg_of = [&] (int q, int c0) -> XchgD
    {
        const FrontD &f = fr [ids [q]] ;
        int c1 = std::min (c0 + MB, f.nscol) ;
        int g = grpn [ids [q]], r = rank - grp0 [ids [q]] ;
        if (world == 1) { g = 1 ; r = 0 ; }                 // (single-rank self test of the exchange path)
        int mb = f.nsrow - c1 ;
        int R = mb > 0 ? (((mb + g - 1) / g) + 15) / 16 * 16 : 0 ;
        return XchgD {psx_at (ids [q], c0) + c0 + (i64) c0 * f.nsrow, f.nsrow, c1 - c0, mb, R, g, r} ;
    } ;
    auto emit_rs = [&] (int q, int c0, int wait_ev)
    {
        Launch La {K_XCHG_RS, 0, 0, 0, 0, 0} ;
        La.xd = xchg_of (q, c0) ;
        La.bytes = 8.0 * ((double) La.xd.w * La.xd.w + (double) La.xd.R * La.xd.w) * La.xd.g ;
        La.ar_g0 = grp0 [ids [q]] ; La.ar_gn = grpn [ids [q]] ;
        La.wait_ev = wait_ev ;
        S.launches.push_back (La) ;
    } ;
    auto emit_ag = [&] (int q, int c0)
    {
        Launch La {K_XCHG_AG, 0, 0, 0, 0, 0} ;
        La.xd = xchg_of (q, c0) ;
        if (La.xd.R == 0) return ;                          // nothing below the diagonal block
        La.bytes = 8.0 * (double) La.xd.R * La.xd.w * La.xd.g ;
        La.ar_g0 = grp0 [ids [q]] ; La.ar_gn = grpn [ids [q]] ;
        S.launches.push_back (La) ;
    } ;
    // Window of a distributed front.  open: the block column [b0, b0 + MB) -- the owners' stored columns,
    // zero elsewhere (k_win_move), then the contributions of this rank's children to those columns
    // (extend-add into the window): the rank's partial sum, ready for the reduce-scatter.  On stream 1 behind
    // an event when the block column is opened ahead of time (exchange look-ahead).  close: the factored
    // columns of outer block [o0, o1) into the owners' slabs.
    auto emit_win = [&] (int q, int mode, int ca, int cb, int stream, int wait_ev)
    {
        const FrontD &f = fr [ids [q]] ;
        Launch Lw {K_WIN, 0, 0, S.wg.size (), 0, 0} ;
        Lw.stream = stream ; Lw.wait_ev = wait_ev ;
        int blocks = 0 ;
        for (int b0 = ca ; b0 < cb ; b0 += MB)
        {
            int b1 = std::min (b0 + MB, cb) ;
            S.wg.push_back (WinD {f.psx, psx_at (ids [q], b0), f.nsrow, b0, b1, b0, f.nsrow, f.own_w, f.own_g, f.own_r, mode, blocks}) ;
            blocks += (b1 - b0) * ((f.nsrow - b0 + WIN_ROWS - 1) / WIN_ROWS) ;
            Lw.bytes += 16.0 * (double) (b1 - b0) * (f.nsrow - b0) / f.own_g ;
        }
        Lw.ng = (int) (S.wg.size () - Lw.goff) ; Lw.grid = blocks ;
        if (Lw.ng) S.launches.push_back (Lw) ;
        if (mode == 0 && f.child_end != f.child_begin)
        {
            Launch Le {K_EA, 0, 0, S.eg.size (), 0, 0} ;
            Le.stream = stream ;
            Le.aux = f.nsrow >= 2048 ? 4 : EA_TW ;
            S.eg.push_back (EaGroup {ids [q], 0, ca, cb, psx_at (ids [q], ca)}) ;
            Le.ng = 1 ; Le.grid = (cb - ca + Le.aux - 1) / Le.aux ;
            if (child)
                for (int c = f.child_begin ; c < f.child_end ; c++)
                {
                    // (the part of the child's block that lands in these columns: priced by its share of the columns)
                    double r = fr [child [c]].ncb ;
                    Le.bytes += ((r * (r + 1) / 2) * 24.0 + 4.0 * r) * (double) (cb - ca) / f.nsrow ;
                }
            S.launches.push_back (Le) ;
        }
    } ;
    std::vector<int> early_open (nf, -1) ;  // block column of front q opened ahead of time
    const bool balance_cb = !use_big && !getenv ("CHOLMOD_HIP_NO_CB_BALANCE") ;
    // One trailing-update step: for every listed front, columns [kc, kc+kk) update
    // the in-front columns [t0, t1) (all rows from t0 down) and, if cb, the
    // contribution block.  Steps with kk >= MB are `wide`: their tiles are dealt
    // over the rank group of a shared front and they feed the exchange look-ahead.
    struct Upd { int q, kc, kk, t0, t1 ; bool cb, wide ; } ;
    std::vector<Upd> step ;
    std::function<void ()> emit_step ;
    auto emit_outer_la = [&] (std::vector<Upd> &outer)
    {
        // outer updates of a look-ahead batch, on the main stream behind the chain that produced their panels: first the
        // part that completes the NEXT outer block column of every front (the chain goes on beside what follows), then
        // the rest of the front and the contribution block
        int evc = -1 ;
        if (la_last >= 0)
        {
            if (S.launches [la_last].rec_ev < 0) S.launches [la_last].rec_ev = S.nevents++ ;
            evc = S.launches [la_last].rec_ev ;
        }
        const size_t first = S.launches.size () ;
        for (const Upd &x : outer)
        {
            const FrontD &f = fr [ids [x.q]] ;
            if (x.t1 <= x.t0) continue ;
            int tn = std::min (x.t0 + ob_of (f), x.t1) ;
            add_update (big, small, f, ids [x.q], x.t0, x.kc, x.kk, f.nsrow - x.t0, tn - x.t0, false) ;
        }
        flush_updates (big, small) ;
        const bool any_next = S.launches.size () > first ;
        int eva = any_next ? record_last () : -1 ;
        for (const Upd &x : outer)
        {
            const FrontD &f = fr [ids [x.q]] ;
            int tn = std::min (x.t0 + ob_of (f), x.t1) ;
            if (x.t1 > tn && x.t1 > x.t0) add_update (big, small, f, ids [x.q], tn, x.kc, x.kk, f.nsrow - tn, x.t1 - tn, false) ;
            add_update (big, small, f, ids [x.q], f.nscol, x.kc, x.kk, f.ncb, f.ncb, true) ;
        }
        const size_t first_rest = S.launches.size () ;
        flush_updates (big, small) ;
        // what runs beside the next chain leaves it room: the one-wave-per-tile launches of the rest in their persistent form
        if (any_next && la_persistent)
            for (size_t q = first_rest ; q < S.launches.size () ; q++)
                if (S.launches [q].kind == K_UPD_W) S.launches [q].pcnt = S.npcnt++ ;
        if (S.launches.size () > first && evc >= 0) S.launches [first].wait_ev = evc ;
        if (any_next) la_wait = eva ;
    } ;
    emit_step = [&] ()
    {
        if (la_on)
        {
            std::vector<Upd> outer, inner ;
            for (const Upd &x : step) (x.cb ? outer : inner).push_back (x) ;
            if (!outer.empty ())
            {
                step.swap (inner) ;
                if (!step.empty ()) emit_step () ;      // (the chain's own updates: second stream, through the code below)
                emit_outer_la (outer) ;
                step.clear () ;
                return ;
            }
        }
        const int chain_stream = la_on ? 1 : 0 ;
        bool any_next = false ;
        if (xla)
            for (const Upd &x : step)
            {
                if (!x.wide || !is_shared (ids [x.q]) || x.t1 <= x.t0) continue ;
                const FrontD &f = fr [ids [x.q]] ;
                int tn = std::min (x.t0 + MB, x.t1) ;
                if (x.cb && windowed (ids [x.q])) add_outer_slabs (small, f, ids [x.q], x.kc, x.kk, x.t0, tn) ;
                else add_update (big, small, f, ids [x.q], x.t0, x.kc, x.kk, f.nsrow - x.t0, tn - x.t0, false, true) ;
                any_next = true ;
            }
        int ev_next = -1 ;
        if (any_next) { flush_updates (big, small) ; ev_next = record_last () ; }
        // The fused "update + dpotrf of the next diagonal block" (k_update2f) saves a 17 us launch and runs the
        // update in the four-wave kernel.  Where the K >= 512 chain updates of the fronts of this step are a
        // matrix-core-sized piece of work together (4096 tiles: the mid-size fronts of one level, each below
        // the per-region threshold), the update goes to k_update3 and the diagonal blocks to a dpotrf launch
        // of their own -- as a single big region does (CHOLMOD_HIP_UPD3_BY_LAUNCH=0: by region only).
        const bool ff_by_launch = by_launch ;
        i64 ff_pooled = 0 ;
        if (ff_by_launch && fuse_potrf && w_min_tiles > 0 && !use_big)
            for (const Upd &x : step)
            {
                const FrontD &f = fr [ids [x.q]] ;
                if (is_shared (ids [x.q]) || x.cb || x.kk < 512 || f.nscol - x.t0 < NB || x.t1 - x.t0 < NB) continue ;
                i64 mt = (f.nsrow - x.t0 + SMALL - 1) / SMALL, nt = (x.t1 - x.t0 + SMALL - 1) / SMALL ;
                ff_pooled += nt * (nt + 1) / 2 + (mt - nt) * nt ;
            }
        const bool ff_unfuse_wide_k = ff_pooled >= 2 * w_min_tiles ;
        for (const Upd &x : step)
        {
            const FrontD &f = fr [ids [x.q]] ;
            int c0 = x.t0 ;
            bool ahead = any_next && x.wide && is_shared (ids [x.q]) && x.t1 > x.t0 ;
            if (ahead) c0 = std::min (x.t0 + MB, x.t1) ;
            // a narrow update of the panel chain ends on the next diagonal block: its first
            // tile is that block, and the workgroup that updates it factors it (k_update2f)
            // (also the K >= 512 doubling updates inside an outer block, unless their tiles are
            // dealt over the ranks of a shared front: the factor must exist on every rank)
            bool ff = fuse_potrf && !(x.wide && is_shared (ids [x.q])) && !x.cb && c0 == x.t0 && f.nscol - x.t0 >= NB && x.t1 - c0 >= NB ;
            if (ff && w_min_tiles > 0 && !use_big && !is_shared (ids [x.q]))
            {
                // a region big enough for k_update3 is not fused with the next dpotrf
                i64 mt = (f.nsrow - c0 + SMALL - 1) / SMALL, nt = (x.t1 - c0 + SMALL - 1) / SMALL ;
                if (nt * (nt + 1) / 2 + (mt - nt) * nt >= w_min_tiles) ff = false ;
                if (ff_unfuse_wide_k && x.kk >= 512) ff = false ;
            }
            if (ff) pf_done [x.q] = x.t0 ;
            if (!x.wide && is_shared (ids [x.q]) && x.t1 > c0)
            {
                // a narrow update inside a block column of a shared front: its rows have been
                // dealt to the ranks of the group (reduce-scatter by row chunks, emit_rs) -- the
                // rows of the diagonal block (every rank) and this rank's chunk of the rows below
                XchgD X = xchg_of (x.q, (x.kc / MB) * MB) ;
                int b1 = (x.kc / MB) * MB + X.w ;
                add_update (big, small, f, ids [x.q], c0, x.kc, x.kk, b1 - c0, x.t1 - c0, false, false, ff) ;
                int lo = b1 + X.r * X.R, hi = std::min (lo + X.R, f.nsrow) ;
                if (hi > lo)
                {
                    GemmGroup G ;
                    memset (&G, 0, sizeof (G)) ;
                    const i64 wpsx = psx_at (ids [x.q], x.kc) ;
                    G.a_off = wpsx + lo + co (x.kc, f.nsrow) ;
                    G.b_off = wpsx + c0 + co (x.kc, f.nsrow) ;
                    G.c_off = wpsx + lo + co (c0, f.nsrow) ;
                    G.lda = f.nsrow ; G.ldc = f.nsrow ;
                    G.m = hi - lo ; G.n = x.t1 - c0 ; G.k = x.kk ; G.tri = 0 ; G.front = ids [x.q] ;
                    G.tile_mul = 1 ; G.tile_add = 0 ;
                    if (twin) twin_operands (G, (lo | c0), x.kc) ;
                    small.push_back (G) ;
                }
                continue ;
            }
            if (x.cb && windowed (ids [x.q]))
            {
                // the outer update of a distributed front: its in-front columns slab by slab on their owners,
                // the contribution block as before (partial sums, tiles dealt over the group)
                if (x.t1 > c0) add_outer_slabs (small, f, ids [x.q], x.kc, x.kk, c0, x.t1) ;
            }
            else if (x.t1 > c0) add_update (big, small, f, ids [x.q], c0, x.kc, x.kk, f.nsrow - c0, x.t1 - c0, false, x.wide, ff) ;
            if (x.cb && f.cbd)
            {
                // a distributed contribution block: this rank's block of columns, one region that starts on the diagonal;
                // the first outer block assigns (nothing else ever writes there)
                if (f.cb_hi > f.cb_lo)
                {
                    const int a = f.nscol + f.cb_lo, b = f.nscol + f.cb_hi ;
                    GemmGroup G ;
                    memset (&G, 0, sizeof (G)) ;
                    G.a_off = psx_at (ids [x.q], x.kc) + a + co (x.kc, f.nsrow) ;
                    G.b_off = G.a_off ;
                    G.lda = f.nsrow ;
                    G.c_off = f.cb + f.cb_lo ; G.ldc = f.ncb ; G.c_in_cb = 1 ; G.assign = (x.kc == 0) ? 1 : 0 ;
                    G.m = f.nsrow - a ; G.n = b - a ; G.k = x.kk ; G.tri = 1 ; G.front = ids [x.q] ;
                    if (twin) twin_operands (G, a, x.kc) ;
                    G.tile_mul = 1 ; G.tile_add = 0 ;
                    small.push_back (G) ;
                }
            }
            else if (x.cb)
            {
                size_t nsm = small.size () ;
                add_update (big, small, f, ids [x.q], f.nscol, x.kc, x.kk, f.ncb, f.ncb, true, x.wide) ;
                if (windowed (ids [x.q]) && f.own_g > 1 && balance_cb && small.size () > nsm)
                {
                    // The members' own slabs of this step differ (a member has a slab more, or taller ones): the tiles
                    // of the contribution block -- partial sums, anybody may compute any of them -- are dealt so that
                    // every member ends up with the same number of tiles: member r takes the range of 64-tile chunks
                    // [lo, lo + cnt) that fills it up to the common level.
                    const int g = f.own_g ;
                    std::vector<double> tin (g, 0.0), fill (g, 0.0) ;
                    for (int c0 = (x.t0 / f.own_w) * f.own_w ; c0 < x.t1 ; c0 += f.own_w)
                    {
                        int a = std::max (c0, x.t0), b = std::min ({c0 + f.own_w, x.t1, (int) f.nscol}) ;
                        if (b <= a) continue ;
                        double mt = (f.nsrow - a + SMALL - 1) / SMALL, nt = (b - a + SMALL - 1) / SMALL ;
                        tin [(c0 / f.own_w) % g] += nt * (nt + 1) / 2 + (mt - nt) * nt ;
                    }
                    GemmGroup &G = small.back () ;
                    const i64 nch = (region_tiles (G) + 63) / 64 ;
                    // water level: sum_r max (0, level - tin [r]) = 64 nch
                    std::vector<double> srt (tin) ;
                    std::sort (srt.begin (), srt.end ()) ;
                    double need = 64.0 * nch, level = srt [0] ;
                    for (int q = 0 ; q < g ; q++)
                    {
                        double next = q + 1 < g ? srt [q + 1] : 1e300 ;
                        double room = (next - level) * (q + 1) ;
                        if (room >= need) { level += need / (q + 1) ; need = 0 ; break ; }
                        need -= room ; level = next ;
                    }
                    double cum = 0 ;
                    i64 lo = 0, hi = 0 ;
                    for (int r = 0 ; r <= f.own_r ; r++)
                    {
                        lo = hi ;
                        cum += std::max (0.0, level - tin [r]) ;
                        hi = r + 1 == g ? nch : std::min<i64> (nch, (i64) std::llround (cum / 64.0)) ;
                    }
                    if (hi <= lo) small.pop_back () ;
                    else { G.tile_mul = 1 ; G.tile_add = (i32) lo ; G.tile_cnt = (i32) (hi - lo) ; }
                }
            }
        }
        flush_updates (big, small, chain_stream) ;
        if (any_next)
            for (const Upd &x : step)
            {
                if (!x.wide || !is_shared (ids [x.q]) || x.t1 <= x.t0) continue ;
                int wev = ev_next ;
                if (x.cb && windowed (ids [x.q]))
                {
                    // the first block column of the next outer block: into its window ahead of time, on the
                    // exchange stream behind the update that completed it
                    emit_win (x.q, 0, x.t0, std::min (x.t0 + MB, x.t1), 1, ev_next) ;
                    early_open [x.q] = x.t0 ;
                }
                emit_rs (x.q, x.t0, wev) ;
                early [x.q] = x.t0 ;
            }
        step.clear () ;
    } ;
    // ---- the panel chain in 256-column sub-blocks (kernels.hip.h: k_diag / k_rowsolve).
    // Per sub-block [i0, b1) of a front: one workgroup factors the diagonal sub-block, one launch
    // solves every row below it, and -- recursive doubling over the sub-blocks of the outer block
    // column, as before over 64-column steps -- with e sub-blocks done and p the largest power of
    // two dividing e, the last p sub-blocks (K = 256 p) update the next p; the K = OB update
    // closes the outer block column.  Opt-in (CHOLMOD_HIP_CHAIN256): measured no faster than the 64-column chain below, see DESIGN.md section 4.
    if (chain256)
    {
        const int SB = DG_W ;
        for (int i0 = 0 ; i0 < maxnscol ; i0 += SB)
        {
            if (i0 % MB == 0)
            {
                // a distributed front entering a new outer block column: its block columns into the window (as in the 64-column chain below)
                for (int q = 0 ; q < nf ; q++)
                {
                    const FrontD &f = fr [ids [q]] ;
                    if (f.nscol <= i0 || !windowed (ids [q]) || i0 % ob_of (f) != 0) continue ;
                    int o1 = std::min (i0 + ob_of (f), (int) f.nscol) ;
                    int from = early_open [q] == i0 ? std::min (i0 + MB, o1) : i0 ;
                    if (o1 > from) emit_win (q, 0, from, o1, 0, -1) ;
                }
                for (int q = 0 ; q < nf ; q++)
                {
                    const FrontD &f = fr [ids [q]] ;
                    if (f.nscol <= i0 || !is_shared (ids [q]) || early [q] == i0) continue ;
                    emit_rs (q, i0, -1) ;
                }
            }
            // fused (default with the 256-column chain; CHOLMOD_HIP_NO_CHAINF: off): diagonal and row workgroups of a sub-block in
            // ONE launch, the diagonal sub-block spread over up to four workgroups that hand their row block of L on
            // through flags (k_chainf).  A front shared between ranks: the diagonal sub-block on every rank, below it the rest of
            // the 512-wide diagonal block (every rank) and this rank's chunk of the rows (two row ranges)
            bool fused256 = !getenv ("CHOLMOD_HIP_NO_CHAINF") ;
            if (fused256)
            {
                Launch Lc {K_CHAINF, 0, 0, S.cg.size (), 0, 0} ;
                int dblocks = 0, bblocks = 0, wmax = 0 ;
                for (int q = 0 ; q < nf ; q++)
                {
                    const FrontD &f = fr [ids [q]] ;
                    if (f.nscol <= i0) continue ;
                    int OBq = ob_of (f) ;
                    int o0 = (i0 / OBq) * OBq ;
                    int o1 = std::min (o0 + OBq, (int) f.nscol) ;
                    int b1 = std::min (i0 + SB, o1) ;
                    int w = b1 - i0 ;
                    int slot = (int) (S.cg.size () - Lc.goff) ;
                    int m1 = f.nsrow - b1, off2 = 0, m2 = 0 ;
                    if (is_shared (ids [q]))
                    {
                        XchgD X = xchg_of (q, (i0 / MB) * MB) ;
                        int e1 = (i0 / MB) * MB + X.w ;
                        m1 = e1 - b1 ;
                        int lo2 = e1 + X.r * X.R, hi2 = std::min (lo2 + X.R, (int) f.nsrow) ;
                        off2 = lo2 - i0 ; m2 = std::max (hi2 - lo2, 0) ;
                    }
                    S.cg.push_back (CfGroup {psx_at (ids [q], i0) + i0 + (i64) i0 * f.nsrow, f.nsrow, w, ids [q], i0, m1, slot, S.ncflags++, dblocks, bblocks, off2, m2, 0}) ;
                    dblocks += (w + 63) / 64 ;
                    bblocks += (m1 + 63) / 64 + (m2 + 63) / 64 ;
                    wmax = std::max (wmax, w) ;
                    Lc.flops += (double) w * w * w / 3.0 + (double) (m1 + m2) * w * w ;
                    Lc.bytes += 16.0 * (m1 + m2) * w ;
                }
                Lc.ng = (int) (S.cg.size () - Lc.goff) ; Lc.grid = dblocks + bblocks ; Lc.ndiag = dblocks ; Lc.aux = wmax ;
                S.max_dinv_slots = std::max (S.max_dinv_slots, Lc.ng) ;
                if (Lc.ng) S.launches.push_back (Lc) ;
            }
            Launch Ld {K_DIAG, 0, 0, S.dg.size (), 0, 0} ;
            Launch Lr {K_ROWSOLVE, 0, 0, S.rg.size (), 0, 0} ;
            int rblocks = 0 ;
            for (int q = 0 ; q < nf && !fused256 ; q++)
            {
                const FrontD &f = fr [ids [q]] ;
                if (f.nscol <= i0) continue ;
                int OBq = ob_of (f) ;
                int o0 = (i0 / OBq) * OBq ;
                int o1 = std::min (o0 + OBq, f.nscol) ;
                int b1 = std::min (i0 + SB, o1) ;
                int w = b1 - i0 ;
                int slot = (int) (S.dg.size () - Ld.goff) ;
                S.dg.push_back (DgGroup {psx_at (ids [q], i0) + i0 + (i64) i0 * f.nsrow, f.nsrow, w, ids [q], i0, slot, 0}) ;
                Ld.flops += (double) w * w * w / 3.0 ;
                // rows to solve: everything below the sub-block -- of a shared front the rest of the
                // 512-wide diagonal block (every rank of the group) and this rank's chunk below it
                int lo [2] = {b1, 0}, hi [2] = {f.nsrow, 0} ;
                if (is_shared (ids [q]))
                {
                    XchgD X = xchg_of (q, (i0 / MB) * MB) ;
                    int e1 = (i0 / MB) * MB + X.w ;
                    hi [0] = e1 ;
                    lo [1] = e1 + X.r * X.R ; hi [1] = std::min (lo [1] + X.R, f.nsrow) ;
                }
                for (int part = 0 ; part < 2 ; part++)
                {
                    int m = hi [part] - lo [part] ;
                    if (m <= 0) continue ;
                    S.rg.push_back (RsGroup {psx_at (ids [q], i0) + i0 + (i64) i0 * f.nsrow, psx_at (ids [q], i0) + lo [part] + (i64) i0 * f.nsrow,
                        f.nsrow, m, w, ids [q], i0, rblocks, slot, 0}) ;
                    rblocks += (m + RS_ROWS - 1) / RS_ROWS ;
                    Lr.flops += (double) m * w * w ;
                    Lr.bytes += 16.0 * m * w ;
                }
            }
            Ld.ng = Ld.grid = (int) (S.dg.size () - Ld.goff) ;
            S.max_dinv_slots = std::max (S.max_dinv_slots, Ld.ng) ;
            if (Ld.ng) S.launches.push_back (Ld) ;
            Lr.ng = (int) (S.rg.size () - Lr.goff) ; Lr.grid = rblocks ;
            if (Lr.ng) S.launches.push_back (Lr) ;
            for (int q = 0 ; q < nf ; q++)
            {
                const FrontD &f = fr [ids [q]] ;
                if (f.nscol <= i0 || !is_shared (ids [q])) continue ;
                int b0 = (i0 / MB) * MB ;
                if (i0 + SB >= std::min (b0 + MB, f.nscol)) emit_ag (q, b0) ;
                if (windowed (ids [q]))
                {
                    int OBq = ob_of (f), o0 = (i0 / OBq) * OBq, o1 = std::min (o0 + OBq, (int) f.nscol) ;
                    if (i0 + SB >= o1) emit_win (q, 1, o0, o1, 0, -1) ;
                }
            }
            for (int q = 0 ; q < nf ; q++)
            {
                const FrontD &f = fr [ids [q]] ;
                if (f.nscol <= i0) continue ;
                int OBq = ob_of (f) ;
                int o0 = (i0 / OBq) * OBq ;
                int o1 = std::min (o0 + OBq, f.nscol) ;
                if (i0 + SB >= o1)
                {
                    step.push_back (Upd {q, o0, o1 - o0, o1, f.nscol, true, true}) ;
                    continue ;
                }
                int e = (i0 - o0) / SB + 1 ;
                int p = e & -e ;
                int t0 = o0 + e * SB ;
                int t1 = std::min (o0 + (e + p) * SB, o1) ;
                int kc = o0 + (e - p) * SB ;
                step.push_back (Upd {q, kc, t0 - kc, t0, t1, false, p * SB >= MB}) ;
            }
            emit_step () ;
        }
        return ;
    }
    // ---- panel look-ahead on one GPU (CHOLMOD_HIP_LOOKAHEAD=1): the chain of outer block column k + 1 (dpotrf, panel solves,
    // the doubling updates inside the block column: launches of a few dozen to a few hundred workgroups, 25-40 us each whatever
    // their size) runs on the second stream as soon as the part of outer update k that completes block column k + 1 is done,
    // beside the rest of that update on the main stream; the outer update k + 1 waits for the chain.  The update kernel runs
    // four tiles per workgroup on such plans (k_update3, WPB = 4), so the chain's four-wave workgroups find room beside it.
    if (world == 1 && lookahead_enabled ())
    {
        bool any_shared = false, any_two = false ;
        for (int q = 0 ; q < nf ; q++)
        {
            if (is_shared (ids [q]) || windowed (ids [q])) any_shared = true ;
            if (fr [ids [q]].nscol > ob_of (fr [ids [q]])) any_two = true ;
        }
        la_on = !any_shared && any_two ;
        if (la_on && !S.launches.empty ()) la_wait = record_last () ;       // (whatever assembled these fronts)
    }
    for (int i0 = 0 ; i0 < maxnscol ; i0 += NB)
    {
        // ---- multi-GPU: a 512-column block column of a shared front holds per-rank
        // partial sums (extend-adds of the rank's own subtrees + its share of the
        // earlier wide update tiles); sum them before it is factored.  Only rows
        // >= i0 carry data (above lies the dead upper triangle): they are packed
        // into a staging buffer, halving the volume for the root.
        if (i0 % MB == 0)
        {
            // a distributed front entering a new outer block column: its block columns into the window
            // (all but one opened ahead of time)
            for (int q = 0 ; q < nf ; q++)
            {
                const FrontD &f = fr [ids [q]] ;
                if (f.nscol <= i0 || !windowed (ids [q]) || i0 % ob_of (f) != 0) continue ;
                int o1 = std::min (i0 + ob_of (f), (int) f.nscol) ;
                int from = early_open [q] == i0 ? std::min (i0 + MB, o1) : i0 ;
                if (o1 > from) emit_win (q, 0, from, o1, 0, -1) ;
            }
            for (int q = 0 ; q < nf ; q++)
            {
                const FrontD &f = fr [ids [q]] ;
                if (f.nscol <= i0 || !is_shared (ids [q]) || early [q] == i0) continue ;
                emit_rs (q, i0, -1) ;
            }
        }
        // potrf of the diagonal blocks
        Launch Lp {K_POTRF, 0, 0, S.pg.size (), 0, 0} ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0) continue ;
            if (pf_done [q] == i0) continue ;       // factored by the update that preceded it
            int nb = std::min (NB, f.nscol - i0) ;
            PfGroup G {psx_at (ids [q], i0) + i0 + co (i0, f.nsrow), f.nsrow, nb, ids [q], i0} ;
            S.pg.push_back (G) ;
            Lp.flops += (double) nb * nb * nb / 3.0 ;
        }
        Lp.ng = Lp.grid = (int) (S.pg.size () - Lp.goff) ;
        if (Lp.ng) { tag_chain (Lp) ; S.launches.push_back (Lp) ; }
        // Fronts whose step is "solve, K = 64 update of the next 64 columns, factor the next
        // diagonal block" (every other step of the doubling schedule) take all three in one
        // launch (k_trsm_upd): a full panel, a full next block inside the same outer block
        // column, the front not shared between ranks.
        std::vector<char> fused (nf, 0) ;
        if (fuse_trsm)
        {
            Launch Lf_ {K_TRSM_UPD, 0, 0, S.tg.size (), 0, 0} ;
            int fblocks = 0 ;
            for (int q = 0 ; q < nf ; q++)
            {
                const FrontD &f = fr [ids [q]] ;
                if (f.nscol < i0 + 2 * NB || is_shared (ids [q])) continue ;
                int OBq = ob_of (f) ;
                int o0 = (i0 / OBq) * OBq ;
                int o1 = std::min (o0 + OBq, f.nscol) ;
                if (i0 + 2 * NB > o1) continue ;             // the next block belongs to the outer update
                int e = (i0 - o0) / NB + 1 ;
                if ((e & -e) != 1) continue ;               // p = 1 steps only
                int m = f.nsrow - (i0 + NB) ;
                TrGroup G {f.psx + i0 + co (i0, f.nsrow),
                           f.psx + (i0 + NB) + co (i0, f.nsrow), f.nsrow, m, NB,
                           ids [q], i0, fblocks} ;
                fblocks += (m + TRM_ROWS - 1) / TRM_ROWS ;
                S.tg.push_back (G) ;
                Lf_.flops += (double) m * NB * NB + 2.0 * ((double) m * NB - (double) NB * (NB - 1) / 2) * NB + (double) NB * NB * NB / 3.0 ;
                Lf_.bytes += 8.0 * (3.0 * m * NB) ;
                fused [q] = 1 ;
                pf_done [q] = i0 + NB ;
            }
            Lf_.ng = (int) (S.tg.size () - Lf_.goff) ; Lf_.grid = fblocks ; Lf_.aux = NB ;
            if (Lf_.ng) { tag_chain (Lf_) ; S.launches.push_back (Lf_) ; }
        }
        // trsm of the rows below
        Launch Lt {K_TRSM, 0, 0, S.tg.size (), 0, 0} ;
        int blocks = 0 ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0 || fused [q]) continue ;
            int nb = std::min (NB, f.nscol - i0) ;
            // rows to solve: everything below the diagonal block -- of a shared front, whose
            // block column has been dealt to the ranks by row chunks: the rest of the 512-wide
            // diagonal block (every rank of the group) and this rank's chunk below it
            int lo [2] = {i0 + nb, 0}, hi [2] = {f.nsrow, 0} ;
            if (is_shared (ids [q]))
            {
                XchgD X = xchg_of (q, (i0 / MB) * MB) ;
                int b1 = (i0 / MB) * MB + X.w ;
                hi [0] = b1 ;
                lo [1] = b1 + X.r * X.R ; hi [1] = std::min (lo [1] + X.R, f.nsrow) ;
            }
            for (int part = 0 ; part < 2 ; part++)
            {
                int m = hi [part] - lo [part] ;
                if (m <= 0) continue ;
                TrGroup G {psx_at (ids [q], i0) + i0 + co (i0, f.nsrow),
                           psx_at (ids [q], i0) + lo [part] + co (i0, f.nsrow), f.nsrow, m, nb,
                           ids [q], i0, blocks} ;
                blocks += (m + TRM_ROWS - 1) / TRM_ROWS ;
                S.tg.push_back (G) ;
                Lt.flops += (double) m * nb * nb ;
                Lt.aux = std::max (Lt.aux, (nb + 15) / 16 * 16) ;     // widest panel, in 16-column blocks
            }
        }
        Lt.ng = (int) (S.tg.size () - Lt.goff) ; Lt.grid = blocks ;
        if (Lt.ng) { tag_chain (Lt) ; S.launches.push_back (Lt) ; }
        // a block column of a shared front is complete on the rows of its owners: gather the
        // solved row chunks on every rank of the group before anything uses it as an operand
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0 || !is_shared (ids [q])) continue ;
            int b0 = (i0 / MB) * MB ;
            if (i0 + NB >= std::min (b0 + MB, f.nscol)) emit_ag (q, b0) ;
            // the outer block column of a distributed front is complete in its window: into the owners' slabs
            if (windowed (ids [q]))
            {
                int OBq = ob_of (f), o0 = (i0 / OBq) * OBq, o1 = std::min (o0 + OBq, (int) f.nscol) ;
                if (i0 + NB >= o1) emit_win (q, 1, o0, o1, 0, -1) ;
            }
        }
        // ---- trailing updates.  Inside an outer block column of the front (OB
        // columns, ob_of): recursive doubling -- with e 64-column blocks of it
        // factored and p the largest power of two dividing e, the last p blocks
        // (K = 64 p) update the next p blocks only; every column block is then
        // read-modified-written log2 times instead of once per 64-column step (768
        // instead of 1792 column sweeps per 512 columns), with K up to OB/2 on the
        // matrix cores.  When the outer block column (or the front) is complete:
        // one K = OB update of everything to its right, contribution block included.
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0 || fused [q]) continue ;
            int OBq = ob_of (f) ;
            int o0 = (i0 / OBq) * OBq ;
            int o1 = std::min (o0 + OBq, f.nscol) ;
            if (i0 + NB >= o1)
            {
                step.push_back (Upd {q, o0, o1 - o0, o1, f.nscol, true, true}) ;
                continue ;
            }
            int e = (i0 - o0) / NB + 1 ;
            int p = e & -e ;
            int t0 = o0 + e * NB ;
            int t1 = std::min (o0 + (e + p) * NB, o1) ;
            int kc = o0 + (e - p) * NB ;
            step.push_back (Upd {q, kc, t0 - kc, t0, t1, false, p * NB >= MB}) ;
        }
        emit_step () ;
    }
    if (la_on && la_last >= 0)
    {
        // whatever follows the batch on the main stream comes after the chain's last launch
        if (S.launches [la_last].rec_ev < 0) S.launches [la_last].rec_ev = S.nevents++ ;
        Launch Lj {K_JOIN, 0, 0, 0, 0, 0} ;
        Lj.wait_ev = S.launches [la_last].rec_ev ;
        S.launches.push_back (Lj) ;
    }
}

static int build_host (cholmod_hip_plan *P)
{
    i64 n = P->n, nsuper = P->nsuper ;
    P->fr.resize (nsuper) ;
    P->supermap.resize (std::max<i64> (n, 1)) ;
    P->level.assign (nsuper, 0) ;
    for (i64 s = 0 ; s < nsuper ; s++)
    {
        FrontD &f = P->fr [s] ;
        memset (&f, 0, sizeof (f)) ;
        if (P->super [s+1] - P->super [s] <= 0 || P->pi [s+1] - P->pi [s] > INT32_MAX)
            return CHOLMOD_HIP_INVALID ;
        f.psx = P->px [s] ; f.psi = P->pi [s] ;
        f.k1 = (i32) P->super [s] ;
        f.nscol = (i32) (P->super [s+1] - P->super [s]) ;
        f.nsrow = (i32) (P->pi [s+1] - P->pi [s]) ;
        if (f.nsrow < f.nscol) return CHOLMOD_HIP_INVALID ;
        f.ncb = f.nsrow - f.nscol ;
        f.rel = P->pi [s] - P->super [s] ;      // compact offset, sum of earlier ncb
        for (i64 k = P->super [s] ; k < P->super [s+1] ; k++) P->supermap [k] = (i32) s ;
    }
    P->relsize = P->ssize - n ;
    // supernodal etree (reference t_cholmod_super_numeric.c:1025) and levels
    std::vector<i32> nchild (nsuper, 0) ;
    for (i64 s = 0 ; s < nsuper ; s++)
    {
        FrontD &f = P->fr [s] ;
        f.parent = f.ncb > 0 ? P->supermap [P->Ls [f.psi + f.nscol]] : -1 ;
        if (f.parent >= 0)
        {
            if (f.parent <= s) return CHOLMOD_HIP_INVALID ;
            nchild [f.parent]++ ;
            P->level [f.parent] = std::max (P->level [f.parent], P->level [s] + 1) ;
        }
    }
    // full child lists (tree order), used for the arena lifetimes
    std::vector<i32> cptr (nsuper + 1, 0), call (std::max<i64> (nsuper, 1), 0) ;
    for (i64 s = 0 ; s < nsuper ; s++) cptr [s+1] = cptr [s] + nchild [s] ;
    {
        std::vector<i32> pos (cptr.begin (), cptr.end () - 1) ;
        for (i64 s = 0 ; s < nsuper ; s++)
        {
            i32 p = P->fr [s].parent ;
            if (p >= 0) call [pos [p]++] = (i32) s ;
        }
    }
    // every member of the subtree rooted at `root`, through the real child lists:
    // supernodes need not be numbered in etree postorder (Common->postorder = FALSE,
    // arbitrary maps handed to cholmod_hip_plan_create), so a subtree is not an
    // index range in general -- only parent > child is guaranteed
    std::vector<i32> st_stack ;
    auto for_subtree = [&] (i32 root, auto &&fn)
    {
        st_stack.clear () ;
        st_stack.push_back (root) ;
        while (!st_stack.empty ())
        {
            i32 t = st_stack.back () ; st_stack.pop_back () ;
            fn (t) ;
            for (i32 c = cptr [t] ; c < cptr [t+1] ; c++) st_stack.push_back (call [c]) ;
        }
    } ;
    int nlev = 0 ;
    for (i64 s = 0 ; s < nsuper ; s++) nlev = std::max (nlev, P->level [s] + 1) ;
    P->nlevels = nlev ;
    P->lvl_ptr.assign (nlev + 1, 0) ;
    for (i64 s = 0 ; s < nsuper ; s++) P->lvl_ptr [P->level [s] + 1]++ ;
    for (int l = 0 ; l < nlev ; l++) P->lvl_ptr [l+1] += P->lvl_ptr [l] ;
    P->lvl_list.assign (std::max<i64> (nsuper, 1), 0) ;
    {
        std::vector<i32> pos (P->lvl_ptr.begin (), P->lvl_ptr.end () - 1) ;
        for (i64 s = 0 ; s < nsuper ; s++) P->lvl_list [pos [P->level [s]]++] = (i32) s ;
    }
    // executed flops: sum_s nscol^3/3 + ncb nscol^2 + ncb^2 nscol
    std::vector<double> wsub (nsuper, 0.0) ;
    P->exec_flops = 0 ;
    for (i64 s = 0 ; s < nsuper ; s++)
    {
        double c = P->fr [s].nscol, r = P->fr [s].ncb ;
        double own = c * c * c / 3.0 + r * c * c + r * r * c ;
        P->exec_flops += own ;
        wsub [s] += own ;
        if (P->fr [s].parent >= 0) wsub [P->fr [s].parent] += wsub [s] ;
    }
    // ---- ownership: proportional mapping of the supernodal etree.
    // A front is *shared* while its subtree outweighs 1/(6 world) of the whole
    // factorization; a shared front belongs to a contiguous group of ranks
    // [grp0, grp0+grpn) (the root's group is everybody).  Where the heavy
    // children of a shared front can split its group in proportion to their
    // weights without unbalancing it (<= 10 % above the mean) they get disjoint
    // sub-groups -- a sub-group of one rank owns the child's whole subtree --
    // otherwise they inherit the parent's group.  The light subtrees hanging
    // off the shared region are dealt, largest first, to the least loaded rank
    // of their parent's group (LPT).  Every rank derives the same map.
    P->owner.assign (std::max<i64> (nsuper, 1), 0) ;
    P->assign_cb.assign (std::max<i64> (nsuper, 1), 0) ;
    P->grp0.assign (std::max<i64> (nsuper, 1), 0) ;
    P->grpn.assign (std::max<i64> (nsuper, 1), 1) ;
    // Self test of the exchange path on one GPU: CHOLMOD_HIP_SHARE_AS_WORLD=k with
    // world == 1 marks the fronts a k-rank run would share, so the pack /
    // all-reduce callback / unpack launches run (summing over the single rank).
    int share_world = P->world ;
    if (P->world == 1)
    {
        const char *e = getenv ("CHOLMOD_HIP_SHARE_AS_WORLD") ;
        if (e && atoi (e) > 1) share_world = atoi (e) ;
    }
    P->force_shared = (P->world == 1 && share_world > 1) ;
    if (share_world > 1 && nsuper > 0)
    {
        double total = 0 ;
        for (i64 s = 0 ; s < nsuper ; s++) if (P->fr [s].parent < 0) total += wsub [s] ;
        // (1 / (6 world): with 1 / (4 world) two subtrees of 6 % each stayed atomic at four ranks
        // on Poisson 200^3 and left one rank 7.7 % above the mean; now within 1.2 %)
        double thr_div = 6.0 ;
        if (const char *e = getenv ("CHOLMOD_HIP_SHARE_DIV")) if (atof (e) >= 1.0) thr_div = atof (e) ;
        const double thr = total / (thr_div * share_world) ;
        const bool subgroups = !getenv ("CHOLMOD_HIP_NO_SUBGROUPS") ;
        double split_tol = 1.10 ;
        if (const char *e = getenv ("CHOLMOD_HIP_SPLIT_TOL")) if (atof (e) >= 1.0) split_tol = atof (e) ;
        std::vector<char> shared (nsuper, 0) ;
        std::vector<double> load (share_world, 0.0) ;
        struct Solo { double w ; i32 root, g0, gn ; } ;
        std::vector<Solo> solo ;
        // top-down over the shared region (explicit stack; roots get everybody)
        struct Item { i32 t, g0, gn ; } ;
        std::vector<Item> stack ;
        for (i64 s = nsuper ; s-- > 0 ; )
            if (P->fr [s].parent < 0) stack.push_back (Item {(i32) s, 0, (i32) share_world}) ;
        while (!stack.empty ())
        {
            Item it = stack.back () ; stack.pop_back () ;
            if (wsub [it.t] <= thr || it.gn == 1)
            {
                solo.push_back (Solo {wsub [it.t], it.t, it.g0, it.gn}) ;
                continue ;
            }
            shared [it.t] = 1 ;
            P->grp0 [it.t] = it.g0 ; P->grpn [it.t] = it.gn ;
            {
                double c = P->fr [it.t].nscol, r = P->fr [it.t].ncb ;
                double own = c * c * c / 3.0 + r * c * c + r * r * c ;
                for (int q = it.g0 ; q < it.g0 + it.gn ; q++) load [q] += own / it.gn ;
            }
            // heavy children, heaviest first (ties: lower index)
            std::vector<i32> heavy ;
            for (i32 c = cptr [it.t] ; c < cptr [it.t+1] ; c++)
            {
                if (wsub [call [c]] > thr) heavy.push_back (call [c]) ;
                else solo.push_back (Solo {wsub [call [c]], call [c], it.g0, it.gn}) ;
            }
            std::sort (heavy.begin (), heavy.end (), [&] (i32 x, i32 y)
                { return wsub [x] != wsub [y] ? wsub [x] > wsub [y] : x < y ; }) ;
            int nh = (int) heavy.size () ;
            std::vector<i32> cnt (nh, 0) ;
            bool split = subgroups && nh >= 2 && nh <= it.gn ;
            if (split)
            {
                // largest-remainder apportionment of the gn ranks, at least one each
                double wh = 0 ;
                for (i32 h : heavy) wh += wsub [h] ;
                int left = it.gn ;
                std::vector<double> rem (nh) ;
                for (int q = 0 ; q < nh ; q++)
                {
                    double x = it.gn * wsub [heavy [q]] / wh ;
                    cnt [q] = std::max (1, (int) x) ;
                    rem [q] = x - cnt [q] ;
                    left -= cnt [q] ;
                }
                while (left > 0)
                {
                    int best = 0 ;
                    for (int q = 1 ; q < nh ; q++) if (rem [q] > rem [best]) best = q ;
                    cnt [best]++ ; rem [best] -= 1.0 ; left-- ;
                }
                while (left < 0)
                {
                    int best = -1 ;
                    for (int q = 0 ; q < nh ; q++)
                        if (cnt [q] > 1 && (best < 0 || rem [q] < rem [best])) best = q ;
                    if (best < 0) break ;
                    cnt [best]-- ; rem [best] += 1.0 ; left++ ;
                }
                double worst = 0 ;
                for (int q = 0 ; q < nh ; q++) worst = std::max (worst, wsub [heavy [q]] / cnt [q]) ;
                split = (left == 0) && worst <= split_tol * wh / it.gn ;
            }
            // children are pushed so that they pop in the apportionment order
            int g = it.g0 + it.gn ;
            for (int q = nh ; q-- > 0 ; )
            {
                if (split) { g -= cnt [q] ; stack.push_back (Item {heavy [q], (i32) g, cnt [q]}) ; }
                else stack.push_back (Item {heavy [q], it.g0, it.gn}) ;
            }
        }
        std::stable_sort (solo.begin (), solo.end (), [] (const Solo &x, const Solo &y)
            { return x.w != y.w ? x.w > y.w : x.root < y.root ; }) ;
        for (const Solo &e : solo)
        {
            int best = e.g0 ;
            for (int r = e.g0 + 1 ; r < e.g0 + e.gn ; r++) if (load [r] < load [best]) best = r ;
            load [best] += e.w ;
            for_subtree (e.root, [&] (i32 q)
            {
                P->owner [q] = P->world == 1 ? 0 : best ;
                P->grp0 [q] = P->owner [q] ; P->grpn [q] = 1 ;
            }) ;
        }
        for (i64 s = 0 ; s < nsuper ; s++)
        {
            if (!shared [s]) continue ;
            P->owner [s] = -1 ;
            if (P->world == 1) { P->grp0 [s] = 0 ; P->grpn [s] = 1 ; }
        }
    }
    auto mine = [&] (i64 s) { return P->rank >= P->grp0 [s] && P->rank < P->grp0 [s] + P->grpn [s] ; } ;
    // the rank's own L: the fronts it holds, packed in supernode order (see cholmod_hip_plan::lpx)
    // (a shared front: only the column slabs this rank owns; with one rank -- the self test of the exchange
    // path -- that is every slab, and the front stays where the reference layout has it)
    const bool distribute = !getenv ("CHOLMOD_HIP_NO_DISTRIBUTED_FRONTS") && !(P->flags & CHOLMOD_HIP_CX_STORAGE) ;
    const int ownw = (P->flags & CHOLMOD_HIP_PHI_TWIN) ? std::max (own_width (), 64) : own_width () ;
    // The contribution block of a shared front distributed like its panel (the slabs continue past column nscol, the outer
    // updates of a slab run on its owner), and NOTHING extend-added into it: the contributions of a rank's fronts to a shared
    // ancestor are routed past the blocks in between, straight into the ancestor whose PANEL holds the column -- every entry
    // travels once, no member keeps a full square of partial sums, no replicated extend-add.  CHOLMOD_HIP_NO_CB_PASSTHROUGH=1:
    // the layout of the first half of round 4 (full squares of partial sums, pulled level by level).
    const bool passthru = distribute && !getenv ("CHOLMOD_HIP_NO_CB_PASSTHROUGH") ;
    P->passthru = passthru ;
    // member q's block of a distributed contribution block (ncb columns, group of g) starts where q / g of the lower
    // triangle's area lies to its left (multiples of 64); a function of (ncb, g, q) only: every rank can evaluate it for
    // every member of every group
    auto cb_bound = [] (int ncb, int g, int q) -> int
    {
        if (q <= 0) return 0 ;
        if (q >= g) return ncb ;
        const double T = 0.5 * (double) ncb * (ncb + 1) * q / g ;
        // area left of column j: j ncb - j (j - 1) / 2
        double j = ncb + 0.5 - std::sqrt (std::max (0.0, (ncb + 0.5) * (ncb + 0.5) - 2.0 * T)) ;
        int b = (int) (j / 64.0 + 0.5) * 64 ;
        return std::min (std::max (b, 0), ncb) ;
    } ;
    P->lpx.assign (std::max<i64> (nsuper, 1), -1) ;
    P->win_off.assign (std::max<i64> (nsuper, 1), -1) ;
    P->lx_local = 0 ;
    for (i64 s = 0 ; s < nsuper ; s++)
    {
        FrontD &f = P->fr [s] ;
        if (P->world > 1 && !mine (s)) { f.psx = 0 ; continue ; }
        P->lpx [s] = P->world > 1 ? P->lx_local : P->px [s] ;
        f.psx = P->lpx [s] ;
        i64 cols = f.nscol ;
        if (P->owner [s] < 0 && distribute)
        {
            f.own_w = ownw ; f.own_g = P->world > 1 ? P->grpn [s] : 1 ; f.own_r = P->world > 1 ? P->rank - P->grp0 [s] : 0 ;
            cols = 0 ;
            for (int c0 = 0 ; c0 < f.nscol ; c0 += ownw) if (col_owned (f, c0)) cols += std::min (ownw, f.nscol - c0) ;
            if (passthru && f.ncb > 0)
            {
                // member q's block of contribution-block columns starts where q / g of the lower triangle's area lies to its left
                f.cbd = 1 ;
                f.cb_lo = cb_bound (f.ncb, f.own_g, f.own_r) ; f.cb_hi = cb_bound (f.ncb, f.own_g, f.own_r + 1) ;
            }
        }
        P->lx_local += cols * f.nsrow ;
    }
    if (P->world == 1) P->lx_local = P->xsize ;
    P->lx_fronts = P->lx_local ;
    // thin fronts (fused LDS-resident kernel): their contribution blocks are packed
    // lower triangles, the generic fronts' full squares
    for (i64 s = 0 ; s < nsuper ; s++)
    {
        FrontD &f = P->fr [s] ;
        f.cbp = (!(P->flags & CHOLMOD_HIP_NO_SMALL_FRONTS) && f.nsrow <= SM_MAX && !(P->owner [s] < 0)) ? 1 : 0 ;
        // (a complex front in its own storage: thin all the same, its block the even columns of a square like everybody's: 2)
        if (f.cbp && (P->flags & CHOLMOD_HIP_CX_STORAGE)) f.cbp = 2 ;
    }
    const bool cx_storage = (P->flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;
    auto cb_len = [&] (const FrontD &f) -> i64
    {
        // (a distributed block: this rank's block of columns; a complex front in its own storage: the even columns of the
        // twin's square)
        if (f.cbd) return (i64) f.ncb * (f.cb_hi - f.cb_lo) ;
        return f.cbp == 1 ? (i64) f.ncb * (f.ncb + 1) / 2 : cx_storage ? (i64) f.ncb * (f.ncb / 2) : (i64) f.ncb * f.ncb ;
    } ;
    // The same length in the layout over ALL fronts, from which the batch split is chosen: every rank must derive the same
    // number for every front, whether it holds the front or not -- so nothing here may read f.cbd / f.own_g / f.cb_lo (set
    // for the fronts of THIS rank only; round-4 advisor item: a member of a sub-group counted ncb^2 / g, a non-member ncb^2,
    // and `8 A.top <= budget` could pick different splits on different ranks).  A distributed block counts as its LARGEST
    // member share, ncb * max_q (cb_hi - cb_lo): what the neediest member really allocates.
    std::vector<i64> cb_len_all (std::max<i64> (nsuper, 1), 0) ;
    for (i64 s = 0 ; s < nsuper ; s++)
    {
        const FrontD &f = P->fr [s] ;
        if (P->world > 1 && passthru && P->owner [s] < 0 && f.ncb > 0)
        {
            const int g = P->grpn [s] ;
            int widest = 0 ;
            for (int q = 0 ; q < g ; q++) widest = std::max (widest, cb_bound (f.ncb, g, q + 1) - cb_bound (f.ncb, g, q)) ;
            cb_len_all [s] = (i64) f.ncb * widest ;
        }
        else cb_len_all [s] = f.cbp == 1 ? (i64) f.ncb * (f.ncb + 1) / 2 : cx_storage ? (i64) f.ncb * (f.ncb / 2) : (i64) f.ncb * f.ncb ;
    }
    // who releases whose block: the parent, once it has pulled it -- or, for a front whose parent is shared and whose
    // contributions are routed to the ancestors' panels, the root of its tree (it contributes until then)
    std::vector<std::vector<i32>> rel_list (std::max<i64> (nsuper, 1)) ;
    for (i64 s = 0 ; s < nsuper ; s++)
    {
        i32 p = P->fr [s].parent ;
        if (p < 0) continue ;
        i32 t = p ;
        if (passthru && P->owner [p] < 0) while (P->fr [t].parent >= 0) t = P->fr [t].parent ;
        rel_list [t].push_back ((i32) s) ;
    }
    // this rank's view of the child lists: a shared parent pulls only the
    // contribution blocks this rank computed (its own subtrees and its partial
    // copies of shared children); the other ranks add theirs on their side and
    // the sums meet in the all-reduce of the parent's block columns
    // (passthru: the list of a shared front holds its CONTRIBUTORS -- every front of its subtree this rank holds whose
    // parent is shared, i.e. its own children and the contributors of its shared children -- each with the map of its pair)
    std::vector<std::vector<i32>> contrib (passthru ? nsuper : 0) ;
    P->child.clear () ; P->crel.clear () ; P->relpairs.clear () ;
    P->relsize_all = P->relsize ;
    for (i64 s = 0 ; s < nsuper ; s++)
    {
        FrontD &f = P->fr [s] ;
        f.child_begin = (i32) P->child.size () ;
        if (mine (s))
        {
            const bool route = passthru && P->owner [s] < 0 ;
            for (i32 c = cptr [s] ; c < cptr [s+1] ; c++)
            {
                const i32 d = call [c] ;
                if (!mine (d)) continue ;
                if (!route) { P->child.push_back (d) ; P->crel.push_back (P->fr [d].rel) ; continue ; }
                contrib [s].push_back (d) ;
                if (P->owner [d] < 0) contrib [s].insert (contrib [s].end (), contrib [d].begin (), contrib [d].end ()) ;
            }
            if (route)
                for (i32 d : contrib [s])
                {
                    if (P->fr [d].ncb == 0) continue ;
                    P->child.push_back (d) ;
                    P->crel.push_back (P->relsize_all) ;
                    P->relpairs.push_back (RelPair {d, (i32) s, P->relsize_all}) ;
                    P->relsize_all += P->fr [d].ncb ;
                }
        }
        f.child_end = (i32) P->child.size () ;
        // one rank of the group adds A: the first one -- or, column by column, the owner (distributed fronts)
        f.assemble = (f.own_w ? mine (s) : P->rank == P->grp0 [s]) ? 1 : 0 ;
    }
    if (P->child.empty ()) { P->child.push_back (0) ; P->crel.push_back (0) ; }
    P->my_lvl_ptr.assign (nlev + 1, 0) ;
    P->my_lvl_list.clear () ;
    for (int l = 0 ; l < nlev ; l++)
    {
        for (int q = P->lvl_ptr [l] ; q < P->lvl_ptr [l+1] ; q++)
            if (mine (P->lvl_list [q])) P->my_lvl_list.push_back (P->lvl_list [q]) ;
        P->my_lvl_ptr [l+1] = (i32) P->my_lvl_list.size () ;
    }
    if (P->my_lvl_list.empty ()) P->my_lvl_list.push_back (0) ;
    // ---- execution order and contribution-block arena ---------------------------
    // A *batch* = fronts factored together (one set of launches); a CB lives from
    // its batch until the batch of its parent.  The plain order is one batch per
    // etree level (all fronts of equal height at once): best batching, but every CB
    // of two adjacent levels is alive at the same time (Poisson 200^3: 178 GB).
    // When that does not fit next to L, the tree is cut into `nsplit` subtrees that
    // are swept one after the other (level by level inside each), followed by the
    // top part: the live set shrinks to one subtree's working set + the finished
    // subtree roots + the top levels, at the price of more, smaller launches low in
    // the tree.  nsplit doubles until the arena fits the budget.
    std::vector<std::vector<i32>> batches, best_batches ;
    std::vector<i64> best_cb (std::max<i64> (nsuper, 1), 0) ;
    i64 best_arena = 0 ; int best_nsplit = 1 ;
    if (P->arena_budget < 0)
    {
        // Several ranks must derive the same batch order, so the budget is nominal, not the momentary free memory: what a
        // 288 GB part has left next to the LARGEST part of L any rank of this partition holds (its subtrees, its slabs of
        // the shared fronts; every rank computes all of them), the index maps and a margin for windows and staging.
        // (Rounds 1-3 budgeted the whole factor on every rank: Poisson 200^3 then swept subtrees one after the other -- more,
        // smaller launches -- although a rank of 8 holds 27 of the 181.6 GB.)
        std::vector<double> lxr (P->world, 0.0) ;
        for (i64 s = 0 ; s < nsuper ; s++)
        {
            const double cols = P->fr [s].nscol, rows = P->fr [s].nsrow ;
            if (P->owner [s] >= 0) lxr [P->owner [s]] += cols * rows ;
            else for (int q = P->grp0 [s] ; q < P->grp0 [s] + P->grpn [s] ; q++)
                lxr [q] += (distribute ? std::ceil (cols / (double) (ownw * P->grpn [s])) * ownw : cols) * rows ;
        }
        double worst = 0 ;
        for (double v : lxr) worst = std::max (worst, v) ;
        P->arena_budget = (i64) std::max (1e9, 270e9 - (8.0 * worst + 8.0 * P->ssize + 4.0 * (P->ssize - n) + 12e9)) ;
    }
    i64 budget = P->arena_budget ;
    for (int nsplit = 1 ; ; nsplit *= 2)
    {
        batches.clear () ;
        std::vector<i32> group (std::max<i64> (nsuper, 1), -1) ;     // -1 = top part
        int ngroups = 0 ;
        if (nsplit > 1 && nsuper > 0)
        {
            typedef std::pair<double, i32> WS ;
            std::priority_queue<WS> pq ;
            for (i64 s = 0 ; s < nsuper ; s++) if (P->fr [s].parent < 0) pq.push (WS (wsub [s], (i32) -s)) ;
            while (!pq.empty () && (int) pq.size () < nsplit)
            {
                i32 t = -pq.top ().second ;
                if (cptr [t+1] == cptr [t]) break ;                 // heaviest subtree is a leaf
                pq.pop () ;
                for (i32 c = cptr [t] ; c < cptr [t+1] ; c++) pq.push (WS (wsub [call [c]], -call [c])) ;
            }
            std::vector<i32> roots ;
            while (!pq.empty ()) { roots.push_back (-pq.top ().second) ; pq.pop () ; }
            std::sort (roots.begin (), roots.end ()) ;
            for (i32 r : roots) { for_subtree (r, [&] (i32 q) { group [q] = ngroups ; }) ; ngroups++ ; }
        }
        // batches in postorder: a subtree group contributes its levels in increasing
        // height when its root is reached, a front of the top part is a batch of its
        // own right after its last descendant (supernodes are numbered in postorder,
        // so "increasing index of the unit's last member" is a valid order and frees
        // every contribution block as early as possible)
        {
            std::vector<std::vector<std::vector<i32>>> by (ngroups, std::vector<std::vector<i32>> (nlev)) ;
            std::vector<i32> group_last (ngroups, -1) ;
            for (i64 s = 0 ; s < nsuper ; s++)
                if (group [s] >= 0) { by [group [s]][P->level [s]].push_back ((i32) s) ; group_last [group [s]] = (i32) s ; }
            if (ngroups == 0)
            {
                std::vector<std::vector<i32>> lv (nlev) ;
                for (i64 s = 0 ; s < nsuper ; s++) lv [P->level [s]].push_back ((i32) s) ;
                for (auto &l : lv) if (!l.empty ()) batches.push_back (std::move (l)) ;
            }
            else
            {
                for (i64 s = 0 ; s < nsuper ; s++)
                {
                    if (group [s] < 0) batches.push_back (std::vector<i32> (1, (i32) s)) ;
                    else if (group_last [group [s]] == (i32) s)
                        for (auto &l : by [group [s]]) if (!l.empty ()) batches.push_back (std::move (l)) ;
                }
            }
        }
        Arena A ;
        for (const auto &bt : batches)
        {
            for (i32 sf : bt) { FrontD &f = P->fr [sf] ; f.cb = A.alloc (P->world > 1 ? cb_len_all [sf] : cb_len (f)) ; }
            for (i32 sf : bt)
                for (i32 c : rel_list [sf])
                {
                    FrontD &g = P->fr [c] ;
                    A.release (g.cb, P->world > 1 ? cb_len_all [c] : cb_len (g)) ;
                }
        }
        // keep the first split that fits; if none does, the one with the smallest
        // arena (upload_plan then reports the shortage)
        if (nsplit == 1 || A.top < best_arena)
        {
            best_arena = A.top ; best_nsplit = nsplit ; best_batches = batches ;
            for (i64 s = 0 ; s < nsuper ; s++) best_cb [s] = P->fr [s].cb ;
        }
        if (budget <= 0 || 8 * A.top <= budget || nsplit >= 256 || (nsplit > 1 && ngroups < nsplit / 2)) break ;
    }
    batches.swap (best_batches) ;
    for (i64 s = 0 ; s < nsuper ; s++) P->fr [s].cb = best_cb [s] ;
    P->arena = best_arena ;
    P->nsplit = best_nsplit ;
    P->global_arena = best_arena ;
    P->batch_of.assign (std::max<i64> (nsuper, 1), -1) ;
    for (size_t b = 0 ; b < batches.size () ; b++) for (i32 sf : batches [b]) P->batch_of [sf] = (i32) b ;
    if (P->world > 1)
    {
        // The batch order above is laid out over ALL fronts so that every rank takes
        // the same split decision; the arena itself only has to hold the contribution
        // blocks of this rank's fronts (its subtrees + its shared fronts): lay them
        // out again over the chosen batches, a fraction of the global footprint.
        Arena A ;
        for (const auto &bt : batches)
        {
            for (i32 sf : bt) if (mine (sf)) { FrontD &f = P->fr [sf] ; f.cb = A.alloc (cb_len (f)) ; }
            for (i32 sf : bt)
                if (mine (sf))
                    for (i32 c : rel_list [sf])
                        if (mine (c))
                        {
                            FrontD &g = P->fr [c] ;
                            A.release (g.cb, cb_len (g)) ;
                        }
        }
        P->arena = A.top ;
    }
    // solve tasks (all supernodes: after cholmod_hip_gather_factor every rank holds L)
    P->sv_tasks.clear () ; P->sv_ptr.assign (nlev + 1, 0) ; P->sv_big.assign (nlev, {}) ;
    for (int l = 0 ; l < nlev ; l++)
    {
        for (int q = P->lvl_ptr [l] ; q < P->lvl_ptr [l+1] ; q++)
        {
            i32 sid = P->lvl_list [q] ;
            const FrontD &f = P->fr [sid] ;
            // one workgroup streams ~50-100 GB/s: anything above 512 KB of L gets the
            // multi-workgroup block walk
            if (f.nscol > SOLVE_BIG_COLS || (i64) f.nsrow * f.nscol > ((i64) 1 << 16)) P->sv_big [l].push_back (sid) ;
            else P->sv_tasks.push_back (SolveTask {sid, 0, f.nscol, 1}) ;
        }
        P->sv_ptr [l+1] = (i32) P->sv_tasks.size () ;
    }
    P->inv_tasks.clear () ; P->inv_first.assign (std::max<i64> (nsuper, 1), -1) ;
    P->sb_tasks.clear () ; P->sb_commit.clear () ; P->sb_launch.clear () ; P->sb_commit_launch.clear () ;
    P->sb_lvl_ptr.assign (nlev + 1, 0) ; P->sb_max_tasks = 0 ;
    for (int l = 0 ; l < nlev ; l++)
    {
        int maxblk = 0 ;
        cholmod_hip_plan::SbLaunch Lc {l, (i32) P->sb_commit.size (), 0, 0} ;
        for (i32 sid : P->sv_big [l])
        {
            const FrontD &f = P->fr [sid] ;
            P->inv_first [sid] = (i64) P->inv_tasks.size () ;
            for (int jb = 0 ; jb < f.nscol ; jb += SOLVE_IB)
                P->inv_tasks.push_back (InvTask {sid, jb, (i64) P->inv_tasks.size () * 8192}) ;
            maxblk = std::max (maxblk, (f.nscol + SOLVE_SB - 1) / SOLVE_SB) ;
            P->sb_commit.push_back (SolveBlk {sid, 0, f.nscol, Lc.grid, 0, Lc.ntasks}) ;
            Lc.grid += (f.nscol + 255) / 256 ; Lc.ntasks++ ;
        }
        P->sb_commit_launch.push_back (Lc) ;
        for (int b = 0 ; b < maxblk ; b++)
        {
            cholmod_hip_plan::SbLaunch Lb {l, (i32) P->sb_tasks.size (), 0, 0} ;
            for (i32 sid : P->sv_big [l])
            {
                const FrontD &f = P->fr [sid] ;
                int jb = b * SOLVE_SB ;
                if (jb >= f.nscol) continue ;
                int w = std::min (SOLVE_SB, f.nscol - jb) ;
                int rest = f.nsrow - (jb + w) ;
                P->sb_tasks.push_back (SolveBlk {sid, jb, w, Lb.grid, (i32) (P->inv_first [sid] + jb / SOLVE_IB), Lb.ntasks}) ;
                // workgroups: 256-row chunks below the block x 64-column sub-blocks
                Lb.grid += rest > 0 ? ((rest + 255) / 256) * ((w + 63) / 64) : 0 ; Lb.ntasks++ ;
            }
            P->sb_max_tasks = std::max (P->sb_max_tasks, (int) Lb.ntasks) ;
            P->sb_launch.push_back (Lb) ;
        }
        P->sb_lvl_ptr [l+1] = (i32) P->sb_launch.size () ;
    }
    // windows of the distributed fronts: at the tail of the rank's array, alive for the front's batch only
    // (the region is as long as the neediest batch)
    {
        const ObThresholds obt = outer_block_thresholds () ;
        i64 longest = 0 ;
        for (const auto &bt : batches)
        {
            i64 at = 0 ;
            for (i32 sf : bt)
            {
                const FrontD &f = P->fr [sf] ;
                if (!mine (sf) || !f.own_w) continue ;
                int ob = front_ob (f, P->flags, obt) ;
                P->win_off [sf] = P->lx_fronts + at ;
                at += window_count (f, ob) * window_len (f, ob) ;
            }
            longest = std::max (longest, at) ;
        }
        P->lx_local = P->lx_fronts + longest ;
    }
    // launch schedule of this rank
    Schedule &S = P->sch ;
    std::vector<i32> mine_ids ;
    for (const auto &bt : batches)
    {
        mine_ids.clear () ;
        for (i32 sf : bt) if (mine (sf)) mine_ids.push_back (sf) ;
        const i32 *all_ids = mine_ids.data () ;
        int all_nf = (int) mine_ids.size () ;
        if (all_nf == 0) continue ;
        // thin fronts go to the fused LDS-resident kernel, in three size classes
        // so that the dynamic LDS of a launch fits its widest member
        std::vector<i32> gen ;
        {
            // size classes by rows (the LDS of a launch is sized by its widest member;
            // packed triangle: 4.4 / 9.6 / 16.9 KB, one wave per front -> the 32-wave cap
            // or 16 / 9 fronts per CU; 37.7 / 75.7 KB, four waves per front -> 4 / 2 per CU)
            static const int NCLS = 5 ;
            static const int cls [NCLS] = {32, 48, 64, 96, SM_MAX} ;
            std::vector<i32> bucket [NCLS] ;
            for (int q = 0 ; q < all_nf ; q++)
            {
                i32 sid = all_ids [q] ;
                const FrontD &f = P->fr [sid] ;
                if (!f.cbp) { gen.push_back (sid) ; continue ; }
                int c = 0 ;
                while (c < NCLS - 1 && f.nsrow > cls [c]) c++ ;
                bucket [c].push_back (sid) ;
            }
            // a class too thin to fill the chip rides with the next larger one (a launch
            // costs more than the occupancy it would win)
            for (int c = 0 ; c < NCLS - 1 ; c++)
            {
                if (bucket [c].empty () || bucket [c].size () >= 256) continue ;
                int up = c + 1 ;
                while (up < NCLS - 1 && bucket [up].empty ()) up++ ;
                if (bucket [up].empty ()) continue ;
                bucket [up].insert (bucket [up].end (), bucket [c].begin (), bucket [c].end ()) ;
                std::sort (bucket [up].begin (), bucket [up].end ()) ;
                bucket [c].clear () ;
            }
            for (int c = 0 ; c < NCLS ; c++)
            {
                if (bucket [c].empty ()) continue ;
                Launch Ls_ {K_SMALL, (int) bucket [c].size (), (int) bucket [c].size (), S.sm.size (), 0, 0} ;
                int mx = 0, mxc = 0, mxt = 0 ;
                bool leaves = !(P->flags & CHOLMOD_HIP_NO_LEAF_PAIRS) && !(P->flags & CHOLMOD_HIP_CX_STORAGE) ;
                for (i32 sid : bucket [c])
                {
                    FrontD &f = P->fr [sid] ;
                    if (f.assemble == 1) f.assemble = 2 ;
                    mx = std::max (mx, f.nsrow) ;
                    mxc = std::max (mxc, f.nscol) ;
                    mxt = std::max (mxt, f.nscol * f.nsrow - f.nscol * (f.nscol - 1) / 2) ;
                    if (f.child_end != f.child_begin) leaves = false ;
                    double cc = f.nscol, r = f.ncb ;
                    Ls_.flops += cc * cc * cc / 3.0 + r * cc * cc + r * r * cc ;
                    Ls_.bytes += 8.0 * (f.nsrow * cc + r * (r + 1) / 2) ;
                    for (int ch = f.child_begin ; ch < f.child_end ; ch++)
                    {
                        double rc = P->fr [P->child [ch]].ncb ;
                        Ls_.bytes += 8.0 * rc * (rc + 1) / 2 + 4.0 * rc ;
                    }
                    S.sm.push_back (sid) ;
                }
                Ls_.aux = mx ;                              // widest member: LDS sizing, waves per front
                if (leaves && mx <= 32 && mxc <= 16) { Ls_.leaf_pw = (mxc + 3) / 4 * 4 ; Ls_.leaf_T = (mxt + 31) / 32 * 32 ; }
                S.launches.push_back (Ls_) ;
            }
        }
        const i32 *ids = gen.data () ;
        int nf = (int) gen.size () ;
        if (nf == 0) continue ;
        // Contribution blocks of unshared fronts are never zero-filled: their first
        // trailing update writes C = -L21*L21' (GemmGroup.assign) and the children's
        // contributions to the CB part are extend-added after the dense phase.  Only
        // the children's contributions to the PANEL must be in place before it.
        // (Shared fronts keep the zero-fill: a rank writes only its share of the CB
        // tiles, the rest must read as zero in its partial sum.)
        bool can_assign = !(P->flags & CHOLMOD_HIP_NO_CB_ASSIGN) ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = P->fr [ids [q]] ;
            // (a distributed contribution block is written slab by slab by its owners and takes nothing else: assigned too)
            P->assign_cb [ids [q]] = (f.cbd || (can_assign && f.ncb > 0 && P->owner [ids [q]] >= 0)) ? 1 : 0 ;
        }
        Launch Lz {K_ZERO, 0, 0, S.zg.size (), 0, 0} ;
        int blocks = 0 ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = P->fr [ids [q]] ;
            if (f.ncb == 0 || P->assign_cb [ids [q]]) continue ;
            S.zg.push_back (ZeroGroup {f.cb, (i64) f.ncb, blocks, (P->flags & CHOLMOD_HIP_CX_STORAGE) ? 1 : 0}) ;
            blocks += (((P->flags & CHOLMOD_HIP_CX_STORAGE) ? f.ncb / 2 : f.ncb) + ZERO_COLS - 1) / ZERO_COLS ;
            Lz.bytes += 4.0 * (double) f.ncb * f.ncb ;
        }
        Lz.ng = (int) (S.zg.size () - Lz.goff) ; Lz.grid = blocks ;
        if (Lz.ng) S.launches.push_back (Lz) ;
        for (int phase = 0 ; phase < 2 ; phase++)
        {
            // phase 0 (before the dense phase): everything into the panel columns, and
            // into the CB columns of the zero-filled fronts; phase 1 (after it): the CB
            // columns of the assign fronts
            Launch Le {K_EA, 0, 0, S.eg.size (), 0, 0} ;
            blocks = 0 ;
            // target columns per workgroup: 8, or 4 when the launch holds a big front (measured at
            // 4 / 8 / 16 / 32: the nd24k stand-in and Poisson 100^3 like 4, the 2D problem 8)
            int tw = EA_TW ;
            for (int q = 0 ; q < nf ; q++)
                if (P->fr [ids [q]].child_end != P->fr [ids [q]].child_begin && P->fr [ids [q]].nsrow >= 2048) tw = getenv ("CHOLMOD_HIP_EA_TW_BIG") ? atoi (getenv ("CHOLMOD_HIP_EA_TW_BIG")) : 4 ;
            Le.aux = tw ;
            for (int q = 0 ; q < nf ; q++)
            {
                const FrontD &f = P->fr [ids [q]] ;
                if (f.child_end == f.child_begin) continue ;
                bool asg = P->assign_cb [ids [q]] != 0 ;
                if (f.cbd) continue ;               // (nothing is extend-added into a distributed contribution block)
                // (a distributed front takes the contributions to its panel block column by block column,
                // when the block column enters the window: schedule_dense, emit_win)
                int lo = (phase == 0 && !f.own_w) ? 0 : f.nscol ;
                int hi = phase == 0 ? (asg ? f.nscol : f.nsrow) : f.nsrow ;
                if (phase == 1 && !asg) continue ;
                if (hi <= lo) continue ;
                S.eg.push_back (EaGroup {ids [q], blocks, lo, hi, EA_NO_PBASE}) ;
                blocks += (hi - lo + tw - 1) / tw ;
                if (phase == 0)
                    for (int c = f.child_begin ; c < f.child_end ; c++)
                    {
                        double r = P->fr [P->child [c]].ncb ;
                        Le.bytes += (r * (r + 1) / 2) * 24.0 + 4.0 * r ;   // CB read + target RMW + map
                    }
            }
            Le.ng = (int) (S.eg.size () - Le.goff) ; Le.grid = blocks ;
            if (Le.ng) S.launches.push_back (Le) ;
            if (phase == 0)
                schedule_dense (P->fr, ids, nf, S, P->flags, P->owner.data (), P->grp0.data (), P->grpn.data (),
                    P->rank, P->world, P->assign_cb.data (), P->win_off.data (), P->child.data ()) ;
        }
    }
    return CHOLMOD_HIP_OK ;
}

static void free_device (cholmod_hip_plan *P)
{
    if (P->prog_dev) { (void) hipHostFree (P->prog_dev) ; P->prog_dev = nullptr ; }
    if (RcclApi *R = (P->nccl_world ? rccl_api () : nullptr))
    {
        for (auto &g : P->nccl_group) (void) R->CommDestroy (g.second) ;
        (void) R->CommDestroy (P->nccl_world) ;
        P->nccl_group.clear () ; P->nccl_world = nullptr ;
    }
    if (P->ar_done) (void) hipEventDestroy (P->ar_done) ;
    void *ptrs [] = {P->d_Ls, P->d_fr, P->d_supermap, P->d_child, P->d_relmap, P->d_info,
        P->d_lvl_list, P->d_Lx, P->d_cb, P->d_zg, P->d_eg, P->d_pg, P->d_tg, P->d_tu_cnt, P->d_cdesc, P->d_smd, P->d_sp01, P->d_gg, P->d_sm,
        P->d_Sp, P->d_Si, P->d_Snz, P->d_Sx, P->d_amap, P->d_X, P->d_Y, P->d_perm, P->d_xchg, P->d_stage, P->d_ag, P->d_Lx_full, P->d_fr_full, P->d_dg, P->d_rg, P->d_wg, P->d_cg, P->d_cflags, P->d_pcnt, P->d_crel, P->d_relpairs, P->d_dinv, P->d_sv,
        P->d_inv_tasks, P->d_winv, P->d_solved, P->d_sv_acc, P->d_ticket, P->d_chk, P->d_chk_out, P->d_thin_tim, P->d_sb_tasks, P->d_sb_commit, P->d_first_fail, P->d_vsrc, P->d_vals} ;
    for (void *p : ptrs) if (p) (void) hipFree (p) ;
    for (auto e : P->evpool) (void) hipEventDestroy (e) ;
    for (auto e : P->sync_ev) (void) hipEventDestroy (e) ;
    if (P->stream2) (void) hipStreamDestroy (P->stream2) ;
    if (P->ev0) (void) hipEventDestroy (P->ev0) ;
    if (P->ev1) (void) hipEventDestroy (P->ev1) ;
    if (P->stream) (void) hipStreamDestroy (P->stream) ;
}

static int upload_plan (cholmod_hip_plan *P)
{
    hipError_t e ;
    const bool ptiming = getenv ("CHOLMOD_HIP_PLAN_TIMING") != nullptr ;
    auto pnow = [] () { return std::chrono::duration<double> (std::chrono::steady_clock::now ().time_since_epoch ()).count () ; } ;
    double tu0 = pnow () ;
    // tuning (CHOLMOD_HIP_CU_MASK_32THS = k, 1 .. 31): the main stream on k / 32 of the CUs of every XCD
    // (mask bit b <-> CU b / 8 of XCD b % 8 on this part, tools/cumask.py): what the kernels of a
    // factorization cost on a share of the chip (look-ahead arithmetic)
    if (const char *e = getenv ("CHOLMOD_HIP_CU_MASK_32THS"))
    {
        int k = atoi (e) ;
        if (k < 1 || k > 31) return CHOLMOD_HIP_INVALID ;
        uint32_t m [8] ;
        for (int w = 0 ; w < 8 ; w++) m [w] = (4 * w + 4 <= k) ? 0xFFFFFFFFu : (4 * w >= k) ? 0u : (uint32_t) ((1ull << (8 * (k - 4 * w))) - 1) ;
        HIPCHK (hipExtStreamCreateWithCUMask (&P->stream, 8, m)) ;
    }
    else HIPCHK (hipStreamCreate (&P->stream)) ;
    if (const char *e = TEST_ENV ("CHOLMOD_HIP_TEST_JITTER"))
    {
        unsigned long long seed = 0 ; int mx = 2000 ;
        if (sscanf (e, "%llu:%d", &seed, &mx) >= 1 && mx > 0)
        {
            P->jitter_us = mx ;
            P->jitter_state = seed * 0x9E3779B97F4A7C15ull + (unsigned long long) (P->rank + 1) * 0xD1B54A32D192ED03ull ;
        }
    }
    P->test_drop_waits = TEST_ENV ("CHOLMOD_HIP_TEST_DROP_WAITS") != nullptr ;
    if (const char *e = TEST_ENV ("CHOLMOD_HIP_TEST_HANG_EXCHANGE")) (void) sscanf (e, "%d:%ld", &P->test_hang_rank, &P->test_hang_xchg) ;
    // several ranks: k_update3 with four tiles per workgroup, so that the exchange stream's (and RCCL's) four-wave workgroups
    // find room beside a trailing update (rocprofv3, rank 0 of 8 at 200^3: k_win_move 959 -> 94 ms in all, longest launch
    // 79 -> 1.2 ms; the update itself 2351 -> 2378 ms).  CHOLMOD_HIP_UPD3_WG4=0 / 1 forces either form.
    P->upd3_wg4 = P->world > 1 || P->force_shared || lookahead_enabled () ;
    if (const char *e = getenv ("CHOLMOD_HIP_UPD3_WG4")) P->upd3_wg4 = atoi (e) != 0 ;
    {
        // The exchange stream runs BESIDE the rest of a trailing update (look-ahead: window open, extend-add, pack, the
        // collective): its small workgroups must get the wave slots the update's tiles free up, ahead of the update's own
        // remaining tiles -- at default priority rocprofv3 shows k_win_move stretched over the whole 78 ms of the update it
        // was meant to hide behind (round 4).  CHOLMOD_HIP_NO_STREAM_PRIORITY=1: as before.
        int least = 0, greatest = 0 ;
        if (!getenv ("CHOLMOD_HIP_NO_STREAM_PRIORITY") && hipDeviceGetStreamPriorityRange (&least, &greatest) == hipSuccess && greatest != least)
            HIPCHK (hipStreamCreateWithPriority (&P->stream2, hipStreamDefault, greatest)) ;
        else HIPCHK (hipStreamCreate (&P->stream2)) ;
    }
    for (int q = 0 ; q < P->sch.nevents ; q++)
    {
        hipEvent_t e ;
        HIPCHK (hipEventCreateWithFlags (&e, hipEventDisableTiming)) ;
        P->sync_ev.push_back (e) ;
    }
    HIPCHK (hipEventCreate (&P->ev0)) ;
    HIPCHK (hipEventCreate (&P->ev1)) ;
    size_t freeb = 0, totalb = 0 ;
    HIPCHK (hipMemGetInfo (&freeb, &totalb)) ;
    double need = 8.0 * P->lx_local + 8.0 * P->arena + 8.0 * P->ssize + 4.0 * P->relsize
        + sizeof (GemmGroup) * (double) P->sch.gg.size () + 64.0 * P->nsuper + (double) (64 << 20) ;
    if (need > (double) freeb)
    {
        fprintf (stderr, "cholmod_hip: factor needs %.2f GB of HBM, %.2f GB free\n",
            need / 1e9, freeb / 1e9) ;
        return CHOLMOD_HIP_OUT_OF_MEMORY ;
    }
    double tu1 = pnow () ;
    P->d_Ls = dupload (P->Ls, e) ; HIPCHK (e) ;
    P->d_fr = dupload (P->fr, e) ; HIPCHK (e) ;
    P->d_supermap = dupload (P->supermap, e) ; HIPCHK (e) ;
    P->d_child = dupload (P->child, e) ; HIPCHK (e) ;
    P->d_crel = dupload (P->crel, e) ; HIPCHK (e) ;
    P->d_relpairs = dupload (P->relpairs, e) ; HIPCHK (e) ;
    {
        std::vector<ChildD> cdv (P->child.size ()) ;
        for (size_t q = 0 ; q < P->child.size () ; q++)
        {
            if (P->child [q] < 0 || (size_t) P->child [q] >= P->fr.size ()) { cdv [q] = ChildD {0, 0, 0, 1} ; continue ; }   // (padding entry of an empty list)
            const FrontD &cf = P->fr [P->child [q]] ;
            cdv [q] = ChildD {cf.cb, cf.rel, cf.ncb, cf.cbp} ;
        }
        P->d_cdesc = dupload (cdv, e) ; HIPCHK (e) ;
    }
    P->d_lvl_list = dupload (P->lvl_list, e) ; HIPCHK (e) ;
    P->d_zg = dupload (P->sch.zg, e) ; HIPCHK (e) ;
    P->d_eg = dupload (P->sch.eg, e) ; HIPCHK (e) ;
    P->d_pg = dupload (P->sch.pg, e) ; HIPCHK (e) ;
    P->d_tg = dupload (P->sch.tg, e) ; HIPCHK (e) ;
    HIPCHK (hipMalloc ((void **) &P->d_tu_cnt, std::max<size_t> (P->sch.tg.size (), 1) * sizeof (i32))) ;
    if (P->sch.npcnt > 0)
    {
        HIPCHK (hipMalloc ((void **) &P->d_pcnt, 8 * (size_t) P->sch.npcnt * sizeof (int))) ;
        int dev = 0, ncu = 0 ;
        if (hipGetDevice (&dev) == hipSuccess && hipDeviceGetAttribute (&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) P->ncu = ncu ;
        if (const char *e = getenv ("CHOLMOD_HIP_LA_RESERVE")) P->la_reserve = atoi (e) ;
        if (const char *e = getenv ("CHOLMOD_HIP_LA_RESERVE_CU")) P->la_reserve_cu = atoi (e) ;
    }
    P->d_gg = dupload (P->sch.gg, e) ; HIPCHK (e) ;
    P->d_dg = dupload (P->sch.dg, e) ; HIPCHK (e) ;
    P->d_rg = dupload (P->sch.rg, e) ; HIPCHK (e) ;
    P->d_wg = dupload (P->sch.wg, e) ; HIPCHK (e) ;
    P->d_cg = dupload (P->sch.cg, e) ; HIPCHK (e) ;
    HIPCHK (hipMalloc ((void **) &P->d_cflags, (4 * (size_t) P->sch.ncflags + 4) * sizeof (int))) ;
    HIPCHK (hipMalloc ((void **) &P->d_dinv, (size_t) std::max (P->sch.max_dinv_slots, 1) * 4096 * sizeof (double))) ;
    P->d_sm = dupload (P->sch.sm, e) ; HIPCHK (e) ;
    {
        std::vector<FrontD> smd (P->sch.sm.size ()) ;
        for (size_t q = 0 ; q < smd.size () ; q++) smd [q] = P->fr [P->sch.sm [q]] ;
        P->d_smd = dupload (smd, e) ; HIPCHK (e) ;
        HIPCHK (hipMalloc ((void **) &P->d_sp01, std::max<size_t> (2 * smd.size (), 1) * sizeof (i64))) ;
    }
    P->d_sv = dupload (P->sv_tasks, e) ; HIPCHK (e) ;
    double tu2 = pnow () ;
    HIPCHK (hipMalloc ((void **) &P->d_relmap, std::max<i64> (std::max (P->relsize, P->relsize_all), 1) * sizeof (i32))) ;
    HIPCHK (hipMalloc ((void **) &P->d_info, std::max<i64> (P->nsuper, 1) * sizeof (i32))) ;
    HIPCHK (hipMalloc ((void **) &P->d_first_fail, sizeof (int))) ;
    // test hook: behave as if the reservation of L failed (degradation tests)
    if (TEST_ENV ("CHOLMOD_HIP_TEST_FAIL_ALLOC")) return CHOLMOD_HIP_OUT_OF_MEMORY ;
    HIPCHK (hipMalloc ((void **) &P->d_Lx, std::max<i64> (P->lx_local, 1) * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &P->d_cb, std::max<i64> (P->arena, 1) * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &P->d_xchg, 3 * (size_t) P->world * sizeof (double))) ;
    {
        i64 mx = 1, mxg = 1 ;
        for (const Launch &L : P->sch.launches)
            if (L.kind == K_XCHG_RS || L.kind == K_XCHG_AG)
            {
                mx = std::max (mx, ((i64) L.xd.w * L.xd.w + (i64) L.xd.R * L.xd.w) * L.xd.g) ;
                mxg = std::max (mxg, (i64) L.xd.R * L.xd.w * L.xd.g) ;
            }
        HIPCHK (hipMalloc ((void **) &P->d_stage, (size_t) mx * sizeof (double))) ;
        HIPCHK (hipMalloc ((void **) &P->d_ag, (size_t) mxg * sizeof (double))) ;
        P->stage_len = mx ; P->ag_len = mxg ;
    }
    double tu3 = pnow () ;
    if (getenv ("CHOLMOD_HIP_THIN_TIMING"))
    {
        HIPCHK (hipMalloc ((void **) &P->d_thin_tim, (P->sch.launches.size () + 1) * 10 * sizeof (long long))) ;
        HIPCHK (hipMemset (P->d_thin_tim, 0, (P->sch.launches.size () + 1) * 10 * sizeof (long long))) ;
    }
    if (P->nsuper > 0)
    {
        int grid = (int) ((P->nsuper * 64 + 255) / 256) ;
        hipLaunchKernelGGL (k_relmap, dim3 (grid), dim3 (256), 0, P->stream,
            (int) P->nsuper, P->d_fr, P->d_Ls, P->d_relmap) ;
        if (!P->relpairs.empty ())
            hipLaunchKernelGGL (k_relmap_pairs, dim3 ((unsigned) ((P->relpairs.size () * 64 + 255) / 256)), dim3 (256), 0, P->stream,
                (int) P->relpairs.size (), P->d_relpairs, P->d_fr, P->d_Ls, P->d_relmap) ;
        HIPCHK (hipGetLastError ()) ;
        HIPCHK (hipStreamSynchronize (P->stream)) ;
    }
    if (ptiming) fprintf (stderr, "cholmod_hip upload_plan: streams/events %.3f s, maps + schedule H2D %.3f s, hipMalloc (L %.1f GB, arena %.1f GB) %.3f s, relmap kernel %.3f s\n",
        tu1 - tu0, tu2 - tu1, 8e-9 * P->lx_local, 8e-9 * P->arena, tu3 - tu2, pnow () - tu3) ;
    return CHOLMOD_HIP_OK ;
}

// k_trsm's dynamic LDS exceeds the 64 KB default limit for 64-wide panels
static int raise_lds_limits ()
{
    static bool done = false ;
    if (done) return CHOLMOD_HIP_OK ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_trsm_mfma<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_trsm_mfma<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_trsm_mfma<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_trsm_upd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_trsm_upd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_rowsolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int) rowsolve_lds_bytes ())) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_chainf, hipFuncAttributeMaxDynamicSharedMemorySize, (int) chainf_lds_bytes ())) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_thin_front<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_thin_front<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_thin_front<4, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    done = true ;
    return CHOLMOD_HIP_OK ;
}

// waves per SIMD the one-wave thin-front kernel is compiled for, by size class (<= 32 / 48 / 64
// rows); CHOLMOD_HIP_THIN_MINW = "a" or "a,b,c" overrides (3 .. 6)
static int thin_minw (int cls)
{
    static int v [3] = {4, 4, 3} ;      // (measured on the 2D 1259^2 problem: 6 everywhere 1.044 ms, 4 / 4 / 3 1.004 ms)
    static const bool init = [] ()
    {
        if (const char *e = getenv ("CHOLMOD_HIP_THIN_MINW"))
        {
            int a = 0, b = 0, c = 0 ;
            int k = sscanf (e, "%d,%d,%d", &a, &b, &c) ;
            if (k == 1) b = c = a ;
            if (k == 1 || k == 3) { int w [3] = {a, b, c} ; for (int q = 0 ; q < 3 ; q++) if (w [q] >= 3 && w [q] <= 6) v [q] = w [q] ; }
        }
        return true ;
    } () ;
    (void) init ;
    return v [cls] ;
}

// CHOLMOD_HIP_NARROW_EXCHANGE_KERNELS=1 (tuning): the kernels of the exchange stream (window open, extend-add into the window,
// pack) as ONE-wave workgroups.  Beside a trailing update whose one-wave tiles refill every register-file slot as it frees
// up, a four-wave workgroup needs room on all four SIMDs of a CU at once and starves until the update ends (rocprofv3:
// k_win_move stretched over the update's 78 ms); one-wave workgroups do get in (0.3 ms) -- but run at a fraction of the
// four-wave kernels' rate and cost a rank of 8 30 ms of compute (1.083 against 1.053 s), more than the early reduce-scatter of
// one block column in eight can return.  Measured, left off.
static bool narrow_xs ()
{
    static const bool v = [] () { const char *e = getenv ("CHOLMOD_HIP_NARROW_EXCHANGE_KERNELS") ; return e && atoi (e) != 0 ; } () ;
    return v ;
}

static int run_launch (cholmod_hip_plan *P, const Launch &L, bool serial)
{
    hipStream_t st = (serial || L.stream == 0 || !P->stream2) ? P->stream : P->stream2 ;
    const bool cx = (P->flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;      // a complex factor in its own storage (kernels.hip.h: ldcx / stcx)
    const bool twin = !cx && (P->flags & CHOLMOD_HIP_PHI_TWIN) != 0 ;      // update kernels contract over the even columns
// K<true> for a complex factor in its own storage, K<false> otherwise
#define CX_LAUNCH(K, ...) do { if (cx) hipLaunchKernelGGL (K<true>, __VA_ARGS__) ; else hipLaunchKernelGGL (K<false>, __VA_ARGS__) ; } while (0)
// the update kernels: 0 real, 1 twin (even-column contraction), 2 complex storage
#define TW_LAUNCH(KA, KB, ...) do { if (cx) hipLaunchKernelGGL ((KA 2 KB), __VA_ARGS__) ; else if (twin) hipLaunchKernelGGL ((KA 1 KB), __VA_ARGS__) ; \
                                    else hipLaunchKernelGGL ((KA 0 KB), __VA_ARGS__) ; } while (0)
    // (test hook CHOLMOD_HIP_TEST_DROP_WAITS=1: the cross-stream waits of the schedule are skipped -- the mutation the jitter
    // test must catch, tests/test_gpu_scale.py::test_stream_jitter_catches_a_dropped_wait)
    if (!serial && L.wait_ev >= 0 && L.kind != K_XCHG_RS && !P->test_drop_waits) HIPCHK (hipStreamWaitEvent (st, P->sync_ev [L.wait_ev], 0)) ;
    if (P->jitter_us > 0 && !serial)
    {
        // test hook CHOLMOD_HIP_TEST_JITTER=seed[:max_us]: ahead of one launch in three, its stream is held up for a random
        // time (mostly tens of microseconds, now and then max_us): a launch whose input comes from the OTHER stream without
        // an event between them then runs before its producer -- with the arena and the windows poisoned that shows in
        // the factor (tests/test_dist.py, tools/dist_soak.py).  The schedule itself is what ships.
        P->jitter_state = P->jitter_state * 6364136223846793005ull + 1442695040888963407ull ;
        const unsigned r = (unsigned) (P->jitter_state >> 33) ;
        if (r % 3 == 0)
        {
            const unsigned u = (r >> 8) % 100 ;
            const long long us = u < 90 ? 5 + (long long) ((r >> 16) % 60) : (long long) ((r >> 16) % (unsigned) P->jitter_us) ;
            hipLaunchKernelGGL (k_spin, dim3 (1), dim3 (64), 0, st, us * 100) ;
        }
    }
    switch (L.kind)
    {
        case K_JOIN: break ;
        case K_SMALL:
            { int rl = raise_lds_limits () ; if (rl != CHOLMOD_HIP_OK) return rl ; }
            // fronts of <= 64 rows run one wave per front (lane = row, no cross-wave
            // hand-off), wider ones four
            {
                // tuning: per-phase shader cycles of one front per launch (CHOLMOD_HIP_THIN_TIMING)
                long long *tim = P->d_thin_tim ? P->d_thin_tim + 10 * (size_t) (&L - P->sch.launches.data ()) : nullptr ;
#define THIN_LAUNCH(NW_, TIMED_, MINW_) \
                hipLaunchKernelGGL ((k_thin_front<NW_, TIMED_, MINW_>), dim3 (L.grid), dim3 (64 * NW_), thin_front_lds_bytes (L.aux), st, \
                    P->d_sm + L.goff, P->d_smd + L.goff, P->d_sp01 + 2 * L.goff, P->d_cdesc, P->d_relmap, P->d_Ls, P->d_Sp, \
                    P->s_unpacked ? P->d_Snz : nullptr, P->d_Si, P->d_Sx, P->cur_beta, \
                    P->d_Lx, P->d_cb, P->d_info, L.aux, P->d_amap, P->cur_mapped, tim)
                if (cx)
                {
                    // complex storage: one wave per front up to 64 twin rows, four above
#define THIN_LAUNCH_CX(NW_, MINW_) \
                    hipLaunchKernelGGL ((k_thin_front<NW_, false, MINW_, true>), dim3 (L.grid), dim3 (64 * NW_), thin_front_lds_bytes (L.aux), st, \
                        P->d_sm + L.goff, P->d_smd + L.goff, P->d_sp01 + 2 * L.goff, P->d_cdesc, P->d_relmap, P->d_Ls, P->d_Sp, \
                        P->s_unpacked ? P->d_Snz : nullptr, P->d_Si, P->d_Sx, P->cur_beta, \
                        P->d_Lx, P->d_cb, P->d_info, L.aux, P->d_amap, P->cur_mapped, (long long *) nullptr)
                    if (L.aux > 64) THIN_LAUNCH_CX (4, 2) ; else THIN_LAUNCH_CX (1, 4) ;
#undef THIN_LAUNCH_CX
                }
                else if (L.leaf_pw && P->cur_mapped && !P->s_unpacked && !tim)
                {
                    // leaf fronts two to a wave, once the assembly map of the resident S exists
#define LEAF_LAUNCH(PW_, MINW_) \
                    hipLaunchKernelGGL ((k_leaf_pair<PW_, MINW_>), dim3 ((L.grid + 1) / 2), dim3 (64), (size_t) (2 * L.leaf_T + 128) * sizeof (double), st, \
                        P->d_sm + L.goff, L.grid, P->d_smd + L.goff, P->d_sp01 + 2 * L.goff, P->d_Sx, P->d_amap, P->cur_beta, P->d_Lx, P->d_cb, P->d_info, L.leaf_T)
                    static const int lw = [] () { const char *e = getenv ("CHOLMOD_HIP_LEAF_MINW") ; int w = e ? atoi (e) : 4 ; return (w == 2 || w == 3 || w == 5 || w == 6) ? w : 4 ; } () ;
#define LEAF_PW(MINW_) \
                    { if (L.leaf_pw <= 4) LEAF_LAUNCH (4, MINW_) ; else if (L.leaf_pw <= 8) LEAF_LAUNCH (8, MINW_) ; \
                      else if (L.leaf_pw <= 12) LEAF_LAUNCH (12, MINW_) ; else LEAF_LAUNCH (16, MINW_) ; }
                    // (measured on the 2D 1259^2 leaf level: 4 waves per SIMD 0.112 ms, 5: 0.191, 6: 0.245)
                    if (lw == 2) LEAF_PW (2) else if (lw == 3) LEAF_PW (3) else if (lw == 5) LEAF_PW (5) else if (lw == 6) LEAF_PW (6) else LEAF_PW (4)
#undef LEAF_PW
#undef LEAF_LAUNCH
                }
                else if (tim) { if (L.aux <= 64) THIN_LAUNCH (1, true, 4) ; else THIN_LAUNCH (4, true, 2) ; }
                else if (L.aux > 64) THIN_LAUNCH (4, false, 2) ;
                else
                {
                    // one wave per front: the register budget (waves per SIMD the compiler must
                    // allow) by size class -- the LDS of the wider classes caps the occupancy
                    // anyway (9.6 / 16.9 KB per front), and 80 registers spill
                    const int w = thin_minw (L.aux <= 32 ? 0 : L.aux <= 48 ? 1 : 2) ;
                    if (w == 6) THIN_LAUNCH (1, false, 6) ;
                    else if (w == 5) THIN_LAUNCH (1, false, 5) ;
                    else if (w == 3) THIN_LAUNCH (1, false, 3) ;
                    else THIN_LAUNCH (1, false, 4) ;
                }
#undef THIN_LAUNCH
            }
            break ;
        case K_XCHG_RS:
        case K_XCHG_AG:
            {
                // Native exchange (cholmod_hip_rccl_attach): everything stream-ordered, the host
                // never waits.  A reduce-scatter ahead of time (wait_ev >= 0) runs on the second
                // stream behind the event of the update that completed the block column, while
                // the main stream goes on with the rest of the trailing update; the main stream
                // then waits (on the device) for the sum before it touches the block column.
                // Host callback (gloo tests, --exchange callback): it only knows a sum
                // all-reduce, so the reduce-scatter is an all-reduce of all segments and the
                // all-gather a sum of buffers that are zero outside the sender's chunk.
                const bool rs = L.kind == K_XCHG_RS ;
                const XchgD &X = L.xd ;
                RcclApi *R = P->nccl_world ? rccl_api () : nullptr ;
                if (!R && !P->ar_fn) return CHOLMOD_HIP_INVALID ;
                bool ahead = rs && !serial && L.wait_ev >= 0 && P->stream2 ;
                hipStream_t cs = ahead ? P->stream2 : st ;
                if (ahead)
                {
                    if (R) HIPCHK (hipStreamWaitEvent (cs, P->sync_ev [L.wait_ev], 0)) ;
                    else HIPCHK (hipEventSynchronize (P->sync_ev [L.wait_ev])) ;
                }
                ncclComm_t comm = P->nccl_world ;
                if (R && L.ar_gn != P->world)
                {
                    auto it = P->nccl_group.find (((i64) L.ar_g0 << 16) | (i64) L.ar_gn) ;
                    if (it == P->nccl_group.end ()) return CHOLMOD_HIP_INVALID ;
                    comm = it->second ;
                }
                const i64 seg = (i64) X.w * X.w + (i64) X.R * X.w, chunk = (i64) X.R * X.w ;
                const long long xseq = ++P->prog_xchg_enq ;
                if (P->prog_dev) hipLaunchKernelGGL (k_mark, dim3 (1), dim3 (1), 0, cs, P->prog_dev, (P->prog_fact << 32) | xseq) ;
#ifdef CHOLMOD_HIP_TEST_HOOKS
                // test hook CHOLMOD_HIP_TEST_HANG_EXCHANGE=rank:seq: that rank never issues exchange `seq` of its second
                // factorization (its host thread sleeps here; nothing hangs on the device) -- its peers then wait for it in
                // the collective, which is what bench.py's watchdog must turn into an error line (tests/test_bench_contract.py)
                if (P->rank == P->test_hang_rank && xseq == P->test_hang_xchg && P->prog_fact >= 2)
                {
                    (void) hipStreamSynchronize (cs) ;
                    for ( ; ; ) sleep (3600) ;
                }
#endif
                auto move = [&] (int mode, i64 total)
                {
                    if (total <= 0) return ;
                    // (one workgroup per column and part: k_xchg_move)
                    const unsigned parts = mode == 0 ? (unsigned) X.g + 1 : mode == 3 ? (unsigned) X.g : mode == 1 ? 2u : 1u ;
                    hipLaunchKernelGGL (k_xchg_move, dim3 ((unsigned) X.w * parts), dim3 (ahead && narrow_xs () ? 64 : 256), 0, cs, X, mode, P->d_Lx, P->d_stage, P->d_ag) ;
                } ;
                if (rs)
                {
                    move (0, seg * X.g) ;
                    if (R)
                    {
                        RCCLCHK (R->ReduceScatter (P->d_stage, P->d_stage + (i64) X.r * seg, (size_t) seg, ncclDouble, ncclSum, comm, cs)) ;
                        // The w x w diagonal block travels in every segment, and a ring sums every segment in another
                        // order: the members' copies of it would differ in their last bits, each would factor its own,
                        // and a borderline pivot could fail on one member only.  One copy for all: the first member's
                        // (2 MB at w = 512, next to the block column's 8 (w + rows) w bytes).
                        if (X.g > 1) RCCLCHK (R->Broadcast (P->d_stage, P->d_stage + (i64) X.r * seg, (size_t) X.w * X.w, ncclDouble, 0, comm, cs)) ;
                    }
                    else
                    {
                        HIPCHK (hipStreamSynchronize (cs)) ;
                        if (P->ar_fn (P->d_stage, seg * X.g, L.ar_g0, L.ar_gn, P->ar_user) != 0) return CHOLMOD_HIP_GPU_PROBLEM ;
                    }
                    move (1, seg) ;
                }
                else
                {
                    if (!R) HIPCHK (hipMemsetAsync (P->d_ag, 0, (size_t) (chunk * X.g) * sizeof (double), cs)) ;
                    move (2, chunk) ;
                    if (R) RCCLCHK (R->AllGather (P->d_ag + (i64) X.r * chunk, P->d_ag, (size_t) chunk, ncclDouble, comm, cs)) ;
                    else
                    {
                        HIPCHK (hipStreamSynchronize (cs)) ;
                        if (P->ar_fn (P->d_ag, chunk * X.g, L.ar_g0, L.ar_gn, P->ar_user) != 0) return CHOLMOD_HIP_GPU_PROBLEM ;
                    }
                    move (3, chunk * X.g) ;
                }
                if (P->prog_dev) hipLaunchKernelGGL (k_mark, dim3 (1), dim3 (1), 0, cs, P->prog_dev + 1, (P->prog_fact << 32) | xseq) ;
                if (ahead)
                {
                    if (R)
                    {
                        HIPCHK (hipEventRecord (P->ar_done, cs)) ;
                        HIPCHK (hipStreamWaitEvent (st, P->ar_done, 0)) ;
                    }
                    else HIPCHK (hipStreamSynchronize (cs)) ;
                }
            }
            break ;
        case K_CHAINF:
            { int rl = raise_lds_limits () ; if (rl != CHOLMOD_HIP_OK) return rl ; }
            hipLaunchKernelGGL (k_chainf, dim3 (L.grid), dim3 (256), chainf_lds_bytes (), st,
                P->d_cg + L.goff, L.ng, L.ndiag, P->d_Lx, P->d_info, P->d_dinv, P->d_cflags, P->d_cflags + 4 * (size_t) P->sch.ncflags) ;
            break ;
        case K_WIN:
            // (one-wave workgroups on the exchange stream: see k_extend_add)
            hipLaunchKernelGGL (k_win_move, dim3 (L.grid), dim3 (L.stream == 1 && !serial && narrow_xs () ? 64 : 256), 0, st, P->d_wg + L.goff, L.ng, P->d_Lx) ; break ;
        case K_ZERO:
            hipLaunchKernelGGL (k_zero, dim3 (L.grid), dim3 (256), 0, st,
                P->d_zg + L.goff, L.ng, P->d_cb) ; break ;
        case K_EA:
            CX_LAUNCH (k_extend_add, dim3 (L.grid), dim3 (L.stream == 1 && !serial && narrow_xs () ? 64 : 256), 0, st,
                P->d_eg + L.goff, L.ng, P->d_fr, P->d_child, P->d_crel, P->d_relmap, P->d_Lx, P->d_cb, L.aux > 0 ? L.aux : EA_TW) ; break ;
        case K_POTRF:
            if (cx) hipLaunchKernelGGL ((k_potrf_mfma<false, true>), dim3 (L.grid), dim3 (256), 0, st,
                P->d_pg + L.goff, P->d_Lx, P->d_info, (long long *) nullptr) ;
            else hipLaunchKernelGGL ((k_potrf_mfma<false, false>), dim3 (L.grid), dim3 (256), 0, st,
                P->d_pg + L.goff, P->d_Lx, P->d_info, (long long *) nullptr) ;
            break ;
        case K_TRSM:
            { int rl = raise_lds_limits () ; if (rl != CHOLMOD_HIP_OK) return rl ; }
            if (cx) hipLaunchKernelGGL ((k_trsm_mfma<false, true>), dim3 (L.grid), dim3 (256), trsm_mfma_lds_bytes (L.aux), st,
                P->d_tg + L.goff, L.ng, P->d_Lx, P->d_info, L.aux, (long long *) nullptr) ;
            else hipLaunchKernelGGL ((k_trsm_mfma<false, false>), dim3 (L.grid), dim3 (256), trsm_mfma_lds_bytes (L.aux), st,
                P->d_tg + L.goff, L.ng, P->d_Lx, P->d_info, L.aux, (long long *) nullptr) ;
            break ;
        case K_TRSM_UPD:
            { int rl = raise_lds_limits () ; if (rl != CHOLMOD_HIP_OK) return rl ; }
            CX_LAUNCH (k_trsm_upd, dim3 (L.grid), dim3 (256), trsm_upd_lds_bytes (), st,
                P->d_tg + L.goff, L.ng, P->d_Lx, P->d_info, P->d_tu_cnt + L.goff) ;
            break ;
        case K_UPD_BIG:
            hipLaunchKernelGGL ((k_update2<BIG, BIG, BKK, 2, false>), dim3 (L.grid), dim3 (256), 0, st,
                P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
            break ;
        case K_UPD_SMALL:
            TW_LAUNCH (k_update2<SMALL COMMA SMALL COMMA BKK COMMA 2 COMMA false COMMA, >, dim3 (L.grid), dim3 (256), 0, st,
                P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
            break ;
        case K_DIAG:
            hipLaunchKernelGGL (k_diag<false>, dim3 (L.grid), dim3 (256), 0, st, P->d_dg + L.goff, P->d_Lx, P->d_info, P->d_dinv, (long long *) nullptr) ;
            break ;
        case K_ROWSOLVE:
            { int rl = raise_lds_limits () ; if (rl != CHOLMOD_HIP_OK) return rl ; }
            hipLaunchKernelGGL (k_rowsolve, dim3 (L.grid), dim3 (256), rowsolve_lds_bytes (), st,
                P->d_rg + L.goff, L.ng, P->d_Lx, P->d_info, P->d_dinv) ;
            break ;
        case K_UPD_W:
            // operand sets in flight: four for long contractions, two for short ones (tools/upd3.py)
            if (L.pcnt >= 0 && !serial)
            {
                // persistent form beside a panel chain: 2 workgroups per CU minus the reserve (CHOLMOD_HIP_LA_RESERVE, default 64)
                int slots = 2 * P->ncu - P->la_reserve ;
                if (slots < 8) slots = 8 ;
                unsigned gp = (unsigned) std::min<long> (((long) L.grid + 3) / 4, (long) slots) ;
                const int rsv = P->la_reserve_cu ;
                if (rsv > 0) gp = (unsigned) (4 * P->ncu) ;         // (2 per CU stay, the others are burnt on the reserved CUs)
                int *cnt = P->d_pcnt + 8 * (size_t) L.pcnt ;
                if (L.aux >= 1024) TW_LAUNCH (k_update3p<4 COMMA, >, dim3 (gp), dim3 (256), 0, st, P->d_gg + L.goff, L.ng, L.grid, cnt, P->d_Lx, P->d_cb, rsv) ;
                else TW_LAUNCH (k_update3p<2 COMMA, >, dim3 (gp), dim3 (256), 0, st, P->d_gg + L.goff, L.ng, L.grid, cnt, P->d_Lx, P->d_cb, rsv) ;
            }
            else if (P->upd3_wg4)
            {
                // (several ranks) four tiles per workgroup: see k_update3
                const unsigned g4 = (unsigned) (((L.grid + 31) / 32) * 8) ;
                if (L.aux >= 1024) TW_LAUNCH (k_update3<4 COMMA, COMMA 4>, dim3 (g4), dim3 (256), 0, st, P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
                else TW_LAUNCH (k_update3<2 COMMA, COMMA 4>, dim3 (g4), dim3 (256), 0, st, P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
            }
            else if (L.aux >= 1024) TW_LAUNCH (k_update3<4 COMMA, >, dim3 (L.grid), dim3 (64), 0, st, P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
            else TW_LAUNCH (k_update3<2 COMMA, >, dim3 (L.grid), dim3 (64), 0, st, P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
            break ;
        case K_UPD_PF:
            TW_LAUNCH (k_update2f<, >, dim3 (L.grid), dim3 (256), 0, st,
                P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb, P->d_info) ;
            break ;
    }
    if (!serial && L.rec_ev >= 0) HIPCHK (hipEventRecord (P->sync_ev [L.rec_ev], st)) ;
    return CHOLMOD_HIP_OK ;
}

// Run the numeric factorization on the resident S.  Leaves Lx on the device.
static int run_factorize (cholmod_hip_plan *P, double beta, int quick, i64 *minor)
{
    hipStream_t st = P->stream ;
    if (!P->d_Sp) return CHOLMOD_HIP_INVALID ;
    bool prof = P->profiling ;
    static const bool host_timing = getenv ("CHOLMOD_HIP_HOST_TIMING") != nullptr ;
    auto now = [] () { return std::chrono::duration<double> (std::chrono::steady_clock::now ().time_since_epoch ()).count () ; } ;
    double th0 = now () ;
    P->cur_beta = beta ;
    P->cur_mapped = P->amap_valid ? 1 : 0 ;     // thin fronts: A through the map the first assembly recorded
    size_t nl = P->sch.launches.size () ;
    if (prof)
    {
        while (P->evpool.size () < 2 * (nl + 1))
        {
            hipEvent_t e ; HIPCHK (hipEventCreate (&e)) ; P->evpool.push_back (e) ;
        }
    }
    P->winv_valid = false ;                 // the diagonal-block inverses follow the factor
    HIPCHK (hipEventRecord (P->ev0, st)) ;
    int poisoned = CHOLMOD_HIP_OK ;
    bool building_map = false ;
    if (prof) HIPCHK (hipEventRecord (P->evpool [0], st)) ;
    // Lx := 0 (several ranks: the rank's own fronts -- its d_Lx holds nothing else); the complete
    // factor a gather may have left in d_Lx_full is stale from here on
    P->full_valid = false ;
    if (!P->d_Lx)
    {
        // (the gather went through the host and released the rank's own array: cholmod_hip_gather_factor)
        if (P->d_Lx_full) { (void) hipFree (P->d_Lx_full) ; P->d_Lx_full = nullptr ; }
        HIPCHK (hipMalloc ((void **) &P->d_Lx, std::max<i64> (P->lx_local, 1) * sizeof (double))) ;
    }
    if (!P->d_cb)
    {
        // (the arena made room for the gathered factor, see cholmod_hip_gather_factor)
        if (hipMalloc ((void **) &P->d_cb, std::max<i64> (P->arena, 1) * sizeof (double)) != hipSuccess)
        {
            (void) hipGetLastError () ;
            if (P->d_Lx_full) { (void) hipFree (P->d_Lx_full) ; P->d_Lx_full = nullptr ; }
            HIPCHK (hipMalloc ((void **) &P->d_cb, std::max<i64> (P->arena, 1) * sizeof (double))) ;
        }
    }
    // test hook CHOLMOD_HIP_TEST_POISON_ARENA=1: the contribution-block arena (and the exchange staging) start every
    // factorization as NaNs -- an entry somebody reads before anybody has written it then shows in the factor,
    // whatever a fresh allocation happens to hold (tests/test_gpu_parity.py, tests/test_dist.py)
    const bool poison = TEST_ENV ("CHOLMOD_HIP_TEST_POISON_ARENA") != nullptr ;
    if (poison)
    {
        HIPCHK (hipMemsetAsync (P->d_cb, 0xFF, std::max<i64> (P->arena, 1) * sizeof (double), st)) ;
        if (P->d_stage) HIPCHK (hipMemsetAsync (P->d_stage, 0xFF, (size_t) P->stage_len * sizeof (double), st)) ;
        if (P->lx_local > P->lx_fronts) HIPCHK (hipMemsetAsync (P->d_Lx + P->lx_fronts, 0xFF, (size_t) (P->lx_local - P->lx_fronts) * sizeof (double), st)) ;
        if (P->d_ag) HIPCHK (hipMemsetAsync (P->d_ag, 0xFF, (size_t) P->ag_len * sizeof (double), st)) ;
    }
    HIPCHK (hipMemsetAsync (P->d_Lx, 0, std::max<i64> (poison ? P->lx_fronts : P->lx_local, 1) * sizeof (double), st)) ;
    HIPCHK (hipMemsetAsync (P->d_info, 0, std::max<i64> (P->nsuper, 1) * sizeof (i32), st)) ;
    HIPCHK (hipMemsetAsync (P->d_tu_cnt, 0, std::max<size_t> (P->sch.tg.size (), 1) * sizeof (i32), st)) ;
    if (P->d_pcnt) HIPCHK (hipMemsetAsync (P->d_pcnt, 0, 8 * (size_t) P->sch.npcnt * sizeof (int), st)) ;
    if (P->d_cflags) HIPCHK (hipMemsetAsync (P->d_cflags, 0, (4 * (size_t) P->sch.ncflags + 4) * sizeof (int), st)) ;
    if (P->n > 0 && P->amap_valid)
    {
        // the resident S was assembled before: stream it through its map
        if (P->s_nz > 0)
            hipLaunchKernelGGL (k_assemble_mapped, dim3 ((unsigned) ((P->s_nz + 255) / 256)), dim3 (256), 0, st,
                P->s_nz, P->d_amap, P->d_Sx, P->d_Lx) ;
        if (beta != 0.0)
        {
            if (P->flags & CHOLMOD_HIP_CX_STORAGE) hipLaunchKernelGGL (k_add_beta<true>, dim3 ((unsigned) ((P->n + 255) / 256)), dim3 (256), 0, st,
                P->n, P->d_supermap, P->d_fr, P->d_Lx, beta) ;
            else hipLaunchKernelGGL (k_add_beta<false>, dim3 ((unsigned) ((P->n + 255) / 256)), dim3 (256), 0, st,
                P->n, P->d_supermap, P->d_fr, P->d_Lx, beta) ;
        }
    }
    else if (P->n > 0)
    {
        HIPCHK (hipMemsetAsync (P->d_amap, 0xFF, std::max<i64> (P->s_nz, 1) * sizeof (i64), st)) ;     // -1: not in L
        if (P->flags & CHOLMOD_HIP_CX_STORAGE) hipLaunchKernelGGL (k_assemble<true>, dim3 ((unsigned) ((P->n + 255) / 256)), dim3 (256), 0, st,
            P->n, P->d_Sp, P->s_unpacked ? P->d_Snz : nullptr, P->d_Si, P->d_Sx,
            P->d_supermap, P->d_fr, P->d_Ls, P->d_Lx, beta, P->d_amap) ;
        else hipLaunchKernelGGL (k_assemble<false>, dim3 ((unsigned) ((P->n + 255) / 256)), dim3 (256), 0, st,
            P->n, P->d_Sp, P->s_unpacked ? P->d_Snz : nullptr, P->d_Si, P->d_Sx,
            P->d_supermap, P->d_fr, P->d_Ls, P->d_Lx, beta, P->d_amap) ;
        building_map = true ;       // (the thin-front kernels record their part; valid once every launch has run)
    }
    if (prof) HIPCHK (hipEventRecord (P->evpool [1], st)) ;
    int fail_rank = -1 ; long fail_launch = -1 ;
    if (const char *e = TEST_ENV ("CHOLMOD_HIP_TEST_FAIL_LAUNCH")) (void) sscanf (e, "%d:%ld", &fail_rank, &fail_launch) ;
    P->prog_fact = P->prog_fact + 1 ; P->prog_launch = 0 ; P->prog_xchg_enq = 0 ;
    for (size_t q = 0 ; q < nl ; q++)
    {
        const Launch &L = P->sch.launches [q] ;
        P->prog_launch = (long long) q + 1 ;
        if (prof && poisoned == CHOLMOD_HIP_OK) HIPCHK (hipEventRecord (P->evpool [2 * (q + 1)], st)) ;
        if (poisoned != CHOLMOD_HIP_OK)
        {
            // A launch of this rank failed.  The other ranks of its groups are about to
            // block in the collectives that follow: keep taking part in them (the data no
            // longer matters) and report the failure through the agreement exchange at
            // the end, so that every rank returns an error instead of hanging.
            if (L.kind == K_XCHG_RS || L.kind == K_XCHG_AG) (void) run_launch (P, L, true) ;
            continue ;
        }
        int rl = run_launch (P, L, prof) ;
        // test hook "rank:launch": that launch of that rank reports a failure
        if (fail_rank == P->rank && fail_launch == (long) q) rl = CHOLMOD_HIP_GPU_PROBLEM ;
        if (rl != CHOLMOD_HIP_OK)
        {
            if (P->world == 1) return rl ;
            poisoned = rl ;
            continue ;
        }
        if (prof) HIPCHK (hipEventRecord (P->evpool [2 * (q + 1) + 1], st)) ;
    }
    if (poisoned == CHOLMOD_HIP_OK && hipGetLastError () != hipSuccess) poisoned = CHOLMOD_HIP_GPU_PROBLEM ;
    if (poisoned != CHOLMOD_HIP_OK && P->world == 1) return poisoned ;
    (void) hipEventRecord (P->ev1, st) ;
    double th1 = now () ;
    // not-positive-definite protocol (t_cholmod_super_numeric.c:905-968): the first
    // failing supernode and its info, reduced on the device
    float ms = 0 ;
    i64 sbad = -1, binfo = 0 ;
    if (poisoned == CHOLMOD_HIP_OK)
    {
        int first = (int) P->nsuper ;
        i32 inf = 0 ;
        bool ok = hipMemcpyAsync (P->d_first_fail, &first, sizeof (int), hipMemcpyHostToDevice, st) == hipSuccess ;
        if (ok && P->nsuper > 0)
            hipLaunchKernelGGL (k_first_fail, dim3 ((unsigned) ((P->nsuper + 255) / 256)), dim3 (256), 0, st,
                P->nsuper, P->d_info, P->d_first_fail) ;
        // (k_chainf: a workgroup that waited longer than its budget for another one's flag says so here)
        int chain_err = 0 ;
        if (P->sch.ncflags > 0)
            ok = ok && hipMemcpyAsync (&chain_err, P->d_cflags + 4 * (size_t) P->sch.ncflags, sizeof (int), hipMemcpyDeviceToHost, st) == hipSuccess ;
        ok = ok && hipMemcpyAsync (&first, P->d_first_fail, sizeof (int), hipMemcpyDeviceToHost, st) == hipSuccess
            && hipStreamSynchronize (st) == hipSuccess
            && hipEventElapsedTime (&ms, P->ev0, P->ev1) == hipSuccess ;
        if (ok && first < (int) P->nsuper)
        {
            ok = hipMemcpy (&inf, P->d_info + first, sizeof (i32), hipMemcpyDeviceToHost) == hipSuccess ;
            sbad = first ; binfo = inf ;
        }
        if (chain_err != 0)
        {
            fprintf (stderr, "cholmod_hip: k_chainf: a workgroup waited for a flag beyond its budget (hand-off between workgroups failed)\n") ;
            ok = false ;
        }
        if (!ok)
        {
            if (P->world == 1) return CHOLMOD_HIP_GPU_PROBLEM ;
            poisoned = CHOLMOD_HIP_GPU_PROBLEM ;
        }
        else if (building_map) P->amap_valid = true ;
    }
    double th2 = now () ;
    if (host_timing)
        fprintf (stderr, "cholmod_hip: host enqueue %.3f ms, wait+info copy %.3f ms, device %.3f ms, %zu launches\n",
            1e3 * (th1 - th0), 1e3 * (th2 - th1), (double) ms, nl) ;
    double *S = P->stats ;
    for (int q = 0 ; q < CHOLMOD_HIP_NSTATS ; q++) S [q] = 0 ;
    S [0] = ms * 1e-3 ;
    S [1] = P->exec_flops ;
    S [2] = (double) nl + 3 ;
    S [3] = P->nlevels ;
    S [4] = 8.0 * P->arena ;
    S [5] = 8.0 * P->xsize ;
    S [36] = 8.0 * P->lx_local ;
    for (size_t q = 0 ; q < nl ; q++)
    {
        const Launch &L = P->sch.launches [q] ;
        if (L.kind == K_UPD_SMALL) { S [7] += 1 ; S [8] += L.flops ; S [16] += L.bytes ; }
        if (L.kind == K_UPD_W) { S [33] += 1 ; S [34] += L.flops ; S [35] += L.bytes ; }
        if (L.kind == K_UPD_PF) { S [26] += 1 ; S [28] += L.flops ; S [29] += L.bytes ; }
        if (L.kind == K_TRSM_UPD) S [31] += 1 ;
        if (L.kind == K_XCHG_RS || L.kind == K_XCHG_AG) { S [17] += 1 ; S [18] += L.bytes ; }
        if (L.kind == K_SMALL) { S [20] += L.bytes ; S [21] += L.ng ; }
        S [22] = P->nsplit ;
        if (L.kind == K_UPD_BIG) { S [15] += L.flops ; }
        if (L.kind == K_EA) S [10] += L.bytes ;
        if (L.kind == K_WIN)
            for (int w = 0 ; w < L.ng ; w++)
            {
                const WinD &Wd = P->sch.wg [L.goff + w] ;
                if (Wd.mode == 0) { S [37] += 1 ; if (Wd.win < 0) S [38] += 1 ; }
            }
    }
    if (prof && poisoned == CHOLMOD_HIP_OK)
    {
        float t = 0 ;
        HIPCHK (hipEventElapsedTime (&t, P->evpool [0], P->evpool [1])) ;
        S [13] = t * 1e-3 ;
        P->launch_ms.assign (nl, 0.0f) ;
        for (size_t q = 0 ; q < nl ; q++)
        {
            const Launch &L = P->sch.launches [q] ;
            HIPCHK (hipEventElapsedTime (&t, P->evpool [2 * (q + 1)], P->evpool [2 * (q + 1) + 1])) ;
            P->launch_ms [q] = t ;
            double sec = t * 1e-3 ;
            switch (L.kind)
            {
                case K_UPD_SMALL: S [6] += sec ; if (L.aux < MB) { S [23] += sec ; } break ;
                case K_UPD_PF: S [27] += sec ; break ;
                case K_UPD_W: S [32] += sec ; break ;
                case K_UPD_BIG: S [14] += sec ; break ;
                case K_EA: case K_ZERO: case K_WIN: S [9] += sec ; break ;
                case K_POTRF: case K_DIAG: case K_CHAINF: S [11] += sec ; break ;
                case K_ROWSOLVE: S [12] += sec ; break ;
                case K_SMALL: S [19] += sec ; break ;
                case K_TRSM: S [12] += sec ; break ;
                case K_TRSM_UPD: S [30] += sec ; break ;
            }
        }
    }
    if (P->world > 1 || P->force_shared)
    {
        // agree on the first failing supernode: every rank publishes its own
        // candidate in its slot of a small device array, the sum-all-reduce
        // makes all slots visible everywhere
        if (!P->ar_fn && !P->nccl_world) return CHOLMOD_HIP_INVALID ;
        std::vector<double> x (3 * (size_t) P->world, 0.0) ;
        x [P->rank] = (double) (sbad >= 0 ? sbad : P->nsuper) ;
        x [P->world + P->rank] = (double) binfo ;
        x [2 * P->world + P->rank] = (poisoned != CHOLMOD_HIP_OK) ? 1.0 : 0.0 ;     // a launch of this rank failed
        HIPCHK (hipMemcpy (P->d_xchg, x.data (), x.size () * sizeof (double), hipMemcpyHostToDevice)) ;
        if (P->nccl_world)
        {
            RCCLCHK (rccl_api ()->AllReduce (P->d_xchg, P->d_xchg, x.size (), ncclDouble, ncclSum, P->nccl_world, st)) ;
            HIPCHK (hipStreamSynchronize (st)) ;
        }
        else if (P->ar_fn (P->d_xchg, (i64) x.size (), 0, P->world, P->ar_user) != 0) return CHOLMOD_HIP_GPU_PROBLEM ;
        HIPCHK (hipMemcpy (x.data (), P->d_xchg, x.size () * sizeof (double), hipMemcpyDeviceToHost)) ;
        for (int r = 0 ; r < P->world ; r++)
            if (x [2 * P->world + r] != 0.0) return poisoned != CHOLMOD_HIP_OK ? poisoned : CHOLMOD_HIP_GPU_PROBLEM ;
        i64 best = P->nsuper ;
        for (int r = 0 ; r < P->world ; r++)
            if ((i64) x [r] < best) { best = (i64) x [r] ; binfo = (i64) x [P->world + r] ; }
        sbad = best < P->nsuper ? best : -1 ;
    }
    *minor = P->n ;
    if (sbad < 0) return CHOLMOD_HIP_OK ;
    const FrontD &f = P->fr [sbad] ;
    *minor = f.k1 + binfo - 1 ;
    // everything from supernode sbad + 1 on (from sbad itself when it has no valid column, or on
    // a quick return) is zero: in the rank's own array, the fronts it holds with those indices
    // (held fronts are packed in supernode order, so that is one tail of the array)
    i64 first_zero = sbad + ((binfo == 1 || quick) ? 0 : 1) ;
    i64 zero_from = P->lx_fronts ;
    for (i64 q = first_zero ; q < P->nsuper ; q++) if (P->lpx [q] >= 0) { zero_from = P->lpx [q] ; break ; }
    if (zero_from < P->lx_fronts)
        HIPCHK (hipMemsetAsync (P->d_Lx + zero_from, 0, (P->lx_fronts - zero_from) * sizeof (double), st)) ;
    HIPCHK (hipStreamSynchronize (st)) ;
    return CHOLMOD_HIP_NOT_POSDEF ;
}

} // namespace

// The complete numeric factor in the reference layout and the descriptors that go with it: the
// rank's own array when there is one rank, the gathered copy otherwise (nullptr before a gather).
static double *whole_factor (cholmod_hip_plan *P) { return P->world == 1 ? P->d_Lx : (P->full_valid ? P->d_Lx_full : nullptr) ; }
static const FrontD *whole_fronts (cholmod_hip_plan *P) { return P->world == 1 ? P->d_fr : P->d_fr_full ; }

// ============================================================================
// extern "C" shim
// ============================================================================

extern "C" {

const char *cholmod_hip_version (void) { return "suitesparse_amd cholmod_hip 0.1 (gfx950)" ; }

int cholmod_hip_probe (void)
{
    int cnt = 0 ;
    if (hipGetDeviceCount (&cnt) != hipSuccess) return 0 ;
    return cnt > 0 ? 1 : 0 ;
}

int cholmod_hip_device_count (int *count)
{
    if (!count) return CHOLMOD_HIP_INVALID ;
    *count = 0 ;
    int cnt = 0 ;
    if (hipGetDeviceCount (&cnt) != hipSuccess) { (void) hipGetLastError () ; return CHOLMOD_HIP_NO_DEVICE ; }
    *count = cnt ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_memorysize (size_t *total_mem, size_t *available_mem)
{
    if (total_mem) *total_mem = 0 ;
    if (available_mem) *available_mem = 0 ;
    if (!cholmod_hip_probe ()) return 1 ;
    size_t f = 0, t = 0 ;
    if (hipMemGetInfo (&f, &t) != hipSuccess) return 1 ;
    if (total_mem) *total_mem = t ;
    if (available_mem) *available_mem = f ;
    return 0 ;
}

int cholmod_hip_set_device (int device)
{
    return hipSetDevice (device) == hipSuccess ? CHOLMOD_HIP_OK : CHOLMOD_HIP_NO_DEVICE ;
}

cholmod_hip_plan *cholmod_hip_plan_create_dist (int64_t n, int64_t nsuper,
    const int64_t *super, const int64_t *pi, const int64_t *px, const int64_t *s,
    int flags, int rank, int world, int *status)
{
    int st_local ;
    if (!status) status = &st_local ;
    *status = CHOLMOD_HIP_OK ;
    if (n < 0 || nsuper < 0 || !super || !pi || !px || !s || world < 1 || rank < 0 || rank >= world)
    { *status = CHOLMOD_HIP_INVALID ; return nullptr ; }
    // front descriptors, relative maps and the supernode map are 32-bit on the device
    if (n > INT32_MAX || nsuper > INT32_MAX) { *status = CHOLMOD_HIP_TOO_LARGE ; return nullptr ; }
    bool host_only = (flags & CHOLMOD_HIP_PLAN_HOST_ONLY) != 0 ;
    if (!host_only && !cholmod_hip_probe ()) { *status = CHOLMOD_HIP_NO_DEVICE ; return nullptr ; }
    const double tpc = std::chrono::duration<double> (std::chrono::steady_clock::now ().time_since_epoch ()).count () ;
    if (flags & CHOLMOD_HIP_CX_STORAGE)
    {
        // complex storage: the twin's index space, the generic kernels only (the LDS-resident thin-front
        // kernels and the 256-column chain have no complex-storage form), one rank
        if (world > 1) { if (status) *status = CHOLMOD_HIP_INVALID ; return nullptr ; }
        // (round 4: the thin-front kernel has a complex-storage form, k_thin_front<..., CX>; CHOLMOD_HIP_CX_NO_THIN=1: the
        // generic kernels for every front, as before)
        flags |= CHOLMOD_HIP_PHI_TWIN ;
        if (getenv ("CHOLMOD_HIP_CX_NO_THIN")) flags |= CHOLMOD_HIP_NO_SMALL_FRONTS ;
        flags &= ~(CHOLMOD_HIP_CHAIN256 | CHOLMOD_HIP_TILE128) ;
    }
    if (flags & CHOLMOD_HIP_PHI_TWIN)
    {
        // a twin has every supernode boundary and row in (2i, 2i+1) pairs
        bool ok = (n % 2 == 0) ;
        for (i64 q = 0 ; ok && q <= nsuper ; q++) ok = (super [q] % 2 == 0) && (pi [q] % 2 == 0) ;
        for (i64 q = 0 ; ok && q + 1 < pi [nsuper] ; q += 2) ok = (s [q] % 2 == 0) && (s [q + 1] == s [q] + 1) ;
        if (!ok) { if (status) *status = CHOLMOD_HIP_INVALID ; return nullptr ; }
    }
    // (the plan is allocated only once the flags and the twin structure are known to be valid:
    // nothing to release on the early returns above)
    cholmod_hip_plan *P = new (std::nothrow) cholmod_hip_plan ;
    if (!P) { *status = CHOLMOD_HIP_OUT_OF_MEMORY ; return nullptr ; }
    P->n = n ; P->nsuper = nsuper ; P->flags = flags ; P->host_only = host_only ;
    P->rank = rank ; P->world = world ;
    P->super.assign (super, super + nsuper + 1) ;
    P->pi.assign (pi, pi + nsuper + 1) ;
    P->px.assign (px, px + nsuper + 1) ;
    P->ssize = pi [nsuper] ; P->xsize = px [nsuper] ;
    P->Ls.assign (s, s + std::max<i64> (P->ssize, 1)) ;
    // memory budget of the contribution-block arena (see build_host): what is left
    // of the HBM next to L and the index maps.  Several ranks must derive the same
    // schedule, so they use a nominal capacity instead of their momentary free
    // memory.  CHOLMOD_HIP_ARENA_BUDGET_MB overrides (tests, tuning).
    {
        const char *e = getenv ("CHOLMOD_HIP_ARENA_BUDGET_MB") ;
        double fixed = 8.0 * P->xsize + 8.0 * P->ssize + 4.0 * (P->ssize - n) + 3e9 ;
        if (e && atof (e) > 0) P->arena_budget = (i64) (atof (e) * 1048576.0) ;
        else if (world > 1) P->arena_budget = -1 ;         // (build_host: from the largest part of L any rank holds)
        else if (!host_only)
        {
            size_t freeb = 0, totalb = 0 ;
            if (hipMemGetInfo (&freeb, &totalb) == hipSuccess)
                P->arena_budget = (i64) std::max (1e9, (double) freeb - fixed) ;
        }
    }
    const bool ptiming = getenv ("CHOLMOD_HIP_PLAN_TIMING") != nullptr ;
    auto pnow = [] () { return std::chrono::duration<double> (std::chrono::steady_clock::now ().time_since_epoch ()).count () ; } ;
    double tp0 = pnow () ;
    *status = build_host (P) ;
    double tp1 = pnow () ;
    if (*status == CHOLMOD_HIP_OK && !host_only) *status = upload_plan (P) ;
    if (ptiming) fprintf (stderr, "cholmod_hip_plan_create: copy maps %.3f s, build_host %.3f s, upload_plan %.3f s\n",
        tp0 - tpc, tp1 - tp0, pnow () - tp1) ;
    if (*status != CHOLMOD_HIP_OK)
    {
        free_device (P) ; delete P ; return nullptr ;
    }
    return P ;
}

cholmod_hip_plan *cholmod_hip_plan_create (int64_t n, int64_t nsuper,
    const int64_t *super, const int64_t *pi, const int64_t *px, const int64_t *s,
    int flags, int *status)
{
    return cholmod_hip_plan_create_dist (n, nsuper, super, pi, px, s, flags, 0, 1, status) ;
}

int cholmod_hip_set_allreduce (cholmod_hip_plan *P, cholmod_hip_allreduce_fn fn, void *user)
{
    if (!P) return CHOLMOD_HIP_INVALID ;
    P->ar_fn = fn ; P->ar_user = user ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_get_groups (cholmod_hip_plan *P, int64_t *first, int64_t *size)
{
    if (!P || !first || !size) return CHOLMOD_HIP_INVALID ;
    for (i64 q = 0 ; q < P->nsuper ; q++) { first [q] = P->grp0 [q] ; size [q] = P->grpn [q] ; }
    return CHOLMOD_HIP_OK ;
}

/* test hook: the (contributor, ancestor) pairs of this rank's plan -- front d extend-adds into the windows of the shared
 * front a through a map of its own (contributions routed past the contribution blocks in between); and per front the block
 * [cb_lo, cb_hi) of contribution-block columns this rank stores (-1, -1: the whole block, or not held).  Returns the number of
 * pairs; fills at most cap of them. */
int64_t cholmod_hip_debug_routing (cholmod_hip_plan *P, int64_t cap, int64_t *pair_d, int64_t *pair_a, int64_t *cb_lo, int64_t *cb_hi)
{
    if (!P) return CHOLMOD_HIP_INVALID ;
    for (size_t q = 0 ; q < P->relpairs.size () && (i64) q < cap ; q++)
    {
        if (pair_d) pair_d [q] = P->relpairs [q].d ;
        if (pair_a) pair_a [q] = P->relpairs [q].a ;
    }
    for (i64 s = 0 ; s < P->nsuper ; s++)
    {
        const FrontD &f = P->fr [s] ;
        if (cb_lo) cb_lo [s] = f.cbd ? f.cb_lo : -1 ;
        if (cb_hi) cb_hi [s] = f.cbd ? f.cb_hi : -1 ;
    }
    return (int64_t) P->relpairs.size () ;
}

/* Progress of the factorization that is running (or ran last), for a watchdog thread of the caller:
 * enable != 0 allocates two words of pinned host memory the device marks, in stream order, around every block-column
 * exchange (two one-thread kernels per exchange: microseconds next to the collective). */
int cholmod_hip_progress_enable (cholmod_hip_plan *P, int enable)
{
    if (!P || P->host_only) return CHOLMOD_HIP_INVALID ;
    if (enable && !P->prog_dev)
    {
        HIPCHK (hipHostMalloc ((void **) &P->prog_dev, 2 * sizeof (long long), hipHostMallocDefault)) ;
        P->prog_dev [0] = P->prog_dev [1] = 0 ;
    }
    else if (!enable && P->prog_dev)
    {
        (void) hipStreamSynchronize (P->stream) ;
        if (P->stream2) (void) hipStreamSynchronize (P->stream2) ;
        (void) hipHostFree (P->prog_dev) ; P->prog_dev = nullptr ;
    }
    return CHOLMOD_HIP_OK ;
}

/* out [0] factorizations started on this plan, [1] launches of the schedule the host has enqueued in the current one,
 * [2] launches in the schedule, [3] exchanges enqueued, [4] exchanges in the schedule, [5] / [6] exchange the DEVICE has
 * entered / left in the current factorization (markers; -1 without cholmod_hip_progress_enable), and of the exchange
 * entered and not left ([5] > [6]): [7] kind (7 = reduce-scatter + broadcast, 11 = all-gather), [8] first rank and [9]
 * size of its group, [10] columns of its block column, [11] rows below that block column's diagonal block.  Safe to call from another thread while a factorization runs. */
int cholmod_hip_progress (cholmod_hip_plan *P, int64_t *out)
{
    if (!P || !out) return CHOLMOD_HIP_INVALID ;
    for (int q = 0 ; q < 12 ; q++) out [q] = 0 ;
    const long long fact = P->prog_fact ;
    out [0] = fact ; out [1] = P->prog_launch ; out [2] = (int64_t) P->sch.launches.size () ; out [3] = P->prog_xchg_enq ;
    i64 nx = 0 ;
    for (const Launch &L : P->sch.launches) if (L.kind == K_XCHG_RS || L.kind == K_XCHG_AG) nx++ ;
    out [4] = nx ; out [5] = out [6] = -1 ;
    if (P->prog_dev)
    {
        const long long a = ((volatile long long *) P->prog_dev) [0], b = ((volatile long long *) P->prog_dev) [1] ;
        out [5] = (a >> 32) == fact ? (a & 0xFFFFFFFFll) : 0 ;
        out [6] = (b >> 32) == fact ? (b & 0xFFFFFFFFll) : 0 ;
        if (out [5] > out [6])
        {
            i64 seq = 0 ;
            for (const Launch &L : P->sch.launches)
                if ((L.kind == K_XCHG_RS || L.kind == K_XCHG_AG) && ++seq == out [5])
                {
                    out [7] = L.kind ; out [8] = L.ar_g0 ; out [9] = L.ar_gn ; out [10] = L.xd.w ; out [11] = L.xd.mb ;
                    break ;
                }
        }
    }
    return CHOLMOD_HIP_OK ;
}

int64_t cholmod_hip_get_batches (cholmod_hip_plan *P, int64_t *batch_of, int64_t *global_arena)
{
    if (!P) return CHOLMOD_HIP_INVALID ;
    if (batch_of) for (i64 q = 0 ; q < P->nsuper ; q++) batch_of [q] = P->batch_of [q] ;
    if (global_arena) *global_arena = P->global_arena ;
    return P->nsplit ;
}

int cholmod_hip_get_partition (cholmod_hip_plan *P, int64_t *owner)
{
    if (!P || !owner) return CHOLMOD_HIP_INVALID ;
    for (i64 q = 0 ; q < P->nsuper ; q++) owner [q] = P->owner [q] ;
    return CHOLMOD_HIP_OK ;
}

/* After a distributed factorization every rank holds, in its own packed array, the shared
 * fronts of its groups and its own subtrees.  The gather builds the complete factor in the
 * reference layout (L->px) on EVERY rank: each front is written into a zeroed full-size array by
 * exactly one rank (the first of its group), and a sum over all ranks completes it everywhere.
 * The full array needs 8 xsize bytes next to the rank's own part; if that does not fit, the
 * contribution-block arena (dead between factorizations) makes room and is allocated again by
 * the next factorization. */
int cholmod_hip_gather_factor (cholmod_hip_plan *P)
{
    if (!P || P->host_only) return CHOLMOD_HIP_INVALID ;
    if (P->world == 1) return CHOLMOD_HIP_OK ;
    if (!P->ar_fn && !P->nccl_world) return CHOLMOD_HIP_INVALID ;
    P->winv_valid = false ;
    HIPCHK (hipStreamSynchronize (P->stream)) ;
    // test hooks: CHOLMOD_HIP_TEST_FAIL_GATHER=r: rank r finds no room for the complete factor at all;
    // CHOLMOD_HIP_TEST_GATHER_STAGED=r: rank r finds none next to its own part (the host-staged way below)
    const char *tfg = TEST_ENV ("CHOLMOD_HIP_TEST_FAIL_GATHER"), *tgs = TEST_ENV ("CHOLMOD_HIP_TEST_GATHER_STAGED") ;
    const bool fail_here = tfg && atoi (tfg) == P->rank, staged_here = tgs && atoi (tgs) == P->rank ;
    // (nothing newer than the gathered copy: the same on every rank -- full_valid is set by a complete gather
    // and cleared by a factorization, collectively both)
    if (P->full_valid && P->d_Lx_full && !fail_here && !staged_here) return CHOLMOD_HIP_OK ;
    if ((fail_here || staged_here) && P->d_Lx_full) { (void) hipFree (P->d_Lx_full) ; P->d_Lx_full = nullptr ; }
    std::unique_ptr<double []> own_host ;   // the rank's own part of L on the host (staged way only)
    bool staged = false ;
    auto restore_own = [&] () -> bool       // the rank's own array back from the host copy
    {
        if (hipMalloc ((void **) &P->d_Lx, std::max<i64> (P->lx_local, 1) * sizeof (double)) == hipSuccess
            && hipMemcpy (P->d_Lx, own_host.get (), (size_t) P->lx_local * sizeof (double), hipMemcpyHostToDevice) == hipSuccess) return true ;
        (void) hipGetLastError () ;
        fprintf (stderr, "cholmod_hip_gather_factor: rank %d lost its part of the factor\n", P->rank) ;
        if (P->d_Lx) { (void) hipFree (P->d_Lx) ; P->d_Lx = nullptr ; }
        return false ;
    } ;
    auto try_full = [&] () -> bool
    {
        if (hipMalloc ((void **) &P->d_Lx_full, std::max<i64> (P->xsize, 1) * sizeof (double)) == hipSuccess) return true ;
        (void) hipGetLastError () ;
        P->d_Lx_full = nullptr ;
        return false ;
    } ;
    if (!P->d_Lx_full && !fail_here && P->d_Lx)
    {
        bool got = !staged_here && try_full () ;
        if (!got && !staged_here)
        {
            // the contribution-block arena is dead between factorizations: it makes room
            if (P->d_cb) { (void) hipFree (P->d_cb) ; P->d_cb = nullptr ; }
            got = try_full () ;
        }
        if (!got)
        {
            // Still no room next to the rank's own part (two ranks at Poisson 200^3: 117 GB + 181.6 GB): the own
            // part takes a detour through host memory -- download, release it (and the arena), reserve the complete
            // array, upload the fronts this rank contributes straight into their places.  The next factorization
            // reserves the rank's own array again (run_factorize).
            // (only with plenty of host memory to spare -- the other ranks of the node may be doing the same, and an
            // over-committed allocation fails when it is touched, not here: 40 % of what /proc/meminfo calls available)
            double avail = 0 ;
            if (FILE *mf = fopen ("/proc/meminfo", "r"))
            {
                char line [256] ;
                while (fgets (line, sizeof (line), mf))
                    if (strncmp (line, "MemAvailable:", 13) == 0) { avail = 1024.0 * atof (line + 13) ; break ; }
                fclose (mf) ;
            }
            if (8.0 * (double) P->lx_local <= 0.4 * avail)
                own_host.reset (new (std::nothrow) double [(size_t) std::max<i64> (P->lx_local, 1)]) ;
            if (own_host && hipMemcpy (own_host.get (), P->d_Lx, (size_t) P->lx_local * sizeof (double), hipMemcpyDeviceToHost) == hipSuccess)
            {
                if (P->d_cb) { (void) hipFree (P->d_cb) ; P->d_cb = nullptr ; }
                (void) hipFree (P->d_Lx) ; P->d_Lx = nullptr ;
                if (try_full ()) staged = true ;
                else (void) restore_own () ;        // (not even alone: the rank's part goes back where it was)
            }
            else (void) hipGetLastError () ;
            if (!P->d_Lx_full)
                fprintf (stderr, "cholmod_hip_gather_factor: no room for the complete factor (%.1f GB) on rank %d\n", 8e-9 * P->xsize, P->rank) ;
        }
    }
    // everything local that can still fail comes BEFORE the agreement, and its outcome is part of the vote
    bool setup_ok = true ;
    if (!P->d_fr_full)
    {
        // (descriptors of the complete factor: the global offsets, every column in place)
        std::vector<FrontD> ff (P->fr) ;
        for (i64 q = 0 ; q < P->nsuper ; q++) { ff [q].psx = P->px [q] ; ff [q].own_w = 0 ; ff [q].own_g = 1 ; ff [q].own_r = 0 ; }
        hipError_t e ;
        P->d_fr_full = dupload (ff, e) ;
        if (e != hipSuccess) { (void) hipGetLastError () ; if (P->d_fr_full) (void) hipFree (P->d_fr_full) ; P->d_fr_full = nullptr ; setup_ok = false ; }
    }
    // one number summed over all ranks, in d_xchg; < 0: the exchange itself failed
    auto agree = [&] (double mine) -> double
    {
        double any = 0.0 ;
        if (hipMemcpy (P->d_xchg, &mine, sizeof (double), hipMemcpyHostToDevice) != hipSuccess) { (void) hipGetLastError () ; mine = 1.0 ; }
        if (P->nccl_world)
        {
            if (rccl_api ()->AllReduce (P->d_xchg, P->d_xchg, 1, ncclDouble, ncclSum, P->nccl_world, P->stream) != ncclSuccess) return -1.0 ;
            if (hipStreamSynchronize (P->stream) != hipSuccess) { (void) hipGetLastError () ; return -1.0 ; }
        }
        else if (P->ar_fn (P->d_xchg, 1, 0, P->world, P->ar_user) != 0) return -1.0 ;
        if (hipMemcpy (&any, P->d_xchg, sizeof (double), hipMemcpyDeviceToHost) != hipSuccess) { (void) hipGetLastError () ; return -1.0 ; }
        return any + (mine != 0.0 && any == 0.0 ? 1.0 : 0.0) ;
    } ;
    {
        // every rank must enter the sums below or none: a rank without room tells the others first
        // (a rank that returned on its own would leave them waiting in the collective)
        double any = agree ((P->d_Lx_full && setup_ok) ? 0.0 : 1.0) ;
        if (any != 0.0)
        {
            if (P->d_Lx_full) { (void) hipFree (P->d_Lx_full) ; P->d_Lx_full = nullptr ; }
            P->full_valid = false ;
            if (staged && !restore_own ()) return CHOLMOD_HIP_GPU_PROBLEM ;     // (another rank had no room: this one keeps its part)
            return any < 0.0 ? CHOLMOD_HIP_GPU_PROBLEM : (setup_ok ? CHOLMOD_HIP_OUT_OF_MEMORY : CHOLMOD_HIP_GPU_PROBLEM) ;
        }
    }
    // From here on every rank enters every collective, whatever happens to it locally: a failure is kept in
    // `bad`, the remaining sums are still entered (their data no longer matters), and a trailing agreement
    // tells everybody.
    int bad = CHOLMOD_HIP_OK ;
    auto hold = [&] (hipError_t e) { if (e != hipSuccess) { (void) hipGetLastError () ; if (bad == CHOLMOD_HIP_OK) { bad = CHOLMOD_HIP_GPU_PROBLEM ;
        fprintf (stderr, "cholmod_hip_gather_factor: %s on rank %d\n", hipGetErrorString (e), P->rank) ; } } } ;
    hold (hipMemsetAsync (P->d_Lx_full, 0, std::max<i64> (P->xsize, 1) * sizeof (double), P->stream)) ;
    auto put = [&] (i64 dst, i64 src, i64 len)      // a piece of the rank's part into its place in the complete factor
    {
        if (len <= 0) return ;
        if (staged) hold (hipMemcpyAsync (P->d_Lx_full + dst, own_host.get () + src, (size_t) len * sizeof (double), hipMemcpyHostToDevice, P->stream)) ;
        else hold (hipMemcpyAsync (P->d_Lx_full + dst, P->d_Lx + src, (size_t) len * sizeof (double), hipMemcpyDeviceToDevice, P->stream)) ;
    } ;
    for (i64 q = 0 ; q < P->nsuper ; )
    {
        if (P->lpx [q] < 0) { q++ ; continue ; }
        const FrontD &f = P->fr [q] ;
        if (f.own_w)
        {
            // a distributed front: the slabs this rank owns (whole columns, contiguous in both arrays)
            for (int c0 = 0 ; c0 < f.nscol ; c0 += f.own_w)
                if (col_owned (f, c0))
                    put (P->px [q] + (i64) c0 * f.nsrow, P->lpx [q] + (i64) col_local (f, c0) * f.nsrow, (i64) std::min (f.own_w, f.nscol - c0) * f.nsrow) ;
            q++ ;
            continue ;
        }
        // runs of consecutive fronts this rank contributes: contiguous in both arrays
        if (P->rank != P->grp0 [q]) { q++ ; continue ; }
        i64 e = q ;
        while (e < P->nsuper && P->lpx [e] >= 0 && !P->fr [e].own_w && P->rank == P->grp0 [e] && P->lpx [e] - P->lpx [q] == P->px [e] - P->px [q]) e++ ;
        put (P->px [q], P->lpx [q], P->px [e] - P->px [q]) ;
        q = e ;
    }
    hold (hipStreamSynchronize (P->stream)) ;
    const i64 chunk = (i64) 1 << 27 ;
    for (i64 off = 0 ; off < P->xsize ; off += chunk)
    {
        i64 cnt = std::min (chunk, P->xsize - off) ;
        if (P->nccl_world)
        {
            if (rccl_api ()->AllReduce (P->d_Lx_full + off, P->d_Lx_full + off, (size_t) cnt, ncclDouble, ncclSum, P->nccl_world, P->stream) != ncclSuccess
                && bad == CHOLMOD_HIP_OK) bad = CHOLMOD_HIP_GPU_PROBLEM ;
        }
        else if (P->ar_fn (P->d_Lx_full + off, cnt, 0, P->world, P->ar_user) != 0 && bad == CHOLMOD_HIP_OK) bad = CHOLMOD_HIP_GPU_PROBLEM ;
    }
    hold (hipStreamSynchronize (P->stream)) ;
    {
        double any = agree (bad == CHOLMOD_HIP_OK ? 0.0 : 1.0) ;
        if (any != 0.0)
        {
            // somebody's copy is not the factor: nobody keeps one; a rank that staged its part through the host gets it back
            (void) hipFree (P->d_Lx_full) ; P->d_Lx_full = nullptr ;
            P->full_valid = false ;
            if (staged) (void) restore_own () ;
            return bad != CHOLMOD_HIP_OK ? bad : CHOLMOD_HIP_GPU_PROBLEM ;
        }
    }
    P->full_valid = true ;
    return CHOLMOD_HIP_OK ;
}

/* ---- native exchange over RCCL ------------------------------------------------- */

int cholmod_hip_rccl_unique_id (void *id128)
{
    RcclApi *R = rccl_api () ;
    if (!R || !id128) return CHOLMOD_HIP_NO_DEVICE ;
    static_assert (sizeof (ncclUniqueId) == 128, "ncclUniqueId is 128 bytes") ;
    ncclUniqueId id ;
    RCCLCHK (R->GetUniqueId (&id)) ;
    memcpy (id128, &id, sizeof (id)) ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_rccl_attach (cholmod_hip_plan *P, const void *id128)
{
    if (!P || P->host_only || !id128) return CHOLMOD_HIP_INVALID ;
    RcclApi *R = rccl_api () ;
    if (!R) return CHOLMOD_HIP_NO_DEVICE ;
    if (P->nccl_world) return CHOLMOD_HIP_OK ;
    ncclUniqueId id ;
    memcpy (&id, id128, sizeof (id)) ;
    RCCLCHK (R->CommInitRank (&P->nccl_world, P->world, id, P->rank)) ;
    // one communicator per rank group the plan shares fronts over; every rank of
    // the world walks the same sorted list (the split is collective over the world)
    std::map<i64, int> groups ;
    for (i64 s = 0 ; s < P->nsuper ; s++)
        if (P->grpn [s] > 1 && P->grpn [s] < P->world) groups [((i64) P->grp0 [s] << 16) | (i64) P->grpn [s]] = 1 ;
    int color = 0 ;
    for (auto &g : groups)
    {
        int g0 = (int) (g.first >> 16), gn = (int) (g.first & 0xffff) ;
        bool member = P->rank >= g0 && P->rank < g0 + gn ;
        ncclComm_t sub = nullptr ;
        RCCLCHK (R->CommSplit (P->nccl_world, member ? color : NCCL_SPLIT_NOCOLOR, P->rank, &sub, nullptr)) ;
        if (member) P->nccl_group [g.first] = sub ;
        color++ ;
    }
    if (!P->ar_done) HIPCHK (hipEventCreateWithFlags (&P->ar_done, hipEventDisableTiming)) ;
    // self check: a sum of ones over every communicator this rank belongs to must give
    // the size of its group (catches a wrong split before any factor data moves)
    {
        std::vector<std::pair<ncclComm_t, int>> mine ;
        mine.push_back ({P->nccl_world, P->world}) ;
        for (auto &g : P->nccl_group) mine.push_back ({g.second, (int) (g.first & 0xffff)}) ;
        for (auto &c : mine)
        {
            double one = 1.0, got = 0.0 ;
            HIPCHK (hipMemcpyAsync (P->d_xchg, &one, sizeof (double), hipMemcpyHostToDevice, P->stream)) ;
            RCCLCHK (R->AllReduce (P->d_xchg, P->d_xchg, 1, ncclDouble, ncclSum, c.first, P->stream)) ;
            HIPCHK (hipMemcpyAsync (&got, P->d_xchg, sizeof (double), hipMemcpyDeviceToHost, P->stream)) ;
            HIPCHK (hipStreamSynchronize (P->stream)) ;
            if (got != (double) c.second)
            {
                fprintf (stderr, "cholmod_hip_rccl_attach: self check failed (sum %g over a group of %d)\n", got, c.second) ;
                return CHOLMOD_HIP_GPU_PROBLEM ;
            }
        }
    }
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_rccl_detach (cholmod_hip_plan *P)
{
    if (!P) return CHOLMOD_HIP_INVALID ;
    if (RcclApi *R = (P->nccl_world ? rccl_api () : nullptr))
    {
        for (auto &g : P->nccl_group) (void) R->CommDestroy (g.second) ;
        (void) R->CommDestroy (P->nccl_world) ;
    }
    P->nccl_group.clear () ; P->nccl_world = nullptr ;
    return CHOLMOD_HIP_OK ;
}

void cholmod_hip_plan_destroy (cholmod_hip_plan *P)
{
    if (!P) return ;
    free_device (P) ;
    delete P ;
}

int cholmod_hip_upload_matrix (cholmod_hip_plan *P, const int64_t *Sp, const int64_t *Si,
    const int64_t *Snz, const double *Sx)
{
    if (!P || P->host_only || !Sp || !Si || !Sx) return CHOLMOD_HIP_INVALID ;
    i64 n = P->n ;
    i64 nz = 0 ;
    if (Snz) { for (i64 j = 0 ; j < n ; j++) nz = std::max<i64> (nz, Sp [j] + Snz [j]) ; }
    else nz = Sp [n] ;
    if (nz > P->s_nz || !P->d_Sp)
    {
        if (P->d_Sp) { (void) hipFree (P->d_Sp) ; (void) hipFree (P->d_Si) ; (void) hipFree (P->d_Sx) ; (void) hipFree (P->d_Snz) ; (void) hipFree (P->d_amap) ; }
        P->d_Sp = P->d_Si = P->d_Snz = P->d_amap = nullptr ; P->d_Sx = nullptr ;
        HIPCHK (hipMalloc ((void **) &P->d_Sp, (n + 1) * sizeof (i64))) ;
        HIPCHK (hipMalloc ((void **) &P->d_Snz, std::max<i64> (n, 1) * sizeof (i64))) ;
        HIPCHK (hipMalloc ((void **) &P->d_Si, std::max<i64> (nz, 1) * sizeof (i64))) ;
        HIPCHK (hipMalloc ((void **) &P->d_Sx, std::max<i64> (nz, 1) * sizeof (double))) ;
        HIPCHK (hipMalloc ((void **) &P->d_amap, std::max<i64> (nz, 1) * sizeof (i64))) ;
        P->s_nz = nz ;
    }
    HIPCHK (hipMemcpyAsync (P->d_Sp, Sp, (n + 1) * sizeof (i64), hipMemcpyHostToDevice, P->stream)) ;
    if (Snz) HIPCHK (hipMemcpyAsync (P->d_Snz, Snz, n * sizeof (i64), hipMemcpyHostToDevice, P->stream)) ;
    if (nz) HIPCHK (hipMemcpyAsync (P->d_Si, Si, nz * sizeof (i64), hipMemcpyHostToDevice, P->stream)) ;
    if (nz) HIPCHK (hipMemcpyAsync (P->d_Sx, Sx, nz * sizeof (double), hipMemcpyHostToDevice, P->stream)) ;
    if (!P->sch.sm.empty ())
    {
        // the range of S every thin front's columns occupy, in the block order of its launch
        std::vector<i64> sp01 (2 * P->sch.sm.size ()) ;
        for (size_t q = 0 ; q < P->sch.sm.size () ; q++)
        {
            const FrontD &f = P->fr [P->sch.sm [q]] ;
            sp01 [2 * q] = Sp [f.k1] ; sp01 [2 * q + 1] = Sp [f.k1 + f.nscol] ;
        }
        HIPCHK (hipMemcpyAsync (P->d_sp01, sp01.data (), sp01.size () * sizeof (i64), hipMemcpyHostToDevice, P->stream)) ;
        HIPCHK (hipStreamSynchronize (P->stream)) ;     // (sp01 is a local)
    }
    HIPCHK (hipStreamSynchronize (P->stream)) ;
    P->s_unpacked = (Snz != nullptr) ;
    P->amap_valid = false ;         // a new pattern may have come with the new values
    P->s_cur_nz = nz ;
    P->vsrc_nz = 0 ;                // ... and the value map of the previous one is void
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_set_value_map (cholmod_hip_plan *P, const int64_t *src, int64_t snz, int64_t nvalues)
{
    if (!P || P->host_only || !src || !P->d_Sp || P->s_unpacked || snz != P->s_cur_nz || nvalues < 0) return CHOLMOD_HIP_INVALID ;
    for (i64 q = 0 ; q < snz ; q++) if (src [q] < 0 || src [q] >= nvalues) return CHOLMOD_HIP_INVALID ;
    if (P->d_vsrc) { (void) hipFree (P->d_vsrc) ; P->d_vsrc = nullptr ; }
    if (P->d_vals) { (void) hipFree (P->d_vals) ; P->d_vals = nullptr ; }
    HIPCHK (hipMalloc ((void **) &P->d_vsrc, std::max<i64> (snz, 1) * sizeof (i64))) ;
    HIPCHK (hipMalloc ((void **) &P->d_vals, std::max<i64> (nvalues, 1) * sizeof (double))) ;
    if (snz) HIPCHK (hipMemcpy (P->d_vsrc, src, snz * sizeof (i64), hipMemcpyHostToDevice)) ;
    P->vsrc_nz = snz ; P->vals_n = nvalues ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_refresh_values (cholmod_hip_plan *P, const double *values, int64_t nvalues)
{
    if (!P || P->host_only || !values || !P->d_vsrc || P->vsrc_nz != P->s_cur_nz || nvalues != P->vals_n)
        return CHOLMOD_HIP_INVALID ;
    if (nvalues) HIPCHK (hipMemcpyAsync (P->d_vals, values, nvalues * sizeof (double), hipMemcpyHostToDevice, P->stream)) ;
    if (P->vsrc_nz)
        hipLaunchKernelGGL (k_gather_values, dim3 ((unsigned) ((P->vsrc_nz + 255) / 256)), dim3 (256), 0, P->stream,
            P->vsrc_nz, P->d_vsrc, P->d_vals, P->d_Sx) ;
    HIPCHK (hipStreamSynchronize (P->stream)) ;     // the caller may reuse `values` at once
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_factorize_resident (cholmod_hip_plan *P, double beta,
    int quick_return_if_not_posdef, int64_t *minor)
{
    if (!P || P->host_only) return CHOLMOD_HIP_INVALID ;
    i64 m = P->n ;
    int rc = run_factorize (P, beta, quick_return_if_not_posdef, &m) ;
    if (minor) *minor = m ;
    return rc ;
}

int cholmod_hip_download_factor (cholmod_hip_plan *P, double *Lx_host)
{
    if (!P || P->host_only || !Lx_host) return CHOLMOD_HIP_INVALID ;
    const double *Lw = whole_factor (P) ;
    if (!Lw) return CHOLMOD_HIP_INVALID ;           // several ranks: cholmod_hip_gather_factor first
    HIPCHK (hipMemcpy (Lx_host, Lw, P->xsize * sizeof (double), hipMemcpyDeviceToHost)) ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_upload_factor (cholmod_hip_plan *P, const double *Lx_host)
{
    if (!P || P->host_only || !Lx_host) return CHOLMOD_HIP_INVALID ;
    double *Lw = P->d_Lx ;
    if (P->world > 1)
    {
        // several ranks: the complete factor lives beside the rank's own part
        if (!P->d_Lx_full) HIPCHK (hipMalloc ((void **) &P->d_Lx_full, std::max<i64> (P->xsize, 1) * sizeof (double))) ;
        if (!P->d_fr_full)
        {
            std::vector<FrontD> ff (P->fr) ;
            for (i64 q = 0 ; q < P->nsuper ; q++) { ff [q].psx = P->px [q] ; ff [q].own_w = 0 ; ff [q].own_g = 1 ; ff [q].own_r = 0 ; }
            hipError_t e ;
            P->d_fr_full = dupload (ff, e) ; HIPCHK (e) ;
        }
        Lw = P->d_Lx_full ;
    }
    HIPCHK (hipMemcpy (Lw, Lx_host, P->xsize * sizeof (double), hipMemcpyHostToDevice)) ;
    P->winv_valid = false ;
    P->full_valid = P->world > 1 ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_factorize (cholmod_hip_plan *P, const int64_t *Sp, const int64_t *Si,
    const int64_t *Snz, const double *Sx, double beta, int quick_return_if_not_posdef,
    double *Lx_host, int64_t *minor)
{
    int rc = cholmod_hip_upload_matrix (P, Sp, Si, Snz, Sx) ;
    if (rc != CHOLMOD_HIP_OK) return rc ;
    rc = cholmod_hip_factorize_resident (P, beta, quick_return_if_not_posdef, minor) ;
    if (rc < 0) return rc ;
    if (Lx_host)
    {
        int rc2 = cholmod_hip_download_factor (P, Lx_host) ;
        if (rc2 != CHOLMOD_HIP_OK) return rc2 ;
    }
    return rc ;
}

int cholmod_hip_solve (cholmod_hip_plan *P, int which, double *X, int64_t nrhs, int64_t ldx)
{
    if (!P || P->host_only || !X || nrhs < 0 || ldx < P->n) return CHOLMOD_HIP_INVALID ;
    if (nrhs == 0 || P->n == 0) return CHOLMOD_HIP_OK ;
    const double *Lw = whole_factor (P) ;
    const FrontD *frw = whole_fronts (P) ;
    if (!Lw || !frw) return CHOLMOD_HIP_INVALID ;  // several ranks: cholmod_hip_gather_factor first
    const bool cxs = (P->flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;
    i64 need = ldx * nrhs ;
    if (need > P->x_cap)
    {
        if (P->d_X) (void) hipFree (P->d_X) ;
        P->d_X = nullptr ;
        HIPCHK (hipMalloc ((void **) &P->d_X, need * sizeof (double))) ;
        P->x_cap = need ;
    }
    hipStream_t st = P->stream ;
    // workspace of the big-supernode walk
    if (!P->inv_tasks.empty ())
    {
        if (!P->d_winv)
        {
            hipError_t e ;
            HIPCHK (hipMalloc ((void **) &P->d_winv, P->inv_tasks.size () * 8192 * sizeof (double))) ;
            P->d_inv_tasks = dupload (P->inv_tasks, e) ; HIPCHK (e) ;
            P->d_sb_tasks = dupload (P->sb_tasks, e) ; HIPCHK (e) ;
            P->d_sb_commit = dupload (P->sb_commit, e) ; HIPCHK (e) ;
            HIPCHK (hipMalloc ((void **) &P->d_ticket, (size_t) std::max (P->sb_max_tasks, 1) * sizeof (unsigned int))) ;
            HIPCHK (hipMemset (P->d_ticket, 0, (size_t) std::max (P->sb_max_tasks, 1) * sizeof (unsigned int))) ;
            P->winv_valid = false ;
        }
        if (need > P->solved_cap)
        {
            if (P->d_solved) (void) hipFree (P->d_solved) ;
            P->d_solved = nullptr ;
            P->solved_cap = need ;
            HIPCHK (hipMalloc ((void **) &P->d_solved, P->solved_cap * sizeof (double))) ;
        }
        if ((i64) P->sb_max_tasks * SOLVE_SB * nrhs > P->sv_acc_cap)
        {
            if (P->d_sv_acc) (void) hipFree (P->d_sv_acc) ;
            P->d_sv_acc = nullptr ;
            P->sv_acc_cap = (i64) P->sb_max_tasks * SOLVE_SB * nrhs ;
            HIPCHK (hipMalloc ((void **) &P->d_sv_acc, P->sv_acc_cap * sizeof (double))) ;
            HIPCHK (hipMemset (P->d_sv_acc, 0, P->sv_acc_cap * sizeof (double))) ;
        }
        if (!P->winv_valid)
        {
            CXS_LAUNCH (k_diag_inv64, dim3 ((unsigned) P->inv_tasks.size ()), dim3 (64), 0, st,
                P->d_inv_tasks, frw, Lw, P->d_winv) ;
            P->winv_valid = true ;
        }
    }
    HIPCHK (hipMemcpyAsync (P->d_X, X, need * sizeof (double), hipMemcpyHostToDevice, st)) ;
    HIPCHK (hipEventRecord (P->ev0, st)) ;
    if (which == 0 || which == 1)
    {
        for (int l = 0 ; l < P->nlevels ; l++)
        {
            int nf = P->sv_ptr [l+1] - P->sv_ptr [l] ;
            if (nf) CXS_LAUNCH (k_lsolve, dim3 (nf), dim3 (256), 0, st,
                P->d_sv + P->sv_ptr [l], frw, P->d_Ls, Lw, P->d_X, (i64) ldx, (int) nrhs) ;
            for (int q = P->sb_lvl_ptr [l] ; q < P->sb_lvl_ptr [l+1] ; q++)
            {
                const auto &B = P->sb_launch [q] ;
                CXS_LAUNCH (k_solve_fwd_diag, dim3 (B.ntasks), dim3 (256), 0, st,
                    P->d_sb_tasks + B.first, frw, Lw, P->d_winv, P->d_X, (i64) ldx, (int) nrhs, P->d_solved) ;
                if (B.grid > 0) CXS_LAUNCH (k_solve_fwd_apply, dim3 (B.grid), dim3 (256), 0, st,
                    P->d_sb_tasks + B.first, (int) B.ntasks, frw, P->d_Ls, Lw,
                    P->d_X, (i64) ldx, (int) nrhs, P->d_solved) ;
            }
            const auto &Cm = P->sb_commit_launch [l] ;
            if (Cm.ntasks) hipLaunchKernelGGL (k_solve_commit, dim3 (Cm.grid), dim3 (256), 0, st,
                P->d_sb_commit + Cm.first, (int) Cm.ntasks, frw, P->d_X, (i64) ldx, (int) nrhs, P->d_solved) ;
        }
    }
    if (which == 0 || which == 2)
    {
        for (int l = P->nlevels - 1 ; l >= 0 ; l--)
        {
            for (int q = P->sb_lvl_ptr [l+1] - 1 ; q >= P->sb_lvl_ptr [l] ; q--)
            {
                const auto &B = P->sb_launch [q] ;
                if (B.grid > 0) CXS_LAUNCH (k_solve_bwd_apply, dim3 (B.grid), dim3 (256), 0, st,
                    P->d_sb_tasks + B.first, (int) B.ntasks, frw, P->d_Ls, Lw,
                    P->d_X, (i64) ldx, (int) nrhs, P->d_sv_acc) ;
                CXS_LAUNCH (k_solve_bwd_diag, dim3 (B.ntasks), dim3 (256), 0, st,
                    P->d_sb_tasks + B.first, frw, Lw, P->d_winv, P->d_X, (i64) ldx, (int) nrhs, P->d_sv_acc) ;
            }
            int nf = P->sv_ptr [l+1] - P->sv_ptr [l] ;
            if (nf) CXS_LAUNCH (k_ltsolve, dim3 (nf), dim3 (256), 0, st,
                P->d_sv + P->sv_ptr [l], frw, P->d_Ls, Lw, P->d_X, (i64) ldx, (int) nrhs) ;
        }
    }
    HIPCHK (hipGetLastError ()) ;
    HIPCHK (hipEventRecord (P->ev1, st)) ;
    HIPCHK (hipMemcpyAsync (X, P->d_X, need * sizeof (double), hipMemcpyDeviceToHost, st)) ;
    HIPCHK (hipStreamSynchronize (st)) ;
    {
        float ms = 0 ;
        if (hipEventElapsedTime (&ms, P->ev0, P->ev1) == hipSuccess) P->solve_seconds = 1e-3 * ms ;
    }
    return CHOLMOD_HIP_OK ;
}

static int ensure_check_tasks (cholmod_hip_plan *P) ;

int cholmod_hip_factor_checks (cholmod_hip_plan *P, double *out5)
{
    if (!P || P->host_only || !out5) return CHOLMOD_HIP_INVALID ;
    for (int q = 0 ; q < 5 ; q++) out5 [q] = 0 ;
    if (P->nsuper == 0) return CHOLMOD_HIP_OK ;
    { int rc = ensure_check_tasks (P) ; if (rc != CHOLMOD_HIP_OK) return rc ; }
    HIPCHK (hipMemsetAsync (P->d_chk_out, 0, 5 * sizeof (double), P->stream)) ;
    // (several ranks: the complete factor exists only after cholmod_hip_gather_factor)
    if (!whole_factor (P) || !whole_fronts (P)) return CHOLMOD_HIP_INVALID ;
    const bool cxs = (P->flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;
    CXS_LAUNCH (k_factor_checks, dim3 ((unsigned) P->nchk), dim3 (256), 0, P->stream,
        P->d_chk, whole_fronts (P), whole_factor (P), P->d_chk_out) ;
    HIPCHK (hipGetLastError ()) ;
    HIPCHK (hipMemcpyAsync (out5, P->d_chk_out, 5 * sizeof (double), hipMemcpyDeviceToHost, P->stream)) ;
    HIPCHK (hipStreamSynchronize (P->stream)) ;
    return CHOLMOD_HIP_OK ;
}

/* The same invariants over the fronts THIS rank answers for in a distributed factor (every front
 * belongs to the first rank of its group), read from the rank's own part of L -- no gathered copy
 * needed: the sums of out5 over all ranks are the invariants of the complete factor.  (One rank:
 * the same numbers as cholmod_hip_factor_checks.) */
int cholmod_hip_factor_checks_local (cholmod_hip_plan *P, double *out5)
{
    if (!P || P->host_only || !out5) return CHOLMOD_HIP_INVALID ;
    for (int q = 0 ; q < 5 ; q++) out5 [q] = 0 ;
    if (P->nsuper == 0) return CHOLMOD_HIP_OK ;
    if (!P->d_Lx) return CHOLMOD_HIP_INVALID ;          // (released by a host-staged gather: the gathered factor is what there is)
    std::vector<CheckTask> t ;
    for (i64 s = 0 ; s < P->nsuper ; s++)
        if (P->lpx [s] >= 0 && (P->fr [s].own_w || P->rank == P->grp0 [s]))
            for (int c0 = 0 ; c0 < P->fr [s].nscol ; c0 += CHK_COLS)
                if (col_owned (P->fr [s], c0)) t.push_back (CheckTask {(i32) s, c0}) ;       // (a slab is a multiple of CHK_COLS columns)
    if (t.empty ()) return CHOLMOD_HIP_OK ;
    hipError_t e ;
    CheckTask *dt = dupload (t, e) ; HIPCHK (e) ;
    double *dout = nullptr ;
    if (hipMalloc ((void **) &dout, 5 * sizeof (double)) != hipSuccess) { (void) hipGetLastError () ; (void) hipFree (dt) ; return CHOLMOD_HIP_OUT_OF_MEMORY ; }
    (void) hipMemsetAsync (dout, 0, 5 * sizeof (double), P->stream) ;
    const bool cxs = (P->flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;
    CXS_LAUNCH (k_factor_checks, dim3 ((unsigned) t.size ()), dim3 (256), 0, P->stream, dt, P->d_fr, P->d_Lx, dout) ;
    e = hipGetLastError () ;
    if (e == hipSuccess) e = hipMemcpyAsync (out5, dout, 5 * sizeof (double), hipMemcpyDeviceToHost, P->stream) ;
    if (e == hipSuccess) e = hipStreamSynchronize (P->stream) ;
    (void) hipFree (dt) ; (void) hipFree (dout) ;
    return e == hipSuccess ? CHOLMOD_HIP_OK : CHOLMOD_HIP_GPU_PROBLEM ;
}

static int ensure_check_tasks (cholmod_hip_plan *P)
{
    if (P->d_chk) return CHOLMOD_HIP_OK ;
    std::vector<CheckTask> t ;
    for (i64 s = 0 ; s < P->nsuper ; s++)
        for (int c0 = 0 ; c0 < P->fr [s].nscol ; c0 += CHK_COLS) t.push_back (CheckTask {(i32) s, c0}) ;
    hipError_t e ;
    P->d_chk = dupload (t, e) ; HIPCHK (e) ;
    P->nchk = (i64) t.size () ;
    HIPCHK (hipMalloc ((void **) &P->d_chk_out, 5 * sizeof (double))) ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_download_even_columns (cholmod_hip_plan *P, double *out_host)
{
    if (!P || P->host_only || !out_host) return CHOLMOD_HIP_INVALID ;
    if (P->nsuper == 0 || P->xsize == 0) return CHOLMOD_HIP_OK ;
    if (P->flags & CHOLMOD_HIP_CX_STORAGE)
    {
        // the factor is stored as its even columns: it IS the interleaved complex factor
        if (!whole_factor (P)) return CHOLMOD_HIP_INVALID ;
        HIPCHK (hipMemcpy (out_host, whole_factor (P), (size_t) P->xsize * sizeof (double), hipMemcpyDeviceToHost)) ;
        return CHOLMOD_HIP_OK ;
    }
    { int rc = ensure_check_tasks (P) ; if (rc != CHOLMOD_HIP_OK) return rc ; }
    double *tmp = nullptr ;
    const size_t bytes = (size_t) (P->xsize / 2) * sizeof (double) ;
    if (hipMalloc ((void **) &tmp, bytes) != hipSuccess) { (void) hipGetLastError () ; return CHOLMOD_HIP_OUT_OF_MEMORY ; }
    if (!whole_factor (P)) { (void) hipFree (tmp) ; return CHOLMOD_HIP_INVALID ; }
    hipLaunchKernelGGL (k_even_columns, dim3 ((unsigned) P->nchk), dim3 (256), 0, P->stream,
        P->d_chk, whole_fronts (P), whole_factor (P), tmp) ;
    hipError_t e = hipGetLastError () ;
    if (e == hipSuccess) e = hipMemcpyAsync (out_host, tmp, bytes, hipMemcpyDeviceToHost, P->stream) ;
    if (e == hipSuccess) e = hipStreamSynchronize (P->stream) ;
    (void) hipFree (tmp) ;
    return e == hipSuccess ? CHOLMOD_HIP_OK : CHOLMOD_HIP_GPU_PROBLEM ;
}

int cholmod_hip_get_maps (cholmod_hip_plan *P, int64_t *sparent, int64_t *level, int64_t *relmap)
{
    if (!P) return CHOLMOD_HIP_INVALID ;
    for (i64 s = 0 ; s < P->nsuper ; s++)
    {
        if (sparent) sparent [s] = P->fr [s].parent ;
        if (level) level [s] = P->level [s] ;
    }
    if (relmap)
    {
        if (P->host_only) return CHOLMOD_HIP_NO_DEVICE ;
        std::vector<i32> tmp (std::max<i64> (P->relsize, 1)) ;
        HIPCHK (hipMemcpy (tmp.data (), P->d_relmap, tmp.size () * sizeof (i32), hipMemcpyDeviceToHost)) ;
        // fronts without a parent have no entries; the compact layout has none either
        for (i64 q = 0 ; q < P->relsize ; q++) relmap [q] = tmp [q] ;
    }
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_get_stats (cholmod_hip_plan *P, double *stats)
{
    if (!P || !stats) return CHOLMOD_HIP_INVALID ;
    P->stats [1] = P->exec_flops ;
    P->stats [2] = (double) P->sch.launches.size () + 3 ;
    P->stats [3] = P->nlevels ;
    P->stats [4] = 8.0 * P->arena ;
    P->stats [5] = 8.0 * P->xsize ;
    P->stats [36] = 8.0 * P->lx_local ;
    P->stats [22] = P->nsplit ;
    P->stats [24] = P->solve_seconds ;
    for (int q = 0 ; q < CHOLMOD_HIP_NSTATS ; q++) stats [q] = P->stats [q] ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_debug_thin_cycles (cholmod_hip_plan *P, int64_t launch, long long *out10)
{
    if (!P || !P->d_thin_tim || launch < 0 || launch >= (i64) P->sch.launches.size ()) return CHOLMOD_HIP_INVALID ;
    HIPCHK (hipMemcpy (out10, P->d_thin_tim + 10 * launch, 10 * sizeof (long long), hipMemcpyDeviceToHost)) ;
    return CHOLMOD_HIP_OK ;
}

int64_t cholmod_hip_get_launch_profile (cholmod_hip_plan *P, int64_t cap, int32_t *kind, int32_t *grid,
    int32_t *aux, double *ms, double *flops, double *bytes)
{
    if (!P) return CHOLMOD_HIP_INVALID ;
    i64 nl = (i64) P->sch.launches.size () ;
    for (i64 q = 0 ; q < nl && q < cap ; q++)
    {
        const Launch &L = P->sch.launches [q] ;
        if (kind) kind [q] = L.kind ;
        if (grid) grid [q] = L.grid ;
        if (aux) aux [q] = L.aux ;
        if (ms) ms [q] = q < (i64) P->launch_ms.size () ? P->launch_ms [q] : 0.0 ;
        if (flops) flops [q] = L.flops ;
        if (bytes) bytes [q] = L.bytes ;
    }
    return nl ;
}

// test hook: a fingerprint of everything build_host derives for this rank -- fronts, child lists and routing pairs, the
// rank's layout of L and of the arena, every group array and the launch list (FNV-1a over the fields).  Host-only plans
// have it too: tests/test_schedule_fingerprint.py pins the schedules of a battery of problems, worlds and flags, so that a
// restructuring of the scheduler that changes any launch is caught without a GPU.
int cholmod_hip_debug_schedule_hash (cholmod_hip_plan *P, uint64_t *out16)
{
    if (!P || !out16) return CHOLMOD_HIP_INVALID ;
    auto fnv = [] (uint64_t h, const void *p, size_t nbytes) -> uint64_t
    {
        const unsigned char *b = (const unsigned char *) p ;
        for (size_t q = 0 ; q < nbytes ; q++) { h ^= b [q] ; h *= 0x100000001b3ull ; }
        return h ;
    } ;
    const uint64_t H0 = 0xcbf29ce484222325ull ;
    auto hv = [&] (const auto &v) -> uint64_t
    {
        return v.empty () ? H0 : fnv (H0, v.data (), v.size () * sizeof (v [0])) ;
    } ;
    const Schedule &S = P->sch ;
    out16 [0] = hv (S.zg) ; out16 [1] = hv (S.eg) ; out16 [2] = hv (S.pg) ; out16 [3] = hv (S.tg) ;
    out16 [4] = hv (S.gg) ; out16 [5] = hv (S.dg) ; out16 [6] = hv (S.rg) ; out16 [7] = hv (S.wg) ;
    out16 [8] = hv (S.cg) ; out16 [9] = hv (S.sm) ;
    uint64_t h = H0 ;
    for (const Launch &L : S.launches)
    {
        const i64 v [] = {L.kind, L.grid, L.ng, (i64) L.goff, L.stream, L.wait_ev, L.rec_ev, L.ar_g0, L.ar_gn, L.aux, L.leaf_T, L.leaf_pw, L.ndiag,
            L.xd.slab, L.xd.lda, L.xd.w, L.xd.mb, L.xd.R, L.xd.g, L.xd.r} ;
        h = fnv (h, v, sizeof (v)) ;
        const double d [] = {L.flops, L.bytes} ;
        h = fnv (h, d, sizeof (d)) ;
    }
    out16 [10] = h ;
    out16 [11] = hv (P->fr) ;
    out16 [12] = fnv (hv (P->child), P->crel.data (), P->crel.size () * sizeof (i64)) ;
    out16 [13] = P->relpairs.empty () ? H0 : fnv (H0, P->relpairs.data (), P->relpairs.size () * sizeof (RelPair)) ;
    const i64 w [] = {S.ncflags, S.max_dinv_slots, S.nevents, P->arena, P->lx_local, P->lx_fronts, P->nsplit, P->relsize_all, P->global_arena} ;
    out16 [14] = fnv (H0, w, sizeof (w)) ;
    out16 [15] = fnv (fnv (hv (P->lpx), P->win_off.data (), P->win_off.size () * sizeof (i64)), P->assign_cb.data (), P->assign_cb.size ()) ;
    return CHOLMOD_HIP_OK ;
}

// tuning: the update regions of launch `launch` (an update launch of any kind), 12 numbers per
// region: m, n, k, tri, c_in_cb, lda, ldc, ntiles, nblk, front, assign, swz
int64_t cholmod_hip_debug_launch_regions (cholmod_hip_plan *P, int64_t launch, int64_t cap, int64_t *out)
{
    if (!P || launch < 0 || launch >= (i64) P->sch.launches.size ()) return CHOLMOD_HIP_INVALID ;
    const Launch &L = P->sch.launches [launch] ;
    if (L.kind != K_UPD_W && L.kind != K_UPD_SMALL && L.kind != K_UPD_BIG && L.kind != K_UPD_PF) return 0 ;
    for (i64 q = 0 ; q < L.ng && q < cap ; q++)
    {
        const GemmGroup &G = P->sch.gg [L.goff + q] ;
        const i64 v [12] = {G.m, G.n, G.k, G.tri, G.c_in_cb, G.lda, G.ldc, G.ntiles, G.nblk, G.front, G.assign, G.swz} ;
        for (int t = 0 ; t < 12 ; t++) out [12 * q + t] = v [t] ;
    }
    return L.ng ;
}

int cholmod_hip_set_profiling (cholmod_hip_plan *P, int on)
{
    if (!P) return CHOLMOD_HIP_INVALID ;
    P->profiling = on != 0 ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_dense_partial_factor (double *F, int64_t nsrow, int64_t nscol, int flags,
    int64_t *info_out)
{
    if (!F || nsrow <= 0 || nscol <= 0 || nscol > nsrow) return CHOLMOD_HIP_INVALID ;
    if (!cholmod_hip_probe ()) return CHOLMOD_HIP_NO_DEVICE ;
    FrontD f ;
    memset (&f, 0, sizeof (f)) ;
    f.nscol = (i32) nscol ; f.nsrow = (i32) nsrow ; f.ncb = (i32) (nsrow - nscol) ; f.parent = -1 ;
    std::vector<FrontD> fr (1, f) ;
    Schedule S ;
    i32 id = 0 ;
    double fl = 0 ;
    schedule_dense (fr, &id, 1, S, flags, nullptr, nullptr, nullptr, 0, 1) ;
    (void) fl ;
    cholmod_hip_plan P ;
    P.flags = flags ;
    hipError_t e ;
    HIPCHK (hipStreamCreate (&P.stream)) ;
    HIPCHK (hipStreamCreate (&P.stream2)) ;
    for (int q = 0 ; q < S.nevents ; q++)
    {
        hipEvent_t ev ;
        HIPCHK (hipEventCreateWithFlags (&ev, hipEventDisableTiming)) ;
        P.sync_ev.push_back (ev) ;
    }
    i64 ncb = nsrow - nscol ;
    HIPCHK (hipMalloc ((void **) &P.d_Lx, nsrow * nscol * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &P.d_cb, std::max<i64> (ncb * ncb, 1) * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &P.d_info, sizeof (i32))) ;
    HIPCHK (hipMemset (P.d_info, 0, sizeof (i32))) ;
    P.d_pg = dupload (S.pg, e) ; HIPCHK (e) ;
    P.d_tg = dupload (S.tg, e) ; HIPCHK (e) ;
    HIPCHK (hipMalloc ((void **) &P.d_tu_cnt, std::max<size_t> (S.tg.size (), 1) * sizeof (i32))) ;
    HIPCHK (hipMemset (P.d_tu_cnt, 0, std::max<size_t> (S.tg.size (), 1) * sizeof (i32))) ;
    P.d_gg = dupload (S.gg, e) ; HIPCHK (e) ;
    P.d_dg = dupload (S.dg, e) ; HIPCHK (e) ;
    P.d_rg = dupload (S.rg, e) ; HIPCHK (e) ;
    P.d_cg = dupload (S.cg, e) ; HIPCHK (e) ;
    P.sch.ncflags = S.ncflags ;
    HIPCHK (hipMalloc ((void **) &P.d_cflags, (4 * (size_t) S.ncflags + 4) * sizeof (int))) ;
    HIPCHK (hipMemset (P.d_cflags, 0, (4 * (size_t) S.ncflags + 4) * sizeof (int))) ;
    HIPCHK (hipMalloc ((void **) &P.d_dinv, (size_t) std::max (S.max_dinv_slots, 1) * 4096 * sizeof (double))) ;
    P.sch.npcnt = S.npcnt ;
    if (S.npcnt > 0)
    {
        HIPCHK (hipMalloc ((void **) &P.d_pcnt, 8 * (size_t) S.npcnt * sizeof (int))) ;
        HIPCHK (hipMemset (P.d_pcnt, 0, 8 * (size_t) S.npcnt * sizeof (int))) ;
    }
    P.upd3_wg4 = lookahead_enabled () ;
    HIPCHK (hipMemcpy (P.d_Lx, F, nsrow * nscol * sizeof (double), hipMemcpyHostToDevice)) ;
    if (ncb > 0)
        HIPCHK (hipMemcpy2D (P.d_cb, ncb * sizeof (double), F + nscol + nscol * nsrow,
            nsrow * sizeof (double), ncb * sizeof (double), ncb, hipMemcpyHostToDevice)) ;
    HIPCHK (hipDeviceSynchronize ()) ;       // uploads above used the null stream
    for (const Launch &L : S.launches) run_launch (&P, L, false) ;
    HIPCHK (hipGetLastError ()) ;
    HIPCHK (hipStreamSynchronize (P.stream2)) ;
    HIPCHK (hipStreamSynchronize (P.stream)) ;
    HIPCHK (hipMemcpy (F, P.d_Lx, nsrow * nscol * sizeof (double), hipMemcpyDeviceToHost)) ;
    if (ncb > 0)
        HIPCHK (hipMemcpy2D (F + nscol + nscol * nsrow, nsrow * sizeof (double), P.d_cb,
            ncb * sizeof (double), ncb * sizeof (double), ncb, hipMemcpyDeviceToHost)) ;
    i32 inf = 0 ;
    HIPCHK (hipMemcpy (&inf, P.d_info, sizeof (i32), hipMemcpyDeviceToHost)) ;
    if (info_out) *info_out = inf ;
    free_device (&P) ;
    P.stream = nullptr ; P.stream2 = nullptr ; P.sync_ev.clear () ;
    P.d_Lx = P.d_cb = nullptr ; P.d_info = nullptr ;
    P.d_pg = nullptr ; P.d_tg = nullptr ; P.d_gg = nullptr ; P.d_tu_cnt = nullptr ;
    P.d_dg = nullptr ; P.d_rg = nullptr ; P.d_dinv = nullptr ; P.d_cg = nullptr ; P.d_cflags = nullptr ; P.d_pcnt = nullptr ;
    return CHOLMOD_HIP_OK ;
}

} // extern "C"
