// engine.hip -- host side of the MI355X supernodal Cholesky engine that touches the device: upload of a plan
// (plan_build.hip / schedule_dense.hip derive it), the runner of its launch list, the exchange over RCCL, the gather,
// the triangular solves, and the extern "C" shim declared in include/cholmod_hip.h.  One process drives one GPU.
#include "kernels.hip.h"
#include "plan.hip.h"
#include <thread>

#include <dlfcn.h>
#include <unistd.h>

namespace sship {
RcclApi *rccl_api ()
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
} // namespace sship
#define COMMA ,
// (solves, checks) K<true> for a complex factor in its own storage; needs `cxs` in scope
#define CXS_LAUNCH(K, ...) do { if (cxs) hipLaunchKernelGGL (K<true>, __VA_ARGS__) ; else hipLaunchKernelGGL (K<false>, __VA_ARGS__) ; } while (0)
#define RCCLCHK(call) do { ncclResult_t r_ = (call) ; if (r_ != ncclSuccess) { \
    fprintf (stderr, "cholmod_hip: %s failed: %s (%s:%d)\n", #call, \
        (rccl_api () && rccl_api ()->GetErrorString) ? rccl_api ()->GetErrorString (r_) : "?", __FILE__, __LINE__) ; \
    return CHOLMOD_HIP_GPU_PROBLEM ; } } while (0)


namespace {

static void free_device (cholmod_hip_plan *P)
{
    if (P->prog_dev) { (void) hipHostFree (P->prog_dev) ; P->prog_dev = nullptr ; }
    if (P->h_vals) { (void) hipHostFree (P->h_vals) ; P->h_vals = nullptr ; }
    if (P->values_ev) { (void) hipEventDestroy (P->values_ev) ; P->values_ev = nullptr ; }
    for (auto e : P->chunk_ev) (void) hipEventDestroy (e) ;
    P->chunk_ev.clear () ;
    if (P->d_vorder) { (void) hipFree (P->d_vorder) ; P->d_vorder = nullptr ; }
    if (RcclApi *R = (P->nccl_world ? rccl_api () : nullptr))
    {
        for (auto &g : P->nccl_group) (void) R->CommDestroy (g.second) ;
        (void) R->CommDestroy (P->nccl_world) ;
        P->nccl_group.clear () ; P->nccl_world = nullptr ;
    }
    if (P->ar_done) (void) hipEventDestroy (P->ar_done) ;
    void *ptrs [] = {P->d_Ls, P->d_fr, P->d_supermap, P->d_child, P->d_relmap, P->d_info,
        P->d_lvl_list, P->d_Lx, P->d_cb, P->d_zg, P->d_eg, P->d_pg, P->d_tg, P->d_tu_cnt, P->d_cdesc, P->d_smd, P->d_sp01, P->d_gg, P->d_sm,
        P->d_Sp, P->d_Si, P->d_Snz, P->d_Sx, P->d_amap, P->d_X, P->d_Y, P->d_perm, P->d_xchg, P->d_stage, P->d_ag, P->d_agf, P->d_Lx_full, P->d_fr_full, P->d_dg, P->d_rg, P->d_wg, P->d_cg, P->d_cflags, P->d_crel, P->d_relpairs, P->d_dinv, P->d_sv,
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
    HIPCHK (hipStreamCreate (&P->stream)) ;
    if (const char *e = TEST_ENV ("CHOLMOD_HIP_TEST_JITTER"))
    {
        unsigned long long seed = 0 ; int mx = 2000 ;
        if (sscanf (e, "%llu:%d", &seed, &mx) >= 1 && mx > 0)
        {
            P->jitter_us = mx ;
            P->jitter_state = seed * 0x9E3779B97F4A7C15ull + (unsigned long long) (P->rank + 1) * 0xD1B54A32D192ED03ull ;
        }
    }
    { const char *e = TEST_ENV ("CHOLMOD_HIP_TEST_DROP_WAITS") ; P->test_drop_waits = e ? std::max (atoi (e), 1) : 0 ; }
    if (const char *e = TEST_ENV ("CHOLMOD_HIP_TEST_HANG_EXCHANGE")) (void) sscanf (e, "%d:%ld:%ld", &P->test_hang_rank, &P->test_hang_xchg, &P->test_hang_fact) ;
    // several ranks: k_update3 with four tiles per workgroup, so that the exchange stream's (and RCCL's) four-wave workgroups
    // find room beside a trailing update (rocprofv3, rank 0 of 8 at 200^3: k_win_move 959 -> 94 ms in all, longest launch
    // 79 -> 1.2 ms; the update itself 2351 -> 2378 ms).  (its knob, measured both ways in round 4, is gone.)
    P->upd3_wg4 = P->world > 1 || P->force_shared ;
    {
        // The exchange stream runs BESIDE the rest of a trailing update (look-ahead: window open, extend-add, pack, the
        // collective): its small workgroups must get the wave slots the update's tiles free up, ahead of the update's own
        // remaining tiles -- at default priority rocprofv3 shows k_win_move stretched over the whole 78 ms of the update it
        // was meant to hide behind (round 4).  
        int least = 0, greatest = 0 ;
        if (hipDeviceGetStreamPriorityRange (&least, &greatest) == hipSuccess && greatest != least)
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
    HIPCHK (hipMalloc ((void **) &P->d_Lx, (std::max<i64> (P->lx_local, 1) + UPD3_LX_PAD) * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &P->d_cb, std::max<i64> (P->arena, 1) * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &P->d_xchg, 3 * (size_t) P->world * sizeof (double))) ;
    {
        i64 mx = 1, mxg = 1, mxf = 1 ;
        for (const Launch &L : P->sch.launches)
            if (L.kind == K_XCHG_RS || L.kind == K_XCHG_AG)
            {
                mx = std::max (mx, ((i64) L.xd.w * L.xd.w + ((i64) L.xd.R + L.xd.Rf) * L.xd.w) * L.xd.g) ;
                mxg = std::max (mxg, (i64) L.xd.R * L.xd.w * L.xd.g) ;
                mxf = std::max (mxf, (i64) L.xd.Rf * L.xd.w * L.xd.g) ;
            }
        HIPCHK (hipMalloc ((void **) &P->d_stage, (size_t) mx * sizeof (double))) ;
        HIPCHK (hipMalloc ((void **) &P->d_ag, (size_t) mxg * sizeof (double))) ;
        HIPCHK (hipMalloc ((void **) &P->d_agf, (size_t) mxf * sizeof (double))) ;
        P->agf_len = mxf ;
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

// waves per SIMD the one-wave thin-front kernel is compiled for, by size class (<= 32 / 48 / 64 rows): 4 / 4 / 3
// (measured on the 2D 1259^2 problem in round 2: 6 everywhere 1.044 ms, 4 / 4 / 3 1.004 ms; the knob and the other
// instantiations are gone)
static int thin_minw (int cls) { return cls == 2 ? 3 : 4 ; }

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
    // test must catch, tests/test_gpu_scale.py::test_stream_jitter_catches_a_dropped_wait; =2: only the joins with the far-row
    // gathers of the shared fronts, tests/test_dist.py)
    const bool drop_wait = P->test_drop_waits == 1 || (P->test_drop_waits == 2 && L.kind == K_JOIN) ;
    if (!serial && L.wait_ev >= 0 && L.kind != K_XCHG_RS && !drop_wait) HIPCHK (hipStreamWaitEvent (st, P->sync_ev [L.wait_ev], 0)) ;
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
        case K_JOIN: break ;        // (no launch of its own: a cross-stream wait, done above)
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
#define LEAF_PW(MINW_) \
                    { if (L.leaf_pw <= 4) LEAF_LAUNCH (4, MINW_) ; else if (L.leaf_pw <= 8) LEAF_LAUNCH (8, MINW_) ; \
                      else if (L.leaf_pw <= 12) LEAF_LAUNCH (12, MINW_) ; else LEAF_LAUNCH (16, MINW_) ; }
                    // (measured on the 2D 1259^2 leaf level: 4 waves per SIMD 0.112 ms, 5: 0.191, 6: 0.245)
                    LEAF_PW (4)
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
                    if (w == 3) THIN_LAUNCH (1, false, 3) ;
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
                // The all-gather of the far chunks (L.far, L.stream == 1) runs on the second stream
                // behind the event of the block column's chain (waited for above, like any launch's);
                // the main stream meets it at a K_JOIN ahead of the outer update.
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
                const i64 seg = (i64) X.w * X.w + ((i64) X.R + X.Rf) * X.w, chunk = (i64) (L.far ? X.Rf : X.R) * X.w ;
                double *agb = L.far ? P->d_agf : P->d_ag ;
                const long long xseq = ++P->prog_xchg_enq ;
                if (P->prog_dev) hipLaunchKernelGGL (k_mark, dim3 (1), dim3 (1), 0, cs, P->prog_dev, (P->prog_fact << 32) | xseq) ;
#ifdef CHOLMOD_HIP_TEST_HOOKS
                // test hook CHOLMOD_HIP_TEST_HANG_EXCHANGE=rank:seq[:fact]: that rank never issues exchange `seq` of its
                // factorization number `fact` (default: the second; its host thread sleeps here, nothing hangs on the
                // device) -- its peers then wait for it in the collective, which is what bench.py's watchdog must turn into
                // an error line (tests/test_bench_contract.py)
                if (P->rank == P->test_hang_rank && xseq == P->test_hang_xchg && P->prog_fact >= P->test_hang_fact)
                {
                    (void) hipStreamSynchronize (cs) ;
                    for ( ; ; ) sleep (3600) ;
                }
#endif
                auto move = [&] (int mode, i64 total)
                {
                    if (total <= 0) return ;
                    // (one workgroup per column and part: k_xchg_move)
                    const unsigned parts = mode == 0 ? 2u * (unsigned) X.g + 1 : (mode == 3 || mode == 5) ? (unsigned) X.g : mode == 1 ? 3u : 1u ;
                    const bool narrow = (ahead || (L.stream == 1 && !serial)) && narrow_xs () ;
                    hipLaunchKernelGGL (k_xchg_move, dim3 ((unsigned) X.w * parts), dim3 (narrow ? 64 : 256), 0, cs, X, mode, P->d_Lx, P->d_stage, agb) ;
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
                    if (!R) HIPCHK (hipMemsetAsync (agb, 0, (size_t) (chunk * X.g) * sizeof (double), cs)) ;
                    move (L.far ? 4 : 2, chunk) ;
                    if (R) RCCLCHK (R->AllGather (agb + (i64) X.r * chunk, agb, (size_t) chunk, ncclDouble, comm, cs)) ;
                    else
                    {
                        HIPCHK (hipStreamSynchronize (cs)) ;
                        if (P->ar_fn (agb, chunk * X.g, L.ar_g0, L.ar_gn, P->ar_user) != 0) return CHOLMOD_HIP_GPU_PROBLEM ;
                    }
                    move (L.far ? 5 : 3, chunk * X.g) ;
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
            if (P->upd3_wg4)
            {
                // (several ranks) four tiles per workgroup: see k_update3
                const unsigned g4 = (unsigned) (((L.grid + 31) / 32) * 8) ;
                if (L.aux >= 1024) TW_LAUNCH (k_update3<4 COMMA, COMMA 4>, dim3 (g4), dim3 (256), 0, st, P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
                else TW_LAUNCH (k_update3<2 COMMA, COMMA 4>, dim3 (g4), dim3 (256), 0, st, P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
            }
            else if (L.half)
            {
                // two waves per tile (k_update3 <..., HALF>): the launches of 512 .. 10 240 tiles, schedule_dense.hip
                const unsigned gh = 2u * (unsigned) ((L.grid + 7) / 8 * 8) ;
                if (L.aux >= 1024) TW_LAUNCH (k_update3<4 COMMA, COMMA 1 COMMA true>, dim3 (gh), dim3 (64), 0, st, P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
                else TW_LAUNCH (k_update3<2 COMMA, COMMA 1 COMMA true>, dim3 (gh), dim3 (64), 0, st, P->d_gg + L.goff, L.ng, P->d_Lx, P->d_cb) ;
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

// what the plan exchanges per factorization (stats [17], [18], [25], [39]; a property of the launch list)
static void exchange_volume (const cholmod_hip_plan *P, double *S)
{
    S [17] = S [18] = S [25] = S [39] = 0 ;
    const size_t nl = P->sch.launches.size () ;
    for (size_t q = 0 ; q < nl ; q++)
    {
        const Launch &L = P->sch.launches [q] ;
        if (L.kind != K_XCHG_RS && L.kind != K_XCHG_AG) continue ;
        S [17] += 1 ; S [18] += L.bytes ;
        if (L.kind != K_XCHG_AG) continue ;
        // an all-gather the main stream waits for at once: in line, or on the exchange stream with the join right behind it
        S [25] += L.bytes ;
        bool inl = L.stream == 0 ;
        for (size_t p = q + 1 ; !inl && p < nl ; p++)
        {
            const Launch &N = P->sch.launches [p] ;
            if (N.kind == K_JOIN) inl = true ;
            else if (!(N.kind == K_XCHG_AG && N.stream == 1)) break ;
        }
        if (inl) S [39] += L.bytes ;
    }
}

// Run the numeric factorization on the resident S.  Leaves Lx on the device.
/* the chunks 0 .. need-1 into S and L, on the main stream, as soon as each has been pushed and copied */
static int apply_value_chunks (cholmod_hip_plan *P, long need)
{
    while (P->chunks_applied < need)
    {
        const long c = P->chunks_applied ;
        for (long spin = 0 ; ; spin++)
        {
            const long have = P->chunks_pushed.load (std::memory_order_acquire) ;
            if (have < 0) return CHOLMOD_HIP_GPU_PROBLEM ;
            if (have > c) break ;
            if ((spin & 1023) == 1023) std::this_thread::yield () ;
        }
        HIPCHK (hipStreamWaitEvent (P->stream, P->chunk_ev [c], 0)) ;
        const i64 k0 = c * P->chunk_len, k1 = std::min<i64> (k0 + P->chunk_len, P->s_cur_nz) ;
        hipLaunchKernelGGL (k_values_chunk, dim3 ((unsigned) ((k1 - k0 + 255) / 256)), dim3 (256), 0, P->stream,
            k0, k1, P->d_vorder, P->d_vals, P->d_amap, P->d_Sx, P->d_Lx) ;
        P->chunks_applied = c + 1 ;
    }
    return CHOLMOD_HIP_OK ;
}

/* What a factorization enqueues before it touches a value: the start event, (re)allocation of what a gather released, the
 * clearing of L and of the per-factorization counters.  cholmod_hip_values_begin calls it ahead of time, so that the
 * clearing runs beside the upload of the values. */
static int factorize_prologue (cholmod_hip_plan *P)
{
    hipStream_t st = P->stream ;
    P->winv_valid = false ;                 // the diagonal-block inverses follow the factor
    HIPCHK (hipEventRecord (P->ev0, st)) ;
    // Lx := 0 (several ranks: the rank's own fronts -- its d_Lx holds nothing else); the complete
    // factor a gather may have left in d_Lx_full is stale from here on
    P->full_valid = false ;
    if (!P->d_Lx)
    {
        // (the gather went through the host and released the rank's own array: cholmod_hip_gather_factor)
        if (P->d_Lx_full) { (void) hipFree (P->d_Lx_full) ; P->d_Lx_full = nullptr ; }
        HIPCHK (hipMalloc ((void **) &P->d_Lx, (std::max<i64> (P->lx_local, 1) + UPD3_LX_PAD) * sizeof (double))) ;
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
        if (P->d_agf) HIPCHK (hipMemsetAsync (P->d_agf, 0xFF, (size_t) P->agf_len * sizeof (double), st)) ;
        // (... and the padding behind L that partial tiles of k_update3 may read -- into accumulators nobody stores: a NaN
        // that reaches the factor from there shows in every poisoned parity test; round-5 advisor)
        HIPCHK (hipMemsetAsync (P->d_Lx + std::max<i64> (P->lx_local, 1), 0xFF, (size_t) UPD3_LX_PAD * sizeof (double), st)) ;
    }
    HIPCHK (hipMemsetAsync (P->d_Lx, 0, std::max<i64> (poison ? P->lx_fronts : P->lx_local, 1) * sizeof (double), st)) ;
    HIPCHK (hipMemsetAsync (P->d_info, 0, std::max<i64> (P->nsuper, 1) * sizeof (i32), st)) ;
    HIPCHK (hipMemsetAsync (P->d_tu_cnt, 0, std::max<size_t> (P->sch.tg.size (), 1) * sizeof (i32), st)) ;
    if (P->d_cflags) HIPCHK (hipMemsetAsync (P->d_cflags, 0, (4 * (size_t) P->sch.ncflags + 4) * sizeof (int), st)) ;
    return CHOLMOD_HIP_OK ;
}

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
    int poisoned = CHOLMOD_HIP_OK ;
    bool building_map = false ;
    if (prof) HIPCHK (hipEventRecord (P->evpool [0], st)) ;
    if (!P->prologue_done)
    {
        const int rp = factorize_prologue (P) ;
        if (rp != CHOLMOD_HIP_OK) return rp ;
    }
    P->prologue_done = false ;
    const bool chunked = P->values_chunked ;
    P->values_chunked = false ;
    if (chunked)
    {
        // (cholmod_hip_values_begin: the values arrive chunk by chunk in batch order; the diagonal shift first, every chunk
        // is added into the cleared L right before the first launch of the first batch that needs it)
        if (beta != 0.0 && P->n > 0)
            hipLaunchKernelGGL (k_add_beta<false>, dim3 ((unsigned) ((P->n + 255) / 256)), dim3 (256), 0, st,
                P->n, P->d_supermap, P->d_fr, P->d_Lx, beta) ;
    }
    if (!chunked && P->values_pending)
    {
        // (cholmod_hip_values_commit: the new values of S arrive on the exchange stream; everything above -- the clearing of
        // L -- ran beside their upload, the assembly is the first to read them)
        HIPCHK (hipStreamWaitEvent (st, P->values_ev, 0)) ;
        P->values_pending = false ;
    }
    if (chunked) { }
    else if (P->n > 0 && P->amap_valid)
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
        if (chunked)
        {
            const int rv = apply_value_chunks (P, P->launch_need_chunks [q]) ;
            if (rv != CHOLMOD_HIP_OK) return rv ;
        }
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
    if (chunked)
    {
        // (whatever the schedule did not ask for -- nothing, normally -- so that the resident S is complete)
        const int rv = apply_value_chunks (P, P->nchunks) ;
        if (rv != CHOLMOD_HIP_OK) return rv ;
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
    exchange_volume (P, S) ;
    for (size_t q = 0 ; q < nl ; q++)
    {
        const Launch &L = P->sch.launches [q] ;
        if (L.kind == K_UPD_SMALL) { S [7] += 1 ; S [8] += L.flops ; S [16] += L.bytes ; }
        if (L.kind == K_UPD_W) { S [33] += 1 ; S [34] += L.flops ; S [35] += L.bytes ; }
        if (L.kind == K_UPD_PF) { S [26] += 1 ; S [28] += L.flops ; S [29] += L.bytes ; }
        if (L.kind == K_TRSM_UPD) S [31] += 1 ;
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

static cholmod_hip_plan *plan_create_impl (int64_t n, int64_t nsuper,
    const int64_t *super, const int64_t *pi, const int64_t *px, const int64_t *s,
    int flags, int rank, int world, int *status) ;

cholmod_hip_plan *cholmod_hip_plan_create_dist (int64_t n, int64_t nsuper,
    const int64_t *super, const int64_t *pi, const int64_t *px, const int64_t *s,
    int flags, int rank, int world, int *status)
{
    int st_local ;
    if (!status) status = &st_local ;
    // the plan builder works in C++ containers: their allocation failures end here, as a status, not in the caller's C frames
    try { return plan_create_impl (n, nsuper, super, pi, px, s, flags, rank, world, status) ; }
    catch (const std::bad_alloc &) { *status = CHOLMOD_HIP_OUT_OF_MEMORY ; return nullptr ; }
}

static cholmod_hip_plan *plan_create_impl (int64_t n, int64_t nsuper,
    const int64_t *super, const int64_t *pi, const int64_t *px, const int64_t *s,
    int flags, int rank, int world, int *status)
{
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
    // (whatever way this function is left without handing P over -- a failed step, a std::bad_alloc on its way to the
    // caller's catch -- the plan and what it holds on the device are released)
    struct Guard { cholmod_hip_plan *P ; ~Guard () { if (P) { free_device (P) ; delete P ; } } } guard {P} ;
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
    if (*status != CHOLMOD_HIP_OK) return nullptr ;
    guard.P = nullptr ;
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
                    out [7] = L.kind ; out [8] = L.ar_g0 ; out [9] = L.ar_gn ; out [10] = L.xd.w ; out [11] = L.xd.mb + L.xd.mf ;
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
        if (hipMalloc ((void **) &P->d_Lx, (std::max<i64> (P->lx_local, 1) + UPD3_LX_PAD) * sizeof (double)) == hipSuccess
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
    P->h_vgather.clear () ; P->values_chunked = false ;
    if (!Snz && P->world == 1) P->h_Sp.assign (Sp, Sp + n + 1) ; else P->h_Sp.clear () ;
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
    // the same map in the order the factorization asks for the values: entries of S sorted by the LAUNCH that needs them
    // first -- a thin front's entries by its own launch (the thin-front kernels read their columns of S themselves), a
    // generic front's by the first launch of its batch behind the thin ones (k_values_chunk has then put them into L) --
    // and by source position inside such a class.  One rank, packed S.
    P->h_vgather.clear () ; P->batch_entries_end.clear () ; P->launch_need_chunks.clear () ;
    if (P->d_vorder) { (void) hipFree (P->d_vorder) ; P->d_vorder = nullptr ; }
    static const bool chunked_off = getenv ("CHOLMOD_HIP_VALUES_IN_BATCH_ORDER") && !strcmp (getenv ("CHOLMOD_HIP_VALUES_IN_BATCH_ORDER"), "0") ;
    if (!chunked_off && P->world == 1 && !(P->flags & CHOLMOD_HIP_CX_STORAGE) && (i64) P->h_Sp.size () == P->n + 1 && snz > 0 && snz <= nvalues
        && !P->batch_launch0.empty () && !P->force_shared)
    {
        const size_t nb = P->batch_launch0.size (), nl = P->sch.launches.size () ;
        // need [sf]: the launch before which front sf's entries must have been applied
        std::vector<i32> need ((size_t) std::max<i64> (P->nsuper, 1), -1) ;
        std::vector<i32> generic_launch (nb, 0) ;
        bool okn = true ;
        for (size_t b = 0 ; b < nb ; b++)
        {
            const size_t q0 = P->batch_launch0 [b], q1 = (b + 1 < nb) ? P->batch_launch0 [b + 1] : nl ;
            size_t q = q0 ;
            for ( ; q < q1 && P->sch.launches [q].kind == K_SMALL ; q++)
            {
                const Launch &Lq = P->sch.launches [q] ;
                for (size_t e = Lq.goff ; e < Lq.goff + (size_t) Lq.ng ; e++) need [P->sch.sm [e]] = (i32) q ;
            }
            generic_launch [b] = (i32) std::min (q, q1 > 0 ? q1 - 1 : 0) ;
        }
        for (i64 sf = 0 ; sf < P->nsuper && okn ; sf++)
        {
            const i32 b = P->batch_of [sf] ;
            if (b < 0 || (size_t) b >= nb) { okn = false ; break ; }
            if (need [sf] < 0) need [sf] = generic_launch [b] ;
        }
        if (okn && nl > 0)
        {
            // buckets by need, filled in source order on a few threads: the caller's staging copy (stage [k] = values
            // [index [k]]) then walks its value array front to back once per class -- sequential reads with gaps -- and the
            // scatter to S's order is left to the device, where it costs nothing.  (S's own order made that copy a random
            // gather: 1.7 ms for the 4.75 M entries of the 2D stand-in against 0.4 ms for a plain copy.)
            std::vector<i64> order ((size_t) snz) ;
            P->h_vgather.resize ((size_t) snz) ;
            std::vector<i64> cnt (nl + 1, 0) ;
            {
                std::vector<i32> needq ((size_t) snz) ;
                std::vector<i64> inv ((size_t) nvalues, -1) ;
                const int T = (int) std::max<i64> (1, std::min<i64> (8, snz / 200000)) ;
                auto par = [&] (auto fn) {
                    std::vector<std::thread> th ;
                    for (int t = 1 ; t < T ; t++) th.emplace_back (fn, t) ;
                    fn (0) ;
                    for (auto &x : th) x.join () ;
                } ;
                par ([&] (int t) {
                    const i64 s0 = P->nsuper * t / T, s1 = P->nsuper * (t + 1) / T ;
                    for (i64 sf = s0 ; sf < s1 ; sf++)
                        for (i64 q = P->h_Sp [P->super [sf]] ; q < P->h_Sp [P->super [sf + 1]] ; q++) { needq [q] = need [sf] ; inv [src [q]] = q ; }
                }) ;
                std::vector<std::vector<i64>> tcnt (T, std::vector<i64> (nl, 0)) ;
                par ([&] (int t) {
                    const i64 a0 = nvalues * t / T, a1 = nvalues * (t + 1) / T ;
                    for (i64 a = a0 ; a < a1 ; a++) if (inv [a] >= 0) tcnt [t][needq [inv [a]]]++ ;
                }) ;
                for (size_t q = 0 ; q < nl ; q++)
                {
                    i64 off = cnt [q] ;
                    for (int t = 0 ; t < T ; t++) { const i64 c = tcnt [t][q] ; tcnt [t][q] = off ; off += c ; }
                    cnt [q + 1] = off ;
                }
                par ([&] (int t) {
                    const i64 a0 = nvalues * t / T, a1 = nvalues * (t + 1) / T ;
                    for (i64 a = a0 ; a < a1 ; a++)
                    {
                        const i64 q = inv [a] ;
                        if (q < 0) continue ;
                        const i64 k = tcnt [t][needq [q]]++ ;
                        order [k] = q ; P->h_vgather [k] = a ;
                    }
                }) ;
            }
            HIPCHK (hipMalloc ((void **) &P->d_vorder, (size_t) snz * sizeof (i64))) ;
            HIPCHK (hipMemcpy (P->d_vorder, order.data (), (size_t) snz * sizeof (i64), hipMemcpyHostToDevice)) ;
            P->nchunks = (long) ((snz + P->chunk_len - 1) / P->chunk_len) ;
            // chunks a launch needs = those that hold the entries of every class up to its own
            P->launch_need_chunks.assign (nl + 1, (i32) P->nchunks) ;
            for (size_t q = 0 ; q < nl ; q++) P->launch_need_chunks [q] = (i32) ((cnt [q + 1] + P->chunk_len - 1) / P->chunk_len) ;
            while ((long) P->chunk_ev.size () < P->nchunks)
            {
                hipEvent_t e ; HIPCHK (hipEventCreateWithFlags (&e, hipEventDisableTiming)) ; P->chunk_ev.push_back (e) ;
            }
        }
    }
    return CHOLMOD_HIP_OK ;
}

/* Round 6 -- the value upload in the order the factorization needs it.  cholmod_hip_values_gather_index: where staged
 * position k comes from in the caller's value array (NULL: no batch order for this plan -- use staging / push / commit as
 * before) and the chunk length.  The caller then runs three roles on threads of its own:
 *   * cholmod_hip_values_begin (plan): the clearing of L is enqueued at once (it runs beside the upload), the chunk
 *     counters are reset; then cholmod_hip_factorize_resident, whose launch loop waits -- host side for the push, device
 *     side for the copy -- only for the chunks the next batch needs, applies them (k_values_chunk) and goes on;
 *   * staging: host_buffer [k] = values [index [k]], chunk by chunk;
 *   * cholmod_hip_values_push_chunk (plan, c) for c = 0, 1, ... as chunks are staged (one thread, in order);
 *     cholmod_hip_values_push_chunk (plan, -1) if the staging failed: the waiting factorization returns an error.
 * Nothing of this touches the resident S before the factorization applies it, chunk by chunk. */
int cholmod_hip_values_gather_index (cholmod_hip_plan *P, const int64_t **index, int64_t *chunk_len, int64_t *count)
{
    if (!P || !index || !chunk_len || !count) return CHOLMOD_HIP_INVALID ;
    const bool ok = !P->h_vgather.empty () && (i64) P->h_vgather.size () == P->s_cur_nz && P->vsrc_nz == P->s_cur_nz
        && P->amap_valid && P->d_vorder && P->launch_need_chunks.size () == P->sch.launches.size () + 1 ;
    *index = ok ? P->h_vgather.data () : nullptr ;
    *chunk_len = P->chunk_len ;
    *count = P->s_cur_nz ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_values_begin (cholmod_hip_plan *P)
{
    if (!P || !P->h_vals || P->h_vgather.empty () || !P->amap_valid) return CHOLMOD_HIP_INVALID ;
    P->chunks_pushed.store (0) ;
    P->chunks_applied = 0 ;
    P->values_chunked = true ;
    P->values_pending = false ;
    const int rc = factorize_prologue (P) ;
    if (rc != CHOLMOD_HIP_OK) { P->values_chunked = false ; return rc ; }
    P->prologue_done = true ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_values_push_chunk (cholmod_hip_plan *P, int64_t c)
{
    if (!P || !P->h_vals) return CHOLMOD_HIP_INVALID ;
    if (c < 0 || c >= P->nchunks) { P->chunks_pushed.store (-1) ; return CHOLMOD_HIP_INVALID ; }
    const i64 k0 = c * P->chunk_len, k1 = std::min<i64> (k0 + P->chunk_len, P->s_cur_nz) ;
    if (hipMemcpyAsync (P->d_vals + k0, P->h_vals + k0, (size_t) (k1 - k0) * sizeof (double), hipMemcpyHostToDevice, P->stream2) != hipSuccess
        || hipEventRecord (P->chunk_ev [c], P->stream2) != hipSuccess)
    {
        (void) hipGetLastError () ;
        P->chunks_pushed.store (-1) ;
        return CHOLMOD_HIP_GPU_PROBLEM ;
    }
    P->chunks_pushed.store ((long) c + 1, std::memory_order_release) ;
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

/* The same, pipelined (round 5: the API step of the small configurations is a third H2D + host work): the caller copies
 * A->x chunk by chunk into a pinned staging buffer of the plan (its own threads, cholmod_hip_values_staging) and pushes every
 * chunk as soon as it is filled (cholmod_hip_values_push: DMA on the exchange stream, no host wait); while the last chunks
 * travel it does what else it has to do (the pattern hash), then commits (gather into the resident S, an event the next
 * factorization's ASSEMBLY waits for -- its clearing of L, 1 to 10 GB of memset, runs beside the upload) or cancels. */
int cholmod_hip_values_staging (cholmod_hip_plan *P, double **host_buffer, int64_t *nvalues)
{
    if (!P || P->host_only || !host_buffer || !nvalues || !P->d_vsrc || P->vsrc_nz != P->s_cur_nz) return CHOLMOD_HIP_INVALID ;
    if (!P->h_vals || P->h_vals_n != P->vals_n)
    {
        if (P->h_vals) { (void) hipHostFree (P->h_vals) ; P->h_vals = nullptr ; }
        if (hipHostMalloc ((void **) &P->h_vals, (size_t) std::max<i64> (P->vals_n, 1) * sizeof (double), hipHostMallocDefault) != hipSuccess)
        {
            (void) hipGetLastError () ;
            P->h_vals = nullptr ;
            return CHOLMOD_HIP_OUT_OF_MEMORY ;
        }
        P->h_vals_n = P->vals_n ;
    }
    if (!P->values_ev) HIPCHK (hipEventCreateWithFlags (&P->values_ev, hipEventDisableTiming)) ;
    *host_buffer = P->h_vals ; *nvalues = P->vals_n ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_values_push (cholmod_hip_plan *P, int64_t offset, int64_t count)
{
    if (!P || !P->h_vals || offset < 0 || count < 0 || offset + count > P->vals_n) return CHOLMOD_HIP_INVALID ;
    if (count) HIPCHK (hipMemcpyAsync (P->d_vals + offset, P->h_vals + offset, (size_t) count * sizeof (double), hipMemcpyHostToDevice, P->stream2)) ;
    return CHOLMOD_HIP_OK ;
}

int cholmod_hip_values_commit (cholmod_hip_plan *P, int commit)
{
    if (!P || !P->h_vals) return CHOLMOD_HIP_INVALID ;
    if (!commit)
    {
        // (the pattern turned out to be another one: nothing of the staged values may reach the resident S, and whoever
        // rewrites S next must not meet a copy in flight)
        HIPCHK (hipStreamSynchronize (P->stream2)) ;
        P->values_pending = false ;
        return CHOLMOD_HIP_OK ;
    }
    if (P->vsrc_nz)
        hipLaunchKernelGGL (k_gather_values, dim3 ((unsigned) ((P->vsrc_nz + 255) / 256)), dim3 (256), 0, P->stream2,
            P->vsrc_nz, P->d_vsrc, P->d_vals, P->d_Sx) ;
    HIPCHK (hipGetLastError ()) ;
    HIPCHK (hipEventRecord (P->values_ev, P->stream2)) ;
    P->values_pending = true ;
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

/* Smallest and largest diagonal entry of the resident factor, and the number of NaN / negative ones: what
 * cholmod_l_rcond needs (reference CHOLMOD/Cholesky/cholmod_rcond.c:102-124 walks Lx [psx + jj + jj nsrow] on the host;
 * here L stays in HBM and three numbers come back).  Several ranks: after cholmod_hip_gather_factor. */
int cholmod_hip_diag_minmax (cholmod_hip_plan *P, double *out3)
{
    if (!P || P->host_only || !out3) return CHOLMOD_HIP_INVALID ;
    out3 [0] = out3 [1] = out3 [2] = 0 ;
    if (P->n == 0) return CHOLMOD_HIP_OK ;
    if (!whole_factor (P) || !whole_fronts (P)) return CHOLMOD_HIP_INVALID ;
    { int rc = ensure_check_tasks (P) ; if (rc != CHOLMOD_HIP_OK) return rc ; }     // (d_chk_out: five doubles of scratch)
    unsigned long long seed [3] ;
    const double inf = INFINITY, zero = 0.0 ;
    memcpy (&seed [0], &inf, 8) ; memcpy (&seed [1], &zero, 8) ; seed [2] = 0 ;
    HIPCHK (hipMemcpyAsync (P->d_chk_out, seed, sizeof (seed), hipMemcpyHostToDevice, P->stream)) ;
    const bool cxs = (P->flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;
    CXS_LAUNCH (k_diag_minmax, dim3 ((unsigned) ((P->n + 255) / 256)), dim3 (256), 0, P->stream,
        P->n, P->d_supermap, whole_fronts (P), whole_factor (P), (unsigned long long *) P->d_chk_out) ;
    HIPCHK (hipGetLastError ()) ;
    HIPCHK (hipMemcpyAsync (seed, P->d_chk_out, sizeof (seed), hipMemcpyDeviceToHost, P->stream)) ;
    HIPCHK (hipStreamSynchronize (P->stream)) ;
    memcpy (&out3 [0], &seed [0], 8) ; memcpy (&out3 [1], &seed [1], 8) ; out3 [2] = (double) seed [2] ;
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
    exchange_volume (P, P->stats) ;
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
        const i64 v [] = {L.kind, L.grid, L.ng, (i64) L.goff, L.stream, L.wait_ev, L.rec_ev, L.ar_g0, L.ar_gn, L.aux, L.leaf_T, L.leaf_pw, L.ndiag, L.half,
            L.xd.slab, L.xd.lda, L.xd.w, L.xd.mb, L.xd.R, L.xd.g, L.xd.r, L.xd.fo, L.xd.mf, L.xd.Rf, L.far} ;
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
    HIPCHK (hipMalloc ((void **) &P.d_Lx, (nsrow * nscol + UPD3_LX_PAD) * sizeof (double))) ;
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
    P.d_dg = nullptr ; P.d_rg = nullptr ; P.d_dinv = nullptr ; P.d_cg = nullptr ; P.d_cflags = nullptr ;
    return CHOLMOD_HIP_OK ;
}

} // extern "C"
