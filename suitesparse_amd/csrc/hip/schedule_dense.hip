// schedule_dense.hip -- the launches of the dense partial factorization of one batch of fronts (all of one etree level, or
// one front of the top part): two-level blocked right-looking Cholesky of the first nscol columns of every front
// [panel | contribution block].  Host code only: it appends groups and launches to a Schedule, engine.hip runs them.
//
//   outer block columns (OB = 512 ... 4096 columns, by the front's row count) closed by one K = OB update of everything to
//   their right, contribution block included; inside one, a chain of 64-column steps (dpotrf of the diagonal block, dtrsm of
//   the rows below, fused with their neighbours where the step allows: k_update2f, k_trsm_upd) with recursive-doubling
//   trailing updates -- or the 256-column chain (k_chainf; the default on fronts shared between ranks).
//   Reference: the dpotrf / dtrsm per supernode of CHOLMOD/Supernodal/t_cholmod_super_numeric.c:864-867, :997-1002 and the
//   dsyrk / dgemm updates of :682-717, regrouped.
//
// Several ranks: a front shared by a rank group has its panel distributed by column slabs; the outer block
// column being factored lives in a window on every member, is summed by a reduce-scatter by row chunks before its chain and
// gathered after it (emit_rs / emit_ag), the next one opened and summed ahead of time beside the rest of the outer update.
#include "plan.hip.h"

namespace sship {
namespace {

// One trailing-update step of a front: columns [kc, kc + kk) update the in-front columns [t0, t1) (all rows from t0 down)
// and, if cb, the contribution block.  Steps with kk >= MB are `wide`: their tiles are dealt over the rank group of a shared
// front and they feed the exchange look-ahead.
struct Upd { int q, kc, kk, t0, t1 ; bool cb, wide ; } ;

struct DenseScheduler
{
    // ---- what the caller hands over
    const std::vector<FrontD> &fr ;
    const i32 *ids ; const int nf ;
    Schedule &S ;
    const int flags ;
    const i32 *owner, *grp0, *grpn ;
    const int rank, world ;
    const char *assign_cb ;         // front's contribution block is written (not updated) by its first outer update
    const i64 *win ;                // window of a distributed front (offset in the rank's L), -1 / nullptr: none
    const i32 *child ;              // the rank's child lists (pricing of the extend-add into a window)
    const bool allow_half ;         // one GPU: launches of a few thousand tiles may run two waves per tile (below)
    // ---- derived once per batch
    // The real twin of a complex factor (phi embedding, host/complex.c): every row / column pair (2i, 2i+1) is (re, im) of
    // one complex row, the odd columns of a panel are the rotations of the even ones.  The update kernels then contract over
    // the EVEN columns only -- column stride 2 nsrow, K / 2 -- and rebuild the 2 x 2 blocks from the four real products in
    // the lanes (kernels.hip.h: update_tile / update_tile_w, TW): half the flops of the embedding.  A complex factor in its
    // own storage (CHOLMOD_HIP_CX_STORAGE; kernels.hip.h: ldcx / stcx): the index space is still the twin's, but only its
    // even columns exist -- column c of a front or of a contribution block lives at (c >> 1) ld, the panels ARE their even
    // columns (operand stride ld, K / 2 contraction steps).
    bool cx, twin, use_big ;
    ObThresholds obt ;
    int maxnscol = 0, maxrows = 0 ;
    // A region goes to k_update3 (one wave per tile: 75 TFLOP/s at K = 4096 against 64 for the four-wave k_update2, 58
    // against 49 at K = 512; measured, tools/upd3.py) when it has enough tiles to put two waves on every SIMD; below that
    // the four waves per tile of k_update2 fill the chip better.  CHOLMOD_HIP_UPD3_MIN_TILES overrides (0 = never).
    // (every knob is read per batch: tests and in-process A/B runs change them between plans)
    i64 w_min_tiles ;
    // Round 5: HALF tiles -- two waves per 64 x 64 tile, 64 x 32 each (kernels.hip.h: update_tile_w, NQ = 1).  A launch of T tiles
    // fills the chip's 2048 wave slots T / 2048 times; with one or two and a bit fillings the last one is mostly empty (a
    // triangular 4 488^2 region at K = 1024, 2 556 tiles: 48 TFLOP/s), and below 2048 tiles the four-wave kernel was all there
    // was (35 - 50).  Measured standalone (tools/upd3.py half, profiles/r05_upd3_half_tiles.json): half tiles win from ~500 to
    // ~8 000 tiles at every K >= 256 (64.5 against 48.3 / 58.3 for whole tiles / four-wave tiles at 2 556 tiles, 45.8 against
    // 32.2 / 41.1 at 780), whole tiles from ~12 000 on (operand traffic per flop is 1.5 x).  So: a one-wave-per-tile launch of
    // fewer than w_half_max tiles runs in half tiles, and the regions of a flush go to it from w_half_min pooled tiles on
    // instead of 2048.  CHOLMOD_HIP_UPD3_HALF_MAX=0: off (whole tiles from 2048 on, as in rounds 3-4).
    i64 w_half_min, w_half_max ;
    int w_min_k ;                   // shortest contraction a pooled region may have
    i64 unfuse_tiles ;              // a chain update of this many tiles is not fused with the next dpotrf
    bool by_launch ;                // ... pooled over the regions of a launch (CHOLMOD_HIP_UPD3_BY_LAUNCH=0: by region only)
    bool one_region ;               // tuning (CHOLMOD_HIP_UPDW_ONE_REGION=1): every region of a k_update3 launch a launch of its own
    bool swz16 ;                    // tuning (CHOLMOD_HIP_SWZ16=1): 16 x 16 super-tiles for the one-wave-per-tile walk
    bool xla ;                      // exchange look-ahead (several ranks)
    bool chain256 = false ;         // the 256-column chain instead of the 64-column one
    bool fused256 = true ;          // ... as ONE launch per sub-block (k_chainf) rather than k_diag + k_rowsolve
    bool fuse_potrf = false, fuse_trsm = false ;
    bool balance_cb ;
    // ---- state of the batch
    std::vector<GemmGroup> big, small ;     // regions collected for the next flush: 128 x 128 tiles (opt-in) / 64 x 64
    std::vector<GemmGroup> pfv ;            // narrow updates whose first tile is factored on the spot (k_update2f)
    std::vector<GemmGroup> wav ;            // regions big enough for one wave per 64 x 64 tile (k_update3)
    std::vector<int> early ;                // block column of front q already summed ahead of time
    std::vector<int> early_open ;           // block column of front q opened (window) ahead of time
    std::vector<int> pf_done ;              // column whose diagonal block a fused update / solve has factored
    std::vector<Upd> step ;

    DenseScheduler (const std::vector<FrontD> &fr_, const i32 *ids_, int nf_, Schedule &S_, int flags_, const i32 *owner_,
        const i32 *grp0_, const i32 *grpn_, int rank_, int world_, const char *assign_cb_, const i64 *win_, const i32 *child_,
        bool allow_half_)
        : fr (fr_), ids (ids_), nf (nf_), S (S_), flags (flags_), owner (owner_), grp0 (grp0_), grpn (grpn_), rank (rank_),
          world (world_), assign_cb (assign_cb_), win (win_), child (child_), allow_half (allow_half_)
    {
        cx = (flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;
        twin = (flags & CHOLMOD_HIP_PHI_TWIN) != 0 || cx ;
        use_big = (flags & CHOLMOD_HIP_TILE128) != 0 && !twin ;
        obt = outer_block_thresholds () ;
        for (int q = 0 ; q < nf ; q++)
        {
            maxnscol = std::max (maxnscol, fr [ids [q]].nscol) ;
            maxrows = std::max (maxrows, fr [ids [q]].nsrow) ;
        }
        { const char *e = getenv ("CHOLMOD_HIP_UPD3_MIN_TILES") ; w_min_tiles = e ? (i64) atoll (e) : (i64) 2048 ; }
        by_launch = true ;
        { const char *e = getenv ("CHOLMOD_HIP_UPD3_HALF_MAX") ; w_half_max = e ? (i64) atoll (e) : (i64) 10240 ; }      // (=0: whole tiles only, kept for the schedule fingerprints that cover that form)
        // (measured flat within the noise and fixed: pooled regions from 256 / 512 / 1024 tiles on, shortest pooled contraction
        // 256 / 128 / 64, chain updates un-fused from 2048 ... 256 tiles on -- profiles/r05_ab_upd3_half_knobs.log)
        w_half_min = 512 ; w_min_k = 256 ; unfuse_tiles = w_min_tiles ;
        if (!allow_half || w_min_tiles <= 0) w_half_max = 0 ;
        if (w_half_max <= 0 || w_half_min > w_min_tiles) w_half_min = w_min_tiles ;
        one_region = false ;
        swz16 = false ;
        xla = !(flags & CHOLMOD_HIP_NO_EXCHANGE_LOOKAHEAD) ;
        balance_cb = !use_big && !getenv ("CHOLMOD_HIP_NO_CB_BALANCE") ;
        early.assign (nf, -1) ; early_open.assign (nf, -1) ; pf_done.assign (nf, -1) ;
        choose_chain () ;
    }

    // ---- geometry ------------------------------------------------------------------------------------------------------
    i64 co (int c, i64 ld) const { return cx ? (i64) (c >> 1) * ld : (i64) c * ld ; }
    void twin_operands (GemmGroup &G, int origin, int kc) const
    {
        // (everything is even in a doubled structure; a plan that claims to be a twin and is not would silently drop a column)
        if ((origin | kc | G.k | G.m | G.n | G.lda | G.ldc) & 1) { fprintf (stderr, "cholmod_hip: twin plan with an odd region\n") ; abort () ; }
        if (!cx) G.lda *= 2 ;
        G.k /= 2 ;
    }
    // Outer block width: a property of the FRONT (its row count), not of the batch -- the ranks of a multi-GPU group see
    // different batches around the same shared front and must cut its updates into the same regions.
    int ob_of (const FrontD &f) const { return front_ob (f, flags, obt) ; }
    // (owner [] < 0 only occurs with world > 1, or in the single-rank self test CHOLMOD_HIP_SHARE_AS_WORLD that drives the
    // exchange path with one rank)
    bool is_shared (int fid) const { return owner && owner [fid] < 0 ; }
    // A distributed front (several GPUs: win [fid] >= 0) has its panel stored by column slabs on their owners; the outer
    // block column being factored lives in a window of nsrow x OB doubles (two of them, used alternately), addressed as if
    // the whole front were there: psx_at (fid, c) is the base to use for anything that touches column c of the front during
    // the panel chain of c's outer block.
    bool windowed (int fid) const { return win && win [fid] >= 0 ; }
    i64 psx_at (int fid, int col) const
    {
        const FrontD &f = fr [fid] ;
        if (!windowed (fid)) return f.psx ;
        int OBq = ob_of (f), ob = col / OBq ;
        return win [fid] + (i64) (ob & 1) * window_len (f, OBq) - (i64) ob * OBq * f.nsrow ;
    }
    static i64 region_tiles (const GemmGroup &G)
    {
        i64 mt = (G.m + SMALL - 1) / SMALL, nt = (G.n + SMALL - 1) / SMALL ;
        return G.tri ? nt * (nt + 1) / 2 + (mt - nt) * nt : mt * nt ;
    }
    static i64 tri_tiles (int rows, int cols)
    {
        i64 mt = (rows + SMALL - 1) / SMALL, nt = (cols + SMALL - 1) / SMALL ;
        return nt * (nt + 1) / 2 + (mt - nt) * nt ;
    }
    // The exchange of the block column [c0, c1) of shared front q: geometry of its row chunks (descriptors.hip.h: XchgD).
    // This rank keeps the diagonal block, its chunk of the NEAR rows [c1, o1) -- the rest of the outer block column: the
    // later block columns of the outer block need them as an operand -- and its chunk of the FAR rows [o1, nsrow), which are
    // cut the same way for every block column of the outer block.
    XchgD xchg_of (int q, int c0) const
    {
        const FrontD &f = fr [ids [q]] ;
        int c1 = std::min (c0 + MB, f.nscol) ;
        int g = grpn [ids [q]], r = rank - grp0 [ids [q]] ;
        if (world == 1) { g = 1 ; r = 0 ; }                 // (single-rank self test of the exchange path)
        const int OBq = ob_of (f), o1 = std::min ((c0 / OBq) * OBq + OBq, (int) f.nscol) ;
        auto chunk = [g] (int rows) { return rows > 0 ? (((rows + g - 1) / g) + 15) / 16 * 16 : 0 ; } ;
        const int mb = o1 - c1, mf = f.nsrow - o1 ;
        return XchgD {psx_at (ids [q], c0) + c0 + (i64) c0 * f.nsrow, f.nsrow, c1 - c0, mb, chunk (mb), g, r, o1 - c0, mf, chunk (mf)} ;
    }
    // rows of a shared front's block column (the one that holds column i0) this rank works on below a sub-block that ends
    // at column b1: the rest of the 512-wide diagonal block (every member), the rank's chunk of the near rows and its chunk
    // of the far rows
    void chunk_rows (int q, int i0, int b1, int lo [3], int hi [3]) const
    {
        const FrontD &f = fr [ids [q]] ;
        lo [0] = b1 ; hi [0] = f.nsrow ; lo [1] = hi [1] = lo [2] = hi [2] = 0 ;
        if (!is_shared (ids [q])) return ;
        const int b0 = (i0 / MB) * MB ;
        XchgD X = xchg_of (q, b0) ;
        hi [0] = b0 + X.w ;
        lo [1] = b0 + X.w + X.r * X.R ; hi [1] = std::min (lo [1] + X.R, b0 + X.w + X.mb) ;
        lo [2] = b0 + X.fo + X.r * X.Rf ; hi [2] = std::min (lo [2] + X.Rf, (int) f.nsrow) ;
        for (int p = 1 ; p < 3 ; p++) if (hi [p] < lo [p]) hi [p] = lo [p] ;
    }
    int record_last ()
    {
        if (S.launches.size () == 0) return -1 ;
        if (S.launches.back ().rec_ev < 0) S.launches.back ().rec_ev = S.nevents++ ;
        return S.launches.back ().rec_ev ;
    }

    // ---- update regions ------------------------------------------------------------------------------------------------
    GemmGroup blank_region (int fid) const
    {
        GemmGroup G ;
        memset (&G, 0, sizeof (G)) ;
        G.front = fid ; G.tile_mul = 1 ; G.tile_add = 0 ;
        return G ;
    }
    // target region: rows r0.., cols r0.. of the front (starts on the diagonal), or its contribution block
    void add_update (const FrontD &f, int fid, int r0, int kc, int kk, int m, int ncols, bool to_cb,
        bool split = false, bool factor_first = false)
    {
        if (m <= 0 || ncols <= 0 || kk <= 0) return ;
        GemmGroup G = blank_region (fid) ;
        G.a_off = psx_at (fid, kc) + r0 + co (kc, f.nsrow) ;
        G.b_off = G.a_off ;
        G.lda = f.nsrow ;
        if (to_cb) { G.c_off = f.cb ; G.ldc = f.ncb ; G.c_in_cb = 1 ; }
        else { G.c_off = psx_at (fid, r0) + r0 + co (r0, f.nsrow) ; G.ldc = f.nsrow ; }
        G.m = m ; G.n = ncols ; G.k = kk ; G.tri = 1 ;
        if (twin) twin_operands (G, r0, kc) ;
        // first update of a contribution block nobody zeroed: C = -A*B'
        G.assign = (to_cb && kc == 0 && assign_cb && assign_cb [fid]) ? 1 : 0 ;
        if (split && is_shared (fid)) { G.tile_mul = grpn [fid] ; G.tile_add = rank - grp0 [fid] ; }
        if (factor_first) { G.pf_next = 1 ; G.pf_col0 = r0 ; pfv.push_back (G) ; return ; }
        bool isbig = use_big && ncols >= BIG && m >= 2 * BIG ;
        (isbig ? big : small).push_back (G) ;
    }
    // The outer update (K = OB, operands in the window of outer block kc / OB) of the in-front columns [ca, cb) of a
    // distributed front: owner-computes -- this rank updates the slabs it stores, in place, every one a region of its own
    // that starts on the diagonal.
    void add_outer_slabs (const FrontD &f, int fid, int kc, int kk, int ca, int cb)
    {
        if (kk <= 0) return ;
        for (int c0 = (ca / f.own_w) * f.own_w ; c0 < cb ; c0 += f.own_w)
        {
            if (!col_owned (f, c0)) continue ;
            int a = std::max (c0, ca), b = std::min ({c0 + f.own_w, cb, (int) f.nscol}) ;
            if (b <= a) continue ;
            GemmGroup G = blank_region (fid) ;
            G.a_off = psx_at (fid, kc) + a + co (kc, f.nsrow) ;
            G.b_off = G.a_off ;
            G.lda = f.nsrow ;
            G.c_off = f.psx + a + (i64) col_local (f, a) * f.nsrow ; G.ldc = f.nsrow ;
            G.m = f.nsrow - a ; G.n = b - a ; G.k = kk ; G.tri = 1 ;
            if (twin) twin_operands (G, a, kc) ;
            small.push_back (G) ;
        }
    }
    // rows [lo, hi) x columns [c0, c1) of a front's window, K = [kc, kc + kk): a rectangle below the diagonal
    void add_rows_update (int fid, int lo, int hi, int c0, int c1, int kc, int kk)
    {
        if (hi <= lo || c1 <= c0 || kk <= 0) return ;
        const FrontD &f = fr [fid] ;
        GemmGroup G = blank_region (fid) ;
        const i64 wpsx = psx_at (fid, kc) ;
        G.a_off = wpsx + lo + co (kc, f.nsrow) ;
        G.b_off = wpsx + c0 + co (kc, f.nsrow) ;
        G.c_off = wpsx + lo + co (c0, f.nsrow) ;
        G.lda = f.nsrow ; G.ldc = f.nsrow ;
        G.m = hi - lo ; G.n = c1 - c0 ; G.k = kk ; G.tri = 0 ;
        if (twin) twin_operands (G, (lo | c0), kc) ;
        small.push_back (G) ;
    }
    // a narrow update inside a block column of a shared front: its rows have been dealt to the ranks of the group
    // (reduce-scatter by row chunks, emit_rs) -- the rows of the diagonal block (every rank) and this rank's chunks below
    void add_chunk_update (const Upd &x, int c0, bool ff)
    {
        const FrontD &f = fr [ids [x.q]] ;
        int lo [3], hi [3] ;
        chunk_rows (x.q, x.kc, c0, lo, hi) ;
        add_update (f, ids [x.q], c0, x.kc, x.kk, hi [0] - c0, x.t1 - c0, false, false, ff) ;
        for (int p = 1 ; p < 3 ; p++) add_rows_update (ids [x.q], lo [p], hi [p], c0, x.t1, x.kc, x.kk) ;
    }
    // a wide update (K >= 512) between the block columns of an outer block of a shared front, columns [c0, c1): the near
    // rows (down to the end of the outer block column; every member has them since the in-line all-gathers of the K
    // columns) by tiles dealt over the group, the far rows CHUNK-LOCALLY -- every member the rows it has solved itself,
    // so that nobody needs the others' far rows before the outer update (their all-gather runs beside the chain)
    void add_wide_shared (const Upd &x, int c0, int c1)
    {
        const FrontD &f = fr [ids [x.q]] ;
        const int fid = ids [x.q], OBq = ob_of (f), o1 = std::min ((x.kc / OBq) * OBq + OBq, (int) f.nscol) ;
        add_update (f, fid, c0, x.kc, x.kk, o1 - c0, c1 - c0, false, true) ;
        int lo [3], hi [3] ;
        chunk_rows (x.q, x.kc, c0, lo, hi) ;
        add_rows_update (fid, lo [2], hi [2], c0, c1, x.kc, x.kk) ;
    }
    // a distributed contribution block: this rank's block of columns, one region that starts on the diagonal; the first
    // outer block assigns (nothing else ever writes there)
    void add_distributed_cb (const Upd &x)
    {
        const FrontD &f = fr [ids [x.q]] ;
        if (f.cb_hi <= f.cb_lo) return ;
        const int a = f.nscol + f.cb_lo, b = f.nscol + f.cb_hi ;
        GemmGroup G = blank_region (ids [x.q]) ;
        G.a_off = psx_at (ids [x.q], x.kc) + a + co (x.kc, f.nsrow) ;
        G.b_off = G.a_off ;
        G.lda = f.nsrow ;
        G.c_off = f.cb + f.cb_lo ; G.ldc = f.ncb ; G.c_in_cb = 1 ; G.assign = (x.kc == 0) ? 1 : 0 ;
        G.m = f.nsrow - a ; G.n = b - a ; G.k = x.kk ; G.tri = 1 ;
        if (twin) twin_operands (G, a, x.kc) ;
        small.push_back (G) ;
    }
    // (layout of the first half of round 4, CHOLMOD_HIP_NO_CB_PASSTHROUGH) The contribution block of a windowed front as
    // partial sums on every member: the members' own slabs of this step differ (a member has a slab more, or taller ones),
    // so the tiles of the block -- anybody may compute any of them -- are dealt so that every member ends up with the same
    // number of tiles: member r takes the range of 64-tile chunks [lo, lo + cnt) that fills it up to the common level.
    void balance_cb_tiles (const Upd &x)
    {
        const FrontD &f = fr [ids [x.q]] ;
        const int g = f.own_g ;
        std::vector<double> tin (g, 0.0) ;
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

    // ---- collected regions -> launches ---------------------------------------------------------------------------------
    // tile grid of a region and the blocks this rank spends on it (kernels.hip.h: decode_tile); returns 0 if none.
    // `tiles` = blocks of the launch so far (XCD alignment of a swizzled group)
    i64 place_region (GemmGroup &G, int T, bool wave_tiles, i64 &tiles) const
    {
        G.mt = (G.m + T - 1) / T ; G.nt = (G.n + T - 1) / T ;
        i64 cnt = G.tri ? (i64) G.nt * (G.nt + 1) / 2 + (i64) (G.mt - G.nt) * G.nt : (i64) G.mt * G.nt ;
        G.ntiles = (i32) cnt ;
        i64 mine ;
        G.swz = 0 ;
        if (G.tile_mul == 1 && cnt < 1024 && !G.tile_cnt) return cnt ;
        i64 nch = (cnt + 63) / 64 ;
        i64 mych = nch > G.tile_add ? (nch - G.tile_add + G.tile_mul - 1) / G.tile_mul : 0 ;
        if (G.tile_cnt) mych = std::min<i64> (G.tile_cnt, nch > G.tile_add ? nch - G.tile_add : 0) ;     // a range of chunks
        mine = mych * 64 ;
        G.swz = !(flags & CHOLMOD_HIP_NO_XCD_SWIZZLE) && mine >= 1024 ;
        // one wave per tile: an XCD runs 256 tiles at a time, so a 16 x 16 super-tile (32 operand panels per 256 tiles) is
        // one XCD's load.  Opt-in (CHOLMOD_HIP_SWZ16=1): standalone on a triangular 49 152^2 region, K = 4096, it is 74.6
        // against 74.4 TFLOP/s and 200 against 220 GB fetched, inside the 200^3 factorization 105.4 against 105.1 ms per
        // launch and 478 against 462 GB (same box, top-48 launches): the tiles of an XCD drift apart in k either way, and
        // the wider strip only widens what they drift over.
        if (G.swz && wave_tiles && G.tile_mul == 1 && !G.tile_cnt && cnt >= 8192 && swz16)
        {
            G.swz = 2 ;
            mine = (cnt + 255) / 256 * 256 ;
        }
        if (G.swz) tiles = (tiles + 7) / 8 * 8 ;     // keep block % 8 == XCD aligned
        else if (G.tile_mul == 1 && !G.tile_cnt) mine = cnt ;
        return mine ;
    }
    // one kind of update launch out of the regions collected in v (kind: K_UPD_BIG / _SMALL / _PF / _W)
    void flush_kind (std::vector<GemmGroup> &v, int kind)
    {
        if (v.empty ()) return ;
        const int T = kind == K_UPD_BIG ? BIG : SMALL ;
        Launch L {kind, 0, (int) v.size (), S.gg.size (), 0, 0} ;
        i64 tiles = 0 ;
        auto close_launch = [&] ()
        {
            L.ng = (int) (S.gg.size () - L.goff) ;
            L.grid = (int) tiles ;
            L.half = (kind == K_UPD_W && tiles < w_half_max) ? 1 : 0 ;
            if (L.ng) S.launches.push_back (L) ;
            L = Launch {L.kind, 0, 0, S.gg.size (), 0, 0} ;
            tiles = 0 ;
        } ;
        // (One wave per tile keeps its L2 reuse only while the waves of a super-tile run in step -- DESIGN section 9 item 3: what
        // broke the step were the partial tiles' slow loads, fixed in the kernel; re-ordering the regions of a launch, squares
        // ahead of trapezoids, or giving the big ones launches of their own then measured +-0.3 % and is not done.)
        for (auto &G : v)
        {
            if (one_region && kind == K_UPD_W && S.gg.size () > L.goff) close_launch () ;
            const i64 mine = place_region (G, T, kind == K_UPD_W, tiles) ;
            if (mine == 0) continue ;
            const i64 cnt = G.ntiles ;
            G.nblk = (i32) mine ;
            G.tile_start = (i32) tiles ;
            tiles += mine ;
            double elems = G.tri ? (double) G.n * (G.n + 1) / 2 + (double) (G.m - G.n) * G.n : (double) G.m * G.n ;
            double share = (G.tile_mul == 1 && !G.tile_cnt) ? 1.0 : std::min (1.0, (double) mine / (double) cnt) ;
            L.flops += 2.0 * elems * G.k * share ;
            L.aux = std::max (L.aux, (int) G.k) ;
            L.bytes += ((G.assign ? 8.0 : 16.0) * elems + 8.0 * ((double) G.m + G.n) * G.k) * share ;
            S.gg.push_back (G) ;
        }
        close_launch () ;
        v.clear () ;
    }
    void flush_updates ()
    {
        if (w_min_tiles > 0 && !use_big)
        {
            // (a factored-first update of a big region: the diagonal block is factored by a separate launch instead -- 17 us
            // next to milliseconds)
            // ... and what fills the chip is the LAUNCH, not the region: the regions of many mid-size fronts of one level (a
            // few hundred tiles each, K >= 256) together are tens of thousands of tiles -- they go with the big ones when
            // their sum reaches the threshold.  Below K = 256 the update is bound by the read-modify-write of C and the two
            // kernels are on par.
            i64 pooled = 0 ;
            if (by_launch) for (auto &G : small) if (G.k >= w_min_k || region_tiles (G) >= w_min_tiles) pooled += region_tiles (G) ;
            std::vector<GemmGroup> keep ;
            for (auto &G : small)
            {
                // (w_half_min == w_min_tiles unless half tiles are on: then the pooled regions go from 512 tiles on)
                const bool w = region_tiles (G) >= w_min_tiles || (by_launch && pooled >= w_half_min && G.k >= w_min_k) ;
                if (w) wav.push_back (G) ; else keep.push_back (G) ;
            }
            small.swap (keep) ;
        }
        flush_kind (big, K_UPD_BIG) ;
        flush_kind (small, K_UPD_SMALL) ;
        flush_kind (pfv, K_UPD_PF) ;
        flush_kind (wav, K_UPD_W) ;
    }

    // ---- exchange and windows of the shared fronts -----------------------------------------------------------------------
    void emit_rs (int q, int c0, int wait_ev)
    {
        Launch La {K_XCHG_RS, 0, 0, 0, 0, 0} ;
        La.xd = xchg_of (q, c0) ;
        La.bytes = 8.0 * ((double) La.xd.w * La.xd.w + ((double) La.xd.R + La.xd.Rf) * La.xd.w) * La.xd.g ;
        La.ar_g0 = grp0 [ids [q]] ; La.ar_gn = grpn [ids [q]] ;
        La.wait_ev = wait_ev ;
        S.launches.push_back (La) ;
    }
    // The solved chunks of block column c0 of shared front q back to everybody.  The near chunks in line: the next block
    // column's updates need those rows as an operand.  The far chunks are needed by the outer update only (the updates
    // between block columns are chunk-local there, add_wide_shared): their gather is put aside (`pending`) and issued on
    // the exchange stream behind the block column's chain -- after the next block column's reduce-scatter, which the chain
    // is waiting for -- so that the main stream runs the following block columns beside it and meets it at join_far_gathers.
    // Without the exchange look-ahead (CHOLMOD_HIP_NO_EXCHANGE_LOOKAHEAD): both in line.
    struct FarGather { int q, c0, ev ; } ;
    std::vector<FarGather> pending ;
    int far_ev = -1 ;                   // event behind the last far gather issued on the exchange stream, not joined yet
    Launch ag_launch (int q, int c0, int far) const
    {
        Launch La {K_XCHG_AG, 0, 0, 0, 0, 0} ;
        La.xd = xchg_of (q, c0) ;
        La.far = far ;
        La.bytes = 8.0 * (double) (far ? La.xd.Rf : La.xd.R) * La.xd.w * La.xd.g ;
        La.ar_g0 = grp0 [ids [q]] ; La.ar_gn = grpn [ids [q]] ;
        return La ;
    }
    void emit_ag (int q, int c0)
    {
        Launch Ln = ag_launch (q, c0, 0), Lf = ag_launch (q, c0, 1) ;
        if (Ln.xd.R > 0) S.launches.push_back (Ln) ;
        if (Lf.xd.Rf == 0) return ;                         // nothing below the outer block column
        if (xla) pending.push_back (FarGather {q, c0, -1}) ;
        else S.launches.push_back (Lf) ;
    }
    void issue_far_gathers ()
    {
        for (const FarGather &p : pending)
        {
            Launch Lf = ag_launch (p.q, p.c0, 1) ;
            Lf.stream = 1 ; Lf.wait_ev = p.ev ;
            S.launches.push_back (Lf) ;
            far_ev = -2 ;
        }
        if (far_ev == -2) far_ev = record_last () ;
        pending.clear () ;
    }
    // the main stream needs the far rows of every block column gathered so far: the outer update, the window's way home
    void join_far_gathers ()
    {
        issue_far_gathers () ;
        if (far_ev < 0) return ;
        Launch Lj {K_JOIN, 0, 0, 0, 0, 0} ;
        Lj.wait_ev = far_ev ;
        S.launches.push_back (Lj) ;
        far_ev = -1 ;
    }
    // Window of a distributed front.  open (mode 0): the block columns of [ca, cb) -- the owners' stored columns, zero
    // elsewhere (k_win_move), then the contributions of this rank's children to those columns (extend-add into the window):
    // the rank's partial sum, ready for the reduce-scatter.  On stream 1 behind an event when the block column is opened
    // ahead of time (exchange look-ahead).  close (mode 1): the factored columns of outer block [ca, cb) into the owners' slabs.
    void emit_win (int q, int mode, int ca, int cb, int stream, int wait_ev)
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
        if (mode != 0 || f.child_end == f.child_begin) return ;
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
    // The chain enters a new 512-column block column at i0: a distributed front entering a new OUTER block column gets its
    // block columns into the window (all but the one opened ahead of time), and every shared front's block column that has
    // not been summed ahead of time is summed now (per-rank partial sums: the extend-adds of the rank's own subtrees + its
    // share of the earlier wide update tiles; only rows >= i0 carry data, they are packed into a staging buffer).
    void enter_block_column (int i0)
    {
        if (i0 % MB != 0) return ;
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
    // ... and leaves the sub-block [i0, i0 + W): a block column of a shared front that is complete on the rows of its owners
    // is gathered on every rank of the group before anything uses it as an operand; the outer block column of a distributed
    // front that is complete in its window goes into the owners' slabs
    void leave_sub_block (int i0, int W)
    {
        const size_t np = pending.size () ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0 || !is_shared (ids [q])) continue ;
            int b0 = (i0 / MB) * MB ;
            if (i0 + W >= std::min (b0 + MB, (int) f.nscol)) emit_ag (q, b0) ;
        }
        if (pending.size () > np)
        {
            // (the event behind the chain of these block columns and their in-line gathers, on the main stream)
            const int ev = record_last () ;
            for (size_t p = np ; p < pending.size () ; p++) pending [p].ev = ev ;
        }
        bool closing = false ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0 || !windowed (ids [q])) continue ;
            int OBq = ob_of (f), o0 = (i0 / OBq) * OBq, o1 = std::min (o0 + OBq, (int) f.nscol) ;
            if (i0 + W < o1) continue ;
            if (!closing) join_far_gathers () ;
            closing = true ;
            emit_win (q, 1, o0, o1, 0, -1) ;
        }
    }

    // ---- one trailing-update step of the batch -----------------------------------------------------------------------------
    // Exchange look-ahead (several ranks): the update that completes the NEXT 512-column block column of a shared front is
    // issued first (U_next), the rest of the trailing update (U_rest) right behind it, and the block column's exchange then
    // runs on the second stream while U_rest keeps the chip busy.  Returns the event behind U_next, -1 if there is none.
    int emit_next_columns_first ()
    {
        bool any_next = false ;
        if (xla)
            for (const Upd &x : step)
            {
                if (!x.wide || !is_shared (ids [x.q]) || x.t1 <= x.t0) continue ;
                const FrontD &f = fr [ids [x.q]] ;
                int tn = std::min (x.t0 + MB, x.t1) ;
                if (x.cb && windowed (ids [x.q])) add_outer_slabs (f, ids [x.q], x.kc, x.kk, x.t0, tn) ;
                else if (x.cb) add_update (f, ids [x.q], x.t0, x.kc, x.kk, f.nsrow - x.t0, tn - x.t0, false, true) ;
                else add_wide_shared (x, x.t0, tn) ;
                any_next = true ;
            }
        if (!any_next) return -2 ;
        flush_updates () ;
        return record_last () ;
    }
    // The fused "update + dpotrf of the next diagonal block" (k_update2f) saves a 17 us launch and runs the update in the
    // four-wave kernel.  Where the K >= 512 chain updates of the fronts of this step are a matrix-core-sized piece of work
    // together (4096 tiles: the mid-size fronts of one level, each below the per-region threshold), the update goes to
    // k_update3 and the diagonal blocks to a dpotrf launch of their own -- as a single big region does.
    bool unfuse_wide_k () const
    {
        i64 pooled = 0 ;
        if (by_launch && fuse_potrf && w_min_tiles > 0 && !use_big)
            for (const Upd &x : step)
            {
                const FrontD &f = fr [ids [x.q]] ;
                if (is_shared (ids [x.q]) || x.cb || x.kk < 512 || f.nscol - x.t0 < NB || x.t1 - x.t0 < NB) continue ;
                pooled += tri_tiles (f.nsrow - x.t0, x.t1 - x.t0) ;
            }
        return pooled >= 2 * unfuse_tiles ;
    }
    void emit_step ()
    {
        for (const Upd &x : step) if (x.cb && is_shared (ids [x.q])) { join_far_gathers () ; break ; }
        const int ev_next = emit_next_columns_first () ;
        const bool any_next = ev_next != -2 ;
        const bool ff_unfuse_wide_k = unfuse_wide_k () ;
        for (const Upd &x : step)
        {
            const FrontD &f = fr [ids [x.q]] ;
            const int fid = ids [x.q] ;
            int c0 = x.t0 ;
            bool ahead = any_next && x.wide && is_shared (fid) && x.t1 > x.t0 ;
            if (ahead) c0 = std::min (x.t0 + MB, x.t1) ;
            // a narrow update of the panel chain ends on the next diagonal block: its first tile is that block, and the
            // workgroup that updates it factors it (k_update2f) (also the K >= 512 doubling updates inside an outer block,
            // unless their tiles are dealt over the ranks of a shared front: the factor must exist on every rank)
            bool ff = fuse_potrf && !(x.wide && is_shared (fid)) && !x.cb && c0 == x.t0 && f.nscol - x.t0 >= NB && x.t1 - c0 >= NB ;
            if (ff && w_min_tiles > 0 && !use_big && !is_shared (fid))
            {
                // a region big enough for k_update3 is not fused with the next dpotrf
                if (tri_tiles (f.nsrow - c0, x.t1 - c0) >= unfuse_tiles) ff = false ;
                if (ff_unfuse_wide_k && x.kk >= 512) ff = false ;
            }
            if (ff) pf_done [x.q] = x.t0 ;
            if (!x.wide && is_shared (fid) && x.t1 > c0) { add_chunk_update (x, c0, ff) ; continue ; }
            if (x.wide && is_shared (fid) && !x.cb) { if (x.t1 > c0) add_wide_shared (x, c0, x.t1) ; continue ; }
            // the outer update of a distributed front: its in-front columns slab by slab on their owners
            if (x.cb && windowed (fid)) { if (x.t1 > c0) add_outer_slabs (f, fid, x.kc, x.kk, c0, x.t1) ; }
            else if (x.t1 > c0) add_update (f, fid, c0, x.kc, x.kk, f.nsrow - c0, x.t1 - c0, false, x.wide, ff) ;
            if (x.cb && f.cbd) add_distributed_cb (x) ;
            else if (x.cb)
            {
                size_t nsm = small.size () ;
                add_update (f, fid, f.nscol, x.kc, x.kk, f.ncb, f.ncb, true, x.wide) ;
                if (windowed (fid) && f.own_g > 1 && balance_cb && small.size () > nsm) balance_cb_tiles (x) ;
            }
        }
        flush_updates () ;
        if (any_next)
            for (const Upd &x : step)
            {
                if (!x.wide || !is_shared (ids [x.q]) || x.t1 <= x.t0) continue ;
                if (x.cb && windowed (ids [x.q]))
                {
                    // the first block column of the next outer block: into its window ahead of time, on the exchange
                    // stream behind the update that completed it
                    emit_win (x.q, 0, x.t0, std::min (x.t0 + MB, x.t1), 1, ev_next) ;
                    early_open [x.q] = x.t0 ;
                }
                emit_rs (x.q, x.t0, ev_next) ;
                early [x.q] = x.t0 ;
            }
        issue_far_gathers () ;
        step.clear () ;
    }
    // Trailing updates after the sub-block [i0, i0 + W) of every front that still has columns there (`skip [q]`: the step
    // was taken by a fused launch).  Inside an outer block column of the front (OB columns, ob_of): recursive doubling --
    // with e sub-blocks of it factored and p the largest power of two dividing e, the last p sub-blocks (K = W p) update the
    // next p only; every column block is then read-modified-written log2 times instead of once per step (768 instead of 1792
    // column sweeps per 512 columns at W = 64), with K up to OB / 2 on the matrix cores.  When the outer block column (or the
    // front) is complete: one K = OB update of everything to its right, contribution block included.
    void push_doubling_steps (int i0, int W, const std::vector<char> *skip)
    {
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0 || (skip && (*skip) [q])) continue ;
            int OBq = ob_of (f) ;
            int o0 = (i0 / OBq) * OBq ;
            int o1 = std::min (o0 + OBq, (int) f.nscol) ;
            if (i0 + W >= o1)
            {
                step.push_back (Upd {q, o0, o1 - o0, o1, f.nscol, true, true}) ;
                continue ;
            }
            int e = (i0 - o0) / W + 1 ;
            int p = e & -e ;
            int t0 = o0 + e * W ;
            int t1 = std::min (o0 + (e + p) * W, o1) ;
            int kc = o0 + (e - p) * W ;
            step.push_back (Upd {q, kc, t0 - kc, t0, t1, false, p * W >= MB}) ;
        }
        emit_step () ;
    }

    // ---- which chain ---------------------------------------------------------------------------------------------------------
    void choose_chain ()
    {
        // (not with complex storage: the 256-column kernels have no complex form)
        chain256 = (flags & CHOLMOD_HIP_CHAIN256) != 0 && !cx ;
        // CHOLMOD_HIP_CHAINF_AUTO=1 (tuning): the fused 256-column chain (k_chainf) for the batches it is measured to win on
        // -- fronts of at least 192 columns and at most 16 384 rows (one round of row workgroups), none shared between ranks
        if (!chain256 && !cx && !twin && getenv ("CHOLMOD_HIP_CHAINF_AUTO") && atoi (getenv ("CHOLMOD_HIP_CHAINF_AUTO")) != 0 && !(flags & CHOLMOD_HIP_NO_FUSED_POTRF))
        {
            bool any_shared = false ;
            for (int q = 0 ; q < nf ; q++) if (is_shared (ids [q])) any_shared = true ;
            chain256 = !any_shared && maxnscol >= 192 && maxrows <= 16384 ;
        }
        // A batch that holds a front shared between ranks takes the fused 256-column chain by default: the chain of a shared
        // front is the part of a rank's work that does not shrink with the number of ranks, and its 64-column form has no
        // fused kernels there (the diagonal blocks are replicated, the rows dealt by chunks: dpotrf, dtrsm and the narrow
        // updates are separate launches, ~45 us per 64 columns against ~28 in k_chainf).  CHOLMOD_HIP_SHARED_CHAIN64=1: the
        // 64-column chain.
        bool any_win = false ;
        for (int q = 0 ; q < nf ; q++) if (windowed (ids [q])) any_win = true ;
        if (any_win)
        {
            const bool c64 = getenv ("CHOLMOD_HIP_SHARED_CHAIN64") || getenv ("CHOLMOD_HIP_NO_CHAINF") || cx || (flags & CHOLMOD_HIP_NO_FUSED_POTRF) ;
            chain256 = !c64 ;
        }
        fused256 = !getenv ("CHOLMOD_HIP_NO_CHAINF") ;
        fuse_potrf = !(flags & CHOLMOD_HIP_NO_FUSED_POTRF) && !chain256 ;     // (the 256-column chain has no separate dpotrf launches to fuse)
        fuse_trsm = fuse_potrf && !(flags & CHOLMOD_HIP_NO_FUSED_TRSM) ;
    }

    // ---- the panel chain in 256-column sub-blocks (kernels.hip.h: k_chainf, or k_diag / k_rowsolve) -----------------------
    // Per sub-block [i0, b1) of a front: the diagonal sub-block is factored and every row below it solved -- in ONE launch
    // (k_chainf: the diagonal sub-block spread over up to four workgroups that hand their row block of L on through flags),
    // or by one workgroup (k_diag) and a solve launch (k_rowsolve; CHOLMOD_HIP_NO_CHAINF).  A front shared between ranks: the
    // diagonal sub-block on every rank, below it the rest of the 512-wide diagonal block (every rank) and this rank's chunk.
    void emit_chainf (int i0)
    {
        const int SB = DG_W ;
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
            int lo [3], hi [3] ;
            chunk_rows (q, i0, b1, lo, hi) ;
            const int m1 = hi [0] - b1, m2 = hi [1] - lo [1], m3 = hi [2] - lo [2] ;
            S.cg.push_back (CfGroup {psx_at (ids [q], i0) + i0 + (i64) i0 * f.nsrow, f.nsrow, w, ids [q], i0, m1, slot, S.ncflags++, dblocks, bblocks,
                m2 ? lo [1] - i0 : 0, m2, m3 ? lo [2] - i0 : 0, m3, 0}) ;
            dblocks += (w + 63) / 64 ;
            bblocks += (m1 + 63) / 64 + (m2 + 63) / 64 + (m3 + 63) / 64 ;
            wmax = std::max (wmax, w) ;
            Lc.flops += (double) w * w * w / 3.0 + (double) (m1 + m2 + m3) * w * w ;
            Lc.bytes += 16.0 * (m1 + m2 + m3) * w ;
        }
        Lc.ng = (int) (S.cg.size () - Lc.goff) ; Lc.grid = dblocks + bblocks ; Lc.ndiag = dblocks ; Lc.aux = wmax ;
        S.max_dinv_slots = std::max (S.max_dinv_slots, Lc.ng) ;
        if (Lc.ng) S.launches.push_back (Lc) ;
    }
    void emit_diag_rowsolve (int i0)
    {
        const int SB = DG_W ;
        Launch Ld {K_DIAG, 0, 0, S.dg.size (), 0, 0} ;
        Launch Lr {K_ROWSOLVE, 0, 0, S.rg.size (), 0, 0} ;
        int rblocks = 0 ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0) continue ;
            int OBq = ob_of (f) ;
            int o0 = (i0 / OBq) * OBq ;
            int o1 = std::min (o0 + OBq, (int) f.nscol) ;
            int b1 = std::min (i0 + SB, o1) ;
            int w = b1 - i0 ;
            int slot = (int) (S.dg.size () - Ld.goff) ;
            S.dg.push_back (DgGroup {psx_at (ids [q], i0) + i0 + (i64) i0 * f.nsrow, f.nsrow, w, ids [q], i0, slot, 0}) ;
            Ld.flops += (double) w * w * w / 3.0 ;
            int lo [3], hi [3] ;
            chunk_rows (q, i0, b1, lo, hi) ;
            for (int part = 0 ; part < 3 ; part++)
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
    }
    void run_chain256 ()
    {
        for (int i0 = 0 ; i0 < maxnscol ; i0 += DG_W)
        {
            enter_block_column (i0) ;
            if (fused256) emit_chainf (i0) ; else emit_diag_rowsolve (i0) ;
            leave_sub_block (i0, DG_W) ;
            push_doubling_steps (i0, DG_W, nullptr) ;
        }
    }

    // ---- the panel chain in 64-column steps --------------------------------------------------------------------------------
    // dpotrf of the diagonal blocks nobody has factored yet (a fused update or solve may have: pf_done)
    void emit_potrf (int i0)
    {
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
        if (Lp.ng) S.launches.push_back (Lp) ;
    }
    // Fronts whose step is "solve, K = 64 update of the next 64 columns, factor the next diagonal block" (every other step
    // of the doubling schedule) take all three in one launch (k_trsm_upd): a full panel, a full next block inside the same
    // outer block column, the front not shared between ranks.
    void emit_trsm_upd (int i0, std::vector<char> &fused)
    {
        if (!fuse_trsm) return ;
        Launch Lf_ {K_TRSM_UPD, 0, 0, S.tg.size (), 0, 0} ;
        int fblocks = 0 ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol < i0 + 2 * NB || is_shared (ids [q])) continue ;
            int OBq = ob_of (f) ;
            int o0 = (i0 / OBq) * OBq ;
            int o1 = std::min (o0 + OBq, (int) f.nscol) ;
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
        if (Lf_.ng) S.launches.push_back (Lf_) ;
    }
    // dtrsm of the rows below the diagonal block -- of a shared front, whose block column has been dealt to the ranks by row
    // chunks: the rest of the 512-wide diagonal block (every rank of the group) and this rank's chunk below it
    void emit_trsm (int i0, const std::vector<char> &fused)
    {
        Launch Lt {K_TRSM, 0, 0, S.tg.size (), 0, 0} ;
        int blocks = 0 ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = fr [ids [q]] ;
            if (f.nscol <= i0 || fused [q]) continue ;
            int nb = std::min (NB, f.nscol - i0) ;
            int lo [3], hi [3] ;
            chunk_rows (q, i0, i0 + nb, lo, hi) ;
            for (int part = 0 ; part < 3 ; part++)
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
        if (Lt.ng) S.launches.push_back (Lt) ;
    }
    void run_chain64 ()
    {
        for (int i0 = 0 ; i0 < maxnscol ; i0 += NB)
        {
            enter_block_column (i0) ;
            emit_potrf (i0) ;
            std::vector<char> fused (nf, 0) ;
            emit_trsm_upd (i0, fused) ;
            emit_trsm (i0, fused) ;
            leave_sub_block (i0, NB) ;
            push_doubling_steps (i0, NB, &fused) ;
        }
    }

    void run ()
    {
        if (chain256) run_chain256 () ; else run_chain64 () ;
        join_far_gathers () ;       // (nothing left by now: the last block column of a shared front ends with a join)
    }
} ;

} // namespace

void schedule_dense (const std::vector<FrontD> &fr, const i32 *ids, int nf,
    Schedule &S, int flags, const i32 *owner, const i32 *grp0, const i32 *grpn, int rank, int world,
    const char *assign_cb, const i64 *win, const i32 *child, bool allow_half)
{
    DenseScheduler D (fr, ids, nf, S, flags, owner, grp0, grpn, rank, world, assign_cb, win, child, allow_half) ;
    D.run () ;
}

} // namespace sship
