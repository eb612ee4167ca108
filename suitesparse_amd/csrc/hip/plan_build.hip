// plan_build.hip -- everything a rank derives from the symbolic factor on the host, once per symbolic factor: the
// supernodal etree and its levels, who owns which front (proportional mapping), the rank's packed layout of L
// (own subtrees, its column slabs of the shared fronts, windows), the child lists / contribution routing, the batches in
// which fronts are factored (memory-aware), the contribution-block arena, the solve tasks, and -- through schedule_dense.hip
// -- the launch list.  Host code only, deterministic: every rank of a partition runs it on the same maps and must arrive at
// the same global decisions (tests/test_dist.py, tests/test_schedule_fingerprint.py).
#include "plan.hip.h"

namespace sship {
namespace {

struct PlanBuilder
{
    cholmod_hip_plan *P ;
    const i64 n, nsuper ;
    int nlev = 0 ;
    // full child lists of the etree (tree order): children of s = call [cptr [s] .. cptr [s+1])
    std::vector<i32> cptr, call ;
    std::vector<double> wsub ;              // flops of the subtree rooted at s
    std::vector<i32> st_stack ;
    // several ranks
    bool distribute = false, passthru = false, cx_storage = false ;
    int ownw = 128 ;
    std::vector<i64> cb_len_all ;           // length of every front's contribution block in the layout over ALL fronts
    std::vector<std::vector<i32>> rel_list ;    // who releases whose contribution block
    std::vector<std::vector<i32>> batches ;

    explicit PlanBuilder (cholmod_hip_plan *P_) : P (P_), n (P_->n), nsuper (P_->nsuper) {}

    bool mine (i64 s) const { return P->rank >= P->grp0 [s] && P->rank < P->grp0 [s] + P->grpn [s] ; }
    // every member of the subtree rooted at `root`, through the real child lists: supernodes need not be numbered in etree
    // postorder (Common->postorder = FALSE, arbitrary maps handed to cholmod_hip_plan_create), so a subtree is not an index
    // range in general -- only parent > child is guaranteed
    template <typename F> void for_subtree (i32 root, F &&fn)
    {
        st_stack.clear () ;
        st_stack.push_back (root) ;
        while (!st_stack.empty ())
        {
            i32 t = st_stack.back () ; st_stack.pop_back () ;
            fn (t) ;
            for (i32 c = cptr [t] ; c < cptr [t+1] ; c++) st_stack.push_back (call [c]) ;
        }
    }
    static double own_flops (const FrontD &f)
    {
        double c = f.nscol, r = f.ncb ;
        return c * c * c / 3.0 + r * c * c + r * r * c ;
    }
    // member q's block of a distributed contribution block (ncb columns, group of g) starts where q / g of the lower
    // triangle's area lies to its left (multiples of 64); a function of (ncb, g, q) only: every rank can evaluate it for
    // every member of every group
    static int cb_bound (int ncb, int g, int q)
    {
        if (q <= 0) return 0 ;
        if (q >= g) return ncb ;
        const double T = 0.5 * (double) ncb * (ncb + 1) * q / g ;
        // area left of column j: j ncb - j (j - 1) / 2
        double j = ncb + 0.5 - std::sqrt (std::max (0.0, (ncb + 0.5) * (ncb + 0.5) - 2.0 * T)) ;
        int b = (int) (j / 64.0 + 0.5) * 64 ;
        return std::min (std::max (b, 0), ncb) ;
    }
    i64 cb_len (const FrontD &f) const
    {
        // (a distributed block: this rank's block of columns; a complex front in its own storage: the even columns of the
        // twin's square)
        if (f.cbd) return (i64) f.ncb * (f.cb_hi - f.cb_lo) ;
        return f.cbp == 1 ? (i64) f.ncb * (f.ncb + 1) / 2 : cx_storage ? (i64) f.ncb * (f.ncb / 2) : (i64) f.ncb * f.ncb ;
    }

    // ---- fronts, etree, levels (reference t_cholmod_super_numeric.c:1025: Sparent = SuperMap [Ls [pi [s] + nscol]]) -------
    int init_fronts ()
    {
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
        return CHOLMOD_HIP_OK ;
    }
    int build_etree ()
    {
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
        cptr.assign (nsuper + 1, 0) ; call.assign (std::max<i64> (nsuper, 1), 0) ;
        for (i64 s = 0 ; s < nsuper ; s++) cptr [s+1] = cptr [s] + nchild [s] ;
        {
            std::vector<i32> pos (cptr.begin (), cptr.end () - 1) ;
            for (i64 s = 0 ; s < nsuper ; s++)
            {
                i32 p = P->fr [s].parent ;
                if (p >= 0) call [pos [p]++] = (i32) s ;
            }
        }
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
        wsub.assign (nsuper, 0.0) ;
        P->exec_flops = 0 ;
        for (i64 s = 0 ; s < nsuper ; s++)
        {
            double own = own_flops (P->fr [s]) ;
            P->exec_flops += own ;
            wsub [s] += own ;
            if (P->fr [s].parent >= 0) wsub [P->fr [s].parent] += wsub [s] ;
        }
        return CHOLMOD_HIP_OK ;
    }

    // ---- ownership: proportional mapping of the supernodal etree ------------------------------------------
    // A front is *shared* while its subtree outweighs 1/(6 world) of the whole factorization; a shared front belongs to a
    // contiguous group of ranks [grp0, grp0+grpn) (the root's group is everybody).  Where the heavy children of a shared
    // front can split its group in proportion to their weights without unbalancing it (<= 10 % above the mean) they get
    // disjoint sub-groups -- a sub-group of one rank owns the child's whole subtree -- otherwise they inherit the parent's
    // group.  The light subtrees hanging off the shared region are dealt, largest first, to the least loaded rank of their
    // parent's group (LPT).  Every rank derives the same map.
    struct Solo { double w ; i32 root, g0, gn ; } ;
    struct Item { i32 t, g0, gn ; } ;
    // largest-remainder apportionment of gn ranks over the heavy children (at least one each); false if it would put a
    // sub-group more than split_tol above the mean
    bool apportion (const std::vector<i32> &heavy, int gn, double split_tol, std::vector<i32> &cnt) const
    {
        const int nh = (int) heavy.size () ;
        double wh = 0 ;
        for (i32 h : heavy) wh += wsub [h] ;
        int left = gn ;
        std::vector<double> rem (nh) ;
        for (int q = 0 ; q < nh ; q++)
        {
            double x = gn * wsub [heavy [q]] / wh ;
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
        return (left == 0) && worst <= split_tol * wh / gn ;
    }
    // top-down over the shared region (explicit stack; roots get everybody): marks the shared fronts with their groups,
    // collects the solo subtrees hanging off them
    void map_shared_region (int share_world, double thr, std::vector<char> &shared, std::vector<double> &load, std::vector<Solo> &solo)
    {
        const bool subgroups = !getenv ("CHOLMOD_HIP_NO_SUBGROUPS") ;
        double split_tol = 1.10 ;
        if (const char *e = getenv ("CHOLMOD_HIP_SPLIT_TOL")) if (atof (e) >= 1.0) split_tol = atof (e) ;
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
            for (int q = it.g0 ; q < it.g0 + it.gn ; q++) load [q] += own_flops (P->fr [it.t]) / it.gn ;
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
            if (split) split = apportion (heavy, it.gn, split_tol, cnt) ;
            // children are pushed so that they pop in the apportionment order
            int g = it.g0 + it.gn ;
            for (int q = nh ; q-- > 0 ; )
            {
                if (split) { g -= cnt [q] ; stack.push_back (Item {heavy [q], (i32) g, cnt [q]}) ; }
                else stack.push_back (Item {heavy [q], it.g0, it.gn}) ;
            }
        }
    }
    void assign_ownership ()
    {
        P->owner.assign (std::max<i64> (nsuper, 1), 0) ;
        P->assign_cb.assign (std::max<i64> (nsuper, 1), 0) ;
        P->grp0.assign (std::max<i64> (nsuper, 1), 0) ;
        P->grpn.assign (std::max<i64> (nsuper, 1), 1) ;
        // Self test of the exchange path on one GPU: CHOLMOD_HIP_SHARE_AS_WORLD=k with world == 1 marks the fronts a k-rank
        // run would share, so the pack / all-reduce callback / unpack launches run (summing over the single rank).
        int share_world = P->world ;
        if (P->world == 1)
        {
            const char *e = getenv ("CHOLMOD_HIP_SHARE_AS_WORLD") ;
            if (e && atoi (e) > 1) share_world = atoi (e) ;
        }
        P->force_shared = (P->world == 1 && share_world > 1) ;
        if (share_world <= 1 || nsuper <= 0) return ;
        double total = 0 ;
        for (i64 s = 0 ; s < nsuper ; s++) if (P->fr [s].parent < 0) total += wsub [s] ;
        // (1 / (6 world): with 1 / (4 world) two subtrees of 6 % each stayed atomic at four ranks on Poisson 200^3 and left
        // one rank 7.7 % above the mean; now within 1.2 %)
        double thr_div = 6.0 ;
        const double thr = total / (thr_div * share_world) ;
        std::vector<char> shared (nsuper, 0) ;
        std::vector<double> load (share_world, 0.0) ;
        std::vector<Solo> solo ;
        map_shared_region (share_world, thr, shared, load, solo) ;
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

    // ---- the rank's own L: the fronts it holds, packed in supernode order (see cholmod_hip_plan::lpx) ----------------------
    // (a shared front: only the column slabs this rank owns; with one rank -- the self test of the exchange path -- that is
    // every slab, and the front stays where the reference layout has it)
    void layout_local_factor ()
    {
        distribute = !getenv ("CHOLMOD_HIP_NO_DISTRIBUTED_FRONTS") && !(P->flags & CHOLMOD_HIP_CX_STORAGE) ;
        ownw = (P->flags & CHOLMOD_HIP_PHI_TWIN) ? std::max (own_width (), 64) : own_width () ;
        // The contribution block of a shared front distributed like its panel (the slabs continue past column nscol, the
        // outer updates of a slab run on its owner), and NOTHING extend-added into it: the contributions of a rank's fronts
        // to a shared ancestor are routed past the blocks in between, straight into the ancestor whose PANEL holds the column
        // -- every entry travels once, no member keeps a full square of partial sums, no replicated extend-add.
        // CHOLMOD_HIP_NO_CB_PASSTHROUGH=1: the layout of the first half of round 4 (full squares of partial sums, pulled
        // level by level).
        passthru = distribute && !getenv ("CHOLMOD_HIP_NO_CB_PASSTHROUGH") ;
        P->passthru = passthru ;
        cx_storage = (P->flags & CHOLMOD_HIP_CX_STORAGE) != 0 ;
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
                    f.cbd = 1 ;
                    f.cb_lo = cb_bound (f.ncb, f.own_g, f.own_r) ; f.cb_hi = cb_bound (f.ncb, f.own_g, f.own_r + 1) ;
                }
            }
            P->lx_local += cols * f.nsrow ;
        }
        if (P->world == 1) P->lx_local = P->xsize ;
        P->lx_fronts = P->lx_local ;
        // thin fronts (fused LDS-resident kernel): their contribution blocks are packed lower triangles, the generic
        // fronts' full squares
        for (i64 s = 0 ; s < nsuper ; s++)
        {
            FrontD &f = P->fr [s] ;
            f.cbp = (!(P->flags & CHOLMOD_HIP_NO_SMALL_FRONTS) && f.nsrow <= SM_MAX && !(P->owner [s] < 0)) ? 1 : 0 ;
            // (a complex front in its own storage: thin all the same, its block the even columns of a square like everybody's: 2)
            if (f.cbp && (P->flags & CHOLMOD_HIP_CX_STORAGE)) f.cbp = 2 ;
        }
        // The same length in the layout over ALL fronts, from which the batch split is chosen: every rank must derive the
        // same number for every front, whether it holds the front or not -- so nothing here may read f.cbd / f.own_g /
        // f.cb_lo (set for the fronts of THIS rank only; round-4 advisor item: a member of a sub-group counted ncb^2 / g, a
        // non-member ncb^2, and `8 A.top <= budget` could pick different splits on different ranks).  A distributed block
        // counts as its LARGEST member share, ncb * max_q (cb_hi - cb_lo): what the neediest member really allocates.
        cb_len_all.assign (std::max<i64> (nsuper, 1), 0) ;
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
    }

    // ---- child lists and contribution routing -------------------------------------------------------------------------------
    void build_child_lists ()
    {
        // who releases whose block: the parent, once it has pulled it -- or, for a front whose parent is shared and whose
        // contributions are routed to the ancestors' panels, the root of its tree (it contributes until then)
        rel_list.assign (std::max<i64> (nsuper, 1), {}) ;
        for (i64 s = 0 ; s < nsuper ; s++)
        {
            i32 p = P->fr [s].parent ;
            if (p < 0) continue ;
            i32 t = p ;
            if (passthru && P->owner [p] < 0) while (P->fr [t].parent >= 0) t = P->fr [t].parent ;
            rel_list [t].push_back ((i32) s) ;
        }
        // this rank's view of the child lists: a shared parent pulls only the contribution blocks this rank computed (its
        // own subtrees and its partial copies of shared children); the other ranks add theirs on their side and the sums
        // meet in the exchange of the parent's block columns
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
    }

    // ---- execution order and contribution-block arena ------------------------------------------------------------------------
    // A *batch* = fronts factored together (one set of launches); a CB lives from its batch until the batch of its parent.
    // The plain order is one batch per etree level (all fronts of equal height at once): best batching, but every CB of two
    // adjacent levels is alive at the same time (Poisson 200^3: 178 GB).  When that does not fit next to L, the tree is cut
    // into `nsplit` subtrees that are swept one after the other (level by level inside each), followed by the top part: the
    // live set shrinks to one subtree's working set + the finished subtree roots + the top levels, at the price of more,
    // smaller launches low in the tree.  nsplit doubles until the arena fits the budget.
    void nominal_budget ()
    {
        if (P->arena_budget >= 0) return ;
        // Several ranks must derive the same batch order, so the budget is nominal, not the momentary free memory: what a
        // 288 GB part has left next to the LARGEST part of L any rank of this partition holds (its subtrees, its slabs of the
        // shared fronts; every rank computes all of them), the index maps and a margin for windows and staging.
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
    // the batches of a sweep over `nsplit` subtrees; returns the number of subtree groups it found
    int make_batches (int nsplit)
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
        // batches in postorder: a subtree group contributes its levels in increasing height when its root is reached, a
        // front of the top part is a batch of its own right after its last descendant (supernodes are numbered in postorder,
        // so "increasing index of the unit's last member" is a valid order and frees every contribution block as early as
        // possible)
        std::vector<std::vector<std::vector<i32>>> by (ngroups, std::vector<std::vector<i32>> (nlev)) ;
        std::vector<i32> group_last (ngroups, -1) ;
        for (i64 s = 0 ; s < nsuper ; s++)
            if (group [s] >= 0) { by [group [s]][P->level [s]].push_back ((i32) s) ; group_last [group [s]] = (i32) s ; }
        if (ngroups == 0)
        {
            std::vector<std::vector<i32>> lv (nlev) ;
            for (i64 s = 0 ; s < nsuper ; s++) lv [P->level [s]].push_back ((i32) s) ;
            for (auto &l : lv) if (!l.empty ()) batches.push_back (std::move (l)) ;
            return 0 ;
        }
        for (i64 s = 0 ; s < nsuper ; s++)
        {
            if (group [s] < 0) batches.push_back (std::vector<i32> (1, (i32) s)) ;
            else if (group_last [group [s]] == (i32) s)
                for (auto &l : by [group [s]]) if (!l.empty ()) batches.push_back (std::move (l)) ;
        }
        return ngroups ;
    }
    void choose_batches ()
    {
        nominal_budget () ;
        std::vector<std::vector<i32>> best_batches ;
        std::vector<i64> best_cb (std::max<i64> (nsuper, 1), 0) ;
        i64 best_arena = 0 ; int best_nsplit = 1 ;
        const i64 budget = P->arena_budget ;
        for (int nsplit = 1 ; ; nsplit *= 2)
        {
            const int ngroups = make_batches (nsplit) ;
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
            // keep the first split that fits; if none does, the one with the smallest arena (upload_plan then reports the shortage)
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
        if (P->world <= 1) return ;
        // The batch order above is laid out over ALL fronts so that every rank takes the same split decision; the arena
        // itself only has to hold the contribution blocks of this rank's fronts (its subtrees + its shared fronts): lay them
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

    // ---- solve tasks (all supernodes: after cholmod_hip_gather_factor every rank holds L) -------------------------------------
    void build_solve_tasks ()
    {
        P->sv_tasks.clear () ; P->sv_ptr.assign (nlev + 1, 0) ; P->sv_big.assign (nlev, {}) ;
        for (int l = 0 ; l < nlev ; l++)
        {
            for (int q = P->lvl_ptr [l] ; q < P->lvl_ptr [l+1] ; q++)
            {
                i32 sid = P->lvl_list [q] ;
                const FrontD &f = P->fr [sid] ;
                // one workgroup streams ~50-100 GB/s: anything above 512 KB of L gets the multi-workgroup block walk
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
    }

    // ---- windows of the distributed fronts: at the tail of the rank's array, alive for the front's batch only (the region is
    // as long as the neediest batch)
    void layout_windows ()
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

    // ---- launch schedule of this rank ----------------------------------------------------------------------------------------
    // thin fronts of a batch go to the fused LDS-resident kernel, in size classes so that the dynamic LDS of a launch fits its
    // widest member (packed triangle: 4.4 / 9.6 / 16.9 KB, one wave per front -> the 32-wave cap or 16 / 9 fronts per CU;
    // 37.7 / 75.7 KB, four waves per front -> 4 / 2 per CU); the others are returned in `gen`
    void schedule_thin (const std::vector<i32> &mine_ids, std::vector<i32> &gen)
    {
        Schedule &S = P->sch ;
        static const int NCLS = 5 ;
        static const int cls [NCLS] = {32, 48, 64, 96, SM_MAX} ;
        std::vector<i32> bucket [NCLS] ;
        for (i32 sid : mine_ids)
        {
            const FrontD &f = P->fr [sid] ;
            if (!f.cbp) { gen.push_back (sid) ; continue ; }
            int c = 0 ;
            while (c < NCLS - 1 && f.nsrow > cls [c]) c++ ;
            bucket [c].push_back (sid) ;
        }
        // a class too thin to fill the chip rides with the next larger one (a launch costs more than the occupancy it would win)
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
            // (round 6) a class of very many fronts -- the leaves of a 2D / circuit problem: 65 136 in one launch at the
            // G3_circuit stand-in -- goes out as four launches over consecutive fronts: run from a resident S they cost
            // three launch latencies; fed by cholmod_l_factorize from host memory, each waits for its own quarter of the
            // values only (the upload travels in the order of the launches, engine.hip: cholmod_hip_set_value_map)
            const size_t nb_ = bucket [c].size () ;
            const int parts = (P->world == 1 && nb_ >= 16384) ? 4 : 1 ;
            for (int part = 0 ; part < parts ; part++)
            {
            const size_t b0_ = nb_ * part / parts, b1_ = nb_ * (part + 1) / parts ;
            Launch Ls_ {K_SMALL, (int) (b1_ - b0_), (int) (b1_ - b0_), S.sm.size (), 0, 0} ;
            int mx = 0, mxc = 0, mxt = 0 ;
            bool leaves = !(P->flags & CHOLMOD_HIP_NO_LEAF_PAIRS) && !(P->flags & CHOLMOD_HIP_CX_STORAGE) ;
            for (size_t bi = b0_ ; bi < b1_ ; bi++)
            {
                const i32 sid = bucket [c][bi] ;
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
    }
    // zero-fill of the contribution blocks that need it.  Contribution blocks of unshared fronts are never zero-filled: their
    // first trailing update writes C = -L21*L21' (GemmGroup.assign) and the children's contributions to the CB part are
    // extend-added after the dense phase.  Only the children's contributions to the PANEL must be in place before it.
    // (Shared fronts without a distributed block keep the zero-fill: a rank writes only its share of the CB tiles, the rest
    // must read as zero in its partial sum.)
    void schedule_zero (const i32 *ids, int nf)
    {
        Schedule &S = P->sch ;
        const bool can_assign = !(P->flags & CHOLMOD_HIP_NO_CB_ASSIGN) ;
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
    }
    // extend-add of the generic fronts of a batch.  phase 0 (before the dense phase): everything into the panel columns, and
    // into the CB columns of the zero-filled fronts; phase 1 (after it): the CB columns of the assign fronts
    void schedule_extend_add (const i32 *ids, int nf, int phase)
    {
        Schedule &S = P->sch ;
        Launch Le {K_EA, 0, 0, S.eg.size (), 0, 0} ;
        int blocks = 0 ;
        // target columns per workgroup: 8, or 4 when the launch holds a big front (measured at 4 / 8 / 16 / 32: the nd24k
        // stand-in and Poisson 100^3 like 4, the 2D problem 8)
        int tw = EA_TW ;
        for (int q = 0 ; q < nf ; q++)
            if (P->fr [ids [q]].child_end != P->fr [ids [q]].child_begin && P->fr [ids [q]].nsrow >= 2048) tw = 4 ;
        Le.aux = tw ;
        for (int q = 0 ; q < nf ; q++)
        {
            const FrontD &f = P->fr [ids [q]] ;
            if (f.child_end == f.child_begin) continue ;
            bool asg = P->assign_cb [ids [q]] != 0 ;
            if (f.cbd) continue ;               // (nothing is extend-added into a distributed contribution block)
            // (a distributed front takes the contributions to its panel block column by block column, when the block column
            // enters the window: schedule_dense, emit_win)
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
    }
    void schedule_batches ()
    {
        Schedule &S = P->sch ;
        std::vector<i32> mine_ids, gen ;
        P->batch_launch0.clear () ;
        for (const auto &bt : batches)
        {
            P->batch_launch0.push_back (S.launches.size ()) ;      // (the batch's first launch: what its values must have arrived for)
            mine_ids.clear () ; gen.clear () ;
            for (i32 sf : bt) if (mine (sf)) mine_ids.push_back (sf) ;
            if (mine_ids.empty ()) continue ;
            schedule_thin (mine_ids, gen) ;
            const i32 *ids = gen.data () ;
            const int nf = (int) gen.size () ;
            if (nf == 0) continue ;
            schedule_zero (ids, nf) ;
            schedule_extend_add (ids, nf, 0) ;
            schedule_dense (P->fr, ids, nf, S, P->flags, P->owner.data (), P->grp0.data (), P->grpn.data (),
                P->rank, P->world, P->assign_cb.data (), P->win_off.data (), P->child.data (),
                P->world == 1 && !P->force_shared) ;      // (half tiles: one GPU -- plans of several ranks run four tiles per workgroup)
            schedule_extend_add (ids, nf, 1) ;
        }
    }

    int run ()
    {
        int rc = init_fronts () ;
        if (rc != CHOLMOD_HIP_OK) return rc ;
        rc = build_etree () ;
        if (rc != CHOLMOD_HIP_OK) return rc ;
        assign_ownership () ;
        layout_local_factor () ;
        build_child_lists () ;
        choose_batches () ;
        build_solve_tasks () ;
        layout_windows () ;
        schedule_batches () ;
        return CHOLMOD_HIP_OK ;
    }
} ;

} // namespace

int build_host (cholmod_hip_plan *P)
{
    PlanBuilder B (P) ;
    return B.run () ;
}

} // namespace sship
