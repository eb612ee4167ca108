// kernels.hip.h -- gfx950 device kernels of the supernodal Cholesky engine.
//
// The engine is a level-scheduled multifrontal formulation of the reference's
// left-looking loop (CHOLMOD/Supernodal/t_cholmod_super_numeric.c:279-1048):
// each supernode s owns a dense front  F_s = [ panel | contribution block ]
//   panel  = the nsrow x nscol block of the packed Lx array (reference layout,
//            CHOLMOD/Include/cholmod_core.h:1673-1798), factored in place;
//   CB_s   = (nsrow-nscol)^2 Schur contribution, kept in an HBM arena and
//            extend-added into the parent front through the relative row map
//            (the reference's RelativeMap, :743-750, and scatter, :756-772).
// The sum of all CB-borne updates into a front equals the sum of the
// reference's per-descendant dsyrk/dgemm updates (:682-717); the order of
// summation differs, hence parity is to 1e-12, not bitwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "descriptors.hip.h"

namespace sship {

typedef double d4 __attribute__((ext_vector_type(4))) ;
typedef double d2u __attribute__((ext_vector_type(2), aligned(8))) ;

// ---- complex factors in their own storage (CX) -----------------------------------------
// A complex factor is computed on the index space of its real twin (phi embedding: rows /
// columns 2i, 2i+1 = re, im of i; host/complex.c) but STORED as the reference stores it
// (t_cholmod_super_numeric.c:41-83: L complex, interleaved): of every front only the even twin
// columns exist -- column c of the twin lives at (c >> 1) * ld, ld = the twin's row count =
// twice the complex one -- i.e. the panel IS the interleaved complex panel (2 xsize doubles
// instead of the twin's 4 xsize).  The odd columns are the rotations of the even ones,
//     twin (2i, 2j+1) = -twin (2i+1, 2j)      twin (2i+1, 2j+1) = twin (2i, 2j),
// and are rebuilt on the way into LDS / registers by ldcx; stcx keeps the even columns only.
// `base` must address an element with an even row and an even column of the twin.
template <bool CX>
__device__ __forceinline__ double ldcx (const double *base, int r, int c, i64 ld)
{
    if constexpr (!CX) return base [r + (i64) c * ld] ;
    else
    {
        const double v = base [(r ^ (c & 1)) + (i64) (c >> 1) * ld] ;
        return ((c & 1) && !(r & 1)) ? -v : v ;
    }
}
template <bool CX>
__device__ __forceinline__ void stcx (double *base, int r, int c, i64 ld, double v)
{
    if constexpr (!CX) base [r + (i64) c * ld] = v ;
    else if (!(c & 1)) base [r + (i64) (c >> 1) * ld] = v ;
}
// offset of twin column c of a panel with leading dimension ld
template <bool CX> __host__ __device__ __forceinline__ i64 colx (int c, i64 ld) { return CX ? (i64) (c >> 1) * ld : (i64) c * ld ; }

__device__ __forceinline__ int lower_bound_i32 (const i32 *a, int n, int v)
{
    int lo = 0, hi = n ;
    while (lo < hi) { int mid = (lo + hi) >> 1 ; if (a [mid] < v) lo = mid + 1 ; else hi = mid ; }
    return lo ;
}

template <typename G>
__device__ __forceinline__ int find_group (const G *g, int ng, int b, i32 G::*start)
{
    int lo = 0, hi = ng - 1 ;
    while (lo < hi)
    {
        int mid = (lo + hi + 1) >> 1 ;
        if (g [mid].*start <= b) lo = mid ; else hi = mid - 1 ;
    }
    return lo ;
}

// ---- one-time: child -> parent relative row maps ---------------------------
// relmap[rel_d + i] = position of row Ls[pi_d + nscol_d + i] in the parent's
// row list; the reference builds the same numbers through Map[] at
// t_cholmod_super_numeric.c:326-333 and :743-750.  One wave per child, lanes
// stride the rows ("gather" of sorted lists by binary search).
__global__ void __launch_bounds__(256) k_relmap (int nsuper, const FrontD *fr,
    const i64 *Ls, i32 *relmap)
{
    int wave = (blockIdx.x * 256 + threadIdx.x) >> 6 ;
    int lane = threadIdx.x & 63 ;
    if (wave >= nsuper) return ;
    FrontD d = fr [wave] ;
    if (d.parent < 0 || d.ncb == 0) return ;
    FrontD p = fr [d.parent] ;
    const i64 *prow = Ls + p.psi ;
    const i64 *drow = Ls + d.psi + d.nscol ;
    for (int i = lane ; i < d.ncb ; i += 64)
    {
        i64 r = drow [i] ;
        int lo = 0, hi = p.nsrow ;
        while (lo < hi) { int mid = (lo + hi) >> 1 ; if (prow [mid] < r) lo = mid + 1 ; else hi = mid ; }
        relmap [d.rel + i] = lo ;
    }
}

// The same for (contributor, ancestor) pairs (several GPUs: contributions routed past the contribution blocks of shared
// fronts): map [off + i] = position of the contributor's contribution-block row i in the ancestor's row list, -1 for the
// rows below the ancestor's first column (consumed by a front in between).
__global__ void __launch_bounds__(256) k_relmap_pairs (int npairs, const RelPair *pr, const FrontD *fr, const i64 *Ls, i32 *relmap)
{
    int wave = (blockIdx.x * 256 + threadIdx.x) >> 6 ;
    int lane = threadIdx.x & 63 ;
    if (wave >= npairs) return ;
    const RelPair R = pr [wave] ;
    FrontD d = fr [R.d], p = fr [R.a] ;
    const i64 *prow = Ls + p.psi ;
    const i64 *drow = Ls + d.psi + d.nscol ;
    for (int i = lane ; i < d.ncb ; i += 64)
    {
        i64 r = drow [i] ;
        int lo = 0, hi = p.nsrow ;
        while (lo < hi) { int mid = (lo + hi) >> 1 ; if (prow [mid] < r) lo = mid + 1 ; else hi = mid ; }
        relmap [R.off + i] = (r < p.k1) ? -1 : lo ;
    }
}

// ---- assemble A into the panels ---------------------------------------------
// reference: t_cholmod_super_numeric.c:353-431 (ASSIGN semantics, entries not
// in the symbolic pattern are dropped, beta added to the diagonal).
template <bool CX>
__global__ void __launch_bounds__(256) k_assemble (i64 n, const i64 *Sp,
    const i64 *Snz, const i64 *Si, const double *Sx, const i32 *supermap,
    const FrontD *fr, const i64 *Ls, double *Lx, double beta, i64 *amap)
{
    i64 k = blockIdx.x * (i64) 256 + threadIdx.x ;
    if (k >= n) return ;
    const FrontD &f = fr [supermap [k]] ;
    if (f.assemble != 1) return ;      // 0: another rank's, 2: k_small_front does it
    if (!col_owned (f, (int) (k - f.k1))) return ;     // a distributed front: the column's owner assembles it
    i64 psx = f.psx, psi = f.psi ;
    int nsrow = f.nsrow, k1 = f.k1 ;
    const i64 *rows = Ls + psi ;
    i64 p = Sp [k], pend = Snz ? p + Snz [k] : Sp [k+1] ;
    if constexpr (CX)
    {
        // (the odd columns of the embedded matrix are not stored)
        if (k & 1) { for (i64 q = p ; q < pend ; q++) amap [q] = -1 ; return ; }
    }
    double *col = Lx + psx + colx<CX> (col_local (f, (int) (k - k1)), nsrow) ;
    for ( ; p < pend ; p++)
    {
        i64 i = Si [p] ;
        if (i < k) continue ;
        int lo = 0, hi = nsrow ;
        while (lo < hi) { int mid = (lo + hi) >> 1 ; if (rows [mid] < i) lo = mid + 1 ; else hi = mid ; }
        if (lo < nsrow && rows [lo] == i)
        {
            // (a valid cholmod_sparse holds no duplicate entries, Check/cholmod_check.c; if a
            // sorted column carries some all the same, the LAST one wins here as in the
            // reference's assignment loop, and only that one enters the map, so the mapped
            // scatter of later factorizations -- one thread per entry -- has a single writer)
            if (p + 1 < pend && Si [p + 1] == i) continue ;
            col [lo] = Sx [p] ;
            // remember where this entry of S lives in Lx: later factorizations of the same
            // resident S stream through the map instead of searching (k_assemble_mapped)
            amap [p] = (col - Lx) + lo ;
        }
    }
    if (beta != 0.0) col [k - k1] += beta ;
}

// The same scatter for a matrix whose map is known: 8 B of map + 8 B of value in,
// 8 B out per entry, no search (the binary searches of k_assemble cost ~1.5 GB of
// uncoalesced Ls reads at Poisson 100^3, 1.7 ms; this pass: 0.1 ms).
__global__ void __launch_bounds__(256) k_assemble_mapped (i64 nz, const i64 *amap, const double *Sx, double *Lx)
{
    i64 p = blockIdx.x * (i64) 256 + threadIdx.x ;
    if (p >= nz) return ;
    i64 q = amap [p] ;
    if (q >= 0) Lx [q] = Sx [p] ;
}

// Sx [q] = values [src [q]]: new values of the caller's matrix into the resident S
__global__ void __launch_bounds__(256) k_gather_values (i64 nz, const i64 *src, const double *values, double *Sx)
{
    i64 q = blockIdx.x * (i64) 256 + threadIdx.x ;
    if (q < nz) Sx [q] = values [src [q]] ;
}

// One chunk of a value upload in batch order (round 6): staged entry k belongs to entry order [k] of S; it refreshes the
// resident S (the thin-front kernels read their columns from there) and, where the entry has a place in L's generic
// fronts, lands there.  "+=" into the cleared L: k_add_beta has run before, no two entries of S share a target.
__global__ void __launch_bounds__(256) k_values_chunk (i64 k0, i64 k1, const i64 *order, const double *staged, const i64 *amap,
    double *Sx, double *Lx)
{
    i64 k = k0 + blockIdx.x * (i64) 256 + threadIdx.x ;
    if (k >= k1) return ;
    const i64 q = order [k] ;
    const double v = staged [k] ;
    Sx [q] = v ;
    const i64 m = amap [q] ;
    if (m >= 0) Lx [m] += v ;
}

// Lx(k,k) += beta for the columns this rank's k_assemble owns (after k_assemble_mapped)
template <bool CX>
__global__ void __launch_bounds__(256) k_add_beta (i64 n, const i32 *supermap, const FrontD *fr, double *Lx, double beta)
{
    i64 k = blockIdx.x * (i64) 256 + threadIdx.x ;
    if (k >= n) return ;
    const FrontD &f = fr [supermap [k]] ;
    if (f.assemble != 1 || !col_owned (f, (int) (k - f.k1))) return ;
    if (CX && (k & 1)) return ;
    Lx [f.psx + colx<CX> (col_local (f, (int) (k - f.k1)), f.nsrow) + (k - f.k1)] += beta ;
}

// ---- zero the contribution blocks of a level --------------------------------
// Only the lower triangle is ever read (extend-add, k_small_front) or kept, so
// only it is cleared: a block owns ZERO_COLS columns and clears rows j0..ncb-1.
__global__ void __launch_bounds__(256) k_zero (const ZeroGroup *g, int ng, double *CB)
{
    int gi = find_group (g, ng, (int) blockIdx.x, &ZeroGroup::blk_start) ;
    ZeroGroup G = g [gi] ;
    int ncb = (int) G.len ;                 // len holds ncb (square side)
    // (G.pad = 1: a complex front in its own storage, ncb / 2 stored columns, stored column j
    // = twin column 2 j whose live rows start at 2 j)
    int ncol = G.pad ? ncb >> 1 : ncb, cs = G.pad ? 2 : 1 ;
    int j0 = ((int) blockIdx.x - G.blk_start) * ZERO_COLS ;
    int j1 = j0 + ZERO_COLS < ncol ? j0 + ZERO_COLS : ncol ;
    double *dst = CB + G.off ;
    for (int j = j0 ; j < j1 ; j++)
    {
        double *col = dst + (i64) j * ncb ;
        for (int i = cs * j0 + threadIdx.x ; i < ncb ; i += 256) col [i] = 0.0 ;
    }
}

// ---- extend-add: pull the children's contribution blocks into a front -------
// One workgroup owns tw (EA_TW = 8, or 4) consecutive target columns of the parent front and
// visits every child; wave w owns the target columns == w (mod 4), so no two
// waves ever touch the same entry and no atomics are needed.  Reads of a child
// CB column are contiguous (coalesced); the target rows follow the relative
// map.  reference scatter: t_cholmod_super_numeric.c:756-772.
// a child's contribution block is read exactly once: streamed past the caches (EA_NT=0: ordinary loads)
#ifndef EA_NT
#define EA_NT 1
#endif
#if EA_NT
#define EA_LDV(p) __builtin_nontemporal_load (p)
#else
#define EA_LDV(p) (*(p))
#endif
template <bool CX>
__global__ void __launch_bounds__(256) k_extend_add (const EaGroup *g, int ng,
    const FrontD *fr, const i32 *child, const i64 *crel, const i32 *relmap, double *Lx, double *CB, int tw)
{
    int gi = find_group (g, ng, (int) blockIdx.x, &EaGroup::blk_start) ;
    const FrontD &P = fr [g [gi].front] ;
    int c0 = g [gi].c_lo + ((int) blockIdx.x - g [gi].blk_start) * tw ;
    int c1 = c0 + tw < g [gi].c_hi ? c0 + tw : g [gi].c_hi ;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63 ;
    // (launched with four waves, or -- on the exchange stream, beside a trailing update whose one-wave workgroups refill
    // every register file slot as it frees up -- with ONE: a four-wave workgroup needs room on all four SIMDs of a CU at
    // once and starves there; its single wave then takes all the target columns of the block)
    const int nwv = (int) blockDim.x >> 6 ;
    i64 Ppsx = g [gi].pbase != EA_NO_PBASE ? g [gi].pbase : P.psx, Pcb = P.cb ;
    int Pnscol = P.nscol, Pnsrow = P.nsrow, Pncb = P.ncb ;
    int cb = P.child_begin, ce = P.child_end ;
    for (int ci = cb ; ci < ce ; ci++)
    {
        const FrontD &Cc = fr [child [ci]] ;
        // (crel: the map of THIS pair -- child -> parent, or contributor -> ancestor when the contributions of a shared
        // front's descendants are routed past its contribution block; rows below the ancestor's first column map to -1)
        const i32 *rm = relmap + crel [ci] ;
        int nc = Cc.ncb ;
        const double *src = CB + Cc.cb ;
        const bool cbd = Cc.cbd != 0 ;
        int j0 = lower_bound_i32 (rm, nc, c0) ;
        int j1 = lower_bound_i32 (rm, nc, c1) ;
        for (int j = j0 ; j < j1 ; j++)
        {
            int tc = rm [j] ;
            if (cbd && (j < Cc.cb_lo || j >= Cc.cb_hi)) continue ;      // (a distributed block: the columns this rank stores)
            if constexpr (CX) { if ((j & 1) || (((tc >> 1) & 3) % nwv) != wave) continue ; }      // (an even child column lands on an even column)
            else if (((tc & 3) % nwv) != wave) continue ;
            double *dst ;
            int roff ;
            if (tc < Pnscol) { dst = Lx + Ppsx + colx<CX> (tc, Pnsrow) ; roff = 0 ; }
            else { dst = CB + Pcb + colx<CX> (tc - Pnscol, Pncb) ; roff = Pnscol ; }
            const double *sc = cbd ? src + (i64) (j - Cc.cb_lo) * nc
                             : (!CX && Cc.cbp) ? src + tri_col (j, nc) : src + colx<CX> (j, nc) ;
            // eight, then four independent gather / read-modify-write chains in flight per wave
            int i = j + lane ;
            for ( ; i + 448 < nc ; i += 512)
            {
                int r [8] ; double v [8], d [8] ;
#pragma unroll
                for (int q = 0 ; q < 8 ; q++) { r [q] = rm [i + 64 * q] - roff ; v [q] = EA_LDV (sc + i + 64 * q) ; }
#pragma unroll
                for (int q = 0 ; q < 8 ; q++) d [q] = dst [r [q]] ;
#pragma unroll
                for (int q = 0 ; q < 8 ; q++) dst [r [q]] = d [q] + v [q] ;
            }
            for ( ; i + 192 < nc ; i += 256)
            {
                int r0 = rm [i] - roff, r1 = rm [i + 64] - roff, r2 = rm [i + 128] - roff, r3 = rm [i + 192] - roff ;
                double v0 = EA_LDV (sc + i), v1 = EA_LDV (sc + i + 64), v2 = EA_LDV (sc + i + 128), v3 = EA_LDV (sc + i + 192) ;
                double d0 = dst [r0], d1 = dst [r1], d2 = dst [r2], d3 = dst [r3] ;
                dst [r0] = d0 + v0 ; dst [r1] = d1 + v1 ; dst [r2] = d2 + v2 ; dst [r3] = d3 + v3 ;
            }
            for ( ; i < nc ; i += 64)
                dst [rm [i] - roff] += sc [i] ;
        }
    }
}

// ---- diagonal-block Cholesky (nb <= 64), one wave per block ----------------
// LAPACK dpotrf("L") semantics on the nb x nb block (reference call
// t_cholmod_super_numeric.c:864-867): the first pivot <= 0 stops the
// factorization and is reported 1-based, relative to the front, in info[front];
// NaN pivots do not stop it (:907-908).  After a failure every remaining
// column of this front is written as zero (:889-895, :926-931).
// sqrt(d) and 1/sqrt(d) from one v_rsq_f64 seed and a coupled Goldschmidt
// iteration (two quadratic steps + one correction): ~15 dependent FMAs instead
// of the ~70 instructions of sqrt() followed by a full-precision division.
// Both results are within 1 ulp for normal d.
// Branch-free: arguments outside [1e-290, 1e290] are rescaled by 2^(+-512)
// around the iteration (v_ldexp_f64), so there is no slow path for hipcc to
// if-convert onto the caller's dependency chain.  d <= 0 or non-finite gives
// NaN / garbage, which every caller masks (a failed pivot).
__device__ __forceinline__ void sqrt_rsqrt (double d, double &r, double &ri)
{
    int sh = d > 1e290 ? -512 : (d < 1e-290 ? 512 : 0) ;
    d = __builtin_ldexp (d, sh) ;
    double y = __builtin_amdgcn_rsq (d) ;       // ~2^-26 relative accuracy
    double g = d * y, h = 0.5 * y ;
    double e = __builtin_fma (-h, g, 0.5) ;
    g = __builtin_fma (g, e, g) ; h = __builtin_fma (h, e, h) ;
    e = __builtin_fma (-h, g, 0.5) ;
    g = __builtin_fma (g, e, g) ; h = __builtin_fma (h, e, h) ;
    double t = __builtin_fma (-g, g, d) ;       // residual d - g^2
    g = __builtin_fma (t, h, g) ;
    // refine 1/g: ri = 2h, one Newton step on g * ri = 1
    ri = h + h ;
    double u = __builtin_fma (-g, ri, 1.0) ;
    ri = __builtin_fma (u, ri, ri) ;
    r = __builtin_ldexp (g, -(sh >> 1)) ;
    ri = __builtin_ldexp (ri, sh >> 1) ;
}

// ---- diagonal-block Cholesky, second generation --------------------------------
// Same contract as k_potrf.  Four waves; the block is eliminated in 16-column
// panels:
//  (1) wave 0 factors the panel, lane = row of the panel (diagonal-block rows
//      and the rows below ride through the same right-looking elimination), 16
//      registers per lane.  The pivot and the multipliers travel by
//      v_readlane, and the dependent chain per column is kept to
//      readlane -> rcp + one Newton step -> mul -> fma: the square root and its
//      reciprocal (sqrt_rsqrt) are taken off that chain -- the elimination
//      itself runs on the unscaled columns (u = l * sqrt(d), an LDL' step),
//      the scaled column l = u * rsqrt(d) is only what gets stored;
//  (2) all four waves apply the panel to the trailing 16x16 tiles with
//      v_mfma_f64_16x16x4 straight out of the k-major LDS copy.
// 1/d from v_rcp_f64 plus one Newton step is within ~1.5 ulp, the result
// differs from a division-based dpotrf by rounding only (parity tests: 1e-12).
__device__ __forceinline__ double readlane_f64 (double v, int l)
{
    int lo = __double2loint (v), hi = __double2hiint (v) ;
    lo = __builtin_amdgcn_readlane (lo, l) ;
    hi = __builtin_amdgcn_readlane (hi, l) ;
    return __hiloint2double (hi, lo) ;
}
// The elimination of a <= 64 x 64 block held k-major in LDS (T [k * PF2_LD + i] = A(i,k),
// identity-padded to nblk 16-column panels).
// *s_fail (preset to -1) receives the first column with a pivot <= 0.  All 256
// threads of the workgroup call it; it ends with a barrier.
// PHI: the block is the phi embedding of a complex Hermitian block (rows / columns 2k, 2k+1 = re, im; complex storage):
// the panel phase then eliminates COMPLEX columns -- the pair (2p, 2p+1) in one step.  Column 2p+1 is the rotation of
// column 2p (entry (r, 2p+1) = -/+ entry (r ^ 1, 2p)), so it is never eliminated, only rebuilt at the end; with
// t = column 2p / d and s (r) = +t (r + 1) on even rows, -t (r - 1) on odd rows (a lane swap inside the pair), a trailing
// even column 2j takes  a -= t x_j + s y_j  (x_j, y_j = rows 2j, 2j+1 of column 2p: the complex multiply-add of zpotrf's
// rank-1 step): 8 pivots, reciprocals and 56 (read-lane pair, two fma) steps per 16 twin columns instead of 16 and 120.
__device__ __forceinline__ double lane_xor1_f64 (double v)
{
    int lo = __double2loint (v), hi = __double2hiint (v) ;
    lo = __builtin_amdgcn_update_dpp (0, lo, 0xB1, 0xF, 0xF, true) ;       // quad_perm [1, 0, 3, 2]
    hi = __builtin_amdgcn_update_dpp (0, hi, 0xB1, 0xF, 0xF, true) ;
    return __hiloint2double (hi, lo) ;
}
template <bool PHI = false, typename Tick>
__device__ __forceinline__ void pf_eliminate (double *T, int nblk, int *s_fail, int tid, Tick tick)
{
    const int lane = tid & 63, wave = tid >> 6 ;
    int lr = lane & 15, lk = lane >> 4 ;
    __shared__ __attribute__((aligned(16))) double bc [128] ;        // wave 0's broadcast scratch (real panels)
    for (int jb = 0 ; jb < nblk ; jb++)
    {
        int c0 = 16 * jb ;
        if (PHI && wave == 0)
        {
            int row = c0 + lane ;
            int rr = row < PF_NB ? row : PF_NB - 1 ;
            const bool odd = (lane & 1) != 0 ;
            double a [8] ;                      // the even columns c0 + 2 p of the panel
#pragma unroll
            for (int p = 0 ; p < 8 ; p++) a [p] = T [(c0 + 2 * p) * PF2_LD + rr] ;
            int fail = -1 ;
            double dv = 1.0 ;                   // lane 2 p keeps the pivot of complex column p
#pragma unroll
            for (int p = 0 ; p < 8 ; p++)
            {
                double d = readlane_f64 (a [p], 2 * p) ;
                if (fail < 0 && d <= 0.0) fail = c0 + 2 * p ;
                double x = __builtin_amdgcn_rcp (d) ;
                double e = __builtin_fma (-d, x, 1.0) ;
                double t0 = a [p] * x ;
                double u [8], v [8] ;
#pragma unroll
                for (int j = p + 1 ; j < 8 ; j++) { u [j] = readlane_f64 (a [p], 2 * j) ; v [j] = readlane_f64 (a [p], 2 * j + 1) ; }
                __builtin_amdgcn_sched_barrier (0) ;
                double t = __builtin_fma (t0, e, t0) ;
                double sw = lane_xor1_f64 (t) ;
                double sgn = odd ? -sw : sw ;
#pragma unroll
                for (int j = p + 1 ; j < 8 ; j++) { a [j] = __builtin_fma (-t, u [j], a [j]) ; a [j] = __builtin_fma (-sgn, v [j], a [j]) ; }
                if (lane == 2 * p) dv = d ;
                __builtin_amdgcn_sched_barrier (0) ;
            }
            tick (1) ;
            double r, ri ;
            sqrt_rsqrt (dv, r, ri) ;
            if (row < PF_NB)
            {
#pragma unroll
                for (int p = 0 ; p < 8 ; p++)
                {
                    double rc = readlane_f64 (r, 2 * p), ric = readlane_f64 (ri, 2 * p) ;
                    double ev = (lane == 2 * p) ? rc : a [p] * ric ;
                    if (fail >= 0 && c0 + 2 * p >= fail) ev = 0.0 ;
                    double sw = lane_xor1_f64 (ev) ;
                    T [(c0 + 2 * p) * PF2_LD + row] = ev ;
                    T [(c0 + 2 * p + 1) * PF2_LD + row] = odd ? sw : -sw ;        // (its entry above the diagonal is never read)
                }
            }
            if (fail >= 0 && lane == 0) (*s_fail) = fail ;
            tick (2) ;
        }
        else if (wave == 0)
        {
            // panel rows c0 .. 63: lane = row c0 + lane (lanes past the block idle)
            int row = c0 + lane ;
            int rr = row < PF_NB ? row : PF_NB - 1 ;
            double a [16] ;
#pragma unroll
            for (int c = 0 ; c < 16 ; c++) a [c] = T [(c0 + c) * PF2_LD + rr] ;
            int fail = -1 ;
            double dv = 1.0 ;                   // lane c keeps the pivot of column c
            // Round 6: pivot and multipliers of a column travel through a 128-double LDS scratch of the wave, as in the
            // thin-front kernel (tf_panel) since round 2 -- every lane writes its entry of the column (one ds_write_b64; the
            // panel's diagonal rows are lanes 0 .. 15), everybody reads them back as broadcast ds_read_b128 pairs: (15 - c) / 2
            // + (15 - c) + 8 vector instructions per column where a v_readlane pair per value cost 3 (15 - c) + 10.  LDS
            // operations of one wave execute in order: the read behind the write needs no barrier.  The LDS round trip of the
            // multipliers runs beside the reciprocal's chain (v_rcp_f64 + one Newton step on the pivot, which arrives the
            // same way).
            typedef double pf_d2 __attribute__((ext_vector_type(2))) ;
#pragma unroll
            for (int c = 0 ; c < 16 ; c++)
            {
                bc [lane] = a [c] ;
                asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
                double u [16] ;
#pragma unroll
                for (int c2 = c & ~1 ; c2 < 16 ; c2 += 2)
                {
                    pf_d2 v = *(const pf_d2 *) (bc + c2) ;
                    u [c2] = v.x ; u [c2 + 1] = v.y ;
                }
                asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
                const double d = u [c] ;
                if (fail < 0 && d <= 0.0) fail = c0 + c ;
                // (past a failed pivot the columns carry garbage; they are
                // zeroed below, and nothing flows back into earlier columns)
                double x = __builtin_amdgcn_rcp (d) ;
                double e = __builtin_fma (-d, x, 1.0) ;
                double t0 = a [c] * x ;             // beside e, off the chain: t = a x (1 + e)
                double t = __builtin_fma (t0, e, t0) ;      // u(row,c) / d
#pragma unroll
                for (int c2 = c + 1 ; c2 < 16 ; c2++) a [c2] = __builtin_fma (-t, u [c2], a [c2]) ;
                if (lane == c) dv = d ;
                __builtin_amdgcn_sched_barrier (0) ;
            }
            tick (1) ;
            // off the chain: sqrt / rsqrt of the 16 pivots side by side in lanes
            // 0..15, then the stored columns l = u * rsqrt(d)
            double r, ri ;
            sqrt_rsqrt (dv, r, ri) ;
            bc [lane] = r ; bc [64 + lane] = ri ;
            asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
#pragma unroll
            for (int c = 0 ; c < 16 ; c += 2)
            {
                pf_d2 rv = *(const pf_d2 *) (bc + c), iv = *(const pf_d2 *) (bc + 64 + c) ;
                a [c] = (lane == c) ? rv.x : a [c] * iv.x ;
                a [c + 1] = (lane == c + 1) ? rv.y : a [c + 1] * iv.y ;
                if (fail >= 0 && c0 + c >= fail) a [c] = 0.0 ;
                if (fail >= 0 && c0 + c + 1 >= fail) a [c + 1] = 0.0 ;
            }
            asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
            // (entries above the diagonal of the 16x16 block are never read again)
            if (row < PF_NB)
            {
#pragma unroll
                for (int c = 0 ; c < 16 ; c++) T [(c0 + c) * PF2_LD + row] = a [c] ;
            }
            if (fail >= 0 && lane == 0) (*s_fail) = fail ;
            tick (2) ;
        }
        __syncthreads () ;
        tick (3) ;
        if ((*s_fail) >= 0) break ;
        // trailing tiles (ti >= tj > jb), dealt round-robin to the waves
        int nt = nblk - 1 - jb ;
        int ntile = nt * (nt + 1) / 2 ;
        for (int u = wave ; u < ntile ; u += 4)
        {
            // u -> (ti, tj) in the lower triangle of an nt x nt tile grid
            int tj = 0, rem = u ;
            while (rem >= nt - tj) { rem -= nt - tj ; tj++ ; }
            int ti = tj + rem ;
            int i0 = 16 * (jb + 1 + ti), j0 = 16 * (jb + 1 + tj) ;
            d4 acc ;
#pragma unroll
            for (int r = 0 ; r < 4 ; r++) acc [r] = T [(j0 + lk + 4 * r) * PF2_LD + i0 + lr] ;
#pragma unroll
            for (int kk = 0 ; kk < 16 ; kk += 4)
            {
                double av = -T [(c0 + kk + lk) * PF2_LD + i0 + lr] ;
                double bv = T [(c0 + kk + lk) * PF2_LD + j0 + lr] ;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64 (bv, av, acc, 0, 0, 0) ;
            }
#pragma unroll
            for (int r = 0 ; r < 4 ; r++) T [(j0 + lk + 4 * r) * PF2_LD + i0 + lr] = acc [r] ;
        }
        tick (4) ;
        __syncthreads () ;
        tick (5) ;
    }
}

// The factored 64 x 64 block of pf_eliminate back to the front: the lower triangle, zero at /
// beyond a failed pivot.  CX: the even columns only, and entry (2j+1, 2j) -- the imaginary part
// of a diagonal entry, a rounding residue of the embedded elimination -- as the exact zero
// zpotrf leaves there (the rotation then rebuilds an exactly zero (2j, 2j+1)).
template <bool CX>
__device__ __forceinline__ void pf_store (double *A, i64 lda, const double *T, int nb, int fail, int lane, int wave)
{
#pragma unroll
    for (int q = 0 ; q < 16 ; q++)
    {
        int k = wave + 4 * q, i = lane ;
        if (k < nb && i >= k && i < nb)
        {
            double v = (fail >= 0 && k >= fail) ? 0.0 : T [k * PF2_LD + i] ;
            if (CX && i == k + 1) v = 0.0 ;
            stcx<CX> (A, i, k, lda, v) ;
        }
    }
}

template <bool TIMED, bool CX = false>
__global__ void __launch_bounds__(256) k_potrf_mfma (const PfGroup *g, double *Lx, i32 *info, long long *tim)
{
    long long tc [8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0 ;
    auto tick = [&] (int slot) { if constexpr (TIMED) { long long t = __builtin_readcyclecounter () ; tc [slot] += t - t_prev ; t_prev = t ; } } ;
    if constexpr (TIMED) t_prev = __builtin_readcyclecounter () ;
    __shared__ __attribute__((aligned(16))) double T [PF_NB * PF2_LD] ;   // T[k][i] = L(i,k)
    __shared__ int s_fail ;
    __builtin_amdgcn_s_setprio (3) ;
    PfGroup G = g [blockIdx.x] ;
    double *A = Lx + G.off ;
    int nb = G.nb, lda = G.lda ;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6 ;
    if (info [G.front] != 0)
    {
        for (int k = wave ; k < nb ; k += 4)
            if (lane >= k && lane < nb) stcx<CX> (A, lane, k, lda, 0.0) ;
        return ;
    }
    int nbp = (nb + 15) / 16 * 16 ;
    int nblk = nbp / 16 ;
    // stage: thread = (row i, columns wave + 4 q), all 16 loads in flight
    {
        int i = lane ;
        int ic = i < nb ? i : nb - 1 ;
        double tmp [16] ;
#pragma unroll
        for (int q = 0 ; q < 16 ; q++)
        {
            int k = wave + 4 * q ;
            tmp [q] = ldcx<CX> (A, ic, k < nb ? k : nb - 1, lda) ;
        }
#pragma unroll
        for (int q = 0 ; q < 16 ; q++)
        {
            int k = wave + 4 * q ;
            T [k * PF2_LD + i] = (i < nb && k < nb) ? tmp [q] : (i == k ? 1.0 : 0.0) ;
        }
    }
    if (tid == 0) s_fail = -1 ;
    __syncthreads () ;
    tick (0) ;
    pf_eliminate<CX> (T, nblk, &s_fail, tid, tick) ;
    int fail = s_fail ;
    if (fail >= 0 && tid == 0) info [G.front] = G.col0 + fail + 1 ;
    // write-back of the lower triangle (columns at / beyond a failed pivot: zero)
    pf_store<CX> (A, lda, T, nb, fail, lane, wave) ;
    tick (6) ;
    if constexpr (TIMED) { if (tid == 0) for (int q = 0 ; q < 8 ; q++) tim [q] = tc [q] ; }
}

// ---- panel triangular solve on the matrix cores ------------------------------
// Same contract as k_trsm (B := B * inv(L11)', dtrsm("R","L","C","N") of the
// reference, t_cholmod_super_numeric.c:997-1002), organised as a blocked
// substitution over 16-column blocks so that everything but the four 16x16
// diagonal inverses runs on v_mfma_f64_16x16x4:
//     X_j = (B_j - sum_{i<j} X_i * L_ji') * inv(L_jj)'
// One workgroup = 4 waves = 64 rows; a wave owns 16 rows and needs no data of
// the other waves after the staging phase.  LDS (k-major, so that the 16 lanes
// of an MFMA operand read 16 consecutive doubles):
//   Ls [k][c]  = -L11(c,k) below the diagonal, L11(k,k) on it (identity-padded
//                to a multiple of 16 and beyond a failed pivot)
//   Wd [b][k][c] = inv(L_bb)(c,k), computed per workgroup: lane q of wave b
//                solves column q by forward substitution, reciprocals of the
//                diagonal precomputed so the 16-step chain is mul + fma
// Solved blocks and intermediates stay in registers (see the layout note below).
__host__ __device__ inline size_t trsm_mfma_lds_bytes (int ldl)
{
    return (size_t) (ldl * ldl + (ldl / 16) * 256) * sizeof (double) ;
}
// The two inner stages of the panel solve.
// Ls [k * ldl + c] = -L11(c,k) below the diagonal, L11(k,k) on it (identity-padded);
// Wd [b][k][c] receives inv(L_bb)(c,k) of the 16 x 16 diagonal blocks (wave b).
template <typename Tick>
__device__ __forceinline__ void trsm_diag_inverses (const double *Ls, int ldl, double *Wd, int nblk,
    int lane, int wave, Tick tick)
{
    if (wave < nblk)
    {
        int b = wave ;
        int rl = lane & 15 ;
        double Lr [16], acc [16], y [16] ;
#pragma unroll
        for (int e = 0 ; e < 16 ; e++) Lr [e] = Ls [(16 * b + e) * ldl + 16 * b + rl] ;   // -L_bb(rl,e), diag +
        double rdv = 0.0 ;
#pragma unroll
        for (int e = 0 ; e < 16 ; e++) if (rl == e) rdv = Lr [e] ;
        {
            // 1 / diagonal: v_rcp_f64 + two Newton steps
            double x = __builtin_amdgcn_rcp (rdv) ;
            double t = __builtin_fma (-rdv, x, 1.0) ; x = __builtin_fma (x, t, x) ;
            t = __builtin_fma (-rdv, x, 1.0) ; x = __builtin_fma (x, t, x) ;
            rdv = x ;
        }
        tick (1) ;
#pragma unroll
        for (int r = 0 ; r < 16 ; r++) acc [r] = (r == rl) ? 1.0 : 0.0 ;
#pragma unroll
        for (int e = 0 ; e < 16 ; e++)
        {
            double mu [16] ;
#pragma unroll
            for (int r = e + 1 ; r < 16 ; r++) mu [r] = readlane_f64 (Lr [e], r) ;
            __builtin_amdgcn_sched_barrier (0) ;
            y [e] = acc [e] * readlane_f64 (rdv, e) ;
#pragma unroll
            for (int r = e + 1 ; r < 16 ; r++) acc [r] = __builtin_fma (mu [r], y [e], acc [r]) ;
            __builtin_amdgcn_sched_barrier (0) ;
        }
        if (lane < 16)
        {
#pragma unroll
            for (int r = 0 ; r < 16 ; r++) Wd [b * 256 + lane * 16 + r] = y [r] ;     // Wd[b][k=q][c=r]
        }
    }
}

// bj [jj][r] = B(row lr of the wave's 16 rows, column 16 jj + lk + 4 r) in the
// accumulator layout; solves X = B inv(L11)' block column by block column and stores it.
// xr [j] receives the solved block j, again in the accumulator (= A-operand) layout.
// (B: base of the row block, an even row / even column of the twin when CX; brow: this lane's row in it)
template <bool CX, typename Tick>
__device__ __forceinline__ void trsm_solve_rows (const d4 (&bj) [4], int nblk, const double *Ls, int ldl,
    const double *Wd, int lane, int nvalid, bool rok, int nb, double *B, int brow, i64 lda, Tick tick, d4 (&xr) [4])
{
    const int lr = lane & 15, lk = lane >> 4 ;
#pragma unroll
    for (int j = 0 ; j < 4 ; j++)
    {
        if (j < nblk)
        {
            d4 acc = bj [j] ;
#pragma unroll
            for (int i = 0 ; i < 4 ; i++)
            {
                if (i < j)
                {
#pragma unroll
                    for (int s4 = 0 ; s4 < 4 ; s4++)
                    {
                        double b = Ls [(16 * i + 4 * s4 + lk) * ldl + 16 * j + lr] ;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64 (b, xr [i][s4], acc, 0, 0, 0) ;
                    }
                }
            }
            tick (3) ;
            d4 x = (d4) {0.0, 0.0, 0.0, 0.0} ;
#pragma unroll
            for (int s4 = 0 ; s4 < 4 ; s4++)
            {
                double b = Wd [j * 256 + (4 * s4 + lk) * 16 + lr] ;
                x = __builtin_amdgcn_mfma_f64_16x16x4f64 (b, acc [s4], x, 0, 0, 0) ;
            }
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                int c = 16 * j + lk + 4 * r ;
                double v = (c < nvalid) ? x [r] : 0.0 ;
                x [r] = v ;
                if (rok && c < nb) stcx<CX> (B, brow, c, lda, v) ;
            }
            xr [j] = x ;
            tick (5) ;
        }
    }
}

template <bool TIMED, bool CX = false>
__global__ void __launch_bounds__(256) k_trsm_mfma (const TrGroup *g, int ng,
    double *Lx, const i32 *info, int ldl, long long *tim)
{
    long long tc [8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0 ;
    auto tick = [&] (int slot) { if constexpr (TIMED) { long long t = __builtin_readcyclecounter () ; tc [slot] += t - t_prev ; t_prev = t ; } } ;
    if constexpr (TIMED) t_prev = __builtin_readcyclecounter () ;
    extern __shared__ __attribute__((aligned(16))) double trsm_lds [] ;
    double *Ls = trsm_lds ;                         // [ldl][ldl]
    double *Wd = Ls + ldl * ldl ;                   // [ldl/16][16][16]
    __builtin_amdgcn_s_setprio (3) ;
    int gi = find_group (g, ng, (int) blockIdx.x, &TrGroup::blk_start) ;
    TrGroup G = g [gi] ;
    int nb = G.nb, lda = G.lda ;
    int nbp = (nb + 15) / 16 * 16 ;
    int nblk = nbp / 16 ;
    const double *L11 = Lx + G.l_off ;
    int inf = info [G.front] ;
    int nvalid = nb ;
    if (inf != 0)
    {
        nvalid = inf - 1 - G.col0 ;
        if (nvalid < 0) nvalid = 0 ;
        if (nvalid > nb) nvalid = nb ;
    }
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6 ;
    int lr = lane & 15, lk = lane >> 4 ;
    int row0 = ((int) blockIdx.x - G.blk_start) * TRM_ROWS + wave * 16 ;
    int row = row0 + lr ;
    bool rok = row < G.m ;
    double *B = Lx + G.b_off ;
    const int brow = rok ? row : G.m - 1 ;
    // stage L11 (thread = row j of L11, columns 4 apart); the loads of this
    // wave's 16 rows of B are issued right behind and only waited for when the
    // first block needs them
    d4 bj [4] ;
    {
        int j = tid & 63 ;
        int jc = j < nb ? j : nb - 1 ;
        double tmp [16] ;
#pragma unroll
        for (int q = 0 ; q < 16 ; q++)
        {
            int k = (tid >> 6) + 4 * q ;
            tmp [q] = ldcx<CX> (L11, jc, k < nb ? k : nb - 1, lda) ;
        }
        // B straight into the accumulator layout: lane holds B(row lr, col 16 j + lk + 4 r)
#pragma unroll
        for (int jj = 0 ; jj < 4 ; jj++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                int c = 16 * jj + lk + 4 * r ;
                bj [jj][r] = ldcx<CX> (B, brow, c < nb ? c : nb - 1, lda) ;
            }
#pragma unroll
        for (int q = 0 ; q < 16 ; q++)
        {
            int k = (tid >> 6) + 4 * q ;
            double v = (j == k) ? 1.0 : 0.0 ;
            if (j < nvalid && k < j) v = -tmp [q] ;
            if (j < nvalid && k == j) v = tmp [q] ;
            if (j < nbp && k < nbp) Ls [k * ldl + j] = v ;
        }
    }
    __syncthreads () ;
    tick (0) ;
    // inverse of the 16x16 diagonal blocks: wave b; lane r < 16 holds row r of
    // the block in registers and solves column r of the inverse by forward
    // substitution, the (lane-uniform) multipliers travelling by v_readlane
    trsm_diag_inverses (Ls, ldl, Wd, nblk, lane, wave, tick) ;
    __syncthreads () ;
    tick (2) ;
    // The accumulator layout of v_mfma_f64_16x16x4 (lane (lr,lk) holds columns
    // lk + 4 r, r = 0..3) IS its A-operand layout for the k-steps s = r (k = 4 s +
    // lk): a solved block X_i and the intermediate B_j - sum feed the next MFMAs
    // straight from registers, no LDS round trip, no barrier.
    d4 xr [4] ;
    trsm_solve_rows<CX> (bj, nblk, Ls, ldl, Wd, lane, nvalid, rok, nb, B, brow, lda, tick, xr) ;
    if constexpr (TIMED) { if (tid == 0) for (int q = 0 ; q < 8 ; q++) tim [q] = tc [q] ; }
}

// ---- thin fronts: the whole front in LDS, one pass over HBM -------------------
// For thin supernodes (nsrow <= SM_MAX; 98 % of the supernodes of a 2D / circuit
// problem, the leaf levels of every problem) the level-batched generic kernels
// cost a dozen launches and several HBM round trips per front.  Here one
// workgroup (ONE WAVE when nsrow <= 64, four otherwise) builds the front in LDS
// as a packed lower triangle, eliminates its nscol columns and streams the
// result out: HBM traffic = A entries + children's contribution blocks in,
// panel + contribution block out, every byte once and contiguous.
// Reference steps: t_cholmod_super_numeric.c:305-431 (assemble), :743-772
// (relative-map scatter), :864-867 / :997-1002 (dpotrf / dtrsm) and the
// dsyrk/dgemm its ancestors would have pulled (:682-717) as the contribution block.
//  (1) the row list, the column pointers of A and the first batch of the first
//      child's contribution block are requested before anything is waited for;
//  (2) children's contribution blocks are packed triangles: a flat, fully
//      coalesced stream, eight loads per thread in flight, scatter-added into the
//      LDS front through the relative map (one barrier per child: two children
//      may hit the same entry, two entries of one child never do);
//  (3) 16-column panels: wave 0 eliminates the panel with lane = row -- its 64
//      rows ride through the same right-looking step, pivots and multipliers by
//      v_readlane, the chain per column is rcp + Newton -> mul -> fma (LDL' on the
//      unscaled columns; sqrt/rsqrt once per panel, 16 pivots side by side); the
//      other waves solve their rows against the published 16x16 block;
//      finished columns go to Lx straight from registers;
//  (4) the trailing update runs on the matrix cores out of the packed front; the
//      one after the last panel IS the contribution block and is written to HBM
//      from the accumulators (packed), never back to LDS.
__host__ __device__ inline size_t thin_front_lds_bytes (int ns_max)
{
    int nsp = (ns_max + 1) & ~1 ;
    return (size_t) (ns_max * (ns_max + 1) / 2) * sizeof (double) + 3 * (size_t) nsp * sizeof (i32) ;
}
// (i,j) of the e-th entry of a packed lower triangle of order m
__device__ __forceinline__ void tri_decode (int e, int m, int &i, int &j)
{
    float b = (float) (2 * m + 1) ;
    int jj = (int) ((b - __builtin_sqrtf (b * b - 8.0f * (float) e)) * 0.5f) ;
    jj = jj < 0 ? 0 : (jj > m - 1 ? m - 1 : jj) ;
    int st = jj * m - ((jj * (jj - 1)) >> 1) ;             // first entry of column jj
    while (st > e) { jj-- ; st -= m - jj ; }
    while (st + (m - jj) <= e) { st += m - jj ; jj++ ; }
    j = jj ; i = jj + (e - st) ;
}
// LDS-only barrier of the thin-front workgroup: __syncthreads() would also wait
// for the global stores in flight (the finished panel columns on their way to
// Lx), two microseconds per panel.  One wave: LDS operations of a wave are
// ordered, nothing to wait for.
template <int NW> __device__ __forceinline__ void tf_barrier ()
{
    if constexpr (NW == 1) { asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ; }
    else asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory") ;
}
// One PW-column panel (PW = 4, 8, 12, 16 at compile time: no branch inside the
// elimination; columns pc .. PW-1 do not exist and behave as identity columns).
// Every wave carries the PW diagonal rows in its lanes 0 .. PW-1 (the same values
// in all waves) and 64 - PW rows of its own behind them, so each wave runs the
// whole elimination by itself: no wave waits for another, no barrier inside the
// panel.  Pivots and multipliers travel through a 64-double LDS scratch of the wave
// (one ds_write_b64 by every lane, broadcast ds_read_b128 pairs by everybody): (PW-1-c)
// + 7 vector instructions per column where two v_readlane per value cost 3 (PW-1-c)
// + 10 (thin fronts of the 2D 1259^2 problem 1.012 -> 0.982 ms; the kernel is bound
// by instruction issue).  LDS operations of one wave execute in order, so the read
// behind the write needs no barrier; bc = this wave's scratch (128 doubles, 16-byte
// aligned).  On return a [] holds the finished entries L(row, c0 + c) of this
// thread's row; `own` tells whether this thread is the one that stores them
// (diagonal rows: wave 0 only).
typedef double d2 __attribute__((ext_vector_type(2))) ;
// CX (a complex front in its own storage): only the even twin columns go to Lx -- Lp points at stored column c0 / 2, stored
// column (c0 + c) / 2 follows at (c / 2) ns -- and the imaginary part of a diagonal entry (row c + 1 of an even column c, a
// rounding residue of the embedded elimination) as the exact zero zpotrf leaves there (as pf_store<CX>).
template <int PW, int NW, bool CX = false>
__device__ __forceinline__ void tf_panel (double *F, int *s_fail, int ns, int c0, int pc,
    int lane, int wave, int &fail, double *Lp, double *bc)
{
    double a [PW] ;
    const int row = lane < PW ? c0 + lane : c0 + PW + (64 - PW) * wave + (lane - PW) ;
    const bool rok = row < ns ;
    const bool own = rok && (lane >= PW || wave == 0) ;
    const int rr = rok ? row : ns - 1 ;
    {
        int o = tri_col (c0, ns) ;
#pragma unroll
        for (int c = 0 ; c < PW ; c++)
        {
            double v = F [o + rr] ;
            v = (rok && row >= c0 + c) ? v : 0.0 ;
            a [c] = (c < pc) ? v : (lane == c ? 1.0 : 0.0) ;
            if (c + 1 < pc) o += ns - (c0 + c) - 1 ;
        }
    }
    // Every wave holds its own copy of the PW diagonal rows, and wave 0 writes the FINISHED diagonal
    // rows back into F at the end of its elimination: nobody may be that far before everybody has
    // its copy.  Without this barrier a wave that the scheduler held back between two of the loads
    // above read rows that wave 0 had already scaled (seen in world-4 runs with four processes on
    // one GPU: the last columns of one wave's rows off by the factor of a wrong pivot, one run in
    // three; never in a single process -- round 3, tests/test_dist.py, tools/flaky_dist.sh).
    if constexpr (NW > 1) tf_barrier<NW> () ;
    double dv = 1.0 ;                                       // lane c keeps the pivot of column c
    const bool odd_lane = (lane & 1) != 0 ;                 // (= odd row: c0, PW and 64 - PW are even)
    if constexpr (CX)
    {
        // complex columns (as pf_eliminate<PHI>): the pair (c, c + 1) in one step, the odd column never eliminated -- it is
        // the rotation of the even one and is rebuilt below; a trailing even column 2 j takes a -= t x_j + s y_j
#pragma unroll
        for (int c = 0 ; c < PW ; c += 2)
        {
            bc [lane] = a [c] ;
            asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
            double u [PW] ;
#pragma unroll
            for (int c2 = c ; c2 < PW ; c2 += 2)
            {
                d2 v = *(const d2 *) (bc + c2) ;
                u [c2] = v.x ; u [c2 + 1] = v.y ;
            }
            asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
            double d = u [c] ;
            double x = __builtin_amdgcn_rcp (d) ;
            double e = __builtin_fma (-d, x, 1.0) ;
            x = __builtin_fma (x, e, x) ;
            double t = a [c] * x ;
            double sw = lane_xor1_f64 (t) ;
            double sg = odd_lane ? -sw : sw ;
#pragma unroll
            for (int c2 = c + 2 ; c2 < PW ; c2 += 2) { a [c2] = __builtin_fma (-t, u [c2], a [c2]) ; a [c2] = __builtin_fma (-sg, u [c2 + 1], a [c2]) ; }
            if (lane == c) dv = d ;
            __builtin_amdgcn_sched_barrier (0) ;
        }
    }
    else
#pragma unroll
    for (int c = 0 ; c < PW ; c++)
    {
        bc [lane] = a [c] ;                                 // (every lane: no exec mask to set up; rows 0 .. PW-1 matter)
        asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
        double u [PW] ;
#pragma unroll
        for (int c2 = c & ~1 ; c2 < PW ; c2 += 2)
        {
            d2 v = *(const d2 *) (bc + c2) ;
            u [c2] = v.x ; u [c2 + 1] = v.y ;
        }
        asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
        double d = u [c] ;
        double x = __builtin_amdgcn_rcp (d) ;
        double e = __builtin_fma (-d, x, 1.0) ;
        x = __builtin_fma (x, e, x) ;
        double t = a [c] * x ;                              // u(row,c) / d
#pragma unroll
        for (int c2 = c + 1 ; c2 < PW ; c2++) a [c2] = __builtin_fma (-t, u [c2], a [c2]) ;
        if (lane == c) dv = d ;
        // (without this the scheduler defers the updates of columns c+2 .. and keeps every
        // column's multipliers alive: 235 registers)
        __builtin_amdgcn_sched_barrier (0) ;
    }
    // first failing pivot of the panel: lane c holds pivot c (NaN does not trip)
    if (fail < 0)
    {
        unsigned long long bad = __ballot (lane < pc && dv <= 0.0) ;
        if (bad) fail = c0 + (int) __builtin_ctzll (bad) ;
    }
    double r, ri ;
    sqrt_rsqrt (dv, r, ri) ;
    bc [lane] = r ; bc [64 + lane] = ri ;
    asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
#pragma unroll
    for (int c = 0 ; c < PW ; c += 2)
    {
        d2 rv = *(const d2 *) (bc + c), iv = *(const d2 *) (bc + 64 + c) ;
        a [c] = (lane == c) ? rv.x : a [c] * iv.x ;
        if constexpr (CX)
        {
            // the odd column: the rotation of the finished even one (its diagonal entry = the even column's)
            double sw = lane_xor1_f64 (a [c]) ;
            a [c + 1] = odd_lane ? sw : -sw ;
        }
        else a [c + 1] = (lane == c + 1) ? rv.y : a [c + 1] * iv.y ;
        if (fail >= 0 && c0 + c >= fail) a [c] = 0.0 ;
        if (fail >= 0 && c0 + c + 1 >= fail) a [c + 1] = 0.0 ;
    }
    asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
    if (wave == 0 && lane == 0) *s_fail = fail ;
    if (own)
    {
        int o = tri_col (c0, ns) ;
        double *Lr = Lp + row ;
#pragma unroll
        for (int c = 0 ; c < PW ; c++)
        {
            if (c < pc && row >= c0 + c)
            {
                F [o + row] = a [c] ;
                if constexpr (CX) { if (!(c & 1) && (fail < 0 || c0 + c < fail)) Lr [(i64) (c >> 1) * ns] = (row == c0 + c + 1) ? 0.0 : a [c] ; }
                else if (fail < 0 || c0 + c < fail) Lr [(i64) c * ns] = a [c] ;
            }
            if (c + 1 < pc) o += ns - (c0 + c) - 1 ;
        }
    }
}
// G tiles (i0, j0 + 16 g) of the trailing update, G independent MFMA chains:
// C -= L(:, c0 .. c0+pc) L(:, same)'.  TO_CB: the result is the contribution block
// (packed, HBM); otherwise it goes back into the LDS front.  All offsets come from
// 24-bit multiplies and increments (tri_col (j+4) - tri_col (j) = 4 m - 4 j - 10).
__device__ __forceinline__ int tri_col24 (int j, int m) { return __mul24 (j, m) - (__mul24 (j, j + 1) >> 1) ; }
template <int G, int KS, bool TO_CB, bool CX = false>
__device__ __forceinline__ void tf_tiles (double *F, int ns, int c0, int pc, int i0, int j0, int lane,
    double *Co, int nc, int ncb, bool czero)
{
    const int lr = lane & 15, lk = lane >> 4 ;
    const int i = i0 + lr ;
    const int ir = i < ns ? i : ns - 1 ;
    int koff [KS] ;                                         // column of k-step s for this lane (k = 4 s + lk)
    double af [KS] ;
#pragma unroll
    for (int s4 = 0 ; s4 < KS ; s4++)
    {
        int k = 4 * s4 + lk ;
        koff [s4] = tri_col24 (c0 + (k < pc ? k : pc - 1), ns) ;
        double v = F [koff [s4] + ir] ;
        af [s4] = (k < pc) ? -v : 0.0 ;
    }
    // C entries of this lane: (i, j0 + 16 g + lk + 4 r); offsets in the packed front
    d4 acc [G] ;
    if (czero)
    {
        // a front without children: nothing but zeros behind the panel columns
#pragma unroll
        for (int g = 0 ; g < G ; g++) acc [g] = (d4) {0.0, 0.0, 0.0, 0.0} ;
    }
    else
    {
        int j = j0 + lk ;
        int o = tri_col24 (j < ns ? j : ns - 1, ns) ;
#pragma unroll
        for (int g = 0 ; g < G ; g++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                bool ok = (i < ns) && (j <= i) ;
                double v = F [ok ? o + i : 0] ;
                acc [g][r] = ok ? v : 0.0 ;
                o += 4 * ns - 4 * j - 10 ; j += 4 ;
            }
    }
#pragma unroll
    for (int s4 = 0 ; s4 < KS ; s4++)
    {
#pragma unroll
        for (int g = 0 ; g < G ; g++)
        {
            int jr = j0 + 16 * g + lr ; jr = jr < ns ? jr : ns - 1 ;
            // (k >= pc: the clamped column pc-1 is read again and meets af = 0; it is a column of
            // the real product, so even a non-finite entry reaches nothing it does not reach anyway)
            double bv = F [koff [s4] + jr] ;
            acc [g] = __builtin_amdgcn_mfma_f64_16x16x4f64 (bv, af [s4], acc [g], 0, 0, 0) ;
        }
    }
    if constexpr (TO_CB)
    {
        int j = j0 + lk - nc ;                              // column / row inside the contribution block
        const int ic = i - nc ;
        int o = tri_col24 (j < ncb ? j : ncb - 1, ncb) ;
#pragma unroll
        for (int g = 0 ; g < G ; g++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                // (CX: the even columns of the block, a square with ld = ncb: stored column j / 2)
                if constexpr (CX) { if (i < ns && j <= ic && !(j & 1)) Co [(j >> 1) * ncb + ic] = acc [g][r] ; }
                else if (i < ns && j <= ic) Co [o + ic] = acc [g][r] ;
                o += 4 * ncb - 4 * j - 10 ; j += 4 ;
            }
    }
    else
    {
        int j = j0 + lk ;
        int o = tri_col24 (j < ns ? j : ns - 1, ns) ;
#pragma unroll
        for (int g = 0 ; g < G ; g++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                if (i < ns && j <= i) F [o + i] = acc [g][r] ;
                o += 4 * ns - 4 * j - 10 ; j += 4 ;
            }
    }
}
template <int KS, bool TO_CB, bool CX = false>
__device__ __forceinline__ void tf_tile_row (double *F, int ns, int c0, int pc, int t0, int I, int lane,
    double *Co, int nc, int ncb, bool czero)
{
    const int i0 = t0 + 16 * I ;
    for (int J = 0 ; J <= I ; )
    {
        int left = I + 1 - J, j0 = t0 + 16 * J ;
        if (left >= 4) { tf_tiles<4, KS, TO_CB, CX> (F, ns, c0, pc, i0, j0, lane, Co, nc, ncb, czero) ; J += 4 ; }
        else if (left >= 2) { tf_tiles<2, KS, TO_CB, CX> (F, ns, c0, pc, i0, j0, lane, Co, nc, ncb, czero) ; J += 2 ; }
        else { tf_tiles<1, KS, TO_CB, CX> (F, ns, c0, pc, i0, j0, lane, Co, nc, ncb, czero) ; J += 1 ; }
    }
}
// CX: a complex front in its own storage (the twin's index space; the front in LDS is the whole twin block, assembled from
// phi (S) like any real front).  The children's blocks hold their even twin columns only (squares with ld = ncb, written by
// the generic kernels or by this one): every stored entry (i, 2 jj) is added at its place AND as the entry of the odd
// column it rotates into -- (i ^ 1, 2 jj + 1), negated for odd i -- where that lies in the lower triangle.  The panel and
// the contribution block leave as even columns (tf_panel, tf_tiles).
template <int NW, bool TIMED = false, int MINW = (NW == 1 ? 6 : 2), bool CX = false>
__global__ void __launch_bounds__(64 * NW, MINW) k_thin_front (const i32 *fronts,
    const FrontD *frl, const i64 *sp01, const ChildD *cd, const i32 *relmap, const i64 *Ls,
    const i64 *Sp, const i64 *Snz, const i64 *Si, const double *Sx, double beta,
    double *Lx, double *CB, i32 *info, int ns_max, i64 *amap, int mapped, long long *tim = nullptr)
{
    long long tc [10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0 ;
    auto tick = [&] (int slot) { if constexpr (TIMED) { long long t = __builtin_readcyclecounter () ; tc [slot] += t - t_prev ; t_prev = t ; } } ;
    if constexpr (TIMED) t_prev = __builtin_readcyclecounter () ;
    constexpr int NT = 64 * NW ;
    constexpr int NLD = 8 ;                                 // child entries in flight per thread and buffer
    constexpr int NRM = ((NW == 1 ? 64 : SM_MAX) + NT - 1) / NT ;   // relative-map entries per thread (one wave: fronts, hence child blocks, of <= 64 rows)
    extern __shared__ __attribute__((aligned(16))) double tf_lds [] ;
    double *F = tf_lds ;                                    // packed lower triangle of the front
    const int nsp = (ns_max + 1) & ~1 ;
    i32 *rows_l = (i32 *) (F + ns_max * (ns_max + 1) / 2) ; // the front's row list
    i32 *rm_l = rows_l + nsp ;                              // relative maps of two children (ping-pong)
    __shared__ int s_fail ;
    __shared__ __attribute__((aligned(16))) double s_bc [128 * NW] ;    // panel broadcast scratch
    // frl / sp01: the launch's fronts in block order -- descriptor and range of S of block b
    // at index b, so that the first level of loads depends on blockIdx only (the chain
    // front id -> descriptor -> column pointers -> data was four round trips deep)
    const FrontD &f = frl [blockIdx.x] ;
    const int ns = f.nsrow, nc = f.nscol, ncb = f.ncb, k1 = f.k1 ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6 ;
    const i64 psx = f.psx, cbo = f.cb ;
    const int T = ns * (ns + 1) / 2 ;
    const int cbeg = f.child_begin, cend = f.child_end ;
    // ---- (1) requests first: row list, column pointers of A, the first child
    i64 p0 = 0, p1 = 0 ;
    // A of a packed S whose map is known (every factorization of a resident S after the
    // first): the entries of the front's columns are one contiguous range of S, lanes
    // stride it; amap [p] = -2 - (offset in the packed front), -1 = not in L
    const bool flat = mapped && !Snz && f.assemble ;
    const bool asm_col = !flat && f.assemble && tid < nc ;
    const i64 rowv = (!flat && tid < ns) ? Ls [f.psi + tid] : 0 ;     // the row list: only the search path needs it
    i64 fq = -1 ; double fx = 0.0 ;
    if (flat)
    {
        p0 = sp01 [2 * blockIdx.x] + tid ; p1 = sp01 [2 * blockIdx.x + 1] ;
        if (p0 < p1) { fq = amap [p0] ; fx = Sx [p0] ; }
    }
    if (asm_col)
    {
        i64 col = (i64) k1 + tid ;
        p0 = Sp [col] ; p1 = Snz ? p0 + Snz [col] : Sp [col + 1] ;
    }
    // children stream: chunk = NLD * NT consecutive entries of one child's packed CB
    // (or NLD whole columns of a square one, see below); two register buffers
    struct Cur { int ci, base, m, tot, sq ; const double *src ; i64 rel ; } ;
    auto child_at = [&] (int ci, Cur &c)
    {
        const ChildD &cf = cd [ci] ;
        c.ci = ci ; c.base = 0 ; c.m = cf.ncb ; c.sq = CX || cf.cbp != 1 ;
        c.tot = CX ? (cf.ncb >> 1) * cf.ncb : cf.cbp == 1 ? cf.ncb * (cf.ncb + 1) / 2 : cf.ncb * cf.ncb ;
        c.src = CB + cf.cb ; c.rel = cf.rel ;
    } ;
    auto advance = [&] (Cur &c) -> bool                      // next chunk; false when the stream is over
    {
        c.base += NLD * NT ;
        if (c.base < c.tot) return true ;
        if (c.ci + 1 >= cend) return false ;
        child_at (c.ci + 1, c) ;
        return true ;
    } ;
    auto issue = [&] (const Cur &c, double (&v) [NLD], i32 (&rmv) [NRM])
    {
        if (c.base == 0)
        {
            const i32 *rm = relmap + c.rel ;
#pragma unroll
            for (int q = 0 ; q < NRM ; q++) { int e = tid + NT * q ; rmv [q] = rm [e < c.m ? e : c.m - 1] ; }
        }
        // a thread takes NLD CONSECUTIVE entries of the chunk (the wave still covers one
        // contiguous 4 KB piece): (i, j) is decoded once per chunk and then advances by one
        // -- a stride of NT entries crosses several short columns and costs a loop of
        // 25 - 60 instructions per entry.  Waves wholly past the end of a small block
        // (a leaf's 105 entries are 14 threads' worth) neither load nor decode.
        // (the last chunk of a block is dealt nq <= NLD entries per thread, so that all lanes
        // share it: a leaf's 105 entries are two per thread, not eight for 14 threads)
        int nq = (c.tot - c.base + NT - 1) / NT ; nq = nq < NLD ? nq : NLD ;
        const int e0 = c.base + nq * tid ;
        if (__builtin_amdgcn_readfirstlane (e0) < c.tot)
        {
#pragma unroll
            for (int q = 0 ; q < NLD ; q++)
            {
                if (q >= nq) break ;
                int e = e0 + q ;
                v [q] = c.src [e < c.tot ? e : c.tot - 1] ;
            }
        }
    } ;
    auto consume = [&] (const Cur &c, const double (&v) [NLD], const i32 (&rmv) [NRM])
    {
        i32 *rmc = rm_l + (c.ci & 1) * nsp ;
        if (c.base == 0)
        {
            // a new child: its map goes to LDS; the barrier also keeps two children
            // from adding into the same entry at the same time
#pragma unroll
            for (int q = 0 ; q < NRM ; q++) { int e = tid + NT * q ; if (e < c.m) rmc [e] = rmv [q] ; }
            tf_barrier<NW> () ;
        }
        int nq = (c.tot - c.base + NT - 1) / NT ; nq = nq < NLD ? nq : NLD ;
        int e = c.base + nq * tid, i, j ;
        if (__builtin_amdgcn_readfirstlane (e) >= c.tot) return ;
        const int m = c.m ;
        if (c.sq) { j = e / m ; i = e - j * m ; if constexpr (CX) j *= 2 ; }
        else tri_decode (e < c.tot ? e : c.tot - 1, m, i, j) ;
#pragma unroll
        for (int q = 0 ; q < NLD ; q++)
        {
            if (q >= nq) break ;
            // (ds_add_f64: one LDS instruction instead of read / add / write; two entries of
            // one child never meet in the same entry, the barrier above separates children)
            if (e < c.tot && i >= j)
                (void) __hip_atomic_fetch_add (&F [tri_col24 (rmc [j], ns) + rmc [i]], v [q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ;
            if constexpr (CX)
            {
                // the odd twin column this entry rotates into: (i ^ 1, j + 1), negated for odd i
                const int i2 = i ^ 1 ;
                if (e < c.tot && i2 >= j + 1)
                    (void) __hip_atomic_fetch_add (&F [tri_col24 (rmc [j + 1], ns) + rmc [i2]], (i & 1) ? -v [q] : v [q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ;
            }
            e++ ; i++ ;
            const bool wrap = i >= m ;                      // next column (branch-free)
            j = wrap ? j + (CX ? 2 : 1) : j ;
            i = wrap ? (c.sq ? 0 : j) : i ;
            if (j >= m) { j = m - (CX ? 2 : 1) ; i = m - 1 ; e = c.tot ; }     // past the last entry
        }
    } ;
    double vA [NLD], vB [NLD] ;
    i32 rmA [NRM], rmB [NRM] ;
    Cur cA, cB ;
    bool haveA = cbeg < cend, haveB = false ;
    if (haveA) { child_at (cbeg, cA) ; issue (cA, vA, rmA) ; }
    for (int e = tid ; e < T ; e += NT) F [e] = 0.0 ;
    if (tid == 0) s_fail = -1 ;
    if (!flat && tid < ns) rows_l [tid] = (i32) rowv ;
    tf_barrier<NW> () ;
    tick (0) ;
    // ---- A into the panel columns (ASSIGN semantics, entries outside the pattern dropped)
    if (flat)
    {
        if (fq <= -2) F [(int) (-2 - fq)] = fx ;
        for (i64 p = p0 + NT ; p < p1 ; p += NT)
        {
            i64 q = amap [p] ;
            if (q <= -2) F [(int) (-2 - q)] = Sx [p] ;
        }
        if (beta != 0.0)
        {
            tf_barrier<NW> () ;
            if (tid < nc) F [tri_col (tid, ns) + tid] += beta ;
        }
    }
    if (asm_col)
    {
        const int k = tid ;
        const i64 col = (i64) k1 + k ;
        double *Fc = F + tri_col (k, ns) ;
        for (i64 p = p0 ; p < p1 ; p += 4)
        {
            i64 ii [4] ; double xx [4] ;
#pragma unroll
            for (int u = 0 ; u < 4 ; u++)
            {
                i64 q = p + u < p1 ? p + u : p1 - 1 ;
                ii [u] = Si [q] ; xx [u] = Sx [q] ;
            }
#pragma unroll
            for (int u = 0 ; u < 4 ; u++)
            {
                if (p + u >= p1 || ii [u] < col) continue ;
                int lo = k, hi = ns ;
                while (lo < hi) { int mid = (lo + hi) >> 1 ; if (rows_l [mid] < (i32) ii [u]) lo = mid + 1 ; else hi = mid ; }
                if (lo < ns && rows_l [lo] == (i32) ii [u])
                {
                    Fc [lo] = xx [u] ;
                    if (!Snz) amap [p + u] = -2 - (i64) (tri_col (k, ns) + lo) ;
                }
            }
        }
        if (beta != 0.0) Fc [k] += beta ;
    }
    tick (1) ;
    // ---- (2) children: consume one buffer while the next chunk is in flight
    while (haveA)
    {
        cB = cA ; haveB = advance (cB) ;
        if (haveB) issue (cB, vB, rmB) ;
        consume (cA, vA, rmA) ;
        if (!haveB) break ;
        cA = cB ; haveA = advance (cA) ;
        if (haveA) issue (cA, vA, rmA) ;
        consume (cB, vB, rmB) ;
    }
    tf_barrier<NW> () ;
    tick (2) ;
    // ---- (3), (4): panels of up to 16 columns
    int fail = -1 ;
    for (int c0 = 0 ; c0 < nc ; c0 += TF_PW)
    {
        const int pc = nc - c0 < TF_PW ? nc - c0 : TF_PW ;
        double *Lp = Lx + psx + (i64) (CX ? c0 >> 1 : c0) * ns ;
        if (pc <= 4) tf_panel<4, NW, CX> (F, &s_fail, ns, c0, pc, lane, wave, fail, Lp, s_bc + 128 * wave) ;
        else if (pc <= 8) tf_panel<8, NW, CX> (F, &s_fail, ns, c0, pc, lane, wave, fail, Lp, s_bc + 128 * wave) ;
        else if (pc <= 12) tf_panel<12, NW, CX> (F, &s_fail, ns, c0, pc, lane, wave, fail, Lp, s_bc + 128 * wave) ;
        else tf_panel<16, NW, CX> (F, &s_fail, ns, c0, pc, lane, wave, fail, Lp, s_bc + 128 * wave) ;
        tick (3) ;
        tf_barrier<NW> () ;
        tick (4) ;
        fail = s_fail ;
        if (fail >= 0) break ;
        // trailing update of rows / columns t0 .. ns-1 with the pc panel columns; tile
        // rows are dealt in pairs (p, nd-1-p) so that every wave gets the same work
        const int t0 = c0 + pc ;
        const bool last = t0 >= nc ;
        const int nd = (ns - t0 + 15) >> 4 ;
        double *Co = CB + cbo ;
        const bool czero = last && c0 == 0 && cbeg == cend ;    // leaf front, single panel
        for (int pr = wave ; 2 * pr < nd ; pr += NW)
        {
            for (int side = 0 ; side < 2 ; side++)
            {
                int I = side ? nd - 1 - pr : pr ;
                if (side && I == pr) break ;
                if (last)
                {
                    if (pc <= 4) tf_tile_row<1, true, CX> (F, ns, c0, pc, t0, I, lane, Co, nc, ncb, czero) ;
                    else if (pc <= 8) tf_tile_row<2, true, CX> (F, ns, c0, pc, t0, I, lane, Co, nc, ncb, czero) ;
                    else if (pc <= 12) tf_tile_row<3, true, CX> (F, ns, c0, pc, t0, I, lane, Co, nc, ncb, czero) ;
                    else tf_tile_row<4, true, CX> (F, ns, c0, pc, t0, I, lane, Co, nc, ncb, czero) ;
                }
                else        // (only with more than 16 columns: the panel before the last is full)
                    tf_tile_row<4, false, CX> (F, ns, c0, pc, t0, I, lane, Co, nc, ncb, czero) ;
            }
        }
        if (!last) tf_barrier<NW> () ;
        tick (5) ;
    }
    if (fail >= 0)
    {
        if (tid == 0) info [fronts [blockIdx.x]] = fail + 1 ;
        // the ancestors of a failed front compute values nobody keeps; give them zeros
        const int tot = CX ? (ncb >> 1) * ncb : ncb * (ncb + 1) / 2 ;
        for (int e = tid ; e < tot ; e += NT) CB [cbo + e] = 0.0 ;
    }
    if constexpr (TIMED) { if (tid == 0 && blockIdx.x == gridDim.x / 2) for (int q = 0 ; q < 10 ; q++) tim [q] = tc [q] ; }
}

// ---- leaf fronts, two per wave -----------------------------------------------------
// The thin-front kernel is bound by instruction issue, and a leaf front of a 2D / circuit
// problem (26 rows, 12 columns) uses 26 of the 64 lanes of its wave.  Fronts without
// children, of <= 32 rows and <= 16 columns (one panel), therefore go two to a wave once
// the assembly map of the resident S exists: lanes 0..31 carry the rows of one front,
// lanes 32..63 those of the other, and every vector instruction of the scatter of A, of
// the elimination and of the stores of L serves both.  What makes this possible is the
// LDS broadcast of tf_panel: the multipliers of a column are read from an address that
// differs between the two halves, where a v_readlane would be wave-wide.  The
// contribution blocks are then computed front by front on the matrix cores (tf_tiles,
// uniform parameters).  Same packed-triangle LDS layout and same map encoding as
// k_thin_front, which runs the first factorization of a resident S (and records the map).
// LDS: the panel columns of the two packed fronts (LP_T doubles each: the launch's largest
// nscol * nsrow - nscol (nscol - 1) / 2, rounded up to a multiple of 32) + 128 doubles of
// broadcast scratch -- 5 KB per wave for 26 x 12 leaves, so that the occupancy is set by
// the registers (the kernel is latency-bound: four dependent loads in front of the work).
template <int PW, int MINW>
__global__ void __launch_bounds__(64, MINW) k_leaf_pair (const i32 *fronts, int nfronts, const FrontD *frl,
    const i64 *sp01, const double *Sx, const i64 *amap, double beta, double *Lx, double *CB, i32 *info, int LP_T)
{
    extern __shared__ __attribute__((aligned(16))) double lp_lds [] ;
    double *Fs = lp_lds ;
    double *bc = lp_lds + 2 * LP_T ;
    const int lane = threadIdx.x, h = lane >> 5, l = lane & 31 ;
    const int b = blockIdx.x ;
    const bool twin = 2 * b + 1 < nfronts ;
    const int ia = 2 * b, ib = twin ? 2 * b + 1 : 2 * b ;
    const FrontD &fa = frl [ia], &fb = frl [ib] ;
    const int nsA = fa.nsrow, ncA = fa.nscol, nsB = fb.nsrow, ncB = fb.nscol ;
    const bool live = h == 0 || twin ;
    const int ns = h ? nsB : nsA, nc = h ? ncB : ncA ;
    const i64 psx = h ? fb.psx : fa.psx ;
    const bool asmb = live && (h ? fb.assemble : fa.assemble) != 0 ;
    double *F = Fs + h * LP_T ;
    // ---- requests first: this half's range of S and its first entries
    i64 p0, p1 ;
    {
        const i64 a0 = sp01 [2 * ia], a1 = sp01 [2 * ia + 1], b0 = sp01 [2 * ib], b1 = sp01 [2 * ib + 1] ;
        p0 = (h ? b0 : a0) + l ; p1 = h ? b1 : a1 ;
    }
    if (!asmb) p1 = p0 ;
    i64 fq = -1 ; double fx = 0.0 ;
    if (p0 < p1) { fq = amap [p0] ; fx = Sx [p0] ; }
    // ---- zero the panel columns of both packed fronts
    for (int k = 0 ; k < LP_T ; k += 32) F [l + k] = 0.0 ;
    asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
    // ---- A through the map (ASSIGN semantics, entries outside the pattern have no map entry)
    if (fq <= -2) F [(int) (-2 - fq)] = fx ;
    p0 += 32 ;
    while (__any (p0 < p1))
    {
        if (p0 < p1)
        {
            const i64 q = amap [p0] ;
            if (q <= -2) F [(int) (-2 - q)] = Sx [p0] ;
        }
        p0 += 32 ;
    }
    asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
    if (beta != 0.0 && asmb && l < nc) F [l * ns - ((l * (l - 1)) >> 1)] += beta ;
    asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
    // ---- the panel: lane = row l of its half's front
    const bool rok = live && l < ns ;
    const int rr = l < ns ? l : ns - 1 ;
    double a [PW] ;
    {
        int o = 0 ;
#pragma unroll
        for (int c = 0 ; c < PW ; c++)
        {
            const bool in = c < nc ;
            double v = F [in ? o + rr : 0] ;
            v = (in && rok && l >= c) ? v : 0.0 ;
            a [c] = in ? v : (l == c ? 1.0 : 0.0) ;
            o += ns - c - 1 ;
        }
    }
    const double *bch = bc + 32 * h ;
    double dv = 1.0 ;
#pragma unroll
    for (int c = 0 ; c < PW ; c++)
    {
        bc [lane] = a [c] ;
        asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
        double u [PW] ;
#pragma unroll
        for (int c2 = c & ~1 ; c2 < PW ; c2 += 2)
        {
            d2 v = *(const d2 *) (bch + c2) ;
            u [c2] = v.x ; u [c2 + 1] = v.y ;
        }
        asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
        const double d = u [c] ;
        double x = __builtin_amdgcn_rcp (d) ;
        const double e = __builtin_fma (-d, x, 1.0) ;
        x = __builtin_fma (x, e, x) ;
        const double t = a [c] * x ;
#pragma unroll
        for (int c2 = c + 1 ; c2 < PW ; c2++) a [c2] = __builtin_fma (-t, u [c2], a [c2]) ;
        if (l == c) dv = d ;
        __builtin_amdgcn_sched_barrier (0) ;
    }
    // first failing pivot of each front (NaN does not trip)
    const unsigned long long bad = __ballot (live && l < nc && dv <= 0.0) ;
    const unsigned int badA = (unsigned int) bad, badB = (unsigned int) (bad >> 32) ;
    const int failA = badA ? (int) __builtin_ctz (badA) : -1 ;
    const int failB = badB ? (int) __builtin_ctz (badB) : -1 ;
    const int fail = h ? failB : failA ;
    double r, ri ;
    sqrt_rsqrt (dv, r, ri) ;
    bc [lane] = r ; bc [64 + lane] = ri ;
    asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
#pragma unroll
    for (int c = 0 ; c < PW ; c += 2)
    {
        d2 rv = *(const d2 *) (bch + c), iv = *(const d2 *) (bch + 64 + c) ;
        a [c] = (l == c) ? rv.x : a [c] * iv.x ;
        a [c + 1] = (l == c + 1) ? rv.y : a [c + 1] * iv.y ;
        if (fail >= 0 && c >= fail) a [c] = 0.0 ;
        if (fail >= 0 && c + 1 >= fail) a [c + 1] = 0.0 ;
    }
    // finished columns: to Lx from the registers, and back into the LDS fronts (operands of
    // the contribution blocks)
    {
        int o = 0 ;
        double *Lr = Lx + psx + l ;
#pragma unroll
        for (int c = 0 ; c < PW ; c++)
        {
            if (rok && c < nc && l >= c)
            {
                F [o + l] = a [c] ;
                if (fail < 0 || c < fail) Lr [(i64) c * ns] = a [c] ;
            }
            o += ns - c - 1 ;
        }
    }
    asm volatile ("" ::: "memory") ; __builtin_amdgcn_wave_barrier () ;
    // ---- contribution blocks, front by front (uniform parameters, all 64 lanes)
    for (int hh = 0 ; hh < (twin ? 2 : 1) ; hh++)
    {
        const FrontD &f = hh ? fb : fa ;
        const int ns_ = f.nsrow, nc_ = f.nscol, ncb_ = f.ncb ;
        const int fail_ = hh ? failB : failA ;
        double *Fh = Fs + hh * LP_T ;
        double *Co = CB + f.cb ;
        if (fail_ >= 0)
        {
            if (lane == 0) info [fronts [hh ? ib : ia]] = fail_ + 1 ;
            const int tot = ncb_ * (ncb_ + 1) / 2 ;
            for (int e = lane ; e < tot ; e += 64) Co [e] = 0.0 ;
            continue ;
        }
        const int nd = (ncb_ + 15) >> 4 ;
        for (int I = 0 ; I < nd ; I++)
        {
            if (nc_ <= 4) tf_tile_row<1, true> (Fh, ns_, 0, nc_, nc_, I, lane, Co, nc_, ncb_, true) ;
            else if (nc_ <= 8) tf_tile_row<2, true> (Fh, ns_, 0, nc_, nc_, I, lane, Co, nc_, ncb_, true) ;
            else if (nc_ <= 12) tf_tile_row<3, true> (Fh, ns_, 0, nc_, nc_, I, lane, Co, nc_, ncb_, true) ;
            else tf_tile_row<4, true> (Fh, ns_, 0, nc_, nc_, I, lane, Co, nc_, ncb_, true) ;
        }
    }
}

// ---- dense update  C -= A * B'  (fp64 MFMA 16x16x4 tiles) -------------------
// The contraction the reference performs with dsyrk/dgemm per descendant
// (:682-717) and LAPACK performs inside dpotrf.  A and B are row blocks of the
// same packed panel (leading dimension lda), C is a region of a panel or of a
// contribution block.  One workgroup (4 waves, 2x2) owns a BM x BN tile of C;
// A/B k-slabs of BK columns are staged in LDS k-major with a 16-double pad so
// the 16x4 MFMA operand reads are bank-conflict free; the next slab is
// prefetched into registers while the current one feeds the matrix cores.
// The MFMA is issued as D' = Bfrag x Afrag so that a lane holds C(i0+(l&15),
// j0+(l>>4)+4r): consecutive lanes then touch consecutive rows of a column of
// the column-major target and the read-modify-write is coalesced.

// Block -> tile map of an update region.
//  * Tiles are enumerated in strips of 8 tile columns, row by row inside a strip,
//    so 64 consecutive tiles form an 8x8 super-tile (16 operand panels for 64
//    tiles instead of ~34 with a row-major order).
//  * XCD awareness: the dispatcher hands block b to XCD b % 8 (observed, used for
//    speed only); with G.swz the blocks of one XCD walk whole super-tiles, so the
//    panels a super-tile shares meet in ONE L2 instead of in all eight.
//  * Multi-GPU: the 64-tile chunks are dealt round-robin to the ranks
//    (chunk c belongs to rank c % tile_mul == tile_add).
// Returns false for a padding block (past the end of the rank's last chunk).
__device__ __forceinline__ bool decode_tile (const GemmGroup &G, int u, int &I, int &J)
{
    // strips of W tile columns: W = 8 (64 consecutive tiles = an 8 x 8 super-tile), or 16 with
    // G.swz == 2 (256 consecutive tiles = a 16 x 16 super-tile = the tiles ONE XCD runs at a time
    // with one wave per tile: 32 operand panels for 256 tiles)
    const int W = G.swz == 2 ? 16 : 8, LW = G.swz == 2 ? 8 : 6 ;
    int t ;
    if (G.tile_mul == 1 && !G.swz) t = u + (G.tile_add << 6) ;
    else
    {
        int cl, within ;
        if (G.swz && u < (G.nblk >> (LW + 3)) << (LW + 3)) { int q = u >> 3 ; cl = (q >> LW) * 8 + (u & 7) ; within = q & ((1 << LW) - 1) ; }
        else { cl = u >> LW ; within = u & ((1 << LW) - 1) ; }
        t = ((cl * G.tile_mul + G.tile_add) << LW) + within ;
    }
    if (t >= G.ntiles) return false ;
    int S = 0, w, mt = G.mt, nt = G.nt ;
    for ( ; ; S++)
    {
        w = nt - W * S ; if (w > W) w = W ;
        int rows = G.tri ? mt - W * S : mt ;
        int c = G.tri ? w * (w + 1) / 2 + (rows - w) * w : rows * w ;
        if (t < c) break ;
        t -= c ;
    }
    if (!G.tri) { I = t / w ; J = W * S + t % w ; return true ; }
    int tri = w * (w + 1) / 2 ;
    if (t < tri)
    {
        int r = 0 ;
        while ((r + 1) * (r + 2) / 2 <= t) r++ ;
        I = W * S + r ; J = W * S + t - r * (r + 1) / 2 ;
    }
    else
    {
        int t2 = t - tri ;
        I = W * S + w + t2 / w ; J = W * S + t2 % w ;
    }
    return true ;
}

// ---- dense update, second generation ------------------------------------------
// Same contract as k_update.  Differences, all measured on MI355X:
//  * the global loads of a full k-slab are unconditional (row indices clamped
//    once, rows/cols past the tile edge are computed but never stored), so
//    hipcc keeps 2*N loads in flight instead of branching and waiting on each;
//    only the last, partial slab takes a masked path;
//  * __launch_bounds__(256, MINW): with MINW = 2 two workgroups share a CU and
//    one's barrier/LDS-store bubble is covered by the other's MFMA stream (one
//    wave per SIMD tops out at ~35 TFLOP/s, two at ~47 on this part);
//  * DB = true double-buffers the LDS slabs (one barrier per slab).
// One BM x BN tile of an update region (tile row I, tile column J): the whole
// contraction and the read-modify-write (or assignment) of the target.
//
// TW ("twin"): the operands are the EVEN columns of a phi-embedded complex panel (rows 2i, 2i+1 =
// re, im of complex row i; the odd columns are their rotations and are not read: the scheduler
// hands over lda = 2 nsrow and k = K / 2).  With P = A_even B_even' the full-K update of the
// embedding is U (2i, 2j) = U (2i+1, 2j+1) = P (2i, 2j) + P (2i+1, 2j+1) and
// U (2i+1, 2j) = -U (2i, 2j+1) = P (2i+1, 2j) - P (2i, 2j+1) -- a complex multiply-add as four
// real ones instead of the embedding's eight.  The slab rows are dealt so that fragment a of a
// wave holds the rows of parity a (position 32 w + 16 a + t <-> row 32 w + 2 t + a, likewise
// the columns): the four products of one complex entry then sit in the same lane and slot of
// acc [0..1][0..1] and are combined there; the epilogue stores row pairs.
// TW = 2 (CX): the same contraction on a front in complex storage (ldcx / stcx above): the panel IS
// its even twin columns (lda = ld of the front, k = complex columns), and of the target only the
// even twin columns exist, column j at (j >> 1) ldc.
template <int BM, int BN, int BK, bool DB, bool TO_LDS = false, int TW = 0>
__device__ __forceinline__ void update_tile (const GemmGroup &G, int I, int J,
    double *Lx, double *CB, double *sm)
{
    static_assert (!TW || (BM == 64 && BN == 64), "twin tiles are 64 x 64") ;
    constexpr int LDT = BM + 16 ;
    constexpr int LDU = BN + 16 ;
    constexpr int WM = BM / 2, WN = BN / 2 ;
    constexpr int TI = WM / 16, TJ = WN / 16 ;
    constexpr int NA = BM * BK / 256, NB_ = BN * BK / 256 ;
    constexpr int ASZ = BK * LDT, BSZ = BK * LDU ;
    int row0 = I * BM, col0 = J * BN ;
    int mrem = G.m - row0, nrem = G.n - col0 ;
    i64 lda = G.lda ;
    int K = G.k ;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6 ;
    int wm = wave & 1, wn = wave >> 1 ;

    // element (idx % BM, idx / BM) of a slab, idx = tid + 256 q: the row is the
    // same for every q when BM divides 256, the k offset advances by 256/BM
    const double *pa, *pb ;
    {
        int i = tid % BM, j = tid % BN ;
        if constexpr (TW)
        {
            i = (i & 32) + 2 * (i & 15) + ((i >> 4) & 1) ;
            j = (j & 32) + 2 * (j & 15) + ((j >> 4) & 1) ;
        }
        if (i > mrem - 1) i = mrem - 1 ;
        if (j > nrem - 1) j = nrem - 1 ;
        pa = Lx + G.a_off + row0 + i + (i64) (tid / BM) * lda ;
        pb = Lx + G.b_off + col0 + j + (i64) (tid / BN) * lda ;
    }
    constexpr int KSA = 256 / BM, KSB = 256 / BN ;     // k stride between q's
    static_assert (256 % BM == 0 && 256 % BN == 0, "tile must divide the block") ;
    double ra [NA], rb [NB_] ;
    auto gload_full = [&] (int k0)
    {
#pragma unroll
        for (int q = 0 ; q < NA ; q++) ra [q] = pa [(i64) (k0 + q * KSA) * lda] ;
#pragma unroll
        for (int q = 0 ; q < NB_ ; q++) rb [q] = pb [(i64) (k0 + q * KSB) * lda] ;
    } ;
    auto gload_tail = [&] (int k0)
    {
#pragma unroll
        for (int q = 0 ; q < NA ; q++)
        {
            int k = k0 + tid / BM + q * KSA ;
            ra [q] = (k < K) ? pa [(i64) (k0 + q * KSA) * lda] : 0.0 ;
        }
#pragma unroll
        for (int q = 0 ; q < NB_ ; q++)
        {
            int k = k0 + tid / BN + q * KSB ;
            rb [q] = (k < K) ? pb [(i64) (k0 + q * KSB) * lda] : 0.0 ;
        }
    } ;
    auto gload = [&] (int k0) { if (k0 + BK <= K) gload_full (k0) ; else gload_tail (k0) ; } ;
    auto lstore = [&] (int buf)
    {
        double *As = sm + buf * (ASZ + BSZ), *Bs = As + ASZ ;
#pragma unroll
        for (int q = 0 ; q < NA ; q++) As [(tid / BM + q * KSA) * LDT + (tid % BM)] = ra [q] ;
#pragma unroll
        for (int q = 0 ; q < NB_ ; q++) Bs [(tid / BN + q * KSB) * LDU + (tid % BN)] = rb [q] ;
    } ;

    d4 acc [TI][TJ] ;
#pragma unroll
    for (int a = 0 ; a < TI ; a++)
#pragma unroll
        for (int b = 0 ; b < TJ ; b++) acc [a][b] = (d4) {0.0, 0.0, 0.0, 0.0} ;

    auto compute = [&] (int buf)
    {
        const double *As = sm + buf * (ASZ + BSZ), *Bs = As + ASZ ;
        const double *ap = As + (lane >> 4) * LDT + wm * WM + (lane & 15) ;
        const double *bp = Bs + (lane >> 4) * LDU + wn * WN + (lane & 15) ;
#pragma unroll
        for (int kk = 0 ; kk < BK ; kk += 4)
        {
            double af [TI], bf [TJ] ;
#pragma unroll
            for (int a = 0 ; a < TI ; a++) af [a] = ap [kk * LDT + a * 16] ;
#pragma unroll
            for (int b = 0 ; b < TJ ; b++) bf [b] = bp [kk * LDU + b * 16] ;
#pragma unroll
            for (int a = 0 ; a < TI ; a++)
#pragma unroll
                for (int b = 0 ; b < TJ ; b++)
                    acc [a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64 (
                        bf [b], af [a], acc [a][b], 0, 0, 0) ;
        }
    } ;

    double *C = (G.c_in_cb ? CB : Lx) + G.c_off + row0 + colx<TW == 2> (col0, G.ldc) ;
    // k_update2f's diagonal tile: the old values of the block are requested before the
    // contraction starts -- after it they would put one HBM latency on the chain that
    // leads into the elimination
    double cv [TO_LDS ? TI : 1][TO_LDS ? TJ : 1][4] ;
    if constexpr (TO_LDS)
    {
#pragma unroll
        for (int a = 0 ; a < TI ; a++)
#pragma unroll
            for (int b = 0 ; b < TJ ; b++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    int i = wm * WM + a * 16 + (lane & 15), j = wn * WN + b * 16 + (lane >> 4) + 4 * r ;
                    if constexpr (TW) { i = wm * WM + 2 * (lane & 15) + a ; j = wn * WN + 2 * ((lane >> 4) + 4 * r) + b ; }
                    cv [a][b][r] = ldcx<TW == 2> (C, i, j, G.ldc) ;
                }
    }
    gload (0) ;
    if constexpr (DB)
    {
        lstore (0) ;
        __syncthreads () ;
        int buf = 0 ;
        for (int k0 = 0 ; k0 < K ; k0 += BK)
        {
            bool more = k0 + BK < K ;
            if (more) gload (k0 + BK) ;
            compute (buf) ;
            if (more) lstore (buf ^ 1) ;
            __syncthreads () ;
            buf ^= 1 ;
        }
    }
    else
    {
        for (int k0 = 0 ; k0 < K ; k0 += BK)
        {
            __syncthreads () ;
            lstore (0) ;
            __syncthreads () ;
            if (k0 + BK < K) gload (k0 + BK) ;
            compute (0) ;
        }
    }
    if constexpr (TW)
    {
        static_assert (!TW || (TI == 2 && TJ == 2), "one fragment per parity") ;
#pragma unroll
        for (int r = 0 ; r < 4 ; r++)
        {
            const double ee = acc [0][0][r] + acc [TI - 1][TJ - 1][r], oe = acc [TI - 1][0][r] - acc [0][TJ - 1][r] ;
            acc [0][0][r] = ee ; acc [TI - 1][TJ - 1][r] = ee ; acc [TI - 1][0][r] = oe ; acc [0][TJ - 1][r] = -oe ;
        }
    }
    if constexpr (TO_LDS)
    {
        // k_update2f, a full BM x BN tile on the diagonal: the updated block goes to LDS,
        // k-major (sm [j * BM + i], zero above the diagonal), for the elimination that follows
        static_assert (BM == BN && !DB, "diagonal tile") ;
        __syncthreads () ;          // the operand slabs in sm are dead
#pragma unroll
        for (int a = 0 ; a < TI ; a++)
#pragma unroll
            for (int b = 0 ; b < TJ ; b++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    int i = wm * WM + a * 16 + (lane & 15) ;
                    int j = wn * WN + b * 16 + (lane >> 4) + 4 * r ;
                    if constexpr (TW) { i = wm * WM + 2 * (lane & 15) + a ; j = wn * WN + 2 * ((lane >> 4) + 4 * r) + b ; }
                    sm [j * BM + i] = (i >= j) ? cv [a][b][r] - acc [a][b][r] : 0.0 ;
                }
        return ;
    }
    if constexpr (TW)
    {
        // row pairs (2 t, 2 t + 1) of a column: one 16-byte read-modify-write (m, n and the
        // region's origin are even: a pair lies inside the region or outside it)
#pragma unroll
        for (int b = 0 ; b < (TW == 2 ? 1 : TJ) ; b++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                const int i = wm * WM + 2 * (lane & 15) ;
                const int j = wn * WN + 2 * ((lane >> 4) + 4 * r) + b ;
                if (i >= mrem || j >= nrem) continue ;
                const bool both = !G.tri || row0 + i >= col0 + j ;
                const bool second = G.tri && row0 + i + 1 == col0 + j ;
                double *Cj = C + i + colx<TW == 2> (j, G.ldc) ;
                if (both)
                {
                    d2u v ;
                    if (G.assign) { v.x = -acc [0][b][r] ; v.y = -acc [1][b][r] ; }
                    else { v = *(const d2u *) Cj ; v.x -= acc [0][b][r] ; v.y -= acc [1][b][r] ; }
                    *(d2u *) Cj = v ;
                }
                else if (second)
                {
                    if (G.assign) Cj [1] = -acc [1][b][r] ; else Cj [1] -= acc [1][b][r] ;
                }
            }
        return ;
    }
#pragma unroll
    for (int a = 0 ; a < TI ; a++)
#pragma unroll
        for (int b = 0 ; b < TJ ; b++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                int i = wm * WM + a * 16 + (lane & 15) ;
                int j = wn * WN + b * 16 + (lane >> 4) + 4 * r ;
                if (i < mrem && j < nrem && (!G.tri || row0 + i >= col0 + j))
                {
                    if (G.assign) C [i + (i64) j * G.ldc] = -acc [a][b][r] ;
                    else C [i + (i64) j * G.ldc] -= acc [a][b][r] ;
                }
            }
}

template <int BM, int BN, int BK, int MINW, bool DB, int TW = 0>
__global__ void __launch_bounds__(256, MINW) k_update2 (const GemmGroup *g, int ng,
    double *Lx, double *CB)
{
    __shared__ double sm [(DB ? 2 : 1) * BK * (BM + 16 + BN + 16)] ;
    int gi = find_group (g, ng, (int) blockIdx.x, &GemmGroup::tile_start) ;
    GemmGroup G = g [gi] ;
    int I, J ;
    if ((int) blockIdx.x - G.tile_start >= G.nblk) return ;
    if (!decode_tile (G, (int) blockIdx.x - G.tile_start, I, J)) return ;
    update_tile<BM, BN, BK, DB, false, TW> (G, I, J, Lx, CB, sm) ;
}

// ---- dense update, third generation: one wave = one tile, no LDS, no barrier --------
// Same contract as k_update2 (C -= A B' on a 64 x 64 tile of an update region, reference
// dsyrk / dgemm t_cholmod_super_numeric.c:682-717).  k_update2's four waves share the
// operand slabs through LDS and meet at two barriers per 16 columns of K; with five such
// workgroups per CU the matrix pipe still idles 17 % of the time (rocprofv3
// SQ_VALU_MFMA_BUSY_CYCLES) although neither LDS nor HBM is saturated: the waves of a
// workgroup sit on four different SIMDs, progress at the pace of whatever shares those
// SIMDs, and wait for the slowest at every barrier.  Here a wave owns the whole 64 x 64
// tile (sixteen 16 x 16 accumulators = 128 registers) and streams its operands straight
// from L1 / L2 into the MFMA operand layout:
//   * lane (lr, lk) loads rows {2 lr, 2 lr + 1} (+ 32) of column k + lk of the A panel with
//     ONE 16-byte load (rows of a packed panel are contiguous): the two doubles are the
//     lane's entries of two different 16 x 4 operand fragments.  The tile's rows are
//     thereby permuted over the fragments (fragment a, lane row t <-> tile row
//     32 (a >> 1) + 2 t + (a & 1)); the epilogue applies the same permutation, which also
//     turns the read-modify-write of C into 16-byte accesses;
//   * 4 loads feed 16 MFMAs (1024 matrix-pipe cycles): operand traffic per flop as
//     k_update2 (64 + 64 rows per 64 x 64 tile), no ds_write / ds_read / s_barrier at all;
//   * DEPTH operand sets are in flight (3: the loads of a k-step are issued three steps,
//     i.e. >= 3000 cycles, before their MFMAs), two waves per SIMD cover the rest.
// Partial tiles (EDGE) mask their stores (and, with UPD3_CLAMP_LOADS, use 8-byte loads with clamped rows); a K that is no
// multiple of 4 ends with one masked k-step in either path.
// TW: the even columns of a phi-embedded complex panel (see update_tile): the row / column pairs a
// lane loads ARE the (re, im) pairs, fragment parity = row parity, so the four real products of a
// complex entry are acc [2 q + {0,1}][2 p + {0,1}][r] of one lane and are combined in place.
// NQ = 2: the whole 64 x 64 tile; NQ = 1 (round 5): its left or right HALF, 64 rows x 32 columns starting at column cofs (0 or
// 32) of the tile -- two waves share a tile, so a region of T tiles puts 2 T waves on the chip: the launches of 1 000 - 6 000
// tiles (the top fronts of the mid-size problems) that fill the 2048 wave slots one and a bit or two and a bit times then
// waste half a tile time at most instead of a whole one.  Three loads feed eight MFMAs instead of four feeding sixteen; the
// entries are computed in the same order as in the whole tile: bit-identical results.
// Partial tiles load like whole ones (round 5): a lane's rows past the edge of the region are rows of the same panel further
// down, or -- below the front's last row -- the first rows of its next column, of the next front, or the UPD3_LX_PAD doubles
// every allocation of L ends with; what they feed are accumulators of rows / columns the epilogue never stores.  With the
// clamped 8-byte loads of rounds 3-4 (UPD3_CLAMP_LOADS = 1) a partial tile took twice the time of a whole one, and the tiles
// behind it in the launch lost the lockstep their L2 reuse depends on (schedule_dense.hip: flush_kind).
#ifndef UPD3_CLAMP_LOADS
#define UPD3_CLAMP_LOADS 0
#endif
template <int DEPTH, bool EDGE, int TW = 0, int NQ = 2>
__device__ __forceinline__ void update_tile_w (const GemmGroup &G, int I, int J, double *Lx, double *CB, int cofs = 0)
{
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4 ;
    const int row0 = I * 64, col0 = J * 64 + cofs ;
    const int mrem = G.m - row0, nrem = G.n - col0 ;
    const i64 lda = G.lda ;
    const int K = G.k ;
    // operand pointers of this lane: element (row pair, column lk) of the A and B row blocks
    const double *pa [2], *pb [2] ;
    int ra [2][2], rb [2][2] ;          // (EDGE) clamped rows of the pair
#pragma unroll
    for (int q = 0 ; q < 2 ; q++)
    {
        ra [q][0] = 32 * q + 2 * lr ; ra [q][1] = ra [q][0] + 1 ;
        rb [q][0] = 32 * q + 2 * lr ; rb [q][1] = rb [q][0] + 1 ;
        if constexpr (EDGE && UPD3_CLAMP_LOADS)
        {
#pragma unroll
            for (int h = 0 ; h < 2 ; h++)
            {
                if (ra [q][h] > mrem - 1) ra [q][h] = mrem - 1 ;
                if (rb [q][h] > nrem - 1) rb [q][h] = nrem - 1 ;
            }
        }
        pa [q] = Lx + G.a_off + row0 + (i64) lk * lda ;
        pb [q] = Lx + G.b_off + col0 + (i64) lk * lda ;
    }
    struct Frag { double a [4], b [2 * NQ] ; } ;     // a [2 q + h] = A (row pair q, member h), likewise b (NQ column pairs)
    // a full k-step (columns kk .. kk + 3 all below K)
    auto load = [&] (Frag &F, int kk)
    {
        const i64 ko = (i64) kk * lda ;
#pragma unroll
        for (int q = 0 ; q < 2 ; q++)
        {
            if constexpr (EDGE && UPD3_CLAMP_LOADS)
            {
                F.a [2 * q] = pa [q][ko + ra [q][0]] ; F.a [2 * q + 1] = pa [q][ko + ra [q][1]] ;
                if (q < NQ) { F.b [2 * q] = pb [q][ko + rb [q][0]] ; F.b [2 * q + 1] = pb [q][ko + rb [q][1]] ; }
            }
            else
            {
                d2u va = *(const d2u *) (pa [q] + ko + ra [q][0]) ;
                F.a [2 * q] = va.x ; F.a [2 * q + 1] = va.y ;
                if (q < NQ)
                {
                    d2u vb = *(const d2u *) (pb [q] + ko + rb [q][0]) ;
                    F.b [2 * q] = vb.x ; F.b [2 * q + 1] = vb.y ;
                }
            }
        }
    } ;
    // the last step of a K that is no multiple of 4: lanes whose column kk + lk lies past K read
    // column K - 1 instead and contribute zero
    auto load_tail = [&] (Frag &F, int kk)
    {
        const bool live = kk + lk < K ;
        const i64 ko = (i64) kk * lda - (live ? 0 : (i64) (kk + lk - (K - 1)) * lda) ;
#pragma unroll
        for (int q = 0 ; q < 2 ; q++)
        {
            double a0, a1, b0 = 0, b1 = 0 ;
            if constexpr (EDGE && UPD3_CLAMP_LOADS)
            {
                a0 = pa [q][ko + ra [q][0]] ; a1 = pa [q][ko + ra [q][1]] ;
                if (q < NQ) { b0 = pb [q][ko + rb [q][0]] ; b1 = pb [q][ko + rb [q][1]] ; }
            }
            else
            {
                d2u va = *(const d2u *) (pa [q] + ko + ra [q][0]) ;
                a0 = va.x ; a1 = va.y ;
                if (q < NQ) { d2u vb = *(const d2u *) (pb [q] + ko + rb [q][0]) ; b0 = vb.x ; b1 = vb.y ; }
            }
            F.a [2 * q] = live ? a0 : 0.0 ; F.a [2 * q + 1] = live ? a1 : 0.0 ;
            if (q < NQ) { F.b [2 * q] = live ? b0 : 0.0 ; F.b [2 * q + 1] = live ? b1 : 0.0 ; }
        }
    } ;
    d4 acc [4][2 * NQ] ;
#pragma unroll
    for (int a = 0 ; a < 4 ; a++)
#pragma unroll
        for (int b = 0 ; b < 2 * NQ ; b++) acc [a][b] = (d4) {0.0, 0.0, 0.0, 0.0} ;
    auto compute = [&] (const Frag &F)
    {
#pragma unroll
        for (int a = 0 ; a < 4 ; a++)
#pragma unroll
            for (int b = 0 ; b < 2 * NQ ; b++)
                acc [a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64 (F.b [b], F.a [a], acc [a][b], 0, 0, 0) ;
    } ;
    // Operand set d holds k-step s + d at the top of an iteration; the set consumed last is
    // reloaded FIRST in the next iteration, every other set right after its MFMAs: the most
    // recent load at the loop edge is one whole k-step (>= 1024 matrix-pipe cycles) old, so
    // the conservative wait hipcc places at a loop header costs nothing.
    const int nfull = K >> 2, nsteps = (K + 3) >> 2 ;      // (a partial last step: nsteps == nfull + 1)
    auto load_any = [&] (Frag &F, int step) { if (step < nfull) load (F, 4 * step) ; else load_tail (F, 4 * step) ; } ;
    Frag f [DEPTH] ;
#pragma unroll
    for (int d = 0 ; d < DEPTH - 1 ; d++) if (d < nsteps) load_any (f [d], d) ;
    // (nothing in flight at the loop header: hipcc merges the states of the two edges into
    // it and would otherwise wait for all but one load at the top of EVERY iteration; with
    // this the waits inside the loop are the exact counts)
    __builtin_amdgcn_s_waitcnt (0x0F70) ;       // vmcnt(0)
    int s = 0 ;
    for ( ; s + 2 * DEPTH - 1 <= nfull ; s += DEPTH)
    {
        load (f [DEPTH - 1], 4 * (s + DEPTH - 1)) ;
#pragma unroll
        for (int d = 0 ; d < DEPTH ; d++)
        {
            compute (f [d]) ;
            if (d < DEPTH - 1) load (f [d], 4 * (s + DEPTH + d)) ;
        }
    }
    for ( ; s < nsteps ; s += DEPTH)
    {
        if (s + DEPTH - 1 < nsteps) load_any (f [DEPTH - 1], s + DEPTH - 1) ;
#pragma unroll
        for (int d = 0 ; d < DEPTH ; d++)
        {
            if (s + d < nsteps) compute (f [d]) ;
            if (d < DEPTH - 1 && s + DEPTH + d < nsteps) load_any (f [d], s + DEPTH + d) ;
        }
    }
    if constexpr (TW)
    {
#pragma unroll
        for (int q = 0 ; q < 2 ; q++)
#pragma unroll
            for (int p = 0 ; p < NQ ; p++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    const double ee = acc [2 * q][2 * p][r] + acc [2 * q + 1][2 * p + 1][r] ;
                    const double oe = acc [2 * q + 1][2 * p][r] - acc [2 * q][2 * p + 1][r] ;
                    acc [2 * q][2 * p][r] = ee ; acc [2 * q + 1][2 * p + 1][r] = ee ;
                    acc [2 * q + 1][2 * p][r] = oe ; acc [2 * q][2 * p + 1][r] = -oe ;
                }
    }
    // epilogue: acc [a][b][r] of lane (lr, lk) is C (row 32 (a >> 1) + 2 lr + (a & 1),
    // column 32 (b >> 1) + 2 (lk + 4 r) + (b & 1))
    double *C = (G.c_in_cb ? CB : Lx) + G.c_off + row0 + colx<TW == 2> (col0, G.ldc) ;
    const bool diag = G.tri && I == J ;
#pragma unroll
    for (int b = 0 ; b < 2 * NQ ; b++)
#pragma unroll
        for (int r = 0 ; r < 4 ; r++)
        {
            if (TW == 2 && (b & 1)) continue ;          // (complex storage: the even twin columns only)
            const int j = 32 * (b >> 1) + 2 * (lk + 4 * r) + (b & 1) ;      // column inside this wave's part of the tile
            const int jd = j + cofs ;                                       // ... inside the tile (the diagonal test)
            double *Cj = C + colx<TW == 2> (j, G.ldc) ;
#pragma unroll
            for (int q = 0 ; q < 2 ; q++)
            {
                const int i = 32 * q + 2 * lr ;
                // the row pair (i, i + 1) of column j, both rows inside the region
                auto store_pair = [&] ()
                {
                    if (diag && i + 1 < jd) return ;
                    d2u v ;
                    if (G.assign) { v.x = -acc [2 * q][b][r] ; v.y = -acc [2 * q + 1][b][r] ; }
                    else
                    {
                        v = *(const d2u *) (Cj + i) ;
                        v.x -= acc [2 * q][b][r] ; v.y -= acc [2 * q + 1][b][r] ;
                    }
                    if (diag && i < jd) Cj [i + 1] = v.y ;          // (the pair straddles the diagonal)
                    else *(d2u *) (Cj + i) = v ;
                } ;
                if constexpr (EDGE)
                {
                    // (a partial tile stores like a whole one wherever the pair and the column exist: its masked scalar stores made
                    // it slower than its neighbours at short K, and tiles of unequal length cost the launch its lockstep)
                    if (j < nrem)
                    {
                        if (i + 1 < mrem) store_pair () ;
                        else if (i < mrem && (!diag || i >= jd))
                        {
                            if (G.assign) Cj [i] = -acc [2 * q][b][r] ;
                            else Cj [i] -= acc [2 * q][b][r] ;
                        }
                    }
                }
                else store_pair () ;
            }
        }
}

// WPB = 4 (several ranks): four tiles per workgroup, one per wave, nothing shared.  Wave w of block b
// plays block ((b >> 3) * 4 + w) * 8 + (b & 7) of the one-wave launch (same XCD, b & 7, hence the same
// tile walk).  A retiring workgroup then frees a wave slot on EVERY SIMD of its CU at once: the
// four-wave workgroups of the exchange stream (and of RCCL) find room beside the update, which
// one-wave workgroups -- refilled SIMD by SIMD -- never leave them.
// HALF (round 5; one tile per workgroup only): blocks b and b + 8 -- the same XCD, hence the same tile walk -- take the left and
// the right 32 columns of tile ((b >> 4) << 3) | (b & 7) of the one-wave launch; the grid is twice that launch's, rounded up
// to whole groups of eight.
template <int DEPTH, int TW = 0, int WPB = 1, bool HALF = false>
__global__ void __launch_bounds__(64 * WPB, 2) k_update3 (const GemmGroup *g, int ng, double *Lx, double *CB)
{
    int vb = (int) blockIdx.x ;
    int cofs = 0 ;
    if constexpr (WPB > 1) vb = ((vb >> 3) * WPB + (int) (threadIdx.x >> 6)) * 8 + (vb & 7) ;
    if constexpr (HALF) { cofs = ((vb >> 3) & 1) * 32 ; vb = ((vb >> 4) << 3) | (vb & 7) ; }
    int gi = find_group (g, ng, vb, &GemmGroup::tile_start) ;
    GemmGroup G = g [gi] ;
    int I, J ;
    if (vb - G.tile_start >= G.nblk) return ;
    if (!decode_tile (G, vb - G.tile_start, I, J)) return ;
    if constexpr (HALF)
    {
        if (G.n - J * 64 - cofs <= 0) return ;                          // (the last tile column is at most 32 wide)
        if (G.m - I * 64 >= 64 && G.n - J * 64 - cofs >= 32) update_tile_w<DEPTH, false, TW, 1> (G, I, J, Lx, CB, cofs) ;
        else update_tile_w<DEPTH, true, TW, 1> (G, I, J, Lx, CB, cofs) ;
    }
    else
    {
        if (G.m - I * 64 >= 64 && G.n - J * 64 >= 64) update_tile_w<DEPTH, false, TW> (G, I, J, Lx, CB) ;
        else update_tile_w<DEPTH, true, TW> (G, I, J, Lx, CB) ;
    }
}

// ---- trailing update that also factors the next diagonal block ------------------
// The narrow (K < 512) updates of the panel chain are followed, on the same stream, by
// the dpotrf of the block they have just finished updating: tile (0,0) of their region.
// Here the workgroup that owns that tile keeps it in LDS and eliminates it on the spot
// (pf_eliminate, as k_potrf_mfma) while the other workgroups of the launch finish their
// tiles -- one launch of ~17 us less per 64 columns of the chain (potrf -> trsm -> update,
// ~43 us per step, is half the time of a mid-size factorization).  Everything else in
// the launch is k_update2<64,64,16,2,false>.
template <int TW>
__global__ void __launch_bounds__(256, 2) k_update2f (const GemmGroup *g, int ng,
    double *Lx, double *CB, i32 *info)
{
    __shared__ __attribute__((aligned(16))) double sm [PF_NB * PF2_LD] ;   // operand slabs, then the diagonal block
    __shared__ int s_fail ;
    static_assert (PF_NB == 64 && PF2_LD == 64 && 16 * (64 + 16 + 64 + 16) <= PF_NB * PF2_LD, "slabs fit") ;
    int gi = find_group (g, ng, (int) blockIdx.x, &GemmGroup::tile_start) ;
    GemmGroup G = g [gi] ;
    int I, J ;
    if ((int) blockIdx.x - G.tile_start >= G.nblk) return ;
    if (!decode_tile (G, (int) blockIdx.x - G.tile_start, I, J)) return ;
    if (!(G.pf_next && I == 0 && J == 0))
    {
        update_tile<64, 64, 16, false, false, TW> (G, I, J, Lx, CB, sm) ;
        return ;
    }
    __builtin_amdgcn_s_setprio (3) ;
    update_tile<64, 64, 16, false, true, TW> (G, 0, 0, Lx, CB, sm) ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6 ;
    if (tid == 0) s_fail = -1 ;
    __syncthreads () ;
    double *A = Lx + G.c_off ;
    const i64 lda = G.ldc ;
    if (info [G.front] != 0)
    {
        // an earlier pivot of this front failed: its remaining columns are zero
        for (int k = wave ; k < PF_NB ; k += 4)
            if (lane >= k) stcx<TW == 2> (A, lane, k, lda, 0.0) ;
        return ;
    }
    auto tick = [] (int) {} ;
    pf_eliminate<TW == 2> (sm, PF_NB / 16, &s_fail, tid, tick) ;
    const int fail = s_fail ;
    if (fail >= 0 && tid == 0) info [G.front] = G.pf_col0 + fail + 1 ;
    pf_store<TW == 2> (A, lda, sm, PF_NB, fail, lane, wave) ;
}

// ---- panel solve + the K = 64 update of the next block column + its dpotrf ---------
// Every other step of the panel chain (recursive doubling, p = 1) is: dtrsm of the rows
// below a 64 x 64 diagonal block, then the K = 64 update of the NEXT 64 columns with
// exactly those solved rows, then the dpotrf of the next diagonal block -- two launches
// (k_trsm_mfma, k_update2f) of ~15 + ~25 us whose work is a few microseconds.  Here one
// launch does all three: a workgroup solves its 64 rows X_b (as k_trsm_mfma: the solved
// blocks stay in registers, in the A-operand layout of v_mfma_f64_16x16x4) and, beside
// them, the 64 rows X_0 of the next diagonal block (redundantly: no cross-workgroup
// hand-off, which costs an L2 write-back on this part); -X_0 goes to LDS k-major, and
// C_b -= X_b X_0' (64 MFMAs per wave, four independent chains) is read-modified-written
// from the accumulator layout.  Workgroup 0 owns the next diagonal block: it keeps the
// updated block in LDS and eliminates it on the spot (pf_eliminate, as k_update2f).
// Conditions (checked by the scheduler): a full 64-column panel, a full next block,
// the front not shared between ranks.  TrGroup: l_off = the factored diagonal block,
// b_off = first row below it (same columns), m rows, col0 = its first column.
#define TU_LDX 80
__host__ __device__ inline size_t trsm_upd_lds_bytes ()
{
    return (size_t) (64 * 64 + 4 * 256 + 64 * TU_LDX) * sizeof (double) ;
}
template <bool CX>
__global__ void __launch_bounds__(256) k_trsm_upd (const TrGroup *g, int ng, double *Lx, i32 *info, i32 *cnt)
{
    extern __shared__ __attribute__((aligned(16))) double tu_lds [] ;
    double *Ls = tu_lds ;                           // [64][64]  -L11 k-major; later the next diagonal block
    double *Wd = Ls + 64 * 64 ;                     // [4][16][16] inverses of the 16 x 16 diagonal blocks
    double *X0s = Wd + 4 * 256 ;                    // [64][TU_LDX]  -X_0 k-major
    __shared__ int s_fail, s_last ;
    __builtin_amdgcn_s_setprio (3) ;
    const int gi = find_group (g, ng, (int) blockIdx.x, &TrGroup::blk_start) ;
    const TrGroup G = g [gi] ;
    const i64 lda = G.lda ;
    const int ldl = 64 ;
    const double *L11 = Lx + G.l_off ;
    const int inf = info [G.front] ;
    int nvalid = 64 ;
    if (inf != 0)
    {
        nvalid = inf - 1 - G.col0 ;
        if (nvalid < 0) nvalid = 0 ;
        if (nvalid > 64) nvalid = 64 ;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6 ;
    const int lr = lane & 15, lk = lane >> 4 ;
    const int blk = (int) blockIdx.x - G.blk_start ;
    const int row = blk * TRM_ROWS + wave * 16 + lr ;
    const bool rok = row < G.m ;
    double *B = Lx + G.b_off ;                                  // the rows below the panel
    const int brow = rok ? row : G.m - 1 ;                      // this lane's row of them
    const int b0row = wave * 16 + lr ;                          // its row of the next diagonal block's rows
    double *Cr = B + colx<CX> (64, lda) ;                       // the same rows, 64 columns to the right
    d4 bj [4], b0 [4], cj [4] ;
    {
        const int j = tid & 63 ;
        double tmp [16] ;
#pragma unroll
        for (int q = 0 ; q < 16 ; q++) tmp [q] = ldcx<CX> (L11, j, (tid >> 6) + 4 * q, lda) ;
#pragma unroll
        for (int jj = 0 ; jj < 4 ; jj++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                const int c = 16 * jj + lk + 4 * r ;
                bj [jj][r] = ldcx<CX> (B, brow, c, lda) ;
                b0 [jj][r] = ldcx<CX> (B, b0row, c, lda) ;
                cj [jj][r] = ldcx<CX> (Cr, brow, c, lda) ;
            }
#pragma unroll
        for (int q = 0 ; q < 16 ; q++)
        {
            const int k = (tid >> 6) + 4 * q ;
            double v = (j == k) ? 1.0 : 0.0 ;
            if (j < nvalid && k < j) v = -tmp [q] ;
            if (j < nvalid && k == j) v = tmp [q] ;
            Ls [k * ldl + j] = v ;
        }
    }
    if (tid == 0) s_fail = -1 ;
    __syncthreads () ;
    auto tick = [] (int) {} ;
    trsm_diag_inverses (Ls, ldl, Wd, 4, lane, wave, tick) ;
    // The solve is in place, and the rows of the next diagonal block (the first 64 rows below
    // the panel, B0) are read by EVERY workgroup of the group (b0) before they are solved: a
    // write-after-read hazard on those 64 rows, no data passes through memory.  Every
    // workgroup computes the solved X_0 anyway, so the store of X_0 is left to whichever
    // workgroup is the LAST to have its loads in: each counts itself in (one relaxed atomic)
    // once its loads have arrived, and the one that draws nwg - 1 knows that nobody will read
    // the old rows any more.  Nobody waits, nothing depends on the order in which the
    // dispatcher starts workgroups (round 2 had workgroup 0 spin on the counter, bounded, and
    // store regardless on time-out).
    const int nwg = (G.m + TRM_ROWS - 1) / TRM_ROWS ;
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory") ;
    __syncthreads () ;
    if (tid == 0) s_last = (nwg == 1 || __hip_atomic_fetch_add (cnt + gi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) ? 1 : 0 ;
    d4 xr [4], x0 [4] ;
    trsm_solve_rows<CX> (bj, 4, Ls, ldl, Wd, lane, nvalid, rok && blk != 0, 64, B, brow, lda, tick, xr) ;
    if (blk == 0)
    {
#pragma unroll
        for (int j = 0 ; j < 4 ; j++) x0 [j] = xr [j] ;
    }
    else trsm_solve_rows<CX> (b0, 4, Ls, ldl, Wd, lane, nvalid, false, 64, B, b0row, lda, tick, x0) ;
    // -X_0 k-major: X0s [k][j] = -X_0 (j, k); block jj, element r of the accumulator layout is
    // (row lr of the wave's 16 rows, column 16 jj + lk + 4 r)
#pragma unroll
    for (int jj = 0 ; jj < 4 ; jj++)
#pragma unroll
        for (int r = 0 ; r < 4 ; r++)
            X0s [(16 * jj + lk + 4 * r) * TU_LDX + wave * 16 + lr] = -x0 [jj][r] ;
    __syncthreads () ;                      // (also: every wave is done with Ls and Wd; s_last is visible)
    if (s_last)
    {
        // the solved rows of the next diagonal block, by the last workgroup to have read the old ones
#pragma unroll
        for (int jj = 0 ; jj < 4 ; jj++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++) stcx<CX> (B, b0row, 16 * jj + lk + 4 * r, lda, x0 [jj][r]) ;
    }
    // C (row, 16 jt + lk + 4 r) -= sum_k X (row, k) X_0 (16 jt + .., k)
#pragma unroll
    for (int i = 0 ; i < 4 ; i++)
#pragma unroll
        for (int s4 = 0 ; s4 < 4 ; s4++)
#pragma unroll
            for (int jt = 0 ; jt < 4 ; jt++)
            {
                double bv = X0s [(16 * i + 4 * s4 + lk) * TU_LDX + 16 * jt + lr] ;
                cj [jt] = __builtin_amdgcn_mfma_f64_16x16x4f64 (bv, xr [i][s4], cj [jt], 0, 0, 0) ;
            }
    if (blk != 0)
    {
        if (rok)
        {
#pragma unroll
            for (int jt = 0 ; jt < 4 ; jt++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++) stcx<CX> (Cr, brow, 16 * jt + lk + 4 * r, lda, cj [jt][r]) ;
        }
        return ;
    }
    // workgroup 0: the updated tile is the next diagonal block
    double *A = Cr ;
    if (inf != 0)
    {
        // an earlier pivot of this front failed: its remaining columns are zero
        for (int k = wave ; k < PF_NB ; k += 4)
            if (lane >= k) stcx<CX> (A, lane, k, lda, 0.0) ;
        return ;
    }
    double *T = Ls ;                                // T [k * PF2_LD + i] = A (i, k), zero above the diagonal
#pragma unroll
    for (int jt = 0 ; jt < 4 ; jt++)
#pragma unroll
        for (int r = 0 ; r < 4 ; r++)
        {
            const int i = wave * 16 + lr, j = 16 * jt + lk + 4 * r ;
            T [j * PF2_LD + i] = (i >= j) ? cj [jt][r] : 0.0 ;
        }
    __syncthreads () ;
    pf_eliminate<CX> (T, PF_NB / 16, &s_fail, tid, tick) ;
    const int fail = s_fail ;
    if (fail >= 0 && tid == 0) info [G.front] = G.col0 + 64 + fail + 1 ;
    pf_store<CX> (A, lda, T, PF_NB, fail, lane, wave) ;
}

// ---- multi-GPU: packing of a shared front's block column for the row-split exchange ----
// A block column [b0, b0+w) of a front shared by g ranks holds per-rank partial sums in
// its live rows (>= b0).  It is summed with ONE reduce-scatter whose segment q carries
//     [ D : the w x w diagonal block (ld = w) | near chunk q : R x w (ld = R) | far chunk q : Rf x w (ld = Rf) ]
// NEAR rows = the rows below D inside the outer block column [b0 + w, o1): later block columns of the
// outer block need them as an operand; FAR rows = [o1, nsrow), cut into the SAME g chunks for every block
// column of the outer block (chunks zero-padded past their last row).  Every rank receives the summed
// diagonal block and the summed rows of ITS two chunks, runs the panel chain (dpotrf / dtrsm / K < 512
// updates, reference t_cholmod_super_numeric.c:864-867, :997-1002) on those rows only, and the solved
// chunks travel back with two all-gathers: a small one of the near chunks, in line, and a large one of the
// far chunks that nobody needs before the outer update (round 5: the updates between the block columns of
// an outer block are chunk-local on the far rows), so it runs on the exchange stream beside the chain of the
// following block columns.  Same volume as the all-reduce it replaces.
// mode 0: Lx -> stage (all g segments: this rank's partial sums, D repeated per segment)
// mode 1: segment r of stage -> Lx (summed D and own chunks)
// mode 2: own near chunk of Lx -> ag + r R w        mode 3: ag (all near chunks but r) -> Lx
// mode 4: own far chunk of Lx -> ag + r Rf w        mode 5: ag (all far chunks but r) -> Lx   (ag: the buffer of that gather)
// (round 4: one workgroup = one column of the block column x one part -- the diagonal block or one row chunk; rows
// stream contiguously, no division per element: the element-indexed first version moved ~1 TB/s, and a rank of 8 moves
// 67 GB through these copies per factorization of Poisson 200^3)
// grid: w * (2 g + 1) workgroups for mode 0 (part 2 g = the diagonal block), w * 3 for mode 1 (D, own chunks), w for
// modes 2 / 4, w * g for modes 3 / 5.
// test hook (CHOLMOD_HIP_TEST_JITTER): keeps its stream busy for about `ticks` ticks of the 100 MHz wall clock
__global__ void k_spin (long long ticks)
{
    const long long t0 = (long long) wall_clock64 () ;
    while ((long long) wall_clock64 () - t0 < ticks) __builtin_amdgcn_s_sleep (20) ;
}

// progress marker (cholmod_hip_progress_enable): one word into host-visible memory, in stream order -- a watchdog
// thread of the host reads which exchange of the factorization a hung rank has entered and not left
__global__ void k_mark (volatile long long *p, long long v)
{
    *p = v ;
    __threadfence_system () ;
}

// n doubles from src to dst, those at index >= nr as zeros (pad) or not at all: four independent loads per thread in flight
// (the pack of a block column and the unpack of the gathered chunks are 48 GB each per rank of 8 and factorization)
__device__ __forceinline__ void xm_copy (double *dst, const double *src, int n, int nr, int tid, int nt, bool pad)
{
    int i = tid ;
    for ( ; i + 3 * nt < n ; i += 4 * nt)
    {
        double v [4] ;
#pragma unroll
        for (int u = 0 ; u < 4 ; u++) { const int e = i + u * nt ; v [u] = (e < nr) ? __builtin_nontemporal_load (src + e) : 0.0 ; }
#pragma unroll
        for (int u = 0 ; u < 4 ; u++) { const int e = i + u * nt ; if (pad || e < nr) dst [e] = v [u] ; }
    }
    for ( ; i < n ; i += nt) { if (i < nr) dst [i] = __builtin_nontemporal_load (src + i) ; else if (pad) dst [i] = 0.0 ; }
}

__global__ void __launch_bounds__(256) k_xchg_move (XchgD X, int mode, double *Lx, double *stage, double *ag)
{
    const i64 seg = (i64) X.w * X.w + ((i64) X.R + X.Rf) * X.w ;
    const int j = (int) blockIdx.x % X.w, part = (int) blockIdx.x / X.w ;
    double *S = Lx + X.slab + (i64) j * X.lda ;                   // column j of the block column, from the diagonal block's first row
    const int tid = threadIdx.x, nt = (int) blockDim.x ;
    // chunk q of the near (far = false) or far rows: where it starts in the column, its rows that exist, its place in a segment
    auto rows_of = [&] (bool far, int q, int &first, int &nr, int &R, i64 &sofs)
    {
        R = far ? X.Rf : X.R ;
        first = (far ? X.fo : X.w) + q * R ;
        nr = (far ? X.mf : X.mb) - q * R ;
        sofs = (i64) X.w * X.w + (far ? (i64) X.R * X.w : 0) + (i64) j * R ;
    } ;
    int first, nr, R ; i64 sofs ;
    if (mode == 0)
    {
        if (part == 2 * X.g)
        {
            // the diagonal block's column j (lower part, zero above) into every segment
            for (int i = tid ; i < X.w ; i += nt)
            {
                const double v = (i >= j) ? S [i] : 0.0 ;
                for (int q = 0 ; q < X.g ; q++) stage [(i64) q * seg + (i64) j * X.w + i] = v ;
            }
        }
        else
        {
            const int q = part % X.g ;
            rows_of (part >= X.g, q, first, nr, R, sofs) ;
            xm_copy (stage + (i64) q * seg + sofs, S + first, R, nr, tid, nt, true) ;
        }
    }
    else if (mode == 1)
    {
        const double *ps = stage + (i64) X.r * seg ;
        if (part == 0) { for (int i = j + tid ; i < X.w ; i += nt) S [i] = ps [(i64) j * X.w + i] ; }
        else
        {
            rows_of (part == 2, X.r, first, nr, R, sofs) ;
            const double *src = ps + sofs ;
            double *dst = S + first ;
            for (int i = tid ; i < R && i < nr ; i += nt) dst [i] = src [i] ;
        }
    }
    else if (mode == 2 || mode == 4)
    {
        rows_of (mode == 4, X.r, first, nr, R, sofs) ;
        const double *src = S + first ;
        double *dst = ag + (i64) X.r * R * X.w + (i64) j * R ;
        for (int i = tid ; i < R ; i += nt) dst [i] = (i < nr) ? src [i] : 0.0 ;
    }
    else
    {
        const int q = part ;
        if (q == X.r || q >= X.g) return ;
        rows_of (mode == 5, q, first, nr, R, sofs) ;
        const double *src = ag + (i64) q * R * X.w + (i64) j * R ;
        xm_copy (S + first, src, R < nr ? R : nr, nr, tid, nt, false) ;
    }
}

// ---- the window of a distributed front (several GPUs, round 4) --------------------------------
// A shared front's panel is stored by column slabs on their owners (FrontD::own_w); the outer block
// column the group is factoring lives, on every member, in a WINDOW: nsrow x OB doubles addressed as
// if the whole front were there (win = base - o0 nsrow, so column c of the front is win + c ld).
//   mode 0 (open):  window columns [c0, c1), rows r0 .. nsrow-1  :=  the owner's stored column, zero on
//                   the other members -- their partial sum before the children's contributions
//                   (extend-add into the window) and the dealt tiles of the wide updates are added;
//   mode 1 (close): the factored columns back into the owner's slabs.
// One workgroup = one column x WIN_ROWS rows.
__global__ void __launch_bounds__(256) k_win_move (const WinD *g, int ng, double *Lx)
{
    int gi = find_group (g, ng, (int) blockIdx.x, &WinD::blk_start) ;
    const WinD W = g [gi] ;
    const int nchunk = (W.nrows - W.r0 + WIN_ROWS - 1) / WIN_ROWS ;
    const int b = (int) blockIdx.x - W.blk_start ;
    const int c = W.c0 + b / nchunk ;
    const int ra = W.r0 + (b % nchunk) * WIN_ROWS ;
    const int rb = ra + WIN_ROWS < W.nrows ? ra + WIN_ROWS : W.nrows ;
    if (c >= W.c1) return ;
    const int t = c / W.own_w ;
    const bool mine = (t % W.own_g) == W.own_r ;
    double *wc = Lx + W.win + (i64) c * W.ld ;
    double *sc = Lx + W.store + (i64) ((t / W.own_g) * W.own_w + c % W.own_w) * W.ld ;
    if (W.mode == 0)
    {
        if (mine) for (int i = ra + threadIdx.x ; i < rb ; i += (int) blockDim.x) wc [i] = sc [i] ;
        else for (int i = ra + threadIdx.x ; i < rb ; i += (int) blockDim.x) wc [i] = 0.0 ;
    }
    else if (mine) for (int i = ra + threadIdx.x ; i < rb ; i += (int) blockDim.x) sc [i] = wc [i] ;
}

// ---- the panel chain in 256-column sub-blocks (opt-in: CHOLMOD_HIP_CHAIN256) ---------------
// The default chain walks a front's outer block column in 64-column steps: dpotrf of the
// diagonal block, dtrsm of ALL rows below, the K = 64 ... 256 doubling updates -- about fifteen
// dependent launches of 15-30 us per 512 columns.  This variant advances 256 columns per pair of
// launches (reference steps: dpotrf t_cholmod_super_numeric.c:864-867, dtrsm :997-1002, applied
// to a 256-column sub-block):
//   k_diag      ONE workgroup per front factors the w x w (w <= 256) diagonal sub-block, left-
//               looking over 64-column panels: the panel's 64 x 64 diagonal block is updated with
//               the sub-block's earlier columns on the matrix cores (operands straight from L2
//               in MFMA layout), eliminated in LDS (pf_eliminate), its rows inside the sub-block
//               solved as k_trsm_mfma does; the sixteen 16 x 16 diagonal-block inverses it needs
//               for that are published (dinv) for
//   k_rowsolve  every 64 rows below the sub-block, one workgroup: X = B inv(L)' for all w columns
//               at once -- per 64-column panel the rows of L it multiplies with are staged
//               k-major in LDS (negated), the solved 16 x 16 blocks of X stay in registers in
//               the A-operand layout (16 x d4 per wave) and feed the later panels' products; the
//               K < 256 "narrow updates" of the 64-column chain do not exist here, they are the
//               left-looking products inside these two kernels.
// Measured (round 3, MI355X): 2.3x fewer launches (nd24k stand-in 617 -> 273), same parity, but
// k_diag takes 126 us per full sub-block (four 64-column eliminations at ~11 us, whose column
// chain of ~250 cycles per column is what any variant waits for, plus the in-block solves) and
// k_rowsolve 39 us (544 MFMAs per wave, serial on its SIMD): 190 us per 256 columns against
// ~220 us for the 64-column chain's sixteen launches -- and slower where many fronts share a
// launch.  Poisson 100^3 156 against 150 ms, nd24k stand-in 32.9 against 29.4, 2D 1259^2 7.7
// against 5.3.  Kept as an option: a one-workgroup diagonal kernel is what a look-ahead on a
// CU-masked stream would need.
// Not-positive-definite protocol as everywhere: the first pivot <= 0 goes to info [front]
// (1-based, relative to the front), every later column of the front is written as zero.
template <typename Tick>
__device__ __forceinline__ void dg_left_looking (d4 (&acc) [4], const double *A, i64 lda, int rowc, int c0, int w, int lr, int lk, Tick)
{
    // acc [jb] -= L (rowc, 0 : c0) L (c0 + 16 jb + lr', 0 : c0)'   (rows of the wave in lanes lr, columns lk + 4 r)
    const double *pa = A + rowc + (i64) lk * lda ;
    const double *pb [4] ;
#pragma unroll
    for (int jb = 0 ; jb < 4 ; jb++)
    {
        int rb = c0 + 16 * jb + lr ; if (rb > w - 1) rb = w - 1 ;
        pb [jb] = A + rb + (i64) lk * lda ;
    }
    // (c0 is a multiple of 64: batches of 32 columns, all forty loads of a batch in flight before its MFMAs.
    // Fetching the next batch under this batch's MFMAs was tried: 160 more registers, spills, 130 -> 219 us.)
    for (int k0 = 0 ; k0 < c0 ; k0 += 32)
    {
        double af [8], bf [8][4] ;
#pragma unroll
        for (int u = 0 ; u < 8 ; u++)
        {
            af [u] = pa [(i64) (k0 + 4 * u) * lda] ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++) bf [u][jb] = pb [jb][(i64) (k0 + 4 * u) * lda] ;
        }
#pragma unroll
        for (int u = 0 ; u < 8 ; u++)
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
                acc [jb] = __builtin_amdgcn_mfma_f64_16x16x4f64 (-bf [u][jb], af [u], acc [jb], 0, 0, 0) ;
    }
}

template <bool TIMED>
__global__ void __launch_bounds__(256) k_diag (const DgGroup *g, double *Lx, i32 *info, double *dinv, long long *tim)
{
    long long tc [8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0 ;
    auto stamp = [&] (int slot) { if constexpr (TIMED) { long long t = __builtin_readcyclecounter () ; tc [slot] += t - t_prev ; t_prev = t ; } } ;
    if constexpr (TIMED) t_prev = __builtin_readcyclecounter () ;
    __shared__ __attribute__((aligned(16))) double T [PF_NB * PF2_LD] ;     // the panel's 64 x 64 diagonal block, k-major
    __shared__ __attribute__((aligned(16))) double Ls [64 * 64] ;           // -L11 k-major (diagonal positive), identity-padded
    __shared__ __attribute__((aligned(16))) double Wd [4 * 256] ;           // inverses of its four 16 x 16 diagonal blocks
    __shared__ int s_fail ;
    __builtin_amdgcn_s_setprio (3) ;
    const DgGroup G = g [blockIdx.x] ;
    double *A = Lx + G.off ;
    const i64 lda = G.lda ;
    const int w = G.w ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4 ;
    double *DI = dinv + (i64) G.slot * 4096 ;
    auto tick = [] (int) {} ;
    bool dead = info [G.front] != 0 ;        // an earlier pivot of this front failed
    for (int c0 = 0 ; c0 < w ; c0 += 64)
    {
        const int pw = (w - c0 < 64) ? w - c0 : 64 ;
        if (dead)
        {
            // the front's remaining columns are zero (:889-895, :926-931); identity "inverses" so
            // that nothing non-finite reaches k_rowsolve's (masked) products
            for (int k = wave ; k < pw ; k += 4)
                for (int i = c0 + k + lane ; i < w ; i += 64) A [i + (i64) (c0 + k) * lda] = 0.0 ;
            for (int e = tid ; e < 1024 ; e += 256) DI [(c0 >> 6) * 1024 + e] = ((e & 255) >> 4) == (e & 15) ? 1.0 : 0.0 ;
            continue ;
        }
        // ---- the panel's diagonal block: this wave's 16 rows, left-looking update, into T
        {
            const int rowg = c0 + 16 * wave + lr ;
            const int rowc = rowg < w ? rowg : w - 1 ;
            d4 acc [4] ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    int col = c0 + 16 * jb + lk + 4 * r ; if (col > w - 1) col = w - 1 ;
                    acc [jb][r] = A [rowc + (i64) col * lda] ;
                }
            dg_left_looking (acc, A, lda, rowc, c0, w, lr, lk, tick) ;
            stamp (0) ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    const int i = 16 * wave + lr, j = 16 * jb + lk + 4 * r ;
                    T [j * PF2_LD + i] = (i < pw && j < pw) ? (i >= j ? acc [jb][r] : 0.0) : (i == j ? 1.0 : 0.0) ;
                }
        }
        if (tid == 0) s_fail = -1 ;
        __syncthreads () ;
        stamp (1) ;
        pf_eliminate (T, (pw + 15) >> 4, &s_fail, tid, tick) ;
        stamp (2) ;
        const int fail = s_fail ;
        const int nvalid = fail >= 0 ? fail : pw ;
        if (fail >= 0 && tid == 0) info [G.front] = G.col0 + c0 + fail + 1 ;
        // the factored block to Lx (columns at / beyond a failed pivot: zero), and -L11 k-major for the row solves
#pragma unroll
        for (int q = 0 ; q < 16 ; q++)
        {
            const int k = wave + 4 * q, i = lane ;
            if (k < pw && i >= k && i < pw) A [(c0 + i) + (i64) (c0 + k) * lda] = (k < nvalid) ? T [k * PF2_LD + i] : 0.0 ;
            double v = (i == k) ? 1.0 : 0.0 ;
            if (i < nvalid && k < i) v = -T [k * PF2_LD + i] ;
            if (i < nvalid && k == i) v = T [k * PF2_LD + k] ;
            Ls [k * 64 + i] = v ;
        }
        __syncthreads () ;
        stamp (3) ;
        trsm_diag_inverses (Ls, 64, Wd, 4, lane, wave, tick) ;
        __syncthreads () ;
        for (int e = tid ; e < 1024 ; e += 256) DI [(c0 >> 6) * 1024 + e] = Wd [e] ;
        stamp (4) ;
        // ---- the rows of the sub-block below this panel (only behind a full panel)
        for (int rr = c0 + 64 ; rr < w ; rr += 64)
        {
            const int row = rr + 16 * wave + lr ;
            const bool rok = row < w ;
            const int rowc = rok ? row : w - 1 ;
            d4 bj [4], xr [4] ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++) bj [jb][r] = A [rowc + (i64) (c0 + 16 * jb + lk + 4 * r) * lda] ;
            dg_left_looking (bj, A, lda, rowc, c0, w, lr, lk, tick) ;
            stamp (5) ;
            trsm_solve_rows<false> (bj, 4, Ls, 64, Wd, lane, nvalid, rok, 64, A + (i64) c0 * lda, rowc, lda, tick, xr) ;
            stamp (6) ;
        }
        if (fail >= 0) dead = true ;
        __syncthreads () ;          // this panel's columns are in Lx for the next panel's products (workgroup scope)
        stamp (7) ;
    }
    if constexpr (TIMED) { if (tid == 0) for (int q = 0 ; q < 8 ; q++) tim [q] = tc [q] ; }
}

__host__ __device__ inline size_t rowsolve_lds_bytes () { return (size_t) (DG_W * 64 + 4 * 256) * sizeof (double) ; }
__global__ void __launch_bounds__(256) k_rowsolve (const RsGroup *g, int ng, double *Lx, const i32 *info, const double *dinv)
{
    extern __shared__ __attribute__((aligned(16))) double rs_lds [] ;
    double *Lst = rs_lds ;                  // [k][c]: -L (c0 + c, k) for k < c0 + c (k-major, 64 columns wide), identity-padded
    double *Wd = Lst + DG_W * 64 ;          // the panel's four 16 x 16 diagonal-block inverses (from k_diag)
    const int gi = find_group (g, ng, (int) blockIdx.x, &RsGroup::blk_start) ;
    const RsGroup G = g [gi] ;
    const i64 lda = G.lda ;
    const int w = G.w ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4 ;
    const int row = ((int) blockIdx.x - G.blk_start) * RS_ROWS + wave * 16 + lr ;
    const bool rok = row < G.m ;
    double *B = Lx + G.b_off + (rok ? row : G.m - 1) ;       // this lane's row, first column of the sub-block
    const double *L = Lx + G.l_off ;                          // L (0, 0) of the sub-block
    const double *DI = dinv + (i64) G.slot * 4096 ;
    const int inf = info [G.front] ;
    int nvt = w ;                                             // valid columns of the sub-block
    if (inf != 0) { nvt = inf - 1 - G.col0 ; if (nvt < 0) nvt = 0 ; if (nvt > w) nvt = w ; }
    d4 X [16] ;
#pragma unroll
    for (int q = 0 ; q < 16 ; q++) X [q] = (d4) {0.0, 0.0, 0.0, 0.0} ;
    // one 64-column panel; j is a compile-time constant so that X [] stays in registers
    auto panel = [&] (auto jc)
    {
        constexpr int j = decltype (jc)::value ;
        constexpr int c0 = 64 * j ;
        if (c0 < w)
        {
            const int pw = (w - c0 < 64) ? w - c0 : 64 ;
            int nvj = nvt - c0 ; if (nvj < 0) nvj = 0 ; if (nvj > pw) nvj = pw ;
            d4 bj [4] ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    int c = 16 * jb + lk + 4 * r ; if (c > pw - 1) c = pw - 1 ;
                    bj [jb][r] = B [(i64) (c0 + c) * lda] ;
                }
            __syncthreads () ;              // the previous panel's readers are done with Lst / Wd
            {
                // stage the rows c0 .. c0 + 63 of L, columns 0 .. c0 + 63, sixteen loads in flight per thread
                const int c = lane ;
                const int rowL = (c0 + c < w) ? c0 + c : w - 1 ;
                const int nk = c0 + 64 ;
                for (int kb = 0 ; kb < nk ; kb += 64)
                {
                    double tmp [16] ;
#pragma unroll
                    for (int q = 0 ; q < 16 ; q++)
                    {
                        int k = kb + wave + 4 * q ; if (k > w - 1) k = w - 1 ;
                        tmp [q] = L [rowL + (i64) k * lda] ;
                    }
#pragma unroll
                    for (int q = 0 ; q < 16 ; q++)
                    {
                        const int k = kb + wave + 4 * q ;
                        double v = (k == c0 + c) ? 1.0 : 0.0 ;
                        if (c < nvj && k < c0 + c) v = -tmp [q] ;
                        if (c < nvj && k == c0 + c) v = tmp [q] ;
                        Lst [k * 64 + c] = v ;
                    }
                }
                for (int e = tid ; e < 1024 ; e += 256) Wd [e] = DI [j * 1024 + e] ;
            }
            __syncthreads () ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
            {
                // (four partial sums: a dependent fp64 MFMA costs ~190 cycles, an independent one 64)
                d4 acc = bj [jb], a1 = (d4) {0.0, 0.0, 0.0, 0.0}, a2 = a1, a3 = a1 ;
#pragma unroll
                for (int kb = 0 ; kb < 16 ; kb++)
                {
                    if (kb < 4 * j + jb)
                    {
                        double b [4] ;
#pragma unroll
                        for (int s4 = 0 ; s4 < 4 ; s4++) b [s4] = Lst [(16 * kb + 4 * s4 + lk) * 64 + 16 * jb + lr] ;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [0], X [kb][0], acc, 0, 0, 0) ;
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [1], X [kb][1], a1, 0, 0, 0) ;
                        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [2], X [kb][2], a2, 0, 0, 0) ;
                        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [3], X [kb][3], a3, 0, 0, 0) ;
                    }
                }
                acc = (acc + a1) + (a2 + a3) ;
                d4 x = (d4) {0.0, 0.0, 0.0, 0.0} ;
#pragma unroll
                for (int s4 = 0 ; s4 < 4 ; s4++)
                {
                    double b = Wd [jb * 256 + (4 * s4 + lk) * 16 + lr] ;
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64 (b, acc [s4], x, 0, 0, 0) ;
                }
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    const int c = 16 * jb + lk + 4 * r ;
                    const double v = (c < nvj) ? x [r] : 0.0 ;
                    x [r] = v ;
                    if (rok && c < pw) B [(i64) (c0 + c) * lda] = v ;
                }
                X [4 * j + jb] = x ;
            }
        }
    } ;
    panel (std::integral_constant<int, 0> {}) ;
    panel (std::integral_constant<int, 1> {}) ;
    panel (std::integral_constant<int, 2> {}) ;
    panel (std::integral_constant<int, 3> {}) ;
}

// ---- the 256-column chain in ONE launch (round 4): k_chainf -----------------------------------
// k_diag + k_rowsolve fused, with the diagonal sub-block's work spread over four workgroups instead
// of one.  Every workgroup owns 64 rows of a front's sub-block column [col0, col0 + w), w <= 256,
// for the whole launch:
//   * row block rb < nd = ceil (w / 64) lies INSIDE the w x w diagonal sub-block ("diagonal
//     workgroup"): it solves its rows against the panels j < rb exactly as a row workgroup does, then
//     subtracts its own X X' from its 64 x 64 diagonal block, eliminates it (pf_eliminate), inverts
//     the four 16 x 16 diagonal blocks and PUBLISHES row block rb of L (its solved blocks, L_rb,rb,
//     the inverses) -- the critical path per 64 columns is solve + syrk + elimination of ONE
//     workgroup (k_diag: one workgroup did all four row blocks' solves and products as well);
//   * the row blocks below the sub-block ("row workgroups") are k_rowsolve: per panel j they stage
//     row block j of L k-major in LDS and multiply out of registers -- but wait, per panel, for
//     diagonal workgroup j's flag instead of a kernel boundary.
// Hand-off without an L2 write-back: everything a diagonal workgroup publishes is written with
// relaxed agent-scope atomic stores (global_store ... sc1: write-through past the XCD's L2) and read
// with relaxed agent-scope atomic loads (sc1: served at the coherence point); the flag follows the
// data after an explicit s_waitcnt vmcnt(0) in every wave + the workgroup barrier (cf_drain_stores;
// checked in the gfx950 ISA: waitcnt, s_barrier, flag store) -- no buffer_wbl2 / buffer_inv, which
// is what made the cross-workgroup hand-off of round 2 cost 30 us.  A workgroup only ever waits for
// workgroups with a LOWER block index (all diagonal workgroups of a launch come first, in panel
// order per front), so in-order dispatch cannot deadlock; a wait that exceeds its budget raises
// *err (the host reports CHOLMOD_HIP_GPU_PROBLEM) instead of hanging the device.
// flags [4 slot + j]: low byte = stages of row block j published (s <= j: its solved blocks of the panels
// < s; j + 1: its diagonal block and the inverses as well), then bits 8.. = 1 + nvt, nvt = valid columns
// of the sub-block so far (w when no pivot has failed).
// rows below the sub-block: [w, w + m1) and, for a front shared between ranks (its block column dealt by row chunks), two
// more ranges [off2, off2 + m2) and [off3, off3 + m3) -- the rest of the 512-wide diagonal block on every rank, then this
// rank's chunk of the near rows (inside the outer block column) and its chunk of the far rows
__device__ __forceinline__ double ld_coh (const double *p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ; }
__device__ __forceinline__ void st_coh (double *p, double v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ; }
// polls (s_sleep 4 + one coherent load, ~1.5 us each) before a hand-off counts as failed: ~6 s, far beyond any jitter
// hold or time slice another process of the device can cause (the host then reports CHOLMOD_HIP_GPU_PROBLEM on all ranks)
#define CF_SPIN_LIMIT (1 << 22)
// all of this wave's global stores (and loads) acknowledged; also a compiler barrier for memory operations
__device__ __forceinline__ void cf_drain_stores () { asm volatile ("s_waitcnt vmcnt(0)" ::: "memory") ; }
__host__ __device__ inline size_t chainf_lds_bytes () { return (size_t) (DG_W * 64 + 4 * 256 + 16) * sizeof (double) ; }
__global__ void __launch_bounds__(256) k_chainf (const CfGroup *g, int ng, int ndiag_total, double *Lx, i32 *info,
    double *dinv, int *flags, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double cf_lds [] ;
    double *Lst = cf_lds ;                  // [k][c]: -L (c0 + c, k), k-major, 64 columns wide (row workgroup phase)
    double *Wd = Lst + DG_W * 64 ;          // four 16 x 16 diagonal-block inverses of the current panel
    int *s_int = (int *) (Wd + 4 * 256) ;   // [0] broadcast of a polled flag, [1] s_fail
    double *T = Lst ;                       // (diagonal phase, after the panels: the 64 x 64 block, k-major)
    double *Lsd = Lst + PF_NB * PF2_LD ;    // (diagonal phase: -L11 k-major for the inverses)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4 ;
    const bool isdiag = (int) blockIdx.x < ndiag_total ;
    int gi ;
    if (isdiag) gi = find_group (g, ng, (int) blockIdx.x, &CfGroup::dstart) ;
    else gi = find_group (g, ng, (int) blockIdx.x - ndiag_total, &CfGroup::bstart) ;
    const CfGroup G = g [gi] ;
    const i64 lda = G.lda ;
    const int w = G.w ;
    const int nd = (w + 63) >> 6 ;
    const int rb = isdiag ? (int) blockIdx.x - G.dstart : nd ;
    int row0 = 64 * rb, rowend = w ;
    if (!isdiag)
    {
        const int t = (int) blockIdx.x - ndiag_total - G.bstart, n1 = (G.m1 + 63) >> 6, n2 = (G.m2 + 63) >> 6 ;
        if (t < n1) { row0 = w + 64 * t ; rowend = w + G.m1 ; }
        else if (t < n1 + n2) { row0 = G.off2 + 64 * (t - n1) ; rowend = G.off2 + G.m2 ; }
        else { row0 = G.off3 + 64 * (t - n1 - n2) ; rowend = G.off3 + G.m3 ; }
    }
    const int row = row0 + 16 * wave + lr ;
    const bool rok = row < rowend ;
    double *Lb = Lx + G.l_off ;                               // L (0, 0) of the sub-block
    double *B = Lb + (rok ? row : rowend - 1) ;               // this lane's row, first column of the sub-block
    double *DI = dinv + (i64) G.slot * 4096 ;
    int *fl = flags + 4 * (i64) G.fslot ;
    if (isdiag) __builtin_amdgcn_s_setprio (3) ; else __builtin_amdgcn_s_setprio (1) ;
    int nvt = w ;                                             // valid columns of the sub-block
    bool dead = false ;                                       // a hand-off timed out (wait_stage): no store into L from here on
    if (tid == 0) s_int [2] = 0 ;                             // (read after wait_stage's barrier only)
    {
        const int inf = info [G.front] ;                      // a pivot of an EARLIER launch failed
        if (inf != 0) { nvt = inf - 1 - G.col0 ; if (nvt < 0) nvt = 0 ; if (nvt > w) nvt = w ; }
    }
    // Row block j of L is published in stages: flag j = s (1 <= s <= j) once its solved blocks of the panels < s are
    // written, and (j + 1) | (1 + nvt) << 8 once its diagonal block and the inverses are.  Thread 0 polls until the stage
    // exceeds `have`; the value travels through LDS.
    auto wait_stage = [&] (int j, int have) -> int
    {
        if (tid == 0)
        {
            int v = 0, n = 0 ;
            while (((v = __hip_atomic_load (fl + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xFF) <= have)
            {
                __builtin_amdgcn_s_sleep (4) ;
                // budget exceeded (or somebody else's was: no second full wait behind a dead workgroup): tell the host, go
                // on as "nothing valid" and DEAD -- a dead workgroup stores nothing into L any more, it only keeps
                // publishing its flags so that the workgroups behind it come to an end as well
                if (++n > CF_SPIN_LIMIT || ((n & 1023) == 0 && __hip_atomic_load (err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                {
                    atomicExch (err, 1) ; v = (j + 1) | (1 << 8) ; s_int [2] = 1 ; break ;
                }
            }
            s_int [0] = v ;
        }
        __syncthreads () ;
        if (s_int [2]) dead = true ;
        return s_int [0] ;
    } ;
    d4 X [16] ;
#pragma unroll
    for (int q = 0 ; q < 16 ; q++) X [q] = (d4) {0.0, 0.0, 0.0, 0.0} ;
    // a diagonal workgroup: its 64 x 64 diagonal block in registers from the start (accumulator layout), updated panel by
    // panel; Xp = the LDS rows k >= 192 of Lst, which a diagonal workgroup (panels j <= 2: k < 192) never stages into
    double *Xp = Lst + 192 * 64 ;
    d4 dacc [4] ;
#pragma unroll
    for (int jb = 0 ; jb < 4 ; jb++) dacc [jb] = (d4) {0.0, 0.0, 0.0, 0.0} ;
    if (isdiag)
    {
        const int dc0 = 64 * rb ;
        const int rowg = dc0 + 16 * wave + lr ;
        const int rowc = rowg < w ? rowg : w - 1 ;
#pragma unroll
        for (int jb = 0 ; jb < 4 ; jb++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                int col = dc0 + 16 * jb + lk + 4 * r ; if (col > w - 1) col = w - 1 ;
                dacc [jb][r] = Lb [rowc + (i64) col * lda] ;
            }
    }
    // one 64-column panel left of this workgroup's rows (k_rowsolve's panel; j is a compile-time constant so that X [] stays in registers)
    auto panel = [&] (auto jc)
    {
        constexpr int j = decltype (jc)::value ;
        constexpr int c0 = 64 * j ;
        if (j < rb && c0 < w)
        {
            const int pw = (w - c0 < 64) ? w - c0 : 64 ;
            d4 bj [4] ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    int c = 16 * jb + lk + 4 * r ; if (c > pw - 1) c = pw - 1 ;
                    bj [jb][r] = B [(i64) (c0 + c) * lda] ;
                }
            // stage the rows c0 .. c0 + 63 of L, columns 0 .. c0 + 63, following diagonal workgroup j's stages: the
            // columns of the earlier panels while it is still busy with the later ones, so that only its diagonal
            // block (16 loads) and the inverses are left to fetch when its last flag arrives (coherent loads)
            int have = 0, nvj = 0 ;
            const int c = lane ;
            const int rowL = (c0 + c < w) ? c0 + c : w - 1 ;
            // one batch = 64 columns of row block j of L (coherent loads) into Lst
            auto stage_batch = [&] (int p)
            {
                const int kb = 64 * p ;
                double tmp [16] ;
#pragma unroll
                for (int q = 0 ; q < 16 ; q++)
                {
                    int k = kb + wave + 4 * q ; if (k > w - 1) k = w - 1 ;
                    tmp [q] = ld_coh (Lb + rowL + (i64) k * lda) ;
                }
#pragma unroll
                for (int q = 0 ; q < 16 ; q++)
                {
                    const int k = kb + wave + 4 * q ;
                    double v = (k == c0 + c) ? 1.0 : 0.0 ;
                    // (columns of the earlier panels: row c0 + c counts as valid -- what it feeds are output columns that
                    // are masked if it is not, see the note at the products)
                    if ((p < j || c < nvj) && k < c0 + c) v = -tmp [q] ;
                    if (p == j && c < nvj && k == c0 + c) v = tmp [q] ;
                    Lst [k * 64 + c] = v ;
                }
            } ;
            // (1) the columns of the EARLIER panels, as diagonal workgroup j publishes them (it is still busy with the later ones)
            while (have < j)
            {
                const int fv = wait_stage (j, have) ;      // (ends with a barrier: the previous panel's readers are done with Lst / Wd)
                int upto = fv & 0xFF ; if (upto > j) upto = j ;
                for (int p = have ; p < upto ; p++) stage_batch (p) ;
                have = upto ;
            }
            // (2) ... and the products with them, BEFORE the last flag: pre [jb] = B_jb - sum_{kb < 4 j} X_kb L (16 jb .., 16 kb ..)'
            // (a row of L past a failed pivot enters only output columns at / past that pivot, which are written as zero)
            d4 pre [4] ;
            if (j > 0) __syncthreads () ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
            {
                d4 acc = bj [jb], a1 = (d4) {0.0, 0.0, 0.0, 0.0}, a2 = a1, a3 = a1 ;
#pragma unroll
                for (int kb = 0 ; kb < 12 ; kb++)
                {
                    if (kb < 4 * j)
                    {
                        double b [4] ;
#pragma unroll
                        for (int s4 = 0 ; s4 < 4 ; s4++) b [s4] = Lst [(16 * kb + 4 * s4 + lk) * 64 + 16 * jb + lr] ;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [0], X [kb][0], acc, 0, 0, 0) ;
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [1], X [kb][1], a1, 0, 0, 0) ;
                        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [2], X [kb][2], a2, 0, 0, 0) ;
                        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [3], X [kb][3], a3, 0, 0, 0) ;
                    }
                }
                pre [jb] = (acc + a1) + (a2 + a3) ;
            }
            // (3) the last stage: its diagonal block (16 loads) and the inverses
            {
                const int fv = wait_stage (j, j) ;
                const int pv = (fv >> 8) - 1 ;
                if (pv < nvt) nvt = pv ;
                nvj = nvt - c0 ; if (nvj < 0) nvj = 0 ; if (nvj > pw) nvj = pw ;
                stage_batch (j) ;
                for (int e = tid ; e < 1024 ; e += 256) Wd [e] = ld_coh (DI + j * 1024 + e) ;
            }
            __syncthreads () ;
#pragma unroll
            for (int jb = 0 ; jb < 4 ; jb++)
            {
                d4 acc = pre [jb], a1 = (d4) {0.0, 0.0, 0.0, 0.0}, a2 = a1, a3 = a1 ;
#pragma unroll
                for (int kb2 = 0 ; kb2 < 3 ; kb2++)
                {
                    if (kb2 < jb)
                    {
                        const int kb = 4 * j + kb2 ;
                        double b [4] ;
#pragma unroll
                        for (int s4 = 0 ; s4 < 4 ; s4++) b [s4] = Lst [(16 * kb + 4 * s4 + lk) * 64 + 16 * jb + lr] ;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [0], X [kb][0], acc, 0, 0, 0) ;
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [1], X [kb][1], a1, 0, 0, 0) ;
                        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [2], X [kb][2], a2, 0, 0, 0) ;
                        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [3], X [kb][3], a3, 0, 0, 0) ;
                    }
                }
                acc = (acc + a1) + (a2 + a3) ;
                d4 x = (d4) {0.0, 0.0, 0.0, 0.0} ;
#pragma unroll
                for (int s4 = 0 ; s4 < 4 ; s4++)
                {
                    double b = Wd [jb * 256 + (4 * s4 + lk) * 16 + lr] ;
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64 (b, acc [s4], x, 0, 0, 0) ;
                }
#pragma unroll
                for (int r = 0 ; r < 4 ; r++)
                {
                    const int c = 16 * jb + lk + 4 * r ;
                    const double v = (c < nvj) ? x [r] : 0.0 ;
                    x [r] = v ;
                    if (rok && c < pw && !dead)
                    {
                        // (a diagonal workgroup's solved blocks are row block rb of L for everybody after it)
                        if (isdiag) st_coh (B + (i64) (c0 + c) * lda, v) ; else B [(i64) (c0 + c) * lda] = v ;
                    }
                }
                X [4 * j + jb] = x ;
            }
            if (isdiag)
            {
                // this row block's solved blocks through panel j are written (stage j + 1: the workgroups after it prefetch them)
                // ... and its own diagonal block takes their product right away, dacc -= X_j X_j' (K = 64: B operands
                // through LDS, A operands = the registers), so that only the LAST panel's is left when the last flag comes
#pragma unroll
                for (int kb2 = 0 ; kb2 < 4 ; kb2++)
#pragma unroll
                    for (int s4 = 0 ; s4 < 4 ; s4++) Xp [(16 * kb2 + lk + 4 * s4) * 64 + 16 * wave + lr] = X [4 * j + kb2][s4] ;
                __syncthreads () ;
#pragma unroll
                for (int kb2 = 0 ; kb2 < 4 ; kb2++)
#pragma unroll
                    for (int s4 = 0 ; s4 < 4 ; s4++)
#pragma unroll
                        for (int jb = 0 ; jb < 4 ; jb++)
                        {
                            const double bfv = Xp [(16 * kb2 + 4 * s4 + lk) * 64 + 16 * jb + lr] ;
                            dacc [jb] = __builtin_amdgcn_mfma_f64_16x16x4f64 (-bfv, X [4 * j + kb2][s4], dacc [jb], 0, 0, 0) ;
                        }
                // (the stage flag after the product: every wave drains ITS stores -- s_waitcnt vmcnt(0), stores count in
                // vmcnt on gfx9 -- before the barrier the flag store follows; a workgroup-scope release fence emits lgkmcnt only
                // and s_barrier waits for nothing.  By now the stores are acknowledged, so the wait is free.)
                cf_drain_stores () ;
                __syncthreads () ;
                if (tid == 0) __hip_atomic_store (fl + rb, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
            }
        }
    } ;
    panel (std::integral_constant<int, 0> {}) ;
    panel (std::integral_constant<int, 1> {}) ;
    panel (std::integral_constant<int, 2> {}) ;
    panel (std::integral_constant<int, 3> {}) ;
    if (!isdiag) return ;
    // ---- diagonal workgroup rb: its own 64 x 64 diagonal block
    const int c0 = 64 * rb ;
    const int pw = (w - c0 < 64) ? w - c0 : 64 ;
    __syncthreads () ;                      // (the last panel's readers are done with Lst / Xp: T and Lsd overlay Lst)
    auto tick = [] (int) {} ;
    int nv_out = nvt ;
    if (dead) { }
    else if (nvt <= c0)
    {
        // an earlier pivot failed: this panel's columns are zero (:889-895, :926-931), identity "inverses"
        for (int k = wave ; k < pw ; k += 4)
            for (int i = c0 + k + lane ; i < c0 + pw ; i += 64) st_coh (Lb + i + (i64) (c0 + k) * lda, 0.0) ;
        for (int e = tid ; e < 1024 ; e += 256) st_coh (DI + rb * 1024 + e, ((e & 255) >> 4) == (e & 15) ? 1.0 : 0.0) ;
    }
    else
    {
#pragma unroll
        for (int jb = 0 ; jb < 4 ; jb++)
#pragma unroll
            for (int r = 0 ; r < 4 ; r++)
            {
                const int i = 16 * wave + lr, j = 16 * jb + lk + 4 * r ;
                T [j * PF2_LD + i] = (i < pw && j < pw) ? (i >= j ? dacc [jb][r] : 0.0) : (i == j ? 1.0 : 0.0) ;
            }
        if (tid == 0) s_int [1] = -1 ;
        __syncthreads () ;
        pf_eliminate (T, (pw + 15) >> 4, &s_int [1], tid, tick) ;
        const int fail = s_int [1] ;
        const int nvalid = fail >= 0 ? fail : pw ;
        if (fail >= 0) { nv_out = c0 + fail ; if (tid == 0) info [G.front] = G.col0 + c0 + fail + 1 ; }
#pragma unroll
        for (int q = 0 ; q < 16 ; q++)
        {
            const int k = wave + 4 * q, i = lane ;
            if (k < pw && i >= k && i < pw) st_coh (Lb + (c0 + i) + (i64) (c0 + k) * lda, (k < nvalid) ? T [k * PF2_LD + i] : 0.0) ;
            double v = (i == k) ? 1.0 : 0.0 ;
            if (i < nvalid && k < i) v = -T [k * PF2_LD + i] ;
            if (i < nvalid && k == i) v = T [k * PF2_LD + k] ;
            Lsd [k * 64 + i] = v ;
        }
        __syncthreads () ;
        trsm_diag_inverses (Lsd, 64, Wd, 4, lane, wave, tick) ;
        __syncthreads () ;
        for (int e = tid ; e < 1024 ; e += 256) st_coh (DI + rb * 1024 + e, Wd [e]) ;
    }
    // publish: every wave waits for the acknowledgement of its own write-through stores (vmcnt(0)), the barrier collects the
    // four waves, then the flag goes out -- data before flag at the coherence point, still without buffer_wbl2 / buffer_inv
    cf_drain_stores () ;
    __syncthreads () ;
    if (tid == 0) __hip_atomic_store (fl + rb, (rb + 1) | ((1 + nv_out) << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
}

// ---- extreme diagonal entries of L (cholmod_l_rcond on a device-resident factor) ---------------------
// out [0] = min L_jj, out [1] = max L_jj (as ordered bit patterns of NON-NEGATIVE doubles: the caller seeds them with
// +inf / 0), out [2] += number of NaN or negative diagonal entries.  One thread per column; CX: a complex factor in its own
// storage, the diagonal of the complex L is the (real) entry (2j, 2j) of the twin, kept in the even column 2j.
template <bool CX>
__global__ void __launch_bounds__(256) k_diag_minmax (i64 n, const i32 *supermap, const FrontD *fr, const double *Lx, unsigned long long *out)
{
    const i64 k = blockIdx.x * (i64) 256 + threadIdx.x ;
    double v = 0 ;
    bool have = false, bad = false ;
    if (k < n && !(CX && (k & 1)))
    {
        const FrontD f = fr [supermap [k]] ;
        const int jj = (int) (k - f.k1) ;
        v = Lx [f.psx + jj + colx<CX> (jj, f.nsrow)] ;
        have = true ;
        if (!(v >= 0.0)) { bad = true ; have = false ; }        // (NaN or negative: counted, kept out of the extremes)
    }
    double lo = have ? v : __builtin_inf (), hi = have ? v : 0.0 ;
    for (int o = 32 ; o > 0 ; o >>= 1)
    {
        const double l2 = __shfl_xor (lo, o), h2 = __shfl_xor (hi, o) ;
        lo = l2 < lo ? l2 : lo ; hi = h2 > hi ? h2 : hi ;
    }
    const unsigned long long nbad = __popcll (__ballot (bad)) ;
    if ((threadIdx.x & 63) == 0)
    {
        atomicMin (out, (unsigned long long) __double_as_longlong (lo)) ;
        atomicMax (out + 1, (unsigned long long) __double_as_longlong (hi)) ;
        if (nbad) atomicAdd (out + 2, nbad) ;
    }
}

// ---- first failing supernode (not-positive-definite protocol) ---------------------
// out [0] = smallest supernode with info != 0 (nsuper if none), so that the host reads
// 4 bytes per factorization instead of the whole info array (G3_circuit stand-in:
// 114 250 supernodes, 457 KB into pageable memory per refactorization).
__global__ void __launch_bounds__(256) k_first_fail (i64 nsuper, const i32 *info, int *out)
{
    i64 s = blockIdx.x * (i64) 256 + threadIdx.x ;
    if (s < nsuper && info [s] != 0) atomicMin (out, (int) s) ;
}

// ---- triangular solves with the device-resident factor (nrhs columns) -------
// Level-scheduled restatement of cholmod_l_super_lsolve / _ltsolve
// (t_cholmod_super_solve.c:14-220, :222-411).  One workgroup per supernode of
// the level.  Forward: x1 = L1 \ x1 ; X[Ls2] -= L2 * x1 (children of one parent
// may hit the same rows, hence the atomic add).  Backward: x1 = L1' \ (x1 -
// L2' * X[Ls2]) needs no atomics.

template <bool CX>
__global__ void __launch_bounds__(256) k_lsolve (const SolveTask *tasks,
    const FrontD *fr, const i64 *Ls, const double *Lx, double *X, i64 ldx, int nrhs)
{
    __shared__ double xb [64] ;
    __shared__ double Dl [64 * 65] ;
    SolveTask T = tasks [blockIdx.x] ;
    const FrontD &f = fr [T.front] ;
    int nscol = f.nscol, nsrow = f.nsrow, k1 = f.k1, tid = threadIdx.x ;
    int lane = tid & 63, wave = tid >> 6 ;
    int c0 = T.c0, c1 = T.c1 ;
    const double *L = Lx + f.psx ;
    const i64 *rows = Ls + f.psi ;
    for (int r = 0 ; r < nrhs ; r++)
    {
        double *x = X + (i64) r * ldx ;
        for (int jb = c0 ; jb < c1 ; jb += 64)
        {
            int nb = c1 - jb < 64 ? c1 - jb : 64 ;
            // stage the diagonal block in LDS (coalesced), so the sequential
            // substitution below never waits on HBM
            for (int e = tid ; e < 64 * 64 ; e += 256)
            {
                int i = e & 63, j = e >> 6 ;
                Dl [i * 65 + j] = (i < nb && j < nb && j <= i) ? ldcx<CX> (L, jb + i, jb + j, nsrow) : (i == j ? 1.0 : 0.0) ;
            }
            __syncthreads () ;
            if (wave == 0)
            {
                // dtrsv("L","N","N") on the 64-wide diagonal block, one wave
                // (reciprocal diagonal once per lane and v_readlane broadcasts: a
                // division or a ds_bpermute on the 64-step chain costs 5x the rest)
                double xv = (lane < nb) ? x [k1 + jb + lane] : 0.0 ;
                double rdi = 1.0 / Dl [lane * 65 + lane] ;
                for (int j = 0 ; j < nb ; j++)
                {
                    double xj = readlane_f64 (xv, j) * readlane_f64 (rdi, j) ;
                    if (lane == j) xv = xj ;
                    else if (lane > j) xv = __builtin_fma (-Dl [lane * 65 + j], xj, xv) ;
                }
                xb [lane] = xv ;
                if (lane < nb) x [k1 + jb + lane] = xv ;
            }
            __syncthreads () ;
            for (int i = jb + nb + tid ; i < c1 ; i += 256)
            {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0 ;
                int j = 0 ;
                for ( ; j + 4 <= nb ; j += 4)
                {
                    double l0 = ldcx<CX> (L, i, jb + j, nsrow), l1 = ldcx<CX> (L, i, jb + j + 1, nsrow) ;
                    double l2 = ldcx<CX> (L, i, jb + j + 2, nsrow), l3 = ldcx<CX> (L, i, jb + j + 3, nsrow) ;
                    a0 += l0 * xb [j] ; a1 += l1 * xb [j + 1] ; a2 += l2 * xb [j + 2] ; a3 += l3 * xb [j + 3] ;
                }
                for ( ; j < nb ; j++) a0 += ldcx<CX> (L, i, jb + j, nsrow) * xb [j] ;
                x [k1 + i] -= (a0 + a1) + (a2 + a3) ;
            }
            __syncthreads () ;
        }
        if (T.below)
        {
            // dgemv: X[rows2] -= L2 * x1 ; siblings share ancestor rows -> atomic
            for (int i = nscol + tid ; i < nsrow ; i += 256)
            {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0 ;
                int j = 0 ;
                for ( ; j + 4 <= nscol ; j += 4)
                {
                    double l0 = ldcx<CX> (L, i, j, nsrow), l1 = ldcx<CX> (L, i, j + 1, nsrow) ;
                    double l2 = ldcx<CX> (L, i, j + 2, nsrow), l3 = ldcx<CX> (L, i, j + 3, nsrow) ;
                    a0 += l0 * x [k1 + j] ; a1 += l1 * x [k1 + j + 1] ;
                    a2 += l2 * x [k1 + j + 2] ; a3 += l3 * x [k1 + j + 3] ;
                }
                for ( ; j < nscol ; j++) a0 += ldcx<CX> (L, i, j, nsrow) * x [k1 + j] ;
                atomicAdd (&x [rows [i]], -((a0 + a1) + (a2 + a3))) ;
            }
        }
        __syncthreads () ;
    }
}

// column block of the big-supernode walk (k_solve_fwd_blk / k_solve_bwd_blk below)

template <bool CX>
__global__ void __launch_bounds__(256) k_ltsolve (const SolveTask *tasks,
    const FrontD *fr, const i64 *Ls, const double *Lx, double *X, i64 ldx, int nrhs)
{
    __shared__ double xb [64] ;
    __shared__ double ych [512] ;
    __shared__ double Dl [64 * 65] ;
    SolveTask T = tasks [blockIdx.x] ;
    const FrontD &f = fr [T.front] ;
    int nscol = f.nscol, nsrow = f.nsrow, k1 = f.k1, tid = threadIdx.x ;
    int lane = tid & 63, wave = tid >> 6 ;
    int c0 = T.c0, c1 = T.c1 ;
    const double *L = Lx + f.psx ;
    const i64 *rows = Ls + f.psi ;
    for (int r = 0 ; r < nrhs ; r++)
    {
        double *x = X + (i64) r * ldx ;
        if (T.below)
        {
            // dgemv("C"): x1 -= L2' * X[rows2].  The gathered X[rows2] is staged in
            // LDS in chunks of 512 rows (one gather per row instead of one per row
            // and column); a wave walks the columns, 8 independent loads per lane
            for (int i0 = nscol ; i0 < nsrow ; i0 += 512)
            {
                int nr = nsrow - i0 < 512 ? nsrow - i0 : 512 ;
                for (int q = tid ; q < 512 ; q += 256) ych [q] = (q < nr) ? x [rows [i0 + q]] : 0.0 ;
                __syncthreads () ;
                for (int j = wave ; j < nscol ; j += 4)
                {
                    double v [8] ;
#pragma unroll
                    for (int u = 0 ; u < 8 ; u++) { int q = lane + 64 * u ; v [u] = (q < nr) ? ldcx<CX> (L, i0 + q, j, nsrow) : 0.0 ; }
                    double acc = 0.0 ;
#pragma unroll
                    for (int u = 0 ; u < 8 ; u++) acc += v [u] * ych [lane + 64 * u] ;
                    for (int o = 32 ; o > 0 ; o >>= 1) acc += __shfl_down (acc, o) ;
                    if (lane == 0) x [k1 + j] -= acc ;
                }
                __syncthreads () ;
            }
        }
        // dtrsv("L","C","N") by 64-wide blocks from the bottom of [c0,c1)
        int last = c0 + ((c1 - c0 - 1) / 64) * 64 ;
        for (int jb = last ; jb >= c0 ; jb -= 64)
        {
            int nb = c1 - jb < 64 ? c1 - jb : 64 ;
            for (int e = tid ; e < 64 * 64 ; e += 256)
            {
                int i = e & 63, j = e >> 6 ;
                Dl [i * 65 + j] = (i < nb && j < nb && j <= i) ? ldcx<CX> (L, jb + i, jb + j, nsrow) : (i == j ? 1.0 : 0.0) ;
            }
            for (int jj = wave ; jj < nb ; jj += 4)
            {
                int j = jb + jj ;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0 ;
                int i = jb + nb + lane ;
                for ( ; i + 192 < c1 ; i += 256)
                {
                    double l0 = ldcx<CX> (L, i, j, nsrow), l1 = ldcx<CX> (L, i + 64, j, nsrow), l2 = ldcx<CX> (L, i + 128, j, nsrow), l3 = ldcx<CX> (L, i + 192, j, nsrow) ;
                    a0 += l0 * x [k1 + i] ; a1 += l1 * x [k1 + i + 64] ;
                    a2 += l2 * x [k1 + i + 128] ; a3 += l3 * x [k1 + i + 192] ;
                }
                for ( ; i < c1 ; i += 64) a0 += ldcx<CX> (L, i, j, nsrow) * x [k1 + i] ;
                double acc = (a0 + a1) + (a2 + a3) ;
                for (int o = 32 ; o > 0 ; o >>= 1) acc += __shfl_down (acc, o) ;
                if (lane == 0) xb [jj] = x [k1 + j] - acc ;
            }
            __syncthreads () ;
            if (wave == 0)
            {
                double xv = (lane < nb) ? xb [lane] : 0.0 ;
                double rdi = 1.0 / Dl [lane * 65 + lane] ;
                for (int j = nb - 1 ; j >= 0 ; j--)
                {
                    double xj = readlane_f64 (xv, j) * readlane_f64 (rdi, j) ;
                    if (lane == j) xv = xj ;
                    else if (lane < j) xv = __builtin_fma (-Dl [j * 65 + lane], xj, xv) ;
                }
                if (lane < nb) x [k1 + jb + lane] = xv ;
            }
            __syncthreads () ;
        }
    }
}


// ---- big supernodes: a batched walk in 256-column blocks -----------------------
// The 64x64 diagonal blocks of the big supernodes are inverted once per
// factorization (k_diag_inv64, every block independent).  The solve walks a big
// supernode in blocks of SOLVE_SB = 256 columns, and one launch carries the same
// step of EVERY big supernode of the etree level (they are independent), so the
// number of dependent launches is the block count of the level's widest
// supernode, not the sum over its supernodes (Poisson 100^3: 1680 -> 130 per
// direction):
//   forward  k_solve_fwd_diag (one workgroup per task) forms x_b = inv(L_bb) x_b, four
//            64-column sub-blocks with the explicit inverses on the diagonal, into a
//            side vector; k_solve_fwd_apply (256-row x 64-column workgroups) subtracts
//            L[rows, b] x_b from the rows below; k_solve_commit copies x_b back per level;
//   backward k_solve_bwd_apply adds L[rows, b]' x[rows] into the task's accumulator,
//            k_solve_bwd_diag forms x_b = inv(L_bb)' (x_b - acc).
// Inverse layout (per 64-block, 2 x 4096 doubles): Wm [k*64 + r] = W(r,k) and
// WmT [k*64 + c] = W(k,c), both zero outside the lower triangle and
// identity-padded past the supernode's last column.

template <bool CX>
__global__ void __launch_bounds__(64) k_diag_inv64 (const InvTask *tasks, const FrontD *fr,
    const double *Lx, double *Winv)
{
    __shared__ double Lm [64 * 64] ;        // Lm [e*64 + r] = L(r,e)
    __shared__ double Wl [64 * 65] ;        // Wl [k*65 + q] = W(k,q)
    __shared__ double rdl [64] ;
    InvTask T = tasks [blockIdx.x] ;
    const FrontD &f = fr [T.front] ;
    int nsrow = f.nsrow, q = threadIdx.x ;
    int nb = f.nscol - T.jb < 64 ? f.nscol - T.jb : 64 ;
    const double *L = Lx + f.psx + T.jb + colx<CX> (T.jb, nsrow) ;
    for (int e = 0 ; e < 64 ; e++)
        Lm [e * 64 + q] = (q < nb && e < nb && e <= q) ? ldcx<CX> (L, q, e, nsrow) : (q == e ? 1.0 : 0.0) ;
    __syncthreads () ;
    rdl [q] = 1.0 / Lm [q * 64 + q] ;
    __syncthreads () ;
    // lane q = column q of the inverse, rows in blocks of 16
    for (int R = 0 ; R < 4 ; R++)
    {
        double acc [16] ;
#pragma unroll
        for (int r = 0 ; r < 16 ; r++) acc [r] = (16 * R + r == q) ? 1.0 : 0.0 ;
        for (int k = 0 ; k < 16 * R ; k++)
        {
            double wk = Wl [k * 65 + q] ;
#pragma unroll
            for (int r = 0 ; r < 16 ; r++) acc [r] = __builtin_fma (-Lm [k * 64 + 16 * R + r], wk, acc [r]) ;
        }
#pragma unroll
        for (int e = 0 ; e < 16 ; e++)
        {
            double y = acc [e] * rdl [16 * R + e] ;
            Wl [(16 * R + e) * 65 + q] = y ;
#pragma unroll
            for (int r = e + 1 ; r < 16 ; r++) acc [r] = __builtin_fma (-Lm [(16 * R + e) * 64 + 16 * R + r], y, acc [r]) ;
        }
    }
    __syncthreads () ;
    double *Wm = Winv + T.w_off, *WmT = Wm + 4096 ;
    for (int k = 0 ; k < 64 ; k++)
    {
        Wm [k * 64 + q] = Wl [q * 65 + k] ;     // W(r = q, k)
        WmT [k * 64 + q] = Wl [k * 65 + q] ;    // W(k, c = q)
    }
}

// One step of the walk for every big supernode of a level at once: task t = block
// [jb, jb+w) of one supernode, workgroups wg_start .. of the launch belong to it.

__device__ __forceinline__ int find_solve_task (const SolveBlk *t, int nt, int b)
{
    int lo = 0, hi = nt - 1 ;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1 ; if (t [mid].wg_start <= b) lo = mid ; else hi = mid - 1 ; }
    return lo ;
}

// forward, step 1 (one workgroup per task): x_b = inv(L_bb) x_b by 64-column
// sub-blocks -- explicit inverses on the diagonal, matrix-vector products below it.
// The solved x_b goes to the side vector Y (k_solve_commit copies it back per level).
template <bool CX>
__global__ void __launch_bounds__(256) k_solve_fwd_diag (const SolveBlk *tasks,
    const FrontD *fr, const double *Lx, const double *Winv, const double *X, i64 ldx, int nrhs, double *Y)
{
    __shared__ double xs [SOLVE_SB], t [64], part [4][64] ;
    const SolveBlk T = tasks [blockIdx.x] ;
    const FrontD &f = fr [T.front] ;
    const int nsrow = f.nsrow, k1 = f.k1, tid = threadIdx.x ;
    const int jb = T.jb, w = T.w, nsub = (w + 63) >> 6 ;
    const int r = tid & 63, p = tid >> 6 ;
    const double *L = Lx + f.psx ;
    // everything this thread will ever need of L_bb leaves for the registers at once
    // (sub-step k: row 64 k + r, columns == p (mod 4) before the sub-block: 16 k values;
    // 96 in all, one HBM latency instead of one per sub-step); the chain below then
    // only waits for LDS and for the 64 x 64 inverses (L2)
    double l1 [16], l2 [32], l3 [48] ;
    {
#pragma unroll
        for (int u = 0 ; u < 16 ; u++) l1 [u] = (64 + r < w) ? ldcx<CX> (L, jb + 64 + r, jb + p + 4 * u, nsrow) : 0.0 ;
#pragma unroll
        for (int u = 0 ; u < 32 ; u++) l2 [u] = (128 + r < w) ? ldcx<CX> (L, jb + 128 + r, jb + p + 4 * u, nsrow) : 0.0 ;
#pragma unroll
        for (int u = 0 ; u < 48 ; u++) l3 [u] = (192 + r < w) ? ldcx<CX> (L, jb + 192 + r, jb + p + 4 * u, nsrow) : 0.0 ;
    }
    // ... and so do the four 64 x 64 inverses (16 values per thread and sub-step): fetched
    // inside the chain they cost one L2 / HBM latency per sub-step, four per launch
    double w0 [16], w1 [16], w2 [16], w3 [16] ;
    {
        const double *Wm = Winv + (i64) T.inv * 8192 + r ;
#pragma unroll
        for (int u = 0 ; u < 16 ; u++)
        {
            const int o = (p + 4 * u) * 64 ;
            w0 [u] = Wm [o] ;
            w1 [u] = (nsub > 1) ? Wm [8192 + o] : 0.0 ;
            w2 [u] = (nsub > 2) ? Wm [2 * 8192 + o] : 0.0 ;
            w3 [u] = (nsub > 3) ? Wm [3 * 8192 + o] : 0.0 ;
        }
    }
    for (int rhs = 0 ; rhs < nrhs ; rhs++)
    {
        const double *x = X + (i64) rhs * ldx ;
        xs [tid] = (tid < w) ? x [k1 + jb + tid] : 0.0 ;
        __syncthreads () ;
        for (int k = 0 ; k < nsub ; k++)
        {
            // t = x_k - L[k-th row block, columns before it] * (solved part)
            double a0 = 0.0, a1 = 0.0 ;
            if (k == 1)
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u += 2) { a0 = __builtin_fma (l1 [u], xs [p + 4 * u], a0) ; a1 = __builtin_fma (l1 [u + 1], xs [p + 4 * u + 4], a1) ; }
            }
            else if (k == 2)
            {
#pragma unroll
                for (int u = 0 ; u < 32 ; u += 2) { a0 = __builtin_fma (l2 [u], xs [p + 4 * u], a0) ; a1 = __builtin_fma (l2 [u + 1], xs [p + 4 * u + 4], a1) ; }
            }
            else if (k == 3)
            {
#pragma unroll
                for (int u = 0 ; u < 48 ; u += 2) { a0 = __builtin_fma (l3 [u], xs [p + 4 * u], a0) ; a1 = __builtin_fma (l3 [u + 1], xs [p + 4 * u + 4], a1) ; }
            }
            part [p][r] = a0 + a1 ;
            __syncthreads () ;
            if (tid < 64) t [tid] = xs [64 * k + tid] - ((part [0][tid] + part [1][tid]) + (part [2][tid] + part [3][tid])) ;
            __syncthreads () ;
            double a = 0.0 ;                                            // W(r, kk), kk == p (mod 4)
            if (k == 0)
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) a = __builtin_fma (w0 [u], t [p + 4 * u], a) ;
            }
            else if (k == 1)
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) a = __builtin_fma (w1 [u], t [p + 4 * u], a) ;
            }
            else if (k == 2)
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) a = __builtin_fma (w2 [u], t [p + 4 * u], a) ;
            }
            else
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) a = __builtin_fma (w3 [u], t [p + 4 * u], a) ;
            }
            part [p][r] = a ;
            __syncthreads () ;
            if (tid < 64) xs [64 * k + tid] = (part [0][tid] + part [1][tid]) + (part [2][tid] + part [3][tid]) ;
            __syncthreads () ;
        }
        if (tid < w) Y [(i64) rhs * ldx + k1 + jb + tid] = xs [tid] ;
        __syncthreads () ;
    }
}

// forward, step 2: X[rows below] -= L[rows, b] x_b ; workgroup = (256-row chunk) x
// (64-column sub-block), so that a single supernode fills the chip
template <bool CX>
__global__ void __launch_bounds__(256) k_solve_fwd_apply (const SolveBlk *tasks, int ntasks,
    const FrontD *fr, const i64 *Ls, const double *Lx, double *X, i64 ldx, int nrhs, const double *Y)
{
    __shared__ double xs [64] ;
    const SolveBlk T = tasks [find_solve_task (tasks, ntasks, (int) blockIdx.x)] ;
    const FrontD &f = fr [T.front] ;
    const int nscol = f.nscol, nsrow = f.nsrow, k1 = f.k1, tid = threadIdx.x ;
    const int jb = T.jb, w = T.w, nsub = (w + 63) >> 6 ;
    const int wgl = (int) blockIdx.x - T.wg_start ;
    const int q = wgl % nsub, chunk = wgl / nsub ;
    const int c0 = 64 * q, cw = (w - c0 < 64) ? w - c0 : 64 ;
    const double *L = Lx + f.psx ;
    const i64 *rows = Ls + f.psi ;
    const int i = jb + w + chunk * 256 + tid ;
    for (int rhs = 0 ; rhs < nrhs ; rhs++)
    {
        double *x = X + (i64) rhs * ldx ;
        if (tid < 64) xs [tid] = (tid < cw) ? Y [(i64) rhs * ldx + k1 + jb + c0 + tid] : 0.0 ;
        __syncthreads () ;
        if (i < nsrow)
        {
            double acc [4] = {0.0, 0.0, 0.0, 0.0} ;
            int c = 0 ;
            for ( ; c + 16 <= cw ; c += 16)
            {
                double l [16] ;
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) l [u] = ldcx<CX> (L, i, jb + c0 + c + u, nsrow) ;
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) acc [u & 3] = __builtin_fma (l [u], xs [c + u], acc [u & 3]) ;
            }
            for ( ; c < cw ; c++) acc [0] = __builtin_fma (ldcx<CX> (L, i, jb + c0 + c, nsrow), xs [c], acc [0]) ;
            double sm = (acc [0] + acc [1]) + (acc [2] + acc [3]) ;
            // (several column sub-blocks, and siblings of the level, add into the same row)
            atomicAdd (i < nscol ? &x [k1 + i] : &x [rows [i]], -sm) ;
        }
        __syncthreads () ;
    }
}

// solved values of the big supernodes of a level back into X (one task per supernode)
__global__ void __launch_bounds__(256) k_solve_commit (const SolveBlk *tasks, int ntasks, const FrontD *fr,
    double *X, i64 ldx, int nrhs, const double *Y)
{
    const SolveBlk T = tasks [find_solve_task (tasks, ntasks, (int) blockIdx.x)] ;
    const FrontD &f = fr [T.front] ;
    int j = ((int) blockIdx.x - T.wg_start) * 256 + threadIdx.x ;
    if (j >= f.nscol) return ;
    for (int rhs = 0 ; rhs < nrhs ; rhs++) X [(i64) rhs * ldx + f.k1 + j] = Y [(i64) rhs * ldx + f.k1 + j] ;
}

// backward, step 1: acc (task) += L[rows, b]' x[rows] ; workgroup = (256-row chunk) x
// (64-column sub-block); thread = row, the column sums of a wave go through an LDS
// transpose, one atomic add per column and wave
template <bool CX>
__global__ void __launch_bounds__(256) k_solve_bwd_apply (const SolveBlk *tasks, int ntasks,
    const FrontD *fr, const i64 *Ls, const double *Lx, const double *X, i64 ldx, int nrhs, double *accbuf)
{
    __shared__ double Tw [4][64 * 17] ;      // per wave: 64 rows x 16 columns of products
    const SolveBlk T = tasks [find_solve_task (tasks, ntasks, (int) blockIdx.x)] ;
    const FrontD &f = fr [T.front] ;
    const int nscol = f.nscol, nsrow = f.nsrow, k1 = f.k1, tid = threadIdx.x ;
    const int lane = tid & 63, wave = tid >> 6 ;
    const int jb = T.jb, w = T.w, nsub = (w + 63) >> 6 ;
    const int wgl = (int) blockIdx.x - T.wg_start ;
    const int qsub = wgl % nsub, chunk = wgl / nsub ;
    const double *L = Lx + f.psx ;
    const i64 *rows = Ls + f.psi ;
    double *acc = accbuf + (i64) T.slot * nrhs * SOLVE_SB ;
    const int r0 = jb + w + chunk * 256 ;
    const int nr = nsrow - r0 < 256 ? nsrow - r0 : 256 ;
    if (nr <= 0) return ;
    const int i = r0 + tid ;
    const bool ok = tid < nr ;
    const int c16 = lane & 15, seg = lane >> 4 ;
    for (int rhs = 0 ; rhs < nrhs ; rhs++)
    {
        const double *x = X + (i64) rhs * ldx ;
        double y = ok ? ((i < nscol) ? x [k1 + i] : x [rows [i]]) : 0.0 ;
        for (int q = 4 * qsub ; q < 4 * qsub + 4 ; q++)
        {
            double l [16] ;
#pragma unroll
            for (int c = 0 ; c < 16 ; c++) l [c] = ldcx<CX> (L, ok ? i : r0, jb + 16 * q + (16 * q + c < w ? c : 0), nsrow) ;
#pragma unroll
            for (int c = 0 ; c < 16 ; c++) Tw [wave][lane * 17 + c] = (16 * q + c < w) ? l [c] * y : 0.0 ;
            __builtin_amdgcn_s_waitcnt (0xc07f) ;      // lgkmcnt(0): own wave's LDS writes landed
            __builtin_amdgcn_wave_barrier () ;
            double sum = 0.0 ;
#pragma unroll
            for (int rr = 0 ; rr < 16 ; rr++) sum += Tw [wave][(seg * 16 + rr) * 17 + c16] ;
            sum += __shfl_xor (sum, 16) ;
            sum += __shfl_xor (sum, 32) ;
            if (seg == 0 && 16 * q + c16 < w) atomicAdd (&acc [rhs * SOLVE_SB + 16 * q + c16], sum) ;
            __builtin_amdgcn_wave_barrier () ;
        }
    }
}

// backward, step 2 (one workgroup per task): x_b = inv(L_bb)' (x_b - acc) by
// sub-blocks from the bottom; the accumulator is cleared for the next step
template <bool CX>
__global__ void __launch_bounds__(256) k_solve_bwd_diag (const SolveBlk *tasks,
    const FrontD *fr, const double *Lx, const double *Winv, double *X, i64 ldx, int nrhs, double *accbuf)
{
    __shared__ double xs [SOLVE_SB], t [64], part [4][64] ;
    __shared__ double Tw [4][64 * 17] ;
    const SolveBlk T = tasks [blockIdx.x] ;
    const FrontD &f = fr [T.front] ;
    const int nsrow = f.nsrow, k1 = f.k1, tid = threadIdx.x ;
    const int jb = T.jb, w = T.w, nsub = (w + 63) >> 6 ;
    const double *L = Lx + f.psx ;
    double *acc = accbuf + (i64) T.slot * nrhs * SOLVE_SB ;
    const int r = tid & 63, p = tid >> 6 ;
    const int lane = tid & 63, wave = tid >> 6, c16 = lane & 15, seg = lane >> 4 ;
    // sub-step k needs L(rows below sub-block k inside the block, its 64 columns):
    // wave = 16 of the columns, lane = row (coalesced), all 96 values per thread
    // requested at once
    double m2 [16], m1 [32], m0 [48] ;
    {
#pragma unroll
        for (int c = 0 ; c < 16 ; c++)
        {
            m2 [c] = (192 + lane < w) ? ldcx<CX> (L, jb + 192 + lane, jb + 128 + 16 * wave + c, nsrow) : 0.0 ;
#pragma unroll
            for (int j = 0 ; j < 2 ; j++) m1 [2 * c + j] = (128 + 64 * j + lane < w) ? ldcx<CX> (L, jb + 128 + 64 * j + lane, jb + 64 + 16 * wave + c, nsrow) : 0.0 ;
#pragma unroll
            for (int j = 0 ; j < 3 ; j++) m0 [3 * c + j] = (64 + 64 * j + lane < w) ? ldcx<CX> (L, jb + 64 + 64 * j + lane, jb + 16 * wave + c, nsrow) : 0.0 ;
        }
    }
    // the transposed 64 x 64 inverses as well (see k_solve_fwd_diag)
    double w0 [16], w1 [16], w2 [16], w3 [16] ;
    {
        const double *WmT = Winv + (i64) T.inv * 8192 + 4096 + r ;
#pragma unroll
        for (int u = 0 ; u < 16 ; u++)
        {
            const int o = (p + 4 * u) * 64 ;
            w0 [u] = WmT [o] ;
            w1 [u] = (nsub > 1) ? WmT [8192 + o] : 0.0 ;
            w2 [u] = (nsub > 2) ? WmT [2 * 8192 + o] : 0.0 ;
            w3 [u] = (nsub > 3) ? WmT [3 * 8192 + o] : 0.0 ;
        }
    }
    for (int rhs = 0 ; rhs < nrhs ; rhs++)
    {
        double *x = X + (i64) rhs * ldx ;
        xs [tid] = (tid < w) ? x [k1 + jb + tid] - acc [rhs * SOLVE_SB + tid] : 0.0 ;
        acc [rhs * SOLVE_SB + tid] = 0.0 ;
        __syncthreads () ;
        for (int k = nsub - 1 ; k >= 0 ; k--)
        {
            // t[c] = xs[64k + c] - sum over the solved rows below (inside the block) of L(row, 64k + c) xs[row]
            double sm [16] ;
#pragma unroll
            for (int c = 0 ; c < 16 ; c++) sm [c] = 0.0 ;
            if (k == 2)
            {
                double y = xs [192 + lane] ;
#pragma unroll
                for (int c = 0 ; c < 16 ; c++) sm [c] = m2 [c] * y ;
            }
            else if (k == 1)
            {
                double y0 = xs [128 + lane], y1 = xs [192 + lane] ;
#pragma unroll
                for (int c = 0 ; c < 16 ; c++) sm [c] = __builtin_fma (m1 [2 * c + 1], y1, m1 [2 * c] * y0) ;
            }
            else if (k == 0)
            {
                double y0 = xs [64 + lane], y1 = xs [128 + lane], y2 = xs [192 + lane] ;
#pragma unroll
                for (int c = 0 ; c < 16 ; c++) sm [c] = __builtin_fma (m0 [3 * c + 2], y2, __builtin_fma (m0 [3 * c + 1], y1, m0 [3 * c] * y0)) ;
            }
            // column sums over the wave's 64 rows through an LDS transpose
#pragma unroll
            for (int c = 0 ; c < 16 ; c++) Tw [wave][lane * 17 + c] = sm [c] ;
            __builtin_amdgcn_s_waitcnt (0xc07f) ;
            __builtin_amdgcn_wave_barrier () ;
            double sum = 0.0 ;
#pragma unroll
            for (int rr = 0 ; rr < 16 ; rr++) sum += Tw [wave][(seg * 16 + rr) * 17 + c16] ;
            sum += __shfl_xor (sum, 16) ;
            sum += __shfl_xor (sum, 32) ;
            if (seg == 0) t [16 * wave + c16] = xs [64 * k + 16 * wave + c16] - sum ;
            __syncthreads () ;
            double a = 0.0 ;                                            // W(kk, c), kk == p (mod 4)
            if (k == 0)
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) a = __builtin_fma (w0 [u], t [p + 4 * u], a) ;
            }
            else if (k == 1)
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) a = __builtin_fma (w1 [u], t [p + 4 * u], a) ;
            }
            else if (k == 2)
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) a = __builtin_fma (w2 [u], t [p + 4 * u], a) ;
            }
            else
            {
#pragma unroll
                for (int u = 0 ; u < 16 ; u++) a = __builtin_fma (w3 [u], t [p + 4 * u], a) ;
            }
            part [p][r] = a ;
            __syncthreads () ;
            if (tid < 64) xs [64 * k + tid] = (part [0][tid] + part [1][tid]) + (part [2][tid] + part [3][tid]) ;
            __syncthreads () ;
        }
        if (tid < w) x [k1 + jb + tid] = xs [tid] ;
        __syncthreads () ;
    }
}

// gather / scatter by the fill-reducing permutation (cholmod_solve.c:105,:322)
__global__ void k_perm (i64 n, const i64 *perm, const double *src, double *dst,
    int inverse)
{
    i64 k = blockIdx.x * (i64) 256 + threadIdx.x ;
    if (k >= n) return ;
    if (inverse) dst [perm [k]] = src [k] ; else dst [k] = src [perm [k]] ;
}

// ---- size-independent invariants of a device-resident factor ----------------
// One pass over Lx (parity checks at sizes the CPU oracle cannot reach):
//   out[0] += sum_j log L(j,j)          (= logdet(A)/2, analytic for Poisson grids)
//   out[1] += entries != 0 in the dead strictly-upper triangles of the diagonal
//             blocks (the reference never writes them)
//   out[2] += non-finite entries of the lower trapezoids
//   out[3] += sum of squares of the lower trapezoids (||L||_F^2)
//   out[4] += diagonal entries <= 0
// A workgroup owns CHK_COLS columns of one supernode; wave w takes the columns
// == w (mod 4), lanes stride the rows (coalesced).
// ---- even columns of the factor (complex input through the real embedding) ---------
// The engine factors the 2n x 2n embedding of a complex matrix (host/complex.c); the
// interleaved complex column j of a supernode is the even column 2j of the real one.
// out (packed: supernode s at px [s] / 2, i.e. 2 * nsrow_c * nscol_c doubles) receives
// them, the imaginary part of the diagonal as the exact zero zpotrf leaves.
__global__ void __launch_bounds__(256) k_even_columns (const CheckTask *tasks, const FrontD *fr,
    const double *Lx, double *out)
{
    const CheckTask T = tasks [blockIdx.x] ;
    const FrontD &f = fr [T.front] ;
    const int nsrow = f.nsrow, nscol = f.nscol ;
    const double *L = Lx + f.psx ;
    double *O = out + f.psx / 2 ;
    const int c1 = T.c0 + CHK_COLS < nscol ? T.c0 + CHK_COLS : nscol ;
    for (int c = T.c0 + (T.c0 & 1) ; c < c1 ; c += 2)
        for (int i = threadIdx.x ; i < nsrow ; i += 256)
            O [(i64) (c >> 1) * nsrow + i] = (i == c + 1) ? 0.0 : L [(i64) c * nsrow + i] ;
}

// (CX: the checks run on the twin the storage stands for -- ldx rebuilds the odd columns)
template <bool CX>
__global__ void __launch_bounds__(256) k_factor_checks (const CheckTask *tasks, const FrontD *fr,
    const double *Lx, double *out)
{
    CheckTask T = tasks [blockIdx.x] ;
    const FrontD &f = fr [T.front] ;
    int nsrow = f.nsrow, nscol = f.nscol ;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6 ;
    const double *L = Lx + f.psx ;
    double slog = 0.0, sq = 0.0 ;
    double nup = 0.0, nbad = 0.0, nneg = 0.0 ;
    int c1 = T.c0 + CHK_COLS < nscol ? T.c0 + CHK_COLS : nscol ;
    for (int j = T.c0 + wave ; j < c1 ; j += 4)
    {
        for (int i = lane ; i < nsrow ; i += 64)
        {
            double v = ldcx<CX> (L, i, col_local (f, j), nsrow) ;      // (a distributed front: the tasks cover this rank's slabs)
            if (i < j) { if (v != 0.0) nup += 1.0 ; }
            else
            {
                if (!(v - v == 0.0)) nbad += 1.0 ; else sq = __builtin_fma (v, v, sq) ;
                if (i == j) { if (v > 0.0) slog += log (v) ; else nneg += 1.0 ; }
            }
        }
    }
    double acc [5] = {slog, nup, nbad, sq, nneg} ;
    __shared__ double red [4][5] ;
#pragma unroll
    for (int q = 0 ; q < 5 ; q++)
    {
        double v = acc [q] ;
        for (int o = 32 ; o > 0 ; o >>= 1) v += __shfl_down (v, o) ;
        if (lane == 0) red [wave][q] = v ;
    }
    __syncthreads () ;
    if (threadIdx.x < 5)
    {
        double v = (red [0][threadIdx.x] + red [1][threadIdx.x]) + (red [2][threadIdx.x] + red [3][threadIdx.x]) ;
        if (v != 0.0) atomicAdd (&out [threadIdx.x], v) ;
    }
}

} // namespace sship
