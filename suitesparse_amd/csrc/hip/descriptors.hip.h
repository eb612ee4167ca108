// descriptors.hip.h -- what the host side of the engine and its device kernels share: the per-front and per-launch-group
// descriptors the plan builder fills and the kernels read, and the blocking constants both sides must agree on.  No device
// code: the host-only translation units (plan_build.hip, schedule_dense.hip) include this file, not kernels.hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef long long i64;
typedef int i32;

namespace sship {

// ---- device-side descriptors ------------------------------------------------

struct FrontD {
    i64 psx;        // offset of the panel in Lx            (L->px[s])
    i64 psi;        // offset of the row list in Ls         (L->pi[s])
    i64 cb;         // offset of the contribution block in the arena
    i64 rel;        // offset of this front's child->parent relative map
    i32 k1;         // first column                         (L->super[s])
    i32 nscol, nsrow, ncb;
    i32 parent;     // supernodal etree parent or -1
    i32 child_begin, child_end;   // range in the child index array (this rank's view)
    i32 assemble;   // 1: k_assemble scatters A into this front on this rank, 2: the
                    // fused thin-front kernel does, 0: another rank does
    i32 cbp;        // 1: the contribution block is stored as a packed lower triangle
                    // (column j holds rows j..ncb-1; written by k_thin_front), 0: as a
                    // full square with ld = ncb (written by the dense update kernel)
    // Several GPUs, a front shared by a rank group (round 4): its panel is DISTRIBUTED over the group by
    // slabs of own_w columns, slab t on member t % own_g -- a rank stores only its own slabs, packed one
    // behind the other with ld = nsrow (psx = offset of the first one in the rank's array).  own_w == 0:
    // the whole panel is here (private fronts, one GPU, the gathered factor).
    i32 own_w, own_g, own_r;
    // ... and so is its contribution block (cbd = 1), by BLOCKS of columns: member r stores columns [cb_lo, cb_hi) of it
    // (boundaries at equal shares of the lower triangle's area, multiples of 64), ld = ncb, and is the one that applies
    // the outer updates to them.  Nothing is extend-added INTO such a block: what the descendants contribute to it is
    // routed past it, straight into the ancestor whose panel holds the column (engine.hip: contributors).
    i32 cbd;
    i32 cb_lo, cb_hi;
};
// column c of front f: is it stored on this rank, and where (in columns from psx)
__host__ __device__ __forceinline__ bool col_owned (const FrontD &f, int c) { return f.own_w == 0 || ((c / f.own_w) % f.own_g) == f.own_r ; }
__host__ __device__ __forceinline__ int col_local (const FrontD &f, int c) { return f.own_w == 0 ? c : ((c / f.own_w) / f.own_g) * f.own_w + c % f.own_w ; }
// front columns < c stored on this rank (= col_local (f, c) when c itself is)
__host__ __device__ __forceinline__ int owned_before (const FrontD &f, int c)
{
    if (f.own_w == 0) return c ;
    const int t = c / f.own_w ;
    const int full = t > f.own_r ? (t - f.own_r + f.own_g - 1) / f.own_g : 0 ;       // owned slabs before slab t
    return full * f.own_w + ((t % f.own_g) == f.own_r ? c % f.own_w : 0) ;
}

// what a parent needs of a child, in the order of the child lists: one load instead of the chain
// child [ci] -> fr [child] -> (cb, rel, ncb, cbp)
struct ChildD { i64 cb; i64 rel; i32 ncb; i32 cbp; };
struct EaGroup { i32 front; i32 blk_start; i32 c_lo; i32 c_hi;      // extend-add into target columns [c_lo, c_hi)
                 i64 pbase; };                                        // panel columns live at pbase + c ld (EA_NO_PBASE: at the front's psx;
                                                                      // a window's virtual base may well be negative);
                                                                      // a shared front's block column in its window (engine.hip)
#define EA_NO_PBASE INT64_MIN
struct ZeroGroup { i64 off; i64 len; i32 blk_start; i32 pad; };
struct PfGroup { i64 off; i32 lda; i32 nb; i32 front; i32 col0; };
struct TrGroup { i64 l_off; i64 b_off; i32 lda; i32 m; i32 nb; i32 front;
                 i32 col0; i32 blk_start; };
// the 256-column panel chain (k_diag / k_rowsolve): a diagonal sub-block of a front and the
// rows below it
struct DgGroup { i64 off; i32 lda; i32 w; i32 front; i32 col0; i32 slot; i32 pad; };
struct RsGroup { i64 l_off; i64 b_off; i32 lda; i32 m; i32 w; i32 front; i32 col0; i32 blk_start; i32 slot; i32 pad; };
struct GemmGroup {
    i64 a_off, b_off, c_off;    // a/b index Lx; c indexes Lx or the CB arena
    i32 lda, ldc;
    i32 m, n, k;                // target region m x n, contraction length k
    i32 tri;                    // 1: region starts on the diagonal (row0==col0):
                                //    only tiles with I>=J, and i>=j inside
    i32 c_in_cb;                // 1: C lives in the CB arena
    i32 tile_start;             // first block of this group in the launch
    i32 mt, nt;                 // tile grid
    i32 front;
    i32 tile_mul, tile_add;     // multi-GPU: 64-tile chunk c of a shared front's outer
                                // update belongs to the rank with c % tile_mul == tile_add
    i32 ntiles;                 // tiles of the region (all ranks)
    i32 nblk;                   // blocks this launch spends on the group (this rank)
    i32 swz;                    // 1: XCD-aware super-tile walk (big groups)
    i32 assign;                 // 1: C = -A*B' (first update of a contribution block: no zero-fill, no read)
    i32 pf_next, pf_col0;       // k_update2f: tile (0,0) of the region is the next 64 x 64 diagonal block of
                                // the front (its first column: pf_col0) and is factored by the workgroup that updates it
    i32 tile_cnt;               // multi-GPU, > 0: this rank's share of the region is the RANGE of tile_cnt 64-tile chunks
                                // that starts at chunk tile_add (tile_mul = 1): the contribution block of a distributed
                                // front, dealt so that it evens out what the members' own slabs differ by
};

// Contribution blocks of the generic fronts are stored as full squares, ld = ncb
// (lower part used); those of the thin fronts as packed lower triangles:
// element (i,j), i >= j, of a packed triangle of order m lives at tri_col(j,m) + i.
__host__ __device__ __forceinline__ int tri_col (int j, int m) { return j * m - ((j * (j + 1)) >> 1) ; }

// (contributor, ancestor) pair whose relative map is computed next to the child -> parent ones (k_relmap_pairs)
struct RelPair { i32 d ; i32 a ; i64 off ; } ;
// k_zero: a block owns ZERO_COLS columns
#define ZERO_CHUNK 8192
#define ZERO_COLS 8
// k_extend_add: target columns per workgroup (16: 4.5 / 15.8 / 1.08 ms of extend-add at the nd24k stand-in / Poisson 100^3 / 2D 1259^2; 8: 3.9 / 14.5 / 0.96; 4: 3.7 / 14.4 / 1.05; 32: 5.6 / 16.0 / 1.48)
#ifndef EA_TW
#define EA_TW 8           // (16: 4.5 / 15.8 / 1.08 ms of extend-add at the nd24k stand-in / Poisson 100^3 / 2D 1259^2; 8: 3.9 / 14.5 / 0.96; 4: 3.7 / 14.4 / 1.05; 32: 5.6 / 16.0 / 1.48)
#endif
// doubles every device array that holds panels of L ends with: k_update3's partial tiles read up to 63 rows past a column's end
#define UPD3_LX_PAD 64
// inner panel width (k_potrf_mfma, k_trsm_mfma) and the leading dimension of its LDS copy
#define PF_NB 64
#define PF2_LD 64
// rows per workgroup of k_trsm_mfma / k_trsm_upd
#define TRM_ROWS 64
// thin fronts (k_thin_front): most rows, panel width
#define SM_MAX 136
#define TF_PW 16
// the exchange of a block column of a shared front (k_xchg_move; see kernels.hip.h)
struct XchgD {
    i64 slab ;      // offset in Lx of entry (b0, b0) of the front
    i32 lda ;       // nsrow
    i32 w ;         // columns of the block column
    i32 mb ;        // NEAR rows: below the diagonal block, inside the outer block column (o1 - b0 - w)
    i32 R ;         // near rows per chunk (g R >= mb)
    i32 g, r ;      // group size, this rank's index in the group
    i32 fo ;        // FAR rows: first one, counted from the diagonal block's first row (o1 - b0) ...
    i32 mf ;        // ... how many (nsrow - o1) ...
    i32 Rf ;        // ... and per chunk (g Rf >= mf): the same chunks for every block column of the outer block
} ;
// k_win_move: one workgroup = one column x WIN_ROWS rows
#define WIN_ROWS 8192
struct WinD { i64 store ; i64 win ; i32 ld ; i32 c0, c1 ; i32 r0 ; i32 nrows ; i32 own_w, own_g, own_r ; i32 mode ; i32 blk_start ; } ;
// the 256-column chain (k_diag / k_rowsolve / k_chainf)
#define DG_W 256
#define RS_ROWS 64
struct CfGroup { i64 l_off ; i32 lda ; i32 w ; i32 front ; i32 col0 ; i32 m1 ; i32 slot ; i32 fslot ; i32 dstart ; i32 bstart ; i32 off2 ; i32 m2 ; i32 off3 ; i32 m3 ; i32 pad ; } ;
// triangular solves
struct SolveTask { i32 front ; i32 c0, c1 ; i32 below ; } ;   // columns [c0,c1) of a supernode
#define SOLVE_IB 64          /* diagonal blocks with an explicit inverse (k_diag_inv64) */
#define SOLVE_SB 256         /* column block of the big-front walk: four inverse blocks */
#define SOLVE_BIG_COLS 256   /* fronts wider than this (or > 512 KB) take the multi-workgroup walk */
struct InvTask { i32 front ; i32 jb ; i64 w_off ; } ;
struct SolveBlk { i32 front, jb, w, wg_start, inv, slot ; } ;   // inv: index of its first 64 x 64 inverse
// factor checks
#define CHK_COLS 64
struct CheckTask { i32 front ; i32 c0 ; } ;

} // namespace sship
