// probe_kernels.hip.h -- micro-benchmarks, latency probes and the superseded
// first-generation kernels (VALU cross-checks).  NOT part of the product library:
// compiled only into lib/libcholmod_amd_probes.so (tools/, bench.py's measured
// ceilings).
#pragma once
#include "kernels.hip.h"

namespace sship {

#define PF_NB 64
#define PF_LD 66            /* even (16-B aligned pairs) and conflict-free for b128 */
#define PF_CW 8             /* columns eliminated per group */
// One wave, lane = row.  The block is processed in groups of PF_CW columns:
//  (1) left-looking update of the lane's PF_CW entries by all earlier columns
//      (PF_CW independent FMA chains; own row read contiguously from T, the
//      multipliers L(jb..jb+7,k) read as one broadcast line from the k-major
//      copy Tt);
//  (2) the PF_CW x PF_CW diagonal block is factored redundantly by every lane
//      in registers (no cross-lane traffic), (3) each lane solves its own row
//      against it.  Two barriers per group instead of one per column.
// The block is identity-padded to a multiple of PF_CW, so nb < 64 needs no
// special cases.
template <bool TIMED>
__global__ void __launch_bounds__(64) k_potrf (const PfGroup *g, double *Lx, i32 *info, long long *tim)
{
    long long tc [8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0 ;
    auto tick = [&] (int slot) { if constexpr (TIMED) { long long t = __builtin_readcyclecounter () ; tc [slot] += t - t_prev ; t_prev = t ; } } ;
    if constexpr (TIMED) t_prev = __builtin_readcyclecounter () ;
    __shared__ __attribute__((aligned(16))) double T [PF_NB * PF_LD] ;   // T[i][k] = L(i,k)
    __shared__ __attribute__((aligned(16))) double Tt [PF_NB * PF_NB] ;  // Tt[k][i] = L(i,k)
    // latency-critical single wave: outrank the MFMA update waves it may share
    // a SIMD with when the look-ahead stream overlaps it with a trailing update
    __builtin_amdgcn_s_setprio (3) ;
    PfGroup G = g [blockIdx.x] ;
    double *A = Lx + G.off ;
    int nb = G.nb, lda = G.lda, lane = threadIdx.x ;
    if (info [G.front] != 0)
    {
        for (int j = 0 ; j < nb ; j++)
            if (lane >= j && lane < nb) A [lane + (i64) j * lda] = 0.0 ;
        return ;
    }
    int nbp = (nb + PF_CW - 1) / PF_CW * PF_CW ;
    {
        // stage the block: 8 independent column loads in flight per step
        int li = lane < nb ? lane : nb - 1 ;
        for (int j0 = 0 ; j0 < nbp ; j0 += 8)
        {
            double tmp [8] ;
#pragma unroll
            for (int c = 0 ; c < 8 ; c++)
            {
                int j = j0 + c < nb ? j0 + c : nb - 1 ;
                tmp [c] = A [li + (i64) j * lda] ;
            }
#pragma unroll
            for (int c = 0 ; c < 8 ; c++)
            {
                int j = j0 + c ;
                double v = (lane < nb && j < nb) ? tmp [c] : (lane == j ? 1.0 : 0.0) ;
                T [lane * PF_LD + j] = v ;
            }
        }
    }
    __syncthreads () ;
    tick (0) ;
    int fail = -1 ;
    for (int jb = 0 ; jb < nbp ; jb += PF_CW)
    {
        double a [PF_CW] ;
#pragma unroll
        for (int c = 0 ; c < PF_CW ; c++) a [c] = T [lane * PF_LD + jb + c] ;
        for (int k = 0 ; k < jb ; k += 4)
        {
            double x [4] ;
#pragma unroll
            for (int u = 0 ; u < 4 ; u++) x [u] = T [lane * PF_LD + k + u] ;
#pragma unroll
            for (int u = 0 ; u < 4 ; u++)
            {
                const double *lt = Tt + (k + u) * PF_NB + jb ;
#pragma unroll
                for (int c = 0 ; c < PF_CW ; c++) a [c] -= x [u] * lt [c] ;
            }
        }
        tick (1) ;
#pragma unroll
        for (int c = 0 ; c < PF_CW ; c++) T [lane * PF_LD + jb + c] = a [c] ;
        __syncthreads () ;
        tick (2) ;
        // diagonal block, lower part, same values in every lane
        double D [PF_CW][PF_CW], rinv [PF_CW] ;
#pragma unroll
        for (int r = 0 ; r < PF_CW ; r++)
#pragma unroll
            for (int c = 0 ; c <= r ; c++) D [r][c] = T [(jb + r) * PF_LD + jb + c] ;
#pragma unroll
        for (int c = 0 ; c < PF_CW ; c++)
        {
            if (fail < 0)
            {
                double d = D [c][c] ;
                if (d <= 0.0) fail = jb + c ;
                else
                {
                    double r, ri ;
                    sqrt_rsqrt (d, r, ri) ;
                    D [c][c] = r ; rinv [c] = ri ;
#pragma unroll
                    for (int r2 = c + 1 ; r2 < PF_CW ; r2++) D [r2][c] *= ri ;
#pragma unroll
                    for (int c2 = c + 1 ; c2 < PF_CW ; c2++)
#pragma unroll
                        for (int r2 = c2 ; r2 < PF_CW ; r2++) D [r2][c2] -= D [r2][c] * D [c2][c] ;
                }
            }
            if (fail >= 0) { rinv [c] = 0.0 ; }
        }
        tick (3) ;
        // own row against the factored diagonal block
        double x [PF_CW] ;
#pragma unroll
        for (int c = 0 ; c < PF_CW ; c++)
        {
            double v = a [c] ;
#pragma unroll
            for (int e = 0 ; e < c ; e++) v -= x [e] * D [c][e] ;
            x [c] = v * rinv [c] ;
            if (lane == jb + c) x [c] = D [c][c] ;
            if (fail >= 0 && jb + c >= fail) x [c] = 0.0 ;
        }
#pragma unroll
        for (int c = 0 ; c < PF_CW ; c++)
        {
            T [lane * PF_LD + jb + c] = x [c] ;
            Tt [(jb + c) * PF_NB + lane] = x [c] ;
        }
        tick (4) ;
        __syncthreads () ;
        tick (5) ;
        if (fail >= 0) break ;
    }
    if (fail >= 0)
    {
        if (lane == 0) info [G.front] = G.col0 + fail + 1 ;
        for (int j = fail ; j < nb ; j++) T [lane * PF_LD + j] = 0.0 ;
    }
    __syncthreads () ;
    for (int j0 = 0 ; j0 < nb ; j0 += 8)
    {
#pragma unroll
        for (int c = 0 ; c < 8 ; c++)
        {
            int j = j0 + c ;
            if (j < nb && lane >= j && lane < nb) A [lane + (i64) j * lda] = T [lane * PF_LD + j] ;
        }
    }
    tick (6) ;
    if constexpr (TIMED) { if (lane == 0) for (int q = 0 ; q < 8 ; q++) tim [q] = tc [q] ; }
}


// ---- tuning probe: issue / latency of the fp64 ops the panel kernels chain -----
// one wave; out[v] = shader-clock cycles for n repetitions of variant v
__global__ void __launch_bounds__(64) k_latency_probe (double *sink, long long *out, int n)
{
    double x = 1.0 + threadIdx.x * 1e-9, a = 0.999999, b = 1e-7 ;
    long long t0, t1 ;
    // 0: dependent v_fma_f64
    t0 = __builtin_readcyclecounter () ;
    for (int i = 0 ; i < n ; i++) { x = __builtin_fma (x, a, b) ; asm volatile ("" : "+v" (x)) ; }
    t1 = __builtin_readcyclecounter () ; out [0] = t1 - t0 ;
    // 1: 8 independent v_fma_f64 chains (issue rate)
    double y [8] ;
    for (int q = 0 ; q < 8 ; q++) y [q] = x + q ;
    t0 = __builtin_readcyclecounter () ;
    for (int i = 0 ; i < n ; i++)
    {
#pragma unroll
        for (int q = 0 ; q < 8 ; q++) { y [q] = __builtin_fma (y [q], a, b) ; asm volatile ("" : "+v" (y [q])) ; }
    }
    t1 = __builtin_readcyclecounter () ; out [1] = t1 - t0 ;
    for (int q = 0 ; q < 8 ; q++) x += y [q] ;
    // 2: dependent v_rcp_f64
    t0 = __builtin_readcyclecounter () ;
    for (int i = 0 ; i < n ; i++) { x = __builtin_amdgcn_rcp (x) ; asm volatile ("" : "+v" (x)) ; }
    t1 = __builtin_readcyclecounter () ; out [2] = t1 - t0 ;
    // 3: readlane pair -> fma with the scalar, 8 independent accumulators
    t0 = __builtin_readcyclecounter () ;
    for (int i = 0 ; i < n ; i++)
    {
#pragma unroll
        for (int q = 0 ; q < 8 ; q++)
        {
            double sv = readlane_f64 (x, q) ;
            y [q] = __builtin_fma (sv, a, y [q]) ; asm volatile ("" : "+v" (y [q])) ;
        }
    }
    t1 = __builtin_readcyclecounter () ; out [3] = t1 - t0 ;
    for (int q = 0 ; q < 8 ; q++) x += y [q] ;
    // 4: dependent chain through readlane: x -> readlane -> fma -> x
    t0 = __builtin_readcyclecounter () ;
    for (int i = 0 ; i < n ; i++)
    {
        double sv = readlane_f64 (x, 3) ;
        x = __builtin_fma (sv, a, b) ; asm volatile ("" : "+v" (x)) ;
    }
    t1 = __builtin_readcyclecounter () ; out [4] = t1 - t0 ;
    // 5: dependent v_mul_f64
    t0 = __builtin_readcyclecounter () ;
    for (int i = 0 ; i < n ; i++) { x = x * a ; asm volatile ("" : "+v" (x)) ; }
    t1 = __builtin_readcyclecounter () ; out [5] = t1 - t0 ;
    // 6: dependent MFMA accumulate chain
    d4 acc = {x, x, x, x} ;
    t0 = __builtin_readcyclecounter () ;
    for (int i = 0 ; i < n ; i++) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64 (a, b, acc, 0, 0, 0) ; }
    t1 = __builtin_readcyclecounter () ; out [6] = t1 - t0 ;
    x += acc [0] ;
    // 7: LDS broadcast read -> fma dependent chain
    __shared__ double lds [64] ;
    lds [threadIdx.x] = a ;
    __syncthreads () ;
    t0 = __builtin_readcyclecounter () ;
    for (int i = 0 ; i < n ; i++)
    {
        int idx = ((int) x) & 63 ;
        x = __builtin_fma (lds [idx], x, b) ; asm volatile ("" : "+v" (x)) ;
    }
    t1 = __builtin_readcyclecounter () ; out [7] = t1 - t0 ;
    sink [threadIdx.x] = x ;
}


// ---- panel triangular solve: B := B * inv(L11)' , one thread per row --------
// dtrsm("R","L","C","N") of the reference (:997-1002).  L11' (nb <= 64) is
// staged in LDS; a thread solves its row in groups of TR_CW columns held in
// registers: the contribution of earlier column groups comes from LDS copies of
// the solved values (column-major over the workgroup's rows: conflict-free) and
// a broadcast line of L11', the TR_CW x TR_CW diagonal block is applied through
// its explicit inverse (computed once per workgroup, one lane per column), so
// no division sits on the per-row dependency chain.  Columns at or beyond a
// failed pivot are written as zero.
#define TR_ROWS 64
#define TR_CW 8
__global__ void __launch_bounds__(TR_ROWS) k_trsm (const TrGroup *g, int ng,
    double *Lx, const i32 *info, int ldl)
{
    // dynamic LDS, sized by the widest block of the launch (ldl = its width
    // rounded up to TR_CW): levels with thousands of narrow panels then fit many
    // workgroups per CU instead of two
    extern __shared__ __attribute__((aligned(16))) double trsm_lds [] ;
    double *Lt = trsm_lds ;                         // Lt[k][j] = L11(j,k), ld = ldl
    double *xs = Lt + ldl * ldl ;                   // xs[k][t], ld = TR_ROWS
    double *Wi = xs + ldl * TR_ROWS ;               // inverse diagonal blocks
    __builtin_amdgcn_s_setprio (3) ;
    int gi = find_group (g, ng, (int) blockIdx.x, &TrGroup::blk_start) ;
    TrGroup G = g [gi] ;
    int nb = G.nb, lda = G.lda ;
    int nbp = (nb + TR_CW - 1) / TR_CW * TR_CW ;
    const double *L11 = Lx + G.l_off ;
    int inf = info [G.front] ;
    int nvalid = nb ;
    if (inf != 0)
    {
        nvalid = inf - 1 - G.col0 ;
        if (nvalid < 0) nvalid = 0 ;
        if (nvalid > nb) nvalid = nb ;
    }
    int t = threadIdx.x ;
    // stage L11', identity-padded to a multiple of TR_CW and beyond a failed
    // pivot; lane j owns row j of L11, 8 independent column loads in flight
    {
        int j = t ;
        int jc = j < nb ? j : nb - 1 ;
        for (int k0 = 0 ; k0 < nbp ; k0 += 8)
        {
            double tmp [8] ;
#pragma unroll
            for (int c = 0 ; c < 8 ; c++)
            {
                int k = k0 + c < nb ? k0 + c : nb - 1 ;
                tmp [c] = L11 [jc + (i64) k * lda] ;
            }
#pragma unroll
            for (int c = 0 ; c < 8 ; c++)
            {
                int k = k0 + c ;
                double v = (j == k) ? 1.0 : 0.0 ;
                if (j < nvalid && k <= j) v = tmp [c] ;
                if (j < nbp) Lt [k * ldl + j] = v ;
            }
        }
    }
    int row = ((int) blockIdx.x - G.blk_start) * TR_ROWS + t ;
    bool active = row < G.m ;
    double *B = Lx + G.b_off + (active ? row : 0) ;
    // prefetch the whole row of B into LDS (nb independent loads in flight)
    for (int j0 = 0 ; j0 < nbp ; j0 += 8)
    {
        double tmp [8] ;
#pragma unroll
        for (int c = 0 ; c < 8 ; c++)
        {
            int j = j0 + c < nb ? j0 + c : nb - 1 ;
            tmp [c] = B [(i64) j * lda] ;
        }
#pragma unroll
        for (int c = 0 ; c < 8 ; c++) xs [(j0 + c) * TR_ROWS + t] = (j0 + c < nb) ? tmp [c] : 0.0 ;
    }
    __syncthreads () ;
    // inverse of every TR_CW x TR_CW diagonal block: lane (b,q) computes column
    // q of inv(L_bb) by forward substitution in registers
    {
        int b = t >> 3, q = t & 7 ;
        if (b * TR_CW < nbp)
        {
            double y [TR_CW] ;
#pragma unroll
            for (int r = 0 ; r < TR_CW ; r++)
            {
                double acc = (r == q) ? 1.0 : 0.0 ;
#pragma unroll
                for (int e = 0 ; e < r ; e++)
                    acc -= Lt [(b * TR_CW + e) * ldl + b * TR_CW + r] * y [e] ;
                double v = acc / Lt [(b * TR_CW + r) * ldl + b * TR_CW + r] ;
                y [r] = (r >= q) ? v : 0.0 ;
            }
#pragma unroll
            for (int r = 0 ; r < TR_CW ; r++) Wi [(b * TR_CW + r) * TR_CW + q] = y [r] ;
        }
    }
    __syncthreads () ;
    if (!active) return ;
    for (int jb = 0 ; jb < nbp ; jb += TR_CW)
    {
        double xb [TR_CW] ;
#pragma unroll
        for (int c = 0 ; c < TR_CW ; c++) xb [c] = xs [(jb + c) * TR_ROWS + t] ;
        for (int kg = 0 ; kg < jb ; kg += TR_CW)
        {
            double xk [TR_CW] ;
#pragma unroll
            for (int d = 0 ; d < TR_CW ; d++) xk [d] = xs [(kg + d) * TR_ROWS + t] ;
#pragma unroll
            for (int d = 0 ; d < TR_CW ; d++)
            {
                const double *lt = Lt + (kg + d) * ldl + jb ;
#pragma unroll
                for (int c = 0 ; c < TR_CW ; c++) xb [c] -= xk [d] * lt [c] ;
            }
        }
        // x_c = sum_{e<=c} v_e * inv(L_JJ)(c,e)
        const double *w = Wi + jb * TR_CW ;
        double xo [TR_CW] ;
#pragma unroll
        for (int c = 0 ; c < TR_CW ; c++)
        {
            double v = 0.0 ;
#pragma unroll
            for (int e = 0 ; e <= c ; e++) v += xb [e] * w [c * TR_CW + e] ;
            xo [c] = (jb + c < nvalid) ? v : 0.0 ;
        }
#pragma unroll
        for (int c = 0 ; c < TR_CW ; c++)
        {
            xs [(jb + c) * TR_ROWS + t] = xo [c] ;
            if (jb + c < nb) B [(i64) (jb + c) * lda] = xo [c] ;
        }
    }
}


template <int BM, int BN, int BK, bool USE_MFMA>
__global__ void __launch_bounds__(256) k_update (const GemmGroup *g, int ng,
    double *Lx, double *CB)
{
    constexpr int LDT = BM + 16 ;           // k-major LDS row stride (A), see above
    constexpr int LDU = BN + 16 ;
    constexpr int WM = BM / 2, WN = BN / 2 ; // per-wave tile
    constexpr int TI = WM / 16, TJ = WN / 16 ;
    constexpr int NA = BM * BK / 256, NB_ = BN * BK / 256 ;
    __shared__ double As [BK * LDT] ;
    __shared__ double Bs [BK * LDU] ;

    int gi = find_group (g, ng, (int) blockIdx.x, &GemmGroup::tile_start) ;
    GemmGroup G = g [gi] ;
    int I, J ;
    if ((int) blockIdx.x - G.tile_start >= G.nblk) return ;
    if (!decode_tile (G, (int) blockIdx.x - G.tile_start, I, J)) return ;
    int row0 = I * BM, col0 = J * BN ;
    int mrem = G.m - row0, nrem = G.n - col0 ;      // valid rows / cols in tile
    const double *A = Lx + G.a_off + row0 ;
    const double *B = Lx + G.b_off + col0 ;
    double *C = (G.c_in_cb ? CB : Lx) + G.c_off + row0 + (i64) col0 * G.ldc ;
    int lda = G.lda, K = G.k ;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6 ;
    int wm = wave & 1, wn = wave >> 1 ;

    double ra [NA], rb [NB_] ;
    auto gload = [&] (int k0)
    {
#pragma unroll
        for (int q = 0 ; q < NA ; q++)
        {
            int idx = tid + 256 * q ;
            int i = idx % BM, k = idx / BM ;
            ra [q] = (i < mrem && k0 + k < K) ? A [i + (i64) (k0 + k) * lda] : 0.0 ;
        }
#pragma unroll
        for (int q = 0 ; q < NB_ ; q++)
        {
            int idx = tid + 256 * q ;
            int j = idx % BN, k = idx / BN ;
            rb [q] = (j < nrem && k0 + k < K) ? B [j + (i64) (k0 + k) * lda] : 0.0 ;
        }
    } ;
    auto lstore = [&] ()
    {
#pragma unroll
        for (int q = 0 ; q < NA ; q++)
        {
            int idx = tid + 256 * q ;
            As [(idx / BM) * LDT + (idx % BM)] = ra [q] ;
        }
#pragma unroll
        for (int q = 0 ; q < NB_ ; q++)
        {
            int idx = tid + 256 * q ;
            Bs [(idx / BN) * LDU + (idx % BN)] = rb [q] ;
        }
    } ;

    if constexpr (USE_MFMA)
    {
        d4 acc [TI][TJ] ;
#pragma unroll
        for (int a = 0 ; a < TI ; a++)
#pragma unroll
            for (int b = 0 ; b < TJ ; b++) acc [a][b] = (d4) {0.0, 0.0, 0.0, 0.0} ;
        gload (0) ;
        for (int k0 = 0 ; k0 < K ; k0 += BK)
        {
            __syncthreads () ;
            lstore () ;
            __syncthreads () ;
            if (k0 + BK < K) gload (k0 + BK) ;
#pragma unroll
            for (int kk = 0 ; kk < BK ; kk += 4)
            {
                double af [TI], bf [TJ] ;
                int kr = kk + (lane >> 4) ;
#pragma unroll
                for (int a = 0 ; a < TI ; a++)
                    af [a] = As [kr * LDT + wm * WM + a * 16 + (lane & 15)] ;
#pragma unroll
                for (int b = 0 ; b < TJ ; b++)
                    bf [b] = Bs [kr * LDU + wn * WN + b * 16 + (lane & 15)] ;
#pragma unroll
                for (int a = 0 ; a < TI ; a++)
#pragma unroll
                    for (int b = 0 ; b < TJ ; b++)
                        acc [a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64 (
                            bf [b], af [a], acc [a][b], 0, 0, 0) ;
            }
        }
        // lane holds C(i, j) with i = .. + (lane&15), j = .. + (lane>>4) + 4r
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
                        C [i + (i64) j * G.ldc] -= acc [a][b][r] ;
                }
    }
    else
    {
        // VALU reference path (debug / cross-check of the MFMA operand maps):
        // thread (ti,tj) of a 16x16 grid owns a (BM/16) x (BN/16) sub-tile.
        constexpr int RM = BM / 16, RN = BN / 16 ;
        double acc [RM][RN] ;
#pragma unroll
        for (int a = 0 ; a < RM ; a++)
#pragma unroll
            for (int b = 0 ; b < RN ; b++) acc [a][b] = 0.0 ;
        int ti = tid & 15, tj = tid >> 4 ;
        gload (0) ;
        for (int k0 = 0 ; k0 < K ; k0 += BK)
        {
            __syncthreads () ;
            lstore () ;
            __syncthreads () ;
            if (k0 + BK < K) gload (k0 + BK) ;
#pragma unroll
            for (int kk = 0 ; kk < BK ; kk++)
            {
                double af [RM], bf [RN] ;
#pragma unroll
                for (int a = 0 ; a < RM ; a++) af [a] = As [kk * LDT + ti + 16 * a] ;
#pragma unroll
                for (int b = 0 ; b < RN ; b++) bf [b] = Bs [kk * LDU + tj + 16 * b] ;
#pragma unroll
                for (int a = 0 ; a < RM ; a++)
#pragma unroll
                    for (int b = 0 ; b < RN ; b++) acc [a][b] += af [a] * bf [b] ;
            }
        }
#pragma unroll
        for (int a = 0 ; a < RM ; a++)
#pragma unroll
            for (int b = 0 ; b < RN ; b++)
            {
                int i = ti + 16 * a, j = tj + 16 * b ;
                if (i < mrem && j < nrem && (!G.tri || row0 + i >= col0 + j))
                    C [i + (i64) j * G.ldc] -= acc [a][b] ;
            }
    }
}

// issue-bound v_fma_f64 loop (register only): the fp64 VALU ceiling
template <int NACC>
__global__ void __launch_bounds__(256) k_valu_peak (double *out, int iters)
{
    double acc [NACC] ;
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * threadIdx.x ;
#pragma unroll
    for (int q = 0 ; q < NACC ; q++) acc [q] = q ;
    for (int it = 0 ; it < iters ; it++)
    {
#pragma unroll
        for (int q = 0 ; q < NACC ; q++) acc [q] = __builtin_fma (acc [q], a, b) ;
    }
    double sum = 0 ;
#pragma unroll
    for (int q = 0 ; q < NACC ; q++) sum += acc [q] ;
    out [blockIdx.x * 256 + threadIdx.x] = sum ;
}

// mixed issue test: waves 0,1 of a block run the MFMA loop, waves 2,3 the VALU
// loop -- do the fp64 matrix and vector pipes overlap on gfx950?
__global__ void __launch_bounds__(256) k_mixed_peak (double *out, int it_mfma, int it_valu)
{
    int wave = threadIdx.x >> 6 ;
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * threadIdx.x ;
    double sum = 0 ;
    if (wave < 2)
    {
        d4 acc [8] ;
#pragma unroll
        for (int q = 0 ; q < 8 ; q++) acc [q] = (d4) {0.0, 0.0, 0.0, 0.0} ;
        for (int it = 0 ; it < it_mfma ; it++)
        {
#pragma unroll
            for (int q = 0 ; q < 8 ; q++)
                acc [q] = __builtin_amdgcn_mfma_f64_16x16x4f64 (a, b, acc [q], 0, 0, 0) ;
        }
#pragma unroll
        for (int q = 0 ; q < 8 ; q++) sum += acc [q][0] + acc [q][1] + acc [q][2] + acc [q][3] ;
    }
    else
    {
        double acc [16] ;
#pragma unroll
        for (int q = 0 ; q < 16 ; q++) acc [q] = q ;
        for (int it = 0 ; it < it_valu ; it++)
        {
#pragma unroll
            for (int q = 0 ; q < 16 ; q++) acc [q] = __builtin_fma (acc [q], a, b) ;
        }
#pragma unroll
        for (int q = 0 ; q < 16 ; q++) sum += acc [q] ;
    }
    out [blockIdx.x * 256 + threadIdx.x] = sum ;
}


// ---- micro-benchmark: issue-bound v_mfma_f64_16x16x4_f64 loop (no memory) ----
// Measures the fp64 matrix-core ceiling that the roofline is priced against
// (spec 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 2048 flop / 64 cycles x 2.4 GHz).
// FILL: what sits between two MFMAs -- 0 nothing, 1 s_nop 3, 2 one independent
// v_fma_f32, 3 one LDS read (the pattern of a real kernel's operand fetch)
template <int NACC, int FILL = 0>
__global__ void __launch_bounds__(256) k_mfma_peak (double *out, int iters, double scale)
{
    __shared__ double lds_fill [256] ;
    lds_fill [threadIdx.x] = 1.0 ;
    float ff = threadIdx.x ;
    double lacc = 0.0 ;
    d4 acc [NACC] ;
    double a [4], b [4] ;
#pragma unroll
    // scale = 0: all-zero operands (no toggling in the multipliers: the issue rate
    // without the power the data costs); scale = 1: full-mantissa operands
    for (int q = 0 ; q < 4 ; q++) { a [q] = scale * (1.0 + 1e-9 * (threadIdx.x + q)) ; b [q] = scale * (1.0 - 1e-9 * (threadIdx.x + 3 * q)) ; }
#pragma unroll
    for (int q = 0 ; q < NACC ; q++) acc [q] = (d4) {0.0, 0.0, 0.0, 0.0} ;
    for (int it = 0 ; it < iters ; it++)
    {
#pragma unroll
        for (int q = 0 ; q < NACC ; q++)
        {
            acc [q] = __builtin_amdgcn_mfma_f64_16x16x4f64 (a [q & 3], b [(q >> 2) & 3], acc [q], 0, 0, 0) ;
            if constexpr (FILL == 1) asm volatile ("s_nop 3") ;
            if constexpr (FILL == 2) { ff = __builtin_fmaf (ff, 1.0001f, 0.5f) ; asm volatile ("" : "+v" (ff)) ; }
            if constexpr (FILL == 3) { lacc += lds_fill [(threadIdx.x + q + it) & 255] ; }
        }
    }
    double sum = lacc + ff ;
#pragma unroll
    for (int q = 0 ; q < NACC ; q++) sum += acc [q][0] + acc [q][1] + acc [q][2] + acc [q][3] ;
    out [blockIdx.x * 256 + threadIdx.x] = sum ;
}

// second issue-loop family: the operand pattern of the update kernel -- a TI x TJ
// grid of accumulators per wave, A fragment a[i] shared along a row, B fragment
// b[j] along a column, fragments refreshed from LDS every k-step (LDSREAD) or kept
template <int TI, int TJ, bool LDSREAD>
__global__ void __launch_bounds__(256) k_mfma_peak2 (double *out, int iters, double scale)
{
    __shared__ double frag [2][4][128] ;
    for (int e = threadIdx.x ; e < 2 * 4 * 128 ; e += 256) (&frag [0][0][0]) [e] = scale * (1.0 + 1e-9 * e) ;
    __syncthreads () ;
    d4 acc [TI][TJ] ;
#pragma unroll
    for (int i = 0 ; i < TI ; i++)
#pragma unroll
        for (int j = 0 ; j < TJ ; j++) acc [i][j] = (d4) {0.0, 0.0, 0.0, 0.0} ;
    double a [TI], b [TJ] ;
    int lane = threadIdx.x & 63 ;
#pragma unroll
    for (int i = 0 ; i < TI ; i++) a [i] = frag [0][i & 3][lane] ;
#pragma unroll
    for (int j = 0 ; j < TJ ; j++) b [j] = frag [1][j & 3][lane] ;
    for (int it = 0 ; it < iters ; it++)
    {
        if constexpr (LDSREAD)
        {
#pragma unroll
            for (int i = 0 ; i < TI ; i++) a [i] = frag [0][i & 3][(lane + it) & 127] ;
#pragma unroll
            for (int j = 0 ; j < TJ ; j++) b [j] = frag [1][j & 3][(lane + it) & 127] ;
        }
#pragma unroll
        for (int i = 0 ; i < TI ; i++)
#pragma unroll
            for (int j = 0 ; j < TJ ; j++)
                acc [i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64 (b [j], a [i], acc [i][j], 0, 0, 0) ;
    }
    double sum = 0 ;
#pragma unroll
    for (int i = 0 ; i < TI ; i++)
#pragma unroll
        for (int j = 0 ; j < TJ ; j++) sum += acc [i][j][0] + acc [i][j][1] + acc [i][j][2] + acc [i][j][3] ;
    out [blockIdx.x * 256 + threadIdx.x] = sum ;
}


// ---- the fp64 matrix-core ceiling, third attempt (round 3) ---------------------------------
// k_mfma_peak above is mis-built: hipcc keeps its accumulators in VGPRs across the loop edge and
// in AGPRs inside it, i.e. 64 v_accvgpr_write + 64 v_accvgpr_read per 8 MFMAs (seen in the ISA);
// it measures those copies (~104 cycles per MFMA), not the pipe.  Here the loop body is inline
// assembly: NACC independent accumulators (>= 4: an accumulator is reused three MFMAs = 192
// cycles later, no dependent-issue hazard), 4 x NACC MFMAs per iteration, operands from four
// A / four B registers with distinct full-mantissa data, nothing else in the loop but the
// scalar counter.  stamp [2b], [2b+1] = shader cycles (s_memtime) and 100 MHz wall ticks
// (s_memrealtime) wave 0 of block b spent in the loop: cycles / ticks x 100 MHz = the clock
// the chip sustained under this load, cycles / MFMAs issued by that wave's SIMD = issue rate.
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_ceiling (double *out, long long *stamp, int iters, double scale)
{
    static_assert (NACC >= 4 && NACC % 4 == 0, "independent accumulators") ;
    d4 acc [NACC] ;
    double a [4], b [4] ;
#pragma unroll
    for (int q = 0 ; q < 4 ; q++)
    {
        unsigned long long h = (threadIdx.x + 64 * q + 1) * 0x9E3779B97F4A7C15ull ;
        a [q] = scale * (1.0 + (double) (h >> 12) * 0x1p-52) ;
        h = (h ^ (h >> 29)) * 0xBF58476D1CE4E5B9ull ;
        b [q] = scale * (1.0 - (double) (h >> 12) * 0x1p-53) ;
    }
#pragma unroll
    for (int q = 0 ; q < NACC ; q++) acc [q] = (d4) {0.0, 0.0, 0.0, 0.0} ;
    __syncthreads () ;
    long long c0 = __builtin_readcyclecounter (), w0 = wall_clock64 () ;
    for (int it = 0 ; it < iters ; it++)
    {
#pragma unroll
        for (int u = 0 ; u < 4 ; u++)
#pragma unroll
            for (int q = 0 ; q < NACC ; q++)
                asm volatile ("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v" (acc [q]) : "v" (a [(q + u) & 3]), "v" (b [q & 3])) ;
    }
    long long c1 = __builtin_readcyclecounter (), w1 = wall_clock64 () ;
    double sum = 0 ;
#pragma unroll
    for (int q = 0 ; q < NACC ; q++) sum += acc [q][0] + acc [q][1] + acc [q][2] + acc [q][3] ;
    out [blockIdx.x * 256 + threadIdx.x] = sum ;
    if (threadIdx.x == 0) { stamp [2 * blockIdx.x] = c1 - c0 ; stamp [2 * blockIdx.x + 1] = w1 - w0 ; }
}


// Hand-off latency between two workgroups (round 6): block `prod` writes a payload of `ndbl` doubles and raises a flag,
// block `cons` waits for the flag, reads the payload, checks it and raises the flag back; `rounds` ping-pongs, wall clock
// per one-way hand-off out of thread 0 of the producer.  Every other block of the launch leaves at once, so that with
// block -> XCD = blockIdx % 8 (tools/xcd_map.py) the pair (0, 8) shares an XCD's L2 and the pair (0, 1) does not.
// MODE 0: payload and flags by relaxed AGENT-scope atomics (global_store / global_load sc1: what k_chainf does);
// MODE 1: payload by WORKGROUP-scope atomics (sc0: past the L1, served by the XCD's L2), flags agent scope;
// MODE 2: payload by plain stores / loads behind an L1 invalidate (buffer_inv sc0... emitted by a workgroup-scope acquire
//         fence in threadgroup-split-safe code), flags workgroup scope -- only meaningful inside one XCD.
template <int MODE>
__global__ void __launch_bounds__(256) k_handoff (double *payload, int *flags, int prod, int cons, int ndbl, int rounds,
    long long *ticks, int *bad)
{
    const int b = (int) blockIdx.x, tid = threadIdx.x ;
    if (b != prod && b != cons) return ;
    __shared__ int s_v ;
    int errors = 0 ;
    long long w0 = 0 ;
    if (b == prod && tid == 0) w0 = wall_clock64 () ;
    for (int r = 1 ; r <= rounds ; r++)
    {
        // (a wait that ran out ends the probe for both sides: flags [48], agent scope; no second full wait)
        if (tid == 0) s_v = (errors & 2) ? 1 : __hip_atomic_load (flags + 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
        __syncthreads () ;
        if (s_v) break ;
        __syncthreads () ;
        if (b == prod)
        {
            for (int e = tid ; e < ndbl ; e += 256)
            {
                const double v = (double) (r * 4096 + e) ;
                if (MODE == 0) __hip_atomic_store (payload + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
                else if (MODE == 1) __hip_atomic_store (payload + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ;
                else payload [e] = v ;
            }
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory") ;
            __syncthreads () ;
            if (tid == 0)
            {
                if (MODE == 2) __hip_atomic_store (flags, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP) ;
                else __hip_atomic_store (flags, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
                // wait for the echo
                int n = 0 ;
                while ((MODE == 2 ? __hip_atomic_load (flags + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                  : __hip_atomic_load (flags + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < r)
                    if (++n > (1 << 18)) { errors |= 2 ; break ; }
            }
            __syncthreads () ;
        }
        else
        {
            if (tid == 0)
            {
                int n = 0 ;
                while ((MODE == 2 ? __hip_atomic_load (flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                  : __hip_atomic_load (flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < r)
                    if (++n > (1 << 18)) { errors |= 2 ; break ; }
            }
            __syncthreads () ;
            if (MODE == 2) __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup") ;
            for (int e = tid ; e < ndbl ; e += 256)
            {
                double v ;
                if (MODE == 0) v = __hip_atomic_load (payload + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
                else if (MODE == 1) v = __hip_atomic_load (payload + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ;
                else v = ((volatile double *) payload) [e] ;
                if (v != (double) (r * 4096 + e)) errors |= 1 ;
            }
            __syncthreads () ;
            if (tid == 0)
            {
                if (MODE == 2) __hip_atomic_store (flags + 32, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP) ;
                else __hip_atomic_store (flags + 32, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
            }
        }
    }
    if (b == prod && tid == 0) ticks [0] = wall_clock64 () - w0 ;
    if (errors & 2) __hip_atomic_store (flags + 48, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
    if (errors) atomicOr (bad, errors) ;
}

// full-mantissa pseudo-random fill in [-0.5, 0.5) (splitmix64 of the index)
__global__ void __launch_bounds__(256) k_fill_random (double *x, i64 n)
{
    for (i64 e = blockIdx.x * (i64) 256 + threadIdx.x ; e < n ; e += (i64) gridDim.x * 256)
    {
        unsigned long long z = (unsigned long long) e * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull ;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull ; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull ; z ^= z >> 31 ;
        x [e] = (double) (z >> 11) * 0x1p-53 - 0.5 ;
    }
}

} // namespace sship
