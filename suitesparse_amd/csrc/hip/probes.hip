// probes.hip -- micro-benchmarks and tuning probes of the engine's kernels
// (tools/, bench.py's measured ceilings).  Built into lib/libcholmod_amd_probes.so,
// NOT into the product library.
#include "probe_kernels.hip.h"
#include "../../../include/cholmod_hip.h"
#include "../../../include/cholmod_hip_probes.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace sship ;

namespace {
constexpr int BIG = 128, SMALL = 64, BKK = 16 ;
#define HIPCHK(call) do { hipError_t e_ = (call) ; if (e_ != hipSuccess) { \
    fprintf (stderr, "cholmod_hip probes: %s failed: %s (%s:%d)\n", #call, \
        hipGetErrorString (e_), __FILE__, __LINE__) ; return CHOLMOD_HIP_GPU_PROBLEM ; } } while (0)
static int probe_device ()
{
    int cnt = 0 ;
    return (hipGetDeviceCount (&cnt) == hipSuccess && cnt > 0) ? 1 : 0 ;
}
static int raise_lds_limits ()
{
    HIPCHK (hipFuncSetAttribute ((const void *) k_trsm_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_trsm_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) ;
    return CHOLMOD_HIP_OK ;
}
}

extern "C" {

double cholmod_hip_bench_update_kernel (int64_t m, int64_t n, int64_t k, int iters, int flags)
{
    if (m <= 0 || n <= 0 || k <= 0 || iters <= 0) return CHOLMOD_HIP_INVALID ;
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    // A: m x k, B: n x k (both ld = max(m,n)), C: m x n, all in one "Lx" buffer
    // flag 131072: the layout of a real front -- an odd leading dimension and operand offsets that are
    // only 8-byte aligned (packed supernodes start anywhere, nsrow is any number)
    const bool odd = (flags & 131072) != 0 ;
    // CHOLMOD_PROBE_ODD_PARTS (with flag 131072): 1 = only the operands A / B in the odd layout, 2 = only the target C, 3 = both (default)
    int parts = 3 ;
    if (const char *e = getenv ("CHOLMOD_PROBE_ODD_PARTS")) parts = atoi (e) & 3 ;
    const bool odd_ab = odd && (parts & 1), odd_c = odd && (parts & 2) ;
    i64 ld = std::max (m, n) + (odd_ab ? 1 + ((std::max (m, n) + 1) % 2 == 0 ? 1 : 0) : 0) ;
    i64 a_off = odd_ab ? 1 : 0, b_off = a_off + ld * k + (odd_ab ? 2 : 0), c_off = b_off + ld * k ;
    c_off = (c_off + 15) / 16 * 16 + (odd_c ? 1 : 0) ;
    i64 total = c_off + (m + 3) * n ;
    double *d = nullptr ;
    // CHOLMOD_PROBE_OFFSET_GB=g: the operands sit g GB into ONE allocation of g GB + their own size (as the
    // top fronts of a big factor sit deep inside the one Lx allocation): does the rate depend on it?
    i64 skip = 0 ;
    if (const char *e = getenv ("CHOLMOD_PROBE_OFFSET_GB")) skip = (i64) (atof (e) * 1e9 / 8.0) ;
    if (hipMalloc ((void **) &d, (total + skip + UPD3_LX_PAD) * sizeof (double)) != hipSuccess) return CHOLMOD_HIP_OUT_OF_MEMORY ;    // (partial tiles of k_update3 read up to 63 rows past an operand, kernels.hip.h)
    double *d_base = d ;
    d += skip ;
    // operands filled on the device (full-mantissa pseudo-random values in [-0.5, 0.5)): regions of
    // the size of the factorization's top fronts (tens of GB) would take minutes through the host
    hipLaunchKernelGGL (k_fill_random, dim3 (4096), dim3 (256), 0, 0, d, total) ;
    if (hipDeviceSynchronize () != hipSuccess) { (void) hipFree (d_base) ; return CHOLMOD_HIP_GPU_PROBLEM ; }
    bool small = (flags & CHOLMOD_HIP_TILE128) == 0 ;
    int T = small ? SMALL : BIG ;
    GemmGroup G ;
    memset (&G, 0, sizeof (G)) ;
    G.a_off = a_off ; G.b_off = b_off ; G.c_off = c_off ; G.lda = (i32) ld ; G.ldc = (i32) (m + (odd_c ? 3 : 0)) ;
    G.m = (i32) m ; G.n = (i32) n ; G.k = (i32) k ; G.tri = (flags & 65536) ? 1 : 0 ; G.tile_mul = 1 ; G.tile_add = 0 ;
    if (G.tri) { G.b_off = a_off ; if (m < n) { (void) hipFree (d_base) ; return CHOLMOD_HIP_INVALID ; } }    // a syrk-shaped region: B = A, only tiles on / below the diagonal
    int TM = T, TN = T ;
    if (flags & 2048) { TM = 128 ; TN = 64 ; }      // experimental rectangular tiles (non-tri only)
    if (flags & 4096) { TM = 64 ; TN = 128 ; }
    {
        i64 mt_ = (m + TM - 1) / TM, nt_ = (n + TN - 1) / TN ;
        G.ntiles = (i32) (G.tri ? nt_ * (nt_ + 1) / 2 + (mt_ - nt_) * nt_ : mt_ * nt_) ;
        G.nblk = (G.ntiles + 63) / 64 * 64 ;
    }
    G.swz = (flags & CHOLMOD_HIP_NO_XCD_SWIZZLE) ? 0 : 1 ;
    // flag 262144: 16-wide strips, an XCD walks 16 x 16 super-tiles (256 tiles = what it runs at a time)
    if ((flags & 262144) && G.swz) { G.swz = 2 ; G.nblk = (G.ntiles + 255) / 256 * 256 ; }
    G.mt = (i32) ((m + TM - 1) / TM) ; G.nt = (i32) ((n + TN - 1) / TN) ;
    GemmGroup *dg = nullptr ;
    (void) hipMalloc ((void **) &dg, sizeof (G)) ;
    (void) hipMemcpy (dg, &G, sizeof (G), hipMemcpyHostToDevice) ;
    int grid = G.nblk ;
    hipEvent_t e0, e1 ;
    (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ;
    auto launch = [&] ()
    {
        if (flags & 524288)     // (round 5) half tiles: two waves per 64 x 64 tile, 64 x 32 each; depth 4 (flag 32768) or 2
        {
            const unsigned g2 = 2u * (unsigned) ((grid + 7) / 8 * 8) ;
            if (flags & 32768) hipLaunchKernelGGL ((k_update3<4, 0, 1, true>), dim3 (g2), dim3 (64), 0, 0, dg, 1, d, d) ;
            else hipLaunchKernelGGL ((k_update3<2, 0, 1, true>), dim3 (g2), dim3 (64), 0, 0, dg, 1, d, d) ;
        }
        else if (flags & 8192)       // third generation: one wave per tile, no LDS (operand sets in flight: 3 / 2 / 4)
            hipLaunchKernelGGL ((k_update3<3>), dim3 (grid), dim3 (64), 0, 0, dg, 1, d, d) ;
        else if (flags & 16384)
            hipLaunchKernelGGL ((k_update3<2>), dim3 (grid), dim3 (64), 0, 0, dg, 1, d, d) ;
        else if (flags & 32768)
            hipLaunchKernelGGL ((k_update3<4>), dim3 (grid), dim3 (64), 0, 0, dg, 1, d, d) ;
        else if (flags & 1)      // the first-generation VALU kernel (cross-check)
            hipLaunchKernelGGL ((k_update<SMALL, SMALL, BKK, false>), dim3 (grid), dim3 (256), 0, 0, dg, 1, d, d) ;
        else if (small && (flags & 256))
            hipLaunchKernelGGL ((k_update2<SMALL, SMALL, 32, 2, false>), dim3 (grid), dim3 (256), 0, 0, dg, 1, d, d) ;
        else if (small && (flags & 512))
            hipLaunchKernelGGL ((k_update2<SMALL, SMALL, 16, 3, false>), dim3 (grid), dim3 (256), 0, 0, dg, 1, d, d) ;
        else if (small && (flags & 1024))
            hipLaunchKernelGGL ((k_update2<SMALL, SMALL, 16, 2, true>), dim3 (grid), dim3 (256), 0, 0, dg, 1, d, d) ;
        else if (flags & 2048)
            hipLaunchKernelGGL ((k_update2<128, 64, 16, 2, false>), dim3 (grid), dim3 (256), 0, 0, dg, 1, d, d) ;
        else if (flags & 4096)
            hipLaunchKernelGGL ((k_update2<64, 128, 16, 2, false>), dim3 (grid), dim3 (256), 0, 0, dg, 1, d, d) ;
        else if (small)
            hipLaunchKernelGGL ((k_update2<SMALL, SMALL, BKK, 2, false>), dim3 (grid), dim3 (256), 0, 0, dg, 1, d, d) ;
        else
            hipLaunchKernelGGL ((k_update2<BIG, BIG, BKK, 2, false>), dim3 (grid), dim3 (256), 0, 0, dg, 1, d, d) ;
    } ;
    launch () ;
    (void) hipDeviceSynchronize () ;
    (void) hipEventRecord (e0, 0) ;
    for (int it = 0 ; it < iters ; it++) launch () ;
    (void) hipEventRecord (e1, 0) ;
    (void) hipEventSynchronize (e1) ;
    float ms = 0 ;
    (void) hipEventElapsedTime (&ms, e0, e1) ;
    hipError_t err = hipGetLastError () ;
    (void) hipFree (d_base) ; (void) hipFree (dg) ;
    (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ;
    if (err != hipSuccess || ms <= 0) return CHOLMOD_HIP_GPU_PROBLEM ;
    double entries = G.tri ? (double) n * (n + 1) / 2 + (double) (m - n) * n : (double) m * n ;
    return 2.0 * entries * k * iters / (ms * 1e-3) ;
}

/* (round 5) The outer update of a top front as the engine issues it: the trapezoid of the in-front columns right of the outer
 * block (rows [0, m1) x columns [0, n1) of the panel rows, triangular) and the square of the contribution block (the last m2
 * rows, triangular), both contracting the same K columns of ONE panel of leading dimension ld >= m1.
 * mode 0: one launch, trapezoid first (the engine until the end of round 5); 1: two launches; 2: one launch, square first;
 * 3: one launch, the square's blocks padded to start at a multiple of 2048 * 8 blocks.  Returns flop/s over both regions. */
double cholmod_hip_bench_update_pair (int64_t m1, int64_t n1, int64_t m2, int64_t k, int64_t ld, int iters, int mode)
{
    if (m1 <= 0 || n1 <= 0 || m2 <= 0 || m2 > m1 || n1 > m1 || k <= 0 || ld < m1 || iters <= 0) return CHOLMOD_HIP_INVALID ;
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    const i64 a_off = 1 ;                                       // (8-byte aligned only, as a packed front)
    const i64 c1_off = a_off + ld * k + 1 ;                     // C of the trapezoid: m1 x n1, ldc = ld (it lives in the front)
    const i64 c2_off = c1_off + ld * n1 + 1 ;                   // C of the square: m2 x m2
    const i64 total = c2_off + m2 * m2 ;
    double *d = nullptr ;
    if (hipMalloc ((void **) &d, (total + UPD3_LX_PAD) * sizeof (double)) != hipSuccess) return CHOLMOD_HIP_OUT_OF_MEMORY ;
    hipLaunchKernelGGL (k_fill_random, dim3 (4096), dim3 (256), 0, 0, d, total) ;
    if (hipDeviceSynchronize () != hipSuccess) { (void) hipFree (d) ; return CHOLMOD_HIP_GPU_PROBLEM ; }
    auto region = [&] (i64 rows0, i64 m, i64 n, i64 c_off, i64 ldc)
    {
        GemmGroup G ;
        memset (&G, 0, sizeof (G)) ;
        G.a_off = a_off + rows0 ; G.b_off = G.a_off ; G.c_off = c_off ; G.lda = (i32) ld ; G.ldc = (i32) ldc ;
        G.m = (i32) m ; G.n = (i32) n ; G.k = (i32) k ; G.tri = 1 ; G.tile_mul = 1 ;
        G.mt = (i32) ((m + 63) / 64) ; G.nt = (i32) ((n + 63) / 64) ;
        G.ntiles = (i32) ((i64) G.nt * (G.nt + 1) / 2 + (i64) (G.mt - G.nt) * G.nt) ;
        G.nblk = (G.ntiles + 63) / 64 * 64 ;
        G.swz = 1 ;
        return G ;
    } ;
    GemmGroup T = region (0, m1, n1, c1_off, ld), Q = region (m1 - m2, m2, m2, c2_off, m2) ;
    GemmGroup two [2] ;
    if (mode == 2) { two [0] = Q ; two [1] = T ; } else { two [0] = T ; two [1] = Q ; }
    two [0].tile_start = 0 ;
    two [1].tile_start = (two [0].nblk + 7) / 8 * 8 ;
    if (mode == 3) two [1].tile_start = (two [0].nblk + 16383) / 16384 * 16384 ;
    const int grid2 = two [1].tile_start + two [1].nblk ;
    GemmGroup sep [2] = {T, Q} ;
    sep [0].tile_start = sep [1].tile_start = 0 ;
    GemmGroup *dg = nullptr ;
    (void) hipMalloc ((void **) &dg, 4 * sizeof (GemmGroup)) ;
    (void) hipMemcpy (dg, two, 2 * sizeof (GemmGroup), hipMemcpyHostToDevice) ;
    (void) hipMemcpy (dg + 2, sep, 2 * sizeof (GemmGroup), hipMemcpyHostToDevice) ;
    auto launch = [&] ()
    {
        if (mode == 1)
        {
            hipLaunchKernelGGL ((k_update3<4>), dim3 (sep [0].nblk), dim3 (64), 0, 0, dg + 2, 1, d, d) ;
            hipLaunchKernelGGL ((k_update3<4>), dim3 (sep [1].nblk), dim3 (64), 0, 0, dg + 3, 1, d, d) ;
        }
        else hipLaunchKernelGGL ((k_update3<4>), dim3 (grid2), dim3 (64), 0, 0, dg, 2, d, d) ;
    } ;
    hipEvent_t e0, e1 ;
    (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ;
    launch () ;
    (void) hipDeviceSynchronize () ;
    (void) hipEventRecord (e0, 0) ;
    for (int it = 0 ; it < iters ; it++) launch () ;
    (void) hipEventRecord (e1, 0) ;
    (void) hipEventSynchronize (e1) ;
    float ms = 0 ;
    (void) hipEventElapsedTime (&ms, e0, e1) ;
    hipError_t err = hipGetLastError () ;
    (void) hipFree (d) ; (void) hipFree (dg) ;
    (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ;
    if (err != hipSuccess || ms <= 0) return CHOLMOD_HIP_GPU_PROBLEM ;
    const double entries = (double) n1 * (n1 + 1) / 2 + (double) (m1 - n1) * n1 + (double) m2 * (m2 + 1) / 2 ;
    return 2.0 * entries * k * iters / (ms * 1e-3) ;
}

/* mixed MFMA+VALU issue test: returns seconds; flops are computed by the caller */
double cholmod_hip_bench_mixed (int blocks_per_cu, int it_mfma, int it_valu)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    int blocks = 256 * blocks_per_cu ;
    double *d = nullptr ;
    if (hipMalloc ((void **) &d, (size_t) blocks * 256 * sizeof (double)) != hipSuccess) return CHOLMOD_HIP_OUT_OF_MEMORY ;
    hipEvent_t e0, e1 ;
    (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ;
    hipLaunchKernelGGL (k_mixed_peak, dim3 (blocks), dim3 (256), 0, 0, d, 4, 4) ;
    (void) hipDeviceSynchronize () ;
    (void) hipEventRecord (e0, 0) ;
    hipLaunchKernelGGL (k_mixed_peak, dim3 (blocks), dim3 (256), 0, 0, d, it_mfma, it_valu) ;
    (void) hipEventRecord (e1, 0) ;
    (void) hipEventSynchronize (e1) ;
    float ms = 0 ;
    (void) hipEventElapsedTime (&ms, e0, e1) ;
    (void) hipFree (d) ;
    (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ;
    return ms * 1e-3 ;
}

/* tuning probe: per-phase shader-clock cycles of one 64x64 k_potrf (lane 0):
 * out[0] stage, [1] panel update, [2] publish+barrier, [3] 8x8 factor,
 * [4] row solve+store, [5] barrier, [6] write-back */
int cholmod_hip_debug_potrf_cycles (long long *out8)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    const int n = 64 ;
    std::vector<double> A (n * n) ;
    for (int j = 0 ; j < n ; j++) for (int i = 0 ; i < n ; i++) A [i + j * n] = (i == j) ? n + 1.0 : 1.0 / (1.0 + abs (i - j)) ;
    double *d = nullptr ; i32 *dinfo = nullptr ; long long *dt = nullptr ; PfGroup *dg = nullptr ;
    HIPCHK (hipMalloc ((void **) &d, n * n * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &dinfo, sizeof (i32))) ;
    HIPCHK (hipMalloc ((void **) &dt, 8 * sizeof (long long))) ;
    HIPCHK (hipMalloc ((void **) &dg, sizeof (PfGroup))) ;
    PfGroup G {0, n, n, 0, 0} ;
    HIPCHK (hipMemcpy (dg, &G, sizeof (G), hipMemcpyHostToDevice)) ;
    HIPCHK (hipMemset (dinfo, 0, sizeof (i32))) ;
    for (int rep = 0 ; rep < 3 ; rep++)
    {
        HIPCHK (hipMemcpy (d, A.data (), n * n * sizeof (double), hipMemcpyHostToDevice)) ;
        hipLaunchKernelGGL (k_potrf<true>, dim3 (1), dim3 (64), 0, 0, dg, d, dinfo, dt) ;
        HIPCHK (hipDeviceSynchronize ()) ;
    }
    HIPCHK (hipMemcpy (out8, dt, 8 * sizeof (long long), hipMemcpyDeviceToHost)) ;
    (void) hipFree (d) ; (void) hipFree (dinfo) ; (void) hipFree (dt) ; (void) hipFree (dg) ;
    return CHOLMOD_HIP_OK ;
}

/* tuning probe: per-phase shader-clock cycles of the matrix-core panel kernels
 * on one 64x64 diagonal block with 64 rows below it.  out16 [0..7]: k_potrf_mfma
 * (stage, column chain, scale+store, barrier, trailing tiles, barrier,
 * write-back); out16 [8..15]: k_trsm_mfma (stage, reciprocals, diagonal
 * inverses, update MFMAs, barrier, diagonal MFMAs + store, barrier). */
int cholmod_hip_debug_panel_cycles (long long *out16)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    { int rl = raise_lds_limits () ; if (rl != CHOLMOD_HIP_OK) return rl ; }
    const int n = 64, m = 128 ;
    std::vector<double> A ((size_t) m * n) ;
    for (int j = 0 ; j < n ; j++) for (int i = 0 ; i < m ; i++) A [i + (size_t) j * m] = (i == j) ? n + 1.0 : 1.0 / (1.0 + abs (i - j)) ;
    double *d = nullptr ; i32 *dinfo = nullptr ; long long *dt = nullptr ; PfGroup *dg = nullptr ; TrGroup *dtg = nullptr ;
    HIPCHK (hipMalloc ((void **) &d, A.size () * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &dinfo, sizeof (i32))) ;
    HIPCHK (hipMalloc ((void **) &dt, 16 * sizeof (long long))) ;
    HIPCHK (hipMalloc ((void **) &dg, sizeof (PfGroup))) ;
    HIPCHK (hipMalloc ((void **) &dtg, sizeof (TrGroup))) ;
    PfGroup G {0, m, n, 0, 0} ;
    TrGroup T {0, n, m, m - n, n, 0, 0, 0} ;
    HIPCHK (hipMemcpy (dg, &G, sizeof (G), hipMemcpyHostToDevice)) ;
    HIPCHK (hipMemcpy (dtg, &T, sizeof (T), hipMemcpyHostToDevice)) ;
    HIPCHK (hipMemset (dinfo, 0, sizeof (i32))) ;
    for (int rep = 0 ; rep < 3 ; rep++)
    {
        HIPCHK (hipMemcpy (d, A.data (), A.size () * sizeof (double), hipMemcpyHostToDevice)) ;
        hipLaunchKernelGGL (k_potrf_mfma<true>, dim3 (1), dim3 (256), 0, 0, dg, d, dinfo, dt) ;
        hipLaunchKernelGGL (k_trsm_mfma<true>, dim3 (1), dim3 (256), trsm_mfma_lds_bytes (64), 0,
            dtg, 1, d, dinfo, 64, dt + 8) ;
        HIPCHK (hipDeviceSynchronize ()) ;
    }
    HIPCHK (hipMemcpy (out16, dt, 16 * sizeof (long long), hipMemcpyDeviceToHost)) ;
    (void) hipFree (d) ;
    // wall time of whole launches (HIP events, ns): [7] one potrf workgroup,
    // [15] a trsm over 16 000 rows (250 workgroups) as at the top of Poisson 100^3
    {
        const int mm = 16064 ;
        double *big = nullptr ;
        HIPCHK (hipMalloc ((void **) &big, (size_t) mm * n * sizeof (double))) ;
        std::vector<double> Ab ((size_t) mm * n) ;
        for (int j = 0 ; j < n ; j++) for (int i = 0 ; i < mm ; i++) Ab [i + (size_t) j * mm] = (i == j) ? n + 1.0 : 1.0 / (1.0 + abs (i - j) % 97) ;
        PfGroup G2 {0, mm, n, 0, 0} ;
        TrGroup T2 {0, n, mm, mm - n, n, 0, 0, 0} ;
        HIPCHK (hipMemcpy (dg, &G2, sizeof (G2), hipMemcpyHostToDevice)) ;
        HIPCHK (hipMemcpy (dtg, &T2, sizeof (T2), hipMemcpyHostToDevice)) ;
        hipEvent_t e0, e1, e2 ;
        (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ; (void) hipEventCreate (&e2) ;
        float best_p = 1e30f, best_t = 1e30f ;
        for (int rep = 0 ; rep < 5 ; rep++)
        {
            HIPCHK (hipMemcpy (big, Ab.data (), Ab.size () * sizeof (double), hipMemcpyHostToDevice)) ;
            HIPCHK (hipEventRecord (e0, 0)) ;
            hipLaunchKernelGGL (k_potrf_mfma<false>, dim3 (1), dim3 (256), 0, 0, dg, big, dinfo, (long long *) nullptr) ;
            HIPCHK (hipEventRecord (e1, 0)) ;
            hipLaunchKernelGGL (k_trsm_mfma<false>, dim3 ((mm - n + TRM_ROWS - 1) / TRM_ROWS), dim3 (256),
                trsm_mfma_lds_bytes (64), 0, dtg, 1, big, dinfo, 64, (long long *) nullptr) ;
            HIPCHK (hipEventRecord (e2, 0)) ;
            HIPCHK (hipDeviceSynchronize ()) ;
            float a = 0, b = 0 ;
            HIPCHK (hipEventElapsedTime (&a, e0, e1)) ;
            HIPCHK (hipEventElapsedTime (&b, e1, e2)) ;
            best_p = std::min (best_p, a) ; best_t = std::min (best_t, b) ;
        }
        out16 [7] = (long long) (best_p * 1e6) ;
        out16 [15] = (long long) (best_t * 1e6) ;
        (void) hipFree (big) ;
        (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ; (void) hipEventDestroy (e2) ;
    }
    (void) hipFree (dinfo) ; (void) hipFree (dt) ; (void) hipFree (dg) ; (void) hipFree (dtg) ;
    return CHOLMOD_HIP_OK ;
}

/* tuning probe: cycles per repetition of the basic fp64 instruction patterns of
 * the panel kernels, one wave (see k_latency_probe); out8 [v] = cycles for n reps */
int cholmod_hip_debug_latency (long long *out8, int n)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    double *sink = nullptr ; long long *dt = nullptr ;
    HIPCHK (hipMalloc ((void **) &sink, 64 * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &dt, 8 * sizeof (long long))) ;
    for (int rep = 0 ; rep < 2 ; rep++)
    {
        hipLaunchKernelGGL (k_latency_probe, dim3 (1), dim3 (64), 0, 0, sink, dt, n) ;
        HIPCHK (hipDeviceSynchronize ()) ;
    }
    HIPCHK (hipMemcpy (out8, dt, 8 * sizeof (long long), hipMemcpyDeviceToHost)) ;
    (void) hipFree (sink) ; (void) hipFree (dt) ;
    return CHOLMOD_HIP_OK ;
}

/* issue loops with the update kernel's operand pattern: variant = 100 ti + 10 tj + ldsread */
double cholmod_hip_bench_mfma_peak2 (int variant, int waves_per_simd, int iters, int zero_operands)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    if (waves_per_simd < 1) waves_per_simd = 1 ;
    int blocks = 256 * waves_per_simd ;
    double *d = nullptr ;
    if (hipMalloc ((void **) &d, (size_t) blocks * 256 * sizeof (double)) != hipSuccess) return CHOLMOD_HIP_OUT_OF_MEMORY ;
    double scale = zero_operands ? 0.0 : 1.0 ;
    int ti = variant / 100, tj = (variant / 10) % 10, lr = variant % 10 ;
    auto launch = [&] (int it) -> bool
    {
#define P2(TI_, TJ_) if (ti == TI_ && tj == TJ_) { \
        if (lr) hipLaunchKernelGGL ((k_mfma_peak2<TI_, TJ_, true>), dim3 (blocks), dim3 (256), 0, 0, d, it, scale) ; \
        else hipLaunchKernelGGL ((k_mfma_peak2<TI_, TJ_, false>), dim3 (blocks), dim3 (256), 0, 0, d, it, scale) ; return true ; }
        P2 (1, 1) P2 (2, 2) P2 (2, 4) P2 (4, 4) P2 (1, 4) P2 (4, 2)
#undef P2
        return false ;
    } ;
    if (!launch (16)) { (void) hipFree (d) ; return CHOLMOD_HIP_INVALID ; }
    hipEvent_t e0, e1 ;
    (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ;
    (void) hipDeviceSynchronize () ;
    (void) hipEventRecord (e0, 0) ;
    launch (iters) ;
    (void) hipEventRecord (e1, 0) ;
    (void) hipEventSynchronize (e1) ;
    float ms = 0 ;
    (void) hipEventElapsedTime (&ms, e0, e1) ;
    hipError_t err = hipGetLastError () ;
    (void) hipFree (d) ;
    (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ;
    if (err != hipSuccess || ms <= 0) return CHOLMOD_HIP_GPU_PROBLEM ;
    return (double) blocks * 4.0 * iters * (double) (ti * tj) * 2048.0 / (ms * 1e-3) ;
}

double cholmod_hip_bench_mfma_peak (int waves_per_simd, int iters)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    double scale = 1.0 ;                        // 10000 + ...: all-zero operands
    if (waves_per_simd >= 10000) { scale = 0.0 ; waves_per_simd -= 10000 ; }
    int fill = 0 ;                              // 1000 f + w: filler f between the MFMAs
    if (waves_per_simd >= 1000) { fill = waves_per_simd / 1000 ; waves_per_simd %= 1000 ; }
    bool valu = waves_per_simd < 0 ;            // negative: fp64 VALU FMA loop instead
    if (valu) waves_per_simd = -waves_per_simd ;
    bool acc16 = waves_per_simd >= 100 ;        // 100 + w: sixteen accumulators per wave
    if (acc16) waves_per_simd -= 100 ;
    if (waves_per_simd < 1) waves_per_simd = 1 ;
    if (iters < 1) iters = 1 ;
    int blocks = 256 * waves_per_simd ;         // 256 CUs x (4 waves per block = 1 per SIMD)
    double *d = nullptr ;
    if (hipMalloc ((void **) &d, (size_t) blocks * 256 * sizeof (double)) != hipSuccess) return CHOLMOD_HIP_OUT_OF_MEMORY ;
    hipEvent_t e0, e1 ;
    (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ;
    auto launch = [&] (int it)
    {
        if (valu) hipLaunchKernelGGL ((k_valu_peak<16>), dim3 (blocks), dim3 (256), 0, 0, d, it) ;
        else if (acc16) hipLaunchKernelGGL ((k_mfma_peak<16>), dim3 (blocks), dim3 (256), 0, 0, d, it, scale) ;
        else if (fill == 1) hipLaunchKernelGGL ((k_mfma_peak<8, 1>), dim3 (blocks), dim3 (256), 0, 0, d, it, scale) ;
        else if (fill == 2) hipLaunchKernelGGL ((k_mfma_peak<8, 2>), dim3 (blocks), dim3 (256), 0, 0, d, it, scale) ;
        else if (fill == 3) hipLaunchKernelGGL ((k_mfma_peak<8, 3>), dim3 (blocks), dim3 (256), 0, 0, d, it, scale) ;
        else hipLaunchKernelGGL ((k_mfma_peak<8>), dim3 (blocks), dim3 (256), 0, 0, d, it, scale) ;
    } ;
    launch (16) ;
    (void) hipDeviceSynchronize () ;
    (void) hipEventRecord (e0, 0) ;
    launch (iters) ;
    (void) hipEventRecord (e1, 0) ;
    (void) hipEventSynchronize (e1) ;
    float ms = 0 ;
    (void) hipEventElapsedTime (&ms, e0, e1) ;
    hipError_t err = hipGetLastError () ;
    (void) hipFree (d) ;
    (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ;
    if (err != hipSuccess || ms <= 0) return CHOLMOD_HIP_GPU_PROBLEM ;
    if (valu) return (double) blocks * 256.0 * iters * 16.0 * 2.0 / (ms * 1e-3) ;
    return (double) blocks * 4.0 * iters * (acc16 ? 16.0 : 8.0) * 2048.0 / (ms * 1e-3) ;
}

/* The fp64 matrix-core ceiling: an inline-assembly v_mfma_f64_16x16x4_f64 loop (k_mfma_ceiling),
 * `waves_per_simd` resident waves per SIMD on every CU, `nacc` (4 or 8) independent
 * accumulators per wave.  Returns flop/s from HIP events; out3 [0] = shader cycles per MFMA
 * per SIMD (64 = the pipe's issue rate), out3 [1] = the shader clock sustained inside the loop
 * in GHz (s_memtime against the 100 MHz s_memrealtime), out3 [2] = flop/s that the issue rate
 * would give at 2.4 GHz. */
double cholmod_hip_bench_mfma_ceiling (int waves_per_simd, int nacc, int iters, int zero_operands, double *out3)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    if (waves_per_simd < 1) waves_per_simd = 1 ;
    if (iters < 1) iters = 1 ;
    if (nacc != 8) nacc = 4 ;
    int blocks = 256 * waves_per_simd ;
    double *d = nullptr ; long long *st = nullptr ;
    if (hipMalloc ((void **) &d, (size_t) blocks * 256 * sizeof (double)) != hipSuccess) return CHOLMOD_HIP_OUT_OF_MEMORY ;
    if (hipMalloc ((void **) &st, (size_t) blocks * 2 * sizeof (long long)) != hipSuccess) { (void) hipFree (d) ; return CHOLMOD_HIP_OUT_OF_MEMORY ; }
    double scale = zero_operands ? 0.0 : 1.0 ;
    auto launch = [&] (int it)
    {
        if (nacc == 8) hipLaunchKernelGGL ((k_mfma_ceiling<8>), dim3 (blocks), dim3 (256), 0, 0, d, st, it, scale) ;
        else hipLaunchKernelGGL ((k_mfma_ceiling<4>), dim3 (blocks), dim3 (256), 0, 0, d, st, it, scale) ;
    } ;
    hipEvent_t e0, e1 ;
    (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ;
    launch (64) ;
    (void) hipDeviceSynchronize () ;
    (void) hipEventRecord (e0, 0) ;
    launch (iters) ;
    (void) hipEventRecord (e1, 0) ;
    (void) hipEventSynchronize (e1) ;
    float ms = 0 ;
    (void) hipEventElapsedTime (&ms, e0, e1) ;
    hipError_t err = hipGetLastError () ;
    std::vector<long long> h ((size_t) blocks * 2) ;
    if (err == hipSuccess) err = hipMemcpy (h.data (), st, h.size () * sizeof (long long), hipMemcpyDeviceToHost) ;
    (void) hipFree (d) ; (void) hipFree (st) ;
    (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ;
    if (err != hipSuccess || ms <= 0) return CHOLMOD_HIP_GPU_PROBLEM ;
    if (out3)
    {
        double cyc = 0, ticks = 0 ;
        for (int b = 0 ; b < blocks ; b++) { cyc += (double) h [2 * b] ; ticks += (double) h [2 * b + 1] ; }
        double per_wave = 4.0 * nacc * (double) iters ;               // MFMAs one wave issued
        // waves_per_simd waves share a SIMD's pipe while they are co-resident
        out3 [0] = cyc / blocks / (per_wave * waves_per_simd) ;
        out3 [1] = ticks > 0 ? cyc / ticks * 0.1 : 0.0 ;                // cycles per 10 ns tick -> GHz
        out3 [2] = out3 [0] > 0 ? 256.0 * 4.0 * 2048.0 / out3 [0] * 2.4e9 : 0.0 ;
    }
    return (double) blocks * 4.0 * (4.0 * nacc * (double) iters) * 2048.0 / (ms * 1e-3) ;
}


/* Tuning probe: the update kernel selected by `flags` (as cholmod_hip_bench_update_kernel)
 * against k_update2<64,64,16,2,false> on the same random operands, triangular region or not,
 * assign mode or not: returns max |difference| / max |reference| over the target region
 * (negative = a CHOLMOD_HIP_* code). */
double cholmod_hip_debug_update_diff (int64_t m, int64_t n, int64_t k, int tri, int assign, int flags)
{
    if (m <= 0 || n <= 0 || k <= 0 || (tri && m < n)) return CHOLMOD_HIP_INVALID ;
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    i64 ld = std::max (m, n) + 3 ;
    i64 a_off = 1, b_off = 1 + ld * k, c_off = 2 + 2 * ld * k ;      // (odd offsets: 8-byte alignment only)
    i64 total = c_off + (m + 5) * n ;
    std::vector<double> h (total), r0 (total), r1 (total) ;
    unsigned long long sdd = 88172645463325252ull ;
    for (i64 q = 0 ; q < total ; q++)
    {
        sdd ^= sdd << 13 ; sdd ^= sdd >> 7 ; sdd ^= sdd << 17 ;
        h [q] = (double) (sdd >> 11) / 9007199254740992.0 - 0.5 ;
    }
    double *d = nullptr ; GemmGroup *dg = nullptr ;
    if (hipMalloc ((void **) &d, (total + UPD3_LX_PAD) * sizeof (double)) != hipSuccess) return CHOLMOD_HIP_OUT_OF_MEMORY ;
    HIPCHK (hipMalloc ((void **) &dg, sizeof (GemmGroup))) ;
    GemmGroup G ;
    memset (&G, 0, sizeof (G)) ;
    G.a_off = a_off ; G.b_off = tri ? a_off : b_off ; G.c_off = c_off ; G.lda = (i32) ld ; G.ldc = (i32) (m + 5) ;
    G.m = (i32) m ; G.n = (i32) n ; G.k = (i32) k ; G.tri = tri ? 1 : 0 ; G.assign = assign ? 1 : 0 ;
    G.tile_mul = 1 ; G.tile_add = 0 ;
    G.mt = (i32) ((m + 63) / 64) ; G.nt = (i32) ((n + 63) / 64) ;
    G.ntiles = tri ? G.nt * (G.nt + 1) / 2 + (G.mt - G.nt) * G.nt : G.mt * G.nt ;
    G.nblk = (G.ntiles + 63) / 64 * 64 ;
    G.swz = G.nblk >= 1024 ;
    HIPCHK (hipMemcpy (dg, &G, sizeof (G), hipMemcpyHostToDevice)) ;
    for (int pass = 0 ; pass < 2 ; pass++)
    {
        HIPCHK (hipMemcpy (d, h.data (), total * sizeof (double), hipMemcpyHostToDevice)) ;
        if (pass == 1 && (flags & 262144))
        {
            // the one-wave-per-tile kernel walking 16 x 16 super-tiles (swz = 2), against k_update2's 8 x 8 walk
            G.swz = 2 ; G.nblk = (G.ntiles + 255) / 256 * 256 ;
            HIPCHK (hipMemcpy (dg, &G, sizeof (G), hipMemcpyHostToDevice)) ;
        }
        if (pass == 0) hipLaunchKernelGGL ((k_update2<SMALL, SMALL, BKK, 2, false>), dim3 (G.nblk), dim3 (256), 0, 0, dg, 1, d, d) ;
        else if ((flags & 524288) && (flags & 32768)) hipLaunchKernelGGL ((k_update3<4, 0, 1, true>), dim3 (2 * ((G.nblk + 7) / 8 * 8)), dim3 (64), 0, 0, dg, 1, d, d) ;
        else if (flags & 524288) hipLaunchKernelGGL ((k_update3<2, 0, 1, true>), dim3 (2 * ((G.nblk + 7) / 8 * 8)), dim3 (64), 0, 0, dg, 1, d, d) ;
        else if (flags & 16384) hipLaunchKernelGGL ((k_update3<2>), dim3 (G.nblk), dim3 (64), 0, 0, dg, 1, d, d) ;
        else if (flags & 32768) hipLaunchKernelGGL ((k_update3<4>), dim3 (G.nblk), dim3 (64), 0, 0, dg, 1, d, d) ;
        else hipLaunchKernelGGL ((k_update3<3>), dim3 (G.nblk), dim3 (64), 0, 0, dg, 1, d, d) ;
        HIPCHK (hipDeviceSynchronize ()) ;
        HIPCHK (hipMemcpy ((pass ? r1 : r0).data (), d, total * sizeof (double), hipMemcpyDeviceToHost)) ;
    }
    (void) hipFree (d) ; (void) hipFree (dg) ;
    double mx = 0, df = 0 ;
    bool touched = false ;
    for (i64 q = 0 ; q < total ; q++)
    {
        mx = std::max (mx, fabs (r0 [q])) ;
        df = std::max (df, fabs (r0 [q] - r1 [q])) ;
        if (r0 [q] != h [q]) touched = true ;
    }
    if (!touched) return CHOLMOD_HIP_GPU_PROBLEM ;
    return df / mx ;
}

/* Tuning probe: per-phase shader cycles (wave 0) of one k_diag workgroup on a w x w diagonal sub-block:
 * [0] loads + left-looking update of the panel's diagonal block, [1] into LDS + barrier, [2] the 64 x 64
 * elimination, [3] store + negated copy + barrier, [4] 16 x 16 inverses + barrier + publish, [5] loads +
 * left-looking update of the row chunks, [6] their solves, [7] closing barrier; out [8] = whole launch in ns
 * (HIP events), out [9] = k_rowsolve over `m` rows below the same block in ns. */
int cholmod_hip_debug_diag_cycles (long long *out10, int w, int m)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    if (w < 1 || w > DG_W || m < 0) return CHOLMOD_HIP_INVALID ;
    const int n = w + m ;
    std::vector<double> A ((size_t) n * w) ;
    for (int j = 0 ; j < w ; j++) for (int i = 0 ; i < n ; i++) A [i + (size_t) j * n] = (i == j) ? w + 1.0 : 1.0 / (1.0 + abs (i - j) % 61) ;
    double *d = nullptr, *dinv = nullptr ; i32 *dinfo = nullptr ; long long *dt = nullptr ; DgGroup *dg = nullptr ; RsGroup *rg = nullptr ;
    HIPCHK (hipMalloc ((void **) &d, A.size () * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &dinv, 4096 * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &dinfo, sizeof (i32))) ;
    HIPCHK (hipMalloc ((void **) &dt, 8 * sizeof (long long))) ;
    HIPCHK (hipMalloc ((void **) &dg, sizeof (DgGroup))) ;
    HIPCHK (hipMalloc ((void **) &rg, sizeof (RsGroup))) ;
    DgGroup G {0, n, w, 0, 0, 0, 0} ;
    RsGroup R {0, (i64) w, n, m, w, 0, 0, 0, 0, 0} ;
    HIPCHK (hipMemcpy (dg, &G, sizeof (G), hipMemcpyHostToDevice)) ;
    HIPCHK (hipMemcpy (rg, &R, sizeof (R), hipMemcpyHostToDevice)) ;
    HIPCHK (hipMemset (dinfo, 0, sizeof (i32))) ;
    HIPCHK (hipFuncSetAttribute ((const void *) k_rowsolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int) rowsolve_lds_bytes ())) ;
    hipEvent_t e0, e1, e2 ;
    (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ; (void) hipEventCreate (&e2) ;
    float best_d = 1e30f, best_r = 1e30f ;
    for (int rep = 0 ; rep < 4 ; rep++)
    {
        HIPCHK (hipMemcpy (d, A.data (), A.size () * sizeof (double), hipMemcpyHostToDevice)) ;
        HIPCHK (hipEventRecord (e0, 0)) ;
        if (rep == 0) hipLaunchKernelGGL (k_diag<true>, dim3 (1), dim3 (256), 0, 0, dg, d, dinfo, dinv, dt) ;
        else hipLaunchKernelGGL (k_diag<false>, dim3 (1), dim3 (256), 0, 0, dg, d, dinfo, dinv, (long long *) nullptr) ;
        HIPCHK (hipEventRecord (e1, 0)) ;
        if (m > 0) hipLaunchKernelGGL (k_rowsolve, dim3 ((m + RS_ROWS - 1) / RS_ROWS), dim3 (256), rowsolve_lds_bytes (), 0, rg, 1, d, dinfo, dinv) ;
        HIPCHK (hipEventRecord (e2, 0)) ;
        HIPCHK (hipDeviceSynchronize ()) ;
        float a = 0, b = 0 ;
        HIPCHK (hipEventElapsedTime (&a, e0, e1)) ;
        HIPCHK (hipEventElapsedTime (&b, e1, e2)) ;
        if (rep > 0) { best_d = std::min (best_d, a) ; best_r = std::min (best_r, b) ; }
    }
    HIPCHK (hipMemcpy (out10, dt, 8 * sizeof (long long), hipMemcpyDeviceToHost)) ;
    out10 [8] = (long long) (best_d * 1e6) ;
    out10 [9] = (long long) (best_r * 1e6) ;
    i32 inf = 0 ;
    HIPCHK (hipMemcpy (&inf, dinfo, sizeof (i32), hipMemcpyDeviceToHost)) ;
    (void) hipFree (d) ; (void) hipFree (dinv) ; (void) hipFree (dinfo) ; (void) hipFree (dt) ; (void) hipFree (dg) ; (void) hipFree (rg) ;
    (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ; (void) hipEventDestroy (e2) ;
    return inf == 0 ? CHOLMOD_HIP_OK : CHOLMOD_HIP_NOT_POSDEF ;
}


/* ---- CU masks: which compute units a masked stream's workgroups land on ------------------
 * (tuning: can the panel chain of the next outer block run on CUs of its own beside the trailing
 * update?)  Launches `blocks` one-wave workgroups that each spin for `spin_us` on a stream created
 * with hipExtStreamCreateWithCUMask (nwords 32-bit words; nwords == 0: an ordinary stream) and
 * records (XCC_ID, HW_ID) of each: out [b] = (xcc << 32) | hw_id.  Returns the launch's
 * milliseconds or a negative status. */
__global__ void k_where_am_i (long long *out, long long spin_ticks)
{
    if (threadIdx.x == 0)
    {
        unsigned hw = __builtin_amdgcn_s_getreg ((31 << 11) | 4) ;      // HW_REG_HW_ID, 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg ((31 << 11) | 20) ;    // HW_REG_XCC_ID
        out [blockIdx.x] = ((long long) xcc << 32) | hw ;
    }
    // (spin_ticks < 0: uneven durations, 1 .. 4 times |spin_ticks| by a hash of the block index -- does the block -> XCD
    // assignment stay blockIdx % 8 when workgroups retire out of order?  tools/xcd_map.py)
    if (spin_ticks < 0) spin_ticks = -spin_ticks * (1 + (long long) (((unsigned) blockIdx.x * 2654435761u) >> 30)) ;
    long long t0 = __builtin_amdgcn_s_memrealtime () ;
    while (__builtin_amdgcn_s_memrealtime () - t0 < spin_ticks) __builtin_amdgcn_s_sleep (8) ;
}

double cholmod_hip_probe_cu_mask (const uint32_t *mask, int nwords, int blocks, int spin_us, long long *out)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    hipStream_t st = nullptr ;
    hipError_t e = nwords > 0 ? hipExtStreamCreateWithCUMask (&st, (uint32_t) nwords, mask) : hipStreamCreate (&st) ;
    if (e != hipSuccess) { (void) hipGetLastError () ; return CHOLMOD_HIP_INVALID ; }
    long long *d = nullptr ;
    if (hipMalloc ((void **) &d, (size_t) blocks * sizeof (long long)) != hipSuccess) { (void) hipStreamDestroy (st) ; return CHOLMOD_HIP_OUT_OF_MEMORY ; }
    hipEvent_t e0, e1 ;
    (void) hipEventCreate (&e0) ; (void) hipEventCreate (&e1) ;
    (void) hipEventRecord (e0, st) ;
    hipLaunchKernelGGL (k_where_am_i, dim3 (blocks), dim3 (64), 0, st, d, (long long) spin_us * 100) ;
    (void) hipEventRecord (e1, st) ;
    e = hipStreamSynchronize (st) ;
    float ms = 0 ;
    (void) hipEventElapsedTime (&ms, e0, e1) ;
    if (e == hipSuccess) e = hipMemcpy (out, d, (size_t) blocks * sizeof (long long), hipMemcpyDeviceToHost) ;
    (void) hipFree (d) ; (void) hipEventDestroy (e0) ; (void) hipEventDestroy (e1) ; (void) hipStreamDestroy (st) ;
    return e == hipSuccess ? (double) ms : (double) CHOLMOD_HIP_GPU_PROBLEM ;
}

/* Two kernels side by side on two masked streams: the one-wave-per-tile update on a triangular
 * region (m rows, contraction k) on stream A (mask_a), and `nchain` dependent launches of a small
 * kernel that spins chain_us each on stream B (mask_b).  out4: [0] update alone ms, [1] chain
 * alone ms, [2] both started together: update ms, [3] chain ms. */
int cholmod_hip_probe_overlap (const uint32_t *mask_a, const uint32_t *mask_b, int nwords, int64_t m, int64_t k,
    int nchain, int chain_blocks, int chain_us, double *out4)
{
    if (!probe_device ()) return CHOLMOD_HIP_NO_DEVICE ;
    hipStream_t sa = nullptr, sb = nullptr ;
    if (nwords > 0)
    {
        HIPCHK (hipExtStreamCreateWithCUMask (&sa, (uint32_t) nwords, mask_a)) ;
        HIPCHK (hipExtStreamCreateWithCUMask (&sb, (uint32_t) nwords, mask_b)) ;
    }
    else { HIPCHK (hipStreamCreate (&sa)) ; HIPCHK (hipStreamCreate (&sb)) ; }
    const i64 lda = m ;
    double *d = nullptr ; long long *w = nullptr ; GemmGroup *dg = nullptr ;
    HIPCHK (hipMalloc ((void **) &d, (size_t) (lda * (k + m)) * sizeof (double))) ;
    HIPCHK (hipMemset (d, 0, (size_t) (lda * (k + m)) * sizeof (double))) ;
    HIPCHK (hipMalloc ((void **) &w, (size_t) chain_blocks * sizeof (long long))) ;
    GemmGroup G ;
    memset (&G, 0, sizeof (G)) ;
    G.a_off = 0 ; G.b_off = 0 ; G.c_off = lda * k ; G.lda = (i32) lda ; G.ldc = (i32) lda ;
    G.m = (i32) m ; G.n = (i32) m ; G.k = (i32) k ; G.tri = 1 ; G.tile_mul = 1 ;
    G.mt = G.nt = (i32) ((m + 63) / 64) ;
    G.ntiles = (i32) ((i64) G.nt * (G.nt + 1) / 2) ;
    G.nblk = (G.ntiles + 63) / 64 * 64 ; G.swz = G.nblk >= 1024 ;
    HIPCHK (hipMalloc ((void **) &dg, sizeof (G))) ;
    HIPCHK (hipMemcpy (dg, &G, sizeof (G), hipMemcpyHostToDevice)) ;
    hipEvent_t a0, a1, b0, b1 ;
    (void) hipEventCreate (&a0) ; (void) hipEventCreate (&a1) ; (void) hipEventCreate (&b0) ; (void) hipEventCreate (&b1) ;
    auto upd = [&] () { hipLaunchKernelGGL ((k_update3<4>), dim3 (G.nblk), dim3 (64), 0, sa, dg, 1, d, d) ; } ;
    auto chain = [&] () { for (int q = 0 ; q < nchain ; q++) hipLaunchKernelGGL (k_where_am_i, dim3 (chain_blocks), dim3 (64), 0, sb, w, (long long) chain_us * 100) ; } ;
    float t = 0 ;
    upd () ; HIPCHK (hipDeviceSynchronize ()) ;
    (void) hipEventRecord (a0, sa) ; upd () ; (void) hipEventRecord (a1, sa) ; HIPCHK (hipDeviceSynchronize ()) ;
    (void) hipEventElapsedTime (&t, a0, a1) ; out4 [0] = t ;
    (void) hipEventRecord (b0, sb) ; chain () ; (void) hipEventRecord (b1, sb) ; HIPCHK (hipDeviceSynchronize ()) ;
    (void) hipEventElapsedTime (&t, b0, b1) ; out4 [1] = t ;
    (void) hipEventRecord (a0, sa) ; upd () ; (void) hipEventRecord (a1, sa) ;
    (void) hipEventRecord (b0, sb) ; chain () ; (void) hipEventRecord (b1, sb) ;
    HIPCHK (hipDeviceSynchronize ()) ;
    (void) hipEventElapsedTime (&t, a0, a1) ; out4 [2] = t ;
    (void) hipEventElapsedTime (&t, b0, b1) ; out4 [3] = t ;
    (void) hipFree (d) ; (void) hipFree (w) ; (void) hipFree (dg) ;
    (void) hipEventDestroy (a0) ; (void) hipEventDestroy (a1) ; (void) hipEventDestroy (b0) ; (void) hipEventDestroy (b1) ;
    (void) hipStreamDestroy (sa) ; (void) hipStreamDestroy (sb) ;
    return CHOLMOD_HIP_OK ;
}

/* one-way hand-off time in microseconds between blocks `prod` and `cons` of one launch (k_handoff), or -1; *bad: bit 0 = a
 * payload word arrived stale, bit 1 = a wait ran out */
double cholmod_hip_bench_handoff (int mode, int prod, int cons, int ndbl, int rounds, int *bad)
{
    double *payload = nullptr ; int *flags = nullptr, *dbad = nullptr ; long long *ticks = nullptr ;
    if (hipMalloc ((void **) &payload, (size_t) std::max (ndbl, 1) * sizeof (double)) != hipSuccess) return -1 ;
    if (hipMalloc ((void **) &flags, 64 * sizeof (int)) != hipSuccess) return -1 ;
    if (hipMalloc ((void **) &dbad, sizeof (int)) != hipSuccess) return -1 ;
    if (hipMalloc ((void **) &ticks, sizeof (long long)) != hipSuccess) return -1 ;
    (void) hipMemset (flags, 0, 64 * sizeof (int)) ; (void) hipMemset (dbad, 0, sizeof (int)) ; (void) hipMemset (payload, 0, (size_t) std::max (ndbl, 1) * sizeof (double)) ;
    const int grid = std::max (prod, cons) + 1 ;
    if (mode == 0) hipLaunchKernelGGL (k_handoff<0>, dim3 (grid), dim3 (256), 0, 0, payload, flags, prod, cons, ndbl, rounds, ticks, dbad) ;
    else if (mode == 1) hipLaunchKernelGGL (k_handoff<1>, dim3 (grid), dim3 (256), 0, 0, payload, flags, prod, cons, ndbl, rounds, ticks, dbad) ;
    else hipLaunchKernelGGL (k_handoff<2>, dim3 (grid), dim3 (256), 0, 0, payload, flags, prod, cons, ndbl, rounds, ticks, dbad) ;
    long long t = 0 ; int hb = 0 ;
    bool ok = hipDeviceSynchronize () == hipSuccess && hipMemcpy (&t, ticks, sizeof (t), hipMemcpyDeviceToHost) == hipSuccess
        && hipMemcpy (&hb, dbad, sizeof (hb), hipMemcpyDeviceToHost) == hipSuccess ;
    (void) hipFree (payload) ; (void) hipFree (flags) ; (void) hipFree (dbad) ; (void) hipFree (ticks) ;
    if (bad) *bad = hb ;
    if (!ok) return -1 ;
    return (double) t / 100.0 / (2.0 * rounds) ;        // (s_memrealtime: 100 MHz)
}

} // extern "C"
