/* cholmod_hip_probes.h -- micro-benchmarks and tuning probes of the engine's
 * kernels.  They live in lib/libcholmod_amd_probes.so (csrc/hip/probes.hip), not
 * in the product library: bench.py prints their figures beside the spec peak,
 * tools/ use them for tuning.  flop/s or seconds as stated; negative = a
 * CHOLMOD_HIP_* error code. */
#ifndef CHOLMOD_HIP_PROBES_H
#define CHOLMOD_HIP_PROBES_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Dense fp64 C -= A*B' micro-benchmark on the engine's update kernel k_update2 (used by
 * bench.py to print the measured MFMA rate next to the 78.6 TFLOP/s spec).
 * Returns achieved flop/s, or a negative CHOLMOD_HIP_* code. */
double cholmod_hip_bench_update_kernel (int64_t m, int64_t n, int64_t k,
    int iters, int flags) ;
/* the outer update of a top front, trapezoid + square, as one launch or as two (tools/upd3.py pair) */
double cholmod_hip_bench_update_pair (int64_t m1, int64_t n1, int64_t m2, int64_t k, int64_t ld, int iters, int mode) ;

/* Issue-bound v_mfma_f64_16x16x4_f64 loop without memory traffic: the measured
 * fp64 matrix-core ceiling (flop/s) printed next to the 78.6 TFLOP/s spec. */
double cholmod_hip_bench_mfma_peak (int waves_per_simd, int iters) ;
/* The same with the update kernel's operand pattern: a ti x tj grid of accumulators
 * per wave (variant = 100 ti + 10 tj + ldsread; ldsread = 1 refreshes the fragments
 * from LDS every step), all-zero operands on request. */
double cholmod_hip_bench_mfma_peak2 (int variant, int waves_per_simd, int iters, int zero_operands) ;

/* The measured fp64 matrix-core ceiling (round 3): inline-assembly MFMA loop, nothing else in
 * it.  flop/s; out3 = {shader cycles per MFMA per SIMD, sustained shader clock in GHz, flop/s
 * that issue rate gives at 2.4 GHz}.  (cholmod_hip_bench_mfma_peak's loop carries VGPR <-> AGPR
 * copies the compiler inserted and under-reports; it is kept for the round-1/2 records.) */
double cholmod_hip_bench_mfma_ceiling (int waves_per_simd, int nacc, int iters, int zero_operands, double *out3) ;
/* CU masks (tools/cumask.py): where the workgroups of a stream created with hipExtStreamCreateWithCUMask run
 * (out [b] = (XCC_ID << 32) | HW_ID), and the one-wave-per-tile update beside a chain of small dependent
 * launches on two masked streams */
double cholmod_hip_probe_cu_mask (const uint32_t *mask, int nwords, int blocks, int spin_us, long long *out) ;
int cholmod_hip_probe_overlap (const uint32_t *mask_a, const uint32_t *mask_b, int nwords, int64_t m, int64_t k,
    int nchain, int chain_blocks, int chain_us, double *out4) ;

/* Tuning probe: an update kernel selected by `flags` against k_update2 on the same operands;
 * max |difference| / max |reference| (negative = a CHOLMOD_HIP_* code). */
double cholmod_hip_debug_update_diff (int64_t m, int64_t n, int64_t k, int tri, int assign, int flags) ;

/* Tuning probe: per-phase shader cycles of one k_diag workgroup on a w x w sub-block ([0..7]), the launch in ns [8],
 * k_rowsolve over m rows below it in ns [9]. */
int cholmod_hip_debug_diag_cycles (long long *out10, int w, int m) ;

/* Tuning probe: per-phase shader cycles of one 64x64 k_potrf launch. */
int cholmod_hip_debug_potrf_cycles (long long *out8) ;
/* Same for the matrix-core panel kernels: [0..7] k_potrf_mfma, [8..15] k_trsm_mfma. */
int cholmod_hip_debug_panel_cycles (long long *out16) ;
/* Tuning probe: cycles for n repetitions of basic fp64 instruction patterns (one wave). */
int cholmod_hip_debug_latency (long long *out8, int n) ;

/* Tuning probe: waves 0,1 of every block run the MFMA loop, waves 2,3 a
 * v_fma_f64 loop; returns the seconds the launch took. */
double cholmod_hip_bench_mixed (int blocks_per_cu, int it_mfma, int it_valu) ;


#ifdef __cplusplus
}
#endif
#endif
