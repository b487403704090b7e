/*
 * msm_hip_testhooks.h -- test, benchmark and calibration hooks of the MI355X BN254 G1 MSM engine.
 *
 * NOT part of the product ABI: these symbols exist only in libmsm_hip_hooks.so, a second build of the same sources with
 * -DMSM_HIP_TEST_HOOKS (make -C gpu-acceleration_amd/csrc hooks).  That library also contains the complete engine
 * (include/msm_hip.h), so a hooks context runs exactly the kernels the product runs.  Users: tests/, bench.py (input
 * generation, multiplier calibration), tools/.
 */
#ifndef MSM_HIP_TESTHOOKS_H
#define MSM_HIP_TESTHOOKS_H
#include "msm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic inputs: counterpart of test_utils::generate_random_bases_and_scalars
 *      (metal_msm.rs:698-731).  Base i is k_i*G with k_i = SplitMix64 stream (seed, i) reduced below r,
 *      scalar i likewise from scalar_seed; both written to DEVICE memory (Montgomery bases). ------- */
int32_t msm_bn254_g1_generate_device(msm_ctx *ctx, uint64_t base_seed, uint64_t scalar_seed, size_t n,
                                     void *d_bases_mont_out, void *d_scalars_out);
/* the same k_i / s_i streams on the host (8 words each), for closed-form checks */
int32_t msm_bn254_generate_scalars_host(uint64_t seed, size_t n, int nonzero, uint32_t *out);

/* ---- device-math unit-test hooks: counterpart of the reference's test_* kernels (SURVEY C10,
 *      e.g. mont_mul_cios.metal:8-15, jacobian_add_2007_bl.metal:8-40).  Host arrays in/out. ------ */
#define MSM_OP_FP_ADD 0u
#define MSM_OP_FP_SUB 1u
#define MSM_OP_FP_MONT_MUL 2u
#define MSM_OP_FP_TO_MONT 3u
#define MSM_OP_FP_FROM_MONT 4u
#define MSM_OP_FP_INV 5u
int32_t msm_test_fp_op(msm_ctx *ctx, uint32_t op, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n);
#define MSM_OP_G1_MADD 0u /* a: Jacobian(24) + b: affine Montgomery(16) -> Jacobian(24) */
#define MSM_OP_G1_ADD 1u  /* a: Jacobian(24) + b: Jacobian(24)          -> Jacobian(24) */
#define MSM_OP_G1_DBL 2u  /* a: Jacobian(24)                            -> Jacobian(24) */
#define MSM_OP_G1_ADD_WIDE 3u /* as ADD, computed by 8 cooperating lanes (csrc/ec_wide.hpp, the reduction-tree path) */
#define MSM_OP_G1_MADD_M256 4u     /* as MADD, b's arkworks words gathered as they are (fp_unpack_shl5 + xyzz_madd_m32: k_accumulate_pieces<.., M256>) */
#define MSM_OP_G1_MADD_M256_NEG 5u /* a - b by the same path (the digit's sign applied to S2) */
int32_t msm_test_g1_op(msm_ctx *ctx, uint32_t op, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n);
/* signed/unsigned digit decomposition of the planner's choice, digits[w*n + i] as int32 */
int32_t msm_test_decompose(msm_ctx *ctx, const uint32_t *scalars, size_t n, uint32_t window_bits, int32_t *digits);

/* ---- integer-multiplier calibration --------------------------------------------
 * Runs two saturating micro-kernels on the context's device (4 wavefronts per SIMD, dependent chains, ~1 ms each) and
 * reports what THIS device sustains, per second over the whole chip, lane level:
 *   *mad_per_s     v_mad_u64_u32 operations (the instruction the field multiplication is made of)
 *   *fp_mul_per_s  9 x 29-bit Montgomery multiplications (fp_mul of csrc/fp_bn254.hpp, 171 multiplier instructions each)
 * bench.py prices k_accumulate against these instead of a datasheet figure. */
int32_t msm_calibrate(msm_ctx *ctx, double *mad_per_s, double *fp_mul_per_s);

/* ---- stage-level parity (counterpart of the reference's per-kernel tests: tests/cuzk/transpose.rs:6-118,
 *      smvp.rs:119-303, pbpr.rs:26-247).  Runs the whole pipeline once on host inputs (as msm_bn254_g1, single shot) and
 *      copies the intermediates of every stage back.  All output pointers are nullable; sizes follow msm_plan(n, ...)
 *      with W = num_windows, nb = num_buckets, nv = virtual_points, kb = log2(nb):
 *        digits      W x nv     bucket index | negate << 31, 0xFFFFFFFF = skipped (zero digit / base at infinity)
 *        offsets     W*nb + 1   exclusive prefix of the bucket sizes (CSC column pointer)
 *        sorted      W x nv     first offsets[W*nb] entries valid: virtual point index | negate << 31, grouped by bucket
 *        buckets     W*nb x 24  bucket sums, Jacobian Montgomery words (bucket b of window w holds the digit magnitude b + 1; in the TOP
 *                               window, when t = msm_plan_t.top_digit_bits < kb, the magnitude (b mod 2^t) + 1, the index bits above
 *                               t being low bits of the point index)
 *        bit_sums    W x (kb+1) x 24   Q_{w,u} (u < kb: buckets whose index has bit u set) and Q_{w,kb} = all buckets
 *      With MSM_FLAG_WINDOW_TABLE in the context's flags the RESIDENT path runs (upload + window table, then the resident
 *      pipeline): V = msm_plan_t.bucket_arrays arrays instead of W windows -- offsets V*nb + 1, buckets V*nb x 24, sorted entries are
 *      TABLE indices j*nv + i (window j of the group, virtual point i).  Arrays of more than 2^17 buckets are reduced as
 *      P = 2^(kb-16) pseudo-windows of 2^16 buckets: bit_sums then is (V*P) x 17 x 24 (Q_u over the index INSIDE the slice, u < 16,
 *      and the slice's plain sum).
 *      sort_path (nullable) receives 2 = two-level LDS sort, 1 = tiled LDS histogram, 0 = global-atomic fallback;
 *      big_items (nullable) the number of (region, batch) items of oversized sort regions handed to k_big_place. */
int32_t msm_test_stage_dump(msm_ctx *ctx, const uint32_t *bases_xy, uint32_t base_form, const uint8_t *inf_mask,
                            const uint32_t *scalars, size_t n, uint32_t *digits, uint32_t *offsets, uint32_t *sorted,
                            uint32_t *buckets_jacobian_mont, uint32_t *bit_sums_jacobian_mont, uint32_t *sort_path,
                            uint32_t *big_items, uint32_t out_jacobian_mont[24]);

/* ---- a call that FAILS between its sort and its accumulation (the piece-sort histogram and the bin cursors used to be cleaned by
 *      the accumulation of the call that used them, so such a failure left them dirty for the next call on the context).  Runs the
 *      decomposition, the sort and the piece plan of an n-point MSM on `scalars` (host, n x 8 words) and then returns MSM_ERR_HIP as a
 *      failed copy / event wait in front of the accumulation would: the flag words, the list counters, the piece histogram and its cursors
 *      are left exactly as that failure leaves them.  The next call on the context must be right. */
int32_t msm_test_abandon_after_sort(msm_ctx *ctx, const uint32_t *scalars, size_t n);

/* ---- probes of the bucket reduction's dependent chain (tools/wide_level_probe.py -> profiles/r6_wide_level_breakdown.txt).
 *      msm_probe_wide_level: ONE workgroup of `threads` threads runs `iters` pairwise levels e[i] += e[i + m/2] over m XYZZ records in LDS.
 *        mode 0: eight lanes per addition (the body of lds_tree_wide), loop bracketed only
 *        mode 1: the same with shader-cycle marks inside the addition of wavefront 0 (needs m/2 <= threads/8)
 *        mode 2: one lane per addition (xyzz_add, the form of k_pair_level)
 *      out[0] shader cycles of the loop, out[1] constant-rate (100 MHz) ticks of the loop, out[2 + k] (mode 1) cycles between mark k - 1 and
 *      mark k summed over the levels: 1 operands loaded, 2 stage-1 product, 3 exchange + P / R, 4 squares, 5 special-case vote,
 *      6 exchange + stage-3 operands, 7 stage-3 product, 8 exchange + X3 + stage-4 operands, 9 stage-4 product, 10 exchange + Y3,
 *      11 result stored, 12 loop left, 13 barrier passed.
 *      msm_probe_launch_chain: `launches` dependent launches back to back on the context's stream -- k_pair_level_wide over n_adds additions
 *      each, or an empty kernel (n_adds = 0) -- microseconds per launch by hipEvents. */
int32_t msm_probe_wide_level(msm_ctx *ctx, uint32_t threads, uint32_t m, uint32_t iters, uint32_t mode, long long out[18]);
int32_t msm_probe_launch_chain(msm_ctx *ctx, uint32_t n_adds, uint32_t launches, double *us_per_launch);
/* `launches` dependent launches of an EMPTY kernel of blocks x threads with lds_kb of static LDS per workgroup (0, 1 or 36): what the
 * dispatch of a large grid costs when every workgroup leaves at once (k_combine_pieces on uniform scalars) */
int32_t msm_probe_empty_launch(msm_ctx *ctx, uint32_t blocks, uint32_t threads, uint32_t lds_kb, uint32_t launches, double *us_per_launch);
/* k_reduce_bits_wide of the 2^20 shape (8 windows x 16 bit sums = 128 workgroups) on synthetic row / column sums with parts switched off:
 * parts bit 0 = staging (HBM -> LDS), bit 1 = the LDS tree, bit 2 = the XYZZ -> Jacobian -> R = 2^256 conversion; 7 = the kernel as the product runs it */
int32_t msm_probe_reduce_bits(msm_ctx *ctx, uint32_t parts, uint32_t launches, double *us_per_launch);
/* the list counters the last sort chain / accumulation left on the device: out[0] long-list entries, out[1] mid-list entries (buckets of 3 .. 7
 * pieces k_combine_pieces folds), out[2] pieces, out[3] partial-sum slots, out[4] buckets of exactly two pieces (listed apart) */
int32_t msm_test_get_list_counts(msm_ctx *ctx, uint32_t out[5]);

#ifdef __cplusplus
}
#endif
#endif /* MSM_HIP_TESTHOOKS_H */
