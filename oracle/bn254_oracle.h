/*
 * oracle/bn254_oracle.h -- CPU restatement (plain C) of the reference's BN254 G1
 * variable-base MSM path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call this.  The product (gpu-acceleration_amd/) never does.
 *
 * What is restated, and from where (paths relative to
 * /root/reference/mopro-msm/src/msm/metal_msm/):
 *   - Fq Montgomery arithmetic, R = 2^256           shader/mont_backend/mont.metal:105-181,
 *                                                   utils/mont_reduction.rs:15-40
 *   - Jacobian dbl-2009-l / add-2007-bl / madd-2007-bl   shader/curve/jacobian.metal:11-166
 *   - signed-digit scalar decomposition             shader/cuzk/convert_point_coords_and_decompose_scalars.metal:94-121
 *   - CSR->CSC transpose (stable counting sort)     shader/cuzk/transpose.metal:8-65, tests/cuzk/transpose.rs:95-118
 *   - SMVP bucket sums and sign folding             shader/cuzk/smvp.metal:14-107, tests/cuzk/smvp.rs:256-288
 *   - running-sum bucket reduction                  shader/cuzk/pbpr.metal:33-148, tests/cuzk/pbpr.rs:161-216
 *   - Horner window combine                         metal_msm.rs:204-261
 *   - window-size table                             metal_msm.rs:661-673
 * The *definition of correct* is a third-party dependency absent from
 * /root/reference: ark-ec 0.4.1 `VariableBaseMSM::msm` over ark-bn254 0.4.0 /
 * ark-ff 0.4.1 (mopro-msm/Cargo.toml:25-35, Cargo.lock).  Its published
 * algorithm (signed-window Pippenger, c = ln(n)+2, one rayon task per window)
 * is restated in oracle_msm_pippenger() for the CPU baseline.
 *
 * Pinning: field/curve level is pinned by the reference's own literals
 * (tests/golden/reference_constants.json: p, R mod p, mu, n0, R^-1, generator
 * and identity in both forms) and by EFD known answers 2G, 3G, (r-1)G.  MSM
 * level is pinned by tests/golden/msm_*.npz, produced by an independent
 * pure-Python big-integer implementation (tools/gen_golden.py) evaluated two
 * ways (naive sum and closed form).  The reference holds NO committed MSM
 * vectors and arkworks cannot run in this image, so "bit-exact vs arkworks"
 * rests on the group-element argument (canonical affine coordinates are
 * unique), not on arkworks outputs: MSM-level parity against arkworks itself
 * is unpinned.
 *
 * Word formats: a field element is 8 little-endian uint32 words; a point is
 * x[8] || y[8]; a Jacobian point is X[8] || Y[8] || Z[8]  (limbs_conversion.rs:311-378).
 */
#ifndef BN254_ORACLE_H
#define BN254_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_FORM_STD 0u  /* coordinates are plain integers < p                  */
#define ORACLE_FORM_MONT 1u /* coordinates are x*2^256 mod p (== arkworks Fq.0)    */

/* ---- Fq, 8 LE u32 words in/out ------------------------------------------- */
void oracle_fq_constants(uint32_t p[8], uint32_t r_mod_p[8], uint32_t r2_mod_p[8], uint64_t *inv64);
void oracle_fq_to_mont(const uint32_t a[8], uint32_t out[8]);
void oracle_fq_from_mont(const uint32_t a[8], uint32_t out[8]);
void oracle_fq_mont_mul(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]);
void oracle_fq_add(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]);
void oracle_fq_sub(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]);
void oracle_fq_inv_mont(const uint32_t a[8], uint32_t out[8]); /* a, out Montgomery */

/* ---- G1 Jacobian, Montgomery words, identity has Z = 0 -------------------- */
void oracle_g1_dbl(const uint32_t a[24], uint32_t out[24]);
void oracle_g1_dbl_n(const uint32_t a[24], uint32_t k, uint32_t out[24]); /* 2^k * a */
void oracle_g1_add(const uint32_t a[24], const uint32_t b[24], uint32_t out[24]);
void oracle_g1_madd(const uint32_t a[24], const uint32_t b_xy_mont[16], uint32_t out[24]);
/* Jacobian (Montgomery) -> canonical affine standard-form words; returns 1 for infinity */
int oracle_g1_to_affine_std(const uint32_t a[24], uint32_t out_xy[16]);
/* k * (x,y)  for an affine standard-form base and a 256-bit scalar; out Jacobian Montgomery */
void oracle_g1_scalar_mul(const uint32_t base_xy_std[16], const uint32_t k[8], uint32_t out[24]);

/* ---- MSM.  bases: n x 16 words (form per `form`), inf: n bytes or NULL,
 *      scalars: n x 8 words standard form (< r).  Results: canonical affine
 *      standard words + infinity flag, and (optionally, may be NULL) the
 *      Jacobian Montgomery words.  Return 0 on success, <0 on bad args. ------ */
int oracle_msm_naive(const uint32_t *bases, uint32_t form, const uint8_t *inf, const uint32_t *scalars,
                     size_t n, uint32_t out_xy_std[16], uint8_t *out_inf, uint32_t out_jac_mont[24]);
/* arkworks-0.4-algorithm restatement (CPU baseline). threads<=0 => all cores */
int oracle_msm_pippenger(const uint32_t *bases, uint32_t form, const uint8_t *inf, const uint32_t *scalars,
                         size_t n, int threads, uint32_t out_xy_std[16], uint8_t *out_inf,
                         uint32_t out_jac_mont[24]);
/* the reference's own cuZK staging (decompose -> transpose -> smvp -> pbpr -> Horner),
 * window_bits = 0 picks the reference's table (metal_msm.rs:661-673). */
int oracle_msm_cuzk(const uint32_t *bases, uint32_t form, const uint8_t *inf, const uint32_t *scalars,
                    size_t n, uint32_t window_bits, uint32_t out_xy_std[16], uint8_t *out_inf,
                    uint32_t out_jac_mont[24]);
int oracle_threads_available(void);

/* ---- stage mirrors of the reference kernels (for intermediate parity tests) */
uint32_t oracle_ref_window_bits(size_t n);              /* metal_msm.rs:661-673 */
uint32_t oracle_num_windows(uint32_t window_bits);      /* ceil(254 / w), metal_msm.rs:84-85 */
/* chunks[w*n + i] = digit + H in [0, 2H), reference K1 semantics */
void oracle_decompose_signed(const uint32_t *scalars, size_t n, uint32_t window_bits, uint32_t *chunks);
/* per window: col_ptr[w*(C+1) + v], val_idxs[w*n + k], stable (reference K2 semantics) */
void oracle_transpose(const uint32_t *chunks, size_t n, uint32_t num_windows, uint32_t num_cols,
                      uint32_t *col_ptr, uint32_t *val_idxs);

/* bucket sums from signed digits (SMVP sign folding, smvp.metal:46-105; bucket b = digit magnitude b + 1) and the bit sums of
 * the buckets (the plain sums the engine's reduction forms instead of pbpr.metal:33-148's running sums); Jacobian Montgomery */
int oracle_bucket_sums(const uint32_t *bases, uint32_t form, const uint8_t *inf, const int32_t *digits, size_t n, uint32_t W,
                       uint32_t nb, uint32_t *out_jac);
int oracle_bit_sums(const uint32_t *buckets_jac, uint32_t W, uint32_t nb, uint32_t kb, uint32_t *out_jac);

/* ---- deterministic synthetic inputs (SplitMix64), used by tests ----------- */
/* k[i] uniform in [1, r), 8 words each */
void oracle_gen_scalars(uint64_t seed, size_t n, int nonzero, uint32_t *out);
/* P_i = k_i * G as affine Montgomery (form=1) or standard (form=0) words */
void oracle_gen_bases_from_logs(const uint32_t *k, size_t n, uint32_t form, uint32_t *out_xy);

/* ---- arkworks 0.4 compressed G1Affine images (the reference's instance files: utils/preprocess.rs:193-223 write with
 * `serialize_compressed`, 101-131 / 225-256 read).  ark-serialize 0.4.x / ark-ec 0.4.x are third-party crates absent from
 * /root/reference; their published format is restated: 32 bytes = x (standard form, little-endian), bit 255 =
 * SWFlags::YIsNegative (y > p - y as integers), bit 254 = SWFlags::PointAtInfinity; both set = invalid.
 * The reference commits no such file, so this format is UNPINNED by fixtures (stated in DESIGN.md). */
/* returns 0, or 1 + index of the first invalid image; out_xy: n x 16 words in `form`; out_inf: n bytes */
size_t oracle_g1_decompress(const uint8_t *compressed, size_t n, uint32_t form, uint32_t *out_xy, uint8_t *out_inf);
void oracle_g1_compress(const uint32_t *bases_xy, uint32_t form, const uint8_t *inf, size_t n, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
