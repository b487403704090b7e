/*
 * msm_hip.h -- C ABI of the MI355X-native BN254 G1 variable-base MSM engine.
 *
 * This is the drop-in boundary for ONE path of zkmopro/gpu-acceleration (mopro-msm v0.2.0):
 *
 *     pub fn metal_variable_base_msm(bases: &[G1Affine], scalars: &[Fr])
 *         -> Result<G1Projective, Box<dyn Error>>          mopro-msm/src/msm/metal_msm/metal_msm.rs:642-695
 *
 * A Rust shim (rust/mopro-msm-hip) keeps that signature and forwards to
 * msm_bn254_g1() below.  Everything the reference does between that call and its return value --
 * MetalMSMPipeline::new / execute_pipeline / final_reduction (metal_msm.rs:48-261), the Metal
 * dispatch layer (host/metal_wrapper.rs:55-217, host/shader_manager.rs:98-167, host/gpu.rs:3-31) and the
 * five kernels (shader/cuzk/{convert_point_coords_and_decompose_scalars,transpose,smvp,pbpr}.metal) -- is
 * replaced by the implementation behind this header.
 *
 * Word formats (pinned by the reference's packer, utils/limbs_conversion.rs:311-378):
 *   field element  = 8 little-endian uint32 words
 *   affine base    = x[8] || y[8]                         (16 words, 64 bytes)
 *   scalar         = 8 words, standard (non-Montgomery) form, value < r
 *   Jacobian point = X[8] || Y[8] || Z[8], Montgomery form (R = 2^256); identity <=> Z == 0
 * MSM_FORM_MONT coordinates are bit-identical to arkworks' internal `Fq.0.0` limbs, so the shim
 * can hand them over without the 3N CPU-side Montgomery reductions of pack_affine_and_scalars.
 *
 * Determinism.  out_affine_std (and out_is_inf) are CANONICAL: the same inputs give the same bits on every call, every entry point, every
 * plan and any number of GPUs -- this is what "bit-exact against arkworks" means here and what every parity test compares.
 * out_jacobian_mont -- the reference's result type, G::new(x, y, z) (metal_msm.rs:228-241) -- is a projective REPRESENTATIVE of that group
 * element: X : Y : Z depends on the order in which the points of a bucket were added, and the counting sort places the entries of a bucket
 * with LDS atomics, so the 24 words generally DIFFER BETWEEN TWO IDENTICAL CALLS (arkworks compares projectively: `==` on G1Projective
 * holds).  The reference's own sort is serial (shader/cuzk/transpose.metal:8-65), so its limbs repeat; a caller that hashes or memcmp's
 * the Jacobian words must either use out_affine_std or create the context with MSM_FLAG_DETERMINISTIC, which returns the Z = 1
 * representative.  (A deterministic PLACEMENT -- every bucket's run sorted by point index in the fine sort's LDS staging -- was measured in
 * round 5 and is not the default: profiles/NOTES_r5.md.)
 *
 * Threading: a context serialises its own calls with an internal mutex; different contexts are
 * independent.  No caller pointer is retained after a call returns.  Nothing aborts or panics:
 * every failure is a negative status plus msm_last_error().
 */
#ifndef MSM_HIP_H
#define MSM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A consumer must be built against the header of the library it loads: check msm_abi_version() == MSM_HIP_ABI_VERSION once after loading
 * (the Rust shim and the Python binding do) -- msm_timings_t and msm_config_t have grown with the ABI number, and the plain getters write the
 * whole struct of THEIR build.  ABI 7: msm_bn254_g1_combine_flags, msm_get_timings_sized / msm_multi_get_timings_sized, msm_multi_get_clock_stats, msm_set_kernel_timing. */
#define MSM_HIP_ABI_VERSION 7u

/* status codes */
#define MSM_OK 0
#define MSM_ERR_EMPTY (-1)      /* n == 0: reference returns Err("Empty input"), metal_msm.rs:647-649 */
#define MSM_ERR_BAD_ARG (-2)    /* NULL pointer, bad form/flags, scalar >= 2^254, window out of range   */
#define MSM_ERR_NO_DEVICE (-3)  /* no HIP device / extension unusable: the product never falls back to CPU */
#define MSM_ERR_HIP (-4)        /* a HIP runtime call failed; see msm_last_error()                      */
#define MSM_ERR_OOM (-5)
#define MSM_ERR_STATE (-6)      /* e.g. resident call without uploaded bases                            */
#define MSM_ERR_INVALID_DATA (-7) /* a compressed point does not decode (arkworks SerializationError::InvalidData) */
#define MSM_ERR_RCCL (-8)       /* an RCCL call of the multi-GPU exchange failed; see msm_multi_last_error()            */

/* coordinate form of the bases handed in */
#define MSM_FORM_STD 0u  /* plain integers < p   (what pack_affine_and_scalars emits)          */
#define MSM_FORM_MONT 1u /* x*2^256 mod p        (arkworks Fq.0 as is)                          */

/* msm_config_t.flags */
#define MSM_FLAG_UNSIGNED_DIGITS 1u /* plain radix-2^c digits, 2^c-1 buckets/window (BASELINE config "fixed 16-bit window") */
#define MSM_FLAG_NO_GLV 2u          /* do not split scalars with the curve endomorphism (csrc/glv_bn254.hpp); the default splits
                                       k = k1 + lambda*k2, |k_j| < 7*2^123: 2N points phi-extended, half the windows        */

#define MSM_FLAG_WINDOW_TABLE 4u     /* msm_bn254_g1_upload_bases / _upload_compressed also precompute the
                                       WINDOW TABLE of the resident set, T_j[i] = 2^(c*j) P_i for every window j (c doublings and one
                                       inversion per record, once per upload), and msm_bn254_g1_resident / _resident_batch on the whole
                                       set add a digit of window j as +-T_j[i] into ONE bucket array shared by all windows: the buckets are
                                       reduced once per MSM instead of once per window, which moves the optimum to wider windows (c = 20
                                       at 2^20 points: 13 windows instead of 16, 19 % fewer additions, a 19-position host chain instead of
                                       254).  Memory: W x 64 bytes per (virtual) point -- 872 MB at 2^20 -- see msm_plan_t.table_bytes; build 28 ms at
                                       2^20 (what ~300 MSMs gain: for base sets that outlive many calls, e.g. a proving key); none above 2^21.
                                       Every other entry point, and a resident call on fewer scalars than bases, is unaffected.
                                       Results are the same group element either way (bit-exact affine coordinates).            */

#define MSM_FLAG_DETERMINISTIC 8u     /* out_jacobian_mont is the CANONICAL representative of the result: (x*R, y*R, R), the identity (R, R, 0) -- the
                                       same 24 words for the same group element.  Without it the words are A representative that MAY DIFFER
                                       BETWEEN IDENTICAL CALLS (see "Determinism" below).  Costs one field inversion on the host (~10 us;
                                       shared with out_affine_std when both are asked for).  ABI 6.                              */

typedef struct msm_ctx msm_ctx;

typedef struct {
    int32_t device;       /* HIP device ordinal; -1 = current device                                   */
    uint32_t window_bits; /* c; 0 = planner (replaces the N->window table at metal_msm.rs:661-673)     */
    uint32_t flags;       /* MSM_FLAG_*                                                                */
    uint32_t stream_chunk_log2; /* the host-pointer entries cut n >= 2 * 2^this points into chunks of 2^this points: chunk j+1
                                   travels host->HBM (copy stream) while chunk j is sorted and accumulated INTO the shared bucket
                                   array; one bucket reduction and one host finish per MSM (BASELINE config 5).
                                   0 = automatic: from 2^19 points on, uniform chunks of 2^18..2^20 points (a remainder below half a chunk
                                   joins the last one).
                                   Copies run on a stream with a hardware queue of its own: from pinned caller memory at the link
                                   rate, from pageable memory as fast as the runtime stages it (both overlap the kernels) */
    uint64_t max_points;  /* pre-size the HBM workspace for this many points; 0 = grow on demand        */
    uint32_t batch_layout; /* MSM_BATCH_LAYOUT_*: how msm_bn254_g1_resident_batch shares the GPU between its two pipelines (ABI 5) */
    uint32_t host_threads; /* ABI 6 (was `reserved`): CPU threads of the host finish (the Horner chain over the bit sums), the caller included.
                              0 = default (2: the caller + one worker -- every further worker lowers the median by microseconds and raises the
                              mean through 2-8 ms outliers on busy hosts); 1 = the calling thread alone; at most 64 */
} msm_config_t;

/* msm_config_t.batch_layout.  The layout of a batch call is a pure function of (this field, the context's tuned choice, n): nothing
 * is timed behind the caller's back (the reference's whole configuration surface is one struct, metal_msm.rs:16-28).
 *   ONE_STREAM         both pipelines queue whole MSMs on ONE compute stream, MSM after MSM without a gap; uploads and host finishes
 *                      happen beside it.  Needs nothing from HIP's stream -> hardware-queue mapping: the layout that stays within ~12 %
 *                      of its best whatever other contexts and streams the process has created.
 *   ONE_STREAM_REDUCE  the same, and each MSM's bucket reduction runs on a second, high-priority stream beside the next MSM's sort:
 *                      3-6 % faster per MSM from 2^19 points when the three streams get hardware queues of their own -- and 25-50 % SLOWER
 *                      when they do not (three or four other contexts alive: profiles/r3_batch_many_contexts.txt).
 *   TWO_STREAMS        each pipeline keeps its own compute stream and the kernels overlap: best below 2^19 points, where no kernel fills
 *                      the GPU.
 *   AUTO (0)           ONE_STREAM from 2^19 points, TWO_STREAMS below -- unless msm_tune_batch() has measured this context. */
#define MSM_BATCH_LAYOUT_AUTO 0u
#define MSM_BATCH_LAYOUT_ONE_STREAM 1u
#define MSM_BATCH_LAYOUT_ONE_STREAM_REDUCE 2u
#define MSM_BATCH_LAYOUT_TWO_STREAMS 3u

typedef struct {
    uint32_t window_bits;  /* c                                         */
    uint32_t num_windows;  /* W                                         */
    uint32_t num_buckets;  /* buckets per window (2^(c-1) signed, 2^c unsigned incl. an unused slot) */
    uint32_t signed_digits;
    uint64_t workspace_bytes;
    uint64_t virtual_points; /* points each window sorts and accumulates: n, or 2n with the GLV split       */
    uint32_t glv;            /* 1 = scalars are split with the endomorphism                                  */
    uint32_t scalar_bits;    /* bits the windows cover: 254, or 127 with GLV                                 */
    uint32_t table_factor;   /* f: window-table levels per base (MSM_FLAG_WINDOW_TABLE); 1 = no table        */
    uint32_t bucket_arrays;  /* num_windows / table_factor bucket arrays of num_buckets buckets each         */
    uint64_t table_bytes;    /* HBM the window table of the resident set takes (0 without one)               */
    uint32_t top_digit_bits; /* the TOP window's bucket index holds the digit magnitude - 1 in its low top_digit_bits bits; when that is
                                less than log2(num_buckets) -- the top window only holds 254 - c*(W-1) bits (split plans: the halves are
                                below 2^125.81) -- the bits above hold low bits of the point index, so that its buckets are as full as
                                every other window's (csrc/msm_planner.hpp)                                    */
    uint32_t reserved;
} msm_plan_t;

/* per-stage device times of the last call on this context, milliseconds (hipEvent) */
typedef struct {
    float h2d_ms;        /* host->HBM copies (0 for device-resident calls)      */
    float convert_ms;    /* bases to Montgomery / internal layout               */
    float decompose_ms;  /* scalar windowing + signed digits + bucket histogram */
    float sort_ms;       /* bucket offsets (scan) + scatter of point indices    */
    float accumulate_ms; /* per-bucket point accumulation -- the graded kernel  */
    float reduce_ms;     /* bucket reduction (row / column sums, bit sums).  Until ABI 5 this field also held what is now combine_ms */
    float finish_ms;     /* host Horner over the bit sums (+ normalisation if asked) */
    float total_ms;      /* wall clock of the whole call                        */
    uint64_t num_points;
    uint64_t num_adds;   /* mixed additions executed by accumulate (non-zero digits) */
    uint32_t stream_chunks; /* host->HBM chunks the call was cut into (0 = single shot / device-resident)      */
    uint32_t batch_layout;  /* MSM_BATCH_LAYOUT_* the last msm_bn254_g1_resident_batch call of this context ran under (0 before a batch call) */
    float plan_ms;       /* ABI 6: the accumulation's work-item plan (k_piece_count + k_piece_scatter: whole buckets sorted by length), between
                            sort_ms and accumulate_ms -- work the round-3 accumulation kernel did itself                              */
    float combine_ms;    /* ABI 6: k_combine_pieces (buckets cut into pieces), between accumulate_ms and reduce_ms; was counted in reduce_ms */
} msm_timings_t;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* replaces MetalMSMPipeline::with_default_config() (metal_msm.rs:64, rebuilt on EVERY call there) */
int32_t msm_ctx_create(const msm_config_t *cfg /* NULL = defaults */, msm_ctx **out);
void msm_ctx_destroy(msm_ctx *ctx);
/* message of the last failure on ctx (or of the last failed msm_ctx_create when ctx == NULL) */
const char *msm_last_error(const msm_ctx *ctx);
uint32_t msm_abi_version(void);

/* ---- the drop-in call: host pointers in, host results out ----------------------------------
 * Round 6: pageable caller memory is PINNED IN PLACE for the duration of a host-pointer call (hipHostRegister; msm_bn254_g1, _arkworks, _resident(_batch),
 * _upload_bases, the msm_multi forms) and unregistered when the call has consumed it: the copies run at the link rate (53-55 GB/s instead of the ~40 of
 * staged pageable copies -- 2^20 points 2.55 -> 2.45 ms) and registration costs microseconds on this platform.  Memory the caller pinned itself is left alone;
 * a range that cannot be registered travels as before; registrations are reference-counted across contexts and threads of the process. */
/* replaces metal_variable_base_msm (metal_msm.rs:642-695).  bases: n x 16 words; inf_mask: n bytes
 * (non-zero = point at infinity, arkworks G1Affine.infinity) or NULL; scalars: n x 8 words.
 * Any of the three outputs may be NULL.  out_jacobian_mont is the reference's own result type (G::new(x, y, z),
 * metal_msm.rs:228-241): a representative that may differ between identical calls unless the context has MSM_FLAG_DETERMINISTIC.  out_affine_std = canonical affine, standard form (0,0 when the result is the identity and
 * *out_is_inf = 1); it costs the call's only field inversion (~10 us on the host) -- pass NULL to skip it and normalise
 * later with msm_bn254_g1_combine(partial, 1, ...) if needed. */
int32_t msm_bn254_g1(msm_ctx *ctx, const uint32_t *bases_xy, uint32_t base_form, const uint8_t *inf_mask,
                     const uint32_t *scalars, size_t n, uint32_t out_jacobian_mont[24],
                     uint32_t out_affine_std[16], uint8_t *out_is_inf);

/* ---- zero-copy arkworks ingestion ------------------------------------------
 * bases: array of n `ark_bn254::G1Affine` structs exactly as they sit in memory, described by (stride, byte offsets of
 * x, y and the `infinity` bool; inf_off = (size_t)-1 if there is none) -- the Rust shim measures these with
 * size_of / addr_of!, the struct is not repr(C).  Coordinates and scalars are arkworks' internal Montgomery words
 * (Fq.0 / Fr.0, R = 2^256): no CPU-side `into_bigint()`, no repacking (replaces pack_affine_and_scalars,
 * utils/limbs_conversion.rs:311-378: 3 Montgomery reductions + 3 heap allocations per point).
 * Round 6: the call first runs as if no struct had its `infinity` flag set (the words are repacked on the device and gathered as they are, the sort
 * overlaps the transfer of the bases); if one does, the call is repeated internally with the flags as an infinity mask -- same result, about twice
 * the time for such inputs (proving keys and SRS points are never at infinity). */
int32_t msm_bn254_g1_arkworks(msm_ctx *ctx, const void *bases, size_t stride, size_t x_off, size_t y_off, size_t inf_off,
                              const uint32_t *scalars_mont, size_t n, uint32_t out_jacobian_mont[24],
                              uint32_t out_affine_std[16], uint8_t *out_is_inf);

/* ---- bases resident in HBM ------------------------------------- */
int32_t msm_bn254_g1_upload_bases(msm_ctx *ctx, const uint32_t *bases_xy, uint32_t base_form,
                                  const uint8_t *inf_mask, size_t n);
int32_t msm_bn254_g1_resident(msm_ctx *ctx, const uint32_t *scalars, size_t n, uint32_t out_jacobian_mont[24],
                              uint32_t out_affine_std[16], uint8_t *out_is_inf);
/* the same with the scalars already in HBM (ABI 4): d_scalars = n x 8 words of device memory, e.g. the witness of a prover whose
 * earlier stages ran on the GPU; hip_stream = the hipStream_t that produced them (NULL = the context's stream).  Nothing crosses
 * PCIe but the 96-byte result, the bases are not converted again (msm_bn254_g1_device converts its bases on every call), and a
 * context created with MSM_FLAG_WINDOW_TABLE uses the table: 2^20 points 1.28 ms against the 1.41 of msm_bn254_g1_device in the same run (1.35 without the table).
 * Blocks until the result is on the host. */
int32_t msm_bn254_g1_resident_device(msm_ctx *ctx, const void *d_scalars, size_t n, void *hip_stream,
                                     uint32_t out_jacobian_mont[24], uint32_t out_affine_std[16], uint8_t *out_is_inf);

/* `count` MSMs against the same resident bases, TWO in flight: scalars[i] = n x 8 words (host), results out_jacobian_mont[i*24..],
 * out_affine_std[i*16..] (nullable), out_is_inf[i] (nullable).  This is how provers call MSM: several scalar vectors per proof against
 * fixed bases.  A second pipeline inside the context (second host thread, own workspace) uploads the next
 * scalar vector and finishes the previous MSM on the CPU while the GPU computes: from 2^19 points both pipelines feed ONE compute
 * stream, MSM after MSM without a gap (each MSM's bucket reduction on a second, high-priority stream beside the next MSM's sort);
 * below, where no kernel fills the GPU, each keeps its own stream and the kernels overlap.  Which of the three layouts a call runs
 * under is decided by msm_config_t.batch_layout (see MSM_BATCH_LAYOUT_*), deterministically; msm_get_timings().batch_layout reports it.
 * n is clamped to the resident set; a window table (MSM_FLAG_WINDOW_TABLE) serves calls on the WHOLE set only.
 * Per MSM, single calls -> batch (round 4): 2^14 0.32 -> 0.21 ms, 2^17 0.50 -> 0.35, 2^20 2.1 -> 1.49, 2^22 7.75 -> 5.80.
 * Results are identical to `count` msm_bn254_g1_resident calls; on an error the first failing code is returned. */
int32_t msm_bn254_g1_resident_batch(msm_ctx *ctx, const uint32_t *const *scalars, size_t n, size_t count,
                                    uint32_t *out_jacobian_mont, uint32_t *out_affine_std, uint8_t *out_is_inf);
/* Explicit, opt-in measurement of the batch layout (ABI 5): runs the batch `scalars[0..count)` (count >= 2; results discarded) under each
 * of the three layouts on THIS context as the process is now -- once untimed (a layout's first use creates hardware queues and the
 * second pipeline's workspace), then `reps` (0 = 3) timed batches, the minimum counts -- and keeps the fastest for this size class
 * (below / from 2^19 points) until the next msm_bn254_g1_upload_bases / _upload_compressed; AUTO then uses it.  *chosen (nullable)
 * receives the layout, ms_per_msm[3] (nullable) the three measurements in the order ONE_STREAM, ONE_STREAM_REDUCE, TWO_STREAMS.
 * A context whose msm_config_t.batch_layout is not AUTO keeps its configured layout (the measurements are still returned). */
int32_t msm_tune_batch(msm_ctx *ctx, const uint32_t *const *scalars, size_t n, size_t count, uint32_t reps, uint32_t *chosen,
                       double *ms_per_msm);

/* ---- arkworks `serialize_compressed` point images ---------------
 * The reference's benchmark harness keeps its instances on disk as `Vec<G1Affine>::serialize_compressed`
 * (mopro-msm/src/msm/utils/preprocess.rs:193-223 writes, 101-131 / 225-256 read): per point 32 bytes = x in
 * standard form, little-endian, with bit 255 = "y is the larger of (y, p-y)" and bit 254 = point at infinity
 * (ark-ec 0.4 SWFlags).  Reading them back costs arkworks one Fq square root per point on the CPU; here the
 * square root y = (x^3+3)^((p+1)/4) runs on the GPU, one thread per point.
 * compressed: n x 32 bytes (host).  An image with both flag bits set, x >= p, or x^3+3 a non-residue fails the
 * whole call with MSM_ERR_INVALID_DATA and *first_invalid (nullable) = the lowest such index.               */
/* decode to arkworks Montgomery words (R = 2^256): out_xy_mont n x 16 words, out_inf n bytes (both host)   */
int32_t msm_bn254_g1_decompress(msm_ctx *ctx, const uint8_t *compressed, size_t n, uint32_t *out_xy_mont,
                                uint8_t *out_inf, int64_t *first_invalid);
/* decode straight into the resident-bases set (then msm_bn254_g1_resident); nothing returns to the host   */
int32_t msm_bn254_g1_upload_compressed(msm_ctx *ctx, const uint8_t *compressed, size_t n, int64_t *first_invalid);
/* the inverse, on the host (no context, no GPU): bases as for msm_bn254_g1 -> n x 32 bytes                */
int32_t msm_bn254_g1_compress(const uint32_t *bases_xy, uint32_t base_form, const uint8_t *inf_mask, size_t n,
                              uint8_t *out_compressed);

/* ---- everything already in HBM (what bench.py times) ---------------------------------------- */
/* d_bases_mont: n x 16 words, Montgomery form, device memory; d_inf_mask: n bytes device memory or NULL;
 * d_scalars: n x 8 words device memory.  hip_stream: a hipStream_t (NULL = the context's stream).
 * Blocks until the result is on the host. */
int32_t msm_bn254_g1_device(msm_ctx *ctx, const void *d_bases_mont, const void *d_inf_mask, const void *d_scalars,
                            size_t n, void *hip_stream, uint32_t out_jacobian_mont[24],
                            uint32_t out_affine_std[16], uint8_t *out_is_inf);

/* ---- multi-GPU: fold per-rank partial results (host arithmetic, like final_reduction
 *      metal_msm.rs:204-261 runs on the CPU).  partials: k x 24 words Jacobian Montgomery. -------- */
int32_t msm_bn254_g1_combine(const uint32_t *partials_jacobian_mont, size_t k, uint32_t out_jacobian_mont[24],
                             uint32_t out_affine_std[16], uint8_t *out_is_inf);
/* the same with the representative chosen by the caller (ABI 7): flags = 0 behaves as msm_bn254_g1_combine; MSM_FLAG_DETERMINISTIC hands out the
 * Z = 1 representative, i.e. the words a single context or msm_multi with that flag returns for the same group element -- what a
 * one-process-per-GPU job folds its ranks' partials with when its contexts carry the flag (mopro_msm_hip.distributed.all_reduce_msm).
 * Any other flag bit: MSM_ERR_BAD_ARG. */
int32_t msm_bn254_g1_combine_flags(const uint32_t *partials_jacobian_mont, size_t k, uint32_t flags, uint32_t out_jacobian_mont[24],
                                   uint32_t out_affine_std[16], uint8_t *out_is_inf);

/* ---- multi-GPU inside ONE process (the reference has no multi-device code: host/gpu.rs:3-5 opens the
 *      system default device).  The caller-facing signature is the same as the single-GPU calls; the point range is cut
 *      into `ndev` contiguous shards, device g pulls ITS OWN shard over its own PCIe link (one host thread + one context
 *      per device), runs the whole single-GPU pipeline on it, and the 96-byte partial group elements are exchanged:
 *        MSM_MULTI_EXCHANGE_RCCL  ncclAllGather of 24 words per rank over RCCL/xGMI (librccl is dlopen'ed: no link-time
 *                                 dependency), every rank folds in rank order, rank 0's bits are returned;
 *        MSM_MULTI_EXCHANGE_HOST  the partials already sit in pinned host memory: the calling thread folds them.
 *      AUTO (ABI 6): when RCCL is possible (librccl loads, ndev > 1, no device listed twice) msm_multi_create MEASURES both exchanges once --
 *      a few calls on identity partials, the same code path the calls take -- and keeps the faster one for the handle's life
 *      (msm_multi_exchange() says which, msm_multi_get_exchange_probe() what was measured); otherwise HOST.  Each rank's partial is in host
 *      memory when its local call returns, so the RCCL round trip is by construction more work than the host fold; rounds 2-4 preferred it
 *      unmeasured.  (One process PER GPU -- torchrun, mopro_msm_hip.distributed -- always exchanges over RCCL.)  The two exchanges fold
 *      the same partials in the same rank order: identical out_affine_std bits either way, and identical out_jacobian_mont for identical
 *      PARTIALS -- which, like every Jacobian result, repeat between calls only under MSM_FLAG_DETERMINISTIC (see "Determinism").  A device may be listed more than once (tests on a 1-GPU box: {0, 0}). ---------- */
#define MSM_MULTI_EXCHANGE_AUTO 0u
#define MSM_MULTI_EXCHANGE_RCCL 1u
#define MSM_MULTI_EXCHANGE_HOST 2u
typedef struct msm_multi msm_multi;
/* devices == NULL: all visible devices (ndev ignored), or the comma-separated list in MSM_HIP_DEVICES.
 * cfg->device is ignored; the other fields apply to every per-device context.  NULL cfg = defaults. */
int32_t msm_multi_create(const int32_t *devices, int32_t ndev, const msm_config_t *cfg, uint32_t exchange, msm_multi **out);
void msm_multi_destroy(msm_multi *m);
const char *msm_multi_last_error(const msm_multi *m); /* m == NULL: last failed msm_multi_create on this thread */
int32_t msm_multi_num_devices(const msm_multi *m);
uint32_t msm_multi_exchange(const msm_multi *m);      /* MSM_MULTI_EXCHANGE_RCCL or _HOST: what the calls really use */
/* what AUTO measured at creation (ABI 6): ms per exchange on identity partials, best of five; both 0 when nothing was probed (an explicit
 * mode, one device, a duplicated device, no librccl).  Either pointer may be NULL. */
int32_t msm_multi_get_exchange_probe(const msm_multi *m, float *rccl_ms, float *host_ms);
/* same arguments and semantics as msm_bn254_g1 / msm_bn254_g1_arkworks, host pointers */
int32_t msm_bn254_g1_multi(msm_multi *m, const uint32_t *bases_xy, uint32_t base_form, const uint8_t *inf_mask,
                           const uint32_t *scalars, size_t n, uint32_t out_jacobian_mont[24],
                           uint32_t out_affine_std[16], uint8_t *out_is_inf);
int32_t msm_bn254_g1_multi_arkworks(msm_multi *m, const void *bases, size_t stride, size_t x_off, size_t y_off,
                                    size_t inf_off, const uint32_t *scalars_mont, size_t n,
                                    uint32_t out_jacobian_mont[24], uint32_t out_affine_std[16], uint8_t *out_is_inf);
/* shards already resident: d_bases_mont[g] / d_scalars[g] / d_inf_masks[g] (nullable array, nullable entries) are device
 * pointers on device g holding n_per_dev[g] points (as for msm_bn254_g1_device); entries with n_per_dev[g] == 0 are skipped */
int32_t msm_bn254_g1_multi_device(msm_multi *m, const void *const *d_bases_mont, const void *const *d_inf_masks,
                                  const void *const *d_scalars, const size_t *n_per_dev,
                                  uint32_t out_jacobian_mont[24], uint32_t out_affine_std[16], uint8_t *out_is_inf);
/* timings of device g's part of the last call */
int32_t msm_multi_get_timings(const msm_multi *m, int32_t g, msm_timings_t *out);
/* what the exchange cost in the last call: *exchange_ms = rank 0's wall clock from the end of its local MSM to the folded result
 * (RCCL: the rendezvous with the slowest rank, 96 B up, ncclAllGather, 96*ndev B down, fold; host fold: the fold alone);
 * shard_ms[0..nshard) = wall clock of every rank's local MSM.  Either pointer may be NULL.
 * Error behaviour of the multi calls: a rank whose local MSM fails never leaves its peers waiting in the collective -- the ranks
 * rendezvous on the host first and ALL skip the exchange; the call returns the first failing rank's status, and
 * msm_multi_last_error() names the device and rank (metal_msm.rs:647-656: errors are returned, nothing hangs). */
int32_t msm_multi_get_exchange_stats(const msm_multi *m, float *exchange_ms, float *shard_ms, int32_t nshard);
int32_t msm_multi_get_timings_sized(const msm_multi *m, int32_t rank, void *out, size_t out_size); /* ABI 7, see msm_get_timings_sized */
int32_t msm_multi_set_kernel_timing(msm_multi *m, uint32_t every_n); /* ABI 7: msm_set_kernel_timing on every rank's context */
/* msm_get_clock_stats of rank `rank`'s context (ABI 7): a slow rank of a multi-GPU call can be told from a slow CLOCK on its device */
int32_t msm_multi_get_clock_stats(msm_multi *m, int32_t rank, double *sclk_ghz, double *cycles_per_addition, uint64_t *samples);

/* ---- introspection --------------------------------------------------------------------------- */
/* the plan of a call on n points under (window_bits, flags); with MSM_FLAG_WINDOW_TABLE in flags: the plan of a RESIDENT call on a
 * set of n bases uploaded under those flags (window width, table factor, table memory) */
int32_t msm_plan(size_t n, uint32_t window_bits, uint32_t flags, msm_plan_t *out);
int32_t msm_get_timings(const msm_ctx *ctx, msm_timings_t *out);
/* size-checked forms (ABI 7): at most out_size bytes are written -- a consumer built against an older, shorter msm_timings_t passes ITS
 * sizeof and is not overrun; fields beyond the library's struct are left untouched.  out_size == 0: MSM_ERR_BAD_ARG. */
int32_t msm_get_timings_sized(const msm_ctx *ctx, void *out, size_t out_size);
/* per-stage hipEvents are OFF by default (every record costs ~6 us of stream time); when off, the stage fields of
 * msm_timings_t other than accumulate_ms / finish_ms / total_ms read 0 */
int32_t msm_set_stage_timing(msm_ctx *ctx, int32_t enabled);
/* Which launches of the accumulate kernel carry their pair of hipEvents (ABI 7): every `every_n`-th one; 0 = none (the DEFAULT since ABI 7), 1 = all of them (what ABI <= 6 did).
 * The events ride on the kernel's dispatch, but a timed dispatch does not overlap its neighbours' launch latency: ~11 us per MSM at every size.  A caller
 * that reads msm_get_accumulate_kernel_stats / msm_timings_t.accumulate_ms turns them on; bench.py samples every 4th launch of its timed loop.
 * Stage timing (msm_set_stage_timing) times every launch whatever this says.  msm_timings_t.accumulate_ms reads 0 after an untimed launch. */
int32_t msm_set_kernel_timing(msm_ctx *ctx, uint32_t every_n);
/* average duration (ms) of the TIMED accumulate kernel launches since the last reset, measured with
 * hipEvents on the stream the kernel runs on; *launches receives their count */
int32_t msm_get_accumulate_kernel_stats(const msm_ctx *ctx, double *avg_ms, uint64_t *launches);
void msm_reset_kernel_stats(msm_ctx *ctx);
/* Clock probe of the accumulate kernel since the last reset (ABI 5): the first workgroup of every launch brackets its chunk with the
 * shader-cycle counter and the constant-rate counter.  *sclk_ghz = the shader clock the kernel really sustained (cycles / ticks x
 * hipDeviceAttributeWallClockRate); *cycles_per_addition = shader cycles that wavefront needed per mixed addition (three wavefronts
 * share a SIMD): equal on two boxes that execute the same instruction stream, whatever their clocks; *samples = launches sampled.
 * Any pointer may be NULL.  Synchronises the context's stream. */
int32_t msm_get_clock_stats(msm_ctx *ctx, double *sclk_ghz, double *cycles_per_addition, uint64_t *samples);

#ifdef __cplusplus
}
#endif
#endif /* MSM_HIP_H */
