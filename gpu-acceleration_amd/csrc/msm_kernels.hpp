// msm_kernels.hpp -- the gfx950 kernels of the BN254 G1 MSM pipeline.
//
// Stage map against the reference (shader/cuzk/*.metal):
//   K1 convert_point_coords_and_decompose_scalars -> coordinates: NOTHING for arkworks words (round 5: k_accumulate_pieces<.., M256> gathers them as they
//        are; a split plan's phi records: k_decompose_glv<.., PHI> / k_phi_records), k_convert_bases for standard-form words (one Montgomery product
//        per coordinate instead of two Barrett multiplications), k_import_ark / k_decompress for the other input forms; scalars: k_decompose(_glv)
//   K2 transpose (ONE thread per window, serial 2N+C loop) -> two-level counting sort in LDS: k_coarse_hist, k_coarse_prefix,
//        k_coarse_starts, k_coarse_scatter, k_fine_sort (fallbacks: k_tile_* for n > 2^24, device-scope atomics in
//        k_decompose + k_scan_* + k_scatter for windows of more than 2^17 buckets)
//   K3 smvp (full 16-mul Jacobian add, one thread per bucket pair) -> k_piece_count + k_piece_scatter (work items = whole buckets,
//        sorted by length), k_accumulate_pieces (XYZZ mixed add, one thread per piece), k_combine_pieces (buckets split by skew)
//   K4/K5 bpr_stage_1/2 -> k_pair_level / k_pair_level_wide (row/column plain sums, dense pairwise levels) +
//        k_reduce_bits_wide (per-bit sums, LDS trees of eight-lane additions; k_reduce_bits = the one-lane fallback)
//   final_reduction (CPU) -> stays on the CPU: host_g1.hpp
//
// Data layout in HBM (all little-endian u32 words):
//   bases    n x 16   affine x||y, 64 B per point = half a cache line per gather: the caller's arkworks words (R = 2^256 Montgomery; M256 kernels)
//                     or the INTERNAL Montgomery domain (x*2^261 mod p, canonical, packed 8 x 32-bit words per coordinate by k_convert_bases:
//                     standard-form / struct / compressed inputs, resident sets and their window tables)
//   scalars  n x 8    standard form
//   digits   W x n    bucket index (bit 31 = negate, 0xFFFFFFFF = digit 0 / base at infinity), window-major; as 16-bit codes (bit 15, 0xFFFF) where a
//                     window has <= 2^15 buckets and the two-level sort runs (DIGIT16_* below)
//   ranks    W x n    arrival rank inside the bucket (fallback path only: more than 2^17 buckets per window)
//   offsets  W*nb + 1 exclusive prefix sum of bucket sizes == CSC column pointer of the reference
//   sorted   W x n    point index | sign<<31 grouped by bucket        == val_idxs of the reference
//   buckets  W*nb x 36  XYZZ bucket sums, 4 coordinates x 9 limbs of 29 bits (144 B);
//   plist    pieces x 4     (bucket, first sorted entry, length | flags, partial slot), longest pieces first
//   partials  x 36          partial sums of the pieces of buckets longer than the cap (mean + 2 sigma: msmplan::make_piece_plan)
#pragma once
#include "msm_planner.hpp"
#include "ec_bn254.hpp"
#include "ec_wide.hpp"
#include "glv_bn254.hpp"

namespace msmk {
using namespace bn254;

constexpr uint32_t DIGIT_SKIP = 0xFFFFFFFFu;
constexpr uint32_t SIGN_BIT = 0x80000000u;
constexpr int SCALAR_BITS = 254;

constexpr int XW = 36;  // words per XYZZ record in HBM: 4 coordinates x 9 limbs
// flags (u32 words, one set per context): [0] error bits  [4] sorted entries of this (chunk of an) MSM  [6,7] running 64-bit total of
// sorted entries over the chunks of a streamed MSM  [8] long-list entries  [9] mid-list entries  [10] pieces  [11] partial-sum slots
constexpr uint32_t FLAG_ERR = 0, FLAG_PAIRS = 4, FLAG_ADDS64 = 6, FLAG_LONG = 8, FLAG_MID = 9, FLAG_PIECES = 10, FLAG_PARTIALS = 11, FLAG_MID2 = 12, FLAG_NONEMPTY = 13;

// 16-BIT DIGIT CODES (round 5).  Where a window has at most 2^15 buckets and the two-level LDS sort runs (every default plan without a
// window table: c <= 16 signed), a digit travels from k_decompose to the two sort kernels that read it as 16 bits: bucket index in bits
// 0..14, bit 15 = negate, 0xFFFF = digit 0 / base at infinity.  0xFFFF is free: it would be bucket 2^15 - 1 NEGATED, i.e. the digit -2^15,
// and the signed recode never produces it (v > H gives magnitude 2H - v <= H - 1); the spread top window of a split plan holds magnitudes
// <= 7 * 2^(c-5) < 2^top_bits, so its spread code cannot end in all ones either (msmplan::digits16 spells the conditions out).
// Halves the decomposition's write and the two reads of the coarse histogram / scatter: 100 MB of the sort stage's 351 MB at 2^20.
// Rows of the digit array are padded to an even number of entries (digit_row_stride) so that a thread can load two digits as one word.
constexpr uint32_t PIECE_BINS = 1024;  // k_piece_count / k_piece_scatter: one histogram bin per piece length (== msmplan::PIECE_BINS_MAX)
constexpr uint32_t DIGIT16_SKIP = 0xFFFFu;
__host__ __device__ __forceinline__ uint32_t digit16_widen(uint32_t h) {
    return h == DIGIT16_SKIP ? DIGIT_SKIP : ((h & 0x7FFFu) | ((h & 0x8000u) << 16));
}
__host__ __device__ __forceinline__ uint32_t digit16_narrow(uint32_t d) {
    return d == DIGIT_SKIP ? DIGIT16_SKIP : ((d & 0x7FFFu) | ((d >> 16) & 0x8000u));
}
// store digit code d (32-bit form) as entry o of a digit array of the given width
template <bool D16>
__device__ __forceinline__ void digit_store(void* __restrict__ digits, size_t o, uint32_t d) {
    if (D16) reinterpret_cast<uint16_t*>(digits)[o] = (uint16_t)digit16_narrow(d);
    else reinterpret_cast<uint32_t*>(digits)[o] = d;
}
// The piece-sort histogram and its bin cursors (k_piece_count / k_piece_scatter) are zeroed at the HEAD of every (chunk of an) MSM, by the
// first workgroup of the decomposition and of the coarse histogram, next to the list counters: a sort never depends on the kernel that
// ran before it.  (Round 4 left this to workgroup 0 of k_accumulate_pieces, i.e. to the PREVIOUS call having completed: an error between
// a sort and its accumulation left them dirty for the next call -- ADVICE r4.)  PIECE_BINS + 1 words each.
__device__ __forceinline__ void clear_piece_bins(uint32_t* __restrict__ phist, uint32_t* __restrict__ pcursor) {
    for (uint32_t k = threadIdx.x; k <= PIECE_BINS; k += blockDim.x) phist[k] = 0, pcursor[k] = 0;
}

// packed 8-word field element (canonical value) -> 9 x 29-bit limbs
__device__ __forceinline__ fp load_fp_packed(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fp_unpack(w);
}
__device__ __forceinline__ void load_words8(uint32_t w[8], const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    w[0] = a.x, w[1] = a.y, w[2] = a.z, w[3] = a.w, w[4] = b.x, w[5] = b.y, w[6] = b.z, w[7] = b.w;
}
__device__ __forceinline__ void store_words8(uint32_t* p, const uint32_t w[8]) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ affine load_affine(const uint32_t* p) { return affine{load_fp_packed(p), load_fp_packed(p + 8)}; }
// XYZZ record: 36 words = 9 x 16 bytes
__device__ __forceinline__ xyzz load_xyzz(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint32_t w[XW];
#pragma unroll
    for (int i = 0; i < XW / 4; i++) {
        uint4 t = q[i];
        w[4 * i] = t.x;
        w[4 * i + 1] = t.y;
        w[4 * i + 2] = t.z;
        w[4 * i + 3] = t.w;
    }
    xyzz r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.v[i] = w[i];
        r.y.v[i] = w[9 + i];
        r.zz.v[i] = w[18 + i];
        r.zzz.v[i] = w[27 + i];
    }
    return r;
}
__device__ __forceinline__ void store_xyzz(uint32_t* p, const xyzz& v) {
    uint32_t w[XW];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        w[i] = v.x.v[i];
        w[9 + i] = v.y.v[i];
        w[18 + i] = v.zz.v[i];
        w[27 + i] = v.zzz.v[i];
    }
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < XW / 4; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// one coordinate (0 = X, 1 = Y, 2 = ZZ, 3 = ZZZ) of an XYZZ record, HBM or LDS
__device__ __forceinline__ fp load_coord(const uint32_t* rec, uint32_t coord) {
    fp r;
    const uint32_t* p = rec + 9 * coord;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = p[i];
    return r;
}
__device__ __forceinline__ void store_coord(uint32_t* rec, uint32_t coord, const fp& v) {
    uint32_t* p = rec + 9 * coord;
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = v.v[i];
}
// out = a + b by the EIGHT lanes of a group (ec_wide.hpp: 4 multiplications in series instead of 14).  Records in HBM or
// LDS; out may alias a.  Must be reached by whole 8-lane groups.  Special pairs (identity operand, equal or opposite
// points) fall back to the scalar complete addition on the group's first lane.
// the rare pairs (an identity operand, equal or opposite points): the scalar complete addition.  ~40 KB of code wherever it is inlined:
// a kernel keeps ONE eight-lane call site (the instruction cache is 64 KB for two CUs).  Out of line it would cost nothing in code, but
// a callee's register needs become the kernel's: 248 VGPRs, two wavefronts per SIMD for every reduction kernel.
__device__ __forceinline__ void add_records_complete(const uint32_t* a_rec, const uint32_t* b_rec, uint32_t* out_rec) {
    store_xyzz(out_rec, xyzz_add(load_xyzz(a_rec), load_xyzz(b_rec)));
}
template <bool PROBE = false>
__device__ __forceinline__ void wide_add_records(const uint32_t* a_rec, const uint32_t* b_rec, uint32_t* out_rec, long long* ts = nullptr) {
    const uint32_t role = threadIdx.x & (WIDE_LANES - 1);
    wide_mark<PROBE>(ts, 0);
    const fp opa = load_coord(wide_opa_rec(role) ? b_rec : a_rec, wide_opa_coord(role));
    const fp opb = load_coord(wide_opb_rec(role) ? b_rec : a_rec, wide_opb_coord(role));
    const bool ident = role == 4 && (fp_is_zero_exact(opa) || fp_is_zero_exact(opb));  // role 4 holds ZZ1 and ZZ2
    wide_mark<PROBE>(ts, 1);
    fp o;
    if (xyzz_add_wide<PROBE>(opa, opb, ident, o, ts)) {
        if (role == 0) add_records_complete(a_rec, b_rec, out_rec);
        return;
    }
    if (wide_has_out(role)) store_coord(out_rec, wide_out_coord(role), o);
    wide_mark<PROBE>(ts, 11);
}
constexpr uint32_t WIDE_TREE_MAX = 256;  // records one workgroup's LDS tree holds (36 KB)
// pairwise tree over m XYZZ records in LDS, wide additions, result in e[0].  Whole workgroup; blockDim multiple of 64.
__device__ __forceinline__ void lds_tree_wide(uint32_t* e, uint32_t m) {
    const uint32_t g = threadIdx.x / WIDE_LANES, ng = blockDim.x / WIDE_LANES;
    while (m > 1) {  // uniform
        const uint32_t h = (m + 1) >> 1;
        __syncthreads();
        for (uint32_t i = g; i < m - h; i += ng) wide_add_records(e + (size_t)i * XW, e + (size_t)(i + h) * XW, e + (size_t)i * XW);
        m = h;
    }
    __syncthreads();
}
// the same tree stopped at `stop` records (both powers of two, m >= stop): e[i] += e[i + m/2] level by level; records whose index modulo `stop` is
// >= valid are left alone (a workgroup that owns fewer than `stop` outputs: their slots hold nothing)
__device__ __forceinline__ void lds_tree_wide_until(uint32_t* e, uint32_t m, uint32_t stop, uint32_t valid) {
    const uint32_t g = threadIdx.x / WIDE_LANES, ng = blockDim.x / WIDE_LANES;
    while (m > stop) {  // uniform
        const uint32_t h = m >> 1;
        __syncthreads();
        for (uint32_t i = g; i < h; i += ng)
            if ((i & (stop - 1u)) < valid) wide_add_records(e + (size_t)i * XW, e + (size_t)(i + h) * XW, e + (size_t)i * XW);
        m = h;
    }
    __syncthreads();
}
// Jacobian in the C-ABI format: 24 words, canonical R = 2^256 Montgomery
__device__ __forceinline__ void store_jacobian_mont256(uint32_t* o, const jacobian& j) {
    uint32_t w[8];
    fp_to_mont256(w, j.x);
    store_words8(o, w);
    fp_to_mont256(w, j.y);
    store_words8(o + 8, w);
    fp_to_mont256(w, j.z);
    store_words8(o + 16, w);
}
// RESULTS IN PINNED HOST MEMORY, self-validating (round 6).  The host does not wait for the last kernel to retire: it polls what the kernel writes.  A first
// form -- the 24 words of a bit sum, a system-scope fence, then a sequence word per workgroup -- was WRONG about once in 50 000 calls: the host saw every
// sequence word of the call and still read a 32-byte chunk of the previous call's sum (tools/race_hunt.py; the chunk arrived after the word that was stored
// behind the fence -- device stores to host memory are posted writes, and nothing the shader can do orders two of them at their destination).  Now every
// result word travels as an aligned 8-byte PAIR (word, sequence number of the call): a pair is written by one store and read by one load, so a word whose
// tag is the call's number IS this call's word, whatever order the pairs arrive in.  48 words per bit sum instead of 24 + 1; no fence.
__device__ __forceinline__ void store_words8_tagged(uint32_t* p, const uint32_t w[8], uint32_t seq) {  // p: 16 words, 16-byte aligned
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], seq, w[1], seq);
    q[1] = make_uint4(w[2], seq, w[3], seq);
    q[2] = make_uint4(w[4], seq, w[5], seq);
    q[3] = make_uint4(w[6], seq, w[7], seq);
}
__device__ __forceinline__ void store_jacobian_mont256_tagged(uint32_t* o, const jacobian& j, uint32_t seq) {  // o: 48 words
    uint32_t w[8];
    fp_to_mont256(w, j.x);
    store_words8_tagged(o, w, seq);
    fp_to_mont256(w, j.y);
    store_words8_tagged(o + 16, w, seq);
    fp_to_mont256(w, j.z);
    store_words8_tagged(o + 32, w, seq);
}
// The same for the XYZZ record `rec` (LDS), by the first THREE lanes of the calling wavefront (round 6).  The conversion is three independent chains --
//   X' = X * (ZZ^2)^2 * c     Y' = Y * (ZZZ^2)^2 * c     Z' = ZZ * ZZZ * c          (c = 2^256 / 2^261: internal -> arkworks' Montgomery domain)
// -- of 4, 4 and 2 multiplications: lane 0, 1, 2 run one each through ONE fp_mul call site (a loop of four trips), 4 multiplications in series and
// 1.4 KB of code instead of the 10 in series and ~14 KB, executed once and cold, of store_jacobian_mont256(xyzz_to_jacobian()) on one lane: that tail
// was 14 of k_reduce_bits_wide's 40 us (profiles/r6_wide_level_breakdown.txt).  The identity (ZZ == 0) goes out as (R, R, 0), as there.
// Call with threadIdx.x < 64 (whole first wavefront, any lanes beyond 2 idle along); o = 48 words of pinned host memory ((word, seq) pairs, see above).
__device__ __forceinline__ void store_jacobian_mont256_lanes(uint32_t* o, const uint32_t* rec, uint32_t seq) {
    const uint32_t lane = threadIdx.x;
    if (lane >= 3) return;
    const fp zz = load_coord(rec, 2), zzz = load_coord(rec, 3);
    const bool ident = fp_is_zero_exact(zz);
    const fp xy = load_coord(rec, lane == 1 ? 1u : 0u);  // lane 0: X, lane 1: Y (lane 2: unused)
    const fp cm = fp_const(FP29_OUT_MONT);
    fp a = lane == 1 ? zzz : zz, b = lane == 0 ? zz : zzz, r = a;  // trip 0: ZZ^2 | ZZZ^2 | ZZ * ZZZ
#pragma unroll 1
    for (int t = 0; t < 4; t++) {
        const fp m = fp_mul(a, b);
        if (lane < 2 || t < 2) r = m;  // lane 2 is done after two trips (its later products are discarded)
        // next trip:  1: square again | square again | * c      2: * X | * Y | -      3: * c | * c | -
        a = r;
        b = t == 0 ? (lane == 2 ? cm : r) : t == 1 ? xy : cm;
        if (t == 1 && ident && lane < 2) a = fp_one(), b = fp_one();  // identity: X = Y = 1 (internal domain: one * one = one), then * c like any value
    }
    uint32_t w[8];
    fp_pack(w, fp_reduce_lt2p(r));
    store_words8_tagged(o + 16 * lane, w, seq);
}
__device__ __forceinline__ jacobian load_jacobian_mont256(const uint32_t* p) {
    uint32_t w[8];
    jacobian j;
    for (int c = 0; c < 3; c++) {
        for (int k = 0; k < 8; k++) w[k] = p[8 * c + k];
        fp t = fp_from_mont256(w);
        if (c == 0) j.x = t;
        else if (c == 1) j.y = t;
        else j.z = t;
    }
    return j;
}

// ---------------------------------------------------------------------------------------------
// K1's coordinate half: caller coordinates (standard form, or arkworks' R = 2^256 Montgomery words) -> the
// internal domain x*2^261 mod p, canonical, packed.  One Montgomery product per coordinate (the reference
// spends two 17x17-limb Barrett multiplications here, barrett_reduction.metal:84-118).
// With glv != 0 the record of phi(P_i) = (beta * x_i, y_i) is written as well, at index n + i (glv_bn254.hpp): one more
// multiplication per point, and the 2n records are what k_accumulate gathers from.
__device__ __forceinline__ void store_coord_and_phi(uint32_t* __restrict__ out, uint32_t pt, uint32_t which, uint32_t n, const fp& v, uint32_t glv) {
    uint32_t w[8];
    fp_pack(w, fp_reduce_lt2p(v));
    store_words8(out + ((size_t)pt * 2 + which) * 8, w);
    if (glv) {
        if (which == 0) fp_pack(w, fp_reduce_lt2p(fp_mul(v, fp_from_std(glv::BETA_STD))));  // beta * x, < 1.01p
        store_words8(out + ((size_t)(n + pt) * 2 + which) * 8, w);
    }
}
__global__ void k_convert_bases(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, uint32_t mont_form, uint32_t glv) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per coordinate
    if (i >= 2u * n) return;
    fp v = load_fp_packed(in + (size_t)i * 8);
    v = fp_mul(v, mont_form ? fp_const(FP29_IN_MONT) : fp_const(FP29_IN_STD));  // < 1.01p
    store_coord_and_phi(out, i >> 1, i & 1u, n, v, glv);
}

// Round 5: the device and host-pointer calls on arkworks-form bases (R = 2^256 Montgomery words) no longer convert them at all --
// k_accumulate_pieces<.., M256> gathers the caller's records as they are (fp_unpack_shl5: 32 * W is the internal-domain value, unreduced).
// What is left of K1's coordinate half on that path is the phi half of a split plan: record i = (beta * x_i, y_i) in the SAME word form, one
// multiplication and 128 bytes per point where k_convert_bases spent three and 192 (unsplit plans: nothing at all).
// (out of line: inlined into k_decompose_glv the two Montgomery products made the register allocator give that kernel 256 VGPRs -- one
// wavefront per SIMD, 541 us instead of 45 at 2^20 -- wherever in the kernel the block stood)
__device__ __noinline__ void phi_record(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    uint32_t w[8];
    load_words8(w, in);
    const fp bx = fp_reduce_lt2p(fp_mul(fp_unpack(w), fp_from_std(glv::BETA_STD)));  // X * beta (mod p), canonical: beta * x in R = 2^256 form
    fp_pack(w, bx);
    store_words8(out, w);
    load_words8(w, in + 8);
    store_words8(out + 8, w);
}
__global__ void k_phi_records(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    phi_record(in + (size_t)i * 16, out + (size_t)i * 16);
}

// Zero-copy ingestion of an array of arkworks `G1Affine` structs: the struct array is copied to
// HBM as it is and read here through (stride, x offset, y offset, infinity offset) -- the layout is probed by the
// Rust shim with addr_of!, never assumed.  Coordinates are Fq Montgomery words (R = 2^256).
__global__ void k_import_ark(const uint8_t* __restrict__ raw, uint64_t stride, uint32_t x_off, uint32_t y_off, uint32_t inf_off,
                             uint32_t has_inf, uint32_t n, uint32_t* __restrict__ out, uint8_t* __restrict__ inf_out, uint32_t glv) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per coordinate
    if (i >= 2u * n) return;
    const uint32_t pt = i >> 1, which = i & 1u;
    const uint8_t* rec = raw + (size_t)pt * stride;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(rec + (which ? y_off : x_off));  // 4-byte aligned (checked on the host)
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = src[k];
    fp v = fp_mul(fp_unpack(w), fp_const(FP29_IN_MONT));
    store_coord_and_phi(out, pt, which, n, v, glv);
    if (which == 0 && inf_out) inf_out[pt] = has_inf ? (rec[inf_off] != 0) : 0;
}

// The FAST form of the same ingestion (round 6), for struct arrays WITHOUT points at infinity -- every proving key, every SRS: the Fq words are already what
// k_accumulate_pieces<.., M256> gathers (arkworks' own Montgomery words), so the structs are only REPACKED to 64-byte x || y records (no field
// multiplication; with glv the phi record (beta * x, y) of every point behind them, at record n + i) and the call runs like the packed-words call: the sort
// starts when the scalars are there, not when the bases are.  A set `infinity` flag is not handled here: it raises error bit 16 and the host runs the call again
// through k_import_ark (the flags then travel to the decomposition as an infinity mask).
__global__ void __launch_bounds__(256) k_ark_repack(const uint8_t* __restrict__ raw, uint64_t stride, uint32_t x_off, uint32_t y_off, uint32_t inf_off, uint32_t has_inf,
                                                    uint32_t n, uint32_t* __restrict__ out, uint32_t* __restrict__ err, uint32_t glv) {
    const uint32_t pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const uint8_t* rec = raw + (size_t)pt * stride;
    const uint32_t* sx = reinterpret_cast<const uint32_t*>(rec + x_off);  // 4-byte aligned (checked on the host)
    const uint32_t* sy = reinterpret_cast<const uint32_t*>(rec + y_off);
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = sx[k];
    store_words8(out + (size_t)pt * 16, w);
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = sy[k];
    store_words8(out + (size_t)pt * 16 + 8, w);
    if (has_inf && rec[inf_off] != 0) atomicOr(err + FLAG_ERR, 16u);
    if (glv) phi_record(out + (size_t)pt * 16, out + (size_t)(n + pt) * 16);  // (reads back what this thread just wrote)
}

// Row f3: arkworks `serialize_compressed` images of G1Affine (reference utils/preprocess.rs:193-223) -> bases.
// One thread per point: x (standard form, flags in bits 254/255) -> y = (x^3+3)^((p+1)/4)  (p = 3 mod 4), the root
// is checked (y^2 == x^3+3, else the image is invalid) and the sign picked as ark-ec 0.4 does: flag bit 255 set <=> y is
// the larger of (y, p-y) as integers.  The exponent is a constant, so the square-and-multiply branch is wave-uniform.
// out_ark = 0: internal domain, packed (the pipeline's base format);  1: arkworks Montgomery words R = 2^256.
__global__ void k_decompress(const uint32_t* __restrict__ rec, uint32_t n, uint32_t* __restrict__ out, uint8_t* __restrict__ inf_out,
                             uint32_t* __restrict__ first_bad, uint32_t out_ark, uint32_t glv) {
    constexpr uint32_t PW[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    constexpr uint32_t EXP[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};   // (p+1)/4, 252 bits
    constexpr uint32_t HALF[8] = {0x6c3e7ea3u, 0x9e10460bu, 0xb438e546u, 0xcbc0b548u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};  // (p-1)/2
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[8];
    load_words8(w, rec + (size_t)i * 8);
    const uint32_t f_neg = w[7] >> 31, f_inf = (w[7] >> 30) & 1u;
    w[7] &= 0x3FFFFFFFu;
    bool lt = false;  // x < p ?
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        if (w[k] != PW[k]) {
            lt = w[k] < PW[k];
            break;
        }
    }
    uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!lt || (f_neg & f_inf)) {
        atomicMin(first_bad, i);
        store_words8(out + (size_t)i * 16, zero);
        store_words8(out + (size_t)i * 16 + 8, zero);
        inf_out[i] = 0;
        return;
    }
    if (f_inf) {  // identity: coordinates unused downstream
        store_words8(out + (size_t)i * 16, zero);
        store_words8(out + (size_t)i * 16 + 8, zero);
        inf_out[i] = 1;
        return;
    }
    const fp x = fp_from_std(w);                                    // < 1.01p
    const fp three = fp_add(fp_dbl(fp_one()), fp_one());              // < 3p
    const fp rhs = fp_add(fp_mul(fp_sqr(x), x), three);              // < 4.01p
    // fixed 3-bit windows over the 252-bit exponent (84 of them): 7 table entries rhs^1 .. rhs^7 in registers, then per window three
    // squarings and -- unless the window is zero -- one multiplication instead of the 251 + 126 of
    // bit-by-bit square-and-multiply: 252 + 74 for this exponent (-16 % multiplier instructions).  The window value is a constant of the exponent: wave-uniform.
    fp tab[7];
    tab[0] = rhs;                              // < 4.01p: a product's operand, as in the bit-by-bit chain
    tab[1] = fp_sqr(tab[0]);                   // ^2
    tab[2] = fp_mul(tab[1], tab[0]);           // ^3
    tab[3] = fp_sqr(tab[1]);                   // ^4
    tab[4] = fp_mul(tab[3], tab[0]);           // ^5
    tab[5] = fp_sqr(tab[2]);                   // ^6
    tab[6] = fp_mul(tab[5], tab[0]);           // ^7
    auto window = [&](int k) -> uint32_t {  // bits [3k, 3k+3) of the exponent
        const int b = 3 * k, i = b >> 5;
        const uint64_t lo = EXP[i], hi = i < 7 ? EXP[i + 1] : 0u;
        return (uint32_t)(((hi << 32) | lo) >> (b & 31)) & 7u;
    };
    auto entry = [&](uint32_t d) {  // d in 1..7, uniform
        fp r = tab[0];
#pragma unroll
        for (int k = 1; k < 7; k++)
            if (d == (uint32_t)k + 1) r = tab[k];
        return r;
    };
    fp y = entry(window(83));                  // the top window (bits 249..251) is not zero: bit 251 is set
#pragma unroll 1
    for (int k = 82; k >= 0; k--) {
        y = fp_sqr(fp_sqr(fp_sqr(y)));                                // < 1.1p
        const uint32_t d = window(k);
        if (d) y = fp_mul(y, entry(d));                               // < 1.03p
    }
    const fp y2 = fp_canonical(fp_sqr(y)), r2 = fp_canonical(rhs);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 9; k++) ok = ok && (y2.v[k] == r2.v[k]);
    if (!ok) {
        atomicMin(first_bad, i);
        store_words8(out + (size_t)i * 16, zero);
        store_words8(out + (size_t)i * 16 + 8, zero);
        inf_out[i] = 0;
        return;
    }
    uint32_t ys[8];
    fp_to_std(ys, y);
    bool larger = false;  // y > (p-1)/2  <=>  y > p - y
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        if (ys[k] != HALF[k]) {
            larger = ys[k] > HALF[k];
            break;
        }
    }
    fp yc = fp_reduce_lt2p(y);  // canonical
    if ((larger ? 1u : 0u) != f_neg && !fp_is_zero_exact(yc)) yc = fp_reduce_lt2p(fp_neg<2>(yc));  // p - y
    uint32_t o[8];
    if (out_ark) fp_to_mont256(o, x); else fp_pack(o, fp_reduce_lt2p(x));
    store_words8(out + (size_t)i * 16, o);
    if (out_ark) fp_to_mont256(o, yc); else fp_pack(o, yc);
    store_words8(out + (size_t)i * 16 + 8, o);
    if (glv && !out_ark) {  // record of phi(P_i) = (beta*x, y) at index n + i
        store_words8(out + (size_t)(n + i) * 16 + 8, o);
        fp_pack(o, fp_reduce_lt2p(fp_mul(x, fp_from_std(glv::BETA_STD))));
        store_words8(out + (size_t)(n + i) * 16, o);
    }
    inf_out[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// Row f4: the WINDOW TABLE of a resident base set.  T_0 = the bases, T_j[i] = 2^(c*j) * P_i for j < f.
// A digit of window w = v*f + j then adds  +-T_j[i]  into bucket array v, whose weight is 2^(c*f*v): the f windows of a group share
// ONE bucket array, and with f = W every window shares the same one -- which is what allows windows of 20 bits (2^19 buckets) at
// 2^20 points: 13 windows instead of 16, 19 % fewer mixed additions, one array of buckets to reduce.  The reference names this
// family of techniques under "advanced algorithms" (README.md:192-196); its cost model (utils/window_size_optimizer.rs:38-51,
// (n + 2^(s+1)) * ceil(lambda/s)) is what the table changes: the 2^(s+1) term is paid once, not once per window.
// One thread per PAIR of records: c doublings each (XYZZ, dbl-2008-s-1), then ONE inversion for both (Montgomery's trick: 1/(za*zb), times zb,
// times za) back to affine -- the Fermat inversion (254 squarings + 127 multiplications) is two thirds of a record's work; run once per upload
// and level (2^20 points, 12 levels: 39 -> 31 ms; with the windowed exponent of fp_inv: 27.5).  A record whose ZZZ is 0 mod p (a base at infinity -- masked out by its digits -- or a
// garbage record) gives (0, 0), never read, and must not poison its partner: it enters the product as 1.
__global__ void __launch_bounds__(256) k_table_next(const uint32_t* __restrict__ prev, uint32_t* __restrict__ next, uint32_t nv, uint32_t c) {
    const uint32_t i0 = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i0 >= nv) return;
    const bool two = i0 + 1 < nv;
    xyzz ta = xyzz_dbl_affine(load_affine(prev + (size_t)i0 * 16));
    xyzz tb = xyzz_dbl_affine(load_affine(prev + (size_t)(two ? i0 + 1 : i0) * 16));
#pragma unroll 1
    for (uint32_t k = 1; k < c; k++) {
        ta = xyzz_dbl(ta);
        tb = xyzz_dbl(tb);
    }
    const fp za = fp_canonical(ta.zzz), zb = fp_canonical(tb.zzz);  // in [0, p): a value that is 0 mod p is exactly 0 now
    const bool dead_a = fp_is_zero_exact(za) || xyzz_is_identity(ta), dead_b = fp_is_zero_exact(zb) || xyzz_is_identity(tb);
    const fp ma = dead_a ? fp_one() : za, mb = dead_b ? fp_one() : zb;
    const fp ip = fp_inv(fp_mul(ma, mb));
    const fp inv[2] = {fp_mul(ip, mb), fp_mul(ip, ma)};  // 1/ZZZ of a, of b
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h == 1 && !two) break;
        const xyzz& t = h ? tb : ta;
        uint32_t w[8];
        fp x = fp_zero(), y = fp_zero();
        if (!(h ? dead_b : dead_a)) {
            const fp tt = fp_mul(inv[h], t.zz);  // ZZ/ZZZ
            x = fp_mul(t.x, fp_sqr(tt));        // X/ZZ
            y = fp_mul(t.y, inv[h]);            // Y/ZZZ
        }
        fp_pack(w, fp_reduce_lt2p(x));  // (products: normalised, < 2p)
        store_words8(next + (size_t)(i0 + h) * 16, w);
        fp_pack(w, fp_reduce_lt2p(y));
        store_words8(next + (size_t)(i0 + h) * 16 + 8, w);
    }
}

// ---------------------------------------------------------------------------------------------
// K1 scalar half + K2 phase 1.  One thread per point: slice the scalar into W radix-2^c digits,
// recode to signed digits d in [-(H-1), H] (v > H  =>  d = v - 2H, carry 1), and count the bucket.
// bits [off, off+c) of a 256-bit little-endian scalar
__device__ __forceinline__ uint32_t scalar_window(const uint32_t s[8], uint32_t off, uint32_t c) {
    uint32_t wi = off >> 5, sh = off & 31;
    if (wi >= 8) return 0;
    uint64_t lo = s[wi];
    uint64_t hi = (wi + 1 < 8) ? s[wi + 1] : 0u;
    uint64_t v = (lo | (hi << 32)) >> sh;
    return (uint32_t)v & ((1u << c) - 1u);
}

// arkworks holds Fr in Montgomery form (R = 2^256 mod r); the reference converts every scalar on the CPU
// (`into_bigint()` inside pack_affine_and_scalars, utils/limbs_conversion.rs:311-378).  Done here on the device instead
// (SURVEY section 8 row f1): s * 2^-256 mod r by word-serial Montgomery reduction, canonical result.
__device__ __forceinline__ void fr_from_mont(uint32_t s[8]) {
    constexpr uint32_t R[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    constexpr uint32_t INV = 0xefffffffu;  // -r^-1 mod 2^32
    uint32_t t[9] = {s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t m = t[0] * INV;
        uint64_t carry = ((uint64_t)m * R[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            uint64_t v = (uint64_t)m * R[j] + t[j] + carry;
            t[j - 1] = (uint32_t)v;
            carry = v >> 32;
        }
        uint64_t v = (uint64_t)t[8] + carry;
        t[7] = (uint32_t)v;
        t[8] = (uint32_t)(v >> 32);
    }
    // t < 2r: subtract r once if needed
    uint32_t d[8];
    uint64_t bw = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t v = (uint64_t)t[j] - R[j] - bw;
        d[j] = (uint32_t)v;
        bw = (v >> 32) & 1u;
    }
    const bool ge = t[8] != 0 || bw == 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = ge ? d[j] : t[j];
}

// HIST = true also counts buckets with global atomics (fallback when a window's histogram does not fit LDS);
// HIST = false only writes the digits and leaves counting to k_tile_hist.
template <bool SIGNED, bool HIST, bool D16>
__global__ void k_decompose(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_mask, uint32_t n,
                            uint32_t c, uint32_t W, uint32_t nb, uint32_t* __restrict__ hist,
                            void* __restrict__ digits, uint32_t drow, uint32_t* __restrict__ ranks, uint32_t* __restrict__ err,
                            uint32_t scalars_mont, uint32_t top_shift, uint32_t top_bits, uint32_t spread_mask,
                            uint32_t* __restrict__ phist, uint32_t* __restrict__ pcursor) {
    static_assert(!(HIST && D16), "the global-atomic fallback keeps 32-bit digits");
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) err[FLAG_LONG] = 0, err[FLAG_MID] = 0, err[FLAG_MID2] = 0, err[FLAG_PIECES] = 0, err[FLAG_PARTIALS] = 0, err[FLAG_NONEMPTY] = 0;  // list counters of this (chunk of an) MSM: k_piece_count fills them later in the stream
    if (blockIdx.x == 0) clear_piece_bins(phist, pcursor);
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    uint4 a = sp[0], b = sp[1];
    uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (scalars_mont) fr_from_mont(s);
    if (s[7] >> 30) atomicOr(err, 1u);  // scalar >= 2^254 cannot be a canonical Fr
    bool skip = inf_mask != nullptr && inf_mask[i] != 0;
    const uint32_t H = 1u << (c - 1);
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; w++) {
        uint32_t v = scalar_window(s, w * c, c) + carry;
        uint32_t mag = v, neg = 0;
        if (SIGNED) {
            if (v > H) {
                mag = (2u * H) - v;
                neg = SIGN_BIT;
                carry = 1;
            } else {
                carry = 0;
            }
        }
        size_t o = (size_t)w * drow + i;
        // window table with one shared bucket array: the short top window (14 bits of 20 at c = 20) would pile its digits into the
        // lowest buckets -- 32 of the sort's 1024 regions.  Its table level is 2^(c*(W-1) - top_shift) P instead, and the digit goes in
        // as d * 2^top_shift: the same group element, spread over every 2^top_shift-th bucket
        if (w == W - 1) mag <<= top_shift;
        if (mag == 0 || skip) {
            digit_store<D16>(digits, o, DIGIT_SKIP);
        } else {
            uint32_t bkt = mag - 1;
            // the top window of a plan holds 254 - c*(W-1) bits: magnitudes up to 2^top_bits.  spread_mask != 0 (msmplan::top_digit_bits): the index
            // bits above them carry low bits of the point index, so that the window's entries use all of its buckets (see k_decompose_glv)
            if (w == W - 1 && spread_mask) bkt |= (i & spread_mask) << top_bits;
            if (HIST) ranks[o] = atomicAdd(&hist[(size_t)w * nb + bkt], 1u);
            if (D16 && (bkt | neg) == (0x7FFFu | SIGN_BIT)) atomicOr(err, 8u);  // cannot happen: would read as the 16-bit skip code
            digit_store<D16>(digits, o, bkt | neg);
        }
    }
    if (SIGNED && carry) atomicOr(err, 2u);  // cannot happen for scalars < 2^254 with W = 254/c + 1
}

// bits [off, off+c) of a 128-bit little-endian magnitude
__device__ __forceinline__ uint32_t window128(const uint32_t s[4], uint32_t off, uint32_t c) {
    uint32_t wi = off >> 5, sh = off & 31;
    if (wi >= 4) return 0;
    uint64_t lo = s[wi];
    uint64_t hi = (wi + 1 < 4) ? s[wi + 1] : 0u;
    return (uint32_t)(((lo | (hi << 32)) >> sh)) & ((1u << c) - 1u);
}
// k_decompose with the GLV split (glv_bn254.hpp): scalar k_i -> (k1, k2), |k_j| < 7 * 2^123; the digits of |k1| go to virtual point
// i, those of |k2| to virtual point n + i (the record of phi(P_i)), the sign of k_j is folded into every digit's negate flag.
// digits: W x 2n, window-major.
// spread_mask != 0 (msmplan::glv_top_digit_bits): the top window's magnitudes are at most 2^top_bits, fewer than the window has buckets;
// its bucket index is (magnitude - 1) | (i & spread_mask) << top_bits -- the same digit in 2^spread buckets, chosen by the point index, so
// that the top window's buckets are no fuller than the others'.  The host ignores the bit sums of the index bits above top_bits.
// PHI (round 5): the thread also writes the phi record of its point, (beta * x_i, y_i) in the caller's R = 2^256 word form (what k_phi_records
// does) -- the device call on arkworks-form bases has the base array at hand when the scalars are decomposed, and one launch less (and no
// second stream, no cross-stream event) is worth more than the 64 + 64 bytes per point cost this kernel.
template <bool SIGNED, bool D16, bool PHI>
__global__ void __launch_bounds__(256) k_decompose_glv(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_mask, uint32_t n, uint32_t c,
                                uint32_t W, void* __restrict__ digits, uint32_t drow, uint32_t* __restrict__ err, uint32_t scalars_mont,
                                uint32_t top_shift, uint32_t top_bits, uint32_t spread_mask, uint32_t* __restrict__ phist,
                                uint32_t* __restrict__ pcursor, const uint32_t* __restrict__ phi_src, uint32_t* __restrict__ phi_dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) err[FLAG_LONG] = 0, err[FLAG_MID] = 0, err[FLAG_MID2] = 0, err[FLAG_PIECES] = 0, err[FLAG_PARTIALS] = 0, err[FLAG_NONEMPTY] = 0;  // list counters of this (chunk of an) MSM: k_piece_count fills them later in the stream
    if (blockIdx.x == 0) clear_piece_bins(phist, pcursor);
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    uint4 a = sp[0], b = sp[1];
    uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (scalars_mont) fr_from_mont(s);
    if (s[7] >> 30) atomicOr(err, 1u);  // scalar >= 2^254 cannot be a canonical Fr
    const bool skip = inf_mask != nullptr && inf_mask[i] != 0;
    uint32_t k[2][4];
    bool kneg[2];
    if (!glv::split(s, k[0], kneg[0], k[1], kneg[1])) atomicOr(err, 4u);  // a half beyond 126 bits: cannot happen below 2^254
    const uint32_t H = 1u << (c - 1);
    const size_t row = drow;  // >= 2n
    const uint32_t spread = (i & spread_mask) << top_bits;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t carry = 0;
        for (uint32_t w = 0; w < W; w++) {
            uint32_t v = window128(k[h], w * c, c) + carry;
            uint32_t mag = v;
            bool neg = kneg[h];
            if (SIGNED) {
                // v == H may go either way (+H, or -H with a carry).  It goes the way that leaves the FINAL digit (the half's sign folded
                // in) in [-(H-1), H], the range of the unsplit form: a negative half recodes from v >= H on, a positive one from v > H.
                // (Round 5: the 16-bit digit codes have no room for a negated magnitude H.)
                if (v + (kneg[h] ? 1u : 0u) > H) {
                    mag = (2u * H) - v;
                    neg = !neg;
                    carry = 1;
                } else {
                    carry = 0;
                }
            }
            uint32_t bkt = mag - 1;
            if (w == W - 1) {
                if (spread_mask) {
                    if (mag > (1u << top_bits)) atomicOr(err, 4u);  // cannot happen: halves < 7 * 2^123
                    bkt |= spread;
                } else {
                    bkt = (mag << top_shift) - 1;  // (window table, shared bucket array: see k_decompose)
                }
            }
            const uint32_t code = (mag == 0 || skip) ? DIGIT_SKIP : (bkt | (neg ? SIGN_BIT : 0u));
            if (D16 && code == (0x7FFFu | SIGN_BIT)) atomicOr(err, 8u);  // cannot happen: would read as the 16-bit skip code
            digit_store<D16>(digits, (size_t)w * row + (size_t)h * n + i, code);
        }
        if (SIGNED && carry) atomicOr(err, 2u);
    }
    if (PHI) phi_record(phi_src + (size_t)i * 16, phi_dst + (size_t)i * 16);  // (last: nothing of the split is live any more)
}

// ---------------------------------------------------------------------------------------------
// K2 phase 2: exclusive prefix sum of the W*nb bucket counts (three small launches).
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_BLOCK / 64; k++) {
        if (k < wid) base += wsum[k];
        tot += wsum[k];
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// tile-local exclusive scan; tile totals to block_sums
__global__ void k_scan_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ block_sums,
                             uint32_t count) {
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < count) ? in[base + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t ex = block_exclusive_scan(sum, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < count) out[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// single block: exclusive scan of the tile totals (any number of tiles), grand total to *total_out
__global__ void k_scan_block_sums(uint32_t* block_sums, uint32_t nblocks, uint32_t* total_out) {
    uint32_t running = 0;
    for (uint32_t start = 0; start < nblocks; start += SCAN_BLOCK) {
        uint32_t idx = start + threadIdx.x;
        uint32_t v = idx < nblocks ? block_sums[idx] : 0u;
        uint32_t total;
        uint32_t ex = block_exclusive_scan(v, &total);
        if (idx < nblocks) block_sums[idx] = running + ex;
        running += total;
    }
    if (threadIdx.x == 0) *total_out = running;
}
__global__ void k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ block_sums, uint32_t count,
                           const uint32_t* __restrict__ total) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] += block_sums[i / SCAN_TILE];
    if (i == 0) out[count] = *total;
}

// K2 phase 1 (LDS path): one 1024-thread workgroup counts one TILE of points of one window in an LDS
// histogram of all nb buckets (<= 128 KB of the CU's 160 KB).  The LDS atomic returns the arrival rank inside
// (window, tile, bucket) -- used by k_tile_scatter, which repeats the same atomics on absolute cursors;
// counts[w][tile][b] go to HBM with coalesced stores.  16.7 M LDS atomics replace
// 16.7 M scattered device-scope atomics (the reference does this with ONE thread per window, transpose.metal:27-32).
constexpr int TILE_BLOCK = 1024;
__global__ void __launch_bounds__(TILE_BLOCK) k_tile_hist(const uint32_t* __restrict__ digits, uint32_t* __restrict__ counts,
                                                          uint32_t n, uint32_t nb, uint32_t tile_len, uint32_t T) {
    extern __shared__ uint32_t s_tile_hist[];
    const uint32_t tile = blockIdx.x, w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < nb; b += TILE_BLOCK) s_tile_hist[b] = 0;
    __syncthreads();
    const uint32_t i0 = tile * tile_len, i1 = min(n, i0 + tile_len);
    const size_t row = (size_t)w * n;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += TILE_BLOCK) {
        uint32_t d = digits[row + i];
        if (d != DIGIT_SKIP) atomicAdd(&s_tile_hist[d & ~SIGN_BIT], 1u);
    }
    __syncthreads();
    uint32_t* out = counts + ((size_t)w * T + tile) * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += TILE_BLOCK) out[b] = s_tile_hist[b];
}
// per (window, bucket): exclusive prefix over the T tiles in place, bucket total to hist
__global__ void k_tile_prefix(uint32_t* __restrict__ counts, uint32_t* __restrict__ hist, uint32_t nb, uint32_t T,
                              uint32_t total_buckets) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total_buckets) return;
    uint32_t w = k / nb, b = k % nb;
    uint32_t* p = counts + (size_t)w * T * nb + b;
    uint32_t run = 0;
    for (uint32_t t = 0; t < T; t++) {
        uint32_t cnt = p[(size_t)t * nb];
        p[(size_t)t * nb] = run;
        run += cnt;
    }
    hist[k] = run;
}

// K2 phase 3 (LDS path): the same (window, tile) workgroup shape as k_tile_hist.  The LDS array now holds the
// ABSOLUTE write cursor of every bucket for this tile (bucket offset + counts of earlier tiles, both read with
// coalesced loads), and each element takes its slot with one LDS atomic: no rank array, no random reads.
__global__ void __launch_bounds__(TILE_BLOCK) k_tile_scatter(const uint32_t* __restrict__ digits, const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ tile_base, uint32_t* __restrict__ sorted,
                                                             uint32_t n, uint32_t nb, uint32_t tile_len, uint32_t T) {
    extern __shared__ uint32_t s_tile_hist[];
    // (measured: an XCD-aware window->XCD mapping of these workgroups changes nothing -- the stores are 4-byte
    // writes into a 4 MB-per-window region and leave L2 as one partially written line each; profiles/NOTES_r1.md)
    const uint32_t tile = blockIdx.x, w = blockIdx.y;
    const uint32_t* ob = offsets + (size_t)w * nb;
    const uint32_t* tbp = tile_base + ((size_t)w * T + tile) * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += TILE_BLOCK) s_tile_hist[b] = ob[b] + tbp[b];
    __syncthreads();
    const uint32_t i0 = tile * tile_len, i1 = min(n, i0 + tile_len);
    const size_t row = (size_t)w * n;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += TILE_BLOCK) {
        uint32_t d = digits[row + i];
        if (d != DIGIT_SKIP) sorted[atomicAdd(&s_tile_hist[d & ~SIGN_BIT], 1u)] = i | (d & SIGN_BIT);
    }
}

// ---------------------------------------------------------------------------------------------
// K2, two-level LDS counting sort (the default when n <= 2^(31 - fine_bits)).
// Measured (profiles/NOTES_r1.md): stores leave L2 per wave instruction, so a scatter of 4-byte elements costs one
// memory transaction per element whatever its locality (k_tile_scatter: 216 us at N = 2^20).  Here every store
// instruction writes a contiguous run:
//   level 1  bucket index = coarse (top 8..10 bits, so that a region holds ~8192 elements) | fine.  k_coarse_hist
//            counts the coarse bins of every
//            (window, 16384-point sub-tile); k_coarse_prefix / k_coarse_starts turn the counts into write positions;
//            k_coarse_scatter sorts a sub-tile by coarse bin IN LDS and copies each bin's run out contiguously.
//            An element travels as  point index | fine << idx_bits | sign << 31.
//   level 2  k_fine_sort: one workgroup per (window, coarse bin) region (~8192 elements): fine histogram in LDS ->
//            the bucket offsets (this replaces k_tile_hist / k_tile_prefix / k_scan_*), LDS-staged placement,
//            sequential copy-out.  Regions larger than the LDS staging area (skewed data) place directly.
constexpr uint32_t SUBTILE = 16384;        // elements sorted in LDS by one level-1 workgroup (64 KB staging)
constexpr uint32_t FINE_CAP = 16384;       // elements a level-2 workgroup can stage in LDS (64 KB: two workgroups per CU)
constexpr uint32_t COARSE_BINS_MAX = 1024;  // == TILE_BLOCK: one thread per coarse bin

// atomicAdd(&ctr[key], 1u) on an LDS counter for the active lanes of a wavefront.  When ALL of them hit the same counter
// (skewed scalars: one bucket holds most of a window -- many equal witness values, or the adversarial all-equal case) the
// wavefront issues ONE atomic for the whole group instead of 64 conflicting ones; the check is two ballots.
__device__ __forceinline__ uint32_t lds_inc(uint32_t* ctr, uint32_t key) {
    const unsigned long long act = __ballot(1);
    const uint32_t k0 = __builtin_amdgcn_readfirstlane(key);
    if (__ballot(key == k0) == act) {  // wave-uniform
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (lane == (uint32_t)(__ffsll((long long)act) - 1)) base = atomicAdd(&ctr[k0], (uint32_t)__popcll(act));
        base = __builtin_amdgcn_readfirstlane(base);  // the first active lane is the one that added
        return base + (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
    }
    return atomicAdd(&ctr[key], 1u);
}

// The SUBTILE / TILE_BLOCK = 16 digit codes (32-bit form) a thread of a level-1 workgroup owns, all loads in flight at once.  32-bit digits:
// entry m is element i0 + tid + m * TILE_BLOCK.  16-bit digits: eight 4-byte loads of two codes each, entries 2k and 2k+1 are elements
// i0 + 2 * (tid + k * TILE_BLOCK) and the one behind it (rows are padded to an even length and start 4-byte aligned: digit_row_stride).
template <bool D16>
__device__ __forceinline__ uint32_t subtile_element(uint32_t i0, int m) {
    return D16 ? i0 + 2u * (threadIdx.x + (uint32_t)(m >> 1) * TILE_BLOCK) + (uint32_t)(m & 1) : i0 + threadIdx.x + (uint32_t)m * TILE_BLOCK;
}
template <bool D16>
__device__ __forceinline__ void load_subtile_digits(const void* __restrict__ digits, size_t row, uint32_t i0, uint32_t i1,
                                                    uint32_t (&dg)[SUBTILE / TILE_BLOCK]) {
    if (D16) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(digits) + row + i0);
#pragma unroll
        for (int k = 0; k < SUBTILE / TILE_BLOCK / 2; k++) {
            const uint32_t j = threadIdx.x + (uint32_t)k * TILE_BLOCK, i = i0 + 2u * j;
            const uint32_t v = i < i1 ? p[j] : 0xFFFFFFFFu;
            dg[2 * k] = digit16_widen(v & 0xFFFFu);
            dg[2 * k + 1] = i + 1 < i1 ? digit16_widen(v >> 16) : DIGIT_SKIP;  // (the pad entry of an odd row is not a digit)
        }
    } else {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(digits) + row;
#pragma unroll
        for (int k = 0; k < SUBTILE / TILE_BLOCK; k++) {
            const uint32_t i = i0 + threadIdx.x + (uint32_t)k * TILE_BLOCK;
            dg[k] = i < i1 ? p[i] : DIGIT_SKIP;
        }
    }
}

template <bool D16>
__global__ void __launch_bounds__(TILE_BLOCK) k_coarse_hist(const void* __restrict__ digits, uint32_t drow, uint32_t* __restrict__ counts,
                                                            uint32_t n, uint32_t fine_bits, uint32_t ncoarse, uint32_t NS,
                                                            uint32_t* __restrict__ flags, uint32_t* __restrict__ phist,
                                                            uint32_t* __restrict__ pcursor) {
    __shared__ uint32_t s_h[COARSE_BINS_MAX];
    const uint32_t st = blockIdx.x, w = blockIdx.y;
    // list counters and piece bins of THIS sort call (k_piece_count fills them later in the stream; k_decompose zeroes them too -- kept
    // here so that a sort never depends on which kernel ran before it)
    if (st == 0 && w == 0) {
        if (threadIdx.x == 0) flags[FLAG_LONG] = 0, flags[FLAG_MID] = 0, flags[FLAG_MID2] = 0, flags[FLAG_PIECES] = 0, flags[FLAG_PARTIALS] = 0, flags[FLAG_NONEMPTY] = 0;
        clear_piece_bins(phist, pcursor);
    }
    if (threadIdx.x < COARSE_BINS_MAX) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i0 = st * SUBTILE, i1 = min(n, i0 + SUBTILE);
    // SUBTILE / TILE_BLOCK = 16 digits per thread: all loads in flight before the first LDS atomic
    uint32_t dg[SUBTILE / TILE_BLOCK];
    load_subtile_digits<D16>(digits, (size_t)w * drow, i0, i1, dg);
#pragma unroll
    for (int k = 0; k < SUBTILE / TILE_BLOCK; k++)
        if (dg[k] != DIGIT_SKIP) lds_inc(s_h, (dg[k] & ~SIGN_BIT) >> fine_bits);
    __syncthreads();
    if (threadIdx.x < ncoarse) counts[((size_t)w * NS + st) * ncoarse + threadIdx.x] = s_h[threadIdx.x];  // [w][sub-tile][bin]
}
// per (window, bin): exclusive prefix over the sub-tiles in place; total of the region.  One 1024-thread workgroup per PREFIX_REGIONS
// consecutive regions: thread (segment, region) sums its SEGMENT of the sub-tiles (consecutive threads own consecutive regions: every
// step reads 64-byte runs of counts[w][sub-tile][*]), the segment totals meet in LDS, a second pass writes the prefixes.  (Round 2:
// one thread per region walked all NS sub-tiles alone -- 64 dependent steps at 2^20 points, 832 with the 13 x 2^20 entries of a
// window-table MSM: 0.3 ms of latency on four workgroups.)
constexpr uint32_t PREFIX_REGIONS = 16, PREFIX_SEGS = 1024 / PREFIX_REGIONS;
__global__ void __launch_bounds__(1024) k_coarse_prefix(uint32_t* __restrict__ counts, uint32_t* __restrict__ region_total, uint32_t NS,
                                                        uint32_t ncoarse, uint32_t nregions) {
    __shared__ uint32_t s_seg[PREFIX_SEGS][PREFIX_REGIONS + 1];
    const uint32_t rl = threadIdx.x % PREFIX_REGIONS, sg = threadIdx.x / PREFIX_REGIONS;
    const uint32_t r = blockIdx.x * PREFIX_REGIONS + rl;
    const bool live = r < nregions;
    const uint32_t w = live ? r / ncoarse : 0u, b = live ? r % ncoarse : 0u;
    uint32_t* p = counts + (size_t)w * NS * ncoarse + b;
    const uint32_t per = (NS + PREFIX_SEGS - 1) / PREFIX_SEGS;
    const uint32_t s0 = min(NS, sg * per), s1 = min(NS, s0 + per);
    uint32_t sum = 0;
    if (live)
        for (uint32_t s = s0; s < s1; s++) sum += p[(size_t)s * ncoarse];
    s_seg[sg][rl] = sum;
    __syncthreads();
    uint32_t run = 0, tot = 0;
    for (uint32_t k = 0; k < PREFIX_SEGS; k++) {
        const uint32_t v = s_seg[k][rl];
        if (k < sg) run += v;
        tot += v;
    }
    if (live) {
        for (uint32_t s = s0; s < s1; s++) {
            const uint32_t v = p[(size_t)s * ncoarse];
            p[(size_t)s * ncoarse] = run;
            run += v;
        }
        if (sg == 0) region_total[r] = tot;
    }
}
// single workgroup: exclusive scan of the region totals (any count) -> region_start[0..nregions], grand total
// OVERSIZED regions (skewed scalars: one bucket holds a large share of a window, e.g. the ones of a witness vector) used to be
// sorted by their single owner workgroup, element batch after element batch.  They are now cut into (region, batch) work items
// that extra workgroups share: k_coarse_starts lists them, the worker blocks of k_fine_sort count every batch into the region's
// global fine histogram, k_big_place turns the histogram into bucket offsets and places every batch (one global cursor add per
// bucket and batch).  `big` layout (u32): [0] items listed (may exceed BIG_MAX_ITEMS: regions that did not fit keep their owner),
// [1] table slots used, [16 ..) BIG_MAX_ITEMS x (region, batch), then per slot BIG_SLOT_WORDS = 512 counts + 512 cursors.
constexpr uint32_t BIG_NONE = 0xFFFFFFFFu;
constexpr uint32_t BIG_MAX_ITEMS = 4096;
constexpr uint32_t BIG_WORKERS_X = 16;  // worker blocks per window row (grid.y = sort windows), at least BIG_WORKERS_MIN in all
constexpr uint32_t BIG_WORKERS_MIN = 512;
constexpr uint32_t BIG_ITEMS_OFF = 16, BIG_TAB_OFF = BIG_ITEMS_OFF + 2 * BIG_MAX_ITEMS;
constexpr uint32_t BIG_SLOT_WORDS = 1024;  // == 2 * FINE_BINS_MAX: counts, then cursors, of up to 512 fine bins
constexpr size_t BIG_WORDS = (size_t)BIG_TAB_OFF + (size_t)BIG_MAX_ITEMS * BIG_SLOT_WORDS;

__global__ void k_coarse_starts(const uint32_t* __restrict__ region_total, uint32_t* __restrict__ region_start, uint32_t nregions,
                                uint32_t* __restrict__ total_out, uint32_t* __restrict__ offsets_end, uint32_t* __restrict__ bigslot,
                                uint32_t* __restrict__ big, uint32_t big_threshold, uint32_t batch_cap) {
    if (threadIdx.x < 2) big[threadIdx.x] = 0;
    for (uint32_t k = threadIdx.x; k < 2 * BIG_MAX_ITEMS; k += blockDim.x) big[BIG_ITEMS_OFF + k] = BIG_NONE;
    __syncthreads();
    uint32_t running = 0;
    for (uint32_t start = 0; start < nregions; start += SCAN_TILE) {  // SCAN_ITEMS consecutive regions per thread and round
        const uint32_t base = start + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            v[k] = base + k < nregions ? region_total[base + k] : 0u;
            sum += v[k];
        }
        uint32_t total;
        uint32_t ex = running + block_exclusive_scan(sum, &total);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (base + k < nregions) {
                region_start[base + k] = ex;
                uint32_t slot = BIG_NONE;
                if (v[k] > big_threshold) {  // rare: list the region's batches for the worker blocks
                    const uint32_t nit = (v[k] + batch_cap - 1) / batch_cap, ib = atomicAdd(&big[0], nit);  // batches of one staging area
                    if (ib + nit <= BIG_MAX_ITEMS) {
                        slot = atomicAdd(&big[1], 1u);
                        for (uint32_t z = 0; z < nit; z++) {
                            big[BIG_ITEMS_OFF + 2 * (ib + z)] = base + k;
                            big[BIG_ITEMS_OFF + 2 * (ib + z) + 1] = z;
                        }
                    }
                }
                bigslot[base + k] = slot;
            }
            ex += v[k];
        }
        running += total;
    }
    if (threadIdx.x == 0) {
        region_start[nregions] = running;
        *total_out = running;    // number of sorted entries (non-zero digits)
        *offsets_end = running;  // offsets[W * nb]
    }
    __syncthreads();  // big[1] is final: the whole workgroup clears the count / cursor tables of the slots in use
    const uint32_t nslots = min(big[1], BIG_MAX_ITEMS);
    for (size_t q = threadIdx.x; q < (size_t)nslots * BIG_SLOT_WORDS; q += blockDim.x) big[BIG_TAB_OFF + q] = 0;
}
template <bool D16>
__global__ void __launch_bounds__(TILE_BLOCK) k_coarse_scatter(const void* __restrict__ digits, uint32_t drow, const uint32_t* __restrict__ counts,
                                                               const uint32_t* __restrict__ region_start, uint32_t* __restrict__ tmp,
                                                               uint32_t n, uint32_t fine_bits, uint32_t idx_bits, uint32_t ncoarse,
                                                               uint32_t NS) {
    __shared__ uint32_t s_stage[SUBTILE];
    __shared__ uint32_t s_lstart[COARSE_BINS_MAX + 1];  // local start of each bin's run in s_stage
    __shared__ uint32_t s_cur[COARSE_BINS_MAX];
    __shared__ uint32_t s_gbase[COARSE_BINS_MAX];        // where the run goes in tmp
    const uint32_t st = blockIdx.x, w = blockIdx.y;
    const uint32_t i0 = st * SUBTILE, i1 = min(n, i0 + SUBTILE);
    uint32_t dg[SUBTILE / TILE_BLOCK];
    load_subtile_digits<D16>(digits, (size_t)w * drow, i0, i1, dg);
    // this sub-tile's bin counts = differences of the prefixes over sub-tiles (last sub-tile: region total - prefix)
    uint32_t cnt = 0;
    if (threadIdx.x < ncoarse) {
        const size_t r = (size_t)w * ncoarse + threadIdx.x;
        const uint32_t* cw = counts + (size_t)w * NS * ncoarse + threadIdx.x;
        const uint32_t pre = cw[(size_t)st * ncoarse];
        const uint32_t nxt = (st + 1 < NS) ? cw[(size_t)(st + 1) * ncoarse] : (region_start[r + 1] - region_start[r]);
        cnt = nxt - pre;
        s_gbase[threadIdx.x] = region_start[r] + pre;
    }
    // prefix of the bin counts: every wavefront scans its 64 bins with shuffles, the 16 wavefront totals are joined through
    // LDS (two barriers; a Hillis-Steele scan over 1024 entries took twenty)
    __shared__ uint32_t s_wtot[TILE_BLOCK / 64];
    {
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        uint32_t x = cnt;  // 0 for threads beyond ncoarse
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(x, d, 64);
            if (lane >= (uint32_t)d) x += y;
        }
        if (lane == 63) s_wtot[wv] = x;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t k = 0; k < wv; k++) base += s_wtot[k];
        const uint32_t incl = base + x;
        if (threadIdx.x < ncoarse) {
            s_lstart[threadIdx.x + 1] = incl;  // start of the next bin's run in s_stage
            s_cur[threadIdx.x] = incl - cnt;   // this bin's cursor
        }
        if (threadIdx.x == 0) s_lstart[0] = 0;
        __syncthreads();
    }
    const uint32_t fine_mask = (1u << fine_bits) - 1u;
    // the sub-tile's 16 digits per thread were loaded before the prefix phase (loads in flight across the barriers)
#pragma unroll
    for (int k = 0; k < SUBTILE / TILE_BLOCK; k++) {
        const uint32_t d = dg[k];
        if (d == DIGIT_SKIP) continue;
        const uint32_t i = subtile_element<D16>(i0, k);
        uint32_t bkt = d & ~SIGN_BIT;
        uint32_t pos = lds_inc(s_cur, bkt >> fine_bits);
        s_stage[pos] = (i & ((1u << idx_bits) - 1u)) | ((bkt & fine_mask) << idx_bits) | (d & SIGN_BIT);
    }
    __syncthreads();
    // copy-out: a group of gw lanes takes a bin at a time and writes its run with consecutive lanes on consecutive words.  gw follows
    // the mean run length SUBTILE / ncoarse: a whole wavefront per bin with 256 bins (runs of 64), 16 lanes with 1024 bins (runs of
    // 16: a wavefront per bin left three quarters of its lanes idle through 64 trips -- 52 us for the 13.6 M entries of a table MSM)
    const uint32_t gw = ncoarse <= 256 ? 64u : ncoarse <= 512 ? 32u : 16u;
    const uint32_t gl = threadIdx.x % gw, grp = threadIdx.x / gw, ngrp = TILE_BLOCK / gw;
    for (uint32_t b = grp; b < ncoarse; b += ngrp) {
        const uint32_t ls = s_lstart[b], le = s_lstart[b + 1], gb = s_gbase[b];
        for (uint32_t k = ls + gl; k < le; k += gw) tmp[gb + (k - ls)] = s_stage[k];
    }
}
constexpr int FINE_PER_THREAD = 16;  // elements that live in registers between the two phases
constexpr uint32_t FINE_BINS_MAX = 512;  // fine part of the bucket index: up to 9 bits (windows of 2^19 buckets: 10 coarse + 9 fine)
constexpr uint32_t SUPER_MAX = 16;       // super-tiles of 2^idx_bits elements a sort window may span
static_assert(BIG_SLOT_WORDS == 2 * FINE_BINS_MAX, "oversized-region tables hold counts + cursors of every fine bin");
// An element travels through the staging array as  (position in its sort window) mod 2^idx_bits | fine << idx_bits | sign << 31 with
// idx_bits = 31 - fine_bits.  EVERY 32-bit pattern is a valid element -- 0xFFFFFFFF is the last position of a super-tile with all fine
// bits set and a negative digit, which the top window of a full table produces on purpose (digit * 2^top_shift - 1 ends in ones):
// whether a register slot holds an element is decided by its index, never by a sentinel value (round 3; a c = 17 split table at 2^21
// points lost exactly that one entry).  A sort window longer than 2^idx_bits elements (the shared bucket array of a window table: 13 x 2^20
// entries with 9 fine bits) is cut into SUPER-TILES of 2^idx_bits elements; the dropped high bits of the position are recovered from
// where the element sits in its region: k_coarse_scatter lays a region out sub-tile after sub-tile, so the region's elements of
// super-tile h are exactly the positions [bnd[h-1], bnd[h]) with bnd[h] = counts[w][(h+1) * sub_per_super][bin] -- the exclusive prefix
// k_coarse_prefix left there.  nsuper - 1 <= 15 comparisons per element, no extra bytes.
struct sort_hi {
    const uint32_t* counts;  // [w][sub-tile][bin] prefixes (k_coarse_prefix)
    uint32_t NS;             // sub-tiles per sort window
    uint32_t nsuper;         // super-tiles per sort window (1 = positions fit idx_bits: nothing to recover)
    uint32_t sub_per_super;  // 2^idx_bits / SUBTILE
};
// called by the whole workgroup (barrier inside): boundaries of region (w, cb) into s_bnd
__device__ __forceinline__ void sort_hi_load(const sort_hi& hi, uint32_t w, uint32_t cb, uint32_t ncoarse, uint32_t* s_bnd) {
    if (hi.nsuper <= 1) return;  // uniform
    __syncthreads();
    if (threadIdx.x + 1 < hi.nsuper) s_bnd[threadIdx.x] = hi.counts[((size_t)w * hi.NS + (size_t)(threadIdx.x + 1) * hi.sub_per_super) * ncoarse + cb];
    __syncthreads();
}
__device__ __forceinline__ uint32_t sort_hi_of(const sort_hi& hi, const uint32_t* s_bnd, uint32_t pos, uint32_t idx_bits) {
    uint32_t h = 0;
    for (uint32_t k = 0; k + 1 < hi.nsuper; k++) h += pos >= s_bnd[k] ? 1u : 0u;  // uniform trip count
    return h << idx_bits;
}
// exclusive prefix over the first nfine <= FINE_BINS_MAX threads' values (the rest pass 0); whole workgroup, blockDim >= nfine
__device__ __forceinline__ uint32_t fine_scan(uint32_t cnt, uint32_t* s_wtot /* [FINE_BINS_MAX / 64] */) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63 && wv < FINE_BINS_MAX / 64) s_wtot[wv] = x;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wv && k < FINE_BINS_MAX / 64; k++) base += s_wtot[k];
    __syncthreads();
    return base + x - cnt;
}
// FINE_BLOCK threads stage up to FINE_BLOCK * 16 elements: 1024 threads (16384 elements, 64 KB of LDS) for the regions of a
// large instance, 256 threads (4096 elements) when regions only hold a few hundred elements -- a 1024-thread workgroup
// costs its dispatch and barriers whatever it sorts, and below 2^18 points those were most of this kernel's time.
// nfine = 2^fine_bits <= FINE_BINS_MAX fine bins; FINE_BLOCK >= nfine (the host picks the block).
template <int FINE_BLOCK>
__global__ void __launch_bounds__(FINE_BLOCK) k_fine_sort(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ region_start,
                                                          uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted, uint32_t nb,
                                                          uint32_t fine_bits, uint32_t idx_bits, uint32_t ncoarse,
                                                          const uint32_t* __restrict__ bigslot, uint32_t* __restrict__ big, sort_hi hi,
                                                          uint32_t* __restrict__ ne_region /* per (window, coarse bin): its buckets that hold entries (k_place_count adds them up: msmplan::effective_pmax) */) {
    __shared__ uint32_t s_cur[FINE_BINS_MAX];
    __shared__ uint32_t s_wtot[FINE_BINS_MAX / 64];
    __shared__ uint32_t s_bnd[SUPER_MAX];
    constexpr uint32_t CAP = (uint32_t)FINE_BLOCK * FINE_PER_THREAD;
    __shared__ uint32_t s_out[CAP];
    const uint32_t nfine = 1u << fine_bits;
    if (blockIdx.x >= ncoarse) {  // worker block: count the batches of oversized regions into their global fine histograms
        const uint32_t nitems = min(big[0], BIG_MAX_ITEMS), fmask = nfine - 1u;
        const uint32_t wx = gridDim.x - ncoarse;  // worker blocks per window row
        for (uint32_t it = (blockIdx.x - ncoarse) + wx * blockIdx.y; it < nitems; it += wx * gridDim.y) {
            const uint32_t rr = big[BIG_ITEMS_OFF + 2 * it], z = big[BIG_ITEMS_OFF + 2 * it + 1];
            if (rr == BIG_NONE) continue;  // uniform
            const uint32_t b0 = region_start[rr] + z * CAP, b1 = min(region_start[rr + 1], b0 + CAP);
            __syncthreads();
            for (uint32_t f = threadIdx.x; f < nfine; f += FINE_BLOCK) s_cur[f] = 0;
            __syncthreads();
            uint32_t eb[FINE_PER_THREAD];
#pragma unroll
            for (int k = 0; k < FINE_PER_THREAD; k++) {
                uint32_t j = b0 + threadIdx.x + k * FINE_BLOCK;
                eb[k] = j < b1 ? tmp[j] : 0u;
            }
#pragma unroll
            for (int k = 0; k < FINE_PER_THREAD; k++)
                if (b0 + threadIdx.x + k * FINE_BLOCK < b1) lds_inc(s_cur, (eb[k] >> idx_bits) & fmask);
            __syncthreads();
            for (uint32_t f = threadIdx.x; f < nfine; f += FINE_BLOCK)
                if (s_cur[f]) atomicAdd(&big[BIG_TAB_OFF + (size_t)bigslot[rr] * BIG_SLOT_WORDS + f], s_cur[f]);
        }
        return;
    }
    const uint32_t cb = blockIdx.x, w = blockIdx.y;
    const uint32_t r = w * ncoarse + cb;
    __shared__ uint32_t s_ne;
    if (threadIdx.x == 0) s_ne = 0;
    if (bigslot[r] != BIG_NONE) {  // oversized region: its batches are shared out to the worker blocks (uniform per workgroup)
        if (threadIdx.x == 0) ne_region[r] = 0;  // (its buckets are not counted: skewed scalars keep the plan's pmax more often)
        return;
    }
    const uint32_t rs = region_start[r], re = region_start[r + 1], S = re - rs;
    const uint32_t fine_mask = nfine - 1u, idx_mask = (1u << idx_bits) - 1u;
    const bool staged = S <= CAP;
    for (uint32_t f = threadIdx.x; f < nfine; f += FINE_BLOCK) s_cur[f] = 0;
    sort_hi_load(hi, w, cb, ncoarse, s_bnd);
    __syncthreads();
    uint32_t e[FINE_PER_THREAD];
    if (staged) {  // the whole region in registers: FINE_PER_THREAD independent loads per thread
#pragma unroll
        for (int k = 0; k < FINE_PER_THREAD; k++) {
            uint32_t j = rs + threadIdx.x + k * FINE_BLOCK;
            e[k] = j < re ? tmp[j] : 0u;
        }
#pragma unroll
        for (int k = 0; k < FINE_PER_THREAD; k++)
            if (rs + threadIdx.x + k * FINE_BLOCK < re) lds_inc(s_cur, (e[k] >> idx_bits) & fine_mask);
    } else {  // oversized region (one bucket holds a large share of the window): FINE_PER_THREAD loads in flight per thread
        for (uint32_t base = rs; base < re; base += CAP) {
#pragma unroll
            for (int k = 0; k < FINE_PER_THREAD; k++) {
                uint32_t j = base + threadIdx.x + k * FINE_BLOCK;
                e[k] = j < re ? tmp[j] : 0u;
            }
#pragma unroll
            for (int k = 0; k < FINE_PER_THREAD; k++)
                if (base + threadIdx.x + k * FINE_BLOCK < re) lds_inc(s_cur, (e[k] >> idx_bits) & fine_mask);
        }
    }
    __syncthreads();
    // exclusive prefix of the <= 512 fine counts: every wavefront scans its 64 counts with shuffles, the wavefront totals are joined
    // through LDS (a Hillis-Steele scan in LDS cost this kernel 14 workgroup barriers of 1024 threads)
    {   // the region's non-empty buckets (a plain store per region: device-scope adds on shared words cost the sort 5-17 us)
        const unsigned long long any = __ballot(threadIdx.x < nfine && s_cur[threadIdx.x] != 0u);
        if ((threadIdx.x & 63u) == 0u && any) atomicAdd(&s_ne, (uint32_t)__popcll(any));  // (LDS; zeroed before the first barrier)
    }
    const uint32_t ex = fine_scan(threadIdx.x < nfine ? s_cur[threadIdx.x] : 0u, s_wtot);  // (barriers inside)
    if (threadIdx.x == 0) ne_region[r] = s_ne;
    if (threadIdx.x < nfine) {
        s_cur[threadIdx.x] = ex;                                                       // local cursor
        offsets[(size_t)w * nb + ((size_t)cb << fine_bits) + threadIdx.x] = rs + ex;  // the bucket's CSC column pointer
    }
    __syncthreads();
    if (staged) {
#pragma unroll
        for (int k = 0; k < FINE_PER_THREAD; k++) {
            if (rs + threadIdx.x + k * FINE_BLOCK >= re) continue;
            uint32_t pos = lds_inc(s_cur, (e[k] >> idx_bits) & fine_mask);
            s_out[pos] = ((e[k] & idx_mask) + sort_hi_of(hi, s_bnd, threadIdx.x + k * FINE_BLOCK, idx_bits)) | (e[k] & SIGN_BIT);
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < S; k += FINE_BLOCK) sorted[rs + k] = s_out[k];
    } else {  // skewed data: the region does not fit LDS, place directly (aggregated cursors hand consecutive lanes
              // consecutive slots, so the stores of a hot bucket are still coalesced)
        for (uint32_t base = rs; base < re; base += CAP) {
#pragma unroll
            for (int k = 0; k < FINE_PER_THREAD; k++) {
                uint32_t j = base + threadIdx.x + k * FINE_BLOCK;
                e[k] = j < re ? tmp[j] : 0u;
            }
#pragma unroll
            for (int k = 0; k < FINE_PER_THREAD; k++) {
                if (base + threadIdx.x + k * FINE_BLOCK >= re) continue;
                uint32_t pos = lds_inc(s_cur, (e[k] >> idx_bits) & fine_mask);
                sorted[rs + pos] = ((e[k] & idx_mask) + sort_hi_of(hi, s_bnd, base - rs + threadIdx.x + k * FINE_BLOCK, idx_bits)) | (e[k] & SIGN_BIT);
            }
        }
    }
}

// K2 phase 3 (fallback path): place every (point, sign) at offsets[bucket] + rank
__global__ void k_scatter(const uint32_t* __restrict__ digits, const uint32_t* __restrict__ ranks,
                          const uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted, uint32_t n, uint32_t nb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t w = blockIdx.y;
    if (i >= n) return;
    size_t o = (size_t)w * n + i;
    uint32_t d = digits[o];
    if (d == DIGIT_SKIP) return;
    uint32_t bkt = d & ~SIGN_BIT;
    sorted[offsets[(size_t)w * nb + bkt] + ranks[o]] = i | (d & SIGN_BIT);
}

// ---------------------------------------------------------------------------------------------
// K3: bucket accumulation (round 4 form: PIECES SORTED BY LENGTH).
//
// Rounds 1-3 cut the sorted (point, sign) array into fixed-length CHUNKS of L entries, one per thread, whatever buckets they belonged to
// (the reference gives one thread a whole bucket pair, smvp.metal:46-71, and serialises on the longest).  That balances any distribution,
// but (i) almost every bucket is cut by a chunk border (mean bucket = L entries at 2^20 points), so a k_combine pass had to add ~one pair
// of partial sums per bucket (54 us at 2^20, 46 us at 2^17) after heads and tails travelled through HBM, and (ii) the launch was 1.33 rounds
// of identical workgroups, whose last wavefront per SIMD ran alone for a whole chunk (tools/ab_libs.py chunk sweep, profiles/
// r4_pieces_vs_chunks.txt: 2.47 Mcycles at L = 64 against 2.28 at L = 22, where k_combine then cost 0.44 ms instead of 0.20).
// Here a work item is a PIECE: a whole bucket, or a part of one that is longer than `pmax` entries (mean occupancy + two standard
// deviations, msmplan::make_piece_plan: the few per cent of Poisson buckets beyond it become a run of pmax and a short rest) -- runs
// of pmax up to 8 x pmax, runs of `psplit` entries beyond (skewed scalars: psplit plays the part the chunk length used to, short enough that
// an instance made of long buckets only still yields ~2^19 pieces and that the longest item of an under-filled launch does not run alone
// for long).  pmax must stay well below a SIMD lane's share of the launch (256 entries at 2^20): the resident workgroups of the first
// round are placed three per CU whatever their lengths, and with whole top-window buckets of ~128-175 entries among ordinary ones of ~64
// the heaviest CUs carried 13 % more than the mean (2.78 against 2.43 Mcycles, profiles/r4_pieces_vs_chunks.txt).  The pieces are counting-sorted by
// length, longest first, so the 64 lanes of a wavefront run the same trip count (the property the chunks were built for), the long items
// start first and the launch ends on its shortest ones (LPT order: the rests of the cut buckets), and a bucket that is one piece is
// written straight to its slot.  (First form: the top window of a plan was not spread yet and its twice-as-full buckets, cut at 2 x the mean,
// supplied the rests; now the decomposition spreads it -- k_decompose(_glv) -- and the cap itself does.)  Split buckets leave partial sums in `partials` and are listed for k_combine_pieces
// (two pieces: one addition -- eight lanes, or one lane per bucket when tens of thousands are listed; 3..7 pieces: a chain of eight-lane additions; 8 or more: a wavefront per bucket up to
// 512 pieces, LDS trees of eight-lane additions per 1024-piece segment beyond).
// Measured against the chunk form, same build, one box (profiles/r4_pieces_vs_chunks.txt): 2^20 1.557 -> 1.488 ms, 2^17 0.470 -> 0.440,
// 2^22 5.32 -> 5.00.
constexpr uint32_t LONG_SPAN = 8;     // split buckets of this many pieces or more are folded by whole workgroups (k_combine_pieces)
constexpr uint32_t LONG_SEG = 1024;   // pieces of a long bucket folded by one workgroup (round 6: was 2048 -- with 256-thread workgroups four serial one-lane additions per lane instead of eight)
constexpr uint32_t WAVE_ITEM_MAX = 512;      // long buckets of at most this many pieces are folded by one wavefront (k_combine_pieces pass A) ...
constexpr uint32_t COMBINE_BLOCK = 256;      // threads of a k_combine_pieces workgroup: at three wavefronts per SIMD three of them share a CU (512 threads at the kernel's 175 VGPRs: ONE)
constexpr uint32_t WAVE_ITEM_RECORDS = 64;   // ... in a 64-record region of the workgroup's LDS (4 wavefronts x 64 = WIDE_TREE_MAX records)
constexpr uint32_t LONG_BLOCKS = 1024, MID_BLOCKS = 1024;  // k_combine_pieces' grid: (bucket, segment) items grid-stride over the first, listed buckets over the rest
constexpr uint32_t MID2_BLOCKS = 1024;       // ... and the workgroups that fold the two-piece buckets one LANE per bucket when there are more than MID_LANE_MIN of them
constexpr uint32_t MID_LANE_MIN = 32768;     // (more than ~1.3 rounds of resident eight-lane groups)
// (PIECE_BINS, at the top of this file: pmax <= PIECE_BINS, one histogram bin per piece length)
constexpr uint32_t PF_WHOLE = 0x80000000u;            // piece.z: the bucket is this one piece -> the sum goes to buckets[k]
constexpr uint32_t PF_FIRST = 0x40000000u;            // piece.z: first piece of a split bucket (INTO: starts from the bucket's old value)
constexpr uint32_t PF_LEN_MASK = 0x00FFFFFFu;

// The run length of very long buckets as the DEVICE sees the instance (round 6).  The host plans `psplit` for the most entries the instance can have (W * n_v); an
// instance whose scalars are mostly zero or tiny sorts far fewer, and its one or two hot buckets, cut into runs planned for a full instance, are what the
// launch then waits for: a witness-like mix at 2^20 (40 % zeros, 30 % ones: 3.8 M entries instead of 16.8 M, one bucket of 314 000) accumulated in 0.25 ms with
// runs of 32 and in 0.35 with the 64 a full instance wants.  The kernels that cut buckets shorten the runs to the largest power of two <= entries >> shift
// (never below 8, never above the host's plan -- the workspace is sized for that).  The argument packs: bits 0-15 the host's psplit, 16-23 the shift,
// bit 31 = fixed (a forced length of the tests: taken as it is).
constexpr uint32_t PSPLIT_FIXED = 0x80000000u;
__host__ __device__ __forceinline__ uint32_t psplit_arg(uint32_t psplit, uint32_t shift, bool fixed) {
    return (psplit & 0xFFFFu) | ((shift & 0xFFu) << 16) | (fixed ? PSPLIT_FIXED : 0u);
}
constexpr uint32_t PSPLIT_EQUAL = 0x40000000u;  // marker on the DECODED run length: buckets up to LONG_SPAN x pmax are cut into equal runs (piece_split)
// the plan's lengths -> the lengths for THIS instance: pmax raised for sparse instances (msmplan::effective_pmax; what is longer is then cut into equal runs),
// psplit shortened for instances of few entries (effective_psplit).  Every kernel that cuts buckets calls it first, with the same flag words.
__device__ __forceinline__ void decode_piece_lengths(uint32_t& pmax, uint32_t& psplit /* in: packed argument; out: run length (| PSPLIT_EQUAL) */, uint32_t entries,
                                                     uint32_t nonempty) {
    const uint32_t arg = psplit;
    uint32_t run = arg & 0xFFFFu;
    if (!(arg & PSPLIT_FIXED)) {
        const uint32_t raised = msmplan::effective_pmax(pmax, entries, nonempty);
        run = msmplan::effective_psplit(run, (arg >> 16) & 0xFFu, entries);
        if (raised > pmax) pmax = raised, run |= PSPLIT_EQUAL;
    }
    psplit = run;
}

// how a bucket of sz entries is cut: 1 piece up to pmax entries; up to LONG_SPAN * pmax entries into runs of pmax and a remainder (one
// addition to fold them; the short rests are what the launch ends on: cut into EQUAL halves instead, buckets of twice the mean cost
// k_accumulate_pieces 2.63 instead of 2.40 Mcycles at 2^20, the smallest items then being ~35 entries long); beyond that into runs of psplit.
// Returns the number of pieces m; pieces 0 .. m-2 hold *q entries, the last one the rest.
__device__ __forceinline__ uint32_t piece_split(uint32_t sz, uint32_t pmax, uint32_t psplit, uint32_t* q) {
    if (sz <= pmax) {
        *q = sz;
        return 1u;
    }
    if (sz > LONG_SPAN * pmax) *q = psplit & ~PSPLIT_EQUAL;
    else if (psplit & PSPLIT_EQUAL) *q = (sz + (sz + pmax - 1) / pmax - 1) / ((sz + pmax - 1) / pmax);  // ceil(sz / m), m = ceil(sz / pmax): <= pmax
    else *q = pmax;
    return (sz + *q - 1) / *q;
}

// exclusive scan over the 1024 threads of a workgroup (v -> sum of the values of lower threads); *total = sum of all
__device__ __forceinline__ uint32_t block1024_exclusive_scan(uint32_t v, uint32_t* s_wsum /* 16 words of LDS */, uint32_t* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wsum[wid] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < wid) base += s_wsum[k];
        tot += s_wsum[k];
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// pass 1: histogram of the piece lengths (LDS per workgroup, one global add per non-empty bin), the identity for empty buckets, a run of
// partial-sum slots and a list entry for every split bucket.  One thread per bucket.  The steps are functions because two kernels run them:
// k_piece_count (every sort path) and k_place_count (the two-level sort: the same launch also places the batches of oversized regions).
struct piece_tally {
    uint32_t kindl, slot, nseg, m, pslot;  // kindl: 0 = mid list (3 .. LONG_SPAN-1 pieces), 1 = long list, 2 = none, 3 = list of the two-piece buckets
};
__device__ __forceinline__ void piece_tally_begin(uint32_t* s_hist, uint32_t* s_n, uint32_t pmax) {  // whole workgroup
    for (uint32_t i = threadIdx.x; i <= pmax; i += blockDim.x) s_hist[i] = 0;
    if (threadIdx.x < 4) s_n[threadIdx.x] = 0;
    __syncthreads();
}
__device__ __forceinline__ piece_tally piece_tally_bucket(uint32_t k, uint32_t sz, uint32_t pmax, uint32_t psplit, uint32_t* s_hist, uint32_t* s_n,
                                                          uint32_t* __restrict__ buckets, uint32_t into) {
    piece_tally t{2u, 0u, 1u, 0u, 0u};
    if (sz == 0) {
        if (!into) store_xyzz(buckets + (size_t)k * XW, xyzz_identity());  // (into: the bucket keeps the earlier chunks' sum)
    } else {
        uint32_t q;
        t.m = piece_split(sz, pmax, psplit, &q);
        if (t.m == 1) {
            atomicAdd(&s_hist[sz], 1u);
        } else {
            atomicAdd(&s_hist[q], t.m - 1);
            atomicAdd(&s_hist[sz - (t.m - 1) * q], 1u);
            t.pslot = atomicAdd(&s_n[2], t.m);  // (LDS; the short top window of a small instance splits ~8000 buckets: one device-scope
                                                // add each on ONE word took k_piece_count 19 us at 2^17 where 2^20 takes 8)
            if (t.m >= LONG_SPAN) {
                t.kindl = 1;
                t.nseg = (t.m + LONG_SEG - 1) / LONG_SEG;
            } else {
                t.kindl = t.m == 2 ? 3u : 0u;
            }
        }
    }
    if (t.kindl != 2) t.slot = atomicAdd(&s_n[t.kindl], t.nseg);
    return t;
}
// the workgroup's share of the lists, the partial-sum slots and every histogram bin: one device-scope add each (whole workgroup)
__device__ __forceinline__ void piece_tally_reserve(const uint32_t* s_hist, const uint32_t* s_n, uint32_t* s_base, uint32_t* __restrict__ hist,
                                                    uint32_t* __restrict__ flags, uint32_t pmax) {
    __syncthreads();
    if (threadIdx.x == 0 && s_n[0]) s_base[0] = atomicAdd(flags + FLAG_MID, s_n[0]);
    if (threadIdx.x == 1 && s_n[1]) s_base[1] = atomicAdd(flags + FLAG_LONG, s_n[1]);
    if (threadIdx.x == 2 && s_n[2]) s_base[2] = atomicAdd(flags + FLAG_PARTIALS, s_n[2]);
    if (threadIdx.x == 3 && s_n[3]) s_base[3] = atomicAdd(flags + FLAG_MID2, s_n[3]);
    for (uint32_t i = threadIdx.x; i <= pmax; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
    __syncthreads();
}
__device__ __forceinline__ void piece_tally_publish(uint32_t k, const piece_tally& t, const uint32_t* s_base, uint32_t* __restrict__ pbase,
                                                    uint32_t* __restrict__ mid_list, uint32_t* __restrict__ long_list, uint32_t total_buckets) {
    // mid_list holds TWO lists: the buckets of exactly two pieces from entry 0 (one addition each: k_combine_pieces folds them one LANE per bucket when
    // there are many), those of 3 .. LONG_SPAN-1 pieces (a dependent chain per bucket: always eight lanes) from entry total_buckets
    if (t.m > 1) pbase[k] = s_base[2] + t.pslot;
    if (t.kindl == 3) {
        mid_list[s_base[3] + t.slot] = k;
    } else if (t.kindl == 0) {
        mid_list[total_buckets + s_base[0] + t.slot] = k;
    } else if (t.kindl == 1) {  // one entry per LONG_SEG pieces: (bucket, segment)
        for (uint32_t j = 0; j < t.nseg; j++) {
            long_list[2 * (size_t)(s_base[1] + t.slot + j)] = k;
            long_list[2 * (size_t)(s_base[1] + t.slot + j) + 1] = j;
        }
    }
}
__global__ void __launch_bounds__(1024) k_piece_count(const uint32_t* __restrict__ offsets, uint32_t total_buckets, uint32_t pmax, uint32_t psplit,
                                                     uint32_t* __restrict__ hist, uint32_t* __restrict__ flags, uint32_t* __restrict__ long_list,
                                                     uint32_t* __restrict__ mid_list, uint32_t* __restrict__ pbase, uint32_t* __restrict__ buckets,
                                                     uint32_t into) {
    __shared__ uint32_t s_hist[PIECE_BINS + 1];
    __shared__ uint32_t s_n[4], s_base[4];  // [0] mid list, [1] long list, [2] partial-sum slots, [3] two-piece list: reserved once per workgroup
    if (blockIdx.x == 0 && threadIdx.x == 0)  // mixed additions executed so far (msm_timings_t.num_adds), all chunks of a streamed MSM
        *reinterpret_cast<unsigned long long*>(flags + FLAG_ADDS64) += (unsigned long long)flags[FLAG_PAIRS];
    decode_piece_lengths(pmax, psplit, flags[FLAG_PAIRS], flags[FLAG_NONEMPTY]);
    piece_tally_begin(s_hist, s_n, pmax);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    piece_tally t{2u, 0u, 1u, 0u, 0u};
    if (k < total_buckets) t = piece_tally_bucket(k, offsets[k + 1] - offsets[k], pmax, psplit, s_hist, s_n, buckets, into);
    piece_tally_reserve(s_hist, s_n, s_base, hist, flags, pmax);
    if (k < total_buckets) piece_tally_publish(k, t, s_base, pbase, mid_list, long_list, total_buckets);
}

// The two-level sort's LAST launch (round 6): k_big_place and k_piece_count in one.  On uniform scalars no region is oversized and k_big_place was
// an empty launch in the middle of the dependent chain -- 4.5 us at every size.  Workgroups [0, count_blocks) tally 1024
// buckets each, as k_piece_count does, leaving out the buckets of oversized regions (their column pointers are not written yet); the other
// wx * sW workgroups place the batches of those regions as k_big_place did, and the workgroup that publishes a region's column pointers (batch 0)
// tallies its buckets from the counts it holds.  A bucket's END is its successor's column pointer -- which for the last bucket of a region
// belongs to the next region and may be written by this very launch: the region table has it.
constexpr uint32_t PLACE_COUNT_SPAN = 1024;  // buckets per tallying workgroup (one histogram flush per workgroup, as with k_piece_count's 1024 threads)
template <int FINE_BLOCK>
__global__ void __launch_bounds__(FINE_BLOCK) k_place_count(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ region_start,
                                                            uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted, uint32_t nb,
                                                            uint32_t fine_bits, uint32_t idx_bits, uint32_t ncoarse,
                                                            const uint32_t* __restrict__ bigslot, uint32_t* __restrict__ big, sort_hi hi,
                                                            uint32_t count_blocks, uint32_t wx, uint32_t sW, uint32_t total_buckets, uint32_t pmax,
                                                            uint32_t psplit, uint32_t* __restrict__ hist, uint32_t* __restrict__ flags,
                                                            uint32_t* __restrict__ long_list, uint32_t* __restrict__ mid_list,
                                                            uint32_t* __restrict__ pbase, uint32_t* __restrict__ buckets, uint32_t into) {
    __shared__ uint32_t s_hist[PIECE_BINS + 1];
    __shared__ uint32_t s_n[4], s_pbase[4];
    // the non-empty buckets k_fine_sort counted per region (region_start + sW * ncoarse + 2: behind the region table)
    __shared__ uint32_t s_nonempty;
    if (threadIdx.x == 0) s_nonempty = 0;
    __syncthreads();
    {
        const uint32_t nregions = sW * ncoarse;
        const uint32_t* ne_region = region_start + nregions + 2;
        uint32_t part = 0;
        for (uint32_t i = threadIdx.x; i < nregions; i += FINE_BLOCK) part += ne_region[i];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
        if ((threadIdx.x & 63u) == 0u && part) atomicAdd(&s_nonempty, part);
    }
    __syncthreads();
    const uint32_t nonempty = s_nonempty;
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[FLAG_NONEMPTY] = nonempty;  // (k_piece_scatter and k_combine_pieces read it there)
    decode_piece_lengths(pmax, psplit, flags[FLAG_PAIRS], nonempty);
    const uint32_t nfine = 1u << fine_bits, fmask = nfine - 1u;
    if (blockIdx.x < count_blocks) {
        if (blockIdx.x == 0 && threadIdx.x == 0)  // mixed additions executed so far (msm_timings_t.num_adds), all chunks of a streamed MSM
            *reinterpret_cast<unsigned long long*>(flags + FLAG_ADDS64) += (unsigned long long)flags[FLAG_PAIRS];
        piece_tally_begin(s_hist, s_n, pmax);
        constexpr int ITER = PLACE_COUNT_SPAN / FINE_BLOCK;
        piece_tally t[ITER];
#pragma unroll
        for (int it = 0; it < ITER; it++) {
            t[it] = piece_tally{2u, 0u, 1u, 0u, 0u};
            const uint32_t k = blockIdx.x * PLACE_COUNT_SPAN + it * FINE_BLOCK + threadIdx.x;
            if (k < total_buckets) {
                const uint32_t w = k / nb, b = k - w * nb, r = w * ncoarse + (b >> fine_bits);
                if (bigslot[r] == BIG_NONE) {
                    const uint32_t end = (b & fmask) == fmask ? region_start[r + 1] : offsets[k + 1];
                    t[it] = piece_tally_bucket(k, end - offsets[k], pmax, psplit, s_hist, s_n, buckets, into);
                }
            }
        }
        piece_tally_reserve(s_hist, s_n, s_pbase, hist, flags, pmax);
#pragma unroll
        for (int it = 0; it < ITER; it++) {
            const uint32_t k = blockIdx.x * PLACE_COUNT_SPAN + it * FINE_BLOCK + threadIdx.x;
            if (k < total_buckets) piece_tally_publish(k, t[it], s_pbase, pbase, mid_list, long_list, total_buckets);
        }
        return;
    }
    // second half of the oversized-region path: bucket offsets from the region's global fine histogram, then every batch is placed;
    // a batch reserves its share of each bucket with ONE global cursor add per bucket.
    constexpr uint32_t CAP = (uint32_t)FINE_BLOCK * FINE_PER_THREAD;
    __shared__ uint32_t s_ex[FINE_BINS_MAX], s_cnt[FINE_BINS_MAX], s_base[FINE_BINS_MAX], s_wtot[FINE_BINS_MAX / 64], s_bnd[SUPER_MAX];
    const uint32_t nitems = min(big[0], BIG_MAX_ITEMS);
    const uint32_t idx_mask = (1u << idx_bits) - 1u;
    for (uint32_t it = blockIdx.x - count_blocks; it < nitems; it += wx * sW) {
        const uint32_t rr = big[BIG_ITEMS_OFF + 2 * it], z = big[BIG_ITEMS_OFF + 2 * it + 1];
        if (rr == BIG_NONE) continue;  // uniform
        uint32_t* tab = big + BIG_TAB_OFF + (size_t)bigslot[rr] * BIG_SLOT_WORDS;
        const uint32_t r0 = region_start[rr], b0 = r0 + z * CAP, b1 = min(region_start[rr + 1], b0 + CAP);
        __syncthreads();
        sort_hi_load(hi, rr / ncoarse, rr % ncoarse, ncoarse, s_bnd);
        // exclusive prefix of the region's bucket counts
        const uint32_t mycnt = threadIdx.x < nfine ? tab[threadIdx.x] : 0u;
        const uint32_t ex = fine_scan(mycnt, s_wtot);
        if (threadIdx.x < nfine) {
            s_ex[threadIdx.x] = ex;
            s_cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        if (z == 0) {  // the first batch publishes the buckets' CSC column pointers and tallies the region's buckets (uniform)
            const uint32_t kb0 = (rr / ncoarse) * nb + ((rr % ncoarse) << fine_bits);
            if (threadIdx.x < nfine) offsets[(size_t)kb0 + threadIdx.x] = r0 + s_ex[threadIdx.x];
            piece_tally_begin(s_hist, s_n, pmax);
            piece_tally t{2u, 0u, 1u, 0u, 0u};
            if (threadIdx.x < nfine) t = piece_tally_bucket(kb0 + threadIdx.x, mycnt, pmax, psplit, s_hist, s_n, buckets, into);
            piece_tally_reserve(s_hist, s_n, s_pbase, hist, flags, pmax);
            if (threadIdx.x < nfine) piece_tally_publish(kb0 + threadIdx.x, t, s_pbase, pbase, mid_list, long_list, total_buckets);
        }
        uint32_t eb[FINE_PER_THREAD];
#pragma unroll
        for (int k = 0; k < FINE_PER_THREAD; k++) {
            uint32_t j = b0 + threadIdx.x + k * FINE_BLOCK;
            eb[k] = j < b1 ? tmp[j] : 0u;
        }
#pragma unroll
        for (int k = 0; k < FINE_PER_THREAD; k++)
            if (b0 + threadIdx.x + k * FINE_BLOCK < b1) lds_inc(s_cnt, (eb[k] >> idx_bits) & fmask);
        __syncthreads();
        if (threadIdx.x < nfine) {
            const uint32_t cn = s_cnt[threadIdx.x];
            s_base[threadIdx.x] = cn ? atomicAdd(&tab[FINE_BINS_MAX + threadIdx.x], cn) : 0u;  // this batch's slice of every bucket
            s_cnt[threadIdx.x] = 0;                                                             // becomes the local cursor
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FINE_PER_THREAD; k++) {
            if (b0 + threadIdx.x + k * FINE_BLOCK >= b1) continue;
            const uint32_t f = (eb[k] >> idx_bits) & fmask;
            const uint32_t pos = r0 + s_ex[f] + s_base[f] + lds_inc(s_cnt, f);
            sorted[pos] = ((eb[k] & idx_mask) + sort_hi_of(hi, s_bnd, b0 - r0 + threadIdx.x + k * FINE_BLOCK, idx_bits)) | (eb[k] & SIGN_BIT);
        }
    }
}

// pass 2: the piece list, longest pieces first.  Every workgroup derives the bins' start positions from the global histogram (an
// exclusive prefix in DESCENDING length order), reserves its share of every bin with one device-scope add per non-empty bin and places
// its pieces with LDS cursors.  piece = (bucket, first sorted entry, length | flags, partial slot).
__global__ void __launch_bounds__(1024) k_piece_scatter(const uint32_t* __restrict__ offsets, uint32_t total_buckets, uint32_t pmax, uint32_t psplit,
                                                       const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor,
                                                       const uint32_t* __restrict__ pbase, uint4* __restrict__ plist, uint32_t* __restrict__ flags) {
    __shared__ uint32_t s_start[PIECE_BINS + 1], s_cnt[PIECE_BINS + 1], s_cur[PIECE_BINS + 1];
    __shared__ uint32_t s_wsum[16];
    decode_piece_lengths(pmax, psplit, flags[FLAG_PAIRS], flags[FLAG_NONEMPTY]);
    {   // thread r owns the bin of length pmax - r (r < pmax); lengths above pmax do not exist, length 0 is never a piece
        const uint32_t len = threadIdx.x < pmax ? pmax - threadIdx.x : 0u;
        const uint32_t h = len ? hist[len] : 0u;
        uint32_t total;
        const uint32_t ex = block1024_exclusive_scan(h, s_wsum, &total);
        if (len) s_start[len] = ex;
        if (blockIdx.x == 0 && threadIdx.x == 0) flags[FLAG_PIECES] = total;  // number of pieces: k_accumulate_pieces' trip count
    }
    for (uint32_t i = threadIdx.x; i <= pmax; i += blockDim.x) s_cnt[i] = 0, s_cur[i] = 0;
    __syncthreads();
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t beg = 0, sz = 0, m = 0, rem = 0, q = 0;
    if (k < total_buckets) {
        beg = offsets[k];
        sz = offsets[k + 1] - beg;
        if (sz) {
            m = piece_split(sz, pmax, psplit, &q);
            rem = sz - (m - 1) * q;  // (m == 1: q == sz, rem == sz)
            if (m > 1) atomicAdd(&s_cnt[q], m - 1);
            atomicAdd(&s_cnt[rem], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= pmax; i += blockDim.x)
        if (s_cnt[i]) s_start[i] += atomicAdd(&cursor[i], s_cnt[i]);  // this workgroup's run inside bin i
    __syncthreads();
    // a bucket of many pieces (skewed scalars: all scalars equal -> 16 buckets of 2^20 entries = 32768 pieces each) is written out by the
    // whole workgroup: its owner only reserves the run and lists it (one thread writing 32768 records took 1 ms)
    constexpr uint32_t COOP_MIN = 64, COOP_MAX = 64;
    __shared__ uint32_t s_nco;
    __shared__ uint4 s_co[COOP_MAX];  // (bucket, first entry, pieces - 1, position of the run)
    if (threadIdx.x == 0) s_nco = 0;
    __syncthreads();
    if (m == 1) {
        const uint32_t pos = s_start[sz] + atomicAdd(&s_cur[sz], 1u);
        plist[pos] = make_uint4(k, beg, sz | PF_WHOLE, 0u);
    } else if (m > 1) {
        const uint32_t pb = pbase[k];
        uint32_t pos = s_start[q] + atomicAdd(&s_cur[q], m - 1);  // the m - 1 full pieces: one reservation
        uint32_t slot = COOP_MAX;
        if (m >= COOP_MIN) slot = atomicAdd(&s_nco, 1u);
        if (slot < COOP_MAX) {
            s_co[slot] = make_uint4(k, beg, m - 1, pos);
        } else {
            for (uint32_t p = 0; p + 1 < m; p++) plist[pos + p] = make_uint4(k, beg + p * q, q | (p == 0 ? PF_FIRST : 0u), pb + p);
        }
        pos = s_start[rem] + atomicAdd(&s_cur[rem], 1u);
        plist[pos] = make_uint4(k, beg + (m - 1) * q, rem, pb + m - 1);
    }
    __syncthreads();
    const uint32_t nco = min(s_nco, COOP_MAX);
    for (uint32_t i = 0; i < nco; i++) {  // uniform
        const uint4 co = s_co[i];
        const uint32_t pb = pbase[co.x];
        for (uint32_t p = threadIdx.x; p < co.z; p += blockDim.x)  // (listed buckets are long ones: runs of psplit)
            plist[co.w + p] = make_uint4(co.x, co.y + p * (psplit & ~PSPLIT_EQUAL), (psplit & ~PSPLIT_EQUAL) | (p == 0 ? PF_FIRST : 0u), pb + p);
    }
}

// one thread per piece, a loop without bucket switches.  INTO: a whole bucket, or the first piece of a split one, starts
// from the value the bucket holds (earlier chunks of a streamed host call / point ranges of a device-resident instance).
// (The piece histogram and the bin cursors are zeroed at the head of the NEXT chain -- clear_piece_bins -- not here.)
// M256 (round 5): `bases` are the caller's arkworks words (R = 2^256 Montgomery, x || y) for entries below nsplit and `phi` (biased by -nsplit
// records, see `record` below) holds the records nsplit + i of a split plan in the same form (k_phi_records); the record is unpacked with the x 2^5 folded into the shifts and the
// digit's sign goes to S2 (xyzz_madd_m32).  Same instruction count per addition as the internal-domain form, no conversion pass before it.
// THREE wavefronts per SIMD, pinned (amdgpu_waves_per_eu): the M256 form of the INTO kernel fits 128 VGPRs and the compiler then takes FOUR, which this
// instruction stream does not like (round 4: +5.6 % cycles with LDS-staged gathers at four; round 5: the streamed host call 2.56 -> 2.42 ms at 2^20 with the
// INTO kernel back at three, profiles/r5_host_path_waves_ab.txt; the device call is indifferent: 1.368 vs 1.375 ms at 2^20, 4.88 vs 4.74 at 2^22).
#define MSM_ACC_WAVES __attribute__((amdgpu_waves_per_eu(3, 3)))
template <bool INTO, bool CHUNK, bool M256>
__global__ void __launch_bounds__(256) MSM_ACC_WAVES k_accumulate_pieces(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ phi, uint32_t nsplit,
                                                           const uint32_t* __restrict__ sorted,
                                                           const uint4* __restrict__ plist, const uint32_t* __restrict__ npieces_ptr,
                                                           uint32_t* __restrict__ buckets, uint32_t* __restrict__ partials,
                                                           unsigned long long* __restrict__ clk) {
    // Clock probe (msm_get_clock_stats): the first workgroup of every launch brackets its own pieces with the shader-cycle counter
    // (s_memtime: counts at whatever frequency the device sustains) and the constant-rate counter (s_memrealtime); the ratio of the two
    // deltas is the shader clock the kernel really ran at, and the cycle delta says whether two boxes execute the same instruction
    // stream in the same number of cycles.  Both values live in scalar registers; the cost is four atomics per launch.
    const bool probe = blockIdx.x == 0;
    long long clk_c0 = 0, clk_w0 = 0;
    if (probe) clk_c0 = clock64(), clk_w0 = wall_clock64();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *npieces_ptr) return;
    const uint4 pc = plist[t];
    const uint32_t k = pc.x, j0 = pc.y, len = pc.z & PF_LEN_MASK, j1 = j0 + len;
    const bool whole = (pc.z & PF_WHOLE) != 0;
    xyzz acc = xyzz_identity();
    if (INTO && (pc.z & (PF_WHOLE | PF_FIRST))) acc = load_xyzz(buckets + (size_t)k * XW);
    // Software pipeline: the 64-byte record of entry j+1 and the index of entry j+2 are in flight while entry j is folded; only ONE raw
    // record is kept, unpacked to 29-bit limbs before the next gather is issued.  Every load is UNCONDITIONAL (the last entry of a piece
    // fetches its own record once more, indices are clamped) and issued at the top of the iteration: gfx950 counts loads and stores with
    // ONE in-order counter (vmcnt), and a conditional prefetch used to be closed by the compiler with an s_waitcnt a few instructions
    // after the gather was issued (round 3, profiles/NOTES_r3.md section 1).  The only wait sits at the top of the next iteration, one
    // whole mixed addition after everything was issued.
    // (M256: `phi` arrives BIASED by -nsplit records -- entry idx >= nsplit is record idx of it -- so the two sources differ by the base pointer only)
    auto record = [&](uint32_t e) {
        const uint32_t idx = e & ~SIGN_BIT;
        const uint32_t* b0 = M256 && idx >= nsplit ? phi : bases;
        return reinterpret_cast<const uint4*>(b0 + (size_t)idx * 16);
    };
    uint32_t e_cur = sorted[j0];
    uint32_t e_nxt = sorted[min(j0 + 1, j1 - 1)];
    uint4 g[4];
    {
        const uint4* bp = record(e_cur);
#pragma unroll
        for (int i = 0; i < 4; i++) g[i] = bp[i];
    }
    for (uint32_t j = j0; j < j1; j++) {
        uint32_t wx[8] = {g[0].x, g[0].y, g[0].z, g[0].w, g[1].x, g[1].y, g[1].z, g[1].w};
        uint32_t wy[8] = {g[2].x, g[2].y, g[2].z, g[2].w, g[3].x, g[3].y, g[3].z, g[3].w};
        affine q;
        q.x = M256 ? fp_unpack_shl5(wx) : fp_unpack(wx);
        q.y = M256 ? fp_unpack_shl5(wy) : fp_unpack(wy);
        if (!M256 && (e_cur & SIGN_BIT)) q.y = fp_neg_raw<2>(q.y);  // raw: only ever a multiplier in xyzz_madd
        {
            const uint4* bp = record(e_nxt);
#pragma unroll
            for (int i = 0; i < 4; i++) g[i] = bp[i];
        }
        const uint32_t e_nn = sorted[min(j + 2, j1 - 1)];
        if (M256) xyzz_madd_m32(acc, q.x, q.y, (e_cur & SIGN_BIT) != 0);
        else xyzz_madd(acc, q);
        e_cur = e_nxt;
        e_nxt = e_nn;
    }
    store_xyzz(whole ? buckets + (size_t)k * XW : partials + (size_t)pc.w * XW, acc);
    if (probe && threadIdx.x == 0) {  // [0] shader cycles, [1] constant-rate ticks, [2] samples, [3] mixed additions of the sampled thread
        atomicAdd(clk + 0, (unsigned long long)(clock64() - clk_c0));
        atomicAdd(clk + 1, (unsigned long long)(wall_clock64() - clk_w0));
        atomicAdd(clk + 2, 1ull);
        atomicAdd(clk + 3, (unsigned long long)len);
    }
}

// Split buckets (longer than pmax entries: skewed scalars, tiny top windows): partial sums partials[pbase[k] .. + m).  ONE launch, three kinds of workgroups: the leading LONG_BLOCKS workgroups take the long list -- (bucket, segment) items of buckets with
// LONG_SPAN or more pieces: a wavefront per bucket up to WAVE_ITEM_MAX pieces, LDS trees of eight-lane additions per LONG_SEG-piece segment beyond (the last-arriving workgroup of a bucket folds the
// segment sums) --, the next MID_BLOCKS the buckets of 3..LONG_SPAN-1 pieces (and the two-piece ones while they are few) with eight lanes per bucket, the trailing MID2_BLOCKS the two-piece
// buckets one lane each when there are many.  On uniform scalars only the two-piece list is filled (~5 % of the buckets).
__device__ __forceinline__ void fold_partials(uint32_t* e, const uint32_t* partials, uint32_t base, uint32_t first, uint32_t stride,
                                              uint32_t count) {
    // (round 6, measured and not kept: folding down to 64 records with one-lane additions before the tree when thousands of items are in flight --
    // half the instructions on paper -- took k_combine_pieces 0.435 instead of 0.396 ms at 256 distinct scalars: profiles/r6_adversarial_timing_mid_round.txt)
    constexpr uint32_t CAP = WIDE_TREE_MAX;
    __syncthreads();  // e is reused
    if (count <= CAP) {
        for (uint32_t i = threadIdx.x >> 2; i < count; i += blockDim.x >> 2) {
            const uint32_t co = threadIdx.x & 3u;
            store_coord(e + (size_t)i * XW, co, load_coord(partials + (size_t)(base + first + i * stride) * XW, co));
        }
    } else if (threadIdx.x < CAP) {
        xyzz acc = xyzz_identity();
#pragma unroll 1
        for (uint32_t i = threadIdx.x; i < count; i += CAP) acc = xyzz_add(acc, load_xyzz(partials + (size_t)(base + first + i * stride) * XW));
        store_xyzz(e + (size_t)threadIdx.x * XW, acc);
    }
    lds_tree_wide(e, count < CAP ? count : CAP);
}
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3))) k_combine_pieces(const uint32_t* __restrict__ offsets, uint32_t* __restrict__ partials,
                                                        uint32_t* __restrict__ buckets, uint32_t pmax, uint32_t psplit, const uint32_t* __restrict__ pbase,
                                                        const uint32_t* __restrict__ mid_count, const uint32_t* __restrict__ mid2_count,
                                                        const uint32_t* __restrict__ mid_list, uint32_t mid3_off,
                                                        const uint32_t* __restrict__ long_count, const uint32_t* __restrict__ long_list,
                                                        uint32_t* __restrict__ long_done, uint32_t mid_lane_min, const uint32_t* __restrict__ entries) {
    __shared__ uint32_t e[WIDE_TREE_MAX * XW];
    decode_piece_lengths(pmax, psplit, entries[0], entries[FLAG_NONEMPTY - FLAG_PAIRS]);  // (entries = flags + FLAG_PAIRS: what the sort chain of this accumulation counted)
    if (blockIdx.x >= LONG_BLOCKS) {
        // mid list: EIGHT lanes per listed bucket (round 6; one lane folding its 2..7 pieces with complete additions took this launch 19 us on uniform
        // scalars -- a lone wavefront's xyzz_add is 6.6 us warm and its ~40 KB of code arrive cold -- and up to 0.45 ms on skewed ones).  Two pieces: one
        // eight-lane addition straight from the partial sums into the bucket; more: the running sum lives in the group's LDS record (in-order LDS traffic of one wavefront).
        // The buckets of exactly TWO pieces are listed apart (mid_list[0 .. n2), one addition each) from those of 3 .. LONG_SPAN-1 pieces (mid_list[mid3_off ..), a
        // dependent chain each).  Many two-piece buckets -- thousands of distinct scalars repeated: the reference's fixture shape at T = 128 lists 80 000 of them
        // and 23 000 longer ones -- are the THROUGHPUT regime: the groups of eight would take several rounds of resident workgroups, each a dependent eight-lane
        // addition that keeps the multiplier half busy; the trailing MID2_BLOCKS workgroups fold them one LANE per bucket in one round of complete additions
        // (one list for both and the lanes walking 1 .. 6 additions: a wavefront runs its longest chain, 0.125 ms at T = 128 where the eight-lane groups took 0.175).
        const uint32_t n3 = *mid_count, n2 = *mid2_count, g = threadIdx.x / WIDE_LANES, ng = blockDim.x / WIDE_LANES;
        const bool lanes2 = n2 > mid_lane_min;
        if (blockIdx.x >= LONG_BLOCKS + MID_BLOCKS) {
            if (!lanes2) return;
            for (uint32_t i = (blockIdx.x - LONG_BLOCKS - MID_BLOCKS) * blockDim.x + threadIdx.x; i < n2; i += MID2_BLOCKS * blockDim.x) {
                const uint32_t k = mid_list[i];
                const uint32_t* p0 = partials + (size_t)pbase[k] * XW;
                store_xyzz(buckets + (size_t)k * XW, xyzz_add(load_xyzz(p0), load_xyzz(p0 + XW)));
            }
            return;
        }
        const uint32_t nmid = n3 + (lanes2 ? 0u : n2);
        uint32_t* acc = e + (size_t)g * XW;
        for (uint32_t i = (blockIdx.x - LONG_BLOCKS) * ng + g; i < nmid; i += MID_BLOCKS * ng) {  // (whole groups share i; the chains first)
            const uint32_t k = i < n3 ? mid_list[mid3_off + i] : mid_list[i - n3];
            uint32_t q;
            const uint32_t m = piece_split(offsets[k + 1] - offsets[k], pmax, psplit, &q), base = pbase[k];
            const uint32_t* p0 = partials + (size_t)base * XW;
            uint32_t* out = buckets + (size_t)k * XW;
            const uint32_t* a = p0;
#pragma unroll 1
            for (uint32_t p = 1; p < m; p++) {  // ONE call site (an eight-lane addition carries the ~40 KB scalar fallback for the special pairs)
                wide_add_records(a, p0 + (size_t)p * XW, p + 1 == m ? out : acc);
                a = acc;
            }
        }
        return;
    }
    __shared__ uint32_t s_last;
    const uint32_t nlong = *long_count;
    // Pass A (round 6): items of at most WAVE_ITEM_MAX pieces -- hundreds of distinct scalars make THOUSANDS of them (256 distinct at 2^20: 4096
    // buckets of 128 pieces) -- are folded by ONE WAVEFRONT each, eight items in flight per workgroup and no workgroup barrier: 32 lanes fold the
    // pieces down to <= 32 records with one-lane additions, the wavefront's 32-record LDS region is folded by eight-lane additions (LDS traffic of
    // one wavefront is in order).  A workgroup per item spent most of its time in barriers between levels that keep one or two wavefronts busy.
    {
        const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63u, g = lane / WIDE_LANES;
        uint32_t* we = e + (size_t)wid * WAVE_ITEM_RECORDS * XW;
        for (uint32_t item = blockIdx.x * (blockDim.x >> 6) + wid; item < nlong; item += LONG_BLOCKS * (blockDim.x >> 6)) {
            const uint32_t k = long_list[2 * (size_t)item];
            uint32_t q;
            const uint32_t cnt = piece_split(offsets[k + 1] - offsets[k], pmax, psplit, &q);
            if (cnt > WAVE_ITEM_MAX) continue;  // (wavefront-uniform; pass B takes it)
            const uint32_t base = pbase[k];
            if (lane < WAVE_ITEM_RECORDS && lane < cnt) {
                xyzz acc = load_xyzz(partials + (size_t)(base + lane) * XW);
#pragma unroll 1
                for (uint32_t i = lane + WAVE_ITEM_RECORDS; i < cnt; i += WAVE_ITEM_RECORDS) acc = xyzz_add(acc, load_xyzz(partials + (size_t)(base + i) * XW));
                store_xyzz(we + (size_t)lane * XW, acc);
            }
            uint32_t m = cnt < WAVE_ITEM_RECORDS ? cnt : WAVE_ITEM_RECORDS;
#pragma unroll 1
            while (m > 1) {  // wavefront-uniform
                const uint32_t h = (m + 1) >> 1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = g; i < m - h; i += 64 / WIDE_LANES) wide_add_records(we + (size_t)i * XW, we + (size_t)(i + h) * XW, we + (size_t)i * XW);
                m = h;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 4) store_coord(buckets + (size_t)k * XW, lane, load_coord(we, lane));
            __builtin_amdgcn_wave_barrier();  // (the region is reused by the wavefront's next item)
        }
    }
    // Pass B: the segments of longer buckets, a workgroup per (bucket, segment)
    for (uint32_t item = blockIdx.x; item < nlong; item += LONG_BLOCKS) {
        const uint32_t k = long_list[2 * (size_t)item], seg = long_list[2 * (size_t)item + 1];
        uint32_t q;
        const uint32_t cnt = piece_split(offsets[k + 1] - offsets[k], pmax, psplit, &q), nseg = (cnt + LONG_SEG - 1) / LONG_SEG, base = pbase[k];
        if (cnt <= WAVE_ITEM_MAX) continue;  // (uniform: done in pass A)
        const uint32_t first = seg * LONG_SEG, count = min(LONG_SEG, cnt - first);
        fold_partials(e, partials, base, first, 1, count);
        if (nseg == 1) {
            if (threadIdx.x < 4) store_coord(buckets + (size_t)k * XW, threadIdx.x, load_coord(e, threadIdx.x));
            continue;
        }
        // park the segment sum in the segment's first slot (only this workgroup ever read it), then count in
        uint32_t* park = partials + (size_t)(base + first) * XW;
        if (threadIdx.x < 4) store_coord(park, threadIdx.x, load_coord(e, threadIdx.x));
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&long_done[item - seg], 1u) == nseg - 1 ? 1u : 0u;
        __syncthreads();
        if (!s_last) continue;  // uniform
        __threadfence();        // see the other segments' sums
        fold_partials(e, partials, base, 0, LONG_SEG, nseg);
        if (threadIdx.x < 4) store_coord(buckets + (size_t)k * XW, threadIdx.x, load_coord(e, threadIdx.x));
        if (threadIdx.x == 0) long_done[item - seg] = 0;  // ready for the next call
    }
}

// ---------------------------------------------------------------------------------------------
// K4+K5: bucket reduction  S_w = sum_b (b+1) * B[w][b].
//
// The reference (pbpr.metal:33-148) gives each thread a run of buckets, forms running sums and then a
// double-and-add by the run's offset -- ~60 dependent group operations per thread.  One dependent XYZZ add
// costs a lone gfx950 wavefront ~10 us, so dependency DEPTH, not work, is what this stage pays for.  Here the
// weights are pushed to the host instead, and the device only forms PLAIN sums, which are trees:
//   split b = hi * n_lo + lo.  With row sums R_hi = sum_lo B[hi][lo] and column sums C_lo = sum_hi B[hi][lo]
//       S_w = n_lo * sum_hi hi * R_hi  +  sum_lo (lo + 1) * C_lo
//   and with bit sums  Q_u = sum over {lo : bit u of lo set} C_lo          (u <  kb_lo)
//                      Q_u = sum over {hi : bit u-kb_lo of hi set} R_hi    (kb_lo <= u < kb),  Q_all = sum_lo C_lo
//       S_w = Q_all + sum_u 2^u * Q_u
// k_pair_level forms R and C by dense pairwise levels (2 adds per bucket), k_reduce_bits the kb+1 bit sums per window
// (wavefront __shfl_down trees), and the host finishes with one Horner chain per window (host_g1.hpp).
__device__ __forceinline__ fp shfl_down_fp(const fp& a, int d, int width) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = __shfl_down(a.v[i], d, width);
    return r;
}
__device__ __forceinline__ xyzz shfl_down_xyzz(const xyzz& v, int d, int width) {
    return xyzz{shfl_down_fp(v.x, d, width), shfl_down_fp(v.y, d, width), shfl_down_fp(v.zz, d, width),
                shfl_down_fp(v.zzz, d, width)};
}
// One pairwise level of both families in one launch:  out[o] = in[i0] + in[i0 + B],  i0 = 2*(o/B)*B + o%B.
//   rows family (halve the lo dimension): B = 1      -> out[o] = in[2o] + in[2o+1]
//   cols family (halve the hi dimension): B = n_lo   -> out[w][a][b] = in[w][2a][b] + in[w][2a+1][b]
// Every lane does exactly one XYZZ add, so each level is dense; kb_hi launches take the 2^kb buckets of every
// window down to R[w][hi] and C[w][lo].
struct pair_job {
    const uint32_t* in;
    uint32_t* out;
    uint32_t n_out;
    uint32_t B;
};
__global__ void __launch_bounds__(256) k_pair_level(pair_job ja, pair_job jb) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    pair_job j = ja;
    if (t >= ja.n_out) {
        t -= ja.n_out;
        j = jb;
        if (t >= jb.n_out) return;
    }
    size_t i0 = (size_t)2 * (t / j.B) * j.B + (t % j.B);
    xyzz r = xyzz_add(load_xyzz(j.in + i0 * XW), load_xyzz(j.in + (i0 + j.B) * XW));
    store_xyzz(j.out + (size_t)t * XW, r);
}

// k_pair_level for the SMALL levels (fewer additions than the chip has lanes / 8): eight lanes per addition.
__global__ void __launch_bounds__(256) k_pair_level_wide(pair_job ja, pair_job jb) {
    uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / WIDE_LANES;
    pair_job j = ja;
    if (t >= ja.n_out) {
        t -= ja.n_out;
        j = jb;
        if (t >= jb.n_out) return;
    }
    size_t i0 = (size_t)2 * (t / j.B) * j.B + (t % j.B);
    wide_add_records(j.in + i0 * XW, j.in + (i0 + j.B) * XW, j.out + (size_t)t * XW);
}

// ALL the remaining pairwise levels of both families in ONE launch (round 6): a dependent launch of k_pair_level_wide costs 5.2 us of which 2.5 are the
// eight-lane addition, the same level inside a workgroup's LDS 2.7 (profiles/r6_wide_level_breakdown.txt) -- the five launches behind k_pair_level8 at
// 2^20 (12.1 + 7.2 + 3 x 5.2 us) become one.  Rows: output o = sum of the np_r CONSECUTIVE partial sums in[o * np_r ..); columns: output (w, b) = sum
// over a < np_c of in[(w * np_c + a) * n_lo + b].  A workgroup owns opw = WIDE_TREE_MAX / np outputs: their WIDE_TREE_MAX partial sums are staged
// partial-major (record j * opw + oo = partial j of output oo), so that every level is e[i] += e[i + m/2] and the tree simply stops at opw records.
// np_r, np_c: powers of two in [2, WIDE_TREE_MAX] (a family with one partial per output is already done: zero workgroups for it).
__global__ void __launch_bounds__(512) k_pair_tail(const uint32_t* __restrict__ rin, uint32_t* __restrict__ rout, uint32_t n_rout, uint32_t np_r,
                                                   const uint32_t* __restrict__ cin, uint32_t* __restrict__ cout, uint32_t n_cout, uint32_t np_c,
                                                   uint32_t n_lo, uint32_t rblocks) {
    __shared__ uint32_t e[WIDE_TREE_MAX * XW];
    const bool rows = blockIdx.x < rblocks;
    const uint32_t np = rows ? np_r : np_c, opw = WIDE_TREE_MAX / np, n_out = rows ? n_rout : n_cout;
    const uint32_t o0 = (rows ? blockIdx.x : blockIdx.x - rblocks) * opw, valid = min(opw, n_out - o0);
    const uint32_t* in = rows ? rin : cin;
    // staging: thread t takes coordinate t & 3 of record t >> 2 (+ 128 per trip)
    for (uint32_t rec = threadIdx.x >> 2; rec < WIDE_TREE_MAX; rec += blockDim.x >> 2) {
        const uint32_t j = rec / opw, oo = rec - j * opw, co = threadIdx.x & 3u;
        if (oo >= valid) continue;
        const uint32_t o = o0 + oo;
        const size_t src = rows ? (size_t)o * np + j : ((size_t)(o / n_lo) * np + j) * n_lo + (o % n_lo);
        store_coord(e + (size_t)rec * XW, co, load_coord(in + src * XW, co));
    }
    lds_tree_wide_until(e, WIDE_TREE_MAX, opw, valid);
    uint32_t* out = rows ? rout : cout;
    for (uint32_t rec = threadIdx.x >> 2; rec < valid; rec += blockDim.x >> 2) {
        const uint32_t co = threadIdx.x & 3u;
        store_coord(out + (size_t)(o0 + rec) * XW, co, load_coord(e + (size_t)rec * XW, co));
    }
}

// THREE pairwise levels of both families in one launch: out_r[o] = in[8o] + ... + in[8o+7] (rows: the lo dimension shrinks by 8),
// out_c[w][a][b] = in[w][8a][b] + ... + in[w][8a+7][b] (cols: the hi dimension shrinks by 8).  One thread per output, seven dependent
// additions; every bucket is read twice (once per family) and the two intermediate levels are never written: three launches and
// ~2.5x the bucket array of traffic less than k_pair_level x 3 (measured 2^20: 119 -> 103 us, 2^17: 74 -> 60 us).
// G lanes per output (round 3): the seven additions of an output are a dependent CHAIN, and with 2 * tb / 8 threads -- 65 536 at 8 x 32768
// buckets: one wavefront per SIMD -- the kernel is priced by that chain (7 x 6.4 us), not by its work.  G adjacent lanes fold 8 / G buckets
// each and join with log2 G shuffle levels: chains of 4 (G = 2) or 3 (G = 4) additions, the same 7 additions per output.
template <int G>
__global__ void __launch_bounds__(256) k_pair_level8(const uint32_t* __restrict__ in, uint32_t* __restrict__ out_r, uint32_t* __restrict__ out_c,
                                                     uint32_t n_out, uint32_t n_lo) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t t = gt / G;
    const uint32_t sub = gt % G;
    size_t i0, stride;
    uint32_t* out;
    if (t < n_out) {  // rows
        i0 = (size_t)8 * t;
        stride = 1;
        out = out_r + (size_t)t * XW;
    } else {
        t -= n_out;
        if (t >= n_out) return;  // (the G lanes of an output leave together)
        i0 = (size_t)8 * (t / n_lo) * n_lo + (t % n_lo);
        stride = n_lo;
        out = out_c + (size_t)t * XW;
    }
    constexpr uint32_t PER = 8 / G, SERIAL = PER - 1, LEVELS = G == 1 ? 0 : G == 2 ? 1 : 2;
    const size_t first = i0 + (size_t)sub * PER * stride;
    xyzz acc = load_xyzz(in + first * XW);
    // ONE call site for the serial folds and the shuffle levels (an inlined complete addition is ~40 KB of code).
    // The NEXT record is in flight while the current one is added (round 6): the chain is a lone wavefront per SIMD, so every record's load latency
    // used to sit in series with the seven additions.
    xyzz nxt = load_xyzz(in + (first + stride) * XW);
#pragma unroll 1
    for (uint32_t step = 0; step < SERIAL + LEVELS; step++) {
        xyzz other;
        if (step < SERIAL) {
            other = nxt;
            if (step + 1 < SERIAL) nxt = load_xyzz(in + (first + (size_t)(step + 2) * stride) * XW);
        } else {
            const uint32_t d = (G / 2) >> (step - SERIAL);
            other = shfl_down_xyzz(acc, (int)d, G);
            if (sub >= d) other = xyzz_identity();  // spectator lanes: adding their own value would take the doubling branch
        }
        acc = xyzz_add(acc, other);
    }
    if (sub == 0) store_xyzz(out, acc);
}

// k_reduce_bits with wide additions: one 512-thread workgroup per (window, bit); the selected elements are staged in LDS
// and folded by a pairwise tree (log2(nsel) levels of 4 multiplications each).  Needs nsel <= WIDE_TREE_MAX.
// PARTS (hooks build's probe, tools/wide_level_probe.py): 7 = the kernel; bit 0 clear: nothing staged (the tree runs on whatever LDS holds); bit 1 clear: no tree;
// bit 2 clear: no XYZZ -> Jacobian -> R = 2^256 conversion (the raw record's first 24 words go out).  The product instantiates 7 only.
template <int PARTS = 7>
__global__ void __launch_bounds__(512) k_reduce_bits_wide(const uint32_t* __restrict__ R, const uint32_t* __restrict__ C,
                                                          uint32_t* __restrict__ q, uint32_t n_hi, uint32_t n_lo, uint32_t kb_lo,
                                                          uint32_t kb, uint32_t* __restrict__ flags, uint32_t* __restrict__ flags_out,
                                                          uint32_t seq) {
    if (blockIdx.x == 0 && threadIdx.x < 8) {  // q, flags_out: pinned HOST memory, (word, seq) pairs (store_words8_tagged)
        reinterpret_cast<uint2*>(flags_out)[threadIdx.x] = make_uint2(flags[threadIdx.x], seq);
        flags[threadIdx.x] = 0;  // this kernel ends the MSM: the next one starts from clean error / count words
    }
    __shared__ uint32_t e[WIDE_TREE_MAX * XW];
    uint32_t w = blockIdx.x / (kb + 1), u = blockIdx.x % (kb + 1);
    const uint32_t* src;
    uint32_t cnt, bit;
    if (u < kb_lo) {
        src = C + (size_t)w * n_lo * XW;
        cnt = n_lo;
        bit = u;
    } else if (u < kb) {
        src = R + (size_t)w * n_hi * XW;
        cnt = n_hi;
        bit = u - kb_lo;
    } else {
        src = C + (size_t)w * n_lo * XW;
        cnt = n_lo;
        bit = 0xFFFFFFFFu;
    }
    const uint32_t nsel = bit == 0xFFFFFFFFu ? cnt : cnt >> 1;
    // thread t stages coordinate t&3 of the (t>>2)-th SELECTED element (index = m with a 1 inserted at position `bit`)
    if (PARTS & 1) {
        for (uint32_t m = threadIdx.x >> 2; m < nsel; m += blockDim.x >> 2) {
            const uint32_t j = bit == 0xFFFFFFFFu ? m : (((m >> bit) << (bit + 1)) | (1u << bit) | (m & ((1u << bit) - 1u)));
            const uint32_t co = threadIdx.x & 3u;
            store_coord(e + (size_t)m * XW, co, load_coord(src + (size_t)j * XW, co));
        }
        if (nsel == 0 && threadIdx.x == 0) store_xyzz(e, xyzz_identity());
    }
    if (PARTS & 2) lds_tree_wide(e, nsel);
    else __syncthreads();
    if (PARTS & 4) {
        // The host does not wait for the kernel to RETIRE (end-of-kernel cache maintenance, completion signal, the runtime's wake-up): it polls the
        // (word, seq) pairs themselves (finish_sync)
        if (threadIdx.x < 64) store_jacobian_mont256_lanes(q + (size_t)blockIdx.x * 48, e, seq);
    } else if (threadIdx.x == 0) {
        for (int i = 0; i < 24; i++) reinterpret_cast<uint2*>(q)[(size_t)blockIdx.x * 24 + i] = make_uint2(e[i], seq);
    }
}

// one wavefront per (window, bit): R[w][0..n_hi), C[w][0..n_lo);  q[w][u] Jacobian
__global__ void __launch_bounds__(64) k_reduce_bits(const uint32_t* __restrict__ R, const uint32_t* __restrict__ C,
                                                    uint32_t* __restrict__ q, uint32_t n_hi, uint32_t n_lo, uint32_t kb_lo,
                                                    uint32_t kb, uint32_t* __restrict__ flags, uint32_t* __restrict__ flags_out,
                                                    uint32_t seq) {
    if (blockIdx.x == 0 && threadIdx.x < 8) {  // q, flags_out: pinned HOST memory, (word, seq) pairs
        reinterpret_cast<uint2*>(flags_out)[threadIdx.x] = make_uint2(flags[threadIdx.x], seq);
        flags[threadIdx.x] = 0;  // this kernel ends the MSM: the next one starts from clean error / count words
    }
    uint32_t w = blockIdx.x / (kb + 1), u = blockIdx.x % (kb + 1);
    const uint32_t* src;
    uint32_t cnt, bit;
    if (u < kb_lo) {
        src = C + (size_t)w * n_lo * XW;
        cnt = n_lo;
        bit = u;
    } else if (u < kb) {
        src = R + (size_t)w * n_hi * XW;
        cnt = n_hi;
        bit = u - kb_lo;
    } else {
        src = C + (size_t)w * n_lo * XW;
        cnt = n_lo;
        bit = 0xFFFFFFFFu;
    }
    // lane m folds the m-th, (m+64)-th, ... SELECTED element (index = m with a 1 inserted at position `bit`), so the
    // serial part is cnt/128 adds instead of cnt/64 masked ones: dependency depth 1 + 6 for 256 row sums
    // ONE xyzz_add call site for the strided folds and the six shuffle levels: an inlined complete add is ~40 KB of
    // code, and a lone wavefront running six unrolled copies streams every one of them through the 64 KB instruction
    // cache cold (measured: 83 us -> see profiles/NOTES_r1.md)
    xyzz acc = xyzz_identity();
    const uint32_t nsel = bit == 0xFFFFFFFFu ? cnt : cnt >> 1;
    const uint32_t nser = (nsel + 63) / 64;
#pragma unroll 1
    for (uint32_t step = 0; step < nser + 6; step++) {
        xyzz other;
        if (step < nser) {
            const uint32_t m = threadIdx.x + 64 * step;
            if (m < nsel) {
                uint32_t j = bit == 0xFFFFFFFFu ? m : (((m >> bit) << (bit + 1)) | (1u << bit) | (m & ((1u << bit) - 1u)));
                other = load_xyzz(src + (size_t)j * XW);
            } else {
                other = xyzz_identity();
            }
        } else {
            // lanes >= d are spectators: __shfl_down hands them their OWN value, and acc + acc would drag the whole
            // wavefront through the doubling branch on top of the addition (measured: 82 -> see NOTES); give them the identity
            const uint32_t d = 32u >> (step - nser);
            if (d >= nsel) continue;  // lanes >= nsel hold the identity: nothing to fold at this distance (uniform)
            other = shfl_down_xyzz(acc, d, 64);
            if (threadIdx.x >= d) other = xyzz_identity();
        }
        acc = xyzz_add(acc, other);
    }
    if (threadIdx.x == 0) {
        store_jacobian_mont256_tagged(q + (size_t)blockIdx.x * 48, xyzz_to_jacobian(acc), seq);  // (see k_reduce_bits_wide: the host polls the pairs)
    }
}

}  // namespace msmk
