// msm_planner.hpp -- window / digit-form / GLV planner of the MSM engine.  Plain C++ (no HIP): also compiled into the HOST-only
// AddressSanitizer build (tools/host_asan_check.cpp).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "../../include/msm_hip.h"

namespace msmplan {

constexpr uint32_t GLV_SPLIT_BITS = 127;  // == glv::SPLIT_BITS (glv_bn254.hpp); asserted in msm_hip.hip
// The halves of a split scalar are below 7 * 2^123 (= 2^125.81; glv_bn254.hpp HALF_BOUND_*, proved by tools/gen_glv_constants.py), so the TOP
// window of a split plan holds few digit values: 14336 of 32768 at c = 16 (eight windows, 14 bits left), 56 of 512 at c = 10.  Its buckets
// would be 2-10x as full as every other window's -- at 2^20 points 16384 buckets of ~128 entries among 229376 of ~64, which the
// accumulation had to cut into pieces and fold again.  The decomposition therefore SPREADS the top window: bucket index =
// (digit magnitude - 1) | (low bits of the point index) << top_digit_bits.  The reduction is unchanged (it sums by index bits); the host
// leaves the bit sums of the spread bits out (host_finish).  top_digit_bits = the smallest t with max magnitude <= 2^t.
inline uint32_t glv_top_digit_bits(uint32_t c, uint32_t W, bool is_signed, uint32_t kb) {
    const uint32_t low = c * (W - 1);
    if (low >= 127) return kb;
    const unsigned __int128 bound = ((unsigned __int128)7 << 123) - 1;  // largest half
    const uint64_t maxmag = (uint64_t)(bound >> low) + (is_signed ? 1u : 0u);  // (the signed form's carry)
    uint32_t t = 0;
    while (((uint64_t)1 << t) < maxmag) t++;
    return t < kb ? t : kb;
}

// ---- planner: replaces the N -> window_size / scale_factor tables (metal_msm.rs:661-691).  The cuZK cost model
// (utils/window_size_optimizer.rs:38-51: per window N mixed adds plus ~2 full adds per bucket) gives the shape, but two
// measured effects decide the table below (tools/sweep_c.py, profiles/NOTES_r1.md "window sweep"):
//  * r < 2^254, so the top window only holds 254 mod c bits.  For c = 7, 9, 11, 12, 14 that is 1-2 bits: every point
//    lands in one of <= 3 buckets of that window, which serialises the LDS sort cursors and makes those buckets
//    thousands of chunks long (c = 12 at N = 2^19: 6.7 ms against 1.2 ms).  Only c in {8, 10, 13, 15, 16} (6, 4, 7, 14, 14
//    top bits) are used.
//  * below ~2^17 points the per-window fixed costs (dependent reduction levels, launches) outweigh the bucket count:
//    fewer, wider windows win earlier than the arithmetic model says.
// c is capped where the two-level LDS sort still covers a window (<= 2^17 buckets; unsigned digits stay at 15).
inline uint32_t plan_window_bits(size_t n, bool is_signed) {
    // re-measured after the reduction-tree and host-latency work (tools/sweep_c.py): 2^13: c = 8 0.335 ms (13: 0.455);
    // 2^14: c = 10 0.397 (13: 0.431); 2^15: 10 0.412 (13: 0.440); 2^16: 13 0.489; 2^17: 15 0.587; 2^18: 15 0.769 (16: 0.790);
    // 2^19: 16 1.125 (15: 1.197).  c = 10 leaves the top window 4 bits (9 buckets of n/16 points): fine for the long-bucket path.
    // Round 2 (tools/c17_sweep.py, interleaved): c = 17 (15 windows of 65536 buckets; the two-level sort covers them) against 16:
    // 2^20 +4.5 % (1.735 vs 1.661 ms: the doubled bucket reduction outweighs the 6 % fewer additions), 2^21 -2.3 %, 2^22 -4.1 %,
    // 2^23 -6.0 %, 2^24 -5.9 % (22.39 vs 23.79 ms)  => 17 above 2^20 points.
    // Round 3, between the powers of two (tools/odd_size_ab.py, profiles/r3_glv_between_pow2.txt): 2^20 + 1 points c = 16 1.686 ms against
    // 17 1.779 (-5 %), 1 500 000 2.199 / 2.274 (-3 %), 2^21 2.852 / 2.878 (inside the noise)  => 17 from 2^21 points on.
    uint32_t c = n <= ((size_t)1 << 13) ? 8u : n <= ((size_t)1 << 15) ? 10u : n <= ((size_t)1 << 16) ? 13u : n <= ((size_t)1 << 18) ? 15u
               : n < ((size_t)1 << 21) ? 16u : 17u;
    if (!is_signed && c > 15u) c = 15u;
    return c;
}
// GLV (glv_bn254.hpp): 2n virtual points with half-length scalars (|k_j| < 7 * 2^123, windows over 127 bits) -- the same additions in half the windows: half the buckets to
// reduce, half the host's Horner chain.  Interleaved A/B against the unsplit pipeline (tools/ab_glv.py): 2^10 -12.8 %, 2^14 -11.9 %,
// 2^16 -12.3 %, 2^17 -12.2 %, 2^18 -9.6 %, 2^19 +1.1 %, 2^20 -0.4 %, 2^21 +9.2 %, 2^22 +6.8 % (twice the base records to gather
// from, k_accumulate unchanged, and the fixed costs it halves no longer matter).  With the chunk length following the bucket
// occupancy: 2^18 -10.4 %, 2^19 -3.4 %, 2^20 -0.3 %, 2^21 +3.6 %, 2^22 +9.9 %  => on by default up to 2^19 points (round 2).
// Round 3, after k_accumulate stopped waiting for its own gathers (the 2n base records of the split no longer cost latency): 2^19 -2.3 %,
// 2^20 -2.2 / -2.5 % on one box and -5.0 % on another, 2^21 -1.3 % (inside the noise), 2^22 +5.2 % (profiles/r3_glv_ab.txt)
// => on by default up to 2^20 points.
constexpr size_t GLV_MAX_POINTS = (size_t)1 << 20;
inline uint32_t plan_window_bits_glv(size_t n, bool is_signed) {
    // measured (tools/sweep_c.py, split on): 2^10 c = 9/10 0.247/0.249 ms; 2^12 10/11 0.286/0.281; 2^13 10 0.308 (16: 0.364);
    // 2^14 10 0.330 (16: 0.440); 2^15 12 0.365 (16: 0.435); 2^16 16 0.420 (13: 0.439); 2^17 16 0.504 (13: 0.541); 2^18 16 0.671
    // (15: 1.06); 2^20 16 1.681.  127 = 7*16 + 15: eight windows, the top one 15 bits wide -- no degenerate window.
    // Re-measured after k_accumulate lost 11 % of its instructions (round 2, tools/sweep_c.py, ramped-up clock, ms): 2^13 c = 10 0.337,
    // 13 0.327, 16 0.331; 2^14 10 0.362-0.369, 13 0.363, 16 0.331; 2^15 12 0.406-0.409, 15 0.390, 16 0.336-0.346; 2^16 16 0.378-0.382
    // (15: 0.446); 2^17 16 0.451 (13: 0.552); 2^18 16 0.613 (13: 0.760); 2^19 16 0.937 (13: 1.201)  => 16 from 2^14 points on.
    // Round 4, with the top window spread over all of its buckets (glv_top_digit_bits: a 9-bit top window of c = 13 no longer piles 2n points
    // into 448 buckets; tools/sweep_window_widths.sh, profiles/r4_window_width_sweep.txt, ms): 2^11 c = 10 0.246-0.261, 13 0.227; 2^12 10 0.272-0.276, 12 0.240,
    // 13 0.232; 2^13 10 0.310, 12 0.270, 13 0.247, 16 0.293; 2^14 13 0.278, 14 0.289, 16 0.291-0.297; 2^15 13 0.323, 16 0.307-0.319; 2^16 13 0.371, 15 0.367,
    // 16 0.336; 2^17 15 0.463, 16 0.415; 2^19 13 1.059, 16 0.846; 2^8..2^10: 0.20-0.23 whatever the width  => 13 from 2^11 to below 2^15 points.
    uint32_t c = n < ((size_t)1 << 11) ? 10u : n < ((size_t)1 << 15) ? 13u : 16u;
    if (!is_signed && c > 15u) c = 15u;
    return c;
}
// Work items of k_accumulate_pieces (msm_kernels.hpp): a bucket of at most pmax entries is ONE piece; a longer one is cut into runs of pmax
// entries and a rest (up to 8 x pmax) or into runs of psplit entries (beyond).
// pmax = mean occupancy + max(8, 2 sqrt(mean)) -- two standard deviations of a Poisson bucket above the mean: the few per cent of the buckets
// beyond it are cut, and their RESTS (1 .. ~20 entries) are what the launch ends on.  The pieces run longest first, 1.33 rounds of resident
// workgroups at 2^20 points; with whole buckets only, the last workgroups still walk 40-50 entries each while the rest of the chip idles.
// Measured (profiles/r4_top_window_spread.txt; k_accumulate_pieces Mcycles / ms per MSM): 2^20 (mean 64) cap 128: 2.455 / 1.480, 112: 2.461,
// 96: 2.373 / 1.444, 88: 2.340, 80: 2.313 / 1.423, 72: 2.311 / 1.422 (k_combine_pieces +16 us);  2^19 (32) cap 64: 1.281, 48: 1.193, 40:
// 1.175, 36: 1.171;  2^18 (16) cap 32: 0.628 / 0.580, 28: 0.610, 24: 0.596 / 0.563, 20: 0.589 / 0.565;  2^17 (8) 16: 0.311 / 0.415, 12: 0.301 /
// 0.420, 10: 0.294 / 0.421 (the folding costs what the balance gains);  2^16 (4) 16: 0.213 / 0.372, 12: 0.180 / 0.343, 8: 0.152 / 0.345;  unsplit
// 2^21 (32) 64: 4.305 / 2.859, 43: 4.147 / 2.826;  2^22, 2^24 (64, 256): no difference (6+ rounds).
// Round 4 before the top window of split plans was spread (glv_top_digit_bits): pmax = 2 x the mean, and the rests of the top window's
// twice-as-full buckets were that tail.  At most the 1024 histogram bins of the piece sort.  psplit makes an instance of long buckets only
// yield ~2^19 pieces (2.7 rounds of resident workgroups), at least 8 entries per piece.  max_pieces / max_partials bound what ANY
// bucket-size distribution over `pairs` sorted entries and `total_buckets` buckets can produce (workspace sizes).
constexpr uint32_t PIECE_BINS_MAX = 1024;
constexpr size_t SPLIT_PIECES_TARGET = (size_t)1 << 18;  // pieces an instance of very long buckets is cut into, about (make_piece_plan: psplit)
struct piece_plan {
    uint32_t pmax = 0, psplit = 0;
    size_t max_pieces = 0, max_partials = 0;
};
// The kernels shorten psplit for instances that sort FEW entries (msmk::effective_psplit: the largest power of two <= entries >> SPLIT_ENTRIES_SHIFT, at least 8, at
// most psplit): such an instance is cut into fewer than 2^(shift + 1) runs (entries < (want + 1) * 2^shift <= 2 p * 2^shift), which max_pieces / max_partials allow for.
// Measured (tools/split_length_ab.py MSM_HIP_SPLIT_SHIFT=..., 2^20 points, ms): witness-like mix (40 % zeros, 30 % ones) shift 0 (never): 0.759, 15: 0.761, 16: 0.657, 17: 0.662,
// 18: 0.668; all scalars < 2^32: 0.667 / 0.672 / 0.663 / 0.622 / 0.622; every full-size row (uniform, all-equal, 3-distinct, 256-distinct, fixture shapes) unchanged.
constexpr uint32_t SPLIT_ENTRIES_SHIFT = 17;
#ifdef __HIPCC__
#define MSMPLAN_HD __host__ __device__
#else
#define MSMPLAN_HD
#endif
// The longest WHOLE bucket as the device sees the instance (round 6).  pmax = mean + 2 sigma is planned for buckets that are Poisson around n_v / nb; an instance
// whose entries sit in a SMALL FRACTION of the buckets -- the reference's own fixture shape, one (base, scalar) sequence repeated T times (metal_msm.rs:706-730): at
// 2^17 points and T = 128 every window has ~2000 buckets of 128 entries and 30 000 empty ones -- is cut into 8 x the pieces it needs (pmax = 16: eight pieces per
// bucket, 15 853 buckets for k_combine_pieces' long list: 0.38 ms of a 0.80 ms call against 0.36 on uniform scalars).  The two-level sort counts the NON-EMPTY
// buckets (k_fine_sort); when there are at most SPARSE_BUCKETS_MAX of them -- whole buckets are then at most one wavefront per SIMD, which a lone wavefront keeps
// nearly busy -- the kernels that cut buckets raise pmax to the occupancy of those, mean + max(8, 2 sqrt(mean)), but not beyond entries >> 16 (2^16 pieces: a
// wavefront for every SIMD), and cut what is still longer into EQUAL runs (sizes that are multiples of T then give pieces of one length).  Never below the plan;
// uniform scalars fill every bucket and change nothing; raising pmax only lowers the piece and partial-sum counts the workspace was sized for.
// Measured with the count in place but WITHOUT the SPARSE_BUCKETS_MAX bound (tools/fixture_shape_sizes.py, ms, before -> after, sort penalty of the first form
// taken out): T = 128: 2^17 0.795 -> 0.49, 2^18 0.696 -> 0.60, 2^19 1.009 -> 0.85 (57 000 non-empty buckets), 2^20 1.495 -> 1.57 (103 000: two unequal
// wavefronts per SIMD); T = 32 (103 000 - 165 000 non-empty from 2^18 on): 2^19 0.830 -> 0.90 => only below 65 536.
constexpr uint32_t SPARSE_BUCKETS_MAX = 65536;
MSMPLAN_HD inline uint32_t effective_pmax(uint32_t pmax, uint32_t entries, uint32_t nonempty) {
    if (nonempty == 0 || nonempty > SPARSE_BUCKETS_MAX) return pmax;  // (0: not counted -- fallback sorts)
    uint32_t mean = entries / nonempty;
    if (mean > 65536u) mean = 65536u;  // (the result is capped at PIECE_BINS_MAX anyway)
    // floor(2 sqrt(mean)), exactly, without the plan's counting loop: every thread of three kernels evaluates this (64 trips cost k_place_count 18 us)
    uint32_t two_sigma = (uint32_t)sqrtf((float)(4u * mean));
    while (two_sigma * two_sigma > 4u * mean) two_sigma--;
    while ((two_sigma + 1u) * (two_sigma + 1u) <= 4u * mean) two_sigma++;
    uint32_t cand = mean + (two_sigma > 8 ? two_sigma : 8u);
    // (... except where that would still cut a typical bucket into EIGHT or more pieces -- k_combine_pieces' long list: 2^16 points at T = 128, 8049 buckets of
    // 128 entries, 0.21 ms -- : there four pieces per bucket, at most 32 entries each (0.125 ms at a lone wavefront's rate bounds what the fewer pieces can
    // cost): 0.537 -> 0.43 ms.  Not in general: 32-entry pieces at 2^16 points and T = 32 are half the wavefronts for the same work, 0.391 -> 0.438)
    uint32_t cap = entries >> 16;
    if (mean >= 8u * cap) {
        const uint32_t quarter = mean / 4u < 32u ? mean / 4u : 32u;
        if (cap < quarter) cap = quarter;
    }
    if (cand > cap) cand = cap;
    if (cand > PIECE_BINS_MAX) cand = PIECE_BINS_MAX;
    return cand > pmax ? cand : pmax;
}
// the rule itself (the kernels call it through msmk::effective_psplit; tools/host_asan_check.cpp holds make_piece_plan's bounds against it)
MSMPLAN_HD inline uint32_t effective_psplit(uint32_t psplit, uint32_t shift, uint32_t entries) {
    const uint32_t want = entries >> shift;  // (shift 0: never shortened)
    uint32_t p = 8;
    while (p * 2 <= want && p * 2 <= psplit) p *= 2;
    return p < psplit ? p : psplit;
}
inline piece_plan make_piece_plan(size_t pairs, size_t mean_occupancy, size_t total_buckets, uint32_t forced_len = 0, const piece_plan* first = nullptr,
                                  size_t split_target = SPLIT_PIECES_TARGET, uint32_t entries_shift = SPLIT_ENTRIES_SHIFT) {
    piece_plan p;
    size_t two_sigma = 0;
    while ((two_sigma + 1) * (two_sigma + 1) <= 4 * mean_occupancy) two_sigma++;  // floor(2 sqrt(mean))
    p.pmax = (uint32_t)std::min<size_t>(PIECE_BINS_MAX, mean_occupancy + std::max<size_t>(8, two_sigma));
    // FEW buckets (tiny instances on 10-bit windows: a few thousand buckets of 16-64 entries; the shared array of a split window table: 2^15
    // buckets for eight windows' entries) would be as few pieces -- a wavefront on a fraction of the SIMDs, each walking its piece alone; there
    // the pieces shrink until ~2^17 of them exist (at least 8 entries each).  2^17 points with the table: k_accumulate_pieces 0.257 -> 0.149 ms, profiles/r4_table_small.txt
    if (total_buckets < ((size_t)1 << 17)) p.pmax = (uint32_t)std::min<size_t>(p.pmax, std::max<size_t>(8, pairs >> 17));
    // Runs of a bucket too long for the pmax rule (a handful of distinct scalars): a POWER OF TWO of entries, about pairs / 2^18 of them (64 at 2^20 points: 2^18
    // pieces, 1.33 rounds of the 196 608 resident lanes, what uniform scalars give).  Round 4's pairs >> 19 (32 entries, 2^19 pieces) left k_combine_pieces twice the
    // partial sums for the same accumulation time; lengths that are not a power of two cost the accumulation 5-8 % whatever the number of rounds they make
    // (MSM_HIP_SPLIT_TARGET of the hooks build, whole calls at 2^20 on one box, ms: all-equal 32: 1.42, 36: 1.53, 40: 1.50, 44: 1.45, 48: 1.54, 64: 1.41;
    // 3-distinct 32: 1.55, 44: 1.47-1.52, 64: 1.49; 256-distinct 32: 1.44, 44: 1.47, 64: 1.425 -- profiles/NOTES_r6.md section 14).
    {
        const size_t want = std::min<size_t>(p.pmax, std::max<size_t>(8, (pairs + split_target - 1) / split_target));
        p.psplit = 8;
        while ((size_t)p.psplit * 2 <= want) p.psplit *= 2;
    }
    if (first) p.pmax = first->pmax, p.psplit = first->psplit;  // a later chunk of an instance: the lengths of its first, largest chunk
    if (forced_len) p.pmax = p.psplit = std::min<uint32_t>(forced_len, PIECE_BINS_MAX);
    // every non-empty bucket is a piece, a split bucket of sz > pmax entries adds at most sz / psplit more (runs of pmax: ceil(sz / pmax) <=
    // sz / psplit + 1 as well); partial sums: split buckets only (at most sz / psplit + 1 each, and fewer than pairs / pmax buckets can be split)
    // (runs of psplit -- or of the shorter length the kernels pick for an instance of few entries: fewer than 2^(entries_shift + 1) of those)
    const size_t runs = forced_len || entries_shift == 0 ? pairs / p.psplit : std::max<size_t>(pairs / p.psplit, (size_t)1 << std::min<uint32_t>(entries_shift + 1, 40));
    p.max_pieces = std::min(pairs, total_buckets + runs) + 1;
    p.max_partials = std::min(pairs, runs + pairs / ((size_t)p.pmax + 1) + 2) + 1;
    return p;
}

// MSM_HIP_GLV_MAX_LOG2 (A/B knob of the HOOKS build).  Read where a CONTEXT is created (msm_ctx.knobs) and by the context-free
// msm_plan(); never inside make_plan, which a call evaluates several times and which must give an upload and the resident calls
// that follow it the same answer.
inline size_t glv_max_from_env() {
#ifdef MSM_HIP_TEST_HOOKS  // (the product library reads no planner knob from the environment)
    if (const char* e = std::getenv("MSM_HIP_GLV_MAX_LOG2")) return (size_t)1 << std::min(23, std::max(0, std::atoi(e)));
#endif
    return GLV_MAX_POINTS;
}
inline int32_t make_plan(size_t n, uint32_t window_bits, uint32_t flags, msm_plan_t* out, size_t glv_max = GLV_MAX_POINTS) {
    if (flags & ~(MSM_FLAG_UNSIGNED_DIGITS | MSM_FLAG_NO_GLV | MSM_FLAG_WINDOW_TABLE | MSM_FLAG_DETERMINISTIC)) return MSM_ERR_BAD_ARG;
    bool is_signed = !(flags & MSM_FLAG_UNSIGNED_DIGITS);
    bool use_glv = !(flags & MSM_FLAG_NO_GLV) && n <= glv_max;
    uint32_t c = window_bits ? window_bits : (use_glv ? plan_window_bits_glv(n, is_signed) : plan_window_bits(n, is_signed));
    if (c < 2 || c > 20) return MSM_ERR_BAD_ARG;
    if ((is_signed ? c - 1 : c) > 17) use_glv = false;  // windows wider than the LDS sort covers (forced c >= 19) run unsplit
    const uint32_t bits = use_glv ? GLV_SPLIT_BITS : 254u;
    out->window_bits = c;
    out->signed_digits = is_signed;
    out->glv = use_glv ? 1u : 0u;
    out->scalar_bits = bits;
    out->virtual_points = use_glv ? 2 * (uint64_t)n : (uint64_t)n;
    // signed: one spare window position so the top digit never overflows (r < 2^254, |k_j| < 2^126): W = floor(bits/c) + 1
    out->num_windows = is_signed ? (bits / c + 1) : ((bits + c - 1) / c);
    out->num_buckets = is_signed ? (1u << (c - 1)) : (1u << c);
    size_t nv = (size_t)out->virtual_points;
    size_t pairs = (size_t)out->num_windows * nv;
    size_t tb = (size_t)out->num_windows * out->num_buckets;
    {   // records, scalars + infinity bytes, digits / sorted / staging, offsets + bucket sums, reduction levels, the piece list and the partial sums
        const piece_plan pp = make_piece_plan(pairs, nv / out->num_buckets, tb);
        out->workspace_bytes = nv * 64 + n * (32 + 1) + pairs * 12 + tb * (8 + 144) + tb * 144 * 3 / 2 + pp.max_partials * 144 + pp.max_pieces * 16 + tb * 4;
    }
    out->table_factor = 1;
    out->bucket_arrays = out->num_windows;
    out->table_bytes = 0;
    uint32_t kb = 0;
    while ((1u << kb) < out->num_buckets) kb++;
    // unsplit: scalars below 2^254 (checked on the device) leave the top window 254 - c*(W-1) bits: magnitudes up to 2^that (signed carry included)
    const uint32_t top_plain = std::min(kb, 254u - c * (out->num_windows - 1));
    out->top_digit_bits = use_glv ? glv_top_digit_bits(c, out->num_windows, is_signed, kb) : top_plain;
    out->reserved = 0;
    return MSM_OK;
}

// ---- the window table of a RESIDENT base set (MSM_FLAG_WINDOW_TABLE): T_j[i] = 2^(c*j) P_i, j < f.
// The f windows of a group add into ONE bucket array; with f = W (the default here) every window shares the same array, so the
// 2^(c-1) buckets are reduced once per MSM instead of once per window -- the reference's cost model (window_size_optimizer.rs:38-51:
// (n + 2^(s+1)) * ceil(lambda/s)) loses its per-window bucket term and the optimum moves to wider windows: c = 20 at 2^20 points
// (13 windows instead of 16: 19 % fewer additions).  Widths whose top window is only 1-2 bits wide (254 mod c: c = 18, 21) are skipped
// as in plan_window_bits; above 20 bits the LDS sort no longer covers a window.  Memory: f * (n or 2n) * 64 bytes.
struct table_knobs {
    uint32_t c = 0, f = 0;                // MSM_HIP_TABLE_C / MSM_HIP_TABLE_F: force the width / the factor (0 = planner)
    size_t max_bytes = (size_t)64 << 30;  // MSM_HIP_TABLE_MAX_GB: no table beyond this
    size_t glv_max = 0;                   // MSM_HIP_TABLE_GLV_MAX_LOG2: table plans split up to this many points (0 = TABLE_GLV_MAX_POINTS)
};
// A table is made while ONE sort covers the shared array and the table pays: up to 2^21 points (13 x 2^21 = 27 M entries; the regions of
// the fine sort hold entries / 1024 and are sorted by their owner workgroup up to four LDS staging areas of 16384).  At 2^22 points
// (54 M entries, 3.5 GB of records) the gathers cost what the 13 % fewer additions save: -2.3 % on one box, +0.8 % on another; beyond, the
// pipeline would have to cut the windows into ranges that accumulate INTO the array (built, bit-exact, 17 % slower than no table).
constexpr size_t TABLE_MAX_ENTRIES = (size_t)32 << 20;
// Measured (tools/table_sweep.py, profiles/r3_f4_shared_buckets.txt; resident batch, ms per MSM, table vs plain):
//   2^14 GLV c = 16 0.172 vs 0.214;  2^16 0.233 vs 0.280;  2^17 0.321 vs 0.374;  2^18 0.494 vs 0.547 (unsplit c = 20: 0.545)
//   2^19 unsplit c = 20 0.855 vs 0.934 (GLV c = 16: 0.899);  2^20 c = 20 1.395 vs 1.521 (c = 17: 1.487);  2^21 2.83 vs 2.95;  2^22 5.35 vs 5.48
// => up to 2^18 points the GLV split with c = 16 (eight windows, ONE array of 2^15 buckets), from 2^19 unsplit c = 20.
constexpr size_t TABLE_GLV_MAX_POINTS = (size_t)1 << 18;
inline uint32_t plan_table_bits(size_t nv, bool glv) {
    if (glv) return nv <= ((size_t)1 << 13) ? 10u : 16u;
    return nv <= ((size_t)1 << 17) ? 16u : nv <= ((size_t)1 << 18) ? 17u : 20u;
}
// plan of a resident call on all n points of a set uploaded under `flags` (window_bits: the context's forced width or 0).  Without
// MSM_FLAG_WINDOW_TABLE, with plain digits, or when the table would not fit: the ordinary plan, table_factor 1.
inline int32_t make_table_plan(size_t n, uint32_t window_bits, uint32_t flags, msm_plan_t* out, size_t glv_max, const table_knobs& tk) {
    int32_t rc = make_plan(n, window_bits, flags, out, glv_max);
    if (rc != MSM_OK || !(flags & MSM_FLAG_WINDOW_TABLE) || (flags & MSM_FLAG_UNSIGNED_DIGITS)) return rc;
    const size_t tglv = tk.glv_max ? tk.glv_max : TABLE_GLV_MAX_POINTS;
    const uint32_t tflags = n > tglv ? (flags | MSM_FLAG_NO_GLV) : flags;  // (the split pays up to 2^18 points here)
    if (tglv > glv_max) glv_max = tglv;
    const bool glv0 = !(tflags & MSM_FLAG_NO_GLV) && n <= glv_max;
    uint32_t c = tk.c ? tk.c : window_bits ? window_bits : plan_table_bits(glv0 ? 2 * n : n, glv0);
    msm_plan_t t;
    if (make_plan(n, c, tflags, &t, glv_max) != MSM_OK) return rc;  // (a forced width out of range: keep the ordinary plan)
    uint32_t f = tk.f ? tk.f : t.num_windows;
    if (f < 2 || t.num_windows % f) return rc;
    const uint64_t bytes = (uint64_t)f * t.virtual_points * 64;
    if (bytes > tk.max_bytes || (uint64_t)f * t.virtual_points > TABLE_MAX_ENTRIES) return rc;
    t.table_factor = f;
    uint32_t kb = 0;
    while ((1u << kb) < t.num_buckets) kb++;
    t.top_digit_bits = kb;  // (a table's short top window is spread by its table level instead: table_top_shift)
    t.bucket_arrays = t.num_windows / f;
    t.table_bytes = bytes;
    const size_t tb = (size_t)t.bucket_arrays * t.num_buckets, pairs = (size_t)t.num_windows * (size_t)t.virtual_points;
    {
        const piece_plan pp = make_piece_plan(pairs, (size_t)f * (size_t)t.virtual_points / t.num_buckets, tb);
        t.workspace_bytes = bytes + n * (32 + 1) + pairs * 12 + tb * (8 + 144) + tb * 144 * 3 / 2 + pp.max_partials * 144 + pp.max_pieces * 16 + tb * 4;
    }
    *out = t;
    return MSM_OK;
}

// Window table with ONE shared bucket array (table factor == number of windows): the top window only holds
// scalar_bits - c*(W-1) bits (14 of 20 at c = 20), so its digits would all land in the lowest buckets -- a few regions of the sort and
// a few hundred chunks of k_accumulate would carry a whole window.  Its table level is built as 2^(c*(W-1) - s) P and the digit d enters
// as d * 2^s instead (same group element, every 2^s-th bucket): s = (c - 1) - top bits, so that d * 2^s <= 2^(c-1) still holds.
inline uint32_t table_top_shift(const msm_plan_t& pl, uint32_t tf) {
    if (tf <= 1 || tf != pl.num_windows || !pl.signed_digits) return 0;
    const uint32_t top_bits = pl.scalar_bits - pl.window_bits * (pl.num_windows - 1);  // max top digit 2^top_bits (carry included)
    return top_bits < pl.window_bits - 1 ? pl.window_bits - 1 - top_bits : 0u;
}

}  // namespace msmplan
