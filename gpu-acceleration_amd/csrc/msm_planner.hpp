// msm_planner.hpp -- window / digit-form / GLV planner of the MSM engine.  Plain C++ (no HIP): also compiled into the HOST-only
// AddressSanitizer build (tools/host_asan_check.cpp).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>

#include "../../include/msm_hip.h"

namespace msmplan {

constexpr uint32_t GLV_SPLIT_BITS = 127;  // == glv::SPLIT_BITS (glv_bn254.hpp); asserted in msm_hip.hip

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
    uint32_t c = n <= ((size_t)1 << 13) ? 8u : n <= ((size_t)1 << 15) ? 10u : n <= ((size_t)1 << 16) ? 13u : n <= ((size_t)1 << 18) ? 15u
               : n <= ((size_t)1 << 20) ? 16u : 17u;
    if (!is_signed && c > 15u) c = 15u;
    return c;
}
// GLV (glv_bn254.hpp): 2n virtual points with 127-bit scalars -- the same additions in half the windows: half the buckets to
// reduce, half the host's Horner chain.  Interleaved A/B against the unsplit pipeline (tools/ab_glv.py): 2^10 -12.8 %, 2^14 -11.9 %,
// 2^16 -12.3 %, 2^17 -12.2 %, 2^18 -9.6 %, 2^19 +1.1 %, 2^20 -0.4 %, 2^21 +9.2 %, 2^22 +6.8 % (twice the base records to gather
// from, k_accumulate unchanged, and the fixed costs it halves no longer matter).  With the chunk length following the bucket
// occupancy: 2^18 -10.4 %, 2^19 -3.4 %, 2^20 -0.3 %, 2^21 +3.6 %, 2^22 +9.9 %  => on by default up to 2^19 points.
constexpr size_t GLV_MAX_POINTS = (size_t)1 << 19;
inline uint32_t plan_window_bits_glv(size_t n, bool is_signed) {
    // measured (tools/sweep_c.py, split on): 2^10 c = 9/10 0.247/0.249 ms; 2^12 10/11 0.286/0.281; 2^13 10 0.308 (16: 0.364);
    // 2^14 10 0.330 (16: 0.440); 2^15 12 0.365 (16: 0.435); 2^16 16 0.420 (13: 0.439); 2^17 16 0.504 (13: 0.541); 2^18 16 0.671
    // (15: 1.06); 2^20 16 1.681.  127 = 7*16 + 15: eight windows, the top one 15 bits wide -- no degenerate window.
    // Re-measured after k_accumulate lost 11 % of its instructions (round 2, tools/sweep_c.py, ramped-up clock, ms): 2^13 c = 10 0.337,
    // 13 0.327, 16 0.331; 2^14 10 0.362-0.369, 13 0.363, 16 0.331; 2^15 12 0.406-0.409, 15 0.390, 16 0.336-0.346; 2^16 16 0.378-0.382
    // (15: 0.446); 2^17 16 0.451 (13: 0.552); 2^18 16 0.613 (13: 0.760); 2^19 16 0.937 (13: 1.201)  => 16 from 2^14 points on.
    uint32_t c = n < ((size_t)1 << 14) ? 10u : 16u;
    if (!is_signed && c > 15u) c = 15u;
    return c;
}
// MSM_HIP_GLV_MAX_LOG2 (A/B knob).  Read where a CONTEXT is created (msm_ctx.knobs) and by the context-free
// msm_plan(); never inside make_plan, which a call evaluates several times and which must give an upload and the resident calls
// that follow it the same answer.
inline size_t glv_max_from_env() {
    if (const char* e = std::getenv("MSM_HIP_GLV_MAX_LOG2")) return (size_t)1 << std::min(23, std::max(0, std::atoi(e)));
    return GLV_MAX_POINTS;
}
inline int32_t make_plan(size_t n, uint32_t window_bits, uint32_t flags, msm_plan_t* out, size_t glv_max = GLV_MAX_POINTS) {
    if (flags & ~(MSM_FLAG_UNSIGNED_DIGITS | MSM_FLAG_NO_GLV)) return MSM_ERR_BAD_ARG;
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
    // signed: one spare window position so the top digit never overflows (r < 2^254, |k_j| < 2^127): W = floor(bits/c) + 1
    out->num_windows = is_signed ? (bits / c + 1) : ((bits + c - 1) / c);
    out->num_buckets = is_signed ? (1u << (c - 1)) : (1u << c);
    size_t nv = (size_t)out->virtual_points;
    size_t pairs = (size_t)out->num_windows * nv;
    size_t tb = (size_t)out->num_windows * out->num_buckets;
    out->workspace_bytes = nv * 64 + n * (32 + 1) + pairs * 12 + tb * (8 + 144) + tb * 144 * 3 / 2;
    return MSM_OK;
}

}  // namespace msmplan
