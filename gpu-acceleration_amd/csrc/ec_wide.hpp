// ec_wide.hpp -- one XYZZ + XYZZ addition spread over EIGHT adjacent lanes (device only).
//
// Why: the tail of the bucket reduction (small pairwise levels, per-bit trees, long-bucket trees) is a chain of DEPENDENT
// additions with far fewer additions than lanes.  A lone gfx950 wavefront needs 6.4 us for one complete xyzz_add
// (tools/microbench3: 14 multiplications of 0.51 us back to back; a second independent chain in the same wavefront does
// not overlap, the multiplier is already busy), so the chain is priced by multiplications in SERIES.  add-2008-s has
// only four multiplication STAGES:
//     stage 1   U1 = X1*ZZ2   U2 = X2*ZZ1   S1 = Y1*ZZZ2   S2 = Y2*ZZZ1   Za = ZZ1*ZZ2   Zb = ZZZ1*ZZZ2
//     stage 2   PP = P^2  (P = U2-U1)        RR = R^2  (R = S2-S1)
//     stage 3   PPP = P*PP    Q = U1*PP      ZZ3 = Za*PP
//     stage 4   T1 = R*(Q-X3)  (X3 = RR-PPP-2Q)     T2 = S1*PPP     ZZZ3 = Zb*PPP          Y3 = T1 - T2
// Eight lanes (roles 0..7, two spare) run one multiplication per stage each and hand operands round with wave
// shuffles: 4 multiplications in series instead of 14.  Same formulas, same value bounds as xyzz_add (ec_bn254.hpp).
//
// The special cases of the COMPLETE law (an identity operand, P == 0: doubling or inverse points) are only DETECTED
// here: the function then returns true for the whole group and the caller runs the scalar xyzz_add for that pair.
#pragma once
#include "ec_bn254.hpp"

namespace bn254 {

constexpr int WIDE_LANES = 8;

// which coordinate (0 = X, 1 = Y, 2 = ZZ, 3 = ZZZ) of which operand (0 = a, 1 = b) a role multiplies in stage 1
__device__ __forceinline__ uint32_t wide_opa_rec(uint32_t role) { return (0x0Au >> role) & 1u; }          // 0,1,0,1,0,0,0,0
__device__ __forceinline__ uint32_t wide_opa_coord(uint32_t role) { return (0x0E50u >> (2 * role)) & 3u; }  // 0,0,1,1,2,3,0,0
__device__ __forceinline__ uint32_t wide_opb_rec(uint32_t role) { return (0x35u >> role) & 1u; }          // 1,0,1,0,1,1,0,0
__device__ __forceinline__ uint32_t wide_opb_coord(uint32_t role) { return (0x0EFAu >> (2 * role)) & 3u; }  // 2,2,3,3,2,3,0,0

__device__ __forceinline__ fp wide_shfl(const fp& a, int src_lane) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = __shfl(a.v[i], src_lane, 64);
    return r;
}
__device__ __forceinline__ fp wide_select(bool c, const fp& a, const fp& b) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// Probe marks (hooks build, k_probe_wide_level: VERDICT r5 item 1a -- where the time of one eight-lane addition goes).  PROBE = false, the
// only form the product instantiates, compiles to nothing.  A mark waits for every outstanding memory / LDS operation, so that the time
// between two marks belongs to the instructions between them.
constexpr int WIDE_MARKS = 16;
template <bool PROBE>
__device__ __forceinline__ void wide_mark(long long* ts, int i) {
    if constexpr (PROBE) {
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_sched_barrier(0);
        ts[i] = clock64();
        __builtin_amdgcn_sched_barrier(0);
    }
}

// opa, opb: this lane's stage-1 operands (coordinate wide_op?_coord(role) of operand wide_op?_rec(role); roles 6, 7: anything
// normalised).  operand_is_identity: role 4 passes (ZZ1 == 0 || ZZ2 == 0), other roles false.
// Returns true (group-uniform) if the pair needs the scalar complete addition; otherwise ONE output coordinate per role:
//   role 1: out = X3 (< 7p)      role 2: out = Y3 (< 5p)      role 4: out = ZZ3 (< 2p)      role 5: out = ZZZ3 (< 2p)
// (wide_out_coord(role) names the coordinate; round 6: one coordinate per lane -- the result goes out in one predicated store instead of three
// divergent ones -- one subtraction per stage instead of both candidates, X3 with one carry ripple: 6861 -> see profiles/r6_wide_level_breakdown.txt)
__device__ __forceinline__ bool wide_has_out(uint32_t role) { return (0x36u >> role) & 1u; }                // roles 1, 2, 4, 5
__device__ __forceinline__ uint32_t wide_out_coord(uint32_t role) { return (0x0E10u >> (2 * role)) & 3u; }  // -,0,1,-,2,3,-,-
template <bool PROBE = false>
__device__ __forceinline__ bool xyzz_add_wide(const fp& opa, const fp& opb, bool operand_is_identity, fp& out, long long* ts = nullptr) {
    const int lane = (int)(threadIdx.x & 63u);
    const int base = lane & ~(WIDE_LANES - 1);
    const uint32_t role = (uint32_t)lane & (WIDE_LANES - 1);

    // stage 1: r0 U1, r1 U2, r2 S1, r3 S2, r4 Za, r5 Zb                       (operands < 7p, < 2p -> products < 1.09p)
    const fp s1 = fp_mul(opa, opb);
    wide_mark<PROBE>(ts, 2);

    // stage 2: r0 PP = (U2-U1)^2, r1 RR = (S2-S1)^2
    const fp g1 = wide_shfl(s1, base + (role == 0 ? 1 : role == 1 ? 2 : (int)role));  // r0 <- U2, r1 <- S1
    const fp g2 = wide_shfl(s1, base + (role == 1 ? 3 : (int)role));                   // r1 <- S2
    // r0: P = U2 - U1 = g1 - s1;  r1: R = S2 - S1 = g2 - g1 (< 4.09p); others: unused.  Minuend and subtrahend are SELECTED, then subtracted once
    const fp pr = fp_sub<3>(wide_select(role == 0, g1, g2), wide_select(role == 0, s1, g1));
    wide_mark<PROBE>(ts, 3);
    const fp s2 = fp_sqr(pr);                                           // r0 PP, r1 RR  (< 1.1p)
    wide_mark<PROBE>(ts, 4);

    // P == 0 (same x: doubling or inverse points) or an identity operand -> scalar path for this pair
    const bool mine = (role == 0 && fp_is_zero_lt2p(s2)) || (role == 4 && operand_is_identity);
    const unsigned long long votes = __ballot(mine);
    if ((votes >> base) & 0xFFull) return true;
    wide_mark<PROBE>(ts, 5);

    // stage 3: r0 PPP = P*PP, r2 Q = U1*PP, r4 ZZ3 = Za*PP
    const fp pp = wide_shfl(s2, base);  // PP to everyone
    const fp u1 = wide_shfl(s1, base);  // U1 to everyone (r2 needs it)
    const fp a3 = role == 0 ? pr : role == 2 ? u1 : s1;  // r4: Za = its own stage-1 product
    wide_mark<PROBE>(ts, 6);
    const fp s3 = fp_mul(a3, pp);                        // r0 PPP (< 1.03p), r2 Q (< 1.01p), r4 ZZ3
    wide_mark<PROBE>(ts, 7);

    // stage 4: r1 T1 = R*(Q - X3), r2 T2 = S1*PPP, r5 ZZZ3 = Zb*PPP
    const fp ppp = wide_shfl(s3, base);      // PPP
    const fp q = wide_shfl(s3, base + 2);    // Q
    const fp x3 = fp_sub_b_2c(s2, ppp, q);   // r1: RR + 5p - PPP - 2Q, one carry ripple (PPP + 2Q < 3.05p < 4p); X3 < 6.1p
    const fp a4 = role == 1 ? pr : s1;                       // r1: R;  r2: S1;  r5: Zb
    const fp b4 = role == 1 ? fp_sub<8>(q, x3) : ppp;        // r1: Q - X3 (< 9.01p)
    wide_mark<PROBE>(ts, 8);
    const fp s4 = fp_mul(a4, b4);                            // r1 T1 (< 1.22p), r2 T2 (< 1.01p), r5 ZZZ3
    wide_mark<PROBE>(ts, 9);

    // Y3 = T1 - T2 on role 2 (it holds T2 and fetches T1), so that roles 1, 2, 4, 5 hand out one coordinate each
    const fp t1 = wide_shfl(s4, base + 1);
    const fp y3 = fp_sub<3>(t1, s4);                         // r2: (< 4.3p); others: unused
    out = role == 1 ? x3 : role == 2 ? y3 : role == 4 ? s3 : s4;
    wide_mark<PROBE>(ts, 10);
    return false;
}

}  // namespace bn254
