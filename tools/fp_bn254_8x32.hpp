// fp_bn254_8x32.hpp -- the FIRST field implementation of this repo: 8 x 32-bit limbs, CIOS Montgomery with
// v_addc_co_u32 carry chains (R = 2^256).  Superseded in the product by fp_bn254.hpp (9 x 29-bit limbs, carry-free
// column accumulation: 1.5x faster on gfx950 because carry instructions issue at the multiplier rate there --
// profiles/NOTES_r1.md).  Kept only as the A/B baseline of tools/microbench*.hip; not included by any kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn254_8x32 {

#define FP_HD __host__ __device__ __forceinline__

struct fp {
    uint32_t v[8];
};

// p = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47  (SH/constants.metal:30-47)
#define FP_P0 0xd87cfd47u
#define FP_P1 0x3c208c16u
#define FP_P2 0x6871ca8du
#define FP_P3 0x97816a91u
#define FP_P4 0x8181585du
#define FP_P5 0xb85045b6u
#define FP_P6 0xe131a029u
#define FP_P7 0x30644e72u
// -p^-1 mod 2^32 (low 16 bits = N0 = 25481, SH/constants.metal:9)
#define FP_INV32 0xe4866389u

FP_HD uint32_t fp_p(int i) {
    constexpr uint32_t P[8] = {FP_P0, FP_P1, FP_P2, FP_P3, FP_P4, FP_P5, FP_P6, FP_P7};
    return P[i];
}
// R mod p (Montgomery 1), SH/constants.metal:175-192
FP_HD fp fp_one() {
    return fp{{0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}};
}
// R^2 mod p
FP_HD fp fp_r2() {
    return fp{{0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u}};
}
FP_HD fp fp_zero() { return fp{{0, 0, 0, 0, 0, 0, 0, 0}}; }

FP_HD bool fp_is_zero(const fp& a) {
    return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0;
}
FP_HD bool fp_eq(const fp& a, const fp& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.v[i] ^ b.v[i];
    return d == 0;
}

// r = a - p, returns the final borrow (1 => a < p).  __builtin_subc/__builtin_addc lower to
// v_sub_co_u32 / v_subb_co_u32 / v_addc_co_u32 chains on gfx950 (64-bit C arithmetic does not).
FP_HD uint32_t fp_sub_p(fp& r, const fp& a) {
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(a.v[i], fp_p(i), bw, &bw);
    return bw;
}
// canonicalise a value known to be < 2p
FP_HD void fp_reduce_once(fp& a) {
    fp t;
    uint32_t bw = fp_sub_p(t, a);
#pragma unroll
    for (int i = 0; i < 8; i++) a.v[i] = bw ? a.v[i] : t.v[i];
}

// ff_add (SH/field/ff.metal:9-20): a,b < p  ->  (a+b) mod p.   a+b < 2^255 so no 257th bit.
FP_HD fp fp_add(const fp& a, const fp& b) {
    fp r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
    fp_reduce_once(r);
    return r;
}
// ff_sub (SH/field/ff.metal:22-35)
FP_HD fp fp_sub(const fp& a, const fp& b) {
    fp r;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(a.v[i], b.v[i], bw, &bw);
    // add p back iff we borrowed
    uint32_t mask = (uint32_t)0 - bw;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(r.v[i], fp_p(i) & mask, c, &c);
    return r;
}
FP_HD fp fp_dbl(const fp& a) { return fp_add(a, a); }
FP_HD fp fp_neg(const fp& a) {
    fp r = fp_sub(fp_zero(), a);  // 0 - a  (a == 0 -> borrow 0 -> stays 0)
    return r;
}

// Montgomery product a*b*R^-1 mod p, CIOS, one row = 8 independent v_mad_u64_u32 followed by
// one 9-word v_addc_co_u32 carry chain (mont_mul_cios, SH/mont_backend/mont.metal:105-181
// restated on 32-bit limbs).
FP_HD fp fp_mul(const fp& a, const fp& b) {
    uint32_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        // t += a * b[i]
        uint64_t pr[8];
        uint32_t c = 0, t9;
#pragma unroll
        for (int j = 0; j < 8; j++) pr[j] = (uint64_t)a.v[j] * b.v[i] + t[j];
        t[0] = (uint32_t)pr[0];
#pragma unroll
        for (int j = 1; j < 8; j++) t[j] = __builtin_addc((uint32_t)pr[j], (uint32_t)(pr[j - 1] >> 32), c, &c);
        t[8] = __builtin_addc(t[8], (uint32_t)(pr[7] >> 32), c, &c);
        t9 = c;
        // t = (t + m*p) / 2^32
        uint32_t m = t[0] * FP_INV32;
#pragma unroll
        for (int j = 0; j < 8; j++) pr[j] = (uint64_t)m * fp_p(j) + t[j];
        c = 0;  // low word of pr[0] is zero by construction
#pragma unroll
        for (int j = 1; j < 8; j++) t[j - 1] = __builtin_addc((uint32_t)pr[j], (uint32_t)(pr[j - 1] >> 32), c, &c);
        t[7] = __builtin_addc(t[8], (uint32_t)(pr[7] >> 32), c, &c);
        t[8] = t9 + c;
    }
    fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    // a,b < p => result < 2p and t[8] == 0
    fp_reduce_once(r);
    return r;
}
FP_HD fp fp_sqr(const fp& a) { return fp_mul(a, a); }
FP_HD fp fp_to_mont(const fp& a) { return fp_mul(a, fp_r2()); }
// raw_reduction (utils/mont_reduction.rs:15-40)
FP_HD fp fp_from_mont(const fp& a) {
    fp one = fp_zero();
    one.v[0] = 1;
    return fp_mul(a, one);
}
// a^(p-2); a in Montgomery form, result in Montgomery form.  a == 0 -> 0.
__host__ __device__ inline fp fp_inv(const fp& a) {
    fp acc = fp_one(), base = a;
    for (int i = 0; i < 256; i++) {
        uint32_t e = fp_p(i >> 5);
        if ((i >> 5) == 0) e -= 2u;  // p0 - 2 does not borrow (p0 = ...fd47)
        if ((e >> (i & 31)) & 1u) acc = fp_mul(acc, base);
        base = fp_sqr(base);
    }
    return acc;
}

}  // namespace bn254_8x32
