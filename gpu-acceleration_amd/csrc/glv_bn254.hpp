// glv_bn254.hpp -- GLV split of a BN254 scalar for the G1 endomorphism phi(x, y) = (beta*x, y) = lambda*(x, y).
//
//     k = k1 + lambda*k2 (mod r),   |k1|, |k2| < 2^126
// so an N-point MSM with 254-bit scalars becomes a 2N-point MSM (points P_i and phi(P_i)) with 126-bit scalars: the same number
// of bucket additions, HALF the windows -- half the buckets to reduce and half the bit positions of the host's Horner chain.
// The reference has no endomorphism; arkworks' own G1 config carries the same constants.
// Constants and formulas: tools/gen_glv_constants.py (derived and verified with Python integers; tests/golden/glv_constants.json):
//     c1 = (k*G1 + 2^287) >> 288,  d2 = (k*G2 + 2^287) >> 288        (G_i = round(2^288*|b|/r): rounded quotients, within 1/2 + 2^-35 of k*|b|/r;
//                                                                      round 4: 2^256 before -- quotient error 1/8, halves up to 2^126.08)
//     k1 = k - c1*A - d2*A2,       k2 = d2*A - c1*B1                 (A = a1 = b2, A2 = -a2, B1 = b1)
// All in two's complement modulo 2^160; the results are then sign + magnitude.
#pragma once
#include <cstdint>
#ifndef FP_HD
#define FP_HD __host__ __device__ __forceinline__
#endif

namespace glv {

constexpr uint32_t BETA_STD[8] = {0x607cfd48u, 0xe4bd44e5u, 0xbb966e3du, 0xc28f069fu, 0xe0acccb0u, 0x5e6dd9e7u, 0xe131a029u, 0x30644e72u};
constexpr uint32_t G1[4] = {0x6eb9c714u, 0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};                             // round(2^288*|b2|/r), 98 bits
constexpr uint32_t G2[6] = {0x149d5410u, 0x00ff6565u, 0x5398fd03u, 0xa773d2d2u, 0x4ccef014u, 0x00000002u};  // round(2^288*|b1|/r), 162 bits
constexpr uint32_t A[2] = {0x94d213e3u, 0x89d32568u};                                           // a1 = b2 = 9931322734385697763
constexpr uint32_t B1[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};               // b1
constexpr uint32_t A2[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};               // -a2
constexpr int SPLIT_BITS = 127;  // bits the windows of a split plan cover
constexpr int HALF_BITS = 126;   // |k1|, |k2| <= (1/2 + 2^-35)(|a1| + |a2|) = 2^125.80 < 7 * 2^123 < 2^126 (tools/gen_glv_constants.py and tests/test_glv.py
                                 // prove it; split() checks the 126 bits) -- the top window of a split plan therefore holds few digit values, and the
                                 // decomposition spreads them over all of its buckets (msmplan::glv_top_digit_bits)

// r[0..NR) += a[0..NA) * b[0..NB)   modulo 2^(32*NR)
template <int NA, int NB, int NR>
FP_HD void mul_acc(uint32_t (&r)[NR], const uint32_t (&a)[NA], const uint32_t (&b)[NB]) {
#pragma unroll
    for (int i = 0; i < NA; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            if (i + j >= NR) break;
            uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + carry;
            r[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
#pragma unroll
        for (int k = i + NB; k < NR; k++) {  // propagate
            uint64_t t = (uint64_t)r[k] + carry;
            r[k] = (uint32_t)t;
            carry = t >> 32;
        }
    }
}
// (k * g + 2^287) >> 288, words 0..NQ-1 of the quotient
template <int NG, int NQ>
FP_HD void rounded_quotient(uint32_t (&q)[NQ], const uint32_t (&k)[8], const uint32_t (&g)[NG]) {
    uint32_t prod[9 + NQ];
#pragma unroll
    for (int i = 0; i < 9 + NQ; i++) prod[i] = 0;
    prod[8] = 0x80000000u;  // + 2^287
    mul_acc<8, NG, 9 + NQ>(prod, k, g);
#pragma unroll
    for (int i = 0; i < NQ; i++) q[i] = prod[9 + i];
}
// two's complement (mod 2^160) -> sign and 4-word magnitude; returns false if the magnitude does not fit HALF_BITS = 126 bits
FP_HD bool to_sign_magnitude(const uint32_t (&v)[5], uint32_t (&mag)[4], bool& neg) {
    neg = (v[4] >> 31) != 0;
    uint32_t t[5];
    uint64_t carry = neg ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        uint64_t x = (uint64_t)(neg ? ~v[i] : v[i]) + carry;
        t[i] = (uint32_t)x;
        carry = x >> 32;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) mag[i] = t[i];
    return t[4] == 0 && (t[3] >> 30) == 0;
}
// k: 8 little-endian words, < 2^254.  Returns false if a half exceeds 126 bits (cannot happen for k < 2^254; checked anyway).
FP_HD bool split(const uint32_t (&k)[8], uint32_t (&k1)[4], bool& neg1, uint32_t (&k2)[4], bool& neg2) {
    uint32_t c1[3], d2[5];
    rounded_quotient<4, 3>(c1, k, G1);  // < 2^65
    rounded_quotient<6, 5>(d2, k, G2);  // < 2^129
    // t1 = c1*A + d2*A2,  t2 = c1*B1        (mod 2^160)
    uint32_t t1[5] = {0, 0, 0, 0, 0}, t2[5] = {0, 0, 0, 0, 0}, u[5] = {0, 0, 0, 0, 0};
    mul_acc<3, 2, 5>(t1, c1, A);
    mul_acc<5, 4, 5>(t1, d2, A2);
    mul_acc<3, 4, 5>(t2, c1, B1);
    mul_acc<5, 2, 5>(u, d2, A);
    uint32_t v1[5], v2[5];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {  // v1 = k - t1
        uint64_t x = (uint64_t)k[i] - t1[i] - borrow;
        v1[i] = (uint32_t)x;
        borrow = (x >> 32) & 1u;
    }
    borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {  // v2 = u - t2
        uint64_t x = (uint64_t)u[i] - t2[i] - borrow;
        v2[i] = (uint32_t)x;
        borrow = (x >> 32) & 1u;
    }
    const bool ok1 = to_sign_magnitude(v1, k1, neg1), ok2 = to_sign_magnitude(v2, k2, neg2);
    return ok1 && ok2;
}

}  // namespace glv
