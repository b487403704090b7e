// ec_bn254.hpp -- BN254 G1 (y^2 = x^3 + 3) group law for gfx950 on the lazily reduced 9 x 29-bit field.
//
// Replaces the reference's SH/curve/jacobian.metal:11-226 and curve/utils.metal:9-31.
// Differences, all deliberate:
//  * accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
//    a mixed add is 8M+2S instead of the reference's 16-multiplication add-2007-bl with Z2 = R
//    (smvp.metal:61-71), and no field inversion is ever needed in the pipeline;
//  * the group law is COMPLETE: P+P, P+(-P), infinity operands are decided projectively
//    (U2 == X1 etc.), not by limb equality (jacobian.metal:52-53) -- arkworks, the behaviour to
//    match, is complete; the reference is not (SURVEY.md section 4 "gaps");
//  * identity is ZZ == 0 (the reference encodes it as Z == 0 with X = Y = R).
// Formulas: EFD shortw/xyzz  madd-2008-s, add-2008-s, dbl-2008-s-1 / mdbl-2008-s-1 (a = 0).
//
// Value bounds (multiples of p; k = p/2^261 = 0.0059; a product of inputs < a*p, b*p is < (a*b*k + 1)*p):
//   affine operand   x < 1, y < 2 (y may be a negation 2p - y)
//   XYZZ everywhere  X < 7, Y < 5, ZZ < 2, ZZZ < 2        -- every function below re-establishes these
//   (Y is in fact < 2 since Y3 comes out of one fused multiply-add; 5 leaves room for affine inputs 2p - y)
// Each fp_sub<K>(a, b) needs b < (K-1)*p; the bound of b is written next to it.
#pragma once
#include "fp_bn254.hpp"

namespace bn254 {

struct affine {  // internal Montgomery domain; the point at infinity is carried out of band
    fp x, y;
};
struct xyzz {
    fp x, y, zz, zzz;
};
struct jacobian {  // X/Z^2, Y/Z^3; identity <=> Z == 0.  Output format of the C ABI.
    fp x, y, z;
};

FP_HD xyzz xyzz_identity() { return xyzz{fp_one(), fp_one(), fp_zero(), fp_zero()}; }
FP_HD bool xyzz_is_identity(const xyzz& p) { return fp_is_zero_exact(p.zz); }
FP_HD xyzz xyzz_from_affine(const affine& a) { return xyzz{a.x, a.y, fp_one(), fp_one()}; }
FP_HD affine affine_neg(const affine& a) { return affine{a.x, fp_neg<2>(a.y)}; }  // y < 1  ->  2p - y < 2

// 2*(x,y) for an affine point: mdbl-2008-s-1 with a = 0.   (y != 0 on BN254: no 2-torsion)
FP_HD xyzz xyzz_dbl_affine(const affine& a) {
    fp u = fp_dbl(a.y);                        // < 4
    fp v = fp_sqr(u);                          // < 1.1
    fp w = fp_mul(u, v);                       // < 1.03
    fp s = fp_mul(a.x, v);                     // < 1.01
    fp xx = fp_sqr(a.x);                       // < 1.01
    fp m = fp_add(fp_dbl(xx), xx);             // < 3.03
    fp x3 = fp_sub<4>(fp_sqr(m), fp_dbl(s));   // 2s < 2.02;  x3 < 1.06 + 4 = 5.06
    fp y3 = fp_mul_add(m, fp_sub<7>(s, x3), w, fp_neg<3>(a.y));  // m*(s - x3) + w*(3p - y): (3.03*8.01 + 1.03*3)k + 1 < 1.17
    return xyzz{x3, y3, v, w};
}
// dbl-2008-s-1, a = 0
FP_HD xyzz xyzz_dbl(const xyzz& p) {
    if (xyzz_is_identity(p)) return p;
    fp u = fp_dbl(p.y);                        // < 10
    fp v = fp_sqr(u);                          // < 1.6
    fp w = fp_mul(u, v);                       // < 1.1
    fp s = fp_mul(p.x, v);                     // < 1.07
    fp xx = fp_sqr(p.x);                       // < 1.29
    fp m = fp_add(fp_dbl(xx), xx);             // < 3.87
    fp x3 = fp_sub<4>(fp_sqr(m), fp_dbl(s));   // 2s < 2.14;  x3 < 1.09 + 4 = 5.09
    fp y3 = fp_mul_add(m, fp_sub<7>(s, x3), w, fp_neg<6>(p.y));  // m*(s - x3) + w*(6p - y): (3.87*8.07 + 1.1*6)k + 1 < 1.23
    return xyzz{x3, y3, fp_mul(v, p.zz), fp_mul(w, p.zzz)};
}

// acc += (x2,y2) with every intermediate normalised: what xyzz_madd below is checked against (tools/fp_bounds_check.cpp)
FP_HD void xyzz_madd_plain(xyzz& acc, const affine& q) {
    if (xyzz_is_identity(acc)) {
        acc = xyzz_from_affine(q);
        return;
    }
    fp u2 = fp_mul(q.x, acc.zz), s2 = fp_mul(q.y, acc.zzz);
    fp pp_ = fp_sub<8>(u2, acc.x), r = fp_sub<6>(s2, acc.y), pp = fp_sqr(pp_);
    if (fp_is_zero_lt2p(pp)) {
        if (fp_is_zero_lt2p(fp_sqr(r))) acc = xyzz_dbl_affine(q);
        else acc = xyzz_identity();
        return;
    }
    fp ppp = fp_mul(pp_, pp), qv = fp_mul(acc.x, pp);
    fp x3 = fp_sub<5>(fp_sqr(r), fp_add(ppp, fp_dbl(qv)));
    fp y3 = fp_mul_add(r, fp_sub<8>(qv, x3), fp_neg<6>(acc.y), ppp);
    acc = xyzz{x3, y3, fp_mul(acc.zz, pp), fp_mul(acc.zzz, ppp)};
}

// acc += (x2,y2)   madd-2008-s, 8M+2S, complete.  q.y may be RAW (fp_neg_raw<2>: the sign of a digit is applied without a carry
// ripple); three more ripples are saved inside: X3 takes one (fp_sub_b_2c) instead of three, and the two subtractions that only
// feed Y3's fused multiply-add stay raw.  120 of the ~2250 instructions of a mixed addition.
FP_HD void xyzz_madd(xyzz& acc, const affine& q) {
    if (xyzz_is_identity(acc)) {
        acc = xyzz{q.x, fp_normalize(q.y), fp_one(), fp_one()};
        return;
    }
    fp u2 = fp_mul(q.x, acc.zz);       // < 1.02
    fp s2 = fp_mul(q.y, acc.zzz);      // q.y raw, value < 2: < 1.03
    fp pp_ = fp_sub<8>(u2, acc.x);     // P: acc.x < 7;  P < 9.02
    fp r = fp_sub<6>(s2, acc.y);       // R: acc.y < 5;  R < 7.03
    fp pp = fp_sqr(pp_);               // < 1.49
    if (fp_is_zero_lt2p(pp)) {         // P == 0 (mod p): same x
        if (fp_is_zero_lt2p(fp_sqr(r))) acc = xyzz_dbl_affine(affine{q.x, fp_normalize(q.y)});  // same point
        else acc = xyzz_identity();                                                             // opposite points
        return;
    }
    fp ppp = fp_mul(pp_, pp);          // < 1.08
    fp qv = fp_mul(acc.x, pp);         // < 1.07
    fp x3 = fp_sub_b_2c(fp_sqr(r), ppp, qv);  // r^2 < 1.3, ppp + 2 qv < 3.22;  x3 < 6.3
    fp y3 = fp_mul_add(r, fp_sub_raw<8>(qv, x3), fp_neg_raw<6>(acc.y), ppp);  // r*(qv - x3) + (6p - y)*ppp: (7.03*9.07 + 6*1.08)k + 1 < 1.42
    acc.x = x3;
    acc.y = y3;
    acc.zz = fp_mul(acc.zz, pp);       // < 1.02
    acc.zzz = fp_mul(acc.zzz, ppp);    // < 1.02
}
// acc += +-(x2, y2) where the affine operand arrives as qx = 32 * X2, qy = 32 * Y2 (fp_unpack_shl5 of the caller's R = 2^256 Montgomery words:
// congruent to the internal-domain coordinates, normalised limbs, value < 2^261 = 169.4p for ANY 256-bit words, < 32p for canonical ones) and
// the sign of the digit is applied to S2 = Y2 * ZZZ1 instead of to y2 (a negated 32 * Y2 would need a pad of 170p).  Same instruction count as
// xyzz_madd; bounds with the 169.4p multipliers:  U2, S2 < 169.4 * 2k + 1 = 3.0;  P < 11;  R < 4 + 6 = 10;  PP < 1.72;  PPP < 1.12;  Q < 1.08;
// R^2 < 1.59, PPP + 2Q < 3.28 (< 4: fp_sub_b_2c);  X3 < 6.6 (< 7);  Y3 < (10 * 9.08 + 6 * 1.12)k + 1 < 1.58;  ZZ3, ZZZ3 < 1.03.
// No k_convert_bases pass, no second copy of the bases in HBM (round 5).
FP_HD void xyzz_madd_m32(xyzz& acc, const fp& qx, const fp& qy, bool neg) {
    if (xyzz_is_identity(acc)) {  // (every piece starts here: two multiplications by one instead of a mixed addition)
        const fp x = fp_mul(qx, fp_one()), y = fp_mul(qy, fp_one());  // < 169.4k + 1 < 2
        acc = xyzz{x, neg ? fp_neg<4>(y) : y, fp_one(), fp_one()};   // Y < 4
        return;
    }
    fp u2 = fp_mul(qx, acc.zz);        // < 3.0
    fp s2 = fp_mul(qy, acc.zzz);       // < 3.0
    fp s2s = s2;
    if (neg) s2s = fp_neg_raw<4>(s2);  // raw: an addend of the next subtraction only (a branch, not selects: nine subtractions under the lane mask)
    fp pp_ = fp_sub<8>(u2, acc.x);     // P: acc.x < 7;  P < 11
    fp r = fp_sub<6>(s2s, acc.y);      // R: acc.y < 5;  R < 10   (s2s raw: limbs < 2^30, + pad < 2^31: fp_normalize's input range)
    fp pp = fp_sqr(pp_);               // < 1.72
    if (fp_is_zero_lt2p(pp)) {         // P == 0 (mod p): same x
        if (fp_is_zero_lt2p(fp_sqr(r))) {  // same point
            const fp x = fp_reduce_lt2p(fp_mul(qx, fp_one())), y = fp_reduce_lt2p(fp_mul(qy, fp_one()));  // canonical
            acc = xyzz_dbl_affine(neg ? affine_neg(affine{x, y}) : affine{x, y});
        } else {
            acc = xyzz_identity();     // opposite points
        }
        return;
    }
    fp ppp = fp_mul(pp_, pp);          // < 1.12
    fp qv = fp_mul(acc.x, pp);         // < 1.08
    fp x3 = fp_sub_b_2c(fp_sqr(r), ppp, qv);  // r^2 < 1.59, ppp + 2 qv < 3.28;  x3 < 6.6
    fp y3 = fp_mul_add(r, fp_sub_raw<8>(qv, x3), fp_neg_raw<6>(acc.y), ppp);  // r*(qv - x3) + (6p - y)*ppp: (10*9.08 + 6*1.12)k + 1 < 1.58
    acc.x = x3;
    acc.y = y3;
    acc.zz = fp_mul(acc.zz, pp);       // < 1.03
    acc.zzz = fp_mul(acc.zzz, ppp);    // < 1.03
}
// a + b, add-2008-s, 12M+2S, complete.
FP_HD xyzz xyzz_add(const xyzz& a, const xyzz& b) {
    if (xyzz_is_identity(a)) return b;
    if (xyzz_is_identity(b)) return a;
    fp u1 = fp_mul(a.x, b.zz);         // < 1.09
    fp u2 = fp_mul(b.x, a.zz);
    fp s1 = fp_mul(a.y, b.zzz);        // < 1.06
    fp s2 = fp_mul(b.y, a.zzz);
    fp pp_ = fp_sub<3>(u2, u1);        // < 4.09
    fp r = fp_sub<3>(s2, s1);          // < 4.06
    fp pp = fp_sqr(pp_);               // < 1.1
    if (fp_is_zero_lt2p(pp)) {
        if (fp_is_zero_lt2p(fp_sqr(r))) return xyzz_dbl(a);
        return xyzz_identity();
    }
    fp ppp = fp_mul(pp_, pp);          // < 1.03
    fp qv = fp_mul(u1, pp);            // < 1.01
    fp x3 = fp_sub_b_2c(fp_sqr(r), ppp, qv);  // r^2 < 1.1, ppp + 2 qv < 3.05;  x3 < 6.1   (one carry ripple, see xyzz_madd)
    fp y3 = fp_mul_add(r, fp_sub_raw<8>(qv, x3), fp_neg_raw<3>(s1), ppp);  // r*(qv - x3) + (3p - s1)*ppp: (4.06*9.01 + 3*1.03)k + 1 < 1.24
    fp zz3 = fp_mul(fp_mul(a.zz, b.zz), pp);
    fp zzz3 = fp_mul(fp_mul(a.zzz, b.zzz), ppp);
    return xyzz{x3, y3, zz3, zzz3};
}

// XYZZ -> Jacobian without inversion: Z = ZZ*ZZZ, X' = X*ZZ^4, Y' = Y*ZZZ^4
// (then X'/Z^2 = X*ZZ^4/(ZZ^2*ZZZ^2) = X*ZZ^4/ZZ^5 = X/ZZ and Y'/Z^3 = Y*ZZZ^4/ZZZ^5 = Y/ZZZ).
FP_HD jacobian xyzz_to_jacobian(const xyzz& p) {
    if (xyzz_is_identity(p)) return jacobian{fp_one(), fp_one(), fp_zero()};  // (1,1,0): SH/constants.metal:175-228
    fp zz2 = fp_sqr(p.zz), zzz2 = fp_sqr(p.zzz);
    return jacobian{fp_mul(p.x, fp_sqr(zz2)), fp_mul(p.y, fp_sqr(zzz2)), fp_mul(p.zz, p.zzz)};
}
FP_HD xyzz xyzz_from_jacobian(const jacobian& p) {
    if (fp_is_zero_lt2p(p.z)) return xyzz_identity();  // z comes straight from canonical words: < 1.01p
    fp zz = fp_sqr(p.z);
    return xyzz{p.x, p.y, zz, fp_mul(zz, p.z)};
}
// affine coordinates (internal domain, < 2p); returns true for the identity (x = y = 0 then)
__host__ __device__ inline bool xyzz_to_affine(const xyzz& p, affine& out) {
    if (xyzz_is_identity(p)) {
        out.x = fp_zero();
        out.y = fp_zero();
        return true;
    }
    fp iz = fp_inv(p.zzz);    // 1/ZZZ
    fp t = fp_mul(iz, p.zz);  // ZZ/ZZZ
    fp izz = fp_sqr(t);       // ZZ^2/ZZZ^2 = ZZ^2/ZZ^3 = 1/ZZ
    out.x = fp_mul(p.x, izz);
    out.y = fp_mul(p.y, iz);
    return false;
}

}  // namespace bn254
