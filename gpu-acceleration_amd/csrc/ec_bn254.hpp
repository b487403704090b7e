// ec_bn254.hpp -- BN254 G1 (y^2 = x^3 + 3) group law for gfx950 and host.
//
// Replaces the reference's SH/curve/jacobian.metal:11-226 and curve/utils.metal:9-31.
// Differences, all deliberate:
//  * accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
//    a mixed add is 8M+2S instead of the reference's 16-multiplication add-2007-bl with Z2 = R
//    (smvp.metal:61-71), and no field inversion is ever needed on the device;
//  * the group law is COMPLETE: P+P, P+(-P), infinity operands are decided projectively
//    (U2 == X1 etc.), not by limb equality (jacobian.metal:52-53) -- arkworks, the behaviour to
//    match, is complete; the reference is not (SURVEY.md section 4 "gaps");
//  * identity is ZZ == 0 (the reference encodes it as Z == 0 with X = Y = R).
// Formulas: EFD shortw/xyzz  madd-2008-s, add-2008-s, dbl-2008-s-1 / mdbl-2008-s-1 (a = 0).
#pragma once
#include "fp_bn254.hpp"

namespace bn254 {

struct affine {  // Montgomery coordinates; the point at infinity is carried out of band
    fp x, y;
};
struct xyzz {
    fp x, y, zz, zzz;
};
struct jacobian {  // X/Z^2, Y/Z^3; identity <=> Z == 0.  Output format of the C ABI.
    fp x, y, z;
};

FP_HD xyzz xyzz_identity() { return xyzz{fp_one(), fp_one(), fp_zero(), fp_zero()}; }
FP_HD bool xyzz_is_identity(const xyzz& p) { return fp_is_zero(p.zz); }
FP_HD xyzz xyzz_from_affine(const affine& a) { return xyzz{a.x, a.y, fp_one(), fp_one()}; }
FP_HD xyzz xyzz_neg(const xyzz& p) { return xyzz{p.x, fp_neg(p.y), p.zz, p.zzz}; }
FP_HD affine affine_neg(const affine& a) { return affine{a.x, fp_neg(a.y)}; }

// 2*(x,y) for an affine point: mdbl-2008-s-1 with a = 0.   (y != 0 on BN254: no 2-torsion)
FP_HD xyzz xyzz_dbl_affine(const affine& a) {
    fp u = fp_dbl(a.y);
    fp v = fp_sqr(u);
    fp w = fp_mul(u, v);
    fp s = fp_mul(a.x, v);
    fp xx = fp_sqr(a.x);
    fp m = fp_add(fp_dbl(xx), xx);
    fp x3 = fp_sub(fp_sqr(m), fp_dbl(s));
    fp y3 = fp_sub(fp_mul(m, fp_sub(s, x3)), fp_mul(w, a.y));
    return xyzz{x3, y3, v, w};
}
// dbl-2008-s-1, a = 0
FP_HD xyzz xyzz_dbl(const xyzz& p) {
    if (xyzz_is_identity(p)) return p;
    fp u = fp_dbl(p.y);
    fp v = fp_sqr(u);
    fp w = fp_mul(u, v);
    fp s = fp_mul(p.x, v);
    fp xx = fp_sqr(p.x);
    fp m = fp_add(fp_dbl(xx), xx);
    fp x3 = fp_sub(fp_sqr(m), fp_dbl(s));
    fp y3 = fp_sub(fp_mul(m, fp_sub(s, x3)), fp_mul(w, p.y));
    return xyzz{x3, y3, fp_mul(v, p.zz), fp_mul(w, p.zzz)};
}

// acc += (x2,y2)   madd-2008-s, 8M+2S, complete.
FP_HD void xyzz_madd(xyzz& acc, const affine& q) {
    if (xyzz_is_identity(acc)) {
        acc = xyzz_from_affine(q);
        return;
    }
    fp u2 = fp_mul(q.x, acc.zz);
    fp s2 = fp_mul(q.y, acc.zzz);
    fp pp_ = fp_sub(u2, acc.x);  // P
    fp r = fp_sub(s2, acc.y);    // R
    if (fp_is_zero(pp_)) {
        if (fp_is_zero(r)) acc = xyzz_dbl_affine(q);  // same point
        else acc = xyzz_identity();                   // opposite points
        return;
    }
    fp pp = fp_sqr(pp_);
    fp ppp = fp_mul(pp_, pp);
    fp qv = fp_mul(acc.x, pp);
    fp x3 = fp_sub(fp_sub(fp_sqr(r), ppp), fp_dbl(qv));
    fp y3 = fp_sub(fp_mul(r, fp_sub(qv, x3)), fp_mul(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fp_mul(acc.zz, pp);
    acc.zzz = fp_mul(acc.zzz, ppp);
}

// a + b, add-2008-s, 12M+2S, complete.
FP_HD xyzz xyzz_add(const xyzz& a, const xyzz& b) {
    if (xyzz_is_identity(a)) return b;
    if (xyzz_is_identity(b)) return a;
    fp u1 = fp_mul(a.x, b.zz);
    fp u2 = fp_mul(b.x, a.zz);
    fp s1 = fp_mul(a.y, b.zzz);
    fp s2 = fp_mul(b.y, a.zzz);
    fp pp_ = fp_sub(u2, u1);
    fp r = fp_sub(s2, s1);
    if (fp_is_zero(pp_)) {
        if (fp_is_zero(r)) return xyzz_dbl(a);
        return xyzz_identity();
    }
    fp pp = fp_sqr(pp_);
    fp ppp = fp_mul(pp_, pp);
    fp qv = fp_mul(u1, pp);
    fp x3 = fp_sub(fp_sub(fp_sqr(r), ppp), fp_dbl(qv));
    fp y3 = fp_sub(fp_mul(r, fp_sub(qv, x3)), fp_mul(s1, ppp));
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
    if (fp_is_zero(p.z)) return xyzz_identity();
    fp zz = fp_sqr(p.z);
    return xyzz{p.x, p.y, zz, fp_mul(zz, p.z)};
}
// canonical affine (Montgomery); returns true for the identity (x = y = 0 then)
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
