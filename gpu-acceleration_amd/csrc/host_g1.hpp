// host_g1.hpp -- host-side (CPU) BN254 G1 arithmetic of the PRODUCT: the few hundred group
// operations that finish an MSM after the GPU has produced one sum per window.
//
// Counterpart of the reference's CPU finish, MetalMSMPipeline::final_reduction
// (metal_msm.rs:204-261: Montgomery->standard via raw_reduction, utils/mont_reduction.rs:15-40, then
// Horner `result = result*2^w + G_i` with arkworks group ops).  arkworks is not available to a C++
// runtime, so the same operations are provided here on 4 x 64-bit limbs (the arkworks Fq layout,
// R = 2^256) -- a lone CPU core finishes a serial chain of ~300 dependent group operations in
// ~0.1 ms, which a lone GPU wavefront cannot (one dependent 256-bit modmul is ~1 us there).
//
// Independent of oracle/ (which is test infrastructure and never linked into the product).
#pragma once
#include <cstdint>
#include <cstring>

namespace hostg1 {

typedef unsigned __int128 u128;

struct Fq {
    uint64_t l[4];
};

static constexpr Fq MOD = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static constexpr Fq ONE = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};  // R mod p
static constexpr Fq RSQ = {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};  // R^2 mod p
static constexpr uint64_t NINV = 0x87d20782e4866389ULL;  // -p^-1 mod 2^64

inline bool is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline bool geq_mod(const Fq& a) {
    for (int i = 3; i >= 0; --i) {
        if (a.l[i] != MOD.l[i]) return a.l[i] > MOD.l[i];
    }
    return true;
}
inline void sub_mod_inplace(Fq& a) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - MOD.l[i] - borrow;
        a.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 127);
    }
}
inline Fq add(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 s = (u128)a.l[i] + b.l[i] + carry;
        r.l[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    if (geq_mod(r)) sub_mod_inplace(r);
    return r;
}
inline Fq sub(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 127);
    }
    if (borrow) {
        uint64_t carry = 0;
        for (int i = 0; i < 4; ++i) {
            u128 s = (u128)r.l[i] + MOD.l[i] + carry;
            r.l[i] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
    }
    return r;
}
inline Fq dbl(const Fq& a) { return add(a, a); }
// Montgomery product, CIOS with product and reduction interleaved per limb.  The modulus leaves its top bit free
// (p < 2^254), so the running value never needs a fifth word ("no-carry" CIOS): t3 = carry_a + carry_m cannot overflow.
inline Fq mul(const Fq& a, const Fq& b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int i = 0; i < 4; ++i) {
        const uint64_t bi = b.l[i];
        u128 pa = (u128)a.l[0] * bi + t0;               // product chain
        const uint64_t m = (uint64_t)pa * NINV;
        u128 pm = (u128)m * MOD.l[0] + (uint64_t)pa;    // reduction chain (low word becomes 0)
        pa = (u128)a.l[1] * bi + t1 + (uint64_t)(pa >> 64);
        pm = (u128)m * MOD.l[1] + (uint64_t)pa + (uint64_t)(pm >> 64);
        t0 = (uint64_t)pm;
        pa = (u128)a.l[2] * bi + t2 + (uint64_t)(pa >> 64);
        pm = (u128)m * MOD.l[2] + (uint64_t)pa + (uint64_t)(pm >> 64);
        t1 = (uint64_t)pm;
        pa = (u128)a.l[3] * bi + t3 + (uint64_t)(pa >> 64);
        pm = (u128)m * MOD.l[3] + (uint64_t)pa + (uint64_t)(pm >> 64);
        t2 = (uint64_t)pm;
        t3 = (uint64_t)(pa >> 64) + (uint64_t)(pm >> 64);
    }
    Fq r = {{t0, t1, t2, t3}};
    if (geq_mod(r)) sub_mod_inplace(r);
    return r;
}
inline Fq sqr(const Fq& a) { return mul(a, a); }
inline Fq to_mont(const Fq& a) { return mul(a, RSQ); }
inline Fq from_mont(const Fq& a) { return mul(a, Fq{{1, 0, 0, 0}}); }
inline Fq inv(const Fq& a) {  // Fermat, a^(p-2)
    Fq acc = ONE, base = a;
    uint64_t e[4] = {MOD.l[0] - 2, MOD.l[1], MOD.l[2], MOD.l[3]};
    for (int i = 0; i < 254; ++i) {
        if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
        base = sqr(base);
    }
    return acc;
}
inline Fq load_words(const uint32_t* w) {
    Fq r;
    for (int i = 0; i < 4; ++i) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    return r;
}
inline void store_words(uint32_t* w, const Fq& a) {
    for (int i = 0; i < 4; ++i) {
        w[2 * i] = (uint32_t)a.l[i];
        w[2 * i + 1] = (uint32_t)(a.l[i] >> 32);
    }
}

// Jacobian point, identity <=> Z == 0 (X = Y = R as the reference encodes it, constants.metal:175-228)
struct Jac {
    Fq x, y, z;
};
inline Jac identity() { return Jac{ONE, ONE, Fq{{0, 0, 0, 0}}}; }
inline bool is_identity(const Jac& p) { return is_zero(p.z); }

inline Jac jdbl(const Jac& p) {  // dbl-2009-l
    if (is_identity(p)) return p;
    Fq a = sqr(p.x), b = sqr(p.y), c = sqr(b);
    Fq d = dbl(sub(sub(sqr(add(p.x, b)), a), c));
    Fq e = add(dbl(a), a);
    Fq f = sqr(e);
    Fq x3 = sub(f, dbl(d));
    Fq c8 = dbl(dbl(dbl(c)));
    Fq y3 = sub(mul(e, sub(d, x3)), c8);
    Fq z3 = dbl(mul(p.y, p.z));
    return Jac{x3, y3, z3};
}
inline Jac jadd(const Jac& p, const Jac& q) {  // add-2007-bl, complete
    if (is_identity(p)) return q;
    if (is_identity(q)) return p;
    Fq z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    Fq u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    Fq s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
    Fq h = sub(u2, u1), rr = sub(s2, s1);
    if (is_zero(h)) return is_zero(rr) ? jdbl(p) : identity();
    Fq i = sqr(dbl(h));
    Fq j = mul(h, i);
    Fq r = dbl(rr);
    Fq v = mul(u1, i);
    Fq x3 = sub(sub(sqr(r), j), dbl(v));
    Fq y3 = sub(mul(r, sub(v, x3)), dbl(mul(s1, j)));
    Fq z3 = dbl(mul(mul(p.z, q.z), h));
    return Jac{x3, y3, z3};
}
// canonical affine, standard form; returns true for the identity (x = y = 0)
inline bool to_affine_std(const Jac& p, Fq& x, Fq& y) {
    if (is_identity(p)) {
        x = Fq{{0, 0, 0, 0}};
        y = x;
        return true;
    }
    Fq zi = inv(p.z), zi2 = sqr(zi);
    x = from_mont(mul(p.x, zi2));
    y = from_mont(mul(p.y, mul(zi2, zi)));
    return false;
}
inline Jac load_jac(const uint32_t* w) { return Jac{load_words(w), load_words(w + 8), load_words(w + 16)}; }
inline void store_jac(uint32_t* w, const Jac& p) {
    store_words(w, p.x);
    store_words(w + 8, p.y);
    store_words(w + 16, p.z);
}

}  // namespace hostg1
