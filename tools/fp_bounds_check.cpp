// tools/fp_bounds_check.cpp -- host build of the DEVICE field/group code (fp_bn254.hpp / ec_bn254.hpp are
// __host__ __device__) with every limb-range assumption of the lazily reduced 9 x 29-bit arithmetic asserted:
// subtrahend limbs never exceed the K*p pad, normalisation never overflows a limb, no Montgomery column leaves
// 64 bits, values stay below 2^261.  Drives long random chains of xyzz_madd / xyzz_add / xyzz_dbl (the value bounds
// do not depend on the operands being curve points) plus boundary operands, and cross-checks every field result
// against the independent 4 x 64-bit host arithmetic (host_g1.hpp).
// Build+run:  hipcc -O2 -std=c++17 -DFP_BOUNDS_CHECK -x hip --offload-arch=gfx950 tools/fp_bounds_check.cpp -o /tmp/fpchk && /tmp/fpchk
#include <cstdint>
#include <cstdio>
#include <random>
#include "../gpu-acceleration_amd/csrc/ec_bn254.hpp"
#include "../gpu-acceleration_amd/csrc/host_g1.hpp"
using namespace bn254;

static std::mt19937_64 rng(0xB254);
static void rand_words(uint32_t w[8], int mode) {
    const uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    if (mode == 1) { for (int i = 0; i < 8; i++) w[i] = P[i]; w[0] -= 1 + (uint32_t)(rng() % 3); return; }  // p-1..p-3
    if (mode == 2) { for (int i = 0; i < 8; i++) w[i] = 0; w[0] = (uint32_t)(rng() % 3); return; }           // 0..2
    for (;;) {
        for (int i = 0; i < 8; i++) w[i] = (uint32_t)rng();
        w[7] &= 0x3FFFFFFFu;
        bool lt = false;
        for (int i = 7; i >= 0; i--) if (w[i] != P[i]) { lt = w[i] < P[i]; break; }
        if (lt) return;
    }
}
static hostg1::Fq to_host_mont256(const fp& a) { uint32_t w[8]; fp_to_mont256(w, a); return hostg1::load_words(w); }
static bool same(const hostg1::Fq& a, const hostg1::Fq& b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }

static fp add_kp(fp a, int k) {  // a + k*p, limbs re-normalised: the same residue, a larger representative
    for (int j = 0; j < k; j++) {
        for (int i = 0; i < 9; i++) a.v[i] += FP29_P[i];
        a = fp_normalize(a);
    }
    return a;
}
static bool same_residue(const fp& a, const fp& b) {
    uint32_t wa[8], wb[8];
    fp_to_mont256(wa, a);
    fp_to_mont256(wb, b);
    for (int i = 0; i < 8; i++) if (wa[i] != wb[i]) return false;
    return true;
}
static bool same_xyzz(const xyzz& a, const xyzz& b) {
    return same_residue(a.x, b.x) && same_residue(a.y, b.y) && same_residue(a.zz, b.zz) && same_residue(a.zzz, b.zzz);
}

int main() {
    long checks = 0;
    // 1. field ops against the 4x64 host arithmetic, in the R = 2^256 Montgomery domain both sides
    for (int it = 0; it < 200000; it++) {
        uint32_t wa[8], wb[8];
        rand_words(wa, it % 11 == 0 ? 1 : it % 13 == 0 ? 2 : 0);
        rand_words(wb, it % 7 == 0 ? 1 : it % 17 == 0 ? 2 : 0);
        fp a = fp_from_mont256(wa), b = fp_from_mont256(wb);
        hostg1::Fq ha = hostg1::load_words(wa), hb = hostg1::load_words(wb);
        if (!same(to_host_mont256(fp_mul(a, b)), hostg1::mul(ha, hb))) { printf("mul mismatch\n"); return 1; }
        if (!same(to_host_mont256(fp_sqr(a)), hostg1::sqr(ha))) { printf("sqr mismatch\n"); return 1; }
        if (!same(to_host_mont256(fp_add(a, b)), hostg1::add(ha, hb))) { printf("add mismatch\n"); return 1; }
        if (!same(to_host_mont256(fp_sub<3>(a, b)), hostg1::sub(ha, hb))) { printf("sub mismatch\n"); return 1; }
        if (!same(to_host_mont256(fp_mul_add(a, b, b, fp_neg<3>(a))), hostg1::sub(hostg1::mul(ha, hb), hostg1::mul(hb, ha)))) { printf("mul_add mismatch\n"); return 1; }
        checks += 5;
        if (it % 100 == 0) {  // the windowed Fermat inversion (fp_inv) against the host's bit-by-bit one; 0 -> 0
            if (!same(to_host_mont256(fp_inv(a)), hostg1::inv(ha))) { printf("inv mismatch\n"); return 1; }
            checks++;
        }
    }
    {
        uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const fp zero = fp_from_mont256(z);
        if (!same(to_host_mont256(fp_inv(zero)), hostg1::load_words(z))) { printf("inv(0) != 0\n"); return 1; }
    }
    // 2. long chains of group operations on arbitrary field values (bounds are what is being checked)
    for (int chain = 0; chain < 2000; chain++) {
        uint32_t wx[8], wy[8];
        rand_words(wx, chain % 5 == 0 ? 1 : 0);
        rand_words(wy, chain % 7 == 0 ? 1 : chain % 11 == 0 ? 2 : 0);
        affine q{fp_from_mont256(wx), fp_from_mont256(wy)};
        xyzz acc = xyzz_from_affine(q), other = xyzz_identity();
        for (int s = 0; s < 600; s++) {
            rand_words(wx, s % 31 == 0 ? 1 : s % 37 == 0 ? 2 : 0);
            rand_words(wy, s % 29 == 0 ? 1 : 0);
            affine p2{fp_from_mont256(wx), fp_from_mont256(wy)};
            const bool ng = rng() & 1;
            affine p2n = p2, p2r = p2;
            if (ng) p2n.y = fp_neg<2>(p2.y), p2r.y = fp_neg_raw<2>(p2.y);
            p2 = p2n;
            {   // the lazily normalised mixed addition (raw negated y) against the fully normalised one
                xyzz a1 = acc, a2 = acc;
                xyzz_madd(a1, p2r);
                xyzz_madd_plain(a2, p2n);
                if (!same_xyzz(a1, a2) || xyzz_is_identity(a1) != xyzz_is_identity(a2)) { printf("madd: lazy and plain differ (chain %d step %d)\n", chain, s); return 1; }
            }
            xyzz_madd(acc, p2r);
            if (s % 5 == 0) { xyzz_madd(other, p2); other = xyzz_add(other, acc); }
            if (s % 7 == 0) acc = xyzz_dbl(acc);
            if (s % 97 == 0) acc = xyzz_add(acc, acc);  // takes the doubling branch
            if (s % 101 == 0) { jacobian j = xyzz_to_jacobian(other); uint32_t w[8]; fp_to_mont256(w, j.x); fp_to_mont256(w, j.z); }
            checks += 3;
        }
    }
    // 3. worst-case representatives: the same operations on operands pushed to the top of their documented ranges
    //    (X + 6p < 7p, Y + 4p < 5p, ZZ + p, ZZZ + p < 2p; affine y given as 2p - y) must pass every assertion and
    //    produce the same residues as on the reduced operands
    for (int it = 0; it < 100000; it++) {
        uint32_t w[6][8];
        for (int j = 0; j < 6; j++) rand_words(w[j], (it + j) % 19 == 0 ? 1 : (it + j) % 23 == 0 ? 2 : 0);
        // the identity is the EXACT limb pattern ZZ = 0 (only ever produced by xyzz_identity()), so residues 0 are
        // kept out of ZZ/ZZZ here; they are exercised in part 2
        if ((w[2][0] | w[2][1] | w[2][7]) == 0) w[2][0] = 5;
        if ((w[1][0] | w[1][1] | w[1][7]) == 0) w[1][0] = 5;
        xyzz a{fp_from_mont256(w[0]), fp_from_mont256(w[1]), fp_from_mont256(w[2]), fp_from_mont256(w[3])};
        xyzz ai{add_kp(a.x, 6), add_kp(a.y, 4), add_kp(a.zz, 1), add_kp(a.zzz, 1)};
        affine q{fp_from_mont256(w[4]), fp_from_mont256(w[5])};
        affine qi{q.x, add_kp(q.y, 1)};
        xyzz r1 = a, r2 = ai;
        if (it & 1) {  // negated y: normalised on the reduced operands, RAW (no carry ripple) on the inflated ones
            xyzz_madd(r1, affine{q.x, fp_neg<2>(q.y)});
            xyzz_madd(r2, affine{q.x, fp_neg_raw<2>(q.y)});
        } else {
            xyzz_madd(r1, q);
            xyzz_madd(r2, qi);
        }
        if (!same_xyzz(r1, r2)) { printf("madd: inflated operands change the residues (it=%d)\n", it); return 1; }
        if (!same_xyzz(xyzz_dbl(a), xyzz_dbl(ai))) { printf("dbl: inflated operands change the residues\n"); return 1; }
        xyzz b{fp_from_mont256(w[4]), fp_from_mont256(w[5]), fp_from_mont256(w[1]), fp_from_mont256(w[0])};
        xyzz bi{add_kp(b.x, 6), add_kp(b.y, 4), add_kp(b.zz, 1), add_kp(b.zzz, 1)};
        if (!same_xyzz(xyzz_add(a, b), xyzz_add(ai, bi))) { printf("add: inflated operands change the residues\n"); return 1; }
        if (!same_xyzz(xyzz_add(a, bi), xyzz_add(ai, b))) { printf("add (mixed): inflated operands change the residues\n"); return 1; }
        checks += 4;
    }
    // 4. round 5: the mixed addition on the caller's R = 2^256 Montgomery words as they are (fp_unpack_shl5 + xyzz_madd_m32) against the
    //    normalised one on converted operands -- canonical words, ANY 256-bit words (all ones: the largest multiplier, 2^261 - 32), reduced and
    //    inflated accumulators, both signs, the identity start and the doubling / cancelling branches
    for (int it = 0; it < 200000; it++) {
        uint32_t w[6][8];
        for (int j = 0; j < 6; j++) rand_words(w[j], (it + j) % 19 == 0 ? 1 : (it + j) % 23 == 0 ? 2 : 0);
        if ((w[2][0] | w[2][1] | w[2][7]) == 0) w[2][0] = 5;
        if (it % 3 == 1) for (int j = 4; j < 6; j++) for (int k = 0; k < 8; k++) w[j][k] = (uint32_t)rng();  // not canonical: any 256 bits
        if (it % 1000 == 7) for (int j = 4; j < 6; j++) for (int k = 0; k < 8; k++) w[j][k] = 0xFFFFFFFFu;
        xyzz a{fp_from_mont256(w[0]), fp_from_mont256(w[1]), fp_from_mont256(w[2]), fp_from_mont256(w[3])};
        if (it % 50 == 0) a = xyzz_identity();
        const xyzz ai = xyzz_is_identity(a) ? a : xyzz{add_kp(a.x, 6), add_kp(a.y, 4), add_kp(a.zz, 1), add_kp(a.zzz, 1)};
        const bool ng = (it >> 1) & 1;
        const fp qx = fp_unpack_shl5(w[4]), qy = fp_unpack_shl5(w[5]);
        for (int i = 0; i < 8; i++)
            if (qx.v[i] > FP_MASK || qy.v[i] > FP_MASK) { printf("unpack_shl5: limb not normalised\n"); return 1; }
        affine q{fp_from_mont256(w[4]), fp_from_mont256(w[5])};  // (fp_mul takes any 256-bit words: the value mod p)
        q.x = fp_reduce_lt2p(q.x), q.y = fp_reduce_lt2p(q.y);
        if (!same_residue(qx, q.x) || !same_residue(qy, q.y)) { printf("unpack_shl5: 32 * W is not the internal-domain value\n"); return 1; }
        if (ng) q.y = fp_neg<2>(q.y);
        xyzz r0 = a, r1 = a, r2 = ai;
        xyzz_madd_plain(r0, q);
        xyzz_madd_m32(r1, qx, qy, ng);
        xyzz_madd_m32(r2, qx, qy, ng);
        if (!same_xyzz(r0, r1) || !same_xyzz(r0, r2) || xyzz_is_identity(r0) != xyzz_is_identity(r1) || xyzz_is_identity(r0) != xyzz_is_identity(r2)) {
            printf("madd_m32 differs from the plain mixed addition (it=%d)\n", it);
            return 1;
        }
        // same point again (doubling branch), then its inverse (cancels to the identity)
        xyzz d0 = r0, d1 = r1;
        xyzz_madd_plain(d0, q);
        xyzz_madd_m32(d1, qx, qy, ng);
        if (xyzz_is_identity(a)) {  // a was the identity: r = q, so this was q + q
            if (!same_xyzz(d0, d1)) { printf("madd_m32: doubling branch differs (it=%d)\n", it); return 1; }
            xyzz c1 = r1;
            xyzz_madd_m32(c1, qx, qy, !ng);
            const bool y_zero = fp_is_zero_lt2p(fp_mul(qy, fp_one()));  // (y = 0 is its own inverse -- not a curve point, but a boundary operand here)
            if (!y_zero && !xyzz_is_identity(c1)) { printf("madd_m32: q + (-q) is not the identity (it=%d)\n", it); return 1; }
        }
        // a long chain keeps the bounds
        if (it % 100 == 0) {
            xyzz c0 = r0, c1 = r1;
            for (int s = 0; s < 200; s++) {
                uint32_t u[2][8];
                for (int j = 0; j < 2; j++) for (int k = 0; k < 8; k++) u[j][k] = (s % 3) ? (uint32_t)rng() : 0xFFFFFFFFu - (uint32_t)(rng() % 4);
                affine t{fp_reduce_lt2p(fp_from_mont256(u[0])), fp_reduce_lt2p(fp_from_mont256(u[1]))};
                const bool g2 = rng() & 1;
                if (g2) t.y = fp_neg<2>(t.y);
                xyzz_madd_plain(c0, t);
                xyzz_madd_m32(c1, fp_unpack_shl5(u[0]), fp_unpack_shl5(u[1]), g2);
            }
            if (!same_xyzz(c0, c1)) { printf("madd_m32: chain differs (it=%d)\n", it); return 1; }
        }
        checks += 4;
    }
    printf("fp_bounds_check: %ld checked operations, no bound violated, field results identical to the 4x64 host arithmetic\n", checks);
    return 0;
}
