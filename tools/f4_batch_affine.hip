// tools/f4_batch_affine.hip -- SURVEY section 8 row f4 / VERDICT r1 item 6(i): would AFFINE bucket accumulation with batched
// (Montgomery-trick) inversion beat the XYZZ mixed addition of k_accumulate?
//
// One round of the batch-affine scheme on a sorted stream: entries 2i and 2i+1 of the stream (neighbours inside a bucket) are added
// as affine points; a thread owns B independent pair additions and shares ONE inversion with its whole wavefront:
//   pass 1   d_i = x_Q - x_P,  prefix_i = d_0 ... d_(i-1) -> scratch (36 B per addition, coalesced)                  1 M
//   invert   lane totals -> wavefront product (shuffle scans) -> ONE inversion per wavefront (64 x B additions) -> lane inverses
//   pass 2   1/d_i = running * prefix_i, running *= d_i; lambda = (y_Q - y_P)/d_i; x_3 = lambda^2 - x_P - x_Q;
//            y_3 = lambda (x_P - x_3) - y_P -> 64-byte affine record out (both points are gathered AGAIN: 2 x 64 B)    4 M + 1 S
// i.e. 5 M + 1 S per addition + the inversion share, against 8 M + 2 S of xyzz_madd -- but 384 B of traffic per addition against
// 68 B, and a complete accumulation needs ~2 N W such additions' worth of rounds (N W / 2 in the first round, then halving, every
// later round reading and writing 64-byte intermediates).  The first round is the cheapest per addition (its inputs are the
// resident base array); it is what this tool times, with the inversion (a) priced at ZERO -- a lower bound no implementation
// can beat -- and (b) done for real (Fermat, redundantly on all lanes, once per wavefront).
// Baseline in the same tool: the XYZZ mixed addition over the same stream, L = 32 entries per thread (k_accumulate's inner loop).
// The stream is synthetic: uniformly random point indices (what the indices inside a bucket are), 2^20 bases, 16 x 2^20 entries.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value tools/f4_batch_affine.hip -o tools/f4_batch_affine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../gpu-acceleration_amd/csrc/msm_kernels.hpp"
using namespace bn254;
using namespace msmk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fill(uint32_t* bases, uint32_t n, uint32_t* sorted, uint32_t entries) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 16u * n) bases[i] = (i * 2654435761u) ^ (i >> 7) & 0x0FFFFFFFu;  // canonical-looking words (top word small)
    if (i < 16u * n && (i & 7u) == 7u) bases[i] &= 0x1FFFFFFFu;
    if (i < entries) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31;
        sorted[i] = (uint32_t)(z % n) | ((z >> 40) & 1u ? SIGN_BIT : 0u);
    }
}
__global__ void __launch_bounds__(256) k_xyzz_stream(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted, uint32_t entries,
                                                     uint32_t L, uint32_t* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t j0 = t * L;
    if (j0 >= entries) return;
    uint32_t j1 = min(entries, j0 + L);
    xyzz acc = xyzz_identity();
    uint32_t e = sorted[j0];
    affine q = load_affine(bases + (size_t)(e & ~SIGN_BIT) * 16);
    for (uint32_t j = j0; j < j1; j++) {
        affine cur = q;
        const uint32_t ec = e;
        if (j + 1 < j1) {
            e = sorted[j + 1];
            q = load_affine(bases + (size_t)(e & ~SIGN_BIT) * 16);
        }
        if (ec & SIGN_BIT) cur.y = fp_neg<2>(cur.y);
        xyzz_madd(acc, cur);
    }
    store_xyzz(out + (size_t)t * XW, acc);
}
__device__ __forceinline__ fp shfl_fp(const fp& a, int src) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = __shfl(a.v[i], src, 64);
    return r;
}
// MODE 0: inversion priced at zero (the lane total stands in for its own inverse: same instruction mix, wrong numbers);
// MODE 1: one Fermat inversion per wavefront, lane inverses from shuffle scans of the lane totals
template <int MODE>
__global__ void __launch_bounds__(256) k_affine_round(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted, uint32_t npairs,
                                                      uint32_t B, uint32_t* __restrict__ scratch, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    const uint32_t p0 = t * B;
    fp run = fp_one();
    for (uint32_t i = 0; i < B; i++) {  // pass 1
        const uint32_t p = p0 + i;
        fp d = fp_one();
        if (p < npairs) {
            const fp xp = load_fp_packed(bases + (size_t)(sorted[2 * p] & ~SIGN_BIT) * 16);
            const fp xq = load_fp_packed(bases + (size_t)(sorted[2 * p + 1] & ~SIGN_BIT) * 16);
            d = fp_sub<2>(xq, xp);
        }
        uint32_t* s = scratch + ((size_t)i * nthreads + t) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) s[k] = run.v[k];
        run = fp_mul(run, d);
    }
    fp inv = run;
    if (MODE == 1) {
        const int lane = threadIdx.x & 63;
        fp pre = run, suf = run;  // inclusive prefix / suffix products over the lanes
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            fp a = shfl_fp(pre, lane - dd < 0 ? lane : lane - dd), b = shfl_fp(suf, lane + dd > 63 ? lane : lane + dd);
            if (lane >= dd) pre = fp_mul(pre, a);
            if (lane + dd <= 63) suf = fp_mul(suf, b);
        }
        const fp total = shfl_fp(pre, 63);
        const fp tinv = fp_inv(total);  // every lane, redundantly: one inversion per wavefront
        fp ex_pre = shfl_fp(pre, lane == 0 ? 0 : lane - 1), ex_suf = shfl_fp(suf, lane == 63 ? 63 : lane + 1);
        if (lane == 0) ex_pre = fp_one();
        if (lane == 63) ex_suf = fp_one();
        inv = fp_mul(tinv, fp_mul(ex_pre, ex_suf));  // 1 / (this lane's total)
    }
    for (uint32_t i = B; i-- > 0;) {  // pass 2
        const uint32_t p = p0 + i;
        if (p >= npairs) continue;
        const uint32_t ep = sorted[2 * p], eq = sorted[2 * p + 1];
        affine P = load_affine(bases + (size_t)(ep & ~SIGN_BIT) * 16), Q = load_affine(bases + (size_t)(eq & ~SIGN_BIT) * 16);
        if (ep & SIGN_BIT) P.y = fp_neg<2>(P.y);
        if (eq & SIGN_BIT) Q.y = fp_neg<2>(Q.y);
        const fp d = fp_sub<2>(Q.x, P.x);
        fp pref;
        const uint32_t* s = scratch + ((size_t)i * nthreads + t) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) pref.v[k] = s[k];
        const fp dinv = fp_mul(inv, pref);
        inv = fp_mul(inv, d);
        const fp lam = fp_mul(fp_sub<3>(Q.y, P.y), dinv);                     // P.y, Q.y < 2p
        const fp x3 = fp_sub<3>(fp_sqr(lam), fp_add(P.x, Q.x));                // < 1.1p + 3p
        const fp y3 = fp_sub<3>(fp_mul(lam, fp_sub<6>(P.x, x3)), P.y);
        uint32_t w[8];
        fp_pack(w, fp_canonical(x3));
        store_words8(out + (size_t)p * 16, w);
        fp_pack(w, fp_canonical(y3));
        store_words8(out + (size_t)p * 16 + 8, w);
    }
}
template <typename F>
static float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    const uint32_t n = 1u << 20, W = 16, entries = W * n, npairs = entries / 2;
    uint32_t *bases, *sorted, *scratch, *out;
    CK(hipMalloc((void**)&bases, (size_t)n * 64));
    CK(hipMalloc((void**)&sorted, (size_t)entries * 4));
    CK(hipMalloc((void**)&scratch, (size_t)npairs * 36 + (1 << 20) * 36));
    CK(hipMalloc((void**)&out, (size_t)npairs * 64 + (size_t)(entries / 8) * 144));
    k_fill<<<(entries + 255) / 256, 256>>>(bases, n, sorted, entries);
    CK(hipDeviceSynchronize());
    printf("# 2^20 bases (64 MB), %u sorted entries (random point indices), MI355X\n", entries);
    for (uint32_t L : {16u, 32u, 64u}) {
        const uint32_t nt = (entries + L - 1) / L;
        float ms = time_ms([&] { k_xyzz_stream<<<(nt + 255) / 256, 256>>>(bases, sorted, entries, L, out); });
        printf("xyzz_madd stream      L=%3u                      : %8.3f ms for %u additions = %6.1f ps per addition\n", L, ms, entries, ms * 1e9 / entries);
    }
    for (uint32_t B : {4u, 8u, 16u, 32u, 64u}) {
        const uint32_t nt = (npairs + B - 1) / B, blocks = (nt + 255) / 256;
        float m0 = time_ms([&] { k_affine_round<0><<<blocks, 256>>>(bases, sorted, npairs, B, scratch, out); });
        float m1 = time_ms([&] { k_affine_round<1><<<blocks, 256>>>(bases, sorted, npairs, B, scratch, out); });
        printf("batch-affine round 1  B=%3u (%5u adds/inversion): inversion free %8.3f ms = %6.1f ps per addition | Fermat per wavefront %8.3f ms = %6.1f ps per addition\n",
               B, 64 * B, m0, m0 * 1e9 / npairs, m1, m1 * 1e9 / npairs);
    }
    printf("# a complete accumulation runs ~2x the additions of round 1 (N W / 2, N W / 4, ...), the later rounds on 64-byte intermediates\n");
    return 0;
}
