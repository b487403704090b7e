// msm_testhooks_kernels.hpp -- kernels that exist for tests, benchmarks and calibration only (compiled into
// libmsm_hip_hooks.so with -DMSM_HIP_TEST_HOOKS, never into the product library libmsm_hip.so):
//   synthetic inputs (counterpart of test_utils::generate_random_bases_and_scalars, metal_msm.rs:698-731),
//   device-math unit-test kernels (counterpart of the reference's test_* kernels),
//   the integer-multiplier calibration kernels and the XYZZ -> Jacobian dump of the stage tests.
#pragma once
#include "msm_kernels.hpp"

namespace msmk {

// test hook: plain signed digits as int32
template <bool SIGNED>
__global__ void k_decompose_plain(const uint32_t* __restrict__ scalars, uint32_t n, uint32_t c, uint32_t W,
                                  int32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    for (int k = 0; k < 8; k++) s[k] = scalars[(size_t)i * 8 + k];
    const uint32_t H = 1u << (c - 1);
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; w++) {
        uint32_t v = scalar_window(s, w * c, c) + carry;
        int32_t d = (int32_t)v;
        if (SIGNED) {
            if (v > H) {
                d = (int32_t)v - (int32_t)(2u * H);
                carry = 1;
            } else {
                carry = 0;
            }
        }
        out[(size_t)w * n + i] = d;
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic inputs (counterpart of test_utils::generate_random_bases_and_scalars, metal_msm.rs:698-731)
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t& st) {
    uint64_t z = (st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// element i of stream `seed`: 254-bit rejection sampling below r (and != 0 when nonzero)
__host__ __device__ inline void gen_scalar(uint64_t seed, uint64_t i, bool nonzero, uint32_t out[8]) {
    const uint64_t R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    uint64_t st = seed + i * 0xD1342543DE82EF95ULL;
    uint64_t v[4];
    for (;;) {
        for (int k = 0; k < 4; k++) v[k] = splitmix64(st);
        v[3] &= 0x3FFFFFFFFFFFFFFFULL;
        bool lt = false;
        for (int k = 3; k >= 0; k--) {
            if (v[k] != R[k]) {
                lt = v[k] < R[k];
                break;
            }
        }
        if (!lt) continue;
        if (nonzero && (v[0] | v[1] | v[2] | v[3]) == 0) continue;
        break;
    }
    for (int k = 0; k < 4; k++) {
        out[2 * k] = (uint32_t)v[k];
        out[2 * k + 1] = (uint32_t)(v[k] >> 32);
    }
}
__global__ void k_gen_scalars(uint64_t seed, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    gen_scalar(seed, i, false, s);
    uint4* q = reinterpret_cast<uint4*>(out + (size_t)i * 8);
    q[0] = make_uint4(s[0], s[1], s[2], s[3]);
    q[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
// base i = k_i * G with k_i = nonzero stream element; pow2_table[j] = 2^j * G (affine, internal domain, packed,
// 254 entries); the result is written as canonical R = 2^256 Montgomery words (what arkworks holds)
__global__ void __launch_bounds__(128) k_gen_bases(uint64_t seed, uint32_t n, const uint32_t* __restrict__ pow2_table,
                                                   uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8];
    gen_scalar(seed, i, true, k);
    xyzz acc = xyzz_identity();
    for (int j = 0; j < SCALAR_BITS; j++) {
        if ((k[j >> 5] >> (j & 31)) & 1u) xyzz_madd(acc, load_affine(pow2_table + (size_t)j * 16));
    }
    affine a;
    xyzz_to_affine(acc, a);
    uint32_t w[8];
    fp_to_mont256(w, a.x);
    store_words8(out + (size_t)i * 16, w);
    fp_to_mont256(w, a.y);
    store_words8(out + (size_t)i * 16 + 8, w);
}

// ---------------------------------------------------------------------------------------------
// device-math unit-test kernels (counterpart of the reference's test_* kernels)
__global__ void k_test_fp(uint32_t op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                          uint32_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // operands and results cross the ABI as canonical R = 2^256 Montgomery words (ops 0,1,2,5) or as the
    // standard/Montgomery pair of the conversion ops (3: std -> mont, 4: mont -> std)
    uint32_t wa[8], wb[8], wr[8];
    for (int k = 0; k < 8; k++) {
        wa[k] = a[(size_t)i * 8 + k];
        wb[k] = b ? b[(size_t)i * 8 + k] : 0u;
    }
    switch (op) {
        case 0: fp_to_mont256(wr, fp_add(fp_from_mont256(wa), fp_from_mont256(wb))); break;
        case 1: fp_to_mont256(wr, fp_sub<3>(fp_from_mont256(wa), fp_from_mont256(wb))); break;
        case 2: fp_to_mont256(wr, fp_mul(fp_from_mont256(wa), fp_from_mont256(wb))); break;
        case 3: fp_to_mont256(wr, fp_from_std(wa)); break;
        case 4: fp_to_std(wr, fp_from_mont256(wa)); break;
        default: {
            fp x = fp_from_mont256(wa);
            if (fp_is_zero_lt2p(x)) fp_to_mont256(wr, fp_zero());
            else fp_to_mont256(wr, fp_inv(x));
            break;
        }
    }
    for (int k = 0; k < 8; k++) out[(size_t)i * 8 + k] = wr[k];
}
__global__ void __launch_bounds__(64) k_test_g1(uint32_t op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                uint32_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    xyzz p = xyzz_from_jacobian(load_jacobian_mont256(a + (size_t)i * 24));
    xyzz r;
    if (op == 0) {
        const uint32_t* bp = b + (size_t)i * 16;
        uint32_t wx[8], wy[8];
        for (int k = 0; k < 8; k++) {
            wx[k] = bp[k];
            wy[k] = bp[8 + k];
        }
        affine q{fp_from_mont256(wx), fp_from_mont256(wy)};
        r = p;
        xyzz_madd(r, q);
    } else if (op == 1) {
        r = xyzz_add(p, xyzz_from_jacobian(load_jacobian_mont256(b + (size_t)i * 24)));
    } else if (op == 4 || op == 5) {
        const uint32_t* bp = b + (size_t)i * 16;
        uint32_t wx[8], wy[8];
        for (int k = 0; k < 8; k++) {
            wx[k] = bp[k];
            wy[k] = bp[8 + k];
        }
        r = p;
        xyzz_madd_m32(r, fp_unpack_shl5(wx), fp_unpack_shl5(wy), op == 5);
    } else if (op == 3) {
        // never reached: op 3 (wide addition) has its own kernel, k_test_g1_wide
        r = p;
    } else {
        r = xyzz_dbl(p);
    }
    store_jacobian_mont256(out + (size_t)i * 24, xyzz_to_jacobian(r));
}

// Calibration kernels for the integer-multiplier roofline ("against the measured v_mad_u64_u32 peak of
// a calibration micro-kernel on the same device"): dependent chains at 4 wavefronts per SIMD, as k_accumulate runs them.
//   what = 0: raw v_mad_u64_u32 (64 per iteration);  1: fp_mul, the 9 x 29-bit Montgomery multiplication (4 per iteration)
__global__ void __launch_bounds__(256) k_calibrate(uint32_t what, uint32_t iters, uint32_t* __restrict__ sink) {
    uint32_t acc_out = 0;
    if (what == 0) {
        uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + 12345u;
        const uint32_t y = (uint32_t)(a >> 32) | 1u;
        for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 64; k++) a = (uint64_t)(uint32_t)a * y + a;  // one v_mad_u64_u32, operand and addend from the chain (wraps)
        }
        acc_out = (uint32_t)a ^ (uint32_t)(a >> 32);
    } else {
        fp v = fp_one(), m = fp_one();
        v.v[0] += threadIdx.x;
        m.v[1] += 7 + threadIdx.x;
        for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) v = fp_mul(v, m);
        }
#pragma unroll
        for (int k = 0; k < 9; k++) acc_out ^= v.v[k];
    }
    if (acc_out == 0x12345u) sink[0] = acc_out;  // never true in practice: keeps the chains alive
}

// test hook for ec_wide.hpp: pair i is added by the 8 lanes of group i (records staged in LDS by the group's first lane)
__global__ void __launch_bounds__(64) k_test_g1_wide(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                     uint32_t* __restrict__ out, uint32_t n) {
    __shared__ uint32_t e[16 * XW];
    const uint32_t g = threadIdx.x / WIDE_LANES, role = threadIdx.x % WIDE_LANES;
    const uint32_t i = blockIdx.x * 8 + g;
    if (i < n && role == 0) {
        store_xyzz(e + (size_t)(2 * g) * XW, xyzz_from_jacobian(load_jacobian_mont256(a + (size_t)i * 24)));
        store_xyzz(e + (size_t)(2 * g + 1) * XW, xyzz_from_jacobian(load_jacobian_mont256(b + (size_t)i * 24)));
    }
    __syncthreads();
    if (i < n) wide_add_records(e + (size_t)(2 * g) * XW, e + (size_t)(2 * g + 1) * XW, e + (size_t)(2 * g) * XW);
    __syncthreads();
    if (i < n && role == 0) store_jacobian_mont256(out + (size_t)i * 24, xyzz_to_jacobian(load_xyzz(e + (size_t)(2 * g) * XW)));
}

// Probe: ONE workgroup runs `iters` pairwise levels of the same size over m records in LDS -- the body of
// lds_tree_wide: e[i] += e[i + m/2] by eight lanes per addition, a barrier per level.  PROBE: wavefront 0 brackets the parts of its
// addition with the shader-cycle counter (ec_wide.hpp wide_mark; each mark drains the memory counters) and sums the intervals; without
// it only the whole loop is bracketed, so the two totals show what the marks themselves cost.
//   out[0] shader cycles of the loop, out[1] constant-rate ticks of the loop, out[2 + k] cycles between mark k - 1 and mark k summed over the levels
//   marks: 0 entry, 1 operands loaded, 2 stage-1 product, 3 exchange + P / R, 4 squares, 5 special-case vote, 6 exchange + stage-3 operands,
//          7 stage-3 product, 8 exchange + X3 + stage-4 operands, 9 stage-4 product, 10 exchange + Y3, 11 result stored, 12 loop left, 13 barrier passed
template <bool PROBE>
__global__ void __launch_bounds__(512) k_probe_wide_level(const uint32_t* __restrict__ src, uint32_t m, uint32_t iters, long long* __restrict__ out) {
    __shared__ uint32_t e[WIDE_TREE_MAX * XW];
    for (uint32_t i = threadIdx.x; i < m * XW; i += blockDim.x) e[i] = src[i];
    __syncthreads();
    const uint32_t g = threadIdx.x / WIDE_LANES, ng = blockDim.x / WIDE_LANES, h = m >> 1;
    long long acc[WIDE_MARKS], ts[WIDE_MARKS];
#pragma unroll
    for (int k = 0; k < WIDE_MARKS; k++) acc[k] = 0, ts[k] = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (uint32_t it = 0; it < iters; it++) {
        for (uint32_t i = g; i < h; i += ng) wide_add_records<PROBE>(e + (size_t)i * XW, e + (size_t)(i + h) * XW, e + (size_t)i * XW, ts);
        wide_mark<PROBE>(ts, 12);
        __syncthreads();
        wide_mark<PROBE>(ts, 13);
        if (PROBE) {
#pragma unroll
            for (int k = 1; k <= 13; k++) acc[k] += ts[k] - ts[k - 1];
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
#pragma unroll
        for (int k = 0; k < WIDE_MARKS; k++) out[2 + k] = acc[k];
    }
}
// the same m / 2 additions by ONE lane each (the form of k_pair_level / the serial folds of k_pair_level8), for the lone-wavefront price of xyzz_add
__global__ void __launch_bounds__(512) k_probe_scalar_level(const uint32_t* __restrict__ src, uint32_t m, uint32_t iters, long long* __restrict__ out) {
    __shared__ uint32_t e[WIDE_TREE_MAX * XW];
    for (uint32_t i = threadIdx.x; i < m * XW; i += blockDim.x) e[i] = src[i];
    __syncthreads();
    const uint32_t h = m >> 1;
    const long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (uint32_t it = 0; it < iters; it++) {
        if (threadIdx.x < h) add_records_complete(e + (size_t)threadIdx.x * XW, e + (size_t)(threadIdx.x + h) * XW, e + (size_t)threadIdx.x * XW);
        __syncthreads();
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = c1 - c0, out[1] = w1 - w0;
}
__global__ void k_probe_empty(uint32_t* sink) {
    if (sink && threadIdx.x == 12345u) sink[0] = 1;
}
// an empty kernel that OWNS resources: LDS_WORDS of static LDS per workgroup (launched with any grid / block): what does dispatching
// k_combine_pieces' 1152 workgroups of 512 threads and 36 KB cost when its lists are empty?
template <int LDS_WORDS>
__global__ void __launch_bounds__(1024) k_probe_empty_lds(uint32_t* sink) {
    __shared__ uint32_t e[LDS_WORDS];
    if (sink && threadIdx.x == 12345u) {
        e[threadIdx.x % LDS_WORDS] = 1;
        sink[0] = e[0];
    }
}

// stage tests: XYZZ records (bucket sums) -> Jacobian R = 2^256 Montgomery words, the C-ABI point format
__global__ void k_dump_xyzz_as_jacobian(const uint32_t* __restrict__ recs, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_jacobian_mont256(out + (size_t)i * 24, xyzz_to_jacobian(load_xyzz(recs + (size_t)i * XW)));
}

}  // namespace msmk
