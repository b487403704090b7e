// tools/mfma_probe.hip -- VERDICT r1 item 5c: can the constant-operand half of a Montgomery multiplication (m * p, p fixed, and
// m = T * p' mod R) move from the VALU to the idle MFMA pipe?
//
// The constant product is a Toeplitz matrix-vector product.  With signed 8-bit MFMA operands the 261-bit m is 38 digits of 7 bits
// and p 37 digits: [64 lanes x 38] x [38 x 74 Toeplitz(p)] -> 74 column sums per lane (i32).  On v_mfma_i32_32x32x32_i8 that is
// 2 (M: 64 lanes) x 3 (N: 74 -> 96) x 2 (K: 38 -> 64) = 12 instructions per wavefront for m*p (and 6 more for the half product
// m = T*p' mod 2^261).  This probe measures (a) the issue rate of that instruction and of v_mad_u64_u32 on a saturated SIMD and
// (b) the full field multiplication (171 multiplier instructions) for reference, so the two pipes can be priced in the same
// unit: SIMD cycles per wavefront field multiplication.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../gpu-acceleration_amd/csrc/fp_bn254.hpp"
using namespace bn254;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(int* out, int iters) {
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
    v16i c[NACC];
    for (int k = 0; k < NACC; k++) for (int j = 0; j < 16; j++) c[k][j] = k + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < NACC; k++) c[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[k], 0, 0, 0);
    }
    int s = 0;
    for (int k = 0; k < NACC; k++) for (int j = 0; j < 16; j++) s ^= c[k][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mad(uint32_t* out, int iters) {  // 8 independent accumulators, inline asm (the compiler folds a C loop)
    uint64_t x0 = threadIdx.x, x1 = 11, x2 = 22, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m = 12345u + threadIdx.x, n = 0x9E3779B9u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
                "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(m), "v"(n) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
}
__global__ void __launch_bounds__(256) k_fpmul(uint32_t* out, int iters) {
    fp v = fp_one(), m = fp_one();
    v.v[0] += threadIdx.x;
    m.v[1] += 7 + threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) v = fp_mul(v, m);
    }
    uint32_t s = 0;
    for (int k = 0; k < 9; k++) s ^= v.v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// VALU and MFMA together: every wavefront alternates 81 mads (the a*b half) with 18 MFMAs (m*p and the half product) -- do the two
// pipes really run side by side, i.e. is the pair as fast as the slower of the two?
__global__ void __launch_bounds__(256) k_both(uint32_t* out, int iters) {
    uint64_t x0 = threadIdx.x, x1 = 11, x2 = 22, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, x8 = 8;
    uint32_t m = 12345u + threadIdx.x, n = 0x9E3779B9u;
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
    v16i c[6];
    for (int k = 0; k < 6; k++) for (int j = 0; j < 16; j++) c[k][j] = k + j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int k = 0; k < 6; k++) c[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 3; k++)  // 27 mads per third: 81 per round
                asm volatile(
                    "v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %1, vcc, %9, %10, %1\n v_mad_u64_u32 %2, vcc, %9, %10, %2\n"
                    "v_mad_u64_u32 %3, vcc, %9, %10, %3\n v_mad_u64_u32 %4, vcc, %9, %10, %4\n v_mad_u64_u32 %5, vcc, %9, %10, %5\n"
                    "v_mad_u64_u32 %6, vcc, %9, %10, %6\n v_mad_u64_u32 %7, vcc, %9, %10, %7\n v_mad_u64_u32 %8, vcc, %9, %10, %8\n"
                    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8)
                    : "v"(m), "v"(n) : "vcc");
        }
    }
    int s = 0;
    for (int k = 0; k < 6; k++) for (int j = 0; j < 16; j++) s ^= c[k][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + x8);
}
template <typename F>
static float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate / 1e6;
    const int cus = prop.multiProcessorCount;
    int* d; CK(hipMalloc((void**)&d, 64 << 20));
    const unsigned blocks = cus * 4u;  // 4 wavefronts per SIMD (256-thread blocks: one wavefront on each SIMD of a CU)
    const int IT = 2000;
    auto cyc = [&](float ms, double per_wave_instr) { return ms * 1e-3 * ghz * 1e9 / per_wave_instr / 4.0; };  // 4 waves share a SIMD
    float m1 = time_ms([&] { k_mfma<1><<<blocks, 256>>>(d, IT); });
    float m4 = time_ms([&] { k_mfma<4><<<blocks, 256>>>(d, IT); });
    float md = time_ms([&] { k_mad<<<blocks, 256>>>((uint32_t*)d, IT); });  // 64 mads per iteration and wavefront
    float fm = time_ms([&] { k_fpmul<<<blocks, 256>>>((uint32_t*)d, IT / 4); });
    float bo = time_ms([&] { k_both<<<blocks, 256>>>((uint32_t*)d, IT / 4); });
    const double c_mfma1 = cyc(m1, IT), c_mfma4 = cyc(m4, IT * 4.0), c_mad = cyc(md, IT * 64.0), c_fp = cyc(fm, IT);
    printf("device %s, %d CUs, clock %.2f GHz (nominal), 4 wavefronts per SIMD\n", prop.name, cus, ghz);
    printf("v_mfma_i32_32x32x32_i8: %.1f SIMD cycles per instruction (1 accumulator chain), %.1f (4 independent accumulators)\n", c_mfma1, c_mfma4);
    printf("v_mad_u64_u32:          %.2f SIMD cycles per instruction\n", c_mad);
    printf("fp_mul (9x29 Montgomery, 171 multiplier instructions): %.0f SIMD cycles per wavefront multiplication\n", c_fp);
    const double best = c_mfma1 < c_mfma4 ? c_mfma1 : c_mfma4;
    printf("constant-operand work of one wavefront multiplication on the VALU: 90 instructions (81 m*p + 9 digit) = %.0f cycles\n", 90 * c_mad);
    printf("the same on the MFMA pipe: 12 (m*p) + 6 (m = T*p' mod R) instructions = %.0f cycles, BEFORE any operand marshalling\n", 18 * best);
    printf("VALU + MFMA issued from the same wavefronts (81 mads + 18 MFMAs per round): %.0f cycles per round (slower pipe alone: %.0f, sum %.0f)\n",
           cyc(bo, IT / 4.0), (81 * c_mad > 18 * best ? 81 * c_mad : 18 * best), 81 * c_mad + 18 * best);
    // fp_mul today = 171 multiplier instructions + shifts/masks/digit products; offloaded = the a*b half (81 mads) co-issued with the 18
    // MFMAs (measured above) + the same shifts/masks -- before a single instruction of operand marshalling (7-bit digit split of T,
    // the 32x32 accumulator tiles transposed back to one row per lane, 74 column sums carried into 29-bit limbs: >= 130 more
    // multiplier-instruction equivalents by count)
    const double other = c_fp - 171 * c_mad, offl = cyc(bo, IT / 4.0) + other;
    printf("=> fp_mul now %.0f cycles (171 mads = %.0f + %.0f of shifts/masks); with the constant half on the MFMA pipe and marshalling FREE: %.0f + %.0f = %.0f cycles = %.2fx (adoption bar: 1.25x)\n",
           c_fp, 171 * c_mad, other, cyc(bo, IT / 4.0), other, offl, c_fp / offl);
    return 0;
}
