// tools/microbench.hip -- instruction-rate calibration for the MSM kernels on gfx950.
// Measures what the accumulate kernel is actually bound by: v_mad_u64_u32 (the 32x32+64 limb product),
// carry-chain adds, and for comparison v_fma_f64 / v_mul_lo_u32 / v_mul_hi_u32; then the achieved
// Montgomery-multiplication and mixed-add rates of fp_bn254.hpp / ec_bn254.hpp.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fp_bn254_8x32.hpp"  // the r1a field these numbers were taken with
#include "../gpu-acceleration_amd/csrc/ec_bn254.hpp"
using namespace bn254;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 512;

__global__ void k_mad64(uint32_t* out, uint32_t a, uint32_t b) {
    uint64_t x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m = a + threadIdx.x, n = b;
    for (int i = 0; i < ITER; i++) {
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
__global__ void k_mullo(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m = a + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n"
            "v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %8, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %8, %7\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_mulhi(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m = a + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mul_hi_u32 %0, %8, %0\n v_mul_hi_u32 %1, %8, %1\n v_mul_hi_u32 %2, %8, %2\n v_mul_hi_u32 %3, %8, %3\n"
            "v_mul_hi_u32 %4, %8, %4\n v_mul_hi_u32 %5, %8, %5\n v_mul_hi_u32 %6, %8, %6\n v_mul_hi_u32 %7, %8, %7\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_mul24(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m = a + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mad_u32_u24 %0, %8, %0, %1\n v_mad_u32_u24 %1, %8, %1, %2\n v_mad_u32_u24 %2, %8, %2, %3\n v_mad_u32_u24 %3, %8, %3, %4\n"
            "v_mad_u32_u24 %4, %8, %4, %5\n v_mad_u32_u24 %5, %8, %5, %6\n v_mad_u32_u24 %6, %8, %6, %7\n v_mad_u32_u24 %7, %8, %7, %0\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_addc(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m = a + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_add_co_u32 %0, vcc, %8, %0\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n v_addc_co_u32 %2, vcc, %8, %2, vcc\n v_addc_co_u32 %3, vcc, %8, %3, vcc\n"
            "v_addc_co_u32 %4, vcc, %8, %4, vcc\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n v_addc_co_u32 %6, vcc, %8, %6, vcc\n v_addc_co_u32 %7, vcc, %8, %7, vcc\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_addc_nop(uint32_t* out, uint32_t a, uint32_t b) {  // as hipcc schedules it: s_nop between dependent carries
    uint32_t x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m = a + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_add_co_u32 %0, vcc, %8, %0\n s_nop 0\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n s_nop 0\n v_addc_co_u32 %2, vcc, %8, %2, vcc\n s_nop 0\n v_addc_co_u32 %3, vcc, %8, %3, vcc\n s_nop 0\n"
            "v_addc_co_u32 %4, vcc, %8, %4, vcc\n s_nop 0\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n s_nop 0\n v_addc_co_u32 %6, vcc, %8, %6, vcc\n s_nop 0\n v_addc_co_u32 %7, vcc, %8, %7, vcc\n s_nop 0\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_add32(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m = a + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
            "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_fma64(uint32_t* out, uint32_t a, uint32_t b) {
    double x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    double m = 1.0000001, n = 1e-9;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %1, %8, %1, %9\n v_fma_f64 %2, %8, %2, %9\n v_fma_f64 %3, %8, %3, %9\n"
            "v_fma_f64 %4, %8, %4, %9\n v_fma_f64 %5, %8, %5, %9\n v_fma_f64 %6, %8, %6, %9\n v_fma_f64 %7, %8, %7, %9\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(n));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
}
__global__ void k_lshladd64(uint32_t* out, uint32_t a, uint32_t b) {
    uint64_t x0 = threadIdx.x, x1 = a, x2 = b, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint64_t m = a + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
            "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
}

constexpr int FP_ITER = 256;
__global__ void k_fpmul(uint32_t* out, const uint32_t* in) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bn254_8x32::fp x, y;
    for (int k = 0; k < 8; k++) { x.v[k] = in[k] ^ (i * 2654435761u >> (k + 3)); y.v[k] = in[8 + k]; }
    x.v[7] &= 0x0FFFFFFFu; y.v[7] &= 0x0FFFFFFFu;
    for (int k = 0; k < FP_ITER; k++) { x = bn254_8x32::fp_mul(x, y); y = bn254_8x32::fp_mul(y, x); }
    out[i] = x.v[0] ^ y.v[3];
}
__global__ void k_madd(uint32_t* out, const uint32_t* in) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    affine q;
    for (int k = 0; k < 9; k++) { q.x.v[k] = in[k] & FP_MASK; q.y.v[k] = in[8 + k] & FP_MASK; }
    q.x.v[8] &= 0xFFFFF; q.y.v[8] &= 0xFFFFF;
    xyzz acc = xyzz_from_affine(q);
    acc.x.v[0] ^= (i & 0xff);  // not on the curve: fine for timing, the formulas are branch-free in the common case
    for (int k = 0; k < FP_ITER / 4; k++) xyzz_madd(acc, q);
    out[i] = acc.x.v[0] ^ acc.zzz.v[3];
}

template <typename K, typename... A>
double time_kernel(K kern, dim3 g, dim3 b, int reps, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    uint32_t* out; CK(hipMalloc(&out, 64 << 20));
    uint32_t h_in[16] = {0x1234567, 0x89abcdef, 0x13579bdf, 0x2468ace0, 0x0f1e2d3c, 0x4b5a6978, 0x87969fa5, 0x01234567,
                         0x7654321, 0xfedcba98, 0xdb975310, 0x0eca8642, 0xc3d2e1f0, 0x8796a5b4, 0x5af96978, 0x07654321};
    uint32_t* in; CK(hipMalloc(&in, 64)); CK(hipMemcpy(in, h_in, 64, hipMemcpyHostToDevice));
    struct { const char* name; void (*k)(uint32_t*, uint32_t, uint32_t); } raw[] = {
        {"v_mad_u64_u32", k_mad64}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mad_u32_u24", k_mul24},
        {"v_addc_co_u32 chain", k_addc}, {"v_addc_co_u32 + s_nop", k_addc_nop}, {"v_add_u32", k_add32},
        {"v_fma_f64", k_fma64}, {"v_lshl_add_u64", k_lshladd64}};
    for (int wpc : {4, 8, 16, 32}) {  // waves per CU
        dim3 g(cus * wpc / 4), b(256);
        for (auto& r : raw) {
            double ms = time_kernel(r.k, g, b, 5, out, 12345u, 67890u);
            double ops = (double)g.x * 256 * ITER * 8;
            printf("waves/CU %2d  %-24s %8.3f ms  %8.2f Gops/s/CU-lane-agg %7.2f Tops/s  cycles/wave-instr/SIMD @2.4GHz %.2f\n", wpc, r.name, ms,
                   ops / ms / 1e6 / cus, ops / ms / 1e9, (ms * 1e-3 * 2.4e9) / ((double)ITER * 8 * wpc / 4));
        }
    }
    for (int wpc : {4, 8, 16}) {
        dim3 g(cus * wpc / 4), b(256);
        double ms = time_kernel(k_fpmul, g, b, 3, out, (const uint32_t*)in);
        double muls = (double)g.x * 256 * FP_ITER * 2;
        printf("waves/CU %2d  fp_mul   %8.3f ms  %8.2f G modmul/s   (%.0f cycles per wave-modmul per SIMD)\n", wpc, ms, muls / ms / 1e6,
               (ms * 1e-3 * 2.4e9) / ((double)FP_ITER * 2 * wpc / 4));
        ms = time_kernel(k_madd, g, b, 3, out, (const uint32_t*)in);
        double adds = (double)g.x * 256 * (FP_ITER / 4);
        printf("waves/CU %2d  xyzz_madd %8.3f ms  %8.2f G madd/s\n", wpc, ms, adds / ms / 1e6);
    }
    // lone-wave latency
    {
        double ms = time_kernel(k_fpmul, dim3(1), dim3(64), 3, out, (const uint32_t*)in);
        printf("lone wave: fp_mul latency %.3f us\n", ms * 1e3 / (FP_ITER * 2));
        ms = time_kernel(k_madd, dim3(1), dim3(64), 3, out, (const uint32_t*)in);
        printf("lone wave: xyzz_madd latency %.3f us\n", ms * 1e3 / (FP_ITER / 4));
    }
    return 0;
}
