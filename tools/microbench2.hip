// tools/microbench2.hip -- prototype + timing of the 9 x 29-bit carry-free Montgomery multiplication
// (mul29 below is the prototype that became bn254::fp_mul in csrc/fp_bn254.hpp)
// against the 8 x 32-bit CIOS of fp_bn254.hpp (see profiles/NOTES_r1.md: carries cost as much as multiplies
// on gfx950, so fewer carry instructions beat fewer multiplies).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fp_bn254_8x32.hpp"  // A/B baseline: 8 x 32-bit CIOS with carry chains
#include "../gpu-acceleration_amd/csrc/fp_bn254.hpp"       // the product field (this prototype, productised)
using bn254_8x32::fp;
using bn254_8x32::fp_mul;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct fq29 { uint32_t v[9]; };
constexpr uint32_t M29 = (1u << 29) - 1;
__device__ __forceinline__ uint32_t p29(int i) {
    constexpr uint32_t P[9] = {0x187cfd47, 0x10460b6, 0x1c72a34f, 0x2d522d0, 0x1585d978, 0x2db40c0, 0xa6e141, 0xe5c2634, 0x30644e};
    return P[i];
}
constexpr uint32_t INV29 = 0x4866389;
__device__ __forceinline__ uint64_t mad(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }

__device__ __forceinline__ fq29 mul29(const fq29& a, const fq29& b) {
    uint64_t acc = 0;
    uint32_t m[9];
    fq29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mad(a.v[i], b.v[k - i], acc);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mad(m[i], p29(k - i), acc);
        m[k] = ((uint32_t)acc * INV29) & M29;
        acc = mad(m[k], p29(0), acc);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc = mad(a.v[i], b.v[k - i], acc);
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc = mad(m[i], p29(k - i), acc);
        r.v[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
__device__ __forceinline__ fq29 add29(const fq29& a, const fq29& b) {
    fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
__device__ __forceinline__ fq29 norm29(const fq29& a) {
    fq29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t t = a.v[i] + c;
        r.v[i] = t & M29;
        c = t >> 29;
    }
    r.v[8] = a.v[8] + c;
    return r;
}

constexpr int FP_ITER = 256;
__global__ void k_mul32(uint32_t* out, const uint32_t* in) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    fp x, y;
    for (int k = 0; k < 8; k++) { x.v[k] = in[k] ^ (i * 2654435761u >> (k + 3)); y.v[k] = in[8 + k]; }
    x.v[7] &= 0x0FFFFFFFu; y.v[7] &= 0x0FFFFFFFu;
    for (int k = 0; k < FP_ITER; k++) { x = fp_mul(x, y); y = fp_mul(y, x); }
    out[i] = x.v[0] ^ y.v[3];
}
__global__ void k_mul29(uint32_t* out, const uint32_t* in) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    fq29 x, y;
    for (int k = 0; k < 9; k++) { x.v[k] = (in[k] ^ (i * 2654435761u >> (k + 3))) & M29; y.v[k] = in[8 + k] & M29; }
    x.v[8] &= 0xFFFFF; y.v[8] &= 0xFFFFF;
    for (int k = 0; k < FP_ITER; k++) { x = mul29(x, y); y = mul29(y, x); }
    out[i] = x.v[0] ^ y.v[3];
}
__global__ void k_mul29_addnorm(uint32_t* out, const uint32_t* in) {  // mul + lazy add + normalize per step
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    fq29 x, y;
    for (int k = 0; k < 9; k++) { x.v[k] = (in[k] ^ (i * 2654435761u >> (k + 3))) & M29; y.v[k] = in[8 + k] & M29; }
    x.v[8] &= 0xFFFFF; y.v[8] &= 0xFFFFF;
    for (int k = 0; k < FP_ITER; k++) { x = mul29(x, y); y = norm29(add29(mul29(y, x), x)); y.v[8] &= 0xFFFFF; }
    out[i] = x.v[0] ^ y.v[3];
}
// correctness probe: out = mul29(a, b) for host-supplied operands
__global__ void k_mul29_once(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    fq29 x, y;
    for (int k = 0; k < 9; k++) { x.v[k] = a[k]; y.v[k] = b[k]; }
    fq29 r = mul29(x, y);
    for (int k = 0; k < 9; k++) out[k] = r.v[k];
}

template <typename K, typename... A>
double time_kernel(K kern, dim3 g, dim3 b, int reps, A... args) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
typedef unsigned __int128 u128;
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t* out; CK(hipMalloc(&out, 64 << 20));
    uint32_t h_in[17] = {0x1234567, 0x89abcdef, 0x13579bdf, 0x2468ace0, 0x0f1e2d3c, 0x4b5a6978, 0x87969fa5, 0x01234567, 0x7654321,
                         0xfedcba98, 0xdb975310, 0x0eca8642, 0xc3d2e1f0, 0x8796a5b4, 0x5af96978, 0x07654321, 0x00012345};
    uint32_t* in; CK(hipMalloc(&in, 68)); CK(hipMemcpy(in, h_in, 68, hipMemcpyHostToDevice));
    // correctness of mul29 against a host big-integer evaluation: r * 2^261 == a*b (mod p), r < 2p-ish
    {
        uint32_t ha[9], hb[9], hr[9];
        for (int k = 0; k < 9; k++) { ha[k] = (h_in[k] * 2654435761u) & M29; hb[k] = (h_in[8 + k] * 40503u + k) & M29; }
        ha[8] &= 0x3FFFFF; hb[8] &= 0x3FFFFF;
        uint32_t *da, *db, *dr; CK(hipMalloc(&da, 36)); CK(hipMalloc(&db, 36)); CK(hipMalloc(&dr, 36));
        CK(hipMemcpy(da, ha, 36, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 36, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mul29_once, dim3(1), dim3(1), 0, 0, da, db, dr);
        CK(hipMemcpy(hr, dr, 36, hipMemcpyDeviceToHost));
        printf("mul29 probe a="); for (int k = 8; k >= 0; k--) printf("%08x ", ha[k]);
        printf("\n            b="); for (int k = 8; k >= 0; k--) printf("%08x ", hb[k]);
        printf("\n            r="); for (int k = 8; k >= 0; k--) printf("%08x ", hr[k]);
        printf("\n");
    }
    for (int wpc : {4, 8, 16}) {
        dim3 g(cus * wpc / 4), b(256);
        double ms = time_kernel(k_mul32, g, b, 3, out, (const uint32_t*)in);
        printf("waves/CU %2d  fp_mul 8x32      %8.3f ms  %8.2f G modmul/s  (%.0f cycles per wave-modmul per SIMD)\n", wpc, ms,
               (double)g.x * 256 * FP_ITER * 2 / ms / 1e6, (ms * 1e-3 * 2.4e9) / ((double)FP_ITER * 2 * wpc / 4));
        ms = time_kernel(k_mul29, g, b, 3, out, (const uint32_t*)in);
        printf("waves/CU %2d  mul29 9x29       %8.3f ms  %8.2f G modmul/s  (%.0f cycles per wave-modmul per SIMD)\n", wpc, ms,
               (double)g.x * 256 * FP_ITER * 2 / ms / 1e6, (ms * 1e-3 * 2.4e9) / ((double)FP_ITER * 2 * wpc / 4));
        ms = time_kernel(k_mul29_addnorm, g, b, 3, out, (const uint32_t*)in);
        printf("waves/CU %2d  mul29+add+norm   %8.3f ms  %8.2f G modmul/s  (%.0f cycles per wave-modmul per SIMD)\n", wpc, ms,
               (double)g.x * 256 * FP_ITER * 2 / ms / 1e6, (ms * 1e-3 * 2.4e9) / ((double)FP_ITER * 2 * wpc / 4));
    }
    return 0;
}
