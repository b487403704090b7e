// tools/microbench3.hip -- lone-wavefront latency of the group law (what the bucket-reduction tree levels pay):
//   (1) dependent fp_mul chain, 1 / 2 / 4 independent chains per lane (is there ILP left for a single wave?)
//   (2) dependent xyzz_add chain per lane, 1 / 2 independent chains
//   (3) the same with 1, 2, 4 wavefronts per SIMD (workgroups of 64 on one CU are spread over its 4 SIMDs)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench3.hip -o tools/microbench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../gpu-acceleration_amd/csrc/ec_bn254.hpp"
using namespace bn254;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CH>
__global__ void __launch_bounds__(64) k_mul_chain(uint32_t* out, int iters) {
    fp x[CH], y = fp_one();
    y.v[0] += threadIdx.x;
    for (int c = 0; c < CH; c++) { x[c] = fp_one(); x[c].v[1] += c + threadIdx.x; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = fp_mul(x[c], y);
    }
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) for (int k = 0; k < 9; k++) s ^= x[c].v[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH>
__global__ void __launch_bounds__(64) k_add_chain(uint32_t* out, int iters) {
    xyzz a[CH], b;
    b.x = fp_one(); b.x.v[0] += 3 + threadIdx.x; b.y = fp_one(); b.y.v[0] += 5; b.zz = fp_one(); b.zzz = fp_one();
    for (int c = 0; c < CH; c++) { a[c] = b; a[c].x.v[1] += 7 + c; a[c].y.v[2] += 1; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) a[c] = xyzz_add(a[c], b);
    }
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) for (int k = 0; k < 9; k++) s ^= a[c].x.v[k] ^ a[c].y.v[k] ^ a[c].zz.v[k] ^ a[c].zzz.v[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename F>
static float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    uint32_t* d; CK(hipMalloc(&d, 1 << 22));
    const int IT = 2000, ITA = 200;
    for (int blocks : {1, 256, 1024, 2048, 4096}) {  // 256 CUs x 4 SIMDs: 1024 blocks of 64 = one wave per SIMD
        float m1 = time_ms([&] { k_mul_chain<1><<<blocks, 64>>>(d, IT); });
        float m2 = time_ms([&] { k_mul_chain<2><<<blocks, 64>>>(d, IT); });
        float m4 = time_ms([&] { k_mul_chain<4><<<blocks, 64>>>(d, IT); });
        float a1 = time_ms([&] { k_add_chain<1><<<blocks, 64>>>(d, ITA); });
        float a2 = time_ms([&] { k_add_chain<2><<<blocks, 64>>>(d, ITA); });
        printf("blocks %5d | fp_mul per-step latency: 1 chain %.3f us, 2 chains %.3f us, 4 chains %.3f us | xyzz_add: 1 chain %.2f us, 2 chains %.2f us\n",
               blocks, m1 * 1e3 / IT, m2 * 1e3 / IT, m4 * 1e3 / IT, a1 * 1e3 / ITA, a2 * 1e3 / ITA);
    }
    return 0;
}
