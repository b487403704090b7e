// tools/microbench4.hip -- issue rates of the 64-bit helpers hipcc puts around every Montgomery column
// (v_lshrrev_b64 for acc >> 29, v_lshl_add_u64 for the 64-bit carry-in add) against their 32-bit replacements.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench4.hip -o tools/microbench4
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, int iters) {
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + 12345, b = a ^ 0x5555aaaa5555aaaaull;
    uint32_t x = (uint32_t)a, y = (uint32_t)(a >> 32), z = x ^ y;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP64(asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(a));) }
        if (OP == 1) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 2) { REP64(asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(x) : "v"(y));) }
        if (OP == 3) { REP64(asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(x));) }
        if (OP == 4) { REP64(asm volatile("v_mad_u64_u32 %0, s[0:1], %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "s0", "s1");) }
        if (OP == 5) { REP64(asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(y));) }
        if (OP == 6) { REP64(asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(z));) }
        if (OP == 7) { REP64(asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x) : "v"(y));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + x + b;
}
template <int OP>
static double run(uint64_t* d, const char* name, double ref) {
    const int iters = 400, blocks = 1024 * 4;  // 4 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * 64;   // wave-instructions
    double per = ms * 1e-3 * 2.4e9 * 1024 / wave_instr;    // cycles (at 2.4 GHz) per wave-instruction per SIMD
    printf("%-16s %.2f cycles per wave-instruction per SIMD%s\n", name, per, ref > 0 ? "" : "");
    return per;
}
int main() {
    uint64_t* d; CK(hipMalloc(&d, 1 << 24));
    run<4>(d, "v_mad_u64_u32", 0); run<7>(d, "v_mul_lo_u32", 0);
    run<0>(d, "v_lshrrev_b64", 0); run<1>(d, "v_lshl_add_u64", 0);
    run<2>(d, "v_alignbit_b32", 0); run<3>(d, "v_lshrrev_b32", 0); run<5>(d, "v_add_u32", 0); run<6>(d, "v_and_b32", 0);
    return 0;
}
