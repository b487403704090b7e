// tools/glv_check.cpp -- host build of the DEVICE GLV split (csrc/glv_bn254.hpp is __host__ __device__): prints
// "ok k neg1 |k1| neg2 |k2|" (hex) for edge and random scalars below 2^254; tests/test_glv.py checks k1 + lambda*k2 = k (mod r)
// and the 127-bit bound with Python integers against tests/golden/glv_constants.json.
// Build: hipcc -O2 -std=c++17 -x hip --offload-arch=gfx950 tools/glv_check.cpp -o /tmp/glv_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <random>
#define FP_HD __host__ __device__ __forceinline__
#include "../gpu-acceleration_amd/csrc/glv_bn254.hpp"
int main(int argc, char** argv) {
    // prints k, s1, k1, s2, k2 as hex for Python to check
    std::mt19937_64 rng(7);
    for (int it = 0; it < 20000; it++) {
        uint32_t k[8];
        for (int i = 0; i < 8; i++) k[i] = (uint32_t)rng();
        k[7] &= 0x3FFFFFFFu;
        if (it == 0) for (int i = 0; i < 8; i++) k[i] = 0;
        if (it == 1) { for (int i = 0; i < 8; i++) k[i] = 0xFFFFFFFFu; k[7] = 0x3FFFFFFFu; }
        if (it == 2) { for (int i = 0; i < 8; i++) k[i] = 0; k[0] = 1; }
        uint32_t k1[4], k2[4]; bool n1, n2;
        bool ok = glv::split(k, k1, n1, k2, n2);
        printf("%d ", ok ? 1 : 0);
        for (int i = 7; i >= 0; i--) printf("%08x", k[i]);
        printf(" %d ", n1 ? 1 : 0);
        for (int i = 3; i >= 0; i--) printf("%08x", k1[i]);
        printf(" %d ", n2 ? 1 : 0);
        for (int i = 3; i >= 0; i--) printf("%08x", k2[i]);
        printf("\n");
    }
}
