// can the arkworks struct path ship 64 B per point?  72-byte G1Affine structs (x 32 | y 32 | infinity 1 + 7 pad) in PINNED host
// memory -> HBM: (a) linear copy of the whole structs (what msm_bn254_g1_arkworks does), (b) hipMemcpy2DAsync of the 64 coordinate bytes of
// every struct (pitch 72 -> 64), (c) hipMemcpy2DAsync of the infinity byte of every struct (pitch 72 -> 1), (d) linear copy of packed 64-byte records.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/memcpy2d_probe tools/memcpy2d_probe.hip      run: tools/memcpy2d_probe [log_n]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const size_t n = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 20);
    unsigned char *h, *d, *dinf;
    CK(hipHostMalloc((void**)&h, n * 72, hipHostMallocDefault));
    CK(hipMalloc((void**)&d, n * 72));
    CK(hipMalloc((void**)&dinf, n));
    for (size_t i = 0; i < n * 72; i += 4096) h[i] = (unsigned char)i;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run = [&](const char* name, auto&& f, double bytes) {
        double best = 1e9;
        for (int r = 0; r < 6; r++) {
            auto t0 = std::chrono::steady_clock::now();
            f();
            (void)hipStreamSynchronize(s);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (r && ms < best) best = ms;
        }
        printf("%-46s %8.3f ms  %7.1f GB/s of useful bytes\n", name, best, bytes / best / 1e6);
    };
    run("linear, whole 72-byte structs", [&] { (void)hipMemcpyAsync(d, h, n * 72, hipMemcpyHostToDevice, s); }, n * 72.0);
    run("2D, 64 coordinate bytes per struct (72 -> 64)", [&] { (void)hipMemcpy2DAsync(d, 64, h, 72, 64, n, hipMemcpyHostToDevice, s); }, n * 64.0);
    run("2D, the infinity byte per struct (72 -> 1)", [&] { (void)hipMemcpy2DAsync(dinf, 1, h + 64, 72, 1, n, hipMemcpyHostToDevice, s); }, n * 1.0);
    run("linear, packed 64-byte records", [&] { (void)hipMemcpyAsync(d, h, n * 64, hipMemcpyHostToDevice, s); }, n * 64.0);
    return 0;
}
