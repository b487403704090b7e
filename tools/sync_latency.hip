// How long after the last kernel of a stream does the host learn about it?  hipStreamSynchronize / hipEventSynchronize against a host
// that polls a word in PINNED memory which the kernel's last workgroup writes (system-scope store).  A spin kernel of ~T us stands in for
// the MSM; wall time of enqueue + wait, median of 200.   hipcc --offload-arch=gfx950 -O3 -o tools/sync_latency tools/sync_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(long long cycles, unsigned* counter, volatile unsigned* done_host, unsigned token) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(counter, 1u) == gridDim.x - 1) {  // the last workgroup
            *counter = 0;
            __threadfence_system();
            *done_host = token;
        }
    }
}
static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    unsigned *counter, *done_h, *done_d;
    hipMalloc(&counter, 4); hipMemset(counter, 0, 4);
    hipHostMalloc(&done_h, 64, hipHostMallocDefault); *done_h = 0;
    hipHostGetDevicePointer((void**)&done_d, done_h, 0);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
    for (double us : {20.0, 200.0, 1500.0}) {
        const long long cyc = (long long)(us * rate / 1000.0);
        std::vector<double> a, b, c;
        unsigned token = 1;
        for (int mode = 0; mode < 3; mode++)
            for (int it = 0; it < 220; it++) {
                auto t0 = std::chrono::steady_clock::now();
                k_spin<<<256, 256, 0, st>>>(cyc, counter, done_d, ++token);
                if (mode == 0) hipStreamSynchronize(st);
                else if (mode == 1) { hipEventRecord(ev, st); hipEventSynchronize(ev); }
                else { while (*(volatile unsigned*)done_h != token) {} }
                double ms = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (mode == 2) hipStreamSynchronize(st);
                if (it >= 20) (mode == 0 ? a : mode == 1 ? b : c).push_back(ms);
            }
        printf("kernel %.0f us: hipStreamSynchronize %.1f us, hipEventRecord+Synchronize %.1f us, pinned-word poll %.1f us (wall, launch to host wake, median of 200)\n",
               us, med(a), med(b), med(c));
    }
    return 0;
}
