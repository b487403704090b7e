// tools/overlap_probe.hip -- does a host->device copy overlap a running kernel on this box, and how?  (round 2: the streamed
// host path ran SLOWER from pinned caller memory than from pageable memory.)
// Build: hipcc -O2 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void busy(uint32_t* out, uint32_t iters) {  // ALU-bound, fills the chip like k_accumulate (256 threads, many blocks)
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + blockIdx.x;
    uint32_t y = (uint32_t)(a >> 32) | 1u;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) a = (uint64_t)(uint32_t)a * y + a;
    }
    if ((uint32_t)a == 0x12345u) out[0] = (uint32_t)a;
}
__global__ void pull(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {  // zero-copy read of (pinned) host memory
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += st) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = (size_t)96 << 20;
    void *hp, *hpage = malloc(bytes), *d, *d2;
    CK(hipHostMalloc(&hp, bytes, hipHostMallocDefault));
    memset(hp, 1, bytes); memset(hpage, 2, bytes);
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&d2, bytes));
    uint32_t* dout; CK(hipMalloc((void**)&dout, 64));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("HSA_ENABLE_SDMA=%s\n", getenv("HSA_ENABLE_SDMA") ? getenv("HSA_ENABLE_SDMA") : "(unset)");
    const uint32_t iters = 700;  // ~2 ms
    const unsigned blocks = 256 * 4 * 12;
    auto run_busy = [&](hipStream_t s) { busy<<<blocks, 256, 0, s>>>(dout, iters); };
    void* hdev; CK(hipHostGetDevicePointer(&hdev, hp, 0));
    for (int rep = 0; rep < 2; rep++) {
        double t;
        t = now(); run_busy(sa); CK(hipStreamSynchronize(sa)); double tk = now() - t;
        t = now(); CK(hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, sb)); CK(hipStreamSynchronize(sb)); double tc = now() - t;
        t = now(); CK(hipMemcpyAsync(d, hpage, bytes, hipMemcpyHostToDevice, sb)); CK(hipStreamSynchronize(sb)); double tp = now() - t;
        t = now(); pull<<<2048, 256, 0, sb>>>((const uint4*)hdev, (uint4*)d2, bytes / 16); CK(hipStreamSynchronize(sb)); double tz = now() - t;
        t = now(); pull<<<256, 256, 0, sb>>>((const uint4*)hdev, (uint4*)d2, bytes / 16); CK(hipStreamSynchronize(sb)); double tz2 = now() - t;
        // concurrently: kernel first, then the copy
        t = now(); run_busy(sa); CK(hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, sb)); CK(hipStreamSynchronize(sb)); double tcb = now() - t; CK(hipStreamSynchronize(sa)); double t1 = now() - t;
        t = now(); CK(hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, sb)); run_busy(sa); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb)); double t2 = now() - t;
        t = now(); run_busy(sa); CK(hipMemcpyAsync(d, hpage, bytes, hipMemcpyHostToDevice, sb)); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb)); double t3 = now() - t;
        t = now(); run_busy(sa); pull<<<2048, 256, 0, sb>>>((const uint4*)hdev, (uint4*)d2, bytes / 16); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb)); double t4 = now() - t;
        t = now(); run_busy(sa); pull<<<256, 256, 0, sb>>>((const uint4*)hdev, (uint4*)d2, bytes / 16); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb)); double t4b = now() - t;
        // 4 pieces with an event chain like the streamed path: copy piece j on sb, kernel j on sa waits for it
        hipEvent_t ev[4]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        t = now();
        for (int j = 0; j < 4; j++) {
            CK(hipMemcpyAsync((char*)d + j * (bytes / 4), (char*)hp + j * (bytes / 4), bytes / 4, hipMemcpyHostToDevice, sb));
            CK(hipEventRecord(ev[j], sb));
            CK(hipStreamWaitEvent(sa, ev[j], 0));
            busy<<<blocks, 256, 0, sa>>>(dout, iters / 4);
        }
        CK(hipStreamSynchronize(sa)); double t5 = now() - t;
        t = now();
        for (int j = 0; j < 4; j++) {
            CK(hipMemcpyAsync((char*)d + j * (bytes / 4), (char*)hpage + j * (bytes / 4), bytes / 4, hipMemcpyHostToDevice, sb));
            CK(hipEventRecord(ev[j], sb));
            CK(hipStreamWaitEvent(sa, ev[j], 0));
            busy<<<blocks, 256, 0, sa>>>(dout, iters / 4);
        }
        CK(hipStreamSynchronize(sa)); double t6 = now() - t;
        t = now();
        for (int j = 0; j < 4; j++) {
            pull<<<2048, 256, 0, sb>>>((const uint4*)((char*)hdev + j * (bytes / 4)), (uint4*)((char*)d2 + j * (bytes / 4)), bytes / 64);
            CK(hipEventRecord(ev[j], sb));
            CK(hipStreamWaitEvent(sa, ev[j], 0));
            busy<<<blocks, 256, 0, sa>>>(dout, iters / 4);
        }
        CK(hipStreamSynchronize(sa)); double t7 = now() - t;
        if (rep == 1)
            printf("kernel alone %.3f | copy pinned %.3f (%.1f GB/s) pageable %.3f | zero-copy pull 2048 blocks %.3f (%.1f GB/s) 256 blocks %.3f\n"
                   "kernel+pinned copy: copy done %.3f all %.3f | copy-then-kernel %.3f | kernel+pageable %.3f | kernel+pull2048 %.3f kernel+pull256 %.3f\n"
                   "4-piece chain: pinned %.3f pageable %.3f pull %.3f  (ideal ~ max(copy, kernel) + 1/4)\n",
                   tk, tc, bytes / tc / 1e6, tp, tz, bytes / tz / 1e6, tz2, tcb, t1, t2, t3, t4, t4b, t5, t6, t7);
    }
    return 0;
}
