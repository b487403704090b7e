// tools/calib_gather.hip -- calibration of the FETCH_SIZE / WRITE_SIZE counters for THIS repo's access patterns
// (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950; other widths uncalibrated).
//   k_stream : 16 B per lane coalesced read of B bytes              (known bytes = B)
//   k_gather : 64-byte records at random indices, 4 x dwordx4/lane  (known bytes = records * 64), table >> 256 MiB
//   k_store36: 144-byte records written as 9 x dwordx4 per lane     (known bytes = records * 144)
// Run each under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and divide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_stream(const uint4* in, uint32_t* out, size_t n16) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_gather(const uint4* table, uint32_t* out, uint32_t nrec, uint32_t table_rec_mask) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    uint32_t idx = (i * 2654435761u) & table_rec_mask;  // pseudo-random permutation-ish
    const uint4* p = table + (size_t)idx * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    uint32_t acc = a.x ^ b.y ^ c.z ^ d.w;
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_store36(uint4* outp, uint32_t nrec) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    uint4* p = outp + (size_t)i * 9;
    for (int k = 0; k < 9; k++) p[k] = make_uint4(i, k, i ^ k, 7);
}
int main() {
    const size_t bytes = 1ull << 30;
    uint4* buf; uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    k_stream<<<2048, 256>>>(buf, out, bytes / 16);
    const uint32_t nrec = 1u << 22;  // 4 Mi records * 64 B = 256 MiB gathered from a 1 GiB table (16 Mi records)
    k_gather<<<nrec / 256, 256>>>(buf, out, nrec, (uint32_t)(bytes / 64 - 1));
    const uint32_t nst = 1u << 21;   // 2 Mi records * 144 B = 288 MiB written
    k_store36<<<nst / 256, 256>>>(buf, nst);
    (void)hipDeviceSynchronize();
    printf("known bytes: k_stream %zu, k_gather %zu, k_store36 %zu\n", bytes, (size_t)nrec * 64, (size_t)nst * 144);
    return 0;
}
