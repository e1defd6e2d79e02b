// LDS throughput of random full-wave accesses on gfx950, 16 waves per CU (one 1024-thread workgroup holding 128 KB):
// cycles of LDS time per wave-instruction for the access kinds the cluster scan can be built from.  Dev tool.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_ubench.hip -o tools/bin/lds_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
extern __shared__ uint32_t lds[];
template <int MODE>
__global__ __launch_bounds__(1024, 4) void k(uint32_t *out, int iters, uint32_t words) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < words; i += 1024) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = tid * 747796405u + blockIdx.x * 2891336453u + 1u, acc = 0;
    const uint32_t mask = words - 1;
    for (int it = 0; it < iters; ++it) {
        uint32_t idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x = x * 1664525u + 1013904223u; idx[u] = (x >> 10) & mask; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) acc += lds[idx[u]];                                                    // ds_read_b32
            if (MODE == 1) acc += atomicOr(&lds[idx[u]], 0x01000000u);                            // ds_or_rtn_b32
            if (MODE == 2) acc += reinterpret_cast<uint16_t *>(lds)[idx[u] * 2 + (u & 1)];       // ds_read_u16
            if (MODE == 3) acc += reinterpret_cast<uint8_t *>(lds)[idx[u] * 4 + (u & 3)];        // ds_read_u8
            if (MODE == 4) reinterpret_cast<uint8_t *>(lds)[idx[u] * 4 + 2] = (uint8_t)x;        // ds_write_b8, all lanes
            if (MODE == 5) { if ((x >> 3) % 7 == 0) lds[idx[u]] = x; }                            // ds_write_b32, 1 lane in 7
            if (MODE == 6) atomicOr(&lds[idx[u]], 0x01000000u);                                  // ds_or_b32 (no return)
            if (MODE == 7) { const uint2 v = *reinterpret_cast<uint2 *>(&lds[idx[u] & ~1u]); acc += v.x ^ v.y; } // ds_read_b64
        }
        if (MODE == 8) { // dependent chain: latency under load
            uint32_t p = idx[0];
#pragma unroll
            for (int u = 0; u < 8; ++u) p = lds[p & mask];
            acc += p;
        }
    }
    out[blockIdx.x * 1024 + tid] = acc;
}
template <int MODE>
void run(const char *name, uint32_t *d, int iters, uint32_t words) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, words * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 1024, words * 4>>>(d, 64, words);
    hipEventRecord(a);
    k<MODE><<<256, 1024, words * 4>>>(d, iters, words);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)iters * 8 * 16; // wave-instructions per CU
    printf("%-28s %8.3f ms  %7.1f ns per wave-instruction per CU  (~%.1f cycles at 2.1 GHz)\n", name, ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.1);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 1024 * 4);
    const uint32_t words = 32768; // 128 KB: one workgroup per CU
    const int iters = 4096;
    run<0>("ds_read_b32 random", d, iters, words);
    run<1>("ds_or_rtn_b32 random", d, iters, words);
    run<6>("ds_or_b32 random", d, iters, words);
    run<2>("ds_read_u16 random", d, iters, words);
    run<3>("ds_read_u8 random", d, iters, words);
    run<7>("ds_read_b64 random", d, iters, words);
    run<4>("ds_write_b8 random", d, iters, words);
    run<5>("ds_write_b32 1/7 lanes", d, iters, words);
    run<8>("dependent ds_read_b32 chain", d, iters, words);
    return 0;
}
