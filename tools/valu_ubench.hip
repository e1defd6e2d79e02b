// Issue cost of the vector instructions the sweep kernels are made of, gfx950: 16 waves per CU (4 per SIMD), long unrolled chains of
// independent instructions; cycles per wave-instruction per SIMD.  Dev tool (DESIGN.md §7 "instruction issue").
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o tools/bin/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(1024, 4) void k(uint32_t *out, int iters) {
    uint32_t a[8];
    unsigned long long q[8];
    double f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = threadIdx.x * 2654435761u + u; q[u] = a[u] * 0x9E3779B97F4A7C15ull; f[u] = 1.0 + a[u] * 1e-9; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[u]) : "v"(a[u]), "v"(a[(u + 3) & 7]) : "vcc");
                if (MODE == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 4) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[u]) : "v"(a[u]));
                if (MODE == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[u]) : "v"(f[(u + 1) & 7]));
                if (MODE == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u + 1) & 7]));
                if (MODE == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 8) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 9) asm volatile("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %0, s20, 5" : "+v"(a[u]) : : "s20");
                if (MODE == 10) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(f[u]), "v"(f[(u + 1) & 7]) : "vcc");
                if (MODE == 11) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[u]));
                if (MODE == 12) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 13) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[u]) : "v"(q[(u + 1) & 7]));
                if (MODE == 14) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[u]) : "v"(f[(u + 1) & 7]));
                if (MODE == 15) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(f[u]) : "v"(a[u]));
                if (MODE == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 17) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 18) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[u]));
                if (MODE == 19) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(a[(u + 1) & 7]) : "vcc");
                if (MODE == 21) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[u]), "v"(a[(u + 1) & 7]) : "vcc");
                if (MODE == 22) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(a[(u + 1) & 7]), "v"(a[(u + 2) & 7]));
                if (MODE == 23) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 24) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(a[(u + 1) & 7]), "v"(a[(u + 2) & 7]));
                if (MODE == 26) asm volatile("v_mov_b32 %0, %1" : "=v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 27) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 28) asm volatile("v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 29) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 30) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
                if (MODE == 31) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(a[u]), "v"(a[(u + 1) & 7]) : "s20", "s21");
                if (MODE == 32) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[u]) : "v"(a[(u + 1) & 7]) : "s20", "s21");
            }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += a[u] + (uint32_t)q[u] + (uint32_t)f[u];
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
template <int MODE>
void run(const char *name, uint32_t *d, int iters, int per) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 1024>>>(d, 16);
    hipEventRecord(a);
    k<MODE><<<256, 1024>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)iters * 32 * per * 4; // wave-instructions per SIMD (4 waves each)
    printf("%-34s %8.3f ms  %6.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / instr);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000;
    run<0>("v_add_u32", d, iters, 1);
    run<7>("v_xor_b32", d, iters, 1);
    run<11>("v_bfe_u32", d, iters, 1);
    run<12>("v_mbcnt_lo_u32_b32", d, iters, 1);
    run<8>("v_mul_u32_u24", d, iters, 1);
    run<3>("v_mul_lo_u32", d, iters, 1);
    run<2>("v_mul_hi_u32", d, iters, 1);
    run<1>("v_mad_u64_u32", d, iters, 1);
    run<4>("v_lshrrev_b64", d, iters, 1);
    run<13>("v_lshl_add_u64", d, iters, 1);
    run<5>("v_fma_f64", d, iters, 1);
    run<6>("v_mul_f64", d, iters, 1);
    run<14>("v_add_f64", d, iters, 1);
    run<10>("v_cmp_lt_f64", d, iters, 1);
    run<15>("v_cvt_f64_u32", d, iters, 1);
    run<16>("v_and_b32", d, iters, 1);
    run<17>("v_or_b32", d, iters, 1);
    run<27>("v_sub_u32", d, iters, 1);
    run<30>("v_min_u32", d, iters, 1);
    run<26>("v_mov_b32", d, iters, 1);
    run<18>("v_lshlrev_b32 (imm)", d, iters, 1);
    run<19>("v_lshrrev_b32 (reg)", d, iters, 1);
    run<21>("v_cmp_eq_u32 -> vcc", d, iters, 1);
    run<31>("v_cmp_lt_u32 -> sgpr pair", d, iters, 1);
    run<20>("v_cndmask_b32 (vcc)", d, iters, 1);
    run<32>("v_cndmask_b32 (sgpr pair)", d, iters, 1);
    run<22>("v_add3_u32", d, iters, 1);
    run<23>("v_lshl_add_u32", d, iters, 1);
    run<24>("v_and_or_b32", d, iters, 1);
    run<28>("v_mbcnt_hi_u32_b32", d, iters, 1);
    run<29>("v_bcnt_u32_b32", d, iters, 1);
    run<9>("v_readlane + v_writelane (pair)", d, iters, 2);
    return 0;
}
