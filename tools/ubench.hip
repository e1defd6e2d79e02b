// Dev tool: issue cost (shader cycles per wave-instruction) of the vector instructions the SSE kernels lean on, gfx950.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(NAME, ASM)                                                                                         \
    __global__ void k_##NAME(unsigned long long *out, int iters) {                                              \
        unsigned a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x9E3779B9u, c = b * 3u + 1u, d = c ^ a;       \
        double x = (double)a * 1e-3, y = (double)b * 1e-4;                                                      \
        unsigned long long q = ((unsigned long long)a << 32) | b;                                               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                   \
        for (int i = 0; i < iters; ++i) { REP16(ASM) }                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                   \
        if (a == 0xFFFFFFFFu && b == 7u && c == 9u && d == 11u && x == 1.5 && y == 2.5 && q == 3ull) out[1] = 1; \
        if ((threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);                                                    \
    }

// four independent chains (a,b,c,d) so that dependent-issue latency does not dominate
BODY(xor, asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BODY(mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BODY(mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BODY(mad_u64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %0, vcc, %3, %1, 0\n v_mad_u64_u32 %0, vcc, %1, %1, 0" : "+v"(q), "+v"(a), "+v"(b), "+v"(c) : : "vcc");)
BODY(mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BODY(mul_f64, asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %0\n v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %0" : "+v"(x), "+v"(y));)
BODY(fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %0, %0\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %0, %0" : "+v"(x), "+v"(y));)
BODY(cvt_f64_u32, asm volatile("v_cvt_f64_u32 %0, %2\n v_cvt_f64_u32 %1, %3\n v_cvt_f64_u32 %0, %3\n v_cvt_f64_u32 %1, %2" : "+v"(x), "+v"(y), "+v"(a), "+v"(b));)
BODY(cmp_f64, asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0" : "+v"(x), "+v"(y) : : "vcc");)
BODY(cmp_u32, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0" : "+v"(a), "+v"(b) : : "vcc");)
BODY(cmp_u64, asm volatile("v_cmp_lt_u64 vcc, %0, %0\n v_cmp_lt_u64 vcc, %0, %0\n v_cmp_lt_u64 vcc, %0, %0\n v_cmp_lt_u64 vcc, %0, %0" : "+v"(q) : : "vcc");)
BODY(cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");)
BODY(mbcnt, asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %1, %2, %1\n v_mbcnt_lo_u32_b32 %2, %3, %2\n v_mbcnt_hi_u32_b32 %3, %0, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BODY(readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21", "s22", "s23");)
BODY(xor3, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BODY(salu, asm volatile("s_add_u32 s20, s20, 1\n s_xor_b32 s21, s21, s20\n s_and_b64 s[22:23], s[22:23], vcc\n s_bcnt1_i32_b64 s24, s[22:23]" : : : "s20", "s21", "s22", "s23", "s24", "scc", "vcc");)

typedef void (*kern_t)(unsigned long long *, int);
struct Ent { const char *name; kern_t k; };

int main() {
    Ent ents[] = {{"v_xor_b32", k_xor}, {"v_bitop3_b32", k_xor3}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u64_u32", k_mad_u64},
                  {"v_mul_u32_u24", k_mul_u24}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64}, {"v_cvt_f64_u32", k_cvt_f64_u32},
                  {"v_cmp_lt_f64", k_cmp_f64}, {"v_cmp_lt_u32", k_cmp_u32}, {"v_cmp_lt_u64", k_cmp_u64}, {"v_cndmask_b32", k_cndmask},
                  {"v_mbcnt", k_mbcnt}, {"v_readlane_b32", k_readlane}, {"salu(4 mixed)", k_salu}};
    unsigned long long *d;
    hipMalloc(&d, 16);
    const int iters = 2000;
    printf("%-16s %10s %10s %10s   (shader cycles per wave-instruction; waves per SIMD = 1, 2, 4)\n", "instr", "1w", "2w", "4w");
    for (auto &e : ents) {
        double res[3];
        int wi = 0;
        for (int threads : {256, 512, 1024}) { // one workgroup per CU of 4 SIMDs -> 1, 2, 4 waves per SIMD
            hipMemset(d, 0, 16);
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, d, iters);
            hipDeviceSynchronize();
            unsigned long long h[2];
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            const int waves_per_simd = threads / 256;
            // cycles per instruction issued by ONE wave while `waves_per_simd` share the SIMD; throughput per SIMD = this / waves
            res[wi++] = (double)h[0] / (iters * 16.0 * 4.0) / waves_per_simd;
        }
        printf("%-16s %10.2f %10.2f %10.2f\n", e.name, res[0], res[1], res[2]);
    }
    return 0;
}
