// mb_valu: issue cost (cycles per wave64 instruction on one SIMD) of the VALU operations the kernels lean on.
// One wave per SIMD would be ideal; here 1 block of 64 threads per CU and a dependent-free unrolled body, timed with clock64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(x) x x x x x x x x
#define BODY(ASM) \
    for (int it = 0; it < iters; ++it) { REP8(REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)) }

template <int OP> __global__ void k(int iters, double seed, unsigned long long* out) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5, c = 0.25;
    float f0 = (float)seed; unsigned u0 = (unsigned)seed + threadIdx.x, u1 = 12345u;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)) }
        if (OP == 1) { REP8(REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0));)) }
        if (OP == 2) { REP8(REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)) }
        if (OP == 3) { REP8(REP8(asm volatile("v_min_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_max_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)) }
        if (OP == 4) { REP8(REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)) }
        if (OP == 5) { REP8(REP8(asm volatile("v_mul_hi_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2\n v_mul_hi_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(u0), "+v"(u1) : "v"(u1 | 1u));)) }
        if (OP == 6) { REP8(REP8(asm volatile("v_xor_b32 %0, %0, %2\n v_and_b32 %1, %1, %2\n v_xor_b32 %0, %0, %2\n v_or_b32 %1, %1, %2" : "+v"(u0), "+v"(u1) : "v"(u1 | 1u));)) }
        if (OP == 7) { REP8(REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(a0), "+v"(a1) : "v"(u0), "v"(u1) : "vcc");)) }
        if (OP == 8) { REP8(REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)) }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a0 + a1 + a2 + a3 + (double)u0 + (double)u1 == 1.2345e-300) out[0] = 0;
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 8 * 1024));
    const char* names[] = {"v_fma_f64", "v_cvt_f64_f32", "v_mul_f64", "v_min/max_f64", "v_add_f64", "v_mul_hi/lo_u32", "v_xor/and/or_b32", "v_mad_u64_u32", "v_rcp/rsq_f64"};
    const int iters = 200, per_iter = 64 * 4;
    printf("%-20s %s\n", "instruction", "cycles per wave64 instruction (1 wave per CU, clock64 ticks)");
#define RUN(OP) { hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64), 0, 0, iters, 1.5, d); CK(hipDeviceSynchronize()); hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64), 0, 0, iters, 1.5, d); CK(hipDeviceSynchronize()); \
      unsigned long long h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost)); double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i]; printf("%-20s %.2f\n", names[OP], s / 256 / ((double)iters * per_iter)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    return 0;
}
