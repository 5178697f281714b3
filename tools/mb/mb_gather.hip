// Microbenchmark: how fast can a CU gather 128-byte node pairs at per-lane random indices?
//  mode 0: each lane loads its own 128 B with 8 x global_load_dwordx4 (what k_extend_bvh2 does)
//  mode 1: cooperative: 8 lanes fetch one pair as a contiguous 128 B (8 loads per wave cover 64 pairs), staged through LDS
//          (ds_write_b128, XOR-swizzled) and read back by the owning lane (8 x ds_read_b128)
//  mode 2: cooperative with global_load_lds_dwordx4 (LDS-DMA, swizzle applied on the source side)
// The next index depends on the loaded data (dependent chain like a traversal).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct __align__(16) Pair { double v[12]; uint32_t a, b, c, d; uint32_t pad[4]; };   // 128 B
static_assert(sizeof(Pair) == 128, "");

template <int MODE>
__global__ __launch_bounds__(128) void k(const Pair* __restrict__ t, uint32_t n_pairs, int iters, double* out) {
    __shared__ __align__(16) char lds[2][64 * 128];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t idx = (blockIdx.x * 128 + threadIdx.x) * 2654435761u % n_pairs;
    double acc = 0;
    char* base = lds[wv];
    for (int it = 0; it < iters; ++it) {
        double v[12]; uint32_t a, b;
        if (MODE == 0) {
            const Pair p = t[idx];
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] = p.v[i];
            a = p.a; b = p.b;
        } else {
            // instruction k: lane l fetches 16-byte piece ((l&7) ^ owner&7) of the pair owned by lane 8k + (l>>3)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const uint32_t owner = kk * 8 + (lane >> 3);
                const uint32_t oidx = __shfl(idx, owner, 64);
                const uint32_t piece = (lane & 7) ^ (owner & 7);
                const char* src = (const char*)(t + oidx) + piece * 16;
                char* dst = base + kk * 1024 + lane * 16;         // == owner*128 + (lane&7)*16
                if (MODE == 1) {
                    *(uint4*)dst = *(const uint4*)src;
                } else {
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(base + kk * 1024), 16, 0, 0);
                }
            }
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // owner reads its 8 pieces: piece i sits in slot i ^ (lane&7)
            uint4 q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = *(const uint4*)(base + lane * 128 + ((i ^ (lane & 7)) * 16));
            __builtin_amdgcn_wave_barrier();
            const double* dq = (const double*)q;
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] = dq[i];
            a = q[6].x; b = q[6].y;
        }
        // ~ the arithmetic of two slab tests
        double s = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) s = fmax(s, (v[i] - acc) * 0.999);
        acc = s * 1e-9;
        idx = (a + (s > 1e300 ? b : 0u)) % n_pairs;
    }
    out[blockIdx.x * 128 + threadIdx.x] = acc + idx;
}

int main(int argc, char** argv) {
    const int iters = 400;
    for (uint32_t n_pairs : {11000u, 730000u}) {
        std::vector<Pair> h(n_pairs);
        uint32_t s = 12345;
        for (auto& p : h) { for (double& x : p.v) x = (s = s * 1664525u + 1013904223u) * 1e-9; p.a = (s = s * 1664525u + 1013904223u) % n_pairs; p.b = 1; p.c = p.d = 0; }
        Pair* d; double* out; const int blocks = 2048;
        CHECK(hipMalloc(&d, n_pairs * sizeof(Pair))); CHECK(hipMalloc(&out, blocks * 128 * 8));
        CHECK(hipMemcpy(d, h.data(), n_pairs * sizeof(Pair), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int mode = 0; mode < 3; ++mode) for (int bl : {2048, 1024}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            double visits = (double)bl * 128 * iters;
            printf("table %7.1f MB mode %d blocks %4d: %8.3f ms  %7.2f Gvisits/s  %7.2f TB/s\n", n_pairs * 128e-6, mode, bl, best, visits / best / 1e6, visits * 128 / best / 1e9);
        }
        CHECK(hipFree(d)); CHECK(hipFree(out));
    }
    return 0;
}
