// Microbenchmark: how fast can a CU gather 128-byte node pairs at per-lane random indices?
//  mode 0: each lane loads its own 128 B with 8 x global_load_dwordx4 (what k_extend_bvh2 does)
//  mode 1: cooperative: 8 lanes fetch one pair as a contiguous 128 B (8 loads per wave cover 64 pairs), staged through LDS
//          (ds_write_b128, XOR-swizzled) and read back by the owning lane (8 x ds_read_b128)
//  mode 2: cooperative with global_load_lds_dwordx4 (LDS-DMA, swizzle applied on the source side)
//  mode 8 / 9: mode 3 with raw buffer loads (V# + 32-bit offsets), 64 / 32 strided live lanes
//  mode 3: each lane loads a 64-byte pair (12 x f32 box planes + 4 x u32) with 4 x global_load_dwordx4 and widens to f64
// The next index depends on the loaded data (dependent chain like a traversal).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct __align__(16) Pair { double v[12]; uint32_t a, b, c, d; uint32_t pad[4]; };   // 128 B
static_assert(sizeof(Pair) == 128, "");

//  LANES_ON < 64: only lanes with (lane % 64) < LANES_ON (mode 10+: every (64/LANES_ON)-th lane) take part: does a masked lane cost?
template <int MODE, int LANES_ON = 64, bool STRIDED = false>
__global__ __launch_bounds__(128) void k(const Pair* __restrict__ t, uint32_t n_pairs, int iters, double* out) {
    if (STRIDED ? ((threadIdx.x & 63) % (64 / LANES_ON)) != 0 : (int)(threadIdx.x & 63) >= LANES_ON) return;
    __shared__ __align__(16) char lds[2][64 * 128];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t idx = (blockIdx.x * 128 + threadIdx.x) * 2654435761u % n_pairs;
    double acc = 0;
    char* base = lds[wv];
    for (int it = 0; it < iters; ++it) {
        double v[12]; uint32_t a, b;
        if (MODE == 8) {          // mode 3 through buffer loads: one V# for the table, 32-bit byte offsets per lane
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)t, 0, (int)(n_pairs * 64u), 0x00020000);
            const uint32_t off = idx * 64u;
            const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
            const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0), q3 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = (double)__uint_as_float(q0[i]) * 100.0; v[4 + i] = (double)__uint_as_float(q1[i]) * 100.0; v[8 + i] = (double)__uint_as_float(q2[i]) * 100.0; }
            a = q3[0]; b = q3[1];
        } else if (MODE == 3) {
            const uint4* p4 = (const uint4*)((const char*)t + (size_t)idx * 64);
            const uint4 q0 = p4[0], q1 = p4[1], q2 = p4[2], q3 = p4[3];
            const float* f = (const float*)&q0; const float* g = (const float*)&q1; const float* h = (const float*)&q2;
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = (double)f[i] * 100.0; v[4 + i] = (double)g[i] * 100.0; v[8 + i] = (double)h[i] * 100.0; }
            a = q3.x; b = q3.y;
        } else if (MODE == 0) {
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
        // 64-byte records: reuse the buffer, a/b fields of record i live at byte 64*i+48
        for (int mode = 0; mode < 10; ++mode) for (int bl : {2048, 1024}) {
            if (mode == 3 && bl == 2048) {
                std::vector<uint32_t> h64((size_t)n_pairs * 16);
                uint32_t s2 = 777;
                for (uint32_t i = 0; i < n_pairs; ++i) { for (int k = 0; k < 12; ++k) { float fv = (float)((s2 = s2 * 1664525u + 1013904223u) * 1e-9); memcpy(&h64[(size_t)i * 16 + k], &fv, 4); }
                    h64[(size_t)i * 16 + 12] = (s2 = s2 * 1664525u + 1013904223u) % n_pairs; h64[(size_t)i * 16 + 13] = 1; h64[(size_t)i * 16 + 14] = h64[(size_t)i * 16 + 15] = 0; }
                CHECK(hipMemcpy(d, h64.data(), (size_t)n_pairs * 64, hipMemcpyHostToDevice));
            }
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 4) hipLaunchKernelGGL((k<3, 32, false>), dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 5) hipLaunchKernelGGL((k<3, 32, true>), dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 6) hipLaunchKernelGGL((k<3, 16, true>), dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 9) hipLaunchKernelGGL((k<8, 32, true>), dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                if (mode == 7) hipLaunchKernelGGL((k<0, 32, true>), dim3(bl), dim3(128), 0, 0, d, n_pairs, iters, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double frac = mode == 4 || mode == 5 || mode == 7 || mode == 9 ? 0.5 : (mode == 6 ? 0.25 : 1.0);
            double visits = (double)bl * 128 * iters * frac;
            printf("table %7.1f MB mode %d blocks %4d: %8.3f ms  %7.2f Gvisits/s  %6.2f Gcalls/s\n", n_pairs * (mode >= 3 && mode != 7 ? 64e-6 : 128e-6), mode, bl, best, visits / best / 1e6, (double)bl * 2 * iters / best / 1e6);
        }
        CHECK(hipFree(d)); CHECK(hipFree(out));
    }
    return 0;
}
