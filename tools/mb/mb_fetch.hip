// Calibration of rocprofv3's FETCH_SIZE for the access shape of k_extend_bvh2: every lane reads its own record as 16-byte
// loads at an address unrelated to its neighbours'.  MI355X_MICROARCH.md (HBM) calibrates the counter only for wide
// coalesced streaming reads (it reports half of their bytes); this program reads a KNOWN number of distinct bytes from a
// table far larger than L2 + Infinity Cache (so every record really comes over the fabric) in three shapes, one kernel each:
//   k_gather64   one 64-byte record per lane  (4 x global_load_dwordx4), random 64-B-aligned records
//   k_gather128  one 128-byte record per lane (8 x global_load_dwordx4), random 128-B-aligned records
//   k_stream     coalesced: consecutive lanes read consecutive 16-byte words (the guide's calibrated case, as a control)
// Run plain for the byte counts and rates, and under `rocprofv3 --pmc FETCH_SIZE` for the counter per kernel;
// tools/fetch_calibration.py divides one by the other.  Build: hipcc --offload-arch=gfx950 -O3 -o mb_fetch mb_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int REC>   // record bytes: 64 or 128
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ t, uint32_t n_rec, int visits, uint32_t salt, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int v = 0; v < visits; ++v) {
        // a hashed index: every (tid, v) pair names a record of its own up to the collisions of a hash over a table with 4-8x
        // more records than visits; a repeated record is fetched again unless it is still cached (L2 + Infinity Cache hold 3.5 %
        // of the table), so the bytes that cross the fabric are the requested bytes to within a few per cent
        const uint32_t r = mix(tid * 0x9e3779b9u + (uint32_t)v * 0x85ebca6bu + salt) % n_rec;
        const uint4* p = t + (size_t)r * (REC / 16);
#pragma unroll
        for (int i = 0; i < REC / 16; ++i) { const uint4 q = p[i]; acc += q.x ^ q.y ^ q.z ^ q.w; }
    }
    out[tid] = acc;
}
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ t, size_t n_words, uint32_t* out) {
    const size_t T = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += T) { const uint4 q = t[i]; acc += q.x ^ q.y ^ q.z ^ q.w; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void k_fill(uint4* t, size_t n_words) {
    const size_t T = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += T) t[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main() {
    const size_t table_bytes = (size_t)8 << 30;            // 8 GiB: 32x the Infinity Cache
    const int blocks = 4096, visits = 16;                    // 4096 x 256 lanes x 16 visits = 16.8 M records per launch
    uint4* t; uint32_t* out;
    CHECK(hipMalloc(&t, table_bytes)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, t, table_bytes / 16);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double n_visits = (double)blocks * 256 * visits;
    // two footprints: the whole 8 GiB table (every record comes from HBM) and its first 128 MiB (beyond the 4 MiB L2 of an XCD,
    // inside the 256 MiB Infinity Cache once warm: the second launch of each shape shows whether such hits are counted)
    for (size_t foot : {table_bytes, (size_t)128 << 20}) for (int rep = 0; rep < 2; ++rep) {
        float ms;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather<64>, dim3(blocks), dim3(256), 0, 0, t, (uint32_t)(foot / 64), visits, 1000u + rep, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_gather64  footprint %5zu MiB launch %d: %.0f records x 64 B = %.0f bytes requested, %.3f ms, %.1f GB/s\n", foot >> 20, rep, n_visits, n_visits * 64, ms, n_visits * 64 / ms / 1e6);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather<128>, dim3(blocks), dim3(256), 0, 0, t, (uint32_t)(foot / 128), visits, 2000u + rep, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_gather128 footprint %5zu MiB launch %d: %.0f records x 128 B = %.0f bytes requested, %.3f ms, %.1f GB/s\n", foot >> 20, rep, n_visits, n_visits * 128, ms, n_visits * 128 / ms / 1e6);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, t, foot / 16, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_stream    footprint %5zu MiB launch %d: %.0f bytes requested, %.3f ms, %.1f GB/s\n", foot >> 20, rep, (double)foot, ms, (double)foot / ms / 1e6);
    }
    CHECK(hipFree(t)); CHECK(hipFree(out));
    return 0;
}
