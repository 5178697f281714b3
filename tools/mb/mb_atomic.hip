// mb_atomic: throughput of returning atomicAdd on R counters (device scope), one atomic per block-iteration, as the
// queue compaction of k_shade / k_generate issues them.  usage: mb_atomic  (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k(uint32_t* ctr, uint32_t R, uint32_t stride_words, uint32_t iters, uint32_t* sink) {
    uint32_t* c = ctr + (size_t)(blockIdx.x % R) * stride_words;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        __shared__ uint32_t s;
        if (threadIdx.x == 0) s = atomicAdd(c, 1u);
        __syncthreads();
        acc += s;
        __syncthreads();
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

int main() {
    uint32_t *ctr, *sink; const size_t words = 64 * 4096;
    CK(hipMalloc(&ctr, words * 4)); CK(hipMalloc(&sink, 4)); CK(hipMemset(ctr, 0, words * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t blocks = 4096, iters = 64;
    printf("%8s %12s %12s %14s\n", "counters", "stride B", "ms", "atomics/us");
    const uint32_t Rs[] = {1, 2, 8, 8, 8, 64, 64}, strides[] = {1, 1024, 16, 64, 1024, 64, 1024};
    for (int t = 0; t < 7; ++t) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, Rs[t], strides[t], iters, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        }
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%8u %12u %12.3f %14.1f\n", Rs[t], strides[t] * 4, ms, (double)blocks * iters / (ms * 1e3));
    }
    return 0;
}
