// Ceiling for the cascade's table probes: rate of independent random 4-byte loads from a table of S bytes
// (one 64-B sector each), and of a dependent pair (bucket -> position list), on one MI355X.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_random.hip && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

template <int PER, int DEP>
__global__ __launch_bounds__(256) void k_probe(const uint32_t* __restrict__ tab, uint64_t mask, const uint32_t* __restrict__ tab2,
                                               uint64_t mask2, uint64_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) v[q] = tab[mix(i * PER + q + 1) & mask];
        if (DEP) {
#pragma unroll
            for (int q = 0; q < PER; q++) v[q] = tab2[(mix(i * PER + q + 77) + v[q]) & mask2];
        }
#pragma unroll
        for (int q = 0; q < PER; q++) acc += v[q];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint64_t n = 16ull << 20;  // threads' worth of work items
    uint32_t* out; OK(hipMalloc((void**)&out, 4));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    printf("%10s %6s %4s %10s %12s %12s\n", "table", "per", "dep", "ms", "G loads/s", "sector GB/s");
    for (uint64_t mb : {4ull, 32ull, 128ull, 256ull, 1024ull, 4096ull, 16384ull}) {
        const uint64_t words = mb << 18;
        uint32_t* tab; OK(hipMalloc((void**)&tab, words * 4)); OK(hipMemset(tab, 0, words * 4));
        for (int dep = 0; dep < 2; dep++) {
            for (int per : {1, 4}) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; rep++) {
                    OK(hipEventRecord(e0));
                    if (per == 1 && !dep) k_probe<1, 0><<<256 * 8, 256>>>(tab, words - 1, tab, words - 1, n, out);
                    if (per == 4 && !dep) k_probe<4, 0><<<256 * 8, 256>>>(tab, words - 1, tab, words - 1, n, out);
                    if (per == 1 && dep) k_probe<1, 1><<<256 * 8, 256>>>(tab, words - 1, tab, words - 1, n, out);
                    if (per == 4 && dep) k_probe<4, 1><<<256 * 8, 256>>>(tab, words - 1, tab, words - 1, n, out);
                    OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
                    float ms; OK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) best = ms < best ? ms : best;
                }
                const double loads = (double)n * per * (dep ? 2 : 1);
                printf("%8lluMB %6d %4d %10.3f %12.1f %12.1f\n", (unsigned long long)mb, per, dep, best, loads / best / 1e6, loads * 64 / best / 1e6);
            }
        }
        OK(hipFree(tab));
    }
    return 0;
}
