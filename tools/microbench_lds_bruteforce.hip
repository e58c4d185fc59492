// The brief's starting idea, measured: a tiny reference (the mature-miRNA library, ~57 k bases = 14 KiB of 2-bit
// text) staged wholesale into LDS, every read scored against EVERY window with XOR / fold / popcount (<= 2
// mismatches).  This is what the indexed cascade replaces; the number below is why.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_lds tools/microbench_lds_bruteforce.hip && /tmp/mb_lds
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int TEXT_BASES = 57344;            // ~ human mature miRNAs + separators
constexpr int TEXT_WORDS = TEXT_BASES / 32;  // u64 words of 32 bases

__global__ __launch_bounds__(256) void k_brute(const uint64_t* __restrict__ text, const uint64_t* __restrict__ reads,
                                               const uint8_t* __restrict__ len, uint32_t n, uint32_t* __restrict__ best_out) {
    __shared__ uint64_t T[TEXT_WORDS + 2];
    for (int i = threadIdx.x; i < TEXT_WORDS + 2; i += blockDim.x) T[i] = i < TEXT_WORDS ? text[i] : 0ull;
    __syncthreads();
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const uint64_t q = reads[r];
        const int L = len[r];
        const uint64_t mask = L >= 32 ? ~0ull : ((1ull << (2 * L)) - 1ull);
        uint32_t best = 0xFFFFFFFFu;
        for (int w = 0; w < TEXT_WORDS; w++) {
            const uint64_t a = T[w], b = T[w + 1];
#pragma unroll 8
            for (int s = 0; s < 32; s++) {
                const uint64_t t = s ? ((a >> (2 * s)) | (b << (64 - 2 * s))) : a;
                const uint64_t x = (q ^ t);
                const uint64_t m = (x | (x >> 1)) & 0x5555555555555555ull & mask;
                const uint32_t mm = (uint32_t)__popcll(m);
                const uint32_t cand = (mm << 24) | (uint32_t)(w * 32 + s);
                if (mm <= 2 && cand < best) best = cand;
            }
        }
        best_out[r] = best;
    }
}

int main() {
    const uint32_t n = 1u << 20;  // 1 M reads (the isomiR pass sees 1.8 M)
    std::vector<uint64_t> text(TEXT_WORDS), reads(n);
    std::vector<uint8_t> len(n);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (auto& t : text) t = rnd();
    for (uint32_t i = 0; i < n; i++) { reads[i] = rnd(); len[i] = (uint8_t)(16 + rnd() % 10); }
    uint64_t *dt, *dr; uint8_t* dl; uint32_t* db;
    OK(hipMalloc((void**)&dt, TEXT_WORDS * 8)); OK(hipMalloc((void**)&dr, n * 8ull)); OK(hipMalloc((void**)&dl, n)); OK(hipMalloc((void**)&db, n * 4ull));
    OK(hipMemcpy(dt, text.data(), TEXT_WORDS * 8, hipMemcpyHostToDevice));
    OK(hipMemcpy(dr, reads.data(), n * 8ull, hipMemcpyHostToDevice));
    OK(hipMemcpy(dl, len.data(), n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        OK(hipEventRecord(e0));
        k_brute<<<256 * 8, 256>>>(dt, dr, dl, n, db);
        OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double windows = (double)n * TEXT_BASES;
    printf("LDS brute force: %u reads x %d windows: %.2f ms -> %.1f M reads/s, %.2f T windows/s\n", n, TEXT_BASES, best,
           n / best / 1e3, windows / best / 1e9);
    return 0;
}
