// mirge_libbuild.hpp -- host-side construction of a library's device image: 2-bit text,
// invalid bitmap, reference starts and the direct-addressed k-mer tables.  Plain C++ (no HIP) so
// that it is compiled into libmirge_native.so and, for logic tests without a GPU, into
// tests/hostsim.  It stands where `bowtie-build` stands for the reference.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "mirge_core.hpp"


struct MirgeHostLib {
    int64_t n_refs = 0;
    uint64_t total = 0;  // bases including one separator after every reference
    int kmax = 8;
    uint64_t valid_positions = 0;
    std::vector<uint64_t> T;    // (total+31)/32 + 8 words, zero padded
    std::vector<uint64_t> inv;  // (total+63)/64 + 4 words, padding = ones
    std::vector<uint32_t> ref_start;  // n_refs + 1
};

static inline int mirge_base_code(char ch) {
    switch (ch & 0xDF) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case 'U': return 3;
        default: return -1;  // N and every other IUPAC code: no window over it is a valid hit
    }
}

static inline int mirge_hostlib_build(MirgeHostLib& L, const char* seq, const int64_t* off, int64_t n_refs,
                                      std::string& err) {
    L.n_refs = n_refs;
    uint64_t total = 0;
    for (int64_t t = 0; t < n_refs; t++) {
        if (off[t + 1] < off[t]) { err = "offsets not monotone"; return -1; }
        total += (uint64_t)(off[t + 1] - off[t]) + 1;
    }
    if (total >= 0xFFFFFFF0ull) { err = "library larger than 2^32 bases is not supported"; return -5; }
    L.total = total;
    L.T.assign((size_t)((total + 31) / 32) + 8, 0ull);
    L.inv.assign((size_t)((total + 63) / 64) + 4, ~0ull);
    L.ref_start.resize((size_t)n_refs + 1);
    uint64_t g = 0;
    for (int64_t t = 0; t < n_refs; t++) {
        L.ref_start[(size_t)t] = (uint32_t)g;
        g += (uint64_t)(off[t + 1] - off[t]) + 1;  // the separator after every reference stays invalid
    }
    L.ref_start[(size_t)n_refs] = (uint32_t)g;
    // 2-bit packing on all host cores: every thread owns a range of global positions that starts and ends on a
    // multiple of 64 bases, so no word of T (32 bases) or inv (64 bases) is shared.  (The human-sized set used to
    // take 4.8 s of one core per process; this is what a one-sample CLI run waits for before its first kernel.)
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const int NT = (int)std::min<uint64_t>(hw, std::max<uint64_t>(1, total >> 20));
    std::vector<uint64_t> valid((size_t)NT, 0);
    auto work = [&](int w) {
        const uint64_t lo = (total * (uint64_t)w / NT) & ~63ull, hi = w + 1 == NT ? total : ((total * (uint64_t)(w + 1) / NT) & ~63ull);
        if (lo >= hi) return;
        // first reference that reaches into [lo, hi)
        int64_t t = (int64_t)(std::upper_bound(L.ref_start.begin(), L.ref_start.begin() + n_refs, (uint32_t)lo) - L.ref_start.begin()) - 1;
        if (t < 0) t = 0;
        uint64_t nv = 0;
        for (; t < n_refs && (uint64_t)L.ref_start[(size_t)t] < hi; t++) {
            const uint64_t rs = L.ref_start[(size_t)t];
            const int64_t len = off[t + 1] - off[t];
            const int64_t i0 = rs < lo ? (int64_t)(lo - rs) : 0;
            const int64_t i1 = rs + (uint64_t)len > hi ? (int64_t)(hi - rs) : len;
            for (int64_t i = i0; i < i1; i++) {
                const int code = mirge_base_code(seq[off[t] + i]);
                if (code >= 0) {
                    const uint64_t gp = rs + (uint64_t)i;
                    L.T[gp >> 5] |= (uint64_t)code << (2 * (gp & 31));
                    L.inv[gp >> 6] &= ~(1ull << (gp & 63));
                    nv++;
                }
            }
        }
        valid[(size_t)w] = nv;
    };
    {
        std::vector<std::thread> th;
        for (int w = 1; w < NT; w++) th.emplace_back(work, w);
        work(0);
        for (auto& x : th) x.join();
    }
    L.valid_positions = 0;
    for (uint64_t v : valid) L.valid_positions += v;
    // largest probe length: 4^k >= MIRGE_K_OVERSAMPLE x positions, clamped to [8, MIRGE_KMAX]
    int k = 8;
    while (k < MIRGE_KMAX && (1ull << (2 * k)) < (uint64_t)MIRGE_K_OVERSAMPLE * std::max<uint64_t>(L.valid_positions, 1)) k++;
    L.kmax = k;
    return 0;
}

// Every position whose whole span [p, p+k1+gap+k2) is valid, counting-sorted by the little-endian
// 2-bit value of block A (k1 bases at p) | block B (k2 bases at p+k1+gap) << 2*k1; positions ascend
// inside a bucket.  gap = k2 = 0 is the ordinary k1-mer table.
static inline uint64_t mirge_text_kmer(const uint64_t* T, uint64_t g, int k) {
    if (k <= 0) return 0;
    const uint64_t q = g >> 5;
    const int s = (int)(g & 31) * 2;
    uint64_t lo = T[q] >> s;
    if (s) lo |= T[q + 1] << (64 - s);
    return lo & mirge_lowmask2(k);
}

static inline void mirge_hostlib_table(const MirgeHostLib& L, int k1, int gap, int k2, std::vector<uint32_t>& bucket,
                                       std::vector<uint64_t>& entry, std::vector<uint32_t>& pos) {
    const uint64_t nb = 1ull << (2 * (k1 + k2));
    const int span = k1 + (k2 > 0 ? gap + k2 : 0);
    bucket.assign(nb + 1, 0u);
    const uint64_t* T = L.T.data();
    const uint64_t* inv = L.inv.data();
    for (int phase = 0; phase < 2; phase++) {
        if (phase == 1) {
            uint32_t acc = 0;
            for (uint64_t b = 0; b <= nb; b++) { uint32_t c = bucket[b]; bucket[b] = acc; acc += c; }
            pos.assign(std::max<size_t>(acc, 1), 0u);
        }
        int run = 0;  // valid bases ending at g
        for (uint64_t g = 0; g < L.total; g++) {
            if ((inv[g >> 6] >> (g & 63)) & 1ull) { run = 0; continue; }
            if (++run >= span) {
                const uint64_t p0 = g - (uint64_t)span + 1;
                uint64_t key = mirge_text_kmer(T, p0, k1);
                if (k2 > 0) key |= mirge_text_kmer(T, p0 + (uint64_t)(k1 + gap), k2) << (2 * k1);
                if (phase == 0) bucket[key]++;
                else pos[bucket[key]++] = (uint32_t)p0;
            }
        }
    }
    for (uint64_t b = nb; b > 0; b--) bucket[b] = bucket[b - 1];
    bucket[0] = 0;
    entry.assign(nb, 0ull);
    for (uint64_t b = 0; b < nb; b++) {
        const uint32_t cnt = bucket[b + 1] - bucket[b];
        entry[b] = MIRGE_ENTRY(cnt, cnt == 1 ? pos[bucket[b]] : bucket[b]);
    }
}
