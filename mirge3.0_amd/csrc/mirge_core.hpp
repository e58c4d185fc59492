// mirge_core.hpp -- data layout and per-read alignment arithmetic shared by the HIP kernels
// (mirge_kernels.hip) and the host-side index builder.  Everything here is integer work on
// 2-bit packed sequences; there is no floating point on this path.
//
// Layout (DESIGN.md "Data layout in HBM"):
//   * a base is 2 bits, A=0 C=1 G=2 T=3; base j of a sequence sits at bits [2j, 2j+1] of
//     word j/32 (little-endian inside a u64), so "the first k bases" is a low-bit mask;
//   * reads live in width groups W in {1,2,4} words (<=32, <=64, <=128 nt), structure of
//     arrays, word-major: seq[w*n + i]; len[i] (u8); nmask[w*n + i] (bit 2j set = base j is
//     an ambiguous call, its 2-bit code is 0) or nullptr when the group has no N;
//   * a library is ONE concatenated 2-bit string T with one separator base after every
//     reference, an "invalid" bitmap inv (1 bit per base: separator, reference N, padding)
//     and per-k direct-addressed k-mer tables (bucket[4^k+1], pos[]) over the valid k-windows.
//     A hit is reported as its global position in T; lowest position == lowest reference
//     index, then leftmost offset, which is the documented tie-break.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MIRGE_HD __host__ __device__ __forceinline__
#else
#define MIRGE_HD inline
#endif

#define MIRGE_KMAX 14          // largest direct-addressed k (4^14+1 u32 = 1 GiB)
#define MIRGE_MAX_READ_LEN 128
#define MIRGE_NO_HIT 0xFFFFFFFFFFFFFFFFull

// Cascade policy of one pass: the restated bowtie-1 argument string
// (reference: mirge/libs/manifoldAlign.py:85).
struct MirgePolicy {
    int32_t mode;      // 0: -n (seeded)   1: -v (end to end)
    int32_t mm;        // N of -n / V of -v
    int32_t seedlen;   // -l (28)
    int32_t maxtotal;  // -n: floor(70/30) = 2 mismatches overall; -v: V
    int32_t trim5;     // -5
    int32_t trim3;     // -3
    int32_t ttail;     // pass 3: only reads matching T{3,}$, aligned without the T run (:118-126)
    int32_t len_lt;    // >0: only reads with len <  len_lt (pass 0, :93)
    int32_t len_gt;    // >0: only reads with len >  len_gt (pass 1, :104)
    int32_t reserved;
};

struct MirgeKTable {
    const uint32_t* bucket;  // 4^k + 1 entries
    const uint32_t* pos;     // global positions, ascending inside a bucket
};

struct MirgeLibView {
    const uint64_t* T;        // 2-bit text (+ >= 6 words of zero padding)
    const uint64_t* inv;      // invalid bitmap, 1 bit per base (+ padding of ones)
    const uint32_t* ref_start;  // n_refs + 1
    const MirgeKTable* tables;  // [MIRGE_KMAX + 1], device memory
    uint64_t total;           // bases in T including separators
    uint32_t n_refs;
    int32_t kmax;             // largest k this library is probed with
};

template <int W>
struct MirgeRead {
    uint64_t w[W];
    uint64_t nm[W];
    int32_t len;
};

MIRGE_HD uint64_t mirge_lowmask2(int nbases) {  // low 2*nbases bits, nbases in [0,32]
    return nbases >= 32 ? ~0ull : ((1ull << (2 * nbases)) - 1ull);
}

MIRGE_HD int mirge_popc(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// bases [a, a+k) of a packed sequence as an integer (k <= 32), little-endian
template <int W>
MIRGE_HD uint64_t mirge_extract(const uint64_t* w, int a, int k) {
    int q = a >> 5, s = (a & 31) * 2;
    uint64_t lo = w[q] >> s;
    if (W > 1 && s != 0 && q + 1 < W) lo |= w[q + 1] << (64 - s);
    return lo & mirge_lowmask2(k);
}

// drop `n5` bases at the 5' end (n5 < 32) and keep `newlen` bases
template <int W>
MIRGE_HD void mirge_trim(MirgeRead<W>& r, int n5, int newlen) {
    if (n5 > 0) {
        int s = 2 * n5;
#pragma unroll
        for (int i = 0; i < W; i++) {
            uint64_t hi = (i + 1 < W) ? r.w[i + 1] : 0ull;
            uint64_t hn = (i + 1 < W) ? r.nm[i + 1] : 0ull;
            r.w[i] = (r.w[i] >> s) | (hi << (64 - s));
            r.nm[i] = (r.nm[i] >> s) | (hn << (64 - s));
        }
    }
#pragma unroll
    for (int i = 0; i < W; i++) {
        int rem = newlen - 32 * i;
        uint64_t m = rem <= 0 ? 0ull : mirge_lowmask2(rem > 32 ? 32 : rem);
        r.w[i] &= m;
        r.nm[i] &= m;
    }
    r.len = newlen;
}

// length of the terminal run of T (code 3, not N) -- re.search('T{3,}$') of manifoldAlign.py:122
template <int W>
MIRGE_HD int mirge_t_run(const MirgeRead<W>& r) {
    int run = 0;
    for (int j = r.len - 1; j >= 0; j--) {
        uint64_t b = (r.w[j >> 5] >> (2 * (j & 31))) & 3ull;
        uint64_t n = (r.nm[j >> 5] >> (2 * (j & 31))) & 1ull;
        if (b != 3ull || n) break;
        run++;
    }
    return run;
}

// What bowtie is handed for this read under this policy.  false: the read is not in the
// pass's FASTA, or bowtie skips it (empty, or length <= mismatch budget).
template <int W>
MIRGE_HD bool mirge_effective_read(MirgeRead<W>& r, const MirgePolicy& p) {
    int L = r.len;
    if (p.len_lt > 0 && !(L < p.len_lt)) return false;
    if (p.len_gt > 0 && !(L > p.len_gt)) return false;
    int l = L;
    if (p.ttail) {
        int run = mirge_t_run<W>(r);
        if (run < 3) return false;
        l = L - run;
    }
    l -= p.trim5 + p.trim3;
    if (l < 1 || l <= p.mm) return false;
    if (l != L || p.trim5) mirge_trim<W>(r, p.trim5, l);
    return true;
}

// Mismatches of the read against the window of T that starts at global position g.
// Returns total mismatches, or -1 when the policy is violated.  (inv is checked separately.)
template <int W>
MIRGE_HD int mirge_window_mm(const uint64_t* __restrict__ T, uint64_t g, const MirgeRead<W>& r,
                             const MirgePolicy& p) {
    const int L = r.len;
    const uint64_t q = g >> 5;
    const int s = (int)(g & 31) * 2;
    int tot = 0, seedmm = 0;
    const int seed = p.mode == 0 ? (L < p.seedlen ? L : p.seedlen) : L;
    uint64_t cur = T[q];
#pragma unroll
    for (int i = 0; i < W; i++) {
        if (32 * i < L) {
            uint64_t nxt = T[q + i + 1];
            uint64_t t = s ? ((cur >> s) | (nxt << (64 - s))) : cur;
            cur = nxt;
            uint64_t x = r.w[i] ^ t;
            uint64_t m = (x | (x >> 1)) & 0x5555555555555555ull;
            int rem = L - 32 * i;
            m &= mirge_lowmask2(rem > 32 ? 32 : rem);
            m |= r.nm[i];
            tot += mirge_popc(m);
            int srem = seed - 32 * i;
            if (srem > 0) seedmm += mirge_popc(m & mirge_lowmask2(srem > 32 ? 32 : srem));
        }
    }
    if (tot > p.maxtotal || seedmm > p.mm) return -1;
    return tot;
}

// any invalid base (separator / reference N / padding) in [g, g+L) ?
MIRGE_HD bool mirge_window_invalid(const uint64_t* __restrict__ inv, uint64_t g, int L) {
    uint64_t q = g >> 6;
    int s = (int)(g & 63);
    int done = 0;
    while (done < L) {
        uint64_t lo = inv[q] >> s;
        if (s) lo |= inv[q + 1] << (64 - s);
        int n = L - done;
        if (n < 64) lo &= (1ull << n) - 1ull;
        if (lo) return true;
        done += 64;
        q++;
    }
    return false;
}

// Best alignment of an (already trimmed) read in one library through the k-mer tables.
// Pigeonhole: the seed region (first min(seedlen,L) bases in -n mode, the whole read in -v
// mode) is cut into mm+1 segments; an alignment within budget leaves one segment untouched,
// so its first k bases are found by an exact table probe; every candidate window is then
// verified in full.  best = (total mismatches << 32) | global position, minimised.
// Returns false if a needed table is missing (k < 1).
template <int W>
MIRGE_HD bool mirge_align_indexed(const MirgeLibView& lib, const MirgePolicy& p,
                                  const MirgeRead<W>& r, uint64_t& best) {
    best = MIRGE_NO_HIT;
    const int L = r.len;
    const int seed = p.mode == 0 ? (L < p.seedlen ? L : p.seedlen) : L;
    const int nseg = p.mm + 1;
    const int h = seed / nseg;
    const int k = h < lib.kmax ? h : lib.kmax;
    if (k < 1) return false;
    const MirgeKTable tb = lib.tables[k];
    for (int sg = 0; sg < nseg; sg++) {
        const int a = sg * h;
        if (mirge_extract<W>(r.nm, a, k)) continue;  // an N inside the probe: cannot be exact
        const uint64_t key = mirge_extract<W>(r.w, a, k);
        const uint32_t lo = tb.bucket[key], hi = tb.bucket[key + 1];
        for (uint32_t c = lo; c < hi; c++) {
            const uint32_t pz = tb.pos[c];
            if (pz < (uint32_t)a) continue;
            const uint64_t g = (uint64_t)pz - (uint64_t)a;
            const int m = mirge_window_mm<W>(lib.T, g, r, p);
            if (m < 0) continue;
            if (mirge_window_invalid(lib.inv, g, L)) continue;
            const uint64_t cand = ((uint64_t)m << 32) | g;
            if (cand < best) best = cand;
            // positions ascend inside a bucket and a 0-mismatch window is in segment 0's
            // bucket, so the first one seen there is the global minimum
            if (sg == 0 && m == 0) return true;
        }
    }
    return true;
}

MIRGE_HD uint64_t mirge_mix64(uint64_t x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
