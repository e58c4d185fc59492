// mirge_core.hpp -- data layout and per-read alignment arithmetic shared by the HIP kernels
// (mirge_kernels.hip) and the host-side index builder.  Everything here is integer work on
// 2-bit packed sequences; there is no floating point on this path.
//
// Layout (DESIGN.md "Data layout in HBM"):
//   * a base is 2 bits, A=0 C=1 G=2 T=3; base j of a sequence sits at bits [2j, 2j+1] of
//     word j/32 (little-endian inside a u64), so "the first k bases" is a low-bit mask;
//   * reads live in width groups W in {1,2,4,8} words (<=31, <=64, <=128, <=255 nt; longer reads: the long class of
//     kernels_long.hpp, W words with W set per read set and the length in 16 bits), structure of
//     arrays, word-major: seq[w*n + i]; len[i] (u8); nmask[w*n + i] (bit 2j set = base j is
//     an ambiguous call, its 2-bit code is 0) or nullptr when the group has no N;
//   * a library is ONE concatenated 2-bit string T with one separator base after every
//     reference, an "invalid" bitmap inv (1 bit per base: separator, reference N, padding)
//     and per-k direct-addressed k-mer tables (bucket[4^k] entries {count, position | start}, pos[]) over the valid k-windows.
//     A hit is reported as its global position in T; lowest position == lowest reference
//     index, then leftmost offset, which is the documented tie-break.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MIRGE_HD __host__ __device__ __forceinline__
#else
#define MIRGE_HD inline
#endif

#define MIRGE_KMAX 14          // largest direct-addressed k (4^14+1 u32 = 1 GiB) of multi-probe plans
#define MIRGE_KMAX0 15         // ... of the single exact probe of an mm = 0 policy (one 4 GiB table per library)
#ifndef MIRGE_K_OVERSAMPLE
#define MIRGE_K_OVERSAMPLE 4   // 4^K >= this x the library's positions
#endif
#define MIRGE_MAX_READ_LEN 255   // the four TEMPLATED width classes of 1 / 2 / 4 / 8 words, whose length is a byte (longer: kernels_long.hpp)
#define MIRGE_NO_HIT 0xFFFFFFFFFFFFFFFFull
// align_hybrid's answer for a read one of whose probe buckets holds more windows than the pass's MirgePolicy::reserved (> 0): not
// aligned here -- the read goes to k_cascade_heavy, one workgroup per read (kernels_cascade.hpp, round 6)
#define MIRGE_DEFER 0xFFFFFFFFFFFFFFFEull

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

// Two bucket formats, by table size:
//   * tables of <= 4^MIRGE_BITMAP_MAXK buckets (bits != nullptr) keep CSR bounds, uint32 bucket[4^k + 1]: the bitmap
//     answers "empty?" from L2 and only the ~20 % non-empty probes read the two bounds; 4 bytes per bucket keep the
//     tables of the small libraries L2-resident (8-byte entries measured 2-5 % slower on those passes);
//   * larger tables (bits == nullptr) hold ONE self-contained 8-byte entry per bucket: count in the high word; in the
//     low word the position itself when the bucket holds a single window (most non-empty buckets do: 4^K >= 4 x
//     positions), else where its list starts in pos[].  One L2-missing sector answers "empty?", and a singleton goes
//     straight to the text: no second bound, no position list (-11 % on the merged snoRNA/rRNA/ncRNA pass).
#define MIRGE_ENTRY(count, low) (((uint64_t)(count) << 32) | (uint64_t)(uint32_t)(low))
struct MirgeKTable {
    const void* bucket;      // bits ? uint32 CSR bounds [4^(k1+k2) + 1] : uint64 entries [4^(k1+k2)] {count, position | start}
    const uint32_t* pos;     // global positions of block A's first base (entries: of the buckets with several windows)
    const uint32_t* bits;    // 1 bit per bucket (non-empty), or nullptr.  Present for tables of <= 4^10
                             // buckets: the bitmap (<= 128 KiB) stays in L2 while the bucket array does
                             // not, and most probes of an unannotatable read find an empty bucket
};

// A probe is one or two exact blocks of the read: block A = k1 bases at read offset a1, block B =
// k2 bases at a1 + k1 + gap (k2 == 0: none).  Its table is addressed by the shape (k1, gap, k2).
struct MirgeProbe {
    uint8_t a1;  // up to MIRGE_MAX_READ_LEN - 1
    int8_t k1, gap, k2;
};
#define MIRGE_MAX_PROBES 9
#define MIRGE_SHAPE_SLOTS (16 * 32 * 16)
MIRGE_HD int mirge_shape_id(int k1, int gap, int k2) { return (k1 * 32 + gap) * 16 + k2; }

struct MirgeLibView {
    const uint64_t* T;        // 2-bit text (+ >= 6 words of zero padding)
    const uint64_t* inv;      // invalid bitmap, 1 bit per base (+ padding of ones)
    const uint32_t* ref_start;  // n_refs + 1
    const MirgeKTable* tables;  // [MIRGE_SHAPE_SLOTS], indexed by mirge_shape_id, device memory
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

#ifndef MIRGE_WIDE_SELECT
#define MIRGE_WIDE_SELECT 1  // 0: the words of a wide read indexed at run time, as rounds 1-4 (A/B)
#endif
// bases [a, a+k) of a packed sequence as an integer (k <= 32), little-endian
template <int W>
MIRGE_HD uint64_t mirge_extract(const uint64_t* w, int a, int k) {
    // (one word: no run-time index into the read -- an index the compiler cannot fold keeps the read's words out of registers;
    // in a kernel that meant a 24-byte-per-thread copy of the read in LDS, 6 KiB per workgroup)
    if (W == 1) return (w[0] >> ((a & 31) * 2)) & mirge_lowmask2(k);
    // (wider reads, same reason: the two words are SELECTED, one compare per word, instead of indexed -- the index had kept a
    // 24 W-byte copy of the read in per-thread scratch in every kernel of the 32-255-nt groups)
    const int q = a >> 5, s = (a & 31) * 2;
#if MIRGE_WIDE_SELECT
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
        lo = i == q ? w[i] : lo;
        hi = i == q + 1 ? w[i] : hi;
    }
    uint64_t v = lo >> s;
    if (s != 0) v |= hi << (64 - s);
    return v & mirge_lowmask2(k);
#else
    uint64_t lo = w[q] >> s;
    if (W > 1 && s != 0 && q + 1 < W) lo |= w[q + 1] << (64 - s);
    return lo & mirge_lowmask2(k);
#endif
}

// drop `n5` bases at the 5' end (n5 < 32) and keep `newlen` bases
template <int W>
MIRGE_HD void mirge_trim_read(MirgeRead<W>& r, int n5, int newlen) {
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
    if (W == 1) {  // one word: bit arithmetic instead of a loop that indexes the read at run time (see mirge_extract)
        if (r.len <= 0) return 0;
        const uint64_t x = r.w[0];
        // bit 2j = base j is a T (both code bits set) and no ambiguous call; the odd bits filled with ones
        const uint64_t t = ((x & (x >> 1) & ~r.nm[0]) & 0x5555555555555555ull) | 0xAAAAAAAAAAAAAAAAull;
        const uint64_t top = ~(t << (64 - 2 * r.len));  // base len-1 in the two highest bits: 00 for a T, 01 otherwise; below the read: ones
        const int lead = top ? (int)__builtin_clzll(top) : 64;
        const int run1 = lead >> 1;
        return run1 < r.len ? run1 : r.len;
    }
#if !MIRGE_WIDE_SELECT
    int run = 0;
    for (int j = r.len - 1; j >= 0; j--) {
        uint64_t b = (r.w[j >> 5] >> (2 * (j & 31))) & 3ull;
        uint64_t n = (r.nm[j >> 5] >> (2 * (j & 31))) & 1ull;
        if (b != 3ull || n) break;
        run++;
    }
    return run;
#endif
    // wider reads: the last base that is no T (or is an ambiguous call), word by word with compile-time indices
    int last_other = -1;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const int nv = r.len - 32 * i;
        const uint64_t valid = (nv <= 0 ? 0ull : mirge_lowmask2(nv < 32 ? nv : 32)) & 0x5555555555555555ull;
        const uint64_t x = r.w[i];
        const uint64_t other = ~(x & (x >> 1) & ~r.nm[i]) & valid;  // bit 2j: base 32 i + j is not a T
        if (other) last_other = 32 * i + ((63 - (int)__builtin_clzll(other)) >> 1);
    }
    return r.len - 1 - last_other;
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
    if (l != L || p.trim5) mirge_trim_read<W>(r, p.trim5, l);
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

// ---- probe plan ---------------------------------------------------------------------------
// Which exact probes guarantee that no alignment within the mismatch budget is missed.
//   Plain pigeonhole: cut the seed region S (first min(seedlen,L) bases in -n mode, the whole
//   read in -v mode) into mm+1 segments; one of them carries no mismatch.  With short reads the
//   segments are short (a 16-nt read under -v 2 gives 4-mers) and a probe returns hundreds of
//   windows.  Recursive pigeonhole: once segment i is mismatch-free, the other segments hold all
//   <= mm mismatches, so cut THEM into mm+1 parts: one part is mismatch-free too.  Every
//   (segment, part) pair is a probe of up to K exact bases in one or two blocks -- (mm+1)^2
//   probes, each about 4^(part length) times more selective.  Used when a segment is shorter
//   than K, the library's probe length.
MIRGE_HD void mirge_two_blocks(int A0, int A1, int B0, int B1, int K, MirgeProbe& pr) {
    int la = A1 - A0, lb = B1 - B0;
    if (la <= 0 && lb <= 0) { pr.a1 = 0; pr.k1 = 0; pr.gap = 0; pr.k2 = 0; return; }
    if (la <= 0) { A0 = B0; A1 = B1; la = lb; lb = 0; }
    if (lb <= 0 || A1 == B0) {  // one contiguous run
        const int len = lb > 0 ? (B1 - A0) : la;
        pr.a1 = (uint8_t)A0; pr.k1 = (int8_t)(len < K ? len : K); pr.gap = 0; pr.k2 = 0;
        return;
    }
    // two runs: suffix of A + prefix of B, ka + kb = min(la + lb, K); the smaller run gets <= 4
    // bases unless more are needed to reach K (keeps the number of distinct table shapes small)
    const int tot = (la + lb) < K ? (la + lb) : K;
    int ka, kb;
    if (la <= lb) { ka = la < 4 ? la : 4; kb = lb < (tot - ka) ? lb : (tot - ka); ka = la < (tot - kb) ? la : (tot - kb); }
    else { kb = lb < 4 ? lb : 4; ka = la < (tot - kb) ? la : (tot - kb); kb = lb < (tot - ka) ? lb : (tot - ka); }
    pr.a1 = (uint8_t)(A1 - ka); pr.k1 = (int8_t)ka; pr.gap = (int8_t)(B0 - A1); pr.k2 = (int8_t)kb;
}

// Three families of probe plans, all pigeonhole arguments over the seed region S:
//   plain    (mm+1 probes): mm+1 segments, one of them is mismatch-free.
//   recursive ((mm+1)^2): segment i is mismatch-free AND one of the mm+1 parts of the rest is (above).
//   subset   (C(mm+2, 2) = 3 or 6): cut S into mm+2 blocks; at most mm of them carry a mismatch, so some PAIR of
//            blocks is mismatch-free -- every pair is one two-block probe of 2S/(mm+2) exact bases (up to K).
// Fewer probes = fewer dependent table lookups (each an L2-missing sector in the big libraries); longer keys =
// fewer candidate windows to verify.  Which family is cheapest depends on S, K and the library's size, so the
// choice is made per (policy, read length, library) with an expected-cost model in units of random 64-B
// sectors: a probe costs 1 (bucket bounds) + P(bucket not empty) (position list) + MIRGE_VERIFY_SECTORS per
// expected window (text + invalid bitmap; less when the text is L2-resident).  tests/test_hostsim.py proves
// coverage for every family.  Measured: -7 % on the merged -n 1 pass, -5 % on the isomiR pass vs plain/recursive only.
#define MIRGE_SCHEME_PLAIN 0
#define MIRGE_SCHEME_RECURSIVE 1
#define MIRGE_SCHEME_SUBSET 2
#ifndef MIRGE_VERIFY_SECTORS
#define MIRGE_VERIFY_SECTORS 2.2      // text + invalid bitmap of a window in a library larger than L2
#define MIRGE_VERIFY_SECTORS_L2 0.7   // ... of a library whose text stays in L2 (< 4 M bases): a lookup costs latency, a window little
#endif

MIRGE_HD int mirge_scheme_count(const MirgePolicy& p, int scheme) {
    const int nseg = p.mm + 1;
    if (scheme == MIRGE_SCHEME_RECURSIVE) return nseg * nseg;
    if (scheme == MIRGE_SCHEME_SUBSET) return (p.mm + 2) * (p.mm + 1) / 2;
    return nseg < MIRGE_MAX_PROBES ? nseg : MIRGE_MAX_PROBES;
}

// probe number q of one family; written without arrays so that callers keep everything in registers
MIRGE_HD void mirge_scheme_probe(const MirgePolicy& p, int S, int K, int scheme, int q, MirgeProbe& out) {
    const int nseg = p.mm + 1;
    const int h = S / nseg;
    if (scheme == MIRGE_SCHEME_PLAIN) {
        out.a1 = (uint8_t)(q * h); out.k1 = (int8_t)(h < K ? h : K); out.gap = 0; out.k2 = 0;
        return;
    }
    if (scheme == MIRGE_SCHEME_SUBSET) {
        const int B = p.mm + 2;
        int i = 0, j = 1, t = q;  // q-th pair (i < j) in lexicographic order
        while (t >= B - 1 - i) { t -= B - 1 - i; i++; }
        j = i + 1 + t;
        mirge_two_blocks(S * i / B, S * (i + 1) / B, S * j / B, S * (j + 1) / B, K, out);
        return;
    }
    if (p.mm == 1) {
        const int h2a = (S - h) / 2, h2b = h / 2;
        switch (q) {
            case 0: mirge_two_blocks(0, h, h, h + h2a, K, out); break;  // mismatch in 2nd half of segment 1, or none
            case 1: mirge_two_blocks(0, h, h + h2a, S, K, out); break;  // ... in the 1st half of segment 1
            case 2: mirge_two_blocks(h2b, h, h, S, K, out); break;      // ... in the 1st half of segment 0
            default: mirge_two_blocks(0, h2b, h, S, K, out); break;     // ... in the 2nd half of segment 0
        }
        return;
    }
    // mm == 2: segment i = q / 3 is mismatch-free, then third j = q % 3 of the other two segments
    const int i = q / 3, j = q % 3;
    if (i == 0) {  // rest = [h, S)
        const int r = S - h, r1 = r / 3, r2 = r / 3;
        const int t0 = j == 0 ? h : (j == 1 ? h + r1 : h + r1 + r2);
        const int t1 = j == 0 ? h + r1 : (j == 1 ? h + r1 + r2 : S);
        mirge_two_blocks(0, h, t0, t1, K, out);
    } else if (i == 1) {  // rest = [0, 2h)
        const int r = 2 * h, r1 = r / 3, r2 = r / 3;
        const int t0 = j == 0 ? 0 : (j == 1 ? r1 : r1 + r2);
        const int t1 = j == 0 ? r1 : (j == 1 ? r1 + r2 : 2 * h);
        mirge_two_blocks(t0, t1, 2 * h, S, K, out);
    } else {  // segment 1 = [h, 2h); rest = [0, h) then [2h, S); rest coordinate t -> t (t < h) or t + h
        const int r = h + (S - 2 * h), r1 = r / 3, r2 = r / 3;
        const int t0 = j == 0 ? 0 : (j == 1 ? r1 : r1 + r2);
        const int t1 = j == 0 ? r1 : (j == 1 ? r1 + r2 : r);
        if (t0 < h && t1 > h) mirge_two_blocks(t0, t1 + h, 0, 0, K, out);   // straddles segment 1: one run
        else if (t1 <= h) mirge_two_blocks(t0, t1, h, 2 * h, K, out);       // left of segment 1
        else mirge_two_blocks(h, 2 * h, t0 + h, t1 + h, K, out);            // right of segment 1
    }
}

// expected cost of one family, in random sectors (see above); npos = bases of the library's text
MIRGE_HD double mirge_scheme_cost(const MirgePolicy& p, int S, int K, int scheme, uint64_t npos) {
    double cost = 0.0;
    const int n = mirge_scheme_count(p, scheme);
    for (int q = 0; q < n; q++) {
        MirgeProbe pr;
        mirge_scheme_probe(p, S, K, scheme, q, pr);
        const int k = pr.k1 + pr.k2;
        if (k <= 0) return 1e30;  // a degenerate probe (empty block) filters nothing: family unusable here
        double lam = (double)npos;
        for (int t = 0; t < k; t++) lam *= 0.25;
        // 1 - exp(-lam) without libm: lam / (1 + lam) is within 20 % of it and monotone, enough to rank plans
        cost += 1.0 + lam / (1.0 + lam) + (npos > (1ull << 22) ? MIRGE_VERIFY_SECTORS : MIRGE_VERIFY_SECTORS_L2) * lam;
    }
    return cost;
}

// family used for a read whose seed region is S bases, in a library of npos bases probed with up to K exact bases
MIRGE_HD int mirge_plan_scheme(const MirgePolicy& p, int S, int K, uint64_t npos) {
    if (p.mm == 0 || p.mm > 2 || S / (p.mm + 1) < 1) return MIRGE_SCHEME_PLAIN;
    int best = MIRGE_SCHEME_PLAIN;
    double bc = mirge_scheme_cost(p, S, K, MIRGE_SCHEME_PLAIN, npos);
    if (S >= 2 * (p.mm + 1)) {  // every part of the recursive cut holds at least one base
        const double c = mirge_scheme_cost(p, S, K, MIRGE_SCHEME_RECURSIVE, npos);
        if (c < bc) { bc = c; best = MIRGE_SCHEME_RECURSIVE; }
    }
    if (S >= p.mm + 2) {
        const double c = mirge_scheme_cost(p, S, K, MIRGE_SCHEME_SUBSET, npos);
        if (c < bc) { bc = c; best = MIRGE_SCHEME_SUBSET; }
    }
    return best;
}

// The region the probes are cut from: the seed (-n) or the read (-v).  Any PREFIX of it serves as well -- an alignment within
// the budget has at most mm mismatches inside the prefix too -- which MIRGE_SEED_CAP2 uses for the two-mismatch policies
// (experiment, profiles/README.md round 4): every read of at least that length then shares one plan and its few tables.
MIRGE_HD int mirge_seed_region(const MirgePolicy& p, int L) {
    int S = p.mode == 0 ? (L < p.seedlen ? L : p.seedlen) : L;
#ifdef MIRGE_SEED_CAP2
    if (p.mm == 2 && S > MIRGE_SEED_CAP2) S = MIRGE_SEED_CAP2;
#endif
    return S;
}

// An exact-seed policy (mm = 0) asks ONE table per library, so it can afford the next k when the library has
// outgrown K = 14 (human mRNA: 130 M positions in 268 M buckets, 39 % of the lookups go on to a position list
// and a window; at k = 15 11 % do: -45 % sectors in the mRNA pass for 4 GiB of HBM)
MIRGE_HD int mirge_policy_k(const MirgePolicy& p, int K, uint64_t npos) {
    if (p.mm == 0 && K == MIRGE_KMAX && (1ull << (2 * MIRGE_KMAX)) < (uint64_t)MIRGE_K_OVERSAMPLE * npos) return MIRGE_KMAX0;
    return K;
}

// number of probes for a read of (trimmed) length L; scheme < 0 = chosen by cost
MIRGE_HD int mirge_probe_count(const MirgePolicy& p, int L, int K, uint64_t npos, int scheme = -1) {
    const int S = mirge_seed_region(p, L);
    K = mirge_policy_k(p, K, npos);
    return mirge_scheme_count(p, scheme >= 0 ? scheme : mirge_plan_scheme(p, S, K, npos));
}

MIRGE_HD void mirge_probe_at(const MirgePolicy& p, int L, int K, uint64_t npos, int q, MirgeProbe& out, int scheme = -1) {
    const int S = mirge_seed_region(p, L);
    K = mirge_policy_k(p, K, npos);
    mirge_scheme_probe(p, S, K, scheme >= 0 ? scheme : mirge_plan_scheme(p, S, K, npos), q, out);
}

// The plan depends only on (policy, K, trimmed length): the host tabulates it once per pass and the
// kernels read probe q of length L with one 4-byte load instead of redoing the integer divisions.
struct MirgePlanTable {
    uint8_t np[MIRGE_MAX_READ_LEN + 1];
    MirgeProbe pr[MIRGE_MAX_READ_LEN + 1][MIRGE_MAX_PROBES];
};
static inline void mirge_plan_table_fill(const MirgePolicy& p, int K, uint64_t npos, MirgePlanTable& t) {
    for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) {
        const bool ok = L >= 1 && L > p.mm;
        const int n = ok ? mirge_probe_count(p, L, K, npos) : 0;
        t.np[L] = (uint8_t)n;
        for (int q = 0; q < MIRGE_MAX_PROBES; q++) {
            MirgeProbe pr; pr.a1 = 0; pr.k1 = 0; pr.gap = 0; pr.k2 = 0;
            if (q < n) mirge_probe_at(p, L, K, npos, q, pr);
            t.pr[L][q] = pr;
        }
    }
}

// key of a probe in the read, or false if an ambiguous call sits in a block (cannot be exact)
template <int W>
MIRGE_HD bool mirge_probe_key(const MirgeRead<W>& r, const MirgeProbe& pr, uint64_t& key) {
    if (pr.k1 <= 0) return false;
    if (mirge_extract<W>(r.nm, pr.a1, pr.k1)) return false;
    key = mirge_extract<W>(r.w, pr.a1, pr.k1);
    if (pr.k2 > 0) {
        const int b = pr.a1 + pr.k1 + pr.gap;
        if (mirge_extract<W>(r.nm, b, pr.k2)) return false;
        key |= mirge_extract<W>(r.w, b, pr.k2) << (2 * pr.k1);
    }
    return true;
}

// Best alignment of an (already trimmed) read in one library through the probe tables: every
// candidate window of every probe is verified in full.  best = (total mismatches << 32) | global
// position, minimised.  Serial reference form (tests/hostsim); the kernels use align_hybrid.
template <int W>
MIRGE_HD bool mirge_align_indexed(const MirgeLibView& lib, const MirgePolicy& p,
                                  const MirgeRead<W>& r, uint64_t& best) {
    best = MIRGE_NO_HIT;
    const int L = r.len;
    const int np = mirge_probe_count(p, L, lib.kmax, lib.total);
    for (int q = 0; q < np; q++) {
        MirgeProbe pr;
        mirge_probe_at(p, L, lib.kmax, lib.total, q, pr);
        uint64_t key;
        if (!mirge_probe_key<W>(r, pr, key)) continue;
        const MirgeKTable tb = lib.tables[mirge_shape_id(pr.k1, pr.gap, pr.k2)];
        const int a = pr.a1;
        if (tb.bits && !((tb.bits[key >> 5] >> (key & 31)) & 1u)) continue;
        uint32_t cnt, lo;
        bool inl = false;
        if (tb.bits) {
            const uint32_t* b = static_cast<const uint32_t*>(tb.bucket);
            lo = b[key]; cnt = b[key + 1] - lo;
        } else {
            const uint64_t e = static_cast<const uint64_t*>(tb.bucket)[key];
            cnt = (uint32_t)(e >> 32); lo = (uint32_t)e; inl = cnt == 1;
        }
        for (uint32_t c = 0; c < cnt; c++) {
            const uint32_t pz = inl ? lo : tb.pos[lo + c];
            if (pz < (uint32_t)a) continue;
            const uint64_t g = (uint64_t)pz - (uint64_t)a;
            const int m = mirge_window_mm<W>(lib.T, g, r, p);
            if (m < 0) continue;
            if (mirge_window_invalid(lib.inv, g, L)) continue;
            const uint64_t cand = ((uint64_t)m << 32) | g;
            if (cand < best) best = cand;
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
