// kernels_cascade.hpp -- part of mirge_kernels.hpp: alignment (probe, verify), k_pass, k_cascade_fused, k_resolve.
#pragma once
// ------------------------------------------------------------------------------------------
// align_hybrid: the device form of mirge_align_indexed (same probes, same verification, same
// minimum) with the candidate lists balanced over the wave.
//   A probe's bucket holds from 0 to thousands of candidate windows (a 16-nt read under -v 2 is
//   probed with 4-mers: ~220 candidates per probe in a 57 kb library, while a 28-nt read sees ~1).
//   Lane-serial evaluation makes the whole wave wait for its unluckiest lane and walks each list
//   as a chain of dependent loads.  Here a lane verifies only short lists (<= MIRGE_LIGHT) itself;
//   longer lists are taken one at a time by the whole wave: the owner's read is broadcast with
//   v_readlane, the 64 lanes stride through the bucket (coalesced pos[] loads, 64 windows
//   verified per step); the few lanes that found a valid window are read back with v_readlane and
//   their minimum goes to the owner.
//   All 64 lanes of the wave must call this together (inactive lanes pass active = false).
// ------------------------------------------------------------------------------------------
#ifndef MIRGE_LIGHT
#define MIRGE_LIGHT 4
#endif
#ifndef MIRGE_LIGHT_MAX
#define MIRGE_LIGHT_MAX 16
#endif
#define MIRGE_COOP_UNROLL 1
#ifndef MIRGE_SORTED_EXIT
#define MIRGE_SORTED_EXIT 1   // heavy buckets are position-sorted at build time (native_lib.hpp): scans may stop at a 0-mismatch hit
#endif
#ifndef MIRGE_ADAPTIVE_LIGHT
#define MIRGE_ADAPTIVE_LIGHT 1  // many lanes with long lists (Alu-like families): the lanes walk their own lists side by side
#endif
#ifndef MIRGE_MIN_BUCKET
#define MIRGE_MIN_BUCKET 1    // exact-seed policies: a read whose probe bucket is heavy takes the rarest k-mer of its seed instead
#endif
// k_cascade_heavy: a workgroup of 1024 threads per read, eight windows per thread in flight: a bucket of 2 M windows is 256 trips
#define MIRGE_HEAVY_THREADS 1024
#define MIRGE_HEAVY_UNROLL 8
#define MIRGE_HEAVY_RETRY 64  // windows in an exact-seed probe's bucket from which the other k-mers of the seed are asked
#ifndef MIRGE_LDS_PLAN
#define MIRGE_LDS_PLAN 1
#endif

// pointers that came out of memory or a v_readlane have lost their address space; these casts keep
// the loads global_load_* (not flat_load_*, which also ties up lgkmcnt)
typedef const __attribute__((address_space(1))) uint32_t* gptr_u32;
typedef const __attribute__((address_space(1))) uint64_t* gptr_u64;

#ifndef MIRGE_PRESENCE_FILTER
// Round 6 experiment (round 5's review, item 5): a 1-bit "bucket is non-empty" filter in front of the self-contained 8-byte entries
// of the large tables too (k > 10: human mRNA k = 15: 128 MB, the ncRNA shapes 2-32 MB each), so that a probe nothing answers -- most
// probes of an unmappable read -- costs a bit out of the Infinity Cache instead of a random HBM sector.  The table's `bits` pointer
// carries bit 0 as the mark "entries behind a filter".  0: off (what is shipped unless profiles/README.md round 6 says otherwise).
#define MIRGE_PRESENCE_FILTER 0
#endif
// the filter / bitmap pointer of a table without its mark, and whether the table holds self-contained entries
__device__ __forceinline__ gptr_u32 table_bits(const MirgeKTable& tb, bool& entries) {
#if MIRGE_PRESENCE_FILTER
    const uint64_t raw = (uint64_t)tb.bits;
    entries = raw == 0ull || (raw & 1ull);
    return (gptr_u32)(raw & ~3ull);
#else
    entries = tb.bits == nullptr;
    return (gptr_u32)tb.bits;
#endif
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, src);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// Consecutive passes with one and the same policy (snoRNA / rRNA / other ncRNA: all "-n 1") run as ONE
// pass over the concatenation of their libraries.  The cascade's priority -- the first library with
// ANY valid hit wins, whatever its mismatch count -- is kept by ranking candidates on
// (library class, mismatches, position): class = which member's text the window lies in.
struct MergeInfo {
    uint32_t bound[4];  // global position where member c's text starts (bound[0] = 0)
    int32_t n;          // members (1 = ordinary pass)
};
__device__ __forceinline__ uint64_t class_key(const MergeInfo& mi, uint64_t g) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 1; i < 4; i++) c += (i < mi.n && g >= mi.bound[i]) ? 1u : 0u;
    return c << 40;
}

// the two text words under a window: issued for several candidates before any is consumed
template <int W>
struct TextWin { uint64_t w[W + 1]; };

// Two adjacent 8-byte words fetched as ONE 16-byte access (4-byte alignment is all a global load needs): the passes are
// bound by the number of vector-memory instructions and cache accesses a CU's texture addresser can take (TA busy 56-69 %
// of the kernel time in every k_pass, profiles/README.md round 2), so paired loads are merged wherever two words sit
// side by side -- the window's text, its invalid bits, a bucket's two CSR bounds.
typedef uint32_t QuadU32 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t DuoU32 __attribute__((ext_vector_type(2), aligned(4)));
typedef const __attribute__((address_space(1))) QuadU32* gptr_quad32;
typedef const __attribute__((address_space(1))) DuoU32* gptr_duo32;
struct PairU64 { uint64_t a, b; };
struct PairU32 { uint32_t a, b; };
__device__ __forceinline__ PairU64 load_pair64(gptr_u64 p) {
    const QuadU32 v = *(gptr_quad32)p;
    return PairU64{(uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32)};
}
__device__ __forceinline__ PairU32 load_pair32(gptr_u32 p) {
    const DuoU32 v = *(gptr_duo32)p;
    return PairU32{v.x, v.y};
}

template <int W>
__device__ __forceinline__ void load_window(gptr_u64 T, uint64_t g, int L, TextWin<W>& tw) {
    const uint64_t q = g >> 5;
    if (W == 1) {
        const PairU64 p = load_pair64(T + q);
        tw.w[0] = p.a; tw.w[1] = p.b;
        return;
    }
#pragma unroll
    for (int i = 0; i <= W; i++) tw.w[i] = (i == 0 || 32 * (i - 1) < L) ? T[q + i] : 0ull;
}

// mirge_window_invalid for windows of up to 64 bases (W == 1: <= 31): the two bitmap words in one access
__device__ __forceinline__ bool window_invalid_pair(gptr_u64 inv, uint64_t g, int L) {
    const PairU64 p = load_pair64(inv + (g >> 6));
    const int s = (int)(g & 63);
    uint64_t lo = p.a >> s;
    if (s) lo |= p.b << (64 - s);
    if (L < 64) lo &= (1ull << L) - 1ull;
    return lo != 0ull;
}

// same arithmetic as mirge_window_mm, on words that are already in registers
template <int W>
__device__ __forceinline__ int window_mm_regs(const TextWin<W>& tw, uint64_t g, const MirgeRead<W>& r,
                                              const MirgePolicy& p) {
    const int L = r.len;
    const int s = (int)(g & 31) * 2;
    int tot = 0, seedmm = 0;
    const int seed = p.mode == 0 ? (L < p.seedlen ? L : p.seedlen) : L;
#pragma unroll
    for (int i = 0; i < W; i++) {
        if (32 * i < L) {
            const uint64_t t = s ? ((tw.w[i] >> s) | (tw.w[i + 1] << (64 - s))) : tw.w[i];
            const uint64_t x = r.w[i] ^ t;
            uint64_t m = (x | (x >> 1)) & 0x5555555555555555ull;
            const int rem = L - 32 * i;
            m &= mirge_lowmask2(rem > 32 ? 32 : rem);
            m |= r.nm[i];
            tot += mirge_popc(m);
            const int srem = seed - 32 * i;
            if (srem > 0) seedmm += mirge_popc(m & mirge_lowmask2(srem > 32 ? 32 : srem));
        }
    }
    if (tot > p.maxtotal || seedmm > p.mm) return -1;
    return tot;
}

// verify up to N candidate positions at once: all pos loads, then all text loads, then arithmetic
// (pos == nullptr: the bucket held one window and c[0] IS its position)
template <int W, int N>
__device__ __forceinline__ uint64_t eval_batch(const MirgeLibView& lib, const MirgePolicy& pol, const MergeInfo& mi,
                                               const MirgeRead<W>& r, gptr_u32 pos, const uint32_t (&c)[N],
                                               uint32_t hi, int a, uint64_t known = MIRGE_NO_HIT) {
    uint32_t pz[N];
    bool ok[N];
#pragma unroll
    for (int u = 0; u < N; u++) {
        ok[u] = c[u] < hi;
        pz[u] = ok[u] ? (pos ? pos[c[u]] : c[u]) : 0u;
    }
    TextWin<W> tw[N];
    uint64_t g[N];
#pragma unroll
    for (int u = 0; u < N; u++) {
        // the window the lane already holds as its best is found again by every probe whose blocks miss its mismatches:
        // its text is not fetched a second time (the low word of NO_HIT is no position)
        ok[u] = ok[u] && pz[u] >= (uint32_t)a && (pz[u] - (uint32_t)a) != (uint32_t)known;
        g[u] = ok[u] ? (uint64_t)pz[u] - (uint64_t)a : 0ull;
        load_window<W>((gptr_u64)lib.T, g[u], ok[u] ? r.len : 0, tw[u]);
    }
    uint64_t best = MIRGE_NO_HIT;
#pragma unroll
    for (int u = 0; u < N; u++) {
        if (!ok[u]) continue;
        const int m = window_mm_regs<W>(tw[u], g[u], r, pol);
        if (m < 0) continue;
        if (W == 1 ? window_invalid_pair((gptr_u64)lib.inv, g[u], r.len) : mirge_window_invalid(lib.inv, g[u], r.len)) continue;
        const uint64_t cand = class_key(mi, g[u]) | ((uint64_t)m << 32) | g[u];
        if (cand < best) best = cand;
    }
    return best;
}

// Where a lane finds probe q of a read of length L and the table it addresses.  The plan and the table registry
// live in global memory; every lane of a wave asks for a different (L, q) entry, so each probe used to cost three
// uncoalesced lookups (plan entry, 24-byte table descriptor) before the first useful one -- about half of all the
// lane-level cache lookups of the passes over the small libraries, which is what bounds them (per-CU L1 tag rate, not
// latency and not VALU: profiles/README.md round 2).  For the one-word read group (trimmed length <= 31) a workgroup
// therefore copies the (L, q) -> {probe, table} map into LDS once (8 KiB) and the lanes read it from there.
struct LdsPlan {
    MirgeKTable tb[32][MIRGE_MAX_PROBES];
    MirgeProbe pr[32][MIRGE_MAX_PROBES];
    uint8_t np[32];
};
template <bool LDS>
struct PlanSrc {
    const MirgePlanTable* g;
    const LdsPlan* l;
};

__device__ __forceinline__ void lds_plan_fill(LdsPlan& s, const MirgeLibView& lib, const MirgePlanTable* __restrict__ plan) {
    for (int idx = threadIdx.x; idx < 32 * MIRGE_MAX_PROBES; idx += blockDim.x) {
        const int L = idx / MIRGE_MAX_PROBES, q = idx % MIRGE_MAX_PROBES;
        const MirgeProbe pr = plan->pr[L][q];
        MirgeKTable tb;
        tb.bucket = nullptr; tb.pos = nullptr; tb.bits = nullptr;
        if (q < (int)plan->np[L] && pr.k1 > 0) tb = lib.tables[mirge_shape_id(pr.k1, pr.gap, pr.k2)];
        s.pr[L][q] = pr;
        s.tb[L][q] = tb;
    }
    for (int L = threadIdx.x; L < 32; L += blockDim.x) s.np[L] = plan->np[L];
}

// probe q of this lane's read: key and table; false = no such probe / an ambiguous call inside it
template <int W, bool LDS>
__device__ __forceinline__ bool probe_setup(const MirgeLibView& lib, const PlanSrc<LDS>& ps, const MirgeRead<W>& r,
                                            int q, int np, bool active, MirgeProbe& pr, MirgeKTable& tb, uint64_t& key) {
    if (!(active && q < np)) return false;
    if (LDS) pr = ps.l->pr[r.len][q];
    else pr = ps.g->pr[r.len][q];  // tabulated mirge_probe_at(pol, len, K, q)
    if (!mirge_probe_key<W>(r, pr, key)) return false;
    if (LDS) tb = ps.l->tb[r.len][q];
    else tb = lib.tables[mirge_shape_id(pr.k1, pr.gap, pr.k2)];
    return true;
}

// The k-mer of the read's seed region [0, S) with the fewest windows in table tb, if one has fewer than `have` (the count of the
// k-mer at offset 0): {lo, hi, its offset a > 0, inl = the bucket holds one window and lo IS its position}; a = 0: none.  A call,
// not inlined: the branch is rare (a read out of a repeat), and inlined its temporaries cost k_cascade_bulk seven more spilled
// vector registers on every read's path.
struct RareKmer { uint32_t lo, hi; int32_t a; uint32_t inl; };
template <int W>
__device__ __attribute__((noinline)) RareKmer rarest_kmer(MirgeKTable tb, MirgeRead<W> r, int S, int k, uint32_t have) {
    RareKmer out{0u, 0u, 0, 0u};
    bool entries;
    gptr_u32 bits = table_bits(tb, entries);
    for (int a2 = 1; a2 + k <= S; a2++) {
        if (mirge_extract<W>(r.nm, a2, k)) continue;
        const uint64_t key2 = mirge_extract<W>(r.w, a2, k);
        uint32_t lo2 = 0, hi2 = 0, inl = 0;
        if (!bits || ((bits[key2 >> 5] >> (key2 & 31)) & 1u)) {
            if (entries) {
                const uint64_t e = ((gptr_u64)tb.bucket)[key2];
                lo2 = (uint32_t)e; hi2 = lo2 + (uint32_t)(e >> 32);
                inl = (uint32_t)(e >> 32) == 1u ? 1u : 0u;
            } else {
                const PairU32 bd = load_pair32((gptr_u32)tb.bucket + key2);
                lo2 = bd.a; hi2 = bd.b;
            }
        }
        if (hi2 - lo2 < have) { have = hi2 - lo2; out.lo = lo2; out.hi = hi2; out.a = a2; out.inl = inl; }
        if (have < MIRGE_HEAVY_RETRY) break;
    }
    return out;
}

// long candidate lists (repeats, poly-A), one at a time by the whole wave: the owner's read is broadcast, the 64 lanes
// stride through the bucket (coalesced pos[] loads), the few lanes that found a valid window are read back
template <int W, bool REP>
__device__ __forceinline__ void verify_heavy(const MirgeLibView& lib, const MirgePolicy& pol, const MergeInfo& mi, const MirgeRead<W>& r,
                                             gptr_u32 pos, uint32_t lo, uint32_t hi, int a, bool heavy, uint64_t& best) {
    const int lane = threadIdx.x & 63;
    unsigned long long hb = __ballot(heavy);
    while (hb) {
        const int src = __ffsll(hb) - 1;
        hb &= hb - 1;
        MirgeRead<W> rr;
#pragma unroll
        for (int w = 0; w < W; w++) {
            rr.w[w] = readlane_u64(r.w[w], src);
            rr.nm[w] = readlane_u64(r.nm[w], src);
        }
        rr.len = __builtin_amdgcn_readlane(r.len, src);
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)lo, src);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)hi, src);
        const int ba = __builtin_amdgcn_readlane(a, src);
        gptr_u32 bpos = (gptr_u32)readlane_u64((uint64_t)pos, src);
        uint64_t lbest = MIRGE_NO_HIT;
        for (uint32_t c0 = blo + lane; c0 - lane < bhi; c0 += 64 * MIRGE_COOP_UNROLL) {  // (wave-uniform trip count: the ballot below)
            uint32_t c[MIRGE_COOP_UNROLL];
#pragma unroll
            for (int u = 0; u < MIRGE_COOP_UNROLL; u++) c[u] = c0 + 64 * u;
            const uint64_t cand = eval_batch<W, MIRGE_COOP_UNROLL>(lib, pol, mi, rr, bpos, c, bhi, ba);
            if (cand < lbest) lbest = cand;
            if constexpr (REP && MIRGE_SORTED_EXIT) {
            // (round 6) the list ascends (k_table_heavy_list + the segmented sort): candidates rank by (class, mismatches, position),
            // class and position only grow from here, so a window without a mismatch is final -- the scan of a poly-A bucket of 10^6
            // windows ends at the first tail that holds the read
            if (__ballot(cand != MIRGE_NO_HIT && ((cand >> 32) & 0xFFu) == 0u)) break;
            }
        }
        // almost every candidate fails verification: instead of a shuffle tree, visit the few
        // lanes that hold a hit (v_readlane -> scalar min)
        unsigned long long hits = __ballot(lbest != MIRGE_NO_HIT);
        uint64_t tbest = MIRGE_NO_HIT;
        while (hits) {
            const int hl = __ffsll(hits) - 1;
            hits &= hits - 1;
            const uint64_t v = readlane_u64(lbest, hl);
            if (v < tbest) tbest = v;
        }
        if (lane == src && tbest < best) best = tbest;
    }
}

// the candidate list [lo, hi) of one probe, verified: short lists by their own lane, long ones by the whole wave
template <int W, bool REP>
__device__ __forceinline__ void verify_lists(const MirgeLibView& lib, const MirgePolicy& pol, const MergeInfo& mi, const MirgeRead<W>& r,
                                             gptr_u32 pos, uint32_t lo, uint32_t hi, int a, uint64_t& best) {
    // lists of up to MIRGE_LIGHT_MAX windows are verified by their own lane, MIRGE_LIGHT per batch
    // (a cooperative hand-over costs the whole wave ~150 instructions per list; with 5-15 windows
    // per list and many such lanes per probe the lane-serial batches are several times cheaper)
    bool heavy = (hi - lo) > MIRGE_LIGHT_MAX;
    if constexpr (REP && MIRGE_ADAPTIVE_LIGHT) {
    // (round 6) A hand-over serves ONE lane's list at a time: 64 lanes that each hold a list of 300 windows (reads out of an
    // Alu-like family with 800 diverged copies) cost the wave 64 x 5 trips, while the lanes walking their own lists side by side
    // cost it 300 / MIRGE_LIGHT = 75.  So: with the wave's heavy lists summing to n windows the cooperative route takes n / 64 trips,
    // and a lane whose own list is no longer than that (times MIRGE_LIGHT) may as well walk it itself; the truly long lists -- a
    // poly-A bucket -- stay with the whole wave.
    const unsigned long long hb0 = __ballot(heavy);
    if (hb0 & (hb0 - 1)) {  // two heavy lanes or more
        uint32_t tot = heavy ? hi - lo : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += (uint32_t)__shfl_xor((int)tot, o, 64);
        if (heavy && (hi - lo) <= (tot >> 4)) heavy = false;
    }
    }
    if (!heavy) {
        for (uint32_t c0 = lo; c0 < hi; c0 += MIRGE_LIGHT) {
            uint32_t c[MIRGE_LIGHT];
#pragma unroll
            for (int u = 0; u < MIRGE_LIGHT; u++) c[u] = c0 + u;
            const uint64_t cand = eval_batch<W, MIRGE_LIGHT>(lib, pol, mi, r, pos, c, hi, a, best);
            if (cand < best) best = cand;
            if constexpr (REP && MIRGE_SORTED_EXIT)
                if (hi - lo > MIRGE_LIGHT_MAX && cand != MIRGE_NO_HIT && ((cand >> 32) & 0xFFu) == 0u) break;  // (a sorted list: see verify_heavy)
        }
    }
    verify_heavy<W, REP>(lib, pol, mi, r, pos, lo, hi, a, heavy, best);
}

// REP (round 6): the build for configurations whose libraries repeat themselves (an outlier bucket: mirge_lib::max_bucket) -- sorted
// exits, the lane-serial walk of many heavy lists, the rarest k-mer of an exact seed, deferral to k_cascade_heavy.  REP = false is the
// kernel as it was for libraries that hold no such bucket: each of those four costs the uniform-library step 0.5-0.8 % in registers
// and issue slots for nothing it could gain (profiles/r06_ab_rep_features.txt).
template <int W, bool LDS, bool REP>
__device__ __forceinline__ void align_hybrid(const MirgeLibView& lib, const MirgePolicy& pol, const MergeInfo& mi,
                                             const PlanSrc<LDS>& ps, const MirgeRead<W>& r, bool active, uint64_t& best) {
    best = MIRGE_NO_HIT;
#if defined(MIRGE_EXP_SKIP)  // timing experiments only (wrong answers): no alignment at all = what walking the lists costs (profiles/README.md, round 4)
    return;
#endif
    const int np = active ? (int)(LDS ? ps.l->np[r.len] : ps.g->np[r.len]) : 0;
    // wave-uniform bound on the probe count: (mm+1) plain segments or (mm+1)^2 recursive probes
    const int npmax = (pol.mm >= 1 && pol.mm <= 2) ? (pol.mm + 1) * (pol.mm + 1) : pol.mm + 1;
    // (round 6) pol.reserved > 0: a read that meets a bucket of more windows than that is not aligned by this wave -- one such list
    // (a poly-A bucket of the mRNA library: 2 M windows) kept ONE wave busy for 20 ms while the chip idled.  It is answered
    // MIRGE_DEFER and taken by k_cascade_heavy: a whole workgroup per read, its lists strided by 256 threads.
    const uint32_t big_t = REP ? (uint32_t)pol.reserved : 0u;
    bool defer = false;
#pragma unroll 1
    for (int q = 0; q < npmax; q++) {
        uint32_t lo = 0, hi = 0;
        int a = 0;
        gptr_u32 pos = nullptr;
        MirgeProbe pr;
        MirgeKTable tb;
        uint64_t key;
        if (probe_setup<W, LDS>(lib, ps, r, q, np, active, pr, tb, key)) {  // no ambiguous call inside the probe
#if MIRGE_PRESENCE_FILTER
            bool entries;
            gptr_u32 bits = table_bits(tb, entries);
            if (!bits || ((bits[key >> 5] >> (key & 31)) & 1u)) {
                if (entries) {  // large table: one self-contained entry, a single window inline
                    const uint64_t e = ((gptr_u64)tb.bucket)[key];
                    const uint32_t cnt = (uint32_t)(e >> 32);
                    lo = (uint32_t)e;
                    hi = lo + cnt;
                    pos = cnt == 1 ? nullptr : (gptr_u32)tb.pos;  // one window: `lo` is its position
                } else {  // small table: CSR bounds behind its L2-resident "bucket is non-empty" bit
                    const PairU32 bd = load_pair32((gptr_u32)tb.bucket + key);
                    pos = (gptr_u32)tb.pos;
                    lo = bd.a;
                    hi = bd.b;
                }
            }
#else
            gptr_u32 bits = (gptr_u32)tb.bits;
            if (!bits) {  // large table: one self-contained entry, a single window inline
                const uint64_t e = ((gptr_u64)tb.bucket)[key];
                const uint32_t cnt = (uint32_t)(e >> 32);
                lo = (uint32_t)e;
                hi = lo + cnt;
                pos = cnt == 1 ? nullptr : (gptr_u32)tb.pos;  // one window: `lo` is its position
            } else if ((bits[key >> 5] >> (key & 31)) & 1u) {  // small table: L2-resident "bucket is non-empty" bit, then CSR bounds
                const PairU32 bd = load_pair32((gptr_u32)tb.bucket + key);
                pos = (gptr_u32)tb.pos;
                lo = bd.a;
                hi = bd.b;
            }
#endif
            a = pr.a1;
            if constexpr (REP && MIRGE_MIN_BUCKET) {
            // (round 6) An exact-seed policy (mm = 0: mRNA, spike-in) admits no mismatch inside the seed, so ANY k-mer of the seed
            // filters completely -- the plan takes the first.  A read out of a repeat with a sequencing error (poly-A with one G) whose
            // first k-mer misses the error lands in a bucket of 10^5 .. 10^6 windows none of which can verify; a k-mer that holds the
            // error is rare.  So a read whose bucket is heavy asks the other k-mers of its seed for their counts and takes the rarest.
            if (pol.mm == 0 && pr.k2 == 0 && (hi - lo) >= MIRGE_HEAVY_RETRY) {
                const RareKmer rk = rarest_kmer<W>(tb, r, mirge_seed_region(pol, r.len), pr.k1, hi - lo);
                if (rk.a > 0) { lo = rk.lo; hi = rk.hi; a = rk.a; pos = rk.inl ? nullptr : (gptr_u32)tb.pos; }
            }
            }
            if (REP && big_t && (hi - lo) > big_t) { defer = true; active = false; lo = hi = 0; }  // (its other probes need not be looked at either)
        }
        verify_lists<W, REP>(lib, pol, mi, r, pos, lo, hi, a, best);
        // a 0-mismatch window (of the first member library) is in probe 0's bucket and buckets ascend:
        // nothing later can beat it
        if (q == 0 && (best >> 32) == 0) active = false;
    }
    if (REP && defer) best = MIRGE_DEFER;
}

// ------------------------------------------------------------------------------------------
// k_cascade_heavy (round 6): the reads align_hybrid answered MIRGE_DEFER for -- one of their probe buckets holds more windows
// than a wave should walk alone (poly-A, Alu-like and simple-repeat buckets of real libraries: 10^4 .. 10^6 windows) -- with ONE
// WORKGROUP per read: every step of the cascade from the one that deferred it on (its mark names the pass; the passes before it found
// nothing), every candidate list strided by all 1 024 threads, four windows per thread in flight, the scan of a position-sorted list ended by the first window without a mismatch.  Same probes, same
// verification, same minimum as align_hybrid: same answers.  heavy_cnt[0] = reads in heavy_list (appended by the cascade kernels
// of this group), heavy_cnt[1] = workgroups done: the last one resets both for the next launch.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wg_min_u64(uint64_t v, unsigned long long* s_best) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t t = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        if (t < v) v = t;
    }
    if (threadIdx.x == 0) *s_best = MIRGE_NO_HIT;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicMin(s_best, (unsigned long long)v);
    __syncthreads();
    const uint64_t m = *s_best;
    __syncthreads();  // (everybody has read it before the next reduction resets it)
    return m;
}

template <int W>
__device__ __forceinline__ uint64_t align_wg(const MirgeLibView& lib, const MirgePolicy& pol, const MergeInfo& mi,
                                             const MirgePlanTable* __restrict__ plan, const MirgeRead<W>& r, unsigned long long* s_best) {
    uint64_t best = MIRGE_NO_HIT;  // this thread's; the workgroup's minimum at the end
    const int np = (int)plan->np[r.len];
    for (int q = 0; q < np; q++) {  // everything up to the scan is uniform over the workgroup: one read
        const MirgeProbe pr = plan->pr[r.len][q];
        uint64_t key;
        if (!mirge_probe_key<W>(r, pr, key)) continue;
        const MirgeKTable tb = lib.tables[mirge_shape_id(pr.k1, pr.gap, pr.k2)];
        bool entries;
        gptr_u32 bits = table_bits(tb, entries);
        uint32_t lo = 0, hi = 0;
        gptr_u32 pos = nullptr;
        if (!bits || ((bits[key >> 5] >> (key & 31)) & 1u)) {
            if (entries) {
                const uint64_t e = ((gptr_u64)tb.bucket)[key];
                const uint32_t cnt = (uint32_t)(e >> 32);
                lo = (uint32_t)e; hi = lo + cnt;
                pos = cnt == 1 ? nullptr : (gptr_u32)tb.pos;
            } else {
                const PairU32 bd = load_pair32((gptr_u32)tb.bucket + key);
                pos = (gptr_u32)tb.pos; lo = bd.a; hi = bd.b;
            }
        }
        int a = pr.a1;
#if MIRGE_MIN_BUCKET
        if (pol.mm == 0 && pr.k2 == 0 && (hi - lo) >= MIRGE_HEAVY_RETRY) {
            const RareKmer rk = rarest_kmer<W>(tb, r, mirge_seed_region(pol, r.len), pr.k1, hi - lo);
            if (rk.a > 0) { lo = rk.lo; hi = rk.hi; a = rk.a; pos = rk.inl ? nullptr : (gptr_u32)tb.pos; }
        }
#endif
        const bool sorted = (hi - lo) > MIRGE_LIGHT_MAX;  // (what k_table_heavy_list lists)
        // the exit is looked for every eighth trip only: a barrier per trip made every wave wait for the slowest wave's loads, trip after
        // trip -- a list of 2 M windows was 512 trips of 4.7 us; between two checks the waves run ahead of each other
        bool seen_exact = false;
        uint32_t trip = 0;
        for (uint32_t c0 = lo + threadIdx.x; c0 - threadIdx.x < hi; c0 += MIRGE_HEAVY_UNROLL * MIRGE_HEAVY_THREADS) {
            uint32_t c[MIRGE_HEAVY_UNROLL];
#pragma unroll
            for (int u = 0; u < MIRGE_HEAVY_UNROLL; u++) c[u] = c0 + u * MIRGE_HEAVY_THREADS;
            const uint64_t cand = eval_batch<W, MIRGE_HEAVY_UNROLL>(lib, pol, mi, r, pos, c, hi, a);
            if (cand < best) best = cand;
            seen_exact |= cand != MIRGE_NO_HIT && ((cand >> 32) & 0xFFu) == 0u;
            if (MIRGE_SORTED_EXIT && sorted && (++trip & 7u) == 0u && __syncthreads_or(seen_exact)) break;
        }
        // a 0-mismatch window of the first member library is in probe 0's bucket: nothing later can beat it
        if (q == 0 && np > 1 && (wg_min_u64(best, s_best) >> 32) == 0) break;
    }
    return wg_min_u64(best, s_best);
}

// a deferred read: marked (the heavy kernel overwrites the mark), listed
// (the mark names the pass that deferred the read, -2 - pass: the passes before it found nothing and k_cascade_heavy resumes there)
__device__ __forceinline__ void defer_read(uint32_t idx, uint32_t* __restrict__ heavy_cnt, uint32_t* __restrict__ heavy_list,
                                           int8_t* __restrict__ res_pass, int8_t* __restrict__ res_mm, int32_t pass_id) {
    res_pass[idx] = (int8_t)(-2 - pass_id); res_mm[idx] = -1;
    heavy_list[atomicAdd(&heavy_cnt[0], 1u)] = idx;
}

// ------------------------------------------------------------------------------------------
// k_pass: one cascade pass over the still-unannotated reads of one width group.
//   Every workgroup owns a fixed segment of `cap` slots.  First pass (act_in == nullptr):
//   workgroup b takes the contiguous reads [b*cap, (b+1)*cap).  Later passes: workgroup b takes
//   the seg_n_in[b] survivors its own previous pass left in act_in[b*cap ...].
//   A hit writes (pass, global position, mismatches) at the read's slot; every other read (not
//   selected by the pass's subset rule, skipped by bowtie, or unaligned) is appended to the
//   workgroup's segment of act_out, so the next pass sees exactly the rows with annotFlag == 0
//   (manifoldAlign.py:120,129).  The append needs no global atomic: one LDS counter per
//   workgroup, one ds_add per wave (ballot + prefix popcount).  A single global cursor was
//   measured at ~0.35 ms per pass for 2 M reads (33 k same-address returning atomics).
// ------------------------------------------------------------------------------------------
// SLOT is the pass index and only names the symbol (k_pass<1,6> ...), so that rocprofv3's per-kernel
// statistics separate the passes; the policy itself stays a run-time argument.
#ifndef MIRGE_PASS_MIN_WAVES
#define MIRGE_PASS_MIN_WAVES 1
#endif
#ifdef MIRGE_PASS_SGPRS
#define MIRGE_PASS_ATTR __attribute__((amdgpu_num_sgpr(MIRGE_PASS_SGPRS)))
#else
#define MIRGE_PASS_ATTR
#endif
// one pass of one workgroup over its n_in reads: reads [seg_r, seg_r + n_in) of the group (FIRST pass, act_in == nullptr) or
// the survivors act_in[seg ...] its previous pass left; survivors go to act_out[seg ...], counted in s_count (LDS, reset by the
// caller).  COHERENT: act_in was written by this workgroup earlier in the SAME kernel (k_cascade_bulk): the loads go to L2
// (agent scope) instead of a vector L1 that may still hold the lines of two passes ago.
template <int W, bool LDSP, bool COHERENT, bool HASN = true>
__device__ __forceinline__ void pass_segment(const MirgeLibView& lib, const MirgePolicy& pol, const MergeInfo& mi, const PlanSrc<LDSP>& psrc,
                                             const GroupView<W>& g, const uint32_t* act_in, uint32_t n_in, size_t seg, size_t seg_r,
                                             uint32_t* __restrict__ act_out, int32_t pass_id, int8_t* __restrict__ res_pass,
                                             uint32_t* __restrict__ res_pos, int8_t* __restrict__ res_mm, uint32_t* s_count,
                                             uint32_t* __restrict__ heavy_cnt, uint32_t* __restrict__ heavy_list) {
    const int lane = threadIdx.x & 63;
    for (uint32_t base = 0; base < n_in; base += MIRGE_BLOCK) {
        const uint32_t t = base + threadIdx.x;
        const bool valid = t < n_in;
        bool survivor = false;
        uint32_t idx = 0;
        if (valid) {
            if (!act_in) idx = (uint32_t)seg_r + t;
            else idx = COHERENT ? __hip_atomic_load(&act_in[seg + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : act_in[seg + t];
            survivor = true;
        }
        // the wave aligns its 64 reads together (align_hybrid balances the candidate lists)
        MirgeRead<W> r2;
        bool elig = false;
        if (valid) {
            load_read<W, HASN>(g, idx, r2);
            elig = mirge_effective_read<W>(r2, pol);
        } else {
#pragma unroll
            for (int w = 0; w < W; w++) { r2.w[w] = 0; r2.nm[w] = 0; }
            r2.len = 0;
        }
        uint64_t best;
        align_hybrid<W, LDSP, true>(lib, pol, mi, psrc, r2, elig, best);  // (the staged route: the repeat-aware build always)
        if (elig && best == MIRGE_DEFER) {  // a bucket no wave should walk alone: k_cascade_heavy answers for this read
            defer_read(idx, heavy_cnt, heavy_list, res_pass, res_mm, pass_id);
            survivor = false;
        } else
        if (elig && best != MIRGE_NO_HIT) {
            const int cls = (int)(best >> 40);  // member library of a merged pass (0 otherwise)
            res_pass[idx] = (int8_t)(pass_id + cls);
            uint32_t b0 = 0;
#pragma unroll
            for (int i = 1; i < 4; i++) if (i == cls) b0 = mi.bound[i];
            res_pos[idx] = (uint32_t)best - b0;  // position in that member's own text
            res_mm[idx] = (int8_t)((best >> 32) & 0xFF);
            survivor = false;
        } else if (valid && !act_in) {  // the first pass sees every read of the group: it also writes "unannotated"
            res_pass[idx] = -1;         // (two memsets per group and the queue gaps around them, before)
            res_mm[idx] = -1;
        }
        const unsigned long long bal = __ballot(survivor);
        if (bal) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(s_count, (uint32_t)__popcll(bal));
            wbase = __shfl(wbase, 0, 64);
            if (survivor) act_out[seg + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = idx;
        }
    }
}

// the first pass's share of workgroup b: n_dev: the read count is still on the device (a cascade enqueued behind the
// collapse that produces it, before the host has read U back): the U reads are cut into gridDim.x segments here; `cap`,
// sized from the raw read count, only spaces the survivor slices
__device__ __forceinline__ uint32_t first_pass_share(uint32_t n, const uint32_t* __restrict__ n_dev, uint32_t cap, size_t& seg_r) {
    const uint32_t ntot = n_dev ? *n_dev : n;
    uint32_t cap_r = cap;
    if (n_dev) cap_r = ((ntot + gridDim.x - 1) / gridDim.x + MIRGE_BLOCK - 1) / MIRGE_BLOCK * MIRGE_BLOCK;
    seg_r = (size_t)blockIdx.x * cap_r;
    return seg_r < ntot ? (uint32_t)((ntot - seg_r) < cap_r ? (ntot - seg_r) : cap_r) : 0u;
}

template <int W, int SLOT>
__global__ void __launch_bounds__(MIRGE_BLOCK, MIRGE_PASS_MIN_WAVES) MIRGE_PASS_ATTR
k_pass(MirgeLibView lib, MirgePolicy pol, MergeInfo mi, const MirgePlanTable* __restrict__ plan, GroupView<W> g,
       const uint32_t* __restrict__ act_in,
       const uint32_t* __restrict__ seg_n_in, uint32_t* __restrict__ act_out, uint32_t* __restrict__ seg_n_out,
       uint32_t cap, int32_t pass_id, int8_t* __restrict__ res_pass, uint32_t* __restrict__ res_pos,
       int8_t* __restrict__ res_mm, const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ heavy_cnt, uint32_t* __restrict__ heavy_list) {
    __shared__ uint32_t s_count;
    constexpr bool LDSP = (W == 1) && MIRGE_LDS_PLAN;
    __shared__ __attribute__((aligned(16))) unsigned char s_plan_raw[LDSP ? sizeof(LdsPlan) : 16];
    LdsPlan* s_plan = reinterpret_cast<LdsPlan*>(s_plan_raw);
    if (threadIdx.x == 0) s_count = 0;
    if (LDSP) lds_plan_fill(*s_plan, lib, plan);
    __syncthreads();
    PlanSrc<LDSP> psrc;
    psrc.g = plan; psrc.l = LDSP ? s_plan : nullptr;
    const size_t seg = (size_t)blockIdx.x * cap;  // the workgroup's slice of the survivor arrays
    size_t seg_r = seg;
    const uint32_t n_in = act_in ? seg_n_in[blockIdx.x] : first_pass_share(g.n, n_dev, cap, seg_r);
    pass_segment<W, LDSP, false>(lib, pol, mi, psrc, g, act_in, n_in, seg, seg_r, act_out, pass_id, res_pass, res_pos, res_mm, &s_count,
                                 heavy_cnt, heavy_list);
    __syncthreads();
    if (threadIdx.x == 0) seg_n_out[blockIdx.x] = s_count;
}

// global position -> (reference index, offset).  One 8-byte entry per granule of 16 positions answers it with ONE access:
//   word 0 = (last reference that starts at or before the granule's first position) << 5 | code
//   word 1 = that reference's start
//   code   = 1..15: exactly one other reference starts inside the granule, at that offset;  16: none;  17: several (only
//            references shorter than 16 nt, or empty ones, do that) -> a search through ref_start from word 0's reference on
// Round 2 had a reference index per 64 positions and then two or three probes of ref_start plus the start itself: four or
// five lane-level cache accesses per read, each a separate line -- k_resolve ran at the rate the texture path takes lines
// (60 us per 4.2 M reads; four reads per thread in flight changed nothing, so it was not latency).  0.5 B per reference
// position of device memory (T and the probe tables take ~10).
#define MIRGE_COARSE_SHIFT 4
struct ResolveTable {
    const uint32_t* ref_start[MIRGE_MAX_PASSES_K];
    const uint32_t* coarse[MIRGE_MAX_PASSES_K];  // [granules][2]
    uint32_t n_refs[MIRGE_MAX_PASSES_K];
};

// (reference, offset) of position g from its granule's entry c
__device__ __forceinline__ void resolve_entry(const PairU32 c, const uint32_t* rs, uint32_t nr, uint32_t g, int32_t& ref, int32_t& off) {
    uint32_t lo = c.a >> 5, base = c.b;
    const uint32_t code = c.a & 31u;
    if (code < 16u) {
        const uint32_t bpos = ((g >> MIRGE_COARSE_SHIFT) << MIRGE_COARSE_SHIFT) + code;
        if (g >= bpos) { lo++; base = bpos; }
    } else if (code == 17u) {
        uint32_t hi = nr;  // last t with rs[t] <= g
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (rs[mid] <= g) lo = mid; else hi = mid;
        }
        base = rs[lo];
    }
    ref = (int32_t)lo;
    off = (int32_t)(g - base);
}
__device__ __forceinline__ void resolve_one(const ResolveTable& tb, int p, uint32_t g, int32_t& ref, int32_t& off) {
    const uint32_t* rs = nullptr;
    const uint32_t* cs = nullptr;
    uint32_t nr = 0;
#pragma unroll
    for (int q = 0; q < MIRGE_MAX_PASSES_K; q++)
        if (q == p) { rs = tb.ref_start[q]; cs = tb.coarse[q]; nr = tb.n_refs[q]; }
    resolve_entry(load_pair32((gptr_u32)cs + 2 * (size_t)(g >> MIRGE_COARSE_SHIFT)), rs, nr, g, ref, off);
}

// ------------------------------------------------------------------------------------------
// k_cascade_bulk (round 3): ALL passes of the bulk read group in one launch.  A workgroup's segment and its survivor lists
// are its own from the first pass to the last -- nothing ever crosses workgroups -- so the kernel boundaries between the
// passes were barriers over the whole chip that nothing needed: every pass waited for its slowest workgroup, and the chip
// was in ONE regime at a time (the merged pass waits on random sectors with its ALUs idle, the isomiR pass issues VALU and
// cache accesses with the memory system idle).  Here a workgroup walks through the steps on its own; workgroups drift
// apart and the regimes overlap.  Same device functions as k_pass: same answers.
// ------------------------------------------------------------------------------------------
#ifndef MIRGE_SURV_PLAIN_LOADS
// The survivor list a workgroup wrote in one pass is read back by the SAME workgroup in the next, behind a __syncthreads(): a
// workgroup-scope release / acquire, which covers global memory -- the waves of a workgroup share their CU's L1, and that L1
// sees the CU's own write-through stores.  Plain loads are therefore enough.  (Round 3 used agent-scope loads "because the L1
// may hold the lines of two passes ago"; a build with plain loads passes the oracle, brute-force, fuzz and full-size C3 tests
// -- profiles/README.md round 4 -- and the model says it must.  MIRGE_SURV_PLAIN_LOADS=0 brings the agent-scope loads back.)
// ONE assumption sits under this: the waves of a workgroup run on one CU.  In threadgroup-split mode (-mtgsplit) they may
// not, and plain loads would race.  hipcc defines no macro for the mode, so it is checked where it can be seen:
// __graft_entry__.build() refuses the flag and reads compute_pgm_rsrc3.TG_SPLIT of every kernel descriptor of the built
// library (mirge3.0_amd/_codeobj.py; tests/test_host_logic.py::test_no_kernel_is_built_for_threadgroup_split_mode).
#define MIRGE_SURV_PLAIN_LOADS 1
#endif
#ifndef MIRGE_BULK_WAVES
#define MIRGE_BULK_WAVES 6  // waves per SIMD = four-wave workgroups per CU the bulk kernel is built for (native_cascade.hpp sizes its grid with it)
#endif
struct FusedStep {
    MirgeLibView lib;
    MirgePolicy pol;
    MergeInfo mi;
    const MirgePlanTable* plan;
    int32_t pass_id;
};
struct FusedSteps {
    int32_t n;
    FusedStep s[MIRGE_MAX_PASSES_K];
};

// ------------------------------------------------------------------------------------------
// Exact passes as ONE lookup of the whole read (round 5).  A pass whose policy admits no mismatch anywhere in the read it is
// handed -- "-v 0" (pass 3, primary tRNA, manifoldAlign.py:85,118-126), or "-n 0" with a length rule that keeps the read inside
// the seed (pass 0, exact miRNA: len < 26 <= 28, :85,93) -- over a SMALL library asks one question: is this very sequence a
// substring of some reference, and where first?  The probe path answers it with a chain of dependent requests (non-empty bit ->
// CSR bounds -> position list -> text window -> invalid bits: 4-5 L1-missing requests for every read that shares its first 9
// bases with a miRNA, which is most of a sample's isomiRs); k_cascade_bulk is bound by the L1 misses a CU has in flight.
// Here every valid window of the library, for every length the batch can ask for, sits in an open-addressing table keyed by
// the whole sequence: entry = tag (27 bits of the hash) | length (5) | lowest position + 1 (32), 8 bytes, load factor <= 1/3,
// 2 MB for the human miRNA set at read lengths 16-25: ONE request answers a miss, a tag match is confirmed against the text
// (one more request, for the ~0.3 % of collapsed reads that are exact miRNAs), so the answer is exact whatever the tag width.
// The lowest position among equal sequences is the cascade's documented tie-break (fewest mismatches -- 0 --, then lowest
// reference, then leftmost offset); the table is built with atomicMin on entries of one (tag, length, sequence).
// Because such a step costs a lane a few instructions, it does not get a walk over the survivor list of its own: it rides in
// front of the next pass's alignment (`pre`) or behind the previous one's (`post`) -- per read the order of the passes is kept,
// and no read's answer depends on another read's.  Human cascade: [exact miRNA | hairpin] [mature tRNA | primary tRNA]
// [snoRNA+rRNA+ncRNA] [mRNA] [isomiR]: five walks instead of seven.
// ------------------------------------------------------------------------------------------
#ifndef MIRGE_EXACT_LAUNDER
#define MIRGE_EXACT_LAUNDER 0  // 1: the step's descriptor re-read inside the loop through an opaque pointer (measured: its loads become vector loads)
#endif
#ifndef MIRGE_EXACT_RIDE
#define MIRGE_EXACT_RIDE 1     // 1: an exact step rides in a neighbouring alignment pass's walk; 0: walks of their own (a loop without alignment)
#endif
struct ExactStep {
    const uint64_t* slots;  // nullptr: no such step
    const uint64_t* T;      // the library's text (confirmation of a tag match)
    uint32_t mask;          // slots - 1
    int32_t pass_id;
    int32_t step;           // index of the pass in the ordinary step list (survivor accounting)
    MirgePolicy pol;
};
MIRGE_HD uint64_t mirge_exact_tagged(uint64_t key, int l, uint32_t& slot) {  // -> (tag | length) << 32, first slot (unmasked)
    const uint64_t h = mirge_mix64(key);
    slot = (uint32_t)h;
    return ((h >> 37) << 37) | ((uint64_t)l << 32);
}

// bases of T from position g on, as many as 31 of them (the packed text is padded by 8 words)
__device__ __forceinline__ uint64_t text_from(gptr_u64 T, uint64_t g) {
    const PairU64 t = load_pair64(T + (g >> 5));
    const int s = (int)(g & 31) * 2;
    return s ? ((t.a >> s) | (t.b << (64 - s))) : t.a;
}

// every valid window [p, p + l) of the library for the lengths in lmask (bit l), l <= 31: counted (FILL = false: duplicates
// included, an upper bound that sizes the table) or inserted
template <bool FILL>
__global__ void k_exact_table(const uint64_t* __restrict__ T, const uint64_t* __restrict__ inv, uint64_t total, uint32_t lmask,
                              uint64_t* __restrict__ slots, uint32_t mask, unsigned long long* __restrict__ count) {
    unsigned long long mine = 0;
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = p >> 6;
        const int s = (int)(p & 63);
        uint64_t bad = inv[q] >> s;
        if (s) bad |= inv[q + 1] << (64 - s);
        const int run = bad ? (int)__builtin_ctzll(bad) : 64;  // valid bases from p on
        if (!run) continue;
        const uint64_t bits = text_from((gptr_u64)T, p);
        for (int l = 1; l <= 31 && l <= run; l++) {
            if (!((lmask >> l) & 1u)) continue;
            if (!FILL) { mine++; continue; }
            const uint64_t seq = bits & mirge_lowmask2(l);
            uint32_t sl;
            const uint64_t tl = mirge_exact_tagged(seq | (1ull << (2 * l)), l, sl);
            const uint64_t entry = tl | (uint64_t)((uint32_t)p + 1u);
            for (sl &= mask;; sl = (sl + 1) & mask) {
                unsigned long long cur = __hip_atomic_load((unsigned long long*)&slots[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == 0ull) {
                    cur = atomicCAS((unsigned long long*)&slots[sl], 0ull, (unsigned long long)entry);
                    if (cur == 0ull) break;
                }
                // the same (tag, length): the same sequence?  (an entry never changes its tag or length, and every position it
                // has ever held shows the same l bases)
                if ((cur >> 32) == (tl >> 32) && (text_from((gptr_u64)T, (uint64_t)((uint32_t)cur - 1u)) & mirge_lowmask2(l)) == seq) {
                    atomicMin((unsigned long long*)&slots[sl], (unsigned long long)entry);  // the lowest position stays
                    break;
                }
            }
        }
    }
    if (!FILL) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
    }
}

// the step's answer for one read: true + its position when the (trimmed, eligible) read is a substring of a reference
template <bool HASN>
__device__ __forceinline__ bool exact_lookup(const ExactStep& e_in, MirgeRead<1> r, bool open, uint32_t& pos) {
    // the step's descriptor is read HERE, every time: hoisted out of the walk's loop its ~16 scalars would stay live across
    // align_hybrid, which already spills scalar registers (the empty asm hides that the pointer is loop-invariant)
    const ExactStep* ep = &e_in;
#if MIRGE_EXACT_LAUNDER
    asm volatile("" : "+s"(ep));
#endif
    const ExactStep& e = *ep;
    if (!open || !mirge_effective_read<1>(r, e.pol)) return false;
    if (HASN && r.nm[0]) return false;  // an ambiguous call is a mismatch wherever it is aligned
    uint32_t sl;
    const uint64_t tl = mirge_exact_tagged(r.w[0] | (1ull << (2 * r.len)), r.len, sl);
    gptr_u64 slots = (gptr_u64)e.slots;
    for (sl &= e.mask;; sl = (sl + 1) & e.mask) {
        const uint64_t cur = slots[sl];
        if (cur == 0ull) return false;
        if ((cur >> 32) == (tl >> 32)) {
            const uint32_t p = (uint32_t)cur - 1u;
            if ((text_from((gptr_u64)e.T, p) & mirge_lowmask2(r.len)) == r.w[0]) { pos = p; return true; }
        }
    }
}

// One walk of a workgroup over its list: an optional exact step in front (`pre`), an optional alignment pass (`main`: the
// FusedStep, has_main), an optional exact step behind (`post`; only behind a main policy that leaves the read as it is).
struct BulkWalk {
    ExactStep pre, post;
    FusedStep main;
    int32_t has_main;
    int32_t main_step;  // index of the main pass in the ordinary step list
};
struct BulkWalks {
    int32_t n;
    BulkWalk w[MIRGE_MAX_PASSES_K];
};

// reads of this wave that are still open, added to a workgroup counter (survivor accounting of the steps inside a walk)
__device__ __forceinline__ void count_open(bool open, uint32_t* ctr) {
    const unsigned long long b = __ballot(open);
    if (b && (threadIdx.x & 63) == 0) atomicAdd(ctr, (uint32_t)__popcll(b));
}

// pass_segment for a walk: per read pre -> main -> post, in the cascade's order
template <int W, bool LDSP, bool COHERENT, bool HASN, bool REP>
__device__ __forceinline__ void walk_segment(const BulkWalk& wk, const PlanSrc<LDSP>& psrc, const GroupView<W>& g, const uint32_t* act_in,
                                             uint32_t n_in, size_t seg, size_t seg_r, uint32_t* __restrict__ act_out,
                                             int8_t* __restrict__ res_pass, uint32_t* __restrict__ res_pos, int8_t* __restrict__ res_mm,
                                             uint32_t* s_count, uint32_t* s_open, uint32_t* __restrict__ heavy_cnt,
                                             uint32_t* __restrict__ heavy_list) {
    const int lane = threadIdx.x & 63;
    const bool has_main = wk.has_main != 0;
    if (W == 1 && !has_main) {  // a walk of exact steps only: its own short loop, nothing of the alignment's registers alive
        if constexpr (W == 1) {
            const bool two = wk.pre.slots != nullptr && wk.post.slots != nullptr;
            for (uint32_t base = 0; base < n_in; base += MIRGE_BLOCK) {
                const uint32_t t = base + threadIdx.x;
                const bool valid = t < n_in;
                uint32_t idx = 0;
                if (valid) {
                    if (!act_in) idx = (uint32_t)seg_r + t;
                    else idx = COHERENT ? __hip_atomic_load(&act_in[seg + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : act_in[seg + t];
                }
                MirgeRead<1> r2;
                r2.w[0] = 0; r2.nm[0] = 0; r2.len = 0;
                if (valid) load_read<1, HASN>(g, idx, r2);
                bool open = valid;
                uint32_t p;
                if (wk.pre.slots && exact_lookup<HASN>(wk.pre, r2, open, p)) {
                    res_pass[idx] = (int8_t)wk.pre.pass_id; res_pos[idx] = p; res_mm[idx] = 0;
                    open = false;
                }
                if (two) count_open(open, &s_open[0]);
                if (wk.post.slots && exact_lookup<HASN>(wk.post, r2, open, p)) {
                    res_pass[idx] = (int8_t)wk.post.pass_id; res_pos[idx] = p; res_mm[idx] = 0;
                    open = false;
                }
                if (open && !act_in) { res_pass[idx] = -1; res_mm[idx] = -1; }
                const unsigned long long bal = __ballot(open);
                if (bal) {
                    uint32_t wbase = 0;
                    if (lane == 0) wbase = atomicAdd(s_count, (uint32_t)__popcll(bal));
                    wbase = __shfl(wbase, 0, 64);
                    if (open) act_out[seg + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = idx;
                }
            }
        }
        return;
    }
    const bool has_pre = MIRGE_EXACT_RIDE && W == 1 && wk.pre.slots != nullptr, has_post = MIRGE_EXACT_RIDE && W == 1 && wk.post.slots != nullptr;
    for (uint32_t base = 0; base < n_in; base += MIRGE_BLOCK) {
        const uint32_t t = base + threadIdx.x;
        const bool valid = t < n_in;
        uint32_t idx = 0;
        if (valid) {
            if (!act_in) idx = (uint32_t)seg_r + t;
            else idx = COHERENT ? __hip_atomic_load(&act_in[seg + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : act_in[seg + t];
        }
        MirgeRead<W> r2;
        if (valid) load_read<W, HASN>(g, idx, r2);
        else {
#pragma unroll
            for (int w = 0; w < W; w++) { r2.w[w] = 0; r2.nm[w] = 0; }
            r2.len = 0;
        }
        bool open = valid;
        if constexpr (W == 1 && MIRGE_EXACT_RIDE) {
            if (has_pre) {
                uint32_t p;
                if (exact_lookup<HASN>(wk.pre, r2, open, p)) {
                    res_pass[idx] = (int8_t)wk.pre.pass_id; res_pos[idx] = p; res_mm[idx] = 0;
                    open = false;
                }
                if (has_main || has_post) count_open(open, &s_open[0]);
            }
        }
        if (has_main) {
            const bool elig = open && mirge_effective_read<W>(r2, wk.main.pol);  // (a policy `post` follows leaves r2 as it was)
            uint64_t best;
            align_hybrid<W, LDSP, REP>(wk.main.lib, wk.main.pol, wk.main.mi, psrc, r2, elig, best);
            if (REP && elig && best == MIRGE_DEFER) {  // a bucket no wave should walk alone: k_cascade_heavy answers for this read
                defer_read(idx, heavy_cnt, heavy_list, res_pass, res_mm, wk.main.pass_id);
                open = false;
            } else
            if (elig && best != MIRGE_NO_HIT) {
                const int cls = (int)(best >> 40);  // member library of a merged pass (0 otherwise)
                res_pass[idx] = (int8_t)(wk.main.pass_id + cls);
                uint32_t b0 = 0;
#pragma unroll
                for (int i = 1; i < 4; i++) if (i == cls) b0 = wk.main.mi.bound[i];
                res_pos[idx] = (uint32_t)best - b0;  // position in that member's own text
                res_mm[idx] = (int8_t)((best >> 32) & 0xFF);
                open = false;
            }
            if (has_post) count_open(open, &s_open[1]);
        }
        if constexpr (W == 1 && MIRGE_EXACT_RIDE) {
            if (has_post) {
                uint32_t p;
                if (exact_lookup<HASN>(wk.post, r2, open, p)) {
                    res_pass[idx] = (int8_t)wk.post.pass_id; res_pos[idx] = p; res_mm[idx] = 0;
                    open = false;
                }
            }
        }
        if (open && !act_in) {     // the first walk sees every read of the group: it also writes "unannotated"
            res_pass[idx] = -1;
            res_mm[idx] = -1;
        }
        const unsigned long long bal = __ballot(open);
        if (bal) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(s_count, (uint32_t)__popcll(bal));
            wbase = __shfl(wbase, 0, 64);
            if (open) act_out[seg + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = idx;
        }
    }
}

// HASN = false: the build for a group without ambiguous calls -- its N masks are compile-time zeros and fold away in everything
// inlined behind the load (7 fewer spilled scalar registers, -2 % kernel time on the bulk group)
template <int W, bool HASN, bool REP>
__global__ void __launch_bounds__(MIRGE_BLOCK) __attribute__((amdgpu_waves_per_eu(W == 1 ? MIRGE_BULK_WAVES : 1, 8)))  // one-word reads: 6 workgroups per CU must be resident (80 VGPRs; k_pass: 77); wider reads keep their registers
k_cascade_bulk(const BulkWalks* __restrict__ walks, GroupView<W> g, uint32_t* __restrict__ actA, uint32_t* __restrict__ actB,
               uint32_t* __restrict__ seg_n, uint32_t cap, int8_t* __restrict__ res_pass, uint32_t* __restrict__ res_pos,
               int8_t* __restrict__ res_mm, const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ heavy_cnt,
               uint32_t* __restrict__ heavy_list) {
    __shared__ uint32_t s_count;
    __shared__ uint32_t s_open[2];  // reads still open behind the walk's pre step / behind its main pass
    constexpr bool LDSP = (W == 1) && MIRGE_LDS_PLAN;
    __shared__ __attribute__((aligned(16))) unsigned char s_plan_raw[LDSP ? sizeof(LdsPlan) : 16];
    LdsPlan* s_plan = reinterpret_cast<LdsPlan*>(s_plan_raw);
    const int nwalks = walks->n;
    const size_t seg = (size_t)blockIdx.x * cap;
    size_t seg_r = seg;
    // (round 6) when the workgroup started and ended, on the constant-rate clock, in the two rows behind the survivor counts: what
    // says whether ONE workgroup's segment -- a read whose probe bucket holds 10^5 positions -- is what the launch waited for
    if (threadIdx.x == 0) seg_n[(size_t)(MIRGE_MAX_PASSES_K + 1) * gridDim.x + blockIdx.x] = (uint32_t)wall_clock64();
    uint32_t n_in = first_pass_share(g.n, n_dev, cap, seg_r);
    const uint32_t* act_in = nullptr;
    uint32_t* act_out = actA;
    for (int wi = 0; wi < nwalks; wi++) {
        const BulkWalk& wk = walks->w[wi];
        if (threadIdx.x == 0) { s_count = 0; s_open[0] = 0; s_open[1] = 0; }
        if (LDSP && wk.has_main) lds_plan_fill(*s_plan, wk.main.lib, wk.main.plan);
        __syncthreads();
        PlanSrc<LDSP> psrc;
        psrc.g = wk.main.plan; psrc.l = LDSP ? s_plan : nullptr;
        walk_segment<W, LDSP, !MIRGE_SURV_PLAIN_LOADS, HASN, REP>(wk, psrc, g, act_in, n_in, seg, seg_r, act_out, res_pass, res_pos, res_mm, &s_count, s_open,
                                                             heavy_cnt, heavy_list);
        __syncthreads();  // the survivors are written (and visible to this workgroup at L2), the plan is free again
        n_in = s_count;
        if (threadIdx.x == 0) {  // seg_n[step][workgroup] = reads still open behind that pass (= handed to the next one)
            const bool pre = W == 1 && wk.pre.slots, post = W == 1 && wk.post.slots;
            if (pre) seg_n[(size_t)wk.pre.step * gridDim.x + blockIdx.x] = (wk.has_main || post) ? s_open[0] : n_in;  // (exact-only walk of two steps: s_open[0] too)
            if (wk.has_main) seg_n[(size_t)wk.main_step * gridDim.x + blockIdx.x] = post ? s_open[1] : n_in;
            if (post) seg_n[(size_t)wk.post.step * gridDim.x + blockIdx.x] = n_in;
        }
        __syncthreads();  // everybody has read s_count before it is reset
        act_in = act_out;
        act_out = (act_out == actA) ? actB : actA;
    }
    if (threadIdx.x == 0) seg_n[(size_t)(MIRGE_MAX_PASSES_K + 2) * gridDim.x + blockIdx.x] = (uint32_t)wall_clock64();
    // (k_resolve stays a launch of its own: done here, for the workgroup's own reads, it is a chain of dependent loads with
    // four waves to hide it -- the kernel grew by 0.04 ms to save a 0.046 ms launch that the whole chip runs at once)
}

// (Four reads per thread with their loads issued together -- before and after the table became one access -- changed nothing:
// 45 -> 49-53 us.  7.6 M wave-level VALU instructions per launch, most of them the 16-way pointer select, 73 MB moved.)
__global__ void k_resolve(ResolveTable tb, const int8_t* __restrict__ res_pass, const uint32_t* __restrict__ res_pos,
                          uint32_t n, int32_t* __restrict__ res_ref, int32_t* __restrict__ res_off,
                          const uint32_t* __restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int p = res_pass[i];
        int32_t ref = -1, off = -1;
        if (p >= 0) resolve_one(tb, p, res_pos[i], ref, off);
        res_ref[i] = ref;
        res_off[i] = off;
    }
}

// ------------------------------------------------------------------------------------------
// k_cascade_fused: the whole cascade of a SMALL read group in one launch.  The groups beside the bulk
// (reads with an N, 32-128 nt reads: a few hundred to a few 100 k reads) cost one launch per pass plus
// memsets and a resolve each -- ~30 launches whose host-side enqueue time, not their GPU time, kept the
// bulk group's first pass waiting (profiles/r01_timeline.txt).  Here a wave keeps its 64 reads through
// every step; a read that is annotated simply stops being eligible (no compaction: the group is small).
// Same device functions as k_pass, so the same answers.
// ------------------------------------------------------------------------------------------
template <int W, bool HASN, bool REP>
__global__ void __launch_bounds__(MIRGE_BLOCK)
k_cascade_fused(const FusedSteps* __restrict__ steps, ResolveTable tb, GroupView<W> g, int8_t* __restrict__ res_pass,
                uint32_t* __restrict__ res_pos, int8_t* __restrict__ res_mm, int32_t* __restrict__ res_ref,
                int32_t* __restrict__ res_off, uint32_t* __restrict__ heavy_cnt, uint32_t* __restrict__ heavy_list) {
    const uint32_t nrounds = (g.n + MIRGE_BLOCK - 1) / MIRGE_BLOCK;
    const int nsteps = steps->n;
    for (uint32_t round = blockIdx.x; round < nrounds; round += gridDim.x) {
        const uint32_t idx = round * MIRGE_BLOCK + threadIdx.x;
        const bool valid = idx < g.n;
        MirgeRead<W> r0;
        if (valid) load_read<W, HASN>(g, idx, r0);
        else {
#pragma unroll
            for (int w = 0; w < W; w++) { r0.w[w] = 0; r0.nm[w] = 0; }
            r0.len = 0;
        }
        bool open = valid;
        int8_t o_pass = -1, o_mm = -1;
        uint32_t o_pos = 0;
        for (int si = 0; si < nsteps; si++) {
            if (!__ballot(open)) break;  // wave-uniform
            const FusedStep& st = steps->s[si];
            MirgeRead<W> r2 = r0;
            const bool elig = open && mirge_effective_read<W>(r2, st.pol);
            uint64_t best;
            PlanSrc<false> psrc;
            psrc.g = st.plan; psrc.l = nullptr;
            align_hybrid<W, false, REP>(st.lib, st.pol, st.mi, psrc, r2, elig, best);
            if (REP && elig && best == MIRGE_DEFER) {  // k_cascade_heavy answers for this read (and writes all five of its fields)
                o_pass = (int8_t)(-2 - st.pass_id);  // (see defer_read)
                heavy_list[atomicAdd(&heavy_cnt[0], 1u)] = idx;
                open = false;
            } else
            if (elig && best != MIRGE_NO_HIT) {
                const int cls = (int)(best >> 40);
                o_pass = (int8_t)(st.pass_id + cls);
                uint32_t b0 = 0;
#pragma unroll
                for (int i = 1; i < 4; i++) if (i == cls) b0 = st.mi.bound[i];
                o_pos = (uint32_t)best - b0;
                o_mm = (int8_t)((best >> 32) & 0xFF);
                open = false;
            }
        }
        if (valid) {
            int32_t ref = -1, off = -1;
            if (o_pass >= 0) resolve_one(tb, o_pass, o_pos, ref, off);
            res_pass[idx] = o_pass; res_pos[idx] = o_pos; res_mm[idx] = o_mm;
            res_ref[idx] = ref; res_off[idx] = off;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_cascade_spec + k_cascade_pick (round 6): the cascade of a TINY read group (a few hundred to a few thousand reads: the groups with an
// ambiguous call) with all of its passes AT ONCE.  In k_cascade_fused a wave takes its 64 reads through the passes one after the other:
// nine dependent chains of lookups, 35-65 us for a group of 40-700 reads whatever the chip is doing -- and on a sample with few unique
// reads those chains are what the step waits for once the bulk kernel has retired (profiles/r06_timeline_zipf.txt).  A pass's answer for
// a read depends on the read alone (its own subset rule and alignment); the cascade only says which answer COUNTS: the first pass, in
// order, that has one.  So block (x = round of 256 reads, y = step) computes step y's answer for every read of its round as if the read
// were still open -- nine times the work of a group that is a thousandth of the sample -- and k_cascade_pick takes, per read, the first
// step with an answer (or hands the read to k_cascade_heavy when a step that comes before any answer deferred it).  Same device
// functions, same answers.
// ------------------------------------------------------------------------------------------
// one read's place in the cascade from its per-step answers (answers[si * stride + idx]): the first step with one
__device__ __forceinline__ void spec_pick_one(const FusedSteps* __restrict__ steps, const ResolveTable& tb, size_t stride, uint32_t idx,
                                              const unsigned long long* __restrict__ answers, int8_t* __restrict__ res_pass,
                                              uint32_t* __restrict__ res_pos, int8_t* __restrict__ res_mm, int32_t* __restrict__ res_ref,
                                              int32_t* __restrict__ res_off, uint32_t* __restrict__ heavy_cnt, uint32_t* __restrict__ heavy_list) {
    const int nsteps = steps->n;
    int8_t o_pass = -1, o_mm = -1;
    uint32_t o_pos = 0;
    for (int si = 0; si < nsteps; si++) {
        const uint64_t best = answers[(size_t)si * stride + idx];
        if (best == MIRGE_NO_HIT) continue;
        const FusedStep& st = steps->s[si];
        if (best == MIRGE_DEFER) {  // (see defer_read: the mark names the pass)
            o_pass = (int8_t)(-2 - st.pass_id);
            heavy_list[atomicAdd(&heavy_cnt[0], 1u)] = idx;
            break;
        }
        const int cls = (int)(best >> 40);
        o_pass = (int8_t)(st.pass_id + cls);
        uint32_t b0 = 0;
#pragma unroll
        for (int i = 1; i < 4; i++) if (i == cls) b0 = st.mi.bound[i];
        o_pos = (uint32_t)best - b0;
        o_mm = (int8_t)((best >> 32) & 0xFF);
        break;
    }
    int32_t ref = -1, off = -1;
    if (o_pass >= 0) resolve_one(tb, o_pass, o_pos, ref, off);
    res_pass[idx] = o_pass; res_pos[idx] = o_pos; res_mm[idx] = o_mm;
    res_ref[idx] = ref; res_off[idx] = off;
}

// `tickets` (one counter per round of 256 reads, zero between launches) = the pick in the same launch: the workgroup that finishes a
// round's LAST step picks for that round's reads (its ticket is the count of steps before it) and puts the counter back to zero.
// nullptr: the answers only, k_cascade_pick follows.  `stride` = the answers' row length (a multiple of 256: no two rounds share a line).
template <int W, bool HASN, bool REP>
__global__ void __launch_bounds__(MIRGE_BLOCK)
k_cascade_spec(const FusedSteps* __restrict__ steps, ResolveTable tb, GroupView<W> g, unsigned long long* __restrict__ answers /*[nsteps][stride]*/,
               uint32_t stride, uint32_t* __restrict__ tickets, int8_t* __restrict__ res_pass, uint32_t* __restrict__ res_pos,
               int8_t* __restrict__ res_mm, int32_t* __restrict__ res_ref, int32_t* __restrict__ res_off, uint32_t* __restrict__ heavy_cnt,
               uint32_t* __restrict__ heavy_list) {
    __shared__ uint32_t s_last;
    const int si = blockIdx.y;
    const FusedStep& st = steps->s[si];
    const uint32_t idx = blockIdx.x * MIRGE_BLOCK + threadIdx.x;
    const bool valid = idx < g.n;
    MirgeRead<W> r2;
    if (valid) load_read<W, HASN>(g, idx, r2);
    else {
#pragma unroll
        for (int w = 0; w < W; w++) { r2.w[w] = 0; r2.nm[w] = 0; }
        r2.len = 0;
    }
    const bool elig = valid && mirge_effective_read<W>(r2, st.pol);
    uint64_t best;
    PlanSrc<false> psrc;
    psrc.g = st.plan; psrc.l = nullptr;
    align_hybrid<W, false, REP>(st.lib, st.pol, st.mi, psrc, r2, elig, best);
    if (valid) answers[(size_t)si * stride + idx] = elig ? best : MIRGE_NO_HIT;
    if (!tickets) return;
    __threadfence();  // this workgroup's answers are out (agent scope) before its ticket is drawn
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.y - 1 ? 1u : 0u;
        if (s_last) __hip_atomic_store(&tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();  // ... and the other steps' answers are read behind theirs
    if (valid) spec_pick_one(steps, tb, stride, idx, answers, res_pass, res_pos, res_mm, res_ref, res_off, heavy_cnt, heavy_list);
}

template <int W>
__global__ void k_cascade_pick(const FusedSteps* __restrict__ steps, ResolveTable tb, uint32_t n, uint32_t stride,
                               const unsigned long long* __restrict__ answers,
                               int8_t* __restrict__ res_pass, uint32_t* __restrict__ res_pos, int8_t* __restrict__ res_mm, int32_t* __restrict__ res_ref,
                               int32_t* __restrict__ res_off, uint32_t* __restrict__ heavy_cnt, uint32_t* __restrict__ heavy_list) {
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x)
        spec_pick_one(steps, tb, stride, idx, answers, res_pass, res_pos, res_mm, res_ref, res_off, heavy_cnt, heavy_list);
}

// (the reads align_hybrid deferred: see align_wg above)
template <int W, bool HASN>
__global__ void __launch_bounds__(MIRGE_HEAVY_THREADS)
k_cascade_heavy(const FusedSteps* __restrict__ steps, ResolveTable tb, GroupView<W> g, uint32_t* __restrict__ heavy_cnt,
                const uint32_t* __restrict__ heavy_list, int8_t* __restrict__ res_pass, uint32_t* __restrict__ res_pos,
                int8_t* __restrict__ res_mm, int32_t* __restrict__ res_ref, int32_t* __restrict__ res_off) {
    __shared__ unsigned long long s_best;
    const uint32_t n = __hip_atomic_load(&heavy_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nsteps = steps->n;
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        const uint32_t idx = heavy_list[item];
        MirgeRead<W> r0;
        load_read<W, HASN>(g, idx, r0);
        const int from_pass = -2 - (int)res_pass[idx];  // the pass that deferred the read: those before it found nothing
        __syncthreads();  // (every thread has read the mark before thread 0 overwrites it with the answer)
        int8_t o_pass = -1, o_mm = -1;
        uint32_t o_pos = 0;
        for (int si = 0; si < nsteps && o_pass < 0; si++) {
            const FusedStep& st = steps->s[si];
            if (st.pass_id < from_pass) continue;
            MirgeRead<W> r2 = r0;
            if (!mirge_effective_read<W>(r2, st.pol)) continue;
            const uint64_t best = align_wg<W>(st.lib, st.pol, st.mi, st.plan, r2, &s_best);
            if (best != MIRGE_NO_HIT) {
                const int cls = (int)(best >> 40);
                o_pass = (int8_t)(st.pass_id + cls);
                uint32_t b0 = 0;
#pragma unroll
                for (int i = 1; i < 4; i++) if (i == cls) b0 = st.mi.bound[i];
                o_pos = (uint32_t)best - b0;
                o_mm = (int8_t)((best >> 32) & 0xFF);
            }
        }
        if (threadIdx.x == 0) {
            int32_t ref = -1, off = -1;
            if (o_pass >= 0) resolve_one(tb, o_pass, o_pos, ref, off);
            res_pass[idx] = o_pass; res_pos[idx] = o_pos; res_mm[idx] = o_mm;
            if (res_ref) { res_ref[idx] = ref; res_off[idx] = off; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&heavy_cnt[1], 1u) == gridDim.x - 1) {
        __hip_atomic_store(&heavy_cnt[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&heavy_cnt[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
