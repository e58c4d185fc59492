// kernels_trim.hpp -- part of mirge_kernels.hpp: k_trim, the read modifiers the reference runs through cutadapt before
// it counts a read (mirge/libs/digest.py:59-101,320-375), on the text in HBM between k_nl_mark and
// k_seq_class: a record's [start, end) is narrowed, nothing is copied.
//   modifiers, in the reference's order: NextSeq quality trimming, quality trimming (-q, default 10 at the 3' end),
//   adapter removal (one 3' adapter, -a, or one 5' adapter, -g; error rate 0.12 of the aligned adapter length, minimum
//   overlap 3, indels allowed), N trimming at both ends, unconditional cuts (-u).
//   The reference's worker tests the length and counts the read INSIDE its loop over the modifiers (digest.py:354-373):
//   a read is counted once after EVERY modifier.  stages_out = number of modifiers reproduces that (virtual record
//   r * stages_out + s = read r after modifier s); stages_out = 1 keeps only the fully trimmed read.
// Restated from cutadapt's published algorithms (qualtrim.pyx, _align.pyx Aligner.locate, modifiers.py): parity
// unpinned -- cutadapt is third-party and absent here.
#pragma once

#define MIRGE_TRIM_MAX_ADAPTER 64
#define MIRGE_TRIM_MAX_MODS 8
struct TrimOpts {
    int32_t nextseq;        // cutoff, -1 = off
    int32_t q_front, q_back;  // q_back -1 = off
    int32_t base;           // 33 / 64
    int32_t alen;           // adapter length, 0 = none
    int32_t front;          // 1: the adapter is a 5' adapter (-g), everything up to its end is removed; N is not accepted in it
    int32_t min_overlap;
    double rate;            // maximum error rate
    int32_t trim_n;
    int32_t n_cut, cut[2];
    int32_t n_mods;         // modifiers in the chain (after dropping the quality ones for a text without qualities)
    int32_t stages_out;     // n_mods (count after every modifier) or 1
    uint8_t adapter[MIRGE_TRIM_MAX_ADAPTER];
    uint8_t wild[MIRGE_TRIM_MAX_ADAPTER];  // adapter position is N: matches any base, not counted in the error-rate length
    // a second adapter (-a and -g in one run, or two of a kind): AdapterCutter with times = 1 removes, per read, the ONE
    // adapter that matches best -- most matches, then fewest errors, then the first given (cutadapt's _best_match)
    int32_t alen2, front2;
    int32_t times;      // -n COUNT: the search is repeated on what the last removal left, until nothing matches (1 = once)
    int32_t no_indels;  // --no-indels: substitutions only (cutadapt prices an indel at 100 000); general kernel only
    int32_t read_wild;  // --match-read-wildcards: an N in the READ matches every adapter base; general kernel only
    int32_t action_none;  // --action none: the adapter is searched but the read stays as it is (the modifier still counts)
    uint8_t adapter2[MIRGE_TRIM_MAX_ADAPTER];
    uint8_t wild2[MIRGE_TRIM_MAX_ADAPTER];
    // round 5 (general kernel only): anchored adapters and ONE linked adapter (cutadapt's PrefixAdapter / SuffixAdapter /
    // LinkedAdapter).  anch / anch2: the adapter must be taken whole at the read's first base (5': `-g ^ADAPTER`) or up to its
    // last base (3': `-a ADAPTER$`).  linked: `adapter` is the 5' part and `adapter2` the 3' part of `ADAPTER1...ADAPTER2`;
    // req1 / req2: the part must be found for the adapter to match at all (`-a`: 5' part anchored + required, 3' optional;
    // `-g`: both required).
    int32_t anch, anch2, linked, req1, req2;
};

// Aligner.locate for a regular 3' adapter on read[0, n): returns the read position where the adapter starts, or n.
// One DP column lives in registers, one 32-bit entry per adapter row:
//     cost << 24 | choice << 22 | matches << 15 | origin + 64     (cost, matches <= 64; -64 <= origin <= 32703)
// `choice` is 0 in a stored entry; a mismatching cell takes min3(diagonal, insertion | 1 << 22, deletion | 2 << 22), which is
// the lowest cost with cutadapt's preference diagonal > insertion > deletion on ties in ONE v_min3_u32, then clears the
// choice bits and adds one to the cost.  The row loop is fully unrolled (MAXM = 32 covers the adapters in use -- TruSeq
// small RNA is 29 nt; 64 is the general form, a separate kernel so that the common one keeps its registers).
// FRONT (a regular 5' adapter): the alignment may also start inside the adapter -- row i starts at cost 0 with origin -i --
// and must reach the adapter's last base; every column is a candidate end, its aligned adapter length is
// m + min(origin, 0); returns the read position where the adapter ENDS, or 0.
#define MIRGE_TRIM_ORIGIN_BIAS 64u
#define MIRGE_TRIM_COST_SHIFT 24
#define MIRGE_TRIM_MATCH_ONE (1u << 15)
#define MIRGE_TRIM_CHOICE_MASK (3u << 22)
// EXACT: the adapter has exactly MAXM bases and no N -- no per-row predicate is left in the unrolled loop (a kernel per
// adapter length, 1-64).  Otherwise MAXM is a capacity (64) and rows beyond o.alen / wildcard rows are tested at run time.
// WHICH: 0 = o.adapter, 1 = o.adapter2.  hit (optional): {found, matches, cost} of the reported alignment.
template <int MAXM, bool EXACT, bool FRONT, int WHICH = 0>
__device__ __forceinline__ int adapter_cut_point(const TrimOpts& o, const uint8_t* __restrict__ read, int n, int* hit = nullptr) {
    const uint8_t* const o_adapter = WHICH ? o.adapter2 : o.adapter;
    const uint8_t* const o_wild = WHICH ? o.wild2 : o.wild;
    const int m = EXACT ? MAXM : (WHICH ? o.alen2 : o.alen);
    // anchored (general kernel): FRONT = PrefixAdapter (flags STOP_WITHIN_SEQ2 alone: read and adapter both start at their
    // first base, first row and column cost their index, candidates stay the last row's cells); back = SuffixAdapter
    // (START_WITHIN_SEQ2 alone: the one candidate is the whole adapter ending at the read's last base)
    const bool anch = !EXACT && (WHICH ? o.anch2 : o.anch) != 0;
    uint32_t e[MAXM + 1];
    uint8_t nw[MAXM + 1];
    nw[0] = 0;
    // --no-indels (general kernel): the two indel candidates of a cell are raised above every diagonal one, and a 3' adapter
    // cannot lose bases in front of the read: rows of the first column start at cost 128 (never accepted, and 128 + 64
    // mismatches still fit the entry's 8 cost bits)
    const uint32_t no_indel_or = (!EXACT && o.no_indels) ? 0xFF000000u : 0u;
#pragma unroll
    for (int i = 0; i <= MAXM; i++) {
        e[i] = (FRONT && !anch) ? MIRGE_TRIM_ORIGIN_BIAS - (uint32_t)i
                                : (((no_indel_or && i ? 128u : (uint32_t)i) << MIRGE_TRIM_COST_SHIFT) | MIRGE_TRIM_ORIGIN_BIAS);
        if (i) nw[i] = EXACT ? (uint8_t)0 : (uint8_t)(nw[i - 1] + (i <= m ? o_wild[i - 1] : 0));
    }
    int b_mat = -1, b_cost = 0, b_val = 0;
    bool found = false, exact = false;
    // i: adapter rows the entry has passed; j: read column it ends in (FRONT only)
    auto consider = [&](uint32_t ent, int i, int j) {
        const int cost = (int)(ent >> MIRGE_TRIM_COST_SHIFT), mat = (int)((ent >> 15) & 0x7F);
        const int origin = (int)(ent & 0x7FFF) - (int)MIRGE_TRIM_ORIGIN_BIAS;
        const int length = FRONT ? i + (origin < 0 ? origin : 0) : i;
        if (length >= o.min_overlap && (double)cost <= (double)(length - (FRONT ? 0 : nw[i])) * o.rate &&
            (!found || mat > b_mat || (mat == b_mat && cost < b_cost))) {
            found = true; b_mat = mat; b_cost = cost; b_val = FRONT ? j : origin;
        }
    };
    for (int j = 1; j <= n && !exact; j++) {
        const uint8_t ch = read[j - 1] & 0xDF;
        uint32_t diag = e[0];
        e[0] = (uint32_t)j + MIRGE_TRIM_ORIGIN_BIAS;  // cost 0, matches 0, origin j
        // (anchored 5': j read bases in front of the adapter cost j -- saturated at 128, beyond every budget and within the
        // entry's 8 cost bits with the <= 64 a column can add --, origin 0)
        if (FRONT && anch) e[0] = ((no_indel_or ? 128u : (uint32_t)(j < 128 ? j : 128)) << MIRGE_TRIM_COST_SHIFT) | MIRGE_TRIM_ORIGIN_BIAS;
        uint32_t last = e[0];
        // (round 6) m is the same number in every lane and every column; hidden from the optimiser per column, `i <= mj` is one scalar
        // compare and a branch per row INSIDE the column loop, and the column's last row computed is row m.  Left to itself the compiler
        // kept 2 x MAXM lane masks for `i <= m` / `i == m` alive across the column loop, most of them spilled and read back with two
        // v_readlane per row and column.
        int mj = m;
        if (!EXACT) asm volatile("" : "+s"(mj));
#pragma unroll
        for (int i = 1; i <= MAXM; i++) {
            if (EXACT || i <= mj) {
                const uint32_t left = e[i];  // previous column, same row
                const uint32_t best3 = EXACT ? min(min(diag, e[i - 1] | (1u << 22)), left | (2u << 22))
                                             : min(min(diag, e[i - 1] | (1u << 22) | no_indel_or), left | (2u << 22) | no_indel_or);
                const uint32_t miss = (best3 & ~MIRGE_TRIM_CHOICE_MASK) + (1u << MIRGE_TRIM_COST_SHIFT);
                const bool same = EXACT ? o_adapter[i - 1] == ch : (o_wild[i - 1] || o_adapter[i - 1] == ch || (o.read_wild && ch == 'N'));
                const uint32_t v = same ? diag + MIRGE_TRIM_MATCH_ONE : miss;
                diag = left;
                e[i] = v;
                last = v;
            }
        }
        if (FRONT || !anch || j == n) consider(last, m, j);
        exact = found && b_cost == 0 && b_mat == m;
    }
    if (!exact && !FRONT && !anch) {  // the adapter may run off the read's end: every prefix of it, in the last column, longest first
#pragma unroll                // (cutadapt: `for i in reversed(range(first_i, m + 1))` -- on equal (matches, cost) the longer prefix stays)
        for (int i = MAXM; i >= 0; i--)
            if (EXACT || i <= m) consider(e[i], i, n);
    }
    if (hit) { hit[0] = found ? 1 : 0; hit[1] = b_mat; hit[2] = b_cost; }
    return found ? b_val : (FRONT ? 0 : n);
}

// The same search with the adapter's kind and number as RUN-TIME (wave-uniform) values: the general 3' instance of k_trim calls it from
// its linked / best-of-two / repeated branches.  Seven inlined compile-time variants of the function above, each with its own DP
// column, had left that instance with 139-1101 spilled registers; this is one body.
// (round 6) The adapter as the search reads it: for each of A C G T the rows that hold that letter, and the wildcard rows, as 64-bit
// masks.  The rows' letters used to be compared one by one with bytes of the kernel's argument block, which the compiler kept in
// scalar registers -- 64 + 64 bytes per adapter, two adapters: most of the general instance's thousand spilled scalar registers.  A
// column now selects ONE mask by its read letter and a row tests its bit.
struct AdapterMasks { uint64_t eq[4]; uint64_t wild; };
__device__ __forceinline__ AdapterMasks adapter_masks(const uint8_t* adapter, const uint8_t* wild, int m) {
    AdapterMasks am{{0ull, 0ull, 0ull, 0ull}, 0ull};
    for (int i = 0; i < m && i < MIRGE_TRIM_MAX_ADAPTER; i++) {
        const uint8_t c = adapter[i];
        const uint64_t bit = 1ull << i;
        if (c == 'A') am.eq[0] |= bit; else if (c == 'C') am.eq[1] |= bit; else if (c == 'G') am.eq[2] |= bit; else if (c == 'T') am.eq[3] |= bit;
        if (wild[i]) am.wild |= bit;
    }
    return am;
}

template <int MAXM>
__device__ __forceinline__ int adapter_cut_point_rt(const TrimOpts& o, const AdapterMasks& am1, const AdapterMasks& am2, const uint8_t* __restrict__ read, int n,
                                                    const bool FRONT, const int WHICH, int* hit) {
    constexpr bool EXACT = false;
    const uint8_t* const o_adapter = WHICH ? o.adapter2 : o.adapter;
    const uint64_t eqA = WHICH ? am2.eq[0] : am1.eq[0], eqC = WHICH ? am2.eq[1] : am1.eq[1], eqG = WHICH ? am2.eq[2] : am1.eq[2],
                   eqT = WHICH ? am2.eq[3] : am1.eq[3], wildm = WHICH ? am2.wild : am1.wild;
    const int m = EXACT ? MAXM : (WHICH ? o.alen2 : o.alen);
    // anchored (general kernel): FRONT = PrefixAdapter (flags STOP_WITHIN_SEQ2 alone: read and adapter both start at their
    // first base, first row and column cost their index, candidates stay the last row's cells); back = SuffixAdapter
    // (START_WITHIN_SEQ2 alone: the one candidate is the whole adapter ending at the read's last base)
    const bool anch = !EXACT && (WHICH ? o.anch2 : o.anch) != 0;
    uint32_t e[MAXM + 1];
    // (round 6) wildcard rows up to row i, which the error-rate length leaves out: only nw(m) is needed while the columns run, and the
    // prefixes' counts come out of a running count in the last loop -- the table of MAXM + 1 of them was MAXM + 1 live scalar
    // registers of this instance's thousand spilled ones
    int nw_m = 0;
    // --no-indels (general kernel): the two indel candidates of a cell are raised above every diagonal one, and a 3' adapter
    // cannot lose bases in front of the read: rows of the first column start at cost 128 (never accepted, and 128 + 64
    // mismatches still fit the entry's 8 cost bits)
    const uint32_t no_indel_or = (!EXACT && o.no_indels) ? 0xFF000000u : 0u;
#pragma unroll
    for (int i = 0; i <= MAXM; i++) {
        e[i] = (FRONT && !anch) ? MIRGE_TRIM_ORIGIN_BIAS - (uint32_t)i
                                : (((no_indel_or && i ? 128u : (uint32_t)i) << MIRGE_TRIM_COST_SHIFT) | MIRGE_TRIM_ORIGIN_BIAS);
    }
    nw_m = __popcll(wildm);
    int b_mat = -1, b_cost = 0, b_val = 0;
    bool found = false, exact = false;
    // i: adapter rows the entry has passed; j: read column it ends in (FRONT only)
    auto consider = [&](uint32_t ent, int i, int j, int nw_i) {
        const int cost = (int)(ent >> MIRGE_TRIM_COST_SHIFT), mat = (int)((ent >> 15) & 0x7F);
        const int origin = (int)(ent & 0x7FFF) - (int)MIRGE_TRIM_ORIGIN_BIAS;
        const int length = FRONT ? i + (origin < 0 ? origin : 0) : i;
        if (length >= o.min_overlap && (double)cost <= (double)(length - (FRONT ? 0 : nw_i)) * o.rate &&
            (!found || mat > b_mat || (mat == b_mat && cost < b_cost))) {
            found = true; b_mat = mat; b_cost = cost; b_val = FRONT ? j : origin;
        }
    };
    for (int j = 1; j <= n && !exact; j++) {
        const uint8_t ch = read[j - 1] & 0xDF;
        // the rows this column's letter matches: the letter's mask, the wildcard rows, every row for an N of the read under
        // --match-read-wildcards; a letter that is none of A C G T (an IUPAC code in the read) is compared with the adapter's bytes
        uint64_t eqm = ch == 'A' ? eqA : ch == 'C' ? eqC : ch == 'G' ? eqG : ch == 'T' ? eqT : 0ull;
        if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T')
            for (int i = 0; i < m; i++) eqm |= (uint64_t)(o_adapter[i] == ch) << i;
        eqm |= wildm;
        if (o.read_wild && ch == 'N') eqm = ~0ull;
        uint32_t diag = e[0];
        e[0] = (uint32_t)j + MIRGE_TRIM_ORIGIN_BIAS;  // cost 0, matches 0, origin j
        // (anchored 5': j read bases in front of the adapter cost j -- saturated at 128, beyond every budget and within the
        // entry's 8 cost bits with the <= 64 a column can add --, origin 0)
        if (FRONT && anch) e[0] = ((no_indel_or ? 128u : (uint32_t)(j < 128 ? j : 128)) << MIRGE_TRIM_COST_SHIFT) | MIRGE_TRIM_ORIGIN_BIAS;
        uint32_t last = e[0];
        int mj = m;  // (as in adapter_cut_point: one scalar compare per row inside the column loop, no lane masks kept across it)
        asm volatile("" : "+s"(mj));
#pragma unroll
        for (int i = 1; i <= MAXM; i++) {
            if (EXACT || i <= mj) {  // (rows run 1..m: the last one computed is row m)
                const uint32_t left = e[i];  // previous column, same row
                const uint32_t best3 = EXACT ? min(min(diag, e[i - 1] | (1u << 22)), left | (2u << 22))
                                             : min(min(diag, e[i - 1] | (1u << 22) | no_indel_or), left | (2u << 22) | no_indel_or);
                const uint32_t miss = (best3 & ~MIRGE_TRIM_CHOICE_MASK) + (1u << MIRGE_TRIM_COST_SHIFT);
                const bool same = (eqm >> (i - 1)) & 1ull;
                const uint32_t v = same ? diag + MIRGE_TRIM_MATCH_ONE : miss;
                diag = left;
                e[i] = v;
                last = v;
            }
        }
        if (FRONT || !anch || j == n) consider(last, m, j, nw_m);
        exact = found && b_cost == 0 && b_mat == m;
    }
    if (!exact && !FRONT && !anch) {  // the adapter may run off the read's end: every prefix of it, in the last column, longest first
        int nw_i = nw_m;
#pragma unroll                // (cutadapt: `for i in reversed(range(first_i, m + 1))` -- on equal (matches, cost) the longer prefix stays)
        for (int i = MAXM; i >= 0; i--)
            if (EXACT || i <= m) {
                consider(e[i], i, n, nw_i);
                if (i >= 1) nw_i -= (int)((wildm >> (i - 1)) & 1ull);
            }
    }
    if (hit) { hit[0] = found ? 1 : 0; hit[1] = b_mat; hit[2] = b_cost; }
    return found ? b_val : (FRONT ? 0 : n);
}

// lstart/lend: the sequence line of every record (after '\r' stripping here); qstart/qend: its quality line (FASTQ) or null.
// vstart/vend[r * stages_out + s]: the read after modifier s (stages_out == n_mods) or after the last one.
// flags[5] |= 1: a record whose quality line is not as long as its sequence line (dnaio raises on it; here q[i] would
// otherwise be taken from the next record's bytes).
// (registers: the launch is MIRGE_BLOCK threads; the general 3' instance -- four inlined DP variants -- and the exact instances of
//  more than 40 rows may take up to 256 registers (two waves per SIMD): a DP column in scratch costs 8 x the kernel's time)
template <int MAXM, bool EXACT, bool FRONT>
__global__ void __launch_bounds__(MIRGE_BLOCK) __attribute__((amdgpu_waves_per_eu((!EXACT && !FRONT) || MAXM > 40 ? 2 : 4, 8)))
k_trim(const uint8_t* __restrict__ text, const int64_t* __restrict__ lstart, const int64_t* __restrict__ lend,
                       const int64_t* __restrict__ qstart, const int64_t* __restrict__ qend, uint32_t n_seq, TrimOpts o,
                       int64_t* __restrict__ vstart, int64_t* __restrict__ vend, uint32_t* __restrict__ flags) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_seq; r += gridDim.x * blockDim.x) {
        int64_t b = lstart[r], e = lend[r];
        if (e > b && text[e - 1] == 13) e--;
        const uint8_t* q = qstart ? text + qstart[r] : nullptr;  // quality of base k of the LINE: q[k]
        if (q && qend) {
            int64_t qe = qend[r];
            if (qe > qstart[r] && text[qe - 1] == 13) qe--;
            if (qe - qstart[r] != e - b) { atomicOr(&flags[5], 1u); q = nullptr; }
        }
        int a0 = 0, a1 = (int)((e - b) > 0x7F00 ? 0x7F00 : (e - b));  // current read = line[a0, a1)
        const uint8_t* s = text + b;
        int stage = 0;
        auto emit = [&]() {
            if (o.stages_out > 1) { vstart[(size_t)r * o.stages_out + stage] = b + a0; vend[(size_t)r * o.stages_out + stage] = b + a1; }
            stage++;
        };
        if (o.nextseq >= 0 && q) {
            int sum = 0, mx = 0, stop = a1;
            for (int i = a1 - 1; i >= a0; i--) {
                int qq = (int)q[i] - o.base;
                if (s[i] == 'G') qq = o.nextseq - 1;  // (cutadapt's `bases[i] == 'G'`: a lower-case g keeps its own quality)
                sum += o.nextseq - qq;
                if (sum < 0) break;
                if (sum > mx) { mx = sum; stop = i; }
            }
            a1 = stop;
            emit();
        }
        if (o.q_back >= 0 && q) {
            int sum = 0, mx = 0, st = a0;
            for (int i = a0; i < a1; i++) {
                sum += o.q_front - ((int)q[i] - o.base);
                if (sum < 0) break;
                if (sum > mx) { mx = sum; st = i + 1; }
            }
            int stop = a1;
            sum = 0; mx = 0;
            for (int i = a1 - 1; i >= a0; i--) {
                sum += o.q_back - ((int)q[i] - o.base);
                if (sum < 0) break;
                if (sum > mx) { mx = sum; stop = i; }
            }
            if (st >= stop) { st = a0; stop = a0; }  // (0, 0) of the current read: nothing left
            a0 = st; a1 = stop;
            emit();
        }
        if (o.alen > 0) {
            bool done = false;
            if constexpr (!EXACT && !FRONT) {
                // The general 3' instance (round 6: ONE copy of the search in its code -- the linked form, the best-of-two form and the
                // plain one-adapter form each had their own, with their own DP column: 1 149 spilled scalar registers, 2.7 x the exact
                // instance's time).  Per round up to two searches, then what the form makes of them:
                //   linked (LinkedAdapter.match_to): the 5' part first; required and absent = no match; the 3' part in what follows
                //     the 5' match (the whole read when an optional 5' part is absent); no 3' match is still a match when that part is
                //     optional and the 5' part was found; a match removes whichever parts were found
                //   otherwise (AdapterCutter: `for _ in range(times): best_match ... break if None`): one or two adapters of either
                //     kind, the better match removed -- most matches, then fewest errors, then the first given
                if (!o.action_none) {
                    const AdapterMasks am1 = adapter_masks(o.adapter, o.wild, o.alen), am2 = adapter_masks(o.adapter2, o.wild2, o.alen2);
                    const int n_search = (o.linked || o.alen2 > 0) ? 2 : 1;
                    for (int it = 0; it < (o.times > 1 ? o.times : 1); it++) {
                        int f1 = 0, m1 = 0, c1 = 0, v1 = 0, f2 = 0, m2 = 0, c2 = 0, v2 = 0;
                        int na0 = a0;
                        bool none = false;
                        for (int k = 0; k < n_search && !none; k++) {
                            const int from = (o.linked && k) ? na0 : a0;
                            const bool fr = o.linked ? k == 0 : (k ? o.front2 : o.front) != 0;
                            int hh[3] = {0, 0, 0};
                            const int v = adapter_cut_point_rt<MAXM>(o, am1, am2, s + from, a1 - from, fr, k, hh);
                            if (k) { f2 = hh[0]; m2 = hh[1]; c2 = hh[2]; v2 = v; } else { f1 = hh[0]; m1 = hh[1]; c1 = hh[2]; v1 = v; }
                            if (o.linked && k == 0) {
                                if (!f1 && o.req1) none = true;
                                else na0 = f1 ? a0 + v1 : a0;
                            }
                        }
                        if (o.linked) {
                            if (none || (!f2 && (o.req2 || !f1))) break;
                            a0 = na0;
                            if (f2) a1 = na0 + v2;
                        } else {
                            const bool second = f2 && (!f1 || m2 > m1 || (m2 == m1 && c2 < c1));
                            if (!(second ? f2 : f1)) break;
                            const bool fr = second ? o.front2 != 0 : o.front != 0;
                            const int v = second ? v2 : v1;
                            if (fr) a0 = a0 + v; else a1 = a0 + v;
                        }
                    }
                }
                done = true;
            }
            if (!done) {
                if (FRONT) a0 = a0 + adapter_cut_point<MAXM, EXACT, true>(o, s + a0, a1 - a0);
                else a1 = a0 + adapter_cut_point<MAXM, EXACT, false>(o, s + a0, a1 - a0);
            }
            emit();
        }
        if (o.trim_n) {
            while (a0 < a1 && (s[a0] & 0xDF) == 'N') a0++;
            while (a1 > a0 && (s[a1 - 1] & 0xDF) == 'N') a1--;
            emit();
        }
        for (int k = 0; k < o.n_cut; k++) {
            const int cval = o.cut[k];
            if (cval > 0) a0 = min(a0 + cval, a1);
            else if (cval < 0) a1 = max(a1 + cval, a0);
            if (cval != 0) emit();
        }
        if (o.stages_out <= 1) { vstart[r] = b + a0; vend[r] = b + a1; }
    }
}
