// native_gz.hpp -- pure host (no HIP): a gzip member inflated on all cores.
//
// Why it is here: a user's sample is `sample.fastq.gz` (mirge/libs/digest.py:136-140 reads it through xopen, which hands the
// inflation to pigz / igzip threads when it finds them).  Everything behind the text takes 0.07 s per 10 M-read sample on the
// GPU; ONE zlib inflate stream takes 1.2 s for the same sample (bench.py cli_path.gz_libraries_resident) -- the input, not the
// path, is what a run from compressed FASTQ waits for.  A deflate stream has no index, but it can still be cut:
//   1. the compressed bytes are cut into chunks; in every chunk but the first a thread FINDS the start of a deflate block (a bit
//      position where a dynamic-Huffman header parses into complete codes, the block decodes into text and is followed by another
//      well-formed header);
//   2. every chunk is decoded from its block start to the next chunk's, in parallel, WITHOUT the 32 KiB of history it may refer
//      to: output symbols are 16 bits, a byte, or -- for a copy that reaches back before the chunk -- the index of the unknown
//      history byte (copies of such symbols copy the symbol);
//   3. the last 32 KiB of every chunk are resolved in order (chunk 0 has no history, so its end is known; that is chunk 1's
//      history, ...: 32 K table lookups per chunk, handed from worker to worker), and every worker turns its chunk into bytes
//      at its offset of the text as soon as it knows its history;
//   4. the CRC-32 and length of the gzip trailer must match (per-chunk CRCs combined), else the caller inflates the ordinary way.
// (The approach of pugz / rapidgzip, written from the deflate specification RFC 1951.)  BGZF files (bgzip: members of <= 64 KiB
// that carry their size) are inflated member by member on the same threads with zlib; a file of several ordinary members (lanes
// merged with `cat`) member after member, each cut as above when it is large enough.
// Nothing here decides an answer: a text that fails any check is simply inflated serially by the caller (collapse.GzipRecordStream).
#pragma once
#include <immintrin.h>
#include <sys/mman.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace mirge_gz {

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf = 0;
    int nbits = 0;  // valid bits in buf; negative once bits behind the end of the data have been consumed (they read as zeros)
    BitReader(const uint8_t* data, size_t n, uint64_t start_bit) : p(data + std::min<size_t>((size_t)(start_bit >> 3), n)), end(data + n) {
        fill();
        drop((int)(start_bit & 7));
    }
    inline void fill() {
        if (end - p >= 8 && nbits >= 0) {  // eight bytes at once: the whole ones that fit are counted, the partial one is OR-ed again next time
            uint64_t w;
            std::memcpy(&w, p, 8);
            buf |= w << nbits;
            const int adv = (63 - nbits) >> 3;
            p += adv; nbits += adv * 8;
            return;
        }
        while (nbits <= 56 && p < end) { buf |= (uint64_t)(*p++) << nbits; nbits += 8; }
    }
    inline uint32_t peek(int n) { if (nbits < n) fill(); return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    inline void drop(int n) { buf >>= n; nbits -= n; }
    inline uint32_t bits(int n) { const uint32_t v = peek(n); drop(n); return v; }
    inline bool ran_off() const { return nbits < 0; }
};

// position (in bits from the start of `data`) of the next unread bit
static inline uint64_t bit_position(const BitReader& br, const uint8_t* data) {
    return (uint64_t)((int64_t)(br.p - data) * 8 - (int64_t)br.nbits);
}

#define MIRGE_GZ_PRIMARY 10  // bits of the first-level table (4 KiB: it stays in L1; a flat 2^15-entry table does not)
struct Huff {
    // first level: indexed by the next MIRGE_GZ_PRIMARY bits (LSB first).  An entry is (symbol << 4) | code length, or -- bit 31 --
    // a pointer for codes longer than the first level: (offset of a second-level table << 4), indexed by the bits behind them
    // (a fixed array, filled as far as the code needs: the block-start search builds some hundred codes per chunk for candidates
    // that turn out to be noise, and a heap allocation for each was most of what was left of its time.  Room: the first level and
    // one second-level table of 2^(15 - 10) entries for every symbol that could open one.)
    struct Table {
        uint32_t v[((size_t)1 << MIRGE_GZ_PRIMARY) + (size_t)288 * 32];
        size_t used = 0;
        const uint32_t* data() const { return v; }
        uint32_t& operator[](size_t i) { return v[i]; }
        const uint32_t& operator[](size_t i) const { return v[i]; }
        size_t size() const { return used; }
        void assign(size_t n, uint32_t x) { used = n; std::fill(v, v + n, x); }
        bool grow(size_t n, uint32_t x) { if (used + n > sizeof(v) / sizeof(v[0])) return false; std::fill(v + used, v + used + n, x); used += n; return true; }
        void clear() { used = 0; }
    } table;
    int pbits = 0, maxlen = 0;
    bool ok = false;
};

// canonical Huffman code from code lengths (RFC 1951 3.2.2); complete codes only (a single-symbol code is allowed where zlib
// allows it)
static bool build_huff(const uint8_t* lens, int n, Huff& h, bool allow_single) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int maxlen = 0, used = 0;
    for (int l = 1; l <= 15; l++) if (count[l]) { maxlen = l; used += count[l]; }
    h.ok = false;
    if (!used) return false;
    long left = 1;
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
    if (left > 0 && !(allow_single && used == 1 && maxlen == 1)) return false;  // incomplete
    uint32_t next[16]; uint32_t code = 0;
    for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    const int P = std::min(maxlen, MIRGE_GZ_PRIMARY), sub = maxlen - P;
    h.maxlen = maxlen; h.pbits = P;
    h.table.assign((size_t)1 << P, 0u);
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t c = next[l]++, r = 0;
        for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);  // bit-reversed: the stream is LSB first
        const uint32_t e = ((uint32_t)s << 4) | (uint32_t)l;
        if (l <= P) {
            for (uint32_t k = r; k < ((uint32_t)1 << P); k += (uint32_t)1 << l) h.table[k] = e;
        } else {
            const uint32_t pre = r & (((uint32_t)1 << P) - 1u);
            if (!(h.table[pre] & 0x80000000u)) {  // this prefix's second-level table: 2^sub entries behind what is there
                const uint32_t off = (uint32_t)h.table.size();
                if (!h.table.grow((size_t)1 << sub, 0u)) return false;
                h.table[pre] = 0x80000000u | (off << 4);
            }
            const uint32_t off = (h.table[pre] & 0x7FFFFFFFu) >> 4;
            for (uint32_t k = r >> P; k < ((uint32_t)1 << sub); k += (uint32_t)1 << (l - P)) h.table[off + k] = e;
        }
    }
    h.ok = true;
    return true;
}
static inline int decode_sym(BitReader& br, const Huff& h) {
    uint32_t e = h.table[br.peek(h.pbits)];
    if (e & 0x80000000u) e = h.table[((e & 0x7FFFFFFFu) >> 4) + (br.peek(h.maxlen) >> h.pbits)];
    if (!(e & 15u)) return -1;  // (the unused half of a single-symbol code, or a hole of an -- impossible -- incomplete code)
    br.drop((int)(e & 15u));
    return (int)(e >> 4);
}

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// the header of a dynamic block (RFC 1951 3.2.7) behind BFINAL / BTYPE; false: not a well-formed one
static bool read_dynamic_header(BitReader& br, Huff& lit, Huff& dist) {
    const int hlit = (int)br.bits(5) + 257, hdist = (int)br.bits(5) + 1, hclen = (int)br.bits(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) cl[order[i]] = (uint8_t)br.bits(3);
    Huff clh;
    if (!build_huff(cl, 19, clh, false)) return false;
    uint8_t lens[286 + 30];
    int i = 0;
    while (i < hlit + hdist) {
        const int s = decode_sym(br, clh);
        if (s < 0 || br.ran_off()) return false;
        if (s < 16) lens[i++] = (uint8_t)s;
        else {
            int rep, val = 0;
            if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)br.bits(2); }
            else if (s == 17) rep = 3 + (int)br.bits(3);
            else rep = 11 + (int)br.bits(7);
            if (i + rep > hlit + hdist) return false;
            while (rep--) lens[i++] = (uint8_t)val;
        }
    }
    if (!lens[256]) return false;  // no end-of-block code
    if (!build_huff(lens, hlit, lit, true)) return false;
    if (!build_huff(lens + hlit, hdist, dist, true)) {
        // zlib accepts a block without any distance code when it holds literals only
        bool any = false;
        for (int k = 0; k < hdist; k++) any |= lens[hlit + k] != 0;
        if (any) return false;
        dist.ok = false; dist.maxlen = 0; dist.table.clear();
    }
    return true;
}
static void fixed_codes(Huff& lit, Huff& dist) {
    uint8_t l[288];
    for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
    build_huff(l, 288, lit, false);
    uint8_t d[32];
    for (int i = 0; i < 32; i++) d[i] = 5;
    build_huff(d, 32, dist, false);
}

#define MIRGE_GZ_WINDOW 32768
#ifndef MIRGE_GZ_TRIAL_SYMBOLS
#define MIRGE_GZ_TRIAL_SYMBOLS 16384  // symbols of text a candidate block must decode into before the search accepts it
#endif
#ifndef MIRGE_GZ_RATIO_GUESS
#define MIRGE_GZ_RATIO_GUESS 7  // symbols reserved per compressed byte of a chunk (FASTQ at level 6: 4-6); more grows the buffer
#endif
// growable array of symbols WITHOUT value-initialisation (a std::vector zero-fills twice the text's size before it is written),
// in 2 MiB-aligned anonymous mappings that ask for huge pages: a run's first call touches ~3 bytes of fresh memory per byte of
// text, and 4 KiB page faults from every thread at once were 0.7 s of a 0.9 s decode phase
struct SymBuf {
    uint16_t* p = nullptr;
    size_t cap = 0;
    SymBuf() = default;
    SymBuf(const SymBuf&) = delete;
    SymBuf& operator=(const SymBuf&) = delete;
    ~SymBuf() { release(); }
    size_t size() const { return cap; }
    static size_t bytes_for(size_t n) { return (n * sizeof(uint16_t) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); }
    bool resize(size_t n) {
        if (n <= cap) return true;
        const size_t nb = bytes_for(n);
#ifdef MIRGE_GZ_POPULATE
        void* q = ::mmap(nullptr, nb + ((size_t)2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
#else
        void* q = ::mmap(nullptr, nb + ((size_t)2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
#endif
        if (q == MAP_FAILED) return false;
        // keep the 2 MiB-aligned part
        uint8_t* base = (uint8_t*)q;
        uint8_t* al = (uint8_t*)(((uintptr_t)base + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1));
        if (al > base) ::munmap(base, (size_t)(al - base));
        const size_t tail = (size_t)((base + nb + ((size_t)2 << 20)) - (al + nb));
        if (tail) ::munmap(al + nb, tail);
#if defined(MADV_HUGEPAGE) && !defined(MIRGE_GZ_NOHUGE)
        (void)::madvise(al, nb, MADV_HUGEPAGE);
#endif
        if (p) { std::memcpy(al, p, cap * sizeof(uint16_t)); ::munmap(p, bytes_for(cap)); }
        p = (uint16_t*)al; cap = nb / sizeof(uint16_t);
        return true;
    }
    void release() { if (p) ::munmap(p, bytes_for(cap)); p = nullptr; cap = 0; }
    // Buffers of finished calls are kept for the next one (a process inflates sample after sample): 64 workers touching 16 MiB of
    // fresh memory each is a gigabyte of page faults per file.  At most MIRGE_GZ_KEEP_SYM_BYTES (default 2 GiB) stay mapped.
    struct Pool {
        std::mutex mu;
        std::vector<std::pair<uint16_t*, size_t>> kept;
        size_t bytes = 0, limit = (size_t)2 << 30;
        Pool() { if (const char* e = std::getenv("MIRGE_GZ_KEEP_SYM_BYTES")) limit = (size_t)std::strtoull(e, nullptr, 10); }
    };
    static Pool& pool() { static Pool* pl = new Pool; return *pl; }  // (never destroyed: threads may still give back at exit)
    void take(bool largest = true) {  // the largest (a chunk's decode) or the smallest (a block-start search) kept buffer, if any
        Pool& pl = pool();
        std::lock_guard<std::mutex> g(pl.mu);
        if (p || pl.kept.empty()) return;
        size_t best = 0;
        for (size_t i = 1; i < pl.kept.size(); i++) if ((pl.kept[i].second > pl.kept[best].second) == largest && pl.kept[i].second != pl.kept[best].second) best = i;
        p = pl.kept[best].first; cap = pl.kept[best].second;
        pl.bytes -= bytes_for(cap);
        pl.kept[best] = pl.kept.back();
        pl.kept.pop_back();
    }
    void give() {
        if (!p) return;
        Pool& pl = pool();
        {
            std::lock_guard<std::mutex> g(pl.mu);
            if (pl.bytes + bytes_for(cap) <= pl.limit) {
                pl.kept.emplace_back(p, cap);
                pl.bytes += bytes_for(cap);
                p = nullptr; cap = 0;
                return;
            }
        }
        release();
    }
    uint16_t& operator[](size_t i) { return p[i]; }
    const uint16_t& operator[](size_t i) const { return p[i]; }
    uint16_t* data() { return p; }
};
// One deflate block decoded into 16-bit symbols appended to `out`: a byte, or 256 + (index into the MIRGE_GZ_WINDOW bytes in front
// of out[0]) for what a copy took from there.  known_history: out[0] is the stream's first byte (reaching back is an error).
// text_only: a byte outside printable ASCII / tab / line ends is an error (the block-start search's plausibility test).
// Returns 0 = block done, 1 = it was the final block, < 0 = malformed.
// `out` is kept AHEAD of the symbols written (`o`, the caller's running count): it grows geometrically and never shrinks here.
static int decode_block(BitReader& br, SymBuf& out, size_t& o, bool known_history, bool text_only, size_t max_out) {
    const uint32_t bfinal = br.bits(1), btype = br.bits(2);
    if (btype == 3) return -1;
    if (btype == 0) {
        br.drop(br.nbits & 7);  // to the byte boundary
        const uint32_t len = br.bits(16), nlen = br.bits(16);
        if ((len ^ nlen) != 0xFFFFu || br.ran_off()) return -1;
        if (o + len > max_out) return -1;
        if (o + len > out.size() && !out.resize(std::max(out.size() * 2, o + len + (size_t)(1 << 16)))) return -1;
        for (uint32_t k = 0; k < len; k++) {
            const uint32_t c = br.bits(8);
            if (text_only && !(c == 9 || c == 10 || c == 13 || (c >= 32 && c < 127))) return -1;
            out[o++] = (uint16_t)c;
        }
        if (br.ran_off()) return -1;
        return (int)bfinal;
    }
    Huff lit, dist;
    if (btype == 2) { if (!read_dynamic_header(br, lit, dist)) return -1; }
    else fixed_codes(lit, dist);
    auto done = [&](int rc) { return rc; };
    const uint32_t* const lt = lit.table.data();
    const uint32_t* const dt = dist.ok ? dist.table.data() : nullptr;
    const uint32_t lpm = ((uint32_t)1 << lit.pbits) - 1u, lmm = ((uint32_t)1 << lit.maxlen) - 1u;
    const uint32_t dpm = dist.ok ? ((uint32_t)1 << dist.pbits) - 1u : 0u, dmm = dist.ok ? ((uint32_t)1 << dist.maxlen) - 1u : 0u;
    const int lpb = lit.pbits, dpb = dist.pbits;
    for (;;) {
        if (o + 512 > max_out) return done(text_only ? -3 : -1);  // -3: a trial decode reached its limit without a fault
        if (o + 512 > out.size() && !out.resize(std::max(out.size() * 2, o + (size_t)(1 << 16)))) return done(-1);
        if (br.ran_off()) return done(-1);
        if (!text_only && br.nbits >= 0 && br.end - br.p >= 16) {
            // The fast loop (the real decode; the block-start search keeps to the careful step below): the reader's state in
            // locals, ONE refill per symbol or match -- 56 bits cover a length code, its extra bits, a distance code and its extra
            // bits (15 + 5 + 15 + 13) --, no per-symbol bounds checks: it runs while 16 bytes of input and 300 symbols of room
            // are left and hands the rest to the careful step.
            const uint8_t* p = br.p;
            const uint8_t* const safe_end = br.end - 16;
            uint64_t buf = br.buf;
            int nb = br.nbits;
            uint16_t* const w = out.data();
            const size_t o_lim = out.size() - 300;
            int rc = 2;  // 2 = out of input or room: not an end
            while (p <= safe_end && o < o_lim) {
                {
                    uint64_t x;
                    std::memcpy(&x, p, 8);
                    buf |= x << nb;
                    const int adv = (63 - nb) >> 3;
                    p += adv;
                    nb += adv * 8;
                }
                uint32_t e = lt[(uint32_t)buf & lpm];
                if (e & 0x80000000u) e = lt[((e & 0x7FFFFFFFu) >> 4) + (((uint32_t)buf & lmm) >> lpb)];
                const int l = (int)(e & 15u);
                if (!l) { rc = -1; break; }
                buf >>= l; nb -= l;
                const int s = (int)(e >> 4);
                if (s < 256) { w[o++] = (uint16_t)s; continue; }
                if (s == 256) { rc = (int)bfinal; break; }
                if (s > 285 || !dt) { rc = -1; break; }
                const int xl = kLenExtra[s - 257];
                const int len = kLenBase[s - 257] + (int)((uint32_t)buf & (((uint32_t)1 << xl) - 1u));
                buf >>= xl; nb -= xl;
                uint32_t de = dt[(uint32_t)buf & dpm];
                if (de & 0x80000000u) de = dt[((de & 0x7FFFFFFFu) >> 4) + (((uint32_t)buf & dmm) >> dpb)];
                const int dl = (int)(de & 15u), ds = (int)(de >> 4);
                if (!dl || ds > 29) { rc = -1; break; }
                buf >>= dl; nb -= dl;
                const int xd = kDistExtra[ds];
                const size_t d = (size_t)kDistBase[ds] + (size_t)((uint32_t)buf & (((uint32_t)1 << xd) - 1u));
                buf >>= xd; nb -= xd;
                if (d > o + (known_history ? 0 : MIRGE_GZ_WINDOW)) { rc = -1; break; }
                if (o >= d) {
                    const uint16_t* src = w + (o - d);
                    uint16_t* dst = w + o;
                    for (int k = 0; k < len; k++) dst[k] = src[k];  // forward, element by element: the ranges may overlap (d < len repeats)
                    o += (size_t)len;
                } else {
                    for (int k = 0; k < len; k++, o++)
                        w[o] = o >= d ? w[o - d] : (uint16_t)(256 + (MIRGE_GZ_WINDOW + o - d));
                }
            }
            br.p = p; br.buf = buf; br.nbits = nb;
            if (rc != 2) return done(rc);
            if (o + 512 > out.size()) continue;  // room first, then on
        }
        const int s = decode_sym(br, lit);
        if (s < 0) return done(-1);
        if (s < 256) {
            if (text_only && !(s == 9 || s == 10 || s == 13 || (s >= 32 && s < 127))) return done(-1);
            out[o++] = (uint16_t)s;
            continue;
        }
        if (s == 256) return done((int)bfinal);
        if (s > 285 || !dist.ok) return done(-1);
        const int len = kLenBase[s - 257] + (int)br.bits(kLenExtra[s - 257]);
        const int ds = decode_sym(br, dist);
        if (ds < 0 || ds > 29) return done(-1);
        const size_t d = (size_t)kDistBase[ds] + br.bits(kDistExtra[ds]);
        if (d > o + (known_history ? 0 : MIRGE_GZ_WINDOW)) return done(-1);
        uint16_t* w = out.data();
        for (int k = 0; k < len; k++, o++)
            w[o] = o >= d ? w[o - d] : (uint16_t)(256 + (MIRGE_GZ_WINDOW + o - d));  // before out[0]: an index into the history
    }
}

// is there a well-formed block header at this bit position (the check that follows a candidate block in the search)?
static bool plausible_header(const uint8_t* data, size_t n, uint64_t bit) {
    BitReader br(data, n, bit);
    (void)br.bits(1);
    const uint32_t btype = br.bits(2);
    if (btype == 3) return false;
    if (btype == 1) return true;
    if (btype == 0) { br.drop(br.nbits & 7); const uint32_t len = br.bits(16), nlen = br.bits(16); return (len ^ nlen) == 0xFFFFu; }
    Huff a, b;
    return read_dynamic_header(br, a, b);
}

// Kraft sums (in 1/128ths) of four 3-bit code lengths at once: the code-length code of a dynamic header is complete iff its
// (up to 19) lengths sum to 128
struct KraftTable {
    uint16_t v[4096];
    KraftTable() {
        for (uint32_t x = 0; x < 4096; x++) {
            uint32_t k = 0;
            for (int j = 0; j < 4; j++) { const uint32_t l = (x >> (3 * j)) & 7u; if (l) k += 128u >> l; }
            v[x] = (uint16_t)k;
        }
    }
};
// first bit position >= from_bit (and < to_bit) at which a non-final dynamic block starts, decodes into MIRGE_GZ_TRIAL_SYMBOLS symbols
// of text -- or, when it is shorter, into text to its end with a well-formed header behind it --; UINT64_MAX if none
static uint64_t find_block_start(const uint8_t* data, size_t n, uint64_t from_bit, uint64_t to_bit) {
    static const KraftTable kraft_of;
    SymBuf tmp;
    tmp.take(false);  // (a mapping of its own per search was most of the search phase: 180 mmap / fault / munmap rounds at once)
    struct GiveBack { SymBuf& b; ~GiveBack() { b.give(); } } give_back{tmp};
    for (uint64_t byte = from_bit >> 3; byte * 8 < to_bit; byte++) {
        if (byte + 8 >= n) break;
        uint64_t w;
        std::memcpy(&w, data + byte, 8);
        // bit s of `cand`: the three bits at shift s read BFINAL = 0, BTYPE = 10b (LSB first: 0 | 0 1)
        uint32_t cand = (uint32_t)(~w & ~(w >> 1) & (w >> 2)) & 0xFFu;
        if (byte == (from_bit >> 3)) cand &= 0xFFu << (from_bit & 7);
        while (cand) {
            const unsigned sh = (unsigned)__builtin_ctz(cand);
            cand &= cand - 1;
            const uint64_t bit = byte * 8 + sh;
            if (bit >= to_bit) break;
            if (byte + 16 <= n) {
                // the cheap part of the header test, on two words of the data and without a reader or a table: HLIT / HDIST in
                // range and the code-length code complete (what build_huff would find out) -- 99 % of the candidates end here
                uint64_t w2;
                std::memcpy(&w2, data + byte + 8, 8);
                const unsigned __int128 full = (((unsigned __int128)w2 << 64) | (unsigned __int128)w) >> sh;
                const uint32_t lo = (uint32_t)full;
                const uint32_t hlit = (lo >> 3) & 31u, hdist = (lo >> 8) & 31u, hclen = ((lo >> 13) & 15u) + 4u;
                if (hlit > 29u || hdist > 29u) continue;
                const uint64_t cl = (uint64_t)(full >> 17) & ((1ull << (3 * hclen)) - 1ull);  // the 3-bit lengths of the code-length code (<= 57 bits)
                const uint32_t kraft = (uint32_t)kraft_of.v[cl & 4095u] + kraft_of.v[(cl >> 12) & 4095u] + kraft_of.v[(cl >> 24) & 4095u] +
                                       kraft_of.v[(cl >> 36) & 4095u] + kraft_of.v[(cl >> 48) & 4095u];
                if (kraft != 128u) continue;
            }
            BitReader br(data, n, bit);
            size_t to = 0;
            // (the trial decode stops after MIRGE_GZ_TRIAL_SYMBOLS symbols of text: enough to tell a block from noise, and the
            // real decode -- and in the end the CRC-32 -- check the rest; a whole block is ~150 k symbols, a third of the
            // search's time)
            const int rc = decode_block(br, tmp, to, false, true, (size_t)MIRGE_GZ_TRIAL_SYMBOLS);
            if (rc == 0) {
                if (to < 1024) continue;  // (a real block of a text file holds thousands of symbols)
                const uint64_t next = bit_position(br, data);
                if (next + 64 > (uint64_t)n * 8 || !plausible_header(data, n, next)) continue;
            } else if (rc != -3) continue;
            return bit;
        }
    }
    return ~0ull;
}

struct GzHeader { size_t body = 0; bool bgzf = false; size_t bsize = 0; };
static bool parse_gz_header(const uint8_t* p, size_t n, GzHeader& h) {
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return false;
    const uint8_t flg = p[3];
    if (flg & 0xe0) return false;  // reserved bits: zlib refuses such a header, so does this route
    size_t at = 10;
    if (flg & 4) {
        if (at + 2 > n) return false;
        const size_t xlen = p[at] | (p[at + 1] << 8);
        at += 2;
        if (at + xlen > n) return false;
        for (size_t q = at; q + 4 <= at + xlen;) {  // subfields: SI1 SI2 LEN data
            const size_t sl = p[q + 2] | (p[q + 3] << 8);
            if (p[q] == 'B' && p[q + 1] == 'C' && sl == 2 && q + 6 <= at + xlen) { h.bgzf = true; h.bsize = (size_t)(p[q + 4] | (p[q + 5] << 8)) + 1; }
            q += 4 + sl;
        }
        at += xlen;
    }
    if (flg & 8) { while (at < n && p[at]) at++; at++; }
    if (flg & 16) { while (at < n && p[at]) at++; at++; }
    if (flg & 2) {  // FHCRC: the low 16 bits of the CRC-32 of the header so far, checked as zlib checks it
        if (at + 2 > n || (uint16_t)(crc32(0L, p, (uInt)at) & 0xffff) != (uint16_t)(p[at] | (p[at + 1] << 8))) return false;
        at += 2;
    }
    if (at >= n) return false;
    h.body = at;
    return true;
}

template <typename F>
static void parallel_for(int n_items, int threads, F&& fn) {
    std::atomic<int> next{0};
    auto work = [&]() { for (int i; (i = next.fetch_add(1)) < n_items;) fn(i); };
    std::vector<std::thread> th;
    for (int t = 1; t < std::min(threads, n_items); t++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
}

// a BGZF / multi-member file whose members carry their sizes: every member on its own with zlib
static int inflate_bgzf(const uint8_t* gz, size_t n, uint8_t* out, size_t cap, size_t* n_out, int threads) {
    struct Member { size_t at, size, body, isize, out_at; };
    std::vector<Member> ms;
    size_t at = 0, total = 0;
    while (at < n) {
        GzHeader h;
        if (!parse_gz_header(gz + at, n - at, h) || !h.bgzf || at + h.bsize > n || h.bsize < h.body + 8) return -1;
        const uint8_t* tr = gz + at + h.bsize - 4;
        const size_t isize = (size_t)tr[0] | ((size_t)tr[1] << 8) | ((size_t)tr[2] << 16) | ((size_t)tr[3] << 24);
        ms.push_back(Member{at, h.bsize, h.body, isize, total});
        total += isize;
        at += h.bsize;
    }
    if (total > cap) return -2;
    std::atomic<int> bad{0};
    // members are small (64 KiB): a thread takes runs of them
    const int run = 64, n_runs = (int)((ms.size() + run - 1) / run);
    parallel_for(n_runs, threads, [&](int r) {
        for (size_t k = (size_t)r * run; k < std::min(ms.size(), (size_t)(r + 1) * run); k++) {
            const Member& m = ms[k];
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            zs.next_in = const_cast<Bytef*>(gz + m.at + m.body);
            zs.avail_in = (uInt)(m.size - m.body - 8);
            zs.next_out = out + m.out_at;
            zs.avail_out = (uInt)m.isize;
            const int rc = inflate(&zs, Z_FINISH);
            const bool ok = rc == Z_STREAM_END && zs.total_out == m.isize;
            inflateEnd(&zs);
            const uint8_t* tr = gz + m.at + m.size - 8;
            const uint32_t want = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
            if (!ok || (uint32_t)crc32(0L, out + m.out_at, (uInt)m.isize) != want) { bad = 1; return; }
        }
    });
    if (bad) return -1;
    *n_out = total;
    return 0;
}

// CRC-32 (gzip polynomial, reflected) by carry-less multiplication: Gopal et al., "Fast CRC Computation for Generic Polynomials
// Using PCLMULQDQ Instruction" (Intel, 2009).  64 bytes are folded per step with the constants x^(512+64) mod P, x^512 mod P
// (bit-reflected), then 4 -> 1 lanes with x^(128+64), x^128, then 128 -> 64 -> 32 bits and a Barrett reduction.

// n >= 64, n % 16 == 0
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul_blocks(const uint8_t* buf, size_t len, uint32_t crc) {
    // constants for the reflected polynomial 0xEDB88320
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};  // fold by 512 bits
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};  // fold by 128 bits
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};  // 96 -> 64 bits
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};  // P' and mu for the Barrett step
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64;
    len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    // four lanes -> one
    x0 = _mm_load_si128((const __m128i*)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    // the remaining 16-byte blocks
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // Barrett reduction 64 -> 32 bits
    x0 = _mm_load_si128((const __m128i*)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

// CRC-32 (the gzip polynomial) of n bytes: carry-less multiplication where the CPU has it (5 GB/s against the 1 GB/s of this
// image's zlib -- the checksum was 40 % of a chunk's time), zlib for the tail and for CPUs without it
static uint32_t crc32_bytes(const uint8_t* p, size_t n) {
    uint32_t c = 0;
    static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !std::getenv("MIRGE_GZ_NO_CLMUL");
    if (clmul && n >= 64) {
        const size_t m = n & ~(size_t)15;
        c = ~crc32_clmul_blocks(p, m, ~c);
        p += m;
        n -= m;
    }
    for (size_t done = 0; done < n;) {  // zlib's crc32 takes a 32-bit length
        const size_t m = std::min<size_t>(n - done, (size_t)1 << 30);
        c = (uint32_t)crc32(c, p + done, (uInt)m);
        done += m;
    }
    return c;
}

// MIRGE_GZ_TIMING=1: the phases of every member inflated in parallel, to stderr (seconds from the call's start)
static bool gz_timing() { static const bool on = std::getenv("MIRGE_GZ_TIMING") != nullptr; return on; }
static double gz_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one ordinary member (the usual sample.fastq.gz), cut into chunks as described at the top
// *consumed = bytes of `gz` the member occupies (another member, or padding, may follow)
// `progress` (may be null): advanced -- with release order -- to progress_base + the number of leading bytes of `out` that are final
// (decoded, their history resolved, written); the CRC is only known at the end, so a consumer that reads ahead must be ready to
// throw everything away when the call fails.
static int inflate_member_parallel(const uint8_t* gz, size_t n, uint8_t* out, size_t cap, size_t* n_out, size_t* consumed, int threads,
                                   int64_t* progress = nullptr, int64_t progress_base = 0) {
    GzHeader h;
    if (!parse_gz_header(gz, n, h) || n < h.body + 8) return -1;
    const double t_call = gz_now();
    const uint8_t* d = gz + h.body;           // the deflate stream ... and the trailer behind it, somewhere
    const size_t dn = n - h.body;
    // chunk starts: ~8 per thread, at least 512 KiB of compressed data each
    int n_chunks = (int)std::min<size_t>((size_t)threads * 8, std::max<size_t>(1, dn >> 19));
    if (n_chunks < 2) return -3;  // too small to be worth it: the caller's serial route
    std::vector<uint64_t> start((size_t)n_chunks, ~0ull);
    start[0] = 0;
    parallel_for(n_chunks - 1, threads, [&](int k) {
        const int i = k + 1;
        const uint64_t from = (uint64_t)((dn / (size_t)n_chunks) * (size_t)i) * 8ull, to = (uint64_t)((dn / (size_t)n_chunks) * (size_t)(i + 1)) * 8ull;
        start[(size_t)i] = find_block_start(d, dn, from, to);
    });
    std::vector<uint64_t> st;
    for (uint64_t s : start) if (s != ~0ull) st.push_back(s);
    const int C = (int)st.size();
    if (C < 2) return -3;
    const double t_search = gz_now();
    std::vector<double> t_dec((size_t)C, 0.0), t_link((size_t)C, 0.0), t_fin((size_t)C, 0.0), t_start((size_t)C, 0.0);
    // Workers take the chunks in order, each with ONE symbol buffer it keeps (fresh memory is touched once per worker, not
    // once per chunk): decode chunk i without its history; wait until chunk i - 1 has published where chunk i's bytes go and
    // what its history is -- it usually has, it started earlier --; publish the same for chunk i + 1 (32 K lookups through
    // the history: microseconds, so the chain through all chunks is milliseconds); then turn the chunk's symbols into bytes at
    // their place of the text and take their CRC-32.
    struct Link {  // what chunk i needs from its predecessors
        std::atomic<int> ready{0};
        bool stop = false;             // the member ended in an earlier chunk: this one belongs to what follows it, its work is dropped
        size_t at = 0;                 // offset of the chunk's first byte in the text
        std::vector<uint8_t> hist;     // the MIRGE_GZ_WINDOW bytes in front of it (empty: the stream starts here)
    };
    std::vector<Link> link((size_t)C + 1);
    link[0].ready.store(1);
    std::vector<uint32_t> crcs((size_t)C, 0);
    std::vector<size_t> n_sym((size_t)C, 0);
    uint64_t final_end_bit = 0;
    int C_eff = -1;  // chunks of THIS member (the data may go on with another member: its chunks are found, decoded and dropped)
    std::atomic<int> failed{0}, next{0}, over_cap{0};
    std::vector<char> finished((size_t)C, 0);  // chunk i's bytes are in place (under front_mu)
    int front = 0;                              // chunks 0 .. front - 1 are
    std::mutex front_mu;
    auto worker = [&]() {
        SymBuf o;
        o.take();
        struct GiveBack { SymBuf& b; ~GiveBack() { b.give(); } } give_back{o};
        for (int i; (i = next.fetch_add(1)) < C;) {
            int st_i = 0;  // 0 ok, 1 ended with the final block, < 0 failed
            size_t no = 0;
            uint64_t endb = 0;
            t_start[(size_t)i] = gz_now();
            if (!failed.load()) {
                BitReader br(d, dn, st[(size_t)i]);
                const uint64_t stop = i + 1 < C ? st[(size_t)i + 1] : ~0ull;
                if (!o.resize((size_t)((i + 1 < C ? (stop - st[(size_t)i]) / 8 : dn - st[(size_t)i] / 8) * MIRGE_GZ_RATIO_GUESS + (1 << 16)))) st_i = -1;
                while (st_i == 0) {
                    // no chunk can be longer than the whole text: a hostile or very repetitive stream (deflate reaches 1032:1)
                    // stops at cap + 1 symbols instead of mapping gigabytes per worker before the size test below
                    const int rc = decode_block(br, o, no, i == 0, false, std::min<size_t>(cap + 1, (size_t)1 << 36));
                    if (rc < 0 && no + 512 > cap) over_cap.store(1);
                    const uint64_t pos = bit_position(br, d);
                    if (rc < 0) st_i = -1;
                    else if (rc == 1) { st_i = 1; endb = pos; }
                    else if (pos == stop) { endb = pos; break; }
                    else if (pos > stop) st_i = -2;  // the next chunk's start was no block boundary of this stream
                }
            }
            t_dec[(size_t)i] = gz_now();
            // the hand-over happens whatever happened: nobody may wait for ever
            while (!link[(size_t)i].ready.load(std::memory_order_acquire)) std::this_thread::yield();
            const Link& me = link[(size_t)i];
            Link& nx = link[(size_t)i + 1];
            if (me.stop) {  // behind the member's end: not this member's business (a failure here is the next member's to find)
                nx.stop = true;
                nx.ready.store(1, std::memory_order_release);
                continue;
            }
            if (st_i < 0 || (st_i == 0 && i == C - 1)) failed.store(1);  // malformed, or the data ended without a final block
            const bool ok = !failed.load();
            nx.at = me.at + (ok ? no : 0);
            if (ok && nx.at > cap) failed.store(2);
            if (ok && st_i == 1) {  // the final block: the member ends here
                C_eff = i + 1;
                final_end_bit = endb;
                nx.stop = true;
            } else if (ok) {
                nx.hist.assign(MIRGE_GZ_WINDOW, 0);
                for (size_t k = 0; k < MIRGE_GZ_WINDOW; k++) {
                    // byte k of the next history = position (no - WINDOW + k) of this chunk, or of ITS history when the chunk is short
                    const long long q = (long long)no - MIRGE_GZ_WINDOW + (long long)k;
                    if (q >= 0) {
                        const uint16_t sv = o[(size_t)q];
                        if (sv < 256) nx.hist[k] = (uint8_t)sv;
                        else if (!me.hist.empty()) nx.hist[k] = me.hist[(size_t)sv - 256];
                        else failed.store(1);
                    } else if (!me.hist.empty()) nx.hist[k] = me.hist[(size_t)(MIRGE_GZ_WINDOW + q)];
                }
            }
            n_sym[(size_t)i] = no;
            nx.ready.store(1, std::memory_order_release);
            t_link[(size_t)i] = gz_now();
            if (failed.load()) continue;
            uint8_t* ob = out + me.at;
            const uint8_t* hw = me.hist.empty() ? nullptr : me.hist.data();
            bool bad = false;
            {   // symbols -> bytes.  In FASTQ the references into the history do NOT die out -- a header is a copy of the header
                // before it, placeholders included, to the chunk's end -- so the translation is a table look-up per symbol (a byte
                // maps to itself, 256 + q to history byte q: 33 KiB, cache-resident) instead of a branch; runs of 64 plain bytes
                // are narrowed without one (two loops the compiler turns into vector code)
                const uint16_t* sy = o.data();
                std::vector<uint8_t> tab((size_t)256 + MIRGE_GZ_WINDOW, 0);
                for (int v = 0; v < 256; v++) tab[(size_t)v] = (uint8_t)v;
                if (hw) std::memcpy(tab.data() + 256, hw, MIRGE_GZ_WINDOW);
                const uint8_t* tb = tab.data();
                size_t k = 0;
                for (; k + 64 <= no; k += 64) {
                    uint16_t any = 0;
                    for (int j = 0; j < 64; j++) any |= sy[k + (size_t)j];
                    if (any < 256) {
                        for (int j = 0; j < 64; j++) ob[k + (size_t)j] = (uint8_t)sy[k + (size_t)j];
                    } else {
                        if (!hw) { bad = true; break; }  // the stream's first chunk has no history to refer to
                        for (int j = 0; j < 64; j++) ob[k + (size_t)j] = tb[sy[k + (size_t)j]];
                    }
                }
                for (; k < no && !bad; k++) {
                    const uint16_t v = sy[k];
                    if (v >= 256 && !hw) bad = true;
                    else ob[k] = tb[v];
                }
            }
            if (bad) { failed.store(1); continue; }
            crcs[(size_t)i] = crc32_bytes(ob, no);
            t_fin[(size_t)i] = gz_now();
            if (progress) {
                std::lock_guard<std::mutex> g(front_mu);
                finished[(size_t)i] = 1;
                const int before = front;
                while (front < C && finished[(size_t)front]) front++;
                // (link[front].at was published by chunk front - 1 before it wrote its bytes)
                if (front > before) __atomic_store_n(progress, progress_base + (int64_t)link[(size_t)front].at, __ATOMIC_RELEASE);
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < std::min(threads, C); t++) th.emplace_back(worker);
        worker();
        for (auto& x : th) x.join();
    }
    if (gz_timing()) {
        double dec_sum = 0, dec_max = 0, link_last = 0, fin_last = 0, wait_sum = 0, conv_sum = 0;
        for (int i = 0; i < C; i++) {
            dec_sum += t_dec[(size_t)i] - t_start[(size_t)i];
            dec_max = std::max(dec_max, t_dec[(size_t)i] - t_start[(size_t)i]);
            link_last = std::max(link_last, t_link[(size_t)i] - t_call);
            fin_last = std::max(fin_last, t_fin[(size_t)i] - t_call);
            if (t_link[(size_t)i] > 0) wait_sum += t_link[(size_t)i] - t_dec[(size_t)i];
            if (t_fin[(size_t)i] > 0) conv_sum += t_fin[(size_t)i] - t_link[(size_t)i];
        }
        std::fprintf(stderr, "mirge_gz: %d chunks on %d threads: search %.4f, last hand-over %.4f, last chunk in place %.4f, join %.4f | per chunk: decode avg %.4f max %.4f, "
                             "wait+history avg %.4f, bytes+crc avg %.4f | first chunk starts %.4f, last chunk starts %.4f\n",
                     C, threads, t_search - t_call, link_last, fin_last, gz_now() - t_call, dec_sum / C, dec_max, wait_sum / C, conv_sum / C,
                     t_start[0] - t_call, t_start[(size_t)C - 1] - t_call);
    }
    if (failed.load() == 2 || over_cap.load()) return -2;  // the text does not fit `cap` (also: a chunk alone outgrew it): the caller streams the file
    if (failed.load() || C_eff < 1) return -1;
    const size_t total = link[(size_t)C_eff].at;
    // the trailer: CRC-32 and length (mod 2^32) behind the final block's last byte
    const size_t tr = (size_t)((final_end_bit + 7) / 8);
    if (tr + 8 > dn) return -1;
    const uint32_t want_crc = (uint32_t)d[tr] | ((uint32_t)d[tr + 1] << 8) | ((uint32_t)d[tr + 2] << 16) | ((uint32_t)d[tr + 3] << 24);
    const uint32_t want_len = (uint32_t)d[tr + 4] | ((uint32_t)d[tr + 5] << 8) | ((uint32_t)d[tr + 6] << 16) | ((uint32_t)d[tr + 7] << 24);
    if ((uint32_t)total != want_len) return -1;
    uint32_t crc = crcs[0];
    for (int i = 1; i < C_eff; i++) crc = (uint32_t)crc32_combine(crc, crcs[(size_t)i], (z_off_t)n_sym[(size_t)i]);
    if (crc != want_crc) return -1;
    *n_out = total;
    *consumed = h.body + tr + 8;
    return 0;
}

// a member too small to cut (or one the search found no second block start in): zlib, on this thread
static int inflate_member_serial(const uint8_t* gz, size_t n, uint8_t* out, size_t cap, size_t* n_out, size_t* consumed) {
    GzHeader h;
    if (!parse_gz_header(gz, n, h) || n < h.body + 8) return -1;
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return -1;
    size_t in_at = h.body, out_at = 0;
    int rc = Z_OK;
    while (rc == Z_OK) {  // zlib's counters are 32 bits wide: feed and drain in pieces
        const size_t in_m = std::min<size_t>(n - in_at, (size_t)1 << 30), out_m = std::min<size_t>(cap - out_at, (size_t)1 << 30);
        zs.next_in = const_cast<Bytef*>(gz + in_at); zs.avail_in = (uInt)in_m;
        zs.next_out = out + out_at; zs.avail_out = (uInt)out_m;
        rc = inflate(&zs, Z_NO_FLUSH);
        in_at += in_m - zs.avail_in; out_at += out_m - zs.avail_out;
        if (rc == Z_OK && zs.avail_out == 0 && out_at == cap) { inflateEnd(&zs); return -2; }
        if (rc == Z_BUF_ERROR || (rc == Z_OK && in_m - zs.avail_in == 0 && out_m - zs.avail_out == 0)) break;
    }
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || in_at + 8 > n) return -1;
    const uint8_t* tr = gz + in_at;
    const uint32_t want_crc = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
    const uint32_t want_len = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
    uint32_t c = 0;
    for (size_t done = 0; done < out_at;) {
        const size_t m = std::min<size_t>(out_at - done, (size_t)1 << 30);
        c = (uint32_t)crc32(c, out + done, (uInt)m);
        done += m;
    }
    if ((uint32_t)out_at != want_len || c != want_crc) return -1;
    *n_out = out_at;
    *consumed = in_at + 8;
    return 0;
}

}  // namespace mirge_gz

// A whole .gz file's bytes -> its text, on `threads` host threads (0: all).  out[cap]; *n_out = bytes written.
// 0: done, and every member verified against the CRC-32 / length it carries.  Negative: not done -- the file is not of a kind
// this route takes (too small, not text), is damaged, or `cap` is too small (-2): the caller inflates it the ordinary way, which
// also reports what is wrong with a damaged file.
// mirge_gz_inflate_progress: the same, and `progress` (may be null) is advanced while the call runs to the number of leading bytes
// of `out` that are final -- a caller on another thread may upload and parse whole records of that prefix meanwhile.  The file's
// CRC-32 is only verified at the end: when the call fails, whatever was read ahead must be dropped.
extern "C" int mirge_gz_inflate_progress(const uint8_t* gz, int64_t n_gz, uint8_t* out, int64_t cap, int64_t* n_out, int32_t threads, int64_t* progress) {
    if (!gz || n_gz < 18 || !out || cap < 0 || !n_out) return fail(-1, "mirge_gz_inflate: bad argument");
    // (beyond ~64 threads nothing is gained: 82 chunks of a 10 M-read sample decode in 0.35 s on 16, 64 or 256 threads of a
    // 2 x 64-core host -- concurrent first-touch page faults, not decoding, set the floor; profiles/README.md round 4)
    int T = threads > 0 ? threads : (int)std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
    T = std::min(T, 256);
    mirge_gz::GzHeader h;
    if (!mirge_gz::parse_gz_header(gz, (size_t)n_gz, h)) return fail(-1, "mirge_gz_inflate: not a gzip file");
#ifdef MADV_HUGEPAGE
    {   // the caller's buffer is fresh memory too: huge pages for its 2 MiB-aligned interior (a hint; failure changes nothing)
        const uintptr_t lo = ((uintptr_t)out + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1), hi = ((uintptr_t)out + (uintptr_t)cap) & ~(((uintptr_t)2 << 20) - 1);
        if (hi > lo) (void)::madvise((void*)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
    }
#endif
    size_t n = 0;
    int rc = 0;
    if (h.bgzf) rc = mirge_gz::inflate_bgzf(gz, (size_t)n_gz, out, (size_t)cap, &n, T);
    else {
        // member after member (`cat lane1.fastq.gz lane2.fastq.gz > sample.fastq.gz` is how lanes are merged): the large ones cut
        // and decoded in parallel, the small ones by zlib; zero padding behind the last is skipped, anything else is an error
        size_t at = 0;
        bool any_parallel = false;
        while (rc == 0 && at < (size_t)n_gz) {
            if (gz[at] == 0) { at++; continue; }
            size_t got = 0, used = 0;
            rc = mirge_gz::inflate_member_parallel(gz + at, (size_t)n_gz - at, out + n, (size_t)cap - n, &got, &used, T, progress, (int64_t)n);
            if (rc == 0) any_parallel = true;
            else if (rc == -3) rc = mirge_gz::inflate_member_serial(gz + at, (size_t)n_gz - at, out + n, (size_t)cap - n, &got, &used);
            if (rc == 0) {
                n += got; at += used;
                if (progress) __atomic_store_n(progress, (int64_t)n, __ATOMIC_RELEASE);
            }
        }
        if (rc == 0 && !any_parallel) rc = -3;  // nothing here was worth the threads: the caller's streamed route does as well
    }
    if (rc) return fail(rc, "mirge_gz_inflate: not inflated in parallel (code " + std::to_string(rc) + "): the serial route applies");
    *n_out = (int64_t)n;
    if (progress) __atomic_store_n(progress, (int64_t)n, __ATOMIC_RELEASE);
    return 0;
}

extern "C" int mirge_gz_inflate(const uint8_t* gz, int64_t n_gz, uint8_t* out, int64_t cap, int64_t* n_out, int32_t threads) {
    return mirge_gz_inflate_progress(gz, n_gz, out, cap, n_out, threads, nullptr);
}
