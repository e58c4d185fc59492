// hostsim.cpp -- TEST INFRASTRUCTURE ONLY, never linked into libmirge_native.so.
//
// This container has no GPU.  To debug the device arithmetic before spending GPU minutes, the
// exact header the kernels are built from (mirge3.0_amd/csrc/mirge_core.hpp: packing, trimming,
// window verification, pigeonhole probing) and the exact table builder (mirge_libbuild.hpp) are
// compiled here with g++ and driven by a plain loop that does what k_pass does per thread.
// tests/test_hostsim.py compares the outcome with the oracle.  It is not a product path and not
// a fallback: the product fails loudly without the HIP library and a GPU.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../mirge3.0_amd/csrc/mirge_core.hpp"
#include "../../mirge3.0_amd/csrc/mirge_libbuild.hpp"

struct SimLib {
    MirgeHostLib h;
    std::vector<std::vector<uint64_t>> entry;
    std::vector<std::vector<uint32_t>> bucket, pos, bits;
    std::vector<MirgeKTable> tables;
    SimLib() : entry(MIRGE_SHAPE_SLOTS), bucket(MIRGE_SHAPE_SLOTS), pos(MIRGE_SHAPE_SLOTS), bits(MIRGE_SHAPE_SLOTS), tables(MIRGE_SHAPE_SLOTS) {
        for (auto& t : tables) { t.bucket = nullptr; t.pos = nullptr; t.bits = nullptr; }
    }
    MirgeLibView view() {
        MirgeLibView v;
        v.T = h.T.data(); v.inv = h.inv.data(); v.ref_start = h.ref_start.data(); v.tables = tables.data();
        v.total = h.total; v.n_refs = (uint32_t)h.n_refs; v.kmax = h.kmax;
        return v;
    }
};

template <int W>
static bool pack_read(const char* s, int L, MirgeRead<W>& r) {
    for (int i = 0; i < W; i++) { r.w[i] = 0; r.nm[i] = 0; }
    for (int p = 0; p < L; p++) {
        int code = mirge_base_code(s[p]);
        if (code < 0) r.nm[p >> 5] |= 1ull << (2 * (p & 31));
        else r.w[p >> 5] |= (uint64_t)code << (2 * (p & 31));
    }
    r.len = L;
    return true;
}

template <int W>
static void sim_one(const char* s, int L, std::vector<SimLib>& libs, const std::vector<bool>& present,
                    const MirgePolicy* pol, int n_pass, int8_t* o_pass, int32_t* o_ref, int32_t* o_off, int8_t* o_mm) {
    *o_pass = -1; *o_ref = -1; *o_off = -1; *o_mm = -1;
    for (int p = 0; p < n_pass; p++) {
        if (!present[p]) continue;
        MirgeRead<W> r;
        pack_read<W>(s, L, r);
        if (!mirge_effective_read<W>(r, pol[p])) continue;
        // same table-on-demand rule as mirge_cascade_run: every shape the read's probe plan names
        const int np = mirge_probe_count(pol[p], r.len, libs[p].h.kmax, libs[p].h.total);
        for (int q = 0; q < np; q++) {
            MirgeProbe pr;
            mirge_probe_at(pol[p], r.len, libs[p].h.kmax, libs[p].h.total, q, pr);
            if (pr.k1 <= 0) continue;
            const int sid = mirge_shape_id(pr.k1, pr.gap, pr.k2);
            if (!libs[p].tables[sid].bucket) {
                mirge_hostlib_table(libs[p].h, pr.k1, pr.gap, pr.k2, libs[p].bucket[sid], libs[p].entry[sid], libs[p].pos[sid]);
                libs[p].tables[sid].bucket = libs[p].entry[sid].data();  // replaced by the CSR bounds below for a bitmap table
                libs[p].tables[sid].pos = libs[p].pos[sid].data();
                if (pr.k1 + pr.k2 <= 10) {  // same non-empty-bucket bitmap as mirge_native.hip builds
                    const auto& bk = libs[p].bucket[sid];
                    auto& bt = libs[p].bits[sid];
                    bt.assign((bk.size() - 1 + 31) / 32, 0u);
                    for (size_t b = 0; b + 1 < bk.size(); b++) if (bk[b + 1] > bk[b]) bt[b >> 5] |= 1u << (b & 31);
                    libs[p].tables[sid].bits = bt.data();
                    libs[p].tables[sid].bucket = bk.data();
                }
            }
        }
        uint64_t best;
        MirgeLibView v = libs[p].view();
        mirge_align_indexed<W>(v, pol[p], r, best);
        if (best != MIRGE_NO_HIT) {
            const uint32_t g = (uint32_t)best;
            const std::vector<uint32_t>& rs = libs[p].h.ref_start;
            uint32_t lo = 0, hi = (uint32_t)libs[p].h.n_refs;
            while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (rs[mid] <= g) lo = mid; else hi = mid; }
            *o_pass = (int8_t)p; *o_ref = (int32_t)lo; *o_off = (int32_t)(g - rs[lo]); *o_mm = (int8_t)(best >> 32);
            return;
        }
    }
}

extern "C" int hostsim_cascade(const char* reads, const int64_t* roff, int64_t n, const char* const* lib_seq,
                               const int64_t* const* lib_off, const int64_t* lib_n, const MirgePolicy* pol,
                               int32_t n_pass, int8_t* o_pass, int32_t* o_ref, int32_t* o_off, int8_t* o_mm) {
    std::vector<SimLib> libs((size_t)n_pass);
    std::vector<bool> present((size_t)n_pass, false);
    for (int p = 0; p < n_pass; p++) {
        if (!lib_seq[p]) continue;
        std::string err;
        if (mirge_hostlib_build(libs[p].h, lib_seq[p], lib_off[p], lib_n[p], err)) return -1;
        present[p] = true;
    }
    for (int64_t i = 0; i < n; i++) {
        const char* s = reads + roff[i];
        const int L = (int)(roff[i + 1] - roff[i]);
        if (L > MIRGE_MAX_READ_LEN) return -6;
        if (L <= 32) sim_one<1>(s, L, libs, present, pol, n_pass, o_pass + i, o_ref + i, o_off + i, o_mm + i);
        else if (L <= 64) sim_one<2>(s, L, libs, present, pol, n_pass, o_pass + i, o_ref + i, o_off + i, o_mm + i);
        else sim_one<4>(s, L, libs, present, pol, n_pass, o_pass + i, o_ref + i, o_off + i, o_mm + i);
    }
    return 0;
}

// probe plan of a read of (trimmed) length L: out[4*q + {0,1,2,3}] = a1, k1, gap, k2
// scheme: -1 = the family the cost model picks for a library of npos bases, else MIRGE_SCHEME_*
extern "C" int hostsim_probe_plan(int32_t mode, int32_t mm, int32_t seedlen, int32_t L, int32_t K, int64_t npos,
                                  int32_t scheme, int8_t* out) {
    MirgePolicy p;
    std::memset(&p, 0, sizeof(p));
    p.mode = mode; p.mm = mm; p.seedlen = seedlen; p.maxtotal = mode == 0 ? 2 : mm;
    const int n = mirge_probe_count(p, L, K, (uint64_t)npos, scheme);
    for (int q = 0; q < n; q++) {
        MirgeProbe pr;
        mirge_probe_at(p, L, K, (uint64_t)npos, q, pr, scheme);
        out[4 * q] = pr.a1; out[4 * q + 1] = pr.k1; out[4 * q + 2] = pr.gap; out[4 * q + 3] = pr.k2;
    }
    return n;
}

extern "C" int hostsim_plan_scheme(int32_t mode, int32_t mm, int32_t seedlen, int32_t L, int32_t K, int64_t npos) {
    MirgePolicy p;
    std::memset(&p, 0, sizeof(p));
    p.mode = mode; p.mm = mm; p.seedlen = seedlen; p.maxtotal = mode == 0 ? 2 : mm;
    return mirge_plan_scheme(p, mirge_seed_region(p, L), K, (uint64_t)npos);
}

// ---- isomiR typing (mirge_isotype.hpp, what k_isotype runs per read) on the CPU, one pair per call
#include "../../mirge3.0_amd/csrc/mirge_isotype.hpp"
extern "C" int hostsim_isotype(const char* a, int32_t la, const char* b, int32_t lb, const char* pre, int32_t lpre, int32_t start0,
                               int32_t* kind, int32_t* start, int32_t* end, char* variant, char* cigar) {
    if (la > MIRGE_ISO_MAXA || lb > MIRGE_ISO_MAXB) return -1;
    MirgeIsoRec r;
    r.kind = 0; r.vlen = r.clen = 0; r.start = r.end = 0;
    mirge_isotype(a, la, b, lb, pre, lpre, start0, r);
    *kind = r.kind; *start = r.start; *end = r.end;
    for (int k = 0; k < r.vlen; k++) variant[k] = r.text[k];
    variant[r.vlen] = 0;
    for (int k = 0; k < r.clen; k++) cigar[k] = r.text[r.vlen + k];
    cigar[r.clen] = 0;
    return 0;
}

// the register-resident form of the same record (mirge_isotype_fast): -2 when the pair is not one for that path
struct HostWs {
    uint32_t b[MIRGE_ISO_FAST_BLOCKS], q[MIRGE_ISO_FAST_BLOCKS];
    uint32_t& blk(int k) { return b[k]; }
    uint32_t& que(int k) { return q[k]; }
};
static int isotype_fast_rec(const char* a, int32_t la, const char* b, int32_t lb, const char* pre, int32_t lpre, int32_t start0, MirgeIsoRec& r) {
    mirge_iso::Seq sa, sb;
    if (la > MIRGE_ISO_MAXA || lb > MIRGE_ISO_MAXB) return -1;
    if (!mirge_iso::seq_of_ascii(a, la, sa) || !mirge_iso::seq_of_ascii(b, lb, sb)) return -2;
    HostWs ws;
    r.kind = 0; r.vlen = r.clen = 0; r.start = r.end = 0;
    return mirge_isotype_fast(sa, sb, pre, lpre, start0, ws, &r.start, &r.end, &r.kind, &r.vlen, &r.clen, r.text) ? 0 : -2;
}
extern "C" int hostsim_isotype_fast(const char* a, int32_t la, const char* b, int32_t lb, const char* pre, int32_t lpre, int32_t start0,
                                    int32_t* kind, int32_t* start, int32_t* end, char* variant, char* cigar) {
    MirgeIsoRec r;
    const int rc = isotype_fast_rec(a, la, b, lb, pre, lpre, start0, r);
    if (rc) return rc;
    *kind = r.kind; *start = r.start; *end = r.end;
    for (int k = 0; k < r.vlen; k++) variant[k] = r.text[k];
    variant[r.vlen] = 0;
    for (int k = 0; k < r.clen; k++) cigar[k] = r.text[r.vlen + k];
    cigar[r.clen] = 0;
    return 0;
}
// n random pairs through both forms: reads made from the canonical by shifts, templated and untemplated extensions, substitutions,
// N calls, short indels, homopolymer runs, plus unrelated reads; precursors with the canonical at the start, the end, absent, empty.
// Returns the number of pairs the fast form took; *bad = pairs on which a field differed (first one copied out).
extern "C" int64_t hostsim_isotype_fuzz(uint64_t seed, int64_t n, int64_t* bad, char* first_a, char* first_b, char* first_pre) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto below = [&](int m) { return (int)(rnd() % (uint64_t)m); };
    int64_t took = 0;
    *bad = 0;
    char a[80], b[160], pre[200], src[200];
    for (int64_t it = 0; it < n; it++) {
        const int la = 14 + below(20);
        for (int k = 0; k < la; k++) a[k] = "ACGT"[below(4)];
        if (it % 9 == 0 && la > 12) for (int k = 6; k < 10; k++) a[k] = a[5];
        if (it % 31 == 0) { const int per = 2 + below(2); for (int k = per; k < la; k++) a[k] = a[k - per]; }  // short tandem repeats
        if (it % 977 == 0) a[below(la)] = 'N';
        const int mode = (int)(it % 7);
        const int nl = mode != 1 ? below(12) : 0, nt = mode != 2 ? below(12) : 0;
        int ls = 0;
        for (int k = 0; k < nl; k++) src[ls++] = "ACGT"[below(4)];
        const int o = ls;
        for (int k = 0; k < la; k++) src[ls++] = a[k];
        for (int k = 0; k < nt; k++) src[ls++] = "ACGT"[below(4)];
        int lpre = 0;
        if (mode == 3) lpre = 0;
        else if (mode == 4) { for (int k = 0; k < nl; k++) pre[lpre++] = src[k]; for (int k = 0; k < nt; k++) pre[lpre++] = src[o + la + k]; }
        else { for (int k = 0; k < ls; k++) pre[lpre++] = src[k]; }
        int start0 = 1;
        if (lpre > 0) {  // precursor.find(canonical) + 1
            int f = -1;
            for (int p = 0; p + la <= lpre && f < 0; p++) { bool eq = true; for (int k = 0; k < la && eq; k++) eq = pre[p + k] == a[k]; if (eq) f = p; }
            start0 = f + 1;
        }
        const int d5 = below(7) - 3, d3 = below(9) - 4;
        int lb = 0;
        const bool t5 = below(10) < 6, t3 = below(10) < 6;
        for (int k = 0; k < -d5; k++) { const int p = o + d5 + k; b[lb++] = (t5 && p >= 0) ? src[p] : "ACGT"[below(4)]; }
        for (int k = (d5 > 0 ? d5 : 0); k < la + (d3 < 0 ? d3 : 0); k++) b[lb++] = a[k];
        for (int k = 0; k < d3; k++) { const int p = o + la + k; b[lb++] = (t3 && p < ls) ? src[p] : "ACGT"[below(4)]; }
        const int nmut = (int)("\0\0\1\1\2\3"[below(6)]);
        for (int m = 0; m < nmut && lb > 0; m++) b[below(lb)] = "ACGTN"[below(5)];
        if (it % 13 == 0 && lb > 8) { const int p = 1 + below(lb - 2); for (int k = p; k + 1 < lb; k++) b[k] = b[k + 1]; lb--; }          // a deleted base
        if (it % 17 == 0 && lb > 8 && lb < 60) { const int p = 1 + below(lb - 1); for (int k = lb; k > p; k--) b[k] = b[k - 1]; b[p] = "ACGT"[below(4)]; lb++; }  // an inserted one
        if (it % 50 == 0) { lb = 14 + below(17); for (int k = 0; k < lb; k++) b[k] = "ACGT"[below(4)]; }
        if (lb < 1) continue;
        MirgeIsoRec r0, r1;
        r0.kind = 0; r0.vlen = r0.clen = 0; r0.start = r0.end = 0;
        mirge_isotype(a, la, b, lb, pre, lpre, start0, r0);
        const int rc = isotype_fast_rec(a, la, b, lb, pre, lpre, start0, r1);
        if (rc == -2) continue;
        took++;
        bool same = rc == 0 && r0.kind == r1.kind && r0.vlen == r1.vlen && r0.clen == r1.clen;
        if (same && r0.kind) same = r0.start == r1.start && r0.end == r1.end;
        for (int k = 0; same && k < r0.vlen + r0.clen; k++) same = r0.text[k] == r1.text[k];
        if (!same) {
            if (!*bad) {
                std::memcpy(first_a, a, la); first_a[la] = 0;
                std::memcpy(first_b, b, lb); first_b[lb] = 0;
                std::memcpy(first_pre, pre, lpre); first_pre[lpre] = 0;
            }
            ++*bad;
        }
    }
    return took;
}
