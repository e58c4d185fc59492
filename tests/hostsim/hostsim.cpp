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
