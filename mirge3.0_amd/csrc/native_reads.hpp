// native_reads.hpp -- part of mirge_native.hip (one translation unit): read sets: pack, parse (FASTQ/FASTA text on the device), concat, unpack.
#pragma once
// ------------------------------------------------------------------------------------------
// reads
// ------------------------------------------------------------------------------------------
// Ten read groups: width class (<=31, <=64, <=128, <=255 nt, longer) x (no ambiguous call | has an N).  Reads with
// an N are rare (~0.1 %); keeping them apart lets the big groups run without an nmask array and lets
// the <=31-nt group collapse on a 64-bit key (sequence bits + length sentinel).  The fifth class holds what the templated
// kernels have no width for -- reads of 256 to 65535 nt, which the reference would hand to bowtie like any other (no upper
// bound: parse.py:102, digest.py:348,368): W words per read with W set by the longest read of the set, the length in 16
// bits, kernels of its own that take any length (kernels_long.hpp).
#define MIRGE_NGROUPS 10
#define MIRGE_NWIDTHS (MIRGE_NGROUPS / 2)
static_assert(MIRGE_NGROUPS == MIRGE_NCLS, "read groups = classes of k_seq_class");
static const int kGroupW[MIRGE_NGROUPS] = {1, 2, 4, 8, 0, 1, 2, 4, 8, 0};  // 0: the long class (ReadGroup::W at run time)
static inline bool is_long_group(int gi) { return kGroupW[gi] == 0; }
static inline size_t len_bytes(int gi) { return is_long_group(gi) ? 2 : 1; }  // ReadGroup::len holds uint16 for the long class
static inline int width_class(int64_t L) { return L <= 31 ? 0 : (L <= 64 ? 1 : (L <= 128 ? 2 : (L <= MIRGE_MAX_READ_LEN ? 3 : 4))); }
// run `call` with W = the width of read group gi (the four templated classes: callers take the long class aside first)
#define MIRGE_BY_WIDTH(gi, rc, call)                         \
    do {                                                     \
        switch (kGroupW[gi]) {                               \
            case 1: { constexpr int W = 1; rc = call; } break; \
            case 2: { constexpr int W = 2; rc = call; } break; \
            case 4: { constexpr int W = 4; rc = call; } break; \
            case 8: { constexpr int W = 8; rc = call; } break; \
            default: rc = fail(-1, "internal: the long read class has no templated kernel"); break; \
        }                                                    \
    } while (0)

struct ReadGroup {
    int W = 1;
    uint32_t n = 0;
    uint64_t* seq = nullptr;
    uint8_t* len = nullptr;    // [n]; the long class: uint16 [n] behind the same pointer (len_bytes)
    uint64_t* nmask = nullptr;
    uint32_t* orig = nullptr;    // handle-order index of each read (nullptr: base + j)
    uint32_t base = 0;
    uint32_t* counts = nullptr;  // [n][S]
    uint32_t* first = nullptr;   // [n] raw index of first appearance (collapse output)
};

struct mirge_reads {
    mirge_ctx* ctx = nullptr;
    int64_t n = 0;
    int64_t total_bases = 0;
    int32_t n_samples = 0;  // 0: no count matrix attached
    ReadGroup g[MIRGE_NGROUPS];
    int32_t len_hist[MIRGE_MAX_READ_LEN + 1];  // lengths present (host), for table preparation
    bool hist_valid = false;
    bool iupac_seen = false;  // some read held an IUPAC code other than N: packed (and printed) as N
    int32_t long_max = 0;     // longest read of the long class (0: none): its groups hold (long_max + 31) / 32 words per read
};

static LongView long_view_of(const ReadGroup& g) {
    LongView v; v.seq = g.seq; v.nmask = g.nmask; v.len = reinterpret_cast<const uint16_t*>(g.len); v.n = g.n; v.W = (uint32_t)g.W; return v;
}

static int largest_group(const mirge_reads* R) {
    int best = 0;
    for (int gi = 1; gi < MIRGE_NGROUPS; gi++) if (R->g[gi].n > R->g[best].n) best = gi;
    return best;
}

template <int W>
static GroupView<W> view_of(const ReadGroup& g) {
    GroupView<W> v; v.seq = g.seq; v.len = g.len; v.nmask = g.nmask; v.n = g.n; return v;
}

extern "C" void mirge_reads_destroy(mirge_reads* r) {
    if (!r) return;
    (void)join_pending_now(r->ctx);
    for (auto& g : r->g) {
        r->ctx->release(g.seq); r->ctx->release(g.len); r->ctx->release(g.nmask);
        r->ctx->release(g.orig); r->ctx->release(g.counts); r->ctx->release(g.first);
    }
    delete r;
}
// Several raw read sets (the samples of a run, each parsed from its own file) as one, in the order given: read j of
// part p gets handle index (reads of the parts before p) + j.  No count matrices; the parts stay valid.
static int reads_concat_impl(mirge_ctx* c, const mirge_reads* const* parts, int32_t n_parts, mirge_reads** out, bool collapsed_ok);
extern "C" int mirge_reads_concat(mirge_ctx* c, const mirge_reads* const* parts, int32_t n_parts, mirge_reads** out) {
    return reads_concat_impl(c, parts, n_parts, out, false);
}
// collapsed_ok: the parts may be collapse results (mirge_collapse_merge: their entries appended as reads; the counts stay behind)
static int reads_concat_impl(mirge_ctx* c, const mirge_reads* const* parts, int32_t n_parts, mirge_reads** out, bool collapsed_ok) {
    if (!c || !parts || n_parts < 1 || !out) return fail(-1, "mirge_reads_concat: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    int64_t total = 0;
    for (int p = 0; p < n_parts; p++) {
        if (!parts[p] || parts[p]->ctx != c) return fail(-1, "mirge_reads_concat: foreign or NULL read set");
        if (parts[p]->n_samples && !collapsed_ok) return fail(-1, "mirge_reads_concat: collapsed read sets cannot be appended");
        total += parts[p]->n;
    }
    if (total >= 0xFFFFFFF0ll) return fail(-5, "more than 2^32 reads in one set is not supported");
    auto R = std::make_unique<mirge_reads>();
    R->ctx = c; R->n = total; R->hist_valid = true;
    for (int p = 0; p < n_parts; p++) R->iupac_seen = R->iupac_seen || (parts[p] && parts[p]->iupac_seen);
    std::memset(R->len_hist, 0, sizeof(R->len_hist));
    for (int p = 0; p < n_parts; p++) {
        R->total_bases += parts[p]->total_bases;
        R->hist_valid = R->hist_valid && parts[p]->hist_valid;
        for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) R->len_hist[L] += parts[p]->len_hist[L];
        R->long_max = std::max(R->long_max, parts[p]->long_max);
    }
    int rc = 0;
    for (int gi = 0; gi < MIRGE_NGROUPS && rc == 0; gi++) {
        ReadGroup& g = R->g[gi];
        g.W = is_long_group(gi) ? (R->long_max + 31) / 32 : kGroupW[gi];
        uint64_t n = 0;
        bool mask = false;
        for (int p = 0; p < n_parts; p++) { n += parts[p]->g[gi].n; mask = mask || parts[p]->g[gi].nmask; }
        g.n = (uint32_t)n;
        if (!g.n) continue;
        if ((rc = dalloc(c, &g.seq, (size_t)g.W * g.n))) break;
        if ((rc = dalloc(c, &g.len, (size_t)g.n * len_bytes(gi)))) break;
        if ((rc = dalloc(c, &g.orig, (size_t)g.n))) break;
        if (mask && (rc = dalloc(c, &g.nmask, (size_t)g.W * g.n))) break;
        uint32_t at = 0, before = 0;
        hipError_t e = hipSuccess;
        for (int p = 0; p < n_parts && e == hipSuccess; p++) {
            const ReadGroup& q = parts[p]->g[gi];
            if (q.n && is_long_group(gi)) {  // the parts may hold different numbers of words per read
                hipLaunchKernelGGL(k_long_copy, dim3(grid_for(c, q.n)), dim3(MIRGE_BLOCK), 0, c->stream, long_view_of(q), (uint32_t)g.W, g.n, at,
                                   g.seq, g.nmask, reinterpret_cast<uint16_t*>(g.len));
                hipLaunchKernelGGL(k_index_shift, dim3(grid_for(c, q.n)), dim3(MIRGE_BLOCK), 0, c->stream, (const uint32_t*)q.orig, q.base,
                                   q.n, before, g.orig + at);
                at += q.n;
            } else if (q.n) {
                for (int w = 0; w < g.W && e == hipSuccess; w++) {  // word-major arrays: one copy per word plane
                    e = hipMemcpyAsync(g.seq + (size_t)w * g.n + at, q.seq + (size_t)w * q.n, (size_t)q.n * 8, hipMemcpyDeviceToDevice, c->stream);
                    if (e == hipSuccess && g.nmask) {
                        if (q.nmask) e = hipMemcpyAsync(g.nmask + (size_t)w * g.n + at, q.nmask + (size_t)w * q.n, (size_t)q.n * 8, hipMemcpyDeviceToDevice, c->stream);
                        else e = hipMemsetAsync(g.nmask + (size_t)w * g.n + at, 0, (size_t)q.n * 8, c->stream);
                    }
                }
                if (e == hipSuccess) e = hipMemcpyAsync(g.len + at, q.len, (size_t)q.n, hipMemcpyDeviceToDevice, c->stream);
                hipLaunchKernelGGL(k_index_shift, dim3(grid_for(c, q.n)), dim3(MIRGE_BLOCK), 0, c->stream, (const uint32_t*)q.orig, q.base,
                                   q.n, before, g.orig + at);
                at += q.n;
            }
            before += (uint32_t)parts[p]->n;
        }
        if (e != hipSuccess) rc = fail(-2, std::string("mirge_reads_concat: ") + hipGetErrorString(e));
    }
    if (rc == 0) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(-2, std::string("mirge_reads_concat: ") + hipGetErrorString(e));
    }
    if (rc) { mirge_reads_destroy(R.release()); return rc; }
    *out = R.release();
    return 0;
}

extern "C" int64_t mirge_reads_count(const mirge_reads* r) { return r ? r->n : -1; }
extern "C" int64_t mirge_reads_total_bases(const mirge_reads* r) { return r ? r->total_bases : -1; }
extern "C" int32_t mirge_reads_n_samples(const mirge_reads* r) { return r ? r->n_samples : -1; }
extern "C" int32_t mirge_reads_group_counts(const mirge_reads* r, int64_t* out, int32_t cap) {
    if (!r) return -1;
    for (int gi = 0; gi < MIRGE_NGROUPS && out && gi < cap; gi++) out[gi] = r->g[gi].n;
    return MIRGE_NGROUPS;
}
extern "C" int32_t mirge_reads_iupac_seen(const mirge_reads* r) { return r ? (r->iupac_seen ? 1 : 0) : -1; }

template <int W>
static int launch_pack(mirge_ctx* c, const uint8_t* dascii, const int64_t* dstart, const int64_t* dend, const uint32_t* didx,
                        ReadGroup& g, uint32_t* dflags, const int64_t* s2start = nullptr, const int32_t* s2len = nullptr) {
    LaunchScope ls(c, "k_pack", g.n);
    hipLaunchKernelGGL(k_pack<W>, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream,
                       dascii, dstart, dend, didx, g.n, g.seq, g.len, g.nmask, dflags, s2start, s2len);
    return 0;
}

static int launch_pack_long(mirge_ctx* c, const uint8_t* dascii, const int64_t* dstart, const int64_t* dend, const uint32_t* didx,
                            ReadGroup& g, uint32_t* dflags, const int64_t* s2start = nullptr, const int32_t* s2len = nullptr) {
    LaunchScope ls(c, "k_pack_long", g.n);
    hipLaunchKernelGGL(k_pack_long, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, dascii, dstart, dend, didx, g.n, (uint32_t)g.W,
                       g.seq, reinterpret_cast<uint16_t*>(g.len), g.nmask, dflags, s2start, s2len);
    return 0;
}

struct ParseJob;
static int pack_from_host_offsets(mirge_ctx* c, const char* ascii, const int64_t* off, int64_t n, mirge_reads* R);

// Reads in host memory (one ASCII buffer + n + 1 offsets) -> packed reads.  The host only checks the offsets (monotone, lengths
// within the limit: a few ms on a few threads); the letters go to the device as they are and take the text parser's route from
// there -- classification, stable placement by read group and 2-bit packing are k_seq_class / k_seq_place / k_pack with
// [off[i], off[i + 1]) for the line bounds.  (Until round 4 the host classified every read on up to 32 threads and uploaded
// an index per group: 40 of the 69 ms of the PCIe-inclusive path.)
extern "C" int mirge_reads_pack(mirge_ctx* c, const char* ascii, const int64_t* off, int64_t n, mirge_reads** out) {
    if (!c || !out || !off || n < 0 || (n > 0 && !ascii)) return fail(-1, "mirge_reads_pack: bad argument");
    if (n >= 0xFFFFFFF0ll) return fail(-5, "more than 2^32 reads in one set is not supported");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    {
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const int T = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 1000000));
        std::vector<int64_t> bad((size_t)T, -1), badlen((size_t)T, 0);
        auto work = [&](int t) {
            const int64_t lo = n * t / T, hi = n * (t + 1) / T;
            for (int64_t i = lo; i < hi; i++) {
                const int64_t L = off[i + 1] - off[i];
                if ((L < 0 || L > MIRGE_LONG_MAX_LEN) && bad[(size_t)t] < 0) { bad[(size_t)t] = i; badlen[(size_t)t] = L; }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        for (int t = 0; t < T; t++)
            if (bad[(size_t)t] >= 0) {
                if (badlen[(size_t)t] < 0) return fail(-1, "mirge_reads_pack: offsets not monotone");
                return fail(-6, "read " + std::to_string(bad[(size_t)t]) + " is " + std::to_string(badlen[(size_t)t]) + " nt; the limit is " +
                                std::to_string(MIRGE_LONG_MAX_LEN));
            }
    }
    auto R = std::make_unique<mirge_reads>();
    R->ctx = c; R->n = 0;
    std::memset(R->len_hist, 0, sizeof(R->len_hist));
    R->hist_valid = true;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) R->g[gi].W = kGroupW[gi];
    if (n > 0) {
        const int rc = pack_from_host_offsets(c, ascii, off, n, R.get());
        if (rc) { mirge_reads_destroy(R.release()); return rc; }
    }
    *out = R.release();
    return 0;
}

// Sequence text -> packed reads, parsed on the device (k_nl_count / k_nl_mark / k_seq_class / k_seq_place, then
// k_pack straight from the text).  format: 1 = FASTQ (4-line records), 2 = FASTA (one sequence line per record),
// 3 = one sequence per line, 0 = by the first byte ('@', '>', else 3).  Reads shorter than min_len are dropped
// (digest.py:348,368); *n_records = records seen before the filter (digest.py:326 `count`).
// mirge_trim (C ABI) -> the kernel's options; the two quality modifiers exist only where the text has qualities
static int trim_options(const mirge_trim* t, int32_t format, TrimOpts& o) {
    std::memset(&o, 0, sizeof(o));
    o.nextseq = t->nextseq_cutoff; o.q_front = t->quality_front; o.q_back = t->quality_back; o.base = t->phred_base ? t->phred_base : 33;
    o.min_overlap = t->min_overlap; o.rate = t->error_rate; o.trim_n = t->trim_n ? 1 : 0;
    if (format != 1) { o.nextseq = -1; o.q_back = -1; }
    if (t->adapter_len < 0 || t->adapter_len > MIRGE_TRIM_MAX_ADAPTER || (t->adapter_len > 0 && !t->adapter))
        return fail(-1, "mirge_reads_parse_trim: the 3' adapter must be 1-" + std::to_string(MIRGE_TRIM_MAX_ADAPTER) + " nt");
    o.alen = t->adapter_len;
    o.front = t->adapter_front ? 1 : 0;
    for (int i = 0; i < o.alen; i++) {
        const char ch = (char)(t->adapter[i] & 0xDF);
        if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N')
            return fail(-1, "mirge_reads_parse_trim: adapter characters other than A/C/G/T/N are not supported");
        if (ch == 'N' && o.front && !t->no_adapter_wildcards) return fail(-1, "mirge_reads_parse_trim: N in a 5' adapter is not supported");
        o.adapter[i] = (uint8_t)ch; o.wild[i] = ch == 'N';
    }
    if (t->adapter2_len < 0 || t->adapter2_len > MIRGE_TRIM_MAX_ADAPTER || (t->adapter2_len > 0 && (!t->adapter2 || !o.alen)))
        return fail(-1, "mirge_reads_parse_trim: the second adapter must be 1-" + std::to_string(MIRGE_TRIM_MAX_ADAPTER) + " nt and follow a first");
    o.alen2 = t->adapter2_len;
    o.front2 = t->adapter2_front ? 1 : 0;
    for (int i = 0; i < o.alen2; i++) {
        const char ch = (char)(t->adapter2[i] & 0xDF);
        if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N')
            return fail(-1, "mirge_reads_parse_trim: adapter characters other than A/C/G/T/N are not supported");
        if (ch == 'N' && o.front2 && !t->no_adapter_wildcards) return fail(-1, "mirge_reads_parse_trim: N in a 5' adapter is not supported");
        o.adapter2[i] = (uint8_t)ch; o.wild2[i] = ch == 'N';
    }
    o.times = t->times > 1 ? t->times : 1;
    o.no_indels = t->no_indels ? 1 : 0;
    o.read_wild = t->match_read_wildcards ? 1 : 0;
    o.action_none = t->action_none ? 1 : 0;
    o.anch = t->adapter_anchored ? 1 : 0;
    o.anch2 = t->adapter2_anchored ? 1 : 0;
    o.linked = t->linked ? 1 : 0;
    if (o.linked) {
        if (!o.alen || !o.alen2 || !o.front || o.front2)
            return fail(-1, "mirge_reads_parse_trim: a linked adapter is a 5' part (adapter, adapter_front = 1) and a 3' part (adapter2, adapter2_front = 0)");
        o.req1 = (t->linked_required & 1) ? 1 : 0;
        o.req2 = (t->linked_required & 2) ? 1 : 0;
    }
    if (t->no_adapter_wildcards)  // -N: an N in the adapter is a letter like any other (it matches an N of the read only)
        for (int i = 0; i < MIRGE_TRIM_MAX_ADAPTER; i++) { o.wild[i] = 0; o.wild2[i] = 0; }
    if (o.times > 16) return fail(-1, "mirge_reads_parse_trim: -n above 16 is not supported");
    if (o.alen && (!(o.rate >= 0.0) || o.rate > 1.0 || o.min_overlap < 1)) return fail(-1, "mirge_reads_parse_trim: error rate / overlap out of range");
    if (t->n_cut < 0 || t->n_cut > 2) return fail(-1, "mirge_reads_parse_trim: at most two unconditional cuts");
    o.n_cut = t->n_cut; o.cut[0] = t->cut[0]; o.cut[1] = t->cut[1];
    o.n_mods = (o.nextseq >= 0) + (o.q_back >= 0) + (o.alen > 0) + o.trim_n;
    for (int k = 0; k < o.n_cut; k++) o.n_mods += o.cut[k] != 0;
    o.stages_out = (t->count_per_modifier && o.n_mods > 1) ? o.n_mods : 1;
    if (o.n_mods > MIRGE_TRIM_MAX_MODS) return fail(-1, "mirge_reads_parse_trim: too many modifiers");
    return 0;
}

// k_trim<M, true, false>: one kernel per 3' adapter length without N (the unrolled DP rows carry no run-time test);
// k_trim<64, false, *>: any adapter of up to 64 bases -- a 3' adapter with N, or a 5' adapter
template <int M>
static int launch_trim_exact(mirge_ctx* c, const TrimOpts& o, const uint8_t* dtext, const int64_t* lstart, const int64_t* lend,
                             const int64_t* qstart, const int64_t* qend, uint32_t n_raw, int64_t* dstart, int64_t* dend, uint32_t* dflags) {
    if constexpr (M > MIRGE_TRIM_MAX_ADAPTER) {
        return fail(-1, "mirge_reads_parse_trim: adapter length");
    } else {
        if (o.alen == M) {
            hipLaunchKernelGGL((k_trim<M, true, false>), dim3(grid_for(c, n_raw)), dim3(MIRGE_BLOCK), 0, c->stream, dtext, lstart, lend, qstart, qend, n_raw, o, dstart, dend, dflags);
            return 0;
        }
        return launch_trim_exact<M + 1>(c, o, dtext, lstart, lend, qstart, qend, n_raw, dstart, dend, dflags);
    }
}
// (the general instances come in two heights: the DP column of 64 + 1 entries does not fit the registers beside the rest of the
//  kernel -- 1 101 spilled registers, 8 x the exact instance's time -- while adapters of up to 32 bases, nearly all of them, fit)
template <bool FRONT>
static void launch_trim_general(mirge_ctx* c, const TrimOpts& o, const uint8_t* dtext, const int64_t* lstart, const int64_t* lend,
                                const int64_t* qstart, const int64_t* qend, uint32_t n_raw, int64_t* dstart, int64_t* dend, uint32_t* dflags) {
    static const bool tall_only = std::getenv("MIRGE_TRIM_TALL") && std::atoi(std::getenv("MIRGE_TRIM_TALL")) != 0;  // (A/B, tests)
    if (!tall_only && o.alen <= 32 && o.alen2 <= 32)
        hipLaunchKernelGGL((k_trim<32, false, FRONT>), dim3(grid_for(c, n_raw)), dim3(MIRGE_BLOCK), 0, c->stream, dtext, lstart, lend, qstart, qend, n_raw, o, dstart, dend, dflags);
    else
        hipLaunchKernelGGL((k_trim<MIRGE_TRIM_MAX_ADAPTER, false, FRONT>), dim3(grid_for(c, n_raw)), dim3(MIRGE_BLOCK), 0, c->stream, dtext, lstart, lend, qstart, qend, n_raw, o, dstart, dend, dflags);
}
static int launch_trim(mirge_ctx* c, const TrimOpts& o, const uint8_t* dtext, const int64_t* lstart, const int64_t* lend,
                       const int64_t* qstart, const int64_t* qend, uint32_t n_raw, int64_t* dstart, int64_t* dend, uint32_t* dflags) {
    bool wild = false;
    for (int i = 0; i < o.alen; i++) wild = wild || o.wild[i];
    if (o.alen2 > 0 || o.times > 1 || o.no_indels || o.read_wild || o.action_none || o.anch || o.linked) {  // the general 3' instance carries these branches
        launch_trim_general<false>(c, o, dtext, lstart, lend, qstart, qend, n_raw, dstart, dend, dflags);
        return 0;
    }
    if (o.front) {  // a 5' adapter is the rare case: the general kernel
        launch_trim_general<true>(c, o, dtext, lstart, lend, qstart, qend, n_raw, dstart, dend, dflags);
        return 0;
    }
    if (o.alen >= 1 && !wild) return launch_trim_exact<1>(c, o, dtext, lstart, lend, qstart, qend, n_raw, dstart, dend, dflags);
    launch_trim_general<false>(c, o, dtext, lstart, lend, qstart, qend, n_raw, dstart, dend, dflags);
    return 0;
}

// ------------------------------------------------------------------------------------------
// text -> records -> packed reads, in two steps that share a ParseJob:
//   parse_lines : the text in HBM, its lines found, the modifiers applied (k_trim): [dstart, dend) of every virtual
//                 record (a record after every modifier, or after the last), for --qiagenumi the UMI stretch behind
//                 the adapter as a second segment
//   parse_pack  : length tests + UMI slice (k_seq_class), stable placement by read group (k_seq_place), 2-bit packing
//                 straight from the text (k_pack) -> a mirge_reads
// mirge_reads_parse[_trim] = one of each; mirge_reads_parse_umi with -udd runs parse_pack twice (the tagged reads, then
// the inserts of the distinct ones) around a collapse.
// ------------------------------------------------------------------------------------------
struct ParseJob {
    mirge_ctx* c = nullptr;
    int format = 0;
    uint64_t n = 0;      // bytes of dtext, with the final newline
    uint32_t n_raw = 0;  // records of the text
    uint32_t n_seq = 0;  // virtual records
    uint8_t* dtext = nullptr;
    uint32_t *tile_cnt = nullptr, *tile_off = nullptr, *lflags = nullptr;
    void* tmp = nullptr;
    size_t tmp_cap = 0;
    int64_t *dstart = nullptr, *dend = nullptr, *lstart = nullptr, *lend = nullptr, *qstart = nullptr, *qend = nullptr;
    int64_t* s2start = nullptr;
    int32_t* s2len = nullptr;
    bool line_flags_checked = false;
    ~ParseJob() {
        if (!c) return;
        (void)hipStreamSynchronize(c->stream);
        c->release(dtext); c->release(tile_cnt); c->release(tile_off); c->release(lflags); c->release(tmp);
        c->release(dstart); c->release(dend); c->release(lstart); c->release(lend); c->release(qstart); c->release(qend);
        c->release(s2start); c->release(s2len);
    }
};
#define MIRGE_LFLAG_WORDS 8  // [3] record structure broken, [5] quality line length != sequence line length

// spare_blank: empty lines that stood behind text[nbytes - 1] in the caller's buffer (the entry point strips the blank end of a
// file): as many of them as complete the last record are lines after all -- a record whose sequence (and quality) is empty, which a
// file trimmed without a minimum length ends with now and then.
static int parse_lines(ParseJob& J, const char* text, int64_t nbytes, const TrimOpts* topt, const mirge_umi* umi, int spare_blank = 0) {
    mirge_ctx* c = J.c;
    const int format = J.format;
    const bool trimming = topt && topt->n_mods > 0;
    const int period = format == 1 ? 4 : (format == 2 ? 2 : 1), sphase = format == 3 ? 0 : 1;
    // the text, with a final newline
    uint64_t n = (uint64_t)nbytes + 1;
    J.n = n;
    // lines are counted in 32 bits: below 8 GiB a text would need lines of less than two bytes to wrap the counter
    if (n >= (1ull << 33)) return fail(-5, "mirge_reads_parse: a text of 8 GiB or more must be passed in parts (mirge_reads_concat)");
    uint32_t ntile = (uint32_t)((n + MIRGE_PARSE_TILE - 1) / MIRGE_PARSE_TILE);
    CHECK(dalloc(c, &J.dtext, (size_t)n + 16));
    CHECK(dalloc(c, &J.tile_cnt, (size_t)ntile + 2));  // (+ 1: the few line ends the spare blank lines may add can open a tile)
    CHECK(dalloc(c, &J.tile_off, (size_t)ntile + 2));
    CHECK(dalloc(c, &J.lflags, (size_t)MIRGE_LFLAG_WORDS));
    HIPOK(hipMemcpyAsync(J.dtext, text, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemsetAsync(J.dtext + nbytes, '\n', 1, c->stream));
    HIPOK(hipMemsetAsync(J.tile_cnt + ntile, 0, 4, c->stream));
    HIPOK(hipMemsetAsync(J.lflags, 0, MIRGE_LFLAG_WORDS * 4, c->stream));
    hipLaunchKernelGGL(k_nl_count, dim3(ntile), dim3(MIRGE_BLOCK), 0, c->stream, J.dtext, n, J.tile_cnt);
    size_t tmp_bytes = 0;
    HIPOK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, J.tile_cnt, J.tile_off, (int)(ntile + 1), c->stream));
    J.tmp_cap = std::max<size_t>(tmp_bytes, 1 << 16);
    CHECK(dalloc(c, (uint8_t**)&J.tmp, J.tmp_cap));
    HIPOK(hipcub::DeviceScan::ExclusiveSum(J.tmp, tmp_bytes, J.tile_cnt, J.tile_off, (int)(ntile + 1), c->stream));
    uint32_t n_lines = 0;
    HIPOK(hipMemcpyAsync(&n_lines, J.tile_off + ntile, 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    if (n_lines % (uint32_t)period != 0 && (int)(period - n_lines % (uint32_t)period) <= spare_blank) {
        // the last record's empty lines: give them back (they fit the 16 bytes of slack) and count again
        const uint32_t need = (uint32_t)period - n_lines % (uint32_t)period;
        HIPOK(hipMemsetAsync(J.dtext + n, '\n', need, c->stream));
        n += need;
        J.n = n;
        ntile = (uint32_t)((n + MIRGE_PARSE_TILE - 1) / MIRGE_PARSE_TILE);
        HIPOK(hipMemsetAsync(J.tile_cnt + ntile, 0, 4, c->stream));
        hipLaunchKernelGGL(k_nl_count, dim3(ntile), dim3(MIRGE_BLOCK), 0, c->stream, J.dtext, n, J.tile_cnt);
        size_t tb2 = J.tmp_cap;
        HIPOK(hipcub::DeviceScan::ExclusiveSum(J.tmp, tb2, J.tile_cnt, J.tile_off, (int)(ntile + 1), c->stream));
        HIPOK(hipMemcpyAsync(&n_lines, J.tile_off + ntile, 4, hipMemcpyDeviceToHost, c->stream));
        HIPOK(hipStreamSynchronize(c->stream));
    }
    if (n_lines % (uint32_t)period != 0)
        return fail(-9, "mirge_reads_parse: " + std::to_string(n_lines) + " lines is not a whole number of " + std::to_string(period) +
                            "-line records (truncated file, blank line, or a FASTA with wrapped sequences)");
    // sequence lines: li in [0, n_lines) with li % period == sphase
    const uint64_t n_seq64 = n_lines > (uint32_t)sphase ? ((uint64_t)n_lines - sphase + period - 1) / period : 0;
    if (n_seq64 >= 0xFFFFFFF0ull) return fail(-5, "more than 2^32 reads in one set is not supported");
    J.n_raw = (uint32_t)n_seq64;
    // with trimming a record becomes `stages_out` virtual records (the read after every modifier, or after the last)
    const uint64_t n_virt64 = (uint64_t)J.n_raw * (trimming ? (uint64_t)topt->stages_out : 1ull);
    if (n_virt64 >= 0xFFFFFFF0ull) return fail(-5, "more than 2^32 reads in one set is not supported");
    J.n_seq = (uint32_t)n_virt64;
    if (!J.n_seq) return 0;
    CHECK(dalloc(c, &J.dstart, (size_t)J.n_seq));
    CHECK(dalloc(c, &J.dend, (size_t)J.n_seq));
    if (trimming) {
        CHECK(dalloc(c, &J.lstart, (size_t)J.n_raw));
        CHECK(dalloc(c, &J.lend, (size_t)J.n_raw));
        if (format == 1) { CHECK(dalloc(c, &J.qstart, (size_t)J.n_raw)); CHECK(dalloc(c, &J.qend, (size_t)J.n_raw)); }
    }
    hipLaunchKernelGGL(k_nl_mark, dim3(ntile), dim3(MIRGE_BLOCK), 0, c->stream, J.dtext, n, J.tile_off, period, sphase,
                       trimming ? J.lstart : J.dstart, trimming ? J.lend : J.dend, (uint64_t)J.n_raw, (int)format, J.lflags, J.qstart, J.qend);
    if (trimming) {
        LaunchScope ls(c, "k_trim", J.n_raw);
        CHECK(launch_trim(c, *topt, J.dtext, J.lstart, J.lend, J.qstart, J.qend, J.n_raw, J.dstart, J.dend, J.lflags));
    }
    if (umi && umi->qiagen) {  // (trimming with one output stage: checked by the caller)
        CHECK(dalloc(c, &J.s2start, (size_t)J.n_raw));
        CHECK(dalloc(c, &J.s2len, (size_t)J.n_raw));
        LaunchScope ls(c, "k_qiagen_umi", J.n_raw);
        hipLaunchKernelGGL(k_qiagen_umi, dim3(grid_for(c, J.n_raw)), dim3(MIRGE_BLOCK), 0, c->stream, J.dtext, J.lstart, J.lend, J.dstart,
                           J.dend, J.n_raw, topt->alen, umi->back, J.s2start, J.s2len);
    }
    return 0;
}

// records [start, end) (+ second segments) of J's text -> R (empty on entry).  rec_of_kept (optional, caller releases): the
// record of every read of R, by handle index.
static int parse_pack(ParseJob& J, int64_t* start, int64_t* end, int64_t* s2start, int32_t* s2len, uint32_t n_seq, const SliceOpts& so,
                      mirge_reads* R, uint32_t** rec_of_kept) {
    mirge_ctx* c = J.c;
    const int format = J.format;
    R->n = 0; R->total_bases = 0; R->hist_valid = true;
    std::memset(R->len_hist, 0, sizeof(R->len_hist));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) R->g[gi].W = kGroupW[gi];
    if (!n_seq) return 0;
    uint8_t* dcls = nullptr;
    uint32_t *blk = nullptr, *blk_off = nullptr, *keep = nullptr, *keep_off = nullptr, *dmeta = nullptr, *src_all = nullptr,
             *orig_all = nullptr, *rok = nullptr;
    const uint32_t nblk = std::max<uint32_t>(1, (n_seq + MIRGE_BLOCK - 1) / MIRGE_BLOCK);
    const size_t meta_words = 8 + MIRGE_MAX_READ_LEN + 1;  // [0..4] flags, [8..] length histogram
    int rc = 0;
    do {
        if ((rc = dalloc(c, &dcls, (size_t)n_seq))) break;
        if ((rc = dalloc(c, &blk, (size_t)MIRGE_NGROUPS * nblk + 1))) break;
        if ((rc = dalloc(c, &blk_off, (size_t)MIRGE_NGROUPS * nblk + 1))) break;
        if ((rc = dalloc(c, &keep, (size_t)nblk + 1))) break;
        if ((rc = dalloc(c, &keep_off, (size_t)nblk + 1))) break;
        if ((rc = dalloc(c, &dmeta, meta_words))) break;
        hipError_t e = hipMemsetAsync(dmeta, 0, meta_words * 4, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(blk + (size_t)MIRGE_NGROUPS * nblk, 0, 4, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(keep + nblk, 0, 4, c->stream);
        if (e != hipSuccess) { rc = fail(-2, hipGetErrorString(e)); break; }
        hipLaunchKernelGGL(k_seq_class, dim3(nblk), dim3(MIRGE_BLOCK), 0, c->stream, J.dtext, start, end, s2start, s2len, n_seq, so, dcls, blk,
                           keep, nblk, dmeta + 8, dmeta);
        size_t need = 0;
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, blk, blk_off, (int)(MIRGE_NGROUPS * nblk + 1), c->stream);
        if (e == hipSuccess && need > J.tmp_cap) { c->release(J.tmp); J.tmp = nullptr; J.tmp_cap = need; if ((rc = dalloc(c, (uint8_t**)&J.tmp, J.tmp_cap))) break; }
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(J.tmp, need, blk, blk_off, (int)(MIRGE_NGROUPS * nblk + 1), c->stream);
        size_t need2 = J.tmp_cap;
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(J.tmp, need2, keep, keep_off, (int)(nblk + 1), c->stream);
        // group bounds = blk_off at the first block of every class, and the total
        uint32_t bounds[MIRGE_NGROUPS + 1];
        for (int q = 0; q < MIRGE_NGROUPS && e == hipSuccess; q++)
            e = hipMemcpyAsync(&bounds[q], blk_off + (size_t)q * nblk, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&bounds[MIRGE_NGROUPS], blk_off + (size_t)MIRGE_NGROUPS * nblk, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->pinned, dmeta, meta_words * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->pinned + 512, J.lflags, MIRGE_LFLAG_WORDS * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_reads_parse: ") + hipGetErrorString(e)); break; }
        if (c->pinned[512 + 3]) { rc = fail(-9, std::string("mirge_reads_parse: a record does not start with '") + (format == 1 ? "@' / its third line with '+'" : ">'") +
                                                  " (truncated file, blank line, or a FASTA with wrapped sequences)"); break; }
        if (c->pinned[512 + 5]) { rc = fail(-9, "mirge_reads_parse: a record's quality line is not as long as its sequence line"); break; }
        if (c->pinned[1]) { rc = fail(-6, "a read is " + std::to_string(c->pinned[2]) + " nt; the limit is " + std::to_string(MIRGE_LONG_MAX_LEN)); break; }
        if (c->pinned[0]) { rc = fail(-7, "a read contains a character that is no nucleotide code (A/C/G/T/U/N or an IUPAC ambiguity code)"); break; }
        R->iupac_seen = c->pinned[4] != 0;
        const uint32_t kept = bounds[MIRGE_NGROUPS];
        R->n = kept;
        for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) {
            R->len_hist[L] = (int32_t)c->pinned[8 + L];
            R->total_bases += (int64_t)L * c->pinned[8 + L];
        }
        // the long class (reads beyond MIRGE_MAX_READ_LEN: outside the histogram): its longest read sets the words per read
        R->long_max = (int32_t)c->pinned[5];
        R->total_bases += (int64_t)((uint64_t)c->pinned[6] | ((uint64_t)c->pinned[7] << 32));
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
            if (is_long_group(gi)) R->g[gi].W = (R->long_max + 31) / 32;
        if (!kept) break;
        if ((rc = dalloc(c, &src_all, (size_t)kept))) break;
        if ((rc = dalloc(c, &orig_all, (size_t)kept))) break;
        if (rec_of_kept && (rc = dalloc(c, &rok, (size_t)kept))) break;
        hipLaunchKernelGGL(k_seq_place, dim3(nblk), dim3(MIRGE_BLOCK), 0, c->stream, dcls, n_seq, blk_off, keep_off, nblk, src_all,
                           orig_all, rok);
        uint32_t* dflags = dmeta;  // reused: k_pack's per-group (saw N, bad byte) pairs
        e = hipMemsetAsync(dflags, 0, 2 * MIRGE_NGROUPS * 4, c->stream);
        for (int gi = 0; gi < MIRGE_NGROUPS && rc == 0 && e == hipSuccess; gi++) {
            ReadGroup& g = R->g[gi];
            g.n = bounds[gi + 1] - bounds[gi];
            if (!g.n) continue;
            if ((rc = dalloc(c, &g.seq, (size_t)g.W * g.n))) break;
            if ((rc = dalloc(c, &g.nmask, (size_t)g.W * g.n))) break;
            if ((rc = dalloc(c, &g.len, (size_t)g.n * len_bytes(gi)))) break;
            if ((rc = dalloc(c, &g.orig, (size_t)g.n))) break;
            e = hipMemcpyAsync(g.orig, orig_all + bounds[gi], (size_t)g.n * 4, hipMemcpyDeviceToDevice, c->stream);
            const uint32_t* src = src_all + bounds[gi];
            int prc = 0;
            if (is_long_group(gi)) prc = launch_pack_long(c, J.dtext, start, end, src, g, dflags + 2 * gi, s2start, s2len);
            else MIRGE_BY_WIDTH(gi, prc, launch_pack<W>(c, J.dtext, start, end, src, g, dflags + 2 * gi, s2start, s2len));
            (void)prc;
        }
        if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (rc == 0 && e != hipSuccess) rc = fail(-2, std::string("mirge_reads_parse: ") + hipGetErrorString(e));
        if (rc == 0)
            for (int gi = 0; gi < MIRGE_NWIDTHS; gi++)  // the groups without an ambiguous call carry no mask
                if (R->g[gi].nmask) { c->release(R->g[gi].nmask); R->g[gi].nmask = nullptr; }
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    c->release(dcls); c->release(blk); c->release(blk_off); c->release(keep); c->release(keep_off); c->release(dmeta);
    c->release(src_all); c->release(orig_all);
    if (rc) { c->release(rok); return rc; }
    if (rec_of_kept) *rec_of_kept = rok;
    return 0;
}

static int pack_from_host_offsets(mirge_ctx* c, const char* ascii, const int64_t* off, int64_t n, mirge_reads* R) {
    const int64_t nbytes = off[n] - off[0];
    if ((uint64_t)nbytes + 1 >= (1ull << 33)) return fail(-5, "mirge_reads_pack: 8 GiB of letters or more must be passed in parts (mirge_reads_concat)");
    ParseJob J;
    J.c = c; J.format = 3;
    J.n = (uint64_t)nbytes + 1;
    J.n_raw = J.n_seq = (uint32_t)n;
    CHECK(dalloc(c, &J.dtext, (size_t)nbytes + 16));
    CHECK(dalloc(c, &J.dstart, (size_t)n + 1));  // the offsets, relative to the first letter
    CHECK(dalloc(c, &J.dend, (size_t)n));         // (k_seq_class writes the bounds back: two arrays, not one read at two offsets)
    CHECK(dalloc(c, &J.lflags, (size_t)MIRGE_LFLAG_WORDS));
    HIPOK(hipMemsetAsync(J.lflags, 0, MIRGE_LFLAG_WORDS * 4, c->stream));
    if (nbytes) HIPOK(hipMemcpyAsync(J.dtext, ascii + off[0], (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    std::vector<int64_t> rel;
    const int64_t* src = off;
    if (off[0] != 0) {
        rel.resize((size_t)n + 1);
        for (int64_t i = 0; i <= n; i++) rel[(size_t)i] = off[i] - off[0];
        src = rel.data();
    }
    HIPOK(hipMemcpyAsync(J.dstart, src, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(J.dend, J.dstart + 1, (size_t)n * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));  // (`rel` is read by the copy)
    const SliceOpts so{0, 0, 0, 0, 1};  // every read is kept, whatever its length, exactly as the offsets bound it
    return parse_pack(J, J.dstart, J.dend, nullptr, nullptr, (uint32_t)n, so, R, nullptr);
}

static void reads_clear_groups(mirge_reads* r) {
    for (auto& g : r->g) {
        r->ctx->release(g.seq); r->ctx->release(g.len); r->ctx->release(g.nmask);
        r->ctx->release(g.orig); r->ctx->release(g.counts); r->ctx->release(g.first);
        g = ReadGroup();
    }
}

extern "C" int mirge_collapse(mirge_ctx* c, const mirge_reads* raw, const int32_t* sample_ids, int32_t S, mirge_reads** uniq, int64_t* n_uniq);

extern "C" int mirge_reads_parse_umi(mirge_ctx* c, const char* text, int64_t nbytes, int32_t format, int32_t min_len,
                                     const mirge_trim* trim, const mirge_umi* umi, mirge_reads** out, int64_t* n_records,
                                     mirge_reads** tagged_out) {
    if (!c || !out || nbytes < 0 || (nbytes > 0 && !text) || format < 0 || format > 3)
        return fail(-1, "mirge_reads_parse: bad argument");
    if (tagged_out) *tagged_out = nullptr;
    if (umi && (umi->front < 0 || umi->back < 0 || umi->front > MIRGE_MAX_READ_LEN || umi->back > MIRGE_MAX_READ_LEN))
        return fail(-1, "mirge_reads_parse_umi: -umi f,b out of range");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    if (format == 0) format = nbytes == 0 ? 3 : (text[0] == '@' ? 1 : (text[0] == '>' ? 2 : 3));
    TrimOpts topt;
    std::memset(&topt, 0, sizeof(topt));
    if (trim) CHECK(trim_options(trim, format, topt));
    if (umi && umi->qiagen) {
        // the reference reads the UMI behind args.adapters[0] (digest.py:120-121,342-347) and counts such a read once,
        // after the last modifier (the length test and the dictionary update follow the loop in this branch, :349-353)
        if (!trim || topt.alen <= 0 || topt.front) return fail(-1, "mirge_reads_parse_umi: --qiagenumi needs the 3' adapter the UMI follows (-a)");
        topt.stages_out = 1;
    }
    auto R = std::make_unique<mirge_reads>();
    R->ctx = c; R->n = 0;
    std::memset(R->len_hist, 0, sizeof(R->len_hist));
    R->hist_valid = true;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) R->g[gi].W = kGroupW[gi];
    if (n_records) *n_records = 0;
    // blank lines at the end of the file are not records (dnaio stops at them too)
    int tail_line_ends = 0;
    while (nbytes > 0 && (text[nbytes - 1] == '\n' || text[nbytes - 1] == '\r' || text[nbytes - 1] == ' ' || text[nbytes - 1] == '\t')) {
        if (text[nbytes - 1] == '\n' && tail_line_ends < 8) tail_line_ends++;
        nbytes--;
    }
    if (nbytes == 0) { *out = R.release(); return 0; }
    ParseJob J;
    J.c = c; J.format = format;
    // (the first of the stripped line ends closed the last line with a letter in it; the others closed empty lines)
    int rc = parse_lines(J, text, nbytes, trim ? &topt : nullptr, umi, format == 3 ? 0 : std::max(0, tail_line_ends - 1));
    if (rc) { mirge_reads_destroy(R.release()); return rc; }
    if (n_records) *n_records = J.n_raw;
    const int32_t f = umi ? umi->front : 0, b = umi ? umi->back : 0;
    // the worker's own length test (digest.py:348 / :360 / :368): on the read as the modifiers left it, with -umi (no
    // --qiagenumi) on what UMIParser leaves of it, max(0, len - f - b)
    const int32_t pre = !umi ? 0 : (umi->qiagen ? min_len : (min_len > 0 ? min_len + f + b : 0));
    if (!umi || !umi->dedup) {
        // counts add up: slicing every counted read == slicing the keys of the merged dictionary (digest.py:164-181)
        const SliceOpts so{pre, min_len, f, b};
        rc = parse_pack(J, J.dstart, J.dend, J.s2start, J.s2len, J.n_seq, so, R.get(), nullptr);
        if (rc) { mirge_reads_destroy(R.release()); return rc; }
        *out = R.release();
        return 0;
    }
    // -udd (digest.py:183-205): the distinct UMI-tagged reads first ...
    uint32_t* rec_of_kept = nullptr;
    {
        const SliceOpts so{pre, 0, 0, 0};
        rc = parse_pack(J, J.dstart, J.dend, J.s2start, J.s2len, J.n_seq, so, R.get(), &rec_of_kept);
    }
    mirge_reads* T = nullptr;
    int64_t U = 0;
    if (rc == 0) rc = mirge_collapse(c, R.get(), nullptr, 1, &T, &U);
    std::unique_ptr<mirge_reads, void (*)(mirge_reads*)> Tg(T, mirge_reads_destroy);
    // ... then ONE read per distinct tagged read, its insert, in the order the tagged reads first appeared: the second
    // collapse (the caller's) counts molecules, and its first indices are ranks in the reference's dictionary order
    uint32_t *keys = nullptr, *keys2 = nullptr;
    int64_t *st2 = nullptr, *en2 = nullptr, *s2s2 = nullptr;
    int32_t* s2l2 = nullptr;
    void* stmp = nullptr;
    do {
        if (rc) break;
        reads_clear_groups(R.get());
        if (!U) break;
        if (U >= 0x7FFFFFFFll) { rc = fail(-5, "mirge_reads_parse_umi: 2^31 distinct tagged reads or more (hipCUB's sort takes an int count)"); break; }
        if ((rc = dalloc(c, &keys, (size_t)U))) break;
        if ((rc = dalloc(c, &keys2, (size_t)U))) break;
        hipError_t e = hipSuccess;
        for (int gi = 0; gi < MIRGE_NGROUPS && e == hipSuccess; gi++) {
            const ReadGroup& g = T->g[gi];
            if (g.n) e = hipMemcpyAsync(keys + g.base, g.first, (size_t)g.n * 4, hipMemcpyDeviceToDevice, c->stream);
        }
        size_t sb = 0;
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortKeys(nullptr, sb, keys, keys2, (int)U, 0, 32, c->stream);
        if (e == hipSuccess && (rc = dalloc(c, (uint8_t**)&stmp, std::max<size_t>(sb, 16)))) break;
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortKeys(stmp, sb, keys, keys2, (int)U, 0, 32, c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_reads_parse_umi: ") + hipGetErrorString(e)); break; }
        if ((rc = dalloc(c, &st2, (size_t)U))) break;
        if ((rc = dalloc(c, &en2, (size_t)U))) break;
        if (J.s2len) {
            if ((rc = dalloc(c, &s2s2, (size_t)U))) break;
            if ((rc = dalloc(c, &s2l2, (size_t)U))) break;
        }
        hipLaunchKernelGGL(k_gather_records, dim3(grid_for(c, (size_t)U)), dim3(MIRGE_BLOCK), 0, c->stream, keys2, (uint32_t)U, rec_of_kept,
                           J.dstart, J.dend, J.s2start, J.s2len, st2, en2, s2s2, s2l2);
        const SliceOpts so{0, min_len, f, b};
        rc = parse_pack(J, st2, en2, s2s2, s2l2, (uint32_t)U, so, R.get(), nullptr);
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    c->release(rec_of_kept); c->release(keys); c->release(keys2); c->release(stmp);
    c->release(st2); c->release(en2); c->release(s2s2); c->release(s2l2);
    if (rc) { mirge_reads_destroy(R.release()); return rc; }
    if (tagged_out) *tagged_out = Tg.release();
    *out = R.release();
    return 0;
}

extern "C" int mirge_reads_parse_trim(mirge_ctx* c, const char* text, int64_t nbytes, int32_t format, int32_t min_len,
                                      const mirge_trim* trim, mirge_reads** out, int64_t* n_records) {
    return mirge_reads_parse_umi(c, text, nbytes, format, min_len, trim, nullptr, out, n_records, nullptr);
}
extern "C" int mirge_reads_parse(mirge_ctx* c, const char* text, int64_t nbytes, int32_t format, int32_t min_len,
                                 mirge_reads** out, int64_t* n_records) {
    return mirge_reads_parse_umi(c, text, nbytes, format, min_len, nullptr, nullptr, out, n_records, nullptr);
}


template <int W>
static int launch_unpack(mirge_ctx* c, const ReadGroup& g, const int64_t* doff, uint8_t* dout) {
    hipLaunchKernelGGL(k_unpack<W>, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, view_of<W>(g), doff, g.base, g.orig, dout);
    return 0;
}

// `dpos` (device, [n], or nullptr): read with handle index i goes to row dpos[i] of the output instead of row i (the range
// split of a dictionary, native_csv.hpp; the handle must then be a collapse result -- no `orig` lists)
static int reads_unpack_impl(mirge_ctx* c, const mirge_reads* R, char* ascii_out, int64_t* off_out, const uint32_t* dpos);
extern "C" int mirge_reads_unpack(mirge_ctx* c, const mirge_reads* R, char* ascii_out, int64_t* off_out) {
    if (!c || !R || !off_out || (R->total_bases > 0 && !ascii_out)) return fail(-1, "mirge_reads_unpack: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    return reads_unpack_impl(c, R, ascii_out, off_out, nullptr);
}
static int reads_unpack_impl(mirge_ctx* c, const mirge_reads* R0, char* ascii_out, int64_t* off_out, const uint32_t* dpos) {
    mirge_reads Rv = *R0;  // (a view: with dpos every group's `orig` is its stretch of the position list)
    if (dpos)
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
            if (Rv.g[gi].n && Rv.g[gi].orig) return fail(-1, "reads_unpack_impl: a row order needs a collapse result");
            Rv.g[gi].orig = const_cast<uint32_t*>(dpos) + Rv.g[gi].base;
        }
    const mirge_reads* R = &Rv;
    const int64_t n = R->n;
    int32_t* dlen = nullptr;
    CHECK(dalloc(c, &dlen, (size_t)std::max<int64_t>(n, 1)));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = R->g[gi];
        if (!g.n) continue;
        LaunchScope ls(c, "k_scatter_len", g.n);
        if (is_long_group(gi))
            hipLaunchKernelGGL(k_scatter_len16, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream,
                               reinterpret_cast<const uint16_t*>(g.len), g.n, g.base, g.orig, dlen);
        else
            hipLaunchKernelGGL(k_scatter_len, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream,
                               g.len, g.n, g.base, g.orig, dlen);
    }
    std::vector<int32_t> hlen((size_t)std::max<int64_t>(n, 1));
    if (n) HIPOK(hipMemcpyAsync(hlen.data(), dlen, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    off_out[0] = 0;
    for (int64_t i = 0; i < n; i++) off_out[i + 1] = off_out[i] + hlen[(size_t)i];
    const int64_t total = off_out[n];
    int64_t* doff = nullptr; uint8_t* dout = nullptr;
    CHECK(dalloc(c, &doff, (size_t)n + 1));
    CHECK(dalloc(c, &dout, (size_t)std::max<int64_t>(total, 1)));
    HIPOK(hipMemcpyAsync(doff, off_out, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = R->g[gi];
        if (!g.n) continue;
        LaunchScope ls(c, "k_unpack", g.n);
        int urc = 0;
        if (is_long_group(gi))
            hipLaunchKernelGGL(k_unpack_long, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, long_view_of(g), doff, g.base, g.orig, dout);
        else MIRGE_BY_WIDTH(gi, urc, launch_unpack<W>(c, g, doff, dout));
        (void)urc;
    }
    if (total) HIPOK(hipMemcpyAsync(ascii_out, dout, (size_t)total, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    c->release(dlen); c->release(doff); c->release(dout);
    return 0;
}

extern "C" int mirge_reads_set_counts(mirge_ctx* c, mirge_reads* R, const uint32_t* counts, int32_t S) {
    if (!c || !R || !counts || S < 1) return fail(-1, "mirge_reads_set_counts: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    // counts are in handle order; each group wants its rows contiguous -> gather on the host
    // through the group's orig list (small: U x S)
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        ReadGroup& g = R->g[gi];
        if (!g.n) continue;
        std::vector<uint32_t> horig(g.n);
        if (g.orig) { HIPOK(hipMemcpyAsync(horig.data(), g.orig, (size_t)g.n * 4, hipMemcpyDeviceToHost, c->stream)); HIPOK(hipStreamSynchronize(c->stream)); }
        else for (uint32_t j = 0; j < g.n; j++) horig[j] = g.base + j;
        std::vector<uint32_t> rows((size_t)g.n * S);
        for (uint32_t j = 0; j < g.n; j++)
            std::memcpy(&rows[(size_t)j * S], &counts[(size_t)horig[j] * S], (size_t)S * 4);
        c->release(g.counts); g.counts = nullptr;
        CHECK(dalloc(c, &g.counts, (size_t)g.n * S));
        HIPOK(hipMemcpyAsync(g.counts, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPOK(hipStreamSynchronize(c->stream));
    }
    R->n_samples = S;
    return 0;
}
