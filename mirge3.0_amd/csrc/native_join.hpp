// native_join.hpp -- part of mirge_native.hip (one translation unit): count join, variant tally, timers and profile access.
#pragma once
// ------------------------------------------------------------------------------------------
// count join
// ------------------------------------------------------------------------------------------
extern "C" int mirge_count_join(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, int32_t exact_pass,
                                int32_t iso_pass, int64_t n_mirna, int64_t* class_sums, int64_t* exact, int64_t* iso) {
    HostClock hc("count_join");
    if (!c || !U || !res || !class_sums || !exact || !iso || n_mirna < 0) return fail(-1, "mirge_count_join: bad argument");
    if (U->n_samples < 1) return fail(-1, "read set has no count matrix (collapse it or mirge_reads_set_counts)");
    if (res->n != U->n) return fail(-1, "result and read set differ in size");
    // k_join adds into exact[ref] / iso[ref] with ref taken from the cascade: the tables must cover the libraries
    // those two passes ran against (a pass index outside the cascade simply selects nothing)
    for (int32_t p : {exact_pass, iso_pass})
        if (p >= 0 && p < res->n_pass && (int64_t)res->n_refs[p] > n_mirna)
            return fail(-1, "mirge_count_join: n_mirna (" + std::to_string(n_mirna) + ") is smaller than the " +
                                std::to_string(res->n_refs[p]) + " references of pass " + std::to_string(p));
    HIPOK(hipSetDevice(c->device));
    const int32_t S = U->n_samples, P = res->n_pass;
    const size_t n_cls = (size_t)P * S, n_tab = (size_t)std::max<int64_t>(n_mirna, 1) * S;
    const size_t words = n_cls + 2 * n_tab;
    if (c->join_dev_words < words) {
        if (c->join_dev) { HIPOK(hipStreamSynchronize(c->stream)); (void)hipFree(c->join_dev); }
        c->join_dev = nullptr; c->join_dev_words = 0; c->join_dev_clean = false;
        HIPOK(hipMalloc((void**)&c->join_dev, words * 8));
        c->join_dev_words = words;
    }
    unsigned long long* d = c->join_dev;
    if (!c->join_dev_clean) HIPOK(hipMemsetAsync(d, 0, c->join_dev_words * 8, c->stream));
    c->join_dev_clean = false;
    if (words * 8 > c->join_pinned_bytes) {
        if (c->join_pinned) { HIPOK(hipStreamSynchronize(c->stream)); (void)hipHostFree(c->join_pinned); }
        c->join_pinned = nullptr; c->join_pinned_bytes = 0;
        HIPOK(hipHostMalloc((void**)&c->join_pinned, words * 8 * 2, hipHostMallocDefault));
        c->join_pinned_bytes = words * 8 * 2;
    }
    unsigned long long* partial = nullptr;
    // Tables that fit LDS (one sample, up to ~3 000 miRNA references): 1024-thread workgroups keep them there while they walk
    // their reads, write one row each, and ONE column sum adds all rows to the ctx's tables -- no global atomic per read.
    // The bulk group's rows are computed before the cascade's side streams are joined (its annotation is complete in the
    // main stream's own order): k_join_rows of 4.2 M reads runs beside the tail of the small groups' cascades instead of
    // behind the wait for them.  Otherwise: everything in one launch with atomics (k_join_multi).
    static const bool rows_off = std::getenv("MIRGE_JOIN_ROWS") && std::atoi(std::getenv("MIRGE_JOIN_ROWS")) == 0;  // A/B
    // `join_pending` stays set until stream_join has actually been issued: an early return below (argument check, out of
    // memory) then leaves the side streams to the next entry point's join_pending_now instead of unjoined for good
    const bool lazy = c->join_pending;
    auto join_side_streams = [&]() -> int {
        if (!lazy || !c->join_pending) return 0;
        c->join_pending = false;
        return stream_join(c);
    };
    struct PoolBlock {  // `partial` goes back to the pool on every way out
        mirge_ctx* c; unsigned long long*& p;
        ~PoolBlock() { c->release(p); p = nullptr; }
    } partial_guard{c, partial};
    const int big = largest_group(U);
    auto group_list = [&](bool bulk, JoinGroups& gs) -> uint64_t {
        std::memset(&gs, 0, sizeof(gs));
        uint64_t total = 0;
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
            const ResGroup& g = res->g[gi];
            if (!g.n || (gi == big) != bulk) continue;
            gs.pass[gs.n_groups] = g.pass; gs.ref[gs.n_groups] = g.ref; gs.counts[gs.n_groups] = U->g[gi].counts;
            gs.start[gs.n_groups++] = (uint32_t)total;
            total += g.n;
        }
        gs.start[gs.n_groups] = (uint32_t)std::min<uint64_t>(total, 0xFFFFFFFFull);
        return total;
    };
    JoinGroups gb, gsm;
    const uint64_t n_bulk = group_list(true, gb), n_small = group_list(false, gsm);
    if (n_small >= 0xFFFFFFF0ull) return fail(-5, "mirge_count_join: more than 2^32 reads outside the bulk group");
    const bool by_rows = !rows_off && n_bulk + n_small >= 65536 && words <= MIRGE_JOIN_ROWS_CELLS && n_mirna * (int64_t)S < 0x7FFFFFFF;
    if (by_rows) {
        const uint32_t rows_b = n_bulk ? (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu, (n_bulk + 4095) / 4096) : 0u;
        const uint32_t rows_s = n_small ? (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu, (n_small + 4095) / 4096) : 0u;
        CHECK(dalloc(c, &partial, (size_t)(rows_b + rows_s) * words));
        if (rows_b) {
            LaunchScope ls(c, "k_join", (double)n_bulk);
            hipLaunchKernelGGL(k_join_rows, dim3(rows_b), dim3(MIRGE_JOIN_ROWS_THREADS), words * 8, c->stream, gb, S, P, exact_pass, iso_pass,
                               (uint32_t)n_tab, partial);
        }
        static const bool rows_on_aux = std::getenv("MIRGE_JOIN_SMALL_ON_AUX") && std::atoi(std::getenv("MIRGE_JOIN_SMALL_ON_AUX")) == 1;
        if (rows_s && rows_on_aux && lazy && c->join_pending) {
            // MIRGE_JOIN_SMALL_ON_AUX=1 (round 5 experiment, OFF by default): the small groups' rows on `aux`, right behind their
            // cascades -- `aux` collects the extra streams, runs the rows kernel, and the main stream waits for `aux` once --
            // instead of on the main stream behind the wait.  Meant to hide the kernel and a queue-to-queue hop (~24 us) behind
            // k_resolve; measured WORSE, 1.214 vs 1.196 ms per C3 step (profiles/r05_ab_join_on_aux.txt): the small groups'
            // cascades end ~50 us after the bulk kernel (they run on what its workgroups free), so the rows kernel on `aux`
            // starts no earlier, and the main stream's wait then includes it.
            c->join_pending = false;
            hipError_t e = hipSuccess;
            if (c->xaux_used) {
                for (int k = 0; k < MIRGE_N_XAUX && e == hipSuccess; k++) {
                    e = hipEventRecord(c->ev_xjoin[k], c->xaux[k]);
                    if (e == hipSuccess) e = hipStreamWaitEvent(c->aux, c->ev_xjoin[k], 0);
                }
                c->xaux_used = false;
            }
            if (e != hipSuccess) return fail(-2, std::string("mirge_count_join: ") + hipGetErrorString(e));
            c->cur = c->aux;
            {
                LaunchScope ls(c, "k_join", (double)n_small);
                hipLaunchKernelGGL(k_join_rows, dim3(rows_s), dim3(MIRGE_JOIN_ROWS_THREADS), words * 8, c->aux, gsm, S, P, exact_pass, iso_pass,
                                   (uint32_t)n_tab, partial + (size_t)rows_b * words);
            }
            c->cur = c->stream;
            e = hipEventRecord(c->ev_join, c->aux);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
            if (e != hipSuccess) return fail(-2, std::string("mirge_count_join: ") + hipGetErrorString(e));
            c->flush_deferred();  // reused only by work queued on the main stream after the wait
        } else {
            CHECK(join_side_streams());
            if (rows_s) {
                LaunchScope ls(c, "k_join", (double)n_small);
                hipLaunchKernelGGL(k_join_rows, dim3(rows_s), dim3(MIRGE_JOIN_ROWS_THREADS), words * 8, c->stream, gsm, S, P, exact_pass, iso_pass,
                                   (uint32_t)n_tab, partial + (size_t)rows_b * words);
            }
        }
        LaunchScope ls(c, "k_join_reduce", (double)words);
        // ... written by the kernel into the page-locked tables the host reads (no copy behind it, nothing to clear)
        hipLaunchKernelGGL(k_join_reduce, dim3((unsigned)((words + 63) / 64)), dim3(1024), 0, c->stream, partial, rows_b + rows_s, (uint32_t)words, d,
                           c->join_pinned);
    } else {
        CHECK(join_side_streams());
        for (JoinGroups* gs : {&gb, &gsm}) {
            const uint64_t total = gs == &gb ? n_bulk : n_small;
            if (!total) continue;
            LaunchScope ls(c, "k_join", (double)total);
            hipLaunchKernelGGL(k_join_multi, dim3(grid_for(c, (size_t)total)), dim3(MIRGE_BLOCK), 0, c->stream, *gs, S, P, exact_pass, iso_pass,
                               d, d + n_cls, d + n_cls + n_tab);
        }
    }
    bool cleared = true;  // the column sums went straight to the page-locked tables; the device tables are still zero
    if (!by_rows) {
        // one device-to-host copy through pinned memory for all three tables (they are contiguous)
        HIPOK(hipMemcpyAsync(c->join_pinned, d, words * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPOK(hipEventRecord(c->ev_meta, c->stream));
    // cleared for the next call now, behind the copy: the host waits for the copy only
    if (!by_rows) cleared = hipMemsetAsync(d, 0, c->join_dev_words * 8, c->stream) == hipSuccess;
    hc.lap("enqueue");
    HIPOK(hipEventSynchronize(c->ev_meta));
    hc.lap("wait for the tables");
    c->join_dev_clean = cleared;
    std::memcpy(class_sums, c->join_pinned, n_cls * 8);
    if (n_mirna) {
        std::memcpy(exact, c->join_pinned + n_cls, (size_t)n_mirna * S * 8);
        std::memcpy(iso, c->join_pinned + n_cls + n_tab, (size_t)n_mirna * S * 8);
    }
    c->drain();
    hc.lap("copy out + drain");
    return 0;
}

extern "C" int mirge_count_join_host(mirge_ctx* c, const int8_t* pass, const int32_t* ref, const uint32_t* counts,
                                     int64_t n, int32_t S, int32_t P, int32_t exact_pass, int32_t iso_pass,
                                     int64_t n_mirna, int64_t* class_sums, int64_t* exact, int64_t* iso) {
    if (!c || !class_sums || !exact || !iso || n < 0 || S < 1 || P < 1 || P > MIRGE_MAX_PASSES || n_mirna < 0 ||
        (n > 0 && (!pass || !ref || !counts)))
        return fail(-1, "mirge_count_join_host: bad argument");
    if (n >= 0xFFFFFFF0ll) return fail(-5, "more than 2^32 rows");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    for (int64_t i = 0; i < n; i++) {
        if (pass[i] >= P) return fail(-1, "pass index out of range");
        if ((pass[i] == exact_pass || pass[i] == iso_pass) && (ref[i] < 0 || ref[i] >= n_mirna))
            return fail(-1, "miRNA reference index out of range");
    }
    const size_t n_cls = (size_t)P * S, n_tab = (size_t)std::max<int64_t>(n_mirna, 1) * S;
    unsigned long long* d = nullptr; int8_t* dp = nullptr; int32_t* dr = nullptr; uint32_t* dc = nullptr;
    CHECK(dalloc(c, &d, n_cls + 2 * n_tab));
    CHECK(dalloc(c, &dp, (size_t)std::max<int64_t>(n, 1)));
    CHECK(dalloc(c, &dr, (size_t)std::max<int64_t>(n, 1)));
    CHECK(dalloc(c, &dc, (size_t)std::max<int64_t>(n, 1) * S));
    HIPOK(hipMemsetAsync(d, 0, (n_cls + 2 * n_tab) * 8, c->stream));
    if (n) {
        HIPOK(hipMemcpyAsync(dp, pass, (size_t)n, hipMemcpyHostToDevice, c->stream));
        HIPOK(hipMemcpyAsync(dr, ref, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        HIPOK(hipMemcpyAsync(dc, counts, (size_t)n * S * 4, hipMemcpyHostToDevice, c->stream));
        LaunchScope ls(c, "k_join", (double)n);
        hipLaunchKernelGGL(k_join, dim3(grid_for(c, (size_t)n)), dim3(MIRGE_BLOCK), 0, c->stream, dp, dr, dc, (uint32_t)n,
                           S, P, exact_pass, iso_pass, d, d + n_cls, d + n_cls + n_tab);
    }
    // one device-to-host copy through pinned memory for all three tables (they are contiguous)
    const size_t words = n_cls + 2 * n_tab;
    if (words * 8 > c->join_pinned_bytes) {
        if (c->join_pinned) (void)hipHostFree(c->join_pinned);
        c->join_pinned = nullptr; c->join_pinned_bytes = 0;
        HIPOK(hipHostMalloc((void**)&c->join_pinned, words * 8 * 2, hipHostMallocDefault));
        c->join_pinned_bytes = words * 8 * 2;
    }
    HIPOK(hipMemcpyAsync(c->join_pinned, d, words * 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    std::memcpy(class_sums, c->join_pinned, n_cls * 8);
    if (n_mirna) {
        std::memcpy(exact, c->join_pinned + n_cls, (size_t)n_mirna * S * 8);
        std::memcpy(iso, c->join_pinned + n_cls + n_tab, (size_t)n_mirna * S * 8);
    }
    c->drain();
    c->release(d); c->release(dp); c->release(dr); c->release(dc);
    return 0;
}

// per-position variant tally / A-to-I counting core: see k_tally.  Everything per (family, sample) comes back in
// ONE block: [n_seqs | seq_true | count_true | canon | kept_exact] (n_fam * S each) then census
// [n_fam][32][16][3][S]; diag / state per read in handle order.
extern "C" int mirge_variant_tally(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, int32_t exact_pass,
                                   int32_t iso_pass, const int32_t* fam_of_ref, int64_t n_mirna, const char* target_ascii,
                                   const int64_t* target_off, int64_t n_fam, const uint8_t* retained, const double* freq,
                                   int64_t* fam_tables, int64_t* census, int8_t* diag_out, int8_t* state_out) {
    static_assert(MIRGE_TALLY_POSITIONS == MIRGE_TALLY_MAXPOS, "tally positions");
    if (!c || !U || !res || !fam_of_ref || n_mirna < 0 || n_fam < 0 || !target_off || (n_fam > 0 && !target_ascii) || !freq ||
        !fam_tables || !census)
        return fail(-1, "mirge_variant_tally: bad argument");
    if (U->n_samples < 1) return fail(-1, "read set has no count matrix");
    if (res->n != U->n) return fail(-1, "result and read set differ in size");
    for (int32_t p : {exact_pass, iso_pass})
        if (p >= 0 && p < res->n_pass && (int64_t)res->n_refs[p] > n_mirna)
            return fail(-1, "mirge_variant_tally: fam_of_ref is shorter than the miRNA library of pass " + std::to_string(p));
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    const int32_t S = U->n_samples;
    // family targets, 2-bit packed on the host (a few thousand sequences of <= 32 nt)
    std::vector<uint64_t> tb((size_t)std::max<int64_t>(n_fam, 1), 0ull);
    std::vector<uint8_t> tl((size_t)std::max<int64_t>(n_fam, 1), 0);
    for (int64_t f = 0; f < n_fam; f++) {
        const int64_t L = target_off[f + 1] - target_off[f];
        if (L < 0 || L > MIRGE_TALLY_MAXPOS) return fail(-6, "family " + std::to_string(f) + ": canonical sequence of " + std::to_string(L) + " nt; the limit is 32");
        for (int64_t k = 0; k < L; k++) {
            const int code = mirge_base_code(target_ascii[target_off[f] + k]);
            if (code < 0) return fail(-7, "family " + std::to_string(f) + ": canonical sequence holds a character other than A/C/G/T/U");
            tb[(size_t)f] |= (uint64_t)code << (2 * k);
        }
        tl[(size_t)f] = (uint8_t)L;
    }
    for (int64_t r = 0; r < n_mirna; r++)
        if (fam_of_ref[r] >= n_fam) return fail(-1, "fam_of_ref holds a family index out of range");
    const size_t n_fs = (size_t)std::max<int64_t>(n_fam, 1) * S, n_cen = n_fs * MIRGE_TALLY_MAXPOS * 16 * 3;
    const size_t n_words = 5 * n_fs + n_cen;
    unsigned long long* d = nullptr;
    uint64_t* dtb = nullptr; uint8_t* dtl = nullptr; int32_t* dfam = nullptr; uint8_t* dret = nullptr; double* dfreq = nullptr;
    int8_t *ddiag = nullptr, *dstate = nullptr;
    const size_t n = (size_t)std::max<int64_t>(U->n, 1);
    CHECK(dalloc(c, &d, n_words));
    CHECK(dalloc(c, &dtb, tb.size()));
    CHECK(dalloc(c, &dtl, tl.size()));
    CHECK(dalloc(c, &dfam, (size_t)std::max<int64_t>(n_mirna, 1)));
    CHECK(dalloc(c, &dfreq, (size_t)S));
    CHECK(dalloc(c, &ddiag, n));
    CHECK(dalloc(c, &dstate, n));
    if (retained) CHECK(dalloc(c, &dret, n));
    HIPOK(hipMemsetAsync(d, 0, n_words * 8, c->stream));
    HIPOK(hipMemcpyAsync(dtb, tb.data(), tb.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(dtl, tl.data(), tl.size(), hipMemcpyHostToDevice, c->stream));
    if (n_mirna) HIPOK(hipMemcpyAsync(dfam, fam_of_ref, (size_t)n_mirna * 4, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(dfreq, freq, (size_t)S * 8, hipMemcpyHostToDevice, c->stream));
    if (retained && U->n) HIPOK(hipMemcpyAsync(dret, retained, (size_t)U->n, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemsetAsync(dstate, 0xFF, n, c->stream));  // reads of the wide groups are never members
    HIPOK(hipMemsetAsync(ddiag, 0, n, c->stream));
    TallyOut o;
    o.n_seqs = d; o.seq_true = d + n_fs; o.count_true = d + 2 * n_fs; o.canon = d + 3 * n_fs; o.kept_exact = d + 4 * n_fs;
    o.census = d + 5 * n_fs; o.diag = ddiag; o.state = dstate;
    std::vector<uint32_t*> lists;  // member lists of the groups, released behind the synchronisation below
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        if (kGroupW[gi] != 1) continue;  // a read annotated to a miRNA is at most 3 nt longer than it: the <= 31-nt groups
        const ResGroup& g = res->g[gi];
        const ReadGroup& rg = U->g[gi];
        if (!g.n) continue;
        // both kernels run one workgroup per chunk of reads: the members of chunk b are list[b * chunk ...], n_list[b] of them
        const uint32_t tgrid = (uint32_t)grid_for(c, g.n);
        uint32_t chunk = ((uint32_t)g.n + tgrid - 1) / tgrid;
        chunk = (chunk + MIRGE_BLOCK - 1) / MIRGE_BLOCK * MIRGE_BLOCK;
        uint32_t* dlist = nullptr;
        CHECK(dalloc(c, &dlist, (size_t)tgrid * chunk + tgrid));
        uint32_t* dnlist = dlist + (size_t)tgrid * chunk;
        {
            LaunchScope ls(c, "k_member_list", g.n);
            const TallyMember pred{g.pass, g.ref, rg.counts, S, exact_pass, iso_pass, dfam, dfreq};
            hipLaunchKernelGGL((k_member_list<TallyMember, MIRGE_BLOCK>), dim3(tgrid), dim3(MIRGE_BLOCK), 0, c->stream, (uint32_t)g.n, chunk,
                               pred, dlist, dnlist);
        }
        LaunchScope ls(c, "k_tally", g.n);
        hipLaunchKernelGGL(k_tally, dim3(tgrid), dim3(MIRGE_BLOCK), 0, c->stream, view_of<1>(rg), rg.base,
                           (const uint32_t*)rg.orig, g.pass, g.ref, rg.counts, S, exact_pass, iso_pass, dfam, dtb, dtl,
                           (const uint8_t*)dret, dfreq, o, (const uint32_t*)dlist, (const uint32_t*)dnlist, chunk);
        lists.push_back(dlist);
    }
    // straight into the caller's arrays (the census alone is 12 KiB per family and sample: no staging copy of it)
    for (int t = 0; t < 5 && n_fam; t++)  // the device block is sized for max(n_fam, 1)
        HIPOK(hipMemcpyAsync(fam_tables + (size_t)t * n_fam * S, d + (size_t)t * n_fs, (size_t)n_fam * S * 8, hipMemcpyDeviceToHost, c->stream));
    if (n_fam) HIPOK(hipMemcpyAsync(census, d + 5 * n_fs, (size_t)n_fam * S * MIRGE_TALLY_MAXPOS * 16 * 3 * 8, hipMemcpyDeviceToHost, c->stream));
    if (diag_out && U->n) HIPOK(hipMemcpyAsync(diag_out, ddiag, (size_t)U->n, hipMemcpyDeviceToHost, c->stream));
    if (state_out && U->n) HIPOK(hipMemcpyAsync(state_out, dstate, (size_t)U->n, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    c->drain();
    c->release(d); c->release(dtb); c->release(dtl); c->release(dfam); c->release(dret); c->release(dfreq);
    c->release(ddiag); c->release(dstate);
    for (uint32_t* l : lists) c->release(l);
    return 0;
}
