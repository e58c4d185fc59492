// native_join.hpp -- part of mirge_native.hip (one translation unit): count join, variant tally, timers and profile access.
#pragma once
// ------------------------------------------------------------------------------------------
// count join
// ------------------------------------------------------------------------------------------
extern "C" int mirge_count_join(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, int32_t exact_pass,
                                int32_t iso_pass, int64_t n_mirna, int64_t* class_sums, int64_t* exact, int64_t* iso) {
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
    unsigned long long* d = nullptr;
    CHECK(dalloc(c, &d, n_cls + 2 * n_tab));
    HIPOK(hipMemsetAsync(d, 0, (n_cls + 2 * n_tab) * 8, c->stream));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ResGroup& g = res->g[gi];
        if (!g.n) continue;
        LaunchScope ls(c, "k_join", g.n);
        hipLaunchKernelGGL(k_join, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, g.pass, g.ref,
                           U->g[gi].counts, g.n, S, P, exact_pass, iso_pass, d, d + n_cls, d + n_cls + n_tab);
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
    c->release(d);
    return 0;
}

extern "C" int mirge_count_join_host(mirge_ctx* c, const int8_t* pass, const int32_t* ref, const uint32_t* counts,
                                     int64_t n, int32_t S, int32_t P, int32_t exact_pass, int32_t iso_pass,
                                     int64_t n_mirna, int64_t* class_sums, int64_t* exact, int64_t* iso) {
    if (!c || !class_sums || !exact || !iso || n < 0 || S < 1 || P < 1 || P > MIRGE_MAX_PASSES || n_mirna < 0 ||
        (n > 0 && (!pass || !ref || !counts)))
        return fail(-1, "mirge_count_join_host: bad argument");
    if (n >= 0xFFFFFFF0ll) return fail(-5, "more than 2^32 rows");
    HIPOK(hipSetDevice(c->device));
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

extern "C" int mirge_variant_tally(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, const mirge_lib* mirna,
                                   int32_t exact_pass, int32_t iso_pass, int32_t iso_trim5, int64_t n_mirna,
                                   int64_t* accepted, int64_t* canonical, int64_t* census) {
    static_assert(MIRGE_TALLY_POSITIONS == MIRGE_TALLY_MAXPOS, "tally positions");
    if (!c || !U || !res || !mirna || !accepted || !canonical || !census || n_mirna != mirna->n_refs)
        return fail(-1, "mirge_variant_tally: bad argument");
    if (U->n_samples < 1) return fail(-1, "read set has no count matrix");
    if (res->n != U->n) return fail(-1, "result and read set differ in size");
    HIPOK(hipSetDevice(c->device));
    const int32_t S = U->n_samples;
    const size_t n_rs = (size_t)std::max<int64_t>(n_mirna, 1) * S, n_cen = n_rs * MIRGE_TALLY_MAXPOS * 16;
    unsigned long long* d = nullptr;
    CHECK(dalloc(c, &d, 2 * n_rs + n_cen));
    HIPOK(hipMemsetAsync(d, 0, (2 * n_rs + n_cen) * 8, c->stream));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        if (kGroupW[gi] != 1) continue;  // a read annotated to a miRNA is at most 3 nt longer than it
        const ResGroup& g = res->g[gi];
        if (!g.n) continue;
        LaunchScope ls(c, "k_tally", g.n);
        hipLaunchKernelGGL(k_tally, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, view_of<1>(U->g[gi]), g.pass, g.ref,
                           g.off, U->g[gi].counts, S, mirna->view(), exact_pass, iso_pass, iso_trim5, d, d + n_rs, d + 2 * n_rs);
    }
    std::vector<unsigned long long> h(2 * n_rs + n_cen);
    HIPOK(hipMemcpyAsync(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    c->drain();
    std::memcpy(accepted, h.data(), (size_t)n_mirna * S * 8);
    std::memcpy(canonical, h.data() + n_rs, (size_t)n_mirna * S * 8);
    std::memcpy(census, h.data() + 2 * n_rs, (size_t)n_mirna * S * MIRGE_TALLY_MAXPOS * 16 * 8);
    c->release(d);
    return 0;
}
