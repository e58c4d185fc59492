// native_cascade.hpp -- part of mirge_native.hip (one translation unit): cascade: merged steps, plan tables, staged and fused launches.
#pragma once
// ------------------------------------------------------------------------------------------
// cascade
// ------------------------------------------------------------------------------------------
struct ResGroup {
    uint32_t n = 0;
    int8_t* pass = nullptr;
    uint32_t* pos = nullptr;
    int8_t* mm = nullptr;
    int32_t* ref = nullptr;
    int32_t* off = nullptr;
};
struct mirge_result {
    mirge_ctx* ctx = nullptr;
    int64_t n = 0;
    int32_t n_pass = 0;
    ResGroup g[MIRGE_NGROUPS];
    const mirge_reads* reads = nullptr;  // borrowed: orig/base mapping (must outlive the fetch)
    uint32_t n_refs[MIRGE_MAX_PASSES] = {0};  // references of each pass's library (bounds of res.ref; checked by the join)
    uint32_t* dmeta = nullptr;           // mirge_collapse_cascade: the device-side read counts its kernels read
};

extern "C" void mirge_result_destroy(mirge_result* r) {
    if (!r) return;
    (void)join_pending_now(r->ctx);  // the side streams may still be writing what goes back to the pool here
    for (auto& g : r->g) {
        r->ctx->release(g.pass); r->ctx->release(g.pos); r->ctx->release(g.mm);
        r->ctx->release(g.ref); r->ctx->release(g.off);
    }
    r->ctx->release(r->dmeta);
    delete r;
}

// One library = the members' references in order (same bases, same separators), so that a position in
// the merged text minus the member's start is the position in the member's own text.
static int merged_library(mirge_ctx* c, const mirge_lib* const* members, int n, mirge_lib** out) {
    std::vector<uint64_t> uids;
    for (int i = 0; i < n; i++) uids.push_back(members[i]->uid);
    for (auto& m : c->merged)
        if (m.uids == uids) { *out = m.lib; return 0; }
    std::string seq;
    std::vector<int64_t> off;
    std::vector<const MirgeHostLib*> hl;
    for (int i = 0; i < n; i++) hl.push_back(&members[i]->h);
    merged_library_text(hl.data(), n, seq, off);  // (native_host.hpp)
    mirge_lib* L = nullptr;
    CHECK(mirge_lib_create(c, seq.data(), off.data(), (int64_t)off.size() - 1, &L));
    c->merged.push_back(mirge_ctx::Merged{uids, L});
    *out = L;
    return 0;
}

// build every probe table pass `p` can ask for, given the read lengths present
static int prepare_tables(mirge_lib* lib, const mirge_policy& pol, const int32_t* hist, bool long_present) {
    MirgePolicy p;
    std::memcpy(&p, &pol, sizeof(p));
    std::vector<ShapeJob> wanted;
    std::vector<bool> seen(MIRGE_SHAPE_SLOTS, false);
    // L = MIRGE_MAX_READ_LEN + 1 stands for the long class (k_cascade_long): a read of 256 nt or more is probed with the plan of
    // its first 31 bases -- of its whole head when a T run is stripped down to less -- so lengths 31 + trims (any below with -ttail)
    for (int L = 1; L <= MIRGE_MAX_READ_LEN + 1; L++) {
        const bool lng = L > MIRGE_MAX_READ_LEN;
        if (lng ? !long_present : !hist[L]) continue;
        // (the long class stands for EVERY length beyond 255: it fails a `len <` rule of up to 256, and no `len >` rule at all --
        // with len_gt >= 256 the stand-in 256 failed the test, no table was built, and a 400-nt read probed a null table)
        if (p.len_lt > 0 && !(L < p.len_lt) && !(lng && p.len_lt > MIRGE_MAX_READ_LEN + 1)) continue;
        if (p.len_gt > 0 && !(L > p.len_gt) && !lng) continue;
        int lo = L, hi = L;
        if (lng) lo = hi = 31 + p.trim5 + p.trim3;
        if (p.ttail) { lo = 1; hi = (lng ? 31 + p.trim5 + p.trim3 : L - 3); }  // any head length once the T run is gone
        for (int l0 = lo; l0 <= hi; l0++) {
            const int l = l0 - p.trim5 - p.trim3;
            if (l < 1 || l <= p.mm) continue;
            const int np = mirge_probe_count(p, l, lib->kmax, lib->h.total);
            for (int q = 0; q < np; q++) {
                MirgeProbe pr;
                mirge_probe_at(p, l, lib->kmax, lib->h.total, q, pr);
                if (pr.k1 <= 0) continue;
                const int sid = mirge_shape_id(pr.k1, pr.gap, pr.k2);
                if (seen[sid]) continue;
                seen[sid] = true;
                wanted.emplace_back();
                wanted.back().k1 = pr.k1; wanted.back().gap = pr.gap; wanted.back().k2 = pr.k2;
            }
        }
    }
    return lib_prepare_shapes(lib, wanted);
}

// the two counters k_cascade_heavy works with (reads listed, workgroups done; zero between launches: the kernel's last workgroup
// resets them) for the group with this tag, or nullptr when no library of the configuration holds a bucket beyond casc_big_t
static uint32_t* heavy_counters(mirge_ctx* c, const char* gtag) {
    if (!c->casc_big_t || !c->heavy_cnt) return nullptr;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
        if (std::strcmp(group_tag(gi), gtag) == 0) return c->heavy_cnt + 2 * gi;
    return c->heavy_cnt;
}
template <int W>
static void launch_heavy(mirge_ctx* c, const ReadGroup& rg, const ResGroup& out, const FusedSteps* dsteps, const ResolveTable& rt, const char* gtag,
                         uint32_t* hcnt, const uint32_t* hlist, const GroupView<W>& v, bool with_resolve) {
    char name[40];
    std::snprintf(name, sizeof(name), "k_cascade_heavy%s", gtag);
    LaunchScope ls(c, name, 0.0);
    const uint32_t grid = (uint32_t)c->n_cu * 2u;  // two 1024-thread workgroups per CU
    int32_t* rr = with_resolve ? out.ref : nullptr;
    int32_t* ro = with_resolve ? out.off : nullptr;
    if (rg.nmask)
        hipLaunchKernelGGL((k_cascade_heavy<W, true>), dim3(grid), dim3(MIRGE_HEAVY_THREADS), 0, c->cur, dsteps, rt, v, hcnt, hlist, out.pass, out.pos, out.mm, rr, ro);
    else
        hipLaunchKernelGGL((k_cascade_heavy<W, false>), dim3(grid), dim3(MIRGE_HEAVY_THREADS), 0, c->cur, dsteps, rt, v, hcnt, hlist, out.pass, out.pos, out.mm, rr, ro);
}

template <int W>
static int cascade_group(mirge_ctx* c, const ReadGroup& rg, ResGroup& out, const std::vector<PassStep>& steps,
                         const mirge_policy* pol, const ResolveTable& rt, const char* gtag,
                         const uint32_t* n_dev = nullptr, uint32_t n_cap = 0, const BulkWalks* dwalks = nullptr) {
    // n_dev != nullptr: the group's read count is not on the host yet (see k_pass); everything is sized for
    // n_cap >= the count, the caller fills out.n in later
    out.n = n_dev ? 0 : rg.n;
    if (!n_dev && !rg.n) return 0;
    const uint32_t n = n_dev ? n_cap : rg.n;
    CHECK(dalloc(c, &out.pass, n));
    CHECK(dalloc(c, &out.pos, n));
    CHECK(dalloc(c, &out.mm, n));
    CHECK(dalloc(c, &out.ref, n));
    CHECK(dalloc(c, &out.off, n));
    // every workgroup keeps its own survivor segment through all passes: no global cursor.  All workgroups
    // must be resident at once: a grid of 8 per CU ran the workgroups that did not fit as a second round on an
    // almost empty machine (average occupancy 47 %, SQ_WAVE_CYCLES; -20 % kernel time with the right grid).
    // Measured on MI355X (tools/occ_sweep.sh, profiles/README.md): kernel time falls up to 6 workgroups per CU
    // and jumps back by 30 % at 7 and beyond -- for the 69-VGPR build and for 57/63-VGPR builds alike, although
    // hipOccupancyMaxActiveBlocksPerMultiprocessor reports 7.  The limiter is the scalar register file: k_pass
    // holds its library view / policy / pointers in ~104 SGPRs (+ VCC etc.), 6 such waves fit a SIMD; builds
    // capped with amdgpu_num_sgpr(96) / (80) move the cliff to 8 / 9 workgroups, but the extra waves buy
    // nothing here (the heavy passes are sector-bound) and the SGPR spills cost a little.  The runtime reports
    // VGPRs only, hence the explicit 6: grid = min(occupancy query, register bound, 6) per CU.
    // MIRGE_WG_PER_CU overrides (sweeps).
    static int wg_per_cu[9] = {0};
    if (!wg_per_cu[W]) {
        int nb = 0;
        hipFuncAttributes fa;
        // (the bulk kernel's own attributes -- until round 6 k_pass<W, 15>'s stood in for them; that kernel is now always the repeat-aware
        //  build and may take a few registers more than the attribute-capped bulk kernel)
        HIPOK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_cascade_bulk<W, false, true>), MIRGE_BLOCK, 0));
        HIPOK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_cascade_bulk<W, false, true>)));
        const int by_regs = 512 / std::max(16, (fa.numRegs + 15) / 16 * 16);  // waves per SIMD = 4-wave workgroups per CU
        wg_per_cu[W] = std::max(1, std::min({nb, by_regs, W == 1 ? MIRGE_BULK_WAVES : 6}));
        if (std::getenv("MIRGE_WG_PER_CU")) wg_per_cu[W] = std::max(1, std::atoi(std::getenv("MIRGE_WG_PER_CU")));
        if (std::getenv("MIRGE_HOST_TIMING"))
            std::fprintf(stderr, "[host] k_pass<%d>: %d VGPRs, occupancy query %d -> %d workgroups per CU\n", W, fa.numRegs, nb, wg_per_cu[W]);
    }
    const uint32_t grid = (uint32_t)std::min<size_t>((size_t)grid_for(c, n), (size_t)c->n_cu * wg_per_cu[W]);
    uint32_t cap = (n + grid - 1) / grid;
    cap = (cap + MIRGE_BLOCK - 1) / MIRGE_BLOCK * MIRGE_BLOCK;
    uint32_t *actA = nullptr, *actB = nullptr, *seg_n = nullptr;
    // (round 6) reads whose probe bucket no wave should walk alone go to k_cascade_heavy: their list, this group's counters
    uint32_t* const hcnt = heavy_counters(c, gtag);
    uint32_t* hlist = nullptr;
    if (hcnt) CHECK(dalloc(c, &hlist, (size_t)n));
    CHECK(dalloc(c, &actA, (size_t)grid * cap));
    CHECK(dalloc(c, &actB, (size_t)grid * cap));
    CHECK(dalloc(c, &seg_n, (size_t)grid * (MIRGE_MAX_PASSES + 3)));  // (+ 2 rows: the workgroups' start / end clocks, k_cascade_bulk)
    if (steps.empty()) {  // no pass runs: nothing will write the "unannotated" marks
        HIPOK(hipMemsetAsync(out.pass, 0xFF, n, c->cur));
        HIPOK(hipMemsetAsync(out.mm, 0xFF, n, c->cur));
    }
    GroupView<W> v = view_of<W>(rg);
    if (n_dev) v.n = n;  // only a stride for W > 1; the deferred-count path is the one-word bulk group
    const uint32_t* act_in = nullptr;
    uint32_t* act_out = actA;
    int stage = 0;
    char name[32];
    std::vector<std::pair<int, int>> stage_of_pass;  // (profile record, stage) for unit accounting
    // all passes in ONE launch (k_cascade_bulk): a workgroup's segment and survivor lists are its own through the whole
    // cascade, so the launches between the passes were chip-wide barriers nothing needed.  MIRGE_BULK_FUSED=0: one launch
    // per pass (A/B, per-pass profiles).
    static const bool bulk_fused = !(std::getenv("MIRGE_BULK_FUSED") && std::atoi(std::getenv("MIRGE_BULK_FUSED")) == 0);
    if (bulk_fused && dwalks && steps.size() > 1) {
        std::snprintf(name, sizeof(name), "k_cascade_bulk%s", gtag);
        LaunchScope ls(c, name, 0.0);
        if (ls.rec >= 0)
            for (size_t k = 0; k < steps.size(); k++) stage_of_pass.emplace_back(ls.rec, (int)k);  // units = reads handed to every pass
        // (a group without ambiguous calls runs the build of the kernel in which the N masks are compile-time zeros)
        // (REP: the repeat-aware build, for configurations with an outlier bucket -- align_hybrid)
#define MIRGE_LAUNCH_BULK(HASN_, REP_)                                                                                                              \
    hipLaunchKernelGGL((k_cascade_bulk<W, HASN_, REP_>), dim3(grid), dim3(MIRGE_BLOCK), 0, c->cur, dwalks, v, actA, actB, seg_n, cap, out.pass, out.pos, \
                       out.mm, n_dev, hcnt, hlist)
        if (rg.nmask) { if (c->casc_rep) MIRGE_LAUNCH_BULK(true, true); else MIRGE_LAUNCH_BULK(true, false); }
        else { if (c->casc_rep) MIRGE_LAUNCH_BULK(false, true); else MIRGE_LAUNCH_BULK(false, false); }
#undef MIRGE_LAUNCH_BULK
        stage = (int)steps.size();
        if (c->profiling && W == 1) {  // the workgroups' clocks of this launch, for mirge_cascade_wg_times (stream-ordered copy)
            if (c->wg_pinned_words < 2 * (size_t)grid) {
                if (c->wg_pinned) (void)hipHostFree(c->wg_pinned);
                c->wg_pinned = nullptr; c->wg_pinned_words = 0;
                if (hipHostMalloc((void**)&c->wg_pinned, 2 * (size_t)grid * 4, hipHostMallocDefault) == hipSuccess) c->wg_pinned_words = 2 * (size_t)grid;
            }
            if (c->wg_pinned_words >= 2 * (size_t)grid) {
                HIPOK(hipMemcpyAsync(c->wg_pinned, seg_n + (size_t)grid * (MIRGE_MAX_PASSES + 1), 2 * (size_t)grid * 4, hipMemcpyDeviceToHost, c->cur));
                c->wg_grid = grid;
            }
        }
    } else
    for (const PassStep& st : steps) {
        const int32_t p = st.p0;
        MirgePolicy mp;
        std::memcpy(&mp, &pol[p], sizeof(mp));
        mp.reserved = hcnt ? (int32_t)c->casc_big_t : 0;
        {
            if (st.np > 1) std::snprintf(name, sizeof(name), "k_pass[%d-%d]%s", (int)p, (int)(p + st.np - 1), gtag);
            else std::snprintf(name, sizeof(name), "k_pass[%d]%s", (int)p, gtag);
            LaunchScope ls(c, name, 0.0);
            if (ls.rec >= 0) stage_of_pass.emplace_back(ls.rec, stage);
            const uint32_t* sn_in = seg_n + (size_t)grid * (stage > 0 ? stage - 1 : 0);
            uint32_t* sn_out = seg_n + (size_t)grid * stage;
#define MIRGE_LAUNCH_PASS(SLOT)                                                                                       \
    hipLaunchKernelGGL((k_pass<W, SLOT>), dim3(grid), dim3(MIRGE_BLOCK), 0, c->cur, st.lib->view(), mp, st.mi, st.dplan, v, act_in, \
                       sn_in, act_out, sn_out, cap, p, out.pass, out.pos, out.mm, n_dev, hcnt, hlist)
            switch (p) {
                case 0: MIRGE_LAUNCH_PASS(0); break;
                case 1: MIRGE_LAUNCH_PASS(1); break;
                case 2: MIRGE_LAUNCH_PASS(2); break;
                case 3: MIRGE_LAUNCH_PASS(3); break;
                case 4: MIRGE_LAUNCH_PASS(4); break;
                case 5: MIRGE_LAUNCH_PASS(5); break;
                case 6: MIRGE_LAUNCH_PASS(6); break;
                case 7: MIRGE_LAUNCH_PASS(7); break;
                case 8: MIRGE_LAUNCH_PASS(8); break;
                case 9: MIRGE_LAUNCH_PASS(9); break;
                default: MIRGE_LAUNCH_PASS(15); break;
            }
#undef MIRGE_LAUNCH_PASS
        }
        act_in = act_out;
        act_out = (act_out == actA) ? actB : actA;
        stage++;
    }
    if (hcnt && stage > 0) launch_heavy<W>(c, rg, out, c->casc_dsteps, rt, gtag, hcnt, hlist, v, false);  // (k_resolve follows for every read)
    {
        std::snprintf(name, sizeof(name), "k_resolve%s", gtag);
        LaunchScope ls(c, name, n);
        hipLaunchKernelGGL(k_resolve, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->cur, rt, out.pass, out.pos, n, out.ref, out.off, n_dev);
    }
    if (c->profiling && c->prof_units && stage > 0 && !stage_of_pass.empty()) {  // units of a pass = reads it was handed = survivors of the stage before
        // copied now (stream-ordered), summed after the one synchronisation at the end of the call
        const size_t words = (size_t)grid * stage;
        if (c->prof_used + words <= MIRGE_PROF_PINNED_WORDS) {
            uint32_t* dst = c->prof_pinned + c->prof_used;
            HIPOK(hipMemcpyAsync(dst, seg_n, words * 4, hipMemcpyDeviceToHost, c->cur));
            for (auto& sp : stage_of_pass) c->prof_pending.push_back(ProfUnits{sp.first, sp.second, grid, dst, (double)n});
            c->prof_used += words;
        }
    }
    c->defer(actA); c->defer(actB); c->defer(seg_n); c->defer(hlist);
    return 0;
}

// a small group's whole cascade as one launch (k_cascade_fused)
template <int W>
static int cascade_group_fused(mirge_ctx* c, const ReadGroup& rg, ResGroup& out, const FusedSteps* dsteps,
                               const ResolveTable& rt, const char* gtag) {
    out.n = rg.n;
    if (!rg.n) return 0;
    const uint32_t n = rg.n;
    CHECK(dalloc(c, &out.pass, n));
    CHECK(dalloc(c, &out.pos, n));
    CHECK(dalloc(c, &out.mm, n));
    CHECK(dalloc(c, &out.ref, n));
    CHECK(dalloc(c, &out.off, n));
    uint32_t* const hcnt = heavy_counters(c, gtag);
    uint32_t* hlist = nullptr;
    if (hcnt) CHECK(dalloc(c, &hlist, (size_t)n));
    char name[32];
    // (round 6) a tiny group: all passes at once (k_cascade_spec), then the first answer per read (k_cascade_pick).  MIRGE_SPEC_MAX: the
    // largest group (reads) that takes this route, 0 = never (tests, A/B)
    static const uint32_t spec_max = std::getenv("MIRGE_SPEC_MAX") ? (uint32_t)std::strtoul(std::getenv("MIRGE_SPEC_MAX"), nullptr, 10) : 32768u;
    const int nsteps = (int)c->casc_steps.size();
    if (n <= spec_max && nsteps > 1) {
        unsigned long long* answers = nullptr;
        const uint32_t rounds = (n + MIRGE_BLOCK - 1) / MIRGE_BLOCK, stride = rounds * MIRGE_BLOCK;
        CHECK(dalloc(c, &answers, (size_t)nsteps * stride));
        const dim3 grid(rounds, (unsigned)nsteps);
        // MIRGE_SPEC_TICKETS=1 (round 6 experiment, OFF by default): the pick in the same launch (the workgroup that ends a round's last
        // step does it: `tickets`, k_cascade_spec) -- one kernel less in a chain of launches the step's end waits for on a sample with few
        // unique reads.  Measured WORSE (profiles/r06_ab_spec_tickets.txt): 0.539 vs 0.408 ms on that sample, 1.259 vs 1.190 ms on the
        // default draw -- the two agent-scope fences per workgroup write back and invalidate the XCD's L2 under the bulk kernel that runs
        // beside it (its own launch time: 0.84 vs 0.78 ms); a kernel boundary is the cheaper fence here.
        static const bool tickets_on = std::getenv("MIRGE_SPEC_TICKETS") && std::atoi(std::getenv("MIRGE_SPEC_TICKETS")) == 1;
        uint32_t* tickets = nullptr;
        if (tickets_on && c->spec_tickets && rounds <= MIRGE_SPEC_TICKET_ROUNDS)
            for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
                if (std::strcmp(group_tag(gi), gtag) == 0) tickets = c->spec_tickets + (size_t)gi * MIRGE_SPEC_TICKET_ROUNDS;
        {
            std::snprintf(name, sizeof(name), "k_cascade_spec%s", gtag);
            LaunchScope ls(c, name, n);
#define MIRGE_LAUNCH_SPEC(HASN_, REP_)                                                                                                          \
    hipLaunchKernelGGL((k_cascade_spec<W, HASN_, REP_>), grid, dim3(MIRGE_BLOCK), 0, c->cur, dsteps, rt, view_of<W>(rg), answers, stride, tickets, \
                       out.pass, out.pos, out.mm, out.ref, out.off, hcnt, hlist)
            if (rg.nmask) { if (c->casc_rep) MIRGE_LAUNCH_SPEC(true, true); else MIRGE_LAUNCH_SPEC(true, false); }
            else { if (c->casc_rep) MIRGE_LAUNCH_SPEC(false, true); else MIRGE_LAUNCH_SPEC(false, false); }
#undef MIRGE_LAUNCH_SPEC
            if (!tickets)
                hipLaunchKernelGGL(k_cascade_pick<W>, dim3(rounds), dim3(MIRGE_BLOCK), 0, c->cur, dsteps, rt, n, stride,
                                   (const unsigned long long*)answers, out.pass, out.pos, out.mm, out.ref, out.off, hcnt, hlist);
        }
        if (hcnt) launch_heavy<W>(c, rg, out, dsteps, rt, gtag, hcnt, hlist, view_of<W>(rg), true);
        c->defer(hlist); c->defer(answers);
        return 0;
    }
    std::snprintf(name, sizeof(name), "k_cascade_fused%s", gtag);
    {
        LaunchScope ls(c, name, n);
        const uint32_t grid = std::min<uint32_t>((n + MIRGE_BLOCK - 1) / MIRGE_BLOCK, (uint32_t)c->n_cu * 8);
        // (HASN = false: no ambiguous calls in the group: the build whose N masks are compile-time zeros; REP: see align_hybrid)
#define MIRGE_LAUNCH_FUSED(HASN_, REP_)                                                                                                        \
    hipLaunchKernelGGL((k_cascade_fused<W, HASN_, REP_>), dim3(grid), dim3(MIRGE_BLOCK), 0, c->cur, dsteps, rt, view_of<W>(rg), out.pass, out.pos, \
                       out.mm, out.ref, out.off, hcnt, hlist)
        if (rg.nmask) { if (c->casc_rep) MIRGE_LAUNCH_FUSED(true, true); else MIRGE_LAUNCH_FUSED(true, false); }
        else { if (c->casc_rep) MIRGE_LAUNCH_FUSED(false, true); else MIRGE_LAUNCH_FUSED(false, false); }
#undef MIRGE_LAUNCH_FUSED
    }
    if (hcnt) launch_heavy<W>(c, rg, out, dsteps, rt, gtag, hcnt, hlist, view_of<W>(rg), true);
    c->defer(hlist);
    return 0;
}

// the long class (reads beyond 255 nt): one launch, one thread per read (k_cascade_long)
static int cascade_group_long(mirge_ctx* c, const ReadGroup& rg, ResGroup& out, const FusedSteps* dsteps, const ResolveTable& rt,
                              const char* gtag) {
    out.n = rg.n;
    if (!rg.n) return 0;
    const uint32_t n = rg.n;
    CHECK(dalloc(c, &out.pass, n));
    CHECK(dalloc(c, &out.pos, n));
    CHECK(dalloc(c, &out.mm, n));
    CHECK(dalloc(c, &out.ref, n));
    CHECK(dalloc(c, &out.off, n));
    char name[32];
    std::snprintf(name, sizeof(name), "k_cascade_long%s", gtag);
    LaunchScope ls(c, name, n);
    const uint32_t grid = std::min<uint32_t>((n + 63) / 64, (uint32_t)c->n_cu * 8);
    hipLaunchKernelGGL(k_cascade_long, dim3(grid), dim3(64), 0, c->cur, dsteps, rt, long_view_of(rg), out.pass, out.pos, out.mm, out.ref, out.off);
    return 0;
}

// steps (merged runs, probe tables, plan tables), resolve table and the fused kernel's device step list for
// one (libraries, policies, read-length set) configuration
static int cascade_prepare(mirge_ctx* c, const mirge_lib* const* libs, const mirge_policy* pol, int32_t n_pass,
                           const int32_t* hist, bool long_present, std::vector<PassStep>& steps, ResolveTable& rt, const FusedSteps** dsteps_out,
                           const BulkWalks** dwalks_out) {
    for (int p = 0; p < MIRGE_MAX_PASSES; p++) { rt.ref_start[p] = nullptr; rt.coarse[p] = nullptr; rt.n_refs[p] = 0; }
    steps.clear();
    for (int32_t p = 0; p < n_pass; p++) {
        if (!libs[p]) continue;
        if (libs[p]->ctx->device != c->device) return fail(-1, "library lives on another device");
        if (pol[p].mm < 0 || pol[p].mm > 3 || pol[p].trim5 < 0 || pol[p].trim5 > 31 || pol[p].trim3 < 0)
            return fail(-1, "unsupported policy");
        rt.ref_start[p] = libs[p]->dref_start;
        rt.coarse[p] = libs[p]->dcoarse;
        rt.n_refs[p] = (uint32_t)libs[p]->n_refs;
    }
    for (int32_t p = 0; p < n_pass;) {
        if (!libs[p]) { p++; continue; }
        // a run of consecutive passes with one and the same policy over distinct libraries becomes ONE
        // launch over their concatenation (k_pass ranks candidates by member library first, so the
        // cascade's "first library with a hit wins" is unchanged): human set -> passes 4,5,6
        int np = 1;
        uint64_t total = libs[p]->h.total;
        static const bool merge_on = !(std::getenv("MIRGE_MERGE_PASSES") && std::getenv("MIRGE_MERGE_PASSES")[0] == '0');
        while (merge_on && np < 4 && p + np < n_pass && libs[p + np] && std::memcmp(&pol[p + np], &pol[p], sizeof(mirge_policy)) == 0 &&
               total + libs[p + np]->h.total < 0xFFFFFFF0ull) {
            bool distinct = true;
            for (int q = 0; q < np; q++) distinct &= libs[p + q] != libs[p + np];
            if (!distinct) break;
            total += libs[p + np]->h.total;
            np++;
        }
        PassStep st;
        st.p0 = p; st.np = np;
        st.mi.n = np;
        for (int i = 0; i < 4; i++) st.mi.bound[i] = 0;
        if (np == 1) st.lib = libs[p];
        else {
            mirge_lib* m = nullptr;
            CHECK(merged_library(c, libs + p, np, &m));
            st.lib = m;
            uint64_t b = 0;
            for (int i = 0; i < np; i++) { st.mi.bound[i] = (uint32_t)b; b += libs[p + i]->h.total; }
        }
        CHECK(prepare_tables(const_cast<mirge_lib*>(st.lib), pol[p], hist, long_present));
        steps.push_back(st);
        p += np;
    }
    // tabulated probe plans: built and uploaded once per (library, policy), then reused by every call
    for (auto& st : steps) {
        MirgePolicy mp;
        std::memcpy(&mp, &pol[st.p0], sizeof(mp));
        const MirgePlanTable* dp = nullptr;
        for (auto& e : c->plans)
            if (e.uid == st.lib->uid && std::memcmp(&e.pol, &mp, sizeof(mp)) == 0) { dp = e.dplan; break; }
        if (!dp) {
            auto h = std::make_unique<MirgePlanTable>();
            mirge_plan_table_fill(mp, st.lib->kmax, st.lib->h.total, *h);
            MirgePlanTable* d = nullptr;
            HIPOK(hipMalloc((void**)&d, sizeof(MirgePlanTable)));
            HIPOK(hipMemcpy(d, h.get(), sizeof(MirgePlanTable), hipMemcpyHostToDevice));
            c->plans.push_back(mirge_ctx::PlanEntry{st.lib->uid, mp, d});
            dp = d;
        }
        st.dplan = dp;
    }
    // (round 6) Does any library of this configuration hold a probe bucket beyond MIRGE_BIG_T windows (default 8192; 0: never defer)?
    // Then the steps carry that threshold (MirgePolicy::reserved): a read that meets such a bucket is answered by k_cascade_heavy, a
    // workgroup per read.  Uniform-random libraries hold none: their cascades run as before, without the extra launch.
    static const uint32_t big_t_env = std::getenv("MIRGE_BIG_T") ? (uint32_t)std::strtoul(std::getenv("MIRGE_BIG_T"), nullptr, 10) : 8192u;
    static const bool big_t_forced = std::getenv("MIRGE_BIG_T") != nullptr;  // (tests: every bucket beyond the given size defers, outlier or not)
    // MIRGE_CASCADE_REP=1 / 0: the repeat-aware build of the cascade kernels always / never (tests, A/B); by itself: when a library holds an
    // outlier bucket (>= 1024 windows and 32 x what a uniform text would put there)
    static const int rep_env = std::getenv("MIRGE_CASCADE_REP") ? std::atoi(std::getenv("MIRGE_CASCADE_REP")) : -1;
    c->casc_big_t = big_t_forced ? big_t_env : 0;
    c->casc_rep = rep_env > 0 || big_t_forced;
    for (auto& st : steps) {
        if (st.lib->max_bucket > 0 && rep_env != 0) c->casc_rep = true;
        if (big_t_env && st.lib->max_bucket > big_t_env) c->casc_big_t = big_t_env;
    }
    if (!c->casc_rep) c->casc_big_t = 0;  // (only the repeat-aware build defers)
    if (!c->spec_tickets) {  // (k_cascade_spec's per-round counters, one row per read group; zero between launches)
        HIPOK(hipMalloc((void**)&c->spec_tickets, (size_t)MIRGE_NGROUPS * MIRGE_SPEC_TICKET_ROUNDS * 4));
        HIPOK(hipMemset(c->spec_tickets, 0, (size_t)MIRGE_NGROUPS * MIRGE_SPEC_TICKET_ROUNDS * 4));
    }
    if (c->casc_big_t && !c->heavy_cnt) {
        HIPOK(hipMalloc((void**)&c->heavy_cnt, 2 * MIRGE_NGROUPS * 4));
        HIPOK(hipMemset(c->heavy_cnt, 0, 2 * MIRGE_NGROUPS * 4));
    }
    // the step list of the fused small-group kernel lives in device memory; uploaded when it changes
    auto fs = std::make_unique<FusedSteps>();
    std::memset(fs.get(), 0, sizeof(FusedSteps));
    fs->n = (int32_t)steps.size();
    for (size_t i = 0; i < steps.size(); i++) {
        FusedStep& f = fs->s[i];
        f.lib = steps[i].lib->view();
        std::memcpy(&f.pol, &pol[steps[i].p0], sizeof(MirgePolicy));
        f.pol.reserved = (int32_t)c->casc_big_t;
        f.mi = steps[i].mi;
        f.plan = steps[i].dplan;
        f.pass_id = steps[i].p0;
    }
    const FusedSteps* dsteps = nullptr;
    const FusedSteps* fs_host = fs.get();  // (stays valid: either `fs` itself or its new home in c->fused)
    for (auto& e : c->fused)
        if (std::memcmp(e.host.get(), fs.get(), sizeof(FusedSteps)) == 0) { dsteps = e.dev; break; }
    if (!dsteps) {
        if (c->fused.size() >= 64) {  // callers cycling through libraries: start over
            HIPOK(hipDeviceSynchronize());
            for (auto& e : c->fused) (void)hipFree(e.dev);
            c->fused.clear();
        }
        FusedSteps* d = nullptr;
        HIPOK(hipMalloc((void**)&d, sizeof(FusedSteps)));
        HIPOK(hipMemcpy(d, fs.get(), sizeof(FusedSteps), hipMemcpyHostToDevice));
        c->fused.push_back(mirge_ctx::FusedEntry{std::move(fs), d});
        dsteps = d;
    }
    *dsteps_out = dsteps;
    // The walk lists of k_cascade_bulk (kernels_cascade.hpp): the same steps; in the list of the ONE-WORD read group every pass
    // that one whole-read lookup can answer rides in front of the next alignment pass or behind the previous one.  Such a pass:
    // one library (no merged run), no mismatch allowed anywhere in the read it is handed -- "-v 0", or "-n 0" when every
    // eligible read lies inside the seed --, a library small enough (lib_exact_table).  Wider groups walk one pass per walk.
    // MIRGE_EXACT_WALKS=0: one walk per pass for every group, as round 4 (A/B; the staged k_pass path never uses the tables).
    static const bool exact_on = !(std::getenv("MIRGE_EXACT_WALKS") && std::atoi(std::getenv("MIRGE_EXACT_WALKS")) == 0);
    std::vector<ExactStep> ex(steps.size());
    std::vector<char> is_exact(steps.size(), 0);
    {
        bool dropped = false;
        // (only the context a library was made in builds, names and drops its whole-read tables: a library borrowed from
        //  another context of the same device keeps the probe path, so no other context's walk lists can name a dropped table)
        for (size_t i = 0; i < steps.size() && exact_on; i++)
            if (steps[i].lib->ctx == c) dropped |= lib_exact_trim(const_cast<mirge_lib*>(steps[i].lib));
        if (dropped) {  // walk lists kept from earlier configurations point into the dropped tables
            for (auto& e : c->walks) (void)hipFree(e.dev);
            c->walks.clear();
        }
    }
    for (size_t i = 0; i < steps.size() && exact_on; i++) {
        const PassStep& st = steps[i];
        MirgePolicy p;
        std::memcpy(&p, &pol[st.p0], sizeof(p));
        if (st.np != 1 || p.mm != 0 || st.lib->ctx != c) continue;
        uint32_t lmask = 0;  // lengths the one-word reads of this batch can have when they reach the pass's lookup
        int lmax = 0;
        for (int L = 1; L <= 31; L++) {
            if (!hist[L]) continue;
            if (p.len_lt > 0 && !(L < p.len_lt)) continue;
            if (p.len_gt > 0 && !(L > p.len_gt)) continue;
            const int lo = p.ttail ? 1 : L, hi = p.ttail ? L - 3 : L;
            for (int l0 = lo; l0 <= hi; l0++) {
                const int l = l0 - p.trim5 - p.trim3;
                if (l < 1 || l <= p.mm) continue;
                lmask |= 1u << l;
                lmax = std::max(lmax, l);
            }
        }
        if (p.mode == 0 && lmax > p.seedlen) continue;  // "-n 0" beyond the seed admits mismatches: not exact
        mirge_lib::ExactTab tab;
        CHECK(lib_exact_table(const_cast<mirge_lib*>(st.lib), lmask, &tab));
        if (!tab.slots) continue;  // no one-word read can reach the pass, or the library is too large: the probe path
        ExactStep& e = ex[i];
        e.slots = tab.slots; e.T = st.lib->dT; e.mask = tab.mask; e.pass_id = st.p0; e.step = (int32_t)i; e.pol = p;
        is_exact[i] = 1;
    }
    if (c->walks.size() + 2 > 64) {  // callers cycling through libraries: start over (room for both lists of this configuration)
        HIPOK(hipDeviceSynchronize());
        for (auto& e : c->walks) (void)hipFree(e.dev);
        c->walks.clear();
    }
    for (int wide = 0; wide < 2; wide++) {
        auto bw = std::make_unique<BulkWalks>();
        std::memset(bw.get(), 0, sizeof(BulkWalks));
        for (size_t i = 0; i < steps.size();) {
            BulkWalk& w = bw->w[bw->n++];
            if (!wide && is_exact[i]) w.pre = ex[i++];
            if (!wide && !MIRGE_EXACT_RIDE && w.pre.slots) {  // exact steps in walks of their own (two consecutive ones share one)
                if (i < steps.size() && is_exact[i]) w.post = ex[i++];
                continue;
            }
            if (i < steps.size() && (wide || !is_exact[i])) {
                w.has_main = 1;
                w.main_step = (int32_t)i;
                w.main = fs_host->s[i];
                i++;
                const MirgePolicy& mp = w.main.pol;  // (`post` looks the read up as the main pass left it: untrimmed)
                if (!wide && i < steps.size() && is_exact[i] && !mp.trim5 && !mp.trim3 && !mp.ttail) w.post = ex[i++];
            }
        }
        const BulkWalks* dwalks = nullptr;
        for (auto& e : c->walks)
            if (std::memcmp(e.host.get(), bw.get(), sizeof(BulkWalks)) == 0) { dwalks = e.dev; break; }
        if (!dwalks) {
            BulkWalks* d = nullptr;
            HIPOK(hipMalloc((void**)&d, sizeof(BulkWalks)));
            HIPOK(hipMemcpy(d, bw.get(), sizeof(BulkWalks), hipMemcpyHostToDevice));
            c->walks.push_back(mirge_ctx::WalksEntry{std::move(bw), d});
            dwalks = d;
        }
        dwalks_out[wide] = dwalks;
    }
    return 0;
}

// (libraries, policies, read lengths present) -> c->casc_steps / casc_rt / casc_dsteps, kept between calls
static int cascade_config(mirge_ctx* c, const mirge_lib* const* libs, const mirge_policy* pol, int32_t n_pass, const int32_t* hist,
                          bool long_present) {
    // Everything up to the launches depends only on (libraries, policies, read lengths present): it is
    // kept from the previous call and reused when those are unchanged (~45 us of host time per call otherwise,
    // on the critical path between the collapse's synchronisation and the first pass)
    std::string key;
    key.append(reinterpret_cast<const char*>(&n_pass), sizeof(n_pass));
    for (int32_t p = 0; p < n_pass; p++) {
        const uint64_t uid = libs[p] ? libs[p]->uid : 0;
        key.append(reinterpret_cast<const char*>(&uid), sizeof(uid));
        key.append(reinterpret_cast<const char*>(&pol[p]), sizeof(mirge_policy));
    }
    for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) key.push_back(hist[L] ? 1 : 0);
    key.push_back(long_present ? 1 : 0);
    if (c->casc_key != key) {
        c->casc_key.clear();
        CHECK(cascade_prepare(c, libs, pol, n_pass, hist, long_present, c->casc_steps, c->casc_rt, &c->casc_dsteps, c->casc_dwalks));
        c->casc_key = key;
    }
    return 0;
}

// launches of every group but `skip` (already queued by the caller; -1 = none): small groups first, the bulk last
static int cascade_launch_groups(mirge_ctx* c, const mirge_reads* R, mirge_result* res, const mirge_policy* pol, int skip, int big_is = -1) {
    const std::vector<PassStep>& steps = c->casc_steps;
    const ResolveTable& rt = c->casc_rt;
    const FusedSteps* dsteps = c->casc_dsteps;
    // MIRGE_FUSED_MAX: largest group (reads) that takes the one-launch path; 0 = always staged (tests).  Beyond ~0.5 M reads
    // the staged passes' compaction pays for their launches (20 M-read sample, 0.7 M reads of 32-64 nt: 3.17 -> 3.08 ms)
    const uint32_t fused_max = small_fused_max();  // (native_collapse.hpp: the one reading of the variable)
    int rc = 0;
    const int big = big_is >= 0 ? big_is : largest_group(R);  // (big_is: the bulk group of a read set whose bulk count is not known yet)
    CHECK(stream_fork(c));
    // enqueue order: the small groups first (one fused launch each, or the staged launches if a group is too
    // large for that), the bulk group last: measured, its 2048-workgroup launches otherwise hold every CU and the
    // small kernels squeeze in between them, stretching single passes of the bulk group by 30 %
    int order[MIRGE_NGROUPS], no = 0;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) if (gi != big) order[no++] = gi;
    // the largest small group first: it gets extra stream 0, the one the main stream waits for directly (stream_join)
    std::stable_sort(order, order + no, [&](int a, int b) { return R->g[a].n > R->g[b].n; });
    order[no++] = big;
    // every small group's one-launch cascade on a stream of its own, behind whatever `aux` still holds for them
    static const bool xaux_on = !(std::getenv("MIRGE_XAUX") && std::atoi(std::getenv("MIRGE_XAUX")) == 0);
    int n_small = 0;
    for (int k = 0; k < MIRGE_NGROUPS; k++)
        if (order[k] != skip && order[k] != big && R->g[order[k]].n && R->g[order[k]].n <= fused_max) n_small++;
    const bool spread = xaux_on && n_small > 1;
    if (spread && !c->xaux_forked) CHECK(xaux_fork(c));
    // (the same assignment as small_group_slots, native_collapse.hpp; with `xaux_forked` the caller's own: small_slot)
    int slot = 0;
    for (int k = 0; k < MIRGE_NGROUPS && rc == 0; k++) {
        const int gi = order[k];
        if (gi == skip) continue;
        c->cur = gi == big ? c->stream : c->aux;
        if (is_long_group(gi)) {
            rc = cascade_group_long(c, R->g[gi], res->g[gi], dsteps, rt, group_tag(gi));
            continue;
        }
        if (gi != big && R->g[gi].n <= fused_max) {
            if (c->xaux_forked) { if (c->small_slot[gi] >= 0) c->cur = c->xaux[c->small_slot[gi]]; }  // behind its own scatter kernel
            else if (spread && R->g[gi].n) c->cur = c->xaux[xaux_slot_of(slot++)];
            MIRGE_BY_WIDTH(gi, rc, cascade_group_fused<W>(c, R->g[gi], res->g[gi], dsteps, rt, group_tag(gi)));
            continue;
        }
        MIRGE_BY_WIDTH(gi, rc, cascade_group<W>(c, R->g[gi], res->g[gi], steps, pol, rt, group_tag(gi), nullptr, 0, c->casc_dwalks[W == 1 ? 0 : 1]));
    }
    // (round 6, the mirge_collapse_cascade route) extra stream 0 collects the other extra streams now, behind its own last cascade: the
    // join then has ONE stream to wait for (stream_join)
    c->x0_gathered = false;
    if (rc == 0 && c->xaux_forked && c->xaux_used) {
        hipError_t e = hipSuccess;
        for (int k = 1; k < xaux_slots() && e == hipSuccess; k++) {  // (the streams small_slot names)
            e = hipEventRecord(c->ev_xjoin[k], c->xaux[k]);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->xaux[0], c->ev_xjoin[k], 0);
        }
        c->x0_gathered = e == hipSuccess;
    }
    // no join here: the next entry point that needs one makes it (join_pending_now); mirge_count_join puts the bulk
    // group's part of its work in front of it
    c->cur = c->stream;
    if (rc == 0) c->join_pending = true;
    else { int jr = stream_join(c); (void)jr; }
    return rc;
}

// read lengths present: the host histogram pack / parse / collapse keep; without one, every length the width groups in use can hold
static void reads_lengths_present(const mirge_reads* R, int32_t* hist) {
    if (R->hist_valid) { std::memcpy(hist, R->len_hist, sizeof(int32_t) * (MIRGE_MAX_READ_LEN + 1)); return; }
    std::memset(hist, 0, sizeof(int32_t) * (MIRGE_MAX_READ_LEN + 1));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = R->g[gi];
        if (!g.n || is_long_group(gi)) continue;  // (the long class: mirge_reads::long_max)
        const int w = kGroupW[gi];
        int lo = w == 1 ? 1 : (w == 2 ? 32 : (w == 4 ? 65 : 129)), hi = w == 1 ? 31 : (w == 2 ? 64 : (w == 4 ? 128 : MIRGE_MAX_READ_LEN));
        for (int L = lo; L <= hi; L++) hist[L] = 1;
    }
}

extern "C" int mirge_cascade_run(mirge_ctx* c, const mirge_reads* R, const mirge_lib* const* libs,
                                 const mirge_policy* pol, int32_t n_pass, mirge_result** out) {
    HostClock hc("cascade");
    if (!c || !R || !libs || !pol || !out || n_pass < 1 || n_pass > MIRGE_MAX_PASSES)
        return fail(-1, "mirge_cascade_run: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    int32_t hist[MIRGE_MAX_READ_LEN + 1];
    reads_lengths_present(R, hist);
    CHECK(cascade_config(c, libs, pol, n_pass, hist, R->long_max > 0));
    hc.lap("plans+fused");
    auto res = std::make_unique<mirge_result>();
    res->ctx = c; res->n = R->n; res->n_pass = n_pass; res->reads = R;
    for (int32_t p = 0; p < n_pass; p++) res->n_refs[p] = libs[p] ? (uint32_t)libs[p]->n_refs : 0u;
    const int rc = cascade_launch_groups(c, R, res.get(), pol, -1);
    hc.lap("enqueue+join");
    if (rc) { mirge_result_destroy(res.release()); return rc; }
    *out = res.release();
    return 0;
}

// What a cascade over `reads` needs besides the reads -- merged libraries, probe tables for the read lengths present, plan
// tables, the device step list -- built now and kept in the ctx (mirge_cascade_run / mirge_collapse_cascade find it there).
// Returns when the tables exist: callers that want to know what a process's first sample spends on them (bench.py's
// cli_path) time this call; nobody has to make it.
extern "C" int mirge_cascade_prepare(mirge_ctx* c, const mirge_reads* R, const mirge_lib* const* libs, const mirge_policy* pol,
                                     int32_t n_pass) {
    if (!c || !R || !libs || !pol || n_pass < 1 || n_pass > MIRGE_MAX_PASSES) return fail(-1, "mirge_cascade_prepare: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    int32_t hist[MIRGE_MAX_READ_LEN + 1];
    reads_lengths_present(R, hist);
    CHECK(cascade_config(c, libs, pol, n_pass, hist, R->long_max > 0));
    HIPOK(hipStreamSynchronize(c->stream));
    return 0;
}

// ticks_out[b] = constant-rate clock ticks workgroup b of the LAST profiled k_cascade_bulk launch (one-word bulk group) took,
// *khz_out = that clock's rate; returns the grid in *grid_out (0: no such launch yet).  Profiling must have been on.
extern "C" int mirge_cascade_wg_times(mirge_ctx* c, uint32_t* ticks_out, int32_t cap, int32_t* grid_out, int32_t* khz_out) {
    if (!c || !grid_out) return fail(-1, "mirge_cascade_wg_times: bad argument");
    CHECK(mirge_ctx_sync(c));
    *grid_out = (int32_t)c->wg_grid;
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device);
    if (khz_out) *khz_out = khz;
    for (uint32_t b = 0; ticks_out && b < c->wg_grid && (int32_t)b < cap; b++) ticks_out[b] = c->wg_pinned[c->wg_grid + b] - c->wg_pinned[b];
    return 0;
}

extern "C" int mirge_cascade_walks(mirge_ctx* c, int32_t* walks) {
    if (!c || !walks) return fail(-1, "mirge_cascade_walks: bad argument");
    walks[0] = walks[1] = walks[2] = 0;
    for (auto& e : c->walks) {
        if (e.dev != c->casc_dwalks[0]) continue;
        walks[0] = e.host->n;
        for (int i = 0; i < e.host->n; i++) {
            const BulkWalk& w = e.host->w[i];
            walks[1] += (w.pre.slots != nullptr) + (w.post.slots != nullptr);
            walks[2] += (w.pre.slots != nullptr) + (w.post.slots != nullptr) + (w.has_main != 0);
        }
    }
    return 0;
}

// Collapse and cascade of one sample as ONE call: the bulk read group's passes are queued on the GPU right behind
// the collapse kernels, BEFORE the host has read the unique counts back (the kernels take the count from device
// memory: k_pass / k_resolve `n_dev`), so the GPU does not idle while the host wakes up, finishes the small groups'
// collapse and enqueues the cascade (~0.1 ms of a 2 ms step).  Same results as mirge_collapse followed by
// mirge_cascade_run, which is also what runs when the bulk group is not on the partitioned key path or its
// partition overflowed (MIRGE_NO_PRESYNC=1 forces that, for A/B).
extern "C" int mirge_collapse_cascade(mirge_ctx* c, const mirge_reads* raw, const mirge_lib* const* libs, const mirge_policy* pol,
                                      int32_t n_pass, mirge_reads** uniq, int64_t* n_uniq, mirge_result** out) {
    if (!c || !raw || !libs || !pol || !uniq || !out || n_pass < 1 || n_pass > MIRGE_MAX_PASSES)
        return fail(-1, "mirge_collapse_cascade: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    static const bool presync_off = std::getenv("MIRGE_NO_PRESYNC") != nullptr;
    mirge_reads* U = nullptr;
    if (presync_off || !raw->hist_valid || raw->n == 0) {
        CHECK(mirge_collapse(c, raw, nullptr, 1, &U, n_uniq));
        const int rc = mirge_cascade_run(c, U, libs, pol, n_pass, out);
        if (rc) { mirge_reads_destroy(U); return rc; }
        *uniq = U;
        return 0;
    }
    // the unique reads have the raw reads' lengths: the cascade can be configured before they exist
    CHECK(cascade_config(c, libs, pol, n_pass, raw->len_hist, raw->long_max > 0));
    auto res = std::make_unique<mirge_result>();
    res->ctx = c; res->n_pass = n_pass;
    for (int32_t p = 0; p < n_pass; p++) res->n_refs[p] = libs[p] ? (uint32_t)libs[p]->n_refs : 0u;
    const size_t prof_mark = c->prof_pending.size();
    int hooked_group = -1;
    CollapseHook hook;
    hook.pre_sync = [&](mirge_reads* partial, const CollapseTmp*, uint32_t* dmeta, int big) -> int {
        ReadGroup rg = partial->g[big];  // the unique reads' arrays, allocated for the raw count
        rg.n = raw->g[big].n;
        c->cur = c->stream;
        const int rc = cascade_group<1>(c, rg, res->g[big], c->casc_steps, pol, c->casc_rt, group_tag(big), dmeta + big, raw->g[big].n, c->casc_dwalks[0]);
        if (rc == 0) { hooked_group = big; c->overlap_mode = true; }
        return rc;
    };
    hook.small_ready = [&](mirge_reads* partial, int big) -> int {
        // the small groups' cascades, queued while the bulk group's collapse still runs (collapse_impl, round 5)
        return cascade_launch_groups(c, partial, res.get(), pol, big, big);
    };
    hook.discard = [&]() {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->aux);
        for (int k = 0; k < MIRGE_N_XAUX; k++) (void)hipStreamSynchronize(c->xaux[k]);
        c->overlap_mode = false;
        c->join_pending = false;  // (everything has drained: nothing is left to join)
        c->xaux_used = false;
        c->x0_gathered = c->aux_drained = false;
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
            if (gi != hooked_group && !res->g[gi].pass) continue;
            ResGroup& g = res->g[gi];
            c->release(g.pass); c->release(g.pos); c->release(g.mm); c->release(g.ref); c->release(g.off);
            g = ResGroup();
        }
        hooked_group = -1;
        c->prof_pending.resize(std::min(c->prof_pending.size(), prof_mark));
        c->flush_deferred();
    };
    int rc = collapse_impl(c, raw, nullptr, 1, &U, n_uniq, &hook);
    if (rc) { c->overlap_mode = false; mirge_result_destroy(res.release()); return rc; }
    if (!hook.ran) {  // general path or overflow: the ordinary sequence
        res.reset();
        rc = mirge_cascade_run(c, U, libs, pol, n_pass, out);
        if (rc) { mirge_reads_destroy(U); return rc; }
        *uniq = U;
        return 0;
    }
    res->n = U->n; res->reads = U; res->dmeta = hook.dmeta;
    res->g[hooked_group].n = U->g[hooked_group].n;
    for (size_t i = prof_mark; i < c->prof_pending.size(); i++) c->prof_pending[i].n_first = (double)U->g[hooked_group].n;
    if (!hook.small_ran) rc = cascade_launch_groups(c, U, res.get(), pol, hooked_group);  // small groups; the join is left pending
    c->overlap_mode = false;
    if (rc) { (void)hipStreamSynchronize(c->stream); mirge_result_destroy(res.release()); mirge_reads_destroy(U); return rc; }
    *uniq = U;
    *out = res.release();
    return 0;
}

template <typename T>
static int fetch_field(mirge_ctx* c, const mirge_result* res, T* host_out, T* ResGroup::*field) {
    if (!host_out || !res->n) return 0;
    T* dfull = nullptr;
    CHECK(dalloc(c, &dfull, (size_t)res->n));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ResGroup& g = res->g[gi];
        const ReadGroup& rg = res->reads->g[gi];
        if (!g.n) continue;
        hipLaunchKernelGGL(k_scatter_out<T>, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream,
                           (const T*)(g.*field), g.n, rg.base, (const uint32_t*)rg.orig, dfull);
    }
    HIPOK(hipMemcpyAsync(host_out, dfull, (size_t)res->n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    c->release(dfull);
    return 0;
}

extern "C" int mirge_result_fetch(mirge_ctx* c, const mirge_result* res, int8_t* pass_out, int32_t* ref_out,
                                  int32_t* off_out, int8_t* mm_out) {
    if (!c || !res) return fail(-1, "mirge_result_fetch: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    CHECK(fetch_field<int8_t>(c, res, pass_out, &ResGroup::pass));
    CHECK(fetch_field<int32_t>(c, res, ref_out, &ResGroup::ref));
    CHECK(fetch_field<int32_t>(c, res, off_out, &ResGroup::off));
    CHECK(fetch_field<int8_t>(c, res, mm_out, &ResGroup::mm));
    return 0;
}
