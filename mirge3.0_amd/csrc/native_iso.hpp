// native_iso.hpp -- part of mirge_native.hip (one translation unit): isomiR typing on the device (k_isotype) and the
// miRTop GFF3 written from its records (row N2; create_gff, mirge/libs/summary.py:48-606).
#pragma once

static_assert(sizeof(MirgeIsoRec) == MIRGE_ISO_RECORD_BYTES, "isomiR record layout (include/mirge_native.h)");

// One typed record per row: rows are the reads of the two miRNA classes in the order the caller wants them printed;
// slot_of_read[read (handle order)] = row or -1.  Tables: see IsoTables.
extern "C" int mirge_isomir_type(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, int32_t exact_pass, int32_t iso_pass,
                                 const int32_t* master_of_ref, int64_t n_mirna, const char* master_ascii, const int32_t* master_off,
                                 const int32_t* pre_of_master, const int32_t* start0, int64_t n_master, const char* pre_ascii,
                                 const int32_t* pre_off, int64_t n_pre, const int32_t* slot_of_read, int64_t n_rows,
                                 void* records_out) {
    if (!c || !U || !res || !master_of_ref || !master_off || !pre_of_master || !start0 || !pre_off || !slot_of_read || n_mirna < 0 ||
        n_master < 0 || n_pre < 0 || n_rows < 0 || (n_rows > 0 && !records_out))
        return fail(-1, "mirge_isomir_type: bad argument");
    if (res->n != U->n) return fail(-1, "result and read set differ in size");
    for (int32_t p : {exact_pass, iso_pass})
        if (p >= 0 && p < res->n_pass && (int64_t)res->n_refs[p] > n_mirna)
            return fail(-1, "mirge_isomir_type: master_of_ref is shorter than the miRNA library of pass " + std::to_string(p));
    for (int64_t r = 0; r < n_mirna; r++)
        if (master_of_ref[r] >= n_master) return fail(-1, "master_of_ref out of range");
    for (int64_t m = 0; m < n_master; m++)
        if (pre_of_master[m] < 0 || pre_of_master[m] >= n_pre) return fail(-1, "pre_of_master out of range");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    const size_t nm = (size_t)std::max<int64_t>(n_master, 1), np = (size_t)std::max<int64_t>(n_pre, 1);
    const size_t mbytes = (size_t)std::max<int32_t>(master_off[n_master], 1), pbytes = (size_t)std::max<int32_t>(pre_off[n_pre], 1);
    int32_t *d_mof = nullptr, *d_moff = nullptr, *d_pom = nullptr, *d_s0 = nullptr, *d_poff = nullptr, *d_slot = nullptr;
    char *d_m = nullptr, *d_p = nullptr;
    MirgeIsoRec* d_out = nullptr;
    CHECK(dalloc(c, &d_mof, (size_t)std::max<int64_t>(n_mirna, 1)));
    CHECK(dalloc(c, &d_moff, nm + 1));
    CHECK(dalloc(c, &d_pom, nm));
    CHECK(dalloc(c, &d_s0, nm));
    CHECK(dalloc(c, &d_poff, np + 1));
    CHECK(dalloc(c, &d_slot, (size_t)std::max<int64_t>(U->n, 1)));
    CHECK(dalloc(c, &d_m, mbytes));
    CHECK(dalloc(c, &d_p, pbytes));
    CHECK(dalloc(c, &d_out, (size_t)std::max<int64_t>(n_rows, 1)));
    if (n_mirna) HIPOK(hipMemcpyAsync(d_mof, master_of_ref, (size_t)n_mirna * 4, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(d_moff, master_off, ((size_t)n_master + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (n_master) {
        HIPOK(hipMemcpyAsync(d_pom, pre_of_master, (size_t)n_master * 4, hipMemcpyHostToDevice, c->stream));
        HIPOK(hipMemcpyAsync(d_s0, start0, (size_t)n_master * 4, hipMemcpyHostToDevice, c->stream));
        if (master_off[n_master]) HIPOK(hipMemcpyAsync(d_m, master_ascii, (size_t)master_off[n_master], hipMemcpyHostToDevice, c->stream));
    }
    HIPOK(hipMemcpyAsync(d_poff, pre_off, ((size_t)n_pre + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (n_pre && pre_off[n_pre]) HIPOK(hipMemcpyAsync(d_p, pre_ascii, (size_t)pre_off[n_pre], hipMemcpyHostToDevice, c->stream));
    if (U->n) HIPOK(hipMemcpyAsync(d_slot, slot_of_read, (size_t)U->n * 4, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemsetAsync(d_out, 0, (size_t)std::max<int64_t>(n_rows, 1) * sizeof(MirgeIsoRec), c->stream));
    IsoTables tb;
    tb.master_of_ref = d_mof; tb.master = d_m; tb.master_off = d_moff; tb.pre_of_master = d_pom; tb.start0 = d_s0;
    tb.pre = d_p; tb.pre_off = d_poff;
    // MIRGE_ISO_FAST=0: every read through the array form of the typing (the tests' second implementation on the device, A/B)
    static const int32_t iso_fast = !(std::getenv("MIRGE_ISO_FAST") && std::atoi(std::getenv("MIRGE_ISO_FAST")) == 0);
    std::vector<uint32_t*> lists;  // the groups' row lists, released behind the synchronisation below
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ResGroup& g = res->g[gi];
        const ReadGroup& rg = U->g[gi];
        if (!g.n || kGroupW[gi] > 2 || is_long_group(gi)) continue;  // a read annotated to a miRNA is at most 3 nt longer than it
        // one 64-thread workgroup per chunk of reads, in both kernels: chunk b's miRNA rows are list[b * chunk ...]
        const uint32_t tgrid = (uint32_t)grid_for(c, g.n, 64);
        uint32_t chunk = ((uint32_t)g.n + tgrid - 1) / tgrid;
        chunk = (chunk + 63) / 64 * 64;
        uint32_t* dlist = nullptr;
        CHECK(dalloc(c, &dlist, (size_t)tgrid * chunk + tgrid));
        uint32_t* dnlist = dlist + (size_t)tgrid * chunk;
        lists.push_back(dlist);
        {
            LaunchScope ls(c, "k_member_list", g.n);
            const IsoMember pred{g.pass, exact_pass, iso_pass, (const uint32_t*)rg.orig, rg.base, d_slot};
            hipLaunchKernelGGL((k_member_list<IsoMember, 64>), dim3(tgrid), dim3(64), 0, c->stream, (uint32_t)g.n, chunk, pred, dlist, dnlist);
        }
        LaunchScope ls(c, "k_isotype", g.n);
        if (kGroupW[gi] == 1)
            hipLaunchKernelGGL(k_isotype<1>, dim3(tgrid), dim3(64), 0, c->stream, view_of<1>(rg), rg.base, (const uint32_t*)rg.orig,
                               g.ref, tb, d_slot, d_out, (const uint32_t*)dlist, (const uint32_t*)dnlist, chunk, iso_fast);
        else
            hipLaunchKernelGGL(k_isotype<2>, dim3(tgrid), dim3(64), 0, c->stream, view_of<2>(rg), rg.base, (const uint32_t*)rg.orig,
                               g.ref, tb, d_slot, d_out, (const uint32_t*)dlist, (const uint32_t*)dnlist, chunk, iso_fast);
    }
    if (n_rows) HIPOK(hipMemcpyAsync(records_out, d_out, (size_t)n_rows * sizeof(MirgeIsoRec), hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    HIPOK(hipGetLastError());
    c->drain();
    c->release(d_mof); c->release(d_moff); c->release(d_pom); c->release(d_s0); c->release(d_poff); c->release(d_slot);
    c->release(d_m); c->release(d_p); c->release(d_out);
    for (uint32_t* l : lists) c->release(l);
    return 0;
}


// (mirge_gff_write and the UID rule: native_host.hpp)
