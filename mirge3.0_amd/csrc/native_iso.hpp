// native_iso.hpp -- part of mirge_native.hip (one translation unit): isomiR typing on the device (k_isotype) and the
// miRTop GFF3 written from its records (row N2; create_gff, mirge/libs/summary.py:48-606).
#pragma once

static_assert(sizeof(MirgeIsoRec) == MIRGE_ISO_RECORD_BYTES, "isomiR record layout (include/mirge_native.h)");

struct IsoDevice {
    int32_t *d_mof = nullptr, *d_moff = nullptr, *d_pom = nullptr, *d_s0 = nullptr, *d_poff = nullptr, *d_slot = nullptr;
    char *d_m = nullptr, *d_p = nullptr;
    MirgeIsoRec* d_out = nullptr;
    IsoTables tb;
    std::vector<uint32_t*> lists;  // the groups' row lists, released behind a synchronisation
};
static int iso_device_tables(mirge_ctx* c, IsoDevice& dv, const mirge_reads* U, const int32_t* master_of_ref, int64_t n_mirna, const char* master_ascii,
                             const int32_t* master_off, const int32_t* pre_of_master, const int32_t* start0, int64_t n_master, const char* pre_ascii,
                             const int32_t* pre_off, int64_t n_pre, int64_t n_rows);
static int iso_device_run(mirge_ctx* c, IsoDevice& dv, const mirge_reads* U, const mirge_result* res, int32_t exact_pass, int32_t iso_pass);
static void iso_device_release(mirge_ctx* c, IsoDevice& dv);

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
    IsoDevice dv;
    int rc = iso_device_tables(c, dv, U, master_of_ref, n_mirna, master_ascii, master_off, pre_of_master, start0, n_master, pre_ascii, pre_off, n_pre, n_rows);
    if (rc == 0 && U->n) {
        const hipError_t e = hipMemcpyAsync(dv.d_slot, slot_of_read, (size_t)U->n * 4, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) rc = fail(-2, std::string("mirge_isomir_type: ") + hipGetErrorString(e));
    }
    if (rc == 0) rc = iso_device_run(c, dv, U, res, exact_pass, iso_pass);
    if (rc == 0 && n_rows) {
        const hipError_t e = hipMemcpyAsync(records_out, dv.d_out, (size_t)n_rows * sizeof(MirgeIsoRec), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(-2, std::string("mirge_isomir_type: ") + hipGetErrorString(e));
    }
    {
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (rc == 0 && e != hipSuccess) rc = fail(-2, std::string("mirge_isomir_type: ") + hipGetErrorString(e));
        const hipError_t e2 = hipGetLastError();
        if (rc == 0 && e2 != hipSuccess) rc = fail(-2, std::string("mirge_isomir_type: ") + hipGetErrorString(e2));
    }
    c->drain();
    iso_device_release(c, dv);
    return rc;
}

// the typing on the device, shared by mirge_isomir_type (records to the host) and mirge_gff_write_device (records stay): the name
// tables uploaded, the slot map and the record array allocated (iso_device_tables), k_member_list + k_isotype per read group
// (iso_device_run; d_slot must hold slot_of_read by then), everything handed back (iso_device_release, behind a synchronisation)
static int iso_device_tables(mirge_ctx* c, IsoDevice& dv, const mirge_reads* U, const int32_t* master_of_ref, int64_t n_mirna, const char* master_ascii,
                             const int32_t* master_off, const int32_t* pre_of_master, const int32_t* start0, int64_t n_master, const char* pre_ascii,
                             const int32_t* pre_off, int64_t n_pre, int64_t n_rows) {
    int32_t*& d_mof = dv.d_mof; int32_t*& d_moff = dv.d_moff; int32_t*& d_pom = dv.d_pom; int32_t*& d_s0 = dv.d_s0; int32_t*& d_poff = dv.d_poff;
    int32_t*& d_slot = dv.d_slot; char*& d_m = dv.d_m; char*& d_p = dv.d_p; MirgeIsoRec*& d_out = dv.d_out;
    const size_t nm = (size_t)std::max<int64_t>(n_master, 1), np = (size_t)std::max<int64_t>(n_pre, 1);
    const size_t mbytes = (size_t)std::max<int32_t>(master_off[n_master], 1), pbytes = (size_t)std::max<int32_t>(pre_off[n_pre], 1);
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
    HIPOK(hipMemsetAsync(d_out, 0, (size_t)std::max<int64_t>(n_rows, 1) * sizeof(MirgeIsoRec), c->stream));
    dv.tb.master_of_ref = d_mof; dv.tb.master = d_m; dv.tb.master_off = d_moff; dv.tb.pre_of_master = d_pom; dv.tb.start0 = d_s0;
    dv.tb.pre = d_p; dv.tb.pre_off = d_poff;
    return 0;
}

static int iso_device_run(mirge_ctx* c, IsoDevice& dv, const mirge_reads* U, const mirge_result* res, int32_t exact_pass, int32_t iso_pass) {
    const IsoTables& tb = dv.tb;
    int32_t* const d_slot = dv.d_slot;
    MirgeIsoRec* const d_out = dv.d_out;
    std::vector<uint32_t*>& lists = dv.lists;
    // MIRGE_ISO_FAST=0: every read through the array form of the typing (the tests' second implementation on the device, A/B)
    static const int32_t iso_fast = !(std::getenv("MIRGE_ISO_FAST") && std::atoi(std::getenv("MIRGE_ISO_FAST")) == 0);
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
    return 0;
}

static void iso_device_release(mirge_ctx* c, IsoDevice& dv) {
    c->release(dv.d_mof); c->release(dv.d_moff); c->release(dv.d_pom); c->release(dv.d_s0); c->release(dv.d_poff); c->release(dv.d_slot);
    c->release(dv.d_m); c->release(dv.d_p); c->release(dv.d_out);
    for (uint32_t* l : dv.lists) c->release(l);
    dv = IsoDevice();
}


// (mirge_gff_write and the UID rule: native_host.hpp)

// sample_miRge3.gff from the device-resident run (round 6): rows chosen (k_gff_select / k_gff_rows: the exact-miRNA rows of the mapped
// frame in frame order, then its isomiR rows -- summary.py:50-60), typed (k_isotype), measured and formatted (k_gff_line) on the
// device; the file's text crosses PCIe once and is written with positional writes.  `order[k]` = handle index of the read in row k of
// the run's frame (mirge_collapse_order / _order_sorted).  name_of_ref / parent_of_ref [n_mirna]: the printed name and the precursor
// of every miRNA reference (-1: the reference drops reads of that name), as indexes into the two string tables.  *n_lines_out = lines
// written below the head.  Replaces create_gff's per-read loop and its file writes (summary.py:204-470).
extern "C" int mirge_gff_write_device(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, int32_t exact_pass, int32_t iso_pass,
                                      const int32_t* master_of_ref, int64_t n_mirna, const char* master_ascii, const int32_t* master_off,
                                      const int32_t* pre_of_master, const int32_t* start0, int64_t n_master, const char* pre_ascii,
                                      const int32_t* pre_off, int64_t n_pre, const int32_t* name_of_ref, const char* name_data,
                                      const int64_t* name_off, int64_t n_names, const int32_t* parent_of_ref, const char* parent_data,
                                      const int64_t* parent_off, int64_t n_parents, const int64_t* order, const char* path, const char* head,
                                      const char* source, int64_t* n_lines_out) {
    if (!c || !U || !res || !master_of_ref || !master_off || !pre_of_master || !start0 || !pre_off || n_mirna < 0 || n_master < 0 || n_pre < 0 ||
        !name_of_ref || !name_off || !parent_of_ref || !parent_off || n_names < 0 || n_parents < 0 || (U->n && !order) || !path || !head || !source ||
        U->n_samples < 1 || res->n != U->n || U->n >= 0x7FFFFFF0ll)
        return fail(-1, "mirge_gff_write_device: bad argument");
    for (int32_t p : {exact_pass, iso_pass})
        if (p >= 0 && p < res->n_pass && (int64_t)res->n_refs[p] > n_mirna)
            return fail(-1, "mirge_gff_write_device: the name tables are shorter than the miRNA library of pass " + std::to_string(p));
    for (int64_t r = 0; r < n_mirna; r++)
        if (master_of_ref[r] >= n_master) return fail(-1, "master_of_ref out of range");
    for (int64_t m = 0; m < n_master; m++)
        if (pre_of_master[m] < 0 || pre_of_master[m] >= n_pre) return fail(-1, "pre_of_master out of range");
    if (name_off[n_names] >= 0xFFFFFFF0ll || parent_off[n_parents] >= 0xFFFFFFF0ll) return fail(-5, "mirge_gff_write_device: name table too large");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    HostClock hc("gff_write_device");
    const size_t n = (size_t)U->n;
    const size_t hl = std::strlen(head), sl = std::strlen(source);
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
        if (U->g[gi].n && U->g[gi].orig) return fail(-1, "mirge_gff_write_device: the read set is not a collapse result");
    CsvTables t;
    csv_tables_of(U, res, t);
    uint32_t *d_order = nullptr, *fe = nullptr, *fi = nullptr, *pe = nullptr, *pi = nullptr, *d_rows = nullptr, *d_noff = nullptr, *d_poff2 = nullptr,
             *d_flags = nullptr;
    int32_t *d_nof = nullptr, *d_pof = nullptr;
    uint8_t *d_nd = nullptr, *d_pd = nullptr, *d_src = nullptr, *d_text = nullptr;
    unsigned long long *d_len = nullptr, *d_off = nullptr;
    void* tmp = nullptr;
    IsoDevice dv;
    int rc = 0;
    int64_t n_lines = 0;
    size_t bytes = 0;
    do {
        std::vector<uint32_t> o32(std::max<size_t>(n, 1)), noff32((size_t)n_names + 1), poff32((size_t)n_parents + 1);
        for (size_t k = 0; k < n; k++) {
            if (order[k] < 0 || order[k] >= U->n) { rc = fail(-1, "mirge_gff_write_device: row index out of range"); break; }
            o32[k] = (uint32_t)order[k];
        }
        if (rc) break;
        for (int64_t k = 0; k <= n_names; k++) noff32[(size_t)k] = (uint32_t)(name_off[k] - name_off[0]);
        for (int64_t k = 0; k <= n_parents; k++) poff32[(size_t)k] = (uint32_t)(parent_off[k] - parent_off[0]);
        const size_t nm1 = (size_t)std::max<int64_t>(n_mirna, 1);
        if ((rc = dalloc(c, &d_order, std::max<size_t>(n, 1)))) break;
        if ((rc = dalloc(c, &fe, n + 1))) break;
        if ((rc = dalloc(c, &fi, n + 1))) break;
        if ((rc = dalloc(c, &pe, n + 1))) break;
        if ((rc = dalloc(c, &pi, n + 1))) break;
        if ((rc = dalloc(c, &d_rows, std::max<size_t>(n, 1)))) break;
        if ((rc = dalloc(c, &d_nof, nm1))) break;
        if ((rc = dalloc(c, &d_pof, nm1))) break;
        if ((rc = dalloc(c, &d_noff, noff32.size()))) break;
        if ((rc = dalloc(c, &d_poff2, poff32.size()))) break;
        if ((rc = dalloc(c, &d_nd, (size_t)noff32.back() + 16))) break;
        if ((rc = dalloc(c, &d_pd, (size_t)poff32.back() + 16))) break;
        if ((rc = dalloc(c, &d_src, sl + 16))) break;
        if ((rc = dalloc(c, &d_flags, 16))) break;
        hipError_t e = hipSuccess;
        if (n) e = hipMemcpyAsync(d_order, o32.data(), n * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && n_mirna) e = hipMemcpyAsync(d_nof, name_of_ref, (size_t)n_mirna * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && n_mirna) e = hipMemcpyAsync(d_pof, parent_of_ref, (size_t)n_mirna * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_noff, noff32.data(), noff32.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_poff2, poff32.data(), poff32.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && noff32.back()) e = hipMemcpyAsync(d_nd, name_data + name_off[0], noff32.back(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && poff32.back()) e = hipMemcpyAsync(d_pd, parent_data + parent_off[0], poff32.back(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && sl) e = hipMemcpyAsync(d_src, source, sl, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, 64, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(fe + n, 0, 4, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(fi + n, 0, 4, c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_gff_write_device: ") + hipGetErrorString(e)); break; }
        // ---- the file's rows: exact-miRNA rows of the frame, then its isomiR rows
        if (n) hipLaunchKernelGGL(k_gff_select, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, t, (const uint32_t*)d_order, (uint32_t)n, exact_pass,
                                  iso_pass, fe, fi);
        size_t tb = 0;
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, fe, pe, (int)(n + 1), c->stream);
        if (e == hipSuccess && (rc = dalloc(c, (uint8_t**)&tmp, std::max<size_t>(tb, 16)))) break;
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, fe, pe, (int)(n + 1), c->stream);
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, fi, pi, (int)(n + 1), c->stream);
        uint32_t tot[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpyAsync(&tot[0], pe + n, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&tot[1], pi + n, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_gff_write_device: ") + hipGetErrorString(e)); break; }
        const size_t n_rows = (size_t)tot[0] + tot[1];
        hc.lap("rows chosen");
        // ---- typing: the records stay on the device
        if ((rc = iso_device_tables(c, dv, U, master_of_ref, n_mirna, master_ascii, master_off, pre_of_master, start0, n_master, pre_ascii, pre_off,
                                    n_pre, (int64_t)n_rows))) break;
        if (n) e = hipMemsetAsync(dv.d_slot, 0xFF, n * 4, c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_gff_write_device: ") + hipGetErrorString(e)); break; }
        if (n) hipLaunchKernelGGL(k_gff_rows, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, (const uint32_t*)d_order, (uint32_t)n, (const uint32_t*)fe,
                                  (const uint32_t*)fi, (const uint32_t*)pe, (const uint32_t*)pi, tot[0], d_rows, dv.d_slot);
        if ((rc = iso_device_run(c, dv, U, res, exact_pass, iso_pass))) break;
        // ---- the lines: measured, placed, written
        GffTables gt;
        gt.name_of_ref = d_nof; gt.parent_of_ref = d_pof; gt.name_data = d_nd; gt.name_off = d_noff; gt.parent_data = d_pd; gt.parent_off = d_poff2;
        gt.n_names = (uint32_t)n_names; gt.n_parents = (uint32_t)n_parents; gt.n_mirna = (uint32_t)n_mirna; gt.source = d_src; gt.source_len = (uint32_t)sl;
        if ((rc = dalloc(c, &d_len, n_rows + 1))) break;
        if ((rc = dalloc(c, &d_off, n_rows + 1))) break;
        e = hipMemsetAsync(d_len + n_rows, 0, 8, c->stream);
        if (e == hipSuccess && n_rows) {
            LaunchScope ls(c, "k_gff_line.len", (double)n_rows);
            hipLaunchKernelGGL(k_gff_line<false>, dim3(grid_for(c, n_rows)), dim3(MIRGE_BLOCK), 0, c->stream, t, gt, (const uint32_t*)d_rows, (uint32_t)n_rows,
                               (const MirgeIsoRec*)dv.d_out, d_len, (const unsigned long long*)nullptr, (uint8_t*)nullptr, d_flags);
        }
        size_t tb2 = 0;
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, d_len, d_off, (int)(n_rows + 1), c->stream);
        if (e == hipSuccess && tb2 > tb) { c->release(tmp); tmp = nullptr; if ((rc = dalloc(c, (uint8_t**)&tmp, tb2))) break; }
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, tb2, d_len, d_off, (int)(n_rows + 1), c->stream);
        unsigned long long total = 0;
        uint32_t hflag = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&total, d_off + n_rows, 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&hflag, d_flags, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_gff_write_device: ") + hipGetErrorString(e)); break; }
        if (hflag) { rc = fail(-1, "mirge_gff_write_device: name index or record out of range"); break; }
        bytes = (size_t)total;
        if (bytes && (rc = dalloc(c, &d_text, bytes))) break;
        if (n_rows && bytes) {
            LaunchScope ls(c, "k_gff_line.text", (double)n_rows);
            hipLaunchKernelGGL(k_gff_line<true>, dim3(grid_for(c, n_rows)), dim3(MIRGE_BLOCK), 0, c->stream, t, gt, (const uint32_t*)d_rows, (uint32_t)n_rows,
                               (const MirgeIsoRec*)dv.d_out, (unsigned long long*)nullptr, (const unsigned long long*)d_off, d_text, d_flags);
        }
        hc.lap("typed + formatted (device)");
        // lines written = rows with a line: counted from the lengths would need another pass; the host counts newlines only when asked
        if (bytes > c->csv_pinned_bytes) {  // page-locked staging shared with the per-read CSVs
            if (c->csv_pinned) (void)hipHostFree(c->csv_pinned);
            c->csv_pinned = nullptr; c->csv_pinned_bytes = 0;
            const size_t want = bytes + bytes / 8 + (1u << 20);
            if (hipHostMalloc((void**)&c->csv_pinned, want, hipHostMallocDefault) != hipSuccess) {
                rc = fail(-3, "mirge_gff_write_device: cannot page-lock " + std::to_string(want) + " bytes"); break;
            }
            c->csv_pinned_bytes = want;
        }
        const size_t CH = 8u << 20;
        struct Chunk { size_t at, n; hipEvent_t ev; };
        std::vector<Chunk> chunks;
        for (size_t at = 0; at < bytes && e == hipSuccess; at += CH) {
            const size_t nn = std::min(CH, bytes - at);
            e = hipMemcpyAsync(c->csv_pinned + at, d_text + at, nn, hipMemcpyDeviceToHost, c->stream);
            hipEvent_t ev = c->get_evt();
            if (e == hipSuccess) e = hipEventRecord(ev, c->stream);
            chunks.push_back(Chunk{at, nn, ev});
        }
        if (e != hipSuccess) rc = fail(-2, std::string("mirge_gff_write_device: ") + hipGetErrorString(e));
        int fd = -1;
        if (rc == 0) {
            fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0 || ::pwrite(fd, head, hl, 0) != (ssize_t)hl) rc = fail(-8, std::string("cannot write ") + path);
        }
        std::atomic<size_t> next{0};
        std::atomic<int> werr{0};
        std::atomic<long long> lines{0};
        auto writer = [&]() {
            (void)hipSetDevice(c->device);
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= chunks.size()) return;
                const Chunk& ck = chunks[k];
                if (hipEventSynchronize(ck.ev) != hipSuccess) { werr = 1; continue; }
                if (rc != 0 || fd < 0) continue;
                long long nl = 0;
                const uint8_t* p0 = c->csv_pinned + ck.at;
                for (size_t q = 0; q < ck.n; q++) nl += p0[q] == '\n';
                lines += nl;
                size_t done = 0;
                while (done < ck.n) {
                    const ssize_t w = ::pwrite(fd, p0 + done, ck.n - done, (off_t)(hl + ck.at + done));
                    if (w <= 0) { werr = 1; break; }
                    done += (size_t)w;
                }
            }
        };
        const int T = (int)std::max<size_t>(1, std::min<size_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), chunks.size()));
        std::vector<std::thread> wt;
        for (int k = 1; k < T; k++) wt.emplace_back(writer);
        writer();
        for (auto& x : wt) x.join();
        (void)hipStreamSynchronize(c->stream);
        for (auto& ck : chunks) c->evt_pool.push_back(ck.ev);
        if (fd >= 0 && ::close(fd) != 0) werr = 1;
        if (rc == 0 && werr) rc = fail(-8, "mirge_gff_write_device: write error");
        n_lines = (int64_t)lines.load();
        hc.lap("copy + write");
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    c->drain();
    iso_device_release(c, dv);
    c->release(d_order); c->release(fe); c->release(fi); c->release(pe); c->release(pi); c->release(d_rows); c->release(d_nof); c->release(d_pof);
    c->release(d_noff); c->release(d_poff2); c->release(d_nd); c->release(d_pd); c->release(d_src); c->release(d_flags); c->release(d_len);
    c->release(d_off); c->release(d_text); c->release(tmp);
    if (rc == 0 && n_lines_out) *n_lines_out = n_lines;
    return rc;
}
