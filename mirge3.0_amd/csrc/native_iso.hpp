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
                               g.ref, tb, d_slot, d_out, (const uint32_t*)dlist, (const uint32_t*)dnlist, chunk);
        else
            hipLaunchKernelGGL(k_isotype<2>, dim3(tgrid), dim3(64), 0, c->stream, view_of<2>(rg), rg.base, (const uint32_t*)rg.orig,
                               g.ref, tb, d_slot, d_out, (const uint32_t*)dlist, (const uint32_t*)dnlist, chunk);
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

namespace {
// miRgeEssential.UID (:364-370): prefix-length-, then two symbols of a 32-letter alphabet per 5-mer (value / 32,
// value % 32, A C G T = 0..3, first base most significant); a last chunk of k < 5 bases is numbered after all shorter
// k-mers (offsets 0, 4, 20, 84) and printed as one symbol below 32, two from there on
inline void uid_append(std::string& out, const char* s, size_t n) {
    static const char AL[] = "BD0EF1HI2JK3LM4NO5PQ6RS7UV8WX9YZ";
    static const int OFFS[5] = {0, 0, 4, 20, 84};
    for (size_t at = 0; at < n; at += 5) {
        const size_t k = std::min<size_t>(5, n - at);
        int v = 0;
        for (size_t t = 0; t < k; t++) v = v * 4 + (s[at + t] == 'A' ? 0 : s[at + t] == 'C' ? 1 : s[at + t] == 'G' ? 2 : 3);
        if (k == 5) { out.push_back(AL[v / 32]); out.push_back(AL[v % 32]); }
        else {
            v += OFFS[k];
            if (v < 32) out.push_back(AL[v]);
            else { out.push_back(AL[v / 32]); out.push_back(AL[v % 32]); }
        }
    }
}
}  // namespace

// The GFF3 body: one line per row with kind != 0, in row order (summary.py:204, :465).  name_of_row / parent_of_row
// index two string tables (the miRNA name as printed, its precursor's name); `head` = the four '#' lines.
extern "C" int mirge_gff_write(const char* path, const char* head, const char* source, const void* records, int64_t n_rows,
                               const char* read_ascii, const int64_t* read_off, const uint32_t* counts, int32_t S,
                               const int32_t* name_of_row, const char* name_data, const int64_t* name_off, int64_t n_names,
                               const int32_t* parent_of_row, const char* parent_data, const int64_t* parent_off, int64_t n_parents) {
    if (!path || !head || !source || n_rows < 0 || S < 1 || !read_off || !name_off || !parent_off ||
        (n_rows > 0 && (!records || !read_ascii || !counts || !name_of_row || !parent_of_row)))
        return fail(-1, "mirge_gff_write: bad argument");
    const MirgeIsoRec* rec = static_cast<const MirgeIsoRec*>(records);
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const int T = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n_rows / 8192));
    std::vector<std::string> buf((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    const std::string src(source);
    auto work = [&](int t) {
        std::string& out = buf[(size_t)t];
        for (int64_t k = n_rows * t / T; k < n_rows * (t + 1) / T; k++) {
            const MirgeIsoRec& r = rec[k];
            if (r.kind == 0) continue;
            const int32_t ni = name_of_row[k], pi = parent_of_row[k];
            if (ni < 0 || ni >= n_names || pi < 0 || pi >= n_parents || (size_t)r.vlen + r.clen > MIRGE_ISO_TEXT) { bad[(size_t)t] = 1; continue; }
            const char* nm = name_data + name_off[ni];
            const size_t nl = (size_t)(name_off[ni + 1] - name_off[ni]);
            const char* rd = read_ascii + read_off[k];
            const size_t rl = (size_t)(read_off[k + 1] - read_off[k]);
            out.append(nm, nl); out.push_back('\t'); out += src; out.push_back('\t');
            out += r.kind == 1 ? "ref_miRNA" : "isomiR";
            out.push_back('\t'); out += std::to_string(r.start); out.push_back('\t'); out += std::to_string(r.end);
            out += "\t.\t+\t.\tRead="; out.append(rd, rl); out += "; UID=";
            bool has_n = false;
            for (size_t q = 0; q < rl; q++) has_n |= rd[q] == 'N';
            if (has_n) out.push_back('.');
            else { out += r.kind == 1 ? "ref-" : "iso-"; out += std::to_string(rl); out.push_back('-'); uid_append(out, rd, rl); }
            out += "; Name="; out.append(nm, nl);
            out += "; Parent="; out.append(parent_data + parent_off[pi], (size_t)(parent_off[pi + 1] - parent_off[pi]));
            out += "; Variant="; out.append(r.text, r.vlen);
            out += "; Cigar="; out.append(r.text + r.vlen, r.clen);
            std::string ex;
            for (int s = 0; s < S; s++) { if (s) ex.push_back(','); csv_uint(ex, counts[(size_t)k * S + s]); }
            out += "; Expression="; out += ex; out += "; Filter=Pass; Hits="; out += ex; out.push_back('\n');
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < T; t++) if (bad[(size_t)t]) return fail(-1, "mirge_gff_write: name index or record out of range");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(-8, std::string("cannot write ") + path);
    bool ok = std::fputs(head, f) >= 0;
    for (int t = 0; t < T && ok; t++) ok = buf[(size_t)t].empty() || std::fwrite(buf[(size_t)t].data(), 1, buf[(size_t)t].size(), f) == buf[(size_t)t].size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) return fail(-8, std::string("write error on ") + path);
    return 0;
}
