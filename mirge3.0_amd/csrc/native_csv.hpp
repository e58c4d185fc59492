// native_csv.hpp -- part of mirge_native.hip (one translation unit): the per-read tables mapped.csv / unmapped.csv
// written from flat arrays on the host's cores.  Replaces the two DataFrame.to_csv calls of mirge/__main__.py:164-173
// (row a15): the reference builds a U-row pandas frame of Python strings to print it; here the rows are formatted
// straight from what the device returned (sequence bytes, pass, reference index, count matrix), same bytes out.
#pragma once

// (the host formatter mirge_annotation_csv and its helpers: native_host.hpp)

// The same two files from the DEVICE-resident run: rows are formatted by k_csv_rowlen / k_csv_rows from the packed unique
// reads, their count matrix and the cascade's annotation where they lie; what crosses PCIe is the files' text (page-locked
// staging kept in the ctx), written by several threads with positional writes while later chunks are still arriving.
// Reference names that need CSV quoting are refused (-4): the caller then formats on the host (mirge_annotation_csv).
// `mode` 0: the two files whole (created, header first).  1: only the bytes the rows would take (bytes_out[0] mapped, [1]
// unmapped; nothing is formatted or written).  2: the rows' text at file offsets file_off[0] / [1] of files that exist
// (no header, no truncation): ONE rank's stretch of a sharded run's files (round 6, fastpath.run_sharded_ranges) -- rank 0
// has created them at their final size from the ranks' mode-1 answers.
static int annotation_csv_device_impl(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, const char* mapped_path,
                                      const char* unmapped_path, const char* header, const int64_t* rows, int64_t n_rows,
                                      int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                      const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n,
                                      int mode, int64_t* bytes_out, const int64_t* file_off);
extern "C" int mirge_annotation_csv_device(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, const char* mapped_path,
                                           const char* unmapped_path, const char* header, const int64_t* rows, int64_t n_rows,
                                           int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                           const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n) {
    return annotation_csv_device_impl(c, U, res, mapped_path, unmapped_path, header, rows, n_rows, n_pass, col_of_pass, n_name_cols,
                                      name_data, name_off, name_n, 0, nullptr, nullptr);
}
extern "C" int mirge_annotation_csv_device_sizes(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, const int64_t* rows,
                                                 int64_t n_rows, int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                                 const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n,
                                                 int64_t* bytes_out) {
    if (!bytes_out) return fail(-1, "mirge_annotation_csv_device_sizes: bad argument");
    return annotation_csv_device_impl(c, U, res, "", "", "", rows, n_rows, n_pass, col_of_pass, n_name_cols, name_data, name_off, name_n,
                                      1, bytes_out, nullptr);
}
extern "C" int mirge_annotation_csv_device_at(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, const char* mapped_path,
                                              const char* unmapped_path, int64_t mapped_off, int64_t unmapped_off, const int64_t* rows,
                                              int64_t n_rows, int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                              const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n) {
    if (!mapped_path || !unmapped_path || mapped_off < 0 || unmapped_off < 0) return fail(-1, "mirge_annotation_csv_device_at: bad argument");
    const int64_t off[2] = {mapped_off, unmapped_off};
    return annotation_csv_device_impl(c, U, res, mapped_path, unmapped_path, "", rows, n_rows, n_pass, col_of_pass, n_name_cols, name_data,
                                      name_off, name_n, 2, nullptr, off);
}
static int annotation_csv_device_impl(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, const char* mapped_path,
                                      const char* unmapped_path, const char* header, const int64_t* rows, int64_t n_rows,
                                      int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                      const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n,
                                      int mode, int64_t* bytes_out, const int64_t* file_off) {
    if (!c || !U || !res || !header || (!rows && n_rows) || n_rows < 0 || n_pass < 1 || n_pass > MIRGE_MAX_PASSES || !col_of_pass ||
        n_name_cols < 0 || !name_data || !name_off || !name_n || U->n_samples < 1 || res->n != U->n || n_rows >= 0xFFFFFFF0ll)
        return fail(-1, "mirge_annotation_csv_device: bad argument");
    // hipCUB's scans take an `int` item count: beyond 2^31 - 2 rows the caller formats on the host (mirge_annotation_csv), as
    // it does for names that need quoting
    if (n_rows >= 0x7FFFFFFEll) return fail(-4, "mirge_annotation_csv_device: 2^31 rows or more (host route)");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    HostClock hc("annotation_csv_device");
    CsvTables t;
    std::memset(&t, 0, sizeof(t));
    t.n_pass = n_pass; t.n_name_cols = n_name_cols; t.S = U->n_samples;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = U->g[gi];
        const ResGroup& r = res->g[gi];
        if (g.orig) return fail(-1, "mirge_annotation_csv_device: the read set is not a collapse result");
        if (g.n != r.n && !(res->dmeta)) return fail(-1, "mirge_annotation_csv_device: result and read set differ");
        t.g[gi] = CsvGroup{g.seq, g.nmask, g.len, g.counts, r.pass, r.ref, g.base, g.n, g.W, is_long_group(gi) ? 1 : 0};
    }
    // names: one blob + 32-bit offsets per pass, uploaded per call (a human library set: ~0.2 M names, a few MB)
    std::vector<uint8_t> blob;
    std::vector<uint32_t> offs;
    size_t blob_at[MIRGE_MAX_PASSES], off_at[MIRGE_MAX_PASSES];
    for (int p = 0; p < n_pass; p++) {
        t.col_of_pass[p] = col_of_pass[p];
        if (col_of_pass[p] >= n_name_cols) return fail(-1, "mirge_annotation_csv_device: column index out of range");
        blob_at[p] = blob.size(); off_at[p] = offs.size();
        t.name_n[p] = 0;
        if (!name_data[p] || !name_off[p] || name_n[p] <= 0) continue;
        const int64_t nb = name_off[p][name_n[p]] - name_off[p][0];
        if (nb >= 0xFFFFFFF0ll || name_n[p] >= 0xFFFFFFF0ll) return fail(-5, "mirge_annotation_csv_device: name table too large");
        const char* src = name_data[p] + name_off[p][0];
        for (int64_t x = 0; x < nb; x++)
            if (src[x] == ',' || src[x] == '"' || src[x] == '\n' || src[x] == '\r')
                return fail(-4, "mirge_annotation_csv_device: a reference name needs CSV quoting (host route)");
        blob.insert(blob.end(), src, src + nb);
        for (int64_t r = 0; r <= name_n[p]; r++) offs.push_back((uint32_t)(name_off[p][r] - name_off[p][0]));
        t.name_n[p] = (uint32_t)name_n[p];
    }
    uint8_t* dblob = nullptr; uint32_t *doffs = nullptr, *drows = nullptr, *dflags = nullptr;
    unsigned long long *len_m = nullptr, *len_u = nullptr, *off_m = nullptr, *off_u = nullptr;
    uint8_t *out_m = nullptr, *out_u = nullptr;
    void* tmp = nullptr;
    std::vector<uint32_t> rows32((size_t)n_rows);
    for (int64_t k = 0; k < n_rows; k++) {
        if (rows[k] < 0 || rows[k] >= U->n) return fail(-1, "mirge_annotation_csv_device: row index out of range");
        rows32[(size_t)k] = (uint32_t)rows[k];
    }
    int rc = 0;
    size_t bytes_m = 0, bytes_u = 0;
    const size_t hl = std::strlen(header);
    do {
        if ((rc = dalloc(c, &dblob, blob.size() + 16))) break;
        if ((rc = dalloc(c, &doffs, offs.size() + 4))) break;
        if ((rc = dalloc(c, &drows, (size_t)n_rows + 1))) break;
        if ((rc = dalloc(c, &dflags, 16))) break;
        if ((rc = dalloc(c, &len_m, (size_t)n_rows + 1))) break;
        if ((rc = dalloc(c, &len_u, (size_t)n_rows + 1))) break;
        if ((rc = dalloc(c, &off_m, (size_t)n_rows + 1))) break;
        if ((rc = dalloc(c, &off_u, (size_t)n_rows + 1))) break;
        hipError_t e = hipSuccess;
        if (!blob.empty()) e = hipMemcpyAsync(dblob, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && !offs.empty()) e = hipMemcpyAsync(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && n_rows) e = hipMemcpyAsync(drows, rows32.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(dflags, 0, 64, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(len_m + n_rows, 0, 8, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(len_u + n_rows, 0, 8, c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_annotation_csv_device: ") + hipGetErrorString(e)); break; }
        for (int p = 0; p < n_pass; p++)
            if (t.name_n[p]) { t.name_data[p] = dblob + blob_at[p]; t.name_off[p] = doffs + off_at[p]; }
        if (n_rows) {
            LaunchScope ls(c, "k_csv_rowlen", (double)n_rows);
            hipLaunchKernelGGL(k_csv_rowlen, dim3(grid_for(c, (size_t)n_rows)), dim3(MIRGE_BLOCK), 0, c->stream, t, drows, (uint32_t)n_rows, len_m, len_u, dflags);
        }
        size_t tb = 0;
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, len_m, off_m, (int)(n_rows + 1), c->stream);
        if (e == hipSuccess && (rc = dalloc(c, (uint8_t**)&tmp, std::max<size_t>(tb, 16)))) break;
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, len_m, off_m, (int)(n_rows + 1), c->stream);
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, len_u, off_u, (int)(n_rows + 1), c->stream);
        unsigned long long tot[2] = {0, 0};
        uint32_t hflag = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&tot[0], off_m + n_rows, 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&tot[1], off_u + n_rows, 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&hflag, dflags, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_annotation_csv_device: ") + hipGetErrorString(e)); break; }
        if (hflag) { rc = fail(-1, "mirge_annotation_csv_device: pass or reference index out of range"); break; }
        bytes_m = mapped_path ? (size_t)tot[0] : 0;
        bytes_u = unmapped_path ? (size_t)tot[1] : 0;
        if (mode == 1) { bytes_out[0] = (int64_t)tot[0]; bytes_out[1] = (int64_t)tot[1]; break; }
        if (bytes_m && (rc = dalloc(c, &out_m, bytes_m))) break;
        if (bytes_u && (rc = dalloc(c, &out_u, bytes_u))) break;
        if (n_rows && (bytes_m || bytes_u)) {
            LaunchScope ls(c, "k_csv_rows", (double)n_rows);
            hipLaunchKernelGGL(k_csv_rows, dim3(grid_for(c, (size_t)n_rows)), dim3(MIRGE_BLOCK), 0, c->stream, t, drows, (uint32_t)n_rows, off_m, off_u, out_m, out_u);
        }
        hc.lap("format (device)");
        // page-locked staging, grown when a sample needs more (the first sample of a process pays for it)
        const size_t need = bytes_m + bytes_u;
        if (need > c->csv_pinned_bytes) {
            if (c->csv_pinned) (void)hipHostFree(c->csv_pinned);
            c->csv_pinned = nullptr; c->csv_pinned_bytes = 0;
            const size_t want = need + need / 8 + (1u << 20);
            if (hipHostMalloc((void**)&c->csv_pinned, want, hipHostMallocDefault) != hipSuccess) {
                rc = fail(-3, "mirge_annotation_csv_device: cannot page-lock " + std::to_string(want) + " bytes"); break;
            }
            c->csv_pinned_bytes = want;
            hc.lap("page-locked staging");
        }
        // copies in chunks, each followed by an event; writer threads put a chunk into its file as soon as it has arrived
        const size_t CH = 8u << 20;
        struct Chunk { int which; size_t src_off, file_off, n; hipEvent_t ev; };
        std::vector<Chunk> chunks;
        for (int which = 0; which < 2; which++) {
            const size_t nb = which == 0 ? bytes_m : bytes_u;
            const uint8_t* dsrc = which == 0 ? out_m : out_u;
            const size_t pin0 = which == 0 ? 0 : bytes_m;
            for (size_t at = 0; at < nb && e == hipSuccess; at += CH) {
                const size_t nn = std::min(CH, nb - at);
                e = hipMemcpyAsync(c->csv_pinned + pin0 + at, dsrc + at, nn, hipMemcpyDeviceToHost, c->stream);
                hipEvent_t ev = c->get_evt();
                if (e == hipSuccess) e = hipEventRecord(ev, c->stream);
                chunks.push_back(Chunk{which, pin0 + at, (mode == 2 ? (size_t)file_off[which] : hl) + at, nn, ev});
            }
        }
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_annotation_csv_device: ") + hipGetErrorString(e)); }
        int fd[2] = {-1, -1};
        const char* paths[2] = {mapped_path, unmapped_path};
        for (int which = 0; which < 2 && rc == 0; which++) {
            if (!paths[which]) continue;
            if (mode == 2) {  // a stretch of a file that exists at its final size: no header, nothing truncated
                if ((which == 0 ? bytes_m : bytes_u) == 0) continue;
                fd[which] = ::open(paths[which], O_WRONLY, 0644);
                if (fd[which] < 0) rc = fail(-8, std::string("cannot open ") + paths[which]);
                continue;
            }
            fd[which] = ::open(paths[which], O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd[which] < 0 || ::pwrite(fd[which], header, hl, 0) != (ssize_t)hl) rc = fail(-8, std::string("cannot write ") + paths[which]);
        }
        std::atomic<size_t> next{0};
        std::atomic<int> werr{0};
        auto writer = [&]() {
            (void)hipSetDevice(c->device);
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= chunks.size()) return;
                const Chunk& ck = chunks[k];
                if (hipEventSynchronize(ck.ev) != hipSuccess) { werr = 1; continue; }
                if (rc != 0 || fd[ck.which] < 0) continue;
                size_t done = 0;
                while (done < ck.n) {
                    const ssize_t w = ::pwrite(fd[ck.which], c->csv_pinned + ck.src_off + done, ck.n - done, (off_t)(ck.file_off + done));
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
        for (int which = 0; which < 2; which++)
            if (fd[which] >= 0 && ::close(fd[which]) != 0) werr = 1;
        if (rc == 0 && werr) rc = fail(-8, "mirge_annotation_csv_device: write error");
        hc.lap("copy + write");
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    c->release(dblob); c->release(doffs); c->release(drows); c->release(dflags); c->release(len_m); c->release(len_u);
    c->release(off_m); c->release(off_u); c->release(out_m); c->release(out_u); c->release(tmp);
    return rc;
}


static void csv_tables_of(const mirge_reads* U, const mirge_result* res, CsvTables& t) {
    std::memset(&t, 0, sizeof(t));
    t.S = U->n_samples;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = U->g[gi];
        t.g[gi] = CsvGroup{g.seq, g.nmask, g.len, g.counts, res ? res->g[gi].pass : nullptr, res ? res->g[gi].ref : nullptr, g.base, g.n, g.W,
                           is_long_group(gi) ? 1 : 0};
    }
}

// order_out[k] (k < U) = index of the unique read in row k of the SORTED union of the sequences: the row order of the
// reference's sample matrix for several samples (pandas `join(how='outer')` sorts its index, digest.py:243), from a radix
// sort on the device (k_lexkey).
extern "C" int mirge_collapse_order_sorted(mirge_ctx* c, const mirge_reads* U, int64_t* order_out) {
    if (!c || !U || (!order_out && U->n)) return fail(-1, "mirge_collapse_order_sorted: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    const size_t n = (size_t)U->n;
    if (!n) return 0;
    if (n >= 0x7FFFFFFFull) return fail(-5, "mirge_collapse_order_sorted: 2^31 unique reads or more (hipCUB's sort takes an int count)");
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
        if (U->g[gi].n && U->g[gi].orig) return fail(-1, "mirge_collapse_order_sorted: handle is not a collapse result");
    int maxlen = MIRGE_MAX_READ_LEN;
    if (U->hist_valid) {
        maxlen = 1;
        for (int L = 1; L <= MIRGE_MAX_READ_LEN; L++) if (U->len_hist[L]) maxlen = L;
    }
    maxlen = std::max(maxlen, (int)U->long_max);  // the long read class lies outside the histogram
    const int n_words = (maxlen + MIRGE_LEX_BASES - 1) / MIRGE_LEX_BASES;
    CsvTables t;
    csv_tables_of(U, nullptr, t);
    unsigned long long *keys = nullptr, *keys2 = nullptr;
    uint32_t *perm = nullptr, *perm2 = nullptr;
    void* tmp = nullptr;
    int rc = 0;
    do {
        if ((rc = dalloc(c, &keys, n))) break;
        if ((rc = dalloc(c, &keys2, n))) break;
        if ((rc = dalloc(c, &perm, n))) break;
        if ((rc = dalloc(c, &perm2, n))) break;
        size_t tb = 0;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys, keys2, perm, perm2, (int)n, 0, 63, c->stream);
        if (e == hipSuccess && (rc = dalloc(c, (uint8_t**)&tmp, std::max<size_t>(tb, 16)))) break;
        hipLaunchKernelGGL(k_iota, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, perm, (uint32_t)n);
        for (int w = n_words - 1; w >= 0 && e == hipSuccess; w--) {  // least significant word first; every pair sort is stable
            hipLaunchKernelGGL(k_lexkey, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, t, perm, (uint32_t)n, w, keys);
            e = hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, keys2, perm, perm2, (int)n, 0, 63, c->stream);
            std::swap(perm, perm2);
        }
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_collapse_order_sorted: ") + hipGetErrorString(e)); break; }
        std::vector<uint32_t> h(n);
        e = hipMemcpyAsync(h.data(), perm, n * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_collapse_order_sorted: ") + hipGetErrorString(e)); break; }
        for (size_t k = 0; k < n; k++) order_out[k] = (int64_t)h[k];
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    c->release(keys); c->release(keys2); c->release(perm); c->release(perm2); c->release(tmp);
    return rc;
}

// The sharded run's parallel tail (round 6; multigpu.py, fastpath.run_sharded_ranges).  The run's ONE mapped.csv / unmapped.csv
// hold the sorted union of all samples' sequences (digest.py:243, mirge/__main__.py:164-173); instead of rank 0 merging every
// dictionary alone, the joint key space is cut into one range per rank by the word-0 key of that order (k_lexkey, 21 bases).
//   mirge_reads_range_sample : k evenly spaced quantiles of this dictionary's word-0 keys (the ranks pool them: the splitters)
//   mirge_reads_range_split  : the dictionary's reads, lengths and counts as host arrays ordered by (owner range, handle index),
//                              bounds_out[q] .. bounds_out[q + 1] = the rows of range q -- each stretch goes to its owner
extern "C" int mirge_reads_range_sample(mirge_ctx* c, const mirge_reads* U, int32_t k, uint64_t* keys_out) {
    if (!c || !U || k < 1 || k > (1 << 20) || !keys_out) return fail(-1, "mirge_reads_range_sample: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    const size_t n = (size_t)U->n;
    if (n >= 0x7FFFFFFFull) return fail(-5, "mirge_reads_range_sample: 2^31 unique reads or more");
    if (!n) { for (int32_t i = 0; i < k; i++) keys_out[i] = ~0ull; return 0; }  // (an empty dictionary votes for nothing: see multigpu.choose_splitters)
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
        if (U->g[gi].n && U->g[gi].orig) return fail(-1, "mirge_reads_range_sample: handle is not a collapse result");
    CsvTables t;
    csv_tables_of(U, nullptr, t);
    unsigned long long *keys = nullptr, *keys2 = nullptr;
    uint32_t* perm = nullptr;
    void* tmp = nullptr;
    int rc = 0;
    do {
        if ((rc = dalloc(c, &keys, n))) break;
        if ((rc = dalloc(c, &keys2, n))) break;
        if ((rc = dalloc(c, &perm, n))) break;
        size_t tb = 0;
        hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, tb, keys, keys2, (int)n, 0, 63, c->stream);
        if (e == hipSuccess && (rc = dalloc(c, (uint8_t**)&tmp, std::max<size_t>(tb, 16)))) break;
        hipLaunchKernelGGL(k_iota, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, perm, (uint32_t)n);
        hipLaunchKernelGGL(k_lexkey, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, t, perm, (uint32_t)n, 0, keys);
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortKeys(tmp, tb, keys, keys2, (int)n, 0, 63, c->stream);
        for (int32_t i = 0; i < k && e == hipSuccess; i++) {
            const size_t at = std::min(n - 1, (size_t)((2 * (uint64_t)i + 1) * (uint64_t)n / (2 * (uint64_t)k)));  // (n < 2^31, k small)
            e = hipMemcpyAsync(&keys_out[i], keys2 + at, 8, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_reads_range_sample: ") + hipGetErrorString(e)); break; }
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    c->release(keys); c->release(keys2); c->release(perm); c->release(tmp);
    return rc;
}

extern "C" int mirge_reads_range_split(mirge_ctx* c, const mirge_reads* U, const uint64_t* splitters, int32_t n_parts, char* ascii_out,
                                       int64_t* off_out, uint32_t* counts_out, int64_t* bounds_out) {
    if (!c || !U || n_parts < 1 || n_parts > 256 || (n_parts > 1 && !splitters) || !off_out || !bounds_out || U->n_samples < 1 ||
        (U->n && !counts_out) || (U->total_bases > 0 && !ascii_out))
        return fail(-1, "mirge_reads_range_split: bad argument");
    for (int32_t q = 1; q + 1 < n_parts; q++)
        if (splitters[q] < splitters[q - 1]) return fail(-1, "mirge_reads_range_split: splitters must ascend");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    const size_t n = (size_t)U->n;
    if (n >= 0x7FFFFFFFull) return fail(-5, "mirge_reads_range_split: 2^31 unique reads or more");
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
        if (U->g[gi].n && U->g[gi].orig) return fail(-1, "mirge_reads_range_split: handle is not a collapse result");
    if (!n) { off_out[0] = 0; for (int32_t q = 0; q <= n_parts; q++) bounds_out[q] = 0; return 0; }
    const int32_t S = U->n_samples;
    CsvTables t;
    csv_tables_of(U, nullptr, t);
    unsigned long long *keys = nullptr, *dsplit = nullptr, *dbounds = nullptr;
    uint32_t *iota = nullptr, *owner = nullptr, *owner2 = nullptr, *perm = nullptr, *pos = nullptr, *dcnt = nullptr;
    void* tmp = nullptr;
    int rc = 0;
    do {
        if ((rc = dalloc(c, &keys, n))) break;
        if ((rc = dalloc(c, &dsplit, (size_t)n_parts))) break;
        if ((rc = dalloc(c, &dbounds, (size_t)n_parts + 1))) break;
        if ((rc = dalloc(c, &iota, n))) break;
        if ((rc = dalloc(c, &owner, n))) break;
        if ((rc = dalloc(c, &owner2, n))) break;
        if ((rc = dalloc(c, &perm, n))) break;
        if ((rc = dalloc(c, &pos, n))) break;
        if ((rc = dalloc(c, &dcnt, n * (size_t)S))) break;
        int bits = 1;
        while ((1 << bits) < n_parts) bits++;
        size_t tb = 0;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb, owner, owner2, iota, perm, (int)n, 0, bits, c->stream);
        if (e == hipSuccess && (rc = dalloc(c, (uint8_t**)&tmp, std::max<size_t>(tb, 16)))) break;
        std::vector<unsigned long long> hb((size_t)n_parts + 1, (unsigned long long)n);
        if (e == hipSuccess) e = hipMemcpyAsync(dbounds, hb.data(), hb.size() * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && n_parts > 1) e = hipMemcpyAsync(dsplit, splitters, (size_t)(n_parts - 1) * 8, hipMemcpyHostToDevice, c->stream);
        hipLaunchKernelGGL(k_iota, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, iota, (uint32_t)n);
        hipLaunchKernelGGL(k_lexkey, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, t, iota, (uint32_t)n, 0, keys);
        hipLaunchKernelGGL(k_range_owner, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, keys, (uint32_t)n, dsplit, n_parts - 1, owner);
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(tmp, tb, owner, owner2, iota, perm, (int)n, 0, bits, c->stream);  // stable: handle order inside a part
        hipLaunchKernelGGL(k_invert_perm, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, perm, (uint32_t)n, pos);
        hipLaunchKernelGGL(k_part_bounds, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->stream, owner2, (uint32_t)n, dbounds);
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
            const ReadGroup& g = U->g[gi];
            if (g.n) hipLaunchKernelGGL(k_scatter_rows, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, g.counts, g.n, S, pos + g.base, dcnt);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(hb.data(), dbounds, hb.size() * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(counts_out, dcnt, n * (size_t)S * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_reads_range_split: ") + hipGetErrorString(e)); break; }
        for (int32_t q = 0; q <= n_parts; q++) bounds_out[q] = (int64_t)hb[(size_t)q];
        bounds_out[0] = 0;
        rc = reads_unpack_impl(c, U, ascii_out, off_out, pos);
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    c->release(keys); c->release(dsplit); c->release(dbounds); c->release(iota); c->release(owner); c->release(owner2); c->release(perm);
    c->release(pos); c->release(dcnt); c->release(tmp);
    return rc;
}

// nonzero_out[s] = unique reads with a count in sample s: 'Trimmed Reads (unique)' per sample (digest.py:214) without
// fetching the U x S matrix
extern "C" int mirge_collapse_nonzero(mirge_ctx* c, const mirge_reads* U, int64_t* nonzero_out) {
    if (!c || !U || !nonzero_out || U->n_samples < 1) return fail(-1, "mirge_collapse_nonzero: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    const int32_t S = U->n_samples;
    unsigned long long* d = nullptr;
    CHECK(dalloc(c, &d, (size_t)S));
    HIPOK(hipMemsetAsync(d, 0, (size_t)S * 8, c->stream));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = U->g[gi];
        if (g.n) hipLaunchKernelGGL(k_count_nonzero_cols, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, g.counts, g.n, S, d);
    }
    std::vector<unsigned long long> h((size_t)S);
    HIPOK(hipMemcpyAsync(h.data(), d, (size_t)S * 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    for (int32_t s = 0; s < S; s++) nonzero_out[s] = (int64_t)h[(size_t)s];
    c->release(d);
    return 0;
}
