// native_csv.hpp -- part of mirge_native.hip (one translation unit): the per-read tables mapped.csv / unmapped.csv
// written from flat arrays on the host's cores.  Replaces the two DataFrame.to_csv calls of mirge/__main__.py:164-173
// (row a15): the reference builds a U-row pandas frame of Python strings to print it; here the rows are formatted
// straight from what the device returned (sequence bytes, pass, reference index, count matrix), same bytes out.
#pragma once

namespace {
struct CsvNames {  // reference names of one pass: one ASCII blob + offsets (n + 1); data == nullptr: pass has no library
    const char* data = nullptr;
    const int64_t* off = nullptr;
    int64_t n = 0;
};

// output of one formatting thread: a flat buffer written through a raw cursor (std::string::push_back per comma was a
// third of the formatting time)
struct CsvBuf {
    char* mem = nullptr;   // malloc'ed: growing must not zero-fill 200 MB that are about to be overwritten
    size_t cap = 0, n = 0;
    CsvBuf() = default;
    CsvBuf(const CsvBuf&) = delete;
    CsvBuf& operator=(const CsvBuf&) = delete;
    ~CsvBuf() { std::free(mem); }
    bool room(size_t extra) {
        if (n + extra <= cap) return true;
        const size_t want = std::max(cap * 2, n + extra + (1u << 16));
        char* m = (char*)std::realloc(mem, want);
        if (!m) return false;
        mem = m; cap = want;
        return true;
    }
    void put(char c) { mem[n++] = c; }
    void put(const char* s, size_t len) { std::memcpy(mem + n, s, len); n += len; }
    const char* data() const { return mem; }
    size_t size() const { return n; }
};

// pandas.to_csv quoting (csv.QUOTE_MINIMAL): quote a field that holds the delimiter, a quote or a line break.
// The caller has made room for 2 * len + 2 bytes.
inline void csv_field(CsvBuf& out, const char* s, size_t len) {
    bool q = false;
    for (size_t i = 0; i < len; i++) q |= s[i] == ',' || s[i] == '"' || s[i] == '\n' || s[i] == '\r';
    if (!q) { out.put(s, len); return; }
    out.put('"');
    for (size_t i = 0; i < len; i++) { if (s[i] == '"') out.put('"'); out.put(s[i]); }
    out.put('"');
}
inline void csv_uint(CsvBuf& out, uint64_t v) {  // room for 20 digits
    char buf[24];
    int k = 24;
    do { buf[--k] = (char)('0' + v % 10); v /= 10; } while (v);
    out.put(buf + k, (size_t)(24 - k));
}
inline void csv_uint(std::string& out, uint64_t v) {  // (the GFF writer's small tables)
    char buf[24];
    int k = 24;
    do { buf[--k] = (char)('0' + v % 10); v /= 10; } while (v);
    out.append(buf + k, (size_t)(24 - k));
}
}  // namespace

// rows[k] (k < n_rows) = index of the read printed in row k (the caller's row order: first appearance for one sample,
// sorted sequences for several).  A read goes to `mapped_path` when pass[i] >= 0, else to `unmapped_path` (either may
// be NULL).  Columns: Sequence, annotFlag, one name column per pass column (col_of_pass[p] = which column pass p
// writes, -1 = none; n_name_cols columns in all), then the S counts.  `header` is the first line, written as given.
extern "C" int mirge_annotation_csv(const char* mapped_path, const char* unmapped_path, const char* header,
                                    const char* seq_ascii, const int64_t* seq_off, const int8_t* pass, const int32_t* ref,
                                    const uint32_t* counts, int32_t S, const int64_t* rows, int64_t n_rows,
                                    int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                    const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n) {
    if (!header || !seq_off || !pass || !ref || !counts || !rows || S < 1 || n_rows < 0 || n_pass < 1 || n_pass > MIRGE_MAX_PASSES ||
        !col_of_pass || n_name_cols < 0 || !name_data || !name_off || !name_n || (n_rows > 0 && !seq_ascii))
        return fail(-1, "mirge_annotation_csv: bad argument");
    CsvNames nm[MIRGE_MAX_PASSES];
    for (int p = 0; p < n_pass; p++) {
        nm[p].data = name_data[p]; nm[p].off = name_off[p]; nm[p].n = name_n[p];
        if (col_of_pass[p] >= n_name_cols) return fail(-1, "mirge_annotation_csv: column index out of range");
    }
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const int T = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n_rows / 16384));
    std::vector<CsvBuf> bufm((size_t)T), bufu((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    auto work = [&](int t) {
        const int64_t lo = n_rows * t / T, hi = n_rows * (t + 1) / T;
        CsvBuf &M = bufm[(size_t)t], &U = bufu[(size_t)t];
        if (!M.room((size_t)(hi - lo) * 56)) { bad[(size_t)t] = 2; return; }
        const size_t fixed = 2 + (size_t)n_name_cols + (size_t)S * 21 + 1 + 4;  // flag, commas, counts, newline, the quotes of two quoted fields
        for (int64_t k = lo; k < hi; k++) {
            // the rows come in the order of first appearance, the arrays in the order the device emitted the reads: every row
            // is four cache misses unless they are asked for ahead
            if (k + 16 < hi) {
                const int64_t j = rows[k + 16];
                __builtin_prefetch(&pass[j]); __builtin_prefetch(&ref[j]); __builtin_prefetch(&counts[(size_t)j * S]);
                __builtin_prefetch(&seq_off[j]);
            }
            if (k + 8 < hi) __builtin_prefetch(seq_ascii + seq_off[rows[k + 8]]);
            const int64_t i = rows[k];
            const int p = pass[i];
            if (p >= n_pass) { bad[(size_t)t] = 1; continue; }
            CsvBuf& out = p >= 0 ? M : U;
            if ((p >= 0 ? mapped_path : unmapped_path) == nullptr) continue;
            const size_t slen = (size_t)(seq_off[i + 1] - seq_off[i]);
            const int col = p >= 0 ? col_of_pass[p] : -1;
            const char* name = nullptr;
            size_t nlen = 0;
            if (col >= 0) {
                const int32_t r = ref[i];
                if (!nm[p].data || r < 0 || r >= nm[p].n) { bad[(size_t)t] = 1; continue; }
                name = nm[p].data + nm[p].off[r];
                nlen = (size_t)(nm[p].off[r + 1] - nm[p].off[r]);
            }
            if (!out.room(2 * slen + 2 * nlen + fixed)) { bad[(size_t)t] = 2; return; }
            csv_field(out, seq_ascii + seq_off[i], slen);
            out.put(',');
            out.put(p >= 0 ? '1' : '0');
            for (int cidx = 0; cidx < n_name_cols; cidx++) {
                out.put(',');
                if (cidx == col) csv_field(out, name, nlen);
            }
            for (int s = 0; s < S; s++) { out.put(','); csv_uint(out, counts[(size_t)i * S + s]); }
            out.put('\n');
        }
    };
    HostClock hc("annotation_csv");
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    hc.lap("format");
    for (int t = 0; t < T; t++) {
        if (bad[(size_t)t] == 2) return fail(-3, "mirge_annotation_csv: out of host memory");
        if (bad[(size_t)t]) return fail(-1, "mirge_annotation_csv: pass or reference index out of range");
    }
    for (int which = 0; which < 2; which++) {
        const char* path = which == 0 ? mapped_path : unmapped_path;
        if (!path) continue;
        // every thread writes its own chunk at its own offset (pwrite): a 200 MB table is bound by the copy into the
        // page cache, which one thread does at a fraction of the machine's memory bandwidth.  (A shared mapping of the
        // sized file with one memcpy per thread was 3.5 x slower on the 256-thread host of the GPU box: page faults on
        // one mapping contend more than positional writes do.)
        const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return fail(-8, std::string("cannot write ") + path);
        const size_t hl = std::strlen(header);
        std::vector<size_t> at((size_t)T + 1, hl);
        for (int t = 0; t < T; t++) at[(size_t)t + 1] = at[(size_t)t] + (which == 0 ? bufm[(size_t)t] : bufu[(size_t)t]).size();
        bool ok = ::pwrite(fd, header, hl, 0) == (ssize_t)hl;
        std::vector<int> werr((size_t)T, 0);
        auto put = [&](int t) {
            const CsvBuf& b = which == 0 ? bufm[(size_t)t] : bufu[(size_t)t];
            size_t done = 0;
            while (done < b.size()) {
                const ssize_t w = ::pwrite(fd, b.data() + done, b.size() - done, (off_t)(at[(size_t)t] + done));
                if (w <= 0) { werr[(size_t)t] = 1; return; }
                done += (size_t)w;
            }
        };
        std::vector<std::thread> wt;
        for (int t = 1; t < T; t++) wt.emplace_back(put, t);
        put(0);
        for (auto& x : wt) x.join();
        for (int t = 0; t < T; t++) ok = ok && !werr[(size_t)t];
        ok = (::close(fd) == 0) && ok;
        if (!ok) return fail(-8, std::string("write error on ") + path);
        hc.lap("write");
    }
    return 0;
}
